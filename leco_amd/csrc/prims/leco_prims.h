// Device primitives for the gfx950 (CDNA4, wave64) kernels: raw-bit bf16 helpers, the MFMA
// tile op, wave shuffles.  Kernel sources include this as <leco_prims.h>; the host-emulation
// test harness (tests/emu/) substitutes its own header of the same name at build time so the
// kernel sources themselves stay pure HIP.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace leco {
typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 hw_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// float -> bf16, round-to-nearest-even, through the gfx950 hardware converter (v_cvt_pk_bf16_f32:
// one instruction per PAIR of values; a hand-rolled integer rounding costs ~7 VALU ops per value).
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hw_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

// D(16x16,f32) = A(16x32,bf16) * B(32x16,bf16) + C   -- v_mfma_f32_16x16x32_bf16.
// Lane l supplies A[l&15][8*(l>>4)+t] and B[8*(l>>4)+t][l&15], t=0..7, and holds
// C/D[4*(l>>4)+r][l&15], r=0..3 (cdna_hip_programming.md section 3).
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(hw_bf16x8, a),
                                                   __builtin_bit_cast(hw_bf16x8, b), c, 0, 0, 0);
}

// D(16x16,f32) = A(16x4,f32) * B(4x16,f32) + C -- v_mfma_f32_16x16x4_f32: f32 inputs, exact f32 products and sums (the fp32
// compute mode, f32.hip).  Lane l supplies A[l&15][l>>4] and B[l>>4][l&15] and holds C/D[4*(l>>4)+r][l&15], r=0..3.
__device__ __forceinline__ f32x4 mfma16x4_f32(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Asynchronous 16-byte global -> LDS copy (global_load_lds_dwordx4): lane l of the wave writes
// lds_wave_base + 16*l; the SOURCE address is per lane.  Completion is tracked by vmcnt
// (__syncthreads() drains it).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The same DMA through a buffer resource descriptor (buffer_load_dwordx4 ... offen lds): the source is
// base + voff (per lane, bytes) + soff (wave-uniform, bytes) -- 32-bit offsets, no 64-bit per-lane pointer math -- and a
// lane whose voff lies beyond the descriptor's extent reads ZEROS (hardware range check): conv halos / padding rows need
// neither a zero page nor a select.  `bytes` < 2^31 so that DMA_OOB is out of range for every soff.
typedef __amdgpu_buffer_rsrc_t buf_rsrc;
constexpr unsigned DMA_OOB = 0x80000000u;
__device__ __forceinline__ buf_rsrc make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void glds16_buf(buf_rsrc r, unsigned voff, unsigned soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
// 16-byte load through a buffer descriptor into registers (buffer_load_dwordx4 ... offen): per-lane byte offset + a
// wave-uniform byte offset; counted by the compiler's own s_waitcnt vmcnt bookkeeping (unlike the LDS-DMA forms above it has
// a register destination the compiler tracks), so a software-pipelined prefetch array needs no hand-written waits.
__device__ __forceinline__ bf16x8 buf_load16(buf_rsrc r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
// Counted wait on this wave's outstanding global/LDS-DMA operations (s_waitcnt vmcnt(N)): the N most
// recently issued may still be in flight.  Lets several tile DMAs span a barrier.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// workgroup barrier that orders LDS accesses but does NOT drain in-flight LDS-DMA (unlike __syncthreads())
__device__ __forceinline__ void barrier_keep_dma() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
// 16-byte LDS read the compiler's s_waitcnt insertion does not track (ds_read_b128 through inline asm).  With
// LDS-DMA in flight the compiler waits lgkmcnt(0) before the first use of ANY ds_read result, which would
// serialise software-pipelined fragment reads; these reads are completed explicitly with lds_wait<N>() and
// handed back to the compiler with lds_tie() (an empty asm that makes later uses depend on the wait).
__device__ __forceinline__ bf16x8 lds_read16_async(const void* lds_ptr) {
    bf16x8 v;
    const unsigned a = (unsigned)(unsigned long long)lds_ptr;   // generic LDS pointer: low 32 bits = LDS byte address
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
    return v;
}
// LDS stores the compiler's s_waitcnt insertion does not track (ds_write_b64 / ds_write_b32 through inline asm): with
// LDS-DMA in flight hipcc puts a vmcnt wait for EVERY outstanding DMA in front of a C++-level LDS access (it cannot
// tell the DMA's destination from the store's), which drains the weight stream at every phase boundary of the stripe
// kernels (stripe.hip).  Completion: lgkmcnt (barrier_keep_dma() waits lgkmcnt(0)).  The data registers are read at
// issue, so they may be overwritten afterwards (64-bit data: no store-data hazard, cdna_hip_programming.md 5.7).
__device__ __forceinline__ void lds_write8_async(void* lds_ptr, u32x2 v) {
    const unsigned a = (unsigned)(unsigned long long)lds_ptr;
    asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write4_async(void* lds_ptr, float v) {
    const unsigned a = (unsigned)(unsigned long long)lds_ptr;
    asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
}
// The same by LDS byte ADDRESS (what the ds_* instructions take) with a compile-time displacement in the instruction's
// 16-bit offset field: one base register serves every fragment of a tile (no per-read address arithmetic).
__device__ __forceinline__ unsigned lds_addr(const void* lds_ptr) { return (unsigned)(unsigned long long)lds_ptr; }
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read16_at(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ void lds_write8_at(unsigned addr, u32x2 v) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_write4_at(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// Hardware transpose read (ds_read_b64_tr_b16): every lane passes the address of 4 consecutive bf16; inside each
// group of 16 lanes the 16 x 4 elements are exchanged so that lane c receives element (c & 3) of lanes
// c/4, 4 + c/4, 8 + c/4, 12 + c/4 -- i.e. with lane i addressing row i/4, columns 4(i%4).. of a [4][16] block,
// lane c gets column c (4 rows): an MFMA operand gathered from a row-major tile without a transposing store.
__device__ __forceinline__ u32x2 lds_read_tr16(const void* lds_ptr) {
    typedef __attribute__((ext_vector_type(4))) short s4;
    const s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)lds_ptr);
    return __builtin_bit_cast(u32x2, v);
}
// The transpose read through inline asm, by LDS byte address + compile-time displacement: not tracked by the compiler's
// s_waitcnt insertion (see lds_read16_async) -- complete with lds_wait<N>() + lds_tie2().
template <int OFF>
__device__ __forceinline__ u32x2 lds_read_tr16_at(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void lds_tie2(u32x2& v) { asm volatile("" : "+v"(v)); }
// wait until at most N LDS (lgkm) operations of this wave are outstanding
template <int N>
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lds_tie(bf16x8& v) { asm volatile("" : "+v"(v)); }
// makes a value opaque to the optimiser at this point (no instruction is emitted): stops loop-invariant code motion
// from hoisting -- and spilling -- whole tables of addresses derived from it
__device__ __forceinline__ void opaque(int& v) { asm volatile("" : "+v"(v)); }
// scheduling fence: the compiler moves no instruction across it
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// Occupancy floor for a kernel (waves per SIMD).  Besides capping the register budget it makes the compiler
// keep MFMA accumulators in VGPRs: with a budget above 256 registers it places them in AGPRs and every VALU
// touch of an accumulator (softmax rescale of O) costs a v_accvgpr_read / v_accvgpr_write pair.
#define LECO_MIN_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n)))
// launch-time sized LDS (up to 160 KB per workgroup on gfx950)
extern __shared__ __attribute__((aligned(16))) unsigned char leco_dyn_lds_[];
__device__ __forceinline__ unsigned char* dyn_lds() { return leco_dyn_lds_; }

// Pointer into a read-only table in global memory that is read through the SCALAR data cache (constant address space:
// s_load, tracked by lgkmcnt).  A by-value kernel argument that is indexed dynamically is copied to scratch by hipcc, and
// an ordinary global load would be a vector-memory operation -- its wait would drain every LDS-DMA in flight.
#define LECO_CONST_AS __attribute__((address_space(4)))
#define LECO_CONST_CAST(T, p) ((const LECO_CONST_AS T*)(unsigned long long)(p))
// tells the compiler a value is wave-uniform (v_readfirstlane): needed for values derived from
// threadIdx (e.g. the wave index) that feed scalar operands such as the LDS-DMA base (M0)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// 24-bit x 24-bit -> low 32 bits (v_mul_u32_u24 / v_mad_u32_u24: full-rate, unlike the 32-bit multiply)
__device__ __forceinline__ unsigned mul24(unsigned a, unsigned b) { return __umul24(a, b); }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ float shfl(float v, int src) { return __shfl(v, src, 64); }
// Sum over lanes without LDS traffic: butterfly inside each row of 16 lanes by DPP (v_add_f32_dpp: quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror -- after each step both partners hold the same partial, so the pairing is that of an xor
// butterfly with masks 1, 2, 4, 8), then the four row totals through v_readlane as (r0 + r1) + (r2 + r3).  Every lane
// returns the same bits; a 6-step __shfl_xor loop is six dependent ds_bpermute round trips (~100 cycles each).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// all 16 lanes of a DPP row (lanes 16 r .. 16 r + 15) receive the row's sum
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1>(v);      // quad_perm [1, 0, 3, 2]
    v += dpp_f32<0x4E>(v);      // quad_perm [2, 3, 0, 1]
    v += dpp_f32<0x141>(v);     // row_half_mirror
    v += dpp_f32<0x140>(v);     // row_mirror
    return v;
}
// sum over 8 consecutive lanes (lanes 8 q .. 8 q + 7)
__device__ __forceinline__ float row8_sum(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    const int bits = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 48));
    return (r0 + r1) + (r2 + r3);
}
// Reductions over the FOUR 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48 -- the lanes that hold one query row's
// scores in the swapped-operand MFMA layout) with gfx950's v_permlane16_swap / v_permlane32_swap: a VALU exchange of two
// registers' rows instead of two ds_bpermute round trips through the LDS pipe per reduction.  With both operands = v,
// permlane16_swap returns {rows v0 v0 v2 v2, rows v1 v1 v3 v3}, permlane32_swap {lo lo, hi hi}: combining the pair gives
// both partners the same value.
// (Inline asm: with `__builtin_amdgcn_permlane16_swap` hipcc 7.2 folds the two results of the swap into one -- max(a0, a1)
// became a0 and a0 + a1 became a1 + a1 in the ISA -- although they differ lane by lane.  `s_nop 1`: the wait states hipcc
// itself puts between a VALU write of an operand and the swap.)
__device__ __forceinline__ void permlane16_swap(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void permlane32_swap(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float rows4_max(float v) {
    // (the max rides in the asm block: on asm outputs the compiler would first canonicalise both operands of an fmaxf)
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
    float c = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(c));
    return a;
}
__device__ __forceinline__ float rows4_sum(float v) {
    float a = v, b = v;
    permlane16_swap(a, b);
    float c = a + b, d = c;
    permlane32_swap(c, d);
    return c + d;
}
// two fp32 fused multiply-adds in one instruction (v_pk_fma_f32); true when the predicate holds in any lane of the wave
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ bool wave_any(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0ull; }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
}  // namespace leco
