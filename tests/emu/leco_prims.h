// TEST INFRASTRUCTURE ONLY -- host-emulation twin of leco_amd/csrc/prims/leco_prims.h.
// Same names and semantics; the wave64 collectives are implemented over emu::wave_gather.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace leco {
typedef unsigned short bf16_t;
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

static inline float __uint_as_float_emu(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint_emu(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
#define __uint_as_float leco::__uint_as_float_emu
#define __float_as_uint leco::__float_as_uint_emu

static inline float bf2f(bf16_t h) { return __uint_as_float_emu(((unsigned)h) << 16); }
static inline bf16_t f2bf(float f) {
    unsigned u = __float_as_uint_emu(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
static inline unsigned pack_bf2(float lo, float hi) {
    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}

// v_mfma_f32_16x16x32_bf16 lane maps (cdna_hip_programming.md section 3):
// lane l supplies A[l&15][8*(l>>4)+t], B[8*(l>>4)+t][l&15]; holds D[4*(l>>4)+r][l&15].
static inline f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    struct Dep { short a[8]; short b[8]; } d;
    for (int t = 0; t < 8; ++t) { d.a[t] = a[t]; d.b[t] = b[t]; }
    const unsigned char* all = emu::wave_gather(&d, sizeof(d));
    int l = emu::lane();
    int j = l & 15, g = l >> 4;
    f32x4 out = c;
    for (int r = 0; r < 4; ++r) {
        int i = 4 * g + r;
        float acc = 0.f;
        for (int k = 0; k < 32; ++k) {
            const Dep* la = reinterpret_cast<const Dep*>(all + (size_t)(i + 16 * (k >> 3)) * emu::kSlot);
            const Dep* lb = reinterpret_cast<const Dep*>(all + (size_t)(j + 16 * (k >> 3)) * emu::kSlot);
            acc += bf2f((bf16_t)la->a[k & 7]) * bf2f((bf16_t)lb->b[k & 7]);
        }
        out[r] = c[r] + acc;
    }
    return out;
}

// v_mfma_f32_16x16x4_f32: lane l supplies A[l&15][l>>4], B[l>>4][l&15]; holds D[4*(l>>4)+r][l&15]
static inline f32x4 mfma16x4_f32(float a, float b, f32x4 c) {
    struct Dep { float a, b; } d{a, b};
    const unsigned char* all = emu::wave_gather(&d, sizeof(d));
    const int l = emu::lane(), j = l & 15, g = l >> 4;
    f32x4 out = c;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            const Dep* la = reinterpret_cast<const Dep*>(all + (size_t)(i + 16 * k) * emu::kSlot);
            const Dep* lb = reinterpret_cast<const Dep*>(all + (size_t)(j + 16 * k) * emu::kSlot);
            acc = fmaf(la->a, lb->b, acc);
        }
        out[r] = acc;
    }
    return out;
}

// LDS-DMA model.  Default: the copy completes at issue (earliest possible landing: exposes a DMA that overwrites
// a ring slot other waves still read).  With LECO_EMU_DMA=late in the environment every copy is DEFERRED until
// the issuing lane's counted wait (`wait_vmcnt<N>` completes all but its N youngest) or a full
// `__syncthreads()` (which drains vmcnt on the hardware): the latest possible landing, so a fragment read that
// is not covered by the right `vmcnt` + barrier sees stale LDS and the parity tests fail.  Kernels must pass in
// both modes (tests/test_kernels.py::test_gemm_dma_protocol_under_late_completion).
namespace emu_dma {
struct Pending { const void* src; void* dst; };
struct Queue { Pending q[64]; int head = 0, count = 0; };
static inline bool late() {
    static const bool v = [] { const char* e = getenv("LECO_EMU_DMA"); return e && !strcmp(e, "late"); }();
    return v;
}
static inline Queue& mine() {
    static thread_local Queue qs[1024];      // one workgroup at a time per OS thread; indexed by work-item
    return qs[emu::t_idx.x + emu::b_dim.x * (emu::t_idx.y + emu::b_dim.y * emu::t_idx.z)];
}
static inline void complete_all_but(int keep) {
    Queue& Q = mine();
    while (Q.count > keep) {
        memcpy(Q.q[Q.head].dst, Q.q[Q.head].src, 16);
        Q.head = (Q.head + 1) & 63;
        --Q.count;
    }
}
}  // namespace emu_dma
static inline void glds16(const void* gsrc, void* lds_wave_base) {
    void* dst = (unsigned char*)lds_wave_base + 16 * emu::lane();
    if (!emu_dma::late()) { memcpy(dst, gsrc, 16); return; }
    emu_dma::Queue& Q = emu_dma::mine();
    if (Q.count == 64) emu_dma::complete_all_but(63);       // vmcnt saturates: the oldest has landed by then
    Q.q[(Q.head + Q.count) & 63] = emu_dma::Pending{gsrc, dst};
    ++Q.count;
}
// buffer-descriptor form: 32-bit offsets, out-of-range lanes read zeros
struct buf_rsrc { const unsigned char* base; unsigned bytes; };
constexpr unsigned DMA_OOB = 0x80000000u;
static inline buf_rsrc make_rsrc(const void* base, unsigned bytes) { return buf_rsrc{(const unsigned char*)base, bytes}; }
static inline void glds16_buf(buf_rsrc r, unsigned voff, unsigned soff, void* lds_wave_base) {
    static const unsigned char zeros[16] = {0};
    // range check on the per-lane offset alone (gfx9 raw buffers: buffer_offset >= num_records - soffset is out of range)
    const bool oob = voff >= r.bytes || (unsigned long long)voff + soff + 16 > r.bytes;
    glds16(oob ? (const void*)zeros : (const void*)(r.base + voff + soff), lds_wave_base);
}
static inline bf16x8 buf_load16(buf_rsrc r, unsigned voff, unsigned soff) {
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool oob = voff >= r.bytes || (unsigned long long)voff + soff + 16 > r.bytes;
    if (!oob) memcpy(&v, r.base + voff + soff, 16);
    return v;
}
template <int N>
static inline void wait_vmcnt() { if (emu_dma::late()) emu_dma::complete_all_but(N); }
#undef __syncthreads
#define __syncthreads() (emu_dma::complete_all_but(0), emu::sync_block())
#define LECO_MIN_WAVES_PER_SIMD(n)
static inline bf16x8 lds_read16_async(const void* lds_ptr) { return *(const bf16x8*)lds_ptr; }
static inline void lds_write8_async(void* lds_ptr, u32x2 v) { memcpy(lds_ptr, &v, 8); }
static inline void lds_write4_async(void* lds_ptr, float v) { memcpy(lds_ptr, &v, 4); }
static inline unsigned char* dyn_lds();
static inline unsigned lds_addr(const void* lds_ptr) { return (unsigned)((const unsigned char*)lds_ptr - dyn_lds()); }
template <int OFF>
static inline bf16x8 lds_read16_at(unsigned addr) { return *(const bf16x8*)(dyn_lds() + addr + OFF); }
template <int OFF>
static inline void lds_write8_at(unsigned addr, u32x2 v) { memcpy(dyn_lds() + addr + OFF, &v, 8); }
static inline void lds_write4_at(unsigned addr, float v) { memcpy(dyn_lds() + addr, &v, 4); }
static inline u32x2 lds_read_tr16(const void* lds_ptr) {
    unsigned long long mine;
    memcpy(&mine, lds_ptr, 8);
    const unsigned char* all = emu::wave_gather(&mine, 8);
    const int l = emu::lane(), g = l >> 4, c = l & 15;
    unsigned short out[4];
    for (int j = 0; j < 4; ++j) memcpy(&out[j], all + (size_t)(16 * g + 4 * j + (c >> 2)) * emu::kSlot + 2 * (c & 3), 2);
    u32x2 r;
    r[0] = (unsigned)out[0] | ((unsigned)out[1] << 16);
    r[1] = (unsigned)out[2] | ((unsigned)out[3] << 16);
    return r;
}
template <int OFF>
static inline u32x2 lds_read_tr16_at(unsigned addr) { return lds_read_tr16(dyn_lds() + addr + OFF); }
static inline void lds_tie2(u32x2&) {}
template <int N>
static inline void lds_wait() {}
static inline void lds_tie(bf16x8&) {}
static inline void opaque(int&) {}
static inline void sched_fence() {}
static inline void barrier_keep_dma() { emu::sync_block(); }
// The hardware hands a workgroup whatever its predecessor left in LDS.  LECO_EMU_LDS=poison fills the dynamic LDS with a
// NaN pattern (0x7FC0: a NaN as bf16 and, doubled, as fp32) at each workgroup's first touch, so a read of never-written LDS
// that reaches a result shows up as a NaN instead of passing on a zeroed / left-over buffer.
static inline unsigned char* dyn_lds() {
    static thread_local __attribute__((aligned(16))) unsigned char buf[160 * 1024];
    static thread_local unsigned long seen = 0;
    if (emu::lds_poison() && seen != emu::block_serial()) {
        seen = emu::block_serial();
        unsigned short* p = (unsigned short*)buf;
        for (int i = 0; i < 80 * 1024; ++i) p[i] = 0x7FC0;
    }
    return buf;
}
#define LECO_CONST_AS
#define LECO_CONST_CAST(T, p) ((const T*)(p))
static inline int uniform(int v) { return v; }
static inline unsigned mul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline int lane_id() { return emu::lane(); }
static inline f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return f32x2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])}; }
static inline bool wave_any(bool pred) {
    const int v = pred ? 1 : 0;
    const unsigned char* all = emu::wave_gather(&v, 4);
    int any = 0;
    for (int l = 0; l < 64; ++l) { int t; memcpy(&t, all + (size_t)l * emu::kSlot, 4); any |= t; }
    return any != 0;
}
static inline float shfl_xor(float v, int mask) {
    const unsigned char* all = emu::wave_gather(&v, 4);
    float r;
    memcpy(&r, all + (size_t)(emu::lane() ^ mask) * emu::kSlot, 4);
    return r;
}
static inline float shfl(float v, int src) {
    const unsigned char* all = emu::wave_gather(&v, 4);
    float r;
    memcpy(&r, all + (size_t)(src & 63) * emu::kSlot, 4);
    return r;
}
// the DPP reductions of the hardware header as xor butterflies (same pairing, same bits)
static inline float row8_sum(float v) { v += shfl_xor(v, 1); v += shfl_xor(v, 2); v += shfl_xor(v, 4); return v; }
static inline float row16_sum(float v) { v = row8_sum(v); v += shfl_xor(v, 8); return v; }
static inline float wave_sum(float v) {
    v = row16_sum(v);
    const float r0 = shfl(v, 0), r1 = shfl(v, 16), r2 = shfl(v, 32), r3 = shfl(v, 48);
    return (r0 + r1) + (r2 + r3);
}
// v_permlane16_swap / v_permlane32_swap reductions over the four 16-lane rows (same pairing: rows (0,1), (2,3), then halves)
static inline float rows4_max(float v) { v = fmaxf(v, shfl_xor(v, 16)); return fmaxf(v, shfl_xor(v, 32)); }
static inline float rows4_sum(float v) {
    const float a = shfl_xor(v, 16);
    const int l = emu::lane();
    // the hardware adds (even row's value) + (odd row's value), then (lower half) + (upper half): the same operand order here
    float m = ((l >> 4) & 1) ? a + v : v + a;
    const float b = shfl_xor(m, 32);
    return (l >> 5) ? b + m : m + b;
}
static inline float fast_exp2(float x) { return exp2f(x); }
static inline float fast_rcp(float x) { return 1.0f / x; }
}  // namespace leco
