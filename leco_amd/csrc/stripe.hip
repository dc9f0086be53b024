// Row-stripe fused kernels for the transformer blocks of the 64^2 level (forward-only plans), gfx950 / wave64,
// v_mfma_f32_16x16x32_bf16.
//
// Everything a diffusers BasicTransformerBlock does AFTER its self-attention core is local to a stripe of token rows
// (call site train_util.py:156-160; LoRA term lora.py:102-106):
//
//   h1 = attn1.to_out(a1) + h0 ;  q2 = attn2.to_q(LN2(h1)) ;  a2 = softmax(q2 K2^T) V2   (77 prompt keys, projected once per step)
//   h2 = attn2.to_out(a2) + h1 ;  h3 = ff.net.2(GEGLU(ff.net.0.proj(LN3(h2)))) + h2 ;   out = proj_out(h3) + x
//
// and so is everything BEFORE it (`xblock_head_kernel`: GroupNorm apply, proj_in, LN1, q|k|v).  As separate launches these
// are 14 short-K GEMM / LayerNorm / cross-attention kernels per block, each near its own launch + prologue + epilogue floor
// (55-65 % of a K = 320 projection is launch, first-tile latency and epilogue).  Here ONE workgroup (8 waves)
// owns a stripe of 64 token rows for the whole chain:
//
//   * ACTIVATIONS ARE STATIONARY IN LDS, WEIGHTS STREAM THROUGH REGISTERS.  The GEMM operand on the activation side is a
//     [64][320] bf16 image in LDS (XOR-swizzled 16-byte chunks) that the phases hand to each other in place
//     (a1 -> l2 -> q2 -> a2 -> l3 -> h3).  Every wave owns 2-3 of the 20 sixteen-column fragments of a 320-wide result for
//     ALL 64 rows ("1 x 8" layout), so a weight fragment is needed by exactly one wave: it is loaded global -> VGPR in MFMA
//     operand layout (buffer_load_dwordx4, 3 k-steps ahead), never through LDS.  A K loop has NO barrier and no LDS write:
//     per 32-wide k-step a wave issues 4 ds_read_b128 (activation fragments), 2-3 buffer loads and 8-12 MFMAs.  (The first
//     version of this kernel streamed weight tiles through an LDS ring shared by the waves -- one barrier, 2-3 LDS-DMA
//     issues and ~60 scalar / address instructions per 10 MFMAs: 125-150 us per block against ~160 us for the launches it
//     replaced; profiles/r04_stripe_ablate_v*.txt.)
//   * the residual stream h lives in REGISTERS as fp32 in the accumulator layout: a residual GEMM accumulates straight into
//     it, LayerNorm reads it (row statistics: lane-local + 2 shuffles + one LDS exchange between the 8 waves) -- h1 / h2 /
//     h3 are never rounded, never stored;
//   * LoRA: the stacked lora_down rows are one more 16-column fragment ("T duty" of the waves that own only two weight
//     fragments); T is rounded to bf16, shared through LDS, and the K-extension step T (scale up)^T adds the low-rank term
//     in fp32 (as gemm.hip);
//   * the feed-forward runs in 10 chunks of 128 hidden units: FF1 chunk (a wave owns a value fragment and its gate
//     fragment) -> GEGLU in registers -> bf16 chunk in a double-buffered LDS image -> FF2 partial sums into h: the [M][8C]
//     and [M][4C] intermediates never exist;
//   * cross-attention: one wave per head, in place; K fragments / pre-transposed V fragments straight from global
//     (L2-resident, prepared once per step by `xattn_prep_kernel`), P stays in registers (swapped products, as attention.hip).
//
// bf16-only fast path of the forward-only plans (denoising passes, batched frozen pass); the training plan and the fp32
// compute mode keep the per-op kernels.
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <hip/hip_runtime.h>
#include <leco_prims.h>

#include "common.h"

namespace leco {
namespace {

// Tuning aids (tools/ablate_stripe.py builds side libraries): LECO_STRIPE_ABLATE bit mask -- 1 = no MFMA, 2 = no weight
// loads, 4 = no activation-fragment reads, 8 = no cross-attention; results are garbage with any bit set, only the time means
// something.  LECO_STRIPE_TIMING: workgroup 0 stamps the shader clock at every phase boundary into a device array
// (leco_xblock_debug_times).  Both 0 / undefined in the product build.
#ifndef LECO_STRIPE_ABLATE
#define LECO_STRIPE_ABLATE 0
#endif
#ifdef LECO_STRIPE_TIMING
__device__ unsigned long long g_xtimes[32];
#define XSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_xtimes[i] = clock64(); } while (0)
#else
#define XSTAMP(i) do { } while (0)
#endif

constexpr int XBM = 64;      // token rows per stripe
constexpr int XKT = 32;      // k per step
constexpr int XPF = 4;       // weight fragments are fetched this many k-steps ahead
constexpr int XNKEY = 80;    // padded prompt keys of K (5 fragments)
constexpr int XNPOS = 96;    // padded (permuted) key positions of V^T

struct XLin {       // device view of one Linear
    const void* w; const void* dn; const void* up; const float* bias;
    unsigned w_bytes, dn_bytes, up_bytes;
    unsigned ldw_b, lddn_b, ldup_b;     // row strides in bytes
    int tf;                             // 0: LoRA off; 1 / 2: 16 / 32 stacked lora_down rows
    unsigned kfrag_b;                   // 0: w is row-major [N][K]; else w is in MFMA fragment order (leco_xlin.packed) and
                                        // this is the byte size of one 16-row fragment over all of K: (K / 32) KB
};

template <int C>
struct XCfg {
    static constexpr int NFR = C / 16;             // 16-column fragments of a C-wide result
    static constexpr int NFW = (NFR + 7) / 8;      // ... per wave (waves 0 .. NFR % 8 - 1 own NFW, the others NFW - 1)
    static constexpr int ARS = C * 2;              // activation image row stride (bytes)
    static constexpr int ABUF = XBM * ARS;
    static constexpr int GRS = 256, GBUF = XBM * GRS;   // GEGLU chunk image: [64][128] bf16, two of them
    static constexpr int KS = C / XKT;
    static constexpr int OFF_A = 0, OFF_G = ABUF, OFF_T = OFF_G + 2 * GBUF, OFF_SCR = OFF_T + XBM * 64;
    static constexpr int OFF_RED = XBM * C * 4;    // column-sum scratch of the staged output store (behind the fp32 staging tile)
    static constexpr int LDS_USED = OFF_SCR + 4096;
    static constexpr int LDS_STAGE = OFF_RED + (512 / (C / 8)) * C * 8 + C * 8;
    static constexpr int LDS_BYTES = LDS_STAGE > LDS_USED ? LDS_STAGE : LDS_USED;
    static_assert(C % 64 == 0 && (C / 8) % 16 == 8, "activation swizzle assumes a row of 8 (mod 16) 16-byte chunks");
    static_assert(NFW == 3 && NFR % 8 == 4, "fragment ownership below is written for 20 fragments: 4 waves x 3 + 4 waves x 2");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS layout does not fit");
};

struct XTailArgs {
    int m, heads, skv, rows_per_sample;
    int src_rows;     // attn / h_in / res: row r reads row r % src_rows (src_rows % 64 == 0; == m: no wrap)
    const bf16_t* attn; int64_t ld_attn;
    const bf16_t* h_in; int64_t ld_h;
    const float *ln2_g, *ln2_b, *ln3_g, *ln3_b; float ln_eps;
    const bf16_t* kp; const bf16_t* vt; float scale_log2;
    const bf16_t* res; int64_t ld_res;
    bf16_t* out; int64_t ld_out;
    float* col_stats; int stats_atom;
    int has_po;
    XLin lin[6];      // to_out1, to_q2, to_out2, ff1 (GEGLU-interleaved), ff2, proj_out
};

struct XHeadArgs {
    int m, rows_per_sample;
    const bf16_t* x; int64_t ld_x;
    const float* gn_cstats; int stats_atom, groups; const float *gn_g, *gn_b; float gn_eps;
    const float *ln1_g, *ln1_b; float ln_eps;
    bf16_t* h_out; int64_t ld_hout;
    bf16_t* qkv_out; int64_t ld_qkv;
    XLin lin[2];      // proj_in, qkv
};

// ------------------------------------------------------------------------------------------------------------------
// Shared machinery of the stripe kernels.
// ------------------------------------------------------------------------------------------------------------------
template <int C>
struct Stripe {
    using Cf = XCfg<C>;
    static constexpr int NFW = Cf::NFW, ARS = Cf::ARS, GRS = Cf::GRS, KS = Cf::KS;
    typedef f32x4 Acc[4][NFW];       // [row fragment][own column fragment]: lane holds rows 16 i + fr, columns 16 (f0 + j) + 4 fg + r

    unsigned char* lds;
    unsigned char *bufA, *bufG, *bufT;
    float* scr;
    int lane, wave, fr, fg;
    int f0, nf;          // this wave's fragments of a C-wide result: f0 .. f0 + nf - 1
    int a_sw;            // activation-image chunk swizzle of this lane's rows (row & 7 == fr & 7)
    int t_sw;            // chunk swizzle of 64-byte rows (T image): {0, 3, 2, 1}[(row >> 2) & 3]

    __device__ __forceinline__ Stripe() {
        lds = dyn_lds();
        bufA = lds + Cf::OFF_A; bufG = lds + Cf::OFF_G; bufT = lds + Cf::OFF_T;
        scr = (float*)(lds + Cf::OFF_SCR);
        const int tid = (int)threadIdx.x;
        lane = tid & 63; wave = uniform(tid >> 6);
        fr = lane & 15; fg = lane >> 4;
        f0 = wave < 4 ? 3 * wave : 12 + 2 * (wave - 4);
        nf = wave < 4 ? 3 : 2;
        a_sw = fr & 7;
        t_sw = (4 - (fr >> 2)) & 3;
    }
    __device__ __forceinline__ static void zero(Acc& a) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NFW; ++j) a[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- activation image (bufA: [64][C] bf16, chunk c of row r at position c ^ (r & 7)) ------------------------------------
    __device__ __forceinline__ int a_chunk(int chunk) const { return (chunk ^ a_sw) << 4; }
    // lane's 4 consecutive columns n .. n + 3 (n % 4 == 0) of row `row` (row & 7 == fr & 7) as bf16
    __device__ __forceinline__ void put4(int row, int n, float v0, float v1, float v2, float v3) const {
        const u32x2 w = {pack_bf2(v0, v1), pack_bf2(v2, v3)};
        *(u32x2*)(bufA + row * ARS + a_chunk(n >> 3) + ((n & 4) << 1)) = w;
    }
    // an accumulator set as bf16 into the activation image (own fragments only)
    __device__ __forceinline__ void store_a(const Acc& v) const {
#pragma unroll
        for (int j = 0; j < NFW; ++j)
            if (j < nf) {
#pragma unroll
                for (int i = 0; i < 4; ++i) put4(16 * i + fr, 16 * (f0 + j) + 4 * fg, v[i][j][0], v[i][j][1], v[i][j][2], v[i][j][3]);
            }
    }
    // an accumulator set as bf16 to global memory: columns col0 .. col0 + C of out (8-byte stores)
    __device__ __forceinline__ void store_global(const Acc& v, bf16_t* out, int64_t ld, int col0, int m0, int m) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + 16 * i + fr;
            if (row < m) {
#pragma unroll
                for (int j = 0; j < NFW; ++j)
                    if (j < nf) {
                        const u32x2 w = {pack_bf2(v[i][j][0], v[i][j][1]), pack_bf2(v[i][j][2], v[i][j][3])};
                        *(u32x2*)(out + (int64_t)row * ld + col0 + 16 * (f0 + j) + 4 * fg) = w;
                    }
            }
        }
    }
    // a [64][C] bf16 stripe (rows m0 .. m0 + 63 of a row-major matrix; rows >= m: zeros) into the activation image
    __device__ __forceinline__ void load_stripe(const bf16_t* src, int64_t ld, int m0, int m) const {
        constexpr int NCH = C / 8, NIT = (XBM * NCH + 511) / 512;
        const int tid = (int)threadIdx.x;
        u32x4 raw[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + 512 * it, row = e / NCH, ch = e - row * NCH;
            raw[it] = u32x4{0u, 0u, 0u, 0u};
            if (e < XBM * NCH && m0 + row < m) raw[it] = *(const u32x4*)(src + (int64_t)(m0 + row) * ld + ch * 8);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + 512 * it, row = e / NCH, ch = e - row * NCH;
            if (e < XBM * NCH) *(u32x4*)(bufA + row * ARS + ((ch ^ (row & 7)) << 4)) = raw[it];
        }
    }

    // ---- K loop: acc[i][j] += A[rows 16 i .., k] W_j[k]^T over NK k-steps of 32.  A = an LDS image (row stride RS; G16: the
    // GEGLU chunk image, swizzled by row & 15) starting at k-step ka0; weight fragment j of this wave = 16 rows starting
    // s<j> bytes into `rw` (j < 2) / `r2` (j == 2, wave-uniform `has2`: a third weight fragment or the lora_down rows),
    // fetched XPF k-steps ahead into registers.  No barrier, no LDS write.
    // kinc / kinc2: bytes from one k-step to the next (row-major: 64; fragment order: 1024 -- a fragment's k-steps are
    // consecutive 1 KB blocks, each lane's 16 bytes at lane * 16: one load = 8 whole cache lines)
    struct Src { buf_rsrc rw, r2; unsigned vw, v2; unsigned s0, s1, s2; unsigned kinc, kinc2; bool has2; };
    typedef bf16x8 WF[XPF][3];       // weight fragments in flight: [k-step % XPF][fragment]
    __device__ __forceinline__ void fetch(WF& wf, int slot, int kt, const Src& S) const {
        if (LECO_STRIPE_ABLATE & 2) return;
        const unsigned kb = (unsigned)kt * S.kinc;
        wf[slot][0] = buf_load16(S.rw, S.vw, S.s0 + kb);
        wf[slot][1] = buf_load16(S.rw, S.vw, S.s1 + kb);
        if (S.has2) wf[slot][2] = buf_load16(S.r2, S.v2, S.s2 + (unsigned)kt * S.kinc2);
    }
    // the first XPF k-steps of a K loop: issued by the caller AHEAD of the phase boundary in front of the loop (LayerNorm,
    // GEGLU, a barrier ...), so the loop does not start with an exposed load latency
    template <int NK, int PF = XPF>
    __device__ __forceinline__ void prefetch(WF& wf, const Src& S) const {
#pragma unroll
        for (int pf = 0; pf < PF; ++pf)
            if (pf < NK) fetch(wf, pf, pf, S);
    }
    // (PF: depth of the fragment ring of this loop; must match the prefetch<NK, PF> that fed it)
    template <int NK, int RS, bool G16, int PF = XPF>
    __device__ __forceinline__ void kloop(Acc& acc, const unsigned char* abuf, int ka0, const Src& S, WF& wf) const {
        static_assert(PF <= XPF, "fragment ring depth");
        const unsigned char* arow = abuf + fr * RS;
        bf16x8 a[4];
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            if (!(LECO_STRIPE_ABLATE & 4) || kt == 0) {
                const int ch = 4 * (ka0 + kt) + fg;
                const unsigned char* ap = arow + (G16 ? ((ch ^ fr) << 4) : a_chunk(ch));
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(ap + 16 * i * RS);
            }
            const int slot = kt % PF;
            if (LECO_STRIPE_ABLATE & 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][0][0] += __uint_as_float((unsigned)(wf[slot][0][0] ^ wf[slot][1][0] ^ a[i][0]));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][0] = mfma16(wf[slot][0], a[i], acc[i][0]);
                    acc[i][1] = mfma16(wf[slot][1], a[i], acc[i][1]);
                }
                if (S.has2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][2] = mfma16(wf[slot][2], a[i], acc[i][2]);
                }
            }
            if (kt + PF < NK) fetch(wf, slot, kt + PF, S);
        }
    }
    // weight-fragment sources of this wave for rows [n0, n0 + C) of Linear L at k offset k0 (elements); `t_here`: the stacked
    // lora_down rows ride along as the third fragment of the waves on T duty (wave 4: rows 0 .. 15, wave 5: rows 16 .. 31)
    __device__ __forceinline__ Src src_c(const XLin& L, int n0, int k0, bool t_here) const {
        Src S;
        S.rw = make_rsrc(L.w, L.w_bytes);
        const bool pk = L.kfrag_b != 0;
        // fragment fi of the weight at k offset k0: row-major -> row 16 fi, column k0; fragment order -> block fi, k-step k0 / 32
        const unsigned fstride = pk ? L.kfrag_b : 16u * L.ldw_b;
        const unsigned kb = pk ? (unsigned)(k0 / XKT) * 1024u : (unsigned)k0 * 2u;
        S.vw = pk ? (unsigned)(lane << 4) : (unsigned)fr * L.ldw_b + (unsigned)(fg << 4);
        S.kinc = pk ? 1024u : (unsigned)(XKT * 2);
        S.s0 = (unsigned)(n0 / 16 + f0) * fstride + kb;
        S.s1 = S.s0 + fstride;
        const bool duty = t_here && L.tf && (wave == 4 || (wave == 5 && L.tf == 2));
        if (nf == 3) {
            S.r2 = S.rw; S.v2 = S.vw; S.s2 = S.s1 + fstride; S.kinc2 = S.kinc; S.has2 = true;
        } else {
            S.r2 = make_rsrc(duty ? L.dn : L.w, duty ? L.dn_bytes : L.w_bytes);
            S.v2 = (unsigned)fr * L.lddn_b + (unsigned)(fg << 4);
            S.s2 = (unsigned)(16 * (wave - 4)) * L.lddn_b + (unsigned)k0 * 2u;
            S.kinc2 = (unsigned)(XKT * 2);
            S.has2 = duty;
        }
        // (keep the scalar offsets provably wave-uniform: otherwise the loads are issued through waterfall loops)
        S.s0 = (unsigned)uniform((int)S.s0);
        S.s1 = (unsigned)uniform((int)S.s1);
        S.s2 = (unsigned)uniform((int)S.s2);
        S.kinc2 = (unsigned)uniform((int)S.kinc2);
        return S;
    }
    // T (fp32: third fragment of the waves on duty) -> bf16 into the T image ([64][32], 64-byte rows, chunk c of row r at
    // position c ^ {0, 3, 2, 1}[(r >> 2) & 3])
    __device__ __forceinline__ void put_t(const f32x4& t, int row, int tq) const {
        const u32x2 w = {pack_bf2(t[0], t[1]), pack_bf2(t[2], t[3])};
        *(u32x2*)(bufT + row * 64 + (((2 * tq + (fg >> 1)) ^ t_sw) << 4) + ((fg & 1) << 3)) = w;
    }
    __device__ __forceinline__ void write_t(Acc& acc, int tf) const {
        if (nf == 2 && tf && (wave == 4 || (wave == 5 && tf == 2))) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                put_t(acc[i][2], 16 * i + fr, wave - 4);
                acc[i][2] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    // K-extension: acc += T (scale up)^T for rows [n0, n0 + C) of up.  The up fragments are loaded ahead (ext_load, before
    // the K loop); T must be visible (barrier) before ext_apply.
    __device__ __forceinline__ void ext_load(bf16x8 (&uf)[NFW], const XLin& L, int n0) const {
        if (!L.tf) return;
        const buf_rsrc ru = make_rsrc(L.up, L.up_bytes);
        const unsigned vu = (unsigned)fr * L.ldup_b + (unsigned)(fg << 4);
        const unsigned s0 = (unsigned)uniform((int)((unsigned)(n0 + 16 * f0) * L.ldup_b));
#pragma unroll
        for (int j = 0; j < NFW; ++j)
            if (j < nf) uf[j] = buf_load16(ru, vu, s0 + (unsigned)(16 * j) * L.ldup_b);
    }
    __device__ __forceinline__ void ext_apply(Acc& acc, const bf16x8 (&uf)[NFW]) const {
        const unsigned char* tp = bufT + fr * 64 + ((fg ^ t_sw) << 4);
        bf16x8 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(tp + 16 * i * 64);
#pragma unroll
        for (int j = 0; j < NFW; ++j)
            if (j < nf) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = mfma16(uf[j], a[i], acc[i][j]);
            }
    }
    // a whole Linear with N = K = C on the activation image, accumulated into acc (+ bias).  `wf` holds the first k-steps
    // (prefetch<KS>(wf, S) by the caller, ahead of whatever precedes the Linear).  Contains one barrier when the Linear carries
    // a LoRA (all waves have then finished reading the image).
    __device__ __forceinline__ void linear(Acc& acc, const XLin& L, const Src& S, WF& wf) const {
        f32x4 b[NFW];
        bf16x8 uf[NFW];
        load_bias(b, L.bias, 0);             // (issued ahead of the K loop: their latency hides behind it)
        ext_load(uf, L, 0);
        kloop<KS, ARS, false>(acc, bufA, 0, S, wf);
        if (L.tf) {
            write_t(acc, L.tf);
            barrier_keep_dma();
            ext_apply(acc, uf);
        }
        add_bias(acc, b);
    }
    __device__ __forceinline__ void load_bias(f32x4 (&b)[NFW], const float* bias, int n0) const {
#pragma unroll
        for (int j = 0; j < NFW; ++j) {
            b[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (bias && j < nf) b[j] = *(const f32x4*)(bias + n0 + 16 * (f0 + j) + 4 * fg);
        }
    }
    __device__ __forceinline__ void add_bias(Acc& acc, const f32x4 (&b)[NFW]) const {
#pragma unroll
        for (int j = 0; j < NFW; ++j)
            if (j < nf) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[i][j][0] += b[j][0]; acc[i][j][1] += b[j][1]; acc[i][j][2] += b[j][2]; acc[i][j][3] += b[j][3]; }
            }
    }
    __device__ __forceinline__ void add_bias(Acc& acc, const float* bias, int n0) const {
        if (!bias) return;
#pragma unroll
        for (int j = 0; j < NFW; ++j)
            if (j < nf) {
                const f32x4 b = *(const f32x4*)(bias + n0 + 16 * (f0 + j) + 4 * fg);
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[i][j][0] += b[0]; acc[i][j][1] += b[1]; acc[i][j][2] += b[2]; acc[i][j][3] += b[3]; }
            }
    }

    // ---- LayerNorm of the register-resident stream -> bf16 activation image.  Every wave must have finished reading the
    // image's previous contents when it calls this (the first barrier in here then makes that true for all of them).
    // (gamma / beta of the lane's columns: loaded by the caller AHEAD of the weight prefetch that precedes the LayerNorm --
    // vector-memory operations complete in order, behind a prefetch they would arrive last)
    __device__ __forceinline__ void ln_params(f32x4 (&g)[NFW], f32x4 (&b)[NFW], const float* gamma, const float* beta) const {
#pragma unroll
        for (int j = 0; j < NFW; ++j) {
            g[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            b[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < nf) {
                g[j] = *(const f32x4*)(gamma + 16 * (f0 + j) + 4 * fg);
                b[j] = *(const f32x4*)(beta + 16 * (f0 + j) + 4 * fg);
            }
        }
    }
    __device__ __forceinline__ void layernorm(const Acc& h, const f32x4 (&g)[NFW], const f32x4 (&b)[NFW], float eps) const {
        float mean[4], rstd[4];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            float* part = scr + pass * (XBM * 8);              // [64][8]
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NFW; ++j)
                    if (j < nf) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float d = pass ? h[i][j][r] - mean[i] : h[i][j][r];
                            s += pass ? d * d : d;
                        }
                    }
                s = rows4_sum(s);
                if (fg == 0) part[(16 * i + fr) * 8 + wave] = s;
            }
            barrier_keep_dma();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 p0 = *(const f32x4*)(part + (16 * i + fr) * 8), p1 = *(const f32x4*)(part + (16 * i + fr) * 8 + 4);
                const float t = ((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]));
                if (pass == 0) mean[i] = t * (1.0f / (float)C);
                else rstd[i] = rsqrtf(t * (1.0f / (float)C) + eps);
            }
        }
#pragma unroll
        for (int j = 0; j < NFW; ++j)
            if (j < nf) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    put4(16 * i + fr, 16 * (f0 + j) + 4 * fg, (h[i][j][0] - mean[i]) * rstd[i] * g[j][0] + b[j][0],
                         (h[i][j][1] - mean[i]) * rstd[i] * g[j][1] + b[j][1], (h[i][j][2] - mean[i]) * rstd[i] * g[j][2] + b[j][2],
                         (h[i][j][3] - mean[i]) * rstd[i] * g[j][3] + b[j][3]);
            }
    }

    // ---- staged output: the [64][C] fp32 result goes through LDS (fp32 [64][C] from LDS offset 0, 16-byte chunks
    // XOR-swizzled by row & 15) so that the global side moves whole 16-byte row segments: + bias + residual, bf16 store, and
    // the per-atom {sum, sumsq} of the stored values for the GroupNorm that follows (leco_gemm_args.col_stats).
    // Must be called by all waves; overlays every LDS image.
    __device__ __forceinline__ void store_out(const Acc& v, const float* bias, const bf16_t* res, int64_t ld_res, bf16_t* out,
                                              int64_t ld_out, int m0, int m, float* col_stats, int stats_atom,
                                              int rows_per_sample) const {
        constexpr int NCH = C / 8, RG = 512 / NCH;        // 16-byte output chunks per row; row groups (threads NCH * RG)
        unsigned char* stg = lds;                         // fp32 [64][C], row stride 4 C bytes
        const int tid = (int)threadIdx.x;
        const int ch = tid % NCH, rg = tid / NCH;
        constexpr int NIT = (XBM + RG - 1) / RG;
        u32x4 rres[NIT];
        float bs[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) bs[r] = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) rres[it] = u32x4{0u, 0u, 0u, 0u};
        if (rg < RG) {       // residual / bias loads first: their latency hides behind the staging
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = rg + RG * it;
                if (res && row < XBM && m0 + row < m) rres[it] = *(const u32x4*)(res + (int64_t)(m0 + row) * ld_res + ch * 8);
            }
            if (bias) {
                const f32x4 b0 = *(const f32x4*)(bias + ch * 8), b1 = *(const f32x4*)(bias + ch * 8 + 4);
                bs[0] = b0[0]; bs[1] = b0[1]; bs[2] = b0[2]; bs[3] = b0[3]; bs[4] = b1[0]; bs[5] = b1[1]; bs[6] = b1[2]; bs[7] = b1[3];
            }
        }
        barrier_keep_dma();                               // every wave is done with the LDS images
#pragma unroll
        for (int j = 0; j < NFW; ++j)
            if (j < nf) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 16 * i + fr, c16 = (16 * (f0 + j) + 4 * fg) >> 2;
                    *(f32x4*)(stg + row * (4 * C) + ((c16 ^ (row & 15)) << 4)) = v[i][j];
                }
            }
        barrier_keep_dma();
        float s1[8], s2[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
        if (rg < RG) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = rg + RG * it;
                if (row >= XBM || m0 + row >= m) continue;
                const f32x4 v0 = *(const f32x4*)(stg + row * (4 * C) + (((2 * ch) ^ (row & 15)) << 4));
                const f32x4 v1 = *(const f32x4*)(stg + row * (4 * C) + (((2 * ch + 1) ^ (row & 15)) << 4));
                float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    x[2 * r] += bs[2 * r] + bf2f((bf16_t)(rres[it][r] & 0xffffu));
                    x[2 * r + 1] += bs[2 * r + 1] + bf2f((bf16_t)(rres[it][r] >> 16));
                }
                const u32x4 o = {pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7])};
                *(u32x4*)(out + (int64_t)(m0 + row) * ld_out + ch * 8) = o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = bf2f((bf16_t)(o[r] & 0xffffu)), b = bf2f((bf16_t)(o[r] >> 16));
                    s1[2 * r] += a; s2[2 * r] += a * a;
                    s1[2 * r + 1] += b; s2[2 * r + 1] += b * b;
                }
            }
        }
        if (!col_stats) return;
        // column sums: the RG row groups meet in LDS (fp32 [RG][C][2] behind the staging tile), thread c sums columns 2 c,
        // 2 c + 1 over the row groups, then one thread per atom adds its columns and sends one pair of atomics
        float* red = (float*)(lds + Cf::OFF_RED);
        if (rg < RG) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const u32x2 w = {__float_as_uint(s1[r]), __float_as_uint(s2[r])};
                *(u32x2*)(red + (rg * C + ch * 8 + r) * 2) = w;
            }
        }
        barrier_keep_dma();
        float* csum = red + RG * C * 2;                   // fp32 [C][2]
        if (tid < C / 2) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                const f32x4 a = *(const f32x4*)(red + (g * C + 2 * tid) * 2);
                t[0] += a[0]; t[1] += a[1]; t[2] += a[2]; t[3] += a[3];
            }
            *(f32x4*)(csum + 4 * tid) = t;
        }
        barrier_keep_dma();
        const int natom = C / stats_atom;
        if (tid < natom && m0 < m) {
            float t1 = 0.f, t2 = 0.f;
            for (int c = tid * stats_atom; c < (tid + 1) * stats_atom; c += 2) {       // (stats_atom is even)
                const f32x4 a = *(const f32x4*)(csum + 2 * c);
                t1 += a[0] + a[2];
                t2 += a[1] + a[3];
            }
            const int b = m0 / rows_per_sample;
            atomicAdd(col_stats + ((int64_t)b * natom + tid) * 2, t1);
            atomicAdd(col_stats + ((int64_t)b * natom + tid) * 2 + 1, t2);
        }
    }
};

// ======================================================================================================================
// Tail of a BasicTransformerBlock (+ Transformer2DModel.proj_out): see the file header.
// D = head dim of the cross-attention (40: SD1.x, 64: SD2.x at C = 320).
// ======================================================================================================================
template <int C, int D>
__global__ __launch_bounds__(512) void xblock_tail_kernel(const XTailArgs p) {
    using St = Stripe<C>;
    using Cf = XCfg<C>;
    using Acc = typename St::Acc;
    constexpr int NFW = Cf::NFW, ARS = Cf::ARS, KS = Cf::KS, GRS = Cf::GRS;
    constexpr int DV = (D + 15) / 16 * 16, NFD = DV / 16;
    constexpr bool ONES = DV > D;
    constexpr int NCHUNK = 4 * C / 128;             // feed-forward chunks of 128 hidden units
    St st;
    const int wave = st.wave, fr = st.fr, fg = st.fg, f0 = st.f0, nf = st.nf;
    const int m0 = (int)blockIdx.x * XBM, M = p.m;
    const int ms = m0 % p.src_rows;             // first row of the stripe in the (possibly batch-shared) input tensors

    // ---- prologue: the first weight fragments of sweep 1; the self-attention output stripe -> activation image; the T image
    // zeroed (its columns 16 .. 31 are only written by rank-stacks > 16); the residual stream h0 -> registers
    typename St::WF wf;
    typename St::Src S = st.src_c(p.lin[0], 0, 0, true);
    st.template prefetch<KS>(wf, S);
    st.load_stripe(p.attn, p.ld_attn, ms, ms + (M - m0));
    *(u32x2*)(st.bufT + (int)threadIdx.x * 8) = u32x2{0u, 0u};
    Acc h;
    St::zero(h);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + 16 * i + fr;
#pragma unroll
        for (int j = 0; j < NFW; ++j)
            if (j < nf && row < M) {
                const u32x2 raw = *(const u32x2*)(p.h_in + (int64_t)(ms + 16 * i + fr) * p.ld_h + 16 * (f0 + j) + 4 * fg);
                h[i][j] = f32x4{bf2f((bf16_t)(raw[0] & 0xffffu)), bf2f((bf16_t)(raw[0] >> 16)), bf2f((bf16_t)(raw[1] & 0xffffu)),
                                bf2f((bf16_t)(raw[1] >> 16))};
            }
    }
    barrier_keep_dma();

    // ---- 1. h1 = a1 Wo1^T + bo1 + h0
    XSTAMP(0);
    st.linear(h, p.lin[0], S, wf);
    XSTAMP(1);
    // ---- 2. l2 = LN2(h1) -> activation image (a1 is dead behind LN's first barrier); sweep 2's first fragments fly meanwhile
    f32x4 lg[NFW], lb[NFW];
    st.ln_params(lg, lb, p.ln2_g, p.ln2_b);
    S = st.src_c(p.lin[1], 0, 0, true);
    st.template prefetch<KS>(wf, S);
    st.layernorm(h, lg, lb, p.ln_eps);
    barrier_keep_dma();
    XSTAMP(2);
    // ---- 3. q2 = l2 Wq2^T -> activation image (bf16), in place of l2.  K / V^T fragments of the wave's first head are fetched
    // behind the K loop, ahead of the barriers in front of the cross-attention.
    constexpr int DVc = (D + 15) / 16 * 16, NFDc = DVc / 16;
    bf16x8 kf[5][2], vf[NFDc][3];
    auto load_kv = [&](int hd) {
        const int b = m0 / p.rows_per_sample;
        const bf16_t* kp = p.kp + (int64_t)(b * p.heads + hd) * (XNKEY * 64);
        const bf16_t* vt = p.vt + (int64_t)(b * p.heads + hd) * (DVc * XNPOS);
#pragma unroll
        for (int f = 0; f < 5; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kf[f][ks] = *(const bf16x8*)(kp + (16 * f + fr) * 64 + 32 * ks + 8 * fg);
#pragma unroll
        for (int fd = 0; fd < NFDc; ++fd)
#pragma unroll
            for (int s = 0; s < 3; ++s) vf[fd][s] = *(const bf16x8*)(vt + (16 * fd + fr) * XNPOS + 32 * s + 8 * fg);
    };
    {
        Acc q;
        St::zero(q);
        st.linear(q, p.lin[1], S, wf);
        if (wave < p.heads && !(LECO_STRIPE_ABLATE & 8)) load_kv(wave);
        if (!p.lin[1].tf) barrier_keep_dma();       // (with a LoRA, linear()'s own barrier already follows every wave's K loop)
        st.store_a(q);
    }
    barrier_keep_dma();             // q2 complete
    XSTAMP(3);

    // ---- 4. cross-attention: one wave per head, IN PLACE (a wave reads and writes only its head's columns);
    // S^T = K Q^T and O^T = V^T P^T (swapped: a lane owns one query row)
    XSTAMP(4);
    if (!(LECO_STRIPE_ABLATE & 8)) {
        for (int hd = wave; hd < p.heads; hd += 8) {
            if (hd != wave) load_kv(hd);
#pragma unroll 1
            for (int u = 0; u < 4; ++u) {
                // Q fragments (MFMA B operand: column = query row, k = head dim); k-groups beyond D are zero
                bf16x8 qf[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bool ok = 32 * ks + 8 * fg < D;
                    const int chunk = ok ? hd * (D / 8) + 4 * ks + fg : 0;
                    qf[ks] = *(const bf16x8*)(st.bufA + (16 * u + fr) * ARS + st.a_chunk(chunk));
                    if (!ok) qf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                }
                f32x4 sc[5];
                float mx = -INFINITY;
#pragma unroll
                for (int f = 0; f < 5; ++f) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
                    a = mfma16(kf[f][0], qf[0], a);
                    a = mfma16(kf[f][1], qf[1], a);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (16 * f + 4 * fg + r >= p.skv) a[r] = -INFINITY;
                        mx = fmaxf(mx, a[r]);
                    }
                    sc[f] = a;
                }
                mx = rows4_max(mx);
                const float mneg = mx * p.scale_log2;
                u32x4 pw[3];
                float rs = 0.f;
#pragma unroll
                for (int f = 0; f < 5; ++f) {
                    const float e0 = fast_exp2(fmaf(sc[f][0], p.scale_log2, -mneg)), e1 = fast_exp2(fmaf(sc[f][1], p.scale_log2, -mneg));
                    const float e2 = fast_exp2(fmaf(sc[f][2], p.scale_log2, -mneg)), e3 = fast_exp2(fmaf(sc[f][3], p.scale_log2, -mneg));
                    if (!ONES) rs += (e0 + e1) + (e2 + e3);
                    pw[f >> 1][(f & 1) * 2] = pack_bf2(e0, e1);
                    pw[f >> 1][(f & 1) * 2 + 1] = pack_bf2(e2, e3);
                }
                pw[2][2] = 0u;
                pw[2][3] = 0u;
                f32x4 o[NFD];
#pragma unroll
                for (int fd = 0; fd < NFD; ++fd) {
                    o[fd] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 3; ++s) o[fd] = mfma16(vf[fd][s], __builtin_bit_cast(bf16x8, pw[s]), o[fd]);
                }
                float l;
                if constexpr (ONES) {
                    l = shfl(o[D / 16][D % 4], fr + 16 * ((D % 16) / 4));
                } else {
                    l = rs;
                    l = rows4_sum(l);
                }
                const float inv = 1.f / l;
#pragma unroll
                for (int fd = 0; fd < NFD; ++fd) {
                    const int d = 16 * fd + 4 * fg;
                    if (d < D) st.put4(16 * u + fr, hd * D + d, o[fd][0] * inv, o[fd][1] * inv, o[fd][2] * inv, o[fd][3] * inv);
                }
            }
        }
    }
    S = st.src_c(p.lin[2], 0, 0, true);
    st.template prefetch<KS>(wf, S);
    barrier_keep_dma();             // a2 complete
    // ---- 5. h2 = a2 Wo2^T + bo2 + h1
    XSTAMP(5);
    st.linear(h, p.lin[2], S, wf);
    XSTAMP(6);

    // ---- 6 + 7. l3 = LN3(h2) -> activation image, then the feed-forward in chunks of 128 hidden units.  FF1 chunk: wave w
    // owns hidden columns 16 w .. + 16 of the chunk: weight rows vb + fr (value) and vb + 64 + fr (gate) of the 64-interleaved
    // GEGLU weight image; the stacked lora_down rows ride in the first chunk as a third fragment of waves 0 / 1.  The bf16
    // chunk goes to one of two LDS images ([64][128], chunk c of row r at position c ^ (r & 15)); FF2 reads it from there (its
    // lora_down projection accumulates over the chunks in the spare third fragment of waves 4 / 5).  Weight fragments of the
    // NEXT K loop are always in flight across the phase boundary in front of it.
    {
        const XLin& L1 = p.lin[3];
        const XLin& L2 = p.lin[4];
        const int vb = 16 * wave + (wave >= 4 ? 64 : 0);
        typename St::Src S1;
        S1.rw = make_rsrc(L1.w, L1.w_bytes);
        const bool pk1 = L1.kfrag_b != 0;
        const unsigned fstride1 = pk1 ? L1.kfrag_b : 16u * L1.ldw_b;
        S1.vw = pk1 ? (unsigned)(st.lane << 4) : (unsigned)fr * L1.ldw_b + (unsigned)(fg << 4);
        S1.kinc = pk1 ? 1024u : (unsigned)(XKT * 2);
        S1.kinc2 = (unsigned)(XKT * 2);
        const bool duty1 = L1.tf && (wave == 0 || (wave == 1 && L1.tf == 2));
        S1.r2 = make_rsrc(duty1 ? L1.dn : L1.w, duty1 ? L1.dn_bytes : L1.w_bytes);
        S1.v2 = (unsigned)fr * L1.lddn_b + (unsigned)(fg << 4);
        S1.s2 = (unsigned)uniform((int)((unsigned)(16 * wave) * L1.lddn_b));
        auto s1_chunk = [&](int c) {
            S1.s0 = (unsigned)uniform((int)((unsigned)((256 * c + vb) / 16) * fstride1));
            S1.s1 = (unsigned)uniform((int)(S1.s0 + 4u * fstride1));
            S1.has2 = duty1 && c == 0;
        };
        typename St::WF wf2;
        s1_chunk(0);
        f32x4 b2[NFW];
        st.ln_params(lg, lb, p.ln3_g, p.ln3_b);
        st.load_bias(b2, L2.bias, 0);
        st.template prefetch<KS>(wf, S1);           // (flies during LayerNorm)
        st.layernorm(h, lg, lb, p.ln_eps);
        st.add_bias(h, b2);                         // ff.net.2 bias: h becomes the accumulator of h3
        barrier_keep_dma();
        XSTAMP(7);
        typename St::Src S2 = st.src_c(L2, 0, 0, true);
        const unsigned s2_0 = S2.s0, s2_1 = S2.s1, s2_2 = S2.s2;
        const unsigned cinc = 4u * S2.kinc, cinc2 = 4u * S2.kinc2;       // a chunk = 4 k-steps
        const buf_rsrc ru1 = make_rsrc(L1.tf ? L1.up : L1.w, L1.tf ? L1.up_bytes : L1.w_bytes);
        const unsigned vu1 = (unsigned)fr * L1.ldup_b + (unsigned)(fg << 4);
#pragma unroll 1
        for (int c = 0; c < NCHUNK; ++c) {
            Acc u;
            St::zero(u);
            f32x4 bv = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
            if (L1.bias) {
                bv = *(const f32x4*)(L1.bias + 256 * c + vb + 4 * fg);
                bg = *(const f32x4*)(L1.bias + 256 * c + vb + 64 + 4 * fg);
            }
            bf16x8 uv = {0, 0, 0, 0, 0, 0, 0, 0}, ug = {0, 0, 0, 0, 0, 0, 0, 0};
            if (L1.tf) {
                const unsigned su = (unsigned)uniform((int)((unsigned)(256 * c + vb) * L1.ldup_b));
                uv = buf_load16(ru1, vu1, su);
                ug = buf_load16(ru1, vu1, su + 64u * L1.ldup_b);
            }
            st.template kloop<KS, ARS, false>(u, st.bufA, 0, S1, wf);
            // FF2's first fragments of this chunk fly during the K-extension, GEGLU and the barrier
            S2.s0 = (unsigned)uniform((int)(s2_0 + cinc * (unsigned)c));
            S2.s1 = (unsigned)uniform((int)(s2_1 + cinc * (unsigned)c));
            S2.s2 = (unsigned)uniform((int)(s2_2 + cinc2 * (unsigned)c));
            st.template prefetch<4, 3>(wf2, S2);
            if (L1.tf) {
                if (c == 0) {
                    if (duty1) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) st.put_t(u[i][2], 16 * i + fr, wave);
                    }
                    barrier_keep_dma();
                }
                // K-extension of the chunk: T (scale up)^T for its value / gate rows
                const unsigned char* tp = st.bufT + fr * 64 + ((fg ^ st.t_sw) << 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16x8 a = *(const bf16x8*)(tp + 16 * i * 64);
                    u[i][0] = mfma16(uv, a, u[i][0]);
                    u[i][1] = mfma16(ug, a, u[i][1]);
                }
            }
            if (c == 0) XSTAMP(11);
            // GEGLU: value * gelu(gate) -> bf16 chunk image c & 1
            {
                unsigned char* gb = st.bufG + (c & 1) * Cf::GBUF;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float g[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = u[i][1][r] + bg[r];
                        // gelu (erf form, F.gelu) with erf by Abramowitz-Stegun 7.1.26, as gemm.hip's fused GEGLU epilogue
                        const float z = fabsf(x) * 0.7071067811865476f;
                        const float t = fast_rcp(1.f + 0.3275911f * z);
                        const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
                        const float hq = 0.5f * x * poly * fast_exp2(-z * z * 1.4426950408889634f);
                        g[r] = (u[i][0][r] + bv[r]) * (x >= 0.f ? x - hq : hq);
                    }
                    const u32x2 w = {pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3])};
                    *(u32x2*)(gb + (16 * i + fr) * GRS + (((2 * wave + (fg >> 1)) ^ fr) << 4) + ((fg & 1) << 3)) = w;
                }
            }
            // the next chunk's FF1 fragments fly during FF2
            if (c + 1 < NCHUNK) {
                s1_chunk(c + 1);
                st.template prefetch<KS>(wf, S1);
            }
            // chunk visible; every wave has finished FF2 of chunk c - 1, i.e. image (c + 1) & 1 is free again
            barrier_keep_dma();
            if (c == 0) XSTAMP(12);
            // FF2 partial sums: h += g[:, chunk] W2[:, 128 c .. + 128]^T
            st.template kloop<4, GRS, true, 3>(h, st.bufG + (c & 1) * Cf::GBUF, 0, S2, wf2);
            if (c == 0) XSTAMP(13);
        }
        // proj_out's first fragments fly during the last K-extension and the hand-over of h3
        if (p.has_po) {
            S = st.src_c(p.lin[5], 0, 0, true);
            st.template prefetch<KS>(wf, S);
        }
        if (L2.tf) {
            bf16x8 uf[NFW];
            st.ext_load(uf, L2, 0);
            st.write_t(h, L2.tf);
            barrier_keep_dma();
            st.ext_apply(h, uf);
        }
    }

    // ---- 8. out = proj_out(h3) + x   (or h3 itself when the Transformer2DModel goes on with another block)
    XSTAMP(8);
    if (p.has_po) {
        barrier_keep_dma();                     // every wave is done with FF1's reads of l3 (and with the T image)
        st.store_a(h);
        barrier_keep_dma();
        Acc y;
        St::zero(y);
        const XLin& L = p.lin[5];
        bf16x8 uf[NFW];
        st.ext_load(uf, L, 0);
        st.template kloop<KS, ARS, false>(y, st.bufA, 0, S, wf);
        if (L.tf) {
            st.write_t(y, L.tf);
            barrier_keep_dma();
            st.ext_apply(y, uf);
        }
        XSTAMP(9);
        st.store_out(y, L.bias, p.res ? p.res + (int64_t)(ms - m0) * p.ld_res : nullptr, p.ld_res, p.out, p.ld_out, m0, M, p.col_stats,
                     p.stats_atom, p.rows_per_sample);
        XSTAMP(10);
    } else {
        st.store_out(h, nullptr, nullptr, 0, p.out, p.ld_out, m0, M, p.col_stats, p.stats_atom, p.rows_per_sample);
    }
}

// ======================================================================================================================
// Head of a Transformer2DModel + its first BasicTransformerBlock up to the self-attention core:
//   n = GroupNorm(x) (from the statistics the producer of x left: leco_gemm_args.col_stats) ;  p = proj_in(n) -> h_out ;
//   qkv = attn1.to_q|to_k|to_v(LN1(p)) -> qkv_out   (three 320-column sweeps sharing one lora_down projection)
// ======================================================================================================================
template <int C>
__global__ __launch_bounds__(512) void xblock_head_kernel(const XHeadArgs p) {
    using St = Stripe<C>;
    using Cf = XCfg<C>;
    using Acc = typename St::Acc;
    constexpr int ARS = Cf::ARS, KS = Cf::KS;
    St st;
    const int m0 = (int)blockIdx.x * XBM, M = p.m;
    const int tid = (int)threadIdx.x;
    typename St::WF wf;
    typename St::Src S = st.src_c(p.lin[0], 0, 0, true);
    st.template prefetch<KS>(wf, S);
    *(u32x2*)(st.bufT + tid * 8) = u32x2{0u, 0u};
    if (p.gn_cstats) {
        // ---- GroupNorm apply: per-channel {mean, rstd * gamma, beta} of this stripe's sample in LDS, then one pass over the
        // stripe (16-byte chunks, coalesced rows) -> bf16 into the activation image
        float* cmean = st.scr;              // [C]   (3 C floats <= 4 KB of scratch)
        float* cscale = st.scr + C;
        float* cbeta = st.scr + 2 * C;
        static_assert(3 * C * 4 <= 4096, "GroupNorm scratch");
        const int b = m0 / p.rows_per_sample, A = p.stats_atom, cg = C / p.groups, ag = cg / A, natom = C / A;
        for (int c = tid; c < C; c += 512) {
            const int g = c / cg;
            float a0 = 0.f, a1 = 0.f;
            for (int a = g * ag; a < (g + 1) * ag; ++a) {
                a0 += p.gn_cstats[((int64_t)b * natom + a) * 2];
                a1 += p.gn_cstats[((int64_t)b * natom + a) * 2 + 1];
            }
            const float inv_n = 1.f / ((float)p.rows_per_sample * (float)cg);
            const float mu = a0 * inv_n, var = a1 * inv_n - mu * mu;
            cmean[c] = mu;
            cscale[c] = rsqrtf(fmaxf(var, 0.f) + p.gn_eps) * p.gn_g[c];
            cbeta[c] = p.gn_b[c];
        }
        constexpr int NCH = C / 8, NIT = (XBM * NCH + 511) / 512;
        u32x4 raw[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + 512 * it, row = e / NCH, ch = e - row * NCH;
            raw[it] = u32x4{0u, 0u, 0u, 0u};
            if (e < XBM * NCH && m0 + row < M) raw[it] = *(const u32x4*)(p.x + (int64_t)(m0 + row) * p.ld_x + ch * 8);
        }
        barrier_keep_dma();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + 512 * it, row = e / NCH, ch = e - row * NCH;
            if (e >= XBM * NCH) break;
            float o[8];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const f32x4 mu = *(const f32x4*)(cmean + ch * 8 + 4 * k), sc = *(const f32x4*)(cscale + ch * 8 + 4 * k);
                const f32x4 be = *(const f32x4*)(cbeta + ch * 8 + 4 * k);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned w = raw[it][2 * k + (r >> 1)];
                    const float xv = bf2f((bf16_t)((r & 1) ? (w >> 16) : (w & 0xffffu)));
                    o[4 * k + r] = (xv - mu[r]) * sc[r] + be[r];
                }
            }
            const u32x4 w = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
            *(u32x4*)(st.bufA + row * ARS + ((ch ^ (row & 7)) << 4)) = w;
        }
    } else {
        st.load_stripe(p.x, p.ld_x, m0, M);
    }
    barrier_keep_dma();
    // ---- p = proj_in(n) + bias -> h_out (the residual stream the tail kernel starts from)
    Acc h;
    St::zero(h);
    st.linear(h, p.lin[0], S, wf);
    st.store_global(h, p.h_out, p.ld_hout, 0, m0, M);
    // ---- l1 = LN1(p) -> activation image (n is dead behind LN's first barrier); the q sweep's first fragments fly meanwhile
    f32x4 lg[Cf::NFW], lb[Cf::NFW];
    st.ln_params(lg, lb, p.ln1_g, p.ln1_b);
    S = st.src_c(p.lin[1], 0, 0, true);
    st.template prefetch<KS>(wf, S);
    st.layernorm(h, lg, lb, p.ln_eps);
    barrier_keep_dma();
    // ---- q | k | v: three sweeps over 320 weight rows each; the stacked lora_down rows ride in the first one only
    {
        const XLin& L = p.lin[1];
#pragma unroll 1
        for (int c = 0; c < 3; ++c) {
            Acc y;
            St::zero(y);
            f32x4 b[Cf::NFW];
            bf16x8 uf[Cf::NFW];
            st.load_bias(b, L.bias, C * c);
            st.ext_load(uf, L, C * c);
            st.template kloop<KS, ARS, false>(y, st.bufA, 0, S, wf);
            if (c + 1 < 3) {        // the next sweep's first fragments fly during the K-extension and the stores
                S = st.src_c(L, C * (c + 1), 0, false);
                st.template prefetch<KS>(wf, S);
            }
            if (L.tf) {
                if (c == 0) {
                    st.write_t(y, L.tf);
                    barrier_keep_dma();
                }
                st.ext_apply(y, uf);
            }
            st.add_bias(y, b);
            st.store_global(y, p.qkv_out, p.ld_qkv, C * c, m0, M);
        }
    }
}

// K / V of the cross-attention, once per step (they depend only on the prompt embeddings and the LoRA weights): from the
// fused projection output kv [B * skv][2 C] (K | V) to the fragment-friendly images the tail kernel loads straight into
// MFMA operands:
//   kp [B][H][80][64]  : K rows (keys >= skv and head-dim columns >= D zero)
//   vt [B][H][DV][96]  : V^T with the key order of every 32-key block permuted to the P^T register order of the swapped
//                        products (position 8 g + t of block s  <->  key 32 s + 16 (t >> 2) + 4 g + (t & 3)); DV > D: row D
//                        holds 1.0 for keys < skv (the PV MFMA then accumulates the softmax row sum), the rest zero
__global__ __launch_bounds__(256) void xattn_prep_kernel(const bf16_t* kv, int64_t ld_kv, bf16_t* kp, bf16_t* vt, int batch,
                                                         int heads, int skv, int d) {
    const int dv = (d + 15) / 16 * 16, c = heads * d;
    const int64_t nk = (int64_t)batch * heads * XNKEY * 64, nv = (int64_t)batch * heads * dv * XNPOS;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nk + nv; e += (int64_t)gridDim.x * 256) {
        if (e < nk) {
            const int col = (int)(e % 64), key = (int)((e / 64) % XNKEY);
            const int hd = (int)((e / (64 * XNKEY)) % heads), b = (int)(e / ((int64_t)64 * XNKEY * heads));
            kp[e] = (key < skv && col < d) ? kv[((int64_t)b * skv + key) * ld_kv + hd * d + col] : (bf16_t)0;
        } else {
            const int64_t t = e - nk;
            const int pos = (int)(t % XNPOS), row = (int)((t / XNPOS) % dv);
            const int hd = (int)((t / ((int64_t)XNPOS * dv)) % heads), b = (int)(t / ((int64_t)XNPOS * dv * heads));
            const int s = pos >> 5, g = (pos >> 3) & 3, tt = pos & 7;
            const int key = 32 * s + 16 * (tt >> 2) + 4 * g + (tt & 3);
            bf16_t val = 0;
            if (key < skv) {
                if (row < d) val = kv[((int64_t)b * skv + key) * ld_kv + c + hd * d + row];
                else if (row == d) val = (bf16_t)0x3f80;
            }
            vt[t] = val;
        }
    }
}

int fill_lin(XLin& L, const leco_xlin& a, int n, int k, const char* what) {
    if (!a.w) return fail(-EINVAL, "%s: null weight", what);
    if (a.ldw % 8 || (a.dn && (a.ld_dn % 8 || a.ld_up % 8))) return fail(-EINVAL, "%s: operand strides must be multiples of 8 elements", what);
    if (a.dn && (!a.up || (a.t_rows != 16 && a.t_rows != 32))) return fail(-EINVAL, "%s: LoRA needs up and t_rows in {16, 32}", what);
    if ((int64_t)n * a.ldw * 2 >= ((int64_t)1 << 31)) return fail(-EINVAL, "%s: weight too large for a buffer descriptor", what);
    memset(&L, 0, sizeof(L));
    L.w = a.w; L.w_bytes = (unsigned)((int64_t)n * a.ldw * 2); L.ldw_b = (unsigned)(a.ldw * 2);
    if (a.packed) {
        if (a.ldw != k || n % 16 || k % 32) return fail(-EINVAL, "%s: fragment-order weights need ldw == K, N %% 16 == 0, K %% 32 == 0", what);
        L.kfrag_b = (unsigned)(k / 32) * 1024u;
    }
    L.bias = a.bias;
    if (a.dn) {
        L.dn = a.dn; L.dn_bytes = (unsigned)((int64_t)a.t_rows * a.ld_dn * 2); L.lddn_b = (unsigned)(a.ld_dn * 2);
        L.up = a.up; L.up_bytes = (unsigned)((int64_t)n * a.ld_up * 2); L.ldup_b = (unsigned)(a.ld_up * 2);
        L.tf = a.t_rows / 16;
    }
    return 0;
}

// > 64 KB of dynamic LDS needs the opt-in attribute: once per kernel (ID) AND device
template <int ID, class K>
void set_lds(K kern, int bytes) {
    static bool attr_set[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (dev_id < 0 || dev_id >= 64 || !attr_set[dev_id]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (dev_id >= 0 && dev_id < 64) attr_set[dev_id] = true;
    }
}
}  // namespace
}  // namespace leco

extern "C" int leco_xattn_prep(const void* kv, int64_t ld_kv, void* kp, void* vt, int32_t batch, int32_t heads, int32_t skv,
                               int32_t head_dim, leco_stream_t stream) {
    using namespace leco;
    if (!kv || !kp || !vt || batch <= 0 || heads <= 0) return fail(-EINVAL, "leco_xattn_prep: bad arguments");
    if (skv <= 0 || skv > XNKEY) return fail(-EINVAL, "leco_xattn_prep: skv=%d (1..%d)", skv, XNKEY);
    if (head_dim != 40 && head_dim != 64) return fail(-EINVAL, "leco_xattn_prep: head_dim %d unsupported (40, 64)", head_dim);
    const int dv = (head_dim + 15) / 16 * 16;
    const int64_t n = (int64_t)batch * heads * (XNKEY * 64 + dv * XNPOS);
    const int g = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(xattn_prep_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)kv, ld_kv, (bf16_t*)kp,
                       (bf16_t*)vt, batch, heads, skv, head_dim);
    return check_launch("leco_xattn_prep");
}

extern "C" int leco_xblock_supported(int32_t c, int32_t heads, int32_t skv, int32_t rows_per_sample) {
    if (c != 320 || heads <= 0 || c % heads) return 0;
    const int d = c / heads;
    if (d != 40 && d != 64) return 0;
    if (skv <= 0 || skv > leco::XNKEY) return 0;
    if (rows_per_sample <= 0 || rows_per_sample % leco::XBM) return 0;
    return 1;
}

extern "C" int leco_xblock_tail(const leco_xblock_tail_args* a, leco_stream_t stream) {
    using namespace leco;
    if (!a) return fail(-EINVAL, "leco_xblock_tail: null args");
    if (!leco_xblock_supported(a->c, a->heads, a->skv, a->rows_per_sample))
        return fail(-EINVAL, "leco_xblock_tail: unsupported shape c=%d heads=%d skv=%d rows_per_sample=%d", a->c, a->heads, a->skv,
                    a->rows_per_sample);
    if (a->m <= 0 || !a->attn || !a->h_in || !a->out || !a->kp || !a->vt || !a->ln2_g || !a->ln2_b || !a->ln3_g || !a->ln3_b)
        return fail(-EINVAL, "leco_xblock_tail: null operand");
    if (a->ld_attn % 8 || a->ld_h % 4 || a->ld_out % 8 || (a->res && a->ld_res % 8))
        return fail(-EINVAL, "leco_xblock_tail: activation strides must keep 16-byte alignment");
    if (a->col_stats && (a->stats_atom <= 0 || a->c % a->stats_atom || (a->stats_atom & 1)))
        return fail(-EINVAL, "leco_xblock_tail: col_stats needs an even stats_atom that divides c");
    const int C = a->c, F = 4 * C;
    XTailArgs p;
    memset(&p, 0, sizeof(p));
    int rc;
    if ((rc = fill_lin(p.lin[0], a->to_out1, C, C, "leco_xblock_tail.to_out1"))) return rc;
    if ((rc = fill_lin(p.lin[1], a->to_q2, C, C, "leco_xblock_tail.to_q2"))) return rc;
    if ((rc = fill_lin(p.lin[2], a->to_out2, C, C, "leco_xblock_tail.to_out2"))) return rc;
    if ((rc = fill_lin(p.lin[3], a->ff1, 2 * F, C, "leco_xblock_tail.ff1"))) return rc;
    if ((rc = fill_lin(p.lin[4], a->ff2, C, F, "leco_xblock_tail.ff2"))) return rc;
    const bool has_po = a->proj_out.w != nullptr;
    if (has_po && (rc = fill_lin(p.lin[5], a->proj_out, C, C, "leco_xblock_tail.proj_out"))) return rc;
    p.m = a->m; p.heads = a->heads; p.skv = a->skv; p.rows_per_sample = a->rows_per_sample;
    if (a->src_rows && (a->src_rows % XBM || a->m % a->src_rows)) return fail(-EINVAL, "leco_xblock_tail: src_rows=%d must be a multiple of 64 dividing m=%d", a->src_rows, a->m);
    p.src_rows = a->src_rows ? a->src_rows : (a->m + XBM - 1) / XBM * XBM;
    p.attn = (const bf16_t*)a->attn; p.ld_attn = a->ld_attn;
    p.h_in = (const bf16_t*)a->h_in; p.ld_h = a->ld_h;
    p.ln2_g = a->ln2_g; p.ln2_b = a->ln2_b; p.ln3_g = a->ln3_g; p.ln3_b = a->ln3_b; p.ln_eps = a->ln_eps;
    p.kp = (const bf16_t*)a->kp; p.vt = (const bf16_t*)a->vt; p.scale_log2 = a->attn_scale * 1.4426950408889634f;
    p.res = has_po ? (const bf16_t*)a->res : nullptr; p.ld_res = a->ld_res;
    p.out = (bf16_t*)a->out; p.ld_out = a->ld_out;
    p.col_stats = a->col_stats; p.stats_atom = a->stats_atom;
    p.has_po = has_po ? 1 : 0;
    const dim3 grid((unsigned)cdiv(a->m, XBM));
    constexpr int lds_bytes = XCfg<320>::LDS_BYTES;
    if (C / a->heads == 40) {
        set_lds<0>(&xblock_tail_kernel<320, 40>, lds_bytes);
        hipLaunchKernelGGL((xblock_tail_kernel<320, 40>), grid, dim3(512), lds_bytes, (hipStream_t)stream, p);
    } else {
        set_lds<1>(&xblock_tail_kernel<320, 64>, lds_bytes);
        hipLaunchKernelGGL((xblock_tail_kernel<320, 64>), grid, dim3(512), lds_bytes, (hipStream_t)stream, p);
    }
    return check_launch("leco_xblock_tail");
}

extern "C" int leco_xblock_head(const leco_xblock_head_args* a, leco_stream_t stream) {
    using namespace leco;
    if (!a) return fail(-EINVAL, "leco_xblock_head: null args");
    if (a->c != 320 || a->rows_per_sample <= 0 || a->rows_per_sample % XBM)
        return fail(-EINVAL, "leco_xblock_head: unsupported shape c=%d rows_per_sample=%d", a->c, a->rows_per_sample);
    if (a->m <= 0 || !a->x || !a->h_out || !a->qkv_out || !a->ln1_g || !a->ln1_b) return fail(-EINVAL, "leco_xblock_head: null operand");
    if (a->ld_x % 8 || a->ld_hout % 4 || a->ld_qkv % 4) return fail(-EINVAL, "leco_xblock_head: activation strides must keep 8-byte alignment");
    if (a->gn_cstats) {
        if (!a->gn_g || !a->gn_b || a->groups <= 0 || a->c % a->groups || a->stats_atom <= 0 || (a->c / a->groups) % a->stats_atom)
            return fail(-EINVAL, "leco_xblock_head: GroupNorm needs gamma / beta, groups | c and stats_atom | c / groups");
    }
    const int C = a->c;
    XHeadArgs p;
    memset(&p, 0, sizeof(p));
    int rc;
    if ((rc = fill_lin(p.lin[0], a->proj_in, C, C, "leco_xblock_head.proj_in"))) return rc;
    if ((rc = fill_lin(p.lin[1], a->qkv, 3 * C, C, "leco_xblock_head.qkv"))) return rc;
    p.m = a->m; p.rows_per_sample = a->rows_per_sample;
    p.x = (const bf16_t*)a->x; p.ld_x = a->ld_x;
    p.gn_cstats = a->gn_cstats; p.stats_atom = a->stats_atom; p.groups = a->groups; p.gn_g = a->gn_g; p.gn_b = a->gn_b; p.gn_eps = a->gn_eps;
    p.ln1_g = a->ln1_g; p.ln1_b = a->ln1_b; p.ln_eps = a->ln_eps;
    p.h_out = (bf16_t*)a->h_out; p.ld_hout = a->ld_hout;
    p.qkv_out = (bf16_t*)a->qkv_out; p.ld_qkv = a->ld_qkv;
    constexpr int lds_bytes = XCfg<320>::LDS_BYTES;
    set_lds<2>(&xblock_head_kernel<320>, lds_bytes);
    hipLaunchKernelGGL((xblock_head_kernel<320>), dim3((unsigned)cdiv(a->m, XBM)), dim3(512), lds_bytes, (hipStream_t)stream, p);
    return check_launch("leco_xblock_head");
}

#ifdef LECO_STRIPE_TIMING
// side builds only (tools/ablate_stripe.py): the phase stamps of workgroup 0 of the last tail launch
extern "C" int leco_xblock_debug_times(unsigned long long* out, int n) {
    if (n > 32) n = 32;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(leco::g_xtimes), sizeof(unsigned long long) * n);
}
#endif
