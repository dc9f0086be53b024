// bf16 MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (wave64, v_mfma_f32_16x16x32_bf16).
//
//   C[M][N] = epilogue( A[M][K] * W[N][K]^T  (+ A_ext[M][ext_k] * W_ext[N][ext_k]^T) )
//
// One kernel family serves every contraction of the UNet pass (SURVEY.md section 2.2 K1-K3,
// K6, K11) and, with pre-transposed weights, their dgrads:
//   * A is either a row-major matrix (Linear, 1x1 conv; optionally two K-ranges from two
//     tensors = the skip-connection concat without materialising it), or gathered on the fly
//     from a channels-last image (3x3 conv stride 1 / stride 2 / on the nearest-2x upsampled
//     image / transposed stride 2).  Cin % 64 == 0 so a 64-wide K tile never straddles a tap.
//   * the rank-r LoRA product is one extra K tile (A_ext = x*down^T, W_ext = scale*up), so
//     it is accumulated in fp32 inside the same MFMA accumulator (lora.py:102-106).
//   * epilogue: + bias[n] + rowbias[sample][n] (time embedding) + residual, SiLU, bf16 store.
//
// Structure: 256 threads = 4 waves in a 2x2 grid over a BM x BN tile, BK = 64.  Operand tiles
// go global -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no ds_write): each wave
// instruction fills 8 tile rows (64 lanes x 16 B); the 16-byte-chunk XOR swizzle
// (chunk ^= row & 7, which makes the ds_read_b128 fragment reads conflict-free) is applied on
// the per-lane SOURCE address because the LDS destination of the DMA is lane-linear.  Rows
// outside the problem / conv padding read a 16-byte zero page.  The tiles form a 4-deep LDS ring
// (up to 144 KB of the 160 KB): three tile DMAs stay in flight across the barriers (counted
// s_waitcnt vmcnt, raw s_barrier) because with ~1 workgroup per CU the global->LDS latency
// (~1.5 us under load), not MFMA time, bounds a K step.
// Operands are fed "swapped" (W fragment as MFMA-A, activation fragment as MFMA-B) so each
// lane ends up with 4 consecutive n of one output row -> 8-byte stores.  Workgroup ids are
// remapped so each XCD (private L2) owns a contiguous run of tiles.  Deep-K / small-M problems
// (the 8x8 and 16x16 levels) are split over K into fp32 partial slabs + a finishing kernel.
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <hip/hip_runtime.h>
#include <leco_prims.h>

#include "common.h"

namespace leco {
namespace {

constexpr int BK = 64;

// Tuning aid (tools/ablate_gemm.py builds side libraries with -DLECO_GEMM_ABLATE=<bit mask>): 1 = no MFMA,
// 2 = no steady-state DMA, 4 = no W-operand DMA, 8 = no A-operand DMA, 16 = no steady-state fragment
// reads, 32 = no LDS swizzle, 64 = no steady-state waits / barriers, 128 = no epilogue.  Results are garbage with any bit
// set; only the timing is meaningful.  0 in the product build.
#ifndef LECO_GEMM_ABLATE
#define LECO_GEMM_ABLATE 0
#endif

__device__ const u32x4 g_zero_page[4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * BK + ((chunk ^ (row & 7)) << 3);
}

// gelu(x) = x Phi(x) (the erf form diffusers' GEGLU uses, F.gelu) with erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7,
// far below the bf16 rounding of the result): one v_rcp + one v_exp instead of libm's branchy erff, which made the fused
// GEGLU epilogue cost MORE than the K loop it follows (tools/ablate_gemm.py --plain: 48 of 86 us on the level-0 tile).
// x < 0 uses q = 1 - erf directly: no cancellation in the tail.
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.7071067811865476f;
    const float t = fast_rcp(1.f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float hq = 0.5f * x * poly * fast_exp2(-z * z * 1.4426950408889634f);   // 0.5 x (1 - erf(|x| / sqrt 2))
    return x >= 0.f ? x - hq : hq;
}

struct GemmRt {      // launch-time extras (not part of the C ABI struct)
    int tiles_n;
    int split_k;     // >1: write raw fp32 partials to ws[split][M][N], epilogue done by splitk_finish
    float* ws;
};

// NS in {2, 4}: LDS ring depth.  NWM in {2, 4}: waves along M (block = NWM x 2 waves).  With NWM = 4 a
// 128-row tile runs on 8 waves = 2 per SIMD, so one wave's LDS-read latency hides behind the other's MFMAs
// even when only one workgroup fits on a CU.
//
// TF in {0, 1, 2}: fused LoRA down-projection.  The W operand tile carries 16*TF extra rows (the stacked
// lora_down matrix t_w [16*TF][K]); T = A * t_w^T is accumulated next to the main accumulators during the
// same K sweep (the activation tile is already in LDS), rounded to bf16 and written into LDS as the A side
// of the K-extension tile -- the separate skinny T GEMM (one launch and one full re-read of A per LoRA
// site) disappears.  TF = 1: every wave computes the 16 T columns for half of its row fragments;
// TF = 2: wave column wave_n computes T columns [16 wave_n, 16 wave_n + 16).
//
template <int BM, int BN, bool CONV, int NS, int NWM, int TF = 0>
__global__ __launch_bounds__(NWM * 128) void gemm_kernel(const leco_gemm_args p, const GemmRt rt) {
    constexpr int NW = NWM * 2, NT = NW * 64;
    constexpr int WM = BM / NWM, WN = BN / 2, FM = WM / 16, FN = WN / 16;
    constexpr int BNT = BN + 16 * TF;                     // W tile rows including the fused lora_down rows
    constexpr int TFM = TF == 0 ? 0 : (TF == 1 ? FM / 2 : FM);   // T accumulator fragments per wave
    constexpr int FNT = FN + (TF ? 1 : 0);                // W-side fragments read per half step
    static_assert(!(TF && CONV), "the fused down-projection is for plain (Linear / 1x1) sites");
    static_assert(TF != 1 || FM % 2 == 0, "TF = 1 splits the row fragments of a wave in two");
    constexpr int GA = BM / 8 / NW;                       // 8-row groups staged per wave (A), exact
    constexpr int GWT = BNT / 8, GW = (GWT + NW - 1) / NW; // W groups: total / per wave (last may be absent)
    constexpr bool RAGGED = (GWT % NW) != 0;
    static_assert((BM / 8) % NW == 0, "A row groups must divide evenly over the waves");
    constexpr int TILE = (BM + BNT) * BK;      // elements per LDS buffer
    bf16_t* smem = (bf16_t*)dyn_lds();

    // XCD-aware bijective remap.  The hardware places linear workgroup id b on XCD b % 8 (each XCD has a
    // private 4 MB L2).  Give each XCD a contiguous run of the logical work list, ordered split-major and,
    // inside a K split, so that the LARGER operand is the one an XCD keeps to itself: for N > M (the deep
    // 16x16 / 8x8 levels: weights >> activations) tiles are walked m-fastest, i.e. an XCD owns a slice of
    // W's rows and streams it once instead of every XCD streaming all of W; otherwise n-fastest (an XCD
    // owns a slice of the activations).
    const int tiles_m_ = (int)gridDim.x / rt.tiles_n;
    const int tiles = (int)gridDim.x;
    const int nwg = tiles * (int)gridDim.y, bid = (int)blockIdx.x + (int)blockIdx.y * tiles;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int split = wg / tiles, t_in = wg - split * tiles;
    int tile_m, tile_n;
    if (p.n > p.m) { tile_n = t_in / tiles_m_; tile_m = t_in - tile_n * tiles_m_; }
    else { tile_m = t_in / rt.tiles_n; tile_n = t_in - tile_m * rt.tiles_n; }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = (int)threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    // DMA instructions this wave issues per tile (ragged: the last W group exists only for the low waves)
    const bool w_last = !RAGGED || (wave + NW * (GW - 1) < GWT);
    const int st_row = lane >> 3;           // row inside an 8-row group this lane stages
    const int st_pos = lane & 7;            // 16-byte slot it fills

    const bf16_t* a0 = (const bf16_t*)p.a0;
    const bf16_t* a1 = (const bf16_t*)p.a1;
    const bf16_t* wp = (const bf16_t*)p.w;
    const bf16_t* aext = (const bf16_t*)p.a_ext;
    const bf16_t* wext = (const bf16_t*)p.w_ext;
    const bf16_t* zero = (const bf16_t*)g_zero_page;
    const int M = p.m, N = p.n;
    const int nk_main = p.k / BK;
    const bool has_ext = TF == 0 && aext != nullptr && p.ext_k > 0;
    // K range of this split (the LoRA extension tile belongs to the last split)
    const int kt_begin = (int)(((int64_t)nk_main * split) / rt.split_k);
    const int kt_end = (int)(((int64_t)nk_main * (split + 1)) / rt.split_k);
    const int nk = (kt_end - kt_begin) + ((has_ext && split == rt.split_k - 1) ? 1 : 0);
    const int k_split = a1 ? p.k_split : 0x7fffffff;

    // Per-lane addressing state, built once.  Every staged A row of this lane (GA of them) keeps
    //   plain: its row index;  conv: qy = oy*sy - 1, qx = ox*sy - 1 (tap (kh,kw) reads u = q + k, source
    //   pixel u >> dv), the sample's first pixel row b*h_in, and a 6-bit validity mask over the 3 vertical /
    //   3 horizontal taps (padding, stride-2 parity of the transposed gather, m < M) -- so the per-K-step
    //   address is a few adds / shifts, two 24-bit multiply-adds and one select against the zero page: no
    //   64-bit multiplies, no divergent branches.
    // Offsets are 32-bit element offsets (validated on the host: pixels < 2^24, elements < 2^32).
    const int cpos8 = (st_pos ^ ((LECO_GEMM_ABLATE & 32) ? 0 : st_row)) * 8;   // swizzled 16-byte slot of this lane
    int qy[GA], qx[GA], pbh[GA];
    unsigned vmask[GA];
    unsigned arow[GA];
    const int cin = CONV ? p.k / 9 : 1;
    const int dv = (CONV && (p.a_mode == LECO_A_CONV3_UP2 || p.a_mode == LECO_A_CONV3_TR2)) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int m = m0 + (wave + NW * i) * 8 + st_row;
        arow[i] = (unsigned)(m < M ? m : 0);
        vmask[i] = m < M ? 0x3fu : 0u;
        qy[i] = qx[i] = pbh[i] = 0;
        if (CONV) {
            const int hw = p.h_out * p.w_out;
            const int pb = (int)arow[i] / hw;
            const int rem = (int)arow[i] - pb * hw;
            const int py = rem / p.w_out, px = rem - py * p.w_out;
            const int sy = (p.a_mode == LECO_A_CONV3_S2) ? 2 : 1;
            const int odd_mask = (p.a_mode == LECO_A_CONV3_TR2) ? 1 : 0;
            const int lim_y = p.h_in << dv, lim_x = p.w_in << dv;
            qy[i] = py * sy - 1;
            qx[i] = px * sy - 1;
            pbh[i] = pb * p.h_in;
            unsigned vm = 0;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int uy = qy[i] + t, ux = qx[i] + t;
                const bool vy = (uy >= 0) & (uy < lim_y) & ((uy & odd_mask) == 0);
                const bool vx = (ux >= 0) & (ux < lim_x) & ((ux & odd_mask) == 0);
                vm |= (vy ? 1u : 0u) << t | (vx ? 1u : 0u) << (3 + t);
            }
            vmask[i] &= vm;
        }
    }
    // W rows of this lane: row pointers (loop invariant); rows n >= N point at the zero page and ignore the
    // K offset (mask 0)
    const bf16_t* wrow[GW];
    const bf16_t* wxrow[GW];
    unsigned wmask[GW];
#pragma unroll
    for (int i = 0; i < GW; ++i) {
        const int rl = (wave + NW * i) * 8 + st_row;      // row inside the W tile
        const int n = n0 + rl;
        const bool main_row = rl < BN && n < N;
        wrow[i] = main_row ? wp + (int64_t)n * p.ldw + cpos8 : zero;
        wmask[i] = main_row ? 0xffffffffu : 0u;
        if (TF && rl >= BN && rl < BNT) {                 // fused lora_down rows
            wrow[i] = (const bf16_t*)p.t_w + (int64_t)(rl - BN) * p.ld_tw + cpos8;
            wmask[i] = 0xffffffffu;
        }
        wxrow[i] = (main_row && cpos8 < p.ext_k && wext) ? wext + (int64_t)n * p.ld_wext + cpos8 : zero;
    }
    // select "src + off" or the zero page without a branch: zero + ((src - zero) + 2*off) & mask
    auto pick = [&](const bf16_t* src, unsigned off, unsigned ok) -> const bf16_t* {
        const long long delta = (const char*)src - (const char*)zero;           // wave-uniform
        const long long d = (delta + ((long long)off << 1)) & -(long long)ok;
        return (const bf16_t*)((const char*)zero + d);
    };
    const int n_main = kt_end - kt_begin;
    // tiles that are STAGED: with the fused down-projection the [0 | scale*up] image of the K-extension step rides in the
    // ring as one more tile (A side: zero page, overwritten by T after the K loop), so that its DMA latency is hidden
    // behind the last K steps instead of being exposed after them (~1.5 us per LoRA GEMM)
    const int nstage = nk + (TF ? 1 : 0);

    struct Addr { const bf16_t* a[GA]; const bf16_t* w[GW]; };
    // source addresses of main K tile `it` (local index): straight-line VALU/SALU, no memory traffic
    auto addr_main = [&](int it, Addr& ad) {
        const int kt = kt_begin + it;
        const int k0 = kt * BK;
        const bf16_t* src;
        unsigned ld;
        int kk, wcol = k0;
        if (!CONV) {
            if (k0 < k_split) { src = a0; ld = (unsigned)p.lda0; kk = k0; }
            else { src = a1; ld = (unsigned)p.lda1; kk = k0 - k_split; }
#pragma unroll
            for (int i = 0; i < GA; ++i) {
                const unsigned off = mul24(arow[i], ld) + (unsigned)(kk + cpos8);
                ad.a[i] = pick(src, off, vmask[i] & 1u);
            }
        } else {
            // K order of a conv is channel-chunk major / tap minor: the 9 taps of one 64-channel chunk
            // re-read (shifted) the same input pixels back to back, so they hit in L1/L2 instead of
            // sweeping the whole input patch once per tap.
            const int chunk = kt / 9, tap = kt - chunk * 9;
            const int cch = chunk * BK;
            const int kh = tap / 3, kw = tap - kh * 3;
            if (cch < k_split) { src = a0; ld = (unsigned)p.lda0; kk = cch; }
            else { src = a1; ld = (unsigned)p.lda1; kk = cch - k_split; }
            wcol = tap * cin + cch;
#pragma unroll
            for (int i = 0; i < GA; ++i) {
                const int iy = (qy[i] + kh) >> dv, ix = (qx[i] + kw) >> dv;
                const unsigned pix = mul24((unsigned)(pbh[i] + iy), (unsigned)p.w_in) + (unsigned)ix;
                const unsigned ok = (vmask[i] >> kh) & (vmask[i] >> (3 + kw)) & 1u;
                const unsigned off = mul24(pix, ld) + (unsigned)(kk + cpos8);
                ad.a[i] = pick(src, off, ok);
            }
        }
#pragma unroll
        for (int i = 0; i < GW; ++i) ad.w[i] = wrow[i] + ((unsigned)wcol & wmask[i]);
    };
    auto addr_ext = [&](Addr& ad) {   // LoRA K-extension tile
#pragma unroll
        for (int i = 0; i < GA; ++i)
            ad.a[i] = (!TF && cpos8 < p.ext_k && vmask[i]) ? aext + (int64_t)arow[i] * p.ld_aext + cpos8 : zero;
#pragma unroll
        for (int i = 0; i < GW; ++i) ad.w[i] = wxrow[i];
    };
    auto issue = [&](const Addr& ad, int buf, bool wl) {
        bf16_t* sA = smem + buf * TILE;
        bf16_t* sB = sA + BM * BK;
#pragma unroll
        for (int i = 0; i < GA; ++i)
            if (!(LECO_GEMM_ABLATE & 8)) glds16(ad.a[i], sA + (wave + NW * i) * 8 * BK);
#pragma unroll
        for (int i = 0; i < GW; ++i) {
            if (RAGGED && i == GW - 1 && !wl) break;
            if (!(LECO_GEMM_ABLATE & 4)) glds16(ad.w[i], sB + (wave + NW * i) * 8 * BK);
        }
    };
    auto stage = [&](int it, int buf) {
        Addr ad;
        if (it < n_main) addr_main(it, ad);
        else addr_ext(ad);
        issue(ad, buf, w_last);
    };

    f32x4 acc[FM][FN];
    f32x4 acct[TFM ? TFM : 1];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (TFM ? TFM : 1); ++i) acct[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int t_i0 = TF == 1 ? wave_n * (FM / 2) : 0;          // first row fragment this wave projects
    const int t_row = BN + (TF == 2 ? wave_n * 16 : 0);        // W-tile row of its lora_down fragment

    // NS-deep DMA ring + software-pipelined fragment reads.  A K tile is consumed in two 32-wide half
    // steps; the ds_read_b128 of half step h+1 are issued BEFORE the MFMAs of half step h (two fragment
    // register sets), so LDS latency overlaps MFMA issue inside a wave.  Per iteration `it`:
    //     read frags(it, ks1) ; addresses of tile it+NS  | MFMA frags(it, ks0)
    //     wait "tile it+1 landed" (counted vmcnt: this wave's pieces) ; barrier
    //         -> every wave has now finished reading tile `it` (both halves are in registers), so its ring
    //            slot can be re-filled; the barrier does not drain the younger DMAs
    //     issue DMA(tile it+NS) into that slot ; read frags(it+1, ks0) | MFMA frags(it, ks1)
    // The steady-state body (tile it+NS is a main tile) is one straight-line basic block -- no branches,
    // waitcnt immediates fixed per instantiation -- so the address arithmetic and the DMA issue interleave
    // with the MFMAs instead of forming a separate phase after the barrier.
    constexpr int LDS_RING = NS * (BM + BN + 16 * TF) * BK * 2;     // bytes of the ring = what the epilogue staging may reuse
#pragma unroll
    for (int s0 = 0; s0 < NS; ++s0)
        if (s0 < nstage) stage(s0, s0);

    const int fr = lane & 15, fg = lane >> 4;
    bf16x8 afA[FM], wfA[FNT], afB[FM], wfB[FNT];
    auto read_frags = [&](int it, int ks, bf16x8 (&af)[FM], bf16x8 (&wf)[FNT]) {
        const bf16_t* sA = smem + (it % NS) * TILE;
        const bf16_t* sB = sA + BM * BK;
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = lds_read16_async(sA + lds_off(wave_m * WM + i * 16 + fr, ks * 4 + fg));
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[j] = lds_read16_async(sB + lds_off(wave_n * WN + j * 16 + fr, ks * 4 + fg));
        if constexpr (TF != 0) wf[FN] = lds_read16_async(sB + lds_off(t_row + fr, ks * 4 + fg));
    };
    // fragment reads are explicit asynchronous ds_reads: `landed` hands a set back to the compiler once a
    // wait has covered it
    auto landed = [&](bf16x8 (&af)[FM], bf16x8 (&wf)[FNT]) {
#pragma unroll
        for (int i = 0; i < FM; ++i) lds_tie(af[i]);
#pragma unroll
        for (int j = 0; j < FNT; ++j) lds_tie(wf[j]);
    };
    auto mma = [&](const bf16x8 (&af)[FM], const bf16x8 (&wf)[FNT]) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if (!(LECO_GEMM_ABLATE & 1)) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
                else acc[i][j][0] += __uint_as_float((unsigned)(wf[j][0] ^ af[i][0]));  // keep the LDS reads live
            }
        if constexpr (TF != 0) {
#pragma unroll
            for (int i = 0; i < TFM; ++i) {
                // TF = 1: wave column 1 projects the upper half of the row fragments (uniform select, not a
                // dynamically indexed register array)
                const bf16x8 a = (TF == 1 && wave_n) ? af[(TF == 1 ? FM / 2 : 0) + i] : af[i];
                acct[i] = mfma16(wf[FN], a, acct[i]);
            }
        }
    };
    // wait until this wave's DMA pieces of tile `t` have landed, given that tiles [0, staged) were issued:
    // the `staged - 1 - t` younger tiles may stay in flight
    auto wait_tile = [&](int t, int staged) {
        const int younger = staged - 1 - t;
        if (NS >= 4 && younger >= 3) { if (w_last) wait_vmcnt<3 * (GA + GW)>(); else wait_vmcnt<3 * (GA + GW - 1)>(); }
        else if (NS >= 3 && younger == 2) { if (w_last) wait_vmcnt<2 * (GA + GW)>(); else wait_vmcnt<2 * (GA + GW - 1)>(); }
        else if (NS >= 2 && younger == 1) { if (w_last) wait_vmcnt<GA + GW>(); else wait_vmcnt<GA + GW - 1>(); }
        else wait_vmcnt<0>();
    };
    // INVARIANT (tools/audit_async_lds.py proves it on the ISA): an asynchronous fragment read is in flight only
    // inside ONE loop -- never across a loop entry or exit.  The compiler treats the asm ds_read's destination as
    // written at the end of the statement, so where two loops keep a fragment set in different registers it copies
    // the registers on the connecting edge, i.e. BEFORE the data has landed (round-1 full-size NaN: the w_last
    // steady loop of the 64x64 TF=1 kernel vs its drain loop).  Completing the set on both sides of every loop
    // costs one LDS latency per tile.
    if (nk > 0) {
        wait_tile(0, nstage < NS ? nstage : NS);
        barrier_keep_dma();
        read_frags(0, 0, afA, wfA);
        lds_wait<0>();
        landed(afA, wfA);
    }
    int it = 0;
    // steady state: iterations whose refill (tile it+NS) is a main tile.  Outstanding LDS reads when set A is
    // needed: the FM+FN reads of set B issued after it -> lgkmcnt(FM+FN) completes A without waiting for B.
    auto steady = [&](auto wl_c) {
        constexpr bool WL = decltype(wl_c)::value;
        constexpr int PIECES = GA + (WL ? GW : GW - 1);
        for (; it + NS < n_main; ++it) {
            Addr ad;
            if (!(LECO_GEMM_ABLATE & 16) || it == 0) read_frags(it, 1, afB, wfB);
            if (!(LECO_GEMM_ABLATE & 2)) addr_main(it + NS, ad);
            lds_wait<FM + FNT>();
            landed(afA, wfA);
            mma(afA, wfA);
            sched_fence();
            if (!(LECO_GEMM_ABLATE & 64)) {
                wait_vmcnt<(NS - 2) * PIECES>();      // tile it+1 landed; NS-2 younger tiles stay in flight
                barrier_keep_dma();                   // (also completes set B)
            } else lds_wait<0>();
            landed(afB, wfB);
            if (!(LECO_GEMM_ABLATE & 2)) issue(ad, it % NS, WL);
            if (!(LECO_GEMM_ABLATE & 16)) read_frags(it + 1, 0, afA, wfA);
            mma(afB, wfB);
            sched_fence();
        }
        lds_wait<0>();          // set A of tile `it` complete before the loop exit (see INVARIANT above)
        landed(afA, wfA);
    };
    if (RAGGED && !w_last) steady(std::false_type{});
    else steady(std::true_type{});
    for (; it < nk; ++it) {   // drain: last NS tiles (and the LoRA extension tile)
        read_frags(it, 1, afB, wfB);
        lds_wait<FM + FNT>();
        landed(afA, wfA);
        mma(afA, wfA);
        sched_fence();
        if (it + 1 < nk) wait_tile(it + 1, nstage < it + NS ? nstage : it + NS);
        barrier_keep_dma();
        landed(afB, wfB);
        if (it + NS < nstage && !(LECO_GEMM_ABLATE & 2)) stage(it + NS, it % NS);
        if (it + 1 < nk) read_frags(it + 1, 0, afA, wfA);
        mma(afB, wfB);
        sched_fence();
    }
    lds_wait<0>();

    if (TF) {
        // ---- fused K-extension: T (this wave's fragments, fp32) -> bf16 into the A side of the extension tile's ring
        // slot (its scale*up rows were DMA'd during the last K steps), one 32-wide MFMA step on the main accumulators.
        // Lane l holds T[row 16 i + (l & 15)][column 4 (l >> 4) + r] of its fragment (the swapped-operand D layout).
        wait_vmcnt<0>();                          // this wave's pieces of the extension tile have landed
        barrier_keep_dma();                       // ... everyone's; and every wave is done with the K tiles
        bf16_t* sA = smem + (n_main % NS) * TILE;   // ring slot of the extension tile: sB already holds scale*up
        bf16_t* sB = sA + BM * BK;
        bf16_t* tout = (bf16_t*)p.t_out;
        const int tcol = (TF == 2 ? wave_n * 16 : 0) + 4 * fg;
#pragma unroll
        for (int i = 0; i < TFM; ++i) {
            const int row = wave_m * WM + (t_i0 + i) * 16 + fr;
            const unsigned lo = pack_bf2(acct[i][0], acct[i][1]), hi = pack_bf2(acct[i][2], acct[i][3]);
            unsigned* d = (unsigned*)(sA + lds_off(row, tcol >> 3) + (tcol & 7));
            d[0] = lo;
            d[1] = hi;
            if (TF == 1) {   // columns 16..31 of the 32-wide step: explicit zeros (up_p is zero there too)
                unsigned* z = (unsigned*)(sA + lds_off(row, (tcol + 16) >> 3) + (tcol & 7));
                z[0] = 0u;
                z[1] = 0u;
            }
            if (tout && tile_n == 0 && m0 + row < M) {
                unsigned* g = (unsigned*)(tout + (int64_t)(m0 + row) * p.ld_tout + tcol);
                g[0] = lo;
                g[1] = hi;
                if (TF == 1) { g[8] = 0u; g[9] = 0u; }   // columns 16..31 of the [M][32] image
            }
        }
        barrier_keep_dma();                       // T is in LDS (ds_write: lgkmcnt); the t_out stores may still be in flight
        bf16x8 af[FM], wf[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8*)(sA + lds_off(wave_m * WM + i * 16 + fr, fg));
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[j] = *(const bf16x8*)(sB + lds_off(wave_n * WN + j * 16 + fr, fg));
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
    }

    // ---- epilogue through LDS: the accumulators (lane = one row x 4 consecutive n) are staged as fp32 -- the whole tile at
    // once where the ring's LDS holds it (RR rows per round, else 64) -- so that the global side of the epilogue (residual /
    // bias reads, bf16 or fp32-partial stores) moves whole 16..32-byte row segments per lane with full-line coalescing.
    // Every thread owns ITEMS (row, 8-column) items of a round; their residual loads are all issued before the first is
    // used: one exposed global latency per round instead of one per item (the short-K projections spend a third of their
    // time here).
    constexpr int SROW = BN + 4;                 // padded fp32 row (bank spread for the f32x4 writes)
    constexpr int NC8 = BN / 8;
    constexpr int LDSB = LDS_RING;
#ifdef LECO_GEMM_EPI64      // A/B aid (tools/_ablate builds): the round-2 form, 64 rows per round
    constexpr int RR = 64;
#else
    constexpr int RR = ((BM * SROW + 2 * BN) * 4 <= LDSB && (BM * NC8) % NT == 0) ? BM : 64;
#endif
    constexpr bool EVEN = (RR * NC8) % NT == 0;
    constexpr int ITEMS = (RR * NC8 + NT - 1) / NT;
    float* stg = (float*)dyn_lds();
    bf16_t* cp = (bf16_t*)p.c;
    const bf16_t* res = (const bf16_t*)p.residual;
    float* wsp = rt.split_k > 1 ? rt.ws + (int64_t)split * M * N : nullptr;
    if (LECO_GEMM_ABLATE & 128) {   // timing aid: no epilogue (one store per lane keeps the accumulators live)
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) v += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (v == 12345.678f && cp) cp[tid] = (bf16_t)1;
        return;
    }
#pragma unroll
    for (int h = 0; h < BM / RR; ++h) {
        barrier_keep_dma();                      // ring buffers / previous round no longer read
        if (RR == BM || (wave_m * WM) / RR == h) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int rl = (wave_m * WM) % RR + i * 16 + fr;   // row inside this round
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    *(f32x4*)(stg + rl * SROW + wave_n * WN + j * 16 + 4 * fg) = acc[i][j];
            }
        }
        barrier_keep_dma();
        if (BN == 128 && p.act == LECO_ACT_GEGLU) {
            // columns [0, 64) of the tile: value block, [64, 128): its gate block (interleaved weight rows); every
            // thread owns 8 value columns of one row and their 8 gates
            for (int e = tid; e < RR * 8; e += NT) {
                const int rl = e >> 3, cc = e & 7;
                const int m = m0 + h * RR + rl, n = n0 + cc * 8;
                if (m >= M) continue;                    // (N % 128 == 0 is validated: the tile is inside the problem)
                const float* sr = stg + rl * SROW + cc * 8;
                const f32x4 v0 = *(const f32x4*)sr, v1 = *(const f32x4*)(sr + 4);
                const f32x4 g0 = *(const f32x4*)(sr + 64), g1 = *(const f32x4*)(sr + 68);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                float gt[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
                if (p.bias) {
                    const f32x4 a0 = *(const f32x4*)(p.bias + n), a1 = *(const f32x4*)(p.bias + n + 4);
                    const f32x4 b0 = *(const f32x4*)(p.bias + n + 64), b1 = *(const f32x4*)(p.bias + n + 68);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] += a0[r]; v[4 + r] += a1[r]; gt[r] += b0[r]; gt[4 + r] += b1[r]; }
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] *= gelu_fast(gt[r]);
                const u32x4 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
                *(u32x4*)(cp + (int64_t)m * p.ldc + (n0 >> 1) + cc * 8) = o;
            }
            continue;
        }
        u32x4 rres[ITEMS];
        if (res && !wsp) {
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int e = tid + it * NT;
                const int rl = e / NC8, cc = e - rl * NC8;
                const int m = m0 + h * RR + rl, n = n0 + cc * 8;
                rres[it] = u32x4{0u, 0u, 0u, 0u};
                if ((EVEN || e < RR * NC8) && m < M && n < N) rres[it] = *(const u32x4*)(res + (int64_t)m * p.ldr + n);
            }
        }
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int e = tid + it * NT;
            if (!EVEN && e >= RR * NC8) break;
            const int rl = e / NC8, cc = e - rl * NC8;
            const int m = m0 + h * RR + rl, n = n0 + cc * 8;
            if (m >= M || n >= N) continue;
            const f32x4 v0 = *(const f32x4*)(stg + rl * SROW + cc * 8);
            const f32x4 v1 = *(const f32x4*)(stg + rl * SROW + cc * 8 + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (wsp) {   // split-K: raw partial sums, the epilogue runs in splitk_finish_kernel
                *(f32x4*)(wsp + (int64_t)m * N + n) = v0;
                *(f32x4*)(wsp + (int64_t)m * N + n + 4) = v1;
                continue;
            }
            if (p.bias) {
                const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] += b0[r]; v[4 + r] += b1[r]; }
            }
            if (p.rowbias) {
                const float* rb = p.rowbias + (int64_t)(m / p.rows_per_group) * p.ld_rowbias + n;
                const f32x4 b0 = *(const f32x4*)rb, b1 = *(const f32x4*)(rb + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] += b0[r]; v[4 + r] += b1[r]; }
            }
            if (res) {
                const u32x4 rr = rres[it];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[2 * r] += bf2f((bf16_t)(rr[r] & 0xffffu));
                    v[2 * r + 1] += bf2f((bf16_t)(rr[r] >> 16));
                }
            }
            if (p.act == LECO_ACT_SILU) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = v[r] / (1.f + __expf(-v[r]));
            }
            if (cp) {
                const u32x4 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
                *(u32x4*)(cp + (int64_t)m * p.ldc + n) = o;
                if (p.col_stats) {   // the values as stored (bf16-rounded) go back to the staging tile for the column sums
                    float* sr = stg + rl * SROW + cc * 8;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sr[2 * r] = bf2f((bf16_t)(o[r] & 0xffffu));
                        sr[2 * r + 1] = bf2f((bf16_t)(o[r] >> 16));
                    }
                }
            }
            if (p.c_f32) {
                const f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                *(f32x4*)(p.c_f32 + (int64_t)m * p.ldc32 + n) = o0;
                *(f32x4*)(p.c_f32 + (int64_t)m * p.ldc32 + n + 4) = o1;
            }
        }
        if (p.col_stats && !wsp) {
            // GroupNorm statistics of the tensor this GEMM produces (leco_hip.h): {sum, sumsq} per sample and ATOM of
            // stats_atom columns.  One thread per column walks down the staged tile (bank-conflict free); the column sums
            // meet in LDS and one thread per atom sends ONE pair of fp32 atomics (a tile whose rows belong to several samples
            // -- the small levels -- sends one pair per column and sample instead).
            barrier_keep_dma();
            const int A = p.stats_atom, NA = N / A;
            const int mlo = m0 + h * RR, mhi = (mlo + RR < M ? mlo + RR : M);
            float* csum = stg + RR * SROW;                      // [BN][2], behind the staging rows
            const bool single = mhi > mlo && mlo / p.stats_rows == (mhi - 1) / p.stats_rows;
            for (int col = tid; col < BN; col += NT) {
                const int n = n0 + col;
                float s1 = 0.f, s2 = 0.f;
                int bcur = -1;
                for (int rl = 0; rl < mhi - mlo; ++rl) {
                    const float x = stg[rl * SROW + col];
                    if (!single) {
                        const int b = (mlo + rl) / p.stats_rows;
                        if (b != bcur) {
                            if (bcur >= 0 && n < N) {
                                atomicAdd(p.col_stats + ((int64_t)bcur * NA + n / A) * 2, s1);
                                atomicAdd(p.col_stats + ((int64_t)bcur * NA + n / A) * 2 + 1, s2);
                            }
                            bcur = b; s1 = 0.f; s2 = 0.f;
                        }
                    }
                    s1 += x;
                    s2 += x * x;
                }
                if (single) {
                    csum[2 * col] = s1;
                    csum[2 * col + 1] = s2;
                } else if (bcur >= 0 && n < N) {
                    atomicAdd(p.col_stats + ((int64_t)bcur * NA + n / A) * 2, s1);
                    atomicAdd(p.col_stats + ((int64_t)bcur * NA + n / A) * 2 + 1, s2);
                }
            }
            if (single) {
                barrier_keep_dma();
                const int b = mlo / p.stats_rows, nhi = (n0 + BN < N ? n0 + BN : N);
                const int a0 = n0 / A, a1 = (nhi - 1) / A;        // atoms this tile's columns touch (edge atoms partially)
                for (int a = a0 + tid; a <= a1; a += NT) {
                    const int c0 = a * A > n0 ? a * A : n0, c1 = (a + 1) * A < nhi ? (a + 1) * A : nhi;
                    float s1 = 0.f, s2 = 0.f;
                    for (int c = c0; c < c1; ++c) { s1 += csum[2 * (c - n0)]; s2 += csum[2 * (c - n0) + 1]; }
                    atomicAdd(p.col_stats + ((int64_t)b * NA + a) * 2, s1);
                    atomicAdd(p.col_stats + ((int64_t)b * NA + a) * 2 + 1, s2);
                }
            }
        }
    }
}

// sums the split-K partial slabs and applies the epilogue (4 consecutive n per thread)
__global__ __launch_bounds__(256) void splitk_finish_kernel(const leco_gemm_args p, const float* ws, int splits) {
    const int M = p.m, N = p.n, nq = N / 4;
    const int64_t total = (int64_t)M * nq;
    bf16_t* cp = (bf16_t*)p.c;
    const bf16_t* res = (const bf16_t*)p.residual;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int m = (int)(e / nq), n = (int)(e - (int64_t)m * nq) * 4;
        f32x4 a = *(const f32x4*)(ws + (int64_t)m * N + n);
        for (int s = 1; s < splits; ++s) {
            f32x4 b = *(const f32x4*)(ws + ((int64_t)s * M + m) * N + n);
            a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
        }
        float v[4] = {a[0], a[1], a[2], a[3]};
        if (p.bias) {
            f32x4 b = *(const f32x4*)(p.bias + n);
            v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
        }
        if (p.rowbias) {
            f32x4 b = *(const f32x4*)(p.rowbias + (int64_t)(m / p.rows_per_group) * p.ld_rowbias + n);
            v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
        }
        if (res) {
            u32x2 rr = *(const u32x2*)(res + (int64_t)m * p.ldr + n);
            v[0] += bf2f((bf16_t)(rr[0] & 0xffffu)); v[1] += bf2f((bf16_t)(rr[0] >> 16));
            v[2] += bf2f((bf16_t)(rr[1] & 0xffffu)); v[3] += bf2f((bf16_t)(rr[1] >> 16));
        }
        if (p.act == LECO_ACT_SILU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.f + __expf(-v[r]));
        }
        if (cp) {
            u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            *(u32x2*)(cp + (int64_t)m * p.ldc + n) = o;
        }
        if (p.c_f32) {
            f32x4 o = {v[0], v[1], v[2], v[3]};
            *(f32x4*)(p.c_f32 + (int64_t)m * p.ldc32 + n) = o;
        }
    }
}

// Split-K finish that also leaves the GroupNorm statistics of its output (leco_gemm_args.col_stats): one block = a 64-row x
// 64-column tile, thread (row lane t / 16, 4 columns t % 16) over rows lane, lane + 16, ...; per-column {sum, sumsq} of the
// bf16-rounded values are combined over the 16 row lanes in LDS and leave as one pair of fp32 atomics per atom (tiles that
// span samples: one pair per column, row lane and sample).
__global__ __launch_bounds__(256) void splitk_finish_stats_kernel(const leco_gemm_args p, const float* ws, int splits) {
    __shared__ float red[16 * 64 * 2];
    const int M = p.m, N = p.n, A = p.stats_atom, NA = N / A;
    const int tid = (int)threadIdx.x, rlane = tid >> 4, c4 = tid & 15;
    const int m0 = (int)blockIdx.y * 64, n = (int)blockIdx.x * 64 + c4 * 4;
    const int mhi = m0 + 64 < M ? m0 + 64 : M;
    const bool single = m0 / p.stats_rows == (mhi - 1) / p.stats_rows;
    bf16_t* cp = (bf16_t*)p.c;
    const bf16_t* res = (const bf16_t*)p.residual;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    int bcur = -1;
    auto flush = [&](int b) {
        for (int r = 0; r < 4; ++r)
            if (n + r < N) {
                atomicAdd(p.col_stats + ((int64_t)b * NA + (n + r) / A) * 2, s1[r]);
                atomicAdd(p.col_stats + ((int64_t)b * NA + (n + r) / A) * 2 + 1, s2[r]);
            }
    };
    if (n < N) {
        for (int m = m0 + rlane; m < mhi; m += 16) {
            f32x4 a = *(const f32x4*)(ws + (int64_t)m * N + n);
            for (int sp = 1; sp < splits; ++sp) {
                const f32x4 b = *(const f32x4*)(ws + ((int64_t)sp * M + m) * N + n);
                a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
            }
            float v[4] = {a[0], a[1], a[2], a[3]};
            if (p.bias) {
                const f32x4 b = *(const f32x4*)(p.bias + n);
                v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
            }
            if (p.rowbias) {
                const f32x4 b = *(const f32x4*)(p.rowbias + (int64_t)(m / p.rows_per_group) * p.ld_rowbias + n);
                v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
            }
            if (res) {
                const u32x2 rr = *(const u32x2*)(res + (int64_t)m * p.ldr + n);
                v[0] += bf2f((bf16_t)(rr[0] & 0xffffu)); v[1] += bf2f((bf16_t)(rr[0] >> 16));
                v[2] += bf2f((bf16_t)(rr[1] & 0xffffu)); v[3] += bf2f((bf16_t)(rr[1] >> 16));
            }
            if (p.act == LECO_ACT_SILU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.f + __expf(-v[r]));
            }
            const u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            *(u32x2*)(cp + (int64_t)m * p.ldc + n) = o;
            if (p.c_f32) {
                const f32x4 of = {v[0], v[1], v[2], v[3]};
                *(f32x4*)(p.c_f32 + (int64_t)m * p.ldc32 + n) = of;
            }
            if (!single) {
                const int b = m / p.stats_rows;
                if (b != bcur) {
                    if (bcur >= 0) flush(bcur);
                    bcur = b;
                    for (int r = 0; r < 4; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
                }
            }
            const float q[4] = {bf2f((bf16_t)(o[0] & 0xffffu)), bf2f((bf16_t)(o[0] >> 16)), bf2f((bf16_t)(o[1] & 0xffffu)),
                                bf2f((bf16_t)(o[1] >> 16))};
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[r] += q[r]; s2[r] += q[r] * q[r]; }
        }
        if (!single && bcur >= 0) flush(bcur);
    }
    if (!single) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[(rlane * 64 + c4 * 4 + r) * 2] = s1[r];
        red[(rlane * 64 + c4 * 4 + r) * 2 + 1] = s2[r];
    }
    __syncthreads();
    const int n0 = (int)blockIdx.x * 64, nhi = n0 + 64 < N ? n0 + 64 : N, b = m0 / p.stats_rows;
    const int a0 = n0 / A, a1 = (nhi - 1) / A;
    for (int a = a0 + tid; a <= a1; a += 256) {
        const int c0 = a * A > n0 ? a * A : n0, c1 = (a + 1) * A < nhi ? (a + 1) * A : nhi;
        float t1 = 0.f, t2 = 0.f;
        for (int c = c0; c < c1; ++c)
#pragma unroll
            for (int l = 0; l < 16; ++l) { t1 += red[(l * 64 + c - n0) * 2]; t2 += red[(l * 64 + c - n0) * 2 + 1]; }
        atomicAdd(p.col_stats + ((int64_t)b * NA + a) * 2, t1);
        atomicAdd(p.col_stats + ((int64_t)b * NA + a) * 2 + 1, t2);
    }
}

// leco_gemm_describe: when set, launch_k records the instantiation it WOULD launch (the name rocprofv3 prints) and the
// grid instead of launching -- measurement tools attribute plan launches to profile rows with it
thread_local char* tl_describe = nullptr;
thread_local int tl_describe_len = 0;

template <int BM, int BN, bool CONV, int NS, int NWM, int TF>
void launch_k(const leco_gemm_args& a, const GemmRt& rt, dim3 grid, hipStream_t s) {
    constexpr int lds_bytes = NS * (BM + BN + 16 * TF) * BK * (int)sizeof(bf16_t);
    static_assert(lds_bytes <= 160 * 1024, "LDS ring does not fit");
    if (tl_describe) {
        const int used = (int)strlen(tl_describe);
        snprintf(tl_describe + used, tl_describe_len - used, "%sgemm_kernel<%d, %d, %s, %d, %d, %d> grid=%u split=%d",
                 used ? " ; " : "", BM, BN, CONV ? "true" : "false", NS, NWM, TF, grid.x, rt.split_k);
        return;
    }
    // > 64 KB of dynamic LDS needs the opt-in attribute: once per instantiation AND device (the attribute lives on
    // the device's copy of the function; one process per GPU is the rule, but nothing here may depend on it)
    static bool attr_set[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (dev_id < 0 || dev_id >= 64 || !attr_set[dev_id]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, CONV, NS, NWM, TF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (dev_id >= 0 && dev_id < 64) attr_set[dev_id] = true;
    }
    hipLaunchKernelGGL((gemm_kernel<BM, BN, CONV, NS, NWM, TF>), grid, dim3(NWM * 128), lds_bytes, s, a, rt);
}
// fused LoRA down-projection variants exist for plain operands only; the ring loses one slot where the
// extra lora_down rows would not fit in 160 KB
template <int BM, int BN, bool CONV, int NS, int NWM>
void launch_ns(const leco_gemm_args& a, const GemmRt& rt, dim3 grid, hipStream_t s) {
    if constexpr (!CONV) {
        constexpr int NS2 = (NS * (BM + BN + 32) * BK * 2 <= 156 * 1024) ? NS : NS - 1;
        if (a.t_w && a.t_rows == 16) return launch_k<BM, BN, CONV, NS, NWM, 1>(a, rt, grid, s);
        if (a.t_w) return launch_k<BM, BN, CONV, NS2, NWM, 2>(a, rt, grid, s);
    }
    launch_k<BM, BN, CONV, NS, NWM, 0>(a, rt, grid, s);
}

// Pipeline shape by grid size.  One 8-wave workgroup per CU with the 4-deep DMA ring (4 x 36 KB for 128x160; two
// waves per SIMD) is the default.  Plain 128x128 grids of >= 512 workgroups (the many-round short-K level-0
// projections, prologue / epilogue bound) run as 4-wave workgroups with 64x64 wave tiles and a 2-deep ring: ~70 KB of
// LDS and <= 256 VGPRs, so TWO workgroups share a CU and one's prologue / epilogue overlaps the other's K loop
// (measured on the whole step, profiles/r02_switch_sweep.txt: +1.5 %; two 8-wave 2-buffer workgroups per CU: -0.2 %,
// a persistent cross-tile-prefetching variant: -2..3.5 % -- both removed).  LECO_GEMM_W4_MIN_BLOCKS overrides the
// threshold (tools/switch_sweep.sh).
template <int BM, int BN, bool CONV>
void launch_one(const leco_gemm_args& a, const GemmRt& rt, dim3 grid, hipStream_t s, int shape) {
    if constexpr (BM == 64) {
        launch_ns<BM, BN, CONV, 4, 2>(a, rt, grid, s);
    } else {
        if constexpr (BM == 128 && BN == 128 && !CONV) {
            static const long w4_min_blocks = [] {
                const char* e = getenv("LECO_GEMM_W4_MIN_BLOCKS");
                return e ? atol(e) : 512L;
            }();
            // shape: 0 = by grid size, 1 = force the 4-wave / two-workgroups-per-CU form, 2 = force the 8-wave form
            if (shape == 1 || (shape == 0 && (long)grid.x * grid.y >= w4_min_blocks))
                return launch_ns<BM, BN, CONV, 2, 2>(a, rt, grid, s);
        }
        launch_ns<BM, BN, CONV, 4, 4>(a, rt, grid, s);
    }
}

template <int BM, int BN>
int launch(const leco_gemm_args& a, int split_k, float* ws, hipStream_t s, int shape = 0) {
    const int tm = cdiv(a.m, BM), tn = cdiv(a.n, BN);
    GemmRt rt{tn, split_k, ws};
    dim3 grid((unsigned)(tm * tn), (unsigned)split_k);
    if constexpr (BM == 256) {   // 256x128 tile: 48 KB per ring slot -> 3-deep ring (144 KB), 8 waves (4 x 2)
        if (a.a_mode == LECO_A_PLAIN) launch_ns<BM, BN, false, 3, 4>(a, rt, grid, s);
        else launch_ns<BM, BN, true, 3, 4>(a, rt, grid, s);
    } else {
        if (a.a_mode == LECO_A_PLAIN) launch_one<BM, BN, false>(a, rt, grid, s, shape);
        else launch_one<BM, BN, true>(a, rt, grid, s, shape);
    }
    if (split_k > 1 && !tl_describe) splitk_finish_launch(a, ws, split_k, s);
    if (tl_describe) return 0;
    return check_launch("leco_gemm");
}

int validate(const leco_gemm_args& a) {
    if (a.m <= 0 || a.n <= 0 || a.k <= 0) return fail(-EINVAL, "leco_gemm: empty problem m=%d n=%d k=%d", a.m, a.n, a.k);
    if (a.k % BK) return fail(-EINVAL, "leco_gemm: k=%d not a multiple of 64", a.k);
    if (a.n % 8) return fail(-EINVAL, "leco_gemm: n=%d not a multiple of 8", a.n);
    if ((a.c && a.ldc % 8) || (a.residual && a.ldr % 8) || (a.c_f32 && a.ldc32 % 4) || (a.rowbias && a.ld_rowbias % 4))
        return fail(-EINVAL, "leco_gemm: output / residual strides must keep 16-byte alignment");
    if (!a.a0 || !a.w) return fail(-EINVAL, "leco_gemm: null operand");
    if (!a.c && !a.c_f32) return fail(-EINVAL, "leco_gemm: no output");
    if (a.a_ext && a.ext_k != 32 && a.ext_k != 64) return fail(-EINVAL, "leco_gemm: ext_k=%d must be 32 or 64", a.ext_k);
    if (a.a_ext && !a.w_ext) return fail(-EINVAL, "leco_gemm: a_ext without w_ext");
    if (a.t_w) {
        if (a.a_ext) return fail(-EINVAL, "leco_gemm: t_w (fused down-projection) and a_ext are exclusive");
        if (a.a_mode != LECO_A_PLAIN) return fail(-EINVAL, "leco_gemm: t_w needs a plain A operand");
        if (!a.w_ext || a.ext_k != 32 || (a.t_rows != 16 && a.t_rows != 32) || a.ld_tw % 8 || a.ld_wext % 8 ||
            (a.t_out && a.ld_tout % 8))
            return fail(-EINVAL, "leco_gemm: t_w needs w_ext, ext_k == 32, t_rows in {16, 32}, 16-byte aligned strides");
    }
    if (a.rowbias && a.rows_per_group <= 0) return fail(-EINVAL, "leco_gemm: rowbias needs rows_per_group");
    if (a.col_stats && (!a.c || a.stats_rows <= 0 || a.stats_atom <= 0 || a.n % a.stats_atom || a.m % a.stats_rows ||
                        a.act == LECO_ACT_GEGLU))
        return fail(-EINVAL, "leco_gemm: col_stats needs a bf16 output, stats_rows | m, stats_atom | n and no fused GEGLU");
    if (a.act == LECO_ACT_GEGLU && (a.n % 128 || !a.c || a.residual || a.rowbias || a.c_f32 || a.ldc % 8))
        return fail(-EINVAL, "leco_gemm: LECO_ACT_GEGLU needs n %% 128 == 0, a bf16 output and no residual / rowbias / fp32 copy");
    if (a.a_mode < LECO_A_PLAIN || a.a_mode > LECO_A_CONV3_TR2) return fail(-EINVAL, "leco_gemm: bad a_mode %d", a.a_mode);
    if ((a.lda0 | a.ldw | (a.a1 ? a.lda1 : 0) | (a.a_ext ? (a.ld_aext | a.ld_wext) : 0)) % 8)
        return fail(-EINVAL, "leco_gemm: operand strides must be multiples of 8 elements (16-byte DMA)");
    if (a.a_mode != LECO_A_PLAIN) {
        if (a.k % 9 || (a.k / 9) % BK) return fail(-EINVAL, "leco_gemm: conv needs k = 9*Cin, Cin %% 64 == 0 (k=%d)", a.k);
        if ((int64_t)a.batch * a.h_out * a.w_out != a.m) return fail(-EINVAL, "leco_gemm: conv m != batch*h_out*w_out");
        if (a.a1 && (a.k_split % BK)) return fail(-EINVAL, "leco_gemm: conv k_split %% 64 != 0");
    } else if (a.a1 && (a.k_split % BK)) {
        return fail(-EINVAL, "leco_gemm: k_split %% 64 != 0");
    }
    {   // the A loader forms 32-bit element offsets with 24-bit multiplies (row or pixel index x row stride)
        const int64_t rows = a.a_mode == LECO_A_PLAIN ? (int64_t)a.m : (int64_t)a.batch * a.h_in * a.w_in;
        const int64_t ld = a.lda0 > (a.a1 ? a.lda1 : 0) ? a.lda0 : a.lda1;
        if (rows >= (1 << 24) || ld >= (1 << 24) || rows * ld + a.k >= ((int64_t)1 << 32))
            return fail(-EINVAL, "leco_gemm: activation operand too large for 32-bit element offsets (rows=%lld ld=%lld)",
                        (long long)rows, (long long)ld);
    }
    return 0;
}
}  // namespace

void splitk_finish_launch(const leco_gemm_args& a, const float* ws, int splits, hipStream_t s) {
    if (a.col_stats && a.c) {
        // tiled form: the same sums + epilogue, plus the GroupNorm statistics of the output it writes
        hipLaunchKernelGGL(splitk_finish_stats_kernel, dim3(cdiv(a.n, 64), cdiv(a.m, 64)), dim3(256), 0, s, a, ws, splits);
        return;
    }
    const int64_t quads = (int64_t)a.m * a.n / 4;
    const int g = (int)((quads + 255) / 256 < 2048 ? (quads + 255) / 256 : 2048);
    hipLaunchKernelGGL(splitk_finish_kernel, dim3(g), dim3(256), 0, s, a, ws, splits);
}
}  // namespace leco

// (round 6) 11 = 128x64: 160 workgroups where 64x64 tiles need 320 on 256 CUs and 128x128 tiles leave two thirds of the chip idle
// (M = 1024, N = 1280): 28 % fewer operand bytes through the busiest CU's port than the 64x64 tiling.
// tile: 0 = heuristic (-1: heuristic restricted to the implicit-GEMM kernel), 1 = 128x128 (wave shape by grid size), 2 = 128x160, 3 = 64x64, 4 = 256x128, 5 = 128x128 as
// 4-wave workgroups (two per CU), 6 = 128x128 as one 8-wave workgroup per CU; 7..10 = the patch-staged 3x3 / stride-1
// convolution (conv_patch.hip) on 256x128 / 128x160 / 128x128 / 256x160 tiles -- problems it does not cover (other
// gathers, a LoRA K-extension, a patch that does not fit) fall back to the heuristic.  split_k: 0 = heuristic (needs a
// workspace), 1 = none, >1 = that many K slices.  workspace: fp32 scratch for split-K partials.
extern "C" int leco_gemm_ex(const leco_gemm_args* args, int tile, int split_k, void* workspace,
                            int64_t workspace_bytes, leco_stream_t stream) {
    using namespace leco;
    if (!args) return fail(-EINVAL, "leco_gemm: null args");
    int rc = validate(*args);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int tile_in = tile;
    const int m = args->m, n = args->n, nk = args->k / BK;
    if (args->act == LECO_ACT_GEGLU) {   // value / gate pairing lives inside one 128-column tile
        if (tile != 0 && tile != 1 && tile != 4 && tile != 5 && tile != 6)
            return fail(-EINVAL, "leco_gemm: LECO_ACT_GEGLU needs a 128-column tile");
        if (tile == 0) tile = 1;
        split_k = 1;
    }
    static const bool no_patch = getenv("LECO_NO_CONV_PATCH") != nullptr;      // A/B switch for measurements
    if (tile == 0 && (args->a_mode == LECO_A_CONV3_S1 || args->a_mode == LECO_A_CONV3_UP2) && !no_patch) {
        // 3x3 / stride-1 convolutions (plain or on the 2x upsampled input): the patch-staged kernel (conv_patch.hip) with the
        // launch shape of its cost model
        int sp = 1;
        const int t = conv_patch_choose(*args, workspace ? workspace_bytes : 0, split_k, &sp);
        if (t) { tile = t; split_k = sp; }
    }
    if (tile <= 0) {      // (-1: the implicit-GEMM heuristic, whatever the gather)
        if (n <= 64 || m <= 64) tile = 3;
        else {
            const int bn = (n % 128 == 0) ? 128 : ((n % 160 == 0) ? 160 : 128);
            const long blocks = (long)cdiv(m, 128) * cdiv(n, bn);
            const bool can_split = workspace != nullptr && nk >= 32;
            tile = (blocks >= 128 || (can_split && m >= 128)) ? (bn == 160 ? 2 : 1) : 3;
            // deep-K, DMA-bound shapes (the 640/1280-channel convs): the 256-row tile moves ~25% fewer
            // global->LDS bytes per MFMA and halves the tile count so split-K can fill one round exactly
            if (tile == 1 && workspace != nullptr && m >= 1024 && nk >= 64) tile = 4;
        }
    }
    const int bm = tile == 3 ? 64 : ((tile == 4 || tile == 7 || tile == 10) ? 256 : 128);
    const int bn = (tile == 3 || tile == 11) ? 64 : ((tile == 2 || tile == 8 || tile == 10) ? 160 : 128);   // 1, 5, 6, 9: 128x128; 11: 128x64
    const long tiles = (long)cdiv(m, bm) * cdiv(n, bn);
    if (split_k == 0) {
        split_k = 1;
        if (workspace && tiles <= 128 && nk >= 32) {
            // one workgroup per CU in this regime: fill the 256 CUs in ONE round (no ragged second wave)
            split_k = (int)(256 / tiles);
            if (split_k > nk / 8) split_k = nk / 8;
            if (split_k > 16) split_k = 16;
        }
    }
    if (split_k > 1) {
        const int64_t need = (int64_t)split_k * m * n * 4;
        if (!workspace || need > workspace_bytes) {
            const int fit = workspace ? (int)(workspace_bytes / ((int64_t)m * n * 4)) : 1;
            split_k = fit < 1 ? 1 : (fit < split_k ? fit : split_k);
        }
        if (split_k > nk) split_k = nk;
    }
    if (split_k < 1) split_k = 1;
    if (args->t_w && split_k > 1 && tile_in == 0 && (long)cdiv(m, 64) * cdiv(n, 64) >= 192) {
        // heuristic mode: a grid of 64x64 tiles that fills the chip keeps the projection fused and needs no split
        // (measured, M=1024 N=1280 K=5120: 39 us vs 84 us for split-K + the separate skinny projection below)
        tile = 3;
        split_k = 1;
    }
    if (args->t_w && split_k > 1) {
        // the fused down-projection cannot span K slices: keep the split (these are the latency-bound deep
        // levels) and run the projection as its own skinny GEMM into t_out first
        if (!args->t_out) split_k = 1;
        else {
            leco_gemm_args t{};
            t.a0 = args->a0; t.a1 = args->a1; t.lda0 = args->lda0; t.lda1 = args->lda1; t.k_split = args->k_split;
            t.a_mode = LECO_A_PLAIN;
            t.w = args->t_w; t.ldw = args->ld_tw;
            t.m = m; t.n = 32; t.k = args->k;
            t.c = args->t_out; t.ldc = args->ld_tout;
            rc = leco_gemm_ex(&t, 0, 1, nullptr, 0, stream);
            if (rc) return rc;
            leco_gemm_args main_args = *args;
            main_args.t_w = nullptr;
            main_args.a_ext = args->t_out;
            main_args.ld_aext = args->ld_tout;
            return leco_gemm_ex(&main_args, tile, split_k, workspace, workspace_bytes, stream);
        }
    }
    if (tile >= 7 && tile <= 10) {
        rc = conv_patch_try(*args, tile, split_k, (float*)workspace, s, tl_describe, tl_describe_len);
        if (rc == 1) return leco_gemm_ex(args, -1, 0, workspace, workspace_bytes, stream);
        if (rc < 0 || tl_describe) return rc;
        // (the patch kernel splits over whole 64-channel chunks: at most nchunks slabs exist -- with one chunk the launch
        // did not split at all and has written its finished output)
        const int nchunks = args->k / 9 / BK;
        const int eff = split_k < nchunks ? split_k : nchunks;
        if (eff > 1) splitk_finish_launch(*args, (const float*)workspace, eff, s);
        return check_launch("leco_gemm");
    }
    switch (tile) {
        case 1: return launch<128, 128>(*args, split_k, (float*)workspace, s);
        case 2: return launch<128, 160>(*args, split_k, (float*)workspace, s);
        case 3: return launch<64, 64>(*args, split_k, (float*)workspace, s);
        case 4: return launch<256, 128>(*args, split_k, (float*)workspace, s);
        case 5: return launch<128, 128>(*args, split_k, (float*)workspace, s, 1);
        case 6: return launch<128, 128>(*args, split_k, (float*)workspace, s, 2);
        case 11: return launch<128, 64>(*args, split_k, (float*)workspace, s);
        default: return fail(-EINVAL, "leco_gemm: bad tile id %d", tile);
    }
}

extern "C" int leco_gemm_describe(const leco_gemm_args* args, int tile, int split_k, void* workspace,
                                  int64_t workspace_bytes, char* out, int32_t out_len) {
    if (!out || out_len < 64) return leco::fail(-EINVAL, "leco_gemm_describe: buffer too small");
    out[0] = 0;
    leco::tl_describe = out;
    leco::tl_describe_len = out_len;
    const int rc = leco_gemm_ex(args, tile, split_k, workspace, workspace_bytes, nullptr);
    leco::tl_describe = nullptr;
    return rc;
}

extern "C" int leco_gemm_tile(const leco_gemm_args* args, int tile, leco_stream_t stream) {
    return leco_gemm_ex(args, tile, 1, nullptr, 0, stream);
}

extern "C" int leco_gemm(const leco_gemm_args* args, leco_stream_t stream) {
    return leco_gemm_ex(args, 0, 1, nullptr, 0, stream);
}
