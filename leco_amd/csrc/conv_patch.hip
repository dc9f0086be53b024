// Patch-staged 3x3 / stride-1 convolution for gfx950 (wave64, v_mfma_f32_16x16x32_bf16).
//
//   C[m][n] = epilogue( sum_{tap, c} X[pixel(m) + tap][c] * W[n][tap][c] ),   m = (b, y, x), K = 9 Cin
//
// Replaces the 3x3 convolutions of diffusers' ResnetBlock2D (conv1 / conv2; call site train_util.py:156-160)
// and, with the flipped / transposed weight image, their dgrad (train_lora.py:279).  Same operands and
// epilogue as `gemm_kernel<..., CONV = true>` (gemm.hip); what differs is how the activation operand reaches
// the MFMAs.  The implicit-GEMM kernel re-DMAs a BM x 64 activation tile for each of the 9 taps (every input
// pixel travels L2 -> LDS nine times: round 2 measured that path, not the MFMAs, as the bound -- 37.5 of
// 41.7 us on the level-0 conv).  Here a workgroup owns a TH x TW block of output pixels and stages the
// (TH + halo) x (TW + 2) x 64-channel INPUT PATCH once per channel chunk; the nine taps are served from that
// one LDS image by shifted fragment reads:
//
//   * tile = TH x TW output pixels, TH rows of the "global row" space g = b * H + y (a tile may span images),
//     TW in {8, 16}; BM = TH * TW in {128, 256}; BN in {128, 160} output channels.
//   * patch rows live in a VIRTUAL row space v = g + g / H: every image is followed by one all-zero
//     separator row, which is at once the bottom halo of image b and the top halo of image b + 1.  The patch
//     is the contiguous range v0 - 1 .. v_last + 1 with TW + 2 columns (left / right halo); halo and
//     separator entries are out-of-range lanes of the buffer-descriptor DMA (zeros, no traffic).  Tap (kh, kw) of output pixel (ty, tx) is patch entry
//     (v(ty) - v0 + kh) * PW + tx + kw: one scalar offset per tap, no validity masks, no branches.
//   * a patch entry is one pixel's 64 channels = 128 B = 8 x 16-byte slots, XOR-swizzled by (entry & 7) on
//     the DMA source side (the LDS destination of an LDS-DMA is lane-linear); 16 consecutive entries
//     -- an MFMA fragment's 16 pixels of one tile row -- read conflict-free at any start offset.
//   * LDS: two patch buffers (chunk c is consumed while c + 1 lands, spread over the first taps of chunk c)
//     + an NSW-deep ring of BN x 64 weight tiles, one tile per tap step.  Per step a wave issues
//     ceil(BN / 64) weight pieces and (first APW taps only) one patch piece: at 256 x 128 that is 2.7 DMA
//     instructions per wave and step against 6 for the implicit-GEMM tile, 198 KB per chunk against 442 KB.
//   * the 9 taps of a chunk are unrolled, so every s_waitcnt immediate, tap offset and ring slot rotation is a
//     compile-time constant; DMAs past the end of the K range are issued out of range (zeros) so that the
//     counts stay uniform (no drain loop, one loop body).
//   * no asynchronous fragment read crosses the loop back edge (tools/audit_async_lds.py).
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

// Tuning aid (tools/ablate_conv.py builds a side library with -DLECO_CONV_ABLATE=<bit mask>): 1 = no steady-state fragment
// reads (the LDS -> register traffic of the tap loop), 2 = no MFMAs, 4 = no wait for the weight DMA in the tap step, 8 = no
// workgroup barrier in the tap step, 16 = no steady-state DMA.  Results are garbage with any bit set; only the timing
// is meaningful.  0 in the product build.
#ifndef LECO_CONV_ABLATE
#define LECO_CONV_ABLATE 0
#endif

struct PatchRt {
    int tiles_n, tiles_x, tiles_g;   // grid.x = tiles_g * tiles_x * tiles_n
    int split_k;                     // > 1: raw fp32 partials to ws[split][M][N], epilogue by splitk_finish
    int tw_log2;                     // TW = 8 or 16
    unsigned a0_bytes, a1_bytes, w_bytes;   // operand extents (buffer descriptors; < 2^31)
    unsigned ax_bytes, wx_bytes;            // ... of the K-extension operands a_ext / w_ext (0: none)
    float* ws;
};

template <int BM, int BN, int NSW>
struct PatchCfg {
    static constexpr int NW = 8;
    static constexpr int APW = BM == 256 ? 6 : 4;          // patch pieces (8 entries each) per wave and chunk
    static constexpr int PCAP = APW * NW * 8;              // patch capacity in entries (pixels)
    static constexpr int GWT = BN / 8, GW = (GWT + NW - 1) / NW;
    static constexpr bool RAGW = (GWT % NW) != 0;          // the last weight piece exists only for the low waves
    static constexpr int WTILE = BN * BK * 2;              // bytes per weight ring slot
    static constexpr int PBUF = PCAP * BK * 2;             // bytes per patch buffer
    static constexpr int OFF_A = NSW * WTILE, OFF_DUMP = OFF_A + 2 * PBUF;
    static constexpr int LDS_BYTES = OFF_DUMP + (RAGW ? 1024 : 0);
    static constexpr int SROW = BN + 4;                    // epilogue staging row (fp32)
    static_assert(APW <= 11 - NSW, "the next chunk's patch must be covered by the wait of tap 8");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS layout does not fit");
    static_assert(64 * SROW * 4 <= LDS_BYTES, "epilogue staging does not fit");
};

// UP2: the convolution runs on the nearest-2x UPSAMPLED input (Upsample2D.conv, LECO_A_CONV3_UP2) -- the patch then lives
// at INPUT resolution ((TH / 2 + 2) x (TW / 2 + 2) entries for a TH x TW output block: a quarter of the stride-1 patch) and
// tap (kh, kw) of output pixel (y, x) reads input pixel ((y + kh - 1) >> 1, (x + kw - 1) >> 1); -1 and H_in are the same
// zero separator rows.
template <int BM, int BN, int NSW, bool UP2>
__global__ __launch_bounds__(512) void conv_patch_kernel(const leco_gemm_args p, const PatchRt rt) {
    using Cf = PatchCfg<BM, BN, NSW>;
    constexpr int NW = Cf::NW, NT = NW * 64;
    constexpr int WM = BM / 4, WN = BN / 2, FM = WM / 16, FN = WN / 16;
    constexpr int APW = Cf::APW, GW = Cf::GW, GWT = Cf::GWT;
    constexpr bool RAGW = Cf::RAGW;
    constexpr int WTILE = Cf::WTILE, PBUF = Cf::PBUF, OFF_A = Cf::OFF_A, OFF_DUMP = Cf::OFF_DUMP;
    unsigned char* lds = dyn_lds();
#if LECO_CONV_ABLATE & 32
    unsigned long long clk[4] = {}, rtc[2] = {};     // shader-clock / 100 MHz stamps of workgroup 0
    clk[0] = __builtin_amdgcn_s_memtime(); rtc[0] = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- work decomposition: XCD-aware bijective remap (hardware places linear workgroup id b on XCD b % 8), split
    // major; inside a split the LARGER operand is what an XCD keeps to itself (N > M: m-fastest walk, an XCD owns a
    // slice of W's rows; otherwise n-fastest: an XCD owns a slice of the activations) -- as gemm.hip
    const int tiles_m = rt.tiles_g * rt.tiles_x;
    const int tiles = (int)gridDim.x;
    const int nwg = tiles * (int)gridDim.y, bid = (int)blockIdx.x + (int)blockIdx.y * tiles;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int split = wg / tiles, t_in = wg - split * tiles;
    int tile_m, tile_n;
    if (p.n > p.m) { tile_n = t_in / tiles_m; tile_m = t_in - tile_n * tiles_m; }
    else { tile_m = t_in / rt.tiles_n; tile_n = t_in - tile_m * rt.tiles_n; }
    const int tile_g = tile_m / rt.tiles_x, tile_x = tile_m - tile_g * rt.tiles_x;

    const int tid = (int)threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int st_row = lane >> 3, st_pos = lane & 7;
    const int fr = lane & 15, fg = lane >> 4;

    const int TWl = rt.tw_log2, TW = 1 << TWl, TH = BM >> TWl;
    const int H = p.h_out, W = p.w_out, GROWS = p.batch * H;
    const int g0 = tile_g * TH, x0 = tile_x * TW, n0 = tile_n * BN;
    const int g_last = (g0 + TH < GROWS ? g0 + TH : GROWS) - 1;
    // patch geometry in the INPUT image: HD x WD pixels per sample, PW columns starting at input column XLO, virtual rows
    // VLO .. VLO + PR - 1 (virtual row of input row (b, iy) = b (HD + 1) + iy; iy = -1 / HD are the zero separators)
    const int HD = UP2 ? p.h_in : H, WD = UP2 ? p.w_in : W;
    const int PW = UP2 ? (TW >> 1) + 2 : TW + 2, XLO = UP2 ? (x0 >> 1) - 1 : x0 - 1;
    const int b_first = g0 / H, b_last = g_last / H;
    const int VLO = UP2 ? b_first * (HD + 1) + ((g0 - b_first * H - 1) >> 1) : g0 + b_first - 1;
    const int VHI = UP2 ? b_last * (HD + 1) + ((g_last - b_last * H + 1) >> 1) : g_last + b_last + 1;
    const int PR = VHI - VLO + 1;
    const int M = p.m, N = p.n;
    const int cin = p.k / 9, nchunks = cin / BK;
    const int chunk_begin = (int)(((int64_t)nchunks * split) / rt.split_k);
    const int chunk_end = (int)(((int64_t)nchunks * (split + 1)) / rt.split_k);
    const int k_split = p.a1 ? p.k_split : 0x7fffffff;

    // operands through buffer descriptors: 32-bit per-lane offsets, and an out-of-range offset (DMA_OOB) reads zeros --
    // halo / separator entries, rows n >= N, the padding piece of a ragged weight tile and every DMA issued past the end
    // of the K range cost no memory traffic and need no zero page or select
    const buf_rsrc ra0 = make_rsrc(p.a0, rt.a0_bytes);
    const buf_rsrc ra1 = make_rsrc(p.a1 ? p.a1 : p.a0, p.a1 ? rt.a1_bytes : rt.a0_bytes);
    const buf_rsrc rw = make_rsrc(p.w, rt.w_bytes);
    const unsigned cpos = (unsigned)((st_pos ^ st_row) * 16);   // swizzled 16-byte slot this lane fills: entry & 7 == st_row

    // ---- patch loader state: the APW entries this lane stages per chunk: pixel index in bits 0..23 (v_mul_u32_u24
    // ignores the rest), bit 31 set = halo / separator / beyond the patch: the DMA offset gets that bit OR-ed in, which
    // puts it out of the descriptor's range.  Filled in the prologue, behind the first weight DMAs.
    unsigned ppix[APW];
    // weight rows of this lane: byte offset of (row, swizzled slot), loop invariant
    unsigned wvoff[GW];
#pragma unroll
    for (int i = 0; i < GW; ++i) {
        const int rl = (wave + NW * i) * 8 + st_row;
        const int n = n0 + rl;
        wvoff[i] = (rl < BN && n < N) ? (unsigned)n * (unsigned)p.ldw * 2u + cpos : DMA_OOB;
    }
    // patch piece j of chunk `cabs` (absolute channel chunk; chunks >= chunk_end: zeros) into patch buffer `pb`
    auto issue_a = [&](int j, int cabs, int pb) {
        const int cch = cabs * BK;
        const bool first = cch < k_split;
        const unsigned ld = first ? (unsigned)p.lda0 : (unsigned)p.lda1;
        const unsigned soff = (unsigned)(first ? cch : cch - k_split) * 2u;
        const unsigned dead = cabs < chunk_end ? 0u : DMA_OOB;
        const unsigned voff = (mul24(ppix[j], ld) * 2u + cpos) | (ppix[j] & DMA_OOB) | dead;
        glds16_buf(first ? ra0 : ra1, voff, soff, lds + OFF_A + pb * PBUF + (wave + NW * j) * (8 * BK * 2));
    };
    // weight tile of (chunk cabs, tap) into ring slot `slot`
    auto issue_w = [&](int cabs, int tap, int slot) {
        const bool live = cabs < chunk_end;
        const unsigned soff = live ? (unsigned)(tap * cin + cabs * BK) * 2u : 0u;
        const unsigned dead = live ? 0u : DMA_OOB;
#pragma unroll
        for (int i = 0; i < GW; ++i) {
            const bool real = !RAGW || i < GW - 1 || (wave + NW * i) < GWT;     // wave-uniform
            unsigned char* dst = real ? lds + slot * WTILE + (wave + NW * i) * (8 * BK * 2) : lds + OFF_DUMP;
            glds16_buf(rw, wvoff[i] | dead | (real ? 0u : DMA_OOB), soff, dst);
        }
    };

    auto issue_w1 = [&](int cabs, int tap, int slot, int i) {      // piece i of that tile
        const bool live = cabs < chunk_end;
        const unsigned soff = live ? (unsigned)(tap * cin + cabs * BK) * 2u : 0u;
        const bool real = !RAGW || i < GW - 1 || (wave + NW * i) < GWT;         // wave-uniform
        unsigned char* dst = real ? lds + slot * WTILE + (wave + NW * i) * (8 * BK * 2) : lds + OFF_DUMP;
        glds16_buf(rw, wvoff[i] | (live ? 0u : DMA_OOB) | (real ? 0u : DMA_OOB), soff, dst);
    };

    // ---- fragment addressing.  Activation fragment i of this wave = tile rows r = wave_m * WM + 16 i + fr; its tap
    // (0, 0) patch entry is abase[i]; tap (kh, kw) adds kh * PW + kw.  Weight fragment j = tile rows wave_n * WN + 16 j + fr.
    // stride 1: entry = abase[i] + kh PW + kw.  UP2: entry = rp[i][kh] + cq[kw] (row part per vertical tap, column part
    // ((tx + kw - 1) >> 1) + 1 per horizontal tap).
    int abase[FM], rp[UP2 ? FM : 1][3], cq[3];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int r = wave_m * WM + i * 16 + fr;
        const int ty = r >> TWl, tx = r & (TW - 1);
        const int g = g0 + ty;
        const int b = (int)(((float)g + 0.5f) * (1.0f / (float)H));
        abase[i] = g < GROWS ? (g + b - VLO - 1) * PW + tx : 0;
        if (UP2) {
            const int y = g - b * H;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) rp[i][kh] = g < GROWS ? (b * (HD + 1) + ((y + kh - 1) >> 1) - VLO) * PW : 0;
        }
    }
    // (the column part is the same for every fragment of a lane -- tx = fr & (TW - 1) -- and stays inside the patch's first
    // row for rows beyond the image stack, whose row part is 0)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) cq[kw] = UP2 ? (((fr & (TW - 1)) + kw - 1) >> 1) + 1 : 0;
    const int wl0 = (wave_n * WN + fr) * (BK * 2) + ((fg ^ (fr & 7)) << 4);       // ks = 0; ks = 1 flips bit 6

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 afA[FM], wfA[FN], afB[FM], wfB[FN];
    int aoff[FM];                                   // byte offset of the ks = 0 read of the current step (ks = 1: ^ 64)
    auto read_a0 = [&](int pb, int dtap, bf16x8 (&af)[FM]) {     // (prologue: tap (0, 0))
        const unsigned char* base = lds + OFF_A + pb * PBUF;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int P = UP2 ? rp[UP2 ? i : 0][0] + cq[0] : abase[i] + dtap;
            aoff[i] = (P << 7) + ((fg ^ (P & 7)) << 4);
            af[i] = lds_read16_async(base + aoff[i]);
        }
    };
    auto read_a1 = [&](int pb, bf16x8 (&af)[FM]) {
        const unsigned char* base = lds + OFF_A + pb * PBUF;
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = lds_read16_async(base + (aoff[i] ^ 64));
    };
    auto read_w = [&](int slot, int ks, bf16x8 (&wf)[FN]) {
        const unsigned char* base = lds + slot * WTILE + (wl0 ^ (ks << 6));
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[j] = lds_read16_async(base + j * (16 * BK * 2));
    };
    auto landed = [&](bf16x8 (&af)[FM], bf16x8 (&wf)[FN]) {
#pragma unroll
        for (int i = 0; i < FM; ++i) lds_tie(af[i]);
#pragma unroll
        for (int j = 0; j < FN; ++j) lds_tie(wf[j]);
    };
    auto mma = [&](const bf16x8 (&af)[FM], const bf16x8 (&wf)[FN]) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
    };

    // ---- prologue: weight tiles of steps 0 .. NSW - 2 first (their addresses are cheap), then -- while those fly --
    // the patch loader table, then the patch of the first chunk.  Small-operand integer divisions go through an exact
    // float reciprocal (operands < 2^18, validated on the host): a 32-bit division is ~40 instructions, and the table
    // needs 2 APW of them per lane.
    auto divf = [](int a, float rcp) { return (int)(((float)a + 0.5f) * rcp); };
    if (chunk_begin < chunk_end) {
#pragma unroll
        for (int s_ = 0; s_ < NSW - 1; ++s_) issue_w(chunk_begin, s_, s_);
        const float rpw = 1.0f / (float)PW, rh1 = 1.0f / (float)(HD + 1);
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            const int q = (wave + NW * j) * 8 + st_row;
            const int srow = divf(q, rpw), scol = q - srow * PW;
            const int v = VLO + srow;
            const int vb = v >= 0 ? divf(v, rh1) : 0, vy = v - vb * (HD + 1);
            const int xx = XLO + scol;
            const bool ok = (srow < PR) & (v >= 0) & (vy < HD) & (vb < p.batch) & (xx >= 0) & (xx < WD);
            ppix[j] = ok ? (unsigned)((vb * HD + vy) * WD + xx) : DMA_OOB;
            // The halo / separator entries are the same LDS slots for every chunk and each 16-byte slot is only ever
            // written by its own lane: zero them once, so the halo does not depend on whether the hardware writes zeros
            // for an out-of-range LDS-DMA lane or skips the lane
            if (!ok) {
                const u32x4 z = {0u, 0u, 0u, 0u};
                unsigned char* d = lds + OFF_A + (wave + NW * j) * (8 * BK * 2) + lane * 16;
                *(u32x4*)d = z;
                *(u32x4*)(d + PBUF) = z;
            }
        }
#pragma unroll
        for (int j = 0; j < APW; ++j) issue_a(j, chunk_begin, 0);
        wait_vmcnt<0>();
        barrier_keep_dma();
        read_a0(0, 0, afA);
        read_w(0, 0, wfA);
        lds_wait<0>();
        landed(afA, wfA);
    }
#if LECO_CONV_ABLATE & 32
    clk[1] = __builtin_amdgcn_s_memtime();
#endif
    int slot = 0, pb = 0;
    for (int c = chunk_begin; c < chunk_end; ++c) {
        // the tap offsets are loop invariant: without this the compiler hoists the fragment addresses of all 9 taps x FM
        // fragments (x 2 patch buffers) out of the loop and spills; recomputing them costs 5 VALU per read beside 2 FN MFMAs
#pragma unroll
        for (int i = 0; i < FM; ++i) opaque(abase[i]);
        if (UP2) { opaque(cq[0]); opaque(cq[1]); opaque(cq[2]); }
        // One tap step; everything that depends on `tap` is a compile-time constant.  Step t consumes weight tile t and
        // issues tile t + NSW - 1 into the slot tile t - 1 vacated at the PREVIOUS step's barrier, so its DMA pieces may go
        // out anywhere in the step.  They are spread evenly over the step's 2 FM FN MFMAs: a vector-memory instruction
        // occupies its wave for ~100-150 cycles (MI355X probe: ~14 GB/s of LDS-DMA per wave whatever the depth), and the two
        // waves of a SIMD run this stream in lockstep behind the barrier -- back-to-back pieces stall both, the matrix pipe idles.
        auto step = [&](auto tap_c) {
            constexpr int tap = decltype(tap_c)::value;
            constexpr int D = NSW - 1;
            constexpr int tn = (tap + D) % 9, cn = (tap + D) / 9;                // the weight tile this step issues
            constexpr int ND = GW + (tap < APW ? 1 : 0), P = FM * FN;
            // MFMA index (0 .. 2P - 1) AFTER which piece d is issued
            // (never in the last MFMAs before the barrier, where its issue time would delay the wave's arrival)
            constexpr auto dpos = [](int d, int nd) {
                const int x = ((2 * d + 1) * 2 * P) / (2 * nd) - 1;
                return (x >= P - 4 && x <= P + 1) ? P + 2 : x;
            };
            // pieces of this step issued before the barrier (which sits after MFMA P - 1)
            constexpr auto npre = [dpos](int nd) { int n = 0; for (int d = 0; d < nd; ++d) n += dpos(d, nd) < P ? 1 : 0; return n; };
            // DMAs younger than weight tile t + 1 (issued in step t + 1 - D) at this step's wait: steps t + 2 - D .. t - 1 in
            // full, and this step's pre-barrier pieces
            constexpr int younger = [npre] {
                int n = npre(ND);
                for (int d = 1; d <= D - 2; ++d) n += GW + ((((tap - d) % 9 + 9) % 9) < APW ? 1 : 0);
                return n;
            }();
            const int dnext = ((tap + 1) % 9 / 3) * PW + (tap + 1) % 9 % 3;      // patch offset of the next step's tap
            const int sfill = slot == 0 ? NSW - 1 : slot - 1;                    // slot of tile t - 1 = of tile t + NSW - 1
            const int snext = slot + 1 == NSW ? 0 : slot + 1;
            const int pbn = tap == 8 ? pb ^ 1 : pb;
            const unsigned char* abuf = lds + OFF_A + pbn * PBUF;
            const unsigned char* wbuf = lds + snext * WTILE + wl0;
            if (!(LECO_CONV_ABLATE & 1)) {
                read_a1(pb, afB);
                read_w(slot, 1, wfB);
            }
            // byte offsets of the NEXT step's activation fragments: VALU work beside the MFMAs of set A, so that the reads
            // themselves can go out right behind the barrier
            constexpr int khn = (tap + 1) % 9 / 3, kwn = (tap + 1) % 9 % 3;
#pragma unroll
            for (int q = 0; q < FM; ++q) {
                const int Pq = UP2 ? rp[UP2 ? q : 0][khn] + cq[kwn] : abase[q] + dnext;
                aoff[q] = (Pq << 7) + ((fg ^ (Pq & 7)) << 4);
            }
            lds_wait<FM + FN>();
            landed(afA, wfA);
#pragma unroll
            for (int m = 0; m < 2 * P; ++m) {
                const int mm = m < P ? m : m - P, i = mm / FN, j = mm % FN;
                if (m == P) {
                    sched_fence();
                    if (!(LECO_CONV_ABLATE & 4)) wait_vmcnt<younger>();    // weight tile t + 1 (and, at tap 8, the next chunk's patch) landed
                    if (!(LECO_CONV_ABLATE & 8)) barrier_keep_dma();       // ... for every wave; all waves are done with tile t's slot (completes set B)
                    else lds_wait<0>();
                    landed(afB, wfB);
                    // first fragment reads of the next step (ks = 0) right behind the barrier
                    if (!(LECO_CONV_ABLATE & 1)) {
#pragma unroll
                        for (int q = 0; q < FM; ++q) afA[q] = lds_read16_async(abuf + aoff[q]);
#pragma unroll
                        for (int q = 0; q < FN; ++q) wfA[q] = lds_read16_async(wbuf + q * (16 * BK * 2));
                    }
                    sched_fence();
                }
                if (LECO_CONV_ABLATE & 2) { if (m == 0) acc[i][j][0] += __uint_as_float((unsigned)(wfA[j][0] ^ afA[i][0] ^ wfB[j][0] ^ afB[i][0])); }
                else if (m < P) acc[i][j] = mfma16(wfA[j], afA[i], acc[i][j]);
                else acc[i][j] = mfma16(wfB[j], afB[i], acc[i][j]);
#pragma unroll
                for (int d = 0; d < ND; ++d)
                    if (dpos(d, ND) == m) {
                        sched_fence();
                        if (LECO_CONV_ABLATE & 16) {}
                        else if (tap < APW && d == 0) issue_a(tap, c + 1, pb ^ 1);
                        else issue_w1(c + cn, tn, sfill, d - (tap < APW ? 1 : 0));
                        sched_fence();
                    }
            }
            sched_fence();
            slot = snext;
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
        step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{});
        pb ^= 1;
        lds_wait<0>();                        // set A complete before the back edge / the loop exit
        landed(afA, wfA);
    }
    wait_vmcnt<0>();                          // the out-of-range DMAs issued past the end of the K range
#if LECO_CONV_ABLATE & 32
    clk[2] = __builtin_amdgcn_s_memtime();
#endif

    // ---- K-extension (leco_gemm_args.a_ext / w_ext: the LoRA up-projection of a c3lier convolution, lora.py:102-106; its
    // 3x3 down-projection T = conv(x, down) is a skinny launch of its own): one more step whose activation rows are the
    // tile's OUTPUT pixels of a_ext [M][ext_k] and whose weight rows are w_ext [N][ext_k] (ext_k = 32 or 64: one or two
    // 32-wide MFMA k-steps).  Both land in the layouts of the main loop -- A as patch entries (entry = tile row, 128-byte
    // rows, slot ^= entry & 7) in patch buffer 0, W in ring slot 0 -- so the fragment reads are the main loop's.  Split-K:
    // the first split carries it.
    if (p.a_ext && split == 0) {
        const buf_rsrc rax = make_rsrc(p.a_ext, rt.ax_bytes), rwx = make_rsrc(p.w_ext, rt.wx_bytes);
        const unsigned xk_bytes = (unsigned)p.ext_k * 2u;
        barrier_keep_dma();                   // every wave has issued (and completed) its last fragment reads of the main loop
#pragma unroll
        for (int j = 0; j < BM / 64; ++j) {
            const int r = (wave + NW * j) * 8 + st_row;                       // tile row = patch entry
            const int g = g0 + (r >> TWl), xx = x0 + (r & (TW - 1));
            const bool ok = g < GROWS && xx < W && cpos < xk_bytes;
            const unsigned voff = ok ? (unsigned)(g * W + xx) * (unsigned)p.ld_aext * 2u + cpos : DMA_OOB;
            glds16_buf(rax, voff, 0u, lds + OFF_A + (wave + NW * j) * (8 * BK * 2));
        }
#pragma unroll
        for (int i = 0; i < GW; ++i) {
            const int rl = (wave + NW * i) * 8 + st_row, n = n0 + rl;
            const bool real = !RAGW || i < GW - 1 || (wave + NW * i) < GWT;   // wave-uniform
            const bool ok = real && rl < BN && n < N && cpos < xk_bytes;
            unsigned char* dst = real ? lds + (wave + NW * i) * (8 * BK * 2) : lds + OFF_DUMP;
            glds16_buf(rwx, ok ? (unsigned)n * (unsigned)p.ld_wext * 2u + cpos : DMA_OOB, 0u, dst);
        }
        wait_vmcnt<0>();
        barrier_keep_dma();
        const unsigned char* abase0 = lds + OFF_A;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks * 32 >= p.ext_k) break;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int r = wave_m * WM + i * 16 + fr;
                afA[i] = lds_read16_async(abase0 + (((r << 7) + ((fg ^ (r & 7)) << 4)) ^ (ks << 6)));
            }
            read_w(0, ks, wfA);
            lds_wait<0>();
            landed(afA, wfA);
            mma(afA, wfA);
        }
    }

    // ---- epilogue through LDS: accumulators (lane = one pixel x 4 consecutive n) staged as fp32 -- the whole tile at
    // once where it fits (RR rows per round) -- so bias / residual reads and the bf16 (or fp32 partial) stores move whole
    // 16-byte row segments.  Every thread owns ITEMS (row, 8-column) items of a round; their residual loads are all
    // issued before the first is used (one exposed global latency per round, not one per item).
    constexpr int SROW = Cf::SROW, NC8 = BN / 8;
    constexpr int RR = BM * SROW * 4 <= Cf::LDS_BYTES ? BM : ((BM / 2) * SROW * 4 <= Cf::LDS_BYTES ? BM / 2 : 64);
    constexpr int ITEMS = RR * NC8 / NT;
    static_assert(RR * NC8 % NT == 0 && RR % WM == 0, "epilogue items must divide evenly over the threads");
    static_assert((RR * SROW + (NT / BN) * BN * 2) * 4 <= Cf::LDS_BYTES, "column-sum scratch must fit behind the staging rows");
    float* stg = (float*)lds;
    bf16_t* cp = (bf16_t*)p.c;
    const bf16_t* res = (const bf16_t*)p.residual;
    float* wsp = rt.split_k > 1 ? rt.ws + (int64_t)split * M * N : nullptr;
#pragma unroll
    for (int h = 0; h < BM / RR; ++h) {
        barrier_keep_dma();
        if ((wave_m * WM) / RR == h) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int rl = (wave_m * WM) % RR + i * 16 + fr;
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    *(f32x4*)(stg + rl * SROW + wave_n * WN + j * 16 + 4 * fg) = acc[i][j];
            }
        }
        barrier_keep_dma();
        int mrow[ITEMS];             // output row m of item `it`, or -1
        u32x4 rres[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int e = tid + it * NT;
            const int rl = e / NC8, cc = e - rl * NC8;
            const int r = h * RR + rl;
            const int g = g0 + (r >> TWl), xx = x0 + (r & (TW - 1));
            const int n = n0 + cc * 8;
            mrow[it] = (g < GROWS && xx < W && n < N) ? g * W + xx : -1;
            rres[it] = u32x4{0u, 0u, 0u, 0u};
            if (res && !wsp && mrow[it] >= 0) rres[it] = *(const u32x4*)(res + (int64_t)mrow[it] * p.ldr + n);
        }
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int e = tid + it * NT;
            const int rl = e / NC8, cc = e - rl * NC8;
            const int n = n0 + cc * 8, m = mrow[it];
            if (m < 0) continue;
            const f32x4 v0_ = *(const f32x4*)(stg + rl * SROW + cc * 8);
            const f32x4 v1_ = *(const f32x4*)(stg + rl * SROW + cc * 8 + 4);
            float v[8] = {v0_[0], v0_[1], v0_[2], v0_[3], v1_[0], v1_[1], v1_[2], v1_[3]};
            if (wsp) {
                *(f32x4*)(wsp + (int64_t)m * N + n) = v0_;
                *(f32x4*)(wsp + (int64_t)m * N + n + 4) = v1_;
                continue;
            }
            if (p.bias) {
                const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] += b0[q]; v[4 + q] += b1[q]; }
            }
            if (p.rowbias) {
                const float* rb = p.rowbias + (int64_t)(m / p.rows_per_group) * p.ld_rowbias + n;
                const f32x4 b0 = *(const f32x4*)rb, b1 = *(const f32x4*)(rb + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] += b0[q]; v[4 + q] += b1[q]; }
            }
            if (res) {
                const u32x4 rr = rres[it];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] += bf2f((bf16_t)(rr[q] & 0xffffu));
                    v[2 * q + 1] += bf2f((bf16_t)(rr[q] >> 16));
                }
            }
            if (p.act == LECO_ACT_SILU) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = v[q] / (1.f + __expf(-v[q]));
            }
            if (cp) {
                const u32x4 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
                *(u32x4*)(cp + (int64_t)m * p.ldc + n) = o;
                if (p.col_stats) {   // the values as stored (bf16-rounded) go back to the staging tile for the column sums
                    float* sr = stg + rl * SROW + cc * 8;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        sr[2 * q] = bf2f((bf16_t)(o[q] & 0xffffu));
                        sr[2 * q + 1] = bf2f((bf16_t)(o[q] >> 16));
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
            // GroupNorm statistics of the tensor this convolution produces (leco_hip.h): {sum, sumsq} per sample and ATOM of
            // stats_atom columns.  Thread = (column, one of NSEG row segments) walks down the staged tile; the partial column
            // sums meet in LDS and one thread per atom sends ONE pair of fp32 atomics (tiles whose rows belong to several
            // samples -- the small levels -- send one pair per column, segment and sample instead).
            barrier_keep_dma();
            constexpr int NSEG = NT / BN, SEGR = (RR + NSEG - 1) / NSEG;
            const int A = p.stats_atom, NA = N / A;
            float* csum = stg + RR * SROW;                      // [NSEG][BN][2], behind the staging rows
            // first / last valid row of the round -> one sample?  (rows are TH x TW blocks of the global row space)
            const int glo = g0 + ((h * RR) >> TWl), ghi_ = g0 + ((h * RR + RR - 1) >> TWl);
            const int ghi = ghi_ < GROWS ? ghi_ : GROWS - 1;
            const bool any = glo < GROWS;
            const bool single = any && (glo * W) / p.stats_rows == (ghi * W + W - 1) / p.stats_rows;
            const int col = tid % BN, seg = tid / BN, n = n0 + col;
            if (seg < NSEG) {
                float s1 = 0.f, s2 = 0.f;
                int bcur = -1;
                for (int rl = seg * SEGR; rl < (seg + 1) * SEGR && rl < RR; ++rl) {
                    const int r = h * RR + rl;
                    const int g = g0 + (r >> TWl), xx = x0 + (r & (TW - 1));
                    if (g >= GROWS || xx >= W) continue;
                    if (!single) {
                        const int b = (g * W + xx) / p.stats_rows;
                        if (b != bcur) {
                            if (bcur >= 0 && n < N) {
                                atomicAdd(p.col_stats + ((int64_t)bcur * NA + n / A) * 2, s1);
                                atomicAdd(p.col_stats + ((int64_t)bcur * NA + n / A) * 2 + 1, s2);
                            }
                            bcur = b; s1 = 0.f; s2 = 0.f;
                        }
                    }
                    const float x = stg[rl * SROW + col];
                    s1 += x;
                    s2 += x * x;
                }
                if (single) {
                    csum[(seg * BN + col) * 2] = s1;
                    csum[(seg * BN + col) * 2 + 1] = s2;
                } else if (bcur >= 0 && n < N) {
                    atomicAdd(p.col_stats + ((int64_t)bcur * NA + n / A) * 2, s1);
                    atomicAdd(p.col_stats + ((int64_t)bcur * NA + n / A) * 2 + 1, s2);
                }
            }
            if (single) {
                barrier_keep_dma();
                const int b = (glo * W) / p.stats_rows, nhi = (n0 + BN < N ? n0 + BN : N);
                const int a0 = n0 / A, a1 = (nhi - 1) / A;
                for (int a = a0 + tid; a <= a1; a += NT) {
                    const int c0 = a * A > n0 ? a * A : n0, c1 = (a + 1) * A < nhi ? (a + 1) * A : nhi;
                    float s1 = 0.f, s2 = 0.f;
                    for (int c = c0; c < c1; ++c)
#pragma unroll
                        for (int sg = 0; sg < NSEG; ++sg) {
                            s1 += csum[(sg * BN + c - n0) * 2];
                            s2 += csum[(sg * BN + c - n0) * 2 + 1];
                        }
                    atomicAdd(p.col_stats + ((int64_t)b * NA + a) * 2, s1);
                    atomicAdd(p.col_stats + ((int64_t)b * NA + a) * 2 + 1, s2);
                }
            }
        }
    }
#if LECO_CONV_ABLATE & 32
    __syncthreads();                          // stamps of workgroup 0 over the start of its own output (timing builds only)
    if (wg == 0 && tid == 0) {
        clk[3] = __builtin_amdgcn_s_memtime(); rtc[1] = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = (unsigned long long*)p.c;
        o[0] = clk[1] - clk[0]; o[1] = clk[2] - clk[1]; o[2] = clk[3] - clk[2]; o[3] = rtc[1] - rtc[0];
    }
#endif
}

// tile geometry of one (BM, problem): TW, tile counts, and whether every tile's patch fits the capacity
struct PatchGeom { int tw_log2, tiles_g, tiles_x; bool fits; };

PatchGeom patch_geometry(const leco_gemm_args& a, int bm, int pcap) {
    const bool up2 = a.a_mode == LECO_A_CONV3_UP2;
    const int H = a.h_out, W = a.w_out, grows = a.batch * H, hd = up2 ? a.h_in : H;
    // TW: the candidate that wastes fewer output columns; ties go to 16 (fragment rows = 16 consecutive patch
    // entries: conflict-free reads, smaller halo)
    auto cols = [&](int tw) { return cdiv(W, tw) * tw; };
    const int tw = cols(16) <= cols(8) ? 16 : 8;
    PatchGeom g{tw == 16 ? 4 : 3, 0, cdiv(W, tw), true};
    const int th = bm / tw, pw = up2 ? tw / 2 + 2 : tw + 2;
    g.tiles_g = cdiv(grows, th);
    for (int t = 0; t < g.tiles_g; ++t) {      // the kernel's VLO / VHI
        const int g0 = t * th, gl = (g0 + th < grows ? g0 + th : grows) - 1, bf = g0 / H, bl = gl / H;
        const int vlo = up2 ? bf * (hd + 1) + ((g0 - bf * H - 1) >> 1) : g0 + bf - 1;
        const int vhi = up2 ? bl * (hd + 1) + ((gl - bl * H + 1) >> 1) : gl + bl + 1;
        if ((vhi - vlo + 1) * pw > pcap) g.fits = false;
    }
    return g;
}

template <int BM, int BN, int NSW, bool UP2>
int launch_patch_m(const leco_gemm_args& a, int split_k, float* ws, hipStream_t s, char* describe, int describe_len) {
    using Cf = PatchCfg<BM, BN, NSW>;
    const PatchGeom g = patch_geometry(a, BM, Cf::PCAP);
    if (!g.fits) return 1;
    const int tn = cdiv(a.n, BN);
    // operand extents in bytes (the descriptors' range check makes DMA_OOB lanes read zeros): < 2^31 or no patch path
    const int cin = a.k / 9;
    const int64_t pixels = (int64_t)a.batch * a.h_in * a.w_in;
    const int c0 = a.a1 ? a.k_split : cin, c1 = cin - c0;
    const int64_t e0 = ((pixels - 1) * a.lda0 + c0) * 2, e1 = a.a1 ? ((pixels - 1) * a.lda1 + c1) * 2 : 0;
    const int64_t ew = ((int64_t)(a.n - 1) * a.ldw + a.k) * 2;
    const int64_t eax = a.a_ext ? ((int64_t)(a.m - 1) * a.ld_aext + a.ext_k) * 2 : 0;
    const int64_t ewx = a.a_ext ? ((int64_t)(a.n - 1) * a.ld_wext + a.ext_k) * 2 : 0;
    if (e0 >= (1ll << 31) || e1 >= (1ll << 31) || ew >= (1ll << 31) || eax >= (1ll << 31) || ewx >= (1ll << 31)) return 1;
    if ((int64_t)a.batch * (a.h_out + 1) >= (1 << 18)) return 1;      // exact float-reciprocal divisions in the kernel
    PatchRt rt{tn, g.tiles_x, g.tiles_g, split_k, g.tw_log2, (unsigned)e0, (unsigned)e1, (unsigned)ew, (unsigned)eax, (unsigned)ewx, ws};
    dim3 grid((unsigned)(g.tiles_g * g.tiles_x * tn), (unsigned)split_k);
    if (describe) {
        const int used = (int)strlen(describe);
        snprintf(describe + used, describe_len - used, "%sconv_patch_kernel<%d, %d, %d, %s> grid=%u split=%d", used ? " ; " : "",
                 BM, BN, NSW, UP2 ? "true" : "false", grid.x, split_k);
        return 0;
    }
    static bool attr_set[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (dev_id < 0 || dev_id >= 64 || !attr_set[dev_id]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_patch_kernel<BM, BN, NSW, UP2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS_BYTES);
        if (dev_id >= 0 && dev_id < 64) attr_set[dev_id] = true;
    }
    hipLaunchKernelGGL((conv_patch_kernel<BM, BN, NSW, UP2>), grid, dim3(512), Cf::LDS_BYTES, s, a, rt);
    return 0;
}
template <int BM, int BN, int NSW>
int launch_patch(const leco_gemm_args& a, int split_k, float* ws, hipStream_t s, char* describe, int describe_len) {
    if (a.a_mode == LECO_A_CONV3_UP2) return launch_patch_m<BM, BN, NSW, true>(a, split_k, ws, s, describe, describe_len);
    return launch_patch_m<BM, BN, NSW, false>(a, split_k, ws, s, describe, describe_len);
}
bool patch_applicable(const leco_gemm_args& a) {
    if (a.t_w || a.act == LECO_ACT_GEGLU) return false;
    if (a.a_ext && (!a.w_ext || (a.ext_k != 32 && a.ext_k != 64))) return false;
    if (a.a_mode == LECO_A_CONV3_S1) return a.h_in == a.h_out && a.w_in == a.w_out;
    if (a.a_mode == LECO_A_CONV3_UP2) return a.h_out == 2 * a.h_in && a.w_out == 2 * a.w_in;
    return false;
}
}  // namespace

// 0: launched (or described); 1: this problem / variant cannot take the patch path (the caller falls back to the
// implicit-GEMM kernel); < 0: error.  variant: 7 = 256 x 128, 8 = 128 x 160, 9 = 128 x 128, 10 = 256 x 160.
int conv_patch_try(const leco_gemm_args& a, int variant, int split_k, float* ws, hipStream_t s, char* describe,
                   int describe_len) {
    if (!patch_applicable(a)) return 1;
    const int nchunks = a.k / 9 / BK;
    if (split_k > nchunks) split_k = nchunks;
    if (split_k < 1) split_k = 1;
    switch (variant) {
        case 7: return launch_patch<256, 128, 4>(a, split_k, ws, s, describe, describe_len);
        case 8: return launch_patch<128, 160, 4>(a, split_k, ws, s, describe, describe_len);
        case 9: return launch_patch<128, 128, 4>(a, split_k, ws, s, describe, describe_len);
        case 10: return launch_patch<256, 160, 3>(a, split_k, ws, s, describe, describe_len);
        default: return 1;
    }
}

// Launch shape for a 3x3 / stride-1 convolution without a tuner-table entry: every (variant, split_k) is priced with
// a small model fitted to MI355X measurements (profiles/r03_conv_patch.txt) and the cheapest wins:
//   one workgroup = ~5 us of prologue + epilogue + 9 * ceil(chunks / split) tap steps of 0.33 / 0.49 / 0.70 / 0.93 us
//   (128x128, 128x160, 256x128, 256x160: MFMA issue + the LDS traffic of the fragment reads), rounds = ceil(workgroups /
//   256 CUs); split-K adds the finishing launch (~3 us) and the fp32 slabs' write + read at ~4 TB/s.
// Returns the tile id (7..10) and sets *split, or 0 when the patch path does not apply.
int conv_patch_choose(const leco_gemm_args& a, int64_t ws_bytes, int split_in, int* split) {
    if (!patch_applicable(a)) return 0;
    struct V { int id, bm, bn, pcap; double step_us; };
    const V vs[4] = {{9, 128, 128, PatchCfg<128, 128, 4>::PCAP, 0.33}, {8, 128, 160, PatchCfg<128, 160, 4>::PCAP, 0.49},
                     {7, 256, 128, PatchCfg<256, 128, 4>::PCAP, 0.70}, {10, 256, 160, PatchCfg<256, 160, 3>::PCAP, 0.93}};
    const int nchunks = a.k / 9 / BK;
    const int splits[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};
    double best = 1e30;
    int best_id = 0;
    *split = 1;
    for (const V& v : vs) {
        const PatchGeom g = patch_geometry(a, v.bm, v.pcap);
        if (!g.fits) continue;
        const long wgs1 = (long)g.tiles_g * g.tiles_x * cdiv(a.n, v.bn);
        for (int sp : splits) {
            if (split_in > 0 && sp != split_in) continue;
            if (sp > nchunks || (sp > 1 && (int64_t)sp * a.m * a.n * 4 > ws_bytes)) continue;
            const long rounds = (wgs1 * sp + 255) / 256;
            double us = rounds * (5.0 + 9.0 * cdiv(nchunks, sp) * v.step_us);
            if (sp > 1) us += 3.0 + 2.0 * sp * (double)a.m * a.n * 4.0 / 4.0e6;
            if (us < best) { best = us; best_id = v.id; *split = sp; }
        }
    }
    return best_id;
}
}  // namespace leco
