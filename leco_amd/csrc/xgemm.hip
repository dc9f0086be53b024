// A-stationary bf16 GEMM for the SHORT-K, SMALL-M Linears of the transformer blocks (gfx950, wave64,
// v_mfma_f32_16x16x32_bf16):
//
//   C[M][N] = A[M][K] W[N][K]^T (+ (A down^T)(scale up)^T) + bias (+ residual),   K in {320, 640, 1280}, N % 128 == 0
//
// i.e. attn{1,2}.to_q / to_out.0, the fused q|k|v projection, Transformer2DModel.proj_in / proj_out of the 32^2 / 16^2 /
// 8^2 levels of diffusers' UNet (call site train_util.py:156-160; LoRA term lora.py:102-106) in the forward-only plans
// (the k LoRA-on denoising passes, train_util.py:172-193, and the batched LoRA-off predictions, train_lora.py:202-237).
//
// Why another GEMM (round-4 verdict: `gemm_kernel<64, 64, ...>` = 10 % of the step at 0.10 of the MFMA peak): for these
// shapes (M = 1024 / 4096, N = K = 1280 / 640) every tiling that fills the chip has ~20 K steps of very little MFMA work,
// and gemm.hip pays a DMA -> barrier -> ds_read round trip per step for BOTH operands -- the launches sit at ~3x the time
// their L2 -> CU traffic needs.  Here, following the stripe kernels (stripe.hip):
//   * the workgroup's BM x K activation tile is staged ONCE (LDS-DMA, k-tiled [K / 64][BM][128 B], chunk ^= row & 7 on the
//     source side: conflict-free ds_read_b128 fragment reads) -- one wait, one barrier;
//   * the weights never touch LDS: each of the 8 waves owns one 16-column fragment of the 128-column tile and streams it
//     global -> VGPR from the MFMA-fragment-ordered weight image (`leco_xlin.packed`: one k-step of one fragment = 1 KB
//     contiguous), PF k-steps ahead, no barrier inside the K loop;
//   * LoRA: T = A down^T rides along as one more fragment on one or two "duty" waves, is rounded to bf16 into LDS (aliasing
//     the dead activation tile) and enters as one extra k-step against scale * up -- the K-extension of gemm.hip, fp32
//     accumulation of the low-rank term included;
//   * epilogue from registers: lane = one row x 4 consecutive n -> 8-byte residual loads (issued before the K loop) and
//     8-byte stores.
// BM = 64 for K <= 640, 32 for K = 1280 (the tile is 80 KB either way).
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include <leco_prims.h>

#include "common.h"

namespace leco {
namespace {

struct XgArgs {
    int m, n;
    const bf16_t* a; int64_t lda;
    const void* w; unsigned w_bytes;
    const float* bias;
    const void* dn; unsigned dn_bytes, lddn_b; int tf;      // tf: 0 LoRA off, 1 / 2 = 16 / 32 stacked lora_down rows
    const void* up; unsigned up_bytes, ldup_b;
    const bf16_t* res; int64_t ldr;
    bf16_t* c; int64_t ldc;
    int tiles_m, tiles_n;
};

// VAR (measurement variants, LECO_XGEMM_VAR): 0 = plain loop, PF = 10; 1 = activation fragments of k-step kt + 1 read
// (asynchronously) before the MFMAs of k-step kt; 2 = as 1 with PF = 5 and a 128-register budget (two workgroups per CU)
template <int BM, int K, int VAR>
__global__ __launch_bounds__(512) LECO_MIN_WAVES_PER_SIMD(VAR == 2 ? 4 : 2) void xgemm_kernel(const XgArgs p) {
    constexpr int RF = BM / 16;               // 16-row fragments of the activation tile
    constexpr int KS = K / 32, KT = K / 64;   // k-steps of 32, LDS k-tiles of 64
    constexpr int PF = VAR == 2 ? 5 : (KS < 10 ? KS : 10);     // weight fragments in flight per wave (k-steps)
    constexpr int PIECES = KT * (BM / 8) / 8; // LDS-DMA pieces per wave for the activation tile
    static_assert(KS % PF == 0 && (KT * (BM / 8)) % 8 == 0, "tile geometry");
    static_assert(BM * K * 2 <= 80 * 1024 && BM * 64 <= BM * K * 2, "LDS layout");
    unsigned char* lds = dyn_lds();

    // ---- XCD-aware bijective remap (hardware places linear workgroup id b on XCD b % 8; each XCD has a private L2): an XCD
    // gets a contiguous run of tiles, walked so that it keeps the LARGER operand's slice to itself (gemm.hip)
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    int tile_m, tile_n;
    if (p.n > p.m) { tile_n = wg / p.tiles_m; tile_m = wg - tile_n * p.tiles_m; }
    else { tile_m = wg / p.tiles_n; tile_n = wg - tile_m * p.tiles_n; }
    const int m0 = tile_m * BM, n0 = tile_n * 128;

    const int tid = (int)threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int st_row = lane >> 3, st_pos = lane & 7;
    const int nw = n0 + 16 * wave;            // this wave's 16 output columns
    const bool duty = p.tf && (wave == 0 || (wave == 1 && p.tf == 2));   // wave-uniform

    // ---- 1. activation tile: global -> LDS by DMA, once.  Piece q = k-tile q / (BM / 8), 8-row group q % (BM / 8); the wave
    // instruction fills 1 KB lane-linearly, so lane (row st_row, slot st_pos) fetches chunk st_pos ^ st_row of its row
    // (rows >= M re-read row M - 1: computed, never stored)
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
        const int q = wave + 8 * j;
        const int kt = q / (BM / 8), g = q - kt * (BM / 8);
        int row = m0 + 8 * g + st_row;
        row = row < p.m ? row : p.m - 1;
        glds16(p.a + (int64_t)row * p.lda + kt * 64 + ((st_pos ^ st_row) << 3), lds + q * 1024);
    }
    // ---- 2. the first PF k-steps of this wave's weight fragment (+ the lora_down fragment on the duty waves), the residual
    // and the bias: all in flight together with the tile
    const buf_rsrc rw = make_rsrc(p.w, p.w_bytes);
    const unsigned vw = (unsigned)(lane << 4);
    const unsigned sw = (unsigned)uniform((nw >> 4) * KS * 1024);
    bf16x8 wf[PF], df[PF];
#pragma unroll
    for (int s = 0; s < PF; ++s) wf[s] = buf_load16(rw, vw, sw + (unsigned)s * 1024u);
    const buf_rsrc rd = make_rsrc(duty ? p.dn : p.w, duty ? p.dn_bytes : p.w_bytes);
    const unsigned vd = (unsigned)(16 * wave + fr) * p.lddn_b + (unsigned)(fg << 4);
    if (duty) {
#pragma unroll
        for (int s = 0; s < PF; ++s) df[s] = buf_load16(rd, vd, (unsigned)s * 64u);
    }
    u32x2 rres[RF];
#pragma unroll
    for (int i = 0; i < RF; ++i) {
        const int row = m0 + 16 * i + fr;
        rres[i] = u32x2{0u, 0u};
        if (p.res && row < p.m) rres[i] = *(const u32x2*)(p.res + (int64_t)row * p.ldr + nw + 4 * fg);
    }
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *(const f32x4*)(p.bias + nw + 4 * fg);
    bf16x8 uf = {0, 0, 0, 0, 0, 0, 0, 0};
    if (p.tf) {      // scale * up rows of this wave's columns: [N][32] bf16, row n at n * ldup_b
        const buf_rsrc ru = make_rsrc(p.up, p.up_bytes);
        uf = buf_load16(ru, (unsigned)fr * p.ldup_b + (unsigned)(fg << 4), (unsigned)uniform(nw) * p.ldup_b);
    }
    wait_vmcnt<0>();
    barrier_keep_dma();

    // ---- 3. K loop: no barrier, no LDS write.  Fragment i of k-step kt = rows 16 i + fr, 16-byte chunk 4 (kt & 1) + fg of
    // k-tile kt >> 1
    f32x4 acc[RF], acct[RF];
#pragma unroll
    for (int i = 0; i < RF; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; acct[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const unsigned char* arow = lds + fr * 128;
    const int sw7 = fr & 7;
    auto frag_ptr = [&](int kt) { return arow + (kt >> 1) * (BM * 128) + ((((kt & 1) << 2) + fg) ^ sw7) * 16; };
    if constexpr (VAR == 0) {
#pragma unroll
        for (int kt = 0; kt < KS; ++kt) {
            const unsigned char* ap = frag_ptr(kt);
            bf16x8 a[RF];
#pragma unroll
            for (int i = 0; i < RF; ++i) a[i] = *(const bf16x8*)(ap + i * (16 * 128));
            const int slot = kt % PF;
#pragma unroll
            for (int i = 0; i < RF; ++i) acc[i] = mfma16(wf[slot], a[i], acc[i]);
            if (duty) {
#pragma unroll
                for (int i = 0; i < RF; ++i) acct[i] = mfma16(df[slot], a[i], acct[i]);
            }
            if (kt + PF < KS) {
                wf[slot] = buf_load16(rw, vw, sw + (unsigned)(kt + PF) * 1024u);
                if (duty) df[slot] = buf_load16(rd, vd, (unsigned)(kt + PF) * 64u);
            }
        }
    } else {
        // two fragment sets: the reads of k-step kt + 1 are in flight during the MFMAs of k-step kt (explicit asynchronous
        // ds_reads, counted lgkmcnt: the only LDS traffic of this loop; both sets complete before the loop is left)
        bf16x8 aA[RF], aB[RF];
        auto rd_set = [&](int kt, bf16x8 (&a)[RF]) {
            const unsigned char* ap = frag_ptr(kt);
#pragma unroll
            for (int i = 0; i < RF; ++i) a[i] = lds_read16_async(ap + i * (16 * 128));
        };
        auto tie_set = [&](bf16x8 (&a)[RF]) {
#pragma unroll
            for (int i = 0; i < RF; ++i) lds_tie(a[i]);
        };
        auto step = [&](int kt, const bf16x8 (&a)[RF]) {
            const int slot = kt % PF;
#pragma unroll
            for (int i = 0; i < RF; ++i) acc[i] = mfma16(wf[slot], a[i], acc[i]);
            if (duty) {
#pragma unroll
                for (int i = 0; i < RF; ++i) acct[i] = mfma16(df[slot], a[i], acct[i]);
            }
            if (kt + PF < KS) {
                wf[slot] = buf_load16(rw, vw, sw + (unsigned)(kt + PF) * 1024u);
                if (duty) df[slot] = buf_load16(rd, vd, (unsigned)(kt + PF) * 64u);
            }
        };
        static_assert(KS % 2 == 0, "the pipelined loop consumes k-steps in pairs");
        rd_set(0, aA);
#pragma unroll
        for (int kt = 0; kt < KS; kt += 2) {
            rd_set(kt + 1, aB);
            lds_wait<RF>();
            tie_set(aA);
            step(kt, aA);
            if (kt + 2 < KS) { rd_set(kt + 2, aA); lds_wait<RF>(); } else lds_wait<0>();
            tie_set(aB);
            step(kt + 1, aB);
        }
        lds_wait<0>();
    }

    // ---- 4. LoRA K-extension: T -> bf16 into LDS ([BM][32], 64-byte rows, over the dead activation tile), one k-step
    // against scale * up.  Lane holds T[16 i + fr][16 wave + 4 fg + r].
    if (p.tf) {
        barrier_keep_dma();                    // every wave is done reading the activation tile
        if (p.tf == 1 && wave == 1) {          // columns 16 .. 31 of a 16-row stack: zeros (up is zero there too)
#pragma unroll
            for (int i = 0; i < RF; ++i) *(u32x2*)(lds + (16 * i + fr) * 64 + 32 + 8 * fg) = u32x2{0u, 0u};
        }
        if (duty) {
#pragma unroll
            for (int i = 0; i < RF; ++i) {
                const u32x2 t = {pack_bf2(acct[i][0], acct[i][1]), pack_bf2(acct[i][2], acct[i][3])};
                *(u32x2*)(lds + (16 * i + fr) * 64 + 32 * wave + 8 * fg) = t;
            }
        }
        barrier_keep_dma();
#pragma unroll
        for (int i = 0; i < RF; ++i) {
            const bf16x8 t = *(const bf16x8*)(lds + (16 * i + fr) * 64 + 16 * fg);
            acc[i] = mfma16(uf, t, acc[i]);
        }
    }

    // ---- 5. epilogue from registers: lane = row 16 i + fr, columns nw + 4 fg .. + 3
#pragma unroll
    for (int i = 0; i < RF; ++i) {
        const int row = m0 + 16 * i + fr;
        if (row >= p.m) continue;
        float v0 = acc[i][0] + bias4[0], v1 = acc[i][1] + bias4[1], v2 = acc[i][2] + bias4[2], v3 = acc[i][3] + bias4[3];
        if (p.res) {
            v0 += bf2f((bf16_t)(rres[i][0] & 0xffffu)); v1 += bf2f((bf16_t)(rres[i][0] >> 16));
            v2 += bf2f((bf16_t)(rres[i][1] & 0xffffu)); v3 += bf2f((bf16_t)(rres[i][1] >> 16));
        }
        const u32x2 o = {pack_bf2(v0, v1), pack_bf2(v2, v3)};
        *(u32x2*)(p.c + (int64_t)row * p.ldc + nw + 4 * fg) = o;
    }
}

template <int BM, int K, int VAR>
int launch_v(const XgArgs& p, hipStream_t s) {
    constexpr int lds_bytes = BM * K * 2;
    static bool attr_set[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (dev_id < 0 || dev_id >= 64 || !attr_set[dev_id]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xgemm_kernel<BM, K, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  lds_bytes);
        if (dev_id >= 0 && dev_id < 64) attr_set[dev_id] = true;
    }
    hipLaunchKernelGGL((xgemm_kernel<BM, K, VAR>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds_bytes, s, p);
    return check_launch("leco_xgemm");
}
template <int BM, int K>
int launch(const XgArgs& p, hipStream_t s) {
    static const int var = [] { const char* e = getenv("LECO_XGEMM_VAR"); return e ? atoi(e) : 1; }();
    if (var == 0) return launch_v<BM, K, 0>(p, s);
    if (var == 2) return launch_v<BM, K, 2>(p, s);
    return launch_v<BM, K, 1>(p, s);
}

}  // namespace
}  // namespace leco

extern "C" int leco_xgemm_supported(int32_t m, int32_t n, int32_t k) {
    if (m <= 0 || n <= 0 || n % 128) return 0;
    return (k == 320 || k == 640 || k == 1280) ? 1 : 0;
}

extern "C" int leco_xgemm_rows(int32_t k) { return k == 1280 ? 32 : 64; }

extern "C" int leco_xgemm(const leco_xgemm_args* a, leco_stream_t stream) {
    using namespace leco;
    if (!a) return fail(-EINVAL, "leco_xgemm: null args");
    if (!leco_xgemm_supported(a->m, a->n, a->k))
        return fail(-EINVAL, "leco_xgemm: unsupported shape m=%d n=%d k=%d (n %% 128 == 0, k in {320, 640, 1280})", a->m, a->n, a->k);
    if (!a->a || !a->lin.w || !a->c) return fail(-EINVAL, "leco_xgemm: null operand");
    if (!a->lin.packed || a->lin.ldw != a->k) return fail(-EINVAL, "leco_xgemm: the weight must be in MFMA fragment order (leco_xlin.packed, ldw == k)");
    if (a->lda % 8 || a->ldc % 4 || (a->residual && a->ldr % 4))
        return fail(-EINVAL, "leco_xgemm: activation strides must keep 16-byte (a) / 8-byte (c, residual) alignment");
    const leco_xlin& L = a->lin;
    if (L.dn && (!L.up || (L.t_rows != 16 && L.t_rows != 32) || L.ld_dn % 8 || L.ld_up < 32 || L.ld_up % 8))
        return fail(-EINVAL, "leco_xgemm: LoRA needs dn, up, t_rows in {16, 32}, 16-byte aligned strides, ld_up >= 32");
    XgArgs p;
    memset(&p, 0, sizeof(p));
    p.m = a->m; p.n = a->n;
    p.a = (const bf16_t*)a->a; p.lda = a->lda;
    p.w = L.w; p.w_bytes = (unsigned)((int64_t)a->n * a->k * 2);
    p.bias = L.bias;
    if (L.dn) {
        p.dn = L.dn; p.lddn_b = (unsigned)(L.ld_dn * 2); p.dn_bytes = (unsigned)(L.t_rows * L.ld_dn * 2); p.tf = L.t_rows / 16;
        p.up = L.up; p.ldup_b = (unsigned)(L.ld_up * 2); p.up_bytes = (unsigned)((int64_t)a->n * L.ld_up * 2);
    }
    p.res = (const bf16_t*)a->residual; p.ldr = a->ldr;
    p.c = (bf16_t*)a->c; p.ldc = a->ldc;
    const int bm = leco_xgemm_rows(a->k);
    p.tiles_m = cdiv(a->m, bm); p.tiles_n = a->n / 128;
    hipStream_t s = (hipStream_t)stream;
    switch (a->k) {
        case 320: return launch<64, 320>(p, s);
        case 640: return launch<64, 640>(p, s);
        default: return launch<32, 1280>(p, s);
    }
}
