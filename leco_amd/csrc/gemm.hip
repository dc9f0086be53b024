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
// Tiling: 256 threads = 4 waves in a 2x2 grid over a BM x BN tile, BK = 64.  Global -> VGPR
// prefetch of tile t+1 is issued before the MFMAs of tile t; LDS is single-buffered, XOR
// swizzled on 16-byte chunks (chunk ^= row & 7) so that the ds_read_b128 fragment reads are
// conflict-free.  Operands are fed "swapped" (W fragment as MFMA-A, activation fragment as
// MFMA-B) so each lane ends up with 4 consecutive n of one output row -> 8-byte stores.
// Workgroup ids are remapped so each XCD (private L2) owns a contiguous run of tiles.
#include <errno.h>
#include <hip/hip_runtime.h>
#include <leco_prims.h>

#include "common.h"

namespace leco {
namespace {

constexpr int BK = 64;

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * BK + ((chunk ^ (row & 7)) << 3);
}

template <int BM, int BN, bool CONV>
__global__ __launch_bounds__(256) void gemm_kernel(const leco_gemm_args p, int tiles_n) {
    constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
    constexpr int NA = BM / 32, NW = BN / 32;
    __shared__ __attribute__((aligned(16))) bf16_t smem[(BM + BN) * BK];
    bf16_t* sA = smem;
    bf16_t* sB = smem + BM * BK;

    // XCD-aware bijective remap: hardware places workgroup b on XCD b % 8; give each XCD a
    // contiguous run of logical tiles (tile_n fastest) so A/W panels are re-used in its L2.
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tile_m = wg / tiles_n, tile_n = wg % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int ld_row = tid >> 3, ld_c = tid & 7;

    const bf16_t* a0 = (const bf16_t*)p.a0;
    const bf16_t* a1 = (const bf16_t*)p.a1;
    const bf16_t* wp = (const bf16_t*)p.w;
    const bf16_t* aext = (const bf16_t*)p.a_ext;
    const bf16_t* wext = (const bf16_t*)p.w_ext;
    const int M = p.m, N = p.n;
    const int nk_main = p.k / BK;
    const int nk = nk_main + ((aext != nullptr && p.ext_k > 0) ? 1 : 0);
    const int k_split = a1 ? p.k_split : 0x7fffffff;

    // conv: decompose this thread's NA output rows once
    int pb[NA], py[NA], px[NA];
    const int cin = CONV ? p.k / 9 : 1;
    if (CONV) {
        const int hw = p.h_out * p.w_out;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int m = m0 + ld_row + 32 * i;
            int mm = m < M ? m : 0;
            pb[i] = mm / hw;
            int rem = mm - pb[i] * hw;
            py[i] = rem / p.w_out;
            px[i] = rem - py[i] * p.w_out;
        }
    }

    u32x4 ra[NA], rw[NW];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    auto load_tile = [&](int kt) {
        if (kt < nk_main) {
            const int k0 = kt * BK;
            if (!CONV) {
                const bf16_t* src;
                int64_t ld;
                int kk;
                if (k0 < k_split) { src = a0; ld = p.lda0; kk = k0; }
                else { src = a1; ld = p.lda1; kk = k0 - k_split; }
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    int m = m0 + ld_row + 32 * i;
                    ra[i] = (m < M) ? *(const u32x4*)(src + (int64_t)m * ld + kk + ld_c * 8) : zero4;
                }
            } else {
                const int tap = k0 / cin, c = k0 - tap * cin;
                const int kh = tap / 3, kw = tap - kh * 3;
                const bf16_t* src;
                int64_t ld;
                int cc;
                if (c < k_split) { src = a0; ld = p.lda0; cc = c; }
                else { src = a1; ld = p.lda1; cc = c - k_split; }
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    int m = m0 + ld_row + 32 * i;
                    int uy = py[i] + kh - 1, ux = px[i] + kw - 1;
                    bool ok = m < M;
                    int iy, ix;
                    if (p.a_mode == LECO_A_CONV3_S1) {
                        iy = uy; ix = ux;
                        ok = ok && iy >= 0 && iy < p.h_in && ix >= 0 && ix < p.w_in;
                    } else if (p.a_mode == LECO_A_CONV3_S2) {
                        iy = 2 * py[i] + kh - 1; ix = 2 * px[i] + kw - 1;
                        ok = ok && iy >= 0 && iy < p.h_in && ix >= 0 && ix < p.w_in;
                    } else if (p.a_mode == LECO_A_CONV3_UP2) {
                        ok = ok && uy >= 0 && uy < 2 * p.h_in && ux >= 0 && ux < 2 * p.w_in;
                        iy = uy >> 1; ix = ux >> 1;
                    } else {  // LECO_A_CONV3_TR2
                        ok = ok && uy >= 0 && ux >= 0 && ((uy | ux) & 1) == 0;
                        iy = uy >> 1; ix = ux >> 1;
                        ok = ok && iy < p.h_in && ix < p.w_in;
                    }
                    ra[i] = ok ? *(const u32x4*)(src + ((int64_t)(pb[i] * p.h_in + iy) * p.w_in + ix) * ld +
                                                 cc + ld_c * 8)
                               : zero4;
                }
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                int n = n0 + ld_row + 32 * i;
                rw[i] = (n < N) ? *(const u32x4*)(wp + (int64_t)n * p.ldw + k0 + ld_c * 8) : zero4;
            }
        } else {  // LoRA K-extension tile
            const bool cok = ld_c * 8 < p.ext_k;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int m = m0 + ld_row + 32 * i;
                ra[i] = (cok && m < M) ? *(const u32x4*)(aext + (int64_t)m * p.ld_aext + ld_c * 8) : zero4;
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                int n = n0 + ld_row + 32 * i;
                rw[i] = (cok && n < N) ? *(const u32x4*)(wext + (int64_t)n * p.ld_wext + ld_c * 8) : zero4;
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) *(u32x4*)(sA + lds_off(ld_row + 32 * i, ld_c)) = ra[i];
#pragma unroll
        for (int i = 0; i < NW; ++i) *(u32x4*)(sB + lds_off(ld_row + 32 * i, ld_c)) = rw[i];
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_tile(0);
    store_tile();
    __syncthreads();

    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) load_tile(kt + 1);
        const int ksteps = (kt >= nk_main && p.ext_k <= 32) ? 1 : 2;
        for (int ks = 0; ks < ksteps; ++ks) {
            bf16x8 af[FM], wf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i)
                af[i] = *(const bf16x8*)(sA + lds_off(wave_m * WM + i * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int j = 0; j < FN; ++j)
                wf[j] = *(const bf16x8*)(sB + lds_off(wave_n * WN + j * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }

    // epilogue: lane holds C[m = .. + fr][n = .. + 4*fg + r], r = 0..3
    bf16_t* cp = (bf16_t*)p.c;
    const bf16_t* res = (const bf16_t*)p.residual;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wave_m * WM + i * 16 + fr;
        if (m >= M) continue;
        const float* rb = p.rowbias ? p.rowbias + (int64_t)(m / p.rows_per_group) * p.ld_rowbias : nullptr;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wave_n * WN + j * 16 + 4 * fg;
            if (n >= N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.bias) {
                f32x4 b = *(const f32x4*)(p.bias + n);
                v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
            }
            if (rb) {
                f32x4 b = *(const f32x4*)(rb + n);
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
}

template <int BM, int BN>
int launch(const leco_gemm_args& a, hipStream_t s) {
    const int tm = cdiv(a.m, BM), tn = cdiv(a.n, BN);
    dim3 grid((unsigned)(tm * tn)), block(256);
    if (a.a_mode == LECO_A_PLAIN)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, false>), grid, block, 0, s, a, tn);
    else
        hipLaunchKernelGGL((gemm_kernel<BM, BN, true>), grid, block, 0, s, a, tn);
    return check_launch("leco_gemm");
}

int validate(const leco_gemm_args& a) {
    if (a.m <= 0 || a.n <= 0 || a.k <= 0) return fail(-EINVAL, "leco_gemm: empty problem m=%d n=%d k=%d", a.m, a.n, a.k);
    if (a.k % BK) return fail(-EINVAL, "leco_gemm: k=%d not a multiple of 64", a.k);
    if (a.n % 4) return fail(-EINVAL, "leco_gemm: n=%d not a multiple of 4", a.n);
    if (!a.a0 || !a.w) return fail(-EINVAL, "leco_gemm: null operand");
    if (!a.c && !a.c_f32) return fail(-EINVAL, "leco_gemm: no output");
    if (a.a_ext && a.ext_k != 32 && a.ext_k != 64) return fail(-EINVAL, "leco_gemm: ext_k=%d must be 32 or 64", a.ext_k);
    if (a.a_ext && !a.w_ext) return fail(-EINVAL, "leco_gemm: a_ext without w_ext");
    if (a.rowbias && a.rows_per_group <= 0) return fail(-EINVAL, "leco_gemm: rowbias needs rows_per_group");
    if (a.a_mode < LECO_A_PLAIN || a.a_mode > LECO_A_CONV3_TR2) return fail(-EINVAL, "leco_gemm: bad a_mode %d", a.a_mode);
    if (a.a_mode != LECO_A_PLAIN) {
        if (a.k % 9 || (a.k / 9) % BK) return fail(-EINVAL, "leco_gemm: conv needs k = 9*Cin, Cin %% 64 == 0 (k=%d)", a.k);
        if ((int64_t)a.batch * a.h_out * a.w_out != a.m) return fail(-EINVAL, "leco_gemm: conv m != batch*h_out*w_out");
        if (a.a1 && (a.k_split % BK)) return fail(-EINVAL, "leco_gemm: conv k_split %% 64 != 0");
    } else if (a.a1 && (a.k_split % BK)) {
        return fail(-EINVAL, "leco_gemm: k_split %% 64 != 0");
    }
    return 0;
}
}  // namespace
}  // namespace leco

// tile: 0 = heuristic, 1 = 128x128, 2 = 128x160, 3 = 64x64 (exposed for tests/tuning)
extern "C" int leco_gemm_tile(const leco_gemm_args* args, int tile, leco_stream_t stream) {
    using namespace leco;
    if (!args) return fail(-EINVAL, "leco_gemm: null args");
    int rc = validate(*args);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (tile == 0) {
        const int m = args->m, n = args->n;
        if (n <= 64 || m <= 64) tile = 3;
        else {
            const int bn = (n % 128 == 0) ? 128 : ((n % 160 == 0) ? 160 : 128);
            const long blocks = (long)cdiv(m, 128) * cdiv(n, bn);
            tile = blocks >= 192 ? (bn == 160 ? 2 : 1) : 3;
        }
    }
    switch (tile) {
        case 1: return launch<128, 128>(*args, s);
        case 2: return launch<128, 160>(*args, s);
        case 3: return launch<64, 64>(*args, s);
        default: return fail(-EINVAL, "leco_gemm: bad tile id %d", tile);
    }
}

extern "C" int leco_gemm(const leco_gemm_args* args, leco_stream_t stream) {
    return leco_gemm_tile(args, 0, stream);
}
