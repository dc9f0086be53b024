// HBM-bound elementwise / small kernels of the LECO step: GEGLU, conv_in / conv_out (the two
// convolutions that are not GEMM-shaped: Cin = 4 and Cout = 4), timestep sinusoid, gradient adds,
// 2x2 upsample dgrad, CFG-combine + DDIM update, ESD loss + its gradient, fused AdamW.
// All bf16 traffic is moved as 16-byte vectors (8 x bf16 per lane); latent-sized tensors
// ((bs,4,h,w), a few hundred KB) stay fp32.
#include <errno.h>
#include <hip/hip_runtime.h>
#include <leco_prims.h>
#include <math.h>

#include "common.h"

namespace leco {
namespace {

__device__ __forceinline__ void unpack8(const u32x4& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = bf2f((bf16_t)(v[i] & 0xffffu));
        f[2 * i + 1] = bf2f((bf16_t)(v[i] >> 16));
    }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
    return v;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float dgelu_erf(float x) {
    return 0.5f * (1.f + erff(x * 0.7071067811865476f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

inline int grid_for(int64_t n_items) {
    int64_t g = (n_items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// ---- GEGLU: y = a * gelu(g), u = [a | g] -----------------------------------------------------
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const bf16_t* u, int64_t ldu, bf16_t* y, int64_t ldy,
                                                         int M, int F) {
    const int nv = F / 8;
    const int64_t total = (int64_t)M * nv;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / nv;
        const int c = (int)(e - r * nv) * 8;
        float a[8], g[8], o[8];
        unpack8(*(const u32x4*)(u + r * ldu + c), a);
        unpack8(*(const u32x4*)(u + r * ldu + F + c), g);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = a[i] * gelu_erf(g[i]);
        *(u32x4*)(y + r * ldy + c) = pack8(o);
    }
}
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const bf16_t* u, int64_t ldu, const bf16_t* dy,
                                                         int64_t lddy, bf16_t* du, int64_t lddu, int M, int F) {
    const int nv = F / 8;
    const int64_t total = (int64_t)M * nv;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / nv;
        const int c = (int)(e - r * nv) * 8;
        float a[8], g[8], d[8], da[8], dg[8];
        unpack8(*(const u32x4*)(u + r * ldu + c), a);
        unpack8(*(const u32x4*)(u + r * ldu + F + c), g);
        unpack8(*(const u32x4*)(dy + r * lddy + c), d);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            da[i] = d[i] * gelu_erf(g[i]);
            dg[i] = d[i] * a[i] * dgelu_erf(g[i]);
        }
        *(u32x4*)(du + r * lddu + c) = pack8(da);
        *(u32x4*)(du + r * lddu + F + c) = pack8(dg);
    }
}

// ---- out = a + b (+ c), row-strided bf16 -------------------------------------------------------
__global__ __launch_bounds__(256) void add_kernel(const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb,
                                                   const bf16_t* c, int64_t ldc, bf16_t* o, int64_t ldo, int M,
                                                   int C) {
    const int nv = C / 8;
    const int64_t total = (int64_t)M * nv;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / nv;
        const int col = (int)(e - r * nv) * 8;
        float x[8], y[8];
        unpack8(*(const u32x4*)(a + r * lda + col), x);
        unpack8(*(const u32x4*)(b + r * ldb + col), y);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] += y[i];
        if (c) {
            unpack8(*(const u32x4*)(c + r * ldc + col), y);
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] += y[i];
        }
        *(u32x4*)(o + r * ldo + col) = pack8(x);
    }
}

// ---- dgrad of nearest-2x upsample: dx[b][y][x][c] = sum of the 2x2 block of dy --------------------
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const bf16_t* dy, bf16_t* dx, int B, int H, int W,
                                                            int C) {
    const int nv = C / 8;
    const int64_t total = (int64_t)B * H * W * nv;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t pix = e / nv;
        const int col = (int)(e - pix * nv) * 8;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t[8];
#pragma unroll
        for (int dy_ = 0; dy_ < 2; ++dy_)
#pragma unroll
            for (int dx_ = 0; dx_ < 2; ++dx_) {
                const int64_t src = ((int64_t)(b * 2 * H + 2 * y + dy_) * (2 * W) + 2 * x + dx_) * C + col;
                unpack8(*(const u32x4*)(dy + src), t);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += t[i];
            }
        *(u32x4*)(dx + pix * C + col) = pack8(acc);
    }
}

// ---- conv_in: 3x3 pad 1, Cin (<=8) -> Cout, NCHW bf16 in, channels-last bf16 out -------------------
// weights fp32 [Cin][3][3][Cout] (Cout contiguous: 8 adjacent outputs = two float4 loads), bias fp32.
__global__ __launch_bounds__(256) void conv_in_kernel(const bf16_t* x, const float* w, const float* bias,
                                                       bf16_t* y, int B, int H, int W, int Cin, int Cout) {
    const int nv = Cout / 8;
    const int64_t total = (int64_t)B * H * W * nv;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t pix = e / nv;
        const int co = (int)(e - pix * nv) * 8;
        const int px = (int)(pix % W), py = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        float acc[8];
        {
            f32x4 b0 = *(const f32x4*)(bias + co), b1 = *(const f32x4*)(bias + co + 4);
            acc[0] = b0[0]; acc[1] = b0[1]; acc[2] = b0[2]; acc[3] = b0[3];
            acc[4] = b1[0]; acc[5] = b1[1]; acc[6] = b1[2]; acc[7] = b1[3];
        }
        for (int c = 0; c < Cin; ++c)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const float xv = bf2f(x[((int64_t)(b * Cin + c) * H + iy) * W + ix]);
                const float* wr = w + (int64_t)(c * 9 + tap) * Cout + co;
                f32x4 w0 = *(const f32x4*)wr, w1 = *(const f32x4*)(wr + 4);
                acc[0] += xv * w0[0]; acc[1] += xv * w0[1]; acc[2] += xv * w0[2]; acc[3] += xv * w0[3];
                acc[4] += xv * w1[0]; acc[5] += xv * w1[1]; acc[6] += xv * w1[2]; acc[7] += xv * w1[3];
            }
        *(u32x4*)(y + pix * Cout + co) = pack8(acc);
    }
}

// ---- conv_out: 3x3 pad 1, C -> Cout (4), channels-last bf16 in, NCHW fp32 out ------------------------
// one wave per output pixel; weights bf16 [Cout][3][3][C]; lanes split C in vectors of 8.
template <int COUT>
__global__ __launch_bounds__(256) void conv_out_kernel(const bf16_t* x, const bf16_t* w, const float* bias,
                                                        float* y, int B, int H, int W, int C) {
    const int lane = lane_id();
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t npix = (int64_t)B * H * W;
    const bool live = pix < npix;
    const int64_t pp = live ? pix : 0;
    const int px = (int)(pp % W), py = (int)((pp / W) % H), b = (int)(pp / ((int64_t)W * H));
    const int nv = C / 8;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;  // wave-uniform
        const bf16_t* xr = x + ((int64_t)(b * H + iy) * W + ix) * C;
        for (int v = lane; v < nv; v += 64) {
            float xv[8], wv[8];
            unpack8(*(const u32x4*)(xr + v * 8), xv);
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                unpack8(*(const u32x4*)(w + ((int64_t)(o * 9 + tap)) * C + v * 8), wv);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[o] += xv[i] * wv[i];
            }
        }
    }
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc[o] += shfl_xor(acc[o], m);
    }
    if (live && lane == 0) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) y[((int64_t)(b * COUT + o) * H + py) * W + px] = acc[o] + bias[o];
    }
}

// ---- conv_in / conv_out on the matrix cores (round 4) ---------------------------------------------------------------
// The two kernels above re-read their weights once per output element (conv_in: 72 16-byte loads per 8 outputs; conv_out:
// the whole 23 KB weight set per pixel) and cost 28 us each per UNet pass -- as much as a level-0 3x3 convolution with 80x
// the work.  Both are tiny GEMMs:
//   conv_in : [pixels][Cin * 9 (<= 64)] x [Cin * 9][Cout]   -- weights stationary in REGISTERS (fp32 split into bf16 hi + lo:
//             the result keeps the fp32-weight accuracy of the scalar kernel), a wave owns up to 5 fragments of 16 output
//             channels, the activation fragment is gathered from the NCHW input;
//   conv_out: [pixels][9 C] x [9 C][4 (one 16-row fragment, rows >= 4 zero)] -- K split over the 4 waves of a workgroup,
//             activation and weight fragments straight from global / L2, partial sums meet in LDS.
// mfma16(W, X, acc): lane (fr, fg) receives rows n = 4 fg + r of column (pixel) fr.
constexpr int CIN_FPW = 5;     // output fragments per wave (Cout <= 4 * 5 * 16)
__global__ __launch_bounds__(256) void conv_in_mfma_kernel(const bf16_t* x, const float* w, const float* bias, bf16_t* y, int B,
                                                            int H, int W, int Cin, int Cout, int groups_per_block) {
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6), fr = lane & 15, fg = lane >> 4;
    const int nf = Cout / 16, fpw = (nf + 3) / 4, f0 = wave * fpw, f1 = min(nf, f0 + fpw);
    const int K = Cin * 9;
    if (f0 >= f1) return;
    // weight fragments: k = 32 ks + 8 fg + i, row n = 16 f + fr
    bf16x8 whi[CIN_FPW][2], wlo[CIN_FPW][2];
    f32x4 bia[CIN_FPW];
#pragma unroll
    for (int q = 0; q < CIN_FPW; ++q) {
        const int f = f0 + q;
        const bool live = f < f1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                const int k = 32 * ks + 8 * fg + i;
                const float a = (live && k < K) ? w[(int64_t)k * Cout + 16 * f + fr] : 0.f;
                const float b = (live && k + 1 < K) ? w[(int64_t)(k + 1) * Cout + 16 * f + fr] : 0.f;
                const unsigned h2 = pack_bf2(a, b);
                hi[i >> 1] = h2;
                lo[i >> 1] = pack_bf2(a - bf2f((bf16_t)(h2 & 0xffffu)), b - bf2f((bf16_t)(h2 >> 16)));
            }
            whi[q][ks] = __builtin_bit_cast(bf16x8, hi);
            wlo[q][ks] = __builtin_bit_cast(bf16x8, lo);
        }
        bia[q] = live ? *(const f32x4*)(bias + 16 * f + 4 * fg) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int npix = B * H * W;                  // (host: < 2^31; 32-bit divisions only -- a 64-bit one is ~200 instructions)
    for (int g = 0; g < groups_per_block; ++g) {
        const int pix = ((int)blockIdx.x * groups_per_block + g) * 16 + fr;
        if (((int)blockIdx.x * groups_per_block + g) * 16 >= npix) break;
        const bool pv = pix < npix;
        const unsigned pp = pv ? (unsigned)pix : 0u;
        const unsigned row = pp / (unsigned)W;
        const int px = (int)(pp - row * (unsigned)W), b = (int)(row / (unsigned)H), py = (int)(row - (unsigned)b * (unsigned)H);
        bf16x8 xf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 t = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 32 * ks + 8 * fg + i;
                const int c = k / 9, tap = k - 9 * c;
                const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
                unsigned v = 0u;
                if (pv && k < K && iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[((int64_t)(b * Cin + c) * H + iy) * W + ix];
                t[i >> 1] |= v << ((i & 1) * 16);
            }
            xf[ks] = __builtin_bit_cast(bf16x8, t);
        }
#pragma unroll
        for (int q = 0; q < CIN_FPW; ++q) {
            const int f = f0 + q;
            if (f >= f1) break;
            f32x4 acc = bia[q];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                acc = mfma16(whi[q][ks], xf[ks], acc);
                acc = mfma16(wlo[q][ks], xf[ks], acc);
            }
            if (pv) {
                const u32x2 o = {pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3])};
                *(u32x2*)(y + (int64_t)pix * Cout + 16 * f + 4 * fg) = o;
            }
        }
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void conv_out_mfma_kernel(const bf16_t* x, const bf16_t* w, const float* bias, float* y, int B,
                                                             int H, int W, int C) {
    __shared__ f32x4 red[4][64];
    const int lane = lane_id(), wave = uniform((int)(threadIdx.x >> 6)), fr = lane & 15, fg = lane >> 4;
    const int npix = B * H * W, pix = (int)blockIdx.x * 16 + fr;       // (host: < 2^31; 32-bit divisions only)
    const bool pv = pix < npix;
    const unsigned pp = pv ? (unsigned)pix : 0u;
    const unsigned row = pp / (unsigned)W;
    const int px = (int)(pp - row * (unsigned)W), b = (int)(row / (unsigned)H), py = (int)(row - (unsigned)b * (unsigned)H);
    const int nkc = C / 32;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const u32x4 z = {0u, 0u, 0u, 0u};
    // k-steps are handed out in UNITS of (channel block, kernel row): the three horizontal taps of a unit read the same
    // cache lines, and the three kernel rows of a channel block are in flight on the workgroup's waves together -- the input
    // is fetched from L2 ~once instead of once per tap (9x: 94 MB per launch at level 0, the bound of the first version)
    constexpr int UU = 4;                                  // units (x 3 taps) in flight per wave
    const int nunits = 3 * nkc;
    for (int u0 = wave; u0 < nunits; u0 += 4 * UU) {
        u32x4 xv[UU][3], wv[UU][3];
#pragma unroll
        for (int uu = 0; uu < UU; ++uu) {
            const int unit = u0 + 4 * uu;
            const int cb = unit / 3, ky = unit - 3 * cb;
            const int iy = py + ky - 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = px + kx - 1;
                const bool ok = unit < nunits && pv && iy >= 0 && iy < H && ix >= 0 && ix < W;
                xv[uu][kx] = ok ? *(const u32x4*)(x + ((int64_t)(b * H + iy) * W + ix) * C + cb * 32 + fg * 8) : z;
                wv[uu][kx] = (unit < nunits && fr < COUT) ? *(const u32x4*)(w + (int64_t)(fr * 9 + ky * 3 + kx) * C + cb * 32 + fg * 8) : z;
            }
        }
#pragma unroll
        for (int uu = 0; uu < UU; ++uu)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                acc = mfma16(__builtin_bit_cast(bf16x8, wv[uu][kx]), __builtin_bit_cast(bf16x8, xv[uu][kx]), acc);
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && 4 * fg < COUT && pv) {
        const f32x4 a0 = red[0][lane], a1 = red[1][lane], a2 = red[2][lane], a3 = red[3][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 4 * fg + r;
            if (n < COUT) y[((int64_t)(b * COUT + n) * H + py) * W + px] = (a0[r] + a1[r]) + (a2[r] + a3[r]) + bias[n];
        }
    }
}

// dgrad of conv_out: dx[pix][c] = sum_{tap,o} dy[b][o][pix - off(tap)] * w[o][tap][c]
template <int COUT>
__global__ __launch_bounds__(256) void conv_out_bwd_kernel(const float* dy, const bf16_t* w, bf16_t* dx, int B,
                                                            int H, int W, int C) {
    const int nv = C / 8;
    const int64_t total = (int64_t)B * H * W * nv;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t pix = e / nv;
        const int c = (int)(e - pix * nv) * 8;
        const int px = (int)(pix % W), py = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int tap = 0; tap < 9; ++tap) {
            // output pixel (oy, ox) used input (py, px) through tap iff oy + kh - 1 = py
            const int oy = py - (tap / 3 - 1), ox = px - (tap % 3 - 1);
            if (oy < 0 || oy >= H || ox < 0 || ox >= W) continue;
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const float d = dy[((int64_t)(b * COUT + o) * H + oy) * W + ox];
                float wv[8];
                unpack8(*(const u32x4*)(w + ((int64_t)(o * 9 + tap)) * C + c), wv);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += d * wv[i];
            }
        }
        *(u32x4*)(dx + pix * C + c) = pack8(acc);
    }
}

// ---- timestep sinusoid: out[i] = [cos(t_i f) | sin(t_i f)], f_j = exp(-ln(1e4) j / half) -------------
__global__ void timestep_embedding_kernel(const float* t_table, const int* idx, int t_stride, int n, int dim,
                                          bf16_t* out) {
    const int half = dim / 2;
    const int base = idx ? *idx : 0;
    for (int e = (int)(blockIdx.x * blockDim.x + threadIdx.x); e < n * half; e += (int)(gridDim.x * blockDim.x)) {
        const int i = e / half, j = e - i * half;
        const float t = t_table[base + i * t_stride];
        const float f = expf(-9.210340371976184f * (float)j / (float)half);
        const float a = t * f;
        out[i * dim + j] = f2bf(cosf(a));
        out[i * dim + half + j] = f2bf(sinf(a));
    }
}

__global__ void advance_kernel(int* counter) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *counter += 1;
}

// ---- CFG combine + DDIM update (train_util.py:163-166,190; DDIM eta = 0 is linear in (x, model_out)) ----
// pred: fp32 [2*bs][n] raw UNet output (uncond half first); x: fp32 [bs][n] latents, updated in place;
// x2: bf16 [2*bs][n] next UNet input = cat([x]*2).  coef[step] = {c_x, c_e}.
__global__ __launch_bounds__(256) void cfg_ddim_kernel(const float* pred, float* x, bf16_t* x2, const float* coef,
                                                        const int* step, float guidance, int64_t half_n) {
    const int st = step ? *step : 0;
    const float cx = coef[2 * st], ce = coef[2 * st + 1];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < half_n; e += (int64_t)gridDim.x * 256) {
        const float u = pred[e], c = pred[half_n + e];
        const float eps = u + guidance * (c - u);
        const float xn = cx * x[e] + ce * eps;
        x[e] = xn;
        const bf16_t hb = f2bf(xn);
        x2[e] = hb;
        x2[half_n + e] = hb;
    }
}

// ---- CFG combine + generic linear scheduler update (DDPM / LMS / Euler-ancestral, model_util.py:247-274) --------
// row = coef + LECO_SCHED_ROW * step: {c_x, c_e, c_n, c_h1, c_h2, c_h3, s_in, d_x, d_e, ...}
//   x' = c_x x + c_e out + c_n noise + sum_j c_hj h_j ;  d = d_x x + d_e out, history shifted h1 <- d, h2 <- h1, h3 <- h2
//   x2 = bf16(s_in x') duplicated = the next UNet input (scale_model_input of the next step folded in)
__global__ __launch_bounds__(256) void cfg_sched_kernel(const float* pred, float* x, bf16_t* x2, const float* coef,
                                                         const int* step, float guidance, int64_t half_n,
                                                         const float* noise, float* hist, int n_hist) {
    const int st = step ? *step : 0;
    const float* r = coef + LECO_SCHED_ROW * st;
    const float cx = r[0], ce = r[1], cn = r[2], sin_ = r[6], dx = r[7], de = r[8];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < half_n; e += (int64_t)gridDim.x * 256) {
        const float out = pred ? pred[e] + guidance * (pred[half_n + e] - pred[e]) : 0.f;
        const float xo = x[e];
        float xn = cx * xo + ce * out;
        if (noise) xn += cn * noise[e];
        float hprev = dx * xo + de * out;
        for (int j = 0; j < n_hist; ++j) {   // oldest-last; shift while accumulating
            const float hj = hist[(int64_t)j * half_n + e];
            xn += r[3 + j] * hj;
            hist[(int64_t)j * half_n + e] = hprev;
            hprev = hj;
        }
        x[e] = xn;
        const bf16_t hb = f2bf(sin_ * xn);
        x2[e] = hb;
        x2[half_n + e] = hb;
    }
}

// ---- ESD loss (prompt_util.py:107-135 with MSELoss, train_lora.py:96,265-270) and d loss / d target-pass output --
// each *_pred is the raw fp32 [2*bs][n] UNet output; guided = u + g_pred*(c - u) (train_util.py:163-166).
// target_goal = neutral + sign * g_loss * (positive - unconditional); loss = mean((target - goal)^2).
// dpred[2*bs][n]: d loss / d raw target output = {(1-g_pred) * d, g_pred * d}, d = 2 (target-goal)/N.
// Up to ESD_BLOCKS blocks; each leaves its partial sum in g_esd_part and draws a ticket; the block that draws the last
// ticket adds the partials IN BLOCK ORDER, so the loss is bitwise reproducible although the blocks finish in any order
// (one 256-thread block alone serialised the eight fp32 streams of a >= 500k-element batch on one CU).  The scratch is
// per device: launches of this kernel must be stream-ordered (they are: one loss per step).
constexpr int ESD_BLOCKS = 64;
__device__ float g_esd_part[ESD_BLOCKS];
__device__ unsigned g_esd_ticket = 0;
// Every prediction is the guided combination u + g_pred (c - u) of the two halves of a CFG-doubled UNet output
// (train_util.py:163-166).  A NULL unconditional half (`*_u`) means the pass ran on the conditional samples only: the
// prediction IS c -- what the combination evaluates to at g_pred = 1, the value the reference passes for all four passes
// (train_lora.py:202-256), up to the fp32 rounding of u + (c - u).  That is the de-duplicated step of FusedStep.
__global__ __launch_bounds__(256) void esd_loss_kernel(const float* tgt_u, const float* tgt_c, const float* pos_u, const float* pos_c,
                                                        const float* neu_u, const float* neu_c, const float* unc_u, const float* unc_c,
                                                        float g_pred, float g_loss, float sign, int64_t half_n, float* loss,
                                                        float* dpred_u, float* dpred_c) {
    __shared__ float red[4];
    __shared__ bool last;
    float part = 0.f;
    const float inv_n = 1.f / (float)half_n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < half_n; e += (int64_t)gridDim.x * 256) {
        const float t = tgt_u ? tgt_u[e] + g_pred * (tgt_c[e] - tgt_u[e]) : tgt_c[e];
        const float p = pos_u ? pos_u[e] + g_pred * (pos_c[e] - pos_u[e]) : pos_c[e];
        const float n = neu_u ? neu_u[e] + g_pred * (neu_c[e] - neu_u[e]) : neu_c[e];
        const float u = unc_u ? unc_u[e] + g_pred * (unc_c[e] - unc_u[e]) : unc_c[e];
        const float goal = n + sign * g_loss * (p - u);
        const float diff = t - goal;
        part += diff * diff;
        if (dpred_c) {
            const float d = 2.f * diff * inv_n;
            if (dpred_u) dpred_u[e] = (1.f - g_pred) * d;
            dpred_c[e] = (tgt_u ? g_pred : 1.f) * d;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) part += shfl_xor(part, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mine = (red[0] + red[1]) + (red[2] + red[3]);
        if (gridDim.x == 1) {
            *loss = mine * inv_n;
            last = false;
        } else {
            g_esd_part[blockIdx.x] = mine;
            __threadfence();                                           // partial visible device-wide before the ticket
            last = atomicAdd(&g_esd_ticket, 1u) == gridDim.x - 1;
        }
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        float tot = 0.f;
        for (unsigned b = 0; b < gridDim.x; ++b) tot += ((volatile float*)g_esd_part)[b];   // fixed order
        *loss = tot * inv_n;
        g_esd_ticket = 0;                                              // re-armed for the next (stream-ordered) launch
    }
}

// ---- fused AdamW over the flat LoRA slab (torch.optim.AdamW semantics, train_lora.py:280) ----------------
// hyper (device fp32): {lr, bias_correction1, bias_correction2, grad_scale}
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, bf16_t* shadow,
                                                     const float* hyper, float beta1, float beta2, float eps,
                                                     float wd, int64_t n) {
    const float lr = hyper[0], bc1 = hyper[1], bc2 = hyper[2], gs = hyper[3];
    const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float gr = g[e] * gs;
        float pv = p[e] * (1.f - lr * wd);
        const float mm = beta1 * m[e] + (1.f - beta1) * gr;
        const float vv = beta2 * v[e] + (1.f - beta2) * gr * gr;
        const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        pv -= step_size * mm / denom;
        p[e] = pv;
        m[e] = mm;
        v[e] = vv;
        shadow[e] = f2bf(pv);
    }
}

// Lion (Chen et al. 2023; `lion_pytorch.Lion`, train_util.py:362-365): decoupled weight decay, sign of the
// interpolated momentum, then the momentum update with beta2.  hyper = {lr, -, -, grad_scale}.
__global__ __launch_bounds__(256) void lion_kernel(float* p, const float* g, float* m, bf16_t* shadow,
                                                    const float* hyper, float beta1, float beta2, float wd, int64_t n) {
    const float lr = hyper[0], gs = hyper[3];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float gr = g[e] * gs, mo = m[e];
        float pv = p[e] * (1.f - lr * wd);
        const float u = beta1 * mo + (1.f - beta1) * gr;
        pv -= lr * (u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f));
        p[e] = pv;
        m[e] = beta2 * mo + (1.f - beta2) * gr;
        shadow[e] = f2bf(pv);
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* x, bf16_t* y, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) y[e] = f2bf(x[e]);
}

// ---- LoRA operand packing (see leco_hip.h) -------------------------------------------------------------
__global__ __launch_bounds__(256) void lora_pack_kernel(const leco_lora_site* sites) {
    const leco_lora_site s = sites[blockIdx.y];
    const bool conv = s.taps == 9;     // 3x3 conv LoRA: lora_down is [r][Cin][3][3] (lora.py:72-81)
    const int R = s.groups * s.r, R16 = (R + 15) / 16 * 16, Rp = s.rp ? s.rp : (conv ? 64 : (R + 31) / 32 * 32);
    const int gn = s.n / s.groups, cin = conv ? s.k / 9 : s.k;
    const int rows_s = conv ? Rp : R16;   // conv sites use dn_s / up_t as GEMM weight operands of Rp rows
    const int64_t n0 = (int64_t)rows_s * s.k, n1 = (int64_t)s.n * Rp, n2 = (int64_t)rows_s * s.n, n3 = (int64_t)s.k * Rp;
    bf16_t* dn_s = (bf16_t*)s.dn_s;
    bf16_t* up_p = (bf16_t*)s.up_p;
    bf16_t* up_t = (bf16_t*)s.up_t;
    bf16_t* dn_p = (bf16_t*)s.dn_p;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n0 + n1 + n2 + n3; e += (int64_t)gridDim.x * 256) {
        if (e < n0) {  // dn_s[j][k]: linear k = column; conv k = tap * Cin + c  <-  down[j][c][tap]
            const int j = (int)(e / s.k), k = (int)(e - (int64_t)j * s.k);
            bf16_t val = 0;
            if (j < R) {
                const int src = conv ? (k % cin) * 9 + k / cin : k;
                val = ((const bf16_t*)s.down[j / s.r])[(int64_t)(j % s.r) * s.k + src];
            }
            dn_s[e] = val;
        } else if (e < n0 + n1) {  // up_p[n][j] = scale * up[g][n - g*gn][jj] if j in group(n)
            const int64_t t = e - n0;
            const int n = (int)(t / Rp), j = (int)(t - (int64_t)n * Rp), g = n / gn;
            float val = 0.f;
            if (j >= g * s.r && j < (g + 1) * s.r)
                val = s.scale * bf2f(((const bf16_t*)s.up[g])[(int64_t)(n - g * gn) * s.r + (j - g * s.r)]);
            up_p[t] = f2bf(val);
            if (s.up_pg) {   // LECO_ACT_GEGLU row interleave: value row f -> 128 (f/64) + f%64, gate row F + f -> ... + 64
                const int F = s.n / 2, f = n < F ? n : n - F;
                const int64_t row = (int64_t)(f / 64) * 128 + (f % 64) + (n < F ? 0 : 64);
                ((bf16_t*)s.up_pg)[row * Rp + j] = f2bf(val);
            }
        } else if (e < n0 + n1 + n2) {  // up_t[j][n] = up[g][n - g*gn][jj] if group(n) == j / r
            const int64_t t = e - n0 - n1;
            const int j = (int)(t / s.n), n = (int)(t - (int64_t)j * s.n), g = n / gn;
            bf16_t val = 0;
            if (j < R && j / s.r == g) val = ((const bf16_t*)s.up[g])[(int64_t)(n - g * gn) * s.r + (j % s.r)];
            up_t[t] = val;
        } else {
            // linear: dn_p[k][j] = scale * down[j][k]
            // conv:   dn_p[c][tap'][j] = scale * down[j][c][8 - tap']  (flipped, in/out swapped: the dgrad operand)
            const int64_t t = e - n0 - n1 - n2;
            const int kk = (int)(t / Rp), j = (int)(t - (int64_t)kk * Rp);
            float val = 0.f;
            if (j < R) {
                const int src = conv ? (kk / 9) * 9 + (8 - kk % 9) : kk;
                val = s.scale * bf2f(((const bf16_t*)s.down[j / s.r])[(int64_t)(j % s.r) * s.k + src]);
            }
            dn_p[t] = f2bf(val);
        }
    }
}

// ---- LoRA weight gradients: G[j][c] += scale * sum_m P[m][j] Q[m][c] -----------------------------------
// block = 256 threads over a 256-column tile (32 column vectors x 8 row lanes); blockIdx.y walks WG_ROWS-row slabs
// of M (P slab staged in LDS as fp32); 8 independent Q loads in flight per thread.  r <= 16.
constexpr int WG_ROWS = 128;
struct WgradConv {   // a_mode == LECO_A_PLAIN: Q row = m.  Otherwise Q row = source pixel of output row m for tap (kh, kw)
    int a_mode, h_out, w_out, h_in, w_in, kh, kw;
};
template <int R>
__device__ __forceinline__ void lora_wgrad_body(const bf16_t* P, int64_t ldp, const bf16_t* Q, int64_t ldq, float* G,
                                                int64_t g_sj, int64_t g_sc, int M, int r, int cols, float scale,
                                                WgradConv cv, float* part, int bx, int by) {
    // thread (vec = tid & 31, rl = tid >> 5): 8 adjacent columns (one 16-byte load per row) x every 8th row of
    // the slab; the 8 row lanes are then combined through LDS in a fixed order and one atomic per (j, column)
    // leaves the block.
    __shared__ float sp[WG_ROWS * R];
    __shared__ int srow[WG_ROWS];
    __shared__ f32x4 red[8 * 256];
    const int tid = (int)threadIdx.x;
    const int vec = tid & 31, rl = tid >> 5;
    const int c0 = bx * 256;
    const int c = c0 + vec * 8;
    const int m0 = by * WG_ROWS;
    const int rows = min(WG_ROWS, M - m0);
    for (int e = tid; e < WG_ROWS * R; e += 256) {
        const int mm = e / R, j = e - mm * R;
        sp[e] = (mm < rows && j < r) ? bf2f(P[(int64_t)(m0 + mm) * ldp + j]) : 0.f;
    }
    if (tid < WG_ROWS) {   // Q row feeding output row m0 + tid (-1: padding / outside)
        int src = -1;
        if (tid < rows) {
            const int m = m0 + tid;
            if (cv.a_mode == LECO_A_PLAIN) {
                src = m;
            } else {
                const int hw = cv.h_out * cv.w_out;
                const int b = m / hw, rem = m - b * hw, oy = rem / cv.w_out, ox = rem - oy * cv.w_out;
                const int sy = cv.a_mode == LECO_A_CONV3_S2 ? 2 : 1;
                const int dv = (cv.a_mode == LECO_A_CONV3_UP2) ? 1 : 0;
                const int uy = oy * sy + cv.kh - 1, ux = ox * sy + cv.kw - 1;
                if (uy >= 0 && uy < (cv.h_in << dv) && ux >= 0 && ux < (cv.w_in << dv))
                    src = (b * cv.h_in + (uy >> dv)) * cv.w_in + (ux >> dv);
            }
        }
        srow[tid] = src;
    }
    __syncthreads();
    float acc[8][R];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < R; ++j) acc[i][j] = 0.f;
    if (c < cols) {
        for (int mm = rl; mm < WG_ROWS; mm += 32) {   // rows beyond `rows` contribute 0 through sp / srow
            u32x4 qv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sr = srow[mm + 8 * u];
                qv[u] = sr >= 0 ? *(const u32x4*)(Q + (int64_t)sr * ldq + c) : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float q[8];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    q[2 * h] = bf2f((bf16_t)(qv[u][h] & 0xffffu));
                    q[2 * h + 1] = bf2f((bf16_t)(qv[u][h] >> 16));
                }
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const float pj = sp[(mm + 8 * u) * R + j];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i][j] += q[i] * pj;
                }
            }
        }
    }
#pragma unroll
    for (int jb = 0; jb < R; jb += 4) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i)
            red[rl * 256 + vec * 8 + i] = f32x4{acc[i][jb], acc[i][jb + 1], acc[i][jb + 2], acc[i][jb + 3]};
        __syncthreads();
        f32x4 t = red[tid];
#pragma unroll
        for (int l = 1; l < 8; ++l) {
            const f32x4 o = red[l * 256 + tid];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) t[jj] += o[jj];
        }
        if (c0 + tid < cols) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                if (jb + jj < r) {
                    if (part)   // deterministic mode: this slab's contribution, summed in slab order by the reduce kernel
                        part[((int64_t)by * r + jb + jj) * cols + c0 + tid] = t[jj] * scale;
                    else
                        atomicAdd(&G[(jb + jj) * g_sj + (int64_t)(c0 + tid) * g_sc], t[jj] * scale);
                }
        }
    }
}

template <int R>
__global__ __launch_bounds__(256) void lora_wgrad_kernel(const bf16_t* P, int64_t ldp, const bf16_t* Q, int64_t ldq,
                                                          float* G, int64_t g_sj, int64_t g_sc, int M, int r,
                                                          int cols, float scale, WgradConv cv, float* part) {
    lora_wgrad_body<R>(P, ldp, Q, ldq, G, g_sj, g_sc, M, r, cols, scale, cv, part, (int)blockIdx.x, (int)blockIdx.y);
}

// ALL LoRA weight gradients of a backward in one launch: block b finds its problem in the table (block_start is the
// running sum of blocks_x * ceil(m / 128); binary search) -- 384 launches at their ~10 us floor become one.
template <int R>
__global__ __launch_bounds__(256) void lora_wgrad_grouped_kernel(const leco_wgrad_problem* probs, int n) {
    const int b = (int)blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (probs[mid].block_start <= b) lo = mid; else hi = mid - 1;
    }
    const leco_wgrad_problem q = probs[lo];
    const int local = b - q.block_start;
    const int by = local / q.blocks_x, bx = local - by * q.blocks_x;
    lora_wgrad_body<R>((const bf16_t*)q.p, q.ldp, (const bf16_t*)q.q, q.ldq, q.g, q.g_sj, q.g_sc, q.m, q.r, q.cols, q.scale,
                       WgradConv{q.a_mode, q.h_out, q.w_out, q.h_in, q.w_in, q.kh, q.kw}, nullptr, bx, by);
}

// deterministic mode: G[j][c] += sum over the M slabs (in slab order) of part[slab][j][c]
__global__ __launch_bounds__(256) void lora_wgrad_reduce_kernel(const float* part, int nslabs, int r, int cols, float* G,
                                                                 int64_t g_sj, int64_t g_sc) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)r * cols) return;
    const int j = (int)(e / cols), c = (int)(e - (int64_t)j * cols);
    float acc = 0.f;
    for (int sl = 0; sl < nslabs; ++sl) acc += part[((int64_t)sl * r + j) * cols + c];
    G[j * g_sj + (int64_t)c * g_sc] += acc;
}

// ---- per-sample column sums: out[b][c] = sum over the rows of sample b of x[row][c] (fp32 out; d time-embedding bias)
__global__ __launch_bounds__(256) void rowgroup_sum_kernel(const bf16_t* x, int64_t ldx, float* out, int64_t ldo,
                                                            int rows_per_group, int cols) {
    const int c = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int b = (int)blockIdx.y;
    if (c >= cols) return;
    const bf16_t* src = x + (int64_t)b * rows_per_group * ldx + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int r = 0;
    for (; r + 4 <= rows_per_group; r += 4) {
        a0 += bf2f(src[(int64_t)r * ldx]); a1 += bf2f(src[(int64_t)(r + 1) * ldx]);
        a2 += bf2f(src[(int64_t)(r + 2) * ldx]); a3 += bf2f(src[(int64_t)(r + 3) * ldx]);
    }
    for (; r < rows_per_group; ++r) a0 += bf2f(src[(int64_t)r * ldx]);
    out[(int64_t)b * ldo + c] = (a0 + a1) + (a2 + a3);
}
}  // namespace
}  // namespace leco

using namespace leco;
#define LECO_STREAM ((hipStream_t)stream)

extern "C" int leco_geglu_fwd(const void* u, int64_t ldu, void* y, int64_t ldy, int32_t m, int32_t f,
                              leco_stream_t stream) {
    if (f % 8) return fail(-EINVAL, "geglu: F=%d %% 8 != 0", f);
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for((int64_t)m * f / 8)), dim3(256), 0, LECO_STREAM,
                       (const bf16_t*)u, ldu, (bf16_t*)y, ldy, m, f);
    return check_launch("leco_geglu_fwd");
}
extern "C" int leco_geglu_bwd(const void* u, int64_t ldu, const void* dy, int64_t lddy, void* du, int64_t lddu,
                              int32_t m, int32_t f, leco_stream_t stream) {
    if (f % 8) return fail(-EINVAL, "geglu: F=%d %% 8 != 0", f);
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for((int64_t)m * f / 8)), dim3(256), 0, LECO_STREAM,
                       (const bf16_t*)u, ldu, (const bf16_t*)dy, lddy, (bf16_t*)du, lddu, m, f);
    return check_launch("leco_geglu_bwd");
}
extern "C" int leco_add(const void* a, int64_t lda, const void* b, int64_t ldb, const void* c, int64_t ldc,
                        void* out, int64_t ldo, int32_t m, int32_t cols, leco_stream_t stream) {
    if (cols % 8) return fail(-EINVAL, "add: cols=%d %% 8 != 0", cols);
    hipLaunchKernelGGL(add_kernel, dim3(grid_for((int64_t)m * cols / 8)), dim3(256), 0, LECO_STREAM,
                       (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (const bf16_t*)c, ldc, (bf16_t*)out, ldo, m, cols);
    return check_launch("leco_add");
}
extern "C" int leco_upsample2x_bwd(const void* dy, void* dx, int32_t batch, int32_t h, int32_t w, int32_t c,
                                   leco_stream_t stream) {
    if (c % 8) return fail(-EINVAL, "upsample_bwd: C=%d %% 8 != 0", c);
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for((int64_t)batch * h * w * c / 8)), dim3(256), 0, LECO_STREAM,
                       (const bf16_t*)dy, (bf16_t*)dx, batch, h, w, c);
    return check_launch("leco_upsample2x_bwd");
}
extern "C" int leco_conv_in(const void* x, const float* w, const float* bias, void* y, int32_t batch, int32_t h,
                            int32_t wd, int32_t cin, int32_t cout, leco_stream_t stream) {
    if (cout % 8) return fail(-EINVAL, "conv_in: Cout=%d %% 8 != 0", cout);
    if (cout % 16 == 0 && cout <= 64 * CIN_FPW && cin * 9 <= 64 && (int64_t)batch * h * wd < (1ll << 30)) {
        const int64_t groups = ((int64_t)batch * h * wd + 15) / 16;
        const int gpb = groups >= 2048 ? 4 : (groups >= 512 ? 2 : 1);       // >= 256 workgroups where the problem has them
        hipLaunchKernelGGL(conv_in_mfma_kernel, dim3((unsigned)((groups + gpb - 1) / gpb)), dim3(256), 0, LECO_STREAM,
                           (const bf16_t*)x, w, bias, (bf16_t*)y, batch, h, wd, cin, cout, gpb);
        return check_launch("leco_conv_in");
    }
    hipLaunchKernelGGL(conv_in_kernel, dim3(grid_for((int64_t)batch * h * wd * cout / 8)), dim3(256), 0, LECO_STREAM,
                       (const bf16_t*)x, w, bias, (bf16_t*)y, batch, h, wd, cin, cout);
    return check_launch("leco_conv_in");
}
extern "C" int leco_conv_out(const void* x, const void* w, const float* bias, float* y, int32_t batch, int32_t h,
                             int32_t wd, int32_t c, int32_t cout, leco_stream_t stream) {
    if (cout != 4 || c % 8) return fail(-EINVAL, "conv_out: needs Cout=4 (got %d), C %% 8 == 0", cout);
    const int64_t npix = (int64_t)batch * h * wd;
    if (c % 32 == 0 && npix < (1ll << 30)) {
        hipLaunchKernelGGL((conv_out_mfma_kernel<4>), dim3((unsigned)((npix + 15) / 16)), dim3(256), 0, LECO_STREAM,
                           (const bf16_t*)x, (const bf16_t*)w, bias, y, batch, h, wd, c);
        return check_launch("leco_conv_out");
    }
    hipLaunchKernelGGL((conv_out_kernel<4>), dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, LECO_STREAM,
                       (const bf16_t*)x, (const bf16_t*)w, bias, y, batch, h, wd, c);
    return check_launch("leco_conv_out");
}
extern "C" int leco_conv_out_bwd(const float* dy, const void* w, void* dx, int32_t batch, int32_t h, int32_t wd,
                                 int32_t c, int32_t cout, leco_stream_t stream) {
    if (cout != 4 || c % 8) return fail(-EINVAL, "conv_out_bwd: needs Cout=4, C %% 8 == 0");
    hipLaunchKernelGGL((conv_out_bwd_kernel<4>), dim3(grid_for((int64_t)batch * h * wd * c / 8)), dim3(256), 0,
                       LECO_STREAM, dy, (const bf16_t*)w, (bf16_t*)dx, batch, h, wd, c);
    return check_launch("leco_conv_out_bwd");
}
extern "C" int leco_timestep_embedding(const float* t_table, const int32_t* idx, int32_t t_stride, int32_t n,
                                       int32_t dim, void* out, leco_stream_t stream) {
    if (dim % 2) return fail(-EINVAL, "timestep_embedding: odd dim");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(cdiv((long)n * dim / 2, 256)), dim3(256), 0, LECO_STREAM,
                       t_table, (const int*)idx, t_stride, n, dim, (bf16_t*)out);
    return check_launch("leco_timestep_embedding");
}
extern "C" int leco_advance(int32_t* counter, leco_stream_t stream) {
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, LECO_STREAM, (int*)counter);
    return check_launch("leco_advance");
}
extern "C" int leco_cfg_ddim_step(const float* pred, float* x, void* x2, const float* coef, const int32_t* step,
                                  float guidance, int64_t half_n, leco_stream_t stream) {
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid_for(half_n)), dim3(256), 0, LECO_STREAM, pred, x, (bf16_t*)x2,
                       coef, (const int*)step, guidance, half_n);
    return check_launch("leco_cfg_ddim_step");
}
extern "C" int leco_cfg_sched_step(const float* pred, float* x, void* x2, const float* coef, const int32_t* step,
                                   float guidance, int64_t half_n, const float* noise, float* hist, int32_t n_hist,
                                   leco_stream_t stream) {
    if (n_hist < 0 || n_hist > 3 || (n_hist && !hist)) return fail(-EINVAL, "cfg_sched_step: n_hist=%d needs 0..3 history slabs", n_hist);
    hipLaunchKernelGGL(cfg_sched_kernel, dim3(grid_for(half_n)), dim3(256), 0, LECO_STREAM, pred, x, (bf16_t*)x2, coef,
                       (const int*)step, guidance, half_n, noise, hist, n_hist);
    return check_launch("leco_cfg_sched_step");
}
static int esd_launch(const float* tu, const float* tc, const float* pu, const float* pc, const float* nu, const float* nc,
                      const float* uu, const float* uc, float g_pred, float g_loss, float sign, int64_t half_n, float* loss,
                      float* du, float* dc, leco_stream_t stream, const char* what) {
    // one workgroup (bs*4*h*w = 32 k .. 64 k elements): the loss is reduced in a fixed order -- bitwise reproducible
    const int64_t want = (half_n + 4095) / 4096;       // >= 16 elements of each stream per thread
    const int blocks = (int)(want < 1 ? 1 : (want > ESD_BLOCKS ? ESD_BLOCKS : want));
    hipLaunchKernelGGL(esd_loss_kernel, dim3(blocks), dim3(256), 0, LECO_STREAM, tu, tc, pu, pc, nu, nc, uu, uc, g_pred, g_loss,
                       sign, half_n, loss, du, dc);
    return check_launch(what);
}
extern "C" int leco_esd_loss(const float* tgt, const float* pos, const float* neu, const float* unc, float g_pred,
                             float g_loss, float sign, int64_t half_n, float* loss, float* dpred,
                             leco_stream_t stream) {
    return esd_launch(tgt, tgt + half_n, pos, pos + half_n, neu, neu + half_n, unc, unc + half_n, g_pred, g_loss, sign, half_n, loss,
                      dpred, dpred ? dpred + half_n : nullptr, stream, "leco_esd_loss");
}
extern "C" int leco_esd_loss_cond(const float* tgt_c, const float* pos_c, const float* neu_c, const float* unc_c, float g_loss,
                                  float sign, int64_t half_n, float* loss, float* dpred_c, leco_stream_t stream) {
    if (!tgt_c || !pos_c || !neu_c || !unc_c || !loss || half_n <= 0) return fail(-EINVAL, "leco_esd_loss_cond: null operand");
    return esd_launch(nullptr, tgt_c, nullptr, pos_c, nullptr, neu_c, nullptr, unc_c, 1.f, g_loss, sign, half_n, loss, nullptr,
                      dpred_c, stream, "leco_esd_loss_cond");
}
extern "C" int leco_adamw(float* p, const float* g, float* m, float* v, void* shadow, const float* hyper,
                          float beta1, float beta2, float eps, float wd, int64_t n, leco_stream_t stream) {
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, LECO_STREAM, p, g, m, v, (bf16_t*)shadow,
                       hyper, beta1, beta2, eps, wd, n);
    return check_launch("leco_adamw");
}
extern "C" int leco_lion(float* p, const float* g, float* m, void* shadow, const float* hyper, float beta1,
                         float beta2, float wd, int64_t n, leco_stream_t stream) {
    hipLaunchKernelGGL(lion_kernel, dim3(grid_for(n)), dim3(256), 0, LECO_STREAM, p, g, m, (bf16_t*)shadow, hyper, beta1,
                       beta2, wd, n);
    return check_launch("leco_lion");
}
extern "C" int leco_cast_f32_bf16(const float* x, void* y, int64_t n, leco_stream_t stream) {
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, LECO_STREAM, x, (bf16_t*)y, n);
    return check_launch("leco_cast_f32_bf16");
}
// dst[r * bytes + i] = src[i], r < reps (16-byte granules)
__global__ __launch_bounds__(256) void repeat_kernel(const u32x4* src, u32x4* dst, int64_t n16, int reps) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n16; e += (int64_t)gridDim.x * 256) {
        const u32x4 v = src[e];
        for (int r = 0; r < reps; ++r) dst[(int64_t)r * n16 + e] = v;
    }
}
extern "C" int leco_repeat(const void* src, void* dst, int64_t bytes, int32_t reps, leco_stream_t stream) {
    if (!src || !dst || bytes <= 0 || bytes % 16 || reps <= 0) return fail(-EINVAL, "leco_repeat: bytes=%lld (multiple of 16), reps=%d", (long long)bytes, reps);
    hipLaunchKernelGGL(repeat_kernel, dim3(grid_for(bytes / 16)), dim3(256), 0, LECO_STREAM, (const u32x4*)src, (u32x4*)dst, bytes / 16, reps);
    return check_launch("leco_repeat");
}
// ---- step glue (leco_amd/train.py::FusedStep.step): the tiny tensor copies between the launch plans of one optimizer step
// (train_lora.py:175-199: initial latents -> cat([latents] * 2); denoised latents -> the inputs of the four remaining
// passes + their timestep) as TWO launches instead of ~15 framework copies with a 10-25 us bubble each.
template <typename T>
__global__ __launch_bounds__(256) void step_begin_kernel(const float* x, T* x2, float scale, int64_t half_n, int* t_idx) {
    if (t_idx && blockIdx.x == 0 && threadIdx.x == 0) *t_idx = 0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < half_n; e += (int64_t)gridDim.x * 256) {
        const float v = x[e] * scale;
        T o;
        if constexpr (sizeof(T) == 4) o = v; else o = f2bf(v);
        x2[e] = o;
        x2[half_n + e] = o;
    }
}
extern "C" int leco_step_begin(const float* x, void* x2, int32_t x2_is_f32, float scale, int64_t half_n, int32_t* t_idx,
                               leco_stream_t stream) {
    if (!x || !x2 || half_n <= 0) return fail(-EINVAL, "leco_step_begin: half_n=%lld", (long long)half_n);
    if (x2_is_f32)
        hipLaunchKernelGGL(step_begin_kernel<float>, dim3(grid_for(half_n)), dim3(256), 0, LECO_STREAM, x, (float*)x2, scale, half_n, (int*)t_idx);
    else
        hipLaunchKernelGGL(step_begin_kernel<bf16_t>, dim3(grid_for(half_n)), dim3(256), 0, LECO_STREAM, x, (bf16_t*)x2, scale, half_n, (int*)t_idx);
    return check_launch("leco_step_begin");
}
__global__ __launch_bounds__(256) void step_mid_kernel(const u32x4* src, u32x4* dst_a, u32x4* dst_b, int64_t n16, int reps_b,
                                                        float t_cur, float* t_a, float* t_b, int* i_a, int* i_b, int slot) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (t_a) *t_a = t_cur;
        if (t_b) *t_b = t_cur;
        if (i_a) *i_a = slot;
        if (i_b) *i_b = slot;
    }
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n16; e += (int64_t)gridDim.x * 256) {
        const u32x4 v = src[e];
        if (dst_a) dst_a[e] = v;
        for (int r = 0; r < reps_b; ++r) dst_b[(int64_t)r * n16 + e] = v;
    }
}
extern "C" int leco_step_mid(const void* src, void* dst_a, void* dst_b, int64_t bytes, int32_t reps_b, float t_cur,
                             float* t_slot_a, float* t_slot_b, int32_t* t_idx_a, int32_t* t_idx_b, int32_t slot,
                             leco_stream_t stream) {
    if (!src || bytes <= 0 || bytes % 16 || reps_b < 0 || (reps_b && !dst_b))
        return fail(-EINVAL, "leco_step_mid: bytes=%lld (multiple of 16), reps_b=%d", (long long)bytes, reps_b);
    hipLaunchKernelGGL(step_mid_kernel, dim3(grid_for(bytes / 16)), dim3(256), 0, LECO_STREAM, (const u32x4*)src, (u32x4*)dst_a,
                       (u32x4*)dst_b, bytes / 16, reps_b, t_cur, t_slot_a, t_slot_b, (int*)t_idx_a, (int*)t_idx_b, slot);
    return check_launch("leco_step_mid");
}
// byte fill as a KERNEL node.  (hipMemsetAsync captured into a hipGraph becomes a memset node; round 6 met a replayed graph
// -- the de-duplicated frozen pass, a second plan of the denoising plan's exact shape -- whose memset node zeroed the
// GroupNorm statistics arena on the first replay only: profiles/r06_graph_memset_node.txt.)
__global__ __launch_bounds__(256) void fill_kernel(unsigned* p, unsigned v, int64_t n4, unsigned char* tail, int ntail, unsigned char b) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) p[e] = v;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = b;
}
extern "C" int leco_memset(void* p, int32_t value, int64_t bytes, leco_stream_t stream) {
    if (!p || bytes < 0 || ((uintptr_t)p & 3)) return fail(-EINVAL, "leco_memset: pointer must be 4-byte aligned");
    if (bytes == 0) return 0;
    const unsigned b = (unsigned)value & 0xffu, v = b * 0x01010101u;
    const int64_t n4 = bytes / 4;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(256), 0, LECO_STREAM, (unsigned*)p, v, n4,
                       (unsigned char*)p + n4 * 4, (int)(bytes - n4 * 4), (unsigned char)b);
    return check_launch("leco_memset");
}
extern "C" int leco_lora_pack(const leco_lora_site* sites, int32_t nsites, leco_stream_t stream) {
    if (nsites <= 0) return 0;
    hipLaunchKernelGGL(lora_pack_kernel, dim3(64, (unsigned)nsites), dim3(256), 0, LECO_STREAM, sites);
    return check_launch("leco_lora_pack");
}
static int wgrad_launch(const void* p, int64_t ldp, const void* q, int64_t ldq, float* g, int64_t g_sj, int64_t g_sc,
                        int32_t m, int32_t r, int32_t cols, float scale, WgradConv cv, float* part, int64_t part_bytes,
                        hipStream_t s) {
    if (r <= 0 || r > 16) return fail(-EINVAL, "lora_wgrad: rank %d unsupported (1..16)", r);
    const dim3 grid(cdiv(cols, 256), cdiv(m, WG_ROWS));
    if (part && (int64_t)grid.y * r * cols * (int64_t)sizeof(float) > part_bytes)
        return fail(-EINVAL, "lora_wgrad: deterministic mode needs %lld workspace bytes",
                    (long long)((int64_t)grid.y * r * cols * (int64_t)sizeof(float)));
    if (r <= 4)
        hipLaunchKernelGGL((lora_wgrad_kernel<4>), grid, dim3(256), 0, s, (const bf16_t*)p, ldp, (const bf16_t*)q, ldq, g,
                           g_sj, g_sc, m, r, cols, scale, cv, part);
    else if (r <= 8)
        hipLaunchKernelGGL((lora_wgrad_kernel<8>), grid, dim3(256), 0, s, (const bf16_t*)p, ldp, (const bf16_t*)q, ldq, g,
                           g_sj, g_sc, m, r, cols, scale, cv, part);
    else
        hipLaunchKernelGGL((lora_wgrad_kernel<16>), grid, dim3(256), 0, s, (const bf16_t*)p, ldp, (const bf16_t*)q, ldq, g,
                           g_sj, g_sc, m, r, cols, scale, cv, part);
    if (part)
        hipLaunchKernelGGL(lora_wgrad_reduce_kernel, dim3(cdiv((long)r * cols, 256)), dim3(256), 0, s, (const float*)part,
                           (int)grid.y, r, cols, g, g_sj, g_sc);
    return check_launch("leco_lora_wgrad");
}
extern "C" int leco_lora_wgrad(const void* p, int64_t ldp, const void* q, int64_t ldq, float* g, int64_t g_sj,
                               int64_t g_sc, int32_t m, int32_t r, int32_t cols, float scale, float* part,
                               int64_t part_bytes, leco_stream_t stream) {
    return wgrad_launch(p, ldp, q, ldq, g, g_sj, g_sc, m, r, cols, scale, WgradConv{LECO_A_PLAIN, 0, 0, 0, 0, 0, 0}, part,
                        part_bytes, LECO_STREAM);
}
extern "C" int leco_lora_wgrad_conv(const void* p, int64_t ldp, const void* q, int64_t ldq, float* g, int64_t g_sj,
                                    int64_t g_sc, int32_t m, int32_t r, int32_t cols, float scale, int32_t a_mode,
                                    int32_t h_out, int32_t w_out, int32_t h_in, int32_t w_in, int32_t kh, int32_t kw,
                                    float* part, int64_t part_bytes, leco_stream_t stream) {
    if (a_mode < LECO_A_CONV3_S1 || a_mode > LECO_A_CONV3_UP2) return fail(-EINVAL, "lora_wgrad_conv: bad a_mode %d", a_mode);
    return wgrad_launch(p, ldp, q, ldq, g, g_sj, g_sc, m, r, cols, scale, WgradConv{a_mode, h_out, w_out, h_in, w_in, kh, kw},
                        part, part_bytes, LECO_STREAM);
}
extern "C" int leco_lora_wgrad_grouped(const leco_wgrad_problem* problems, int32_t nproblems, int32_t total_blocks,
                                       int32_t max_rank, leco_stream_t stream) {
    if (nproblems <= 0 || total_blocks <= 0) return 0;
    if (max_rank <= 0 || max_rank > 16) return fail(-EINVAL, "lora_wgrad_grouped: rank %d unsupported (1..16)", max_rank);
    const dim3 grid((unsigned)total_blocks);
    if (max_rank <= 4) hipLaunchKernelGGL((lora_wgrad_grouped_kernel<4>), grid, dim3(256), 0, LECO_STREAM, problems, nproblems);
    else if (max_rank <= 8) hipLaunchKernelGGL((lora_wgrad_grouped_kernel<8>), grid, dim3(256), 0, LECO_STREAM, problems, nproblems);
    else hipLaunchKernelGGL((lora_wgrad_grouped_kernel<16>), grid, dim3(256), 0, LECO_STREAM, problems, nproblems);
    return check_launch("leco_lora_wgrad_grouped");
}
extern "C" int leco_rowgroup_sum(const void* x, int64_t ldx, float* out, int64_t ldo, int32_t groups,
                                 int32_t rows_per_group, int32_t cols, leco_stream_t stream) {
    hipLaunchKernelGGL(rowgroup_sum_kernel, dim3(cdiv(cols, 256), groups), dim3(256), 0, LECO_STREAM, (const bf16_t*)x, ldx,
                       out, ldo, rows_per_group, cols);
    return check_launch("leco_rowgroup_sum");
}
