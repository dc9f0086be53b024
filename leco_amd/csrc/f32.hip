// fp32 compute mode: the kernels of a UNet pass / training step with fp32 activations, fp32 weights and exact fp32
// contractions (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate -- the 157 TFLOP/s matrix path of gfx950, 1/16 of the
// bf16 rate; cdna_hip_programming.md section 3).
//
// Why it exists: the reference honours `train.precision: float32` (config_util.py:75-83, train_lora.py:54-67; its one
// published number is an fp32 run) and `north_star` states parity as <= 1e-3 relative on the predicted noise.  With bf16
// activations no implementation reaches that (one bf16 rounding is 2^-9; a pass rounds ~600 tensors): the bf16 MFMA path
// sits at 5e-3 .. 9e-3 of the fp32 oracle.  This mode is the arithmetic the reference runs for such configs; it is
// selected by `train.precision: float32` / `UNet2DConditionModel.to(dtype=torch.float32)` and is what the <= 1e-3
// whole-UNet parity test runs.  It is written for exactness and brevity, not for the roofline: one kernel per op,
// plain grid-stride loops, no fusion beyond the GEMM epilogue -- the bf16 files are the performance path.
//
// Same C-ABI argument lists as the bf16 entry points (leco_hip.h), `leco_f32_` prefix; every activation / weight /
// LoRA-operand pointer is `float*`.
#include <errno.h>
#include <hip/hip_runtime.h>
#include <leco_prims.h>
#include <math.h>

#include "common.h"

#define LECO_STREAM ((hipStream_t)stream)

namespace leco {
namespace {

inline int grid1(int64_t n, int per = 256) {
    int64_t g = (n + per - 1) / per;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}
__device__ __forceinline__ float silu_f(float z) { return z / (1.f + expf(-z)); }
__device__ __forceinline__ float dsilu_f(float z) {
    const float s = 1.f / (1.f + expf(-z));
    return s * (1.f + z * (1.f - s));
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float dgelu_f(float x) {
    return 0.5f * (1.f + erff(x * 0.7071067811865476f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

// ------------------------------------------------------------------------------------------------------------------
// GEMM / implicit-GEMM convolution.  64 x 64 tile, 4 waves (2 x 2), each 32 x 32 = 2 x 2 MFMA tiles of 16 x 16 x 4.
// ------------------------------------------------------------------------------------------------------------------
constexpr int FBK = 16;

// address of 4 consecutive k of activation row m (never straddles a tap or a source: cin, k_split % 4 == 0), or nullptr
// for padding.  Gathers as gemm.hip: stride 1 / stride 2 / nearest-2x upsampled input / transposed stride 2.
__device__ __forceinline__ const float* a_ptr(const leco_gemm_args& p, int m, int k0) {
    if (m >= p.m) return nullptr;
    if (k0 >= p.k) {   // LoRA K-extension columns
        const int j = k0 - p.k;
        return j < p.ext_k ? (const float*)p.a_ext + (int64_t)m * p.ld_aext + j : nullptr;
    }
    const int ksp = p.a1 ? p.k_split : 0x7fffffff;
    if (p.a_mode == LECO_A_PLAIN)
        return k0 < ksp ? (const float*)p.a0 + (int64_t)m * p.lda0 + k0 : (const float*)p.a1 + (int64_t)m * p.lda1 + (k0 - ksp);
    const int cin = p.k / 9, tap = k0 / cin, c = k0 - tap * cin, kh = tap / 3, kw = tap - kh * 3;
    const int hw = p.h_out * p.w_out, b = m / hw, rem = m - b * hw, oy = rem / p.w_out, ox = rem - oy * p.w_out;
    const int sy = p.a_mode == LECO_A_CONV3_S2 ? 2 : 1;
    const int dv = (p.a_mode == LECO_A_CONV3_UP2 || p.a_mode == LECO_A_CONV3_TR2) ? 1 : 0;
    const int odd = p.a_mode == LECO_A_CONV3_TR2 ? 1 : 0;
    const int uy = oy * sy + kh - 1, ux = ox * sy + kw - 1;
    if (uy < 0 || uy >= (p.h_in << dv) || ux < 0 || ux >= (p.w_in << dv) || (uy & odd) || (ux & odd)) return nullptr;
    const int64_t pix = (int64_t)(b * p.h_in + (uy >> dv)) * p.w_in + (ux >> dv);
    return c < ksp ? (const float*)p.a0 + pix * p.lda0 + c : (const float*)p.a1 + pix * p.lda1 + (c - ksp);
}
__device__ __forceinline__ const float* w_ptr(const leco_gemm_args& p, int n, int k0) {
    if (n >= p.n) return nullptr;
    if (k0 >= p.k) {
        const int j = k0 - p.k;
        return j < p.ext_k ? (const float*)p.w_ext + (int64_t)n * p.ld_wext + j : nullptr;
    }
    return (const float*)p.w + (int64_t)n * p.ldw + k0;
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(const leco_gemm_args p) {
    __shared__ float As[64][FBK + 1];
    __shared__ float Ws[64][FBK + 1];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = (int)blockIdx.y * 64, n0 = (int)blockIdx.x * 64;
    const int lr = tid >> 2, lk = (tid & 3) * 4;              // tile element this thread stages
    const int fr = lane & 15, fk = lane >> 4;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ktot = p.k + ((p.a_ext && p.ext_k > 0) ? (p.ext_k + FBK - 1) / FBK * FBK : 0);
    for (int k0 = 0; k0 < ktot; k0 += FBK) {
        const float* ap = a_ptr(p, m0 + lr, k0 + lk);
        const float* wp = w_ptr(p, n0 + lr, k0 + lk);
        const f32x4 av = ap ? *(const f32x4*)ap : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 wv = wp ? *(const f32x4*)wp : f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { As[lr][lk + q] = av[q]; Ws[lr][lk + q] = wv[q]; }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < FBK; ks += 4) {
            float a[2], w[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[wm * 32 + i * 16 + fr][ks + fk];
#pragma unroll
            for (int j = 0; j < 2; ++j) w[j] = Ws[wn * 32 + j * 16 + fr][ks + fk];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16x4_f32(a[i], w[j], acc[i][j]);
        }
    }
    // lane holds D[m = 4 fk + r][n = fr] of each 16 x 16 tile
    float* cp = (float*)p.c;
    const float* res = (const float*)p.residual;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + i * 16 + 4 * fk + r, n = n0 + wn * 32 + j * 16 + fr;
                if (m >= p.m || n >= p.n) continue;
                float v = acc[i][j][r];
                if (p.bias) v += p.bias[n];
                if (p.rowbias) v += p.rowbias[(int64_t)(m / p.rows_per_group) * p.ld_rowbias + n];
                if (res) v += res[(int64_t)m * p.ldr + n];
                if (p.act == LECO_ACT_SILU) v = silu_f(v);
                if (cp) cp[(int64_t)m * p.ldc + n] = v;
                if (p.c_f32) p.c_f32[(int64_t)m * p.ldc32 + n] = v;
            }
}

// ------------------------------------------------------------------------------------------------------------------
// Attention: softmax(scale Q K^T) V per (batch, head), online softmax, fp32 throughout.  One block = 64 queries;
// thread (row = tid >> 2, part = tid & 3): 8 of the 32 scores of a key tile and every 4th output dimension.
// ------------------------------------------------------------------------------------------------------------------
constexpr int AQ = 64, AK = 32, AD_MAX = 160;

__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* q, int64_t ldq, int64_t bsq, const float* k, int64_t ldk,
                                                            int64_t bsk, const float* v, int64_t ldv, int64_t bsv, float* o,
                                                            int64_t ldo, int64_t bso, float* lse, int H, int sq, int skv, int d,
                                                            float scale) {
    float* lds = (float*)dyn_lds();
    float* Qs = lds;                       // [AQ][d]
    float* Ks = Qs + AQ * d;               // [AK][d + 1]
    float* Vs = Ks + AK * (d + 1);         // [AK][d]
    float* Ps = Vs + AK * d;               // [AQ][AK + 1]
    const int tid = (int)threadIdx.x, row = tid >> 2, part = tid & 3;
    const int b = (int)blockIdx.z, h = (int)blockIdx.y, q0 = (int)blockIdx.x * AQ;
    const float* qb = q + b * bsq + h * d;
    const float* kb = k + b * bsk + h * d;
    const float* vb = v + b * bsv + h * d;
    for (int e = tid; e < AQ * d; e += 256) {
        const int r = e / d, c = e - r * d;
        Qs[e] = q0 + r < sq ? qb[(int64_t)(q0 + r) * ldq + c] : 0.f;
    }
    float oacc[AD_MAX / 4];
#pragma unroll
    for (int i = 0; i < AD_MAX / 4; ++i) oacc[i] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    const int nd = d / 4;
    for (int k0 = 0; k0 < skv; k0 += AK) {
        __syncthreads();
        for (int e = tid; e < AK * d; e += 256) {
            const int r = e / d, c = e - r * d;
            const bool ok = k0 + r < skv;
            Ks[r * (d + 1) + c] = ok ? kb[(int64_t)(k0 + r) * ldk + c] : 0.f;
            Vs[r * d + c] = ok ? vb[(int64_t)(k0 + r) * ldv + c] : 0.f;
        }
        __syncthreads();
        float s[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) s[c] = 0.f;
        for (int dd = 0; dd < d; ++dd) {
            const float qv = Qs[row * d + dd];
#pragma unroll
            for (int c = 0; c < 8; ++c) s[c] += qv * Ks[(part * 8 + c) * (d + 1) + dd];
        }
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            s[c] = k0 + part * 8 + c < skv ? s[c] * scale : -INFINITY;
            mx = fmaxf(mx, s[c]);
        }
        mx = fmaxf(mx, shfl_xor(mx, 1));
        mx = fmaxf(mx, shfl_xor(mx, 2));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = mrun == -INFINITY ? 0.f : expf(mrun - mnew);
        float ps = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float pv = s[c] == -INFINITY ? 0.f : expf(s[c] - mnew);
            Ps[row * (AK + 1) + part * 8 + c] = pv;
            ps += pv;
        }
        ps += shfl_xor(ps, 1);
        ps += shfl_xor(ps, 2);
        lrun = lrun * alpha + ps;
        mrun = mnew;
        __syncthreads();
        for (int i = 0; i < nd; ++i) {
            const int dd = part + 4 * i;
            float a = oacc[i] * alpha;
            for (int c = 0; c < AK; ++c) a += Ps[row * (AK + 1) + c] * Vs[c * d + dd];
            oacc[i] = a;
        }
    }
    if (q0 + row < sq) {
        const float inv = 1.f / lrun;
        float* ob = o + b * bso + h * d + (int64_t)(q0 + row) * ldo;
        for (int i = 0; i < nd; ++i) ob[part + 4 * i] = oacc[i] * inv;
        if (part == 0) lse[((int64_t)b * H + h) * sq + q0 + row] = mrun + logf(lrun);
    }
}

// delta[b][h][q] = sum_d dO O
__global__ __launch_bounds__(256) void attn_delta_f32_kernel(const float* o, int64_t ldo, int64_t bso, const float* d_o, int64_t lddo,
                                                              int64_t bsdo, float* delta, int B, int H, int sq, int d) {
    const int64_t total = (int64_t)B * H * sq;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int qi = (int)(e % sq), h = (int)((e / sq) % H), b = (int)(e / ((int64_t)sq * H));
        const float* op = o + b * bso + (int64_t)qi * ldo + h * d;
        const float* dp = d_o + b * bsdo + (int64_t)qi * lddo + h * d;
        float a = 0.f;
        for (int c = 0; c < d; ++c) a += op[c] * dp[c];
        delta[e] = a;
    }
}

// One kernel for both gradient passes.  ROWS_ARE_Q: block rows = queries, streamed tiles = keys -> dQ;
// otherwise block rows = keys, streamed tiles = queries -> dK, dV.  Thread (row, part) as in the forward.
template <bool ROWS_ARE_Q>
__global__ __launch_bounds__(256) void attn_bwd_f32_kernel(const float* q, int64_t ldq, int64_t bsq, const float* k, int64_t ldk,
                                                            int64_t bsk, const float* v, int64_t ldv, int64_t bsv, const float* d_o,
                                                            int64_t lddo, int64_t bsdo, const float* lse, const float* delta,
                                                            float* g0, int64_t ldg0, int64_t bsg0, float* g1, int64_t ldg1,
                                                            int64_t bsg1, int H, int sq, int skv, int d, float scale) {
    float* lds = (float*)dyn_lds();
    float* R0 = lds;                       // [AQ][d]      rows: Q (dQ pass) or K (dK/dV pass)
    float* R1 = R0 + AQ * d;               // [AQ][d]      rows: dO (dQ pass) or V (dK/dV pass)
    float* T0 = R1 + AQ * d;               // [AK][d + 1]  tile: K  or Q
    float* T1 = T0 + AK * (d + 1);         // [AK][d + 1]  tile: V  or dO
    float* Ps = T1 + AK * (d + 1);         // [AQ][AK + 1] p
    float* Ds = Ps + AQ * (AK + 1);        // [AQ][AK + 1] ds
    float* tl = Ds + AQ * (AK + 1);        // [AK] lse of the tile's queries (dK/dV pass)
    float* td = tl + AK;                   // [AK] delta
    const int tid = (int)threadIdx.x, row = tid >> 2, part = tid & 3;
    const int b = (int)blockIdx.z, h = (int)blockIdx.y, r0 = (int)blockIdx.x * AQ;
    const float* qb = q + b * bsq + h * d;
    const float* kb = k + b * bsk + h * d;
    const float* vb = v + b * bsv + h * d;
    const float* dob = d_o + b * bsdo + h * d;
    const float* lb = lse + ((int64_t)b * H + h) * sq;
    const float* db = delta + ((int64_t)b * H + h) * sq;
    const int nrows = ROWS_ARE_Q ? sq : skv, ntile = ROWS_ARE_Q ? skv : sq;
    for (int e = tid; e < AQ * d; e += 256) {
        const int r = e / d, c = e - r * d;
        const bool ok = r0 + r < nrows;
        if (ROWS_ARE_Q) {
            R0[e] = ok ? qb[(int64_t)(r0 + r) * ldq + c] : 0.f;
            R1[e] = ok ? dob[(int64_t)(r0 + r) * lddo + c] : 0.f;
        } else {
            R0[e] = ok ? kb[(int64_t)(r0 + r) * ldk + c] : 0.f;
            R1[e] = ok ? vb[(int64_t)(r0 + r) * ldv + c] : 0.f;
        }
    }
    const float my_lse = (ROWS_ARE_Q && r0 + row < sq) ? lb[r0 + row] : 0.f;
    const float my_delta = (ROWS_ARE_Q && r0 + row < sq) ? db[r0 + row] : 0.f;
    float a0[AD_MAX / 4], a1[AD_MAX / 4];
#pragma unroll
    for (int i = 0; i < AD_MAX / 4; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
    const int nd = d / 4;
    for (int t0 = 0; t0 < ntile; t0 += AK) {
        __syncthreads();
        for (int e = tid; e < AK * d; e += 256) {
            const int r = e / d, c = e - r * d;
            const bool ok = t0 + r < ntile;
            if (ROWS_ARE_Q) {
                T0[r * (d + 1) + c] = ok ? kb[(int64_t)(t0 + r) * ldk + c] : 0.f;
                T1[r * (d + 1) + c] = ok ? vb[(int64_t)(t0 + r) * ldv + c] : 0.f;
            } else {
                T0[r * (d + 1) + c] = ok ? qb[(int64_t)(t0 + r) * ldq + c] : 0.f;
                T1[r * (d + 1) + c] = ok ? dob[(int64_t)(t0 + r) * lddo + c] : 0.f;
            }
        }
        if (!ROWS_ARE_Q && tid < AK) {
            tl[tid] = t0 + tid < sq ? lb[t0 + tid] : 0.f;
            td[tid] = t0 + tid < sq ? db[t0 + tid] : 0.f;
        }
        __syncthreads();
        // s = <R0[row], T0[c]> (q.k either way), dp = <dO, V>: dQ pass <R1[row], T1[c]>, dK/dV pass <T1[c], R1[row]>
        float s[8], dp[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { s[c] = 0.f; dp[c] = 0.f; }
        for (int dd = 0; dd < d; ++dd) {
            const float x0 = R0[row * d + dd], x1 = R1[row * d + dd];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                s[c] += x0 * T0[(part * 8 + c) * (d + 1) + dd];
                dp[c] += x1 * T1[(part * 8 + c) * (d + 1) + dd];
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int tc = part * 8 + c;
            const bool ok = (t0 + tc < ntile) && (r0 + row < nrows);
            const float l = ROWS_ARE_Q ? my_lse : tl[tc], dl = ROWS_ARE_Q ? my_delta : td[tc];
            const float pv = ok ? expf(s[c] * scale - l) : 0.f;
            Ps[row * (AK + 1) + tc] = pv;
            Ds[row * (AK + 1) + tc] = pv * (dp[c] - dl) * scale;
        }
        __syncthreads();
        for (int i = 0; i < nd; ++i) {
            const int dd = part + 4 * i;
            float x = a0[i], y = a1[i];
            for (int c = 0; c < AK; ++c) {
                // dQ pass: dQ += ds K.   dK/dV pass: dK += ds Q, dV += p dO
                x += Ds[row * (AK + 1) + c] * T0[c * (d + 1) + dd];
                if (!ROWS_ARE_Q) y += Ps[row * (AK + 1) + c] * T1[c * (d + 1) + dd];
            }
            a0[i] = x;
            a1[i] = y;
        }
    }
    if (r0 + row < nrows) {
        float* p0 = g0 + b * bsg0 + h * d + (int64_t)(r0 + row) * ldg0;
        for (int i = 0; i < nd; ++i) p0[part + 4 * i] = a0[i];
        if (!ROWS_ARE_Q) {
            float* p1 = g1 + b * bsg1 + h * d + (int64_t)(r0 + row) * ldg1;
            for (int i = 0; i < nd; ++i) p1[part + 4 * i] = a1[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// GroupNorm / LayerNorm.  GroupNorm: one block per (sample, group), two-pass statistics; stats[b][g] = {mean, rstd}.
// ------------------------------------------------------------------------------------------------------------------
struct GnSrcF {
    const float* x0;
    const float* x1;
    int64_t ld0, ld1;
    int c0;
};
__device__ __forceinline__ float gn_at(const GnSrcF& s, int64_t row, int c) {
    return c < s.c0 ? s.x0[row * s.ld0 + c] : s.x1[row * s.ld1 + (c - s.c0)];
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}
__global__ __launch_bounds__(256) void gn_fwd_f32_kernel(GnSrcF src, const float* gamma, const float* beta, int hw, int C, int G,
                                                          float eps, int act, float* stats, float* y, int64_t ldy) {
    __shared__ float red[8];
    const int b = (int)blockIdx.y, g = (int)blockIdx.x, cg = C / G, n = hw * cg;
    float s = 0.f;
    for (int e = (int)threadIdx.x; e < n; e += 256) s += gn_at(src, (int64_t)b * hw + e / cg, g * cg + e % cg);
    const float mean = block_sum(s, red) / (float)n;
    float s2 = 0.f;
    for (int e = (int)threadIdx.x; e < n; e += 256) {
        const float dlt = gn_at(src, (int64_t)b * hw + e / cg, g * cg + e % cg) - mean;
        s2 += dlt * dlt;
    }
    const float rstd = rsqrtf(block_sum(s2, red) / (float)n + eps);
    if (threadIdx.x == 0) { stats[(b * G + g) * 2] = mean; stats[(b * G + g) * 2 + 1] = rstd; }
    for (int e = (int)threadIdx.x; e < n; e += 256) {
        const int c = g * cg + e % cg;
        const int64_t row = (int64_t)b * hw + e / cg;
        const float z = (gn_at(src, row, c) - mean) * rstd * gamma[c] + beta[c];
        y[row * ldy + c] = act == LECO_ACT_SILU ? silu_f(z) : z;
    }
}
// dx = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)), dxhat = dy act'(z) gamma
__global__ __launch_bounds__(256) void gn_bwd_f32_kernel(GnSrcF src, const float* dy, int64_t lddy, const float* gamma,
                                                          const float* beta, const float* stats, int hw, int C, int G, int act,
                                                          float* dx, int64_t lddx) {
    __shared__ float red[8];
    const int b = (int)blockIdx.y, g = (int)blockIdx.x, cg = C / G, n = hw * cg;
    const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
    float s1 = 0.f, s2 = 0.f;
    for (int e = (int)threadIdx.x; e < n; e += 256) {
        const int c = g * cg + e % cg;
        const int64_t row = (int64_t)b * hw + e / cg;
        const float xh = (gn_at(src, row, c) - mean) * rstd;
        float d = dy[row * lddy + c];
        if (act == LECO_ACT_SILU) d *= dsilu_f(xh * gamma[c] + beta[c]);
        d *= gamma[c];
        s1 += d;
        s2 += d * xh;
    }
    const float m1 = block_sum(s1, red) / (float)n;
    const float m2 = block_sum(s2, red) / (float)n;
    for (int e = (int)threadIdx.x; e < n; e += 256) {
        const int c = g * cg + e % cg;
        const int64_t row = (int64_t)b * hw + e / cg;
        const float xh = (gn_at(src, row, c) - mean) * rstd;
        float d = dy[row * lddy + c];
        if (act == LECO_ACT_SILU) d *= dsilu_f(xh * gamma[c] + beta[c]);
        d *= gamma[c];
        dx[row * lddx + c] = rstd * (d - m1 - xh * m2);
    }
}
// LayerNorm: one wave per row
__global__ __launch_bounds__(256) void ln_fwd_f32_kernel(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                                          int M, int C, float* y, int64_t ldy, float* mean, float* rstd) {
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * ldx;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += shfl_xor(s, m);
    const float mu = s / (float)C;
    float s2 = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mu; s2 += d * d; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s2 += shfl_xor(s2, m);
    const float rs = rsqrtf(s2 / (float)C + eps);
    for (int c = lane; c < C; c += 64) y[row * ldy + c] = (xr[c] - mu) * rs * gamma[c] + beta[c];
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}
__global__ __launch_bounds__(256) void ln_bwd_f32_kernel(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* gamma,
                                                          const float* mean, const float* rstd, const float* dres, int64_t ldres,
                                                          int M, int C, float* dx, int64_t lddx) {
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = dy[row * lddy + c] * gamma[c], xh = (x[row * ldx + c] - mu) * rs;
        s1 += d;
        s2 += d * xh;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { s1 += shfl_xor(s1, m); s2 += shfl_xor(s2, m); }
    const float m1 = s1 / (float)C, m2 = s2 / (float)C;
    for (int c = lane; c < C; c += 64) {
        const float d = dy[row * lddy + c] * gamma[c], xh = (x[row * ldx + c] - mu) * rs;
        float o = rs * (d - m1 - xh * m2);
        if (dres) o += dres[row * ldres + c];
        dx[row * lddx + c] = o;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// elementwise family
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void geglu_fwd_f32_kernel(const float* u, int64_t ldu, float* y, int64_t ldy, int M, int F) {
    const int64_t total = (int64_t)M * F;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / F;
        const int c = (int)(e - r * F);
        y[r * ldy + c] = u[r * ldu + c] * gelu_f(u[r * ldu + F + c]);
    }
}
__global__ __launch_bounds__(256) void geglu_bwd_f32_kernel(const float* u, int64_t ldu, const float* dy, int64_t lddy, float* du,
                                                             int64_t lddu, int M, int F) {
    const int64_t total = (int64_t)M * F;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / F;
        const int c = (int)(e - r * F);
        const float a = u[r * ldu + c], g = u[r * ldu + F + c], d = dy[r * lddy + c];
        du[r * lddu + c] = d * gelu_f(g);
        du[r * lddu + F + c] = d * a * dgelu_f(g);
    }
}
__global__ __launch_bounds__(256) void add_f32_kernel(const float* a, int64_t lda, const float* b, int64_t ldb, const float* c,
                                                       int64_t ldc, float* out, int64_t ldo, int M, int N) {
    const int64_t total = (int64_t)M * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / N;
        const int col = (int)(e - r * N);
        float v = a[r * lda + col] + b[r * ldb + col];
        if (c) v += c[r * ldc + col];
        out[r * ldo + col] = v;
    }
}
__global__ __launch_bounds__(256) void upsample_bwd_f32_kernel(const float* dy, float* dx, int B, int H, int W, int C) {
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t pix = e / C;
        const int col = (int)(e - pix * C);
        const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        float a = 0.f;
        for (int j = 0; j < 4; ++j)
            a += dy[((int64_t)(b * 2 * H + 2 * y + (j >> 1)) * (2 * W) + 2 * x + (j & 1)) * C + col];
        dx[e] = a;
    }
}
// conv_in: NCHW fp32 -> channels-last fp32; w fp32 [Cin][3][3][Cout]
__global__ __launch_bounds__(256) void conv_in_f32_kernel(const float* x, const float* w, const float* bias, float* y, int B, int H,
                                                           int W, int Cin, int Cout) {
    const int64_t total = (int64_t)B * H * W * Cout;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t pix = e / Cout;
        const int co = (int)(e - pix * Cout);
        const int px = (int)(pix % W), py = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        float a = bias[co];
        for (int c = 0; c < Cin; ++c)
            for (int tap = 0; tap < 9; ++tap) {
                const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                a += x[((int64_t)(b * Cin + c) * H + iy) * W + ix] * w[(int64_t)(c * 9 + tap) * Cout + co];
            }
        y[e] = a;
    }
}
// conv_out: channels-last fp32 -> NCHW fp32; w fp32 [Cout][3][3][C]; one wave per output pixel
__global__ __launch_bounds__(256) void conv_out_f32_kernel(const float* x, const float* w, const float* bias, float* y, int B, int H,
                                                            int W, int C, int Cout) {
    const int lane = lane_id();
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= (int64_t)B * H * W) return;
    const int px = (int)(pix % W), py = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    for (int o = 0; o < Cout; ++o) {
        float a = 0.f;
        for (int tap = 0; tap < 9; ++tap) {
            const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const float* xr = x + ((int64_t)(b * H + iy) * W + ix) * C;
            const float* wr = w + (int64_t)(o * 9 + tap) * C;
            for (int c = lane; c < C; c += 64) a += xr[c] * wr[c];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) a += shfl_xor(a, m);
        if (lane == 0) y[((int64_t)(b * Cout + o) * H + py) * W + px] = a + bias[o];
    }
}
__global__ __launch_bounds__(256) void conv_out_bwd_f32_kernel(const float* dy, const float* w, float* dx, int B, int H, int W, int C,
                                                                int Cout) {
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t pix = e / C;
        const int c = (int)(e - pix * C);
        const int px = (int)(pix % W), py = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        float a = 0.f;
        for (int tap = 0; tap < 9; ++tap) {
            const int oy = py - (tap / 3 - 1), ox = px - (tap % 3 - 1);
            if (oy < 0 || oy >= H || ox < 0 || ox >= W) continue;
            for (int o = 0; o < Cout; ++o)
                a += dy[((int64_t)(b * Cout + o) * H + oy) * W + ox] * w[(int64_t)(o * 9 + tap) * C + c];
        }
        dx[e] = a;
    }
}
__global__ void timestep_f32_kernel(const float* t_table, const int* idx, int t_stride, int n, int dim, float* out) {
    const int half = dim / 2;
    const int base = idx ? *idx : 0;
    for (int e = (int)(blockIdx.x * blockDim.x + threadIdx.x); e < n * half; e += (int)(gridDim.x * blockDim.x)) {
        const int i = e / half, j = e - i * half;
        const float t = t_table[base + i * t_stride];
        const float a = t * expf(-9.210340371976184f * (float)j / (float)half);
        out[i * dim + j] = cosf(a);
        out[i * dim + half + j] = sinf(a);
    }
}
// CFG combine + scheduler update with an fp32 next-input (the bf16 kernels round x2); DDIM = the two-coefficient row
__global__ __launch_bounds__(256) void cfg_step_f32_kernel(const float* pred, float* x, float* x2, const float* coef, const int* step,
                                                            float guidance, int64_t half_n, const float* noise, float* hist,
                                                            int n_hist, int generic) {
    const int st = step ? *step : 0;
    const float* r = coef + (generic ? LECO_SCHED_ROW : 2) * st;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < half_n; e += (int64_t)gridDim.x * 256) {
        const float out = pred ? pred[e] + guidance * (pred[half_n + e] - pred[e]) : 0.f;
        const float xo = x[e];
        float xn = r[0] * xo + r[1] * out, sin_ = 1.f;
        if (generic) {
            sin_ = r[6];
            if (noise) xn += r[2] * noise[e];
            float hprev = r[7] * xo + r[8] * out;
            for (int j = 0; j < n_hist; ++j) {
                const float hj = hist[(int64_t)j * half_n + e];
                xn += r[3 + j] * hj;
                hist[(int64_t)j * half_n + e] = hprev;
                hprev = hj;
            }
        }
        x[e] = xn;
        x2[e] = sin_ * xn;
        x2[half_n + e] = sin_ * xn;
    }
}
__global__ __launch_bounds__(256) void copy_f32_kernel(const float* x, float* y, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) y[e] = x[e];
}
__global__ __launch_bounds__(256) void rowgroup_sum_f32_kernel(const float* x, int64_t ldx, float* out, int64_t ldo, int groups, int rpg,
                                                                int cols) {
    const int64_t total = (int64_t)groups * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int g = (int)(e / cols), c = (int)(e - (int64_t)g * cols);
        float a = 0.f;
        for (int r = 0; r < rpg; ++r) a += x[((int64_t)g * rpg + r) * ldx + c];
        out[(int64_t)g * ldo + c] = a;
    }
}
// LoRA operand images (leco_hip.h: dn_s, up_p, up_t, dn_p) in fp32 from the fp32 master slab
__global__ __launch_bounds__(256) void lora_pack_f32_kernel(const leco_lora_site* sites) {
    const leco_lora_site s = sites[blockIdx.y];
    const bool conv = s.taps == 9;
    const int R = s.groups * s.r, R16 = (R + 15) / 16 * 16, Rp = s.rp ? s.rp : (conv ? 64 : (R + 31) / 32 * 32);
    const int gn = s.n / s.groups, cin = conv ? s.k / 9 : s.k;
    const int rows_s = conv ? Rp : R16;
    const int64_t n0 = (int64_t)rows_s * s.k, n1 = (int64_t)s.n * Rp, n2 = (int64_t)rows_s * s.n, n3 = (int64_t)s.k * Rp;
    float* dn_s = (float*)s.dn_s;
    float* up_p = (float*)s.up_p;
    float* up_t = (float*)s.up_t;
    float* dn_p = (float*)s.dn_p;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n0 + n1 + n2 + n3; e += (int64_t)gridDim.x * 256) {
        if (e < n0) {
            const int j = (int)(e / s.k), k = (int)(e - (int64_t)j * s.k);
            float val = 0.f;
            if (j < R) val = ((const float*)s.down[j / s.r])[(int64_t)(j % s.r) * s.k + (conv ? (k % cin) * 9 + k / cin : k)];
            dn_s[e] = val;
        } else if (e < n0 + n1) {
            const int64_t t = e - n0;
            const int n = (int)(t / Rp), j = (int)(t - (int64_t)n * Rp), g = n / gn;
            float val = 0.f;
            if (j >= g * s.r && j < (g + 1) * s.r) val = s.scale * ((const float*)s.up[g])[(int64_t)(n - g * gn) * s.r + (j - g * s.r)];
            up_p[t] = val;
        } else if (e < n0 + n1 + n2) {
            const int64_t t = e - n0 - n1;
            const int j = (int)(t / s.n), n = (int)(t - (int64_t)j * s.n), g = n / gn;
            float val = 0.f;
            if (j < R && j / s.r == g) val = ((const float*)s.up[g])[(int64_t)(n - g * gn) * s.r + (j % s.r)];
            up_t[t] = val;
        } else {
            const int64_t t = e - n0 - n1 - n2;
            const int kk = (int)(t / Rp), j = (int)(t - (int64_t)kk * Rp);
            float val = 0.f;
            if (j < R) val = s.scale * ((const float*)s.down[j / s.r])[(int64_t)(j % s.r) * s.k + (conv ? (kk / 9) * 9 + (8 - kk % 9) : kk)];
            dn_p[t] = val;
        }
    }
}
// LoRA weight gradient: G[j g_sj + c g_sc] += scale sum_m P[m][j] Q[row(m)][c]; one thread per (j, c), fixed order
// (bitwise reproducible).  Q rows gathered like a conv tap when a_mode != PLAIN.
__global__ __launch_bounds__(256) void lora_wgrad_f32_kernel(const float* P, int64_t ldp, const float* Q, int64_t ldq, float* G,
                                                              int64_t g_sj, int64_t g_sc, int M, int r, int cols, float scale, int a_mode,
                                                              int h_out, int w_out, int h_in, int w_in, int kh, int kw) {
    const int64_t total = (int64_t)r * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e / cols), c = (int)(e - (int64_t)j * cols);
        float a = 0.f;
        for (int m = 0; m < M; ++m) {
            int src = m;
            if (a_mode != LECO_A_PLAIN) {
                const int hw = h_out * w_out, b = m / hw, rem = m - b * hw, oy = rem / w_out, ox = rem - oy * w_out;
                const int sy = a_mode == LECO_A_CONV3_S2 ? 2 : 1, dv = a_mode == LECO_A_CONV3_UP2 ? 1 : 0;
                const int uy = oy * sy + kh - 1, ux = ox * sy + kw - 1;
                if (uy < 0 || uy >= (h_in << dv) || ux < 0 || ux >= (w_in << dv)) continue;
                src = (b * h_in + (uy >> dv)) * w_in + (ux >> dv);
            }
            a += P[(int64_t)m * ldp + j] * Q[(int64_t)src * ldq + c];
        }
        G[j * g_sj + (int64_t)c * g_sc] += scale * a;
    }
}

int f32_gemm_validate(const leco_gemm_args& a) {
    if (a.m <= 0 || a.n <= 0 || a.k <= 0) return fail(-EINVAL, "leco_f32_gemm: empty problem");
    if (!a.a0 || !a.w || (!a.c && !a.c_f32)) return fail(-EINVAL, "leco_f32_gemm: null operand / no output");
    if (a.t_w) return fail(-EINVAL, "leco_f32_gemm: the fused down-projection (t_w) is a bf16-path feature");
    if (a.act != LECO_ACT_NONE && a.act != LECO_ACT_SILU) return fail(-EINVAL, "leco_f32_gemm: act %d unsupported", a.act);
    const int cin = a.a_mode == LECO_A_PLAIN ? a.k : a.k / 9;
    if (a.k % 4 || cin % 4 || (a.a1 && a.k_split % 4) || (a.a_mode != LECO_A_PLAIN && a.k % 9))
        return fail(-EINVAL, "leco_f32_gemm: k / channel counts must be multiples of 4 (k=%d)", a.k);
    if ((a.lda0 | a.ldw | (a.a1 ? a.lda1 : 0) | (a.a_ext ? (a.ld_aext | a.ld_wext | a.ext_k) : 0)) % 4)
        return fail(-EINVAL, "leco_f32_gemm: operand strides must keep 16-byte alignment");
    if (a.a_ext && !a.w_ext) return fail(-EINVAL, "leco_f32_gemm: a_ext without w_ext");
    if (a.a_mode < LECO_A_PLAIN || a.a_mode > LECO_A_CONV3_TR2) return fail(-EINVAL, "leco_f32_gemm: bad a_mode");
    if (a.a_mode != LECO_A_PLAIN && (int64_t)a.batch * a.h_out * a.w_out != a.m) return fail(-EINVAL, "leco_f32_gemm: conv m mismatch");
    if (a.rowbias && a.rows_per_group <= 0) return fail(-EINVAL, "leco_f32_gemm: rowbias needs rows_per_group");
    return 0;
}
size_t attn_lds_fwd(int d) { return (size_t)(AQ * d + AK * (d + 1) + AK * d + AQ * (AK + 1)) * 4; }
size_t attn_lds_bwd(int d) { return (size_t)(2 * AQ * d + 2 * AK * (d + 1) + 2 * AQ * (AK + 1) + 2 * AK) * 4; }
template <typename K>
void set_lds(K kernel, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
}  // namespace
}  // namespace leco

using namespace leco;

extern "C" int leco_f32_gemm(const leco_gemm_args* a, leco_stream_t stream) {
    if (!a) return fail(-EINVAL, "leco_f32_gemm: null args");
    const int rc = f32_gemm_validate(*a);
    if (rc) return rc;
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(cdiv(a->n, 64), cdiv(a->m, 64)), dim3(256), 0, LECO_STREAM, *a);
    return check_launch("leco_f32_gemm");
}
extern "C" int leco_f32_attention_fwd(const void* q, int64_t ldq, int64_t bsq, const void* k, int64_t ldk, int64_t bsk, const void* v,
                                      int64_t ldv, int64_t bsv, void* o, int64_t ldo, int64_t bso, float* lse, int32_t batch,
                                      int32_t heads, int32_t sq, int32_t skv, int32_t d, float scale, leco_stream_t stream) {
    if (d % 4 || d > AD_MAX || d <= 0) return fail(-EINVAL, "leco_f32_attention: head_dim %d unsupported (multiple of 4, <= 160)", d);
    set_lds(attn_fwd_f32_kernel, attn_lds_fwd(d));
    hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(cdiv(sq, AQ), heads, batch), dim3(256), attn_lds_fwd(d), LECO_STREAM, (const float*)q,
                       ldq, bsq, (const float*)k, ldk, bsk, (const float*)v, ldv, bsv, (float*)o, ldo, bso, lse, heads, sq, skv, d, scale);
    return check_launch("leco_f32_attention_fwd");
}
extern "C" int leco_f32_attention_bwd(const void* q, int64_t ldq, int64_t bsq, const void* k, int64_t ldk, int64_t bsk, const void* v,
                                      int64_t ldv, int64_t bsv, const void* o, int64_t ldo, int64_t bso, const void* d_o, int64_t lddo,
                                      int64_t bsdo, const float* lse, float* delta, void* dq, int64_t lddq, int64_t bsdq, void* dk,
                                      int64_t lddk, int64_t bsdk, void* dv, int64_t lddv, int64_t bsdv, int32_t batch, int32_t heads,
                                      int32_t sq, int32_t skv, int32_t d, float scale, leco_stream_t stream) {
    if (d % 4 || d > AD_MAX || d <= 0) return fail(-EINVAL, "leco_f32_attention: head_dim %d unsupported", d);
    hipLaunchKernelGGL(attn_delta_f32_kernel, dim3(grid1((int64_t)batch * heads * sq)), dim3(256), 0, LECO_STREAM, (const float*)o, ldo,
                       bso, (const float*)d_o, lddo, bsdo, delta, batch, heads, sq, d);
    set_lds(attn_bwd_f32_kernel<true>, attn_lds_bwd(d));
    set_lds(attn_bwd_f32_kernel<false>, attn_lds_bwd(d));
    hipLaunchKernelGGL(attn_bwd_f32_kernel<true>, dim3(cdiv(sq, AQ), heads, batch), dim3(256), attn_lds_bwd(d), LECO_STREAM,
                       (const float*)q, ldq, bsq, (const float*)k, ldk, bsk, (const float*)v, ldv, bsv, (const float*)d_o, lddo, bsdo,
                       lse, (const float*)delta, (float*)dq, lddq, bsdq, (float*)nullptr, (int64_t)0, (int64_t)0, heads, sq, skv, d, scale);
    hipLaunchKernelGGL(attn_bwd_f32_kernel<false>, dim3(cdiv(skv, AQ), heads, batch), dim3(256), attn_lds_bwd(d), LECO_STREAM,
                       (const float*)q, ldq, bsq, (const float*)k, ldk, bsk, (const float*)v, ldv, bsv, (const float*)d_o, lddo, bsdo,
                       lse, (const float*)delta, (float*)dk, lddk, bsdk, (float*)dv, lddv, bsdv, heads, sq, skv, d, scale);
    return check_launch("leco_f32_attention_bwd");
}
extern "C" int leco_f32_groupnorm_fwd(const void* x0, int64_t ld0, const void* x1, int64_t ld1, int32_t c0, const float* gamma,
                                      const float* beta, int32_t batch, int32_t hw, int32_t c, int32_t groups, float eps, int32_t act,
                                      float* stats, void* y, int64_t ldy, leco_stream_t stream) {
    if (groups <= 0 || c % groups) return fail(-EINVAL, "leco_f32_groupnorm: c=%d groups=%d", c, groups);
    GnSrcF s{(const float*)x0, (const float*)x1, ld0, ld1, x1 ? c0 : c};
    hipLaunchKernelGGL(gn_fwd_f32_kernel, dim3(groups, batch), dim3(256), 0, LECO_STREAM, s, gamma, beta, hw, c, groups, eps, act, stats,
                       (float*)y, ldy);
    return check_launch("leco_f32_groupnorm_fwd");
}
extern "C" int leco_f32_groupnorm_bwd(const void* x0, int64_t ld0, const void* x1, int64_t ld1, int32_t c0, const void* dy, int64_t lddy,
                                      const float* gamma, const float* beta, const float* stats, int32_t batch, int32_t hw, int32_t c,
                                      int32_t groups, float eps, int32_t act, float* bstats, void* dx, int64_t lddx,
                                      leco_stream_t stream) {
    (void)eps; (void)bstats;
    GnSrcF s{(const float*)x0, (const float*)x1, ld0, ld1, x1 ? c0 : c};
    hipLaunchKernelGGL(gn_bwd_f32_kernel, dim3(groups, batch), dim3(256), 0, LECO_STREAM, s, (const float*)dy, lddy, gamma, beta, stats,
                       hw, c, groups, act, (float*)dx, lddx);
    return check_launch("leco_f32_groupnorm_bwd");
}
extern "C" int leco_f32_layernorm_fwd(const void* x, int64_t ldx, const float* gamma, const float* beta, float eps, int32_t m, int32_t c,
                                      void* y, int64_t ldy, float* mean, float* rstd, leco_stream_t stream) {
    hipLaunchKernelGGL(ln_fwd_f32_kernel, dim3(cdiv(m, 4)), dim3(256), 0, LECO_STREAM, (const float*)x, ldx, gamma, beta, eps, m, c,
                       (float*)y, ldy, mean, rstd);
    return check_launch("leco_f32_layernorm_fwd");
}
extern "C" int leco_f32_layernorm_bwd(const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* gamma, const float* mean,
                                      const float* rstd, const void* dres, int64_t ldres, int32_t m, int32_t c, void* dx, int64_t lddx,
                                      leco_stream_t stream) {
    hipLaunchKernelGGL(ln_bwd_f32_kernel, dim3(cdiv(m, 4)), dim3(256), 0, LECO_STREAM, (const float*)x, ldx, (const float*)dy, lddy,
                       gamma, mean, rstd, (const float*)dres, ldres, m, c, (float*)dx, lddx);
    return check_launch("leco_f32_layernorm_bwd");
}
extern "C" int leco_f32_geglu_fwd(const void* u, int64_t ldu, void* y, int64_t ldy, int32_t m, int32_t f, leco_stream_t stream) {
    hipLaunchKernelGGL(geglu_fwd_f32_kernel, dim3(grid1((int64_t)m * f)), dim3(256), 0, LECO_STREAM, (const float*)u, ldu, (float*)y, ldy, m, f);
    return check_launch("leco_f32_geglu_fwd");
}
extern "C" int leco_f32_geglu_bwd(const void* u, int64_t ldu, const void* dy, int64_t lddy, void* du, int64_t lddu, int32_t m, int32_t f,
                                  leco_stream_t stream) {
    hipLaunchKernelGGL(geglu_bwd_f32_kernel, dim3(grid1((int64_t)m * f)), dim3(256), 0, LECO_STREAM, (const float*)u, ldu, (const float*)dy,
                       lddy, (float*)du, lddu, m, f);
    return check_launch("leco_f32_geglu_bwd");
}
extern "C" int leco_f32_add(const void* a, int64_t lda, const void* b, int64_t ldb, const void* c, int64_t ldc, void* out, int64_t ldo,
                            int32_t m, int32_t cols, leco_stream_t stream) {
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid1((int64_t)m * cols)), dim3(256), 0, LECO_STREAM, (const float*)a, lda, (const float*)b, ldb,
                       (const float*)c, ldc, (float*)out, ldo, m, cols);
    return check_launch("leco_f32_add");
}
extern "C" int leco_f32_upsample2x_bwd(const void* dy, void* dx, int32_t batch, int32_t h, int32_t w, int32_t c, leco_stream_t stream) {
    hipLaunchKernelGGL(upsample_bwd_f32_kernel, dim3(grid1((int64_t)batch * h * w * c)), dim3(256), 0, LECO_STREAM, (const float*)dy,
                       (float*)dx, batch, h, w, c);
    return check_launch("leco_f32_upsample2x_bwd");
}
extern "C" int leco_f32_conv_in(const void* x, const float* w, const float* bias, void* y, int32_t batch, int32_t h, int32_t wd,
                                int32_t cin, int32_t cout, leco_stream_t stream) {
    hipLaunchKernelGGL(conv_in_f32_kernel, dim3(grid1((int64_t)batch * h * wd * cout)), dim3(256), 0, LECO_STREAM, (const float*)x, w, bias,
                       (float*)y, batch, h, wd, cin, cout);
    return check_launch("leco_f32_conv_in");
}
extern "C" int leco_f32_conv_out(const void* x, const void* w, const float* bias, float* y, int32_t batch, int32_t h, int32_t wd,
                                 int32_t c, int32_t cout, leco_stream_t stream) {
    hipLaunchKernelGGL(conv_out_f32_kernel, dim3(cdiv((int64_t)batch * h * wd, 4)), dim3(256), 0, LECO_STREAM, (const float*)x,
                       (const float*)w, bias, y, batch, h, wd, c, cout);
    return check_launch("leco_f32_conv_out");
}
extern "C" int leco_f32_conv_out_bwd(const float* dy, const void* w, void* dx, int32_t batch, int32_t h, int32_t wd, int32_t c,
                                     int32_t cout, leco_stream_t stream) {
    hipLaunchKernelGGL(conv_out_bwd_f32_kernel, dim3(grid1((int64_t)batch * h * wd * c)), dim3(256), 0, LECO_STREAM, dy, (const float*)w,
                       (float*)dx, batch, h, wd, c, cout);
    return check_launch("leco_f32_conv_out_bwd");
}
extern "C" int leco_f32_timestep_embedding(const float* t_table, const int32_t* idx, int32_t t_stride, int32_t n, int32_t dim, void* out,
                                           leco_stream_t stream) {
    hipLaunchKernelGGL(timestep_f32_kernel, dim3(cdiv(n * dim / 2, 256)), dim3(256), 0, LECO_STREAM, t_table, idx, t_stride, n, dim,
                       (float*)out);
    return check_launch("leco_f32_timestep_embedding");
}
extern "C" int leco_f32_cfg_ddim_step(const float* pred, float* x, void* x2, const float* coef, const int32_t* step, float guidance,
                                      int64_t half_n, leco_stream_t stream) {
    hipLaunchKernelGGL(cfg_step_f32_kernel, dim3(grid1(half_n)), dim3(256), 0, LECO_STREAM, pred, x, (float*)x2, coef, step, guidance,
                       half_n, (const float*)nullptr, (float*)nullptr, 0, 0);
    return check_launch("leco_f32_cfg_ddim_step");
}
extern "C" int leco_f32_cfg_sched_step(const float* pred, float* x, void* x2, const float* coef, const int32_t* step, float guidance,
                                       int64_t half_n, const float* noise, float* hist, int32_t n_hist, leco_stream_t stream) {
    if (n_hist < 0 || n_hist > 3) return fail(-EINVAL, "leco_f32_cfg_sched_step: n_hist %d", n_hist);
    hipLaunchKernelGGL(cfg_step_f32_kernel, dim3(grid1(half_n)), dim3(256), 0, LECO_STREAM, pred, x, (float*)x2, coef, step, guidance,
                       half_n, noise, hist, n_hist, 1);
    return check_launch("leco_f32_cfg_sched_step");
}
extern "C" int leco_f32_cast_f32_bf16(const float* x, void* y, int64_t n, leco_stream_t stream) {   // fp32 mode: a copy
    hipLaunchKernelGGL(copy_f32_kernel, dim3(grid1(n)), dim3(256), 0, LECO_STREAM, x, (float*)y, n);
    return check_launch("leco_f32_cast");
}
extern "C" int leco_f32_rowgroup_sum(const void* x, int64_t ldx, float* out, int64_t ldo, int32_t groups, int32_t rows_per_group,
                                     int32_t cols, leco_stream_t stream) {
    hipLaunchKernelGGL(rowgroup_sum_f32_kernel, dim3(grid1((int64_t)groups * cols)), dim3(256), 0, LECO_STREAM, (const float*)x, ldx, out,
                       ldo, groups, rows_per_group, cols);
    return check_launch("leco_f32_rowgroup_sum");
}
extern "C" int leco_f32_lora_pack(const leco_lora_site* sites, int32_t nsites, leco_stream_t stream) {
    if (nsites <= 0) return 0;
    hipLaunchKernelGGL(lora_pack_f32_kernel, dim3(64, (unsigned)nsites), dim3(256), 0, LECO_STREAM, sites);
    return check_launch("leco_f32_lora_pack");
}
extern "C" int leco_f32_lora_wgrad(const void* p, int64_t ldp, const void* q, int64_t ldq, float* g, int64_t g_sj, int64_t g_sc,
                                   int32_t m, int32_t r, int32_t cols, float scale, float* part, int64_t part_bytes,
                                   leco_stream_t stream) {
    (void)part; (void)part_bytes;
    hipLaunchKernelGGL(lora_wgrad_f32_kernel, dim3(grid1((int64_t)r * cols)), dim3(256), 0, LECO_STREAM, (const float*)p, ldp,
                       (const float*)q, ldq, g, g_sj, g_sc, m, r, cols, scale, (int)LECO_A_PLAIN, 0, 0, 0, 0, 0, 0);
    return check_launch("leco_f32_lora_wgrad");
}
extern "C" int leco_f32_lora_wgrad_conv(const void* p, int64_t ldp, const void* q, int64_t ldq, float* g, int64_t g_sj, int64_t g_sc,
                                        int32_t m, int32_t r, int32_t cols, float scale, int32_t a_mode, int32_t h_out, int32_t w_out,
                                        int32_t h_in, int32_t w_in, int32_t kh, int32_t kw, float* part, int64_t part_bytes,
                                        leco_stream_t stream) {
    (void)part; (void)part_bytes;
    hipLaunchKernelGGL(lora_wgrad_f32_kernel, dim3(grid1((int64_t)r * cols)), dim3(256), 0, LECO_STREAM, (const float*)p, ldp,
                       (const float*)q, ldq, g, g_sj, g_sc, m, r, cols, scale, a_mode, h_out, w_out, h_in, w_in, kh, kw);
    return check_launch("leco_f32_lora_wgrad_conv");
}
