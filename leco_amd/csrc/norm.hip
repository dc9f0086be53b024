// GroupNorm(32)(+SiLU) and LayerNorm, forward and dgrad, channels-last bf16.
// HBM-bound: every kernel moves 16-byte (8 x bf16) vectors per lane; statistics are fp32.
//
// GroupNorm on [B][HW][C] with G groups of cg = C/G adjacent channels:
//   one kernel (gn_block_kernel): a block owns all pixels of a few adjacent groups of one sample, so the
//   statistics {sum, sumsq} and y = act((x - mean) * rstd * gamma + beta) (act = SiLU or identity) need no
//   cross-block reduction; the second sweep over the same pixels hits in L2.
// The input may be the channel-concat of two tensors (UNet skip connections): channels
// [0,c0) come from x0, [c0,C) from x1 -- the concat is never materialised.
// Backward (frozen gamma/beta => dgrad only): the same kernel in MODE 1 accumulates {sum dxhat, sum dxhat*xhat}
// per (b,g) and forms dx = rstd*(dxhat - s1/n - xhat*s2/n).
//
// LayerNorm over the last dim (C <= 2048): one wave64 per row, two-pass statistics held in
// registers, wave-shuffle reductions.
#include <errno.h>
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <leco_prims.h>

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
// z sigmoid(z) with the hardware reciprocal (v_rcp_f32, 1 ulp) and exp2: the IEEE fp32 division of `z / (1 + expf(-z))` is a
// ten-instruction sequence per element, and the GroupNorm launches of the 32^2 / 64^2 levels are VALU-bound on it (round 6)
__device__ __forceinline__ float silu(float z) { return z * fast_rcp(1.f + fast_exp2(-1.4426950408889634f * z)); }
__device__ __forceinline__ float dsilu(float z) {
    float s = 1.f / (1.f + __expf(-z));
    return s * (1.f + z * (1.f - s));
}

struct GnSrc {
    const bf16_t* x0;
    const bf16_t* x1;
    int64_t ld0, ld1;
    int c0;  // channels served by x0
};
__device__ __forceinline__ u32x4 gn_load(const GnSrc& s, int64_t row, int c) {
    return c < s.c0 ? *(const u32x4*)(s.x0 + row * s.ld0 + c) : *(const u32x4*)(s.x1 + row * s.ld1 + (c - s.c0));
}

constexpr int GN_MAX_GROUPS = 32;
constexpr int GN_MAX_WAVES = 16;     // 1024 threads

// One block = (sample b, a run of `gpb` adjacent groups whose channel span is a multiple of VEC): the block owns
// every pixel of those channels, so the statistics need no cross-block step -- no atomics, no partial
// buffers, no finishing kernel, bitwise reproducible.  Thread t = (pixel lane t / nv, channel vector t % nv):
// consecutive threads read consecutive 2 VEC-byte pieces of a pixel's channel run.
//   pass 1: per-thread sums over its pixels -> per-group sums in registers -> DPP wave reduction -> one LDS exchange
//           of the wave totals, added by every thread in the same fixed order -> group statistics
//   pass 2: the same pixels again (L2-resident by now) -> normalise (+SiLU) / dgrad -> store.
// MODE 0: forward, statistics {sum x, sum x^2} (also written to stats[b][g][2] for the backward).
// MODE 1: backward, statistics {sum dxhat, sum dxhat*xhat}, dx = rstd*(dxhat - s1/n - xhat*s2/n).
// NVR > 0 (forward only): every thread owns at most NVR pixels of its channel vector and KEEPS them in registers between
// the two passes -- one trip to memory instead of two (the launches are latency-, not bandwidth-bound: 8 - 16 us each, 45
// of them per UNet pass).
// VEC in {8, 4, 2} channels per thread vector (round 6).  With 16-byte vectors a block needs a channel run that is a
// multiple of 8: 40 channels = 2 groups at C = 640, 4 groups at C = 320 -- 64 resp. 32 blocks at UNet batch 4, a quarter /
// an eighth of the chip doing the whole tensor's SiLU arithmetic (14.6 / 13.6 us at HW = 1024,
// profiles/r06_bench_norm.txt).  8- and 4-byte vectors let a block own ONE group (20 resp. 10 channels): 128 blocks, each
// with half / a quarter of the work; the narrower loads cost nothing at these sizes (the tensors are L2-resident).
template <int VEC>
struct GnVec { unsigned w[VEC / 2]; };
template <int VEC>
__device__ __forceinline__ GnVec<VEC> gn_loadv(const GnSrc& s, int64_t row, int c) {
    const bf16_t* p = c < s.c0 ? s.x0 + row * s.ld0 + c : s.x1 + row * s.ld1 + (c - s.c0);
    GnVec<VEC> v;
    if constexpr (VEC == 8) {
        const u32x4 t = *(const u32x4*)p;
        v.w[0] = t[0]; v.w[1] = t[1]; v.w[2] = t[2]; v.w[3] = t[3];
    } else if constexpr (VEC == 4) {
        const u32x2 t = *(const u32x2*)p;
        v.w[0] = t[0]; v.w[1] = t[1];
    } else {
        v.w[0] = *(const unsigned*)p;
    }
    return v;
}
template <int VEC>
__device__ __forceinline__ GnVec<VEC> gn_loadp(const bf16_t* p) {
    GnVec<VEC> v;
    if constexpr (VEC == 8) {
        const u32x4 t = *(const u32x4*)p;
        v.w[0] = t[0]; v.w[1] = t[1]; v.w[2] = t[2]; v.w[3] = t[3];
    } else if constexpr (VEC == 4) {
        const u32x2 t = *(const u32x2*)p;
        v.w[0] = t[0]; v.w[1] = t[1];
    } else {
        v.w[0] = *(const unsigned*)p;
    }
    return v;
}
template <int VEC>
__device__ __forceinline__ void gn_storev(bf16_t* p, const float (&o)[VEC]) {
    if constexpr (VEC == 8) {
        *(u32x4*)p = u32x4{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
    } else if constexpr (VEC == 4) {
        *(u32x2*)p = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
    } else {
        *(unsigned*)p = pack_bf2(o[0], o[1]);
    }
}
template <int VEC>
__device__ __forceinline__ void gn_unpack(const GnVec<VEC>& v, float (&f)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC / 2; ++i) {
        f[2 * i] = bf2f((bf16_t)(v.w[i] & 0xffffu));
        f[2 * i + 1] = bf2f((bf16_t)(v.w[i] >> 16));
    }
}

template <int MODE, int NVR = 0, int VEC = 8>
__global__ __launch_bounds__(1024) void gn_block_kernel(GnSrc src, const bf16_t* dy, int64_t lddy,
                                                         const float* fstats, float* stats_out,
                                                         const float* gamma, const float* beta, int act,
                                                         float eps, int hw, int C, int G, int gpb,
                                                         bf16_t* out, int64_t ldo) {
    __shared__ f32x4 red[GN_MAX_WAVES][2];       // per wave: {sum, sumsq} of the block's <= 4 groups
    const int tid = (int)threadIdx.x, NT = (int)blockDim.x;
    // XCD-aware walk (round 6): a block's channel run is 20 - 80 bytes of every pixel row, so the 128-byte lines of the tensor
    // are shared by the blocks of ADJACENT groups.  The hardware places linear workgroup id L on XCD L % 8; handing each XCD
    // a contiguous run of (sample, group run) items makes neighbours meet in one L2 -- a line is fetched once and its
    // partial-sector stores merge there, instead of three XCDs each fetching it and writing a piece of it back.
    const int nbx = (int)gridDim.x, nitems = nbx * (int)gridDim.y;
    int item = (int)blockIdx.y * nbx + (int)blockIdx.x;
    if ((nitems & 7) == 0) item = (item & 7) * (nitems >> 3) + (item >> 3);
    const int b = item / nbx, g0 = (item - b * nbx) * gpb;
    const int cg = C / G, chunkC = gpb * cg, nv = chunkC / VEC, cbase = g0 * cg;
    const int PL = NT / nv;                       // pixel lanes
    const int pl = tid / nv, v = tid - pl * nv;
    const bool active = pl < PL;
    const int c = cbase + v * VEC;
    const float inv_n = 1.f / ((float)hw * (float)cg);
    int gi[VEC];                                  // group (inside the block's run) of each of this thread's channels
#pragma unroll
    for (int i = 0; i < VEC; ++i) gi[i] = (v * VEC + i) / cg;

    float ga[VEC], be[VEC], mu[VEC], rs[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { ga[i] = 1.f; be[i] = 0.f; mu[i] = 0.f; rs[i] = 1.f; }
    if (active) {
#pragma unroll
        for (int i = 0; i < VEC; i += 2) {
            const f32x2 a = *(const f32x2*)(gamma + c + i), bb = *(const f32x2*)(beta + c + i);
            ga[i] = a[0]; ga[i + 1] = a[1]; be[i] = bb[0]; be[i + 1] = bb[1];
        }
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const int g = (c + i) / cg;
                const float m = fstats[(b * G + g) * 2] * inv_n;
                const float var = fstats[(b * G + g) * 2 + 1] * inv_n - m * m;
                mu[i] = m;
                rs[i] = rsqrtf(fmaxf(var, 0.f) + eps);
            }
        }
    }
    // ---- pass 1
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
    GnVec<VEC> keep[NVR > 0 ? NVR : 1];
    if (active && NVR > 0) {
#pragma unroll
        for (int j = 0; j < NVR; ++j) {
            const int p = pl + j * PL;
            keep[j] = gn_loadv<VEC>(src, (int64_t)b * hw + (p < hw ? p : 0), c);      // (pl itself may lie beyond a small sample)
        }
#pragma unroll
        for (int j = 0; j < NVR; ++j) {
            if (pl + j * PL < hw) {
                float x[VEC];
                gn_unpack<VEC>(keep[j], x);
#pragma unroll
                for (int i = 0; i < VEC; ++i) { s1[i] += x[i]; s2[i] += x[i] * x[i]; }
            }
        }
    } else if (active) {
        for (int p0 = pl; p0 < hw; p0 += 4 * PL) {
            GnVec<VEC> xv[4], dv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + u * PL;
                const int64_t row = (int64_t)b * hw + (p < hw ? p : p0);
                xv[u] = gn_loadv<VEC>(src, row, c);
                if (MODE == 1) dv[u] = gn_loadp<VEC>(dy + row * lddy + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (p0 + u * PL >= hw) break;
                float x[VEC];
                gn_unpack<VEC>(xv[u], x);
                if (MODE == 0) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) { s1[i] += x[i]; s2[i] += x[i] * x[i]; }
                } else {
                    float d[VEC];
                    gn_unpack<VEC>(dv[u], d);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const float xh = (x[i] - mu[i]) * rs[i];
                        float dz = d[i];
                        if (act) dz *= dsilu(xh * ga[i] + be[i]);
                        const float dxh = dz * ga[i];
                        s1[i] += dxh;
                        s2[i] += dxh * xh;
                    }
                }
            }
        }
    }
    // ---- block reduction: the thread's per-channel sums fold into the <= 4 groups of the run (registers), every wave
    // reduces them with DPP adds (wave_sum: no LDS traffic), ONE exchange through LDS, and every thread adds the wave totals
    // in the same fixed order -- one barrier, bitwise reproducible (the round-2..5 form went through three LDS stages and
    // four barriers with serial 13- to 20-term chains of LDS reads: 3 - 5x the latency floor of these launches).
    f32x4 g1 = {0.f, 0.f, 0.f, 0.f}, g2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < VEC; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            g1[g] += gi[i] == g ? s1[i] : 0.f;
            g2[g] += gi[i] == g ? s2[i] : 0.f;
        }
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if (g < gpb) { g1[g] = wave_sum(g1[g]); g2[g] = wave_sum(g2[g]); }
    const int wv = tid >> 6, nwv = NT >> 6;
    if ((tid & 63) == 0) { red[wv][0] = g1; red[wv][1] = g2; }
    __syncthreads();
    f32x4 gt1 = red[0][0], gt2 = red[0][1];
#pragma unroll
    for (int w = 1; w < GN_MAX_WAVES; ++w) {          // (unrolled: all LDS reads in flight at once; same order for every thread)
        if (w < nwv) {
            const f32x4 a = red[w][0], q = red[w][1];
#pragma unroll
            for (int g = 0; g < 4; ++g) { gt1[g] += a[g]; gt2[g] += q[g]; }
        }
    }
    if (tid < gpb) {
        const float t1v = tid == 0 ? gt1[0] : (tid == 1 ? gt1[1] : (tid == 2 ? gt1[2] : gt1[3]));
        const float t2v = tid == 0 ? gt2[0] : (tid == 1 ? gt2[1] : (tid == 2 ? gt2[2] : gt2[3]));
        stats_out[((int64_t)b * G + g0 + tid) * 2] = t1v;
        stats_out[((int64_t)b * G + g0 + tid) * 2 + 1] = t2v;
    }
    if (!active) return;
    // ---- pass 2
    float t1[VEC], t2[VEC];   // MODE 0: mean, rstd.  MODE 1: s1/n, s2/n
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const int g = gi[i];
        const float q1 = (g == 0 ? gt1[0] : (g == 1 ? gt1[1] : (g == 2 ? gt1[2] : gt1[3]))) * inv_n;
        const float q2 = (g == 0 ? gt2[0] : (g == 1 ? gt2[1] : (g == 2 ? gt2[2] : gt2[3]))) * inv_n;
        if (MODE == 0) {
            t1[i] = q1;
            t2[i] = rsqrtf(fmaxf(q2 - q1 * q1, 0.f) + eps);
        } else {
            t1[i] = q1;
            t2[i] = q2;
        }
    }
    if (NVR > 0) {
#pragma unroll
        for (int j = 0; j < NVR; ++j) {
            const int p = pl + j * PL;
            if (p < hw) {
                float x[VEC], o[VEC];
                gn_unpack<VEC>(keep[j], x);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float z = (x[i] - t1[i]) * t2[i] * ga[i] + be[i];
                    o[i] = act ? silu(z) : z;
                }
                gn_storev<VEC>(out + ((int64_t)b * hw + p) * ldo + c, o);
            }
        }
        return;
    }
    for (int p0 = pl; p0 < hw; p0 += 4 * PL) {
        GnVec<VEC> xv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * PL;
            const int64_t row = (int64_t)b * hw + (p < hw ? p : p0);
            xv[u] = gn_loadv<VEC>(src, row, c);
            if (MODE == 1) dv[u] = gn_loadp<VEC>(dy + row * lddy + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * PL;
            if (p >= hw) break;
            float x[VEC], o[VEC];
            gn_unpack<VEC>(xv[u], x);
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float z = (x[i] - t1[i]) * t2[i] * ga[i] + be[i];
                    o[i] = act ? silu(z) : z;
                }
            } else {
                float d[VEC];
                gn_unpack<VEC>(dv[u], d);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float xh = (x[i] - mu[i]) * rs[i];
                    float dz = d[i];
                    if (act) dz *= dsilu(xh * ga[i] + be[i]);
                    const float dxh = dz * ga[i];
                    o[i] = rs[i] * (dxh - t1[i] - xh * t2[i]);
                }
            }
            gn_storev<VEC>(out + ((int64_t)b * hw + p) * ldo + c, o);
        }
    }
}

constexpr int GN_MAX_PARTS = 256;  // pixel-range blocks per sample
constexpr int GN_LDS_FLOATS = 2 * 2560 * 2;  // [rows_par][C][2] staging, rows_par * C <= 5120

// ---- multi-block path for LARGE samples (few (sample, group-run) blocks, > ~200 KB each): three launches.
// Deterministic statistics (no atomics: results are bitwise reproducible run to run):
// gn_stats: block (part, b) reduces its pixel range to part[b][part][g][2]; gn_finish sums the parts in a
// fixed order; gn_apply normalises.
// MODE 0: forward stats {sum x, sum x^2}.  MODE 1: backward stats {sum dxhat, sum dxhat*xhat}.
template <int MODE>
__global__ __launch_bounds__(256) void gn_stats_kernel(GnSrc src, const bf16_t* dy, int64_t lddy,
                                                        const float* fstats, const float* gamma,
                                                        const float* beta, int act, float eps, int hw,
                                                        int C, int G, int pix_per_block, float* part) {
    __shared__ float stage[GN_LDS_FLOATS];
    const int tid = (int)threadIdx.x;
    const int b = (int)blockIdx.y, nparts = (int)gridDim.x;
    const int p0 = (int)blockIdx.x * pix_per_block;
    const int p1 = min(p0 + pix_per_block, hw);
    const int nvec = C / 8, cg = C / G;
    const int cols = nvec < 256 ? nvec : 256, rows_par = 256 / cols;
    const float inv_n = 1.f / ((float)hw * (float)cg);
    if (tid < rows_par * cols) {
        const int r = tid / cols;
        for (int v = tid - r * cols; v < nvec; v += cols) {
            const int c = v * 8;
            float s1[8], s2[8], ga[8], be[8], mu[8], rs[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int g = (c + i) / cg;
                    float m = fstats[(b * G + g) * 2] * inv_n;
                    float var = fstats[(b * G + g) * 2 + 1] * inv_n - m * m;
                    mu[i] = m;
                    rs[i] = rsqrtf(fmaxf(var, 0.f) + eps);
                    ga[i] = gamma[c + i];
                    be[i] = beta[c + i];
                }
            }
            for (int p = p0 + r; p < p1; p += rows_par) {
                const int64_t row = (int64_t)b * hw + p;
                float x[8];
                unpack8(gn_load(src, row, c), x);
                if (MODE == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { s1[i] += x[i]; s2[i] += x[i] * x[i]; }
                } else {
                    float d[8];
                    unpack8(*(const u32x4*)(dy + row * lddy + c), d);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float xh = (x[i] - mu[i]) * rs[i];
                        float dz = d[i];
                        if (act) dz *= dsilu(xh * ga[i] + be[i]);
                        float dxh = dz * ga[i];
                        s1[i] += dxh;
                        s2[i] += dxh * xh;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                stage[((r * C) + c + i) * 2] = s1[i];
                stage[((r * C) + c + i) * 2 + 1] = s2[i];
            }
        }
    }
    __syncthreads();
    if (tid < 2 * G) {   // fixed-order reduction of group g, component (tid & 1)
        const int g = tid >> 1, comp = tid & 1;
        float acc = 0.f;
        for (int r = 0; r < rows_par; ++r)
            for (int c = g * cg; c < (g + 1) * cg; ++c) acc += stage[((r * C) + c) * 2 + comp];
        part[(((int64_t)b * nparts + blockIdx.x) * G + g) * 2 + comp] = acc;
    }
}

// finishes the statistics: stats[b][g][2] = sum over parts (fixed order); one block per sample.
// thread (q = t / 64, v = t % 64) sums parts q, q+4, q+8, ... of value v with independent loads in
// flight, then the four quarter sums are combined in a fixed order.
__global__ __launch_bounds__(256) void gn_finish_kernel(const float* part, int nparts, int G, float* stats) {
    __shared__ float quarter[4 * 64];
    const int b = (int)blockIdx.x, t = (int)threadIdx.x;
    const int q = t >> 6, v = t & 63;
    float acc = 0.f;
    if (v < 2 * G) {
        const float* src = part + (int64_t)b * nparts * G * 2 + v;
#pragma unroll 8
        for (int p = q; p < nparts; p += 4) acc += src[(int64_t)p * G * 2];
    }
    quarter[t] = acc;
    __syncthreads();
    if (t < 2 * G) stats[(int64_t)b * G * 2 + t] = (quarter[t] + quarter[64 + t]) + (quarter[128 + t] + quarter[192 + t]);
}

// MODE 0: forward apply.  MODE 1: backward apply (dx).
template <int MODE>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnSrc src, const bf16_t* dy, int64_t lddy,
                                                        const float* fstats, const float* bstats,
                                                        const float* gamma, const float* beta, int act,
                                                        float eps, int hw, int C, int G, int64_t total_vec,
                                                        bf16_t* out, int64_t ldo) {
    const int nvec = C / 8, cg = C / G;
    const float inv_n = 1.f / ((float)hw * (float)cg);
    const int total = (int)total_vec;
    for (int e = (int)(blockIdx.x * 256 + threadIdx.x); e < total; e += (int)(gridDim.x * 256)) {
        const int row = e / nvec;
        const int c = (e - row * nvec) * 8;
        const int b = row / hw;
        float x[8], o[8], d[8], ga[8], be[8];
        unpack8(gn_load(src, row, c), x);
        if (MODE == 1) unpack8(*(const u32x4*)(dy + (int64_t)row * lddy + c), d);
        {
            f32x4 g0 = *(const f32x4*)(gamma + c), g1 = *(const f32x4*)(gamma + c + 4);
            f32x4 b0 = *(const f32x4*)(beta + c), b1 = *(const f32x4*)(beta + c + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { ga[i] = g0[i]; ga[4 + i] = g1[i]; be[i] = b0[i]; be[4 + i] = b1[i]; }
        }
        // the 8 channels c..c+7 lie in groups gA = c/cg .. gB = (c+7)/cg; cache the statistics of up to 4 groups
        const int gA = c / cg;
        float mu[4], rs[4], s1[4], s2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = min(gA + k, G - 1);
            const float m = fstats[(b * G + g) * 2] * inv_n;
            const float var = fstats[(b * G + g) * 2 + 1] * inv_n - m * m;
            mu[k] = m;
            rs[k] = rsqrtf(fmaxf(var, 0.f) + eps);
            if (MODE == 1) {
                s1[k] = bstats[(b * G + g) * 2] * inv_n;
                s2[k] = bstats[(b * G + g) * 2 + 1] * inv_n;
            }
        }
        int nextb = (gA + 1) * cg - c;   // first channel offset that belongs to the next group
        int k = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i >= nextb) { ++k; nextb += cg; }
            const int kk = k < 3 ? k : 3;
            const float xh = (x[i] - mu[kk]) * rs[kk];
            if (MODE == 0) {
                float z = xh * ga[i] + be[i];
                o[i] = act ? silu(z) : z;
            } else {
                float dz = d[i];
                if (act) dz *= dsilu(xh * ga[i] + be[i]);
                const float dxh = dz * ga[i];
                o[i] = rs[kk] * (dxh - s1[kk] - xh * s2[kk]);
            }
        }
        *(u32x4*)(out + (int64_t)row * ldo + c) = pack8(o);
    }
}

// ---- GroupNorm from producer-side statistics ----------------------------------------------------------------------
// The producers of a tensor (GEMM / conv epilogues, leco_gemm_args.col_stats; colstats_kernel below for the rest) leave
// {sum, sumsq} per (sample, channel).  One block = (sample, a chunk of pixels): it folds the channel sums of its sample
// into group statistics in LDS (C <= a few thousand values) and then normalises its pixels: ONE pass over the tensor, no
// reduction over pixels, one launch -- against read + reduce + read + write in gn_block_kernel, or three launches.
__global__ __launch_bounds__(256) void gn_apply_stats_kernel(GnSrc src, const float* cs0, const float* cs1, int A, const float* gamma,
                                                              const float* beta, int act, float eps, int hw, int C, int G,
                                                              int rows_per_block, float* stats, bf16_t* out, int64_t ldo) {
    __shared__ float gsum[2 * GN_MAX_GROUPS];
    __shared__ float gmr[2 * GN_MAX_GROUPS];
    const int tid = (int)threadIdx.x, b = (int)blockIdx.y, cg = C / G;
    const int na0 = src.c0 / A, na1 = (C - src.c0) / A, ag = cg / A;        // atoms of x0 / x1, atoms per group
    __shared__ float part[GN_MAX_GROUPS * 8 * 2];
    {   // 8 threads per group, each over every 8th channel of the group, combined in a fixed order: all blocks of a sample
        // arrive at bit-identical group statistics
        const int g = tid >> 3, pt = tid & 7;
        float a0 = 0.f, a1 = 0.f;
        if (g < G)
            for (int a = g * ag + pt; a < (g + 1) * ag; a += 8) {       // atom index in the concatenated channel space
                const float* p = a < na0 ? cs0 + ((int64_t)b * na0 + a) * 2 : cs1 + ((int64_t)b * na1 + (a - na0)) * 2;
                a0 += p[0];
                a1 += p[1];
            }
        if (g < GN_MAX_GROUPS) { part[tid * 2] = a0; part[tid * 2 + 1] = a1; }
    }
    __syncthreads();
    if (tid < G) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { a0 += part[(tid * 8 + k) * 2]; a1 += part[(tid * 8 + k) * 2 + 1]; }
        gsum[2 * tid] = a0;
        gsum[2 * tid + 1] = a1;
        const float inv_n = 1.f / ((float)hw * (float)cg);
        const float m = gsum[2 * tid] * inv_n, var = gsum[2 * tid + 1] * inv_n - m * m;
        gmr[2 * tid] = m;
        gmr[2 * tid + 1] = rsqrtf(fmaxf(var, 0.f) + eps);
        if (blockIdx.x == 0 && stats) {       // group sums in the layout the backward kernels read
            stats[((int64_t)b * G + tid) * 2] = gsum[2 * tid];
            stats[((int64_t)b * G + tid) * 2 + 1] = gsum[2 * tid + 1];
        }
    }
    __syncthreads();
    // Thread = (row lane, 16-byte channel vector): the per-channel scale / shift a = rstd gamma, b = beta - mean a of its 8
    // channels live in registers, so the pixel loop is load -> 8 fma (+ SiLU) -> store with no index arithmetic.
    const int nvec = C / 8;
    const int p0 = (int)blockIdx.x * rows_per_block;
    const int p1 = min(hw, p0 + rows_per_block);
    for (int v0 = 0; v0 < nvec; v0 += 256) {
        const int nvc = min(256, nvec - v0), rows_par = 256 / nvc;
        const int rl = tid / nvc, c = (v0 + tid - rl * nvc) * 8;
        if (rl >= rows_par) continue;
        float sa[8], sb[8];
        {
            const f32x4 g0 = *(const f32x4*)(gamma + c), g1 = *(const f32x4*)(gamma + c + 4);
            const f32x4 b0 = *(const f32x4*)(beta + c), b1 = *(const f32x4*)(beta + c + 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int g = (c + i) / cg;
                sa[i] = gmr[2 * g + 1] * (i < 4 ? g0[i] : g1[i - 4]);
                sb[i] = (i < 4 ? b0[i] : b1[i - 4]) - gmr[2 * g] * sa[i];
            }
        }
        for (int pr = p0 + rl; pr < p1; pr += 4 * rows_par) {
            u32x4 xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = pr + u * rows_par;
                xv[u] = gn_load(src, (int64_t)b * hw + (p < p1 ? p : pr), c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = pr + u * rows_par;
                if (p >= p1) break;
                float x[8], o[8];
                unpack8(xv[u], x);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float z = fmaf(x[i], sa[i], sb[i]);
                    o[i] = act ? silu(z) : z;
                }
                *(u32x4*)(out + ((int64_t)b * hw + p) * ldo + c) = pack8(o);
            }
        }
    }
}
// col_stats[b][c / A] += {sum, sumsq} over a chunk of sample b's rows (tensors whose producer has no statistics
// epilogue: conv_in, split-K outputs): column sums in LDS, one pair of atomics per atom and block
__global__ __launch_bounds__(256) void colstats_kernel(const bf16_t* x, int64_t ld, float* cs, int A, int hw, int C, int rows_per_block) {
    float* csum = (float*)dyn_lds();                                   // [C][2]
    const int tid = (int)threadIdx.x, b = (int)blockIdx.y, nvec = C / 8;
    const int p0 = (int)blockIdx.x * rows_per_block, p1 = min(hw, p0 + rows_per_block);
    for (int v = tid; v < nvec; v += 256) {
        float s1[8], s2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
        for (int p = p0; p < p1; p += 4) {          // four row loads in flight (the rolled loop was one dependent round trip per row)
            u32x4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) raw[u] = *(const u32x4*)(x + ((int64_t)b * hw + (p + u < p1 ? p + u : p)) * ld + v * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (p + u >= p1) break;
                float xv[8];
                unpack8(raw[u], xv);
#pragma unroll
                for (int i = 0; i < 8; ++i) { s1[i] += xv[i]; s2[i] += xv[i] * xv[i]; }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { csum[(v * 8 + i) * 2] = s1[i]; csum[(v * 8 + i) * 2 + 1] = s2[i]; }
    }
    __syncthreads();
    const int na = C / A;
    for (int a = tid; a < na; a += 256) {
        float t1 = 0.f, t2 = 0.f;
        for (int c = a * A; c < (a + 1) * A; ++c) { t1 += csum[2 * c]; t2 += csum[2 * c + 1]; }
        atomicAdd(cs + ((int64_t)b * na + a) * 2, t1);
        atomicAdd(cs + ((int64_t)b * na + a) * 2 + 1, t2);
    }
}

constexpr int LN_MAXV = 4;  // vectors of 8 per lane => C <= 2048

__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* x, int64_t ldx, const float* gamma,
                                                      const float* beta, float eps, int M, int C,
                                                      bf16_t* y, int64_t ldy, float* mean, float* rstd) {
    const int lane = lane_id();
    const int row = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    const int nvec = C / 8;
    const bool live = row < M;
    const int64_t r = live ? row : 0;
    float v[LN_MAXV][8];
    f32x4 ga[LN_MAXV][2], be[LN_MAXV][2];      // gamma / beta travel with the row: no third memory round trip behind the statistics
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int vv = lane + 64 * j;
        if (vv < nvec) {
            const u32x4 xv = *(const u32x4*)(x + r * ldx + vv * 8);
            ga[j][0] = *(const f32x4*)(gamma + vv * 8); ga[j][1] = *(const f32x4*)(gamma + vv * 8 + 4);
            be[j][0] = *(const f32x4*)(beta + vv * 8); be[j][1] = *(const f32x4*)(beta + vv * 8 + 4);
            unpack8(xv, v[j]);
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[j][i];
        }
    }
    const float mu = wave_sum(s) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        if (lane + 64 * j < nvec) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { float d = v[j][i] - mu; ss += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(ss) / (float)C + eps);
    if (!live) return;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int vv = lane + 64 * j;
        if (vv < nvec) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (v[j][i] - mu) * rs * ga[j][i >> 2][i & 3] + be[j][i >> 2][i & 3];
            *(u32x4*)(y + r * ldy + vv * 8) = pack8(o);
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// The hidden sizes of every supported UNet are C = 320 * 2^k (320 / 640 / 1280): C / 8 = 5 L vectors with L in {8, 16, 32}.
// A row is then normalised by L lanes x 5 vectors each, a wave works on 64 / L rows at once (x R2 row groups: every load of
// the wave is in flight before the first reduction), the row sums stay inside DPP rows (row8_sum / row16_sum; L = 32 adds
// the two row totals through v_readlane) and gamma / beta are fetched once per lane for all of its rows -- against one
// row per wave, 60 % idle lanes at C = 640, two 6-step ds_bpermute chains and 64 B of gamma / beta per 16 B of x above.
template <int L, int R2>
__global__ __launch_bounds__(256) void ln_fwd5_kernel(const bf16_t* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                                       int M, int C, bf16_t* y, int64_t ldy, float* mean, float* rstd) {
    constexpr int RW = 64 / L;
    const int lane = lane_id(), sub = lane / L, li = lane - sub * L;
    const int wave = (int)blockIdx.x * ((int)blockDim.x >> 6) + ((int)threadIdx.x >> 6);
    const int row0 = wave * (RW * R2) + sub;
    u32x4 xv[R2][5];
#pragma unroll
    for (int q = 0; q < R2; ++q) {
        const int row = row0 + RW * q;
        const bf16_t* xr = x + (int64_t)(row < M ? row : 0) * ldx + li * 8;
#pragma unroll
        for (int j = 0; j < 5; ++j) xv[q][j] = *(const u32x4*)(xr + L * 8 * j);
    }
    f32x4 ga[5][2], be[5][2];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int c = (li + L * j) * 8;
        ga[j][0] = *(const f32x4*)(gamma + c); ga[j][1] = *(const f32x4*)(gamma + c + 4);
        be[j][0] = *(const f32x4*)(beta + c); be[j][1] = *(const f32x4*)(beta + c + 4);
    }
    auto rsum = [&](float v) -> float {
        if constexpr (L == 8) return row8_sum(v);
        else if constexpr (L == 16) return row16_sum(v);
        else {
            v = row16_sum(v);
            const float a = shfl(v, 0) + shfl(v, 16), b2 = shfl(v, 32) + shfl(v, 48);     // (wave-uniform lanes: v_readlane)
            return sub == 0 ? a : b2;
        }
    };
    const float inv_c = 1.f / (float)C;
    float mu[R2], rs[R2];
#pragma unroll
    for (int q = 0; q < R2; ++q) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            float f[8];
            unpack8(xv[q][j], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) s += f[i];
        }
        mu[q] = rsum(s) * inv_c;
    }
#pragma unroll
    for (int q = 0; q < R2; ++q) {
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            float f[8];
            unpack8(xv[q][j], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = f[i] - mu[q]; ss += d * d; }
        }
        rs[q] = rsqrtf(rsum(ss) * inv_c + eps);
    }
#pragma unroll
    for (int q = 0; q < R2; ++q) {
        const int row = row0 + RW * q;
        if (row >= M) continue;
        bf16_t* yr = y + (int64_t)row * ldy + li * 8;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            float f[8], o[8];
            unpack8(xv[q][j], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (f[i] - mu[q]) * rs[q] * ga[j][i >> 2][i & 3] + be[j][i >> 2][i & 3];
            *(u32x4*)(yr + L * 8 * j) = pack8(o);
        }
        if (li == 0) { mean[row] = mu[q]; rstd[row] = rs[q]; }
    }
}
template <int L>
void ln_fwd5_launch(hipStream_t s, const bf16_t* x, int64_t ldx, const float* gamma, const float* beta, float eps, int m, int c,
                    bf16_t* y, int64_t ldy, float* mean, float* rstd) {
    constexpr int RW = 64 / L;
    // two row groups per wave once that still leaves >= 4 waves for every CU; one-wave workgroups while the launch has fewer
    // than 1024 waves (every CU gets work)
    const bool two = cdiv(m, RW * 2) >= 1024;
    const int waves = cdiv(m, RW * (two ? 2 : 1));
    const int wpb = waves >= 1024 ? 4 : 1;
    if (two) hipLaunchKernelGGL((ln_fwd5_kernel<L, 2>), dim3(cdiv(waves, wpb)), dim3(64 * wpb), 0, s, x, ldx, gamma, beta, eps, m, c, y, ldy, mean, rstd);
    else hipLaunchKernelGGL((ln_fwd5_kernel<L, 1>), dim3(cdiv(waves, wpb)), dim3(64 * wpb), 0, s, x, ldx, gamma, beta, eps, m, c, y, ldy, mean, rstd);
}

// dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)) (+ dres), dxhat = dy * gamma
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* x, int64_t ldx, const bf16_t* dy,
                                                      int64_t lddy, const float* gamma, const float* mean,
                                                      const float* rstd, const bf16_t* dres, int64_t ldres,
                                                      int M, int C, bf16_t* dx, int64_t lddx) {
    const int lane = lane_id();
    const int row = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    const int nvec = C / 8;
    const bool live = row < M;
    const int64_t r = live ? row : 0;
    const float mu = mean[r], rs = rstd[r];
    float xh[LN_MAXV][8], dh[LN_MAXV][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int vv = lane + 64 * j;
        if (vv < nvec) {
            float xv[8], dv[8];
            unpack8(*(const u32x4*)(x + r * ldx + vv * 8), xv);
            unpack8(*(const u32x4*)(dy + r * lddy + vv * 8), dv);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xh[j][i] = (xv[i] - mu) * rs;
                dh[j][i] = dv[i] * gamma[vv * 8 + i];
                s1 += dh[j][i];
                s2 += dh[j][i] * xh[j][i];
            }
        }
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
    if (!live) return;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int vv = lane + 64 * j;
        if (vv < nvec) {
            float o[8], rr[8];
            if (dres) unpack8(*(const u32x4*)(dres + r * ldres + vv * 8), rr);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                o[i] = rs * (dh[j][i] - s1 - xh[j][i] * s2);
                if (dres) o[i] += rr[i];
            }
            *(u32x4*)(dx + r * lddx + vv * 8) = pack8(o);
        }
    }
}

int gn_check(int C, int G, int c0, const void* x1) {
    if (G <= 0 || G > GN_MAX_GROUPS || C % G || C % 8 || C > 5120 || C / G < 2)
        return fail(-EINVAL, "groupnorm: bad C=%d G=%d (need C %% 8 == 0, 2 <= C/G, C <= 5120)", C, G);
    if (x1 && (c0 % 8 || c0 <= 0 || c0 >= C)) return fail(-EINVAL, "groupnorm: bad concat split %d of %d", c0, C);
    return 0;
}
int pix_per_block(int batch, int hw, int c) {
    // <= GN_MAX_PARTS pixel ranges per sample; at least 16 pixels each
    int nblk = 2048 / (batch > 0 ? batch : 1);
    if (nblk < 1) nblk = 1;
    if (nblk > GN_MAX_PARTS) nblk = GN_MAX_PARTS;
    int ppb = cdiv(hw, nblk);
    return ppb < 12 ? 12 : ppb;
}
// launch geometry of gn_block_kernel: groups per block (channel span % 8 == 0), threads, LDS bytes
struct GnGeom { int gpb, threads, lds, vec; };
GnGeom gn_geom_vec(int hw, int C, int G, int vec) {
    const int cg = C / G;
    int gpb = 1;
    while ((gpb * cg) % vec) gpb *= 2;               // cg even (C % 8 == 0, G = 2^k): gpb in {1, 2, 4}
    const int nv = gpb * cg / vec;
    int64_t want = (int64_t)hw * nv;                 // one thread per (pixel, vector) if the block allows it
    int threads = want >= 1024 ? 1024 : (int)((want + 63) / 64 * 64);
    if (threads < 256) threads = 256;
    if (threads < nv) threads = (nv + 63) / 64 * 64;
    return GnGeom{gpb, threads, 0, vec};
}
// the 16-byte-vector geometry: what decides between the one-launch and the three-launch / producer-statistics forms
GnGeom gn_geom(int hw, int C, int G) { return gn_geom_vec(hw, C, G, 8); }
// One block per (sample, group run) is the fastest shape while a block's slice stays small or there are
// enough of them to fill the chip; large slices on few blocks (the 64x64 / 32x32 levels at batch 4) are
// bandwidth-starved (measured 41 vs 22 us for 4 x 4096 x 320) and take the pixel-parallel three-launch path.
bool gn_use_block_kernel(const GnGeom& ge, int batch, int hw, int C, int G) {
    const int64_t slice_bytes = (int64_t)hw * (C / G) * ge.gpb * 2;
    const int blocks = batch * (G / ge.gpb);
    return !(slice_bytes > 200 * 1024 && blocks < 96);
}
// The geometry a one-launch shape RUNS on: with 16-byte vectors a block is 40 channels wide at C = 320 / 640 / 960 (4 / 2 / 4
// groups) -- 32 or 64 blocks at UNet batch 4; where fewer than 128 blocks would carry a tensor of >= 256 pixels per sample,
// narrower vectors give one group per block (LECO_GN_VEC=8 keeps the 16-byte form: A/B measurements).
GnGeom gn_geom_run(int batch, int hw, int C, int G) {
    GnGeom ge = gn_geom_vec(hw, C, G, 8);
    static const int forced = [] { const char* e = getenv("LECO_GN_VEC"); return e ? atoi(e) : 0; }();
    if (forced == 8 || hw < 256) return ge;
    for (int vec = 4; vec >= 2 && batch * (G / ge.gpb) < 128 && ge.gpb > 1; vec >>= 1) ge = gn_geom_vec(hw, C, G, vec);
    return ge;
}
template <int MODE, int NVR, int VEC>
void gn_launch_v(const GnGeom& ge, dim3 grid, hipStream_t s, GnSrc src, const bf16_t* dy, int64_t lddy,
                 const float* fstats, float* stats_out, const float* gamma, const float* beta, int act, float eps,
                 int hw, int C, int G, bf16_t* out, int64_t ldo) {
    hipLaunchKernelGGL((gn_block_kernel<MODE, NVR, VEC>), grid, dim3(ge.threads), ge.lds, s, src, dy, lddy, fstats, stats_out,
                       gamma, beta, act, eps, hw, C, G, ge.gpb, out, ldo);
}
template <int MODE, int VEC>
void gn_launch_vec(const GnGeom& ge, dim3 grid, hipStream_t s, GnSrc src, const bf16_t* dy, int64_t lddy,
                   const float* fstats, float* stats_out, const float* gamma, const float* beta, int act, float eps,
                   int hw, int C, int G, bf16_t* out, int64_t ldo) {
    if (MODE == 0) {      // forward: pixels per thread -> the register-resident instantiation that holds them (else two passes)
        const int nv = ge.gpb * (C / G) / VEC, PL = ge.threads / nv, per = (hw + PL - 1) / PL;
        static const bool off = [] { const char* e = getenv("LECO_GN_REGS"); return e && atoi(e) == 0; }();
        if (!off && per <= 2) return gn_launch_v<0, 2, VEC>(ge, grid, s, src, dy, lddy, fstats, stats_out, gamma, beta, act, eps, hw, C, G, out, ldo);
        if (!off && per <= 6) return gn_launch_v<0, 6, VEC>(ge, grid, s, src, dy, lddy, fstats, stats_out, gamma, beta, act, eps, hw, C, G, out, ldo);
        if (!off && per <= 16) return gn_launch_v<0, 16, VEC>(ge, grid, s, src, dy, lddy, fstats, stats_out, gamma, beta, act, eps, hw, C, G, out, ldo);
    }
    return gn_launch_v<MODE, 0, VEC>(ge, grid, s, src, dy, lddy, fstats, stats_out, gamma, beta, act, eps, hw, C, G, out, ldo);
}
template <int MODE>
void gn_launch(const GnGeom& ge, dim3 grid, hipStream_t s, GnSrc src, const bf16_t* dy, int64_t lddy,
               const float* fstats, float* stats_out, const float* gamma, const float* beta, int act, float eps,
               int hw, int C, int G, bf16_t* out, int64_t ldo) {
    if (ge.vec == 4) return gn_launch_vec<MODE, 4>(ge, grid, s, src, dy, lddy, fstats, stats_out, gamma, beta, act, eps, hw, C, G, out, ldo);
    if (ge.vec == 2) return gn_launch_vec<MODE, 2>(ge, grid, s, src, dy, lddy, fstats, stats_out, gamma, beta, act, eps, hw, C, G, out, ldo);
    return gn_launch_vec<MODE, 8>(ge, grid, s, src, dy, lddy, fstats, stats_out, gamma, beta, act, eps, hw, C, G, out, ldo);
}
}  // namespace
}  // namespace leco

using namespace leco;

extern "C" int leco_groupnorm_fwd(const void* x0, int64_t ld0, const void* x1, int64_t ld1, int32_t c0,
                                  const float* gamma, const float* beta, int32_t batch, int32_t hw,
                                  int32_t c, int32_t groups, float eps, int32_t act, float* stats,
                                  void* y, int64_t ldy, leco_stream_t stream) {
    int rc = gn_check(c, groups, c0, x1);
    if (rc) return rc;
    GnSrc src{(const bf16_t*)x0, (const bf16_t*)x1, ld0, ld1, x1 ? c0 : c};
    hipStream_t s = (hipStream_t)stream;
    if (gn_use_block_kernel(gn_geom(hw, c, groups), batch, hw, c, groups)) {
        const GnGeom ge = gn_geom_run(batch, hw, c, groups);
        gn_launch<0>(ge, dim3(groups / ge.gpb, batch), s, src, (const bf16_t*)nullptr, (int64_t)0,
                     (const float*)nullptr, stats, gamma, beta, act, eps, hw, c, groups, (bf16_t*)y, ldy);
        return check_launch("leco_groupnorm_fwd");
    }
    const int ppb = pix_per_block(batch, hw, c);
    const int nparts = cdiv(hw, ppb);
    float* part = stats + (int64_t)batch * groups * 2;   // scratch tail of the stats buffer
    hipLaunchKernelGGL((gn_stats_kernel<0>), dim3(nparts, batch), dim3(256), 0, s, src,
                       (const bf16_t*)nullptr, (int64_t)0, (const float*)nullptr, gamma, beta, act, eps, hw, c,
                       groups, ppb, part);
    hipLaunchKernelGGL(gn_finish_kernel, dim3(batch), dim3(256), 0, s, (const float*)part, nparts, groups, stats);
    const int64_t total = (int64_t)batch * hw * (c / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL((gn_apply_kernel<0>), dim3(grid), dim3(256), 0, s, src, (const bf16_t*)nullptr,
                       (int64_t)0, (const float*)stats, (const float*)nullptr, gamma, beta, act, eps, hw, c,
                       groups, total, (bf16_t*)y, ldy);
    return check_launch("leco_groupnorm_fwd");
}

/* 1: leco_groupnorm_fwd runs this shape as ONE launch (a block owns whole groups); 0: it needs the three-launch
 * statistics -> finish -> apply path (few, large slices).  The plan builder asks producers for statistics only in the
 * second case -- measured on MI355X, that is where the producer-side path pays (profiles/r03_groupnorm_fusion.txt). */
extern "C" int leco_groupnorm_single_launch(int32_t batch, int32_t hw, int32_t c, int32_t groups) {
    if (groups <= 0 || c % groups) return 1;
    return gn_use_block_kernel(gn_geom(hw, c, groups), batch, hw, c, groups) ? 1 : 0;
}
extern "C" int leco_groupnorm_apply_stats(const void* x0, int64_t ld0, const void* x1, int64_t ld1, int32_t c0,
                                          const float* cstats0, const float* cstats1, int32_t atom, const float* gamma,
                                          const float* beta, int32_t batch, int32_t hw, int32_t c, int32_t groups, float eps,
                                          int32_t act, float* stats, void* y, int64_t ldy, leco_stream_t stream) {
    int rc = gn_check(c, groups, c0, x1);
    if (rc) return rc;
    if (!cstats0 || (x1 && !cstats1)) return fail(-EINVAL, "leco_groupnorm_apply_stats: missing channel statistics");
    if (atom <= 0 || (c / groups) % atom || (x1 && c0 % atom))
        return fail(-EINVAL, "leco_groupnorm_apply_stats: atom %d must divide the group size %d and the concat split", atom, c / groups);
    GnSrc src{(const bf16_t*)x0, (const bf16_t*)x1, ld0, ld1, x1 ? c0 : c};
    // pixels per block: ~4 blocks per CU (every block first folds its sample's atom sums into group statistics), at least
    // 16 KB of the tensor each
    int rpb = cdiv((int64_t)batch * hw, 1024);
    const int min_rows = cdiv(16384, (int64_t)c * 2);
    if (rpb < min_rows) rpb = min_rows;
    if (rpb > hw) rpb = hw;
    hipLaunchKernelGGL(gn_apply_stats_kernel, dim3(cdiv(hw, rpb), batch), dim3(256), 0, (hipStream_t)stream, src, cstats0,
                       cstats1, atom, gamma, beta, act, eps, hw, c, groups, rpb, stats, (bf16_t*)y, ldy);
    return check_launch("leco_groupnorm_apply_stats");
}
extern "C" int leco_colstats(const void* x, int64_t ld, float* col_stats, int32_t atom, int32_t batch, int32_t hw, int32_t c,
                             leco_stream_t stream) {
    if (c % 8 || ld % 8 || atom <= 0 || c % atom) return fail(-EINVAL, "leco_colstats: c=%d (multiple of 8 and of atom %d)", c, atom);
    int rpb = cdiv((int64_t)batch * hw, 256);        // ~one block per CU: few atomics
    if (rpb < 16) rpb = 16;
    if (rpb > hw) rpb = hw;
    hipLaunchKernelGGL(colstats_kernel, dim3(cdiv(hw, rpb), batch), dim3(256), (size_t)c * 2 * sizeof(float), (hipStream_t)stream,
                       (const bf16_t*)x, ld, col_stats, atom, hw, c, rpb);
    return check_launch("leco_colstats");
}

extern "C" int leco_groupnorm_bwd(const void* x0, int64_t ld0, const void* x1, int64_t ld1, int32_t c0,
                                  const void* dy, int64_t lddy, const float* gamma, const float* beta,
                                  const float* stats, int32_t batch, int32_t hw, int32_t c, int32_t groups,
                                  float eps, int32_t act, float* bstats, void* dx, int64_t lddx,
                                  leco_stream_t stream) {
    int rc = gn_check(c, groups, c0, x1);
    if (rc) return rc;
    GnSrc src{(const bf16_t*)x0, (const bf16_t*)x1, ld0, ld1, x1 ? c0 : c};
    hipStream_t s = (hipStream_t)stream;
    if (gn_use_block_kernel(gn_geom(hw, c, groups), batch, hw, c, groups)) {
        const GnGeom ge = gn_geom_run(batch, hw, c, groups);
        gn_launch<1>(ge, dim3(groups / ge.gpb, batch), s, src, (const bf16_t*)dy, lddy, stats, bstats,
                     gamma, beta, act, eps, hw, c, groups, (bf16_t*)dx, lddx);
        return check_launch("leco_groupnorm_bwd");
    }
    const int ppb = pix_per_block(batch, hw, c);
    const int nparts = cdiv(hw, ppb);
    float* part = bstats + (int64_t)batch * groups * 2;
    hipLaunchKernelGGL((gn_stats_kernel<1>), dim3(nparts, batch), dim3(256), 0, s, src,
                       (const bf16_t*)dy, lddy, stats, gamma, beta, act, eps, hw, c, groups, ppb, part);
    hipLaunchKernelGGL(gn_finish_kernel, dim3(batch), dim3(256), 0, s, (const float*)part, nparts, groups, bstats);
    const int64_t total = (int64_t)batch * hw * (c / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL((gn_apply_kernel<1>), dim3(grid), dim3(256), 0, s, src, (const bf16_t*)dy, lddy,
                       stats, (const float*)bstats, gamma, beta, act, eps, hw, c, groups, total, (bf16_t*)dx,
                       lddx);
    return check_launch("leco_groupnorm_bwd");
}

extern "C" int leco_layernorm_fwd(const void* x, int64_t ldx, const float* gamma, const float* beta,
                                  float eps, int32_t m, int32_t c, void* y, int64_t ldy, float* mean,
                                  float* rstd, leco_stream_t stream) {
    if (c % 8 || c > 64 * 8 * LN_MAXV) return fail(-EINVAL, "layernorm: unsupported C=%d", c);
    static const bool no5 = [] { const char* e = getenv("LECO_LN5"); return e && atoi(e) == 0; }();     // A/B switch
    if (!no5 && (c == 320 || c == 640 || c == 1280)) {
        hipStream_t s = (hipStream_t)stream;
        if (c == 320) ln_fwd5_launch<8>(s, (const bf16_t*)x, ldx, gamma, beta, eps, m, c, (bf16_t*)y, ldy, mean, rstd);
        else if (c == 640) ln_fwd5_launch<16>(s, (const bf16_t*)x, ldx, gamma, beta, eps, m, c, (bf16_t*)y, ldy, mean, rstd);
        else ln_fwd5_launch<32>(s, (const bf16_t*)x, ldx, gamma, beta, eps, m, c, (bf16_t*)y, ldy, mean, rstd);
        return check_launch("leco_layernorm_fwd");
    }
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(cdiv(m, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       ldx, gamma, beta, eps, m, c, (bf16_t*)y, ldy, mean, rstd);
    return check_launch("leco_layernorm_fwd");
}

extern "C" int leco_layernorm_bwd(const void* x, int64_t ldx, const void* dy, int64_t lddy,
                                  const float* gamma, const float* mean, const float* rstd,
                                  const void* dres, int64_t ldres, int32_t m, int32_t c, void* dx,
                                  int64_t lddx, leco_stream_t stream) {
    if (c % 8 || c > 64 * 8 * LN_MAXV) return fail(-EINVAL, "layernorm: unsupported C=%d", c);
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(cdiv(m, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       ldx, (const bf16_t*)dy, lddy, gamma, mean, rstd, (const bf16_t*)dres, ldres, m, c,
                       (bf16_t*)dx, lddx);
    return check_launch("leco_layernorm_bwd");
}
