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
// (55-65 % of a K = 320 projection is launch, first-tile latency and epilogue).  Here ONE workgroup owns a
// stripe of 64 token rows for the whole chain:
//
//   * the residual stream h lives in REGISTERS as fp32 in the MFMA accumulator layout (2 x 4 waves, 32 x 80 per wave): a
//     residual GEMM accumulates straight into it, LayerNorm reads it (row statistics: lane-local + 2 shuffles + one LDS
//     exchange between the 4 column waves) -- h1 / h2 / h3 are never rounded, never stored;
//   * GEMM operands on the activation side are two 40 KB LDS buffers ([64][320] bf16, XOR-swizzled 16-byte chunks), which
//     the phases hand to each other (a1 -> l2 -> q2 -> a2 -> l3 -> GEGLU chunk -> h3);
//   * the block's weights (3.3 MB incl. LoRA operands) stream ONCE through a 3-slot LDS ring of [N rows][32 k] tiles by
//     buffer-descriptor LDS-DMA, described by a host-built table of sweeps; the issue cursor runs two tiles ahead of the
//     consumer ACROSS phase boundaries, so LayerNorm / softmax / GEGLU phases hide the first-tile latency of the next GEMM;
//   * LoRA: the stacked lora_down rows ride in the weight tiles (T = x down^T accumulated by the waves on "T duty"), the
//     K-extension tile [N][32] of scale*up follows each sweep (fp32 accumulation of the low-rank term, as gemm.hip);
//   * the feed-forward runs in 10 chunks of 128 hidden units: FF1 chunk (1 x 8 waves, value and gate fragment of a
//     hidden column in the same lane) -> GEGLU in registers -> bf16 chunk in LDS -> FF2 partial sums into h: the
//     [M][8C] and [M][4C] intermediates never exist;
//   * cross-attention: one wave per head, K fragments / pre-transposed V fragments straight from global (L2-resident,
//     prepared once per step by `xattn_prep_kernel`), P stays in registers (swapped products, as attention.hip).
//
// All LDS traffic is inline-asm ds_read / ds_write: with LDS-DMA in flight hipcc would otherwise drain the weight stream in
// front of every LDS access (leco_prims.h).  No asynchronous read is in flight across a loop edge (tools/audit_async_lds.py).
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

// Tuning aids (tools/ablate_stripe.py builds side libraries): LECO_STRIPE_ABLATE bit mask -- 1 = no MFMA, 2 = no weight DMA,
// 4 = no fragment reads, 8 = no cross-attention, 16 = no waits / barriers in the tile stream; results are garbage with any bit
// set, only the time means something.  LECO_STRIPE_TIMING: workgroup 0 stamps the shader clock at every phase boundary
// into a device array (leco_xblock_debug_times).  Both 0 / undefined in the product build.
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
constexpr int XKT = 32;      // k per weight tile
constexpr int XNS = 3;       // ring slots
constexpr int XMAXSW = 44;   // sweeps per program (K-extension tiles included)
constexpr int XMAXVEC = 12;  // small fp32 vectors (biases, LayerNorm affine) cached in LDS
constexpr int XNKEY = 80;    // padded prompt keys of K (5 fragments)
constexpr int XNPOS = 96;    // padded (permuted) key positions of V^T

// One pass of the tile stream over `nt` rows of a row-major bf16 matrix and k = k0 + 32 [0, ksteps): self-contained for the
// DMA issuer (no second table look-up on its critical path).  Both sources are pre-offset to the sweep's first row:
//   w: the weight rows themselves (a K-extension tile is a sweep of its own: w = rows of scale*up, one k-step);
//   x: the stacked lora_down rows riding in the tiles (16 tf of them, appended behind the nt weight rows).
struct XSweep {
    const void* w; const void* x;
    unsigned w_bytes, x_bytes;
    unsigned ldw_b, ldx_b;       // row strides in bytes
    int nt, k0, ksteps, tf;
};
struct XVec { const float* src; int off, len; };     // src (or NULL: zeros) -> floats [off, off + len) of the LDS vector cache
struct XProg {
    int nsweeps, nvec;
    int tf[8];                   // per Linear: 0 LoRA off, 1 / 2: 16 / 32 stacked lora_down rows
    XVec vec[XMAXVEC];
    XSweep sw[XMAXSW];
};

template <int C>
struct XCfg {
    static constexpr int FNC = C / 64;             // 16-column fragments per wave of a [64][C] result (2 x 4 waves)
    static constexpr int WN = C / 4;               // columns per wave
    static constexpr int ARS = C * 2;              // activation buffer row stride (bytes)
    static constexpr int ABUF = XBM * ARS;
    static constexpr int GRS = 256, GBUF = XBM * GRS;   // GEGLU chunk buffer: [64][128] bf16
    static constexpr int KS = C / XKT;
    static constexpr int SLOT = (C + 32) * 64;     // ring slot: C weight rows + up to 32 lora_down rows, 64 bytes each
    static constexpr int NVEC = 17 * C;            // floats in the vector cache
    static constexpr int OFF_A = 0, OFF_G = ABUF, OFF_RING = OFF_G + GBUF, OFF_T = OFF_RING + XNS * SLOT;
    static constexpr int OFF_SCR = OFF_T + XBM * 64, OFF_VEC = OFF_SCR + 4096, LDS_BYTES = OFF_VEC + NVEC * 4;
    static constexpr int OFF_RED = XBM * C * 4;    // column-sum scratch of the staged output store (behind the fp32 staging tile)
    static_assert(C % 64 == 0 && (C / 8) % 16 == 8, "activation swizzle assumes a row of 8 (mod 16) 16-byte chunks");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS layout does not fit");
    static_assert(OFF_RED + (512 / (C / 8)) * C * 8 + C * 8 <= OFF_T, "output staging + column-sum scratch must stay below the T buffer");
};
// vector cache slots (float offsets), tail kernel
template <int C> struct XV {
    static constexpr int BO1 = 0, BQ2 = C, BO2 = 2 * C, BFF2 = 3 * C, BPO = 4 * C, LN2G = 5 * C, LN2B = 6 * C, LN3G = 7 * C,
                         LN3B = 8 * C, BFF1 = 9 * C;
    // head kernel
    static constexpr int BPI = 0, BQKV = C, LN1G = 4 * C, LN1B = 5 * C;
};

struct XTailArgs {
    int m, heads, skv, rows_per_sample;
    const bf16_t* attn; unsigned attn_bytes, ld_attn_b;
    const bf16_t* h_in; int64_t ld_h;
    float ln_eps;
    const bf16_t* kp; const bf16_t* vt; float scale_log2;
    const bf16_t* res; int64_t ld_res;
    bf16_t* out; int64_t ld_out;
    float* col_stats; int stats_atom;
    int has_po;
};
struct XTailBlob { XTailArgs p; XProg prog; };     // device-resident launch description (leco_xblock_tail_build)

struct XHeadArgs {
    int m, rows_per_sample;
    const bf16_t* x; int64_t ld_x; unsigned x_bytes;
    const float* gn_cstats; int stats_atom, groups; const float *gn_g, *gn_b; float gn_eps;
    float ln_eps;
    bf16_t* h_out; int64_t ld_hout;
    bf16_t* qkv_out; int64_t ld_qkv;
};
struct XHeadBlob { XHeadArgs p; XProg prog; };

// ------------------------------------------------------------------------------------------------------------------
// Shared machinery of the stripe kernels: lane constants, the weight-tile stream, the two GEMM wave layouts, LayerNorm
// on the register-resident residual stream, the staged output store.  LDS is addressed by 32-bit byte address with
// compile-time displacements (lds_read16_at<OFF>): one base register per tile and operand.
// ------------------------------------------------------------------------------------------------------------------
template <int C>
struct Stripe {
    using Cf = XCfg<C>;
    static constexpr int FNC = Cf::FNC, ARS = Cf::ARS, KS = Cf::KS, SLOT = Cf::SLOT, WN = Cf::WN, GRS = Cf::GRS;
    struct F24 { bf16x8 a[2]; bf16x8 w[FNC]; bf16x8 t, ta; };     // fragments of one tile, 2 x 4 wave layout
    struct F18 { bf16x8 a[4]; bf16x8 w[2]; bf16x8 t, ta; };       // 1 x 8 layout (FF1 chunks: value + gate fragment)

    const LECO_CONST_AS XProg* prog;      // sweep table: read through the scalar cache (dynamic index, wave-uniform)
    unsigned char* lds;
    unsigned aA, aG, aRing, aT, aScr, aVec;       // LDS byte addresses of the regions
    int lane, wave, fr, fg, wm, wn;
    int swz4;          // byte offset of this lane's k-group inside a 64-byte tile row: ((fg ^ g(row)) << 4)
    int a_sw;          // activation-buffer chunk swizzle of this lane's rows (row & 7 == fr & 7)
    int a_rows24;      // byte offset of row wm * 32 + fr in an activation buffer
    int w_ofs24;       // byte offset of tile row wn * WN + fr (+ swizzled k-group)
    int lrow;          // DMA: tile row inside a 16-row piece
    unsigned csrc16;   // DMA: source byte offset of the 16-byte chunk this lane fetches
    // ---- weight stream: the sweep being issued, digested per wave (<= 3 pieces of 1 KB per tile: pieces 0 / 1 are always
    // weight rows, piece 2 weight rows or the lora_down rows), and the raw table entry of the sweep after it (prefetched
    // when the current one is entered, so no scalar load ever sits on the issue path)
    buf_rsrc rw, r2;
    unsigned so0, so1, so2, kb;        // scalar byte offsets of this wave's pieces; k byte offset of the next tile
    unsigned vw, v2;                   // per-lane byte offsets (row-in-piece x stride + swizzled chunk)
    int n_cur, kt_left;                // pieces per tile of this wave; tiles left in the sweep
    const void* nx_w; const void* nx_x;
    unsigned nx_wb, nx_xb, nx_ldw, nx_ldx;
    int nx_nt, nx_k0, nx_ks, nx_tf;
    int s_si = 0, c_next = 0;
    unsigned islot = 0, cslot = 0;     // byte offsets of the next slot to fill / to consume

    __device__ __forceinline__ Stripe(const LECO_CONST_AS XProg* pg) : prog(pg) {
        lds = dyn_lds();
        const unsigned base = lds_addr(lds);
        aA = base + Cf::OFF_A; aG = base + Cf::OFF_G; aRing = base + Cf::OFF_RING; aT = base + Cf::OFF_T;
        aScr = base + Cf::OFF_SCR; aVec = base + Cf::OFF_VEC;
        const int tid = (int)threadIdx.x;
        lane = tid & 63; wave = uniform(tid >> 6);
        fr = lane & 15; fg = lane >> 4; wm = wave >> 2; wn = wave & 3;
        // 64-byte tile rows (4 chunks): chunk c of row r sits at position c ^ g(r), g = {0, 3, 2, 1}[(r >> 2) & 3] --
        // the four 16-lane groups of a ds_read_b128 ({0-3,12-15,20-27}, ...) then touch 16 distinct 16-byte bank slots
        const int g4 = (4 - (fr >> 2)) & 3;
        swz4 = (fg ^ g4) << 4;
        a_sw = fr & 7;
        a_rows24 = (wm * 32 + fr) * ARS;
        w_ofs24 = (wn * WN + fr) * 64 + swz4;
        lrow = lane >> 2;
        csrc16 = (unsigned)(((lane & 3) ^ ((4 - ((lrow >> 2) & 3)) & 3)) << 4);
    }

    // ---- weight-tile stream ------------------------------------------------------------------------------------------
    __device__ __forceinline__ void load_next(int si) {
        const int i = si < prog->nsweeps ? si : 0;       // (past the end: any valid entry; never issued)
        const LECO_CONST_AS XSweep* sw = &prog->sw[i];
        nx_w = sw->w; nx_x = sw->x; nx_wb = sw->w_bytes; nx_xb = sw->x_bytes; nx_ldw = sw->ldw_b; nx_ldx = sw->ldx_b;
        nx_nt = sw->nt; nx_k0 = sw->k0; nx_ks = sw->ksteps; nx_tf = sw->tf;
    }
    // makes the prefetched sweep the current one (s_si) and prefetches the one after it
    __device__ __forceinline__ void enter_sweep() {
        const int npm = nx_nt >> 4, np = npm + nx_tf;          // (npm >= 16: validated on the host)
        const bool live = s_si < prog->nsweeps;
        n_cur = live ? (wave < np ? 1 : 0) + (wave + 8 < np ? 1 : 0) + (wave + 16 < np ? 1 : 0) : 0;
        kt_left = live ? nx_ks : 0x40000000;
        rw = make_rsrc(nx_w, nx_wb);
        so0 = (unsigned)(16 * wave) * nx_ldw;
        so1 = (unsigned)(16 * (wave + 8)) * nx_ldw;
        vw = (unsigned)lrow * nx_ldw + csrc16;
        if (wave + 16 < npm || !nx_x) {
            r2 = rw; so2 = (unsigned)(16 * (wave + 16)) * nx_ldw; v2 = vw;
        } else {
            r2 = make_rsrc(nx_x, nx_xb); so2 = (unsigned)(16 * (wave + 16 - npm)) * nx_ldx; v2 = (unsigned)lrow * nx_ldx + csrc16;
        }
        so2 = (unsigned)uniform((int)so2);      // (assigned beside the per-lane v2: keep it provably scalar, or the DMA's soffset
                                                // is issued through a waterfall loop)
        kb = (unsigned)nx_k0 * 2u;
        load_next(s_si + 1);
    }
    __device__ __forceinline__ void start_stream() {
        load_next(0);
        enter_sweep();
        refill();
        refill();
    }
    // issues the next tile of the program (if any) into the next ring slot.  Piece pc (16 tile rows, 1 KB): lane l fetches 16
    // bytes of row 16 pc + l / 4; the row / k part of the address is scalar (soffset), the lane part a per-sweep constant.
    // Called once after every acquire(), behind the fragment reads and the MFMAs the caller had ready.
    __device__ __forceinline__ void refill() {
        const int c = n_cur;
        unsigned char* dst = lds + Cf::OFF_RING + islot + wave * 1024;
        if (!(LECO_STRIPE_ABLATE & 2)) {
            if (c > 0) glds16_buf(rw, vw, so0 + kb, dst);
            if (c > 1) glds16_buf(rw, vw, so1 + kb, dst + 8192);
            if (c > 2) glds16_buf(r2, v2, so2 + kb, dst + 16384);
        }
        kb += 2 * XKT;
        islot = islot == (XNS - 1) * SLOT ? 0u : islot + SLOT;
        if (--kt_left == 0) { ++s_si; enter_sweep(); }
        c_next = (LECO_STRIPE_ABLATE & 2) ? 0 : c;
    }
    // the next tile has landed for every wave and every wave is done with the tile before it (whose slot the following
    // refill() re-uses).  In flight afterwards: two tiles.  (In-order completion: "at most c_next of my operations outstanding"
    // implies the older pieces of the tile about to be read are complete; other vector-memory operations in between only
    // make the wait stronger.)  Returns the LDS address of the tile.
    __device__ __forceinline__ unsigned acquire() {
        if (!(LECO_STRIPE_ABLATE & 16)) {
            if (c_next >= 3) wait_vmcnt<3>();
            else if (c_next == 2) wait_vmcnt<2>();
            else if (c_next == 1) wait_vmcnt<1>();
            else wait_vmcnt<0>();
            barrier_keep_dma();
        }
        const unsigned s = aRing + cslot;
        cslot = cslot == (XNS - 1) * SLOT ? 0u : cslot + SLOT;
        return s;
    }
    // software pipeline over NK tiles of one sweep: the fragment reads of tile t are issued behind acquire(t) and consumed
    // behind acquire(t + 1) (whose lgkmcnt(0) completes them), so LDS latency hides behind the MFMAs of tile t - 1.  Rolled
    // in pairs (two fragment sets with static names); the set in flight across the back edge lives in the same registers on
    // both sides of it.
    template <int NK, class F, class RD, class TIE, class MM>
    __device__ __forceinline__ void pipeline(F (&f)[2], RD rd, TIE tie, MM mm) {
        {
            const unsigned s = acquire();
            rd(f[0], 0, s);
            refill();
        }
        int kt = 1;
#pragma unroll 1
        for (; kt + 1 < NK; kt += 2) {
            {
                const unsigned s = acquire();
                tie(f[0]);
                rd(f[1], kt, s);
                mm(f[0]);
                refill();
            }
            {
                const unsigned s = acquire();
                tie(f[1]);
                rd(f[0], kt + 1, s);
                mm(f[1]);
                refill();
            }
        }
        if constexpr (NK % 2 == 0) {
            const unsigned s = acquire();
            tie(f[0]);
            rd(f[1], kt, s);
            mm(f[0]);
            refill();
            lds_wait<0>();
            tie(f[1]);
            mm(f[1]);
        } else {
            lds_wait<0>();
            tie(f[0]);
            mm(f[0]);
        }
    }

    // ---- small fp32 vectors (biases, LayerNorm affine): DMA'd once into LDS, read with ds_read (a plain global load would
    // queue behind every weight tile in flight: vector-memory operations complete in order)
    __device__ __forceinline__ void load_vectors() {
        int q = 0;
        for (int v = 0; v < prog->nvec; ++v) {
            const float* src = prog->vec[v].src;
            const int off = prog->vec[v].off, bytes = prog->vec[v].len * 4;
            for (int pc = 0; pc * 1024 < bytes; ++pc, ++q) {
                if ((q & 7) != wave) continue;
                const int o = pc * 1024 + lane * 16;
                if (o < bytes) {
                    if (src) {
                        glds16_buf(make_rsrc(src, (unsigned)bytes), (unsigned)o, 0u, lds + Cf::OFF_VEC + off * 4 + pc * 1024);
                    } else {
                        const u32x2 z = {0u, 0u};
                        lds_write8_at<0>(aVec + off * 4 + o, z);
                        lds_write8_at<8>(aVec + off * 4 + o, z);
                    }
                }
            }
        }
    }
    __device__ __forceinline__ f32x4 vec4(int off) const {     // 4 consecutive floats of the vector cache (off % 4 == 0)
        bf16x8 raw = lds_read16_at<0>(aVec + off * 4);
        lds_wait<0>();
        lds_tie(raw);
        return __builtin_bit_cast(f32x4, raw);
    }

    // ---- activation buffer (bufA: [64][C] bf16, chunk c of row r at position c ^ (r & 7)) -----------------------------------
    __device__ __forceinline__ int a_chunk(int chunk) const { return (chunk ^ a_sw) << 4; }
    // re-derives the lane's chunk swizzle behind an optimisation barrier: the per-k-step chunk offsets are the same in every
    // sweep, and hipcc otherwise keeps all of them live through the whole kernel (and spills)
    __device__ __forceinline__ void fresh_swizzle() { opaque(a_sw); }
    // lane's 4 consecutive columns n .. n + 3 (n % 4 == 0) of row-offset `rowb` as bf16
    __device__ __forceinline__ void put4(int rowb, int n, float v0, float v1, float v2, float v3) const {
        const u32x2 w = {pack_bf2(v0, v1), pack_bf2(v2, v3)};
        lds_write8_at<0>(aA + rowb + a_chunk(n >> 3) + ((n & 4) << 1), w);
    }
    // a [64][C] accumulator set (2 x 4 layout) as bf16 into the activation buffer
    __device__ __forceinline__ void store24(const f32x4 (&v)[2][FNC]) const {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < FNC; ++j)
                put4(a_rows24 + 16 * i * ARS, wn * WN + 16 * j + 4 * fg, v[i][j][0], v[i][j][1], v[i][j][2], v[i][j][3]);
    }
    // a [64][C] accumulator set (2 x 4 layout) as bf16 to global memory: columns col0 .. col0 + C of out (8-byte stores)
    __device__ __forceinline__ void store_global24(const f32x4 (&v)[2][FNC], bf16_t* out, int64_t ld, int col0, int m0, int m) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + wm * 32 + 16 * i + fr;
            if (row < m) {
#pragma unroll
                for (int j = 0; j < FNC; ++j) {
                    const u32x2 w = {pack_bf2(v[i][j][0], v[i][j][1]), pack_bf2(v[i][j][2], v[i][j][3])};
                    *(u32x2*)(out + (int64_t)row * ld + col0 + wn * WN + 16 * j + 4 * fg) = w;
                }
            }
        }
    }
    // DMA of a [64][C] bf16 stripe (rows m0 .. m0 + 63 of a row-major matrix; rows >= m: zeros) into the activation buffer
    __device__ __forceinline__ void load_stripe(const void* base, unsigned bytes, unsigned ld_b, int m0, int m) {
        const buf_rsrc r = make_rsrc(base, bytes);
        constexpr int NCH = C / 8, NP = Cf::ABUF / 1024;
#pragma unroll
        for (int i = 0; i < (NP + 7) / 8; ++i) {
            const int pc = wave + 8 * i;
            if (pc < NP) {
                const int u = 64 * pc + lane;
                const int row = (int)(((float)u + 0.5f) * (1.0f / (float)NCH)), pos = u - row * NCH;
                const unsigned voff = m0 + row < m ? (unsigned)(m0 + row) * ld_b + (unsigned)((pos ^ (row & 7)) << 4) : DMA_OOB;
                glds16_buf(r, voff, 0u, lds + Cf::OFF_A + pc * 1024);
            }
        }
    }

    // ---- 2 x 4 layout: wave (wm, wn) owns rows wm * 32 .. + 32, tile rows wn * WN .. + WN ---------------------------------
    struct T24 { bool has_t; int ta_ofs, t_delta; };     // T duty of this wave in a sweep with `tf` lora_down fragments
    __device__ __forceinline__ T24 duty24(int tf, int nt) const {
        T24 d;
        d.has_t = tf == 2 || (tf == 1 && wn < 2);
        const int ti = tf == 2 ? (wn & 1) : wn, tq = tf == 2 ? (wn >> 1) : 0;
        d.ta_ofs = ti * 16 * ARS;                           // the row fragment this wave projects
        d.t_delta = (nt + 16 * tq - wn * WN) * 64;          // its lora_down rows, relative to the wave's weight rows
        return d;
    }
    template <int ARS_, bool G16>
    __device__ __forceinline__ void read24(F24& f, unsigned abuf, int a_rows, int ka, unsigned s, const T24& d) const {
        if (LECO_STRIPE_ABLATE & 4) return;
        const unsigned ab = abuf + a_rows + (G16 ? ((4 * ka + fg) ^ fr) << 4 : a_chunk(4 * ka + fg));
        f.a[0] = lds_read16_at<0>(ab);
        f.a[1] = lds_read16_at<16 * ARS_>(ab);
        const unsigned wb = s + w_ofs24;
#pragma unroll
        for (int j = 0; j < FNC; ++j) {
            if (j == 0) f.w[0] = lds_read16_at<0>(wb);
            if (j == 1) f.w[1] = lds_read16_at<1024>(wb);
            if (j == 2) f.w[2] = lds_read16_at<2048>(wb);
            if (j == 3) f.w[3] = lds_read16_at<3072>(wb);
            if (j == 4) f.w[4] = lds_read16_at<4096>(wb);
        }
        if (d.has_t) {
            f.t = lds_read16_at<0>(wb + d.t_delta);
            f.ta = lds_read16_at<0>(ab + d.ta_ofs);
        }
    }
    __device__ __forceinline__ void tie24(F24& f, bool has_t) const {
        lds_tie(f.a[0]); lds_tie(f.a[1]);
#pragma unroll
        for (int j = 0; j < FNC; ++j) lds_tie(f.w[j]);
        if (has_t) { lds_tie(f.t); lds_tie(f.ta); }
    }
    __device__ __forceinline__ void mma24(f32x4 (&acc)[2][FNC], f32x4& acct, const F24& f, bool has_t) const {
        if (LECO_STRIPE_ABLATE & 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < FNC; ++j) acc[i][j][0] += __uint_as_float((unsigned)(f.w[j][0] ^ f.a[i][0]));
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < FNC; ++j) acc[i][j] = mfma16(f.w[j], f.a[i], acc[i][j]);
        if (has_t) acct = mfma16(f.t, f.ta, acct);
    }
    // one sweep of NK main tiles on the activation buffer: acc += A[:, 32 ka0 ..] W^T, acct += A t_w^T (waves on T duty)
    template <int NK>
    __device__ __forceinline__ void sweep24(f32x4 (&acc)[2][FNC], f32x4& acct, int tf) {
        static_assert(FNC <= 5, "read24 enumerates the weight fragments");
        const T24 d = duty24(tf, C);
        F24 f[2];
        fresh_swizzle();
        pipeline<NK>(f, [&](F24& x, int kt, unsigned s) { read24<ARS, false>(x, aA, a_rows24, kt, s, d); },
                     [&](F24& x) { tie24(x, d.has_t); }, [&](F24& x) { mma24(acc, acct, x, d.has_t); });
    }
    // T (fp32, waves on T duty) -> bf16 into the T buffer ([64][32], 64-byte rows, tile swizzle)
    __device__ __forceinline__ void put_t(const f32x4& acct, int row, int tq) const {
        const u32x2 w = {pack_bf2(acct[0], acct[1]), pack_bf2(acct[2], acct[3])};
        const int g4 = (4 - (fr >> 2)) & 3;
        lds_write8_at<0>(aT + row * 64 + (((2 * tq + (fg >> 1)) ^ g4) << 4) + ((fg & 1) << 3), w);
    }
    __device__ __forceinline__ void write_t24(const f32x4& acct, int tf) const {
        const bool has_t = tf == 2 || (tf == 1 && wn < 2);
        const int ti = tf == 2 ? (wn & 1) : wn, tq = tf == 2 ? (wn >> 1) : 0;
        if (has_t) put_t(acct, wm * 32 + 16 * ti + fr, tq);
    }
    // K-extension tile: acc += T (scale up)^T.  T must have been written (by any layout) before the call.
    __device__ __forceinline__ void ext24(f32x4 (&acc)[2][FNC]) {
        const unsigned s = acquire();           // barrier: T visible; the [N][32] image of scale*up landed
        F24 f;
        const unsigned tb = aT + (wm * 32 + fr) * 64 + swz4, wb = s + w_ofs24;
        if (!(LECO_STRIPE_ABLATE & 4)) {
            f.a[0] = lds_read16_at<0>(tb);
            f.a[1] = lds_read16_at<16 * 64>(tb);
#pragma unroll
            for (int j = 0; j < FNC; ++j) {
                if (j == 0) f.w[0] = lds_read16_at<0>(wb);
                if (j == 1) f.w[1] = lds_read16_at<1024>(wb);
                if (j == 2) f.w[2] = lds_read16_at<2048>(wb);
                if (j == 3) f.w[3] = lds_read16_at<3072>(wb);
                if (j == 4) f.w[4] = lds_read16_at<4096>(wb);
            }
        }
        lds_wait<0>();
        tie24(f, false);
        f32x4 dummy = {0.f, 0.f, 0.f, 0.f};
        mma24(acc, dummy, f, false);
        refill();
    }
    // a whole Linear on the 2 x 4 layout: K = C, A = the activation buffer, accumulated into acc (bias NOT added)
    __device__ __forceinline__ void linear24(f32x4 (&acc)[2][FNC], int lin) {
        const int tf = prog->tf[lin];
        f32x4 acct = {0.f, 0.f, 0.f, 0.f};
        sweep24<KS>(acc, acct, tf);
        if (tf) {
            write_t24(acct, tf);
            ext24(acc);
        }
    }
    __device__ __forceinline__ void add_bias24(f32x4 (&acc)[2][FNC], int voff) const {
        bf16x8 q[FNC];
        const unsigned a = aVec + (voff + wn * WN + 4 * fg) * 4;
#pragma unroll
        for (int j = 0; j < FNC; ++j) {
            if (j == 0) q[0] = lds_read16_at<0>(a);
            if (j == 1) q[1] = lds_read16_at<64>(a);
            if (j == 2) q[2] = lds_read16_at<128>(a);
            if (j == 3) q[3] = lds_read16_at<192>(a);
            if (j == 4) q[4] = lds_read16_at<256>(a);
        }
        lds_wait<0>();
#pragma unroll
        for (int j = 0; j < FNC; ++j) {
            lds_tie(q[j]);
            const f32x4 b = __builtin_bit_cast(f32x4, q[j]);
#pragma unroll
            for (int i = 0; i < 2; ++i) { acc[i][j][0] += b[0]; acc[i][j][1] += b[1]; acc[i][j][2] += b[2]; acc[i][j][3] += b[3]; }
        }
    }

    // ---- LayerNorm of the register-resident stream (2 x 4 layout) -> bf16 activation buffer.  The caller guarantees that no
    // wave still reads the buffer's previous contents once the FIRST barrier in here has been passed by everyone (it has:
    // every wave finished its reads before it arrives).
    __device__ __forceinline__ void layernorm24(const f32x4 (&h)[2][FNC], int vg, int vb, float eps) {
        float mean[2], rstd[2];
        float part[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < FNC; ++j) s += (h[i][j][0] + h[i][j][1]) + (h[i][j][2] + h[i][j][3]);
            s += shfl_xor(s, 16);
            s += shfl_xor(s, 32);
            part[i] = s;
        }
        const unsigned r0 = aScr + (wm * 32 + fr) * 16;
        if (fg == 0) {
            lds_write4_at(r0 + wn * 4, part[0]);
            lds_write4_at(r0 + 16 * 16 + wn * 4, part[1]);
        }
        barrier_keep_dma();
        {
            bf16x8 q0 = lds_read16_at<0>(r0), q1 = lds_read16_at<16 * 16>(r0);
            lds_wait<0>();
            lds_tie(q0); lds_tie(q1);
            const f32x4 a = __builtin_bit_cast(f32x4, q0), b = __builtin_bit_cast(f32x4, q1);
            mean[0] = ((a[0] + a[1]) + (a[2] + a[3])) * (1.0f / (float)C);
            mean[1] = ((b[0] + b[1]) + (b[2] + b[3])) * (1.0f / (float)C);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < FNC; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = h[i][j][r] - mean[i]; s += d * d; }
            s += shfl_xor(s, 16);
            s += shfl_xor(s, 32);
            part[i] = s;
        }
        const unsigned r1 = r0 + XBM * 16;
        if (fg == 0) {
            lds_write4_at(r1 + wn * 4, part[0]);
            lds_write4_at(r1 + 16 * 16 + wn * 4, part[1]);
        }
        barrier_keep_dma();
        {
            bf16x8 q0 = lds_read16_at<0>(r1), q1 = lds_read16_at<16 * 16>(r1);
            lds_wait<0>();
            lds_tie(q0); lds_tie(q1);
            const f32x4 a = __builtin_bit_cast(f32x4, q0), b = __builtin_bit_cast(f32x4, q1);
            rstd[0] = rsqrtf(((a[0] + a[1]) + (a[2] + a[3])) * (1.0f / (float)C) + eps);
            rstd[1] = rsqrtf(((b[0] + b[1]) + (b[2] + b[3])) * (1.0f / (float)C) + eps);
        }
#pragma unroll
        for (int j = 0; j < FNC; ++j) {
            const int n = wn * WN + 16 * j + 4 * fg;
            bf16x8 qg = lds_read16_at<0>(aVec + (vg + n) * 4), qb = lds_read16_at<0>(aVec + (vb + n) * 4);
            lds_wait<0>();
            lds_tie(qg); lds_tie(qb);
            const f32x4 g = __builtin_bit_cast(f32x4, qg), b = __builtin_bit_cast(f32x4, qb);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                put4(a_rows24 + 16 * i * ARS, n, (h[i][j][0] - mean[i]) * rstd[i] * g[0] + b[0],
                     (h[i][j][1] - mean[i]) * rstd[i] * g[1] + b[1], (h[i][j][2] - mean[i]) * rstd[i] * g[2] + b[2],
                     (h[i][j][3] - mean[i]) * rstd[i] * g[3] + b[3]);
        }
    }

    // ---- staged output: the [64][C] fp32 result (2 x 4 layout) goes through LDS (fp32 [64][C] from LDS address 0, 16-byte
    // chunks XOR-swizzled by row & 15) so that the global side moves whole 16-byte row segments: + bias + residual, bf16 store,
    // and the per-atom {sum, sumsq} of the stored values for the GroupNorm that follows (leco_gemm_args.col_stats).
    // Must be called by all waves after the LAST tile of the program (the staging tile overlays the ring).
    __device__ __forceinline__ void store_out(const f32x4 (&v)[2][FNC], int vbias, const bf16_t* res, int64_t ld_res,
                                              bf16_t* out, int64_t ld_out, int m0, int m, float* col_stats, int stats_atom,
                                              int rows_per_sample) {
        constexpr int NCH = C / 8, RG = 512 / NCH;        // 16-byte output chunks per row; row groups (threads NCH * RG)
        const unsigned stg = lds_addr(lds);               // fp32 [64][C], row stride 4 C bytes
        const int tid = (int)threadIdx.x;
        const int ch = tid % NCH, rg = tid / NCH;
        constexpr int NIT = (XBM + RG - 1) / RG;
        u32x4 rres[NIT];
        if (res && rg < RG) {       // residual loads first: their latency hides behind the staging
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = rg + RG * it;
                rres[it] = u32x4{0u, 0u, 0u, 0u};
                if (row < XBM && m0 + row < m) rres[it] = *(const u32x4*)(res + (int64_t)(m0 + row) * ld_res + ch * 8);
            }
        }
        float bs[8];
        if (vbias >= 0 && rg < RG) {
            const f32x4 b0 = vec4(vbias + ch * 8), b1 = vec4(vbias + ch * 8 + 4);
            bs[0] = b0[0]; bs[1] = b0[1]; bs[2] = b0[2]; bs[3] = b0[3]; bs[4] = b1[0]; bs[5] = b1[1]; bs[6] = b1[2]; bs[7] = b1[3];
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) bs[r] = 0.f;
        }
        barrier_keep_dma();                               // every wave is done reading the activation buffer and the ring
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < FNC; ++j) {
                const int row = wm * 32 + 16 * i + fr, c16 = (wn * WN + 16 * j + 4 * fg) >> 2;
                const u32x4 raw = __builtin_bit_cast(u32x4, v[i][j]);
                const u32x2 lo = {raw[0], raw[1]}, hi = {raw[2], raw[3]};
                const unsigned d = stg + row * (4 * C) + ((c16 ^ (row & 15)) << 4);
                lds_write8_at<0>(d, lo);
                lds_write8_at<8>(d, hi);
            }
        barrier_keep_dma();
        float s1[8], s2[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
        if (rg < RG) {
            bf16x8 q0[NIT], q1[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = rg + RG * it < XBM ? rg + RG * it : 0;
                q0[it] = lds_read16_at<0>(stg + row * (4 * C) + (((2 * ch) ^ (row & 15)) << 4));
                q1[it] = lds_read16_at<0>(stg + row * (4 * C) + (((2 * ch + 1) ^ (row & 15)) << 4));
            }
            lds_wait<0>();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                lds_tie(q0[it]);
                lds_tie(q1[it]);
                const int row = rg + RG * it;
                if (row >= XBM || m0 + row >= m) continue;
                const f32x4 v0 = __builtin_bit_cast(f32x4, q0[it]), v1 = __builtin_bit_cast(f32x4, q1[it]);
                float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    x[2 * r] += bs[2 * r] + (res ? bf2f((bf16_t)(rres[it][r] & 0xffffu)) : 0.f);
                    x[2 * r + 1] += bs[2 * r + 1] + (res ? bf2f((bf16_t)(rres[it][r] >> 16)) : 0.f);
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
        // column sums: the RG row groups meet in LDS (fp32 [RG][C][2] behind the staging tile), thread c sums column c over the
        // row groups, then one thread per atom adds its columns and sends one pair of atomics
        const unsigned red = stg + Cf::OFF_RED;
        if (rg < RG) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const u32x2 w = {__float_as_uint(s1[r]), __float_as_uint(s2[r])};
                lds_write8_at<0>(red + (rg * C + ch * 8 + r) * 8, w);
            }
        }
        barrier_keep_dma();
        const unsigned csum = red + RG * C * 8;           // fp32 [C][2]
        if (tid < C / 2) {                                // two columns per thread (one 16-byte read per row group)
            bf16x8 q[RG];
#pragma unroll
            for (int g = 0; g < RG; ++g) q[g] = lds_read16_at<0>(red + (g * C + 2 * tid) * 8);
            lds_wait<0>();
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                lds_tie(q[g]);
                const f32x4 a = __builtin_bit_cast(f32x4, q[g]);
                t[0] += a[0]; t[1] += a[1]; t[2] += a[2]; t[3] += a[3];
            }
            const u32x4 raw = __builtin_bit_cast(u32x4, t);
            const u32x2 lo = {raw[0], raw[1]}, hi = {raw[2], raw[3]};
            lds_write8_at<0>(csum + tid * 16, lo);
            lds_write8_at<8>(csum + tid * 16, hi);
        }
        barrier_keep_dma();
        const int natom = C / stats_atom;
        if (tid < natom && m0 < m) {
            float t1 = 0.f, t2 = 0.f;
            for (int c = tid * stats_atom; c < (tid + 1) * stats_atom; c += 2) {       // (stats_atom is even)
                bf16x8 raw = lds_read16_at<0>(csum + c * 8);
                lds_wait<0>();
                lds_tie(raw);
                const f32x4 a = __builtin_bit_cast(f32x4, raw);
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
__global__ __launch_bounds__(512) void xblock_tail_kernel(const void* blob) {
    using St = Stripe<C>;
    using Cf = XCfg<C>;
    using V = XV<C>;
    constexpr int FNC = Cf::FNC, ARS = Cf::ARS, KS = Cf::KS, WN = Cf::WN, GRS = Cf::GRS;
    constexpr int DV = (D + 15) / 16 * 16, NFD = DV / 16;
    constexpr bool ONES = DV > D;
    constexpr int NCHUNK = 4 * C / 128;             // feed-forward chunks of 128 hidden units
    const LECO_CONST_AS XTailBlob* B = LECO_CONST_CAST(XTailBlob, blob);
    const LECO_CONST_AS XTailArgs& p = B->p;
    const LECO_CONST_AS XProg* prog = &B->prog;
    St st(prog);
    const int wave = st.wave, fr = st.fr, fg = st.fg, wm = st.wm, wn = st.wn;
    const int m0 = (int)blockIdx.x * XBM, M = p.m;

    // ---- prologue (in vector-memory order): the small vectors and the self-attention output stripe (both older than every
    // weight tile, so the first acquire's counted wait covers them), the first two weight tiles; the T buffer zeroed (its
    // columns 16 .. 31 are only written by rank-stacks > 16); the residual stream h0 -> registers
    st.load_vectors();
    st.load_stripe(p.attn, p.attn_bytes, p.ld_attn_b, m0, M);
    st.start_stream();
    {
        const u32x2 z = {0u, 0u};
        lds_write8_at<0>(st.aT + (int)threadIdx.x * 8, z);
    }
    f32x4 h[2][FNC];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + wm * 32 + 16 * i + fr;
#pragma unroll
        for (int j = 0; j < FNC; ++j) {
            u32x2 raw = {0u, 0u};
            if (row < M) raw = *(const u32x2*)(p.h_in + (int64_t)row * p.ld_h + wn * WN + 16 * j + 4 * fg);
            h[i][j] = f32x4{bf2f((bf16_t)(raw[0] & 0xffffu)), bf2f((bf16_t)(raw[0] >> 16)), bf2f((bf16_t)(raw[1] & 0xffffu)),
                            bf2f((bf16_t)(raw[1] >> 16))};
        }
    }

    // ---- 1. h1 = a1 Wo1^T + bo1 + h0
    XSTAMP(0);
    st.linear24(h, 0);
    st.add_bias24(h, V::BO1);
    XSTAMP(1);
    // ---- 2. l2 = LN2(h1) -> activation buffer (a1 is dead: every wave has finished sweep 0 before LN's first barrier)
    st.layernorm24(h, V::LN2G, V::LN2B, p.ln_eps);
    XSTAMP(2);
    // ---- 3. q2 = l2 Wq2^T -> activation buffer (bf16), in place of l2
    {
        f32x4 q[2][FNC];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < FNC; ++j) q[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        st.linear24(q, 1);
        st.add_bias24(q, V::BQ2);
        barrier_keep_dma();         // every wave is done reading l2
        st.store24(q);
    }
    barrier_keep_dma();             // q2 complete
    XSTAMP(3);

    // ---- 4. cross-attention: one wave per head, IN PLACE (a wave reads and writes only its head's columns);
    // S^T = K Q^T and O^T = V^T P^T (swapped: a lane owns one query row)
    XSTAMP(4);
    if (!(LECO_STRIPE_ABLATE & 8)) {
        const int b = m0 / p.rows_per_sample;
        for (int hd = wave; hd < p.heads; hd += 8) {
            const bf16_t* kp = p.kp + (int64_t)(b * p.heads + hd) * (XNKEY * 64);
            const bf16_t* vt = p.vt + (int64_t)(b * p.heads + hd) * (DV * XNPOS);
            bf16x8 kf[5][2], vf[NFD][3];
#pragma unroll
            for (int f = 0; f < 5; ++f)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) kf[f][ks] = *(const bf16x8*)(kp + (16 * f + fr) * 64 + 32 * ks + 8 * fg);
#pragma unroll
            for (int fd = 0; fd < NFD; ++fd)
#pragma unroll
                for (int s = 0; s < 3; ++s) vf[fd][s] = *(const bf16x8*)(vt + (16 * fd + fr) * XNPOS + 32 * s + 8 * fg);
#pragma unroll 1
            for (int u = 0; u < 4; ++u) {
                // Q fragments (MFMA B operand: column = query row, k = head dim); k-groups beyond D are zero
                bf16x8 qf[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bool ok = 32 * ks + 8 * fg < D;
                    const int chunk = ok ? hd * (D / 8) + 4 * ks + fg : 0;
                    qf[ks] = lds_read16_at<0>(st.aA + (16 * u + fr) * ARS + st.a_chunk(chunk));
                }
                lds_wait<0>();
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    lds_tie(qf[ks]);
                    if (!(32 * ks + 8 * fg < D)) qf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
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
                mx = fmaxf(mx, shfl_xor(mx, 16));
                mx = fmaxf(mx, shfl_xor(mx, 32));
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
                if (ONES) {
                    l = shfl(o[D / 16][D % 4], fr + 16 * ((D % 16) / 4));
                } else {
                    l = rs;
                    l += shfl_xor(l, 16);
                    l += shfl_xor(l, 32);
                }
                const float inv = 1.f / l;
#pragma unroll
                for (int fd = 0; fd < NFD; ++fd) {
                    const int d = 16 * fd + 4 * fg;
                    if (d < D) st.put4((16 * u + fr) * ARS, hd * D + d, o[fd][0] * inv, o[fd][1] * inv, o[fd][2] * inv, o[fd][3] * inv);
                }
            }
        }
    }
    // ---- 5. h2 = a2 Wo2^T + bo2 + h1   (the first acquire's barrier publishes a2)
    XSTAMP(5);
    st.linear24(h, 2);
    st.add_bias24(h, V::BO2);
    XSTAMP(6);
    // ---- 6. l3 = LN3(h2) -> activation buffer
    st.layernorm24(h, V::LN3G, V::LN3B, p.ln_eps);
    st.add_bias24(h, V::BFF2);          // ff.net.2 bias: h becomes the accumulator of h3
    XSTAMP(7);

    // ---- 7. feed-forward in chunks of 128 hidden units.  FF1 chunk: 1 x 8 layout, wave w owns all 64 rows of hidden columns
    // 16 w .. + 16 of the chunk: tile rows vb + fr (value) and vb + 64 + fr (gate) of the 64-interleaved GEGLU weight image.
    // The bf16 chunk goes to its own LDS buffer ([64][128], chunk c of row r at position c ^ (r & 15)); FF2 reads it from there.
    {
        const int tf1 = prog->tf[3], tf2 = prog->tf[4];
        const bool has_t1 = tf1 == 2 || (tf1 == 1 && wave < 4);
        const int ti1 = wave & 3, tq1 = tf1 == 2 ? (wave >> 2) : 0;
        const int vb = 16 * wave + (wave >= 4 ? 64 : 0);
        const int wv_ofs = (vb + fr) * 64 + st.swz4;
        const int t1_delta = (256 + 16 * tq1 - vb) * 64;
        const int a_rows18 = fr * ARS, ta1_ofs = ti1 * 16 * ARS;
        const int g_rows24 = (wm * 32 + fr) * GRS;
        const typename St::T24 d2 = st.duty24(tf2, C);
        typename St::T24 d2g = d2;
        d2g.ta_ofs = d2.ta_ofs / ARS * GRS;
        f32x4 acct2 = {0.f, 0.f, 0.f, 0.f};
        typename St::F18 f1[2];
        typename St::F24 f2[2];
#pragma unroll 1
        for (int c = 0; c < NCHUNK; ++c) {
            f32x4 u[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { u[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; u[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            f32x4 acct1 = {0.f, 0.f, 0.f, 0.f};
            const bool with_t = has_t1 && c == 0;        // the lora_down rows ride in the first chunk's tiles only
            st.fresh_swizzle();
            st.template pipeline<KS>(
                f1,
                [&](typename St::F18& f, int kt, unsigned s) {
                    if (LECO_STRIPE_ABLATE & 4) return;
                    const unsigned ab = st.aA + a_rows18 + st.a_chunk(4 * kt + fg);
                    f.a[0] = lds_read16_at<0>(ab);
                    f.a[1] = lds_read16_at<16 * ARS>(ab);
                    f.a[2] = lds_read16_at<32 * ARS>(ab);
                    f.a[3] = lds_read16_at<48 * ARS>(ab);
                    const unsigned wb = s + wv_ofs;
                    f.w[0] = lds_read16_at<0>(wb);
                    f.w[1] = lds_read16_at<64 * 64>(wb);
                    if (with_t) {
                        f.t = lds_read16_at<0>(wb + t1_delta);
                        f.ta = lds_read16_at<0>(ab + ta1_ofs);
                    }
                },
                [&](typename St::F18& f) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) lds_tie(f.a[i]);
                    lds_tie(f.w[0]);
                    lds_tie(f.w[1]);
                    if (with_t) { lds_tie(f.t); lds_tie(f.ta); }
                },
                [&](typename St::F18& f) {
                    if (LECO_STRIPE_ABLATE & 1) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) u[i][0][0] += __uint_as_float((unsigned)(f.w[0][0] ^ f.w[1][0] ^ f.a[i][0]));
                        return;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        u[i][0] = mfma16(f.w[0], f.a[i], u[i][0]);
                        u[i][1] = mfma16(f.w[1], f.a[i], u[i][1]);
                    }
                    if (with_t) acct1 = mfma16(f.t, f.ta, acct1);
                });
            if (tf1) {
                if (with_t) st.put_t(acct1, 16 * ti1 + fr, tq1);
                const unsigned s = st.acquire();      // barrier: T visible (chunk 0) / still there (later chunks)
                typename St::F18 f;
                const unsigned tb = st.aT + fr * 64 + st.swz4, wb = s + wv_ofs;
                f.a[0] = lds_read16_at<0>(tb);
                f.a[1] = lds_read16_at<16 * 64>(tb);
                f.a[2] = lds_read16_at<32 * 64>(tb);
                f.a[3] = lds_read16_at<48 * 64>(tb);
                f.w[0] = lds_read16_at<0>(wb);
                f.w[1] = lds_read16_at<64 * 64>(wb);
                lds_wait<0>();
#pragma unroll
                for (int i = 0; i < 4; ++i) lds_tie(f.a[i]);
                lds_tie(f.w[0]);
                lds_tie(f.w[1]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    u[i][0] = mfma16(f.w[0], f.a[i], u[i][0]);
                    u[i][1] = mfma16(f.w[1], f.a[i], u[i][1]);
                }
                st.refill();
            }
            if (c == 0) XSTAMP(11);
            // GEGLU: value * gelu(gate) -> bf16 chunk
            {
                const f32x4 bv = st.vec4(V::BFF1 + 256 * c + vb + 4 * fg), bg = st.vec4(V::BFF1 + 256 * c + vb + 64 + 4 * fg);
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
                    lds_write8_at<0>(st.aG + (16 * i + fr) * GRS + (((2 * wave + (fg >> 1)) ^ fr) << 4) + ((fg & 1) << 3), w);
                }
            }
            if (c == 0) XSTAMP(12);
            // FF2 partial sums: h += g[:, chunk] W2[:, 128 c .. + 128]^T   (the first acquire's barrier publishes the chunk)
            st.template pipeline<4>(
                f2, [&](typename St::F24& f, int kt, unsigned s) { st.template read24<GRS, true>(f, st.aG, g_rows24, kt, s, d2g); },
                [&](typename St::F24& f) { st.tie24(f, d2.has_t); }, [&](typename St::F24& f) { st.mma24(h, acct2, f, d2.has_t); });
            if (c == 0) XSTAMP(13);
        }
        if (tf2) {
            st.write_t24(acct2, tf2);
            st.ext24(h);
        }
    }

    // ---- 8. out = proj_out(h3) + x   (or h3 itself when the Transformer2DModel goes on with another block)
    XSTAMP(8);
    if (p.has_po) {
        barrier_keep_dma();                     // every wave is done with the last FF1 reads of l3
        st.store24(h);
        f32x4 y[2][FNC];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < FNC; ++j) y[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        st.linear24(y, 5);
        XSTAMP(9);
        st.store_out(y, V::BPO, p.res, p.ld_res, p.out, p.ld_out, m0, M, p.col_stats, p.stats_atom, p.rows_per_sample);
        XSTAMP(10);
    } else {
        st.store_out(h, -1, nullptr, 0, p.out, p.ld_out, m0, M, p.col_stats, p.stats_atom, p.rows_per_sample);
    }
}

// ======================================================================================================================
// Head of a Transformer2DModel + its first BasicTransformerBlock up to the self-attention core:
//   n = GroupNorm(x) (from the statistics the producer of x left: leco_gemm_args.col_stats) ;  p = proj_in(n) -> h_out ;
//   qkv = attn1.to_q|to_k|to_v(LN1(p)) -> qkv_out   (three 320-column sweeps sharing one lora_down projection)
// ======================================================================================================================
template <int C>
__global__ __launch_bounds__(512) void xblock_head_kernel(const void* blob) {
    using St = Stripe<C>;
    using Cf = XCfg<C>;
    using V = XV<C>;
    constexpr int FNC = Cf::FNC, ARS = Cf::ARS, KS = Cf::KS;
    const LECO_CONST_AS XHeadBlob* B = LECO_CONST_CAST(XHeadBlob, blob);
    const LECO_CONST_AS XHeadArgs& p = B->p;
    const LECO_CONST_AS XProg* prog = &B->prog;
    St st(prog);
    const int m0 = (int)blockIdx.x * XBM, M = p.m;
    const int tid = (int)threadIdx.x;

    // (the vectors' and the stripe's DMA pieces must be OLDER than the weight tiles: the first acquire's counted wait then
    // covers them)
    st.load_vectors();
    if (!p.gn_cstats) st.load_stripe(p.x, p.x_bytes, (unsigned)(p.ld_x * 2), m0, M);
    st.start_stream();
    {
        const u32x2 z = {0u, 0u};
        lds_write8_at<0>(st.aT + tid * 8, z);
    }
    if (p.gn_cstats) {
        // ---- GroupNorm apply: per-channel {mean, rstd * gamma, beta} of this stripe's sample in LDS, then one pass over the
        // stripe (16-byte chunks, coalesced rows) -> bf16 into the activation buffer
        const unsigned cmean = st.aScr, cscale = st.aScr + C * 4, cbeta = st.aScr + 2 * C * 4;
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
            lds_write4_at(cmean + c * 4, mu);
            lds_write4_at(cscale + c * 4, rsqrtf(fmaxf(var, 0.f) + p.gn_eps) * p.gn_g[c]);
            lds_write4_at(cbeta + c * 4, p.gn_b[c]);
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
            bf16x8 q[6];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                q[k] = lds_read16_at<0>(cmean + (ch * 8 + 4 * k) * 4);
                q[2 + k] = lds_read16_at<0>(cscale + (ch * 8 + 4 * k) * 4);
                q[4 + k] = lds_read16_at<0>(cbeta + (ch * 8 + 4 * k) * 4);
            }
            lds_wait<0>();
#pragma unroll
            for (int k = 0; k < 6; ++k) lds_tie(q[k]);
            float o[8];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const f32x4 mu = __builtin_bit_cast(f32x4, q[k]), sc = __builtin_bit_cast(f32x4, q[2 + k]), be = __builtin_bit_cast(f32x4, q[4 + k]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned w = raw[it][2 * k + (r >> 1)];
                    const float xv = bf2f((bf16_t)((r & 1) ? (w >> 16) : (w & 0xffffu)));
                    o[4 * k + r] = (xv - mu[r]) * sc[r] + be[r];
                }
            }
            const unsigned d = st.aA + row * ARS + ((ch ^ (row & 7)) << 4);
            const u32x2 lo = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])}, hi = {pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
            lds_write8_at<0>(d, lo);
            lds_write8_at<8>(d, hi);
        }
    }
    // ---- p = proj_in(n) + bias -> h_out (the residual stream the tail kernel starts from)
    f32x4 h[2][FNC];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < FNC; ++j) h[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    st.linear24(h, 0);
    st.add_bias24(h, V::BPI);
    st.store_global24(h, p.h_out, p.ld_hout, 0, m0, M);
    // ---- l1 = LN1(p) -> activation buffer (n is dead behind LN's first barrier)
    st.layernorm24(h, V::LN1G, V::LN1B, p.ln_eps);
    // ---- q | k | v: three sweeps over 320 weight rows each; the stacked lora_down rows ride in the first one only
    {
        const int tf = prog->tf[1];
#pragma unroll 1
        for (int c = 0; c < 3; ++c) {
            f32x4 y[2][FNC];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < FNC; ++j) y[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 acct = {0.f, 0.f, 0.f, 0.f};
            st.template sweep24<KS>(y, acct, c == 0 ? tf : 0);
            if (tf) {
                if (c == 0) st.write_t24(acct, tf);
                st.ext24(y);
            }
            st.add_bias24(y, V::BQKV + C * c);
            st.store_global24(y, p.qkv_out, p.ld_qkv, C * c, m0, M);
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

struct HostLin { const leco_xlin* a; int n, tf; };
int check_lin(HostLin& L, const leco_xlin& a, int n, const char* what) {
    if (!a.w) return fail(-EINVAL, "%s: null weight", what);
    if (a.ldw % 8 || (a.dn && (a.ld_dn % 8 || a.ld_up % 8))) return fail(-EINVAL, "%s: operand strides must be multiples of 8 elements", what);
    if (a.dn && (!a.up || (a.t_rows != 16 && a.t_rows != 32))) return fail(-EINVAL, "%s: LoRA needs up and t_rows in {16, 32}", what);
    if ((int64_t)n * a.ldw * 2 >= ((int64_t)1 << 31)) return fail(-EINVAL, "%s: weight too large for a buffer descriptor", what);
    L.a = &a; L.n = n; L.tf = a.dn ? a.t_rows / 16 : 0;
    return 0;
}
// sweep over weight rows [n0, n0 + nt) and k = k0 + 32 [0, ksteps) of Linear `L`; with_dn: its lora_down rows ride in the
// tiles; ext: the K-extension tile (rows [n0, n0 + nt) of scale*up, one k-step) follows as a sweep of its own
void fill_sweep(XProg& pg, int& ns, const HostLin& L, int n0, int nt, int k0, int ksteps, bool with_dn, bool ext) {
    const leco_xlin& a = *L.a;
    XSweep& s = pg.sw[ns++];
    memset(&s, 0, sizeof(s));
    s.w = (const char*)a.w + (int64_t)n0 * a.ldw * 2; s.w_bytes = (unsigned)((int64_t)nt * a.ldw * 2); s.ldw_b = (unsigned)(a.ldw * 2);
    if (with_dn && L.tf) { s.x = a.dn; s.x_bytes = (unsigned)((int64_t)a.t_rows * a.ld_dn * 2); s.ldx_b = (unsigned)(a.ld_dn * 2); s.tf = L.tf; }
    s.nt = nt; s.k0 = k0; s.ksteps = ksteps;
    if (ext && L.tf) {
        XSweep& e = pg.sw[ns++];
        memset(&e, 0, sizeof(e));
        e.w = (const char*)a.up + (int64_t)n0 * a.ld_up * 2; e.w_bytes = (unsigned)((int64_t)nt * a.ld_up * 2); e.ldw_b = (unsigned)(a.ld_up * 2);
        e.nt = nt; e.k0 = 0; e.ksteps = 1;
    }
}
void add_vec(XProg& pg, const float* src, int off, int len) {
    XVec& v = pg.vec[pg.nvec++];
    v.src = src; v.off = off; v.len = len;
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

extern "C" int64_t leco_xblock_prog_bytes(void) {
    const size_t a = sizeof(leco::XTailBlob) > sizeof(leco::XHeadBlob) ? sizeof(leco::XTailBlob) : sizeof(leco::XHeadBlob);
    return (int64_t)((a + 255) / 256 * 256);
}

// Validates `a` and writes the launch description of the tail kernel into `host_prog` (leco_xblock_prog_bytes() bytes of HOST
// memory).  The caller copies it to device memory once; leco_xblock_tail_run launches with it.
extern "C" int leco_xblock_tail_build(const leco_xblock_tail_args* a, void* host_prog, int64_t host_bytes) {
    using namespace leco;
    if (!a || !host_prog) return fail(-EINVAL, "leco_xblock_tail_build: null args");
    if (host_bytes < (int64_t)sizeof(XTailBlob)) return fail(-EINVAL, "leco_xblock_tail_build: program buffer too small");
    if (!leco_xblock_supported(a->c, a->heads, a->skv, a->rows_per_sample))
        return fail(-EINVAL, "leco_xblock_tail: unsupported shape c=%d heads=%d skv=%d rows_per_sample=%d", a->c, a->heads, a->skv,
                    a->rows_per_sample);
    if (a->m <= 0 || !a->attn || !a->h_in || !a->out || !a->kp || !a->vt || !a->ln2_g || !a->ln2_b || !a->ln3_g || !a->ln3_b)
        return fail(-EINVAL, "leco_xblock_tail: null operand");
    if (a->ld_attn % 8 || a->ld_h % 4 || a->ld_out % 8 || (a->res && a->ld_res % 8))
        return fail(-EINVAL, "leco_xblock_tail: activation strides must keep 16-byte alignment");
    if ((int64_t)a->m * a->ld_attn * 2 >= ((int64_t)1 << 31)) return fail(-EINVAL, "leco_xblock_tail: activation too large");
    if (a->col_stats && (a->stats_atom <= 0 || a->c % a->stats_atom || (a->stats_atom & 1)))
        return fail(-EINVAL, "leco_xblock_tail: col_stats needs an even stats_atom that divides c");
    const int C = a->c, F = 4 * C;
    XTailBlob* blob = (XTailBlob*)host_prog;
    memset(blob, 0, sizeof(*blob));
    XProg& pg = blob->prog;
    int rc;
    HostLin L[6];
    if ((rc = check_lin(L[0], a->to_out1, C, "leco_xblock_tail.to_out1"))) return rc;
    if ((rc = check_lin(L[1], a->to_q2, C, "leco_xblock_tail.to_q2"))) return rc;
    if ((rc = check_lin(L[2], a->to_out2, C, "leco_xblock_tail.to_out2"))) return rc;
    if ((rc = check_lin(L[3], a->ff1, 2 * F, "leco_xblock_tail.ff1"))) return rc;
    if ((rc = check_lin(L[4], a->ff2, C, "leco_xblock_tail.ff2"))) return rc;
    const bool has_po = a->proj_out.w != nullptr;
    if (has_po && (rc = check_lin(L[5], a->proj_out, C, "leco_xblock_tail.proj_out"))) return rc;
    for (int l = 0; l < (has_po ? 6 : 5); ++l) pg.tf[l] = L[l].tf;
    using V = XV<320>;
    add_vec(pg, a->to_out1.bias, V::BO1, C);
    add_vec(pg, a->to_q2.bias, V::BQ2, C);
    add_vec(pg, a->to_out2.bias, V::BO2, C);
    add_vec(pg, a->ff2.bias, V::BFF2, C);
    add_vec(pg, has_po ? a->proj_out.bias : nullptr, V::BPO, C);
    add_vec(pg, a->ln2_g, V::LN2G, C);
    add_vec(pg, a->ln2_b, V::LN2B, C);
    add_vec(pg, a->ln3_g, V::LN3G, C);
    add_vec(pg, a->ln3_b, V::LN3B, C);
    add_vec(pg, a->ff1.bias, V::BFF1, 2 * F);
    int ns = 0;
    for (int l = 0; l < 3; ++l) fill_sweep(pg, ns, L[l], 0, C, 0, C / XKT, true, true);
    const int nchunk = F / 128;
    for (int c = 0; c < nchunk; ++c) {
        fill_sweep(pg, ns, L[3], 256 * c, 256, 0, C / XKT, c == 0, true);
        fill_sweep(pg, ns, L[4], 0, C, 128 * c, 4, true, c == nchunk - 1);
    }
    if (has_po) fill_sweep(pg, ns, L[5], 0, C, 0, C / XKT, true, true);
    if (ns > XMAXSW) return fail(-EINVAL, "leco_xblock_tail: sweep table overflow");
    pg.nsweeps = ns;
    XTailArgs& p = blob->p;
    p.m = a->m; p.heads = a->heads; p.skv = a->skv; p.rows_per_sample = a->rows_per_sample;
    p.attn = (const bf16_t*)a->attn; p.attn_bytes = (unsigned)((int64_t)a->m * a->ld_attn * 2); p.ld_attn_b = (unsigned)(a->ld_attn * 2);
    p.h_in = (const bf16_t*)a->h_in; p.ld_h = a->ld_h;
    p.ln_eps = a->ln_eps;
    p.kp = (const bf16_t*)a->kp; p.vt = (const bf16_t*)a->vt; p.scale_log2 = a->attn_scale * 1.4426950408889634f;
    p.res = has_po ? (const bf16_t*)a->res : nullptr; p.ld_res = a->ld_res;
    p.out = (bf16_t*)a->out; p.ld_out = a->ld_out;
    p.col_stats = a->col_stats; p.stats_atom = a->stats_atom;
    p.has_po = has_po ? 1 : 0;
    return 0;
}

extern "C" int leco_xblock_tail_run(const void* dev_prog, int32_t m, int32_t head_dim, leco_stream_t stream) {
    using namespace leco;
    if (!dev_prog || m <= 0) return fail(-EINVAL, "leco_xblock_tail_run: bad arguments");
    const dim3 grid((unsigned)cdiv(m, XBM));
    constexpr int lds_bytes = XCfg<320>::LDS_BYTES;
    if (head_dim == 40) {
        set_lds<0>(&xblock_tail_kernel<320, 40>, lds_bytes);
        hipLaunchKernelGGL((xblock_tail_kernel<320, 40>), grid, dim3(512), lds_bytes, (hipStream_t)stream, dev_prog);
    } else if (head_dim == 64) {
        set_lds<1>(&xblock_tail_kernel<320, 64>, lds_bytes);
        hipLaunchKernelGGL((xblock_tail_kernel<320, 64>), grid, dim3(512), lds_bytes, (hipStream_t)stream, dev_prog);
    } else {
        return fail(-EINVAL, "leco_xblock_tail_run: head_dim %d unsupported (40, 64)", head_dim);
    }
    return check_launch("leco_xblock_tail");
}

// Validates `a` and writes the launch description of the head kernel into `host_prog` (see leco_xblock_tail_build).
extern "C" int leco_xblock_head_build(const leco_xblock_head_args* a, void* host_prog, int64_t host_bytes) {
    using namespace leco;
    if (!a || !host_prog) return fail(-EINVAL, "leco_xblock_head_build: null args");
    if (host_bytes < (int64_t)sizeof(XHeadBlob)) return fail(-EINVAL, "leco_xblock_head_build: program buffer too small");
    if (a->c != 320 || a->rows_per_sample <= 0 || a->rows_per_sample % XBM)
        return fail(-EINVAL, "leco_xblock_head: unsupported shape c=%d rows_per_sample=%d", a->c, a->rows_per_sample);
    if (a->m <= 0 || !a->x || !a->h_out || !a->qkv_out || !a->ln1_g || !a->ln1_b) return fail(-EINVAL, "leco_xblock_head: null operand");
    if (a->ld_x % 8 || a->ld_hout % 4 || a->ld_qkv % 4) return fail(-EINVAL, "leco_xblock_head: activation strides must keep 8-byte alignment");
    if ((int64_t)a->m * a->ld_x * 2 >= ((int64_t)1 << 31)) return fail(-EINVAL, "leco_xblock_head: activation too large");
    if (a->gn_cstats) {
        if (!a->gn_g || !a->gn_b || a->groups <= 0 || a->c % a->groups || a->stats_atom <= 0 || (a->c / a->groups) % a->stats_atom)
            return fail(-EINVAL, "leco_xblock_head: GroupNorm needs gamma / beta, groups | c and stats_atom | c / groups");
    }
    const int C = a->c;
    XHeadBlob* blob = (XHeadBlob*)host_prog;
    memset(blob, 0, sizeof(*blob));
    XProg& pg = blob->prog;
    int rc;
    HostLin L[2];
    if ((rc = check_lin(L[0], a->proj_in, C, "leco_xblock_head.proj_in"))) return rc;
    if ((rc = check_lin(L[1], a->qkv, 3 * C, "leco_xblock_head.qkv"))) return rc;
    for (int l = 0; l < 2; ++l) pg.tf[l] = L[l].tf;
    using V = XV<320>;
    add_vec(pg, a->proj_in.bias, V::BPI, C);
    add_vec(pg, a->qkv.bias, V::BQKV, 3 * C);
    add_vec(pg, a->ln1_g, V::LN1G, C);
    add_vec(pg, a->ln1_b, V::LN1B, C);
    int ns = 0;
    fill_sweep(pg, ns, L[0], 0, C, 0, C / XKT, true, true);
    for (int c = 0; c < 3; ++c) fill_sweep(pg, ns, L[1], C * c, C, 0, C / XKT, c == 0, true);
    pg.nsweeps = ns;
    XHeadArgs& p = blob->p;
    p.m = a->m; p.rows_per_sample = a->rows_per_sample;
    p.x = (const bf16_t*)a->x; p.ld_x = a->ld_x; p.x_bytes = (unsigned)((int64_t)a->m * a->ld_x * 2);
    p.gn_cstats = a->gn_cstats; p.stats_atom = a->stats_atom; p.groups = a->groups; p.gn_g = a->gn_g; p.gn_b = a->gn_b; p.gn_eps = a->gn_eps;
    p.ln_eps = a->ln_eps;
    p.h_out = (bf16_t*)a->h_out; p.ld_hout = a->ld_hout;
    p.qkv_out = (bf16_t*)a->qkv_out; p.ld_qkv = a->ld_qkv;
    return 0;
}

extern "C" int leco_xblock_head_run(const void* dev_prog, int32_t m, leco_stream_t stream) {
    using namespace leco;
    if (!dev_prog || m <= 0) return fail(-EINVAL, "leco_xblock_head_run: bad arguments");
    constexpr int lds_bytes = XCfg<320>::LDS_BYTES;
    set_lds<2>(&xblock_head_kernel<320>, lds_bytes);
    hipLaunchKernelGGL((xblock_head_kernel<320>), dim3((unsigned)cdiv(m, XBM)), dim3(512), lds_bytes, (hipStream_t)stream, dev_prog);
    return check_launch("leco_xblock_head");
}

#ifdef LECO_STRIPE_TIMING
// side builds only (tools/ablate_stripe.py): the phase stamps of workgroup 0 of the last tail launch
extern "C" int leco_xblock_debug_times(unsigned long long* out, int n) {
    if (n > 32) n = 32;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(leco::g_xtimes), sizeof(unsigned long long) * n);
}
#endif
