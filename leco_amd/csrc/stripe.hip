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
constexpr int XNKEY = 80;    // padded prompt keys of K (5 fragments)
constexpr int XNPOS = 96;    // padded (permuted) key positions of V^T

struct XLin {       // per-Linear constants the consumer side needs
    const float* bias;
    int tf;                             // 0: LoRA off; 1 / 2: 16 / 32 stacked lora_down rows
    int pad;
};
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
struct XProg {
    int nsweeps;
    int pad;
    XLin lin[7];
    XSweep sw[XMAXSW];
};

template <int C>
struct XCfg {
    static constexpr int FNC = C / 64;             // 16-column fragments per wave of a [64][C] result (2 x 4 waves)
    static constexpr int WN = C / 4;               // columns per wave
    static constexpr int ARS = C * 2;              // activation buffer row stride (bytes)
    static constexpr int ABUF = XBM * ARS;
    static constexpr int KS = C / XKT;
    static constexpr int SLOT = (C + 32) * 64;     // ring slot: C weight rows + up to 32 lora_down rows, 64 bytes each
    static constexpr int OFF_A = 0, OFF_B = ABUF, OFF_RING = 2 * ABUF, OFF_T = OFF_RING + XNS * SLOT;
    static constexpr int OFF_SCR = OFF_T + XBM * 64, LDS_BYTES = OFF_SCR + 4096;
    static_assert(C % 64 == 0 && (C / 8) % 16 == 8, "activation swizzle assumes a row of 8 (mod 16) 16-byte chunks");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS layout does not fit");
    static_assert(2 * ABUF >= XBM * C * 4, "the fp32 output staging uses both activation buffers");
};

struct XTailArgs {
    int m, heads, skv, rows_per_sample;
    const bf16_t* attn; unsigned attn_bytes, ld_attn_b;
    const bf16_t* h_in; int64_t ld_h;
    const float *ln2_g, *ln2_b, *ln3_g, *ln3_b; float ln_eps;
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
    const float *ln1_g, *ln1_b; float ln_eps;
    bf16_t* h_out; int64_t ld_hout;
    bf16_t* qkv_out; int64_t ld_qkv;
};
struct XHeadBlob { XHeadArgs p; XProg prog; };

// ------------------------------------------------------------------------------------------------------------------
// Shared machinery of the stripe kernels: lane constants, the weight-tile stream, the two GEMM wave layouts, LayerNorm
// on the register-resident residual stream, the staged output store.
// ------------------------------------------------------------------------------------------------------------------
template <int C>
struct Stripe {
    using Cf = XCfg<C>;
    static constexpr int FNC = Cf::FNC, ARS = Cf::ARS, KS = Cf::KS, SLOT = Cf::SLOT, WN = Cf::WN;
    struct F24 { bf16x8 a[2]; bf16x8 w[FNC]; bf16x8 t; };     // fragments of one tile, 2 x 4 wave layout
    struct F18 { bf16x8 a[4]; bf16x8 w[2]; bf16x8 t; };       // 1 x 8 layout (FF1 chunks: value + gate fragment)

    const LECO_CONST_AS XProg* prog;      // sweep table: read through the scalar cache (dynamic index, wave-uniform)
    unsigned char* lds;
    unsigned char *bufA, *bufB, *ring, *tbuf;
    float* scr;
    int lane, wave, fr, fg, wm, wn;
    int swz4;          // byte offset of this lane's k-group inside a 64-byte tile row: ((fg ^ g(row)) << 4)
    int a_sw;          // activation-buffer chunk swizzle of this lane's rows (row & 7 == fr & 7)
    int a_rows24;      // byte offset of row wm * 32 + fr in an activation buffer
    int w_ofs24;       // byte offset of tile row wn * WN + fr (+ swizzled k-group)
    int lrow;          // DMA: tile row inside a 16-row piece
    unsigned csrc16;   // DMA: source byte offset of the 16-byte chunk this lane fetches
    // weight stream: the sweep being issued lives in registers (`cur`), the one after it is prefetched (`nxt`) when `cur`
    // is entered, so the scalar loads of the table never sit on the issue path
    struct SweepRegs {
        buf_rsrc rw, rx;
        unsigned ldw_b, ldx_b;
        int npm, k0, ksteps, tf;
    };
    SweepRegs cur, nxt;
    int s_si = 0, s_kt = 0, s_islot = 0, s_cslot = 0, c_next = 0;

    __device__ __forceinline__ Stripe(const LECO_CONST_AS XProg* pg) : prog(pg) {
        lds = dyn_lds();
        bufA = lds + Cf::OFF_A; bufB = lds + Cf::OFF_B; ring = lds + Cf::OFF_RING; tbuf = lds + Cf::OFF_T;
        scr = (float*)(lds + Cf::OFF_SCR);
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

    __device__ __forceinline__ void load_sweep(SweepRegs& r, int si) const {
        const int i = si < prog->nsweeps ? si : 0;       // (past the end: any valid entry; never issued)
        const LECO_CONST_AS XSweep* sw = &prog->sw[i];
        const void* x = sw->x;
        r.rw = make_rsrc(sw->w, sw->w_bytes);
        r.rx = make_rsrc(x ? x : sw->w, x ? sw->x_bytes : sw->w_bytes);
        r.ldw_b = sw->ldw_b; r.ldx_b = sw->ldx_b;
        r.npm = sw->nt >> 4; r.k0 = sw->k0; r.ksteps = sw->ksteps; r.tf = sw->tf;
    }
    // issues the next tile of the program (if any) into the next ring slot; returns this wave's DMA piece count.
    // Piece pc (16 tile rows, 1 KB): lane l fetches 16 bytes of row 16 pc + l / 4; the row / k part of the address is
    // scalar (soffset), the lane part (row-in-piece x stride + swizzled chunk) a per-lane constant.
    __device__ __forceinline__ int issue_tile() {
        if (s_si >= prog->nsweeps) return 0;
        const int npm = cur.npm, np = npm + cur.tf;
        unsigned char* dst = ring + s_islot * SLOT;
        int cnt = 0;
        const unsigned kb = (unsigned)(cur.k0 + XKT * s_kt) * 2u;
        const unsigned vw = (unsigned)lrow * cur.ldw_b + csrc16, vx = (unsigned)lrow * cur.ldx_b + csrc16;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pc = wave + 8 * i;
            if (pc < np) {
                if (!(LECO_STRIPE_ABLATE & 2)) {
                    if (pc < npm) glds16_buf(cur.rw, vw, kb + (unsigned)(16 * pc) * cur.ldw_b, dst + pc * 1024);
                    else glds16_buf(cur.rx, vx, kb + (unsigned)(16 * (pc - npm)) * cur.ldx_b, dst + pc * 1024);
                    ++cnt;
                }
            }
        }
        if (++s_kt == cur.ksteps) {
            ++s_si;
            s_kt = 0;
            cur = nxt;
            load_sweep(nxt, s_si + 1);
        }
        s_islot = s_islot == XNS - 1 ? 0 : s_islot + 1;
        return cnt;
    }
    __device__ __forceinline__ void start_stream() {
        load_sweep(cur, 0);
        load_sweep(nxt, 1);
        (void)issue_tile();
        c_next = issue_tile();
    }
    // the next tile has landed for every wave and every wave is done with the tile before it, whose slot is refilled with
    // the tile after next.  In flight afterwards: two tiles.  (In-order completion: "at most c_next of my operations
    // outstanding" implies the older pieces of the tile about to be read are complete; other vector-memory operations in
    // between only make the wait stronger.)
    __device__ __forceinline__ const unsigned char* acquire() {
        if (!(LECO_STRIPE_ABLATE & 16)) {
            if (c_next >= 3) wait_vmcnt<3>();
            else if (c_next == 2) wait_vmcnt<2>();
            else if (c_next == 1) wait_vmcnt<1>();
            else wait_vmcnt<0>();
            barrier_keep_dma();
        }
        const unsigned char* s = ring + s_cslot * SLOT;
        s_cslot = s_cslot == XNS - 1 ? 0 : s_cslot + 1;
        return s;
    }
    // refill of the slot the last acquire() freed: called once after every acquire(), behind the fragment reads and the
    // MFMAs the caller had ready (the DMA instructions occupy the wave for ~100 cycles each)
    __device__ __forceinline__ void refill() { c_next = issue_tile(); }

    // ---- activation buffers ------------------------------------------------------------------------------------------
    __device__ __forceinline__ int a_chunk(int chunk) const { return (chunk ^ a_sw) << 4; }
    // re-derives the lane's chunk swizzle behind an optimisation barrier: the per-k-step chunk offsets are the same in every
    // sweep, and hipcc otherwise keeps all of them live through the whole kernel (and spills)
    __device__ __forceinline__ void fresh_swizzle() { opaque(a_sw); }
    // lane's 4 consecutive columns n .. n + 3 (n % 4 == 0) of row-offset `rowb` as bf16
    __device__ __forceinline__ void put4(unsigned char* buf, int rowb, int n, float v0, float v1, float v2, float v3) const {
        const u32x2 w = {pack_bf2(v0, v1), pack_bf2(v2, v3)};
        lds_write8_async(buf + rowb + a_chunk(n >> 3) + ((n & 4) << 1), w);
    }
    // a [64][C] accumulator set (2 x 4 layout) as bf16 into an activation buffer
    __device__ __forceinline__ void store24(unsigned char* buf, const f32x4 (&v)[2][FNC]) const {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < FNC; ++j)
                put4(buf, a_rows24 + 16 * i * ARS, wn * WN + 16 * j + 4 * fg, v[i][j][0], v[i][j][1], v[i][j][2], v[i][j][3]);
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
    // DMA of a [64][C] bf16 stripe (rows m0 .. m0 + 63 of a row-major matrix; rows >= m: zeros) into an activation buffer
    __device__ __forceinline__ void load_stripe(unsigned char* buf, const void* base, unsigned bytes, unsigned ld_b, int m0, int m) {
        const buf_rsrc r = make_rsrc(base, bytes);
        constexpr int NCH = C / 8, NP = Cf::ABUF / 1024;
#pragma unroll
        for (int i = 0; i < (NP + 7) / 8; ++i) {
            const int pc = wave + 8 * i;
            if (pc < NP) {
                const int u = 64 * pc + lane;
                const int row = (int)(((float)u + 0.5f) * (1.0f / (float)NCH)), pos = u - row * NCH;
                const unsigned voff = m0 + row < m ? (unsigned)(m0 + row) * ld_b + (unsigned)((pos ^ (row & 7)) << 4) : DMA_OOB;
                glds16_buf(r, voff, 0u, buf + pc * 1024);
            }
        }
    }

    // ---- 2 x 4 layout: wave (wm, wn) owns rows wm * 32 .. + 32, tile rows wn * WN .. + WN ---------------------------------
    __device__ __forceinline__ void read24(F24& f, const unsigned char* abuf, int ka, const unsigned char* s, int t_ofs, bool has_t) const {
        if (LECO_STRIPE_ABLATE & 4) return;
        const int ao = a_chunk(4 * ka + fg);
        f.a[0] = lds_read16_async(abuf + a_rows24 + ao);
        f.a[1] = lds_read16_async(abuf + a_rows24 + 16 * ARS + ao);
#pragma unroll
        for (int j = 0; j < FNC; ++j) f.w[j] = lds_read16_async(s + w_ofs24 + 1024 * j);
        if (has_t) f.t = lds_read16_async(s + t_ofs);
    }
    __device__ __forceinline__ void tie24(F24& f, bool has_t) const {
        lds_tie(f.a[0]); lds_tie(f.a[1]);
#pragma unroll
        for (int j = 0; j < FNC; ++j) lds_tie(f.w[j]);
        if (has_t) lds_tie(f.t);
    }
    __device__ __forceinline__ void mma24(f32x4 (&acc)[2][FNC], f32x4& acct, const F24& f, bool has_t, int ti) const {
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
        if (has_t) acct = mfma16(f.t, ti ? f.a[1] : f.a[0], acct);
    }
    // one sweep of NK main tiles: acc += A[:, 32 ka0 ..] W^T, acct += A t_w^T (waves on T duty).  `nt` = weight rows per tile.
    template <int NK>
    __device__ __forceinline__ void sweep24(f32x4 (&acc)[2][FNC], f32x4& acct, const unsigned char* abuf, int ka0, int nt, int tf) {
        const bool has_t = tf == 2 || (tf == 1 && wn < 2);
        const int ti = tf == 2 ? (wn & 1) : wn, tq = tf == 2 ? (wn >> 1) : 0;
        const int t_ofs = (nt + 16 * tq + fr) * 64 + swz4;
        F24 f[2];
        f[0].t = f[1].t = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        fresh_swizzle();
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            const unsigned char* s = acquire();       // (its barrier completed the previous tile's fragment reads)
            if (kt > 0) tie24(f[(kt - 1) & 1], has_t);
            read24(f[kt & 1], abuf, ka0 + kt, s, t_ofs, has_t);
            if (kt > 0) mma24(acc, acct, f[(kt - 1) & 1], has_t, ti);
            refill();
        }
        lds_wait<0>();
        tie24(f[(NK - 1) & 1], has_t);
        mma24(acc, acct, f[(NK - 1) & 1], has_t, ti);
    }
    // T (fp32, waves on T duty) -> bf16 into the T buffer ([64][32], 64-byte rows, tile swizzle)
    __device__ __forceinline__ void put_t(const f32x4& acct, int row, int tq) const {
        const u32x2 w = {pack_bf2(acct[0], acct[1]), pack_bf2(acct[2], acct[3])};
        const int g4 = (4 - (fr >> 2)) & 3;
        lds_write8_async(tbuf + row * 64 + (((2 * tq + (fg >> 1)) ^ g4) << 4) + ((fg & 1) << 3), w);
    }
    __device__ __forceinline__ void write_t24(const f32x4& acct, int tf) const {
        const bool has_t = tf == 2 || (tf == 1 && wn < 2);
        const int ti = tf == 2 ? (wn & 1) : wn, tq = tf == 2 ? (wn >> 1) : 0;
        if (has_t) put_t(acct, wm * 32 + 16 * ti + fr, tq);
    }
    // K-extension tile: acc += T (scale up)^T.  T must have been written (by any layout) before the call.
    __device__ __forceinline__ void ext24(f32x4 (&acc)[2][FNC]) {
        const unsigned char* s = acquire();           // barrier: T visible; the [N][32] image of scale*up landed
        F24 f;
        f.a[0] = lds_read16_async(tbuf + (wm * 32 + fr) * 64 + swz4);
        f.a[1] = lds_read16_async(tbuf + (wm * 32 + 16 + fr) * 64 + swz4);
#pragma unroll
        for (int j = 0; j < FNC; ++j) f.w[j] = lds_read16_async(s + w_ofs24 + 1024 * j);
        lds_wait<0>();
        tie24(f, false);
        f32x4 dummy = {0.f, 0.f, 0.f, 0.f};
        mma24(acc, dummy, f, false, 0);
        refill();
    }
    // a whole Linear on the 2 x 4 layout: K = C, A = abuf, accumulated into acc (bias NOT added)
    __device__ __forceinline__ void linear24(f32x4 (&acc)[2][FNC], const unsigned char* abuf, int lin) {
        const int tf = prog->lin[lin].tf;
        f32x4 acct = {0.f, 0.f, 0.f, 0.f};
        sweep24<KS>(acc, acct, abuf, 0, C, tf);
        if (tf) {
            write_t24(acct, tf);
            ext24(acc);
        }
    }
    __device__ __forceinline__ void add_bias24(f32x4 (&acc)[2][FNC], const float* bias, int n0) const {
        if (!bias) return;
#pragma unroll
        for (int j = 0; j < FNC; ++j) {
            const f32x4 b = *(const f32x4*)(bias + n0 + wn * WN + 16 * j + 4 * fg);
#pragma unroll
            for (int i = 0; i < 2; ++i) { acc[i][j][0] += b[0]; acc[i][j][1] += b[1]; acc[i][j][2] += b[2]; acc[i][j][3] += b[3]; }
        }
    }

    // ---- LayerNorm of the register-resident stream (2 x 4 layout) -> bf16 activation buffer -------------------------------
    __device__ __forceinline__ void layernorm24(const f32x4 (&h)[2][FNC], const float* gamma, const float* beta, float eps,
                                                unsigned char* dst) {
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
        if (fg == 0) {
            lds_write4_async(scr + (wm * 32 + fr) * 4 + wn, part[0]);
            lds_write4_async(scr + (wm * 32 + 16 + fr) * 4 + wn, part[1]);
        }
        barrier_keep_dma();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x8 raw = lds_read16_async(scr + (wm * 32 + 16 * i + fr) * 4);
            lds_wait<0>();
            lds_tie(raw);
            const f32x4 q = __builtin_bit_cast(f32x4, raw);
            mean[i] = ((q[0] + q[1]) + (q[2] + q[3])) * (1.0f / (float)C);
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
        float* scr2 = scr + XBM * 4;
        if (fg == 0) {
            lds_write4_async(scr2 + (wm * 32 + fr) * 4 + wn, part[0]);
            lds_write4_async(scr2 + (wm * 32 + 16 + fr) * 4 + wn, part[1]);
        }
        barrier_keep_dma();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x8 raw = lds_read16_async(scr2 + (wm * 32 + 16 * i + fr) * 4);
            lds_wait<0>();
            lds_tie(raw);
            const f32x4 q = __builtin_bit_cast(f32x4, raw);
            rstd[i] = rsqrtf(((q[0] + q[1]) + (q[2] + q[3])) * (1.0f / (float)C) + eps);
        }
#pragma unroll
        for (int j = 0; j < FNC; ++j) {
            const int n = wn * WN + 16 * j + 4 * fg;
            const f32x4 g = *(const f32x4*)(gamma + n), b = *(const f32x4*)(beta + n);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                put4(dst, a_rows24 + 16 * i * ARS, n, (h[i][j][0] - mean[i]) * rstd[i] * g[0] + b[0],
                     (h[i][j][1] - mean[i]) * rstd[i] * g[1] + b[1], (h[i][j][2] - mean[i]) * rstd[i] * g[2] + b[2],
                     (h[i][j][3] - mean[i]) * rstd[i] * g[3] + b[3]);
        }
    }

    // ---- staged output: the [64][C] fp32 result (2 x 4 layout) goes through LDS (both activation buffers, 16-byte chunks
    // XOR-swizzled by row & 15) so that the global side moves whole 16-byte row segments: + bias + residual, bf16 store,
    // and the per-atom {sum, sumsq} of the stored values for the GroupNorm that follows (leco_gemm_args.col_stats).
    // Must be called by all waves; no DMA may target the activation buffers or the ring afterwards.
    __device__ __forceinline__ void store_out(const f32x4 (&v)[2][FNC], const float* bias, const bf16_t* res, int64_t ld_res,
                                              bf16_t* out, int64_t ld_out, int m0, int m, float* col_stats, int stats_atom,
                                              int rows_per_sample) {
        constexpr int NCH = C / 8, RG = 512 / NCH;        // 16-byte output chunks per row; row groups (threads NCH * RG)
        unsigned char* stg = lds;                         // fp32 [64][C], row stride 4 C bytes
        barrier_keep_dma();                               // every wave is done reading the activation buffers
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < FNC; ++j) {
                const int row = wm * 32 + 16 * i + fr, c16 = (wn * WN + 16 * j + 4 * fg) >> 2;
                const u32x4 raw = __builtin_bit_cast(u32x4, v[i][j]);
                const u32x2 lo = {raw[0], raw[1]}, hi = {raw[2], raw[3]};
                unsigned char* d = stg + row * (4 * C) + ((c16 ^ (row & 15)) << 4);
                lds_write8_async(d, lo);
                lds_write8_async(d + 8, hi);
            }
        barrier_keep_dma();
        const int tid = (int)threadIdx.x;
        const int ch = tid % NCH, rg = tid / NCH;
        float s1[8], s2[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
        if (rg < RG) {
            constexpr int NIT = (XBM + RG - 1) / RG;
            float bs[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) bs[r] = bias ? bias[ch * 8 + r] : 0.f;
            u32x4 rres[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = rg + RG * it;
                rres[it] = u32x4{0u, 0u, 0u, 0u};
                if (res && row < XBM && m0 + row < m) rres[it] = *(const u32x4*)(res + (int64_t)(m0 + row) * ld_res + ch * 8);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = rg + RG * it;
                if (row >= XBM || m0 + row >= m) continue;
                bf16x8 r0 = lds_read16_async(stg + row * (4 * C) + (((2 * ch) ^ (row & 15)) << 4));
                bf16x8 r1 = lds_read16_async(stg + row * (4 * C) + (((2 * ch + 1) ^ (row & 15)) << 4));
                lds_wait<0>();
                lds_tie(r0);
                lds_tie(r1);
                const f32x4 v0 = __builtin_bit_cast(f32x4, r0), v1 = __builtin_bit_cast(f32x4, r1);
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
        // column sums: row groups meet in the (idle) ring region, fp32 [RG][C][2]; one thread per atom sends one pair of atomics
        float* red = (float*)ring;
        if (rg < RG) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                lds_write4_async(red + (rg * C + ch * 8 + r) * 2, s1[r]);
                lds_write4_async(red + (rg * C + ch * 8 + r) * 2 + 1, s2[r]);
            }
        }
        barrier_keep_dma();
        const int natom = C / stats_atom;
        if (tid < natom && m0 < m) {
            float t1 = 0.f, t2 = 0.f;
            for (int c = tid * stats_atom; c < (tid + 1) * stats_atom; ++c)
                for (int g = 0; g < RG; ++g) {
                    bf16x8 raw = lds_read16_async(red + ((g * C + (c & ~1)) * 2));     // columns c & ~1, (c & ~1) + 1
                    lds_wait<0>();
                    lds_tie(raw);
                    const f32x4 q = __builtin_bit_cast(f32x4, raw);
                    t1 += (c & 1) ? q[2] : q[0];
                    t2 += (c & 1) ? q[3] : q[1];
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
    constexpr int FNC = Cf::FNC, ARS = Cf::ARS, KS = Cf::KS, WN = Cf::WN;
    constexpr int DV = (D + 15) / 16 * 16, NFD = DV / 16;
    constexpr bool ONES = DV > D;
    constexpr int NCHUNK = 4 * C / 128;             // feed-forward chunks of 128 hidden units
    const LECO_CONST_AS XTailBlob* B = LECO_CONST_CAST(XTailBlob, blob);
    const LECO_CONST_AS XTailArgs& p = B->p;
    const LECO_CONST_AS XProg* prog = &B->prog;
    St st(prog);
    const int lane = st.lane, wave = st.wave, fr = st.fr, fg = st.fg, wm = st.wm, wn = st.wn;
    const int m0 = (int)blockIdx.x * XBM, M = p.m;
    (void)lane;

    // ---- prologue: the self-attention output stripe -> bufA, the first two weight tiles, the T buffer zeroed (its columns
    // 16 .. 31 are only written by rank-stacks > 16), the residual stream h0 -> registers
    st.load_stripe(st.bufA, p.attn, p.attn_bytes, p.ld_attn_b, m0, M);
    st.start_stream();
    {
        const u32x2 z = {0u, 0u};
        lds_write8_async(st.tbuf + (int)threadIdx.x * 8, z);
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
    st.linear24(h, st.bufA, 0);
    XSTAMP(1);
    st.add_bias24(h, prog->lin[0].bias, 0);
    // ---- 2. l2 = LN2(h1) -> bufB
    st.layernorm24(h, p.ln2_g, p.ln2_b, p.ln_eps, st.bufB);
    XSTAMP(2);
    // ---- 3. q2 = l2 Wq2^T -> bufA (bf16)
    {
        f32x4 q[2][FNC];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < FNC; ++j) q[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        st.linear24(q, st.bufB, 1);
        st.add_bias24(q, prog->lin[1].bias, 0);
        st.store24(st.bufA, q);     // (bufA: last read by sweep 0, many barriers ago)
    }
    barrier_keep_dma();             // q2 complete; every wave is done reading l2 (bufB is free)
    XSTAMP(3);

    // ---- 4. cross-attention: one wave per head; S^T = K Q^T and O^T = V^T P^T (swapped, a lane owns one query row)
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
                    qf[ks] = lds_read16_async(st.bufA + (16 * u + fr) * ARS + st.a_chunk(chunk));
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
                    if (d < D) st.put4(st.bufB, (16 * u + fr) * ARS, hd * D + d, o[fd][0] * inv, o[fd][1] * inv, o[fd][2] * inv, o[fd][3] * inv);
                }
            }
        }
    }
    // ---- 5. h2 = a2 Wo2^T + bo2 + h1   (the first acquire's barrier publishes a2)
    XSTAMP(5);
    st.linear24(h, st.bufB, 2);
    XSTAMP(6);
    st.add_bias24(h, prog->lin[2].bias, 0);
    // ---- 6. l3 = LN3(h2) -> bufA  (bufA: q2, last read before sweep 2's first barrier)
    st.layernorm24(h, p.ln3_g, p.ln3_b, p.ln_eps, st.bufA);
    st.add_bias24(h, prog->lin[4].bias, 0);          // ff.net.2 bias: h becomes the accumulator of h3
    XSTAMP(7);

    // ---- 7. feed-forward in chunks of 128 hidden units.  FF1 chunk: 1 x 8 layout, wave w owns all 64 rows of hidden columns
    // 16 w .. + 16 of the chunk: tile rows vb + fr (value) and vb + 64 + fr (gate) of the 64-interleaved GEGLU weight image.
    {
        const int tf1 = prog->lin[3].tf, tf2 = prog->lin[4].tf;
        const bool has_t1 = tf1 == 2 || (tf1 == 1 && wave < 4);
        const int ti1 = wave & 3, tq1 = tf1 == 2 ? (wave >> 2) : 0;
        const bool has_t2 = tf2 == 2 || (tf2 == 1 && wn < 2);
        const int ti2 = tf2 == 2 ? (wn & 1) : wn, tq2 = tf2 == 2 ? (wn >> 1) : 0;
        const int vb = 16 * wave + (wave >= 4 ? 64 : 0);
        const int wv_ofs = (vb + fr) * 64 + st.swz4;
        const int t1_ofs = (256 + 16 * tq1 + fr) * 64 + st.swz4;
        const int t2_ofs = (C + 16 * tq2 + fr) * 64 + st.swz4;
        const int a_rows18 = fr * ARS;
        const float* b1 = prog->lin[3].bias;
        f32x4 acct2 = {0.f, 0.f, 0.f, 0.f};
        typename St::F18 f1[2];
        f1[0].t = f1[1].t = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        auto read18 = [&](typename St::F18& f, int ka, const unsigned char* s, bool with_t) {
            if (LECO_STRIPE_ABLATE & 4) return;
            const int ao = st.a_chunk(4 * ka + fg);
#pragma unroll
            for (int i = 0; i < 4; ++i) f.a[i] = lds_read16_async(st.bufA + a_rows18 + 16 * i * ARS + ao);
            f.w[0] = lds_read16_async(s + wv_ofs);
            f.w[1] = lds_read16_async(s + wv_ofs + 64 * 64);
            if (with_t) f.t = lds_read16_async(s + t1_ofs);
        };
        auto tie18 = [&](typename St::F18& f, bool with_t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) lds_tie(f.a[i]);
            lds_tie(f.w[0]);
            lds_tie(f.w[1]);
            if (with_t) lds_tie(f.t);
        };
#pragma unroll 1
        for (int c = 0; c < NCHUNK; ++c) {
            f32x4 u[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { u[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; u[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            f32x4 acct1 = {0.f, 0.f, 0.f, 0.f};
            const bool with_t = has_t1 && c == 0;        // the lora_down rows ride in the first chunk's tiles only
            st.fresh_swizzle();
            auto mma18 = [&](const typename St::F18& f) {
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
                if (with_t) {
                    const bf16x8 a = ti1 == 0 ? f.a[0] : (ti1 == 1 ? f.a[1] : (ti1 == 2 ? f.a[2] : f.a[3]));
                    acct1 = mfma16(f.t, a, acct1);
                }
            };
#pragma unroll
            for (int kt = 0; kt < KS; ++kt) {
                const unsigned char* s = st.acquire();
                if (kt > 0) tie18(f1[(kt - 1) & 1], with_t);
                read18(f1[kt & 1], kt, s, with_t);
                if (kt > 0) mma18(f1[(kt - 1) & 1]);
                st.refill();
            }
            lds_wait<0>();
            tie18(f1[(KS - 1) & 1], with_t);
            mma18(f1[(KS - 1) & 1]);
            if (tf1) {
                if (with_t) st.put_t(acct1, 16 * ti1 + fr, tq1);
                const unsigned char* s = st.acquire();      // barrier: T visible (chunk 0) / still there (later chunks)
                typename St::F18 f;
#pragma unroll
                for (int i = 0; i < 4; ++i) f.a[i] = lds_read16_async(st.tbuf + (16 * i + fr) * 64 + st.swz4);
                f.w[0] = lds_read16_async(s + wv_ofs);
                f.w[1] = lds_read16_async(s + wv_ofs + 64 * 64);
                lds_wait<0>();
                tie18(f, false);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    u[i][0] = mfma16(f.w[0], f.a[i], u[i][0]);
                    u[i][1] = mfma16(f.w[1], f.a[i], u[i][1]);
                }
                st.refill();
            }
            if (c == 0) XSTAMP(11);
            // GEGLU: value * gelu(gate) -> bf16 chunk [64][128] in bufB (columns 0 .. 127)
            {
                f32x4 bv = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
                if (b1) {
                    bv = *(const f32x4*)(b1 + 256 * c + vb + 4 * fg);
                    bg = *(const f32x4*)(b1 + 256 * c + vb + 64 + 4 * fg);
                }
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
                    st.put4(st.bufB, (16 * i + fr) * ARS, 16 * wave + 4 * fg, g[0], g[1], g[2], g[3]);
                }
            }
            if (c == 0) XSTAMP(12);
            // FF2 partial sums: h += g[:, chunk] W2[:, 128 c .. + 128]^T   (the first acquire's barrier publishes the chunk)
            {
                typename St::F24 f2[2];
                f2[0].t = f2[1].t = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                st.fresh_swizzle();
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const unsigned char* s = st.acquire();
                    if (kt > 0) st.tie24(f2[(kt - 1) & 1], has_t2);
                    st.read24(f2[kt & 1], st.bufB, kt, s, t2_ofs, has_t2);
                    if (kt > 0) st.mma24(h, acct2, f2[(kt - 1) & 1], has_t2, ti2);
                    st.refill();
                }
                lds_wait<0>();
                st.tie24(f2[1], has_t2);
                st.mma24(h, acct2, f2[1], has_t2, ti2);
            }
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
        barrier_keep_dma();                     // every wave is done with the last FF1 reads of l3 (bufA)
        st.store24(st.bufA, h);
        f32x4 y[2][FNC];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < FNC; ++j) y[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        st.linear24(y, st.bufA, 5);
        XSTAMP(9);
        st.store_out(y, prog->lin[5].bias, p.res, p.ld_res, p.out, p.ld_out, m0, M, p.col_stats, p.stats_atom, p.rows_per_sample);
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
__global__ __launch_bounds__(512) void xblock_head_kernel(const void* blob) {
    using St = Stripe<C>;
    using Cf = XCfg<C>;
    constexpr int FNC = Cf::FNC, ARS = Cf::ARS, KS = Cf::KS;
    const LECO_CONST_AS XHeadBlob* B = LECO_CONST_CAST(XHeadBlob, blob);
    const LECO_CONST_AS XHeadArgs& p = B->p;
    const LECO_CONST_AS XProg* prog = &B->prog;
    St st(prog);
    const int m0 = (int)blockIdx.x * XBM, M = p.m;
    const int tid = (int)threadIdx.x;

    // (the stripe's DMA pieces must be OLDER than the weight tiles: the first acquire's counted wait then covers them)
    if (!p.gn_cstats) st.load_stripe(st.bufA, p.x, p.x_bytes, (unsigned)(p.ld_x * 2), m0, M);
    st.start_stream();
    {
        const u32x2 z = {0u, 0u};
        lds_write8_async(st.tbuf + tid * 8, z);
    }
    if (p.gn_cstats) {
        // ---- GroupNorm apply: per-channel {mean, rstd * gamma, beta} of this stripe's sample in LDS, then one pass over the
        // stripe (16-byte chunks, coalesced rows) -> bf16 into bufA
        float* cmean = st.scr;              // [C]
        float* cscale = st.scr + C;         // [C]
        float* cbeta = st.scr + 2 * C;      // [C]
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
            lds_write4_async(cmean + c, mu);
            lds_write4_async(cscale + c, rsqrtf(fmaxf(var, 0.f) + p.gn_eps) * p.gn_g[c]);
            lds_write4_async(cbeta + c, p.gn_b[c]);
        }
        barrier_keep_dma();
        constexpr int NCH = C / 8, NIT = (XBM * NCH + 511) / 512;
        u32x4 raw[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + 512 * it, row = e / NCH, ch = e - row * NCH;
            raw[it] = u32x4{0u, 0u, 0u, 0u};
            if (e < XBM * NCH && m0 + row < M) raw[it] = *(const u32x4*)(p.x + (int64_t)(m0 + row) * p.ld_x + ch * 8);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + 512 * it, row = e / NCH, ch = e - row * NCH;
            if (e >= XBM * NCH) break;
            bf16x8 q[6];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                q[k] = lds_read16_async(cmean + ch * 8 + 4 * k);
                q[2 + k] = lds_read16_async(cscale + ch * 8 + 4 * k);
                q[4 + k] = lds_read16_async(cbeta + ch * 8 + 4 * k);
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
            unsigned char* d = st.bufA + row * ARS + ((ch ^ (row & 7)) << 4);
            const u32x2 lo = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])}, hi = {pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
            lds_write8_async(d, lo);
            lds_write8_async(d + 8, hi);
        }
    }
    // ---- p = proj_in(n) + bias -> h_out (the residual stream the tail kernel starts from)
    f32x4 h[2][FNC];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < FNC; ++j) h[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    st.linear24(h, st.bufA, 0);
    st.add_bias24(h, prog->lin[0].bias, 0);
    st.store_global24(h, p.h_out, p.ld_hout, 0, m0, M);
    // ---- l1 = LN1(p) -> bufB
    st.layernorm24(h, p.ln1_g, p.ln1_b, p.ln_eps, st.bufB);
    // ---- q | k | v: three sweeps over 320 weight rows each; the stacked lora_down rows ride in the first one only
    {
        const int tf = prog->lin[1].tf;
        const float* bias = prog->lin[1].bias;
#pragma unroll 1
        for (int c = 0; c < 3; ++c) {
            f32x4 y[2][FNC];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < FNC; ++j) y[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 acct = {0.f, 0.f, 0.f, 0.f};
            st.template sweep24<KS>(y, acct, st.bufB, 0, C, c == 0 ? tf : 0);
            if (tf) {
                if (c == 0) st.write_t24(acct, tf);
                st.ext24(y);
            }
            st.add_bias24(y, bias, C * c);
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
    for (int l = 0; l < (has_po ? 6 : 5); ++l) { pg.lin[l].bias = L[l].a->bias; pg.lin[l].tf = L[l].tf; }
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
    p.ln2_g = a->ln2_g; p.ln2_b = a->ln2_b; p.ln3_g = a->ln3_g; p.ln3_b = a->ln3_b; p.ln_eps = a->ln_eps;
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
    for (int l = 0; l < 2; ++l) { pg.lin[l].bias = L[l].a->bias; pg.lin[l].tf = L[l].tf; }
    int ns = 0;
    fill_sweep(pg, ns, L[0], 0, C, 0, C / XKT, true, true);
    for (int c = 0; c < 3; ++c) fill_sweep(pg, ns, L[1], C * c, C, 0, C / XKT, c == 0, true);
    pg.nsweeps = ns;
    XHeadArgs& p = blob->p;
    p.m = a->m; p.rows_per_sample = a->rows_per_sample;
    p.x = (const bf16_t*)a->x; p.ld_x = a->ld_x; p.x_bytes = (unsigned)((int64_t)a->m * a->ld_x * 2);
    p.gn_cstats = a->gn_cstats; p.stats_atom = a->stats_atom; p.groups = a->groups; p.gn_g = a->gn_g; p.gn_b = a->gn_b; p.gn_eps = a->gn_eps;
    p.ln1_g = a->ln1_g; p.ln1_b = a->ln1_b; p.ln_eps = a->ln_eps;
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
