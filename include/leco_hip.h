/* libleco_hip.so -- C ABI of the MI355X (gfx950) LECO training hot path.
 *
 * The reference (p1atdev/LECO) has no FFI: its per-step arithmetic is
 * diffusers' UNet2DConditionModel.forward / autograd / DDIMScheduler.step /
 * torch.optim.AdamW called from train_util.py:142-193 and train_lora.py:141-290.
 * Each entry point below names the reference call it replaces.  Conventions:
 *   - all pointers are DEVICE pointers owned by the caller (PyTorch allocations);
 *   - activations are bf16, channels-last ([B][H][W][C] == [B*H*W tokens][C]);
 *   - every function only ENQUEUES work on `stream` (a hipStream_t) and never
 *     synchronises, so a sequence of calls can be captured into a hipGraph;
 *   - return 0 on success, a negative errno-style code on bad arguments or a
 *     failed launch; leco_last_error() gives the message (thread-local).
 * No torch types cross this boundary.
 */
#ifndef LECO_HIP_H
#define LECO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* leco_stream_t; /* hipStream_t */

int leco_version(void);
const char* leco_last_error(void);

/* ------------------------------------------------------------------------
 * Fused GEMM / implicit-GEMM 3x3 convolution, bf16 MFMA with fp32 accumulation, with the
 * rank-r LoRA product folded into the same accumulator as one extra K tile
 * ("K-extension": A_ext = x*A^T, W_ext = scale*B):
 *   C[m][n] = act( sum_k A[m][k] W[n][k] + sum_j A_ext[m][j] W_ext[n][j]
 *                  + bias[n] + rowbias[m/rows_per_group][n] + residual[m][n] )
 * Replaces: every nn.Linear / nn.Conv2d forward inside diffusers' UNet (call site
 * train_util.py:156-160) including the LoRA branch lora.py:102-106 (org_forward(x) +
 * lora_up(lora_down(x))*multiplier*scale) and -- with pre-transposed weights -- their
 * autograd dgrad (train_lora.py:279).
 * ---------------------------------------------------------------------- */
enum {
    LECO_A_PLAIN = 0,      /* A is [M][K] row-major (optionally split over two sources) */
    LECO_A_CONV3_S1 = 1,   /* 3x3 pad 1 stride 1 on NHWC input, K = 9*Cin, k = tap*Cin + c */
    LECO_A_CONV3_S2 = 2,   /* 3x3 pad 1 stride 2 (Downsample2D) */
    LECO_A_CONV3_UP2 = 3,  /* 3x3 pad 1 on the nearest-2x upsampled input (Upsample2D) */
    LECO_A_CONV3_TR2 = 4   /* transposed stride-2 gather: dgrad of LECO_A_CONV3_S2 */
};
/* LECO_ACT_GEGLU (leco_gemm only): the GEMM is the GEGLU input projection (diffusers GEGLU.proj, N = 2F) with its
 * weight / bias / w_ext ROWS interleaved in blocks of 64 -- row 128 j + r holds value row 64 j + r (r < 64) or gate
 * row F + 64 j + (r - 64) (r >= 64) -- so one 128-column tile owns a value block and its gate block: the epilogue
 * writes value * gelu(gate) (erf form) as bf16 [M][F] (row stride ldc) and the [M][2F] intermediate never
 * exists.  Needs n % 128 == 0, c != NULL, no residual / rowbias / c_f32, no split-K. */
enum { LECO_ACT_NONE = 0, LECO_ACT_SILU = 1, LECO_ACT_GEGLU = 2 };

typedef struct leco_gemm_args {
    const void* a0;       /* bf16 */
    const void* a1;       /* bf16, second channel/K source (skip-connection concat) or NULL */
    int64_t lda0, lda1;   /* plain: row strides (elements); conv: per-pixel channel strides */
    int32_t k_split;      /* plain: columns [0,k_split) come from a0; conv: channels [0,k_split) */
    int32_t a_mode;
    int32_t batch, h_out, w_out, h_in, w_in; /* conv modes only */
    const void* w;        /* bf16 [N][K], K contiguous; conv: [N][3][3][Cin] */
    int64_t ldw;
    int32_t m, n, k;      /* k % 64 == 0 (conv: Cin % 64 == 0), n % 8 == 0 */
    const void* a_ext;    /* bf16 [M][ext_k] (row stride ld_aext) or NULL */
    int64_t ld_aext;
    const void* w_ext;    /* bf16 [N][ext_k] (row stride ld_wext) */
    int64_t ld_wext;
    int32_t ext_k;        /* 0, 32 or 64 */
    const float* bias;        /* fp32 [N] or NULL */
    const float* rowbias;     /* fp32 [M/rows_per_group][N] or NULL (time-embedding add) */
    int32_t rows_per_group;
    int64_t ld_rowbias;       /* row stride of rowbias (elements) */
    const void* residual;     /* bf16 [M][N] (row stride ldr) or NULL */
    int64_t ldr;
    int32_t act;
    void* c;                  /* bf16 out, row stride ldc (or NULL) */
    int64_t ldc;
    float* c_f32;             /* optional fp32 copy of the pre-rounding result, row stride ldc32 */
    int64_t ldc32;
    /* Fused LoRA down-projection (plain A only; exclusive with a_ext): t_w = stacked lora_down rows, bf16
     * [32][K] (rows >= groups*r zero), row stride ld_tw; t_rows = 16 or 32 = how many of them are non-zero
     * (rounded up to 16).  The kernel forms T = A t_w^T during its own K sweep, rounds it to bf16 and uses it
     * as the A side of the K-extension against w_ext (ext_k must be 32).  t_out (optional, bf16 [M][32], row
     * stride ld_tout) receives T for the backward (lora_up wgrad); it is REQUIRED scratch when the launch
     * heuristic may split K (then T is computed by a separate skinny GEMM into t_out). */
    const void* t_w;
    int64_t ld_tw;
    int32_t t_rows;
    void* t_out;
    int64_t ld_tout;
    /* GroupNorm statistics emitted by the PRODUCER of a tensor (diffusers' GroupNorm re-reads the whole tensor for them;
     * `north_star`: conv2d / GroupNorm fused): col_stats != NULL accumulates {sum, sum of squares} of the bf16-ROUNDED
     * values this call stores to c, per sample and per ATOM of stats_atom adjacent output columns -- fp32
     * [m / stats_rows][n / stats_atom][2], fp32 atomics (a global fp32 atomic costs ~12 ns at its memory channel: one per
     * atom and tile, not one per column), into a buffer the caller has zeroed.  stats_rows = rows (pixels) per sample;
     * stats_atom divides every GroupNorm group the tensor will be part of (SD: 320 / 32 = 10 channels).
     * leco_groupnorm_apply_stats consumes them.  Needs a bf16 output; not with LECO_ACT_GEGLU. */
    float* col_stats;
    int32_t stats_rows;
    int32_t stats_atom;
} leco_gemm_args;

int leco_gemm(const leco_gemm_args* args, leco_stream_t stream);

/* Re-pack every LoRA site (1..3 LoRA modules sharing an input: fused q|k|v) from the bf16
 * shadow of the trainable slab into the four MFMA operand images the hot path consumes:
 *   dn_s [R16][K]  = stacked lora_down rows (zero rows up to R16)          fwd: T = x dn_s^T
 *   up_p [N][Rp]   = scale * lora_up, block-diagonal over groups            fwd K-extension
 *   up_t [R16][N]  = lora_up^T, block-diagonal                              bwd: U = dy up_t^T
 *   dn_p [K][Rp]   = scale * lora_down^T                                    bwd K-extension
 * R = groups*r, R16 = roundup(R,16), Rp = roundup(R,32) unless `rp` overrides it; group g owns output columns
 * [g*N/groups, (g+1)*N/groups).  down[g] is [r][K], up[g] is [N/groups][r] (lora.py:65-66).
 * `sites` is a DEVICE array of descriptors; one launch covers all of them. */
typedef struct leco_lora_site {
    const void* down[3]; /* bf16 */
    const void* up[3];   /* bf16 */
    int32_t groups, r, k, n;
    float scale;         /* multiplier * alpha / r  (lora.py:87,105) */
    int32_t taps;        /* 0/1: Linear or 1x1 conv; 9: 3x3 conv LoRA (k = 9*Cin, lora_down is [r][Cin][3][3]):
                            Rp = 64, dn_s[j][tap*Cin+c], dn_p[c][tap][j] = scale*down[j][c][8-tap] (dgrad operand) */
    void* dn_s;
    void* up_p;
    void* up_t;
    void* dn_p;
    void* up_pg;         /* optional: a second copy of up_p with the LECO_ACT_GEGLU row interleave (or NULL) */
    int32_t rp;          /* 0: Rp = roundup(R, 32) (64 for conv sites); else the column count of up_p / dn_p (a multiple
                            of 64 for R > 64: ranks whose stacked columns need several 64-wide K-extension steps) */
} leco_lora_site;

int leco_lora_pack(const leco_lora_site* sites, int32_t nsites, leco_stream_t stream);


/* G[j*g_sj + c*g_sc] += scale * sum_m P[m][p_off+j] * Q[m][q_off+c], j<r, c<cols (LoRA weight
 * gradients into the flat gradient slab).  P, Q bf16.  Replaces autograd's wgrad of lora_down / lora_up
 * (train_lora.py:279).  part == NULL: the 128-row slabs of M accumulate with fp32 atomics (fast, summation order
 * varies run to run).  part != NULL (DETERMINISTIC mode): every slab writes its contribution to the caller's fp32
 * scratch (ceil(m/128) * r * cols floats <= part_bytes) and a second launch adds them in slab order: bitwise
 * reproducible gradients. */
int leco_lora_wgrad(const void* p, int64_t ldp, const void* q, int64_t ldq, float* g,
                    int64_t g_sj, int64_t g_sc, int32_t m, int32_t r, int32_t cols, float scale,
                    float* part, int64_t part_bytes, leco_stream_t stream);

/* All LoRA weight gradients of a backward in ONE launch (atomic accumulation, like leco_lora_wgrad with part == NULL):
 * `problems` is a DEVICE array; problem i owns the blocks [block_start, block_start + blocks_x * ceil(m / 128)) with
 * blocks_x = ceil(cols / 256); total_blocks = their sum; max_rank = largest r in the table (<= 16).  a_mode
 * LECO_A_PLAIN: a leco_lora_wgrad problem; LECO_A_CONV3_S1/S2/UP2: a leco_lora_wgrad_conv problem (tap kh, kw). */
typedef struct leco_wgrad_problem {
    const void* p; int64_t ldp;
    const void* q; int64_t ldq;
    float* g; int64_t g_sj, g_sc;
    int32_t m, r, cols;
    float scale;
    int32_t a_mode, h_out, w_out, h_in, w_in, kh, kw;
    int32_t block_start, blocks_x;
} leco_wgrad_problem;
int leco_lora_wgrad_grouped(const leco_wgrad_problem* problems, int32_t nproblems, int32_t total_blocks,
                            int32_t max_rank, leco_stream_t stream);

/* same as leco_gemm with an explicit tile choice: 0 heuristic, 1 = 128x128 (wave shape by grid size), 2 = 128x160,
 * 3 = 64x64, 4 = 256x128, 5 = 128x128 as 4-wave workgroups (two per CU), 6 = 128x128 as one 8-wave workgroup per CU,
 * 11 = 128x64 (7..10: the patch-staged 3x3 convolution kernels, conv_patch.hip)
 * (tests / the launch-shape tuner leco_amd/tune.py). */
int leco_gemm_tile(const leco_gemm_args* args, int tile, leco_stream_t stream);
/* full form: additionally split_k (0 = heuristic, 1 = none, n = that many K slices) with a
 * caller-owned fp32 workspace for the partial slabs (split_k * m * n * 4 bytes; a too small
 * workspace only reduces the split).  Deep-K / small-M problems (the 8x8 and 16x16 UNet levels)
 * need this to fill 256 CUs. */
int leco_gemm_ex(const leco_gemm_args* args, int tile, int split_k, void* workspace,
                 int64_t workspace_bytes, leco_stream_t stream);
/* dry run of leco_gemm_ex: writes the kernel instantiation(s) that call would launch -- "gemm_kernel<BM, BN, CONV, NS,
 * NWM, TF> grid=.. split=.." as rocprofv3 names them, " ; "-separated when a call expands to two GEMMs -- into out.
 * Nothing is launched.  Measurement tooling (bench.py attributes plan launches to profile rows with it). */
int leco_gemm_describe(const leco_gemm_args* args, int tile, int split_k, void* workspace,
                       int64_t workspace_bytes, char* out, int32_t out_len);

/* same with Q gathered like the A operand of a 3x3 conv (one call per tap; a_mode LECO_A_CONV3_S1/S2/UP2):
 * the wgrad of a conv lora_down (c3lier, lora.py:72-81).  m = batch*h_out*w_out output rows. */
int leco_lora_wgrad_conv(const void* p, int64_t ldp, const void* q, int64_t ldq, float* g, int64_t g_sj,
                         int64_t g_sc, int32_t m, int32_t r, int32_t cols, float scale, int32_t a_mode,
                         int32_t h_out, int32_t w_out, int32_t h_in, int32_t w_in, int32_t kh, int32_t kw,
                         float* part, int64_t part_bytes, leco_stream_t stream);
/* out[b][c] = sum of the rows_per_group rows of sample b of x[.][c] (fp32): gradient of the per-sample
 * time-embedding bias added by ResnetBlock2D (needed only when time_emb_proj carries a LoRA). */
int leco_rowgroup_sum(const void* x, int64_t ldx, float* out, int64_t ldo, int32_t groups,
                      int32_t rows_per_group, int32_t cols, leco_stream_t stream);

/* ------------------------------------------------------------------------
 * Normalisation (diffusers GroupNorm(32) in ResnetBlock2D / Transformer2DModel / conv_norm_out,
 * LayerNorm in BasicTransformerBlock; call site train_util.py:156-160; dgrads train_lora.py:279).
 * x may be the channel concat [x0 | x1] (x1 == NULL: single source).  stats / bstats are
 * fp32 [batch][groups][2] scratch ({sum, sumsq} resp. {sum dxhat, sum dxhat*xhat}); the
 * forward stats must be kept for the backward.  act: LECO_ACT_NONE / LECO_ACT_SILU.
 * ---------------------------------------------------------------------- */
#define LECO_GN_STATS_FLOATS(batch, groups) ((int64_t)(batch) * (groups) * 2 * 257)
int leco_groupnorm_fwd(const void* x0, int64_t ld0, const void* x1, int64_t ld1, int32_t c0,
                       const float* gamma, const float* beta, int32_t batch, int32_t hw, int32_t c,
                       int32_t groups, float eps, int32_t act, float* stats, void* y, int64_t ldy,
                       leco_stream_t stream);
int leco_groupnorm_bwd(const void* x0, int64_t ld0, const void* x1, int64_t ld1, int32_t c0,
                       const void* dy, int64_t lddy, const float* gamma, const float* beta,
                       const float* stats, int32_t batch, int32_t hw, int32_t c, int32_t groups,
                       float eps, int32_t act, float* bstats, void* dx, int64_t lddx,
                       leco_stream_t stream);
/* 1 if leco_groupnorm_fwd handles this shape in one launch, 0 if it takes three (statistics, finish, apply) */
int leco_groupnorm_single_launch(int32_t batch, int32_t hw, int32_t c, int32_t groups);
/* GroupNorm forward from per-(sample, channel) {sum, sumsq} statistics that the producers of x0 / x1 left behind
 * (leco_gemm_args.col_stats, or leco_colstats for tensors whose producer has no such epilogue): ONE pass over x, no
 * reduction over pixels.  cstats0: fp32 [batch][(c0 or c) / atom][2] for x0's channels, cstats1: [batch][(c - c0) / atom][2]
 * for x1's; atom divides c0 and c / groups.  Also writes stats[b][g] = {sum, sumsq} per group, which leco_groupnorm_bwd
 * expects. */
int leco_groupnorm_apply_stats(const void* x0, int64_t ld0, const void* x1, int64_t ld1, int32_t c0, const float* cstats0,
                               const float* cstats1, int32_t atom, const float* gamma, const float* beta, int32_t batch,
                               int32_t hw, int32_t c, int32_t groups, float eps, int32_t act, float* stats, void* y,
                               int64_t ldy, leco_stream_t stream);
/* col_stats[b][ch / atom] += {sum, sumsq} over the hw rows of sample b of a bf16 [batch * hw][c] tensor (row stride ld) */
int leco_colstats(const void* x, int64_t ld, float* col_stats, int32_t atom, int32_t batch, int32_t hw, int32_t c,
                  leco_stream_t stream);
int leco_layernorm_fwd(const void* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                       int32_t m, int32_t c, void* y, int64_t ldy, float* mean, float* rstd,
                       leco_stream_t stream);
/* dx = LayerNorm dgrad (+ dres if not NULL: the residual-branch gradient) */
int leco_layernorm_bwd(const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* gamma,
                       const float* mean, const float* rstd, const void* dres, int64_t ldres,
                       int32_t m, int32_t c, void* dx, int64_t lddx, leco_stream_t stream);

/* ------------------------------------------------------------------------
 * Attention core softmax(Q K^T * scale) V per (batch, head); replaces diffusers' Attention
 * processor / xformers memory_efficient_attention (train_lora.py:68).  q/k/v/o are bf16 with
 * token stride ld* and batch stride bs* (elements); head h occupies columns [h*d, (h+1)*d).
 * lse: fp32 [batch][heads][sq] (log-sum-exp of the scaled scores), needed by the backward.
 * head_dim in {32, 40, 64, 80, 160}.
 * ---------------------------------------------------------------------- */
int leco_attention_fwd(const void* q, int64_t ldq, int64_t bsq, const void* k, int64_t ldk,
                       int64_t bsk, const void* v, int64_t ldv, int64_t bsv, void* o, int64_t ldo,
                       int64_t bso, float* lse, int32_t batch, int32_t heads, int32_t sq,
                       int32_t skv, int32_t head_dim, float scale, leco_stream_t stream);
/* dQ, dK, dV from dO (flash backward, recomputing P from lse).  delta: fp32 [batch][heads][sq]
 * scratch.  No atomics: dQ and dK/dV are produced by two deterministic kernels. */
int leco_attention_bwd(const void* q, int64_t ldq, int64_t bsq, const void* k, int64_t ldk,
                       int64_t bsk, const void* v, int64_t ldv, int64_t bsv, const void* o,
                       int64_t ldo, int64_t bso, const void* d_o, int64_t lddo, int64_t bsdo,
                       const float* lse, float* delta, void* dq, int64_t lddq,
                       int64_t bsdq, void* dk, int64_t lddk, int64_t bsdk, void* dv, int64_t lddv,
                       int64_t bsdv, int32_t batch, int32_t heads, int32_t sq, int32_t skv,
                       int32_t head_dim, float scale, leco_stream_t stream);

/* ------------------------------------------------------------------------
 * Elementwise / small kernels.
 * ---------------------------------------------------------------------- */
/* GEGLU (diffusers FeedForward ff.net.0): y[m][f] = u[m][f] * gelu_erf(u[m][F+f]) */
int leco_geglu_fwd(const void* u, int64_t ldu, void* y, int64_t ldy, int32_t m, int32_t f,
                   leco_stream_t stream);
int leco_geglu_bwd(const void* u, int64_t ldu, const void* dy, int64_t lddy, void* du, int64_t lddu,
                   int32_t m, int32_t f, leco_stream_t stream);
/* out = a + b (+ c) on row-strided bf16 matrices (gradient fan-in at residual/skip joins) */
int leco_add(const void* a, int64_t lda, const void* b, int64_t ldb, const void* c, int64_t ldc,
             void* out, int64_t ldo, int32_t m, int32_t cols, leco_stream_t stream);
/* dgrad of F.interpolate(scale=2, nearest) (Upsample2D): dx = 2x2 block sums of dy */
int leco_upsample2x_bwd(const void* dy, void* dx, int32_t batch, int32_t h, int32_t w, int32_t c,
                        leco_stream_t stream);
/* UNet conv_in: NCHW bf16 (batch,cin,h,w) -> channels-last bf16; w fp32 [cin][3][3][cout] */
int leco_conv_in(const void* x, const float* w, const float* bias, void* y, int32_t batch, int32_t h,
                 int32_t wd, int32_t cin, int32_t cout, leco_stream_t stream);
/* UNet conv_out: channels-last bf16 -> NCHW fp32 (batch,4,h,w); w bf16 [4][3][3][c]; and its dgrad */
int leco_conv_out(const void* x, const void* w, const float* bias, float* y, int32_t batch, int32_t h,
                  int32_t wd, int32_t c, int32_t cout, leco_stream_t stream);
int leco_conv_out_bwd(const float* dy, const void* w, void* dx, int32_t batch, int32_t h, int32_t wd,
                      int32_t c, int32_t cout, leco_stream_t stream);
/* diffusers Timesteps(flip_sin_to_cos=True, freq_shift=0): out[i] = [cos | sin](t_i * f), bf16
 * [n][dim]; t_i = t_table[*idx + i*t_stride] (idx may be NULL => 0): the timestep is read on
 * the device so a captured graph can be replayed for every denoising step. */
int leco_timestep_embedding(const float* t_table, const int32_t* idx, int32_t t_stride, int32_t n,
                            int32_t dim, void* out, leco_stream_t stream);
int leco_advance(int32_t* counter, leco_stream_t stream);
/* predict_noise's guidance combine (train_util.py:163-166) fused with DDIMScheduler.step
 * (train_util.py:190): eps = u + g (c - u); x <- coef[step][0] x + coef[step][1] eps; also writes
 * the next UNet input cat([x]*2) as bf16 (train_util.py:151). */
int leco_cfg_ddim_step(const float* pred, float* x, void* x2, const float* coef, const int32_t* step,
                       float guidance, int64_t half_n, leco_stream_t stream);

/* Same role for the other schedulers the reference accepts (ddpm / lms / euler_a, model_util.py:247-274), all linear
 * in (sample, model output, fresh noise, previous derivatives).  coef: fp32 [steps][LECO_SCHED_ROW] rows
 * {c_x, c_e, c_n, c_h1, c_h2, c_h3, s_in, d_x, d_e, -, -, -}:
 *   x' = c_x x + c_e out + c_n noise + sum_j c_hj h_j        (out = CFG-combined raw prediction, or 0 if pred == NULL)
 *   d  = d_x x + d_e out, history h1 <- d, h2 <- h1, h3 <- h2  (hist: fp32 [n_hist][half_n], n_hist = 0..3)
 *   x2 = bf16(s_in x') twice = next UNet input (scale_model_input of the NEXT step, train_util.py:153).
 * noise (fp32 [half_n]) may be NULL for deterministic schedulers. */
#define LECO_SCHED_ROW 12
int leco_cfg_sched_step(const float* pred, float* x, void* x2, const float* coef, const int32_t* step,
                        float guidance, int64_t half_n, const float* noise, float* hist, int32_t n_hist,
                        leco_stream_t stream);
/* PromptEmbedsPair.loss (prompt_util.py:107-148) with MSELoss(mean) on device in fp32, plus
 * d loss / d (raw target-pass UNet output) (train_lora.py:279 first autograd step). */
int leco_esd_loss(const float* tgt, const float* pos, const float* neu, const float* unc, float g_pred,
                  float g_loss, float sign, int64_t half_n, float* loss, float* dpred,
                  leco_stream_t stream);
/* The same objective on CONDITIONAL-ONLY predictions (fp32 [half_n] each): at guidance_scale = 1 -- what the reference
 * passes for the three frozen passes and the target pass, train_lora.py:202-256 -- predict_noise's u + 1 (c - u)
 * (train_util.py:163-166) is c, so FusedStep's de-duplicated step never evaluates the unconditional halves.  pos_c / neu_c /
 * unc_c may alias (identical prompts are evaluated once).  dpred_c (may be NULL): d loss / d tgt_c. */
int leco_esd_loss_cond(const float* tgt_c, const float* pos_c, const float* neu_c, const float* unc_c, float g_loss,
                       float sign, int64_t half_n, float* loss, float* dpred_c, leco_stream_t stream);
/* torch.optim.AdamW step on the flat fp32 LoRA slab + refresh of its bf16 shadow
 * (train_lora.py:280).  hyper (device): {lr, 1-beta1^t, 1-beta2^t, grad_scale}. */
int leco_adamw(float* p, const float* g, float* m, float* v, void* shadow, const float* hyper,
               float beta1, float beta2, float eps, float wd, int64_t n, leco_stream_t stream);

/* Lion (`lion_pytorch.Lion`, selected by train.optimizer = "lion", train_util.py:362-365): p *= 1 - lr wd;
 * p -= lr sign(beta1 m + (1-beta1) g); m = beta2 m + (1-beta2) g.  Same hyper block ({lr, -, -, grad_scale}) and
 * bf16 shadow refresh as leco_adamw. */
int leco_lion(float* p, const float* g, float* m, void* shadow, const float* hyper, float beta1, float beta2,
              float wd, int64_t n, leco_stream_t stream);
int leco_cast_f32_bf16(const float* x, void* y, int64_t n, leco_stream_t stream);
int leco_memset(void* p, int32_t value, int64_t bytes, leco_stream_t stream);
/* dst = `reps` copies of the `bytes` (multiple of 16) at src, back to back: batch broadcast of a tensor that is identical for
 * several samples (the two halves of predict_noise's cat([latents] * 2), train_util.py:151, before the first use of the
 * prompt embeddings) */
int leco_repeat(const void* src, void* dst, int64_t bytes, int32_t reps, leco_stream_t stream);
/* Step glue of the reference loop body as two launches (the tensors the reference moves with framework ops between its
 * UNet calls, train_lora.py:175-199):
 *   leco_step_begin: x2 = cat([scale x] * 2) in the activation dtype (bf16, or fp32 when x2_is_f32) -- the first UNet input
 *     of `diffusion` (train_util.py:151-153,172-193; scale = scale_model_input of the first step, 1 for DDIM) -- and
 *     *t_idx = 0 (the denoising pass counter; may be NULL).
 *   leco_step_mid: after the k denoising passes the last UNet input IS cat([denoised] * 2); it becomes the input of the
 *     LoRA-on target pass (dst_a, may be NULL) and, `reps_b` times back to back, of the batched LoRA-off passes (dst_b)
 *     (train_lora.py:202-256); `current_timestep` (train_lora.py:195-199) is stored into both plans' timestep tables
 *     (t_slot_*: the slot's address) and both step indices are pointed at that slot.  bytes: multiple of 16. */
int leco_step_begin(const float* x, void* x2, int32_t x2_is_f32, float scale, int64_t half_n, int32_t* t_idx,
                    leco_stream_t stream);
int leco_step_mid(const void* src, void* dst_a, void* dst_b, int64_t bytes, int32_t reps_b, float t_cur,
                  float* t_slot_a, float* t_slot_b, int32_t* t_idx_a, int32_t* t_idx_b, int32_t slot,
                  leco_stream_t stream);

/* ------------------------------------------------------------------------
 * Row-stripe fused transformer-block kernels (csrc/stripe.hip): forward-only, bf16, C = 320 (the 64^2 level of SD1.x /
 * SD2.x: head dim 40 or 64).  One workgroup owns 64 token rows for a whole chain of Linear / LayerNorm / cross-attention /
 * GEGLU operations of diffusers' BasicTransformerBlock + Transformer2DModel (call site train_util.py:156-160; LoRA term
 * lora.py:102-106): the residual stream stays in registers (fp32), the activation operand of every GEMM in LDS, the block's
 * weights stream once through registers in MFMA operand layout.  Replaces 14 of the per-op launches above per transformer block in the LoRA-on denoising passes
 * (train_util.py:172-193) and the batched LoRA-off predictions (train_lora.py:202-237); the differentiated pass keeps the
 * per-op kernels (it must save every intermediate).
 * ---------------------------------------------------------------------- */
typedef struct leco_xlin {     /* one Linear of a stripe chain: y = x W^T + bias (+ (x down^T) (scale up)^T) */
    const void* w;             /* bf16 [N][K], K contiguous */
    int64_t ldw;
    const float* bias;         /* fp32 [N] or NULL */
    const void* dn;            /* stacked lora_down rows, bf16 [t_rows][K] (leco_lora_site.dn_s); NULL = LoRA off */
    int64_t ld_dn;
    const void* up;            /* scale * lora_up, bf16 [N][32] (leco_lora_site.up_p; columns >= groups*r zero) */
    int64_t ld_up;
    int32_t t_rows;            /* 16 or 32: stacked rank rounded up to 16 */
    int32_t packed;            /* 1: w is stored in MFMA fragment order instead of row-major (frozen weights, re-laid once):
                                  element (n, k) at ((n / 16 * (K / 32) + k / 32) * 64 + (k % 32 / 8) * 16 + n % 16) * 8 + k % 8,
                                  i.e. the 16 rows x 32 columns a wave loads per k-step are ONE contiguous 1 KB block (8 whole
                                  cache lines per load instead of 16 half lines); needs ldw == K */
} leco_xlin;

/* A-stationary GEMM for the short-K / small-M Linears of the transformer blocks (csrc/xgemm.hip; forward-only plans):
 *   c = a lin.w^T (+ (a lin.dn^T)(lin.up)^T) + lin.bias (+ residual)
 * -- attn{1,2}.to_q / to_out.0, the fused q|k|v projection, Transformer2DModel.proj_in / proj_out below the 64^2 level
 * (diffusers' BasicTransformerBlock / Transformer2DModel via train_util.py:156-160; LoRA term lora.py:102-106).  The
 * workgroup's BM x K activation tile is staged in LDS once, lin.w streams global -> VGPR in MFMA fragment order
 * (lin.packed must be 1, lin.ldw == k), the LoRA down-projection rides along in the same K sweep.
 * Shapes: n % 128 == 0, k in {320, 640, 1280} (leco_xgemm_supported); BM = leco_xgemm_rows(k) rows per workgroup. */
typedef struct leco_xgemm_args {
    int32_t m, n, k;
    const void* a;             /* bf16 [m][k], row stride lda (multiple of 8) */
    int64_t lda;
    leco_xlin lin;
    const void* residual;      /* bf16 [m][n] or NULL, row stride ldr (multiple of 4) */
    int64_t ldr;
    void* c;                   /* bf16 [m][n], row stride ldc (multiple of 4) */
    int64_t ldc;
} leco_xgemm_args;
int leco_xgemm_supported(int32_t m, int32_t n, int32_t k);
int leco_xgemm_rows(int32_t k);
int leco_xgemm(const leco_xgemm_args* a, leco_stream_t stream);

/* 1 if the stripe kernels cover this shape (else the caller keeps the per-op launches) */
int leco_xblock_supported(int32_t c, int32_t heads, int32_t skv, int32_t rows_per_sample);

/* K / V of the cross-attention for the stripe kernels, once per step: kv = output of the fused to_k|to_v projection,
 * bf16 [batch * skv][2 C] (row stride ld_kv).  kp: bf16 [batch][heads][80][64] (zero padded); vt: bf16
 * [batch][heads][roundup(head_dim, 16)][96], V^T with the key order permuted to the MFMA operand order (+ a row of ones
 * when head_dim % 16 != 0: the softmax row sum then comes out of the PV product). */
#define LECO_XATTN_KP_ELEMS(batch, heads) ((int64_t)(batch) * (heads) * 80 * 64)
#define LECO_XATTN_VT_ELEMS(batch, heads, head_dim) ((int64_t)(batch) * (heads) * (((head_dim) + 15) / 16 * 16) * 96)
int leco_xattn_prep(const void* kv, int64_t ld_kv, void* kp, void* vt, int32_t batch, int32_t heads, int32_t skv,
                    int32_t head_dim, leco_stream_t stream);

/* Everything of a BasicTransformerBlock after its self-attention core, + Transformer2DModel.proj_out:
 *   h1 = to_out1(attn) + h_in;  q2 = to_q2(LN2(h1));  a2 = softmax(q2 K^T scale) V;  h2 = to_out2(a2) + h1;
 *   h3 = ff2(GEGLU(ff1(LN3(h2)))) + h2;  out = proj_out.w ? proj_out(h3) + res : h3.
 * ff1 carries the LECO_ACT_GEGLU row interleave (weights, bias and up: blocks of 64 value rows followed by their 64 gate
 * rows).  col_stats (optional): {sum, sumsq} of the stored bf16 values per sample and atom of stats_atom (even) columns,
 * like leco_gemm_args.col_stats. */
typedef struct leco_xblock_tail_args {
    int32_t m, c, heads, skv, rows_per_sample;
    int32_t src_rows;                    /* 0: attn / h_in / res have m rows.  Else (a multiple of 64 dividing m): they have
                                            src_rows rows and row r reads row r % src_rows -- samples that are identical up to
                                            the first use of the prompt (classifier-free-guidance duplicates) are computed once */
    const void* attn; int64_t ld_attn;   /* self-attention output, bf16 [m][c] */
    const void* h_in; int64_t ld_h;      /* residual stream entering the block, bf16 [m][c] */
    leco_xlin to_out1, to_q2, to_out2, ff1, ff2, proj_out;
    const float* ln2_g; const float* ln2_b; const float* ln3_g; const float* ln3_b;
    float ln_eps;
    const void* kp; const void* vt;      /* leco_xattn_prep */
    float attn_scale;                    /* head_dim^-0.5 */
    const void* res; int64_t ld_res;     /* proj_out residual (the Transformer2DModel input), bf16 [m][c] */
    void* out; int64_t ld_out;           /* bf16 [m][c] */
    float* col_stats; int32_t stats_atom;
} leco_xblock_tail_args;
int leco_xblock_tail(const leco_xblock_tail_args* args, leco_stream_t stream);

/* Everything of a Transformer2DModel BEFORE the self-attention core of its first BasicTransformerBlock:
 *   n = GroupNorm(x) from the statistics the producer of x left (gn_cstats: fp32 [batch][c / stats_atom][2] {sum, sumsq},
 *       leco_gemm_args.col_stats; NULL: x is already normalised);  h_out = proj_in(n);  qkv_out = to_q|to_k|to_v(LN1(h_out)).
 * qkv is the fused [3 c][c] projection (its stacked lora_down rows are shared by the three 320-column sweeps). */
typedef struct leco_xblock_head_args {
    int32_t m, c, rows_per_sample;
    const void* x; int64_t ld_x;                 /* bf16 [m][c] */
    const float* gn_cstats; int32_t stats_atom; int32_t groups;
    const float* gn_g; const float* gn_b; float gn_eps;
    leco_xlin proj_in, qkv;
    const float* ln1_g; const float* ln1_b; float ln_eps;
    void* h_out; int64_t ld_hout;                /* bf16 [m][c]: the residual stream entering the block */
    void* qkv_out; int64_t ld_qkv;               /* bf16 [m][3 c] */
} leco_xblock_head_args;
int leco_xblock_head(const leco_xblock_head_args* args, leco_stream_t stream);

/* ------------------------------------------------------------------------
 * hipGraph capture of a whole UNet pass (the reference issues ~10^4 eager kernel launches
 * per pass from Python; here a pass is one graph launch).
 * ---------------------------------------------------------------------- */
typedef void* leco_graph_t;
int leco_graph_begin_capture(leco_stream_t stream);
int leco_graph_end_capture(leco_stream_t stream, leco_graph_t* out);
int leco_graph_launch(leco_graph_t graph, leco_stream_t stream);
int leco_graph_destroy(leco_graph_t graph);

/* ---- fp32 compute mode (`train.precision: float32`, config_util.py:75-83; csrc/f32.hip) --------------------------------
 * One twin per entry point above that touches activations: SAME argument list and meaning, but every tensor the bf16
 * entry point takes as bf16 (activations, packed weights, LoRA operand images) is fp32 here, strides still in elements.
 * Contractions run on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32); a whole SD1.5 UNet pass agrees with the fp32
 * reference to 3.7e-6.  Written for exactness, not speed. */
int leco_f32_gemm(const leco_gemm_args* args, leco_stream_t stream);
int leco_f32_attention_fwd(const void* q, int64_t ldq, int64_t bsq, const void* k, int64_t ldk,
                       int64_t bsk, const void* v, int64_t ldv, int64_t bsv, void* o, int64_t ldo,
                       int64_t bso, float* lse, int32_t batch, int32_t heads, int32_t sq,
                       int32_t skv, int32_t head_dim, float scale, leco_stream_t stream);
int leco_f32_attention_bwd(const void* q, int64_t ldq, int64_t bsq, const void* k, int64_t ldk,
                       int64_t bsk, const void* v, int64_t ldv, int64_t bsv, const void* o,
                       int64_t ldo, int64_t bso, const void* d_o, int64_t lddo, int64_t bsdo,
                       const float* lse, float* delta, void* dq, int64_t lddq,
                       int64_t bsdq, void* dk, int64_t lddk, int64_t bsdk, void* dv, int64_t lddv,
                       int64_t bsdv, int32_t batch, int32_t heads, int32_t sq, int32_t skv,
                       int32_t head_dim, float scale, leco_stream_t stream);
int leco_f32_groupnorm_fwd(const void* x0, int64_t ld0, const void* x1, int64_t ld1, int32_t c0,
                       const float* gamma, const float* beta, int32_t batch, int32_t hw, int32_t c,
                       int32_t groups, float eps, int32_t act, float* stats, void* y, int64_t ldy,
                       leco_stream_t stream);
int leco_f32_groupnorm_bwd(const void* x0, int64_t ld0, const void* x1, int64_t ld1, int32_t c0,
                       const void* dy, int64_t lddy, const float* gamma, const float* beta,
                       const float* stats, int32_t batch, int32_t hw, int32_t c, int32_t groups,
                       float eps, int32_t act, float* bstats, void* dx, int64_t lddx,
                       leco_stream_t stream);
int leco_f32_layernorm_fwd(const void* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                       int32_t m, int32_t c, void* y, int64_t ldy, float* mean, float* rstd,
                       leco_stream_t stream);
int leco_f32_layernorm_bwd(const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* gamma,
                       const float* mean, const float* rstd, const void* dres, int64_t ldres,
                       int32_t m, int32_t c, void* dx, int64_t lddx, leco_stream_t stream);
int leco_f32_geglu_fwd(const void* u, int64_t ldu, void* y, int64_t ldy, int32_t m, int32_t f,
                   leco_stream_t stream);
int leco_f32_geglu_bwd(const void* u, int64_t ldu, const void* dy, int64_t lddy, void* du, int64_t lddu,
                   int32_t m, int32_t f, leco_stream_t stream);
int leco_f32_add(const void* a, int64_t lda, const void* b, int64_t ldb, const void* c, int64_t ldc,
             void* out, int64_t ldo, int32_t m, int32_t cols, leco_stream_t stream);
int leco_f32_upsample2x_bwd(const void* dy, void* dx, int32_t batch, int32_t h, int32_t w, int32_t c,
                        leco_stream_t stream);
int leco_f32_conv_in(const void* x, const float* w, const float* bias, void* y, int32_t batch, int32_t h,
                 int32_t wd, int32_t cin, int32_t cout, leco_stream_t stream);
int leco_f32_conv_out(const void* x, const void* w, const float* bias, float* y, int32_t batch, int32_t h,
                  int32_t wd, int32_t c, int32_t cout, leco_stream_t stream);
int leco_f32_conv_out_bwd(const float* dy, const void* w, void* dx, int32_t batch, int32_t h, int32_t wd,
                      int32_t c, int32_t cout, leco_stream_t stream);
int leco_f32_timestep_embedding(const float* t_table, const int32_t* idx, int32_t t_stride, int32_t n,
                            int32_t dim, void* out, leco_stream_t stream);
int leco_f32_cfg_ddim_step(const float* pred, float* x, void* x2, const float* coef, const int32_t* step,
                       float guidance, int64_t half_n, leco_stream_t stream);
int leco_f32_cfg_sched_step(const float* pred, float* x, void* x2, const float* coef, const int32_t* step,
                        float guidance, int64_t half_n, const float* noise, float* hist, int32_t n_hist,
                        leco_stream_t stream);
int leco_f32_cast_f32_bf16(const float* x, void* y, int64_t n, leco_stream_t stream);
int leco_f32_rowgroup_sum(const void* x, int64_t ldx, float* out, int64_t ldo, int32_t groups,
                      int32_t rows_per_group, int32_t cols, leco_stream_t stream);
int leco_f32_lora_pack(const leco_lora_site* sites, int32_t nsites, leco_stream_t stream);
int leco_f32_lora_wgrad(const void* p, int64_t ldp, const void* q, int64_t ldq, float* g,
                    int64_t g_sj, int64_t g_sc, int32_t m, int32_t r, int32_t cols, float scale,
                    float* part, int64_t part_bytes, leco_stream_t stream);
int leco_f32_lora_wgrad_conv(const void* p, int64_t ldp, const void* q, int64_t ldq, float* g, int64_t g_sj,
                         int64_t g_sc, int32_t m, int32_t r, int32_t cols, float scale, int32_t a_mode,
                         int32_t h_out, int32_t w_out, int32_t h_in, int32_t w_in, int32_t kh, int32_t kw,
                         float* part, int64_t part_bytes, leco_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LECO_HIP_H */
