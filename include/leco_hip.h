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
enum { LECO_ACT_NONE = 0, LECO_ACT_SILU = 1 };

typedef struct leco_gemm_args {
    const void* a0;       /* bf16 */
    const void* a1;       /* bf16, second channel/K source (skip-connection concat) or NULL */
    int64_t lda0, lda1;   /* plain: row strides (elements); conv: per-pixel channel strides */
    int32_t k_split;      /* plain: columns [0,k_split) come from a0; conv: channels [0,k_split) */
    int32_t a_mode;
    int32_t batch, h_out, w_out, h_in, w_in; /* conv modes only */
    const void* w;        /* bf16 [N][K], K contiguous; conv: [N][3][3][Cin] */
    int64_t ldw;
    int32_t m, n, k;      /* k % 64 == 0 (conv: Cin % 64 == 0), n % 4 == 0 */
    const void* a_ext;    /* bf16 [M][ext_k] (row stride ld_aext) or NULL */
    int64_t ld_aext;
    const void* w_ext;    /* bf16 [N][ext_k] (row stride ld_wext) */
    int64_t ld_wext;
    int32_t ext_k;        /* 0, 32 or 64 */
    const float* bias;        /* fp32 [N] or NULL */
    const float* rowbias;     /* fp32 [M/rows_per_group][N] or NULL (time-embedding add) */
    int32_t rows_per_group;
    const void* residual;     /* bf16 [M][N] (row stride ldr) or NULL */
    int64_t ldr;
    int32_t act;
    void* c;                  /* bf16 out, row stride ldc (or NULL) */
    int64_t ldc;
    float* c_f32;             /* optional fp32 copy of the pre-rounding result, row stride ldc32 */
    int64_t ldc32;
} leco_gemm_args;

int leco_gemm(const leco_gemm_args* args, leco_stream_t stream);

/* Re-pack every LoRA site (1..3 LoRA modules sharing an input: fused q|k|v) from the bf16
 * shadow of the trainable slab into the four MFMA operand images the hot path consumes:
 *   dn_s [R16][K]  = stacked lora_down rows (zero rows up to R16)          fwd: T = x dn_s^T
 *   up_p [N][Rp]   = scale * lora_up, block-diagonal over groups            fwd K-extension
 *   up_t [R16][N]  = lora_up^T, block-diagonal                              bwd: U = dy up_t^T
 *   dn_p [K][Rp]   = scale * lora_down^T                                    bwd K-extension
 * R = groups*r, R16 = roundup(R,16), Rp = roundup(R,32); group g owns output columns
 * [g*N/groups, (g+1)*N/groups).  down[g] is [r][K], up[g] is [N/groups][r] (lora.py:65-66).
 * `sites` is a DEVICE array of descriptors; one launch covers all of them. */
typedef struct leco_lora_site {
    const void* down[3]; /* bf16 */
    const void* up[3];   /* bf16 */
    int32_t groups, r, k, n;
    float scale;         /* multiplier * alpha / r  (lora.py:87,105) */
    int32_t _pad;
    void* dn_s;
    void* up_p;
    void* up_t;
    void* dn_p;
} leco_lora_site;

int leco_lora_pack(const leco_lora_site* sites, int32_t nsites, leco_stream_t stream);

/* G[j*g_sj + c*g_sc] += scale * sum_m P[m][p_off+j] * Q[m][q_off+c], j<r, c<cols (LoRA weight
 * gradients; fp32 atomics into the flat gradient slab).  P, Q bf16.  Replaces autograd's
 * wgrad of lora_down / lora_up (train_lora.py:279). */
int leco_lora_wgrad(const void* p, int64_t ldp, const void* q, int64_t ldq, float* g,
                    int64_t g_sj, int64_t g_sc, int32_t m, int32_t r, int32_t cols, float scale,
                    leco_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LECO_HIP_H */
