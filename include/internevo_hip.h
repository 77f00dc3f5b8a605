/*
 * libinternevo_hip.so -- C ABI of the MI355X (gfx950) kernels that stand behind the native-op import
 * sites of InternEvo's training step (boundary #2).
 *
 * Conventions
 *   - plain C: raw device pointers + sizes, no torch / C++ types in any signature;
 *   - every entry point is stream-ordered: `stream` is a hipStream_t passed as void* (NULL = default
 *     stream); nothing here calls hipDeviceSynchronize, allocates, or keeps a pointer past the call;
 *   - tensors are row-major; bf16 is the raw 16-bit pattern; `ld*` are row strides in ELEMENTS;
 *   - return value: IE_OK (0) or a negative IE_ERR_* code; ie_last_error() gives the reason
 *     (thread-local, so the autograd thread and the main thread do not clobber each other);
 *   - re-entrant from several host threads (the reference runs backward on the autograd thread).
 *
 * Each function names the reference interface it replaces (paths relative to the InternEvo tree).
 */
#ifndef INTERNEVO_HIP_H
#define INTERNEVO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IE_OK 0
#define IE_ERR_INVALID (-1)     /* bad argument (null pointer, negative size, misaligned) */
#define IE_ERR_UNSUPPORTED (-2) /* shape/dtype this build has no kernel for */
#define IE_ERR_LAUNCH (-3)      /* HIP reported a launch error */

#define IE_BF16 0
#define IE_F32 1

#define IE_ABI_VERSION 1

int ie_abi_version(void);
const char* ie_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * K5  RMSNorm.  Replaces apex.normalization.fused_layer_norm.MixedFusedRMSNorm
 *     (internlm/model/utils.py:662-675) == internlm/model/ops/norm.py:10-23 manual_rms_norm.
 *     y = w * cast_w(x * rsqrt(mean(x^2) + eps)); statistics in fp32; x bf16 or fp32; w bf16 or fp32;
 *     y has w's dtype.  rstd[rows] (fp32) is saved for backward.
 * ---------------------------------------------------------------------------------------------- */
int ie_rmsnorm_fwd(const void* x, int x_dtype, const void* w, int w_dtype, void* y, float* rstd,
                   int64_t rows, int64_t cols, float eps, void* stream);

/* Fused residual add + RMSNorm (modeling_internlm2.py:696-702, 721-727: `_dropped + _residual` then
 * norm): r = bf16(a + b) is written to r_out (may alias a or b), y = RMSNorm(r).  bf16 only. */
int ie_add_rmsnorm_fwd(const void* a, const void* b, void* r_out, const void* w, void* y, float* rstd,
                       int64_t rows, int64_t cols, float eps, void* stream);

/* Backward.  dx = rstd * (dy*w - xhat * mean(dy*w*xhat)); if dres != NULL, dx += dres (the gradient
 * that reaches the same tensor through the residual connection).  dw (w's dtype) = [accumulate ? dw : 0]
 * + sum_rows dy * xhat, reduced deterministically through dw_partial, an fp32 workspace of
 * ie_rmsnorm_bwd_partials(rows) * cols elements. */
int64_t ie_rmsnorm_bwd_partials(int64_t rows);
int ie_rmsnorm_bwd(const void* dy, const void* x, int x_dtype, const void* w, int w_dtype,
                   const float* rstd, const void* dres, void* dx, float* dw_partial, void* dw,
                   int accumulate, int64_t rows, int64_t cols, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K2  Rotary.  ie_apply_rotary replaces rotary_emb.apply_rotary(x1, x2, cos, sin, out1, out2, conj)
 *     (internlm/model/modules/embedding.py:115-120,142-153; torch restatement :63-86).
 *     x1/x2/out1/out2 are [batch, seq, heads, half] views given by element strides (out may alias
 *     in); cos/sin are [seq, half] with row stride cs_ld, broadcast over batch and heads.
 *     conj=0: o1 = x1*cos - x2*sin, o2 = x1*sin + x2*cos;  conj=1: o1 = x1*cos + x2*sin,
 *     o2 = -x1*sin + x2*cos.  Math in fp32, result rounded to dtype.
 * ---------------------------------------------------------------------------------------------- */
int ie_apply_rotary(const void* x1, const void* x2, const void* cos_, const void* sin_, void* out1,
                    void* out2, int dtype, int64_t batch, int64_t seq, int64_t heads, int64_t half,
                    int64_t xs_b, int64_t xs_s, int64_t xs_h, int64_t os_b, int64_t os_s, int64_t os_h,
                    int64_t cs_ld, int conj, void* stream);

/* Fused InternLM2 q/k/v split + even/odd de-interleave + position gather + rotation
 * (modeling_internlm2.py:416-434 + embedding.py:367-371).  qkv [T, hkv, gs=q_per_kv+2, d] bf16;
 * q_out [T, hkv*q_per_kv, d]; kv_out [T, 2, hkv, d]; cos/sin [max_pos, d/2] bf16; pos[T] int64
 * ("indexes").  interleaved=1 is the reference's `not rot_embed_HF_impl` branch.  d must be 128 or 64. */
int ie_qkv_rotary_fwd(const void* qkv, const void* cos_, const void* sin_, const int64_t* pos,
                      void* q_out, void* kv_out, int64_t T, int hkv, int q_per_kv, int d,
                      int interleaved, void* stream);
/* Backward of the above: (dq, dkv) -> dqkv in the wqkv output layout (conjugate rotation,
 * re-interleave). */
int ie_qkv_rotary_bwd(const void* dq, const void* dkv, const void* cos_, const void* sin_,
                      const int64_t* pos, void* dqkv, int64_t T, int hkv, int q_per_kv, int d,
                      int interleaved, void* stream);
/* The same pair with the attention's softmax scale riding on q (MI355X-first fusion; no reference counterpart -- the reference hands
 * softmax_scale to flash_attn, modeling_internlm2.py:446-468): q_out = bf16(q_scale * rotated q), the product taken in fp32 before the
 * one rounding, so that the attention kernels read scores in log2 units straight off the MFMA accumulators (q_scale = softmax_scale *
 * log2 e, attention called with softmax_scale = ln 2: ie_flash_attn_fwd then picks the folded-softmax kernel); the backward multiplies
 * the incoming dq (= dL/dq_out) by dq_scale (the chain rule's q_scale) before the conjugate rotation.  q_scale = dq_scale = 1 is exactly the unscaled pair. */
int ie_qkv_rotary_fwd_scaled(const void* qkv, const void* cos_, const void* sin_, const int64_t* pos,
                             void* q_out, void* kv_out, int64_t T, int hkv, int q_per_kv, int d,
                             int interleaved, float q_scale, void* stream);
int ie_qkv_rotary_bwd_scaled(const void* dq, const void* dkv, const void* cos_, const void* sin_,
                             const int64_t* pos, void* dqkv, int64_t T, int hkv, int q_per_kv, int d,
                             int interleaved, float dq_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K8  SwiGLU gate.  Replaces torch.jit.script(Silu) (internlm/model/utils.py:684-688;
 *     modules/mlp.py:85).  out = bf16(bf16(silu(a)) * b).  a, b, out: [rows, cols] bf16 with row
 *     strides lda/ldb/ldo.
 * ---------------------------------------------------------------------------------------------- */
int ie_swiglu_fwd(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo,
                  int64_t rows, int64_t cols, void* stream);
/* da = dout*b*silu'(a), db = dout*silu(a); act_out (optional, may be NULL) = recomputed forward. */
int ie_swiglu_bwd(const void* dout, int64_t lddo, const void* a, int64_t lda, const void* b,
                  int64_t ldb, void* da, int64_t ldda, void* db, int64_t lddb, void* act_out,
                  int64_t ldact, int64_t rows, int64_t cols, void* stream);

/* The two FFN products with the SwiGLU arithmetic in their epilogues (FeedForward.forward, internlm/model/modules/mlp.py:82-86, and its
 * autograd backward): ONE launch each when the shape rides on the 256x256 refill-schedule GEMM (K % 64 == 0, F % 128 == 0 forward), otherwise
 * the product followed by the kernel above -- bit-identical results either way.
 *   fwd: h13[M, 2F] = x[M, K] @ w13[2F, K]^T (rows 0..F-1 = w1, F..2F-1 = w3);  act[M, F] = bf16(bf16(silu(h13[:, :F])) * h13[:, F:])
 *   bwd: dh13[M, 2F] = (d gate | d up) of the gate at h13 for d(act) = dy[M, K] @ w2[K, F]; d(act) itself is written only on the two-launch path
 *        (dact_scratch [M, F], always required).
 * ie_tune_ffn_fuse(mode): bit 0 = fuse the forward product (default on), bit 1 = fuse the input-gradient product on every eligible shape (default off: on the
 * plain launch measured no faster, profiles/r03_ffn_fuse_ab.jsonl), bit 2 = fuse it where the persistent GEMM frame takes the product (default on since round 6:
 * -0.7 % of the training step, profiles/r06_step_ffn_fuse_bwd_abab.log); 0 forces the two-launch path everywhere.  Default 5. */
int ie_gemm_swiglu_fwd(const void* x, int64_t ldx, const void* w13, int64_t ldw, void* h13, int64_t ldh, void* act, int64_t ld_act,
                       int64_t M, int64_t F, int64_t K, void* stream);
int ie_gemm_swiglu_bwd(const void* dy, int64_t ldy, const void* w2, int64_t ldw, const void* h13, int64_t ldh, void* dh13, int64_t ldd,
                       void* dact_scratch, int64_t ld_scratch, int64_t M, int64_t F, int64_t K, void* stream);
int ie_tune_ffn_fuse(int mode);
/* a3 + a4 in one launch (round 6): q [T, hkv q_per_kv, d], kv [T, 2, hkv, d] = split + de-interleave + rotary (ie_qkv_rotary_fwd_scaled's arithmetic) of
 * x[M, K] @ wqkv[hkv (q_per_kv + 2) d, K]^T -- MHA._packed_forward's wqkv linear, GQA un-interleave and rotary embedding (modeling_internlm2.py:404-445,
 * modules/embedding.py:89-166).  With d = 128 and a product the persistent GEMM frame takes (whole 256-row / 256-column tiles, more than 256 of them) the
 * arithmetic sits in the product's epilogue and the [M, N] product never reaches memory; otherwise the product is written to qkv_scratch [M, N] (contiguous,
 * always required) and ie_qkv_rotary_fwd_scaled follows.  Bit-identical either way.  ie_tune_qkv_rotary_fuse(0) forces the two launches; ie_gemm_qkv_rotary_is_fused
 * says which path a shape takes. */
int ie_gemm_qkv_rotary_fwd(const void* x, int64_t ldx, const void* wqkv, int64_t ldw, const void* cos_table, const void* sin_table, const int64_t* pos, void* q_out,
                           void* kv_out, void* qkv_scratch, int64_t ld_scratch, int64_t M, int hkv, int q_per_kv, int d, int64_t K, int interleaved, float q_scale,
                           void* stream);
int ie_tune_qkv_rotary_fuse(int mode);
int ie_gemm_qkv_rotary_is_fused(int64_t M, int hkv, int q_per_kv, int d, int64_t K);
int ie_gemm_swiglu_is_fused(int bwd, int64_t M, int64_t F, int64_t K);   /* 1 = one launch for this shape */

/* ------------------------------------------------------------------------------------------------
 * K4  Softmax cross-entropy.  Replaces flash_attn.losses.cross_entropy.CrossEntropyLoss
 *     (internlm/model/losses/ce_loss.py:26-36) / nn.CrossEntropyLoss (:37-40).
 *     logits [rows, vocab] bf16 or fp32 (row stride ld); labels int64, ignore_index rows give 0.
 *     fwd: loss_rows[rows] fp32 (per-token loss), lse[rows] fp32.
 *     ie_ce_mean: loss_out[0] = sum(loss_rows)/n_valid, count_out[0] = n_valid (as float).
 *     bwd (in place when dlogits == logits, as inplace_backward=True):
 *       dlogits = (softmax - (1-ls)*onehot - ls/vocab) * gscale, gscale = *dloss * dloss_mul / *count.
 *     dloss and count are DEVICE scalars so the loss scale never round-trips through the host.
 *     count == NULL selects per-row mode: dloss[rows] holds one upstream gradient per token
 *     (gscale = dloss[row] * dloss_mul), which is what an autograd consumer of the per-token losses needs.
 * ---------------------------------------------------------------------------------------------- */
int ie_ce_fwd(const void* logits, int dtype, int64_t ld, const int64_t* labels, float* loss_rows,
              float* lse, int64_t rows, int64_t vocab, int64_t ignore_index, float label_smoothing,
              void* stream);
int ie_ce_mean(const float* loss_rows, const int64_t* labels, int64_t rows, int64_t ignore_index,
               float* loss_out, float* count_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Metric pass fused into K4.  Replaces the per-micro-batch work of AccPerplex.update and
 * LossWithTypeId.update (internlm/model/metrics.py:108-199,281-310), which re-read the [T, vocab] logits three
 * more times (max, argmax, exp-sum) plus one more CE forward, and torch_scatter.scatter(..., reduce="sum") (:92-96).
 *   ie_ce_fwd_metric = ie_ce_fwd that also writes argmax_rows[rows] (int32, FIRST index of the row maximum) and
 *     nll_rows[rows] (plain lse - logit[label]; 0 for ignored rows; independent of label_smoothing).
 *   ie_metric_accumulate adds one micro-batch to device-resident accumulators (single block, fixed order):
 *     facc[5] = {right, total, total_log_probs, loss, token_num} (fp32, the reference's accumulator dtypes),
 *     ds_right / ds_tokens int64[ntypes], ds_loss / ds_token_num fp32[ntypes], indexed by type_ids[rows] (int64).
 *     right counts label == argmax over ALL rows; total / total_log_probs / loss / token_num over labels != ignore.
 *     ntypes == 0: the per-type pointers may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int ie_ce_fwd_metric(const void* logits, int dtype, int64_t ld, const int64_t* labels, float* loss_rows,
                     float* lse, int32_t* argmax_rows, float* nll_rows, int64_t rows, int64_t vocab,
                     int64_t ignore_index, float label_smoothing, void* stream);
int ie_metric_accumulate(const float* nll_rows, const int32_t* argmax_rows, const int64_t* labels,
                         const int64_t* type_ids, int64_t rows, int64_t ignore_index, int ntypes,
                         float* facc, int64_t* ds_right, int64_t* ds_tokens, float* ds_loss,
                         float* ds_token_num, void* stream);
int ie_ce_bwd(const void* logits, void* dlogits, int dtype, int64_t ld, const int64_t* labels,
              const float* lse, const float* dloss, float dloss_mul, const float* count, int64_t rows,
              int64_t vocab, int64_t ignore_index, float label_smoothing, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K6  L2 norm.  Replaces amp_C.multi_tensor_l2norm via multi_tensor_applier
 *     (internlm/solver/optimizer/utils.py:30-37,191-204; torch restatement :177-188).
 *     ie_sumsq_partial: partial[part_offset + i] = sum of squares of a slice of x (fp32 accumulate of
 *     fp32-cast elements); returns through *nparts_out how many partials it wrote.
 *     ie_sumsq_finish:  out[0] = [accumulate ? out[0] : 0] + sum(partial[0..nparts)), fixed order.
 * ---------------------------------------------------------------------------------------------- */
int64_t ie_sumsq_max_partials(void);
int ie_sumsq_partial(const void* x, int dtype, int64_t n, float* partial, int64_t part_offset,
                     int64_t* nparts_out, void* stream);
int ie_sumsq_finish(const float* partial, int64_t nparts, float* out, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a15/a17  Step control on device: DynamicGradScaler.update + overflow check + unscale/clip factor
 *     (internlm/solver/optimizer/hybrid_zero_optim.py:695-779,863-876; optimizer/utils.py:431-543).
 *     One tiny kernel; no host sync.  state layout: see IeStepState.
 * ---------------------------------------------------------------------------------------------- */
typedef struct IeStepState {
    float loss_scale;       /* current dynamic loss scale                                   */
    int growth_step;        /* DynamicGradScaler._growth_step                               */
    int hysteresis_step;    /* DynamicGradScaler._hysteresis_step                           */
    int adam_step;          /* number of successful optimizer steps (bias correction)       */
    int skip;               /* 1 = this step is skipped (inf/nan)                            */
    int found_inf;          /* grad-norm sentinel -1 of compute_norm                         */
    int found_nan;          /* grad-norm sentinel -2                                        */
    float inv_scale;        /* 1/(loss_scale*max(1,clip)) applied to grads by ie_adamw_step  */
    float grad_norm;        /* unscaled global grad norm of this step (for logging)          */
    float loss_scale_used;  /* the loss scale the grads of this step were produced with      */
    int skipped_total;
    int _pad;
} IeStepState;

typedef struct IeScalerConfig {
    float growth_factor, backoff_factor, min_scale, max_scale;
    int growth_interval, hysteresis;
    float clip_grad_norm; /* <= 0: no clipping */
    int dynamic;          /* 0 for fp32 models: scaler frozen, no unscale/clip (hybrid_zero_optim.py:712,773) */
} IeScalerConfig;

int ie_step_state_init(IeStepState* state_dev, float initial_scale, void* stream);
/* sumsq_dev: device scalar = squared L2 norm of the (scaled) grads, already reduced across ranks. */
int ie_step_control(IeStepState* state_dev, const float* sumsq_dev, const IeScalerConfig* cfg_host,
                    void* stream);

/* ------------------------------------------------------------------------------------------------
 * K7  AdamW on a flat partition.  Replaces torch.optim.AdamW(fused=True) == torch._fused_adamw_
 *     (internlm/train/pipeline.py:305-315, stepped at hybrid_zero_optim.py:787) fused with the
 *     fp16->fp32 grad cast, unscale/clip (:749-779) and the fp32->bf16 param copy (:791-797).
 *     g: grads (bf16 or fp32), p32/m/v fp32, p16 (optional) bf16 shadow.  Reads skip, inv_scale,
 *     adam_step from state_dev.
 * ---------------------------------------------------------------------------------------------- */
int ie_adamw_step(const void* g, int g_dtype, float* p32, float* m, float* v, void* p16, int64_t n,
                  const IeStepState* state_dev, double lr, double beta1, double beta2, double eps,
                  double weight_decay, void* stream);
/* How many CUs the following ie_adamw_step / ie_adamw_step_group launches may occupy: 0 (default) = the whole chip (16 384 grid-stride workgroups), n = 1 .. 256 =
 * n workgroups of 1024 threads, each alone on a CU.  Same arithmetic element by element, bit-identical results.  For an update that runs BESIDE the next step's
 * forward (hybrid_zero_optim.py:787 steps after the backward; the reference has nothing beside it): the matrix kernels need whole CUs and only get them between two
 * whole-chip launches.  Read at launch time on the calling thread. */
int ie_tune_adamw_cus(int cus);

/* ------------------------------------------------------------------------------------------------
 * a10 Embedding.  F.embedding fwd (internlm/model/modules/embedding.py:52-60) and its dense
 *     backward (scatter-add), deterministic, fp32 accumulate.  ws: int32[vocab + 1 + T] workspace.
 * ---------------------------------------------------------------------------------------------- */
int ie_embedding_fwd(const void* weight, const int64_t* ids, void* out, int64_t T, int64_t vocab,
                     int64_t dim, void* stream);
int ie_embedding_bwd(const void* dout, const int64_t* ids, void* dweight, int* ws, int64_t T,
                     int64_t vocab, int64_t dim, int accumulate, void* stream);

/* elementwise helpers used around the path */
int ie_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);

/* a18  Send-side layout of the Ulysses sequence<->head exchange.  Replaces the tensor_split + .contiguous() + torch.cat of
 *      _SeqAllToAll (internlm/model/modules/multi_head_attention.py:27-53) around dist.all_to_all: bf16 [A][B][S][C] ->
 *      [S][A][B][C] (inverse = 0) or back (inverse = 1); A = local tokens (or S*... see INTEGRATION.md), B = 1 for q / ctx,
 *      2 for kv, S = sequence-parallel size, C = heads_per_rank * head_dim (multiple of 8).  The receive side of
 *      all_to_all_single needs no copy: [S][A][B][C] is the gathered [S*A tokens][B][C] tensor.
 *      ie_scale_bf16: x *= factor in place (the embed/head gradient re-scale of the ISP averaging rule). */
int ie_seq_head_permute(const void* in, void* out, int64_t A, int B, int S, int64_t C, int inverse, void* stream);
int ie_scale_bf16(void* x, int64_t n, float factor, void* stream);

/* ScaleColumnParallelLinearWithNormHead.forward (internlm/model/ops/linear.py:124-153) and the embedding's gradient scale
 * (modeling_internlm2.py:970-973; embed_grad_scale = weight_scale = s):
 *   ie_grad_scale_mix:  x <- s x + (1 - s) x.detach() in place, the value as the reference's bf16 expression rounds it (embedding output);
 *   ie_head_weight_fwd: out[rows, cols] = the weight the head multiplies by = F.normalize(s w + (1 - s) w.detach()) (norm_head: rows scaled
 *                       to unit length, inv_norm[rows] saved; scale = 1 skips the mix);
 *   ie_head_weight_bwd: dw (= or +=) s * (dy - y (y . dy)) * inv_norm per row, dy = gradient w.r.t. `out` (a weight-gradient GEMM's result),
 *                       y = `out`.  Rows are independent: a vocabulary-parallel head calls these on its own rows. */
int ie_grad_scale_mix(void* x, int64_t n, float scale, void* stream);
int ie_head_weight_fwd(const void* w, int64_t w_ld, void* out, int64_t out_ld, float* inv_norm, int64_t rows, int64_t cols, float scale,
                       int norm_head, void* stream);
int ie_head_weight_bwd(const void* dy, int64_t dy_ld, const void* y, int64_t y_ld, const float* inv_norm, void* dw, int64_t dw_ld, int64_t rows,
                       int64_t cols, float scale, int norm_head, int accumulate, void* stream);
int ie_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K3/K3b  bf16 GEMM on MFMA, fp32 accumulate, bf16 output.
 *     C[M,N] = opA(A)[M,K] * opB(B)[K,N]   (+ C if accumulate)
 *       a_kmajor = 0: A stored [M][K] (lda >= K);  1: A stored [K][M] (lda >= M)
 *       b_kmajor = 0: B stored [N][K] (ldb >= K);  1: B stored [K][N] (ldb >= N)
 *     F.linear fwd  y = x W^T          (model/utils.py:275)          : a_kmajor=0, b_kmajor=0
 *     dgrad         dx = dy W          (model/utils.py:315)          : a_kmajor=0, b_kmajor=1
 *     fused_dense_lib.linear_bias_wgrad  dW = dy^T x (utils.py:293)  : a_kmajor=1, b_kmajor=1
 *     accumulate=1 reproduces autograd's bf16 `param.grad += grad`: C = bf16(C + bf16(acc)).
 * ---------------------------------------------------------------------------------------------- */
int ie_gemm_bf16(const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor,
                 void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate, void* stream);
/* The same forward product on fp8 operands (OCP e4m3; csrc/fp8.hip, schedule -6 of csrc/gemm_bf16_dma.hip on v_mfma_f32_32x32x64_f8f6f4).  The reference has
 * no fp8 linear: per-tensor dynamic scaling, tolerance defined by tests/test_fp8_gpu.py.
 *   ie_fp8_amax:     amax[z] = max(amax[z], max |x_z|) over the n bf16 values of tensor z of `count` tensors laid out back to back (the caller zeroes amax;
 *                    device scalars; count > 1 needs n % 8 == 0).
 *   ie_fp8_quantize: q_z[i] = e4m3(x_z[i] * 448 / amax[z]), round-to-nearest-even, saturating; dequant[z] = amax[z] / 448 (1 when amax[z] == 0).
 *   ie_gemm_fp8:     C[M,N] bf16 = (A[M,K] e4m3)(B[N,K] e4m3)^T * *dequant_a * *dequant_b, fp32 accumulation (+ C if accumulate);
 *                    K % 128 == 0, lda / ldb (elements = bytes) multiples of 16, operands < 4 GiB: IE_ERR_UNSUPPORTED otherwise.
 *   ie_gemm_fp8_batched: `count` such products in ONE launch (the experts of a MoE layer), product z at A + z stride_a, B + z stride_b, C + z stride_c
 *                    (elements) with the scales dequant_a[z], dequant_b[z]. */
int ie_fp8_amax(const void* x, int64_t n, int64_t count, float* amax, void* stream);
int ie_fp8_quantize(const void* x, int64_t n, int64_t count, const float* amax, void* q, float* dequant, void* stream);
int ie_gemm_fp8(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                const float* dequant_a, const float* dequant_b, int accumulate, void* stream);
int ie_gemm_fp8_batched(const void* A, int64_t lda, int64_t stride_a, const void* B, int64_t ldb, int64_t stride_b, void* C, int64_t ldc, int64_t stride_c,
                        int64_t count, int64_t M, int64_t N, int64_t K, const float* dequant_a, const float* dequant_b, int accumulate, void* stream);
/* Same product with an explicit block-tile shape (tuning / A-B benchmarking; ie_gemm_bf16 picks one itself):
 * variant 0 = 128x128 (4 waves), 1 = 256x256 (8 waves), 2 = 256x128, 3 = 128x256 (register-staged);
 * 4 = 256x256 LDS-DMA, 5 = 128x128 LDS-DMA, 6 / 7 = 256x256 LDS-DMA with the DMA issue spread over 2 / 4 k-steps,
 * 8 = 128x128 LDS-DMA spread over 2 k-steps, 9 / 10 = 256x256 / 128x128 LDS-DMA with role-split load/compute
 * phases, 11 = 256x256 with one wave per SIMD (4 waves, 128x128 each), buffer-addressed LDS-DMA and fragments
 * pipelined across k-tiles (operands must each span < 4 GiB), 12 = 128x256 phased, 13 = 256x256 phased with two k-steps per phase and buffer-addressed DMA (operands < 4 GiB), 14 = 128x256 likewise,
 * 15 / 16 / 17 = 256x256 on four 32-deep LDS stages (k32 ring, counted vmcnt): 15 phased 8 waves, 16 one wave per SIMD, 17 = 16 with
 * the DMA pieces of an entry split over both k-steps (operands < 4 GiB);
 * 18 = 11 with all DMA pieces issued two k-steps early, 19 = 256x256, one wave per SIMD, the whole k-tile's fragments in registers and the two
 * 64-deep stages refilled operand by operand behind counted waits (operands < 4 GiB);
 * all LDS-DMA variants need K % 64 == 0; -1 = automatic (which can also cut a half-empty last round of 256x256 tiles off into a second launch
 * of 128x256 tiles, see ie_tune_gemm_tail_split). */
int ie_gemm_bf16_tile(int variant, const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb,
                      int b_kmajor, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate,
                      void* stream);
/* `batch` equal products in ONE launch, operands `stride_*` ELEMENTS apart (multiples of 8): the E experts of a GShard MoE layer --
 * expert e's [C, M] rows of the dispatch buffer x its [2F, M] weights (internlm/model/moe/experts.py: one module call per expert) --
 * fill the chip together instead of one half-empty round each. */
int ie_gemm_bf16_batched(const void* A, int64_t lda, int64_t stride_a, int a_kmajor, const void* B, int64_t ldb,
                         int64_t stride_b, int b_kmajor, void* C, int64_t ldc, int64_t stride_c, int64_t M, int64_t N,
                         int64_t K, int batch, int accumulate, void* stream);
/* column sums of a [rows, cols] bf16 matrix (bias gradient of linear_bias_wgrad, has_bias=True) */
int ie_colsum_bf16(const void* x, int64_t ld, void* out, int64_t rows, int64_t cols, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K1  Flash attention, varlen, causal or full, GQA, head dim 128 (or 64), bf16.
 *     Replaces flash_attn.flash_attn_varlen_kvpacked_func (modeling_internlm2.py:446-468) and
 *     FlashSelfAttention/FlashCrossAttention (multi_head_attention.py:381-392).
 *     q [T, hq, d] (token stride q_ts elements, head stride d), k/v [T, hkv, d] given as separate
 *     base pointers with token stride kv_ts (kv-packed [T,2,hkv,d]: k = kv, v = kv + hkv*d,
 *     kv_ts = 2*hkv*d; qkv-packed likewise).  cu_seqlens int32[nseq+1] (device); seqlen_q ==
 *     seqlen_k per sequence.  out [T, hq, d] (token stride o_ts), lse [hq, T] fp32.
 *     dropout is not supported (the path runs with attn_drop_rate = 0).
 * ---------------------------------------------------------------------------------------------- */
int ie_flash_attn_fwd(const void* q, int64_t q_ts, const void* k, const void* v, int64_t kv_ts,
                      void* out, int64_t o_ts, float* lse, const int32_t* cu_seqlens, int nseq,
                      int64_t T, int max_seqlen, int hq, int hkv, int d, float softmax_scale,
                      int causal, void* stream);
/* delta: fp32 workspace of ie_flash_attn_bwd_workspace(T, hq, hkv, d) floats (delta[hq, T], then -lse / scale [hq, T] and -delta [hq, T],
 * the start values of the dK/dV kernel's S / dP accumulators, then the deterministic per-head-split partial dK / dV sums of the
 * causal-balanced dK/dV kernel).  dq [T,hq,d] (stride dq_ts), dk/dv [T,hkv,d]
 * (stride dkv_ts). */
int64_t ie_flash_attn_bwd_workspace(int64_t T, int hq, int hkv, int d);
int ie_flash_attn_bwd(const void* dout, int64_t do_ts, const void* q, int64_t q_ts, const void* k,
                      const void* v, int64_t kv_ts, const void* out, int64_t o_ts, const float* lse,
                      float* delta, void* dq, int64_t dq_ts, void* dk, void* dv, int64_t dkv_ts,
                      const int32_t* cu_seqlens, int nseq, int64_t T, int max_seqlen, int hq, int hkv,
                      int d, float softmax_scale, int causal, void* stream);
/* The backward of MHA._packed_forward's attention block in one piece (round 6; internlm/model/modeling_internlm2.py:416-468 read upwards): the backward of
 * flash_attn_varlen_kvpacked_func, of the rotary embedding on q and k (ApplyRotaryEmb.backward, modules/embedding.py:150-166) and of the GQA rearrange -- the two
 * attention kernels write dQ, dK (rotated back) and dV straight into dqkv [T][hkv][hq / hkv + 2][d], the output gradient of the wqkv product; no [T, hq, d] /
 * [T, 2, hkv, d] intermediates, no rotary launch.  cos / sin [positions][d / 2] bf16 (the forward's tables), positions int64[T].  Bit-identical to
 * ie_flash_attn_bwd + ie_qkv_rotary_bwd (interleaved = 0, dq_scale = 1).  Only where ..._is_fused says so (head dim 128, causal, no head split of the dK / dV
 * kernel, the default backward variant); elsewhere IE_ERR_UNSUPPORTED and the caller runs the two calls. */
int ie_flash_attn_bwd_qkv_rotary_is_fused(int nseq, int max_seqlen, int hq, int hkv, int d, int causal);
int ie_flash_attn_bwd_qkv_rotary(const void* dout, int64_t do_ts, const void* q, int64_t q_ts, const void* k, const void* v, int64_t kv_ts,
                                 const void* out, int64_t o_ts, const float* lse, float* delta_ws, void* dqkv, const void* cos, const void* sin,
                                 const int64_t* positions, const int32_t* cu_seqlens, int nseq, int64_t T, int max_seqlen, int hq, int hkv, int d,
                                 float softmax_scale, int causal, void* stream);

/* y[rows, cols] += bias[cols] in place (bf16): the attention biases of the InternLM-1 block (multi_head_attention.py:371-408). */
int ie_bias_add_bf16(void* y, int64_t ld, const void* bias, int64_t rows, int64_t cols, void* stream);
/* HybridZeroOptimizer._step with several parameter groups (MoE models: default / fp32 / moe, train/utils.py:25-80): one overflow check and one
 * scaler update over all groups, but every group unscaled and clipped by its OWN norm (hybrid_zero_optim.py:760-779,863-876).
 * sumsq_dev [ngroups]; group_inv_scale_dev / group_norm_dev [ngroups] are outputs (norms already divided by the loss scale). */
int ie_step_control_groups(IeStepState* state_dev, const float* sumsq_dev, int ngroups, const IeScalerConfig* cfg_host,
                           float* group_inv_scale_dev, float* group_norm_dev, void* stream);
/* ie_adamw_step with the gradient factor of a parameter group (one float on the device) instead of the state's. */
int ie_adamw_step_group(const void* g, int g_dtype, float* p32, float* m, float* v, void* p16, int64_t n,
                        const IeStepState* state_dev, const float* inv_scale_group_dev, double lr, double beta1, double beta2,
                        double eps, double weight_decay, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K6  GShard mixture-of-experts layer, top-2 gating, in index form (csrc/moe.hip).
 *     Replaces the routing arithmetic of internlm/model/moe/gshard_layer.py: top2gating (:217-285), the `sec,sm->ecm`
 *     dispatch (:446-448) and `sec,ecm->sm` combine (:482-486) einsums and their autograd; the expert FeedForward
 *     modules (modules/mlp.py:82-86) run on ie_gemm_bf16 / ie_swiglu_* over the [E*C, M] buffers.
 *     S tokens, M hidden, E experts (2..16), C = capacity; x bf16 [S, M]; wg fp32 [E, M] (the gate is an fp32 module);
 *     expert [2, S] int32 (first / second choice); row [2, S] int32 = e*C + slot or -1 (dropped); weight [2, S] fp32
 *     (renormalised; consumers round it to bf16 like NaiveAMP does); token_of [E*C] int32 = 2*token + choice or -1.
 * ---------------------------------------------------------------------------------------------- */
/* Gumbel(0,1) noise of the second choice (gshard_layer.py:63-70,233): counter-based, reproducible from (seed, offset). */
int ie_moe_gumbel_noise(float* out, int64_t n, uint32_t seed, uint64_t offset, void* stream);
/* logits = float(x) wg^T, gates = softmax(logits), expert[0] = argmax(gates), expert[1] = argmax(logits + noise) without the first
 * (noise [S, E] fp32 or NULL). */
int ie_moe_gate_fwd(const void* x, int64_t x_ld, const float* wg, const float* noise, int64_t S, int M, int E,
                    float* logits, float* gates, int32_t* expert, void* stream);
/* slots in token order, capacity drop, renormalised weights, l_aux (1 float, rounded to bf16), exp_counts [E]. */
int ie_moe_route(const float* gates, const int32_t* expert, int64_t S, int E, int capacity, int32_t* row, float* weight,
                 int32_t* token_of, float* l_aux, int32_t* exp_counts, void* stream);
/* The expert buffers' rows in chunk-major order (the expert exchange in `nchunk` pieces, each piece's all_to_all overlapped with the expert products of the piece
 * before it; the reference's exchange is blocking: moe/gshard_layer.py:465-498): slot c of expert e moves from row e C + c to row k E Cn + e Cn + (c mod Cn),
 * Cn = capacity / nchunk, k = c / Cn.  row [2, S] is rewritten in place, token_of_in [E C] copied to token_of_out in the new order (must not alias). */
int ie_moe_chunk_rows(int32_t* row, const int32_t* token_of_in, int32_t* token_of_out, int64_t S, int E, int capacity, int nchunk, void* stream);
int ie_moe_dispatch(const void* x, int64_t x_ld, const int32_t* token_of, int64_t rows, int M, void* expert_in, void* stream);
int ie_moe_combine_fwd(const void* expert_out, const int32_t* row, const float* weight, int64_t S, int M, void* out,
                       int64_t out_ld, void* stream);
/* d_expert_out [rows, M] and d_weight [2, S] (bf16-rounded values, zero where nothing was dispatched). */
int ie_moe_combine_bwd(const void* dout, int64_t d_ld, const void* expert_out, const int32_t* token_of, const float* weight,
                       int64_t rows, int64_t S, int M, void* d_expert_out, float* d_weight, void* stream);
/* dx [S, M] = sum of the token's dispatched d_expert_in rows (overwrites dx). */
int ie_moe_dispatch_bwd(const void* d_expert_in, const int32_t* row, const int32_t* token_of, int64_t S, int M, void* dx,
                        int64_t dx_ld, void* stream);
/* gate backward: d_weight (+ d l_aux = aux_factor * *loss_scale_dev, loss_scale_dev may be NULL = 1) -> d_logits [S, E];
 * dx += bf16(d_logits wg); d_wg (= or +=) d_logits^T float(x).  workspace: ie_moe_dwg_workspace(M, E) floats. */
int64_t ie_moe_dwg_workspace(int M, int E);
int ie_moe_gate_bwd(const void* x, int64_t x_ld, const float* wg, const float* gates, const int32_t* expert, const int32_t* row,
                    const float* d_weight, const int32_t* exp_counts, const float* loss_scale_dev, float aux_factor, int64_t S,
                    int M, int E, float* d_logits, void* dx, int64_t dx_ld, float* d_wg, int accumulate_d_wg, float* workspace,
                    void* stream);

/* Diagnostic: the kernel instantiation the last GEMM launch of this process went to, spelled as rocprofv3 prints it ("gemm_dma_k<256, 256, 2, 2,
 * false, true, -5, 0>"); bench.py compares the set it sees with the kernels of the committed HBM-traffic measurement.  ie_gemm_note_kernel is the
 * library's own recorder (exported because two translation units share it). */
int ie_gemm_last_kernel(char* buf, int n);
int ie_gemm_note_kernel(int kind, int bm, int bn, int wm, int wn, int a_kmajor, int b_kmajor, int schedule, int epilogue);
/* Tuning hook: tile rows per group of the LDS-DMA GEMMs' XCD-aware tile order (0 = default 4). */
int ie_tune_gemm_group(int tile_rows_per_group);
/* Tuning hook: the automatic GEMM's tail split (0 = off [default: neutral inside the training step], 1 = on: remainder tiles by
 * variant 14 when an operand is k-major, else 12; 2 / 3 = always 14 / 12). */
int ie_tune_gemm_tail_split(int mode);
/* Tuning hook: 1 (default) = the forward / input-gradient products on the 16x16x32 refill schedule run in its PERSISTENT frame where that applies (whole
 * 256x256 tiles, an even number of 64-deep k-tiles, more tiles than blocks): 256 blocks walk their tiles, the transfers continue through the tile boundary, the
 * accumulators leave through a wave-private LDS turn.  n >= 8 (a multiple of 8) = the same on n blocks (tests).  Same results bit for bit; 0 = the plain launch. */
int ie_tune_gemm_persistent(int mode);
/* (internal: the block count behind ie_tune_gemm_persistent; exported because two translation units share it) */
int ie_gemm_dma_set_persistent_grid(int blocks);
int ie_gemm_dma_persistent_takes(int64_t M, int64_t N, int64_t K);   /* (internal, shared by two translation units: 1 when the persistent frame takes the product) */
/* Tuning hook (A/B benchmarking): occupancy the dQ kernel of ie_flash_attn_bwd is compiled for, 1 or 2 waves/SIMD. */
int ie_tune_flash_dq_occupancy(int waves_per_simd);
/* Tuning hook: how many blocks share the q heads of one kv head in the dK/dV kernel (0 = automatic, 1, 2 or 4). */
int ie_tune_flash_dkdv_split(int split);
/* Tuning hooks (A/B benchmarking, tools/kbench): kernel variant of the attention forward / backward (0 = default).  Backward: bit 0 = four waves per dK/dV block,
 * bit 1 = the five-product path (needs ie_flash_attn_bwd_set_spill), bit 2 = delta = sum_d dO * O by its own kernel in front of the dQ kernel (the default since
 * round 6 computes it in the dQ kernel's prologue: the same bits). */
int ie_tune_flash_fwd_variant(int variant);
int ie_tune_flash_bwd_variant(int variant);
/* Ring attention (the sequence-parallel attention whose K / V blocks travel around the ranks; internevo_amd/seqpar.py -- the reference has no
 * such mode, the result to match is DistributedAttention's, multi_head_attention.py:56-135): the block = full attention of a rectangle of scores
 * per sequence, queries cu_q[s] .. cu_q[s + 1] (of Tq rows) against keys cu_k[s] .. cu_k[s + 1] (of Tk rows); a sequence without keys gives out = 0,
 * lse = -inf.  The backward takes the lse / out the probabilities are normalised with (for one block: the MERGED lse / out of the whole row, which
 * makes its dq / dk / dv the block's additive share of the whole gradient); dq [Tq], dk / dv [Tk] are overwritten; Tq must be > 0; `delta` is a
 * workspace of ie_flash_attn_bwd_workspace(Tq, hq, hkv, d) floats.  ie_attn_merge folds a block's (out_p bf16 [n, hq, d], lse_p [hq, Tp]) into the running
 * fp32 result (acc [.., hq, d] contiguous, lse_acc [hq, Ta]) of its first n rows; ie_acc_bf16: dst (fp32) += src (bf16), n % 8 == 0. */
int ie_flash_attn_fwd_x(const void* q, int64_t q_ts, const void* k, const void* v, int64_t kv_ts, void* out, int64_t o_ts, float* lse,
                        const int32_t* cu_q, const int32_t* cu_k, int nseq, int64_t Tq, int64_t Tk, int max_seqlen_q, int hq, int hkv,
                        int d, float softmax_scale, void* stream);
int ie_flash_attn_bwd_x(const void* dout, int64_t do_ts, const void* q, int64_t q_ts, const void* k, const void* v, int64_t kv_ts,
                        const void* out, int64_t o_ts, const float* lse, float* delta, void* dq, int64_t dq_ts, void* dk, void* dv,
                        int64_t dkv_ts, const int32_t* cu_q, const int32_t* cu_k, int nseq, int64_t Tq, int64_t Tk, int max_seqlen_q,
                        int max_seqlen_k, int hq, int hkv, int d, float softmax_scale, void* stream);
int ie_attn_merge(float* acc, float* lse_acc, int64_t Ta, const void* out_p, int64_t p_ts, const float* lse_p, int64_t Tp, int64_t n, int hq,
                  int d, void* stream);
int ie_acc_bf16(float* dst, const void* src, int64_t n, void* stream);
/* The five-product attention backward (ie_tune_flash_bwd_variant bit 1; opt-in): the dK/dV kernel also writes dS^T (bf16, 2 bytes per visible
 * (query, key) pair) and dQ is formed from it instead of recomputing S and dP a second time.  It needs a caller-owned buffer of
 * ie_flash_attn_bwd_spill_bytes(nseq, max_seqlen, hq, causal) bytes, 1-KiB aligned, handed over (and taken back with NULL, 0) by
 * ie_flash_attn_bwd_set_spill; without one that is large enough ie_flash_attn_bwd runs its default seven-product path.  Same results to
 * bf16 rounding, deterministic; measured 2-4 % faster than the default at 4 x 4096 tokens and not the default: profiles/r05_flash_bwd_spill.md. */
int64_t ie_flash_attn_bwd_spill_bytes(int nseq, int max_seqlen, int hq, int causal);
int ie_flash_attn_bwd_set_spill(void* buf, int64_t bytes);

/* Diagnostic (bench.py --hold-cus): `blocks` idle workgroups that each occupy one CU for `usec` microseconds on `stream` -- what a
 * collective's kernels take away from the products beside them, reproduced on a one-GPU box (the reference overlaps its gradient reduction with backward the same
 * way: hybrid_zero_optim.py:290-367).  No memory traffic, no result. */
int ie_hold_cus(int blocks, int usec, void* stream);
/* Tuning hook (A/B): products with exactly this many output columns stay on the plain launch although ie_tune_gemm_persistent would take them (0 = none). */
int ie_tune_gemm_persistent_skip_n(int64_t n_cols);
/* Tuning hook (A/B): 1 = every input-gradient product (A k-contiguous, B k-major) takes the refill schedule (and with it the persistent frame where that applies),
 * not only the long / wide ones (K >= 6144 or N >= 8192); 0 (default) = the others on the 8-wave k32 ring. */
int ie_tune_gemm_dgrad_refill_all(int on);
/* out[M, N] = bf16(bf16(x[M, K] w[N, K]^T) + addend[M, N]) (round 6): the block's residual add -- `residual = dropout(hidden) + residual`,
 * internlm/model/modeling_internlm2.py:707-717,728-737 -- in the epilogue of the product in front of it (wo, w2), with the roundings of the two-step form (the
 * product rounded to bf16, the sum formed in fp32 and rounded): bit-identical to ie_gemm_bf16 + the add of ie_add_rmsnorm_fwd.  addend has out's leading dimension
 * and must not be out.  Only where the persistent GEMM frame takes the product (ie_gemm_dma_persistent_takes(M, N, K)); elsewhere IE_ERR_UNSUPPORTED. */
int ie_linear_fwd_add(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* addend, void* out, int64_t ld_out, int64_t M, int64_t N, int64_t K,
                      void* stream);
/* Tuning hook: 1 = every launch of the persistent frame zeroes its tile-queue slot with a 36-byte hipMemsetAsync in stream order first; 0 (default) = the slots are
 * zero at module load, every STREAM has a slot of its own (the first 63 streams; later ones share the last slot and take turns on it) and the LAST block of every
 * launch zeroes its slot again (the kernel does that either way), no memset kernels between the products (-0.5 % of the benchmark step,
 * profiles/r06_step_queue_memset_abab.log). */
int ie_tune_gemm_queue_memset(int on);
/* The weight-gradient products' tail k-split (round 6; internlm/model/utils.py:293-299,336-340 `linear_bias_wgrad`): with a caller-owned workspace registered here
 * (16-byte aligned; 32 MiB covers every remainder of <= 128 tiles; NULL, 0 takes it back) a weight-gradient product (both operands k-major, K % 128 == 0) whose
 * 256x256 tiling ends in a round that is at most half full computes that remainder as two half-k products in ONE launch + a fixed-order fp32 fix-up: the
 * remainder's 128 tiles occupy all 256 CUs for half a tile time instead of half of them for a whole one.  Deterministic; one extra bf16 rounding of the two partial
 * sums against the unsplit product.  The workspace is used in stream order by every such product: one stream at a time. */
int ie_gemm_set_wgrad_ksplit_workspace(void* ws, int64_t bytes);

/* Diagnostic: runs one v_mfma_f32_32x32x16_bf16 with A[i][k], B[k][j] taken from a[32*16], b[16*32]
 * (row-major, bf16) using the operand/accumulator lane maps the kernels assume, writes c[32*32] fp32.
 * Lets the test-suite pin the fragment layout on real hardware. */
int ie_mfma_probe(const void* a, const void* b, float* c, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INTERNEVO_HIP_H */
