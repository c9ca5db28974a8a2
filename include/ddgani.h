/*
 * ddgani.h — C-ABI of libddgani_hip.so, the MI355X (gfx950) implementation of DDGANI's
 * imputation hot path.
 *
 * The reference (xianghuang2/DDGANI) has NO native/FFI boundary: its hot path sits behind
 * plain Python functions that call stock ATen ops.  Each entry point below
 * therefore cites the reference *Python* function (file:line, relative to the reference
 * tree) whose ATen op sequence it replaces.  The Python mirror of those functions
 * (ddgani_amd/model/Diffusion.py, ddgani_amd/utils/util.py) binds them with ctypes.
 *
 * Conventions
 *  - every function returns DDG_OK (0) or a negative DDG_E_* code; nothing throws;
 *  - all data pointers are DEVICE pointers owned by the caller unless marked "host";
 *  - nothing allocates or frees: scratch memory is a caller-provided workspace whose size
 *    the matching ddg_*_workspace_bytes() function returns;
 *  - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = the
 *    default stream); no entry point synchronises;
 *  - matrices are row-major contiguous fp32 unless stated otherwise;
 *  - an empty table (rows == 0 / Nq == 0) returns DDG_OK without a launch; the elementwise, head and
 *    ddg_linear entry points then accept NULL for the row-data pointers (torch hands out NULL for
 *    empty tensors), all other arguments are still checked;
 *  - "field layout": a table column c occupies encoded columns [offsets[c], offsets[c+1])
 *    (width 1 = numeric, width > 1 = one-hot categorical); offsets is int32[C+1] on the
 *    DEVICE, D = offsets[C];  is_cat is uint8[C] on the DEVICE.
 */
#ifndef DDGANI_H
#define DDGANI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDG_OK 0
#define DDG_E_ARG (-1)        /* bad argument (null pointer, negative size, ...)            */
#define DDG_E_UNSUPPORTED (-2) /* shape outside what the kernels are built for               */
#define DDG_E_WORKSPACE (-3)  /* workspace too small                                        */
#define DDG_E_LAUNCH (-4)     /* hipGetLastError() != hipSuccess after a launch             */
#define DDG_E_COLLECTIVE (-6) /* the caller's all-reduce callback reported a failure            */
#define DDG_E_SCRATCH_LIMIT (-5) /* input larger than the static scratch of a *_static entry point */

#define DDG_MAX_D 256         /* widest encoded table the attention kernels are built for   */
#define DDG_MAX_TOPK_REG 30   /* absolute k up to this uses the in-register selection       */

/* library / error helpers */
int ddg_version(void);
const char *ddg_error_string(int code);

/* ---------------------------------------------------------------------------------------
 * a3  Diffusion_setting.initialize          model/Diffusion.py:86-108
 * Host-side: fills out8[8*T] (host memory, fp32) with betas (beta[0]=0), alphas, sqrt_alphas,
 * 1/sqrt_alphas, sqrt_betas, alpha_cumulative, sqrt_alpha_cumulative,
 * sqrt_one_minus_alpha_cumulative, in that order, each of length T.
 * ------------------------------------------------------------------------------------- */
int ddg_schedule(int T, float *out8_host);

/* ---------------------------------------------------------------------------------------
 * a5  forward_diffusion / get_xt            model/Diffusion.py:391-403
 *   x_t = sqrt(abar_t) * x0 + sqrt(1-abar_t) * eps
 * eps: explicit noise [rows,D] or NULL -> generated on chip (Philox4x32-10 + Box-Muller,
 * keyed by seed/stream_id) and, when eps_out != NULL, also written there.
 * t: int64[rows] per-row timesteps (1..T) or NULL -> every row uses t_uniform.
 * sched: DEVICE copy of ddg_schedule's out8 (8*T floats).
 * ------------------------------------------------------------------------------------- */
int ddg_forward_noise(const float *x0, const float *eps, const int64_t *t, int t_uniform,
                      const float *sched, int T, uint64_t seed, uint64_t stream_id,
                      float *x_t, float *eps_out, int64_t rows, int D, void *stream);

/* ---------------------------------------------------------------------------------------
 * a7  G_D_onestep after the generator call  model/Diffusion.py:415-424   (uniform t)
 *   g = (sqrt(abar_{t-1}) b_t pred + sqrt(a_t)(1-abar_{t-1}) x_t)/(1-abar_t)
 *       + (1-abar_{t-1}) b_t z/(1-abar_t)
 *   o = sqrt(abar_{t-1}) x + sqrt(1-abar_{t-1}) z ;   out = M o + (1-M) g
 * z: explicit noise or NULL -> on-chip Philox.  sched_host: HOST copy of ddg_schedule's out8.
 * In-place (out == x_t) is allowed.  Requires 2 <= t <= T (t - 1 indexes abar_{t-1}): the sampler's
 * reverse steps use t >= 3 (model/Diffusion.py:431-438); t = 2 is the posterior of the discriminator
 * tail (:443-445), which ddgani_amd/model/Discriminator_model.py::sampler_disc_tail builds from this call.
 * ------------------------------------------------------------------------------------- */
int ddg_reverse_step(const float *x_t, const float *x, const float *M, const float *pred,
                     const float *z, const float *sched_host, int T, int t,
                     uint64_t seed, uint64_t stream_id, float *out, int64_t rows, int D,
                     void *stream);

/* a6 (head) + a7 fused: G_D_onestep from the generator's PRE-HEAD logits y   model/Diffusion.py:188-201, :415-424
 * out = reverse step of head(y): the per-field sigmoid / softmax and the posterior / re-noising / mask-merge formula in
 * one HBM pass (the prediction is never written).  Bit-identical to ddg_head followed by ddg_reverse_step.
 * offsets / is_cat: DEVICE arrays of the field layout.  In-place (out == x_t) is allowed. */
int ddg_head_reverse_step(const float *y, const float *x_t, const float *x, const float *M, const float *z,
                          const float *sched_host, int T, int t, uint64_t seed, uint64_t stream_id,
                          const int32_t *offsets, const uint8_t *is_cat, int C, float *out, int64_t rows, int D,
                          void *stream);

/* ---------------------------------------------------------------------------------------
 * a6 (head)  per-field sigmoid / softmax   model/Diffusion.py:188-201
 * a9  mask-merge + categorical re-softmax  model/Diffusion.py:598, :606-615
 *   ddg_head:        out = head(y)
 *   ddg_mask_merge:  code = M*x + (1-M)*xhat, then softmax over every categorical block
 * In-place allowed.
 * ------------------------------------------------------------------------------------- */
int ddg_head(const float *y, const int32_t *offsets, const uint8_t *is_cat, int C,
             float *out, int64_t rows, int D, void *stream);
int ddg_mask_merge(const float *x, const float *M, const float *xhat, const int32_t *offsets,
                   const uint8_t *is_cat, int C, float *code, int64_t rows, int D, void *stream);

/* ---------------------------------------------------------------------------------------
 * Dense layer used by a6 (Generator_x0.forward, model/Diffusion.py:179-187) and a10
 * (Discriminator_noise_x.forward, model/Discriminator_model.py:105-115):
 *
 *   Y[rows,out] = f(A)[rows,in] * W[out,in]^T + bias[out]
 *
 * f is the fused A-operand prologue:
 *   DDG_PRO_NONE     f(a) = a
 *   DDG_PRO_ADD_EMB  f(a)[r,k] = a[r,k] + emb[t_r, k]          (time_embedding add, :180)
 *   DDG_PRO_BN_RELU  f(a)[r,k] = relu(a[r,k]*scale[k]+shift[k]) * drop[r,k]
 *                    (BatchNorm apply + ReLU + Dropout of the previous layer, :184-186;
 *                     scale/shift come from ddg_bn_finalize)
 *   DDG_PRO_EMB_RELU f(a)[r,k] = relu(a[r,k] + emb[t_r,k])     (Discriminator_model.py:110-112)
 * A may be the concatenation [A | A2] along columns (torch.cat, Discriminator_model.py:106):
 * columns [0,in1) come from A (row stride in1), [in1,in) from A2 (row stride in-in1);
 * A2 == NULL -> in1 must equal in.
 * drop: uint8[rows,in] keep-mask (1 = keep, scaled by 1/(1-p)) or NULL; when NULL and
 * drop_p > 0 the mask is generated on chip (Philox, seed/stream_id); drop_p == 0 -> none.
 * t: int64[rows] or NULL (t_uniform).  emb: [n_steps,in]; ddg_linear cannot know n_steps: its caller guarantees
 * 0 <= t < n_steps.  The composed entry points (ddg_denoiser_fwd, ddg_disc_fwd, the training steps,
 * ddg_forward_noise) clamp a timestep outside its table instead of reading past it (torch would raise).
 * stats (optional, may be NULL): double[2*out]; the kernel ADDS column sums of Y and Y^2
 * (fp64) into it (zero it first; multi-GPU callers all-reduce it before ddg_bn_finalize).
 * ------------------------------------------------------------------------------------- */
#define DDG_PRO_NONE 0
#define DDG_PRO_ADD_EMB 1
#define DDG_PRO_BN_RELU 2
#define DDG_PRO_EMB_RELU 3

int ddg_linear(const float *A, const float *A2, int in1, const float *W, const float *bias,
               int prologue, const float *emb, const int64_t *t, int t_uniform,
               const float *scale, const float *shift, const uint8_t *drop, float drop_p,
               uint64_t seed, uint64_t stream_id, float *Y, double *stats,
               int64_t rows, int in, int out, void *stream);

/* BatchNorm1d (train mode, biased variance, eps)   model/Diffusion.py:151,156,161
 * scale[k] = gamma[k]/sqrt(var[k]+eps), shift[k] = beta[k]-mean[k]*scale[k] from the fp64
 * column sums produced by ddg_linear over `total_rows` rows (the GLOBAL row count). */
int ddg_bn_finalize(const double *stats, const float *gamma, const float *beta, double total_rows,
                    float eps, float *scale, float *shift, int width, void *stream);

/* ---------------------------------------------------------------------------------------
 * a6  Generator_x0.forward (train mode)     model/Diffusion.py:145-201
 * One call = time-embedding add, 3 x [Linear, BatchNorm(batch stats), ReLU, Dropout(p)],
 * Linear, per-field head.  Single-GPU convenience over ddg_linear/ddg_bn_finalize/ddg_head
 * (row-sharded callers drive the layers themselves and all-reduce the BN sums in between).
 * ------------------------------------------------------------------------------------- */
typedef struct ddg_gen_weights {
    int32_t in_dim, enc_dim, lat_dim, n_steps;  /* D, 2D, D, T+1 (BaseLine/DiffGANI.py:54) */
    const float *W1, *b1, *g1, *be1;            /* linears.0 [enc,in], linears.1 BN [enc]   */
    const float *W2, *b2, *g2, *be2;            /* linears.4 [lat,enc], linears.5 BN [lat]  */
    const float *W3, *b3, *g3, *be3;            /* linears.8 [enc,lat], linears.9 BN [enc]  */
    const float *W4, *b4;                       /* linears.12 [in,enc]                      */
    const float *temb;                          /* time_embedding.weight [n_steps,in]       */
} ddg_gen_weights;

/* (from 8 192 rows on the size includes room for the four weight matrices pre-split into f16 halves) */
size_t ddg_denoiser_workspace_bytes(int64_t rows, int in_dim, int enc_dim, int lat_dim);
/* drop0..2: explicit keep-masks uint8 [rows,enc],[rows,lat],[rows,enc] or NULL (on-chip RNG
 * when drop_p > 0).  apply_head == 0 leaves the pre-head logits in `out`. */
int ddg_denoiser_fwd(const ddg_gen_weights *w, const float *x_t, const int64_t *t, int t_uniform,
                     const uint8_t *drop0, const uint8_t *drop1, const uint8_t *drop2, float drop_p,
                     uint64_t seed, uint64_t stream_id, float bn_eps,
                     const int32_t *offsets, const uint8_t *is_cat, int C, int apply_head,
                     float *out, int64_t rows, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * ddg_denoiser_fwd_static: NO COUNTERPART IN THE REFERENCE.  ddg_denoiser_fwd for callers that
 * cannot allocate device memory or carry a workspace: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  Same arguments and results as ddg_denoiser_fwd minus the
 * workspace; the layer activations and BatchNorm sums live in a static device array of
 * DDG_STATIC_SCRATCH_BYTES that each GPU keeps until the process ends.  It allocates nothing.
 * rows is limited by ddg_denoiser_workspace_bytes(rows, in, enc, lat) <= DDG_STATIC_SCRATCH_BYTES
 * (about 15 400 rows at D = 176); larger inputs return DDG_E_SCRATCH_LIMIT.  All calls on one GPU
 * share that GPU's array: calls that overlap on one GPU (two streams, two threads) race.  The first
 * call on a device looks up the array's device address (hipGetSymbolAddress); make it outside a
 * graph capture.
 * ------------------------------------------------------------------------------------- */
#define DDG_STATIC_SCRATCH_BYTES (32u << 20)
int ddg_denoiser_fwd_static(const ddg_gen_weights *w, const float *x_t, const int64_t *t, int t_uniform,
                            const uint8_t *drop0, const uint8_t *drop1, const uint8_t *drop2, float drop_p,
                            uint64_t seed, uint64_t stream_id, float bn_eps,
                            const int32_t *offsets, const uint8_t *is_cat, int C, int apply_head,
                            float *out, int64_t rows, void *stream);

/* ---------------------------------------------------------------------------------------
 * a10  Discriminator_noise_x.forward        model/Discriminator_model.py:82-115
 * ------------------------------------------------------------------------------------- */
typedef struct ddg_disc_weights {
    int32_t in_dim, lat_dim, out_dim, n_steps;  /* 2D, D/2, C, T+1 (BaseLine/DiffGANI.py:53) */
    const float *W1, *b1, *W2, *b2, *W3, *b3, *W4, *b4;   /* linears.0,2,4,6               */
    const float *e1, *e2, *e3;                  /* step_embeddings.{0,1,2}.weight [n_steps,lat] */
} ddg_disc_weights;

size_t ddg_disc_workspace_bytes(int64_t rows, int lat_dim);
int ddg_disc_fwd(const ddg_disc_weights *w, const float *xtj1, const float *xt, const int64_t *t,
                 int t_uniform, float *out, int64_t rows, void *workspace, size_t workspace_bytes,
                 void *stream);

/* ---------------------------------------------------------------------------------------
 * a11-a15  one minibatch training step      model/Diffusion.py:245-379, :452-526, :571-573
 *
 * The reference's step is: sample_x (128 rows) -> randint t in [2,T] -> forward_diffusion ->
 * Generator_x0 (train mode) -> losses -> autograd backward -> Adam.  Here the caller gathers the
 * batch (rows x0/M/m), draws or uploads the randomness, and calls
 *   ddg_train_gen_grads   loss_G = w_res*res_loss_func + w_gd*G_d_loss_func  (:297-303; the FD and
 *                         Learner terms are constants under no_grad) and the
 *                         gradient of every generator parameter (what loss.backward() leaves in
 *                         .grad after optimizer.zero_grad(), :309-310)
 *   ddg_train_disc_grads  loss_D = loss_scale*D_loss_func (:364-367, loss_scale = D_weight on
 *                         epochs divisible by 100, :338) and the gradient of every discriminator
 *                         parameter (:369-370)
 *   ddg_adam_step         torch.optim.Adam.step(): lr 1e-4, betas (0.9,0.999), eps 1e-8, coupled
 *                         weight_decay 1e-4 for the generator / 0 for the discriminator (:571-573)
 * Gradient structs mirror the weight structs (same tensor shapes).  losses: device floats.
 * eps / drop0..2 == NULL -> drawn on chip (Philox, seed/stream_id) exactly as ddg_denoiser_fwd does.
 * pred_out (optional): the generator's prediction [rows,D] of this step.
 * ------------------------------------------------------------------------------------- */
/* Stream offset in device memory, for callers that replay a captured HIP graph (no reference counterpart: the
 * reference draws torch.randn_like per call).  A graph freezes kernel arguments, the random-stream numbers included.
 * With dev_ptr set (a device uint64 owned by the caller), every on-chip draw of every entry point uses
 * stream_id + *dev_ptr, read when the kernel RUNS: advance the word between replays and the noise is fresh.
 * Process-wide for the current device until called again (NULL = off, the default); call it outside graph capture
 * (it copies to a device symbol).  Explicit-noise arguments (eps, z, drop masks) are not affected. */
int ddg_set_stream_offset(const uint64_t *dev_ptr);

typedef struct ddg_train_batch {
    const float *x0;            /* [rows,D] sampled rows of the encoded table (sample_x, utils/util.py:582) */
    const float *M;             /* [rows,D] expanded mask                                     */
    const float *m;             /* [rows,C] column mask as float (m_data, model/Diffusion.py:574) */
    const int64_t *t;           /* [rows] timesteps in [2,T]                                  */
    const float *eps;           /* [rows,D] forward-diffusion noise or NULL                   */
    const uint8_t *drop0, *drop1, *drop2;   /* keep masks [rows,enc],[rows,lat],[rows,enc] or NULL */
    float drop_p;
    uint64_t seed, stream_id;
    int64_t rows;
} ddg_train_batch;

typedef struct ddg_gen_grads {
    float *W1, *b1, *g1, *be1, *W2, *b2, *g2, *be2, *W3, *b3, *g3, *be3, *W4, *b4, *temb;
} ddg_gen_grads;

typedef struct ddg_disc_grads {
    float *W1, *b1, *W2, *b2, *W3, *b3, *W4, *b4, *e1, *e2, *e3;
} ddg_disc_grads;

size_t ddg_train_workspace_bytes(int64_t batch_rows, int in_dim, int enc_dim, int lat_dim, int disc_lat_dim,
                                 int C);
/* losses_dev: float[2] = {res_loss, g_d_loss} (unweighted values) */
int ddg_train_gen_grads(const ddg_gen_weights *gw, const ddg_disc_weights *dw, const ddg_train_batch *b,
                        const float *sched_dev, int T, const int32_t *offsets_dev, const uint8_t *is_cat_dev,
                        int C, float w_res, float w_gd, float bn_eps, const ddg_gen_grads *grads,
                        float *losses_dev, float *pred_out, void *workspace, size_t workspace_bytes,
                        void *stream);
/* loss_dev: float[1] = D_loss (unscaled value) */
int ddg_train_disc_grads(const ddg_gen_weights *gw, const ddg_disc_weights *dw, const ddg_train_batch *b,
                         const float *sched_dev, int T, const int32_t *offsets_dev, const uint8_t *is_cat_dev,
                         int C, float loss_scale, float bn_eps, const ddg_disc_grads *grads, float *loss_dev,
                         float *pred_out, void *workspace, size_t workspace_bytes, void *stream);
/* ---------------------------------------------------------------------------------------
 * (e) data-parallel training step: the B = 128 minibatch of model/Diffusion.py:287-311 / :356-372
 * sharded by rows over the ranks of one job (one process per GPU), with synchronised BatchNorm.
 * No counterpart in the reference (single device); results equal the single-process step on the same
 * global batch up to fp32 summation order.
 *
 * Each rank calls the *_dp entry point on ITS rows of the global batch (ddg_train_batch.rows = local
 * rows >= 1; eps / drop masks / t are the rank's rows of the global draws).  The library has no RCCL
 * inside: wherever the step reduces over the batch it calls `allreduce` (sum, in place, `count`
 * elements of fp32 (is_f64 = 0) or fp64 (is_f64 = 1) at the DEVICE pointer `buf`, ordered on `stream`;
 * return 0 on success):
 *   forward   3 x BatchNorm column sums (sum z, sum z^2), fp64 [2*width]
 *   losses    res_loss_func's per-field row counts, fp32 [C]            (generator step only)
 *   backward  3 x BatchNorm backward column sums (sum da, sum da*xhat), fp32 [2*width]   (generator step only)
 * `buf` always lies inside the workspace passed to the call.  Outputs are this rank's SHARE: gradients
 * and losses are partial sums already normalised by the global counts, so the caller's one all-reduce
 * (sum) over the gradient tensors and the loss floats completes them (ddgani_amd/dist.py:
 * sharded_train_step keeps them in one flat buffer).  dp == NULL or dp->allreduce == NULL: the
 * single-process step (ddg_train_gen_grads / ddg_train_disc_grads).
 * ------------------------------------------------------------------------------------- */
typedef int (*ddg_allreduce_fn)(void *ctx, void *buf, int64_t count, int is_f64, void *stream);
typedef struct ddg_dp {
    int64_t global_rows;        /* rows of the whole minibatch (128 in the reference) */
    ddg_allreduce_fn allreduce;
    void *ctx;
} ddg_dp;
int ddg_train_gen_grads_dp(const ddg_gen_weights *gw, const ddg_disc_weights *dw, const ddg_train_batch *b,
                           const float *sched_dev, int T, const int32_t *offsets_dev, const uint8_t *is_cat_dev,
                           int C, float w_res, float w_gd, float bn_eps, const ddg_gen_grads *grads,
                           float *losses_dev, float *pred_out, void *workspace, size_t workspace_bytes,
                           void *stream, const ddg_dp *dp);
int ddg_train_disc_grads_dp(const ddg_gen_weights *gw, const ddg_disc_weights *dw, const ddg_train_batch *b,
                            const float *sched_dev, int T, const int32_t *offsets_dev, const uint8_t *is_cat_dev,
                            int C, float loss_scale, float bn_eps, const ddg_disc_grads *grads, float *loss_dev,
                            float *pred_out, void *workspace, size_t workspace_bytes, void *stream,
                            const ddg_dp *dp);

/* The minibatch gather of a training step in one launch: x0 = x[idx], Mb = M[idx] ([B, D]), mb = m[idx] ([B, C]) --
 * sample_x's x[rows, :] (utils/util.py:582-585; model/Diffusion.py:287-289, 356-358 index the table, the expanded mask
 * and the [N, C] mask with the same rows).  idx: int64[B] on the device, clamped into [0, n_rows).  counter / offset
 * (both or neither; no reference counterpart): the book-keeping of a step replayed as a captured graph -- offset[0] =
 * counter[0] before the step's first on-chip draw (see ddg_set_stream_offset); ddg_step_end adds 2^32 to the counter
 * and clears the offset after its last. */
int ddg_gather_batch(const float *x, const float *M, const float *m, const int64_t *idx, int64_t n_rows, int B, int D,
                     int C, float *x0, float *Mb, float *mb, const int64_t *counter, int64_t *offset, void *stream);
int ddg_step_end(int64_t *counter, int64_t *offset, void *stream);

/* host arrays of n_tensors (<= 16) device pointers; step counts from 1 */
int ddg_adam_step(int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                  float *const *exp_avg_sq, const int64_t *numel, int step, double lr, double beta1,
                  double beta2, double eps, double weight_decay, void *stream);

/* ---------------------------------------------------------------------------------------
 * a1 + a2  init_attn_2 core: row-attention top-k fill   utils/util.py:439-496
 *
 * For every local query row j and table column c with mq[j,c]==0:
 *   logits  s_i = sum_d W[c,d] * Xq0[j,d] * Xk0[i,d]     over keys i with mk[i,c]==1, i != j
 *   tau     = k-th largest logit (all valid keys kept when fewer than k exist; ties kept)
 *   fill    = sum_{s_i >= tau} softmax_i * Xk0[i, blk(c)]
 *   out_code[j, blk(c)] = fill          ;  cells with mq[j,c]==1 keep Xq0[j, blk(c)]
 * which is the closed form of softmax(dim=0) -> top-k probability threshold -> renormalise
 * -> P^T X0 -> mask-merge.  Xq0/Xk0 are the encoded rows already multiplied
 * by their expanded masks (X0 = X*M, utils/util.py:439).  Queries are the caller's row shard
 * (q_row_offset = global index of its first row); keys are all rows (after the all-gather
 * in the multi-GPU case).  On one GPU pass the same pointers for queries and keys.
 * mq/mk: uint8 0/1 [rows,C].  W: [C,D].  offsets: int32[C+1] on the HOST here.
 * k >= 1.  k <= DDG_MAX_TOPK_REG selects AND fills in registers (one similarity pass; a two-way
 * tie at the k-th logit keeps both keys right there, pairs with a longer exact tie take a second
 * pass).  Larger k (the reference's top_k = 0.05 means k = 0.05 N) takes the WINDOW selection on
 * tables of >= 16 384 keys: the k-th logit is bracketed on a strided sample of a column's keys (three
 * cheap passes; the sample grows with the key count so that the bracket's bins stay listable), one
 * full pass histograms the bracket (exact count above it, the bin of the k-th logit, its rank in the
 * bin), one full pass accumulates everything above that bin and lists the bin's keys, and the k-th
 * logit is selected exactly inside the list: two full passes on the f16 MFMA with both operands split
 * in two halves (fp32-class products, |error| <= 3 * 2^-22 * sum |q||k|).  Smaller tables, pairs
 * whose bracket missed and DDG_ATTN_RADIX=1 use the exact 5-level radix select on fp32 logits
 * (5 similarity passes + a fill pass).
 * Optional debug outputs (NULL to skip): out_tau [Nq,C] (NaN where mq==1, -inf when all
 * valid keys are kept), out_nvalid int32 [Nq,C].
 * ------------------------------------------------------------------------------------- */
/* mode: how the similarity of the register path (k <= DDG_MAX_TOPK_REG) is evaluated.
 *   DDG_ATTN_F32          fp32 MFMA (v_mfma_f32_32x32x2_f32) for every logit.
 *   DDG_ATTN_F16_FILTER   (= DDG_ATTN_BF16_FILTER, the name of round 1, whose screen was bf16)
 *                         16-bit MFMA (v_mfma_f32_32x32x16_f16 on power-of-two scaled operands) screens
 *                         every logit with a rigorous error margin; every key that could reach the k+2
 *                         largest exact logits is re-scored in fp32, and only fp32 values are ranked and
 *                         weighted: the kept sets and fills are those of fp32 arithmetic (up to fp32
 *                         summation order), at the 16-bit matrix rate.  Larger k ignores the mode. */
#define DDG_ATTN_F32 0
#define DDG_ATTN_BF16_FILTER 1
#define DDG_ATTN_F16_FILTER DDG_ATTN_BF16_FILTER

size_t ddg_attn_workspace_bytes(int64_t n_query_rows, int64_t n_key_rows, int C, int D, int64_t k);
int ddg_attn_topk_fill(const float *Xq0, const uint8_t *mq, int64_t q_row_offset,
                       const float *Xk0, const uint8_t *mk, const float *W,
                       const int32_t *offsets_host, int64_t n_query_rows, int64_t n_key_rows,
                       int C, int D, int64_t k, int mode, float *out_code, float *out_tau, int32_t *out_nvalid,
                       void *workspace, size_t workspace_bytes, void *stream);

/* The three phases of ddg_attn_topk_fill, separately callable on the same workspace (same
 * reference lines; used to time the phases and, multi-GPU, to prepare once per all-gather):
 *   prepare: padded operands + per-column key / query index lists      (utils/util.py:439-448);
 *            on the screening path (filter mode, k <= DDG_MAX_TOPK_REG) the padded key rows are stored in
 *            descending order of their 2-norm: which keys a query keeps does not depend on the order,
 *            how many candidates reach the exact pass does (DDG_KEY_ORDER=row: table order)
 *   select : out_code <- Xq0, then ONE similarity pass                 (utils/util.py:452-466)
 *            k <= DDG_MAX_TOPK_REG: the k+2 largest logits and their key rows are kept in
 *            registers; every pair whose k-th and (k+1)-th logits differ, or tie while the
 *            (k+2)-th is smaller, is filled right there (softmax over the kept k or k+1, gather
 *            of the key values: utils/util.py:484-496); pairs whose tie at the k-th logit may
 *            run past the list are flagged for the fill phase;
 *            larger k: window selection (three sample passes + one full histogram pass), or the exact
 *            5-level radix select of tau (5 similarity passes) below 16 384 keys
 *   fill   : softmax over {s >= tau}, P^T X0, mask-merge (one more similarity pass) for the
 *            pairs select left open: the flagged ones, or all of them for larger k (window path:
 *            the accumulate pass + in-list selection, then the radix path for flagged pairs)
 *                                                                      (utils/util.py:463-496)
 * k and mode must be the same in the three calls.
 * queries_alias_keys: 1 when prepare was called with identical query / key arrays. */
int ddg_attn_prepare(const float *Xq0, const uint8_t *mq, int64_t q_row_offset, const float *Xk0,
                     const uint8_t *mk, const float *W, const int32_t *offsets_host,
                     int64_t n_query_rows, int64_t n_key_rows, int C, int D, int64_t k, int mode,
                     void *workspace, size_t workspace_bytes, void *stream);
int ddg_attn_select(const float *Xq0, int64_t n_query_rows, int64_t n_key_rows, int C, int D, int64_t k,
                    int mode, int queries_alias_keys, float *out_code, void *workspace,
                    size_t workspace_bytes, void *stream);
int ddg_attn_fill(const int32_t *offsets_host, int64_t n_query_rows, int64_t n_key_rows, int C, int D,
                  int64_t k, int mode, int queries_alias_keys, float *out_code, float *out_tau,
                  int32_t *out_nvalid, void *workspace, size_t workspace_bytes, void *stream);
/* Tuning / tests (no reference counterpart): key-chunk schedule of the screening path for all later calls of this
 * process.  positions > 0: fixed chunks of that many key positions (rounded down to a multiple of 128, at least 128);
 * 0: the default growing schedule; < 0: back to what DDG_FILT_CHUNK in the environment said at the first call.
 * Results do not depend on the schedule (tests/test_gpu_parity.py holds two schedules against each other). */
int ddg_attn_set_key_chunk(int64_t positions);
/* Diagnostics (no reference counterpart): out_count_dev[0] <- the number of pairs that the last ddg_attn_fill on
 * this workspace sent to its exact fp32 fallback -- window path: bracket missed or bin above the list capacity;
 * register paths: ties longer than the list, candidate-slot overflow.  Undefined after the plain radix path. */
int ddg_attn_fallback_pairs(int64_t n_query_rows, int64_t n_key_rows, int C, int D, int64_t k,
                            int queries_alias_keys, int32_t *out_count_dev, void *workspace,
                            size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * (f1) the "data-utility" Learner: train_L_code                 model/Learner.py:43-67
 *
 * `epochs` full-batch SGD steps (learning rate `lr`, the reference uses 0.01) of
 *   p = softmax(x W^T + b),  loss = CrossEntropyLoss(p, y)   (log_softmax applied to p again, :36-39, :47)
 * on the training rows, then the same loss and the number of correct arg-max predictions on the
 * validation rows (:61-66).  W [L][D] and b [L] are the fc2 parameters of the reference's `L` module:
 * initial values in (the caller draws them with the reference's seed and construction order, :44-45),
 * trained values out.  x_*: [n][D] fp32, y_*: int32 class indices in [0, L).  One fused pass over
 * x_train per step.  Supported: L <= 16 with D <= 1023, or L <= 64 with D <= 255.
 * out_val_loss [1] and out_val_correct [1] are device pointers.
 * ------------------------------------------------------------------------------------- */
size_t ddg_learner_workspace_bytes(int D, int L);
int ddg_learner_train(const float *x_train, const int32_t *y_train, int64_t n_train,
                      const float *x_valid, const int32_t *y_valid, int64_t n_valid, int D, int L,
                      int epochs, float lr, float *W, float *b, float *out_val_loss,
                      int32_t *out_val_correct, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * (f4) classifiers of the "data-dependency" plug-in   model/FD_model.py:12-65, :76-111
 *
 * FDModel.forward = fc1 [in,in] -> ReLU -> fc3 [out,in] (fc2 exists but is never used, :22-26).
 *   ddg_fd_train_epoch  one pass of train_Model's inner loop (:41-46): SGD (lr, CrossEntropyLoss mean) over the
 *                       rows of X in the order perm[0..n) (the DataLoader's shuffle of this epoch, int64 on the
 *                       device), `batch` rows per step (256 in the reference), last batch ragged; W1/b1/W3/b3 are
 *                       updated in place.
 *   ddg_fd_eval         test_model (:55-65): out_correct = rows whose arg-max equals Y;  get_my_FD_loss's term
 *                       (:106-108): out_ce_mean = CrossEntropyLoss(model(X), Y) (mean).  Either output may be NULL.
 *                       Rows are processed workspace_rows at a time (the workspace is sized for that many).
 * X [n,in] fp32, Y int32 [n] class indices in [0,out).  Workspace: ddg_fd_workspace_bytes(rows per step, in, out).
 * ------------------------------------------------------------------------------------- */
typedef struct ddg_fd_weights {
    int32_t in_dim, out_dim;
    float *W1, *b1, *W3, *b3;
} ddg_fd_weights;
size_t ddg_fd_workspace_bytes(int64_t batch_rows, int in_dim, int out_dim);
int ddg_fd_train_epoch(const ddg_fd_weights *w, const float *X, const int32_t *Y, const int64_t *perm, int64_t n,
                       int batch, float lr, void *workspace, size_t workspace_bytes, void *stream);
int ddg_fd_eval(const ddg_fd_weights *w, const float *X, const int32_t *Y, int64_t n, int32_t *out_correct,
                float *out_ce_mean, void *workspace, size_t workspace_bytes, int64_t workspace_rows, void *stream);

/* ---------------------------------------------------------------------------------------
 * (f1, rest) the plain discriminator D trained inside get_valid_data_index      utils/util.py:623-646
 *                                                                               model/Discriminator_model.py:5-39
 * p = sigmoid(fc3(drop(relu(fc2(drop(relu(fc1(x))))))))  with fc1, fc2: [in][in], fc3: [out][in]; Dropout(0.2) is live in
 * every pass (the module stays in train mode, also when it scores the rows, :638).
 * ddg_dplain_grads = zero_grad + forward + backward of  -mean(m log(p + 1e-8) + (1 - m) log(1 - p + 1e-8))  (:633-637)
 * over ALL `rows` rows (full batch): the gradients of the six tensors go to `grads` (same struct, gradient buffers); the
 * caller applies torch.optim.Adam(lr 0.002) with ddg_adam_step.  ddg_dplain_fwd = the scoring pass: p_out [rows][out].
 * drop1 / drop2: uint8 keep masks [rows][in] (the caller's draws, e.g. torch's CPU generator in the reference's order) or
 * NULL: Philox on chip (seed, stream_id * 2 + layer).  x [rows][in], m [rows][out] fp32.
 * ------------------------------------------------------------------------------------- */
typedef struct ddg_dplain_weights {
    int32_t in_dim, out_dim;
    float *W1, *b1, *W2, *b2, *W3, *b3;
} ddg_dplain_weights;
size_t ddg_dplain_workspace_bytes(int64_t rows, int in_dim, int out_dim);
int ddg_dplain_grads(const ddg_dplain_weights *w, const float *x, const float *m, const uint8_t *drop1,
                     const uint8_t *drop2, float drop_p, uint64_t seed, uint64_t stream_id, int64_t rows,
                     const ddg_dplain_weights *grads, void *workspace, size_t workspace_bytes, void *stream);
int ddg_dplain_fwd(const ddg_dplain_weights *w, const float *x, const uint8_t *drop1, const uint8_t *drop2, float drop_p,
                   uint64_t seed, uint64_t stream_id, int64_t rows, float *p_out, void *workspace, size_t workspace_bytes,
                   void *stream);

/* ---------------------------------------------------------------------------------------
 * (f2) column-association map: the C^2 crosstabs of get_attr_map_Null_kendall
 *                                                   utils/get_attr_attn.py:61-77 (pd.crosstab per column pair)
 *
 * codes_t: uint16 [C][N], COLUMN-major: the discretised table (20-bin numerics :49-51, ordinal categoricals :54),
 * DDG_CODE_MISSING where data_m == 0 (the reference drops those rows per pair, :66-70).  card [C]: codes of column
 * c lie in [0, card[c]).  For every listed pair p = (pair_i[p], pair_j[p]) the contingency table
 *   tables[pair_off[p] + a * card[j] + b] = #{rows r : codes[i][r] == a, codes[j][r] == b, both observed}
 * total_cells = pair_off[n_pairs].  All arrays on the device.  The chi-square / Cramer's-V arithmetic on the
 * tables (:9-18) stays on the host (ddgani_amd/utils/get_attr_attn.py): C^2 tiny tables.
 * ------------------------------------------------------------------------------------- */
#define DDG_CODE_MISSING 65535u
int ddg_contingency(const uint16_t *codes_t, int64_t N, int C, const int32_t *card_dev, const int32_t *pair_i_dev,
                    const int32_t *pair_j_dev, const int64_t *pair_off_dev, int64_t n_pairs, int64_t total_cells,
                    uint32_t *tables, void *stream);

/* ---------------------------------------------------------------------------------------
 * (f3) decode + score of an encoded table: reconvert_data   utils/util.py:228-254
 *                                           errorLoss       utils/util.py:83-139 (normalization :17-48)
 *
 * One pass over the MISSING cells (m == 0) of code [N,D]: decode (numeric x / K + Min, utils/field.py:75-82, in the
 * arithmetic numpy uses for it: dec_f32[c] = 1 when the field's K / Min are Python floats (object-typed column: the
 * fp32 code stays fp32), 0 when they are numpy float64 (fp64); categorical: value of the arg-max position, first
 * maximum), normalise with the ground
 * truth's (min, max + 1e-6), and add per column into sums [C][3] (fp64; zero it first):
 *   [c][0] sum d^2, [c][1] sum |d|   (numeric columns),   [c][2] cells that differ (categorical columns)
 * with d = ori_norm[r,c] - normalised decoded value.  ori_norm [N,C] fp64: the normalised ground truth.
 * dec_K / dec_min / norm_min / norm_den: fp64 [C]; pos_value: fp64 [D] value of every one-hot position.
 * ARMSE / AMAE = means over the columns with missing cells of sqrt(sum d^2 / n_miss) resp. sum |d| / n_miss,
 * categorical columns contributing their mismatch rate to both (utils/util.py:112-139): finished on the host.
 * ------------------------------------------------------------------------------------- */
int ddg_score_table(const float *code, const double *ori_norm, const uint8_t *m, const int32_t *offsets_dev,
                    const uint8_t *is_cat_dev, const double *dec_K, const double *dec_min, const uint8_t *dec_f32,
                    const double *pos_value, const double *norm_min, const double *norm_den, double *sums, int64_t N,
                    int C, int D, void *stream);

/* ---------------------------------------------------------------------------------------
 * (f3) encode of a table: Data_convert                              utils/util.py:52-78
 *      NumericalField.convert  utils/field.py:65-67   mode 0 ("minmax"):   scale * (x - shift)   scale = K, shift = Min
 *                                                      mode 1 ("mean_std"): (x - shift) / scale   scale = sigma, shift = mu
 *      CategoricalField.convert utils/field.py:138-150 one-hot over the sorted distinct values
 * vals [N][C] fp64: numeric value, or the position of a categorical cell's value among its column's sorted distinct
 * values.  fp64 arithmetic rounded once to fp32, as the reference's pandas columns -> torch.tensor(dtype=float):
 * bit-identical to the host mirror for the same field constants.  code [N][D] fp32.  All arrays on the device.
 * ------------------------------------------------------------------------------------- */
int ddg_encode_table(const double *vals, const int32_t *offsets_dev, const uint8_t *is_cat_dev, const uint8_t *mode_dev,
                     const double *scale_dev, const double *shift_dev, float *code, int64_t N, int C, int D,
                     void *stream);

/* ---------------------------------------------------------------------------------------
 * (f3) decode of a whole encoded table: reconvert_data             utils/util.py:228-254
 * out[r,c] = obs[r,c] * m[r,c] + decoded[r,c] * (1 - m[r,c])  in fp64 (:248); decoded as in ddg_score_table (same
 * dec_K / dec_min / dec_f32 / pos_value); obs [N][C] fp64: the table whose observed cells are kept, categorical
 * columns as ordinal codes (the caller maps codes back to labels: enc.inverse_transform, :251-253).
 * ------------------------------------------------------------------------------------- */
int ddg_decode_table(const float *code, const uint8_t *m, const double *obs, const int32_t *offsets_dev,
                     const uint8_t *is_cat_dev, const double *dec_K, const double *dec_min, const uint8_t *dec_f32,
                     const double *pos_value, double *out, int64_t N, int C, int D, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DDGANI_H */
