/*
 * dcv.h -- C-ABI of libdcv.so, the MI355X (gfx950) engine behind deep_cartograph's CV-fit
 * hot path (train_colvars: pca / tica / htica / ae / deep_tica, projection, k-means).
 *
 * The reference has no FFI for this path: the seam is the Python class set in
 * deep_cartograph/modules/cv_learning/cv_calculator.py and the functions of
 * deep_cartograph/modules/statistics/statistics.py.  Each entry point
 * below names the reference code it replaces (file:line relative to
 * /root/reference/deep_cartograph).  INTEGRATION.md shows the ctypes stub a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative DCV_E* code on failure;
 *     dcv_last_error() returns a thread-local, NUL-terminated description;
 *   - pointers named *_d are DEVICE pointers (HIP, current device), *_h are HOST pointers;
 *     the caller owns every buffer, the library never frees caller memory;
 *   - matrices are row-major; `ld` is the row stride in ELEMENTS;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     enqueued asynchronously on it, functions with *_h outputs synchronise that stream;
 *   - workspace sizes are queried with the matching *_workspace() call; workspaces are
 *     plain device memory with no state between calls;
 *   - one process drives one GPU; multi-GPU runs shard frames over ranks and all-reduce the
 *     small result buffers documented per call.
 */
#ifndef DCV_H
#define DCV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCV_OK 0
#define DCV_EINVAL (-1)  /* bad argument / unsupported shape */
#define DCV_EHIP (-2)    /* HIP runtime error */
#define DCV_ENOMEM (-3)  /* workspace too small / allocation failed */
#define DCV_ESTATE (-4)  /* call order violated */
#define DCV_ECALLBACK (-5) /* a host callback of the caller reported failure */

#define DCV_ABI_VERSION 5

int dcv_abi_version(void);
const char* dcv_last_error(void);
/* Number of compute units and bytes of device memory of HIP device `device`. */
int dcv_device_info(int device, int* n_cu, int64_t* hbm_bytes, char* name, size_t name_len);

/* Arithmetic of every matrix product of the library (time-lagged covariances, MLP forward / backward):
 *   DCV_GEMM_NATIVE_F32   v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate (157.3 TFLOP/s peak);
 *   DCV_GEMM_SPLIT_BF16X6 each f32 operand split into three bf16 pieces (8 + 8 + 8 mantissa bits), six
 *                         v_mfma_f32_32x32x16_bf16 products per block, f32 accumulate: the dropped cross terms
 *                         are <= 2^-23 of a product, i.e. the accuracy of an f32 multiply (measured error vs
 *                         float64 at or below the native path's), on the 16x faster BF16 matrix pipe.  Default.
 * Process-wide; also settable with the environment variable DCV_GEMM_MODE=native|split before the first product. */
#define DCV_GEMM_NATIVE_F32 0
#define DCV_GEMM_SPLIT_BF16X6 1
int dcv_set_gemm_mode(int mode);
int dcv_get_gemm_mode(void);

/* ---------------------------------------------------------------- column statistics (a1)
 * Replaces DataFrame.agg(['mean','std','min','max']) in CVCalculator.load_training_data,
 * cv_calculator.py:294-297.  One pass over X (n x F float32).  `out_d` receives 4*F
 * float64: [sum | sum of squares | min | max].  Sums combine over shards by addition,
 * min/max by min/max; mean and std(ddof=1) are finalised on the host
 * (mean = sum/n, var = (sumsq - n*mean^2)/(n-1)).  Algorithmic traffic: 4*n*F bytes read. */
size_t dcv_col_stats_workspace(int64_t n, int32_t F);
int dcv_col_stats(const float* X_d, int64_t n, int32_t F, int64_t ld, double* out_d,
                  void* ws_d, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- normalisation (a3)
 * Replaces LinearCalculator.normalize_data, cv_calculator.py:833-835
 * (data.sub_(mean).div_(range) in float32).  Y may alias X (in place, as the reference). */
int dcv_normalize(const float* X_d, float* Y_d, int64_t n, int32_t F, int64_t ldx, int64_t ldy,
                  const float* mean_d, const float* range_d, void* stream);

/* ---------------------------------------------------------------- lagged covariance (a4-a7)
 * Replaces create_timelagged_dataset + the two torch.einsum("ij,ik,i->jk") of
 * mlcolvar.core.stats.TICA.compute (call sites cv_calculator.py:2247-2261, 2309-2378) and the
 * X^T X of sklearn PCA (cv_calculator.py:2204-2207).
 * Pairs are (i, i+lag), i = 0 .. n_pairs-1; rows 0 .. n_pairs+lag-1 of X must be readable
 * (the last `lag` rows are the halo a shard borrows from its successor).  With
 * z = x - shift (shift_d may be NULL = 0), out_d receives 2F + 2F*F float64:
 *   [ a = sum z_t | b = sum z_lag | A = sum z_t z_t^T | B = sum z_t z_lag^T ]   (raw sums).
 * All four blocks combine over shards by addition.  lag = 0 computes only a and A (PCA);
 * B and b are then zero.  FP32 MFMA (v_mfma_f32_32x32x2_f32), fp32 accumulation over chunks
 * of <= 2048 rows, chunk partials summed in float64 in a fixed order (deterministic).  With a shift the column sums a, b come
 * from the kernel's own fragments, accumulated in float32 inside a chunk: shift_d must then be (close to) the column
 * means -- what the calculators pass -- for those sums to keep float64-grade accuracy; without a shift a separate float64
 * statistics pass forms them.
 * Algorithmic work: 4*n_pairs*F^2 flop (2*n*F^2 for lag 0), 4*n*F bytes. */
size_t dcv_lagged_cov_workspace(int64_t n_pairs, int32_t F, int32_t lag);
int dcv_lagged_cov(const float* X_d, int64_t n_pairs, int32_t F, int64_t ld, int32_t lag,
                   const float* shift_d, double* out_d, void* ws_d, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- linear projection (a13, a14)
 * Replaces LinearCalculator.project_data / normalize_cv, cv_calculator.py:918-991:
 *   p = ((x - fmean)/frange) @ W ; out = (p - cvmean)/cvrange.
 * fmean_d/frange_d NULL => input already normalised; cvmean_d/cvrange_d NULL => raw p.
 * bias_d (d floats or NULL) is added to p before the CV normalisation.
 * W is F x d row-major (the layout of cv_weights.npy), d <= 16.
 * minmax_d (2*d float32: per-column min then max of `out`, or NULL) combines over shards by
 * min/max.  out_d may be NULL when only the extrema are wanted.  HBM-bound:
 * 4*F + 4*d bytes per frame. */
size_t dcv_project_linear_workspace(int64_t n, int32_t F, int32_t d);
int dcv_project_linear(const float* X_d, int64_t n, int32_t F, int64_t ld,
                       const float* fmean_d, const float* frange_d, const float* W_d, int32_t d,
                       const float* bias_d, const float* cvmean_d, const float* cvrange_d,
                       float* out_d, float* minmax_d, void* ws_d, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- MLP engine (a8-a12, a15)
 * Replaces mlcolvar.cvs.DeepTICA / AutoEncoderCV + the lightning training loop driven by
 * NonLinear.train, cv_calculator.py:1456-1553 (models :2471-2492, :2569-2590), and the
 * whole-tensor forward of NonLinear.project_data / normalize_cv (:1735-1754, :1842-1891).
 *
 * A model is a chain of Linear layers over PRE-NORMALISED input (dcv_normalize applies
 * norm_in once; (x-mean)/range is elementwise, so the result is bit-identical to applying it
 * inside the model).  Parameters live in one flat float32 device buffer, layer by layer,
 * weight (out x in, row-major, torch.nn.Linear layout) then bias (out).
 */
#define DCV_ACT_NONE 0
#define DCV_ACT_LEAKY_RELU 1 /* slope 0.01 */
#define DCV_ACT_RELU 2
#define DCV_ACT_TANH 3
#define DCV_ACT_ELU 4
#define DCV_ACT_SOFTPLUS 5
#define DCV_ACT_SHIFTED_SOFTPLUS 6 /* mlcolvar Shifted_Softplus: softplus(z) - softplus(0) */
#define DCV_ACT_CUSTOM_SIGMOID 7   /* mlcolvar Custom_Sigmoid: 1 / (1 + exp(-3 z)); forced on the decoder output for
                                      min_max_range1 features, cv_calculator.py:1193-1198 */
#define DCV_MAX_LAYERS 16

/* torch.optim.<name> as selected by optimizer.name of the training configuration (cv_calculator.py:1377-1380,
 * model.optimizer_name :1511); single-tensor CPU arithmetic of torch 2.x, float32 state. */
#define DCV_OPT_ADAM 0     /* lr, beta1, beta2, eps, weight_decay (L2 into the gradient), amsgrad */
#define DCV_OPT_ADAMW 1    /* decoupled weight decay: p *= 1 - lr * weight_decay */
#define DCV_OPT_SGD 2      /* lr, momentum, dampening, nesterov, weight_decay */
#define DCV_OPT_RMSPROP 3  /* lr, alpha, eps, weight_decay, momentum, centered */
#define DCV_OPT_ADAGRAD 4  /* lr, lr_decay, eps, weight_decay, initial_accumulator_value */
#define DCV_OPT_ADAMAX 5   /* lr, beta1, beta2, eps, weight_decay */
#define DCV_OPT_NADAM 6    /* lr, beta1, beta2, eps, weight_decay, opt_p[0] = momentum_decay, opt_p[1] = decoupled_weight_decay */
#define DCV_OPT_RADAM 7    /* lr, beta1, beta2, eps, weight_decay, opt_p[1] = decoupled_weight_decay */
#define DCV_OPT_ADADELTA 8 /* lr, opt_p[0] = rho, eps, weight_decay */
#define DCV_OPT_ASGD 9     /* lr, opt_p[0] = lambd, opt_p[1] = alpha, opt_p[2] = t0, weight_decay (the averaged copy `ax` is not a model
                              parameter and is not kept) */
#define DCV_OPT_RPROP 10   /* lr, opt_p[0..1] = etas (minus, plus), opt_p[2..3] = step_sizes (min, max) */

#define DCV_MODEL_DEEPTICA 1 /* loss = -sum(eig^2) of the batch TICA of nn(x_t), nn(x_lag) */
#define DCV_MODEL_AE 2       /* loss = mean(((dec(enc(xn)) - xn) * range)^2) */

typedef struct dcv_mlp_desc {
    int32_t model;                      /* DCV_MODEL_* */
    int32_t n_layers;                   /* number of Linear layers (AE: encoder + decoder) */
    int32_t dims[DCV_MAX_LAYERS + 1];   /* dims[0] = F ... dims[n_layers] */
    int32_t act[DCV_MAX_LAYERS];        /* activation after each Linear */
    int32_t latent_layer;               /* AE: index of the Linear whose output is the CV
                                           (encoder depth); Deep-TICA: n_layers */
    int32_t lag;                        /* Deep-TICA pair offset in rows */
    int32_t max_batch;                  /* largest number of samples (pairs / frames) per step */
    double tica_reg;                    /* Deep-TICA: C0 + reg*I */
    /* optimiser (torch.optim semantics, cv_calculator.py:1377-1380); fields a given optimiser does not have are ignored */
    double lr, beta1, beta2, eps, weight_decay;
    int32_t optimizer;                  /* DCV_OPT_* */
    int32_t amsgrad;                    /* Adam / AdamW */
    int32_t nesterov;                   /* SGD */
    int32_t centered;                   /* RMSprop */
    double momentum, dampening;         /* SGD, RMSprop (momentum) */
    double alpha;                       /* RMSprop smoothing constant */
    double lr_decay, initial_accumulator_value; /* Adagrad */
    double opt_p[4];                    /* further constants of the optimiser, see DCV_OPT_* */
    /* torch.nn.Dropout(p) behind the activation of each Linear (mlcolvar FeedForward order: Linear, activation,
     * dropout), active in training steps only; 0 = none.  Masks come from a counter-based generator keyed by `seed`. */
    float dropout[DCV_MAX_LAYERS];
    uint64_t seed;
    /* torch.nn.BatchNorm1d(dims[l + 1]) behind Linear l (mlcolvar FeedForward order: Linear, activation, dropout, batchnorm);
     * training steps normalise with the batch statistics (Deep-TICA: of each half of the batch separately, x_t first, as the
     * reference's two forward calls do) and update the running statistics, evaluation steps and inference use the running
     * ones.  weight / bias of the layer follow its Linear in the flat parameter buffer (dcv_mlp_param_offset which = 2, 3). */
    int32_t batchnorm[DCV_MAX_LAYERS];
    double bn_eps, bn_momentum;         /* torch defaults 1e-5, 0.1 */
    /* torch.optim's `maximize`: the update uses the negated gradient (every optimiser; dcv_mlp_grads still holds the gradient
     * of the loss).  ABI version 4. */
    int32_t maximize;
} dcv_mlp_desc;

typedef struct dcv_mlp dcv_mlp; /* opaque; owns parameters, optimiser state and workspaces */

int dcv_mlp_create(const dcv_mlp_desc* desc, dcv_mlp** out);
void dcv_mlp_destroy(dcv_mlp* m);
/* Length of the flat parameter buffer.  Every tensor starts on a 16-byte boundary, so the
 * buffer may hold padding: dcv_mlp_param_offset(m, layer, 0) is the offset (in floats) of the
 * weight of Linear `layer`, (.., 1) of its bias. */
int64_t dcv_mlp_num_params(const dcv_mlp* m);
int64_t dcv_mlp_param_offset(const dcv_mlp* m, int32_t layer, int32_t which);
/* Flat parameter / gradient buffers (device, float32, dcv_mlp_num_params elements).  The
 * gradient buffer is what a data-parallel run all-reduces (SUM) between backward and apply. */
float* dcv_mlp_params(dcv_mlp* m);
float* dcv_mlp_grads(dcv_mlp* m);
int dcv_mlp_set_params(dcv_mlp* m, const float* params_h, void* stream);
int dcv_mlp_get_params(dcv_mlp* m, float* params_h, void* stream);
int dcv_mlp_set_lr(dcv_mlp* m, double lr);
/* beta1 (Adam / AdamW) or momentum (SGD / RMSprop): what torch's OneCycleLR / CyclicLR cycle next to the
 * learning rate (cv_calculator.py:1228-1273 -> lr_scheduler). */
int dcv_mlp_set_momentum(dcv_mlp* m, double value);
/* Deep-TICA, contiguous batches (idx_d == NULL, 1 <= lag <= batch): x_lag of sample i is x_t of sample
 * i + lag, so by default the network is evaluated once on the batch + lag rows both halves share
 * (identical outputs, about half the matrix work; the gradient of a shared row is the sum of its two
 * roles).  enable = 0 evaluates the two halves separately (2 * batch rows), as gathered batches always
 * are -- used by the parity tests to check that both forms agree. */
int dcv_mlp_set_row_sharing(dcv_mlp* m, int32_t enable);
/* AE only: per-feature range of norm_in (device copy is made); needed by the loss. */
int dcv_mlp_set_feature_range(dcv_mlp* m, const float* range_h, void* stream);

/* One optimisation / evaluation step, split in phases so that a data-parallel caller can
 * all-reduce between them.  Samples of the batch: idx_d != NULL => sample j uses row
 * idx_d[j] (int64, device) else row row0 + j; Deep-TICA also reads row + lag (rows row0 .. row0 +
 * batch + lag - 1 must exist).
 * `global_batch` is the number of samples over ALL ranks (= batch on one GPU).
 *
 *   dcv_mlp_forward   forward pass (train != 0: dropout active, as in model.train(); 0: model.eval());
 *                     Deep-TICA: leaves the batch statistics
 *                     [sum f_t (d) | sum f_lag (d) | sum f_t f_t^T (d*d) | sum f_t f_lag^T (d*d)]
 *                     as float64 in dcv_mlp_stats() (all-reduce SUM across ranks);
 *                     AE: leaves [sum of squared errors] there.
 *   dcv_mlp_backward  loss + gradients of this rank's samples -> dcv_mlp_grads()
 *                     (already scaled for the global batch; all-reduce SUM across ranks).
 *                     Appends one record to the metrics log (see dcv_mlp_read_log).
 *                     train = 0: evaluation only (loss logged, no gradients).
 *   dcv_mlp_apply     optimiser update from dcv_mlp_grads().
 */
int dcv_mlp_forward(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0,
                    int32_t batch, int32_t train, void* stream);
double* dcv_mlp_stats(dcv_mlp* m);
int32_t dcv_mlp_stats_len(const dcv_mlp* m);
int dcv_mlp_backward(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0,
                     int32_t batch, int64_t global_batch, int32_t train, void* stream);
int dcv_mlp_apply(dcv_mlp* m, void* stream);
/* Data-parallel overlap hook.  With a callback set, dcv_mlp_backward(train = 1) reduces the gradients of layers
 * 1 .. n_layers-1 into dcv_mlp_grads() BEFORE it enqueues the layer-0 weight gradient (the largest product of the
 * step) and calls fn(user) on the host at that point: the caller starts the all-reduce of that part of the buffer --
 * floats [dcv_mlp_param_offset(m, 1, 0), dcv_mlp_num_params) -- on a side stream ordered after the launch stream, so
 * the exchange runs under the layer-0 product; the layer-0 part [0, dcv_mlp_param_offset(m, 1, 0)) is reduced when
 * the call returns.  fn = NULL restores the single reduction. */
int dcv_mlp_set_upper_grads_callback(dcv_mlp* m, void (*fn)(void* user), void* user);
/* One data-parallel step behind ONE entry point: forward, all-reduce of the batch statistics, backward, all-reduce of
 * the gradient buffer, optimiser update -- the sequence a caller would otherwise drive phase by phase (five calls across
 * the language boundary per step).  The collectives are the caller's: fn(user, buf_d, count, dtype, phase) must make the
 * SUM over all ranks of the `count` elements at buf_d (device memory, dtype DCV_DTYPE_F32 / DCV_DTYPE_F64) visible to work
 * that is enqueued on `stream` after it returns (an RCCL / torch.distributed all-reduce ordered against that stream does).
 * phase says what is being exchanged: DCV_DP_STATS (dcv_mlp_stats), DCV_DP_GRADS (the gradient buffer, or its layer-0
 * part), and -- with overlap != 0 and more than one layer -- DCV_DP_UPPER_START for the gradients of layers 1.., handed
 * over BEFORE the layer-0 weight gradient is enqueued: that exchange may complete asynchronously under it and is
 * awaited by the closing call fn(user, NULL, 0, DCV_DTYPE_F32, DCV_DP_WAIT).  fn returns 0, or nonzero to abort the step
 * (dcv_mlp_dp_step then returns DCV_ECALLBACK, after a DCV_DP_WAIT call that joins an exchange already started with
 * DCV_DP_UPPER_START).  train = 0: evaluation step (statistics exchanged, loss logged).
 * Batch normalisation (dcv_mlp_desc.batchnorm) is REFUSED when global_batch != batch (DCV_EINVAL): it would normalise
 * with each rank's local rows and let the running statistics of the ranks drift apart, so N ranks would no longer equal
 * one process on the union batch.
 * Replaces, on the reference side, lightning's DDP hooks around cv_calculator.py:1515-1524 (the reference itself is
 * single-process). */
#define DCV_DTYPE_F32 0
#define DCV_DTYPE_F64 1
#define DCV_DP_STATS 0
#define DCV_DP_GRADS 1
#define DCV_DP_UPPER_START 2
#define DCV_DP_WAIT 3
typedef int (*dcv_allreduce_fn)(void* user, void* buf_d, int64_t count, int32_t dtype, int32_t phase);
int dcv_mlp_dp_step(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0, int32_t batch,
                    int64_t global_batch, int32_t train, int32_t overlap, dcv_allreduce_fn fn, void* user, void* stream);
/* ---- RCCL communicator (one process per GPU; RCCL over xGMI inside a node).  EXPERIMENTAL: executed with one rank only
 * (no multi-GPU node has been available to the builder); the default transport of the Python host stays torch.distributed.
 * The collectives of a frame-sharded fit issued
 * by the library itself, in stream order with its kernels.  dcv_comm_unique_id: rank 0 creates the 128-byte id and the
 * caller's bootstrap hands it to every rank; dcv_comm_create: collective over all ranks (ncclCommInitRank on the current
 * device).  dcv_comm_allreduce: in place, op 0 = SUM, 1 = MIN, 2 = MAX, dtype DCV_DTYPE_*, enqueued on `stream`.
 * dcv_comm_dp_allreduce_fn() is an all-reduce callback for dcv_mlp_dp_step with user = the communicator, after
 * dcv_comm_bind_stream(comm, stream) with the step's launch stream: the whole data-parallel step then runs inside the
 * library (the DCV_DP_UPPER_START exchange on a side stream, joined at DCV_DP_WAIT).  librccl.so is opened at run time.
 * Replaces what a torch.distributed / lightning DDP wrapper would do around cv_calculator.py:1515-1524. */
typedef struct dcv_comm dcv_comm;
int dcv_comm_unique_id(void* id_out_128);
int dcv_comm_create(int32_t world, int32_t rank, const void* id_128, dcv_comm** out);
void dcv_comm_destroy(dcv_comm* c);
int dcv_comm_bind_stream(dcv_comm* c, void* stream);
int dcv_comm_allreduce(dcv_comm* c, void* buf_d, int64_t count, int32_t dtype, int32_t op, void* stream);
dcv_allreduce_fn dcv_comm_dp_allreduce_fn(void);
int32_t dcv_comm_world(const dcv_comm* c);
int32_t dcv_comm_rank(const dcv_comm* c);
/* Rank of this engine in a data-parallel run: mixed into the key of the dropout counters, so that ranks holding the same
 * seed mask their local rows independently (default 0). */
int dcv_mlp_set_rank(dcv_mlp* m, int32_t rank);
/* Running statistics of the batch normalisation behind Linear `layer` (dcv_mlp_desc.batchnorm): set != 0 uploads
 * running_mean / running_var (dims[layer + 1] floats each, host) and *num_batches_tracked, set == 0 downloads them.
 * dcv_mlp_set_params resets them to a fresh BatchNorm1d (mean 0, variance 1, 0 batches). */
int dcv_mlp_bn_state(dcv_mlp* m, int32_t layer, float* running_mean_h, float* running_var_h, int64_t* num_batches_tracked,
                     int32_t set, void* stream);
/* Which code path the last forward / step took: 0 = layer by layer (the MFMA block engine), 1 = the fused small-network
 * autoencoder step (one launch: snet.hip), 2 = the fused small-network Deep-TICA kernels (forward + statistics + loss head in
 * one launch, backward in a second: snet_dt.hip).  The fused forms are taken when every weight fits in one CU's LDS
 * (reference-sized networks, cv_calculator.py:2471-2590); DCV_NO_SNET=1 in the environment forces 0. */
int32_t dcv_mlp_last_path(const dcv_mlp* m);
/* Test hook: post-activation output of Linear `layer` in the last forward (rows x dims[layer + 1] floats, dense).
 * DCV_ESTATE after a fused small-network forward (dcv_mlp_last_path != 0: the activations never left LDS). */
int dcv_mlp_layer_output(dcv_mlp* m, int32_t layer, int64_t rows, float* out_d, void* stream);
/* Convenience for one GPU: forward + backward(train=1) + apply. */
int dcv_mlp_train_step(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0,
                       int32_t batch, void* stream);
/* Convenience for one GPU: forward + backward(train=0). */
int dcv_mlp_eval_step(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0,
                      int32_t batch, void* stream);
/* The training part of an epoch (the reference's Lightning training loop over the DictLoader, cv_calculator.py:1456-1553 via
 * trainer.fit) with a constant learning rate: `nsteps` training steps, step j on idx_d[j * batch, (j + 1) * batch) -- or the
 * rows row0 + [j * batch, (j + 1) * batch) when idx_d is null -- identical in launches, parameters and loss records to
 * nsteps calls of dcv_mlp_train_step; the caller's per-call cost is paid once. */
int dcv_mlp_train_steps(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0,
                        int32_t batch, int32_t nsteps, void* stream);
/* A validation pass (the reference's Lightning validation loop over the DictLoader, cv_calculator.py:1456-1553 via
 * trainer.fit): `nbatches` evaluation steps of `batch` samples each, batch j = idx_d[j * batch, (j + 1) * batch) -- or the
 * rows row0 + [j * batch, (j + 1) * batch) when idx_d is null -- appending the nbatches loss records dcv_mlp_eval_step
 * would append, in batch order, bit for bit.  Networks small enough for the fused kernels (dcv_mlp_last_path 1 / 2) are
 * evaluated many batches per launch; the others step by step.  A ragged last batch is a separate dcv_mlp_eval_step.
 * dcv_mlp_stats() is unspecified afterwards. */
int dcv_mlp_eval_steps(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0,
                       int32_t batch, int32_t nbatches, void* stream);

/* Test hook: keep / (1 - p) multipliers (0 or 1 / (1 - p)) that the dropout behind Linear `layer` applies to
 * rows [0, rows) of the batch matrix in training step number `step` (0-based count of training forwards since
 * creation / dcv_mlp_set_params); out_d is rows x dims[layer + 1] floats.  Lets the parity tests hand the oracle
 * the very masks the engine used.  dcv_mlp_dropout_step returns the number of training forwards so far. */
int dcv_mlp_dropout_mask(dcv_mlp* m, int32_t layer, int64_t step, int64_t rows, float* out_d, void* stream);
int64_t dcv_mlp_dropout_step(const dcv_mlp* m);

/* Metrics log: one record of dcv_mlp_log_width() float64 per backward call since the last
 * reset, kept on the device (no host sync inside an epoch).  Record layout:
 *   [loss | batch | Deep-TICA: C0 (d*d) | Ctau symmetrised (d*d) | mean f_t (d)].
 * The host derives eigenvalues / TICA buffers from C0 and Ctau with the reference's own
 * cholesky + eigh recipe -- a d x d solve per record. */
int32_t dcv_mlp_log_width(const dcv_mlp* m);
int dcv_mlp_reset_log(dcv_mlp* m, int32_t capacity, void* stream);
int dcv_mlp_read_log(dcv_mlp* m, double* out_h, int32_t max_records, int32_t* n_records, void* stream);

/* Per-kernel timing for the roofline report: while enabled, HIP events are recorded on the
 * launch stream around the three products of each Linear layer (class = 3*layer + {0 forward,
 * 1 wgrad, 2 dgrad}; level 1 = first layer only, 2 = every layer) for up to max_steps training
 * steps.  dcv_mlp_profile_end synchronises, stops profiling and returns the summed
 * milliseconds and launch counts per class (3*n_layers entries each). */
int dcv_mlp_profile_begin(dcv_mlp* m, int32_t max_steps, int32_t level);
int dcv_mlp_profile_end(dcv_mlp* m, double* ms_h, int32_t* counts_h);
/* Between begin and end: steps issued while paused carry no events (and may go out as graphs), so a timed region can
 * be sampled, e.g. every fourth step.  `paused`: bit 0 = no class is sampled; bits 1 / 2 / 3 = the forward / weight-gradient /
 * input-gradient launches are not sampled (a profiled launch costs the step ~7 us of command-processor work: a caller can
 * stamp the forward product on one step and the weight gradient on another).  Every class counts its own samples (up to
 * max_steps each); dcv_mlp_profile_end reports sums and counts per class. */
int dcv_mlp_profile_pause(dcv_mlp* m, int32_t paused);

/* With DCV_GRAPH=1 in the environment and a non-null stream, dcv_mlp_train_step / forward / backward(train) /
 * eval_step capture their launch sequence into a hipGraph, update the slot's instantiated graph in place (same
 * topology, new arguments) and launch it once.  Off by default: measured, it does not shorten the step on this
 * platform.  Returns how many calls went out as a graph launch. */
int64_t dcv_mlp_graph_launches(const dcv_mlp* m);
int dcv_mlp_set_graph(dcv_mlp* m, int32_t enable);   /* per-engine switch (default: DCV_GRAPH=1 in the environment) */

/* Whole-matrix inference (project / normalize_cv): y = layers[0..latent_layer)(xn);
 * Deep-TICA additionally y = (y - tmean) @ tevecs (both d floats / d*d row-major, device,
 * may be NULL); then out = (y - pmean)/prange when given.  minmax_d as in
 * dcv_project_linear.  out_d (n x d) may be NULL. */
int dcv_mlp_infer(dcv_mlp* m, const float* Xn_d, int64_t n, int64_t ld,
                  const float* tmean_d, const float* tevecs_d, const float* pmean_d,
                  const float* prange_d, float* out_d, float* minmax_d, void* stream);

/* Input-gradient pass of the neural sensitivity analysis.  Replaces
 * mlcolvar.explain.sensitivity_analysis(metric="mean_abs_val") as called by
 * NonLinear.sensitivity_analysis, cv_calculator.py:1893-1921: for the n rows given (n <= the
 * engine's row capacity; chunk larger matrices and add the results)
 *   sens[i] = sum_r | d(sum_j cv_j)/d xn[r][i] | * scale[i]          (float64, F values, overwritten)
 * gout_d (d_latent floats) is the constant gradient of sum_j cv_j with respect to the output of
 * the network layers (the TICA projection and the post-normalisation behind them are affine);
 * scale_d (F floats) is the per-feature factor (dataset std / norm_in range). */
size_t dcv_mlp_input_sensitivity_workspace(const dcv_mlp* m, int64_t n);
int dcv_mlp_input_sensitivity(dcv_mlp* m, const float* Xn_d, int64_t n, int64_t ld, const float* gout_d,
                              const float* scale_d, double* sens_d, void* ws_d, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- k-means (a17-a19)
 * Replaces the Lloyd iterations of sklearn.cluster.KMeans as driven by
 * statistics.kmeans_clustering, statistics.py:159-197 (algorithm: SURVEY.md Appendix A.8).
 * One E+M accumulation pass over P (n x d float64, d <= 16, k <= 64); x_i = P_i - offset
 * when offset_d is given (KMeans.fit centres the data by its mean; centres are then
 * expressed in that frame):
 *   label_i = argmin_j (|c_j|^2 - 2 x_i.c_j)   (first minimum wins),
 *   acc_d   = [sums (k*d) | counts (k) | inertia vs `centers` (1) | labels changed (1)] float64,
 * all of which combine over shards by addition.  labels_d (int32) is read (previous labels,
 * for the `changed` count) and overwritten.  HBM-bound: 8*d + 8 bytes per point. */
size_t dcv_kmeans_workspace(int64_t n, int32_t d, int32_t k);
int dcv_kmeans_step(const double* P_d, int64_t n, int32_t d, const double* offset_d /* d or NULL */,
                    const double* centers_d, int32_t k, int32_t* labels_d, double* acc_d, double* mindist_d /* n or NULL */,
                    void* ws_d, size_t ws_bytes, void* stream);

/* k-means++ seeding passes.  Replace the distance / potential arithmetic of sklearn.cluster._kmeans._kmeans_plusplus as
 * reached from statistics.kmeans_clustering, statistics.py:189 (init='k-means++'); the
 * random draws and the sequential float64 cumsum + searchsorted that turn them into candidate rows stay with the caller.
 * Points P (n x d float64, d <= 8), x_i = P_i - offset; dist(x, c) = max(-2 x.c + |c|^2 + |x|^2, 0).
 *   dcv_kmeanspp_update      closest_i = first ? dist(x_i, centre) : min(closest_i, dist(x_i, centre));
 *                            pot_d[0] = sum_i closest_i (adds over shards);
 *   dcv_kmeanspp_potentials  pot_d[t] = sum_i min(closest_i, dist(x_i, cand_t)) for `trials` <= 8 candidate centres
 *                            (trials x d, already in the offset frame). */
size_t dcv_kmeanspp_workspace(int64_t n, int32_t trials);
int dcv_kmeanspp_potentials(const double* P_d, int64_t n, int32_t d, const double* offset_d, const double* cand_d, int32_t trials,
                            const double* closest_d, double* pot_d, void* ws_d, size_t ws_bytes, void* stream);
int dcv_kmeanspp_update(const double* P_d, int64_t n, int32_t d, const double* offset_d, const double* centre_d, int32_t first,
                        double* closest_d, double* pot_d, void* ws_d, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- k-selection scores
 * Replace sklearn.metrics calinski_harabasz_score / davies_bouldin_score / silhouette_score as called
 * by statistics.optimize_clustering, statistics.py:73-75, on points P (n x d float64, d <= 16) and
 * int32 labels in [0, k), k <= 64 (other labels are ignored).
 * dcv_label_stats: acc_d = [sums k*d | counts k | sum ||x - c_label||^2 k | sum ||x - c_label|| k]
 * (float64; the last two groups about centers_d (k x d) when given, zero otherwise) -- two calls give
 * the label means and then the dispersions both scores are built from; every group combines over
 * shards by addition.
 * dcv_cluster_dist_sums: S[i][c] = sum over the points j of cluster c of ||Q_i - P_j|| for nq query
 * points against the points sorted by cluster (cluster c = rows [start[c], start[c+1]) of Psorted_d;
 * start_d holds k + 1 int64).  The exact all-pairs silhouette: O(nq * n * d) float64.
 * dcv_silhouette_sum: sum_d[0] = sum of the silhouette sample values of the nq queries (labels
 * qlabels_d), sklearn's definition incl. 0 for singleton clusters; divide by n for the score. */
size_t dcv_label_stats_workspace(int64_t n, int32_t d, int32_t k);
int dcv_label_stats(const double* P_d, int64_t n, int32_t d, const int32_t* labels_d, const double* centers_d /* k*d or NULL */,
                    int32_t k, double* acc_d, void* ws_d, size_t ws_bytes, void* stream);
int dcv_cluster_dist_sums(const double* Q_d, int64_t nq, const double* Psorted_d, const int64_t* start_d, int32_t k, int32_t d,
                          double* S_d /* nq x k */, void* stream);
size_t dcv_silhouette_sum_workspace(int64_t nq);
int dcv_silhouette_sum(const double* S_d, int64_t nq, int32_t k, const int32_t* qlabels_d, const int64_t* start_d,
                       double* sum_d, void* ws_d, size_t ws_bytes, void* stream);

/* Free-energy surface of the projected trajectory (second half).  The reference plots
 * mlcolvar.utils.fes.compute_fes(data, backend="KDEpy", num_samples=num_bins, bandwidth, blocks, bounds) (figures.py:95-98):
 * KDEpy's FFTKDE = linear binning of the points onto the num_bins^d grid, convolution of the grid with the kernel,
 * FES = -kB T log(density + eps).  The streaming stage is the binning: each of the n points (row-major float64, row
 * stride ldp, the d <= 2 columns cols_h) spreads unit weight over the 2^d grid nodes around it.  grid_d receives the
 * bins^d node weights (sum = points inside the bounds; d = 2: grid[i][j], i along the first column); outside_h the
 * number of points outside [lo, hi] (ignored).  Deterministic (64-bit fixed-point accumulation).  The d-dimensional
 * convolution on the small grid and the logarithm stay with the caller (statistics.compute_fes). */
size_t dcv_linear_binning_workspace(int32_t d, int32_t bins);
int dcv_linear_binning(const double* P_d, int64_t n, int64_t ldp, int32_t d, const int32_t* cols_h, const double* lo_h,
                       const double* hi_h, int32_t bins, double* grid_d, int64_t* outside_h, void* ws_d, size_t ws_bytes, void* stream);

/* Replaces statistics.find_centroids, statistics.py:370-377: for each of the k centroids the
 * index of the nearest of ALL n points under np.linalg.norm (ties -> lowest index).
 * best_d receives k pairs [distance (float64) | row (float64-encoded int64 is avoided:
 * rows_d int64, dist_d float64)] that combine over shards by lexicographic min. */
size_t dcv_nearest_rows_workspace(int64_t n, int32_t d, int32_t k);
int dcv_nearest_rows(const double* P_d, int64_t n, int32_t d, const double* centers_d, int32_t k,
                     int64_t row_offset, double* dist_d, int64_t* rows_d,
                     void* ws_d, size_t ws_bytes, void* stream);

/* Replaces TrajClusterWorkflow.assign_closest_cluster, traj_cluster_workflow.py:207-238:
 * nn_d[i] = index of the training point nearest to supplementary point i (squared
 * Euclidean distance, ties -> lowest index). */
int dcv_nearest_point(const double* train_d, int64_t n_train, const double* sup_d, int64_t n_sup,
                      int32_t d, int64_t* nn_d, void* stream);

/* ---------------------------------------------------------------- building block
 * C[M,N] = op(A).op(B) on the FP32 MFMA engine that the covariance and MLP kernels are built
 * from (v_mfma_f32_32x32x2_f32, exact f32 products, f32 accumulation).  mode 0 (NT):
 * A[M,K] . B[N,K]^T; 1 (NN): A[M,K] . B[K,N]; 2 (TN): A[K,M]^T . B[K,N].  Exposed so that the
 * parity tests can exercise every operand form / tile shape / ragged edge directly. */
int dcv_gemm_f32(int32_t mode, const float* A_d, int64_t lda, const float* B_d, int64_t ldb, float* C_d,
                 int64_t ldc, int64_t M, int64_t N, int64_t K, void* stream);
/* Split-K form of mode 2 as the weight-gradient / covariance paths use it: the K rows are cut into chunks of
 * k_chunk rows, chunk z writes its partial product to slab_d[z] (M x N floats each).  slab_cap is the number of
 * slabs the buffer holds; a launch that needs more returns DCV_ENOMEM without touching memory. */
int dcv_gemm_tn_split(const float* A_d, int64_t lda, const float* B_d, int64_t ldb, float* slab_d, int64_t slab_cap,
                      int64_t M, int64_t N, int64_t K, int64_t k_chunk, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCV_H */
