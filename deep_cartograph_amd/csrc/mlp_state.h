// State of the MLP engine (dcv_mlp) shared by the translation units that implement its steps.
#pragma once
#include "gemm_kernels.h"
#include <vector>

namespace dcv {

constexpr int kMaxTicaDim = 16;

struct LayerPlan {
    int in, out, act;
    int64_t w_off, b_off;      // offsets into the flat parameter buffer (floats, 16-byte aligned)
    int64_t ldh;               // row stride of the activation buffer
    float* H;                  // [rows][ldh] post-activation output
    // wgrad split-K
    int64_t k_chunk_cap;       // rows per split at full capacity
    int64_t max_splits;
    float* slab;               // [max_splits][out][in]
    float* bpart;              // [bias_blocks_cap][out]
    unsigned long long* mask;  // sign mask of H in the forward epilogue's thread layout (ReLU family), or null
    int64_t mask_rows;         // row count of the training forward that wrote it (-1: stale)
    // batch normalisation behind this Linear (bn.hip), or bn == 0
    int bn;
    int64_t g_off, be_off;     // weight / bias of the normalisation in the flat parameter buffer
    float* Y;                  // [rows][ldh] normalised output (input of the next layer); H keeps the values it was computed from
    float *rm, *rv;            // running mean / variance [out]
    double* bn_stat;           // [2 forward calls][mean | invstd][out] of the last training forward
    double* bn_part;           // statistics partials [blocks][2][out]
    float *bn_gpart, *bn_bpart;   // gradient partials of weight / bias [blocks][out]
    int64_t bn_batches;        // num_batches_tracked
};

}  // namespace dcv

struct dcv_mlp {
    dcv_mlp_desc desc;
    int L;
    int d_out;                 // dims[L]
    int64_t rows_cap;          // rows per step at max_batch
    int64_t n_params;
    std::vector<dcv::LayerPlan> layers;
    float *params, *grads, *adam_m, *adam_v;
    float* opt_aux;            // third optimiser state (amsgrad maximum / centred RMSprop gradient average) or null
    double momentum_rt;        // beta1 (Adam family) or momentum (SGD, RMSprop): dcv_mlp_set_momentum
    double nadam_mu_product;   // NAdam: the float32 state mu_product, as torch reads it back
    double asgd_eta;           // ASGD: the float32 state eta
    bool any_drop;             // some layer has dropout p > 0
    bool any_bn;               // some layer is followed by a batch normalisation
    bool fwd_train;            // the last forward ran in training mode (dropout active): backward must agree
    dcv::TailWs tail;          // workspace of the contraction-split tail tile of row-tiled products (gemm.h: GemmDims::tail_split)
    void (*upper_cb)(void*);   // data-parallel overlap hook (dcv_mlp_set_upper_grads_callback) or null
    void* upper_cb_user;
    bool head_done;            // the last forward already ran the d x d loss head inside its statistics launch (one-GPU steps)
    int64_t drop_step;         // training forwards so far = step field of the next forward's dropout counters
    int64_t cur_step;          // step field of the last training forward
    uint32_t drop_rank;        // mixed into the key of the dropout counters (dcv_mlp_set_rank): the ranks of a data-parallel run draw independent masks
    void* snet;                // plan of the fused small-network step (snet.hip) or null
    bool snet_tried;           // the plan was attempted once (null afterwards = not applicable)
    float* snet_img;           // zero-padded LDS image of every weight / bias of the fused small-network kernels, kept current by the
    int* snet_img_idx;         //   optimiser (img[img_idx[i]] mirrors params[i]; -1: not part of the image); null until a plan builds it
    int snet_img_floats;
    void* snet_dt;             // plan of the fused small-network Deep-TICA forward / backward (snet_dt.hip) or null
    bool snet_dt_tried;
    bool snet_fwd_valid;       // the last forward went through snet_dt_forward and left its blob for the backward
    int last_path;             // path of the last step / forward: 0 layer by layer, 1 fused autoencoder step, 2 fused Deep-TICA kernels
    float* dZ[2];
    int64_t ld_dz;
    double* stats;             // device
    int stats_len;
    double* gradp;             // Deep-TICA: [mu d | Gu d*d | Gv d*d | c d], float64 (see tica_dF_kernel)
    double* spart;             // stats partials
    int spart_blocks;
    double* log;
    int* log_count;
    unsigned* ticket;          // block counter of the single-launch statistics reduction (zero between launches)
    hipGraphExec_t gexec[4];   // instantiated step graphs (train step, forward, backward, eval step) or null
    bool gwarm[4];             // the slot ran once outside capture (lazy module loading, first-use attributes)
    bool graph_on;             // step graphs requested (dcv_mlp_set_graph; default from DCV_GRAPH=1)
    bool graph_off;            // graph instantiation failed once: plain launches from then on
    int64_t graph_launches;    // steps / half-steps that went out as one graph launch
    bool prof_paused;          // profiling armed but skipped for the current calls (dcv_mlp_profile_pause)
    int log_cap, log_width;
    float* feat_range;         // AE
    float *ident, *zeros_d, *ones_d;  // helpers for inference
    float* proj_ws;
    size_t proj_ws_bytes;
    int64_t adam_t;
    double lr;
    // bookkeeping of the last forward (backward must match)
    int32_t last_batch;
    int no_row_sharing;        // diagnostic: evaluate contiguous Deep-TICA batches as two separate halves
    // optional per-kernel timing with HIP events on the launch stream (bench.py roofline)
    int prof_level, prof_cap;
    int prof_kind_off;                // kinds (bit 0 fwd, 1 wgrad, 2 dgrad) whose launches are not sampled right now
    std::vector<int> prof_cnt;        // samples taken per class
    std::vector<hipEvent_t> prof_ev;  // [class][step][2], class = 3*layer + {0 fwd, 1 wgrad, 2 dgrad}
};


namespace dcv {
// gradient partials of a fused small-network step, as the split-K reduction wants them (per layer: [splits][out * in] and
// [bblocks][out])
struct ReduceArgsView {
    const float* slab[DCV_MAX_LAYERS];
    const float* bpart[DCV_MAX_LAYERS];
    int splits[DCV_MAX_LAYERS];
    int bblocks[DCV_MAX_LAYERS];
    int64_t wstride[DCV_MAX_LAYERS], bstride[DCV_MAX_LAYERS];   // floats between consecutive partials (multiples of 4)
};
// a batched validation pass (dcv_mlp_eval_steps) of a fused small-network engine: at most this many batches / workgroups per launch
// (the plans allocate their per-batch tickets and per-workgroup partials for these bounds when they are built: no allocation
// lands in a timed validation pass)
constexpr int kEvalBatchesPerLaunch = 64;
constexpr int64_t kEvalWorkgroupsPerLaunch = 4096;
// snet.hip: the whole autoencoder step in one launch when the network fits in LDS; 1 = not applicable
// R rows of this rank, `batch` = the GLOBAL batch (loss scale 2 / (batch * F)); write_log = false: the caller logs (after an all-reduce)
int snet_ae_step(dcv_mlp* m, const float* Xn_d, int64_t ld, const RowMap& rm, int64_t R, int64_t batch, int train, ReduceArgsView* ra,
                 hipStream_t s, bool write_log = true, int nb = 1);   // nb > 1: that many evaluation batches of R rows in one launch
int snet_ae_tile_rows(dcv_mlp* m, int64_t R = 0);   // rows per workgroup of the fused kernel for batches of R rows (builds the plan on first use); 0: not applicable
void snet_free(dcv_mlp* m);
// the weight image both fused small-network plans stage from (snet.hip); repack: after the parameters were written by anyone
// but the optimiser (dcv_mlp_set_params)
bool snet_image_build(dcv_mlp* m);
int snet_image_repack(dcv_mlp* m, hipStream_t s);
void snet_image_free(dcv_mlp* m);
// snet_dt.hip: Deep-TICA forward (+ statistics, + loss head) and backward of a network that fits in LDS; 1 = not applicable
// nb > 1 (head == 2, no blob): that many evaluation batches of `batch` pairs in one launch, batch j = the pairs [j * batch, (j + 1) * batch)
int snet_dt_forward(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0, int32_t batch, int head, bool keep_blob,
                    hipStream_t s, int nb = 1);
int snet_dt_backward(dcv_mlp* m, int32_t batch, int64_t global_batch, bool head, ReduceArgsView* ra, hipStream_t s);
void snet_dt_free(dcv_mlp* m);
// bn.hip
int bn_forward(dcv_mlp* m, int l, int64_t row0, int64_t rows, bool train, hipStream_t s);
int bn_backward(dcv_mlp* m, int l, float* dz, int64_t ld_dz, int halves, int64_t rows_half, int act, float hscale, const DropCfg& drop,
                int* blocks_out, hipStream_t s);
int bn_eval_backward(dcv_mlp* m, int l, float* dz, int64_t ld_dz, int64_t rows, int act, hipStream_t s);
}  // namespace dcv
