/*
 * rechorus_hip.h -- C ABI of librechorus_hip.so, the MI355X (gfx950) engine for the
 * ReChorus ranking hot path:  embedding gather (user, pos, neg[K]) -> interaction head
 * -> BPR loss -> embedding-gradient segmented scatter + optimizer row update.
 *
 * The reference (THUwangcy/ReChorus) has no FFI: every operation below is an *implicit*
 * ATen kernel group dispatched from Python.  Each entry point cites the reference lines
 * whose arithmetic it replaces (paths relative to the reference's src/).
 *
 * Conventions
 *  - every pointer except `const char*` results and `phase_ms` is a DEVICE pointer
 *    (a torch tensor's data_ptr()); buffers are caller-owned, nothing is allocated here;
 *  - tables are row-major fp32 [n_rows, d]; ids are int64 (the reference layout,
 *    models/BaseModel.py:198) and must lie in [0, n_rows);  ids are NOT range-checked
 *    on the device (the reference asserts this on the host, helpers/BaseReader.py:59);
 *  - kernels are enqueued on `stream` (a hipStream_t passed as void*) and never
 *    synchronise, unless a `phase_ms` host pointer is given (profiling mode);
 *  - every function returns RC_OK (0) or a negative rc_status; the text of the last
 *    failure on the calling thread is rc_last_error_string();
 *  - scratch comes from the caller: rc_*_workspace_bytes() gives the size.
 */
#ifndef RECHORUS_HIP_H
#define RECHORUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rc_stream_t; /* hipStream_t */

enum rc_status {
  RC_OK = 0,
  RC_ERR_INVALID_ARG = -1, /* bad shape / null pointer / unsupported combination      */
  RC_ERR_WORKSPACE = -2,   /* workspace too small                                    */
  RC_ERR_HIP = -3,         /* a HIP runtime call or kernel launch failed             */
  RC_ERR_UNSUPPORTED = -4  /* valid request the engine has no kernel for             */
};

/* optimizers of helpers/BaseRunner.py:110-114 (eval('torch.optim.<name>')) */
enum rc_opt {
  RC_OPT_SGD = 0,     /* torch.optim.SGD, no momentum                                 */
  RC_OPT_ADAM = 1,    /* torch.optim.Adam (amsgrad off)                               */
  RC_OPT_ADAGRAD = 2, /* torch.optim.Adagrad (lr_decay 0, initial accumulator 0)      */
  RC_OPT_ADADELTA = 3 /* torch.optim.Adadelta (rho = beta1, eps 1e-6; m = square_avg, v = acc_delta):
                         dense steps only (rc_dense_update*), the row-wise entry points reject it */
};

/* hyper-parameters of one optimizer step.  `step` is the 1-based step count t used for
 * Adam's bias correction (1 - beta^t); weight decay `l2` is added to the gradient
 * (g += l2 * w) exactly as torch does for weight_decay (helpers/BaseRunner.py:112-113). */
typedef struct rc_opt_hyper {
  int opt;      /* enum rc_opt */
  int reserved; /* 0 */
  double lr;    /* doubles, like the Python floats torch.optim receives; the engine     */
  double l2;    /* narrows the derived scalars to fp32 at the same points torch does    */
  double beta1; /* Adam */
  double beta2; /* Adam */
  double eps;   /* Adam 1e-8, Adagrad 1e-10 */
  int64_t step; /* t >= 1 */
} rc_opt_hyper;

/* ---- library ------------------------------------------------------------------- */
int rc_version(void);                    /* ABI version, currently 1                  */
const char* rc_last_error_string(void);  /* thread-local text of the last failure     */
int rc_device_count(void);               /* >= 0, or a negative rc_status             */

/* ---- forward ------------------------------------------------------------------- */

/* nn.Embedding forward, out[i,:] = W[ids[i],:]      (models/general/BPRMF.py:39-40) */
int rc_gather_rows(const float* W, int d, const int64_t* ids, int64_t n, float* out,
                   rc_stream_t stream);
/* the same for two tables that share the ids, rows side by side: out[i, :] = (Wa[ids[i], :] | Wb[ids[i], :]), out [n, 2 d] -- NeuMF's
 * mf / mlp tables (models/general/NeuMF.py:39-42,61-66) as the block a row-sharded rank serves for the ids it owns; d % 4 == 0.      */
int rc_gather_rows_pair(const float* Wa, const float* Wb, int d, const int64_t* ids, int64_t n, float* out, rc_stream_t stream);

/* BPRMF scores, pred[b,c] = <U[uid[b]], I[iid[b,c]]>  (models/general/BPRMF.py:39-42).
 * Any C >= 1 (eval uses C = 100 or C = n_items-1 with --test_all).                  */
int rc_gather_dot_fwd(const float* U, const float* I, const int64_t* uid,
                      const int64_t* iid, int B, int C, int d, float* pred,
                      rc_stream_t stream);

/* out[b,:] = sum_c coef[b,c] * W[ids[b,c],:] -- the user-row half of autograd's backward of
 * BPRMF.py:42 (d pred / d u_vectors), for callers that get dL/dpred from an arbitrary loss.   */
int rc_weighted_row_sum(const float* W, const int64_t* ids, const float* coef, int B, int C,
                        int d, float* out, rc_stream_t stream);

/* ---- loss ---------------------------------------------------------------------- */

/* GeneralModel.loss (models/BaseModel.py:182-185): softmax-weighted multi-negative BPR.
 * pred [B,C] (column 0 = positive, C >= 2).  Writes the per-row loss
 *   loss_vec[b] = -log(clamp(sum_k softmax(neg)_k * sigmoid(pos - neg_k), 1e-8, 1-1e-8))
 * and, if gpred != NULL, gpred[b,c] = d(inv_b * sum_b loss_vec[b]) / d pred[b,c]
 * (inv_b = 1/B gives the reference's .mean()).                                       */
int rc_bpr_loss_fwd_bwd(const float* pred, int B, int C, float inv_b, float* loss_vec,
                        float* gpred, rc_stream_t stream);

/* Factorization-machine second-order term over stacked field vectors V [n, F, d]
 * (FMBase.forward / DeepFMBase.forward, models/context/FM.py:59-63, DeepFM.py:19-23):
 *   out[i] = sum_k 0.5*((sum_f V[i,f,k])^2 - sum_f V[i,f,k]^2);   dV[i,f,k] = gout[i]*((sum_f' V[i,f',k]) - V[i,f,k]).
 * d in {16,32,64,128}.                                                                              */
int rc_fm_second_order_fwd(const float* V, int64_t n, int F, int d, float* out, rc_stream_t stream);
int rc_fm_second_order_bwd(const float* V, const float* gout, int64_t n, int F, int d, float* dV,
                           rc_stream_t stream);
/* dV = add + d fm2 / dV in one pass: the field vectors' gradient through the FM term on top of their gradient through the deep
 * tower (models/context/DeepFM.py:19-28 feeds the same stacked vectors to both); add [n, F, d] may be dV itself. */
int rc_fm_second_order_bwd_add(const float* V, const float* gout, int64_t n, int F, int d, const float* add, float* dV,
                               rc_stream_t stream);

/* All F categorical field lookups of a context model in one launch (FMBase._get_embeddings_FM,
 * models/context/FM.py:44-57): tables / ids / per_row / row_offset are HOST arrays of length F holding device
 * pointers (tables[f]: [vocab_f, d] fp32; ids[f]: int64 [B] if per_row[f] else [B, C]) and per-field
 * constants.  out [B, C, F, d] = the stacked field vectors (per-row fields broadcast over the C candidates);
 * cid [B, C, F] (optional) = row_offset[f] + id, the composite row index of a virtual table that
 * concatenates all fields -- the sort key for ONE rc_sort_ids + rc_segmented_update(dense_grad) pass that
 * yields every field's dense gradient.  F <= 48.                                                          */
int rc_gather_fields(const float* const* tables, const int64_t* const* ids, const int* per_row,
                     const int64_t* row_offset, int F, int64_t B, int C, int d, float* out, int64_t* cid,
                     rc_stream_t stream);
/* rc_gather_fields for the TWO table families the FM models gather with the same ids (models/context/FM.py:44-57: the [vocab, d]
 * field vectors and the [vocab, 1] first-order weights): out [B, C, F, d] and out1 [B, C, F] in one launch.                      */
int rc_gather_fields_pair(const float* const* tables, const float* const* tables1, const int64_t* const* ids, const int* per_row,
                          const int64_t* row_offset, int F, int64_t B, int C, int d, float* out, float* out1, int64_t* cid,
                          rc_stream_t stream);
/* The same, and every composite row the batch looks up is stamped row_flags[row_offset[f] + id] = (int32) (step_dev[0] + step_add):
 * the rows of THIS step for rc_dense_update_rows_dev (row_flags [sum of vocab sizes], never reset: a stamp of an earlier step is
 * not equal to the current one; step_add = 1 when the optimizer increments its count after the backward pass).                    */
int rc_gather_fields_pair_mark(const float* const* tables, const float* const* tables1, const int64_t* const* ids,
                               const int* per_row, const int64_t* row_offset, int F, int64_t B, int C, int d, float* out,
                               float* out1, int64_t* cid, int32_t* row_flags, const int64_t* step_dev, int step_add,
                               rc_stream_t stream);

/* Field kinds of rc_gather_fields_mixed.  A numeric context feature (a name ending neither '_c' nor '_id', e.g. MIND's c_day_f) is
 * nn.Linear(1, d, bias=False) / nn.Linear(1, 1, bias=False) applied to feed_dict[f].float().unsqueeze(-1)
 * (models/context/FM.py:38-41,47-48,51-52): the value type says how the feature arrives in the feed dict.                          */
enum rc_field_kind { RC_FIELD_IDS = 0, RC_FIELD_F32 = 1, RC_FIELD_F64 = 2, RC_FIELD_I64 = 3 };
/* rc_gather_fields_pair(_mark) for a field list that holds numeric features (FMBase._get_embeddings_FM, models/context/FM.py:44-57,
 * both branches of its conditional expressions): kind[f] = RC_FIELD_IDS as before; otherwise ids[f] points at the feature's VALUES
 * (float / double / int64, [B] if per_row[f] else [B, C]), tables[f] at the d weights of context_embedding[f] ([d, 1] contiguous),
 * tables1[f] at the one weight of linear_embedding[f]: out[b, c, f, :] = x * W[:, 0], out1[b, c, f] = x * w1.  Numeric fields own no
 * row of the virtual concatenated table (row_offset[f] is ignored, no flag is stamped): their occurrences carry numeric_key in cid
 * -- -1 for the groupings that skip negative keys (rc_small_row_sums, rc_bucket_plan), or the total row count, which sorts them
 * behind every real row (rc_sort_ids).  tables1 / out1, cid and row_flags / step_dev are optional (NULL) as in the calls above.     */
int rc_gather_fields_mixed(const float* const* tables, const float* const* tables1, const void* const* ids, const int* per_row,
                           const int* kind, int64_t numeric_key, const int64_t* row_offset, int F, int64_t B, int C, int d, float* out,
                           float* out1, int64_t* cid, int32_t* row_flags, const int64_t* step_dev, int step_add, rc_stream_t stream);
/* rc_gather_fields_mixed with what the rest of a SMALL training step needs from the same ids, in the same launch (each optional):
 *   fm_out [B * C], fm_sum [B * C, d]  the FM pairwise term of every row (models/context/FM.py:61; rc_fm_second_order_fwd's value bit
 *                                      for bit) and the field sum sum_f out[r, f, :] its backward needs;
 *   plan_ws                            the grouping of the composite (field, id) keys that aten::embedding_dense_backward's sort
 *                                      does (rc_small_row_sums' first launch), by 128 workgroups beside the gather's, left where
 *                                      rc_small_row_sums_planned reads it (rc_small_row_sums_workspace_bytes(B * C * F) bytes;
 *                                      B * C * F <= 32,768 keys).
 *   bump                               a device counter that one thread of the launch increments (NULL: none) -- the dropout seed
 *                                      of the deep tower that runs next (utils/layers.py:201-243's nn.Dropout sites), which would
 *                                      otherwise be a one-thread launch of its own; nothing in this launch may read it (not step_dev).
 * d in {16, 32, 64, 128}; kind may be NULL (every field a table); the other arguments as rc_gather_fields_mixed.                     */
int rc_gather_fields_fused(const float* const* tables, const float* const* tables1, const void* const* ids, const int* per_row,
                           const int* kind, int64_t numeric_key, const int64_t* row_offset, int F, int64_t B, int C, int d, float* out,
                           float* out1, int64_t* cid, int32_t* row_flags, const int64_t* step_dev, int step_add, float* fm_out,
                           float* fm_sum, void* plan_ws, size_t plan_ws_bytes, int64_t* bump, rc_stream_t stream);
/* Weight gradients of the numeric fields (autograd's Linear backward behind loss.backward(), helpers/BaseRunner.py:205, for the
 * modules of models/context/FM.py:38-41): dW[j][k] = sum_n x_j[n] * gV[n, field[j], k] and dw1[j][0] = sum_n x_j[n] * gL[n, field[j]]
 * over the n = B * C rows of the per-occurrence gradient blocks gV [n, F, d] / gL [n, F] (either may be NULL with its outputs).
 * values / per_row / kind / field / dW / dw1 are HOST arrays of length n_numeric (device pointers / constants per numeric field).
 * Fixed summation order, no atomics; ws: rc_numeric_field_grads_workspace_bytes (unused up to 1,024 rows).                          */
size_t rc_numeric_field_grads_workspace_bytes(int64_t n, int n_numeric, int d);
int rc_numeric_field_grads(const float* gV, const float* gL, const void* const* values, const int* per_row, const int* kind,
                           const int* field, int n_numeric, int F, int64_t B, int C, int d, float* const* dW, float* const* dw1,
                           void* ws, size_t ws_bytes, rc_stream_t stream);

/* nn.BCELoss on probabilities (CTRModel.loss, models/BaseModel.py:259-267), torch's log clamp (-100) and
 * backward denominator clamp (1e-12): loss_vec[i] = -(y log p + (1-y) log(1-p)); gp[i] = dmean/dp_i
 * with inv_n = 1/n.  The loss is rc_reduce_sum(loss_vec, n, inv_n).                                    */
int rc_bce_prob_fwd_bwd(const float* p, const float* y, int64_t n, float inv_n, float* loss_vec, float* gp,
                        rc_stream_t stream);

/* List-wise softmax cross-entropy over an impression list (ImpressionModel.loss, loss_n='softmaxCE',
 * models/BaseImpressionModel.py:96-107).  pred [B,n]; target [B,n] int64 in {1,0,-1} (-1 = padding),
 * the first max_pos columns are the positive slots.  loss_vec[b] is row b's contribution (their sum
 * is the loss: rc_reduce_sum(loss_vec, B, 1)); h_sum[0] receives sum_b [row b has a negative];
 * gpred (optional) = d loss / d pred.                                                             */
int rc_softmax_ce_fwd_bwd(const float* pred, const int64_t* target, int B, int n, int max_pos,
                          float* loss_vec, float* h_sum, float* gpred, rc_stream_t stream);

/* Point-wise BCE over a ranking list (ContextModel.loss with loss_n 'BCE', models/BaseContextModel.py:53-56):
 * loss_vec[b] = -(log sigmoid(pred[b,0]) + sum_{k>=1} log(1 - sigmoid(pred[b,k]))); the loss is their mean
 * (rc_reduce_sum(loss_vec, B, inv_b)); gpred (optional) = d(mean)/d pred.                                  */
int rc_bce_ranking_fwd_bwd(const float* pred, int64_t B, int C, float inv_b, float* loss_vec, float* gpred,
                           rc_stream_t stream);

/* List-level BPR over an impression list (ImpressionModel.loss with loss_n 'BPR' or 'BPRhard' -- the default
 * loss of every *Impression model, models/BaseImpressionModel.py:50-89): per row
 *   Q = sum_{i pos} a_i sum_{j neg} b_j sigmoid(pred_i - pred_j),  a / b = softmax weights over the valid
 *   positives (of -pred when hard != 0) / negatives;  loss_vec[b] = -log Q_b (the loss is their mean:
 *   rc_reduce_sum(loss_vec, B, inv_b));  gpred (optional) = d(inv_b * sum_b loss_vec[b]) / d pred, through
 *   the softmax weights as well (autograd semantics).  Layout and target as in rc_softmax_ce_fwd_bwd.     */
int rc_list_bpr_fwd_bwd(const float* pred, const int64_t* target, int B, int n, int max_pos, int hard,
                        float inv_b, float* loss_vec, float* gpred, rc_stream_t stream);

/* Every list-wise loss name of ImpressionModel.loss (models/BaseImpressionModel.py:44-129) behind one entry point
 * ('BPR...simple', which the reference returns unreduced, excepted).  kind: */
enum rc_list_kind {
  RC_LIST_BPR = 0,             /* :82-86  re-weighting between the sigmoid and the log (rc_list_bpr_fwd_bwd)        */
  RC_LIST_BPR_HARD = 1,        /* :68-70  the same, lower-scored positives weigh more                              */
  RC_LIST_BPR_AFTER = 2,       /* :76-78  softplus(-(s_i - s_j)) weighted by both softmaxes                          */
  RC_LIST_BPR_HARD_AFTER = 3,
  RC_LIST_BPR_BEFORE = 4,      /* :79-81  softplus of the weighted score difference, summed over all n columns      */
  RC_LIST_BPR_HARD_BEFORE = 5,
  RC_LIST_LISTNET = 6,         /* :84-94  cross entropy between softmax(labels) and softmax(scores)                 */
  RC_LIST_SOFTMAX_CE = 7,      /* :96-107 (rc_softmax_ce_fwd_bwd)                                                   */
  RC_LIST_ATTENTION_RANK = 8,  /* :109-126 listnet + the (1 - t) log(1 - p) term                                    */
  RC_LIST_BPR_SIMPLE = 9       /* :82-83  every valid (pos, neg) pair, no re-weighting; the per-row sums are the result
                                  (unreduced, as the reference returns them): gpred = inv_b * d row / d pred            */
};
/* loss_vec [B]: per-row terms whose fixed-order sum (rc_reduce_sum; scale 1/B for the BPR kinds, which also take
 * inv_b = 1/B for the gradient; scale 1 for the kinds normalised by the rows that have a negative) is the loss;
 * h_sum [1]: scratch of those kinds (may be NULL for the BPR kinds); gpred [B, n] (optional) = dloss/dpred.           */
int rc_list_loss_fwd_bwd(const float* pred, const int64_t* target, int B, int n, int max_pos, int kind, float inv_b,
                         float* loss_vec, float* h_sum, float* gpred, rc_stream_t stream);

/* out[0] = scale * sum_i x[i], fixed summation order (deterministic).  Used for the
 * batch mean of loss_vec (models/BaseModel.py:185 `.mean()`).                         */
int rc_reduce_sum(const float* x, int64_t n, float scale, float* out, rc_stream_t stream);

/* ---- fused BPRMF forward + loss + backward-to-rows ------------------------------- */

/* One pass over the (1+K) candidate rows of every tuple, rows held in registers:
 *   pred (optional, may be NULL), loss_vec[B], gpred[B,C] = dL/dpred,
 *   ugrad[B,d] = sum_c gpred[b,c] * I[iid[b,c]]   (the per-tuple user-row gradient).
 * Replaces BPRMF.forward + GeneralModel.loss + the MulBackward/SumBackward half of
 * autograd (models/general/BPRMF.py:34-45, models/BaseModel.py:182-185).  The item-row
 * gradients g[b,c]*U[uid[b]] are NOT materialised; rc_segmented_update rebuilds them. */
int rc_bprmf_fwd_bwd(const float* U, const float* I, const int64_t* uid,
                     const int64_t* iid, int B, int C, int d, float inv_b, float* pred,
                     float* loss_vec, float* gpred, float* ugrad, rc_stream_t stream);

/* 1 if (d, C) has a register-resident fused kernel (d in {16,32,64,128}, C <= 32*256/d),
 * i.e. rc_bprmf_fwd_bwd_update is available; rc_bprmf_fwd_bwd always has the LDS fall-back. */
int rc_bprmf_fused_supported(int d, int C);

/* rc_bprmf_fwd_bwd that ALSO applies the optimizer `h` to every item row occurring exactly
 * once in the batch (single[b*C+c] != 0, from rc_segment_heads): such a row is read by no
 * other tuple, its whole gradient g[b,c]*U[uid[b]] is known while the row is still in
 * registers, so it is updated and written back here -- one HBM read and one write per step.
 * Rows with several occurrences are left for rc_segmented_update(RC_SEG_SKIP_SINGLETONS).
 * I (and mI, vI) are updated in place; the arithmetic is the one of rc_segmented_update.   */
int rc_bprmf_fwd_bwd_update(const float* U, float* I, float* mI, float* vI,
                            const int64_t* uid, const int64_t* iid, const uint8_t* single,
                            int B, int C, int d, float inv_b, const rc_opt_hyper* h,
                            float* pred, float* loss_vec, float* gpred, float* ugrad,
                            rc_stream_t stream);

/* The same with the singleton information as a bitmap over item ids (rc_bucket_multi_bitmap / the bucket plan of
 * rc_bprmf_train_step): a row is updated here iff its bit is 0.  The kernel looks the tuple's candidates up in the
 * bitmap itself (L2-resident), so no per-position flag array crosses HBM.                                     */
int rc_bprmf_fwd_bwd_update_bitmap(const float* U, float* I, float* mI, float* vI,
                                   const int64_t* uid, const int64_t* iid, const uint32_t* multi,
                                   int B, int C, int d, float inv_b, const rc_opt_hyper* h,
                                   float* pred, float* loss_vec, float* gpred, float* ugrad,
                                   rc_stream_t stream);

/* ---- index sort (the atomic-free replacement of embedding_dense_backward's index_add) */

size_t rc_sort_workspace_bytes(int64_t n);

/* keys_out = ids sorted ascending (as uint32), perm_out = stable sorting permutation
 * (perm_out[j] = position in `ids` of the j-th smallest).  n_rows bounds the ids
 * (only ceil(log2 n_rows) key bits are sorted).  n < 2^31, n_rows <= 2^32.            */
int rc_sort_ids(const int64_t* ids, int64_t n, int64_t n_rows, uint32_t* keys_out,
                uint32_t* perm_out, void* ws, size_t ws_bytes, rc_stream_t stream);

/* Two id lists sorted in ONE call as the virtual concatenation [ids_a ; key_offset_b + ids_b]
 * (keys < key_range).  With key_offset_b >= every id of list a, the first n_a sorted positions are
 * list a's segment and the tail is list b's (perm values n_a .. n_a+n_b-1): a BPRMF step sorts its
 * item and user ids together and hands the two slices to rc_segmented_update2(key_base, occ_base). */
int rc_sort_ids2(const int64_t* ids_a, int64_t n_a, const int64_t* ids_b, int64_t n_b,
                 int64_t key_offset_b, int64_t key_range, uint32_t* keys_out, uint32_t* perm_out,
                 void* ws, size_t ws_bytes, rc_stream_t stream);

/* One pass over the sorted ids (keys/perm from rc_sort_ids) that
 *  - if single != NULL: single[o] = 1 iff occurrence o = perm[j] is the only one of its row;
 *  - if heads  != NULL: appends to heads[] the sorted positions j that start a segment
 *    (only_multi != 0: only segments with >= 2 occurrences) and counts them in n_heads[0]
 *    (device uint32, zeroed here).  The list order is unspecified (integer atomics); no
 *    floating-point result depends on it.  heads needs room for n_occ entries.            */
int rc_segment_heads(const uint32_t* keys, const uint32_t* perm, int64_t n_occ, int only_multi,
                     uint8_t* single, uint32_t* heads, uint32_t* n_heads, rc_stream_t stream);

/* ---- segmented gradient reduction + optimizer row update -------------------------- */

size_t rc_segmented_workspace_bytes(int64_t n_occ, int d);

/* For every distinct row r in sorted `keys` (with occurrences o = perm[j], j in the
 * segment of r, visited in ascending j => fixed summation order, no float atomics):
 *     grad_r = sum_o  coef[o] * Src[ srow(o), : ],
 *     srow(o) = src_index ? src_index[o / div] : o / div        (coef NULL => 1)
 * then, if `dense_grad` != NULL:  dense_grad[r,:] = grad_r   (torch-compatible dense
 * .grad; caller zero-fills it; replaces aten::embedding_dense_backward), else applies
 * the optimizer `h` to row r of W in place (and to rows r of the state tables m, v:
 * Adam exp_avg / exp_avg_sq, Adagrad sum in `m`), i.e. a row-wise ("lazy") version of
 * helpers/BaseRunner.py:206 that only touches rows present in the batch.
 * heads/n_heads: NULL, or the list made by rc_segment_heads (with only_multi set iff
 * RC_SEG_SKIP_SINGLETONS is given) -- saves re-deriving it.
 * flags: RC_SEG_SKIP_SINGLETONS leaves rows with exactly one occurrence untouched (they
 * were updated by rc_bprmf_fwd_bwd_update).
 * Src must not alias W.  ws from rc_segmented_workspace_bytes(n_occ, d).              */
enum rc_seg_flags { RC_SEG_SKIP_SINGLETONS = 1 };
int rc_segmented_update(float* W, float* m, float* v, int d, const uint32_t* keys,
                        const uint32_t* perm, int64_t n_occ, const float* coef,
                        const float* src, const int64_t* src_index, int div,
                        const rc_opt_hyper* h, float* dense_grad, const uint32_t* heads,
                        const uint32_t* n_heads, int flags, void* ws, size_t ws_bytes,
                        rc_stream_t stream);

/* Exact dense optimizer step over all n elements (torch.optim semantics incl. weight
 * decay on every element, helpers/BaseRunner.py:110-114,206).  m/v as above.         */
int rc_dense_update(float* W, const float* G, float* m, float* v, int64_t n,
                    const rc_opt_hyper* h, rc_stream_t stream);

/* The same step over n_tensors tensors in one launch per 36 tensors (a model's whole optimizer.step()):
 * W, G, m, v, n and h are HOST arrays of length n_tensors (m / v may be NULL for SGD; h[t] carries the
 * tensor's lr and weight decay -- 'bias' parameters have l2 = 0, models/BaseModel.py:64-73 -- all h[t].opt
 * equal).  Element arithmetic identical to rc_dense_update.                                              */
int rc_dense_update_multi(float* const* W, const float* const* G, float* const* m, float* const* v,
                          const int64_t* n, const rc_opt_hyper* h, int n_tensors, rc_stream_t stream);

/* hipGraph-capturable variant: for Adam the step count t is read from DEVICE memory (step_dev, int64) and the
 * bias corrections 1 - beta^t are derived in the kernel (h[t].step is ignored), so a captured training step
 * replays correctly; rc_step_increment bumps the counter on the stream.  step_dev NULL = rc_dense_update_multi. */
int rc_dense_update_multi_dev(float* const* W, const float* const* G, float* const* m, float* const* v,
                              const int64_t* n, const rc_opt_hyper* h, int n_tensors, const int64_t* step_dev,
                              rc_stream_t stream);
int rc_step_increment(int64_t* step_dev, rc_stream_t stream);
/* the same for two counters in one launch (a tower's dropout seed and Adam's step count of the same training step) */
int rc_step_increment2(int64_t* a_dev, int64_t* b_dev, rc_stream_t stream);
/* torch.optim.Adam over dense gradients of embedding tables whose batch touches few rows (helpers/BaseRunner.py:110-114,206 with
 * the nn.Embedding tables of models/context/FM.py:33-41 at batch_size 1024: 0.4 % of the rows have a gradient), WITHOUT a dense
 * gradient: tensor t is [n[t] / row_w[t], row_w[t]] with one int32 flag per row, and a row whose flag equals (int32) step_dev[0]
 * was looked up by this step's batch (rc_gather_fields_pair_mark); only those rows of G[t] hold (and are read for) a gradient, the
 * rest of G[t] is never read -- so nothing zero-fills it.  touched = 2: every row takes its Adam step, g = G's row where stamped
 * and 0 elsewhere (Adam still decays m, v and steps along m there): bit-identical to rc_dense_update_multi_dev on a zero-filled
 * dense gradient.  touched = 0 / 1 are the two halves of that pass -- the unstamped rows with g = 0 (G[t] may be NULL) / the
 * stamped rows from G[t] -- for a caller that wants them on different streams.  flags[t] == NULL makes tensor t a plain tensor
 * (all of it updated from G[t] whatever `touched`).  Adam only; step count in device memory, already incremented for this step.
 * max_blocks > 0 caps the grid (workgroups then walk the 4096-element chunks with the grid's stride).                          */
int rc_dense_update_rows_dev(float* const* W, const float* const* G, float* const* m, float* const* v, const int64_t* n,
                             const int32_t* const* flags, const int* row_w, const rc_opt_hyper* h, int n_tensors, int touched,
                             int max_blocks, const int64_t* step_dev, rc_stream_t stream);
/* dst = [a | b | c] (int64): the id tensors of a batch (feed_dict['history_items'], ['lengths'], ['item_id'],
 * models/sequential/SASRec.py:58-60) copied into the static buffer a captured step reads, in one launch. */
int rc_stage_batch(const int64_t* a, int64_t na, const int64_t* b, int64_t nb, const int64_t* c, int64_t nc, int64_t* dst,
                   rc_stream_t stream);

/* ---- whole BPRMF training step ----------------------------------------------------- */

/* rc_segmented_update with a SECOND gradient source: occurrences o >= n_split take the plain row
 * src2[o - n_split, :] instead of coef[o]*Src[srow(o)].  SASRec updates its item table from the
 * candidates (g[b,c] * encoder output, rebuilt on the fly) and from the history positions
 * (gradient rows written by the encoder backward) in ONE pass, so the optimizer sees each row once.
 * key_base / occ_base: keys and perm may be a SLICE of a joint sort (rc_sort_ids2): the table row
 * is keys[j] - key_base, the occurrence index perm[j] - occ_base (0, 0 for a plain sort).           */
int rc_segmented_update2(float* W, float* m, float* v, int d, const uint32_t* keys,
                         const uint32_t* perm, int64_t n_occ, const float* coef, const float* src,
                         const int64_t* src_index, int div, const float* src2, int64_t n_split,
                         int64_t key_base, int64_t occ_base, const rc_opt_hyper* h,
                         float* dense_grad, const uint32_t* heads, const uint32_t* n_heads, int flags,
                         void* ws, size_t ws_bytes, rc_stream_t stream);

/* rc_segmented_update2 for a table of n_rows rows that collect MANY occurrences each -- a small catalogue under a
 * large batch, the reference's own datasets (8.7 K items against 0.5 M candidate + history occurrences of one SASRec
 * step, models/sequential/SASRec.py:51-86 with helpers/BaseRunner.py:193-206): one pass over the sorted keys (plain
 * rc_sort_ids, keys < n_rows) records each row's [start, end), then ONE wave per table row sums its occurrences
 * (fixed order, no float atomics) and applies `h` / writes dense_grad; rows with more than 192 occurrences take the
 * 256-occurrence chunks of rc_segmented_update.  Gradient sources and outputs as there; d in {16, 32, 64, 128, 256},
 * buffers 16-byte aligned.  Worth it from about eight occurrences per table row.                               */
size_t rc_segmented_rows_workspace_bytes(int64_t n_rows, int64_t n_occ, int d);
int rc_segmented_update_rows(float* W, float* m, float* v, int d, int64_t n_rows, const uint32_t* keys,
                             const uint32_t* perm, int64_t n_occ, const float* coef, const float* src,
                             const int64_t* src_index, int div, const float* src2, int64_t n_split,
                             const rc_opt_hyper* h, float* dense_grad, void* ws, size_t ws_bytes, rc_stream_t stream);
/* The same with Adam's step count t read from device memory when the kernels run (step_dev[0] >= 1; h->step is not used; the
 * bias corrections 1 - beta^t are formed in the kernel with fill-in-double arithmetic like the host's): the launch can be
 * captured in a hipGraph and replayed while rc_step_increment advances the counter, as rc_dense_update_multi_dev does for
 * the dense parameters (torch.optim.Adam(capturable=True) semantics, helpers/BaseRunner.py:110-114,206).  SGD / Adagrad: as above. */
int rc_segmented_update_rows_dev(float* W, float* m, float* v, int d, int64_t n_rows, const uint32_t* keys,
                                 const uint32_t* perm, int64_t n_occ, const float* coef, const float* src,
                                 const int64_t* src_index, int div, const float* src2, int64_t n_split,
                                 const rc_opt_hyper* h, const int64_t* step_dev, float* dense_grad, void* ws, size_t ws_bytes,
                                 rc_stream_t stream);

/* rc_segmented_update_rows without the radix sort: the per-row [start, end) and the grouped occurrence list come from a counting
 * sort of the batch's own id tensors, as the reference hands them over (models/sequential/SASRec.py:69-81: feed_dict['item_id']
 * [B, 1 + K], feed_dict['history_items'] [B, history_max] with feed_dict['lengths']; the gradient of i_embeddings reaches both,
 * helpers/BaseRunner.py:205): occurrence o < n_a is ids_a[o], the others ids_b[o - n_a]; with lengths_b, ids_b is [n_b / L_b, L_b]
 * and slots l >= lengths_b[b] are padding -- they take no part, except the first of the batch, which keeps row 0 among the
 * touched rows (its gradient row is zero).  rc_rows_plan_build: three launches (tile histograms in LDS, prefix over the tiles,
 * scatter in occurrence order = the stable sort's order, so that every row sum is bit-identical to the sorted route; the hot
 * rows' chunk list is written on the way), no host synchronisation, capturable.  rc_rows_plan_update: rc_segmented_update_rows
 * on that plan in two launches (n_split = n_a; step_dev as rc_segmented_update_rows_dev, null: h->step).  n_rows <= 12,288;
 * ids outside [0, n_rows) take no part and are counted in status[0] (rc_rows_plan_views; the caller zeroes the workspace once).   */
int rc_rows_plan_supported(int64_t n_rows, int64_t n_occ, int d);
size_t rc_rows_plan_workspace_bytes(int64_t n_rows, int64_t n_occ, int d);
int rc_rows_plan_build(const int64_t* ids_a, int64_t n_a, const int64_t* ids_b, int64_t n_b, const int64_t* lengths_b, int L_b,
                       int64_t n_rows, int d, void* ws, size_t ws_bytes, rc_stream_t stream);
int rc_rows_plan_views(void* ws, int64_t n_rows, int64_t n_occ, int d, const uint32_t** keys, const uint32_t** perm,
                       const uint32_t** start, const uint32_t** end, const uint32_t** status);
int rc_rows_plan_update(float* W, float* m, float* v, int d, int64_t n_rows, int64_t n_occ, const float* coef, const float* src,
                        const int64_t* src_index, int div, const float* src2, int64_t n_split, const rc_opt_hyper* h,
                        const int64_t* step_dev, float* dense_grad, void* ws, size_t ws_bytes, rc_stream_t stream);

/* aten::embedding_dense_backward for a SMALL id list (n <= 32,768; helpers/BaseRunner.py:205 behind loss.backward() at the
 * reference's own batch sizes: 1,024 rows x 8 fields of a CTR step -- models/context/FM.py:49-52 --, 256 x (1 + K) candidates)
 * in TWO launches: 128 workgroups group the ids by row in LDS, one lane-group per touched row sums the gradient rows of its
 * occurrences in ascending position (no float atomics): out[ids[o], :] += ... i.e. out[r, :] = sum over o with ids[o] = r of
 * src[o, :] for every r that occurs; other rows of `out` are left as they are (the caller zero-fills).
 * d in {1..4, 16, 32, 64, 128}; src / out 16-byte aligned for d >= 16.  A plan workgroup's buffers hold 8,192 keys: a list
 * of up to 8,192 ids is grouped in LDS whatever its skew; beyond that a workgroup that owns more than 8,192 keys (one row with
 * a quarter of the list) selects its rows one by one -- correct, slow (the engine keeps such lists on the sort route).     */
int rc_small_row_sums_supported(int64_t n, int64_t n_rows, int d);
size_t rc_small_row_sums_workspace_bytes(int64_t n);
int rc_small_row_sums(const int64_t* ids, int64_t n, int64_t n_rows, const float* src, int d, float* out, void* ws,
                      size_t ws_bytes, rc_stream_t stream);
/* A second table gathered with the SAME ids (the [vocab, d] vectors and the [vocab, 1] first-order weights of the FM family,
 * models/context/FM.py:44-57): the row sums of another src / out pair on the grouping that the preceding rc_small_row_sums call
 * left in `ws` (same n, same n_rows, ws untouched in between) -- one launch instead of two.                                    */
int rc_small_row_sums_again(int64_t n, int64_t n_rows, const float* src, int d, float* out, void* ws, size_t ws_bytes,
                            rc_stream_t stream);
/* ... and where the second table is one float wide (the [vocab, 1] first-order weights, src1 [n], out1 [n_rows]) and d >= 16, both
 * sums in the SAME launch: grouping + one row-sums kernel.                                                                     */
int rc_small_row_sums_pair(const int64_t* ids, int64_t n, int64_t n_rows, const float* src, int d, float* out,
                           const float* src1, float* out1, void* ws, size_t ws_bytes, rc_stream_t stream);
/* rc_small_row_sums_pair over the gradient blocks of a field list that holds numeric features (rc_gather_fields_mixed: their
 * occurrences carry id -1 and take no part in the grouping): src = gV [B * C, F, d] as n = B * C * F occurrence rows, src1 = gL, and
 * the numeric fields' weight gradients (rc_numeric_field_grads: values / per_row / kind / field / dW / dw1, HOST arrays of length
 * n_numeric <= 4) are formed by one extra workgroup each of the SAME launch.  d a multiple of 4, 16 <= d <= 128.                   */
int rc_small_row_sums_pair_numeric(const int64_t* ids, int64_t n, int64_t n_rows, const float* src, int d, float* out,
                                   const float* src1, float* out1, const void* const* values, const int* per_row, const int* kind,
                                   const int* field, int n_numeric, int F, int64_t B, int C, float* const* dW, float* const* dw1,
                                   void* ws, size_t ws_bytes, rc_stream_t stream);

/* The row sums of a backward pass whose grouping rc_gather_fields_fused left in ws (no plan launch): both table families of the FM
 * models (src = gV [B * C, F, d] as n = B * C * F occurrence rows, src1 = gL [n]), with -- each optional -- the numeric fields' weight
 * gradients riding along (n_numeric > 0, arguments as rc_small_row_sums_pair_numeric) and the FM pairwise term's backward folded in
 * (fm_V != NULL: the stacked field vectors [B * C, F, d], fm_S their field sums [B * C, d], fm_g = d loss / d fm [B * C]):
 * occurrence o = r F + f then contributes src[o] + fm_g[r] * (fm_S[r] - fm_V[o]) -- rc_fm_second_order_bwd_add's rows, never
 * written out (models/context/FM.py:61, DeepFM.py:19-28); src may be NULL when nothing else consumed the field vectors.            */
int rc_small_row_sums_planned(int64_t n, int64_t n_rows, const float* src, int d, float* out, const float* src1, float* out1,
                              const void* const* values, const int* per_row, const int* kind, const int* field, int n_numeric, int F,
                              int64_t B, int C, float* const* dW, float* const* dw1, const float* fm_V, const float* fm_S,
                              const float* fm_g, void* ws, size_t ws_bytes, rc_stream_t stream);

/* The CTR head of the context models in one pass: z = bias[0] + sum_f lin[i, f] (+ term1[i]) (+ term2[i])
 * (models/context/FM.py:59-60, DeepFM.py:27, WideDeep.py:46), p = sigmoid(z) (BaseContextModel.py:74-78), the per-row term of
 * nn.BCELoss()(p, label) with torch's clamps (BaseModel.py:259-267; loss = mean of loss_vec) and gz = d loss / d z.
 * lin [n, F] first-order values; term1 / term2 [n] or NULL; label int64 {0, 1}.                                          */
int rc_ctr_head_fwd_bwd(const float* bias, const float* lin, int F, const float* term1, const float* term2,
                        const int64_t* label, int64_t n, float* p, float* loss_vec, float* gz, rc_stream_t stream);
/* rc_ctr_head_fwd_bwd for n <= 65,536 rows in ONE workgroup that also forms sums[0] = the BCE loss (mean of loss_vec, nn.BCELoss's
 * reduction, BaseModel.py:262-267) and sums[1] = sum gz (= d loss / d overall_bias) -- two launches fewer per step; and the
 * backward fan-out of the head: g [n] = gz g_loss[0], g_lin [n, F] = g broadcast over the F first-order weights of a row
 * (contiguous), g_bias [1] = sums[1] g_loss[0] (autograd's mul / sum / expand-copy in one launch).                              */
int rc_ctr_head_fwd_bwd_sums(const float* bias, const float* lin, int F, const float* term1, const float* term2,
                             const int64_t* label, int64_t n, float* p, float* loss_vec, float* gz, float* sums,
                             rc_stream_t stream);
/* rc_ctr_head_fwd_bwd_sums that also leaves the backward fan-out for a seed gradient of exactly one (a whole training step calls
 * loss.backward() on the scalar loss, helpers/BaseRunner.py:205): g_lin [n, F] = gz broadcast over a row's first-order weights,
 * g_bias [1] = sum gz -- rc_ctr_head_bwd's outputs for g_loss = 1, bit for bit (g itself is gz) -- and optionally increments a
 * device counter nothing in this launch reads (bump, may be NULL: Adam's step count, read next by the update kernel).            */
int rc_ctr_head_fwd_full(const float* bias, const float* lin, int F, const float* term1, const float* term2, const int64_t* label,
                         int64_t n, float* p, float* loss_vec, float* gz, float* sums, float* g_lin, float* g_bias, int64_t* bump,
                         rc_stream_t stream);
int rc_ctr_head_bwd(const float* gz, const float* sums, const float* g_loss, int64_t n, int F, float* g, float* g_lin,
                    float* g_bias, rc_stream_t stream);

/* ---- SASRec encoder (models/sequential/SASRec.py:51-86, utils/layers.py:9-63,92-118) ------- */

/* Two tables that share their ids (NeuMF's mf / mlp embedding of a user or an item, models/general/NeuMF.py:37-40)
 * updated in ONE pass over keys / perm / heads; src_a, src_b: per-occurrence gradient rows [n_occ, d]; outputs as
 * in rc_segmented_update (both dense_grad_*, or W_* (+ m, v) with h).  2 d must be one of 16/32/64/128/256.
 * Workspace: rc_segmented_workspace_bytes(n_occ, 2 d).                                                      */
int rc_segmented_update_pair(float* W_a, float* m_a, float* v_a, float* W_b, float* m_b, float* v_b, int d,
                             const uint32_t* keys, const uint32_t* perm, int64_t n_occ, const float* src_a,
                             const float* src_b, const rc_opt_hyper* h, float* dense_grad_a, float* dense_grad_b,
                             const uint32_t* heads, const uint32_t* n_heads, void* ws, size_t ws_bytes,
                             rc_stream_t stream);

/* 1 iff the kernels cover the shape: d in {32,64}, 1..4 layers, heads | d, L <= 64 (every entry point below); also
 * 64 < L <= 128 with ONE layer and 1 / 2 / 4 heads, for rc_sasrec_batch_fwd / _bwd without dropout (their one-row path). */
int rc_sasrec_supported(int d, int n_layers, int n_heads, int L);
/* floats per layer in the dense-gradient block of rc_sasrec_bwd: 5*d*d + 9*d, in the order
 * Wq bq Wk bk Wv bv ln1.w ln1.b W1 b1 W2 b2 ln2.w ln2.b                                          */
int rc_sasrec_dense_param_count(int d);
size_t rc_sasrec_workspace_bytes(int B, int d, int n_layers);

/* hv[b,:] = encoder output at position lengths[b]-1 (SASRec.py:58-76).  layer_params: HOST array of
 * 14*n_layers DEVICE pointers in the order above (nn.Linear weights [out,in]).  hist [B,L] is right
 * padded with 0.  xsave: NULL (inference) or [B, n_layers, L, d] scratch that receives every layer's
 * input for rc_sasrec_bwd.  ws: rc_sasrec_workspace_bytes (holds transposed weight copies).  Scores are then rc_gather_dot_fwd(U=hv, uid=arange(B)) (SASRec.py:80-81). */
int rc_sasrec_fwd(const float* item_emb, const float* pos_emb, const float* const* layer_params,
                  int n_layers, int n_heads, const int64_t* hist, const int64_t* lengths, int B, int L,
                  int d, float* hv, float* xsave, void* ws, size_t ws_bytes, rc_stream_t stream);

/* Backward of rc_sasrec_fwd for dL/dhv = dhv [B,d]: g_hist [B,L,d] = gradient of the layer-0 input
 * rows (= per-occurrence gradients of item_emb[hist] AND pos_emb[position]; 0 past the length) and
 * dense_grads [n_layers, 5*d*d+9*d] in the canonical order.                                        */
int rc_sasrec_bwd(const float* const* layer_params, int n_layers, int n_heads, const int64_t* lengths,
                  int B, int L, int d, const float* xsave, const float* dhv, float* g_hist,
                  float* dense_grads, void* ws, size_t ws_bytes, rc_stream_t stream);

/* The same encoder decomposed at batch level (csrc/sasrec_batch.hip): the valid history rows of the batch form
 * one compact row space [sum len, d]; projections, LayerNorms and the FFN are streaming kernels over it with the
 * weights in LDS, only the attention runs per sequence.  Same arithmetic and results (to fp32 summation order)
 * as rc_sasrec_fwd / rc_sasrec_bwd, several times faster from a few hundred sequences up; the forward pass
 * SAVES the activations the backward needs in `state` (rc_sasrec_batch_state_floats floats, caller-owned)
 * instead of the backward recomputing them.  dense_grads / g_hist / hv as in rc_sasrec_fwd / rc_sasrec_bwd.
 * Without dropout the LAST block is computed for the one position per sequence that is consumed (models/sequential/SASRec.py:76:
 * his_vector = his[arange(B), lengths - 1]; causal mask): one query row per sequence and -- for 1 / 2 / 4 heads and
 * max(3, heads + 1) <= L <= 64 -- no keys and values at all (score_j = (Wk_h^T q_h) . x_j, ctx_h = Wv_h sum_j p_j x_j: one pass
 * over the layer input rows per direction; with one block the rows come straight from the tables and g_hist is written
 * directly), mirrored in the backward -- same results to fp32 rounding (the products are associated differently).               */
size_t rc_sasrec_batch_state_floats(int B, int L, int d, int n_layers);
size_t rc_sasrec_batch_workspace_bytes(int B, int L, int d, int n_layers);
int rc_sasrec_batch_fwd(const float* item_emb, const float* pos_emb, const float* const* layer_params,
                        int n_layers, int n_heads, const int64_t* hist, const int64_t* lengths, int B, int L,
                        int d, float* hv, float* state, void* ws, size_t ws_bytes, rc_stream_t stream);
int rc_sasrec_batch_bwd(const float* const* layer_params, int n_layers, int n_heads, const int64_t* lengths,
                        int B, int L, int d, const float* state, const float* dhv, float* g_hist,
                        float* dense_grads, void* ws, size_t ws_bytes, rc_stream_t stream);

/* Training mode with --dropout p (utils/layers.py:104,110 dropout1 on the attention context, :114,117 dropout2 on the
 * FFN output; SASRec.py:47-49 passes the model's --dropout).  The mask is never stored: element (compact row r,
 * feature f) of site s = 2 * layer + {0: dropout1, 1: dropout2} is dropped iff word (f & 3) of
 * Philox4x32-10(key = *seed_dev, counter = (r, s * d/4 + (f >> 2))) < p * 2^32, kept values are scaled by 1/(1-p);
 * r = (sum of min(len, L) over the sequences before b) + position.  The LayerNorm kernels of both passes
 * regenerate it; the caller bumps *seed_dev once per step (rc_step_increment, capturable).  As for NeuMF, parity
 * with the reference is exact given the mask (tests/golden/sasrecdrop_*.npz: the reference in training mode with
 * its nn.Dropout modules swapped for this mask) and distributional in the mask.  p = 0 (seed_dev may be NULL) is
 * rc_sasrec_batch_fwd / rc_sasrec_batch_bwd bit for bit.                                                       */
int rc_sasrec_batch_fwd_dropout(const float* item_emb, const float* pos_emb, const float* const* layer_params,
                                int n_layers, int n_heads, const int64_t* hist, const int64_t* lengths, int B,
                                int L, int d, float drop_p, const uint64_t* seed_dev, float* hv, float* state,
                                void* ws, size_t ws_bytes, rc_stream_t stream);
int rc_sasrec_batch_bwd_dropout(const float* const* layer_params, int n_layers, int n_heads,
                                const int64_t* lengths, int B, int L, int d, float drop_p,
                                const uint64_t* seed_dev, const float* state, const float* dhv, float* g_hist,
                                float* dense_grads, void* ws, size_t ws_bytes, rc_stream_t stream);
/* rc_sasrec_batch_bwd_dropout in two calls, for a caller that lets the item-table update start (on another stream) as soon as the
 * history rows' gradient g_hist is complete while the encoder's parameter gradients are still being formed: part 1 = every launch
 * up to and including the one that completes g_hist, part 2 = the rest (dense_grads is complete after part 2; both parts take the
 * same arguments).  rc_sasrec_batch_bwd_splits: 1 where part 2 is not empty (one block on the last-row path), else part 1 is the
 * whole backward pass.  Same kernels, same results as the one-call form.                                                       */
int rc_sasrec_batch_bwd_splits(int d, int n_layers, int n_heads, int B, int L, float drop_p);
int rc_sasrec_batch_bwd_part(const float* const* layer_params, int n_layers, int n_heads, const int64_t* lengths, int B, int L, int d,
                             float drop_p, const uint64_t* seed_dev, const float* state, const float* dhv, float* g_hist,
                             float* dense_grads, void* ws, size_t ws_bytes, int part, rc_stream_t stream);

/* Gradient of the position table p_embeddings (SASRec.py:64: position id = length - index on valid slots, 0 on
 * padding): grad_pos[p] = sum_b g_hist[b, len_b - p] over the sequences with len_b >= p, rows 0 and > L are zero.
 * Replaces aten::embedding_dense_backward on the [B, L] position ids; fixed summation order.                  */
size_t rc_sasrec_pos_grad_workspace_bytes(int B, int L, int d);
int rc_sasrec_pos_grad(const float* g_hist, const int64_t* lengths, int B, int L, int d, int n_pos,
                       float* grad_pos, void* ws, size_t ws_bytes, rc_stream_t stream);

/* ---- shape-generic sequence-encoder layers (csrc/seq_layers.hip) --------------------------------------------
 * The pieces of utils/layers.py TransformerLayer :92-118 / MultiHeadAttention :9-63 and of SASRec.py:58-76 that are not GEMMs, for
 * the shapes the register-resident encoders above do not cover (any emb_size that is a multiple of 4, any head count and number of
 * blocks, history up to 1,024 positions, dropout): together with rc_linear_fwd / rc_linear_bwd they replace the plugin's torch
 * layers there.  The batch is the padded [B, L, d] block seen as B * L rows; row (b, i) is VALID iff i < min(len_b, L); rows that
 * are not valid hold zeros at every layer boundary and receive zero gradients (the reference zeroes them at the end, SASRec.py:74,
 * and no valid row attends to one).                                                                                              */

/* off[b] = sum_{b' < b} min(lengths[b'], L), b in [0, B]: a valid row's compact index off[b] + i keys the dropout mask like
 * rc_sasrec_batch_fwd_dropout does, and off[b + 1] - off[b] is the sequence's valid row count.                                   */
int rc_seq_offsets(const int64_t* lengths, int64_t B, int L, int32_t* off, rc_stream_t stream);
/* SASRec.py:58-66: X[b, i, :] = item_emb[hist[b, i]] + pos_emb[len_b - i] for i < len_b, 0 on the padding.                       */
int rc_seq_embed_fwd(const float* item_emb, const float* pos_emb, const int64_t* hist, const int64_t* lengths, int64_t B, int L, int d,
                     float* X, rc_stream_t stream);
/* SASRec.py:76: hv[b, :] = X[b, len_b - 1, :] (0 for an empty sequence); backward: dX [B * L, d] = 0 except those rows = dhv.     */
int rc_seq_pick_last_fwd(const float* X, const int64_t* lengths, int64_t B, int L, int d, float* hv, rc_stream_t stream);
int rc_seq_pick_last_bwd(const float* dhv, const int64_t* lengths, int64_t B, int L, int d, float* dX, rc_stream_t stream);
/* rc_sasrec_pos_grad for any row width: grad_pos [n_pos, d], row p = sum_b dX[b, len_b - p] over the sequences with
 * min(len_b, L) >= p >= 1, every other row 0 (the position ids of SASRec.py:64; fixed summation order, ascending b).               */
int rc_seq_pos_grad(const float* dX, const int64_t* lengths, int64_t B, int L, int d, int n_pos, float* grad_pos, rc_stream_t stream);
/* scaled_dot_product_attention (utils/layers.py:52-63) of H heads over Q / K / V [B * L, H * dk] (head h = columns [h * dk,
 * (h + 1) * dk), the layout head_split :30-32 views): ctx[b, i, h] = softmax_j(q_i . k_j / sqrt(dk)) @ v_j over the keys the mask
 * shows -- mask uint8 [mask_batch ? B : 1][L][L] (0 = hidden; NULL = none), causal != 0 additionally hides j > i (and skips that
 * work), off (NULL = every row valid) restricts queries and keys to valid rows; a row that sees no key yields 0 (the reference's
 * NaN -> 0, :61).  lse [B, H, L] = log-sum-exp of a row's scaled scores, saved for the backward pass, which recomputes the
 * probabilities: Dv [B, H, L] is scratch (sum_j p_ij dP_ij), dQ / dK / dV get every element written.  L <= 1,024, dk <= 256.       */
int rc_seq_attention_supported(int L, int dk);
int rc_seq_attention_fwd(const float* Q, const float* K, const float* V, const int32_t* off, const uint8_t* mask, int mask_batch,
                         int causal, int64_t B, int L, int H, int dk, float* ctx, float* lse, rc_stream_t stream);
int rc_seq_attention_bwd(const float* Q, const float* K, const float* V, const int32_t* off, const uint8_t* mask, int mask_batch,
                         int causal, int64_t B, int L, int H, int dk, const float* lse, const float* dctx, float* Dv, float* dQ,
                         float* dK, float* dV, rc_stream_t stream);
/* TransformerLayer.forward :110,117: Y = LayerNorm(dropout(A) + R) over rows of d floats (R NULL: no residual); xhat [rows, d] and
 * rstd [rows] are kept for the backward pass.  off NULL: every row is valid and its compact index is the row number; otherwise rows
 * = B * L and invalid rows give Y = xhat = 0.  drop_p > 0: element (compact row r, feature f) of `site` is dropped iff word (f & 3)
 * of Philox4x32-10(key = *seed_dev, counter = (r, site * d / 4 + (f >> 2))) < drop_p * 2^32, kept values scaled by 1 / (1 - p) --
 * the stream of rc_sasrec_batch_fwd_dropout (site = 2 * layer + {0: dropout1, 1: dropout2}).  Backward: dA = mask * dZ (NULL:
 * not wanted), dR = dZ, dw = sum_rows dY * xhat, db = sum_rows dY in a fixed order.  d a multiple of 4 up to 1,024.                */
int rc_seq_add_layernorm_fwd(const float* A, const float* R, const float* w, const float* b, const int32_t* off, int64_t rows, int L,
                             int d, float drop_p, const uint64_t* seed_dev, uint32_t site, float* Y, float* xhat, float* rstd,
                             rc_stream_t stream);
size_t rc_seq_add_layernorm_bwd_workspace_bytes(int d);
int rc_seq_add_layernorm_bwd(const float* dY, const float* xhat, const float* rstd, const float* w, const int32_t* off, int64_t rows,
                             int L, int d, float drop_p, const uint64_t* seed_dev, uint32_t site, float* dA, float* dR, float* dw,
                             float* db, void* ws, size_t ws_bytes, rc_stream_t stream);

/* ---- NeuMF head (models/general/NeuMF.py:56-76), one hidden layer ------------------------- */

/* 1 iff the fp32-MFMA kernels cover (emb_size d, hidden size l1): d, l1 in {32,64,128} and the
 * LDS image (W1 + a 64-candidate tile) fits 160 KB.  Other shapes: use rc_gather_rows + any GEMM. */
int rc_neumf_supported(int d, int l1);

/* pred[b,c] = w_out[:d].(mf_u[u_b]*mf_i[i_bc]) + w_out[d:].relu(W1 [mlp_u[u_b];mlp_i[i_bc]] + b1)
 * (NeuMF.py:61-75; W1 = mlp.0.weight [l1, 2d], b1 = mlp.0.bias, w_out = prediction.weight[0]).
 * Without dropout a wave owns 16 candidates: the mlp rows go from global memory straight into the MFMA operand registers,
 * W1 waits in LDS (52 % of the fp32 MFMA peak, gathers at 5 TB/s at d = 128, hidden 64).                                  */
int rc_neumf_fwd(const float* mf_u, const float* mf_i, const float* mlp_u, const float* mlp_i,
                 const float* W1, const float* b1, const float* w_out, const int64_t* uid,
                 const int64_t* iid, int B, int C, int d, int l1, float* pred,
                 rc_stream_t stream);

size_t rc_neumf_workspace_bytes(int B, int C, int d, int l1);

/* Backward of rc_neumf_fwd for dL/dpred = gpred [B,C] (what autograd derives from NeuMF.py:61-75):
 * per-OCCURRENCE gradients of the four table rows, g_*[b*C+c, :]  (feed them to
 * rc_segmented_update with keys = sorted uid-per-candidate / iid), and the dense gradients
 * dW1 [l1,2d], db1 [l1], dw_out [d+l1] (summed over workgroups in fixed order).              */
int rc_neumf_bwd(const float* mf_u, const float* mf_i, const float* mlp_u, const float* mlp_i,
                 const float* W1, const float* b1, const float* w_out, const int64_t* uid,
                 const int64_t* iid, const float* gpred, int B, int C, int d, int l1,
                 float* g_mf_u, float* g_mf_i, float* g_mlp_u, float* g_mlp_i, float* dW1,
                 float* db1, float* dw_out, void* ws, size_t ws_bytes, rc_stream_t stream);

/* The same head in TRAINING mode with dropout p on the hidden layer (NeuMF.py:58 nn.Dropout after the
 * ReLU, the reference's demo runs --dropout 0.2): hidden feature f of candidate n = b*C+c is zeroed iff
 * word (f & 3) of Philox4x32-10(key = *seed_dev, counter = (n, f >> 2)) < p * 2^32, kept values are scaled
 * by 1/(1-p).  The mask is never stored: the backward regenerates it from the same *seed_dev, which
 * therefore must not change between the two calls; bump it (rc_step_increment) once per step, also inside a
 * captured graph.  The stream differs from torch's Philox dropout: parity with the reference is
 * distributional, parity with oracle/neumf_oracle.py (same counter scheme) is exact in the mask.
 * p = 0 (seed_dev may then be NULL) is rc_neumf_fwd / rc_neumf_bwd bit for bit.                          */
int rc_neumf_fwd_dropout(const float* mf_u, const float* mf_i, const float* mlp_u, const float* mlp_i,
                         const float* W1, const float* b1, const float* w_out, const int64_t* uid,
                         const int64_t* iid, int B, int C, int d, int l1, float drop_p,
                         const uint64_t* seed_dev, float* pred, rc_stream_t stream);
int rc_neumf_bwd_dropout(const float* mf_u, const float* mf_i, const float* mlp_u, const float* mlp_i,
                         const float* W1, const float* b1, const float* w_out, const int64_t* uid,
                         const int64_t* iid, const float* gpred, int B, int C, int d, int l1,
                         float drop_p, const uint64_t* seed_dev, float* g_mf_u, float* g_mf_i,
                         float* g_mlp_u, float* g_mlp_i, float* dW1, float* db1, float* dw_out,
                         void* ws, size_t ws_bytes, rc_stream_t stream);

/* ONE BaseRunner.fit iteration of the NeuMF head (helpers/BaseRunner.py:193-206 on models/general/NeuMF.py:56-76 with
 * GeneralModel.loss, models/BaseModel.py:182-185) in one kernel (csrc/neumf_step.hip): forward, BPR loss, backward and the
 * row-wise optimizer update of every ITEM row that a single batch position touches -- those rows never leave the registers
 * they were gathered into.  A wave owns 16 tuples and walks their C candidates, so the user rows, the user half of the hidden
 * layer (W1 [mlp_u ; mlp_i] = W1u mlp_u + W1i mlp_i) and the user-row gradients are per-tuple work.
 *   tables        mf_u, mlp_u are read; mf_i, mlp_i (and their optimizer state m_*, v_*: Adam both, Adagrad m, SGD none -> NULL)
 *                 are updated in place for single-occurrence rows (row-wise semantics: only touched rows move, l2 on them)
 *   marks         rc_neumf_train_step_marks_bytes(n_items) bytes, ALL ZERO before the first call; the call leaves it ready for
 *                 the next one: an owner map (4 B per item id, written before it is read) and a flag byte per item id ("occurs
 *                 at least twice", cleared again) filled by two marking passes with plain stores -- no atomics, and the
 *                 flags do not depend on the schedule
 *   loss_vec [B]  per-tuple loss; dL/dpred uses inv_b (1/B = the reference's .mean()); pred [B, C] or NULL
 *   g_mf_i, g_mlp_i [B C, d]  gradient rows of the positions whose item row occurs more than once (other positions are NOT
 *                 written): feed them to rc_plan_update_pair on a plan of iid built with list_single_a = 0
 *   gu_mf, gu_mlp [B, d]      ONE gradient row per tuple for the user tables (plan the user ids per tuple)
 *   dW1, db1, dw_out          dense gradients, per-workgroup partials summed in fixed order; no float atomics anywhere
 * Shapes: rc_neumf_train_step_supported(C, d, l1): d in {32,64,128}, l1 in {16,32,64} (l1 = 16: d >= 64), C >= 2 and the LDS image <= 160 KB
 * (C <= 136 at d = 128, l1 = 64).  Training-mode dropout (the reference's own NeuMF command line runs --dropout 0.2,
 * docs/demo_scripts_results/Topk_Amazon.sh:8; NeuMF.py:58,70): rc_neumf_train_step_dropout below.                          */
int rc_neumf_train_step_supported(int C, int d, int l1);
size_t rc_neumf_train_step_workspace_bytes(int B, int C, int d, int l1);
size_t rc_neumf_train_step_marks_bytes(int64_t n_items);
int rc_neumf_train_step(float* mf_u, float* mf_i, float* mlp_u, float* mlp_i, float* m_mf_i, float* v_mf_i,
                        float* m_mlp_i, float* v_mlp_i, const float* W1, const float* b1, const float* w_out,
                        const int64_t* uid, const int64_t* iid, int B, int C, int d, int l1, int64_t n_items,
                        void* marks, const rc_opt_hyper* h, float inv_b, float* loss_vec, float* pred,
                        float* g_mf_i, float* g_mlp_i, float* gu_mf, float* gu_mlp, float* dW1, float* db1,
                        float* dw_out, void* ws, size_t ws_bytes, rc_stream_t stream);

/* The marking passes of rc_neumf_train_step on their own, for callers that know the FOLLOWING batch (BaseRunner.fit does: the
 * reference's DataLoader runs ahead of the loop, helpers/BaseRunner.py:182-186): rc_neumf_mark_rows fills a marks buffer for iid
 * [n = B C] on any stream -- e.g. beside the current step's table updates --, rc_neumf_train_step_marked is rc_neumf_train_step
 * without its two marking launches and without the clearing pass (same arguments, same results bit for bit), and
 * rc_neumf_unmark_rows clears the flags again before the buffer is marked for another batch.  Two buffers alternate.          */
int rc_neumf_mark_rows(const int64_t* iid, int64_t n, int64_t n_items, void* marks, rc_stream_t stream);
int rc_neumf_unmark_rows(const int64_t* iid, int64_t n, int64_t n_items, void* marks, rc_stream_t stream);
int rc_neumf_train_step_marked(float* mf_u, float* mf_i, float* mlp_u, float* mlp_i, float* m_mf_i, float* v_mf_i,
                               float* m_mlp_i, float* v_mlp_i, const float* W1, const float* b1, const float* w_out,
                               const int64_t* uid, const int64_t* iid, int B, int C, int d, int l1, int64_t n_items,
                               void* marks, const rc_opt_hyper* h, float inv_b, float* loss_vec, float* pred,
                               float* g_mf_i, float* g_mlp_i, float* gu_mf, float* gu_mlp, float* dW1, float* db1,
                               float* dw_out, void* ws, size_t ws_bytes, rc_stream_t stream);

/* rc_neumf_train_step with nn.Dropout(p) on the hidden layer (models/general/NeuMF.py:58, 70) inside the kernel: the mask of
 * rc_neumf_fwd_dropout / rc_neumf_bwd_dropout -- feature f of candidate n = b C + c is dropped iff word (f & 3) of
 * Philox4x32-10(key = *seed_dev, counter = (n, f >> 2)) < p 2^32, kept values scaled by 1 / (1 - p) -- generated in the forward
 * pass of a candidate and regenerated, not stored, in its backward pass.  *seed_dev is read by the kernel (bump it with
 * rc_step_increment for a fresh mask per step; capturable).  drop_p = 0: exactly rc_neumf_train_step.  marks_prepared != 0: the
 * marks buffer was filled by rc_neumf_mark_rows for this very batch (as rc_neumf_train_step_marked).                          */
int rc_neumf_train_step_dropout(float* mf_u, float* mf_i, float* mlp_u, float* mlp_i, float* m_mf_i, float* v_mf_i,
                                float* m_mlp_i, float* v_mlp_i, const float* W1, const float* b1, const float* w_out,
                                const int64_t* uid, const int64_t* iid, int B, int C, int d, int l1, int64_t n_items,
                                void* marks, int marks_prepared, const rc_opt_hyper* h, float inv_b, float drop_p,
                                const uint64_t* seed_dev, float* loss_vec, float* pred, float* g_mf_i, float* g_mlp_i,
                                float* gu_mf, float* gu_mlp, float* dW1, float* db1, float* dw_out, void* ws,
                                size_t ws_bytes, rc_stream_t stream);

/* The same kernel without table updates, on row blocks with a stride: forward, BPR loss and backward of the NeuMF head for
 * callers that own neither the tables nor the optimizer -- the row-sharded step (rechorus_amd/sharded.py, ShardedNeumf), where
 * the rows of a batch were fetched from their owners into blocks [rows, mf | mlp] (ld = 2 d) and the gradient rows travel back
 * the same way.  uid [B] / iid [B, C] index the row blocks; every position writes its item gradient rows (g_mf_i / g_mlp_i at
 * stride ld_gi), every tuple its two user gradient rows (stride ld_gu); loss_vec, pred, dense gradients, workspace and shapes
 * as rc_neumf_train_step.  Strides are in floats, multiples of 4, at least d.                                                   */
int rc_neumf_head_fwd_bwd(const float* mf_u, const float* mlp_u, int64_t ld_u, const float* mf_i, const float* mlp_i,
                          int64_t ld_i, const float* W1, const float* b1, const float* w_out, const int64_t* uid,
                          const int64_t* iid, int B, int C, int d, int l1, float inv_b, float* loss_vec, float* pred,
                          float* g_mf_i, float* g_mlp_i, int64_t ld_gi, float* gu_mf, float* gu_mlp, int64_t ld_gu,
                          float* dW1, float* db1, float* dw_out, void* ws, size_t ws_bytes, rc_stream_t stream);

/* The head of a row-sharded NeuMF step whose ITEM half of the hidden layer was computed by the rows' owners (csrc/neumf_zhead.hip;
 * models/general/NeuMF.py:61-75: the user ids are tiled over the candidates, so h = relu(W1u mlp_u + W1i mlp_i + b1)): the fetched
 * item rows hold (mf_i [d] | zi = W1i mlp_i [l1]) instead of both table rows, zu [B, l1] = W1u mlp_u + b1 comes from rc_linear_fwd.
 * Forward, GeneralModel.loss (BaseModel.py:182-185) and the backward in one launch: loss_vec [B], pred [B, C] (optional), gi [B C]
 * rows (d mf_i | dz) -- what travels back to the owners, who form d mlp_i = W1i^T dz and their share of dW1i = dz^T mlp_i with
 * rc_linear_bwd --, gu_mf [B, d] = d mf_u, dzu [B, l1] = sum_c dz_c (the caller's rc_linear_bwd turns it into d mlp_u, dW1u, db1),
 * dw_out [d + l1].  Row strides in floats (the blocks of the sharded step hold their halves side by side).  2 <= C <= 256,
 * d, l1 <= 1,024; ws: rc_neumf_zhead_workspace_bytes.                                                                              */
int rc_neumf_zhead_supported(int C, int d, int l1);
size_t rc_neumf_zhead_workspace_bytes(int d, int l1);
int rc_neumf_zhead_fwd_bwd(const float* mf_u, int64_t ld_u, const float* zu, const float* irows, int64_t ld_i, const float* w_out, int B,
                           int C, int d, int l1, float inv_b, float* loss_vec, float* pred, float* gi, int64_t ld_gi, float* gu_mf,
                           int64_t ld_gu, float* dzu, float* dw_out, void* ws, size_t ws_bytes, rc_stream_t stream);

/* ---- dense layers of the heads (csrc/mlp.hip): fp32 MFMA GEMMs ------------------------------------
 * utils/layers.py:201-243 (MLP_Block: Linear -> ReLU -> Dropout per hidden layer + output Linear; the deep part of
 * models/context/DeepFM.py:25 / WideDeep.py:42-47) and the NeuMF tower for any --layers
 * (models/general/NeuMF.py:47-52, 69-72), with nn.Linear's autograd.  Replaces aten::addmm (rocBLAS) + relu + dropout.
 *   rc_linear_fwd: Y [M, N] = drop(relu(X [M, K] W^T + b)); W [N, K] as nn.Linear.weight; b may be NULL; relu 0 / 1.
 *     Dropout (training): element (m, n) is dropped iff word (m & 3) of Philox4x32-10(key = *seed_dev,
 *     counter = (m >> 2, site * 65536 + n)) < drop_p * 2^32, kept values are scaled by 1 / (1 - drop_p); `site`
 *     distinguishes the layers of one forward pass; the caller bumps *seed_dev once per step (rc_step_increment).
 *   rc_linear_bwd: given the layer's saved OUTPUT Y (its own mask: dZ = dY * (Y > 0 ? 1/(1-p) : 0); NULL for a plain
 *     Linear) -> dX [M, K] (NULL: skip), dW [N, K], db [N] (NULL: skip).  The batch reduction of dW / db is cut into
 *     row ranges whose partial sums are combined in a fixed order (no float atomics).                            */
int rc_linear_fwd(const float* X, const float* W, const float* b, int64_t M, int N, int K, int relu, float drop_p,
                  const uint64_t* seed_dev, uint32_t site, float* Y, rc_stream_t stream);
/* rc_linear_fwd with a workspace: a small batch (fewer than 256 output tiles of 64 x 64) is cut along the reduction -- split-K,
 * the partial products summed in split order by a second launch that applies bias / ReLU / dropout -- so that a
 * 1,024 x 512 x 512 product is 512 workgroups of four K steps instead of 128 of sixteen.  ws NULL: no split.             */
size_t rc_linear_fwd_workspace_bytes(int64_t M, int N, int K);
int rc_linear_fwd_ws(const float* X, const float* W, const float* b, int64_t M, int N, int K, int relu, float drop_p,
                     const uint64_t* seed_dev, uint32_t site, float* Y, void* ws, size_t ws_bytes, rc_stream_t stream);
size_t rc_linear_bwd_workspace_bytes(int64_t M, int N, int K);
int rc_linear_bwd(const float* X, const float* W, const float* Y, const float* dY, int64_t M, int N, int K,
                  float drop_p, float* dX, float* dW, float* db, void* ws, size_t ws_bytes, rc_stream_t stream);
/* rc_linear_bwd inside a chain of layers (utils/layers.py:201-243 builds Linear -> ReLU -> Dropout groups): with x_act != 0 the
 * input X is the drop(relu(.)) output of the layer below and dX comes out already multiplied by that layer's mask
 * (X > 0 ? 1 / (1 - x_drop_p) : 0) in the product's epilogue; the layer below is then called with Y = NULL (its dY is its dZ).
 * Either of dX / dW (+ db) may be NULL here: the two products of a layer are independent given dZ, and a caller may issue them in
 * two calls on two streams (the dX chain is the critical path of a small batch; workspaces must then differ). */
int rc_linear_bwd_chain(const float* X, const float* W, const float* Y, const float* dY, int64_t M, int N, int K, float drop_p,
                        int x_act, float x_drop_p, float* dX, float* dW, float* db, void* ws, size_t ws_bytes, rc_stream_t stream);

/* The TAIL of an MLP tower at a small batch (csrc/tower_tail.hip): the last hidden layer Linear(K -> N2) -> ReLU -> Dropout(p) and
 * the output layer Linear(N2 -> 1) of utils/layers.py:201-243 (MLP_Block: models/context/DeepFM.py:25, WideDeep.py:42-47 at
 * --layers [512,64] / [64]; models/general/NeuMF.py:47-52 for multi-layer towers) as ONE forward and ONE backward kernel (+ the
 * fixed-order sum of the per-workgroup partials) instead of eleven GEMM / epilogue launches of 5-15 us each.
 *   rc_tower_tail_fwd: H2 [M, N2] = drop(relu(X [M, K] W2^T + b2)) (the mask of rc_linear_fwd, layer index `site`), z [M] = H2 w3 + b3
 *   rc_tower_tail_bwd: given dz [M] = d loss / d z: dX [M, K] = ((dz w3^T * mask(H2)) W2), multiplied by the mask of the layer below
 *     when x_act != 0 (X > 0 ? 1 / (1 - x_drop_p) : 0, as rc_linear_bwd_chain); dW2 [N2, K], db2 [N2], dw3 [N2], db3 [1].
 *     dX, db2, db3 may be NULL.  ws: rc_tower_tail_workspace_bytes(M, K, N2).  No float atomics.
 * Shapes: rc_tower_tail_supported: N2 in {16, 32, 64}, K in {64, 128, 256, 512}; X, W2, H2, w3 16-byte aligned.                 */
int rc_tower_tail_supported(int64_t M, int K, int N2);
size_t rc_tower_tail_workspace_bytes(int64_t M, int K, int N2);
int rc_tower_tail_fwd(const float* X, const float* W2, const float* b2, const float* w3, const float* b3, int64_t M, int K,
                      int N2, float drop_p, const uint64_t* seed_dev, uint32_t site, float* H2, float* z, rc_stream_t stream);
int rc_tower_tail_bwd(const float* X, const float* W2, const float* w3, const float* H2, const float* dz, int64_t M, int K,
                      int N2, float drop_p, int x_act, float x_drop_p, float* dX, float* dW2, float* db2, float* dw3,
                      float* db3, void* ws, size_t ws_bytes, rc_stream_t stream);

/* ---- training-batch assembly on the device (csrc/sampler.hip) -------------------------------- */

/* GeneralModel.Dataset.actions_before_epoch (models/BaseModel.py:206-214): neg[i,k] ~ uniform over
 * [1, n_items) minus the TRAIN clicked set of users[i] (rejection sampling).  The clicked sets are a CSR
 * over user ids: clicked_items[clicked_ptr[u] .. clicked_ptr[u+1]) sorted ascending (both NULL: no
 * rejection).  Counter-based Philox4x32-10: element e = i*K + k of this call uses the stream
 * (seed, base_index + e), so results do not depend on launch geometry and calls with disjoint
 * [base_index, base_index + n*K) ranges are independent.                                            */
int rc_sample_negatives(const int64_t* users, int64_t n, int K, int64_t n_items, const int64_t* clicked_ptr,
                        const int64_t* clicked_items, uint64_t seed, uint64_t base_index, int64_t* neg,
                        rc_stream_t stream);

/* One shuffled batch of GeneralModel.Dataset rows (models/BaseModel.py:192-203 + collate_batch :135-152):
 * users_out[b] = users[idx[b]]; cand_out[b,0] = items[idx[b]], cand_out[b,1+k] = neg[idx[b],k].       */
int rc_assemble_candidates(const int64_t* idx, int64_t B, int K, const int64_t* users, const int64_t* items,
                           const int64_t* neg, int64_t* users_out, int64_t* cand_out, rc_stream_t stream);

/* SequentialModel.Dataset._get_feed_dict (models/BaseModel.py:236-245): for row i = idx[b] (idx NULL:
 * i = b) with user u = users[i] and history position p = position[i], len = min(p, L):
 * hist[b, t] = his_items[his_ptr[u] + p - len + t] for t < len, 0 beyond (right padding, as
 * pad_sequence in collate_batch); times (optional) likewise from his_times; lengths[b] = len.          */
int rc_gather_history(const int64_t* idx, int64_t B, int L, const int64_t* users, const int64_t* position,
                      const int64_t* his_ptr, const int64_t* his_items, const int64_t* his_times, int64_t* hist,
                      int64_t* times, int64_t* lengths, rc_stream_t stream);

/* ---- evaluation (csrc/eval_rank.hip) ------------------------------------------------------------- */

/* BaseRunner.evaluate_method (helpers/BaseRunner.py:62-63): rank[i] = #{c : pred[i,c] >= pred[i,0]}
 * (ground truth in column 0, ties count against it).  HR@k / NDCG@k are means over rank.             */
int rc_target_rank(const float* pred, int64_t n, int C, int32_t* rank, rc_stream_t stream);

/* ImpressionRunner.evaluate / evaluate_method with HR_at_k / NDCG_at_k / AP_at_k (helpers/ImpressionRunner.py:18-66,74-133,135-168)
 * on the device: pred [N, n] scores of impression lists -- positives in columns [0, pos_num[i]) (pos_num NULL: one per row,
 * :85-86), negatives in [max_pos, max_pos + neg_num[i]), counts clipped to max_pos / n - max_pos (:99-102), every other column
 * ignored (the reference masks it to -inf, :156-168).  A positive that ties with a negative ranks below it (the 1e-6 shift of
 * :89-96), ties inside a group keep column order (the stable merge sort of :98).  per_row [N, 3, n_k] float64 = NDCG@k, MAP@k,
 * HR@k of every row for k = topk[0 .. n_k) (HOST array, n_k <= 16) -- what evaluate_method returns with ret_all = 1; mean
 * [3, n_k] (optional) = their means over the rows (ret_all = 0).  n <= 2,048 (rc_list_metrics_supported).                        */
int rc_list_metrics_supported(int n, int max_pos, int n_k);
int rc_list_metrics(const float* pred, const int64_t* pos_num, const int64_t* neg_num, int64_t N, int n, int max_pos,
                    const int* topk, int n_k, double* per_row, double* mean, rc_stream_t stream);

/* --test_all (models/BaseModel.py:194-195, helpers/BaseRunner.py:243-250) for dot-product heads
 * (BPRMF: Uvec = gathered user rows; SASRec: the encoder outputs) without materialising [N, n_items]:
 * rank[i] = 1 + #{j in [1,n_items), j != targets[i], j not in clicked(users[i]) : <Uvec[i], I[j]> >= <Uvec[i], I[targets[i]]>}
 * clicked = CSR over user ids, sorted (train + residual clicked sets; NULL = no masking).  fp32 MFMA
 * over the catalogue; target_score [N] is an output (scratch).  d in {32, 64, 128}.                   */
int rc_full_catalogue_rank_supported(int d);
int rc_full_catalogue_rank(const float* Uvec, const float* I, const int64_t* users, const int64_t* targets, int64_t N,
                           int64_t n_items, int d, const int64_t* clicked_ptr, const int64_t* clicked_items,
                           float* target_score, int32_t* rank, rc_stream_t stream);

/* ---- row-sharded training: local kernels of the "owner computes" step (csrc/owner_step.hip) ------------
 * New with the multi-GPU engine (the reference is single-device).  Tables are sharded by
 * row: owner(id) = id mod world, local row = id div world.                                               */

size_t rc_route_workspace_bytes(int64_t n, int world);

/* Stable grouping of ids[n] by owner rank: order[j] = position of the j-th id in (owner, position) order,
 * counts[w] = ids owned by rank w (device int64[world]); optional messages for the owners:
 * packed[j] = ((tuple_base + order[j] / div) << 32) | (ids[order[j]] / world)   (items: tuple + local row)
 * local_row[j] = ids[order[j]] / world                                           (users: local row)
 * Placement is computed without atomics, so the result is deterministic.  world <= 64.                   */
int rc_route_by_owner(const int64_t* ids, int64_t n, int world, int64_t tuple_base, int div, uint32_t* order,
                      int64_t* packed, int64_t* local_row, int64_t* counts, void* ws, size_t ws_bytes,
                      rc_stream_t stream);

/* received packed messages -> t_idx[n] (int64 tuple index), rows[n] (int64 local row), t32[n] (uint32 t_idx) */
int rc_owner_unpack(const int64_t* packed, int64_t n, int64_t* t_idx, int64_t* rows, uint32_t* t32,
                    rc_stream_t stream);

size_t rc_owner_backward_workspace_bytes(int64_t n);

/* Owner-side backward of the sharded BPRMF step over the n occurrences this rank received (grouped by
 * tuple: t32 non-decreasing): pug[t] = sum_{j: t32[j]=t} g[j] * I[rows[j]]  (pug [n_tuples,d], zero for
 * tuples with no row here), and -- when `single` is given -- the optimizer update of rows with
 * single[j] != 0 (rows occurring once on this owner this step) with gradient g[j] * Uall[t32[j]], using the
 * pre-step row values for pug.  Rows occurring more than once are NOT updated (rc_segmented_update with
 * RC_SEG_SKIP_SINGLETONS does that).  d in {16,32,64,128,256}.                                           */
int rc_owner_backward(float* I, float* mI, float* vI, int d, const float* Uall, const uint32_t* t32,
                      const int64_t* rows, const float* g, const uint8_t* single, int64_t n, int64_t n_tuples,
                      const rc_opt_hyper* h, float* pug, void* ws, size_t ws_bytes, rc_stream_t stream);

/* ---- bucket plan: the occurrences of a batch's row ids grouped by row, without a device-wide sort ----------
 * (csrc/bucket_plan.hip).  Replaces the rc_sort_ids + rc_segment_heads pair in front of the row updates that
 * stand in for aten::embedding_dense_backward's index_add (helpers/BaseRunner.py:205 -> nn.Embedding tables of
 * models/general/BPRMF.py:31-32).  Two id lists (a: e.g. the B*(1+K) item ids, b: e.g. the B user ids, may be
 * empty) are planned together; positions are p in [0, n_a) for list a and n_a + j for list b.               */
typedef struct rc_plan_row {
  uint32_t row;      /* table row (id)                                                        */
  uint32_t start;    /* occ[start .. start + n) = the positions that touch the row, ascending  */
  uint32_t n;        /* occurrences in the batch                                               */
  uint32_t reserved; /* 0                                                                      */
} rc_plan_row;

/* 1 when a plan geometry exists: id-range buckets (ceil(range_a / 8192) + ceil(range_b / 8192) <= 4096; chosen when the
 * ids are dense enough, range <= 16 n) or hashed buckets with an LDS hash table per bucket (sparse or wider id spaces:
 * ids < 2^32 - 2, at most 8 M keys per list); else 0 -> use rc_sort_ids.  RC_PLAN_HASHED=0/1 forces a geometry (A/B). */
int rc_bucket_plan_supported(int64_t n_a, int64_t n_b, int64_t range_a, int64_t range_b);
size_t rc_bucket_plan_workspace_bytes(int64_t n_a, int64_t n_b);
size_t rc_bucket_plan_flags_bytes(int64_t n_a); /* size of single_a: n_a rounded up to whole 8,192-byte tiles */
/* Device address (inside the workspace rc_bucket_plan was given) of the plan's status word, valid once the plan's kernels have
 * run: 0 = complete; 1 = an id lay outside its table (nn.Embedding would raise a device assert; the key was kept in bounds, the
 * plan is NOT what the caller meant); 2 = a hashed bucket met more distinct rows than its table holds (plan incomplete; never at
 * the sizes rc_bucket_plan_supported admits).  Reading it costs a device -> host copy: for tests and debugging.               */
const uint32_t* rc_bucket_plan_status_ptr(const void* ws, int64_t n_a, int64_t n_b);

/* ids_a[n_a] in [0, range_a), ids_b[n_b] in [0, range_b) (int64, reference layout).
 * list_single_a = 0: rows of list a that occur once are NOT listed; instead single_a[p] = 1 at their position
 *   (0 elsewhere) -- the fused BPRMF kernel updates those rows itself (rc_bprmf_fwd_bwd_update);
 * list_single_a = 1: every distinct row of list a is listed, single_a may be NULL.
 * List b always lists every distinct row.  rows_a / rows_b: capacity n_a / n_b records, *n_rows_a / *n_rows_b
 * (device uint32) receive the counts; the ORDER of the records is unspecified (it influences no result), the
 * positions of a row are ascending (=> fixed summation order in rc_plan_update).  occ: n_a + n_b entries.
 * No float arithmetic, integer atomics only.                                                               */
int rc_bucket_plan(const int64_t* ids_a, int64_t n_a, int64_t range_a, const int64_t* ids_b, int64_t n_b,
                   int64_t range_b, int list_single_a, uint8_t* single_a, rc_plan_row* rows_a,
                   uint32_t* n_rows_a, rc_plan_row* rows_b, uint32_t* n_rows_b, uint32_t* occ, void* ws,
                   size_t ws_bytes, rc_stream_t stream);

/* The singleton information alone, as a bitmap over the ids of ONE list: bit (id & 31) of bitmap[id >> 5] = 1 iff
 * row id occurs at least twice in ids_a[0 .. n_a) (words of id ranges the batch does not touch are left unwritten).
 * This is what the fused BPRMF kernel consumes in the bucket-plan step (rc_bprmf_fwd_bwd_update_bitmap): 1.25 MB for
 * a 10 M-row table, L2-resident, written as one coalesced 1 KB run per 8,192-id bucket -- instead of one flag byte per
 * batch position.  bitmap: rc_bucket_bitmap_bytes(range_a) bytes; ws: rc_bucket_plan_workspace_bytes(n_a, 0).      */
size_t rc_bucket_bitmap_bytes(int64_t range_a);
int rc_bucket_multi_bitmap(const int64_t* ids_a, int64_t n_a, int64_t range_a, uint32_t* bitmap, void* ws,
                           size_t ws_bytes, rc_stream_t stream);

/* Optimizer row update of the rows listed by a plan (csrc/plan_update.hip): per row, the gradient rows of its
 * occurrences are summed in ascending batch position (fixed order, no float atomics), the table row is read once,
 * updated by `h` (row-wise: only listed rows move) and written once; rows with more than 32 occurrences go
 * through 256-occurrence chunks.  Same semantics and arguments as rc_segmented_update2 (which it replaces where a
 * plan exists): the gradient row of the occurrence at position o = occ[slot] is
 *   o <  n_split:  coef[o] * src[src_index ? src_index[o / div] : o / div]     (coef NULL = 1)
 *   o >= n_split:  src2[o - n_split]
 * n_occ = length of occ[] (sizes the hot-row scratch).  d in {16, 32, 64, 128, 256}.                          */
size_t rc_plan_update_workspace_bytes(int64_t n_occ, int d);
int rc_plan_update(float* W, float* m, float* v, int d, const rc_plan_row* rows, const uint32_t* n_rows,
                   const uint32_t* occ, int64_t n_occ, const float* coef, const float* src, const int64_t* src_index,
                   int div, const float* src2, int64_t n_split, const rc_opt_hyper* h, void* ws, size_t ws_bytes,
                   rc_stream_t stream);

/* Two tables that share their ids (NeuMF's mf / mlp embedding of a side, models/general/NeuMF.py:37-40) updated in
 * ONE pass over the plan: per-occurrence gradient rows src_a / src_b [*, d], the occurrence at position o reads row
 * o - occ_base of both.  Workspace: rc_plan_update_workspace_bytes(n_occ, 2 d).  d in {8, 16, 32, 64, 128}.      */
int rc_plan_update_pair(float* W_a, float* m_a, float* v_a, float* W_b, float* m_b, float* v_b, int d,
                        const rc_plan_row* rows, const uint32_t* n_rows, const uint32_t* occ, int64_t n_occ,
                        const float* src_a, const float* src_b, int64_t occ_base, const rc_opt_hyper* h, void* ws,
                        size_t ws_bytes, rc_stream_t stream);
/* rc_plan_update_pair with the update's eight ticket counters supplied by the caller (8 uint32 on the device), ZERO-FILLED where
 * that cost nothing -- with the plan's buffers, beside other work on another stream -- and used by nobody since: the memset in
 * front of the update, a launch of its own between a step's fused kernel and its table updates, is left out.                     */
int rc_plan_update_pair_zeroed(float* W_a, float* m_a, float* v_a, float* W_b, float* m_b, float* v_b, int d,
                               const rc_plan_row* rows, const uint32_t* n_rows, const uint32_t* occ, int64_t n_occ,
                               const float* src_a, const float* src_b, int64_t occ_base, const rc_opt_hyper* h,
                               uint32_t* counters, void* ws, size_t ws_bytes, rc_stream_t stream);
/* rc_plan_update_pair with both gradient sources in ONE block: occurrence o reads src_block[o - occ_base, 0 .. d) for table a and
 * [d .. 2 d) for table b, rows src_ld floats apart (a multiple of 4, >= 2 d) -- the (d mf | d mlp) rows of models/general/NeuMF.py:39-42's
 * two table families as a row-sharded rank receives them, used where they lie.  counters: as rc_plan_update_pair_zeroed, or NULL.  */
int rc_plan_update_pair_block(float* W_a, float* m_a, float* v_a, float* W_b, float* m_b, float* v_b, int d,
                              const rc_plan_row* rows, const uint32_t* n_rows, const uint32_t* occ, int64_t n_occ,
                              const float* src_block, int64_t src_ld, int64_t occ_base, const rc_opt_hyper* h,
                              uint32_t* counters, void* ws, size_t ws_bytes, rc_stream_t stream);

/* The same walk without an optimizer: out[row, :] = the summed gradient row of every LISTED row (other rows of `out` are
 * left as they are) -- aten::embedding_dense_backward's index_add (helpers/BaseRunner.py:205) as a plan consumer, and the
 * "sum rows by inverse index" of the sharded steps' de-duplicated exchanges.  Gradient sources as in rc_plan_update;
 * workspace rc_plan_update_workspace_bytes(n_occ, d).                                                           */
int rc_plan_row_sums(float* out, int d, const rc_plan_row* rows, const uint32_t* n_rows, const uint32_t* occ,
                     int64_t n_occ, const float* coef, const float* src, const int64_t* src_index, int div,
                     const float* src2, int64_t n_split, void* ws, size_t ws_bytes, rc_stream_t stream);

/* The distinct ids of a planned list and the inverse index -- what torch.unique(ids, return_inverse=True) returns, minus
 * the sort (the order of uniq is the plan's record order) and minus the host round trip (the count stays in *n_rows):
 * uniq[r] = id of record r (capacity n_list), inverse[p - occ_base] = r for every position p of record r; occ_base /
 * n_list: first position and length of the list (list a: 0, n_a; list b: n_a, n_b).                               */
int rc_plan_distinct(const rc_plan_row* rows, const uint32_t* n_rows, const uint32_t* occ, int64_t occ_base,
                     int64_t n_list, int64_t* uniq, int64_t* inverse, rc_stream_t stream);

size_t rc_bprmf_step_workspace_bytes(int B, int C, int d);

/* Which pipeline rc_bprmf_train_step uses: 0 = automatic -- batches of at most 32,768 row ids (B (1 + K) + B, e.g. the
 * reference's default --batch_size 256, helpers/BaseRunner.py:33) take the two-launch small-batch step (ids grouped by
 * 128 workgroups beside the fused forward / backward, then one update launch); larger ones the bucket plan where
 * supported, its per-bucket pass -- row records and grouped positions, needed only by the updates -- on a
 * library-owned second stream behind the fused kernel, forked and joined by events, so the call stays capturable;
 * 1 = always the radix-sort pipeline, 2 = bucket plan on the caller's stream only, 3 = bucket plan on two streams
 * whatever the batch size.  1, 2, 3 give bit-identical tables; 0 equals them bit for bit on batches without rows of
 * more than 32 occurrences and to fp32 summation order otherwise (parity tests, A/B timing).  Returns the previous
 * setting; any other `mode` only queries.  Process-wide, initial value 0 (1 / 2 / 3 when the environment has
 * RC_BPRMF_STEP=sort / serial / plan).                                                                      */
int rc_bprmf_step_pipeline(int mode);

/* One BaseRunner.fit iteration for BPRMF (helpers/BaseRunner.py:193-206 with
 * models/general/BPRMF.py:34-45 and models/BaseModel.py:182-185), row-wise optimizer:
 * bucket plan of the item + user ids (rc_bucket_plan; joint radix sort + segment heads where the id space
 * is too wide for it) -> fused fwd/loss/bwd (+ update of single-occurrence item rows) -> update of the
 * remaining item rows -> update of the user rows (+ loss mean).
 * loss_out[0] = mean_b loss (device float).  pred may be NULL.
 * state tables (mU,vU,mI,vI) may be NULL for SGD.
 * phase_ms: NULL, or a HOST float[8] filled with per-phase milliseconds measured with
 * hipEvents on `stream` (the call then synchronises):
 *   [0] sort item ids / partition into buckets [1] sort user ids (0: sorted jointly) [2] fused fwd/bwd
 *   [3] loss mean (0 when folded into the last update launch) [4] item-row update [5] user-row update
 *   [6] total [7] segment heads / per-bucket grouping + singleton flags                  */
int rc_bprmf_train_step(float* U, float* I, float* mU, float* vU, float* mI, float* vI,
                        const int64_t* uid, const int64_t* iid, int B, int C, int d,
                        int64_t n_users, int64_t n_items, const rc_opt_hyper* h,
                        float inv_b, float* loss_out, float* pred, void* ws,
                        size_t ws_bytes, rc_stream_t stream, float* phase_ms);

/* ---- look-ahead across steps ---------------------------------------------------------------------------------
 * The reference's loop knows the following batch while it trains on the current one (its DataLoader runs ahead of
 * helpers/BaseRunner.py:186).  rc_bprmf_train_step_ahead uses that: given the ids of the FOLLOWING call it enqueues
 * that batch's complete bucket plan (histogram, stable partition, multi-occurrence bitmap, row records, grouped
 * positions) on the library's second stream beside THIS step's row updates, into the second plan slot of the same
 * workspace, so that the following call starts directly with its fused kernel.
 *
 * What was prepared, for which batch, is recorded in a CALLER-OWNED ticket -- the library keeps no batch state.
 * Batches are identified by the caller's generation ids (any non-zero number that changes whenever the CONTENTS of
 * the id buffers change: a running batch counter), never by pointer: refilling the same buffers in place with a new
 * batch and a new generation simply misses, and the step plans that batch itself.  Zero-initialise the ticket; one
 * ticket per workspace; while a ticket holds a prepared plan (generation != 0) the workspace must not be used by
 * calls that do not pass this ticket, and next_uid / next_iid must stay alive and unchanged until consumed.        */
typedef struct rc_step_ticket {
  uint64_t generation; /* generation id of the batch whose plan is prepared in the workspace; 0 = none */
  uint64_t ws;         /* address of the workspace it was written into                               */
  int32_t slot;        /* plan slot (0 / 1)                                                          */
  int32_t device;      /* HIP device of the side stream that wrote it                                */
  int32_t B, C, d;     /* geometry it was prepared for                                               */
  int32_t flavour;     /* 1: bitmap + multi-occurrence rows (SGD fast path), 2: every row listed      */
  int64_t n_users, n_items;
  uint64_t reserved[2];
} rc_step_ticket;

/* rc_bprmf_train_step with the look-ahead.  generation: id of THIS batch (0 = unknown: never matches a ticket);
 * next_uid [B] / next_iid [B, C] / next_generation: the following call's batch (same shapes, same workspace), or
 * NULL / 0 for none.  A ticket prepared for exactly (generation, workspace, geometry, optimizer class) is consumed;
 * anything else is discarded (after waiting for the side stream) and the step plans its batch itself.  Results are
 * bit-identical to rc_bprmf_train_step in every case.  No look-ahead is PREPARED under stream capture or for batches
 * that take the small-batch / sort pipeline.  With phase_ms the phases are those of a steady-state step (the plan of
 * the following batch runs on the second stream beside them).                                                      */
int rc_bprmf_train_step_ahead(float* U, float* I, float* mU, float* vU, float* mI, float* vI,
                              const int64_t* uid, const int64_t* iid, uint64_t generation,
                              const int64_t* next_uid, const int64_t* next_iid, uint64_t next_generation,
                              rc_step_ticket* ticket, int B, int C, int d, int64_t n_users, int64_t n_items,
                              const rc_opt_hyper* h, float inv_b, float* loss_out, float* pred, void* ws,
                              size_t ws_bytes, rc_stream_t stream, float* phase_ms);
/* Forget a plan prepared by rc_bprmf_train_step_ahead (call before the workspace it was written into is freed or
 * re-allocated): `stream` waits for the second stream's writes into that workspace; the ticket is cleared.        */
int rc_bprmf_step_ahead_reset(rc_step_ticket* ticket, rc_stream_t stream);

/* ---- measurement aid (bench.py only; not a step of the path) ------------------------------------------------------------------
 * The access mix of the fused BPRMF kernel without its arithmetic, on the caller's table and id list: every occurrence reads
 * its row (d in {32, 64, 128} floats, non-temporal, 8 rows in flight per lane group), the pseudo-random fraction write_frac of
 * the occurrences writes the row back unchanged (the table keeps its contents).  Runs 3 + iters launches on `stream`, times
 * the iters with HIP events, SYNCHRONISES, and returns the average milliseconds per launch in *ms_out (host memory).  bench.py
 * prints the rate as roofline.box_ceiling_gbps: what this box delivers for the kernel's own ids, next to what the kernel
 * reaches.  sink_dev: 4 device bytes (a store that never happens keeps the loads alive).                                     */
int rc_bench_mix(float* table, int d, const int64_t* ids, int64_t n_occ, float write_frac, int iters, float* sink_dev,
                 float* ms_out, rc_stream_t stream);
/* The matrix pipes' sustained fp32 rate on this box (v_mfma_f32_32x32x2_f32 back to back on every SIMD, two waves each, no operand
 * traffic; `iters` x 16 MFMAs per wave, one warm-up launch, one timed launch; SYNCHRONISES) -> *tflops_out (host memory); bench.py
 * prints it as roofline.box_mfma_tflops beside the datasheet's 157.3 TFLOP/s for the MFMA-bound legs.  sink_dev: 4 device bytes. */
int rc_bench_mfma(int iters, float* sink_dev, float* tflops_out, rc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RECHORUS_HIP_H */
