/*
 * hiprec.h — C ABI of libhiprec.so, the MI355X (gfx950) embedding-table training hot path
 * for beta-recsys models (MF / GMF / MLP / NeuMF / LightGCN).
 *
 * The reference (beta-team/beta-recsys v0.3.2) is pure Python/PyTorch and has NO FFI of its own;
 * the boundary it offers is the duck-type of beta_rec/models/torch_engine.py:6-121 (ModelEngine).
 * Every entry point below replaces one stretch of PyTorch ops issued by that engine family; the
 * reference file:line each one stands in for is cited on the declaration.  INTEGRATION.md shows the
 * ctypes stub a beta-recsys maintainer would add.
 *
 * Conventions
 *   - every function returns int: 0 = ok, >0 = hipError_t, <0 = HIPREC_E_*;
 *     hiprec_last_error() returns a thread-local, human-readable message for the last failure.
 *   - no C++ exceptions cross the ABI, no torch types in signatures: plain pointers and sizes.
 *   - every pointer is a DEVICE pointer on the current HIP device unless the name ends in _host.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises.
 *   - the library allocates NOTHING: tables, gradient accumulators, optimizer state, row stamps and
 *     the scratch block are owned by the caller (PyTorch tensors in the Python host layer).
 *   - floats are fp32, indices are int64 (torch.LongTensor, beta_rec/data/base_data.py:247-251).
 *   - index validation: an out-of-range index never touches memory; the triple/sample is skipped
 *     and a bit is OR-ed into the device status word (HIPREC_STATUS_*), which the host layer turns
 *     into IndexError (PyTorch's behaviour for nn.Embedding).
 */
#ifndef HIPREC_H
#define HIPREC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPREC_VERSION 100 /* 0.1.0 */

/* negative library error codes */
#define HIPREC_E_BADARG (-1)   /* null pointer, negative size, unsupported dim ... */
#define HIPREC_E_SCRATCH (-2)  /* scratch block too small */
#define HIPREC_E_UNSUPPORTED (-3)

/* bits of the device status word */
#define HIPREC_STATUS_USER_OOB 1u
#define HIPREC_STATUS_ITEM_OOB 2u
#define HIPREC_STATUS_ROW_OOB 4u
#define HIPREC_STATUS_ROUTE_OVERFLOW 8u /* a fixed-capacity all-to-all bucket was too small */
#define HIPREC_STATUS_NEG_EXHAUSTED 16u /* a user has fewer untouched items than negatives were asked for */
#define HIPREC_STATUS_LAZY_TABLE 32u    /* lazy Adam: a step beyond the scalars table whose bias corrections still move */
#define HIPREC_STATUS_TABLE_FULL 64u    /* a batch's contribution hash partition overflowed: some row's gradient parts were
                                         * not listed (hiprec_batch_row_contrib flags the batch, the step that uses it raises) */

/* optimizer kinds, beta_rec/models/torch_engine.py:23-39 (only `lr` is ever set there) */
#define HIPREC_OPT_SGD 0
#define HIPREC_OPT_ADAM 1
#define HIPREC_OPT_RMSPROP 2

/* The five parameter tensors of beta_rec/models/mf.py:21-25 (MF.__init__), or any buffer set of
 * the same shape (gradient accumulators).  Row-major fp32. */
typedef struct hiprec_mf_tables {
  float* user_emb;    /* [n_users, dim] */
  float* item_emb;    /* [n_items, dim] */
  float* user_bias;   /* [n_users]      (nn.Embedding(n_users, 1)) */
  float* item_bias;   /* [n_items] */
  float* global_bias; /* [1] */
  int64_t n_users;
  int64_t n_items;
  int32_t dim;
  int32_t _pad;
} hiprec_mf_tables;

/* Device-resident step statistics, one per engine.  All fields live on the device so a whole epoch
 * can be enqueued without a host round trip (the reference syncs twice per step for
 * loss.item()/regularizer.item(), beta_rec/models/mf.py:119). */
typedef struct hiprec_stats {
  float loss;        /* last step: BPR / BCE loss (mean over the batch) */
  float reg;         /* last step: MF regularizer, mf.py:49-54 summed over both forward calls */
  double loss_sum;   /* += loss every step (train_an_epoch's total_loss, mf.py:134-139) */
  double reg_sum;    /* += reg every step */
  int64_t step;      /* optimizer step counter t; every *_grad call advances it by one */
  double beta1;      /* Adam betas (set by hiprec_stats_reset) and their running powers */
  double beta2;      /*   beta^t, kept on the device so a captured graph of steps stays valid */
  double beta1_pow;
  double beta2_pow;
  uint32_t status;   /* HIPREC_STATUS_* bits, sticky until cleared by the host */
  uint32_t _pad;
} hiprec_stats;

int hiprec_version(void);
/* sha256 (hex) over the sources the library was built from, "unknown" for a build that was not told (csrc/util.hip);
 * the measurement evidence under profiles/ carries the hash of the build it was taken with. */
const char* hiprec_source_hash(void);
const char* hiprec_last_error(void);
size_t hiprec_stats_bytes(void);            /* sizeof(hiprec_stats) for the host layer */
/* bytes of scratch needed by any *_grad / *_step call on a batch of `batch` units */
size_t hiprec_scratch_bytes(int64_t batch);
/* reset stats to {loss 0, reg 0, sums 0, step 0, betas as given, powers 1, status 0} */
int hiprec_stats_reset(hiprec_stats* stats, double beta1, double beta2, void* stream);
/* t <- t+1 without a *_grad call (stand-alone use of hiprec_opt_dense_step) */
int hiprec_stats_advance_step(hiprec_stats* stats, void* stream);
/* resume: t <- step and the Adam bias-correction powers beta**step (host pow(), as torch.optim computes
 * them) in ONE launch; loss / sums / status are left alone */
int hiprec_stats_set_step(hiprec_stats* stats, int64_t step, double beta1, double beta2, void* stream);
/* zero only the epoch accumulators loss_sum/reg_sum (start of train_an_epoch, mf.py:131-132) */
int hiprec_stats_begin_epoch(hiprec_stats* stats, void* stream);

/* ---- bit-exact row gather: out[k,:] = table[idx[k],:]  (nn.Embedding.forward, mf.py:39-42;
 *      ncf.py:54-57; lightgcn.py:134-136).  dim floats per row.  idx == -1 is a padding slot of a
 *      fixed-capacity exchange: it yields a zero row (hiprec_scatter_add_rows skips it). */
int hiprec_gather_rows(const float* table, int64_t n_rows, int32_t dim, const int64_t* idx,
                       int64_t n, float* out, hiprec_stats* stats, void* stream);

/* ---- bucketing for the fixed-capacity all-to-all of the row-sharded engine:
 *      slot_out[k] = d*cap + (arrival position of key k in bucket d), d = keys[k] mod n_dest;
 *      negative keys are padding (slot -1); counts[d] receives the bucket sizes (zeroed by the
 *      call); a full bucket sets HIPREC_STATUS_ROUTE_OVERFLOW and yields slot -1. */
int hiprec_route_bucket(const int64_t* keys, int64_t n, int32_t n_dest, int64_t cap, int32_t* counts,
                        int64_t* slot_out, hiprec_stats* stats, void* stream);

/* ---- fused routing / packing for the fixed-capacity all-to-alls of the row-sharded engine
 *      (owner(row) = row mod n_dest, bucket d = slots [d*cap, (d+1)*cap), padding -1,
 *      a full bucket sets HIPREC_STATUS_ROUTE_OVERFLOW and drops the entry):
 *  route_triples   A2A-1 send buffer send[n_dest*cap][3] = (user, pos, neg) bucketed by owner(user);
 *  route_items     from the received triples recv[n_slots][3]: A2A-2 request buffer req[n_dest*cap]
 *                  = item ids bucketed by owner(item), slot_pos / slot_neg[n_slots] = the slots the
 *                  rows of each triple come back in, u_loc[n_slots] = local user row (-1 = padding);
 *  gather_payload  owner side: payload[n][dim+1] = [item_emb row | item_bias] of incoming[k] / n_dest
 *                  (zeros for padding), local_idx[n] = that local row or -1;
 *  split / join    [n][dim+1] <-> ([n][dim], [n]) around the gradient kernel.
 *      counts[n_dest] is a caller-owned int32 workspace (bucket fill levels). */
int hiprec_shard_route_triples(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t n,
                               int32_t n_dest, int64_t cap, int32_t* counts, int64_t* send,
                               hiprec_stats* stats, void* stream);
int hiprec_shard_route_items(const int64_t* recv, int64_t n_slots, int32_t n_dest, int64_t cap,
                             int32_t* counts, int64_t* req, int64_t* slot_pos, int64_t* slot_neg,
                             int64_t* u_loc, hiprec_stats* stats, void* stream);
int hiprec_shard_gather_payload(const float* item_emb, const float* item_bias, int64_t n_rows,
                                int32_t dim, const int64_t* incoming, int64_t n, int32_t n_dest,
                                float* payload, int64_t* local_idx, hiprec_stats* stats, void* stream);
int hiprec_shard_split_rows(const float* src, int64_t n, int32_t dim, float* emb, float* bias,
                            void* stream);
int hiprec_shard_join_rows(const float* emb, const float* bias, int64_t n, int32_t dim, float* dst,
                           void* stream);

/* ---- table[idx[k], :] += src[k, 0:dim]  (src rows are src_stride floats apart).  Owner-side
 *      accumulation of the gradient rows that come back through the all-to-all of the row-sharded
 *      engine; the single-process reference does this inside
 *      embedding_dense_backward (mf.py:117). */
int hiprec_scatter_add_rows(float* table, int64_t n_rows, int32_t dim, const int64_t* idx,
                            const float* src, int64_t src_stride, int64_t n, hiprec_stats* stats,
                            void* stream);

/* ---- MF forward for scoring: MF.predict / MF.forward under no_grad (mf.py:32-48, 57-70).
 *      scores[k] = sigmoid(<U[u_k], I[i_k]> + bu[u_k] + bi[i_k] + g) */
int hiprec_mf_predict(const hiprec_mf_tables* w, const int64_t* users, const int64_t* items,
                      int64_t n, float* scores, hiprec_stats* stats, void* stream);
/* MF.forward (beta_rec/models/mf.py:32-55) in one launch: the scores of hiprec_mf_predict and, per sample, sq[k] =
 * |U[u]|^2 + |I[i]|^2 + bu^2 + bi^2 -- the regularizer is sum(sq) / n (the reference squares four gathered tensors). */
int hiprec_mf_forward(const hiprec_mf_tables* w, const int64_t* users, const int64_t* items, int64_t n, float* scores,
                      float* sq, hiprec_stats* stats, void* stream);

/* ---- MF BPR forward + backward (mf.py:101-107,116-117; torch_engine.py:104-105).
 * Accumulates the DENSE gradient of the batch-mean BPR loss into `g` (same layout as `w`; the
 * caller guarantees it is zero on entry, exactly like optimizer.zero_grad()+backward()).
 * A triple whose user index is exactly -1 is padding (fixed-capacity all-to-all) and is skipped
 * silently.  Triple k of the batch is (users[j], pos[j], neg[j]) with j = perm ? perm[k] : k — the optional
 * permutation is the device-side batcher replacing DataLoader(shuffle=True) (base_data.py:253).
 * Writes per-block partial (loss, reg) sums to scratch; the optimizer call that follows (or
 * hiprec_finalize_stats) reduces them into stats.  inv_batch is 1/B of the (global) batch;
 * reg_coef is MFEngine.reg (mf.py:81-83,116 — always 0.0 in the reference).  */
int hiprec_mf_bpr_grad(const hiprec_mf_tables* w, const hiprec_mf_tables* g, const int64_t* users,
                       const int64_t* pos, const int64_t* neg, const int64_t* perm, int64_t batch,
                       float inv_batch, float reg_coef, hiprec_stats* stats, void* scratch,
                       size_t scratch_bytes, void* stream);

/* ---- MF BCE forward + backward (mf.py:108-111; torch_engine.py:108-121, BCELoss mean, log
 * clamped at -100 like PyTorch).  ratings fp32. */
int hiprec_mf_bce_grad(const hiprec_mf_tables* w, const hiprec_mf_tables* g, const int64_t* users,
                       const int64_t* items, const float* ratings, const int64_t* perm,
                       int64_t batch, float inv_batch, float reg_coef, hiprec_stats* stats,
                       void* scratch, size_t scratch_bytes, void* stream);

/* Reduce the per-block partials left in scratch by the last *_grad call into stats
 * (loss, reg, loss_sum += loss, reg_sum += reg) and add the scalar-bias gradient they carry to
 * *g_scalar (the global_bias slot of `g`; may be NULL); loss_reg_out (may be NULL) additionally
 * receives {loss, reg} as two floats (the data-parallel engine appends them to the gradient buffer
 * it all-reduces).  Only needed when no optimizer call follows: hiprec_opt_dense_step /
 * hiprec_mf_sgd_rows do the same reduction themselves. */
int hiprec_finalize_stats(hiprec_stats* stats, const void* scratch, float* g_scalar,
                          float* loss_reg_out, void* stream);

/* ---- dense optimizer step over one flat fp32 buffer (torch.optim.{SGD,Adam,RMSprop}.step with the
 * defaults torch_engine.py:23-39 leaves in place: Adam betas (0.9,0.999) eps 1e-8, RMSprop alpha
 * 0.99 eps 1e-8, no momentum / weight decay).  Applies the update to w[0:n], updates the state
 * (m = exp_avg, v = exp_avg_sq / square_avg; may be NULL for kinds that do not use them) and
 * ZEROES g[0:n] (the next step's zero_grad).  Uses t = stats->step (already advanced by the
 * preceding *_grad call) and the running beta powers in stats (Adam: beta1/beta2 must equal the
 * values given to hiprec_stats_reset; RMSprop: beta2 = alpha).  Hyper-parameters are doubles, as
 * the python floats of the reference are, and are rounded to fp32 exactly where ATen rounds them.
 * When scratch != NULL it also finalizes the partials of the preceding *_grad call; the gradient
 * of the scalar parameter at w[scalar_index] (MF's global_bias; -1 = none) arrives through those
 * partials and is added before that element is updated. */
int hiprec_opt_dense_step(int kind, float* w, float* g, float* m, float* v, int64_t n, double lr,
                          double beta1, double beta2, double eps, hiprec_stats* stats,
                          const void* scratch, int64_t scalar_index, void* stream);

/* ---- exact SGD restricted to the rows a batch touched (SGD with momentum 0 leaves every other
 * row bit-identical: torch_engine.py:26-29).  For each distinct row of the batch:
 * w[row] -= lr*g[row]; g[row] = 0.  `user_stamp` [n_users] / `item_stamp` [n_items] are int32
 * arrays owned by the caller, zero-initialised once; `stamp` must be a value never used before
 * for these arrays (the host passes a running step counter starting at 1).  `users` or `items_a`
 * may be NULL and any entry may be -1 ("no row of that table for this entry": the row-sharded engine
 * passes the user rows it received and the item rows its peers fetched as two separate lists); any
 * other out-of-range index marks a triple the gradient kernel flagged and skipped.  */
int hiprec_mf_sgd_rows(const hiprec_mf_tables* w, const hiprec_mf_tables* g, const int64_t* users,
                       const int64_t* items_a, const int64_t* items_b, const int64_t* perm,
                       int64_t batch, double lr, int32_t* user_stamp, int32_t* item_stamp,
                       int32_t stamp, hiprec_stats* stats, const void* scratch, void* stream);

/* ---- device-side shuffle: out[i] = P_seed(i), a pseudo-random bijection of [0, n) (Feistel network
 * with cycle walking; stateless, no sort).  The epoch order of the device batcher, in place of
 * RandomSampler's torch.randperm (torch/utils/data/sampler.py, used by base_data.py:253). */
int hiprec_random_permutation(int64_t* out, int64_t n, uint64_t seed, void* stream);

/* ---- device-side batcher: lay one epoch out in visiting order.  Batch b = triples
 * perm[b*batch .. (b+1)*batch) (perm NULL = sequential; last batch short), sorted by item id
 * inside the batch, written contiguously to out_*.  `third` is the negative-item array (int64,
 * third_bytes 8) or the rating array (fp32, third_bytes 4).  Replaces
 * DataLoader(PairwiseNegativeDataset|RatingDataset, shuffle=True) + default_collate
 * (data/base_data.py:206-216,247-253; data/data_loaders.py:4-53).  batch <= 8192. */
int hiprec_stage_epoch(const int64_t* users, const int64_t* items, const void* third,
                       int32_t third_bytes, const int64_t* perm, int64_t n, int64_t batch,
                       int64_t* out_users, int64_t* out_items, void* out_third, void* stream);
/* The same with the shuffle folded in: the visiting order is P_seed of hiprec_random_permutation, evaluated on
 * the fly (no perm[] array, one launch per epoch).  Bit-identical to hiprec_random_permutation(seed) followed
 * by hiprec_stage_epoch(perm). */
int hiprec_stage_epoch_shuffled(const int64_t* users, const int64_t* items, const void* third,
                                int32_t third_bytes, uint64_t seed, int64_t n, int64_t batch,
                                int64_t* out_users, int64_t* out_items, void* out_third, void* stream);

/* ---- one whole epoch of MF-BPR training enqueued back to back (MFEngine.train_an_epoch,
 * mf.py:121-139, with the DataLoader replaced by perm[] slices of the resident triple arrays;
 * the last batch is short, drop_last=False as base_data.py:253).  `flat_*` are the flat buffers
 * that hold all five tensors of w / g / state contiguously (n_flat floats) when the optimizer is
 * dense; for kind == SGD with user_stamp != NULL the touched-rows path is used instead.
 * first_stamp.. first_stamp+n_batches-1 are consumed as stamps.  */
int hiprec_mf_bpr_epoch(const hiprec_mf_tables* w, const hiprec_mf_tables* g, const int64_t* users,
                        const int64_t* pos, const int64_t* neg, const int64_t* perm,
                        int64_t n_triples, int64_t batch, float reg_coef, int kind, double lr,
                        double beta1,
                        double beta2, double eps, float* flat_w, float* flat_g, float* flat_m,
                        float* flat_v, int64_t n_flat, int32_t* user_stamp, int32_t* item_stamp,
                        int32_t first_stamp, hiprec_stats* stats, void* scratch,
                        size_t scratch_bytes, void* stream);

/* ---- the same for loss == "bce" (mf.py:108-111): resident (user, item, rating) arrays, the
 * RatingDataset loader of data/base_data.py:182-216 replaced by perm[] slices / a staged layout. */
int hiprec_mf_bce_epoch(const hiprec_mf_tables* w, const hiprec_mf_tables* g, const int64_t* users,
                        const int64_t* items, const float* ratings, const int64_t* perm,
                        int64_t n_samples, int64_t batch, float reg_coef, int kind, double lr,
                        double beta1, double beta2, double eps, float* flat_w, float* flat_g,
                        float* flat_m, float* flat_v, int64_t n_flat, int32_t* user_stamp,
                        int32_t* item_stamp, int32_t first_stamp, hiprec_stats* stats,
                        void* scratch, size_t scratch_bytes, void* stream);

/* ---- one epoch of BPR-MF, ONE kernel per step (mf.py:121-139 around mf.py:92-119 with
 * torch_engine.py:23-39's optimizer).  torch's SGD / Adam / RMSprop updates are elementwise
 * functions of (w, g, m, v), so the update of step k-1 rides inside the gradient kernel of step k:
 * gather blocks evaluate update(W_a, G_prev, M_a, V_a) on the fly for the rows they read (nothing
 * they read is written by the launch, so every gradient of the batch still comes from the pre-step
 * weights, as in mf.py:101-118) while the other blocks of the same grid write W_b, M_b, V_b for the
 * whole buffer.  kind = HIPREC_OPT_*.  w_flat[2], g_flat[3], m_flat[2] (Adam), v_flat[2] (Adam,
 * RMSprop; NULL otherwise) are flat buffers laid out like hiprec_mf_tables (user_emb|item_emb|
 * user_bias|item_bias|global_bias), scratch2[2] two scratch blocks; all caller-owned.  On entry the
 * state is in w/m/v_flat[0] and all g / scratch buffers are zero; on return it is in
 * w/m/v_flat[*final_index] (always 0: the sweep-only flush that ends the epoch updates in place
 * into buffer 0) and every g buffer / scratch header is zero again.  users/pos/neg hold
 * the epoch in visiting order (n_triples, last batch short).  The arithmetic per element is that
 * of hiprec_opt_dense_step, so results equal hiprec_mf_bpr_epoch's up to the order of the atomic
 * gradient sums. */
int hiprec_mf_bpr_epoch_fused(int kind, float* const* w_flat, float* const* g_flat,
                              float* const* m_flat, float* const* v_flat, void* const* scratch2,
                              int64_t n_users, int64_t n_items, int32_t dim, const int64_t* users,
                              const int64_t* pos, const int64_t* neg, int64_t n_triples,
                              int64_t batch, float reg_coef, double lr, double beta1, double beta2,
                              double eps, hiprec_stats* stats, int32_t* final_index, void* stream);
/* Steps [step_begin, step_end) of that epoch (same arguments, same buffers from call to call): an epoch
 * enqueued in pieces, e.g. to put a timestamp between them.  The call whose step_end is the epoch's step
 * count also enqueues the flush (*final_index = 0); before that the state is mid-rotation (*final_index =
 * -1) and must not be read.  step_begin = 0 starts the epoch (epoch sums reset). */
int hiprec_mf_bpr_epoch_fused_range(int kind, float* const* w_flat, float* const* g_flat,
                                    float* const* m_flat, float* const* v_flat, void* const* scratch2,
                                    int64_t n_users, int64_t n_items, int32_t dim, const int64_t* users,
                                    const int64_t* pos, const int64_t* neg, int64_t n_triples,
                                    int64_t batch, int64_t step_begin, int64_t step_end, float reg_coef,
                                    double lr, double beta1, double beta2, double eps, hiprec_stats* stats,
                                    int32_t* final_index, void* stream);
/* The data-parallel spelling (replicated.py: ReplicatedMFEngine): after every step's fused launch, ONE in-place
 * sum all-reduce of that step's [loss partials | gradient] is enqueued on the same stream through the caller's
 * collective -- all_reduce_fn is the address of ncclAllReduce (RCCL; this library does not link it), comm a
 * communicator of `world` ranks that all make this call with equal n_triples / batch.  bufs[3]: the rotating
 * buffers, each [scratch_floats of scratch block | n_flat of gradient], zero on entry of step 0; gradients are
 * scaled by 1 / (batch * world).  Everything else as hiprec_mf_bpr_epoch_fused_range. */
int hiprec_mf_bpr_dp_epoch_fused_range(int kind, float* const* w_flat, float* const* bufs, int64_t scratch_floats,
                                       float* const* m_flat, float* const* v_flat, int64_t n_flat, int64_t n_users,
                                       int64_t n_items, int32_t dim, const int64_t* users, const int64_t* pos,
                                       const int64_t* neg, int64_t n_triples, int64_t batch, int64_t step_begin,
                                       int64_t step_end, int32_t world, float reg_coef, double lr, double beta1,
                                       double beta2, double eps, hiprec_stats* stats, void* all_reduce_fn,
                                       void* comm, int32_t* final_index, void* stream);

/* ---- plain SGD on tables that do not fit the caches (configs[3]): ONE launch per step, no dense gradient
 * buffer, every touched row written once, in place (csrc/mf_owned.hip).  The caller's batcher supplies, for
 * every triple of the epoch (laid out in visiting order, every batch sorted by positive item) and each of
 * its three rows, own_u / own_p / own_n[n_triples] (int32): -1 (or a slot whose total is 1) when the row occurs ONCE in its batch, else a slot
 * id in [0, total_stride) with total[step * total_stride + slot] = the row's number of occurrences in that
 * batch (a pos and a neg occurrence of one item both count; user rows and item rows use distinct slots).
 * arrived[n_slots] (int32) and acc[n_slots * (dim + 1)] (fp32) are zero on entry and zero again on return;
 * gb_pingpong[2], scratch2[2] are caller-owned work space.  Steps [step_begin, step_end) of the epoch are
 * enqueued; the call that reaches the last step also enqueues the flush (scalar bias + stats of the last
 * step), before that the scalar bias element of w_flat is stale.  Semantics: mf.py:92-119 with
 * torch.optim.SGD(lr) -- gradients from the pre-step weights, untouched rows bit-identical. */
int hiprec_mf_bpr_epoch_owned(float* w_flat, int64_t n_users, int64_t n_items, int32_t dim,
                              const int64_t* users, const int64_t* pos, const int64_t* neg,
                              const int32_t* own_u, const int32_t* own_p, const int32_t* own_n,
                              const int32_t* total, int64_t total_stride,
                              int32_t* arrived, float* acc, int64_t n_slots, float* gb_pingpong,
                              void* const* scratch2, int64_t n_triples, int64_t batch,
                              int64_t step_begin, int64_t step_end, float reg_coef, double lr,
                              hiprec_stats* stats, void* stream);

/* Row ownership of a staged epoch for hiprec_mf_bpr_epoch_owned (csrc/ownership.hip): one hash table of
 * 2^table_bits entries per batch (hiprec_ownership_table_bits(batch): >= 4 x batch) built in LDS, the table
 * position of a row IS its slot, so total_stride = n_slots = 2^table_bits.  ws is work space of
 * hiprec_ownership_ws_ints(n, batch, table_bits) int32s (the occurrences' keys bucketed by table partition); outputs:
 * own[3 * n] (role-major: user, positive, negative row of every triple; its three thirds are the own_u / own_p /
 * own_n of the step) = the row's slot, and total[(batch index << table_bits) + slot] = its occurrences in that
 * batch (1 for a row that occurs once: the step treats it like own = -1; 0 for unused entries).  Triples with
 * an out-of-range id get -1.  Integer work, no sort, nothing read back by the host. */
int32_t hiprec_ownership_table_bits(int64_t batch);   /* <= 24: batches of up to 4 M triples */
int64_t hiprec_ownership_ws_ints(int64_t n, int64_t batch, int32_t table_bits);
int hiprec_batch_row_ownership(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t n,
                               int64_t batch, int64_t n_users, int64_t n_items, int32_t table_bits,
                               int32_t* ws, int32_t* total, int32_t* own, void* stream);

/* ---- round 5: batches beyond hiprec_stage_epoch's LDS sort (configs[3]: 65 536 triples), staged WITHOUT a device sort
 * (csrc/ownership.hip).  What hiprec_stage_sort_keys + a radix sort + hiprec_gather_epoch produced -- the epoch in
 * visiting order (perm[] | the Feistel shuffle of `seed`, evaluated on the fly | sequential), every batch sorted by
 * positive item -- as a two-level counting sort in three launches: ranges of 4096 item ids, then the items of a range
 * counted in LDS; the gather of the three arrays is folded into the last launch.  The order inside a group of equal
 * items is unspecified.  ws: hiprec_stage_grouped_ws_ints(n, batch, n_items) int32s (0 = n_items beyond 8 M: not
 * supported, keep the sort).  Replaces, like hiprec_stage_epoch, DataLoader(shuffle=True)'s batching
 * (beta_rec/data/base_data.py:247-253). */
int64_t hiprec_stage_grouped_ws_ints(int64_t n, int64_t batch, int64_t n_items);
int hiprec_stage_epoch_grouped(const int64_t* users, const int64_t* pos, const int64_t* neg, const int64_t* perm,
                               int32_t shuffle, uint64_t seed, int64_t n, int64_t batch, int64_t n_items, int32_t* ws,
                               int64_t* users_out, int64_t* pos_out, int64_t* neg_out, void* stream);

/* ---- round 5: the same step as OWNER PULLS -- two launches per step, no float atomics (csrc/mf_owned.hip,
 * csrc/ownership.hip).  Replaces, like hiprec_mf_bpr_epoch_owned, loss.backward() + torch.optim.SGD.step() of
 * beta_rec/models/mf.py:101-118 and models/torch_engine.py:25-29 for tables beyond the caches.
 * hiprec_batch_row_contrib (staging; integer work, weights not read) counts, per batch, the CONTRIBUTIONS to every
 * row: a user or negative occurrence is one; of the positive occurrences only the head of a run of equal items inside
 * one `chunk` of consecutive triples (chunk = hiprec_mf_pull_chunk(dim): what one wave of the step kernel takes; it
 * sums the run itself).  Outputs: cidx[3 * n] (role-major like hiprec_batch_row_ownership's own[]): -1 = the row has
 * this one contribution (its contributor stores w - lr * g in place), >= 0 = where the contribution goes in the
 * step's contribution buffer, -2 = a positive occurrence that rides in its neighbour's run; rows[n_batches][row_cap]
 * records {row key (user row, or n_users + item row), first contribution, contributions, 0} of the rows with several
 * (row_cap >= hiprec_contrib_row_cap(batch, min_contrib); rows with more than 32 contributions are listed from the
 * END of a batch's records); counts[n_batches][4] = {records from the front, records from the end, contributions, -}.
 * min_contrib = 2: as described; 1: every row gets a record and a range, cidx is never -1 (the lazy Adam / RMSprop
 * form hiprec_mf_epoch_lazy_pull: a row's optimizer step is taken by the apply launch).  ws:
 * hiprec_ownership_ws_ints(n, batch, table_bits) int32s.
 * hiprec_mf_bpr_epoch_pull: per step the gradient launch (rows with one contribution updated in place, the others'
 * parts stored to cbuf [3 * batch, dim] / cbias [3 * batch]: work space, never initialised or cleared; cidx_stride =
 * the n hiprec_batch_row_contrib was called with, n_triples <= cidx_stride being the part of that epoch to run) and the apply
 * launch (one wave per record sums its range and stores w - lr * g; its first block books loss / regulariser, steps
 * the scalar bias and the clock).  Every step is complete when its second launch is.  Same semantics as
 * hiprec_mf_bpr_epoch_owned; the order in which a row's contributions are summed is the order of the range. */
int32_t hiprec_mf_pull_chunk(int32_t dim);
int64_t hiprec_contrib_row_cap(int64_t batch, int32_t min_contrib);
int hiprec_batch_row_contrib(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t n, int64_t batch,
                             int64_t n_users, int64_t n_items, int32_t table_bits, int32_t chunk, int32_t min_contrib,
                             int32_t* ws, int32_t* cidx, int32_t* rows, int64_t row_cap, int32_t* counts, void* stream);
int hiprec_mf_bpr_epoch_pull(float* w_flat, int64_t n_users, int64_t n_items, int32_t dim, const int64_t* users,
                             const int64_t* pos, const int64_t* neg, const int32_t* cidx, int64_t cidx_stride,
                             const int32_t* rows, int64_t row_cap, const int32_t* counts, float* cbuf, float* cbias,
                             void* scratch,
                             int64_t n_triples, int64_t batch, int64_t step_begin, int64_t step_end, float reg_coef,
                             double lr, hiprec_stats* stats, void* stream);

/* ---- the row-sharded engine's epoch-planned SGD step (beta-recsys_amd/sharded.py).  Per step and
 * rank: hiprec_shard_gather_payload (rows of the items peers asked for) -> all-to-all -> this call ->
 * hiprec_shard_publish_partials -> all-to-all back -> hiprec_shard_apply_rows + hiprec_shard_finish_step.
 * hiprec_mf_bpr_owned_remote_step is the owned-rows kernel on (local user shard, FETCHED item rows): users[] are
 * local user rows (-1 = padding), pos_slot / neg_slot index the fetched [n_slots, dim + 1] exchange buffer
 * (row | bias), user rows are updated in place (own_u / total / arrived / acc as in hiprec_mf_bpr_epoch_owned, over
 * the triples this rank received), the gradients of the item slots go into g_send (same layout; zero on entry)
 * for the way back.  The loss partials stay in `scratch`; the optimizer clock is not touched. */
int hiprec_mf_bpr_owned_remote_step(float* w_flat, int64_t n_users, int64_t n_items_local, int32_t dim,
                                    const float* fetched, float* g_send, int64_t n_slots, const int64_t* users,
                                    const int64_t* pos_slot, const int64_t* neg_slot, const int32_t* own_u,
                                    const int32_t* own_p, const int32_t* own_n, const int32_t* total,
                                    int32_t* arrived, float* acc, int64_t batch, float inv_batch, float reg_coef,
                                    double lr, hiprec_stats* stats, void* scratch, void* stream);
/* rows extra_rows[0..n_dest) of g_send ([.., dim + 1]) <- this rank's [loss, reg, d loss / d scalar bias] of the
 * step (from the gradient kernel's partials in scratch): the 3-float all-reduce rides in the all-to-all */
int hiprec_shard_publish_partials(const void* scratch, float* g_send, int32_t dim, const int64_t* extra_rows,
                                  int32_t n_dest, void* stream);
/* owner side: item row idx[k] (+ bias) -= lr * g_recv[k] (fp32 atomics; idx -1 = extra row / padding: skipped) */
int hiprec_shard_apply_rows(float* item_emb, float* item_bias, int64_t n_rows, int32_t dim, const int64_t* idx,
                            const float* g_recv, int64_t n, double lr, hiprec_stats* stats, void* stream);
/* sum the peers' extra rows of g_recv -> stats (loss, reg, epoch sums), scalar bias -= lr * its gradient, t <- t+1 */
int hiprec_shard_finish_step(const float* g_recv, int32_t dim, const int64_t* extra_rows, int32_t n_src,
                             float* global_bias, double lr, int32_t first_of_epoch, hiprec_stats* stats,
                             void* stream);

/* ---- round 3: the epoch planner as kernels (csrc/plan.hip), the planned step in three launches, the dense
 * optimizers on it, and a range of steps -- kernels AND exchanges -- enqueued by one call.  The reference has no
 * counterpart (single device, beta_rec/models/mf.py:92-119 is the step that is distributed here); SURVEY.md 8e.
 *
 * Planner.  All ids and positions are 32-bit inside a plan (n_users, n_items, n, n_steps * cap < 2^31); world <= 64.
 *  hiprec_batch_row_ownership_tables  hiprec_batch_row_ownership that also returns the tables: tab_keys[n_batches <<
 *      table_bits] (the row key in every entry: user row, or n_users + item; -1 = empty), pos_cnt (same shape: an ITEM
 *      entry's number of positive occurrences) and occ[3 * n] (role-major like own: the entry's count when the
 *      occurrence arrived; for a positive occurrence its rank among the item's positive occurrences).
 *  hiprec_plan_route_triples  owner(user) = user mod world.  Triple j of the epoch's visiting order (perm[j], or j when
 *      perm is NULL) belongs to step j / batch.  cnt_ds[world * n_steps] (d-major) = triples per destination and
 *      step; send[3 * n] (int32) = (user / world, pos, neg), (destination, step)-ordered, visiting order inside a
 *      group.  tile_ws: world * hiprec_plan_route_tiles(n, batch) ints.  Out-of-range ids raise USER_OOB / ITEM_OOB
 *      in stats->status and the triple is dropped (the host turns the bits into IndexError, like nn.Embedding).
 *  hiprec_plan_place_triples  recv[3 * n_recv]: what the triple exchange delivered, (source, step)-ordered with
 *      recv_cnt[world * n_steps] (source-major) elements per group -> users / pos / neg[n_steps * cap] (int64), step
 *      s in block [s * cap, (s + 1) * cap), sources in rank order, padding user = -1.  group_ws: 2 * world * n_steps + 1
 *      + n_steps ints; its last n_steps ints return the triples per step (hiprec_plan_item_slots' step_fill: with it
 *      only the padding behind a step's live triples is written, not the whole output arrays).
 *  hiprec_plan_item_slots  from the blocks above (users[] = local user rows) and their ownership tables (batch = cap,
 *      n_users = n_users_local, n_items = the GLOBAL item count; own / occ [3][n_steps * cap], tab_keys / pos_cnt
 *      [n_steps << table_bits]; pos_cnt is overwritten): every distinct item of a step gets one slot of the step's
 *      exchange buffer, whose layout is, owner by owner (owner(item) = item mod world), [rows asked of d ..., 1 extra
 *      row]: req_cnt[n_steps][world], its transpose req_ds[world][n_steps], ex_req[n_steps][world] (the extra rows),
 *      n_slots[n_steps], slot_of[n_steps << table_bits] (slot per table entry, -1 for non-items), req_send (the
 *      rows to ask for, item / world, (destination, step)-ordered; send_base[world * n_steps + 1] = group starts) and
 *      the step blocks re-laid GROUPED BY POSITIVE ITEM with slots for items: users_out / pos_slot / neg_slot (int64,
 *      padding -1 / 0 / 0) and own_out[3][n_steps * cap].  ws: hiprec_plan_slot_ws_ints(...) ints.
 *  hiprec_plan_place_requests  incoming[n_in]: the rows peers will ask for, (source, step)-ordered with
 *      in_cnt[world * n_steps] (source-major) -> in_idx[n_in + world * n_steps] packed step by step, sources in rank
 *      order, each followed by one extra row (-1); step_off[n_steps + 1]; extra_pos[n_steps][world].  dup_ws
 *      (optional, [n_steps][(n_rows_local + 31) / 32] uint32) receives one bit per (step, local row) = "more than one
 *      peer asks for this row in this step".
 *  hiprec_plan_item_slots' slot_shared (optional, uint8 [n_steps][slot_stride >= 2 cap + world], with total = the
 *      ownership tables' counts): 1 for the slots that several triples of the step reference and for the extra rows
 *      -- the only rows of the gradient exchange buffer that must be zero when the step starts. */
int hiprec_batch_row_ownership_tables(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t n,
                                      int64_t batch, int64_t n_users, int64_t n_items, int32_t table_bits,
                                      int32_t* ws, int32_t* total, int32_t* own, int32_t* tab_keys, int32_t* pos_cnt,
                                      int32_t* occ, void* stream);
/* Staging of batches beyond hiprec_stage_epoch's 8192-key LDS sort, around ONE device sort of the caller's (the
 * reference's loader collates per batch on the host, data/base_data.py:247-253).  visit(j) = perm[j], or P_seed(j)
 * (perm NULL, shuffle != 0: hiprec_random_permutation's Feistel shuffle evaluated on the fly), or j.
 *  hiprec_stage_sort_keys  keys[j] = (j / batch) * n_items + items[visit(j)] (int32 when key_bytes == 4: needs
 *      n_batches * n_items < 2^31): sorting them groups every batch by item in ascending row order.
 *  hiprec_gather_epoch     out[j] = in[visit(order ? order[j] : j)] for the three arrays in one launch (order = the
 *      sort's permutation; NULL = the plain visiting order). */
int hiprec_stage_sort_keys(const int64_t* items, const int64_t* perm, int32_t shuffle, uint64_t seed, int64_t n,
                           int64_t batch, int64_t n_items, int32_t key_bytes, void* keys, void* stream);
int hiprec_gather_epoch(const int64_t* users, const int64_t* pos, const int64_t* neg, const int64_t* perm,
                        int32_t shuffle, uint64_t seed, const int64_t* order, int64_t n, int64_t* users_out,
                        int64_t* pos_out, int64_t* neg_out, void* stream);
/* A staged epoch re-laid with every batch GROUPED BY POSITIVE ITEM without any sort (groups in hash-table order, not row
 * order: measured 3 % slower per step than the sorted layout at configs[3], so the engines keep the device sort and
 * this stays an option for callers without one): from what
 * hiprec_batch_row_ownership_tables made of the epoch -- own, occ, tab_keys, pos_cnt -- an exclusive scan of the item
 * entries' pos_cnt per batch (pos_cnt is OVERWRITTEN with every item's first position) and one scatter of the triples
 * and their ownership slots.  invalid_cnt[n_batches]: work space.  Triples with an out-of-range id (own = -1) are
 * parked at the end of their batch.  Not in place. */
int hiprec_group_epoch_by_item(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t n, int64_t batch,
                               int64_t n_users, int32_t table_bits, const int32_t* own, const int32_t* occ,
                               const int32_t* tab_keys, int32_t* pos_cnt, int32_t* invalid_cnt, int64_t* users_out,
                               int64_t* pos_out, int64_t* neg_out, int32_t* own_out, void* stream);
int64_t hiprec_plan_route_tiles(int64_t n, int64_t batch);
int hiprec_plan_route_triples(const int64_t* users, const int64_t* pos, const int64_t* neg, const int64_t* perm,
                              int64_t n, int64_t batch, int32_t world, int64_t n_users, int64_t n_items,
                              int32_t* tile_ws, int32_t* cnt_ds, int32_t* send, hiprec_stats* stats, void* stream);
int hiprec_plan_place_triples(const int32_t* recv, int64_t n_recv, const int32_t* recv_cnt, int32_t world,
                              int64_t n_steps, int64_t cap, int32_t* group_ws, int64_t* users, int64_t* pos,
                              int64_t* neg, void* stream);
int64_t hiprec_plan_slot_ws_ints(int64_t n_steps, int32_t table_bits, int32_t world);
int hiprec_plan_item_slots(const int64_t* users, int64_t n_steps, int64_t cap, int32_t world, int64_t n_users_local,
                           int32_t table_bits, const int32_t* own, const int32_t* occ, const int32_t* tab_keys,
                           int32_t* pos_cnt, int32_t* ws, int32_t* slot_of, int32_t* req_cnt, int32_t* req_ds,
                           int32_t* ex_req, int32_t* n_slots, int32_t* send_base, int32_t* req_send,
                           int64_t* users_out, int64_t* pos_slot, int64_t* neg_slot, int32_t* own_out,
                           const int32_t* total, uint8_t* slot_shared, int64_t slot_stride, const int32_t* step_fill,
                           void* stream);
int hiprec_plan_place_requests(const int32_t* incoming, int64_t n_in, const int32_t* in_cnt, int32_t world,
                               int64_t n_steps, int32_t* group_ws, int32_t* in_idx, int32_t* step_off,
                               int32_t* extra_pos, int64_t n_rows_local, uint32_t* dup_ws, void* stream);

/* The same step as OWNER PULLS (round 5; plain SGD, dim % 4 == 0): two launches, no float atomics.  cidx_* / rows / counts:
 * hiprec_batch_row_contrib's arrays of this step (n_users = local user rows, n_items = a bound of the slot ids,
 * min_contrib = 2).  Launch 1 updates in place the user rows whose only contribution a wave holds and stores the
 * gradient of slots it alone references straight into g_send; the other parts go to cbuf [3 * batch][dim] / cbias
 * [3 * batch].  Launch 2 sums every shared row's range (a user row takes w - lr * sum, a slot's sum goes to g_send) and
 * writes [loss, reg, d loss / d global_bias] of this rank into rows extra_rows[0 .. n_dest) of g_send
 * (hiprec_shard_publish_partials is not needed).  g_send needs no clearing; the optimizer clock is not touched. */
int hiprec_mf_bpr_pull_remote_step(float* w_flat, int64_t n_users, int64_t n_items_local, int32_t dim,
                                   const float* fetched, float* g_send, int64_t n_slots, const int64_t* users,
                                   const int64_t* pos_slot, const int64_t* neg_slot, const int32_t* cidx_u,
                                   const int32_t* cidx_p, const int32_t* cidx_n, const int32_t* rows, int64_t row_cap,
                                   const int32_t* counts, float* cbuf, float* cbias, const int32_t* extra_rows,
                                   int32_t n_dest, int64_t batch, float inv_batch, float reg_coef, double lr,
                                   hiprec_stats* stats, void* scratch, void* stream);
/* The planned step in three launches.  Rows [self_lo, self_hi) of a step's incoming block are the ones this rank
 * asked of itself: they never travel (self_dst / g_self point into the fetched / the send buffer).
 *  hiprec_shard_payload_zero  payload[k] = [item_emb[idx[k]] | item_bias[idx[k]]] (zeros for idx -1) AND
 *      zero[0, zero_floats) = 0 (the gradient exchange buffer; with `shared`, one byte per [dim + 1]-float row of it,
 *      only the rows flagged 1) in ONE launch;
 *  hiprec_mf_bpr_grad_remote_step  hiprec_mf_bpr_owned_remote_step for Adam / RMSprop: nothing is updated, the local
 *      user rows' gradients go into g_flat (dense, laid out like w_flat, zero on entry);
 *  hiprec_shard_apply_finish  target row idx[k] += coef * g_recv[k] (SGD: the item table, coef = -lr; dense
 *      optimizers: the dense gradient, coef = 1) AND the step's bookkeeping (the peers' extra rows -> stats,
 *      *scalar_target += scalar_coef * d loss / d scalar bias, t <- t + 1) in ONE launch; dup_bits (optional): one
 *      bit per target row, 0 = only one of this step's rows names it (plain read-modify-write instead of atomics). */
int hiprec_shard_payload_zero(const float* item_emb, const float* item_bias, int64_t n_rows, int32_t dim,
                              const int32_t* idx, int64_t n, int64_t self_lo, int64_t self_hi, float* payload,
                              float* self_dst, float* zero, int64_t zero_floats, const uint8_t* shared,
                              hiprec_stats* stats, void* stream);
int hiprec_mf_bpr_grad_remote_step(const float* w_flat, float* g_flat, int64_t n_users, int64_t n_items_local,
                                   int32_t dim, const float* fetched, float* g_send, int64_t n_slots,
                                   const int64_t* users, const int64_t* pos_slot, const int64_t* neg_slot,
                                   const int32_t* own_u, const int32_t* own_p, const int32_t* own_n,
                                   const int32_t* total, int64_t batch, float inv_batch, float reg_coef,
                                   hiprec_stats* stats, void* scratch, void* stream);
int hiprec_shard_apply_finish(float* t_emb, float* t_bias, int64_t n_rows, int32_t dim, const int32_t* idx,
                              const float* g_recv, int64_t n, int64_t self_lo, int64_t self_hi, const float* g_self,
                              double coef, const int32_t* extra_pos, int32_t n_src, float* scalar_target,
                              double scalar_coef, int32_t first_of_epoch, const uint32_t* dup_bits, hiprec_stats* stats,
                              void* stream);

/* One epoch plan as the step driver reads it (device arrays as the planner wrote them; the *_host arrays are the
 * exact sizes of the exchanges, host integers: nothing is read back while the steps are enqueued). */
typedef struct hiprec_shard_plan {
  int32_t world, rank;
  int64_t n_steps, cap;          /* steps of the epoch; triples per step block (incl. padding) */
  int64_t local_batch, n_local;  /* the loader's batch size and triples per rank (equal on every rank) */
  const int64_t* users;          /* [n_steps * cap] local user rows, -1 = padding */
  const int64_t* pos_slot;       /* [n_steps * cap] slots of the step's fetched buffer */
  const int64_t* neg_slot;
  const int32_t* own;            /* [3][n_steps * cap] */
  const int32_t* total;          /* [n_steps][total_stride] */
  int64_t total_stride;
  const int32_t* in_idx;         /* packed: local item rows peers ask for, -1 = extra row */
  const int32_t* ex_req;         /* [n_steps][world] extra rows of the fetched / send buffer */
  const int32_t* ex_in;          /* [n_steps][world] extra rows of the incoming block */
  const int64_t* in_off_host;    /* [n_steps + 1] */
  const int64_t* n_slots_host;   /* [n_steps] */
  const int64_t* req_cnt_host;   /* [n_steps][world] rows asked of each owner (extra row not counted) */
  const int64_t* in_cnt_host;    /* [n_steps][world] rows each peer asks for */
  const uint8_t* slot_shared;    /* optional [n_steps][slot_stride]: hiprec_plan_item_slots */
  int64_t slot_stride;
  const uint32_t* dup_bits;      /* optional [n_steps][dup_words]: hiprec_plan_place_requests */
  int64_t dup_words;
  /* optional (round 5, plain SGD, dim % 4 == 0): hiprec_batch_row_contrib's arrays over (users, pos_slot, neg_slot) with
   * batch = cap, n_users = the local user rows, n_items = a bound of the slot ids, min_contrib = 2 -- the step then runs
   * as owner pulls (hiprec_mf_bpr_pull_remote_step): no float atomics, no clearing of the gradient exchange buffer, the
   * partials' publish rides in the pull launch */
  const int32_t* cidx;           /* [3][n_steps * cap] */
  const int32_t* rows;           /* [n_steps][row_cap][4] */
  int64_t row_cap;
  const int32_t* counts;         /* [n_steps][4] */
} hiprec_shard_plan;

typedef struct hiprec_shard_bufs {
  float* w_flat;                 /* this rank's [user_emb | item_emb | user_bias | item_bias | global_bias] */
  int64_t n_users_local, n_items_local;
  int32_t dim, _pad;
  float* payload;                /* [max incoming rows][dim + 1] */
  float* g_recv;
  float* fetched;                /* [max slots][dim + 1] */
  float* g_send;
  int32_t* arrived;              /* plain SGD: the owned-rows step's slot counters / accumulators */
  float* acc;
  void* scratch;
  float* g_flat;                 /* Adam / RMSprop: dense gradient (zero between steps) and moments, like w_flat */
  float* m_flat;
  float* v_flat;
  /* exact lazy Adam / RMSprop (hiprec_lazy_state below): non-NULL stamps replace the dense sweep of every step by
   * catch-up + update of the step's rows; the caller flushes (hiprec_lazy_flush) before anybody reads the tables */
  int32_t* stamp_u;
  int32_t* stamp_i;
  float* lazy_scalars;
  int64_t lazy_scalars_cap;
  float* cbuf;                   /* owner-pulls SGD step: [3 * cap][dim] / [3 * cap] contribution buffers (work space) */
  float* cbias;
} hiprec_shard_bufs;

/* ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd of the RCCL the caller loaded (this library links none) */
typedef struct hiprec_nccl_fns {
  void* send;
  void* recv;
  void* group_start;
  void* group_end;
} hiprec_nccl_fns;

size_t hiprec_shard_plan_bytes(void); /* sizeof the two structs above, for the host layer's layout check */
size_t hiprec_shard_bufs_bytes(void);

/* Steps [step_begin, step_end) of a planned epoch, kernels and exchanges, enqueued on `stream`: per step payload +
 * clear -> grouped send / recv of rows -> gradient kernel -> partials into the extra rows -> grouped send / recv of
 * gradients -> apply + bookkeeping (-> dense sweep for kind != HIPREC_OPT_SGD).  With plain SGD and the plan's
 * contribution lists (cidx / rows / counts, dim % 4 == 0) the gradient side is hiprec_mf_bpr_pull_remote_step: nothing is
 * cleared, no float atomics, the partials are published by its second launch.  world == 1 needs no communicator. */
int hiprec_shard_planned_steps(const hiprec_shard_plan* plan, const hiprec_shard_bufs* bufs, int64_t step_begin,
                               int64_t step_end, int32_t kind, float reg_coef, double lr, double beta1, double beta2,
                               double eps, const hiprec_nccl_fns* nccl, void* comm, hiprec_stats* stats, void* stream);

/* The same with flags.  HIPREC_SHARD_EXCHANGE_SELF: the rank's own segment of both exchanges is sent to and received
 * from ITSELF through the communicator (grouped ncclSend + ncclRecv to its own rank) instead of being handed over in
 * place -- bit-identical results; at world == 1 (which then needs the entry points and a one-rank communicator) this
 * executes the driver's real send / recv path on a single GPU.  New capability; the semantics of the step
 * are beta_rec/models/mf.py:92-119. */
#define HIPREC_SHARD_EXCHANGE_SELF 1u
int hiprec_shard_planned_steps_ex(const hiprec_shard_plan* plan, const hiprec_shard_bufs* bufs, int64_t step_begin,
                                  int64_t step_end, int32_t kind, float reg_coef, double lr, double beta1, double beta2,
                                  double eps, const hiprec_nccl_fns* nccl, void* comm, uint32_t flags,
                                  hiprec_stats* stats, void* stream);

/* ---- exact lazy Adam / RMSprop for tables that live in HBM (csrc/lazy_opt.hip; reference semantics:
 * beta_rec/models/torch_engine.py:30-39 -- nn.Embedding is dense, so torch.optim steps every element every step, a
 * zero gradient for the rows a batch did not touch).  A zero-gradient step of a row depends on the row's own (w, m, v)
 * and the step number only: it is postponed until the row is needed and then REPLAYED, the same fp32 operations in the
 * same order as the dense sweep (hiprec_opt_dense_step) -- bit-identical to it after hiprec_lazy_flush.
 *   stamp_u[n_users], stamp_i[n_items] (int32): the step a row is current as of; -1 = never touched (m = v = 0).
 *     Initialise to -1 together with zeroed moments; after loading optimizer state call hiprec_lazy_mark_current.
 *   scalars[scalars_cap][2] (fp32; Adam only): (lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t) as the step's sweep takes
 *     it -- its v_rcp_f32; libhiprec_ieee.so, which divides, keeps sqrt(1 - beta2^t) itself: the table belongs to the
 *     library that wrote it) of step t, recorded by
 *     hiprec_lazy_update, read by the replays; steps >= scalars_cap must have converged bias corrections
 *     (HIPREC_STATUS_LAZY_TABLE otherwise; 65 536 entries are plenty for the default betas).
 * All buffers are flat and laid out like the MF parameters [user_emb | item_emb | user_bias | item_bias | global_bias];
 * g is the dense gradient, zero except for the rows of the step in flight. */
typedef struct hiprec_lazy_state {
  float* w;
  float* g;
  float* m; /* Adam: exp_avg; NULL for RMSprop */
  float* v; /* exp_avg_sq / square_avg */
  int64_t n_users, n_items;
  int32_t dim;
  int32_t kind; /* HIPREC_OPT_ADAM | HIPREC_OPT_RMSPROP */
  int32_t* stamp_u;
  int32_t* stamp_i;
  float* scalars;
  int32_t scalars_cap, _pad;
  double lr, beta1, beta2, eps;
} hiprec_lazy_state;

/* The rows one step touches, as id lists WITH duplicates (-1 = skip): local user rows, two int64 item lists (a batch's
 * positives and negatives) and one int32 item list (the rows a planned step's peers ask for). */
typedef struct hiprec_lazy_rows {
  const int64_t* users;
  int64_t n_users;
  const int64_t* items_a;
  int64_t n_items_a;
  const int64_t* items_b;
  int64_t n_items_b;
  const int32_t* items_c;
  int64_t n_items_c;
} hiprec_lazy_rows;

size_t hiprec_lazy_state_bytes(void);
/* BEFORE a step reads its rows: the listed rows that lag behind the clock (stats->step = completed steps) are replayed
 * up to it and stored.  No-op for RMSprop (a zero-gradient step leaves w alone; v is replayed by the update).
 * The replay is BOUNDED (round 5): once a row's weights have stopped moving under zero-gradient steps -- exactly:
 * fl(w - x) == w for a sequence of x that only shrinks; ~150-200 steps after the row's last gradient at lr 0.05 --
 * the remaining steps cannot move them either, so a catch-up stops there and a flush only lets the moments decay.
 * Same bits as replaying every step; switched off for betas / eps for which the argument does not hold. */
int hiprec_lazy_catchup(const hiprec_lazy_state* state, const hiprec_lazy_rows* rows, hiprec_stats* stats, void* stream);
/* AFTER the step's gradients are complete in g and the clock has been advanced to the step: every listed row takes the
 * step (g row cleared, stamp = clock), the scalar (last element) too -- its gradient is g's last element plus, if
 * `scratch` is the gradient kernel's scratch block, the per-block partials (which also books loss / reg into stats, as
 * hiprec_opt_dense_step does). */
int hiprec_lazy_update(const hiprec_lazy_state* state, const hiprec_lazy_rows* rows, const void* scratch,
                       hiprec_stats* stats, void* stream);
/* Every lagging row of both tables replayed up to the clock: before predict / state_dict / checkpoints / a dense sweep. */
int hiprec_lazy_flush(const hiprec_lazy_state* state, hiprec_stats* stats, void* stream);
/* After a dense sweep over flushed tables (or after loading optimizer state): EVERY row is current as of the clock
 * (the sweep may have given any row its first gradient; rows whose moments are still zero cost a replay nothing). */
int hiprec_lazy_mark_current(const hiprec_lazy_state* state, const hiprec_stats* stats, void* stream);

/* MFEngine.train_an_epoch (mf.py:121-139) over a staged epoch (users / items_a / third contiguous in visiting order;
 * loss_kind 0 = BPR with third = int64 negatives, 1 = BCE with third = fp32 ratings) with the exact lazy optimizer: per
 * step hiprec_lazy_catchup of the batch's rows, the gradient kernel (hiprec_mf_bpr_grad / _bce_grad: dense gradient in
 * g = state->g), hiprec_lazy_update with the kernel's scratch.  first_of_epoch != 0 clears the epoch sums first.  The
 * caller flushes (hiprec_lazy_flush) before anybody reads the tables. */
int hiprec_mf_epoch_lazy(const hiprec_lazy_state* state, const hiprec_mf_tables* w, const hiprec_mf_tables* g,
                         const int64_t* users, const int64_t* items_a, const void* third, int32_t loss_kind, int64_t n,
                         int64_t batch, int32_t first_of_epoch, float reg_coef, hiprec_stats* stats, void* scratch,
                         size_t scratch_bytes, void* stream);
/* The BPR gradient of one batch on LOCAL tables by the owned-rows kernel (csrc/mf_owned.hip) instead of the atomics of
 * hiprec_mf_bpr_grad: the complete gradient of every row of the batch into g_flat (laid out like w_flat, zero on entry;
 * plain stores for rows with a single writer), nothing updated, the partials left in `scratch`, the optimizer clock
 * advanced by one (like hiprec_mf_bpr_grad).  own_* / total: this
 * batch's slices of hiprec_batch_row_ownership's arrays.  Replaces the autograd backward of mf.py:128-135. */
int hiprec_mf_bpr_grad_owned(const float* w_flat, float* g_flat, int64_t n_users, int64_t n_items, int32_t dim,
                             const int64_t* users, const int64_t* pos, const int64_t* neg, const int32_t* own_u,
                             const int32_t* own_p, const int32_t* own_n, const int32_t* total, int64_t batch,
                             float inv_batch, float reg_coef, hiprec_stats* stats, void* scratch, void* stream);
/* hiprec_mf_epoch_lazy for BPR with that gradient kernel: own_* [n], total [n_batches][total_stride] from
 * hiprec_batch_row_ownership over the staged epoch. */
int hiprec_mf_epoch_lazy_owned(const hiprec_lazy_state* state, const int64_t* users, const int64_t* pos,
                               const int64_t* neg, const int32_t* own_u, const int32_t* own_p, const int32_t* own_n,
                               const int32_t* total, int64_t total_stride, int64_t n, int64_t batch,
                               int32_t first_of_epoch, float reg_coef, hiprec_stats* stats, void* scratch, void* stream);

/* Round 5: the same epoch as OWNER PULLS (csrc/lazy_opt.hip, csrc/mf_owned.hip) -- per step catch-up -> gradient launch
 * (the parts of EVERY row's gradient stored to the contribution buffer with plain stores; counts the step) -> one
 * apply launch: a lane group per row sums the row's parts, replays the moments over the steps the row lagged, takes
 * the real Adam / RMSprop step and stamps the row.  No dense gradient traffic, no float atomics, no claims.  cidx /
 * rows / counts: hiprec_batch_row_contrib's arrays made with min_contrib = 1, offset to this piece's first step
 * (cidx_stride = the n they were made for); cbuf [3 * batch, dim], cbias [3 * batch]: work space.  BPR, dim % 4 == 0.
 * Same semantics as hiprec_mf_epoch_lazy_owned (beta_rec/models/mf.py:121-139 with torch.optim.Adam / RMSprop,
 * models/torch_engine.py:30-39); the caller flushes afterwards. */
int hiprec_mf_epoch_lazy_pull(const hiprec_lazy_state* state, const int64_t* users, const int64_t* pos,
                              const int64_t* neg, const int32_t* cidx, int64_t cidx_stride, const int32_t* rows,
                              int64_t row_cap, const int32_t* counts, float* cbuf, float* cbias, int64_t n,
                              int64_t batch, int32_t first_of_epoch, float reg_coef, hiprec_stats* stats, void* scratch,
                              void* stream);

/* ---- ONE launch of that sequence, with the buffers of this step named explicitly: for callers that
 *      have to do something between two steps -- the data-parallel engine all-reduces
 *      [partials of scratch_cur | g_cur] over RCCL before the next launch consumes them as
 *      scratch_prev / g_prev.  The launch of step k applies update(w_read, g_prev,
 *      m_read, v_read) (the step whose batch had prev_batch triples; prev_batch = 0: nothing pending,
 *      first step of an epoch), accumulates the gradient of its own batch into g_cur (zero on entry)
 *      with the loss partials in scratch_cur, writes w_write / m_write / v_write and clears g_zero.
 *      batch = 0 is the flush that only applies the pending update.  inv_batch is 1/B of the GLOBAL
 *      batch.  hiprec_mf_bpr_epoch_fused is a loop over this call with w/m/v ping-ponging between two
 *      buffers, g rotating through three and scratch between two. */
typedef struct hiprec_fused_step {
  int32_t kind; /* HIPREC_OPT_* */
  int32_t dim;
  int64_t n_users, n_items;
  const float *w_read, *g_prev, *m_read, *v_read; /* flat buffers (layout of hiprec_mf_tables) */
  float *w_write, *m_write, *v_write, *g_cur, *g_zero;
  const void* scratch_prev;
  void* scratch_cur;
  double lr, beta1, beta2, eps;
  float reg_coef;
  int32_t _pad;
} hiprec_fused_step;

size_t hiprec_fused_step_bytes(void);
int hiprec_mf_bpr_fused_step(const hiprec_fused_step* step, const int64_t* users, const int64_t* pos,
                             const int64_t* neg, int64_t batch, int64_t prev_batch, float inv_batch,
                             hiprec_stats* stats, void* stream);

/* The plain-SGD spelling of hiprec_mf_bpr_epoch_fused (first ABI revision). */
int hiprec_mf_bpr_epoch_sgd_fused(float* const* w_flat, float* const* g_flat, void* const* scratch2,
                                  int64_t n_users, int64_t n_items, int32_t dim,
                                  const int64_t* users, const int64_t* pos, const int64_t* neg,
                                  int64_t n_triples, int64_t batch, float reg_coef, double lr,
                                  hiprec_stats* stats, int32_t* final_index, void* stream);

/* ---- the data-parallel (replicated-table) step, split around its one collective:
 *        begin = zero_grad + forward + loss + backward on the local share of the global batch, then
 *                the reduction of the loss partials: the scalar-bias gradient lands in its slot of g,
 *                {loss, reg} in the two floats that follow the gradient (loss_reg_out), so that ONE
 *                all-reduce of [g | loss | reg] moves everything;
 *        (the caller all-reduces that buffer over RCCL)
 *        end   = the dense optimizer sweep over the summed gradient.
 *      Semantically hiprec_mf_{bpr,bce}_grad + hiprec_finalize_stats, and hiprec_opt_dense_step; the
 *      context carries every per-engine constant so that the host pays two short calls per step
 *      (the replicated engine is host-bound at world sizes whose all-reduce is short). */
typedef struct hiprec_dp_step {
  hiprec_mf_tables w, g;
  hiprec_stats* stats;
  void* scratch;
  size_t scratch_bytes;
  float* loss_reg_out; /* the two floats after the gradient in the all-reduced buffer */
  float* w_flat;       /* flat views of w / g / optimizer state (hiprec_opt_dense_step) */
  float* g_flat;
  float* m_flat;
  float* v_flat;
  int64_t n_flat;
  double lr, beta1, beta2, eps;
  float reg_coef;
  int32_t loss_kind; /* 0 = BPR (third = negative items, int64), 1 = BCE (third = ratings, fp32) */
  int32_t opt_kind;  /* HIPREC_OPT_* */
  int32_t _pad;
} hiprec_dp_step;

size_t hiprec_dp_step_bytes(void);
int hiprec_mf_dp_step_begin(const hiprec_dp_step* c, const int64_t* users, const int64_t* items_a,
                            const void* third, int64_t batch, float inv_batch_global, void* stream);
int hiprec_mf_dp_step_end(const hiprec_dp_step* c, void* stream);

/* ======================= NCF family: NeuMF / GMF / MLP (models/ncf.py, gmf.py, mlp.py) ============ */

#define HIPREC_NCF_MAX_LAYERS 8

/* Everything one NCF-family step touches: the four embedding tables, the tower, the head, their
 * gradient accumulators (same shapes, all-zero between steps) and a caller-owned activation
 * workspace for up to max_batch samples.  A half that the model does not have is dim 0 / NULL:
 *   NeuMF (models/ncf.py:24-50)  dim_mlp = emb_dim*2^(L-1), dim_mf = emb_dim, relu_input = 1 (quirk Q7)
 *   MLP   (models/mlp.py:22-38)  dim_mf = 0, relu_input = 0
 *   GMF   (models/gmf.py:19-27)  dim_mlp = 0, n_layers = 0
 * Linear layer l is fc_w[l] [layer_out[l], layer_in[l]] row-major (nn.Linear) + fc_b[l];
 * affine_output is out_w [layer_out[L-1] + dim_mf] (tower part first, as torch.cat([mlp, mf])). */
typedef struct hiprec_ncf_plan {
  float *user_mlp, *item_mlp, *user_mf, *item_mf;
  float *g_user_mlp, *g_item_mlp, *g_user_mf, *g_item_mf;
  int64_t n_users, n_items;
  int32_t dim_mlp, dim_mf, n_layers, relu_input;
  int32_t layer_in[HIPREC_NCF_MAX_LAYERS], layer_out[HIPREC_NCF_MAX_LAYERS];
  float *fc_w[HIPREC_NCF_MAX_LAYERS], *fc_b[HIPREC_NCF_MAX_LAYERS];
  float *g_fc_w[HIPREC_NCF_MAX_LAYERS], *g_fc_b[HIPREC_NCF_MAX_LAYERS];
  float *out_w, *out_b, *g_out_w, *g_out_b;
  int64_t max_batch;
  float* act[HIPREC_NCF_MAX_LAYERS + 1];  /* act[0] = [B, 2*dim_mlp] (ReLU-ed if relu_input), act[l] = H_l */
  float* dact[HIPREC_NCF_MAX_LAYERS + 1]; /* d loss / d (pre-activation), same shapes */
  float *mf, *dmf;                        /* [B, dim_mf] */
  float* scores;                          /* [B] sigmoid outputs */
  /* tower dropout (ncf.py:42-45, mlp.py:30-33: nn.Dropout in front of every Linear), training only:
   * keep[l] = one byte per element of Linear l's input [B, layer_in[l]], or NULL for none; kept entries
   * are scaled by keep_scale = 1 / (1 - p).  With any keep[l] set the step runs one launch per layer
   * (the fused tower kernel has no mask input); act[l] then holds the input AFTER its Dropout. */
  const uint8_t* keep[HIPREC_NCF_MAX_LAYERS];
  float keep_scale;
  int32_t _pad;
} hiprec_ncf_plan;

size_t hiprec_ncf_plan_bytes(void); /* sizeof(hiprec_ncf_plan), for binding-layout checks */

/* ---- exact-fp32 MFMA GEMM with fused epilogue (the nn.Linear forward / backward of the tower):
 * mode 0: C = A[M,K] B[N,K]^T   mode 1: C = A[M,K] B[K,N]   mode 2: C = A[K,M]^T B[K,N];
 * then C += bias[n] (if bias), relu (if relu), C *= [mask[m,n] > 0] (if mask). */
int hiprec_gemm_f32(int mode, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                    float* C, int ldc, const float* bias, int relu, const float* mask, int ldm,
                    void* stream);

/* ---- model.forward / predict under no_grad (ncf.py:52-78): plan->scores[0:batch] = sigmoid(logit) */
int hiprec_ncf_forward(const hiprec_ncf_plan* plan, const int64_t* users, const int64_t* items,
                       int64_t batch, hiprec_stats* stats, void* stream);

/* ---- zero_grad + forward + BCELoss(mean) + backward of {NeuMF,GMF,MLP}Engine.train_single_batch
 * (ncf.py:100-120, gmf.py:60-80, mlp.py:76-96) with dropout 0.  Dense gradients are accumulated
 * into the plan's g_* buffers; the loss partials and d loss/d affine_output.bias are left in
 * scratch for the optimizer call that follows (hiprec_opt_dense_step with scalar_index = position
 * of affine_output.bias in the flat buffer) or for hiprec_finalize_stats. */
int hiprec_ncf_grad(const hiprec_ncf_plan* plan, const int64_t* users, const int64_t* items,
                    const float* ratings, int64_t batch, float inv_batch, hiprec_stats* stats,
                    void* scratch, size_t scratch_bytes, void* stream);
/* hiprec_ncf_grad + optimizer.step() (ncf.py:100-120, torch_engine.py:23-39) in ONE call.  The flat buffers hold
 * [tables | tower | head] (w / g / m / v laid out alike, 16-byte aligned); the first table_floats elements are the
 * embedding tables: their gradients are complete after the forward + chain launch, so their share of the dense sweep
 * rides as extra blocks of the grouped weight-gradient launch and only the tower / head tail keeps a launch of its
 * own.  scalar_index: the element (>= table_floats) whose gradient arrives through the loss partials
 * (affine_output.bias), or -1.  Shapes outside the fused launches take hiprec_ncf_grad + hiprec_opt_dense_step. */
int hiprec_ncf_step(const hiprec_ncf_plan* plan, const int64_t* users, const int64_t* items, const float* ratings,
                    int64_t batch, float inv_batch, int kind, float* w_flat, float* g_flat, float* m_flat,
                    float* v_flat, int64_t n_flat, int64_t table_floats, int64_t scalar_index, double lr, double beta1,
                    double beta2, double eps, hiprec_stats* stats, void* scratch, size_t scratch_bytes, void* stream);


/* ======================= LightGCN (models/lightgcn.py) ============================================ */

/* CSR graph: the reference's norm_adj = D^-1 (A + I) over (n_users + n_items) nodes
 * (utils/common_util.py:24-41, data/deprecated_data_base.py:353), or its transpose.  `eid` (may be
 * NULL = identity) maps an edge of THIS matrix to its index in the forward matrix, so that the
 * transposed graph applies the same edge-dropout bytes. */
typedef struct hiprec_csr {
  const int64_t* rowptr; /* [n_rows + 1] */
  const int32_t* col;    /* [nnz] */
  const float* val;      /* [nnz] */
  const int32_t* eid;    /* [nnz] or NULL */
  int64_t n_rows;
  int64_t nnz;
  const int32_t* slice_row; /* [hiprec_csr_n_slices(nnz)] from hiprec_csr_slice_rows, or NULL: the row of
                             * the first edge of every 256-edge slice (one SpMM wave per slice; NULL makes
                             * every wave binary-search rowptr, 14 dependent loads) */
} hiprec_csr;

/* ---- per-graph preprocessing for hiprec_spmm_csr: out[s] = row that owns edge 256 * s. */
int64_t hiprec_csr_n_slices(int64_t nnz);
int hiprec_csr_slice_rows(const hiprec_csr* a, int32_t* out, int64_t n_out, void* stream);

/* Everything one LightGCN step touches.  e0 / g are the flat parameter / gradient buffers
 * [user_embedding | item_embedding] = [(n_users + n_items), dim]; the rest is caller-owned
 * workspace of the same shape. */
/* A graph stored for the column-sliced SpMM (n_rows < 65536).  Every row's edge list is padded to a multiple of
 * S = lane_slots SLOTS (16; a factored graph may use 24, 32 or 48 -- the host picks the S that makes typical rows few
 * chunks, because a chunk's fixed cost is that of ~16 slots): col16 / val / eid hold n_slots entries (padding: col 0,
 * val 0, eid -1; eid = the edge's index into the keep bytes of a step, i.e. its number in the FORWARD graph's CSR
 * order), n_slots a multiple of 16, and the arrays END with at least S padding slots starting at pad_slot (a multiple
 * of 8): lanes that have no slots of their own read those.  Work items are chunks: at most 4 S consecutive slots of one
 * row (S per lane of a quad), chunks[2 * c] = first slot (a multiple of S), chunks[2 * c + 1] = row |
 * n_slots_of_chunk << 16 | run flags, sorted by row; every subgroup is padded with empty chunks (0 slots) to a
 * multiple of 16 chunks.  The kernel's workgroup of 1024 threads gives a WINDOW of 16 consecutive chunks to a wave (one
 * per quad of lanes) and sums the chunks of one row inside the window (a RUN) before the LDS sees them; the flags say
 * what a quad does: bits 24-25 min(position in the run, quad index inside its 16-lane row), bit 26 the run began in an
 * earlier 16-lane row, bit 27 last chunk of the run (it stores), bit 28 the run is the whole row (plain store instead
 * of an LDS atomic).  The rows are cut into n_groups * subs_per_group subgroups of consecutive rows and about equal
 * chunk count: subgroup k covers rows sub_row[k] .. sub_row[k + 1] (sub_row[0] = 0, the last = n_rows; at most
 * row_cap rows, so that their accumulators fit the LDS next to the slice: hiprec_sliced_row_cap) and chunks
 * sub_chunk[k] .. sub_chunk[k + 1].  n_groups should be a multiple of 8 with (dim / slice width) * n_groups = the
 * CU count.  A graph whose values have the rank-one form of a degree-normalised adjacency carries row_scale /
 * col_scale: the SpMM then streams 2-byte columns only (see hiprec_spmm_sliced).
 * beta-recsys_amd/lightgcn.py: sliced_graph_host builds it. */
typedef struct hiprec_sliced_csr {
  const int32_t* chunks;
  const uint16_t* col16;
  const float* val;
  const int32_t* eid;
  const int32_t* sub_row;
  const int32_t* sub_chunk;
  const int32_t* spill_row;  /* rows that are summed in LDS (several runs), workgroup g's: spill_ptr[g] .. spill_ptr[g + 1] */
  const int32_t* spill_ptr;  /* [n_groups + 1] */
  const int32_t* empty_row;  /* rows without edges (written as zeros), workgroup g's: empty_ptr[g] .. empty_ptr[g + 1] */
  const int32_t* empty_ptr;  /* [n_groups + 1] */
  const float* row_scale; /* both NULL, or the FACTORED form: val of edge (i, j) == row_scale[i] * col_scale[j]; */
  const float* col_scale; /* padding slots then hold column n_rows (an all-zero source row), not 0           */
  int64_t n_rows, n_slots;
  int32_t n_groups, subs_per_group, n_chunks, row_cap;
  int32_t lane_slots; /* S: slots per lane and chunk quarter -- 16, or 24 / 32 / 48 for a factored graph */
  int32_t pad_slot;   /* first of (at least) S all-padding slots at the end of the slot arrays: what a lane without
                       * slots of its own reads */
} hiprec_sliced_csr;

/* slice width (floats) the sliced SpMM uses for n_rows x dim: 4 or 2, the largest that divides dim and lets one
 * slice of the source matrix (n_rows x width x 4 B) fit the LDS; 0 = not applicable (use hiprec_spmm_csr) */
int32_t hiprec_sliced_width(int64_t n_rows, int32_t dim);
/* most rows a subgroup of hiprec_sliced_csr may hold at that width */
int32_t hiprec_sliced_row_cap(int64_t n_rows, int32_t dim);
/* row-major [n_rows][dim] <-> sliced [dim / slice_w][n_rows][slice_w] (add != 0: y += instead of y =);
 * row_scale (may be NULL): xs = row_scale (.) x, the source a factored graph's pass expects (its col_scale) */
int hiprec_to_sliced(const float* x, int64_t n_rows, int32_t dim, int32_t slice_w, const float* row_scale, float* xs,
                     void* stream);
int hiprec_from_sliced(const float* xs, int64_t n_rows, int32_t dim, int32_t slice_w, float* y, int32_t add,
                       void* stream);
/* The edge stream of one step, n_slots entries in `out` (sized for n_slots floats): a general graph's dropped values
 * out[slot] = keep[eid[slot]] ? val[slot] : 0 (float); a factored graph's dropped columns keep[..] ? col16[slot] :
 * n_rows (uint16).  To be passed as step_edges below; the 1 / keep_prob factor goes into `scale`. */
int hiprec_sliced_drop_values(const hiprec_sliced_csr* a, const uint8_t* keep, float* out, void* stream);
/* ys = scale * A xs on SLICED buffers, step_edges NULL = no dropout; acc_mode 0: nothing else, 1: accs += Y,
 * 2: accs = Y.  Every row of ys is written (no zero fill needed).  General graph: xs = X, ys = Y.  Factored graph:
 * xs must hold col_scale (.) X (hiprec_to_sliced with row_scale = a->col_scale) and ys receives col_scale (.) Y,
 * ready to be the next pass's source, while accs gets Y itself. */
int hiprec_spmm_sliced(const hiprec_sliced_csr* a, const void* step_edges, float scale, const float* xs, float* ys,
                       float* accs, int32_t acc_mode, int32_t dim, int32_t slice_w, void* stream);

typedef struct hiprec_lightgcn_plan {
  hiprec_csr a;  /* forward graph  */
  hiprec_csr at; /* its transpose (backward) */
  int64_t n_users, n_items;
  int32_t dim, n_layers;
  float decay; /* regs[0], lightgcn.py:112-113 */
  int32_t _pad;
  float *e0, *g, *xa, *xb, *acc, *da, *db;
  /* Optional: one contiguous caller-owned region of (1 + 2*n_layers) * n_rows * dim floats.  When
   * set, every SpMM of a step writes its own slice of it (and d_out lives in the first slice), so
   * the step zeroes all of them with ONE memset instead of one ~5 us fill launch per SpMM; xa / xb /
   * da / db are then unused. */
  float* zero_ws;
  int64_t zero_ws_floats;
  /* Optional (graphs whose node count fits the LDS, hiprec_sliced_width(n_rows, dim) > 0): the same two graphs
   * stored for the column-sliced SpMM and 4 * n_rows * dim + sa.n_slots + sat.n_slots floats of workspace; propagation then runs on
   * sliced buffers (one transpose in, one out) and needs neither global atomics nor zero fills. */
  hiprec_sliced_csr sa, sat;
  int32_t slice_w;
  int32_t dropped_ready; /* != 0: sliced_ws already holds this step's dropped values (hiprec_lightgcn_step_values) */
  float* sliced_ws;
  int64_t sliced_ws_floats;
} hiprec_lightgcn_plan;

size_t hiprec_lightgcn_plan_bytes(void);

/* ---- y = (A with dropped edges) x ; acc += y.  keep (may be NULL) holds one byte per forward edge,
 * kept edges are scaled by `scale` (= 1/keep_prob, lightgcn.py:27-38).  y is zeroed by the call. */
int hiprec_spmm_csr(const hiprec_csr* a, const uint8_t* keep, float scale, const float* x, float* y,
                    float* acc, int32_t dim, void* stream);

/* ---- keep[e] = uniform(seed, step, e) < keep_prob on the device (counter-based, stateless): the
 * fast alternative to drawing torch.rand(nnz) on the CPU every step as lightgcn.py:32 does. */
int hiprec_edge_dropout_mask(uint8_t* keep, int64_t nnz, float keep_prob, uint64_t seed,
                             uint64_t step, void* stream);

/* ---- (sliced plans) the dropped edge values of one training step for both graphs, into plan->sliced_ws, in ONE
 * launch: draw != 0 draws the mask on the device (the draw of hiprec_edge_dropout_mask for the same seed / step;
 * written to keep[nnz] as well when keep is not NULL), draw == 0 reads keep[].  The same launch lays plan->e0 out in
 * the sliced layout for the first pass.  Set plan->dropped_ready for the propagate / grad calls of that step (same
 * e0, nothing in between); without it they prepare values and layout themselves, one launch each. */
int hiprec_lightgcn_step_values(const hiprec_lightgcn_plan* plan, uint8_t* keep, float keep_prob, int32_t draw,
                                uint64_t seed, uint64_t step, void* stream);

/* ---- (sliced plans of width 4, device draw) optimizer.step() of step t and the preparation of step t + 1 in ONE
 * launch: hiprec_opt_dense_step's arithmetic over the N x dim embedding matrix plan->e0 (g / m / v laid out alike;
 * scratch as there: the loss partials of the step's gradient call, or NULL), with the fresh weights written row-major
 * AND in the sliced layout of the next step's first pass, and the dropped edge streams of `next_step` drawn by the rest
 * of the grid (hiprec_lightgcn_step_values's work).  Start step t + 1 with dropped_ready = 1 if nothing has touched the
 * weights since. */
int hiprec_lightgcn_opt_stage(const hiprec_lightgcn_plan* plan, int32_t kind, float* g, float* m, float* v, double lr,
                              double beta1, double beta2, double eps, hiprec_stats* stats, const void* scratch,
                              float keep_prob, uint64_t seed, uint64_t next_step, void* stream);

/* ---- LightGCN.forward (lightgcn.py:46-78): plan->acc = sum_l A^l E0 (propagated = acc/(L+1)). */
int hiprec_lightgcn_propagate(const hiprec_lightgcn_plan* plan, const uint8_t* keep, float keep_prob,
                              void* stream);

/* ---- scores = sigmoid(<out[u], out[item]>) on the rows left in plan->acc by the last propagate
 * (LightGCN.predict, lightgcn.py:80-101). */
int hiprec_lightgcn_predict(const hiprec_lightgcn_plan* plan, const int64_t* users,
                            const int64_t* items, int64_t n, float* scores, hiprec_stats* stats,
                            void* stream);

/* ---- zero_grad + forward + loss_comput + backward of LightGCNEngine.train_single_batch
 * (lightgcn.py:119-149, 171-191): dense gradient into plan->g, loss partials in scratch. */
int hiprec_lightgcn_grad(const hiprec_lightgcn_plan* plan, const uint8_t* keep, float keep_prob,
                         const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t batch,
                         float inv_batch, hiprec_stats* stats, void* scratch, size_t scratch_bytes,
                         void* stream);

/* ================= Negative sampling ("next": the data step in front of the path) ====
 * data/base_data.py:218-253 instance_bpr_loader (k = 1), :182-216 instance_bce_loader and :254-288
 * instance_mul_neg_loader (k = num_negative): for every row of the training frame, k DISTINCT items
 * drawn uniformly from the items of the pool (ids 0..n_items-1, base_data.py:48) the row's user never
 * interacted with — random.sample(list(set(item_id_pool) - positive_items), k).
 * user_ptr[n_users+1] / pos_sorted: CSR of each user's positive items, ascending and unique.
 * users[n_rows]: the user of each training row.  out[n_rows * k], row-major.  The draw is a pure
 * function of (seed, row, j): out[row*k + j] = the r-th item not in the user's list, r = element j of
 * the Feistel permutation of [0, n_items - deg) keyed by splitmix64(seed ^ splitmix64(row))
 * (oracle/sampler_numpy.py restates it bit for bit).  A user with fewer than k untouched items sets
 * HIPREC_STATUS_NEG_EXHAUSTED (python raises ValueError there) and yields -1. */
int hiprec_sample_negatives(const int64_t* user_ptr, const int64_t* pos_sorted, int64_t n_users,
                            int64_t n_items, const int64_t* users, int64_t n_rows, int32_t k,
                            uint64_t seed, int64_t* out, hiprec_stats* stats, void* stream);

/* ================= Ranking evaluation ("next": the caller after the train step) ====
 * core/eval_engine.py:49-87 evaluate() -> utils/evaluation.py:461-533 merge_ranking_true_pred
 * (relevancy_method "top_k"), :535-583 precision_at_k, :586-629 recall_at_k, :632-689 ndcg_at_k,
 * :692-752 map_at_k.  evaluate() scores the SAME rows it takes the truth from, so the input is one
 * (score, rating) pair per candidate, grouped by user: segment s owns rows seg_ptr[s]..seg_ptr[s+1]
 * (device int64, n_segments+1 entries) in the data frame's original row order (ties in score rank
 * by that order, like nlargest(keep="first") + rank(method="first"), evaluation.py:778-784,516-518).
 * A row is relevant when rating >= 1 (evaluation.py:492); users with no relevant row are not
 * "common users" and are left out of the mean (evaluation.py:495-498).  (user,item) pairs must be
 * unique per user, as in every frame the reference's splitters produce.
 * k_list_host is a HOST array of n_k (<= HIPREC_RANK_MAX_K) cut-offs.  out (device, fp64):
 *   out[0] = number of common users, out[1+4j .. 4+4j] = precision, recall, ndcg, map at k_list[j].
 * workspace: hiprec_rank_metrics_workspace_bytes(n_segments, n_k) bytes of device memory. */
#define HIPREC_RANK_MAX_K 8
size_t hiprec_rank_metrics_workspace_bytes(int64_t n_segments, int32_t n_k);
int hiprec_rank_metrics(const int64_t* seg_ptr, int64_t n_segments, const float* scores,
                        const float* ratings, const int32_t* k_list_host, int32_t n_k,
                        double* workspace, size_t workspace_bytes, double* out, void* stream);

/* ================= PairwiseGMF, the CMN pre-training model (sibling models) ====
 * models/pairwise_gmf.py:28-46 parameters: user_memory [n_users, dim], item_memory [n_items, dim],
 * v = nn.Linear(dim, 1, bias=False).weight [1, dim].  The host keeps them in ONE flat buffer in that
 * order (and the dense gradient in another) so that hiprec_clip_grad_norm and hiprec_opt_dense_step
 * sweep them in one pass each. */
typedef struct hiprec_pgmf_tables {
  float* user_memory; /* [n_users, dim] */
  float* item_memory; /* [n_items, dim] */
  float* v;           /* [dim] */
  int64_t n_users;
  int64_t n_items;
  int32_t dim;        /* <= 256 */
  int32_t _pad;
} hiprec_pgmf_tables;

/* bytes of device workspace hiprec_pgmf_bpr_grad needs (per-block partial sums of grad v) */
size_t hiprec_pgmf_workspace_bytes(int32_t dim);
/* ---- zero_grad + forward + loss + backward of PairwiseGMFEngine.train_single_batch
 * (pairwise_gmf.py:82-112): s = relu(v . (U[u] * I[i])) (pairwise_gmf.py:48-62), loss =
 * mean(-log(sigmoid(s+ - s-) + 1e-12)) (the engine's own bpr_loss, pairwise_gmf.py:144-158)
 * + l2_lambda * ||v||_2 (pairwise_gmf.py:105-108).  Accumulates into the dense gradient g (which the
 * previous hiprec_opt_dense_step left zeroed), leaves the loss partials in scratch for that sweep to
 * fold into stats->loss, and advances the step counter.  Out-of-range ids set the status bits and
 * the triple is skipped. */
int hiprec_pgmf_bpr_grad(const hiprec_pgmf_tables* w, const hiprec_pgmf_tables* g,
                         const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t batch,
                         float inv_batch, float l2_lambda, hiprec_stats* stats, void* scratch,
                         size_t scratch_bytes, void* workspace, size_t workspace_bytes, void* stream);

/* ---- PairwiseGMFEngine.train_an_epoch (pairwise_gmf.py:118-142) over resident (user, pos, neg) arrays
 * in visiting order (n_triples, last batch short): per batch hiprec_pgmf_bpr_grad and hiprec_clip_opt_dense_step
 * (= hiprec_clip_grad_norm(max_norm) + hiprec_opt_dense_step) over the flat buffers [user_memory | item_memory | v]
 * that w / g point into, enqueued back to back with no host work in between. */
int hiprec_pgmf_epoch(const hiprec_pgmf_tables* w, const hiprec_pgmf_tables* g, const int64_t* users,
                      const int64_t* pos, const int64_t* neg, int64_t n_triples, int64_t batch,
                      float l2_lambda, float max_norm, int kind, double lr, double beta1, double beta2,
                      double eps, float* flat_w, float* flat_g, float* flat_m, float* flat_v,
                      int64_t n_flat, hiprec_stats* stats, void* scratch, size_t scratch_bytes,
                      void* workspace, size_t workspace_bytes, void* clip_workspace,
                      size_t clip_workspace_bytes, void* stream);

/* ---- torch.nn.utils.clip_grad_norm_(parameters, max_norm) (pairwise_gmf.py:111, cmn.py:197), L2,
 * over one flat gradient of n floats: total = ||g||, g *= min(max_norm / (total + 1e-6), 1).
 * workspace: hiprec_clip_workspace_bytes() of device memory; afterwards workspace[0] (fp64) holds
 * the total norm (the function's return value in torch) and workspace[1] the coefficient. */
size_t hiprec_clip_workspace_bytes(void);
int hiprec_clip_grad_norm(float* g, int64_t n, float max_norm, void* workspace,
                          size_t workspace_bytes, void* stream);
/* hiprec_clip_grad_norm followed by hiprec_opt_dense_step (pairwise_gmf.py:137-139: clip_grad_norm_ then
 * optimizer.step()) in two launches instead of three: the scaling rides in the optimizer sweep, the scaled gradient is
 * never written.  Same bits in w / m / v / g (cleared) and in workspace[0 .. 1] = (total_norm, coef) as the two calls. */
int hiprec_clip_opt_dense_step(int kind, float* w, float* g, float* m, float* v, int64_t n, double lr, double beta1,
                               double beta2, double eps, hiprec_stats* stats, const void* scratch,
                               int64_t scalar_index, float max_norm, void* workspace, size_t workspace_bytes,
                               void* stream);

/* ================= Triple2vec (sibling models) ===============================
 * models/triple2vec.py:11-34 parameters.  item_emb2 may be the SAME pointer as item_emb1 (in w and in
 * g alike): triple2vec.py:19,38-39 aliases the two tables from the first forward on whenever
 * n_neg != 0, and the kernels then load / update the shared row once. */
typedef struct hiprec_t2v_tables {
  float* user_emb;   /* [n_users, dim] */
  float* item_emb1;  /* [n_items, dim] */
  float* item_emb2;  /* [n_items, dim] or == item_emb1 */
  float* user_bias;  /* [n_users] */
  float* item_bias;  /* [n_items] */
  int64_t n_users;
  int64_t n_items;
  int32_t dim;       /* <= 256 */
  int32_t _pad;
} hiprec_t2v_tables;

/* ---- zero_grad + forward + backward of Triple2vecEngine.train_single_batch (triple2vec.py:36-92,
 * 115-124).  pos_*[batch]; neg_*[batch * n_neg] row-major (the [B, n_neg] tensors of
 * triple2vec.py:145-168).  Both negative item ROWS are gathered with neg_i2 and neg_i1 only selects an
 * item_bias entry, exactly as triple2vec.py:46-47,69-71 do.  scale = 1 / (3 * config batch_size)
 * (triple2vec.py:92 divides by the configured batch size, also for a short last batch).  Accumulates
 * into the dense gradient g, leaves the loss partials in scratch and advances the step counter. */
int hiprec_t2v_grad(const hiprec_t2v_tables* w, const hiprec_t2v_tables* g, const int64_t* pos_u,
                    const int64_t* pos_i1, const int64_t* pos_i2, const int64_t* neg_u,
                    const int64_t* neg_i1, const int64_t* neg_i2, int64_t batch, int32_t n_neg,
                    float scale, hiprec_stats* stats, void* scratch, size_t scratch_bytes, void* stream);

/* ---- Triple2vecEngine.train_an_epoch (triple2vec.py:126-169) over resident arrays in visiting order
 * (pos_*[n_triples], neg_*[n_triples * n_neg], last batch short): per batch hiprec_t2v_grad and
 * hiprec_opt_dense_step over the first n_sweep floats of the flat buffers (the host lays the orphaned
 * item_emb2 out last and leaves it out of the sweep), enqueued back to back. */
int hiprec_t2v_epoch(const hiprec_t2v_tables* w, const hiprec_t2v_tables* g, const int64_t* pos_u,
                     const int64_t* pos_i1, const int64_t* pos_i2, const int64_t* neg_u,
                     const int64_t* neg_i1, const int64_t* neg_i2, int64_t n_triples, int64_t batch,
                     int32_t n_neg, float scale, int kind, double lr, double beta1, double beta2,
                     double eps, float* flat_w, float* flat_g, float* flat_m, float* flat_v,
                     int64_t n_sweep, hiprec_stats* stats, void* scratch, size_t scratch_bytes,
                     void* stream);

/* ---- scores[k] = <U[u_k], (E1[i_k] + E2[i_k]) / 2>  (Triple2vec.predict, triple2vec.py:94-104) */
int hiprec_t2v_predict(const hiprec_t2v_tables* w, const int64_t* users, const int64_t* items,
                       int64_t n, float* scores, hiprec_stats* stats, void* stream);

/* ---- AliasTable.sample (utils/alias_table.py:82-97; called three times per batch by
 * Triple2vecEngine.train_an_epoch, triple2vec.py:145-168) on the device: n draws from the alias table
 * (prob[vocab] fp64 = AliasTable.prob_arr, alias[vocab] = AliasTable.alias_arr, labels[vocab] =
 * AliasTable.index2Label or NULL for the identity).  Draw e is a pure function of (seed, e):
 * h1 = splitmix64(seed ^ splitmix64(e)), h2 = splitmix64(h1), column = mulhi64(h1, vocab),
 * u = (h2 >> 11) * 2^-53, result = u < prob[column] ? column : alias[column]
 * (oracle/triple2vec_numpy.py restates it bit for bit). */
int hiprec_alias_sample(const double* prob, const int64_t* alias, const int64_t* labels, int64_t vocab,
                        uint64_t seed, int64_t* out, int64_t n, void* stream);

/* ================= NGCF (sibling models) =====================================
 * models/ngcf.py:12-46.  Everything one step touches; all buffers are caller-owned device memory.
 * Hop l (0-based, l < n_layers) maps width dim[l] to dim[l+1]; N = n_users + n_items; the concatenated
 * output `all` is [N, dim[0] + ... + dim[n_layers]] (ngcf.py:74-78).  a / at: norm_adj and its transpose
 * as CSR (the same hiprec_csr LightGCN uses). */
#define HIPREC_NGCF_MAX_LAYERS 6
typedef struct hiprec_ngcf_plan {
  hiprec_csr a, at;
  int64_t n_users, n_items;
  int32_t n_layers;
  int32_t dim[HIPREC_NGCF_MAX_LAYERS + 1];   /* each <= 256 */
  float decay;                               /* regs[0] (ngcf.py:108-109) */
  float inv_reg_batch;                       /* 1 / config batch_size (ngcf.py:189) */
  /* parameters and their gradients (zero on entry of hiprec_ngcf_grad, like optimizer.zero_grad) */
  float* e0;                                 /* [N, dim[0]]: user_embedding rows, then item_embedding rows */
  float* g_e0;
  float* gc_w[HIPREC_NGCF_MAX_LAYERS];       /* GC_weights[l].weight [dim[l+1], dim[l]] */
  float* gc_b[HIPREC_NGCF_MAX_LAYERS];       /* GC_weights[l].bias   [dim[l+1]] */
  float* bi_w[HIPREC_NGCF_MAX_LAYERS];       /* Bi_weights[l].* likewise */
  float* bi_b[HIPREC_NGCF_MAX_LAYERS];
  float* g_gc_w[HIPREC_NGCF_MAX_LAYERS];
  float* g_gc_b[HIPREC_NGCF_MAX_LAYERS];
  float* g_bi_w[HIPREC_NGCF_MAX_LAYERS];
  float* g_bi_b[HIPREC_NGCF_MAX_LAYERS];
  /* forward workspace, kept for the backward */
  float* side[HIPREC_NGCF_MAX_LAYERS];       /* [N, dim[l]]   A ego_l */
  float* bi_in[HIPREC_NGCF_MAX_LAYERS];      /* [N, dim[l]]   ego_l * side */
  float* sum_pre[HIPREC_NGCF_MAX_LAYERS];    /* [N, dim[l+1]] GC_l(side) */
  float* bi_pre[HIPREC_NGCF_MAX_LAYERS];     /* [N, dim[l+1]] Bi_l(bi_in) */
  float* ego[HIPREC_NGCF_MAX_LAYERS];        /* [N, dim[l+1]] ego_{l+1} (after dropout, before normalize) */
  float* nrm[HIPREC_NGCF_MAX_LAYERS];        /* [N] row norms of ego_{l+1} */
  float* all;                                /* [N, sum dim]; columns < dim[0] are NOT filled: hop 0's slice is e0 */
  /* message dropout (ngcf.py:70): one keep byte per element of ego_{l+1}, or NULL for none; kept
   * entries are scaled by keep_scale[l] = 1 / (1 - p) */
  uint8_t* keep[HIPREC_NGCF_MAX_LAYERS];
  float keep_scale[HIPREC_NGCF_MAX_LAYERS];
  /* keep_gen != 0: the forward DRAWS the keep bytes of every hop with keep[l] != NULL itself (byte =
   * uniform(keep_seed * 64 + l, keep_step, element) < keep_prob[l], the generator of
   * hiprec_edge_dropout_mask) and stores them in keep[l] for the backward; 0: keep[l] is an input. */
  float keep_prob[HIPREC_NGCF_MAX_LAYERS];
  uint64_t keep_seed, keep_step;
  int32_t keep_gen, _pad;
  /* backward workspace, each [N, max dim] unless noted */
  float* d_all;                              /* [N, sum dim] */
  float* d_sum;
  float* d_bi;
  float* d_side;
  float* d_bi_in;
  float* d_ego[2];
  float* spmm_tmp[HIPREC_NGCF_MAX_LAYERS];   /* output of the transposed SpMM of hop l */
  /* Optional: one contiguous caller-owned region that contains every side[l], spmm_tmp[l] and d_all.
   * When set, a step clears it with ONE fill instead of one ~5 us fill launch per SpMM (their outputs
   * must be zero on entry: heavy rows are accumulated with atomics). */
  float* zero_ws;
  int64_t zero_ws_floats;
  /* Optional (graphs whose node count fits the LDS and hops whose input widths all take the same slice width,
   * hiprec_sliced_width(N, dim[l]) == slice_w): the two graphs stored for the column-sliced SpMM and one buffer of
   * N * max(dim[l]) floats for its sliced source.  The SpMMs of a step then run on hiprec_spmm_sliced's kernel: the
   * source is written in the sliced layout by the kernel that produces it, the result goes straight to side[l] /
   * d_ego; spmm_tmp is unused and NOTHING is cleared per step: d_all (which only the loss scatters into) must be zero
   * when the first step starts -- allocate the workspace zeroed -- and every backward leaves it zero again (its readers
   * clear what they read). */
  hiprec_sliced_csr sa, sat;
  int32_t slice_w, _pad2;
  float* sliced_src;
  int64_t sliced_src_floats;
  /* Optional: one [N, dim[l+1]] buffer per hop for d_sum / d_bi (instead of the shared d_sum / d_bi).  With them (hop
   * widths <= 64) a hop's backward chain is one launch and all weight / bias gradients one grouped launch at the end. */
  float* d_sum_l[HIPREC_NGCF_MAX_LAYERS];
  float* d_bi_l[HIPREC_NGCF_MAX_LAYERS];
} hiprec_ngcf_plan;

size_t hiprec_ngcf_plan_bytes(void);

/* ---- NGCF.forward (ngcf.py:48-80): fills columns >= dim[0] of plan->all (and the per-hop workspaces).  train != 0 applies
 * the keep bytes of the plan, 0 is eval mode. */
int hiprec_ngcf_forward(const hiprec_ngcf_plan* plan, int train, void* stream);

/* ---- scores[k] = <all[u_k], all[n_users + i_k]> on the rows left in plan->all by the last forward
 * (NGCF.predict, ngcf.py:82-100). */
int hiprec_ngcf_predict(const hiprec_ngcf_plan* plan, const int64_t* users, const int64_t* items,
                        int64_t n, float* scores, hiprec_stats* stats, void* stream);

/* ---- zero_grad + forward + bpr_loss + backward of NGCFEngine.train_single_batch (ngcf.py:118-149,
 * 172-199): dense gradients into the plan's g_* buffers, loss partials in scratch (stats->loss =
 * mf_loss + emb_loss, stats->reg = emb_loss), step counter advanced.  inv_batch = 1 / len(users). */
int hiprec_ngcf_grad(const hiprec_ngcf_plan* plan, const int64_t* users, const int64_t* pos,
                     const int64_t* neg, int64_t batch, float inv_batch, hiprec_stats* stats,
                     void* scratch, size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HIPREC_H */
