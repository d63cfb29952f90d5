// Triple2vec on the MF building blocks — SURVEY.md §8f rank 4 (sibling models).
//
//   beta_rec/models/triple2vec.py:11-34   parameters: user_emb [U,D], item_emb1 [I,D], item_emb2 [I,D],
//                                         user_bias [U,1], item_bias [I,1]
//   beta_rec/models/triple2vec.py:36-92   forward = the loss of one batch of (u, i1, i2) triples with
//                                         n_neg sampled (u', i1', i2') per triple
//   beta_rec/models/triple2vec.py:94-104  predict: <U[u], (E1[i] + E2[i]) / 2>
//   beta_rec/models/triple2vec.py:115-124 train_single_batch: zero_grad, forward, backward, step
//
// The loss, with s(.) = logsigmoid, eu = U[u], e1 = E1[i1], e2 = E2[i2]:
//   L = -[ s(eu.(e1+e2) + bu[u])  + sum_j s(-(U[u'_j].eu   + bu[u'_j]))
//        + s(e1.(eu+e2) + bi[i1]) + sum_j s(-(E1[i2'_j].e1 + bi[i1'_j]))     <- rows by i2', bias by i1'
//        + s(e2.(eu+e1) + bi[i2]) + sum_j s(-(E2[i2'_j].e2 + bi[i2'_j])) ] / (3 * batch_size)
// (triple2vec.py:46-47 gathers BOTH negative item rows with neg_i_2; neg_i_1 only reaches item_bias.
//  triple2vec.py:19,38-39: use_bias = n_neg, so item_emb2 IS item_emb1 from the first forward on —
//  the host passes the same pointer twice in that case and the kernel loads the shared row once.)
//
// One wave per (triple, term group): unit 0 of a triple carries its three positive terms, unit 1+j its
// negative j.  Every unit re-reads the three positive rows (cache hits) so that all of a triple's
// 1 + n_neg units run concurrently: at the reference's batch of 256 one wave per TRIPLE is one wave per
// CU walking n_neg dependent load -> reduce -> atomic chains (measured 15.8 us); per unit the chain is
// one deep.  Row gradients are scattered with fp32 atomics into the dense gradient (a negative unit
// also adds its share to the three positive rows).  HBM-bound row traffic.
#include "common.hpp"

namespace hiprec {

constexpr int kT2vMaxNpl = 4;  // dim <= 256
constexpr int kT2vMaxBlocks = 2048;

__device__ __forceinline__ bool in_range(int64_t i, int64_t n) {
  return static_cast<uint64_t>(i) < static_cast<uint64_t>(n);
}

__global__ __launch_bounds__(kBlock) void t2v_grad_kernel(
    hiprec_t2v_tables w, hiprec_t2v_tables g, const int64_t* __restrict__ pos_u,
    const int64_t* __restrict__ pos_i1, const int64_t* __restrict__ pos_i2,
    const int64_t* __restrict__ neg_u, const int64_t* __restrict__ neg_i1,
    const int64_t* __restrict__ neg_i2, int64_t batch, int n_neg, float scale, hiprec_stats* stats,
    Scratch* scratch) {
  const int lane = lane_id();
  const int D = w.dim;
  const bool shared_items = w.item_emb1 == w.item_emb2;
  const int per_triple = n_neg + 1;  // unit 0 of a triple = its three positive terms, unit 1+j = negative j
  const int64_t n_units = batch * per_triple;

  const bool stepper = blockIdx.x == 0 && threadIdx.x == 0;
  StepState step_state{};
  if (stepper) step_state = step_load(stats);

  float loss_acc = 0.f;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block(); q < n_units;
       q += static_cast<int64_t>(gridDim.x) * kWavesPerBlock) {
    const int64_t t = q / per_triple;
    const int j = static_cast<int>(q - t * per_triple) - 1;
    // all six ids in one round trip (a positive unit reads negative 0's ids and ignores them)
    int64_t nu = 0, na = 0, nb = 0;
    if (n_neg > 0) {
      const int64_t qn = t * n_neg + (j < 0 ? 0 : j);
      nu = neg_u[qn];
      na = neg_i1[qn];
      nb = neg_i2[qn];
    }
    const int64_t u = pos_u[t], a = pos_i1[t], b = pos_i2[t];
    const bool u_ok = in_range(u, w.n_users);
    const bool i_ok = in_range(a, w.n_items) && in_range(b, w.n_items);
    if (!(u_ok && i_ok)) {  // the whole triple is dropped; its positive unit reports it
      if (lane == 0 && j < 0)
        atomicOr(&stats->status,
                 (u_ok ? 0u : HIPREC_STATUS_USER_OOB) | (i_ok ? 0u : HIPREC_STATUS_ITEM_OOB));
      continue;
    }
    if (j >= 0) {
      const bool nu_ok = in_range(nu, w.n_users);
      const bool ni_ok = in_range(na, w.n_items) && in_range(nb, w.n_items);
      if (!(nu_ok && ni_ok)) {
        if (lane == 0)
          atomicOr(&stats->status,
                   (nu_ok ? 0u : HIPREC_STATUS_USER_OOB) | (ni_ok ? 0u : HIPREC_STATUS_ITEM_OOB));
        continue;
      }
    }
    float eu[kT2vMaxNpl], e1[kT2vMaxNpl], e2[kT2vMaxNpl];
#pragma unroll
    for (int k = 0; k < kT2vMaxNpl; ++k) {
      const int c = lane + kWave * k;
      const bool in = c < D;
      eu[k] = in ? w.user_emb[u * D + c] : 0.f;
      e1[k] = in ? w.item_emb1[a * D + c] : 0.f;
      e2[k] = in ? w.item_emb2[b * D + c] : 0.f;
    }
    if (j < 0) {
      float dot = 0.f, dot1 = 0.f, dot2 = 0.f;
#pragma unroll
      for (int k = 0; k < kT2vMaxNpl; ++k) {
        dot += eu[k] * (e1[k] + e2[k]);
        dot1 += e1[k] * (eu[k] + e2[k]);
        dot2 += e2[k] * (eu[k] + e1[k]);
      }
      float s, s1, s2;
      loss_acc += neg_logsigmoid(wave_sum(dot) + w.user_bias[u], &s);
      loss_acc += neg_logsigmoid(wave_sum(dot1) + w.item_bias[a], &s1);
      loss_acc += neg_logsigmoid(wave_sum(dot2) + w.item_bias[b], &s2);
      const float dx = -s * scale, dx1 = -s1 * scale, dx2 = -s2 * scale;
#pragma unroll
      for (int k = 0; k < kT2vMaxNpl; ++k) {
        const int c = lane + kWave * k;
        if (c < D) {
          atomic_add_f32(g.user_emb + u * D + c, dx * (e1[k] + e2[k]) + dx1 * e1[k] + dx2 * e2[k]);
          atomic_add_f32(g.item_emb1 + a * D + c, dx * eu[k] + dx1 * (eu[k] + e2[k]) + dx2 * e2[k]);
          atomic_add_f32(g.item_emb2 + b * D + c, dx * eu[k] + dx1 * e1[k] + dx2 * (eu[k] + e1[k]));
        }
      }
      if (lane == 0) {
        atomic_add_f32(g.user_bias + u, dx);
        atomic_add_f32(g.item_bias + a, dx1);
        atomic_add_f32(g.item_bias + b, dx2);
      }
      continue;
    }
    float ru[kT2vMaxNpl], r1[kT2vMaxNpl], r2[kT2vMaxNpl];
    float y = 0.f, y1 = 0.f, y2 = 0.f;
#pragma unroll
    for (int k = 0; k < kT2vMaxNpl; ++k) {
      const int c = lane + kWave * k;
      const bool in = c < D;
      ru[k] = in ? w.user_emb[nu * D + c] : 0.f;
      r1[k] = in ? w.item_emb1[nb * D + c] : 0.f;   // triple2vec.py:46: item_emb1(neg_i_2)
      r2[k] = shared_items ? r1[k] : (in ? w.item_emb2[nb * D + c] : 0.f);
      y += ru[k] * eu[k];
      y1 += r1[k] * e1[k];
      y2 += r2[k] * e2[k];
    }
    float sy, sy1, sy2;  // sigmoid(+y): neg_logsigmoid(-y) returns -logsigmoid(-y) and sigmoid(y)
    loss_acc += neg_logsigmoid(-(wave_sum(y) + w.user_bias[nu]), &sy);
    loss_acc += neg_logsigmoid(-(wave_sum(y1) + w.item_bias[na]), &sy1);
    loss_acc += neg_logsigmoid(-(wave_sum(y2) + w.item_bias[nb]), &sy2);
    const float dy = sy * scale, dy1 = sy1 * scale, dy2 = sy2 * scale;
#pragma unroll
    for (int k = 0; k < kT2vMaxNpl; ++k) {
      const int c = lane + kWave * k;
      if (c < D) {
        atomic_add_f32(g.user_emb + nu * D + c, dy * eu[k]);
        if (shared_items) {  // one row, one atomic
          atomic_add_f32(g.item_emb1 + nb * D + c, dy1 * e1[k] + dy2 * e2[k]);
        } else {
          atomic_add_f32(g.item_emb1 + nb * D + c, dy1 * e1[k]);
          atomic_add_f32(g.item_emb2 + nb * D + c, dy2 * e2[k]);
        }
        atomic_add_f32(g.user_emb + u * D + c, dy * ru[k]);
        atomic_add_f32(g.item_emb1 + a * D + c, dy1 * r1[k]);
        atomic_add_f32(g.item_emb2 + b * D + c, dy2 * r2[k]);
      }
    }
    if (lane == 0) {
      atomic_add_f32(g.user_bias + nu, dy);
      atomic_add_f32(g.item_bias + na, dy1);
      atomic_add_f32(g.item_bias + nb, dy2);
    }
  }
  publish_partials<kWavesPerBlock>(loss_acc, 0.f, 0.f, scale, scratch);
  if (stepper) step_store_advanced(stats, step_state);
}

// scores[k] = <U[u], (E1[i] + E2[i]) / 2>   (Triple2vec.predict, triple2vec.py:94-104)
__global__ __launch_bounds__(kBlock) void t2v_predict_kernel(hiprec_t2v_tables w,
                                                             const int64_t* __restrict__ users,
                                                             const int64_t* __restrict__ items,
                                                             int64_t n, float* __restrict__ scores,
                                                             hiprec_stats* stats) {
  const int lane = lane_id();
  const int D = w.dim;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block(); t < n;
       t += static_cast<int64_t>(gridDim.x) * kWavesPerBlock) {
    const int64_t u = users[t], i = items[t];
    const bool u_ok = in_range(u, w.n_users), i_ok = in_range(i, w.n_items);
    if (!(u_ok && i_ok)) {
      if (lane == 0) {
        atomicOr(&stats->status,
                 (u_ok ? 0u : HIPREC_STATUS_USER_OOB) | (i_ok ? 0u : HIPREC_STATUS_ITEM_OOB));
        scores[t] = __builtin_nanf("");
      }
      continue;
    }
    float dot = 0.f;
    for (int c = lane; c < D; c += kWave)
      dot += w.user_emb[u * D + c] * ((w.item_emb1[i * D + c] + w.item_emb2[i * D + c]) / 2.f);
    dot = wave_sum(dot);
    if (lane == 0) scores[t] = dot;
  }
}

// AliasTable.sample (utils/alias_table.py:82-97) on the device, counter-based: element e draws
// column = floor(u1 * vocab) and keeps it when u2 < prob[column], else takes alias[column]; (u1, u2)
// come from splitmix64(seed ^ splitmix64(e)) and its successor.  labels == NULL: identity.
__global__ __launch_bounds__(kBlock) void alias_sample_kernel(const double* __restrict__ prob,
                                                              const int64_t* __restrict__ alias,
                                                              const int64_t* __restrict__ labels,
                                                              int64_t vocab, uint64_t seed,
                                                              int64_t* __restrict__ out, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < n; e += stride) {
    const uint64_t h1 = splitmix64(seed ^ splitmix64(static_cast<uint64_t>(e)));
    const uint64_t h2 = splitmix64(h1);
    const uint64_t col = __umul64hi(h1, static_cast<uint64_t>(vocab));
    const double u2 = static_cast<double>(h2 >> 11) * (1.0 / 9007199254740992.0);
    const int64_t pick = u2 < prob[col] ? static_cast<int64_t>(col) : alias[col];
    out[e] = labels ? labels[pick] : pick;
  }
}

inline int t2v_grid(int64_t n) {
  const int64_t want = (n + kWavesPerBlock - 1) / kWavesPerBlock;
  return static_cast<int>(want < 1 ? 1 : (want > kT2vMaxBlocks ? kT2vMaxBlocks : want));
}

inline int check_t2v_tables(const hiprec_t2v_tables* w, const char* name) {
  HIPREC_REQUIRE(w != nullptr, "%s is NULL", name);
  HIPREC_REQUIRE(w->user_emb && w->item_emb1 && w->item_emb2 && w->user_bias && w->item_bias,
                 "%s: NULL tensor pointer", name);
  HIPREC_REQUIRE(w->n_users > 0 && w->n_items > 0 && w->dim > 0 && w->dim <= kT2vMaxNpl * kWave,
                 "%s: Triple2vec needs positive sizes and dim <= %d (got %d)", name, kT2vMaxNpl * kWave,
                 w->dim);
  return 0;
}

}  // namespace hiprec

using namespace hiprec;

extern "C" int hiprec_t2v_grad(const hiprec_t2v_tables* w, const hiprec_t2v_tables* g,
                               const int64_t* pos_u, const int64_t* pos_i1, const int64_t* pos_i2,
                               const int64_t* neg_u, const int64_t* neg_i1, const int64_t* neg_i2,
                               int64_t batch, int32_t n_neg, float scale, hiprec_stats* stats,
                               void* scratch, size_t scratch_bytes, void* stream) {
  if (int rc = check_t2v_tables(w, "w")) return rc;
  if (int rc = check_t2v_tables(g, "g")) return rc;
  HIPREC_REQUIRE(w->n_users == g->n_users && w->n_items == g->n_items && w->dim == g->dim,
                 "weight / gradient shapes differ");
  HIPREC_REQUIRE((w->item_emb1 == w->item_emb2) == (g->item_emb1 == g->item_emb2),
                 "item_emb2 must alias item_emb1 in both w and g or in neither");
  HIPREC_REQUIRE(stats && scratch, "NULL stats/scratch");
  HIPREC_REQUIRE(batch >= 0 && n_neg >= 0, "negative batch / n_neg");
  HIPREC_REQUIRE(batch == 0 || (pos_u && pos_i1 && pos_i2), "NULL positive index arrays");
  HIPREC_REQUIRE(batch == 0 || n_neg == 0 || (neg_u && neg_i1 && neg_i2), "NULL negative index arrays");
  if (scratch_bytes < kScratchBytes) {
    set_error("scratch %zu B < %zu B", scratch_bytes, kScratchBytes);
    return HIPREC_E_SCRATCH;
  }
  t2v_grad_kernel<<<t2v_grid(batch * (n_neg + 1)), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      *w, *g, pos_u, pos_i1, pos_i2, neg_u, neg_i1, neg_i2, batch, n_neg, scale, stats,
      static_cast<Scratch*>(scratch));
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_t2v_predict(const hiprec_t2v_tables* w, const int64_t* users,
                                  const int64_t* items, int64_t n, float* scores,
                                  hiprec_stats* stats, void* stream) {
  if (int rc = check_t2v_tables(w, "w")) return rc;
  HIPREC_REQUIRE(n >= 0, "negative n");
  if (n == 0) return 0;
  HIPREC_REQUIRE(users && items && scores && stats, "NULL pointer");
  t2v_predict_kernel<<<t2v_grid(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(*w, users, items, n,
                                                                                  scores, stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_alias_sample(const double* prob, const int64_t* alias, const int64_t* labels,
                                   int64_t vocab, uint64_t seed, int64_t* out, int64_t n,
                                   void* stream) {
  HIPREC_REQUIRE(vocab > 0 && n >= 0, "bad sizes (vocab %lld, n %lld)", (long long)vocab, (long long)n);
  if (n == 0) return 0;
  HIPREC_REQUIRE(prob && alias && out, "NULL pointer");
  alias_sample_kernel<<<grid_for_threads(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      prob, alias, labels, vocab, seed, out, n);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

// Triple2vecEngine.train_an_epoch (triple2vec.py:126-169) over resident arrays in visiting order:
// pos_*[n_triples], neg_*[n_triples * n_neg]; every batch is hiprec_t2v_grad + hiprec_opt_dense_step
// over the first n_sweep floats of the flat buffers, enqueued back to back from C.
extern "C" int hiprec_t2v_epoch(const hiprec_t2v_tables* w, const hiprec_t2v_tables* g,
                                const int64_t* pos_u, const int64_t* pos_i1, const int64_t* pos_i2,
                                const int64_t* neg_u, const int64_t* neg_i1, const int64_t* neg_i2,
                                int64_t n_triples, int64_t batch, int32_t n_neg, float scale, int kind,
                                double lr, double beta1, double beta2, double eps, float* flat_w,
                                float* flat_g, float* flat_m, float* flat_v, int64_t n_sweep,
                                hiprec_stats* stats, void* scratch, size_t scratch_bytes,
                                void* stream) {
  HIPREC_REQUIRE(n_triples >= 0 && batch > 0 && n_neg >= 0, "bad n_triples/batch/n_neg");
  HIPREC_REQUIRE(flat_w && flat_g && n_sweep > 0, "the dense optimizer needs the flat buffers");
  if (int rc = hiprec_stats_begin_epoch(stats, stream)) return rc;
  for (int64_t off = 0; off < n_triples; off += batch) {
    const int64_t b = (n_triples - off < batch) ? (n_triples - off) : batch;
    const int64_t noff = off * n_neg;
    if (int rc = hiprec_t2v_grad(w, g, pos_u + off, pos_i1 + off, pos_i2 + off,
                                 n_neg ? neg_u + noff : nullptr, n_neg ? neg_i1 + noff : nullptr,
                                 n_neg ? neg_i2 + noff : nullptr, b, n_neg, scale, stats, scratch,
                                 scratch_bytes, stream))
      return rc;
    if (int rc = hiprec_opt_dense_step(kind, flat_w, flat_g, flat_m, flat_v, n_sweep, lr, beta1, beta2,
                                       eps, stats, scratch, -1, stream))
      return rc;
  }
  return 0;
}
