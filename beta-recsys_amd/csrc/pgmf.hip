// PairwiseGMF (the CMN pre-training model) on the MF building blocks — SURVEY.md §8f rank 4.
//
//   beta_rec/models/pairwise_gmf.py:48-62   forward: s± = relu(v . (U[u] * I[i±]))
//   beta_rec/models/pairwise_gmf.py:82-116  train_single_batch: loss = mean(-log(sigmoid(s+ - s-) + 1e-12))
//                                           + pretrain_l2_lambda * ||v||_2 ; backward;
//                                           clip_grad_norm_(parameters, grad_clip); optimizer.step()
//   beta_rec/models/pairwise_gmf.py:144-158 the engine's own bpr_loss (eps inside the log)
//
// Layout in HBM: one flat fp32 buffer [user_memory U*D | item_memory I*D | v D] for the weights and
// one of the same shape for the dense gradient, so that the norm clip and the dense optimizer sweep
// (optim.hip) are single streaming passes.  The step is
//   pgmf_bpr_grad_kernel  one wave per triple: 3 row gathers, 2 dots, row gradients scattered with
//                         fp32 atomics, the gradient of v summed per wave in registers and per block
//                         in LDS, one [D] partial per block written to the workspace (no atomics on
//                         the D hot addresses)
//   pgmf_finish_kernel    1 block of 1024 threads: deterministic sum of the per-block partials of
//                         grad v, + the gradient and the value of lambda * ||v||
//   clip_sumsq_kernel / clip_scale_kernel   torch.nn.utils.clip_grad_norm_ over the flat gradient
//   hiprec_opt_dense_step                   the optimizer sweep, shared with MF / NCF / LightGCN
// HBM-bound integer-indexed row traffic: 3 rows read + 3 rows accumulated per triple, like BPR-MF.
#include "common.hpp"

namespace hiprec {

constexpr int kPgmfMaxNpl = 4;       // dim <= 256: columns lane, lane+64, ...
constexpr int kPgmfMaxBlocks = 1024; // 4 triples per block and trip
constexpr int kClipMaxBlocks = 1024;

__global__ __launch_bounds__(kBlock) void pgmf_bpr_grad_kernel(
    hiprec_pgmf_tables w, hiprec_pgmf_tables g, const int64_t* __restrict__ users,
    const int64_t* __restrict__ pos, const int64_t* __restrict__ neg, int64_t batch, float inv_batch,
    hiprec_stats* stats, Scratch* scratch, float* __restrict__ vpart) {
  __shared__ float s_v[kWavesPerBlock][kPgmfMaxNpl * kWave];
  const int lane = lane_id();
  const int wv = wave_in_block();
  const int D = w.dim;

  const bool stepper = blockIdx.x == 0 && threadIdx.x == 0;
  StepState step_state{};
  if (stepper) step_state = step_load(stats);

  float vv[kPgmfMaxNpl], gv[kPgmfMaxNpl];
#pragma unroll
  for (int k = 0; k < kPgmfMaxNpl; ++k) {
    const int c = lane + kWave * k;
    vv[k] = c < D ? w.v[c] : 0.f;
    gv[k] = 0.f;
  }

  float loss_acc = 0.f;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wv; t < batch;
       t += static_cast<int64_t>(gridDim.x) * kWavesPerBlock) {
    const int64_t u = users[t], p = pos[t], n = neg[t];
    const bool u_ok = static_cast<uint64_t>(u) < static_cast<uint64_t>(w.n_users);
    const bool i_ok = static_cast<uint64_t>(p) < static_cast<uint64_t>(w.n_items) &&
                      static_cast<uint64_t>(n) < static_cast<uint64_t>(w.n_items);
    if (!(u_ok && i_ok)) {
      if (lane == 0)
        atomicOr(&stats->status,
                 (u_ok ? 0u : HIPREC_STATUS_USER_OOB) | (i_ok ? 0u : HIPREC_STATUS_ITEM_OOB));
      continue;
    }
    const float* ur = w.user_memory + u * D;
    const float* pr = w.item_memory + p * D;
    const float* nr = w.item_memory + n * D;
    float uu[kPgmfMaxNpl], pp[kPgmfMaxNpl], nn[kPgmfMaxNpl];
    float dot_p = 0.f, dot_n = 0.f;
#pragma unroll
    for (int k = 0; k < kPgmfMaxNpl; ++k) {
      const int c = lane + kWave * k;
      const bool in = c < D;
      uu[k] = in ? ur[c] : 0.f;
      pp[k] = in ? pr[c] : 0.f;
      nn[k] = in ? nr[c] : 0.f;
      dot_p += vv[k] * (uu[k] * pp[k]);
      dot_n += vv[k] * (uu[k] * nn[k]);
    }
    dot_p = wave_sum(dot_p);
    dot_n = wave_sum(dot_n);
    // relu, then the engine's bpr_loss: -log(sigmoid(x) + 1e-12)
    const float x = fmaxf(dot_p, 0.f) - fmaxf(dot_n, 0.f);
    const float y = sigmoid_f32(x);
    loss_acc += -logf(y + 1e-12f);
    // d loss / d x = -(y (1 - y)) / (y + eps) / B   (log backward, then sigmoid backward)
    const float dx = -(inv_batch / (y + 1e-12f)) * ((1.f - y) * y);
    const float dp = dot_p > 0.f ? dx : 0.f;   // threshold_backward: gradient only where s > 0
    const float dn = dot_n > 0.f ? -dx : 0.f;
    if (dp == 0.f && dn == 0.f) continue;      // both relus closed: every gradient term is +0
    float* gur = g.user_memory + u * D;
    float* gpr = g.item_memory + p * D;
    float* gnr = g.item_memory + n * D;
#pragma unroll
    for (int k = 0; k < kPgmfMaxNpl; ++k) {
      const int c = lane + kWave * k;
      if (c < D) {
        const float up = uu[k] * pp[k], un = uu[k] * nn[k];
        gv[k] += dp * up + dn * un;
        atomic_add_f32(gur + c, vv[k] * (dp * pp[k] + dn * nn[k]));
        if (dp != 0.f) atomic_add_f32(gpr + c, dp * (vv[k] * uu[k]));
        if (dn != 0.f) atomic_add_f32(gnr + c, dn * (vv[k] * uu[k]));
      }
    }
  }

  // grad v: per-wave registers -> LDS -> one [D] partial per block
#pragma unroll
  for (int k = 0; k < kPgmfMaxNpl; ++k) s_v[wv][lane + kWave * k] = gv[k];
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += kBlock) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kWavesPerBlock; ++i) s += s_v[i][c];
    vpart[static_cast<int64_t>(blockIdx.x) * D + c] = s;
  }
  publish_partials<kWavesPerBlock>(loss_acc, 0.f, 0.f, inv_batch, scratch);
  if (stepper) step_store_advanced(stats, step_state);
}

// One block of kFinishThreads.  g.v[c] += sum_b vpart[b][c] + lambda * v[c] / ||v||; loss partial +=
// lambda * ||v||.  Thread (s, c) sums the partials of blocks b = s, s + n_slices, ... with four loads in
// flight (one thread per column walking all blocks serially cost 57 us at 256 blocks: 256 dependent
// L2 round trips); the slices are then added in a fixed order, so the result is deterministic.
// (Folding this into the gradient kernel with a "last block done" counter was measured and lost:
// the agent-scope fences it needs write the L2 of every XCD back, 24.6 us vs 4.4 + 4.7 us.)
constexpr int kFinishThreads = 1024;

__global__ __launch_bounds__(kFinishThreads) void pgmf_finish_kernel(hiprec_pgmf_tables w,
                                                                     hiprec_pgmf_tables g,
                                                                     const float* __restrict__ vpart,
                                                                     int n_blocks, float l2_lambda,
                                                                     Scratch* scratch) {
  __shared__ float s_part[kFinishThreads];
  __shared__ float s_sq[kPgmfMaxNpl * kWave];
  const int D = w.dim;
  const int n_slices = kFinishThreads / D;  // >= 4 (D <= 256)
  const int c = threadIdx.x % D;
  const int sl = threadIdx.x / D;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (sl < n_slices) {
    int b = sl;
    for (; b + 3 * n_slices < n_blocks; b += 4 * n_slices) {
      a0 += vpart[static_cast<int64_t>(b) * D + c];
      a1 += vpart[static_cast<int64_t>(b + n_slices) * D + c];
      a2 += vpart[static_cast<int64_t>(b + 2 * n_slices) * D + c];
      a3 += vpart[static_cast<int64_t>(b + 3 * n_slices) * D + c];
    }
    for (; b < n_blocks; b += n_slices) a0 += vpart[static_cast<int64_t>(b) * D + c];
  }
  s_part[threadIdx.x] = (a0 + a1) + (a2 + a3);
  const float vc = static_cast<int>(threadIdx.x) < D ? w.v[threadIdx.x] : 0.f;
  if (threadIdx.x < kPgmfMaxNpl * kWave) s_sq[threadIdx.x] = vc * vc;
  __syncthreads();
  for (int s = kPgmfMaxNpl * kWave / 2; s > 0; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s) s_sq[threadIdx.x] += s_sq[threadIdx.x + s];
    __syncthreads();
  }
  const float l2 = sqrtf(s_sq[0]);
  if (static_cast<int>(threadIdx.x) < D) {
    float sum = 0.f;
    for (int i = 0; i < n_slices; ++i) sum += s_part[i * D + threadIdx.x];
    // sqrt backward then pow backward: lambda / (2 l2) * 2 v  (0/0 -> NaN exactly as autograd gives)
    g.v[threadIdx.x] += sum + (l2_lambda / (2.f * l2)) * (2.f * vc);
  }
  if (threadIdx.x == 0) {
    const uint32_t n = scratch->n_partials;
    scratch->partials[n] = make_float4(l2_lambda * l2, 0.f, 0.f, 0.f);
    scratch->n_partials = n + 1;
  }
}

// torch.nn.utils.clip_grad_norm_(params, max_norm) with norm_type 2 over one flat gradient:
// total = ||g||_2, coef = min(max_norm / (total + 1e-6), 1), g *= coef.
__global__ __launch_bounds__(kBlock) void clip_sumsq_kernel(const float* __restrict__ g, int64_t n,
                                                            double* __restrict__ ws) {
  __shared__ double s_p[kBlock];
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  const int64_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float acc = 0.f;
  double total = 0.0;
  int folded = 0;
  for (int64_t i = tid; i < n4; i += stride) {
    const float4 x = g4[i];
    acc += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
    if (++folded == 64) {  // bound the fp32 run length on very large tables
      total += acc;
      acc = 0.f;
      folded = 0;
    }
  }
  for (int64_t i = (n4 << 2) + tid; i < n; i += stride) acc += g[i] * g[i];
  total += acc;
  s_p[threadIdx.x] = total;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s) s_p[threadIdx.x] += s_p[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) ws[2 + blockIdx.x] = s_p[0];
}

__global__ __launch_bounds__(kBlock) void clip_scale_kernel(float* __restrict__ g, int64_t n,
                                                            float max_norm, int n_partials,
                                                            double* __restrict__ ws) {
  __shared__ double s_p[kBlock];
  double t = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += kBlock) t += ws[2 + i];
  s_p[threadIdx.x] = t;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s) s_p[threadIdx.x] += s_p[threadIdx.x + s];
    __syncthreads();
  }
  const float total_norm = static_cast<float>(sqrt(s_p[0]));
  const float raw = max_norm / (total_norm + 1e-6f);
  const float coef = raw > 1.0f ? 1.0f : raw;  // torch.clamp(max=1.0): a NaN stays a NaN
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ws[0] = static_cast<double>(total_norm);
    ws[1] = static_cast<double>(coef);
  }
  if (coef >= 1.0f) return;  // g * 1.0f is g; a NaN norm falls through and poisons g like torch
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  const int64_t n4 = n >> 2;
  float4* g4 = reinterpret_cast<float4*>(g);
  for (int64_t i = tid; i < n4; i += stride) {
    float4 x = g4[i];
    x.x *= coef;
    x.y *= coef;
    x.z *= coef;
    x.w *= coef;
    g4[i] = x;
  }
  for (int64_t i = (n4 << 2) + tid; i < n; i += stride) g[i] *= coef;
}

inline int pgmf_grid(int64_t batch) {
  const int64_t want = (batch + kWavesPerBlock - 1) / kWavesPerBlock;
  return static_cast<int>(want < 1 ? 1 : (want > kPgmfMaxBlocks ? kPgmfMaxBlocks : want));
}

inline int clip_grid(int64_t n) {
  const int64_t want = ((n + 3) / 4 + kBlock - 1) / kBlock;
  return static_cast<int>(want < 1 ? 1 : (want > kClipMaxBlocks ? kClipMaxBlocks : want));
}

}  // namespace hiprec

using namespace hiprec;

extern "C" size_t hiprec_pgmf_workspace_bytes(int32_t dim) {
  return sizeof(float) * static_cast<size_t>(kPgmfMaxBlocks) * static_cast<size_t>(dim > 0 ? dim : 0);
}

extern "C" size_t hiprec_clip_workspace_bytes(void) { return sizeof(double) * (2 + kClipMaxBlocks); }

extern "C" int hiprec_pgmf_bpr_grad(const hiprec_pgmf_tables* w, const hiprec_pgmf_tables* g,
                                    const int64_t* users, const int64_t* pos, const int64_t* neg,
                                    int64_t batch, float inv_batch, float l2_lambda,
                                    hiprec_stats* stats, void* scratch, size_t scratch_bytes,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  HIPREC_REQUIRE(w && g, "NULL tables");
  HIPREC_REQUIRE(w->user_memory && w->item_memory && w->v && g->user_memory && g->item_memory && g->v,
                 "NULL tensor pointer");
  HIPREC_REQUIRE(w->n_users > 0 && w->n_items > 0 && w->dim > 0 && w->dim <= kPgmfMaxNpl * kWave,
                 "PairwiseGMF needs 0 < dim <= %d (got %d)", kPgmfMaxNpl * kWave, w->dim);
  HIPREC_REQUIRE(w->n_users == g->n_users && w->n_items == g->n_items && w->dim == g->dim,
                 "weight / gradient shapes differ");
  HIPREC_REQUIRE(stats && scratch && workspace, "NULL stats/scratch/workspace");
  HIPREC_REQUIRE(batch >= 0, "negative batch");
  HIPREC_REQUIRE(batch == 0 || (users && pos && neg), "NULL index arrays");
  if (scratch_bytes < kScratchBytes) {
    set_error("scratch %zu B < %zu B", scratch_bytes, kScratchBytes);
    return HIPREC_E_SCRATCH;
  }
  HIPREC_REQUIRE(workspace_bytes >= hiprec_pgmf_workspace_bytes(w->dim), "workspace %zu B < %zu B",
                 workspace_bytes, hiprec_pgmf_workspace_bytes(w->dim));
  auto s = static_cast<hipStream_t>(stream);
  const int grid = pgmf_grid(batch);
  pgmf_bpr_grad_kernel<<<grid, kBlock, 0, s>>>(*w, *g, users, pos, neg, batch, inv_batch, stats,
                                               static_cast<Scratch*>(scratch),
                                               static_cast<float*>(workspace));
  HIPREC_TRY(hipGetLastError());
  pgmf_finish_kernel<<<1, kFinishThreads, 0, s>>>(*w, *g, static_cast<const float*>(workspace), grid,
                                          l2_lambda, static_cast<Scratch*>(scratch));
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_clip_grad_norm(float* g, int64_t n, float max_norm, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  HIPREC_REQUIRE(n >= 0, "negative n");
  HIPREC_REQUIRE(workspace && (n == 0 || g), "NULL pointer");
  HIPREC_REQUIRE((reinterpret_cast<uintptr_t>(g) & 15) == 0, "gradient must be 16-byte aligned");
  HIPREC_REQUIRE(workspace_bytes >= hiprec_clip_workspace_bytes(), "workspace %zu B < %zu B",
                 workspace_bytes, hiprec_clip_workspace_bytes());
  auto s = static_cast<hipStream_t>(stream);
  const int grid = clip_grid(n);
  clip_sumsq_kernel<<<grid, kBlock, 0, s>>>(g, n, static_cast<double*>(workspace));
  HIPREC_TRY(hipGetLastError());
  clip_scale_kernel<<<grid, kBlock, 0, s>>>(g, n, max_norm, grid, static_cast<double*>(workspace));
  HIPREC_TRY(hipGetLastError());
  return 0;
}

// hiprec_clip_grad_norm + hiprec_opt_dense_step as two launches instead of three: the sums of squares, then ONE sweep
// that scales and steps (csrc/optim.hip, opt_dense_kernel<KIND, true>); the scaled gradient is never written.  The
// bits of w / m / v, of the cleared g and of workspace[0 .. 1] are those of the two calls.
extern "C" int hiprec_clip_opt_dense_step(int kind, float* w, float* g, float* m, float* v, int64_t n, double lr,
                                          double beta1, double beta2, double eps, hiprec_stats* stats,
                                          const void* scratch, int64_t scalar_index, float max_norm, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  HIPREC_REQUIRE(n >= 0, "negative n");
  HIPREC_REQUIRE(workspace && (n == 0 || g), "NULL pointer");
  HIPREC_REQUIRE((reinterpret_cast<uintptr_t>(g) & 15) == 0, "gradient must be 16-byte aligned");
  HIPREC_REQUIRE(workspace_bytes >= hiprec_clip_workspace_bytes(), "workspace %zu B < %zu B", workspace_bytes,
                 hiprec_clip_workspace_bytes());
  const int grid = clip_grid(n);
  clip_sumsq_kernel<<<grid, kBlock, 0, static_cast<hipStream_t>(stream)>>>(g, n, static_cast<double*>(workspace));
  HIPREC_TRY(hipGetLastError());
  return opt_dense_step_impl(kind, w, g, m, v, n, lr, beta1, beta2, eps, stats, scratch, scalar_index,
                             static_cast<double*>(workspace), grid, max_norm, stream);
}

// PairwiseGMFEngine.train_an_epoch (pairwise_gmf.py:118-142) over resident (user, pos, neg) arrays in
// visiting order: every batch is hiprec_pgmf_bpr_grad + hiprec_clip_opt_dense_step (clip + sweep in two launches),
// enqueued back to back from C (the python loop costs ~50 us per step, three times the kernels).
extern "C" int hiprec_pgmf_epoch(const hiprec_pgmf_tables* w, const hiprec_pgmf_tables* g,
                                 const int64_t* users, const int64_t* pos, const int64_t* neg,
                                 int64_t n_triples, int64_t batch, float l2_lambda, float max_norm,
                                 int kind, double lr, double beta1, double beta2, double eps,
                                 float* flat_w, float* flat_g, float* flat_m, float* flat_v,
                                 int64_t n_flat, hiprec_stats* stats, void* scratch,
                                 size_t scratch_bytes, void* workspace, size_t workspace_bytes,
                                 void* clip_workspace, size_t clip_workspace_bytes, void* stream) {
  HIPREC_REQUIRE(n_triples >= 0 && batch > 0, "bad n_triples/batch");
  HIPREC_REQUIRE(flat_w && flat_g && n_flat > 0, "the dense optimizer needs the flat buffers");
  if (int rc = hiprec_stats_begin_epoch(stats, stream)) return rc;
  for (int64_t off = 0; off < n_triples; off += batch) {
    const int64_t b = (n_triples - off < batch) ? (n_triples - off) : batch;
    if (int rc = hiprec_pgmf_bpr_grad(w, g, users + off, pos + off, neg + off, b,
                                      1.0f / static_cast<float>(b), l2_lambda, stats, scratch,
                                      scratch_bytes, workspace, workspace_bytes, stream))
      return rc;
    if (int rc = hiprec_clip_opt_dense_step(kind, flat_w, flat_g, flat_m, flat_v, n_flat, lr, beta1, beta2, eps, stats,
                                            scratch, -1, max_norm, clip_workspace, clip_workspace_bytes, stream))
      return rc;
  }
  return 0;
}
