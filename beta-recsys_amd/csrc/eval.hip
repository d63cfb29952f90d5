// Ranking metrics of the reference's evaluation step on the device.
//
// Restates beta_rec/utils/evaluation.py:461-533 (merge_ranking_true_pred, relevancy "top_k"),
// :535-583 precision_at_k, :586-629 recall_at_k, :632-689 ndcg_at_k, :692-752 map_at_k, as they are
// called by core/eval_engine.py:49-87 evaluate(): the truth frame and the prediction frame are the
// SAME rows (eval_engine.py:60-68 builds pred_df from data_df's own user/item columns), so a
// candidate is a (score, rating) pair and "hit" means rating >= 1 (evaluation.py:492).
//
// One wavefront owns one user's candidates (a contiguous segment).  The top-k list is produced by k
// selection passes instead of a sort: pass r picks the smallest 64-bit key that is greater than the
// key picked in pass r-1, where key = (descending-orderable score bits, position in the segment).
// That is exactly pandas' nlargest(keep="first") + rank(method="first") order (evaluation.py:778-784,
// 516-518): score descending, ties by original row order.  k is tens and a segment is usually 101
// (leave-one-out + 100 negatives) to a few thousand rows and sits in L2, so k passes are cheaper than
// a segmented sort and need no workspace.  All metric arithmetic is fp64 like the pandas code.
#include "common.hpp"

namespace hiprec {
namespace {

struct KList {
  int32_t n;
  int32_t k[HIPREC_RANK_MAX_K];
};

__device__ inline uint64_t rank_key(float score, uint32_t pos) {
  if (score == 0.0f) score = 0.0f;  // -0 == +0 for pandas; make the bit patterns agree
  uint32_t u = __float_as_uint(score);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending-orderable
  return (static_cast<uint64_t>(~u) << 32) | pos;  // smaller key == better candidate
}

__device__ inline uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    uint64_t o = __shfl_xor(static_cast<unsigned long long>(v), off, kWave);
    v = o < v ? o : v;
  }
  return v;
}

__device__ inline int wave_sum_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// per_user row layout: [is_common, then for each k: precision, recall, ndcg, map]
__global__ __launch_bounds__(kBlock) void rank_metrics_kernel(
    const int64_t* __restrict__ seg_ptr, int64_t n_seg, const float* __restrict__ scores,
    const float* __restrict__ ratings, KList kl, double* __restrict__ per_user) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int stride = 1 + 4 * kl.n;

  int k_max = 0;
#pragma unroll
  for (int j = 0; j < HIPREC_RANK_MAX_K; ++j)
    if (j < kl.n && kl.k[j] > k_max) k_max = kl.k[j];

  for (int64_t sid = wave; sid < n_seg; sid += n_waves) {
    const int64_t s = seg_ptr[sid];
    const int64_t len = seg_ptr[sid + 1] - s;
    const float* sc = scores + s;
    const float* rt = ratings + s;
    double* row = per_user + sid * stride;

    int actual = 0;
    for (int64_t i = lane; i < len; i += kWave) actual += rt[i] >= 1.0f ? 1 : 0;
    actual = wave_sum_i32(actual);
    if (actual == 0) {  // not in common_users (evaluation.py:495-498): contributes to nothing
      if (lane < stride) row[lane] = 0.0;
      continue;
    }

    int hits[HIPREC_RANK_MAX_K];
    double dcg[HIPREC_RANK_MAX_K], ap[HIPREC_RANK_MAX_K];
#pragma unroll
    for (int j = 0; j < HIPREC_RANK_MAX_K; ++j) {
      hits[j] = 0;
      dcg[j] = 0.0;
      ap[j] = 0.0;
    }

    const int64_t passes = len < k_max ? len : k_max;
    uint64_t lower = 0;
    for (int64_t r = 1; r <= passes; ++r) {
      uint64_t best = ~0ull;
      for (int64_t i = lane; i < len; i += kWave) {
        const uint64_t key = rank_key(sc[i], static_cast<uint32_t>(i));
        if (key >= lower && key < best) best = key;
      }
      best = wave_min_u64(best);
      lower = best + 1;
      const uint32_t pos = static_cast<uint32_t>(best & 0xffffffffu);
      if (rt[pos] >= 1.0f) {
        const double gain = 1.0 / log1p(static_cast<double>(r));
#pragma unroll
        for (int j = 0; j < HIPREC_RANK_MAX_K; ++j) {
          if (j < kl.n && r <= kl.k[j]) {
            hits[j] += 1;
            dcg[j] += gain;
            ap[j] += static_cast<double>(hits[j]) / static_cast<double>(r);
          }
        }
      }
    }

    if (lane == 0) {
      row[0] = 1.0;
#pragma unroll
      for (int j = 0; j < HIPREC_RANK_MAX_K; ++j) {
        if (j >= kl.n) continue;
        const int k = kl.k[j];
        const int ideal = actual < k ? actual : k;
        double idcg = 0.0;
        for (int t = 1; t <= ideal; ++t) idcg += 1.0 / log1p(static_cast<double>(t));
        double* m = row + 1 + 4 * j;
        m[0] = static_cast<double>(hits[j]) / static_cast<double>(k);
        m[1] = static_cast<double>(hits[j]) / static_cast<double>(actual);
        m[2] = dcg[j] / idcg;
        m[3] = ap[j] / static_cast<double>(actual);
      }
    }
  }
}

// One block per column, fixed summation order -> run-to-run identical results.
__global__ __launch_bounds__(kBlock) void rank_reduce_kernel(const double* __restrict__ per_user,
                                                             int64_t n_seg, int stride,
                                                             double* __restrict__ out) {
  __shared__ double part[kBlock];
  __shared__ double n_common;
  const int col = blockIdx.x;
  double acc = 0.0, cnt = 0.0;
  for (int64_t i = threadIdx.x; i < n_seg; i += kBlock) {
    acc += per_user[i * stride + col];
    cnt += per_user[i * stride];
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  const double total = part[0];
  __syncthreads();
  part[threadIdx.x] = cnt;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    n_common = part[0];
    // every metric returns 0.0 when nothing was hit / no common user (evaluation.py:581-582)
    out[col] = col == 0 ? n_common : (n_common > 0.0 ? total / n_common : 0.0);
  }
}

}  // namespace
}  // namespace hiprec

using namespace hiprec;

extern "C" size_t hiprec_rank_metrics_workspace_bytes(int64_t n_segments, int32_t n_k) {
  if (n_segments < 0 || n_k < 0) return 0;
  return sizeof(double) * static_cast<size_t>(n_segments) * (1 + 4 * static_cast<size_t>(n_k));
}

extern "C" int hiprec_rank_metrics(const int64_t* seg_ptr, int64_t n_segments, const float* scores,
                                   const float* ratings, const int32_t* k_list_host, int32_t n_k,
                                   double* workspace, size_t workspace_bytes, double* out,
                                   void* stream) {
  HIPREC_REQUIRE(n_k >= 1 && n_k <= HIPREC_RANK_MAX_K, "rank_metrics: n_k=%d outside 1..%d", n_k,
                 HIPREC_RANK_MAX_K);
  HIPREC_REQUIRE(k_list_host != nullptr && out != nullptr, "rank_metrics: null k_list/out");
  HIPREC_REQUIRE(n_segments >= 0, "rank_metrics: n_segments=%lld", (long long)n_segments);
  KList kl{};
  kl.n = n_k;
  for (int j = 0; j < n_k; ++j) {
    HIPREC_REQUIRE(k_list_host[j] >= 1, "rank_metrics: k[%d]=%d must be >= 1", j, k_list_host[j]);
    kl.k[j] = k_list_host[j];
  }
  const int stride = 1 + 4 * n_k;
  auto s = static_cast<hipStream_t>(stream);
  if (n_segments == 0) {
    HIPREC_TRY(hipMemsetAsync(out, 0, sizeof(double) * stride, s));
    return 0;
  }
  HIPREC_REQUIRE(seg_ptr && scores && ratings && workspace, "rank_metrics: null pointer");
  HIPREC_REQUIRE(workspace_bytes >= hiprec_rank_metrics_workspace_bytes(n_segments, n_k),
                 "rank_metrics: workspace %zu B < %zu B", workspace_bytes,
                 hiprec_rank_metrics_workspace_bytes(n_segments, n_k));
  rank_metrics_kernel<<<grid_for_waves(n_segments), kBlock, 0, s>>>(seg_ptr, n_segments, scores,
                                                                     ratings, kl, workspace);
  HIPREC_TRY(hipGetLastError());
  rank_reduce_kernel<<<stride, kBlock, 0, s>>>(workspace, n_segments, stride, out);
  HIPREC_TRY(hipGetLastError());
  return 0;
}
