// Negative sampling on the device: the data step in front of the training path.
//
// Restates what beta_rec/data/base_data.py builds once per training run on the host:
//   :218-253 instance_bpr_loader      one negative per (user, positive) row,
//   :182-216 instance_bce_loader      num_negative negatives per row (rating 0),
//   :254-288 instance_mul_neg_loader  num_negative negatives per row,
// each drawn with random.sample(list(set(item_id_pool) - positive_items(user)), k): k DISTINCT items,
// uniform over the items of the pool the user never interacted with in the training frame.  The
// reference walks the frame with pandas iterrows (minutes for ML-1M); here one thread produces one
// negative with no rejection loop:
//   r  = the j-th element of a keyed pseudo-random permutation of [0, M), M = n_items - deg(user)
//        (Feistel bijection keyed by splitmix64(seed, row): j = 0..k-1 are distinct by construction);
//   id = the r-th item NOT in the user's sorted positive list = r + t, where t is the smallest index
//        with pos[t] - t > r (binary search; pos[t] - t is non-decreasing for a strictly increasing list).
// Python's Mersenne-Twister stream cannot be replayed on a GPU and set iteration order is an
// implementation detail, so parity is (a) bit-exact against oracle/sampler_numpy.py, which restates
// this generator and the "r-th missing item" map, and (b) distributional against the reference's own
// loader output (tests/golden/sampler_*.npz): support, distinctness, uniformity.
#include "common.hpp"

namespace hiprec {
namespace {

__global__ __launch_bounds__(kBlock) void sample_negatives_kernel(
    const int64_t* __restrict__ user_ptr, const int64_t* __restrict__ pos_sorted, int64_t n_users,
    int64_t n_items, const int64_t* __restrict__ users, int64_t n_rows, int32_t k, uint64_t seed,
    int64_t* __restrict__ out, hiprec_stats* stats) {
  const int64_t total = n_rows * k;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; idx < total; idx += stride) {
    const int64_t row = idx / k;
    const int64_t j = idx - row * k;
    const int64_t u = users[row];
    if (static_cast<uint64_t>(u) >= static_cast<uint64_t>(n_users)) {
      atomicOr(&stats->status, HIPREC_STATUS_USER_OOB);
      out[idx] = -1;
      continue;
    }
    const int64_t beg = user_ptr[u];
    const int64_t deg = user_ptr[u + 1] - beg;
    const int64_t m = n_items - deg;  // items this user never touched
    if (m < k) {                       // random.sample would raise ValueError
      atomicOr(&stats->status, HIPREC_STATUS_NEG_EXHAUSTED);
      out[idx] = -1;
      continue;
    }
    const uint64_t row_seed = splitmix64(seed ^ splitmix64(static_cast<uint64_t>(row)));
    const uint64_t r = feistel_permute(static_cast<uint64_t>(j), static_cast<uint64_t>(m),
                                       feistel_half_bits(static_cast<uint64_t>(m)), row_seed);
    const int64_t* pos = pos_sorted + beg;
    int64_t lo = 0, hi = deg;  // smallest t in [0, deg] with t == deg or pos[t] - t > r
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (pos[mid] - mid > static_cast<int64_t>(r)) {
        hi = mid;
      } else {
        lo = mid + 1;
      }
    }
    out[idx] = static_cast<int64_t>(r) + lo;
  }
}

}  // namespace
}  // namespace hiprec

using namespace hiprec;

extern "C" int hiprec_sample_negatives(const int64_t* user_ptr, const int64_t* pos_sorted,
                                       int64_t n_users, int64_t n_items, const int64_t* users,
                                       int64_t n_rows, int32_t k, uint64_t seed, int64_t* out,
                                       hiprec_stats* stats, void* stream) {
  HIPREC_REQUIRE(n_users > 0 && n_items > 0 && n_rows >= 0 && k >= 1, "bad sizes (n_users %lld, n_items %lld, "
                 "n_rows %lld, k %d)", (long long)n_users, (long long)n_items, (long long)n_rows, k);
  HIPREC_REQUIRE(n_items < (1ll << 61), "n_items too large for the Feistel domain");
  if (n_rows == 0) return 0;
  HIPREC_REQUIRE(user_ptr && users && out && stats, "NULL pointer");
  sample_negatives_kernel<<<grid_for_threads(n_rows * k), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      user_ptr, pos_sorted, n_users, n_items, users, n_rows, k, seed, out, stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}
