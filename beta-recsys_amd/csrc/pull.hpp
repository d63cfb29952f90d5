// Owner-pulls steps (csrc/mf_owned.hip: plain SGD; csrc/lazy_opt.hip: exact lazy Adam / RMSprop): what the apply
// launches share -- the record lists of hiprec_batch_row_contrib (csrc/ownership.hip), the contribution buffer the
// gradient launch fills with plain stores, and the lane-group summation of a row's range.
#pragma once
#include "common.hpp"

namespace hiprec {

constexpr int kPullBlock = 1024;
constexpr int kPullWaves = kPullBlock / kWave;
constexpr int kPullDepth = 8;             // contribution rows a wave keeps in flight

struct PullApply {
  float* w;
  int64_t n_users, o_ie, o_ub, o_ib;
  int32_t dim, begin_epoch, count_step;
  const float* cbuf;
  const float* cbias;
  const int4* rows;              // this batch's records
  int64_t row_cap;
  const int32_t* counts;         // this batch's {short rows, long rows, contributions, -}
  float* gb;                     // the scalar bias
  float lr;
  // the row-sharded step (REMOTE apply, csrc/mf_owned.hip): keys >= n_users are SLOTS of the step's exchange buffer --
  // their sums are stored to slot_out [n_slots][dim + 1] (row | bias) for the way back to the items' owners -- and the
  // stats block writes this rank's loss / regulariser / scalar-bias partials into the n_dest extra rows instead of
  // booking them (the 3-float all-reduce rides in the gradient exchange)
  float* slot_out;
  const int32_t* extra_rows;
  int32_t n_dest;
};

// The gradient launch of an owner-pulls step on local tables (csrc/mf_owned.hip): mf_bpr_owned_kernel<.., PULL> over
// one batch.  cidx_* = the batch's slices of hiprec_batch_row_contrib's cidx (user / positive / negative role);
// count_step: block 0 counts the step in hiprec_stats (the lazy optimizers' apply launch reads the clock, the SGD form
// counts in its apply launch instead).  Loss partials go to `scratch`.
int launch_pull_grad(const float* w_flat, int64_t n_users, int64_t n_items, int32_t dim, const int64_t* users,
                     const int64_t* pos, const int64_t* neg, const int32_t* cidx_u, const int32_t* cidx_p,
                     const int32_t* cidx_n, float* cbuf, float* cbias, int64_t batch, float reg_coef, float lr,
                     int32_t count_step, hiprec_stats* stats, void* scratch, hipStream_t stream);

#if defined(__HIPCC__)

// dim % 4 == 0: LPR lanes x 16 bytes cover one row; a wave works on 64 / LPR rows at once.
// Contributions first, first + stride, ... < cnt of the range at `start` -> g (this lane's 4 columns) and, summed over
// the lane group, gb; DEPTH rows requested per trip.  `trips` is uniform over the wave.
template <int LPR, int DEPTH>
__device__ __forceinline__ void pull_sum_range(const PullApply& f, int start, int first, int cnt, int stride, int trips,
                                               int sl, bool col, float4& g, float& gb) {
  const int D = f.dim;
  for (int t = 0; t < trips; ++t) {
    const int j0 = first + t * DEPTH * stride;
    float4 v[DEPTH];
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) {
      const int j = j0 + q * stride;
      v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col && j < cnt) v[q] = *reinterpret_cast<const float4*>(f.cbuf + static_cast<int64_t>(start + j) * D + sl * 4);
    }
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) {
      g.x += v[q].x;
      g.y += v[q].y;
      g.z += v[q].z;
      g.w += v[q].w;
    }
  }
  float b = 0.f;
  for (int j = first + sl * stride; j < cnt; j += LPR * stride) b += f.cbias[start + j];
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) b += __shfl_xor(b, o);
  gb += b;
}

// where the row of a record's key lives, in floats relative to the flat parameters: the row's first element of this
// lane (sl * 4) and the row's bias element
__device__ __forceinline__ void pull_row_of(const PullApply& f, int key, int sl, int64_t* row, int64_t* bias) {
  const bool user = key < f.n_users;
  const int64_t r = user ? key : key - f.n_users;
  *row = (user ? r * f.dim : f.o_ie + r * f.dim) + sl * 4;
  *bias = user ? f.o_ub + r : f.o_ib + r;
}

#endif  // __HIPCC__

}  // namespace hiprec
