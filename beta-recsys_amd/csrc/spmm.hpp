// CSR SpMM launcher of csrc/lightgcn.hip, shared with NGCF's propagation (ngcf.hip).
#pragma once
#include "common.hpp"

namespace hiprec {

// y = (A with dropped edges) x ; acc += y (acc may be NULL).  The kernel accumulates the slices of heavy
// rows with atomics, so y must be zero: y_is_zero = true when the caller cleared it already (several
// outputs in one memset), false to clear it here.
int launch_spmm(const hiprec_csr* a, const uint8_t* keep, float scale, const float* x, float* y, float* acc,
                int dim, hipStream_t st, bool y_is_zero = false);

// Column-sliced SpMM (csrc/spmm_sliced.hip) on sliced buffers [dim / W][n_rows][W]; acc_mode 0 / 1 (+=) / 2 (=).
// `edges` NULL = the graph's own values / columns (otherwise a step's stream from launch_step_values: float
// values, or uint16 columns for a factored graph, whose source and result are scaled by col_scale).
int sliced_width(int64_t n_rows, int dim);
int sliced_row_cap(int64_t n_rows, int dim);
// What a pass does with its finished rows besides (or instead of) writing ys / accs:
struct SlicedFlush {
  float* zero_out = nullptr;     // sliced buffer cleared row by row as a by-product
  float* final_out = nullptr;    // row-major: instead of ys / accs, the layer sum (accs + Y for acc_mode 1, Y otherwise)
  bool final_set = false;        //   is stored here (true) or added (false)
  int exp = 0;                   // experiment bits, -DHIPREC_SLICED_DEBUG builds only (spmm_sliced.hip)
};
int launch_spmm_sliced(const hiprec_sliced_csr* a, const void* edges, float scale, const float* xs, float* ys,
                       float* accs, int acc_mode, int dim, int W, hipStream_t st, SlicedFlush fl = SlicedFlush{});

// optional second output of an elementwise producer: the same rows in the sliced layout, times col_scale[row]
struct SlicedOut {
  float* xs = nullptr;
  const float* col_scale = nullptr;
  int64_t n_rows = 0;
  int w_shift = 0;
  __device__ __forceinline__ void put(int64_t r, int c, float v) const {
    if (xs == nullptr) return;
    const int64_t o = ((static_cast<int64_t>(c >> w_shift) * n_rows + r) << w_shift) + (c & ((1 << w_shift) - 1));
    xs[o] = col_scale ? v * col_scale[r] : v;
  }
};
// dropped values of one step for one or two graphs in one launch; draw: the device draw keep_draw(seed, step, edge)
// instead of reading keep[] (which `a`'s slots then fill in, when given)
// optional extra work of that launch: x (row-major [n_rows][dim]) into the sliced layout, xs = row_scale (.) x,
// xs_copy = x -- what launch_to_sliced does
struct SlicedInput {
  const float* x = nullptr;
  int64_t n_rows = 0;
  int dim = 0, W = 0;
  const float* row_scale = nullptr;
  float* xs = nullptr;
  float* xs_copy = nullptr;
};
// the dense optimizer riding in the step-values launch (spmm_sliced.hip): w == SlicedInput::x
struct SlicedOpt {
  int kind = -1;
  float* w = nullptr;
  float* g = nullptr;
  float* m = nullptr;
  float* v = nullptr;
  OptScalars s = {};
  hiprec_stats* stats = nullptr;
  const Scratch* scratch = nullptr;
};
int launch_step_values(const hiprec_sliced_csr* a, const hiprec_sliced_csr* b, uint8_t* keep, bool draw,
                       float keep_prob, uint64_t seed, uint64_t step, float* out_a, float* out_b, hipStream_t st,
                       SlicedInput in = SlicedInput{}, const SlicedOpt* fuse = nullptr);
int launch_to_sliced(const float* x, int64_t n_rows, int dim, int W, const float* row_scale, float* xs,
                     float* xs_copy, hipStream_t st);
int launch_from_sliced(const float* xs, int64_t n_rows, int dim, int W, float* y, bool add, hipStream_t st);

}  // namespace hiprec
