// Grouped exact-fp32 MFMA GEMM (csrc/ncf.hip): the problem descriptors and the host-side launcher,
// shared by the NCF tower (ncf.hip) and NGCF's per-hop Linear layers (ngcf.hip).  The kernel itself
// lives in ncf.hip only.
#pragma once
#include "common.hpp"

namespace hiprec {

// One problem of a grouped launch.  mode kNT / kNN / kTNm: C[M,N] = epilogue(op(A) op(B)) with
//   kNT : A is [M,K] (lda), B is [N,K] (ldb)          C = A B^T      (forward: H W^T)
//   kNN : A is [M,K] (lda), B is [K,N] (ldb)          C = A B        (dgrad:   dZ W)
//   kTNm: A is [K,M] (lda), B is [K,N] (ldb)          C = A^T B      (wgrad:   dZ^T H)
//   epilogue: + bias[n] (if bias) ; relu (if relu) ; * [mask[m,n] > 0] (if mask) ; dropout keep bytes (if
//   keep; make_gemm leaves it NULL, callers set keep / ldk / keep_scale on the returned problem)
// mode kColsum: C[n] += sum_m A[m, n] over the block's kColsumRows rows (bias gradients): one fp32 atomic per block
// and column (the order of the blocks' partial sums changes from run to run), or, with a workspace, in two levels
// with a fixed order (make_colsum's ws + launch_colsum_reduce).
enum GemmMode { kNT = 0, kNN = 1, kTNm = 2, kColsum = 3 };

struct GemmProblem {
  int mode, M, N, K;
  const float* A;
  int lda;
  const float* B;
  int ldb;
  float* C;
  int ldc;
  const float* bias;
  int relu;
  const float* mask;
  int ldm;
  // dropout epilogue (after bias / relu / mask): C = keep[m, n] ? C * keep_scale : 0.  NULL = none.
  const uint8_t* keep;
  int ldk;
  float keep_scale;
  int tiles_n, tiles_m, split;  // block decomposition of this problem
  int first_block;              // its first block in the grouped grid
  // kColsum only: when set, the blocks leave their partial sums in ws[slab * N + n] (no atomics) and
  // launch_colsum_reduce adds them up in slab order afterwards -- a column sum that is the same from run to run
  float* ws;
#ifdef HIPREC_TEST_SWITCHES
  int exp_bits;                 // timing experiments (libhiprec_test.so, HIPREC_GEMM_EXP): 1 split-K with plain stores, 2 no GEMM tiles, 4 no column sums
#endif
};

// A dense optimizer sweep that rides in the grouped launch as its first n_blocks blocks: elements [0, 4 * n4) of the
// flat buffers (parameters whose gradient is already complete when the grouped launch starts -- the embedding tables
// of an NCF step -- while the GEMM problems of the same grid still produce the other gradients).  n_blocks = 0: none.
struct SweepArgs {
  float *w, *g, *m, *v;
  int64_t n4;
  int kind, n_blocks, first_block;
  OptScalars s;
  const hiprec_stats* stats;
};

constexpr int kMaxGroup = 16;
struct GemmGroup {
  int n;
  GemmProblem p[kMaxGroup];
  SweepArgs sweep;
};

// split_k: let the launcher split K over several blocks that accumulate with atomics into a
// ZERO-INITIALISED C (only without an epilogue: the weight-gradient GEMMs).
GemmProblem make_gemm(int mode, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                      float* C, int ldc, const float* bias, int relu, const float* mask, int ldm,
                      bool split_k);
// C[n] += sum_m X[m, n].  ws (optional): colsum_ws_floats(M, N) floats of scratch for the two-level, fixed-order form.
GemmProblem make_colsum(const float* X, int M, int N, int ldx, float* out, float* ws = nullptr);
int64_t colsum_ws_floats(int M, int N);
// second level of every kColsum problem of g that has a workspace: out[n] += its partial sums in slab order
int launch_colsum_reduce(const GemmGroup& g, hipStream_t st);
// assigns the blocks of g.p[0..n) and launches them as ONE grid
int launch_group(GemmGroup& g, hipStream_t st);

}  // namespace hiprec
