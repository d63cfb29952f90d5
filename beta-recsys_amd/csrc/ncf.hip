// NCF family (NeuMF / GMF / MLP) training step for gfx950.
//
// Replaces the PyTorch op sequence of beta_rec/models/ncf.py:52-71 (NeuMF.forward),
// models/gmf.py:29-36, models/mlp.py:40-51, the BCELoss of models/ncf.py:92,116 and the autograd
// backward of models/ncf.py:117 (dense gradients for four embedding tables and the tower).
//
//   gather      one wave per sample: H0 = [relu](cat(Um[u], Im[i])),  MF = Ug[u] * Ig[i]
//   tower fwd   H_l = relu(H_{l-1} W_l^T + b_l)        exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)
//   head        logit = [H_L | MF] . w_out + b_out, sigmoid, BCE, d logit, dH_L, dMF, d w_out
//   tower bwd   dW_l = dZ_l^T H_{l-1} ;  db_l = colsum(dZ_l) ;  dZ_{l-1} = (dZ_l W_l) * [H_{l-1} > 0]
//   scatter     atomics of the four embedding-row gradients into the dense gradient tables
//
// The GEMMs are tiny (M = batch 4096, N,K <= 256) and launch-bound; the tile kernel below is a plain
// LDS-staged 64x64x32 MFMA tile with fused bias / ReLU / ReLU-mask epilogues — fp32 in, fp32
// accumulate, bitwise an fmaf chain in k order (no TF32-like path exists on gfx950, and the parity
// tolerance of 1e-5 rules out bf16).
#include <algorithm>

#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "gemm.hpp"

namespace hiprec {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kTM = 64, kTN = 64, kTK = 32;

// Every block-level barrier in this file orders LDS traffic only (operands staged through LDS; global
// results are consumed by LATER launches), so lds_barrier() is used throughout: __syncthreads() also
// waits for vmcnt(0), i.e. for the register-prefetched operands of the next k-step, which serialised
// every k-step on a full memory round trip.


#ifndef HIPREC_SWEEP_CAP
#define HIPREC_SWEEP_CAP 2048
#endif
constexpr int kColsumRows = 128;

// In-kernel timestamps of one split-K weight-gradient tile (-DHIPREC_NCF_DEBUG builds only, tools/build_debug_lib.sh):
// thread 0 of blocks 0 and 200 of the grouped launch; hiprec_debug_gemm_stamps reads them back.
#ifdef HIPREC_NCF_DEBUG
__device__ unsigned long long g_gemm_stamps[2][16];
#define GEMM_STAMP(k)                                                                                       \
  do {                                                                                                      \
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 200) && (k) < 16)                             \
      g_gemm_stamps[blockIdx.x == 0 ? 0 : 1][k] = __builtin_amdgcn_s_memtime();                             \
  } while (0)
#else
#define GEMM_STAMP(k) do {} while (0)
#endif

template <int MODE>
__device__ __forceinline__ void gemm_tile(float (*As)[kTK + 1], float (*Bs)[kTN + 1], const GemmProblem& q,
                                          int tile_m, int tile_n, int z) {
  const int M = q.M, N = q.N, K = q.K, lda = q.lda, ldb = q.ldb, ldc = q.ldc, ldm = q.ldm;
  const float* __restrict__ A = q.A;
  const float* __restrict__ B = q.B;
  float* __restrict__ C = q.C;
  const float* __restrict__ bias = q.bias;
  const float* __restrict__ mask = q.mask;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = tile_m * kTM, n0 = tile_n * kTN;

  GEMM_STAMP(0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // split-K: slice z owns a kTK-aligned stretch of K and adds its partial product atomically
  // (wgrad has K = batch and only a handful of output tiles; C must then be zero on entry)
  const int k_per = ((K + q.split - 1) / q.split + kTK - 1) / kTK * kTK;
  const int k_begin = z * k_per;
  const int k_end = min(K, k_begin + k_per);

  // Each thread stages kPer = 8 elements of the A tile and 8 of the B tile per k-step.  Their
  // (row, k) coordinates inside the tile and the global offsets are fixed across k-steps, so they
  // are computed once; the loads of step t+1 are issued BEFORE the MFMAs of step t and land in
  // registers while the matrix pipe works (software pipelining through registers).
  constexpr int kPer = kTM * kTK / kBlock;  // 8
  int a_m[kPer], a_k[kPer], b_n[kPer], b_k[kPer];
  int64_t a_off[kPer], b_off[kPer];
  bool a_ok[kPer], b_ok[kPer];
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const int e = tid + i * kBlock;
    if (MODE == kTNm) { a_m[i] = e % kTM; a_k[i] = e / kTM; } else { a_k[i] = e % kTK; a_m[i] = e / kTK; }
    if (MODE == kNT) { b_k[i] = e % kTK; b_n[i] = e / kTK; } else { b_n[i] = e % kTN; b_k[i] = e / kTN; }
    const int gm = m0 + a_m[i], gn = n0 + b_n[i];
    a_ok[i] = gm < M;
    b_ok[i] = gn < N;
    a_off[i] = (MODE == kTNm) ? static_cast<int64_t>(a_k[i]) * lda + gm
                              : static_cast<int64_t>(gm) * lda + a_k[i];
    b_off[i] = (MODE == kNT) ? static_cast<int64_t>(gn) * ldb + b_k[i]
                             : static_cast<int64_t>(b_k[i]) * ldb + gn;
  }
  const int64_t a_step = (MODE == kTNm) ? static_cast<int64_t>(lda) : 1;  // per unit of k
  const int64_t b_step = (MODE == kNT) ? 1 : static_cast<int64_t>(ldb);
  // Two register stages: the loads of steps t+1 and t+2 are in flight while step t's MFMAs run.  One
  // stage is not enough here: a k-step is 16 MFMAs (~0.45 us per wave) and an L2 hit under load takes
  // longer than that, so with one stage every step waited for its operands.
  float ra0[kPer], rb0[kPer], ra1[kPer], rb1[kPer];
  auto fetch = [&](float (&ra)[kPer], float (&rb)[kPer], int k0) {
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      ra[i] = (a_ok[i] && k0 + a_k[i] < k_end) ? A[a_off[i] + a_step * k0] : 0.f;
      rb[i] = (b_ok[i] && k0 + b_k[i] < k_end) ? B[b_off[i] + b_step * k0] : 0.f;
    }
  };
  auto step = [&](float (&ra)[kPer], float (&rb)[kPer], int k0) {
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      As[a_m[i]][a_k[i]] = ra[i];
      Bs[b_k[i]][b_n[i]] = rb[i];
    }
    lds_barrier();
    GEMM_STAMP(2 + 2 * ((k0 - k_begin) / kTK));     // this step's operands have arrived and are in LDS
    if (k0 + 2 * kTK < k_end) fetch(ra, rb, k0 + 2 * kTK);  // this stage's registers are free again
    // lane l feeds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]
    const int i = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < kTK; kk += 2) {
      const float a = As[wm * 32 + i][kk + kh];
      const float b = Bs[kk + kh][wn * 32 + i];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    lds_barrier();
    GEMM_STAMP(3 + 2 * ((k0 - k_begin) / kTK));     // ... and have been multiplied
  };
  if (k_begin < k_end) fetch(ra0, rb0, k_begin);
  if (k_begin + kTK < k_end) fetch(ra1, rb1, k_begin + kTK);
  GEMM_STAMP(1);
  for (int k0 = k_begin; k0 < k_end; k0 += 2 * kTK) {
    step(ra0, rb0, k0);
    if (k0 + kTK < k_end) step(ra1, rb1, k0 + kTK);
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int col = lane & 31;
  const int gn = n0 + wn * 32 + col;
  // A split-K slice (the weight-gradient GEMMs) has no epilogue by construction (make_gemm): sixteen adds at fixed row
  // offsets from one base pointer.  The general loop below -- per register the row arithmetic, four epilogue tests
  // and a test for zero around the atomic -- took a third of such a tile's time (in-kernel timestamps,
  // profiles/r04_experiments.md 60: 6.4 k of 20 k cycles).
  // (A tile that sticks out of the matrix -- NeuMF emb 32's last layer has 32 output rows -- only adds a compare per
  // register: those 32 tiles kept the general loop at first and stayed the launch's longest blocks.)
  if (q.split > 1 && !q.keep) {
    const int rb = m0 + wm * 32 + 4 * (lane >> 5);
    float* cp = C + static_cast<int64_t>(rb) * ldc + gn;
    const bool col_ok = gn < N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ro = (r & 3) + 8 * (r >> 2);
      if (col_ok && rb + ro < M) atomic_add_f32(cp + static_cast<int64_t>(ro) * ldc, acc[r]);
    }
    GEMM_STAMP(12);
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int gm = m0 + wm * 32 + row;
    if (gm < M && gn < N) {
      float v = acc[r];
      if (bias) v += bias[gn];
      if (q.relu) v = fmaxf(v, 0.f);
      if (mask) v = mask[static_cast<int64_t>(gm) * ldm + gn] > 0.f ? v : 0.f;
      if (q.keep) v = q.keep[static_cast<int64_t>(gm) * q.ldk + gn] ? v * q.keep_scale : 0.f;
      if (q.split > 1) {
#ifdef HIPREC_TEST_SWITCHES   // timing experiment only (tools/exp_ncf_wgrad_split.py): plain stores, wrong sums
        if (q.exp_bits & 1) {
          C[static_cast<int64_t>(gm) * ldc + gn] = v;
          continue;
        }
#endif
        if (v != 0.f) atomic_add_f32(C + static_cast<int64_t>(gm) * ldc + gn, v);
      } else {
        C[static_cast<int64_t>(gm) * ldc + gn] = v;
      }
    }
  }
  GEMM_STAMP(12);
}

// ---- column sums: out[n] += sum_m X[m, n]   (bias gradients) ---------------------------------------
// A block owns kColsumRows rows: 4 thread groups x 64 columns, each thread sums a quarter of the rows
// of its column, the four partials meet in LDS and ONE atomic per (block, column) leaves.  Few,
// fat blocks on purpose: every block adds into the same N addresses and same-address atomics
// serialise at ~25 ns.
__device__ __forceinline__ void colsum_tile(float* s_flat, const GemmProblem& q, int slab) {
  float (*s_part)[64] = reinterpret_cast<float (*)[64]>(s_flat);
  const float* __restrict__ X = q.A;
  const int M = q.M, N = q.N;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int m0 = slab * kColsumRows;
  const int m1 = min(M, m0 + kColsumRows);
  for (int n0 = 0; n0 < N; n0 += 64) {
    const int n = n0 + tx;
    float s = 0.f;
    if (n < N) {
#pragma unroll 8
      for (int m = m0 + ty; m < m1; m += 4) s += X[static_cast<int64_t>(m) * q.lda + n];
    }
    s_part[ty][tx] = s;
    lds_barrier();
    if (ty == 0 && n < N) {
      const float t = (s_part[0][tx] + s_part[1][tx]) + (s_part[2][tx] + s_part[3][tx]);
      if (q.ws) q.ws[static_cast<int64_t>(slab) * N + n] = t;   // second level: colsum_reduce_kernel, fixed order
      else if (t != 0.f) atomic_add_f32(q.C + n, t);
    }
    lds_barrier();
  }
}

// Grouped launch: up to kMaxGroup independent problems share one grid (the backward pass of one
// Linear layer is three of them reading the same dZ: weight gradient, bias gradient, input
// gradient; each is a ~10 us latency-bound launch on its own at batch 4096).
template <int KIND>
__device__ __forceinline__ void sweep_blocks(const SweepArgs& a) {
  float step_size, bc2_sqrt;
  step_scalars<KIND>(a.s, a.stats, &step_size, &bc2_sqrt);
  float4* w4 = reinterpret_cast<float4*>(a.w);
  float4* g4 = reinterpret_cast<float4*>(a.g);
  float4* m4 = reinterpret_cast<float4*>(a.m);
  float4* v4 = reinterpret_cast<float4*>(a.v);
  const int64_t stride = static_cast<int64_t>(a.n_blocks) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x - a.first_block) * kBlock + threadIdx.x; i < a.n4; i += stride) {
    float4 wv = w4[i], gv = g4[i];
    float4 mv = make_float4(0, 0, 0, 0), vv = make_float4(0, 0, 0, 0);
    if constexpr (KIND == HIPREC_OPT_ADAM) mv = m4[i];
    if constexpr (KIND != HIPREC_OPT_SGD) vv = v4[i];
    opt_update<KIND>(wv.x, gv.x, mv.x, vv.x, a.s, step_size, bc2_sqrt);
    opt_update<KIND>(wv.y, gv.y, mv.y, vv.y, a.s, step_size, bc2_sqrt);
    opt_update<KIND>(wv.z, gv.z, mv.z, vv.z, a.s, step_size, bc2_sqrt);
    opt_update<KIND>(wv.w, gv.w, mv.w, vv.w, a.s, step_size, bc2_sqrt);
    w4[i] = wv;
    g4[i] = gv;
    if constexpr (KIND == HIPREC_OPT_ADAM) m4[i] = mv;
    if constexpr (KIND != HIPREC_OPT_SGD) v4[i] = vv;
  }
}

__global__ __launch_bounds__(kBlock) void gemm_group_kernel(GemmGroup g) {
  __shared__ float As[kTM][kTK + 1];
  __shared__ float Bs[kTK][kTN + 1];
  const int sweep_lo = g.sweep.first_block;
  if (static_cast<int>(blockIdx.x) >= sweep_lo && static_cast<int>(blockIdx.x) < sweep_lo + g.sweep.n_blocks) {
    if (g.sweep.kind == HIPREC_OPT_ADAM) sweep_blocks<HIPREC_OPT_ADAM>(g.sweep);
    else if (g.sweep.kind == HIPREC_OPT_RMSPROP) sweep_blocks<HIPREC_OPT_RMSPROP>(g.sweep);
    else sweep_blocks<HIPREC_OPT_SGD>(g.sweep);
    return;
  }
  int qi = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (i < g.n && static_cast<int>(blockIdx.x) >= g.p[i].first_block) qi = i;
  const GemmProblem& q = g.p[qi];
  const int local = static_cast<int>(blockIdx.x) - q.first_block;
#ifdef HIPREC_TEST_SWITCHES   // timing experiment only: bit 1 = GEMM tiles return at once, bit 2 = column sums do
  if (q.exp_bits & (q.mode == kColsum ? 4 : 2)) return;
#endif
  if (q.mode == kColsum) {
    colsum_tile(&As[0][0], q, local);
    return;
  }
  const int tile_n = local % q.tiles_n;
  const int tile_m = (local / q.tiles_n) % q.tiles_m;
  const int z = local / (q.tiles_n * q.tiles_m);
  switch (q.mode) {
    case kNT:
      gemm_tile<kNT>(As, Bs, q, tile_m, tile_n, z);
      break;
    case kNN:
      gemm_tile<kNN>(As, Bs, q, tile_m, tile_n, z);
      break;
    default:
      gemm_tile<kTNm>(As, Bs, q, tile_m, tile_n, z);
      break;
  }
}

GemmProblem make_gemm(int mode, int M, int N, int K, const float* A, int lda, const float* B,
                             int ldb, float* C, int ldc, const float* bias, int relu,
                             const float* mask, int ldm, bool split_k) {
  GemmProblem q{};
  q.mode = mode; q.M = M; q.N = N; q.K = K; q.A = A; q.lda = lda; q.B = B; q.ldb = ldb; q.C = C; q.ldc = ldc;
  q.bias = bias; q.relu = relu; q.mask = mask; q.ldm = ldm;
  q.tiles_n = (N + kTN - 1) / kTN;
  q.tiles_m = (M + kTM - 1) / kTM;
  q.split = 1;
  // split K when the output has too few tiles to fill 256 CUs (only legal without an epilogue,
  // into a zero-initialised C: the weight-gradient GEMMs)
  if (split_k && !bias && !relu && !mask) {
    const int tiles = q.tiles_n * q.tiles_m;
    // ~384 blocks in all (round 6, same-box A/B: 256 / 384 / 512 / 1024 / 1536 blocks -> NeuMF emb 64 100.0 / 98.3 / 102.4 /
    // 107.5 / 107.5 us per step; emb 32 and NGCF are bound by max_z below and do not move)
    int z = (384 + tiles - 1) / tiles;
    const int max_z = (K + 4 * kTK - 1) / (4 * kTK);  // at least 4 k-tiles per slice (2: NCF 58.8 -> 65.1 us; 8: 60.2, emb 64 127.6 -> 120.0; r06: 3 / 6: 50.7 against 48.8)
    if (z > max_z) z = max_z;
    if (z > 1) q.split = z;
  }
#ifdef HIPREC_TEST_SWITCHES
  static const int plain = getenv("HIPREC_GEMM_EXP") ? atoi(getenv("HIPREC_GEMM_EXP")) : 0;  // 1 plain stores, 2 no GEMM tiles
  q.exp_bits = plain;
#endif
  return q;
}

GemmProblem make_colsum(const float* X, int M, int N, int ldx, float* out, float* ws) {
  GemmProblem q{};
  q.mode = kColsum; q.M = M; q.N = N; q.A = X; q.lda = ldx; q.C = out;
  q.tiles_n = 1;
  q.tiles_m = (M + kColsumRows - 1) / kColsumRows;
  q.split = 1;
  q.ws = q.tiles_m > 1 ? ws : nullptr;   // a single block's one add into a zeroed gradient is already deterministic
#ifdef HIPREC_TEST_SWITCHES
  static const int exp = getenv("HIPREC_GEMM_EXP") ? atoi(getenv("HIPREC_GEMM_EXP")) : 0;   // 4: no column sums
  q.exp_bits = exp;
#endif
  return q;
}

int64_t colsum_ws_floats(int M, int N) { return static_cast<int64_t>((M + kColsumRows - 1) / kColsumRows) * N; }

struct ColsumReduce {
  int n;
  const float* ws[kMaxGroup];
  float* out[kMaxGroup];
  int N[kMaxGroup], slabs[kMaxGroup];
};

// out[n] += the slabs' partial sums of column n, in a FIXED association: 16 thread groups each add every 16th slab
// (their loads are independent: one memory round trip instead of `slabs` dependent ones -- the first version, one
// thread per column walking all slabs, took 19.9 us for NGCF's 77 slabs), then the 16 partials are added in order.
constexpr int kReduceThreads = 1024, kReduceParts = kReduceThreads / 64;
__global__ __launch_bounds__(kReduceThreads) void colsum_reduce_kernel(ColsumReduce r) {
  __shared__ float s_p[kReduceParts][64];
  const int i = blockIdx.x;
  const float* __restrict__ ws = r.ws[i];
  const int N = r.N[i], slabs = r.slabs[i];
  const int tx = threadIdx.x & 63, part = threadIdx.x >> 6;
  for (int n0 = 0; n0 < N; n0 += 64) {
    const int n = n0 + tx;
    float t = 0.f;
    if (n < N) {
#pragma unroll 8
      for (int s = part; s < slabs; s += kReduceParts) t += ws[static_cast<int64_t>(s) * N + n];
    }
    s_p[part][tx] = t;
    __syncthreads();
    if (part == 0 && n < N) {
      float tot = 0.f;
#pragma unroll
      for (int q = 0; q < kReduceParts; ++q) tot += s_p[q][tx];
      r.out[i][n] += tot;
    }
    __syncthreads();
  }
}

int launch_colsum_reduce(const GemmGroup& g, hipStream_t st) {
  ColsumReduce r{};
  for (int i = 0; i < g.n; ++i) {
    const GemmProblem& q = g.p[i];
    if (q.mode != kColsum || !q.ws) continue;
    r.ws[r.n] = q.ws;
    r.out[r.n] = q.C;
    r.N[r.n] = q.N;
    r.slabs[r.n] = q.tiles_m;
    ++r.n;
  }
  if (r.n == 0) return 0;
  colsum_reduce_kernel<<<r.n, kReduceThreads, 0, st>>>(r);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

int launch_group(GemmGroup& g, hipStream_t st) {
  // the sweep's blocks come LAST: the GEMM tiles (chains of dependent cold misses on what the previous launch wrote)
  // are dispatched first and the bandwidth-bound sweep fills in behind them.  Measured on the NCF step (r03 exp. 3):
  // sweep first 64.1 us per step, last 58.1-58.9, no sweep in this launch (a 37 MB sweep of its own) 59.4
  int blocks = 0;
  for (int i = 0; i < g.n; ++i) {
    g.p[i].first_block = blocks;
    blocks += g.p[i].tiles_n * g.p[i].tiles_m * g.p[i].split;
  }
  g.sweep.first_block = blocks;
  blocks += g.sweep.n_blocks;
  if (blocks == 0) return 0;
  gemm_group_kernel<<<blocks, kBlock, 0, st>>>(g);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

static int launch_gemm(int mode, int M, int N, int K, const float* A, int lda, const float* B,
                       int ldb, float* C, int ldc, const float* bias, int relu, const float* mask,
                       int ldm, hipStream_t st, bool split_k = false) {
  if (M <= 0 || N <= 0) return 0;
  if (mode != kNT && mode != kNN && mode != kTNm) {
    set_error("bad gemm mode %d", mode);
    return HIPREC_E_BADARG;
  }
  GemmGroup g{};
  g.n = 1;
  g.p[0] = make_gemm(mode, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, mask, ldm, split_k);
  return launch_group(g, st);
}

// ---- gather: H0[b] = [relu](cat(Um[u], Im[i])),  MF[b] = Ug[u] * Ig[i] ------------------------------
__global__ __launch_bounds__(kBlock) void ncf_gather_kernel(hiprec_ncf_plan p,
                                                            const int64_t* __restrict__ users,
                                                            const int64_t* __restrict__ items,
                                                            int64_t batch, hiprec_stats* stats) {
  const int lane = lane_id();
  const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int Dm = p.dim_mlp, E = p.dim_mf;
  for (int64_t b = wave0; b < batch; b += n_waves) {
    const int64_t u = users[b], i = items[b];
    const bool u_ok = static_cast<uint64_t>(u) < static_cast<uint64_t>(p.n_users);
    const bool i_ok = static_cast<uint64_t>(i) < static_cast<uint64_t>(p.n_items);
    const bool ok = u_ok && i_ok;
    if (!ok && lane == 0)
      atomicOr(&stats->status,
               (u_ok ? 0u : HIPREC_STATUS_USER_OOB) | (i_ok ? 0u : HIPREC_STATUS_ITEM_OOB));
    if (Dm > 0) {
      float* h0 = p.act[0] + b * (2 * Dm);
      for (int c = lane; c < Dm; c += kWave) {
        float a = ok ? p.user_mlp[u * Dm + c] : 0.f;
        float d = ok ? p.item_mlp[i * Dm + c] : 0.f;
        if (p.relu_input) { a = fmaxf(a, 0.f); d = fmaxf(d, 0.f); }
        if (p.keep[0]) {  // the Dropout in front of the first Linear (ncf.py:42-45, mlp.py:30-33)
          const uint8_t* k0 = p.keep[0] + b * (2 * Dm);
          a = k0[c] ? a * p.keep_scale : 0.f;
          d = k0[Dm + c] ? d * p.keep_scale : 0.f;
        }
        h0[c] = a;
        h0[Dm + c] = d;
      }
    }
    if (E > 0) {
      float* mf = p.mf + b * E;
      for (int c = lane; c < E; c += kWave)
        mf[c] = ok ? p.user_mf[u * E + c] * p.item_mf[i * E + c] : 0.f;
    }
  }
}

// ---- head: logit, sigmoid, BCE, d logit, dH_L (masked), dMF, d w_out, d b_out ------------------------
// one wave per sample; d w_out is accumulated per lane over the wave's samples, then one atomic per
// (block, column).  TRAIN = false: scores only.
// 16-wave blocks: at batch 4096 the training launch is 64 blocks whose waves own 4 samples each (one
// trip of the loop below), and every block ends with ONE atomic per column of affine_output.weight --
// 64 same-address atomics (~25 ns each) instead of one per 4-wave block.
constexpr int kHeadWaves = 16;
constexpr int kHeadBlock = kHeadWaves * kWave;

template <bool TRAIN>
__global__ __launch_bounds__(kHeadBlock) void ncf_head_kernel(hiprec_ncf_plan p,
                                                          const float* __restrict__ ratings,
                                                          int64_t batch, float inv_batch,
                                                          hiprec_stats* stats, Scratch* scratch) {
  __shared__ float s_gw[kHeadWaves][256 + 1];
  const int lane = lane_id();
  const int wv = wave_in_block();
  const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * kHeadWaves + wv;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * kHeadWaves;
  const int L = p.n_layers;
  const int nH = (p.dim_mlp > 0) ? p.layer_out[L - 1] : 0;  // width of the tower output
  const int E = p.dim_mf;
  const int nV = nH + E;                                    // affine_output in_features (<= 256+...)
  const float* hL = nH > 0 ? p.act[L] : nullptr;
  const float bo = *p.out_b;
  float loss_acc = 0.f, gb_acc = 0.f;
  float gw[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // columns lane, lane+64, ... (nV <= 320)

  // the stepper's loads travel with the first trip's input loads; its stores go out after them
  const bool stepper = TRAIN && blockIdx.x == 0 && threadIdx.x == 0;
  StepState step_state{};
  if (stepper) step_state = step_load(stats);

  // kUnroll samples per trip: their input loads are requested together, so a wave pays one memory
  // round trip per trip instead of one per sample (a wave owns batch / n_waves = 8 samples at B 4096).
  constexpr int kUnroll = 4;
  float wout[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) wout[k] = lane + kWave * k < nV ? p.out_w[lane + kWave * k] : 0.f;
  for (int64_t b0 = wave0; b0 < batch; b0 += n_waves * kUnroll) {
    float vec[kUnroll][5];
    float rj[kUnroll];
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      const int64_t b = b0 + j * n_waves;
      const bool live = b < batch;
      rj[j] = (TRAIN && live) ? ratings[b] : 0.f;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int c = lane + kWave * k;
        float v = 0.f;
        if (live && c < nH) v = hL[b * nH + c];
        else if (live && c < nV) v = p.mf[b * E + (c - nH)];
        vec[j][k] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      const int64_t b = b0 + j * n_waves;
      if (b >= batch) break;
      float part = 0.f;
#pragma unroll
      for (int k = 0; k < 5; ++k) part += vec[j][k] * wout[k];
      const float logit = wave_sum(part) + bo;
      const float y = sigmoid_f32(logit);
      if (lane == 0) p.scores[b] = y;
      if (!TRAIN) continue;
      const float r = rj[j];
      const float ly = fmaxf(logf(y), -100.f);
      const float l1y = fmaxf(log1pf(-y), -100.f);
      loss_acc += -(r * ly + (1.f - r) * l1y);
      const float gy = (y - r) / fmaxf((1.f - y) * y, 1e-12f) * inv_batch;
      const float dl = gy * ((1.f - y) * y);
      gb_acc += dl;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int c = lane + kWave * k;
        if (c < nV) gw[k] += dl * vec[j][k];
        if (c < nH) {
          // d loss / d z_L = d h_L * [h_L > 0]  (ReLU of the last Linear, applied twice in NeuMF)
          p.dact[L][b * nH + c] = vec[j][k] > 0.f ? dl * wout[k] : 0.f;
        } else if (c < nV) {
          p.dmf[b * E + (c - nH)] = dl * wout[k];
        }
      }
    }
  }
  if (!TRAIN) return;
  if (stepper) step_store_advanced(stats, step_state);
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int c = lane + kWave * k;
    if (c < 256) s_gw[wv][c] = gw[k];
  }
  // loss partial (reg slot unused = 0); d b_out travels in the scalar-gradient slot
  publish_partials<kHeadWaves>(loss_acc, 0.f, gb_acc, inv_batch, scratch);
  for (int c = threadIdx.x; c < nV && c < 256; c += kHeadBlock) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kHeadWaves; ++w) s += s_gw[w][c];
    if (s != 0.f) atomic_add_f32(p.g_out_w + c, s);
  }
  if (nV > 256) {  // rare wide heads: straight per-wave atomics for the tail columns
    const int c = lane + 256;
    if (c < nV && gw[4] != 0.f) atomic_add_f32(p.g_out_w + c, gw[4]);
  }
}

// ---- scatter of the embedding-row gradients ---------------------------------------------------------
__global__ __launch_bounds__(kBlock) void ncf_scatter_kernel(hiprec_ncf_plan p,
                                                             const int64_t* __restrict__ users,
                                                             const int64_t* __restrict__ items,
                                                             int64_t batch) {
  const int lane = lane_id();
  const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int Dm = p.dim_mlp, E = p.dim_mf;
  for (int64_t b = wave0; b < batch; b += n_waves) {
    const int64_t u = users[b], i = items[b];
    if (static_cast<uint64_t>(u) >= static_cast<uint64_t>(p.n_users) ||
        static_cast<uint64_t>(i) >= static_cast<uint64_t>(p.n_items))
      continue;  // flagged by the gather kernel
    if (Dm > 0) {
      const float* dx = p.dact[0] + b * (2 * Dm);  // already masked by [x > 0] when relu_input
      for (int c = lane; c < Dm; c += kWave) {
        atomic_add_f32(p.g_user_mlp + u * Dm + c, dx[c]);
        atomic_add_f32(p.g_item_mlp + i * Dm + c, dx[Dm + c]);
      }
    }
    if (E > 0) {
      const float* dmf = p.dmf + b * E;
      for (int c = lane; c < E; c += kWave) {
        const float d = dmf[c];
        atomic_add_f32(p.g_user_mf + u * E + c, d * p.item_mf[i * E + c]);
        atomic_add_f32(p.g_item_mf + i * E + c, d * p.user_mf[u * E + c]);
      }
    }
  }
}

// ======================= fused tower: the per-sample chain of a training step in one launch ===========
// At batch 4096 every stand-alone launch of this path (gather, a 4096 x 128 x 256 GEMM, the head...)
// costs 8-14 us although its arithmetic is worth 1-2 us: each one starts by missing on what the
// previous launch wrote (rocprofv3, round 1: 11 launches, 106 us per step).  A block of 8 waves that owns 16
// samples carries them through the whole tower, the head, the tower's input-gradient chain and the embedding
// scatter without leaving the CU: the activations stay in LDS between layers and for the backward's ReLU masks
// (they are also written to HBM once, with the dZ_l, for the weight-gradient GEMMs), only the weights stream
// through a 32-k LDS tile.  Layers are fp32 MFMA 16x16x4 (one 16 x 16 output tile per wave and pass).
// Shapes outside the limits below take the unfused path.
constexpr int kFR = 16;                    // samples per block: 16-row MFMA tiles (v_mfma_f32_16x16x4_f32), 256 blocks at
                                           // B 4096 = one per CU (32-row tiles: 128 blocks, half the chip idle)
constexpr int kFWaves = 8;                 // one wave per 16 output columns of a 128-column pass: two waves per SIMD, so
                                           // one's MFMA chain runs under the other's LDS / memory waits (with 4 waves
                                           // of 32 columns a 32-k chunk took ~1800 cycles for 512 cycles of MFMA)
constexpr int kFThreads = kFWaves * kWave; // 512
constexpr int kFK = 32;                    // k-chunk of the weight tiles (64: same step time)
constexpr int kFPre = 3;                   // weight chunks a wave keeps in flight (8 VGPRs each; the kernel has 95 of
                                           // 168).  r06, same box, us per step at emb 32 / 64: 2 chunks 47.8 / 95.6,
                                           // 3: 47.0 / 94.8, 5: 48.2 / 96.8, 8: 49.9 / 101.4 -- more requests in flight
                                           // queue up in front of the L2s.  Every block walks k in the same order on
                                           // purpose: with each block starting at chunk blockIdx % n_chunks (no two
                                           // neighbours asking for the same lines at the same time) 48.8 / 96.6
constexpr int kFMaxIn = 512;               // widest tower input (2 * dim_mlp): emb_dim 64's 512-256-128-64 tower fits
constexpr int kFMaxN = 128;                // output columns of one pass (8 waves x 16)
constexpr int kFMaxW = 256;                // widest layer output (two passes)
constexpr int kFMaxE = 64;                 // GMF width kept in LDS
constexpr int kFLdE = kFMaxE + 1;
constexpr int kFLdN = kFMaxN + 1;          // weight tile rows of the forward ([k][n], written transposed)
constexpr int kFLdB = kFMaxN + 8;          // ... of the input-gradient chain: 16-byte aligned rows
constexpr int kFNarrowIn = 256;            // limits of the stand-alone chain launch (ncf_fused_dgrad_kernel)
constexpr int kFLdIn = kFNarrowIn + 1;
constexpr size_t kFusedMaxLds = 160 * 1024 - 2048;  // (the kernels also hold a few hundred bytes of static LDS)

// LDS of the fused launch, in floats: every layer's input stays ([16][width + 1] each: the tower input, then each
// layer's output) -- the next layer reads it, the backward masks with it, the last phase stores it --, with BWD one
// dZ_l buffer per layer laid out like the activations plus the layer-0 input gradient [16][2 dim_mlp + 1], then the GMF
// tile, the waves' partial sums of d affine_output.weight and the logits / d logits of the 16 samples.
static size_t fused_lds_floats(const hiprec_ncf_plan* p, bool bwd) {
  const size_t a0 = static_cast<size_t>(kFR) * (2 * p->dim_mlp + 1);
  size_t layers = 0;
  for (int l = 0; l < p->n_layers; ++l) layers += static_cast<size_t>(kFR) * (p->layer_out[l] + 1);
  size_t n = a0 + layers + (bwd ? layers + a0 : 0);
  n = (n + 3) / 4 * 4;
  return n + kFR * kFLdE + kFWaves * 3 * kWave + 2 * kFR + (bwd ? 2 * kFR * kFLdE : 0);
}

static bool fusable(const hiprec_ncf_plan* p, bool bwd = false) {
  if (p->dim_mlp <= 0 || p->n_layers < 1) return false;
  if (2 * p->dim_mlp > kFMaxIn || (2 * p->dim_mlp) % kFK) return false;
  if (p->dim_mf > kFMaxE) return false;
  for (int l = 0; l < p->n_layers; ++l) {
    if (p->layer_out[l] > kFMaxW || p->layer_out[l] % 32) return false;
    if (p->layer_in[l] % kFK) return false;
  }
  if (p->layer_out[p->n_layers - 1] + p->dim_mf > 3 * kWave) return false;  // the head keeps 3 values per lane
  return fused_lds_floats(p, bwd) * sizeof(float) <= kFusedMaxLds;
}

// the stand-alone chain launch keeps round 2's first limits (tower input <= 256, layers <= 128 wide)
static bool fusable_narrow(const hiprec_ncf_plan* p) {
  if (!fusable(p) || 2 * p->dim_mlp > kFNarrowIn) return false;
  for (int l = 0; l < p->n_layers; ++l)
    if (p->layer_out[l] > kFMaxN) return false;
  return true;
}

// The tile GEMMs of the fused kernels: out[16][N] (N <= 128: one 16 x 16 tile per wave) = in[16][K] (LDS) x a weight
// matrix that comes STRAIGHT FROM L2 INTO REGISTERS.  A sum over k does not care about the order of its terms, so the
// k index a lane feeds to MFMA j of a 32-k chunk is permuted: lane group kq = lane >> 4 takes the 8 CONSECUTIVE
// k = k0 + 8 kq + j (the 16x16x4 instruction only asks that A and B agree on it).  For nn.Linear's [N][K] weight the
// B operand of a chunk is then two 16-byte loads of this lane's own row -- no transposing LDS tile, no barrier inside a
// layer (rounds 1-2 staged every 32-k chunk through a double-buffered LDS tile: 8 scalar LDS stores per thread and a
// workgroup barrier per chunk, 1 375 cycles per chunk for 512 cycles of MFMA, r02 experiments 29).  Three chunks of B
// are in flight ahead of the MFMAs; begin() issues the first ones and depends on nothing the block computes, so the
// caller places it before whatever produces `in` (the gather, the previous layer's epilogue).
struct FusedGemm {  // forward: W = nn.Linear.weight [N][K] row-major, out = in W^T
  // addresses = a wave-uniform pointer (scalar arithmetic) + ONE 32-bit lane offset: per-lane 64-bit pointer
  // arithmetic for every load was a third of a layer's instructions
  const float* base;  // W (uniform)
  uint32_t lane_off;  // this lane's weight row, at its lane group's first k (BYTES: uniform base + 32-bit offset is
                      // the global_load saddr form)
  bool ok;
  int n_chunks;
  float4 w[kFPre][2];

  __device__ __forceinline__ void fetch(float4 (&d)[2], int t) {
    const float* src = base + t * kFK;  // uniform
    if (!ok) return;  // a scalar branch (ok is wave-uniform): a per-lane select would cost the loads their saddr form
    const char* at = reinterpret_cast<const char*>(src) + lane_off;
    d[0] = *reinterpret_cast<const float4*>(at);
    d[1] = *reinterpret_cast<const float4*>(at + 16);
  }
  __device__ __forceinline__ void begin(const float* __restrict__ W, int K, int N) {
    const int lane = threadIdx.x & 63, wn = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    ok = wn * 16 < N;  // N is a multiple of 16: the wave's 16 columns exist or they do not
    base = W;
    lane_off = static_cast<uint32_t>((ok ? wn * 16 + (lane & 15) : 0) * K + 8 * (lane >> 4)) * 4u;
    n_chunks = K / kFK;
    fetch(w[0], 0);
#pragma unroll
    for (int k = 1; k < kFPre; ++k)
      if (n_chunks > k) fetch(w[k], k);
  }
  __device__ __forceinline__ void run(f32x4& acc, const float* in, int ld_in, float*) {
    const int lane = threadIdx.x & 63;
    const float* a_row = in + (lane & 15) * ld_in + 8 * (lane >> 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = 0.f;
    if (!ok) return;  // wave-uniform (N is a multiple of 16): a wave beyond the pass's columns has nothing to do
    auto chunk = [&](float4 (&b4)[2], int t) {
      float a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = a_row[t * kFK + j];
      const float b[8] = {b4[0].x, b4[0].y, b4[0].z, b4[0].w, b4[1].x, b4[1].y, b4[1].z, b4[1].w};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
      if (t + kFPre < n_chunks) fetch(b4, t + kFPre);
    };
    for (int t = 0; t < n_chunks; t += kFPre) {
      chunk(w[0], t);
#pragma unroll
      for (int k = 1; k < kFPre; ++k)
        if (t + k < n_chunks) chunk(w[k], t + k);
    }
  }
};

// (the head's per-wave partial sums of d affine_output.weight still meet in this LDS region)
constexpr size_t kFusedBwdLdsBytes = sizeof(float) * (kFR * kFLdIn + kFR * kFLdN + 2 * kFK * kFLdB);

struct FusedGemmNN {  // input gradients: W [K][ldw] row-major as it lies in memory, out = in W[:, n_off : n_off + N]
  // B operand k = a row of W: 8 single-dword loads per chunk, each lane group its own 8 rows.  As BUFFER loads: the
  // resource (W + n_off) and the row's byte offset are scalars, the lane's offset one VGPR for the whole layer -- as
  // global loads the compiler kept a 64-bit per-lane pointer per row (~6 VALU instructions per load, a third of the
  // chain's instructions).
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t lane_off;  // this lane's column, at its lane group's first k (bytes)
  bool ok;
  int n_chunks, ldw4;  // row pitch in bytes

  float w[kFPre][8];

  __device__ __forceinline__ void fetch(float (&d)[8], int t) {
    if (!ok) return;  // scalar branch
#pragma unroll
    for (int j = 0; j < 8; ++j)
      d[j] = __builtin_bit_cast(float, static_cast<uint32_t>(__builtin_amdgcn_raw_buffer_load_b32(
                                           rsrc, static_cast<int>(lane_off), (t * kFK + j) * ldw4, 0)));
  }
  // columns n_off .. n_off + N (N <= 128, a multiple of 16) of W[K][ldw]
  __device__ __forceinline__ void begin(const float* __restrict__ W, int ldw_, int K, int N, int n_off) {
    const int lane = threadIdx.x & 63, wn = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    ok = wn * 16 < N;
    ldw4 = ldw_ * 4;
    // raw buffer over the rest of the address space from W + n_off (offsets stay below K * ldw * 4 < 2^31)
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W + n_off), 0, 0x7FFFFFFF, 0x00027000);
    lane_off = static_cast<uint32_t>(8 * (lane >> 4) * ldw_ + (ok ? wn * 16 + (lane & 15) : 0)) * 4u;
    n_chunks = K / kFK;
    fetch(w[0], 0);
#pragma unroll
    for (int k = 1; k < kFPre; ++k)
      if (n_chunks > k) fetch(w[k], k);
  }
  __device__ __forceinline__ void run(f32x4& acc, const float* in, int ld_in, float*) {
    const int lane = threadIdx.x & 63;
    const float* a_row = in + (lane & 15) * ld_in + 8 * (lane >> 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = 0.f;
    if (!ok) return;
    auto chunk = [&](float (&b)[8], int t) {
      float a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = a_row[t * kFK + j];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
      if (t + kFPre < n_chunks) fetch(b, t + kFPre);
    };
    for (int t = 0; t < n_chunks; t += kFPre) {
      chunk(w[0], t);
#pragma unroll
      for (int k = 1; k < kFPre; ++k)
        if (t + k < n_chunks) chunk(w[k], t + k);
    }
  }
};

// In-kernel timestamps of the fused launch (builds with -DHIPREC_NCF_DEBUG only, tools/build_debug_lib.sh): thread 0 of
// two blocks notes the cycle counter at the phase boundaries; hiprec_debug_ncf_stamps reads them back.
#ifdef HIPREC_NCF_DEBUG
__device__ unsigned long long g_ncf_stamps[2][24];
#define NCF_STAMP(k)                                                                                        \
  do {                                                                                                      \
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 131) && (k) < 24)                             \
      g_ncf_stamps[blockIdx.x == 0 ? 0 : 1][k] = __builtin_amdgcn_s_memtime();                              \
  } while (0)
#else
#define NCF_STAMP(k) do {} while (0)
#endif

// The NINTH wave of a chained launch (round 6).  The block's stores used to leave in one sweep at the end -- 16 x 672
// floats of activations and dZ_l, then 16 x 320 embedding-gradient atomics -- and all 256 blocks reached that sweep
// together: 11 MB of stores at ~4 TB/s and 1.3 M float atomics at the L2s' ~300 G/s, 16 k of the launch's 56 k cycles
// (in-kernel timestamps), while the memory system idled through the 40 k before it.  Spreading them over the compute
// waves' phases only moved the cost (gfx950 counts loads and stores in one vmcnt: a wave that has stores in flight waits
// for them at its next weight chunk).  This wave loads no weights and computes nothing: it keeps the compute waves'
// barrier sequence and, behind each barrier that completes a tile in LDS (which nobody overwrites), copies that tile
// out -- act_l during layer l + 1, dZ_l during the chain's next layer.  Plain stores only: atomics issued from here hold
// up the compute waves' weight loads behind them in the CU's memory pipeline.
// Its barriers MUST mirror ncf_fused_forward_kernel<TRAIN, DROP, BWD = true> one for one.
__device__ __forceinline__ void fused_store_wave(const hiprec_ncf_plan& p, const float* lds_raw, int64_t m0,
                                                 int64_t batch, int dz_shift) {
  const int lane = threadIdx.x & 63;
  const int K0 = 2 * p.dim_mlp, L = p.n_layers, ld0 = K0 + 1;
  const int n_rows = static_cast<int>(batch - m0 < kFR ? batch - m0 : kFR);
  // dst[m0 + r][0 .. N) = src[r][0 .. N), r < n_rows.  Few instructions per store, the wave shares its SIMD with two
  // compute waves: eight rows' values of a column read from LDS together, then stored through a buffer resource over
  // the block's rows of dst (row offset = a scalar, column offset = one VGPR; rows beyond n_rows fall outside the
  // resource and are dropped by the hardware, no masks).  The first form -- per store a 64-bit address and a row
  // test -- took 170 cycles per store and held the barriers back (layer 1: 8.0 k -> 11.0 k cycles).
  auto store_tile = [&](float* __restrict__ dst, const float* src, int N, int ld) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst + m0 * N, 0, n_rows * N * 4, 0x00027000);
    for (int c0 = 0; c0 < N; c0 += kWave) {
      const int c = c0 + lane;
      if (c < N) {
        const float* sp = src + c;
#pragma unroll
        for (int r0 = 0; r0 < kFR; r0 += 8) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = sp[(r0 + j) * ld];
#pragma unroll
          for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v[j]), rs, c * 4, (r0 + j) * N * 4, 0);
        }
      }
    }
  };
  lds_barrier();  // the index pairs
  lds_barrier();  // the gathered rows
  store_tile(p.act[0], lds_raw, K0, ld0);
  int off = kFR * ld0;
  for (int l = 0; l < L; ++l) {
    const int N = p.layer_out[l];
    for (int n_off = 0; n_off < N; n_off += kFMaxN) lds_barrier();
    if (l + 1 < L) store_tile(p.act[l + 1], lds_raw + off, N, N + 1);   // (act_L feeds the head only)
    if (l + 1 < L) off += kFR * (N + 1);
  }
  // off = act_L.  The head: logits | d logits | dZ_L and dMF in LDS
  lds_barrier();
  lds_barrier();
  lds_barrier();
  const int nH = p.layer_out[L - 1];
  store_tile(p.dact[L], lds_raw + off + dz_shift, nH, nH + 1);
  for (int l = L - 1; l >= 0; --l) {
    const int nin = p.layer_in[l];
    for (int n_off = 0; n_off < nin; n_off += kFMaxN) {
      lds_barrier();
    }
    if (l > 0) {
      off -= kFR * (nin + 1);   // act_l; dZ_l lies dz_shift behind it
      store_tile(p.dact[l], lds_raw + off + dz_shift, nin, nin + 1);
    }
  }
  lds_barrier();  // publish_partials'
  lds_barrier();  // the last phase's
}

// ---- forward: gather -> tower -> affine_output -> sigmoid; writes act[0..L], scores -------------------------------
// TRAIN: the head's backward half rides along (BCELoss term, d loss / d logit, dZ_L, dMF, d w_out,
// the loss / d b_out partials) -- everything it needs is already in LDS, and a separate head launch
// cost 12 us.  BWD: so does the tower's input-gradient chain and the embedding scatter (see the header).
// A layer wider than 128 columns takes passes of 128 (8 waves x 16); LDS layout: fused_lds_floats.
//
// Round 6: NOTHING is stored to global memory before the block's last phase.  gfx950 counts loads, stores and
// atomics in ONE counter (vmcnt) and the compiler, once both kinds are in flight across a branch, waits for
// vmcnt(0): every weight chunk a layer waited for also waited for the acknowledgement of whatever the previous
// epilogue had stored -- activations, dZ_l, the embedding-gradient atomics of a layer-0 pass, and the 64
// same-address atomics per block of d affine_output.weight (256 blocks serialise on them: the first chain pass waited
// 5-8 k cycles for its 8 weight dwords, in-kernel timestamps r06).  Everything the later launches need stays in LDS
// (activations, one dZ_l buffer per layer, the layer-0 input gradient) and leaves in one sweep at the end (spreading
// the stores over the head and the chain's last layer moved their cost, it did not hide it: same-box A/B 50.5
// against 49.0 us); d w_out leaves as one plain row of per-block partial sums (`gw_ws`) that the grouped launch's
// column-sum path adds up.
// The head's scalar arithmetic (sigmoid, two logs, the BCE quotient: ~300 VALU instructions) ran once per SAMPLE on
// all 64 lanes of a wave, two samples per wave one after the other; now lane r of wave 0 does sample r.
template <bool TRAIN, bool DROP, bool BWD = false>
__global__ __launch_bounds__(kFThreads + (BWD ? kWave : 0)) void ncf_fused_forward_kernel(
    hiprec_ncf_plan p, const int64_t* __restrict__ users, const int64_t* __restrict__ items,
    const float* __restrict__ ratings, int64_t batch, float inv_batch, hiprec_stats* stats,
    Scratch* scratch, float* __restrict__ gw_ws) {
  static_assert(!BWD || TRAIN, "the backward rides on the training forward");
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kFR;
  const int Dm = p.dim_mlp, E = p.dim_mf, K0 = 2 * Dm, n_layers = p.n_layers;
  // LDS: act_0 (the gathered rows) | act_1 | ... | act_L | [BWD: dZ_1 .. dZ_L laid out like act_1 .. act_L | the
  // layer-0 input gradient [kFR][K0 + 1]] | GMF tile | d w_out partials of the waves | logits, d logits
  const int ld0 = K0 + 1;
  int act_floats = kFR * ld0;
  for (int l = 0; l < n_layers; ++l) act_floats += kFR * (p.layer_out[l] + 1);
  const int dz_shift = act_floats - kFR * ld0;          // dZ_l sits dz_shift floats behind act_l (l >= 1)
  float* d0 = lds_raw + act_floats + dz_shift;          // BWD only
  float* s_mf = lds_raw + (act_floats + (BWD ? dz_shift + kFR * ld0 : 0) + 3) / 4 * 4;
  float* s_gw = s_mf + kFR * kFLdE;                     // [kFWaves][3 * kWave]
  float* s_logit = s_gw + kFWaves * 3 * kWave;          // [kFR]
  float* s_dl = s_logit + kFR;                          // [kFR]
  float* s_um = s_dl + kFR;                             // BWD: the GMF factors of the samples, [kFR][kFLdE] each, for the
  float* s_im = s_um + kFR * kFLdE;                     // store wave's GMF-row gradients

  NCF_STAMP(0);
  // The index pairs are fetched by the LAST wave before it requests anything else: it then waits for exactly these two
  // loads (vmcnt counts in order: behind the weight prefetch the fetching wave waited for the tower's cold lines as
  // well, 5.5 k cycles instead of one miss), the other seven go straight to their prefetch.
  // gather, element-parallel: 16 threads fetch the index pairs, then every thread owns one column of 8 (tower input
  // up to 256 wide: the two halves of the block take 8 rows each) or 16 rows; all its loads are requested before
  // anything is stored (one round trip for the whole tile instead of one per row)
  __shared__ long long s_u[kFR], s_i[kFR];
  if constexpr (BWD) {
    if (wave == kFWaves) {  // the store wave (fused_store_wave): this kernel's barriers, none of its arithmetic
      fused_store_wave(p, lds_raw, m0, batch, dz_shift);
      return;
    }
  }
  if (wave == kFWaves - 1 && lane < kFR) {
    const int64_t b = m0 + lane;
    long long u = -1, it = -1;
    if (b < batch) {
      u = users[b];
      it = items[b];
      const bool u_ok = static_cast<uint64_t>(u) < static_cast<uint64_t>(p.n_users);
      const bool i_ok = static_cast<uint64_t>(it) < static_cast<uint64_t>(p.n_items);
      if (!(u_ok && i_ok)) {
        atomicOr(&stats->status,
                 (u_ok ? 0u : HIPREC_STATUS_USER_OOB) | (i_ok ? 0u : HIPREC_STATUS_ITEM_OOB));
        u = it = -1;
      }
    }
    s_u[lane] = u;
    s_i[lane] = it;
  }
  // Loads that depend on nothing the block computes go next, off its serial chain: layer 0's first weight chunks,
  // this lane's bias element of the first pass, the head's weights and targets.
  FusedGemm gemm;
  gemm.begin(p.fc_w[0], p.layer_in[0], min(p.layer_out[0], kFMaxN));
  auto load_bias = [&](int l, int n_off) {  // this lane's bias element of pass (l, n_off)
    const int col = n_off + wn * 16 + (lane & 15);
    return l < n_layers && col < p.layer_out[l] ? p.fc_b[l][col] : 0.f;
  };
  float bias_now = load_bias(0, 0), bias_next = 0.f;
  const int nH = p.layer_out[n_layers - 1], nV = nH + E;
  const float bo = load_scalar_param(p.out_b);
  float wout[3];  // nV <= 192
#pragma unroll
  for (int k = 0; k < 3; ++k) wout[k] = lane + kWave * k < nV ? p.out_w[lane + kWave * k] : 0.f;
  float rt = 0.f;  // lane r of wave 0: the target of sample r
  if (TRAIN && wave == 0 && lane < kFR) {
    const int64_t b = m0 + lane;
    rt = ratings[b < batch ? b : batch - 1];
  }
  const bool stepper = TRAIN && blockIdx.x == 0 && tid == 0;
  StepState step_state{};
  if (stepper) step_state = step_load(stats);

  lds_barrier();
  NCF_STAMP(1);
  constexpr int kPerE = kFR * kFMaxE / kFThreads;  // GMF elements per thread (2)
  {
    static_assert(kFMaxIn == kFThreads && kFMaxE == kWave, "gather mapping");
    const bool halves = K0 <= kFThreads / 2;  // block-uniform
    const int col_t = halves ? tid & (kFThreads / 2 - 1) : tid;
    const int row0 = halves ? (tid / (kFThreads / 2)) * (kFR / 2) : 0, n_rows_t = halves ? kFR / 2 : kFR;
    float v[kFR];
    const int c = col_t < K0 ? col_t : 0;
    const bool c_user = c < Dm;
    const float* base = c_user ? p.user_mlp + c : p.item_mlp + (c - Dm);
#pragma unroll
    for (int r = 0; r < kFR; ++r) {
      v[r] = 0.f;
      if (r < n_rows_t) {
        const long long idx = c_user ? s_u[row0 + r] : s_i[row0 + r];
        v[r] = base[(idx >= 0 ? idx : 0) * Dm];  // flagged samples read row 0 and are zeroed below
      }
    }
    float w[kPerE], gum[kPerE], gim[kPerE];
    const int ce = (tid & 63) < E ? (tid & 63) : 0;
#pragma unroll
    for (int j = 0; j < kPerE; ++j) {
      const int r = j * (kFThreads / kWave) + (tid >> 6);
      const long long u = s_u[r], it = s_i[r];
      const float um = E > 0 ? p.user_mf[(u >= 0 ? u : 0) * E + ce] : 0.f;
      const float im = E > 0 ? p.item_mf[(it >= 0 ? it : 0) * E + ce] : 0.f;
      w[j] = um * im;
      gum[j] = um;
      gim[j] = im;
    }
    // the Dropout in front of the first Linear (ncf.py:42-45, mlp.py:30-33): one keep byte per element
    uint8_t k0[kFR];
    if constexpr (DROP) {
#pragma unroll
      for (int j = 0; j < kFR; ++j) {
        k0[j] = 1;
        if (j < n_rows_t && p.keep[0] && col_t < K0 && m0 + row0 + j < batch)
          k0[j] = p.keep[0][(m0 + row0 + j) * K0 + col_t];
      }
    }
    if (col_t < K0) {
#pragma unroll
      for (int j = 0; j < kFR; ++j) {
        if (j >= n_rows_t) break;
        const int r = row0 + j;
        float x = s_u[r] >= 0 ? v[j] : 0.f;
        if (p.relu_input) x = fmaxf(x, 0.f);
        if constexpr (DROP)
          if (p.keep[0]) x = k0[j] ? x * p.keep_scale : 0.f;
        lds_raw[r * ld0 + col_t] = x;
      }
    }
    if ((tid & 63) < E) {
#pragma unroll
      for (int j = 0; j < kPerE; ++j) {
        const int r = j * (kFThreads / kWave) + (tid >> 6);
        s_mf[r * kFLdE + (tid & 63)] = s_u[r] >= 0 ? w[j] : 0.f;
        if constexpr (BWD) {
          s_um[r * kFLdE + (tid & 63)] = gum[j];
          s_im[r * kFLdE + (tid & 63)] = gim[j];
        }
      }
    }
  }
  lds_barrier();
  NCF_STAMP(2);

  int in_off = 0, ld_in = ld0;  // act_l: lds_raw + in_off, [kFR][ld_in]
  for (int l = 0; l < n_layers; ++l) {
    const int K = p.layer_in[l], N = p.layer_out[l];
    const int out_off = in_off + kFR * ld_in, ld_out = N + 1;
    float* out = lds_raw + out_off;
    for (int n_off = 0; n_off < N; n_off += kFMaxN) {
      // the keep bytes of this pass's outputs (the NEXT Linear's Dropout) are requested before the GEMM
      uint8_t kb[4] = {1, 1, 1, 1};
      if constexpr (DROP) {
        if (l + 1 < n_layers && p.keep[l + 1] && wn * 16 < N - n_off) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t row = m0 + 4 * (lane >> 4) + r;
            if (row < batch) kb[r] = p.keep[l + 1][row * N + n_off + wn * 16 + (lane & 15)];
          }
        }
      }
      f32x4 acc;
      gemm.run(acc, lds_raw + in_off, ld_in, nullptr);
      // the next pass's (or layer's) first weight chunks and bias travel under this epilogue
      if (n_off + kFMaxN < N) {
        gemm.begin(p.fc_w[l] + static_cast<int64_t>(n_off + kFMaxN) * K, K, min(kFMaxN, N - n_off - kFMaxN));
        bias_next = load_bias(l, n_off + kFMaxN);
      } else if (l + 1 < n_layers) {
        gemm.begin(p.fc_w[l + 1], p.layer_in[l + 1], min(p.layer_out[l + 1], kFMaxN));
        bias_next = load_bias(l + 1, 0);
      }
      if (wn * 16 < N - n_off) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // C[row 4 * (lane >> 4) + r][col lane & 15]
          const int col = n_off + wn * 16 + (lane & 15);
          const int row = 4 * (lane >> 4) + r;
          float v = fmaxf(acc[r] + bias_now, 0.f);
          // act[l + 1] is what the NEXT Linear sees: its Dropout is applied here (the backward's dgrad epilogue
          // applies the same keep bytes)
          if constexpr (DROP)
            if (l + 1 < n_layers && p.keep[l + 1]) v = kb[r] ? v * p.keep_scale : 0.f;
          out[row * ld_out + col] = v;
        }
      }
      lds_barrier();
      bias_now = bias_next;
    }
    NCF_STAMP(3 + l);
    in_off = out_off;
    ld_in = ld_out;
  }

  FusedGemmNN gnn;  // BWD: the first weight chunks of the input-gradient chain travel under the head
  if constexpr (BWD)
    gnn.begin(p.fc_w[n_layers - 1], p.layer_in[n_layers - 1], p.layer_out[n_layers - 1],
              min(p.layer_in[n_layers - 1], kFMaxN), 0);
  // affine_output: wave w takes the dot products of rows w, w + 8, ...
  const float* in = lds_raw + in_off;  // act_L
  const int ld_h = ld_in;
  float vec[kFR / kFWaves][3];
#pragma unroll
  for (int j = 0; j < kFR / kFWaves; ++j) {
    const int r = wave + j * kFWaves;
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = lane + kWave * k;
      vec[j][k] = c < nH ? in[r * ld_h + c] : (c < nV ? s_mf[r * kFLdE + (c - nH)] : 0.f);
      part += vec[j][k] * wout[k];
    }
    const float logit = wave_sum(part) + bo;
    if (lane == 0) s_logit[r] = logit;
  }
  lds_barrier();
  // sigmoid, BCELoss (PyTorch's -100 clamp) and d loss / d logit: lane r of wave 0 = sample r
  float y_mine = 0.f, loss_w = 0.f, gb_w = 0.f;
  if (wave == 0) {
    float loss_l = 0.f, dl = 0.f;
    if (lane < kFR && m0 + lane < batch) {
      y_mine = sigmoid_f32(s_logit[lane]);
      if constexpr (TRAIN) {
        const float ly = fmaxf(logf(y_mine), -100.f);
        const float l1y = fmaxf(log1pf(-y_mine), -100.f);
        loss_l = -(rt * ly + (1.f - rt) * l1y);
        const float gy = (y_mine - rt) / fmaxf((1.f - y_mine) * y_mine, 1e-12f) * inv_batch;
        dl = gy * ((1.f - y_mine) * y_mine);
      }
    }
    if constexpr (TRAIN) {
      if (lane < kFR) s_dl[lane] = dl;
      loss_w = wave_sum(loss_l);
      gb_w = wave_sum(dl);
    }
  }
  NCF_STAMP(8);
  if constexpr (!TRAIN) {
    if (wave == 0 && lane < kFR && m0 + lane < batch) p.scores[m0 + lane] = y_mine;
  }
  if constexpr (TRAIN) {
    lds_barrier();
    // dZ_L (the ReLU of the last Linear, applied twice in NeuMF, masks it), dMF, this wave's share of d w_out
    float gw[3] = {0.f, 0.f, 0.f};
    float* dz_top = lds_raw + in_off + dz_shift;   // BWD: dZ_L, laid out like act_L
#pragma unroll
    for (int j = 0; j < kFR / kFWaves; ++j) {
      const int r = wave + j * kFWaves;
      const int64_t b = m0 + r;
      const float dl = s_dl[r];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = lane + kWave * k;
        if (c < nV) gw[k] += dl * vec[j][k];
        if (c < nH) {
          const float dz = vec[j][k] > 0.f ? dl * wout[k] : 0.f;
          if constexpr (BWD) dz_top[r * ld_h + c] = dz;
          else if (b < batch) p.dact[n_layers][b * nH + c] = dz;
        } else if (c < nV) {
          if constexpr (BWD) s_mf[r * kFLdE + (c - nH)] = dl * wout[k];  // (this lane read the product there)
          else if (b < batch) p.dmf[b * E + (c - nH)] = dl * wout[k];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = lane + kWave * k;
      if (c < nV) s_gw[wave * (3 * kWave) + c] = gw[k];
    }
    NCF_STAMP(9);
    if constexpr (BWD) {
      // ---- the input-gradient chain on the same 16 samples: dZ_{l} = (dZ_{l+1} W_l) * [act_l > 0] (and the Dropout
      // in front of Linear l), every dZ_l kept in LDS; layer 0's result is the gradient of the gathered rows ----
      lds_barrier();
      NCF_STAMP(17);
      int h_off = in_off;  // act_{l+1}; act_l sits right before it
      for (int l = n_layers - 1; l >= 0; --l) {
        const int nin = p.layer_in[l], nout = p.layer_out[l];
        const bool masked = l > 0 || p.relu_input;
        const int ld_h_l = nin + 1, ld_c = nout + 1;
        const float* cin = lds_raw + h_off + dz_shift;   // dZ_{l+1}
        h_off -= kFR * ld_h_l;
        const float* h_l = lds_raw + h_off;
        float* cout = l > 0 ? lds_raw + h_off + dz_shift : d0;   // dZ_l like act_l; layer 0: [kFR][K0 + 1]
        for (int n_off = 0; n_off < nin; n_off += kFMaxN) {
          const int n_pass = min(kFMaxN, nin - n_off);
          const int col = n_off + wn * 16 + (lane & 15);
          const bool live_col = wn * 16 < n_pass;
          uint8_t kb[4] = {1, 1, 1, 1};
          if constexpr (DROP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int64_t row = m0 + 4 * (lane >> 4) + r;
              if (p.keep[l] && live_col && row < batch) kb[r] = p.keep[l][row * nin + col];
            }
          }
          f32x4 acc;
          gnn.run(acc, cin, ld_c, nullptr);
#ifdef NCF_PRINTF
          if (blockIdx.x == 0 && (tid == 0 || tid == 17) && n_off == 0)
            printf("l %d tid %d: cin[0..2] %g %g %g (off %d) ld_c %d nchunks %d ok %d acc %g %g %g %g mask %g col %d ld_h_l %d w0 %g %g\n", l, tid,
                   cin[0], cin[1], cin[2], (int)(cin - lds_raw), ld_c, gnn.n_chunks, (int)gnn.ok, acc[0], acc[1], acc[2], acc[3],
                   h_l[(4 * (lane >> 4)) * ld_h_l + col], col, ld_h_l, gnn.w[0][0], gnn.w[0][1]);
#endif
          if (l == n_layers - 1) NCF_STAMP(18);
          if (n_off + kFMaxN < nin)
            gnn.begin(p.fc_w[l], nin, nout, min(kFMaxN, nin - n_off - kFMaxN), n_off + kFMaxN);
          else if (l > 0)
            gnn.begin(p.fc_w[l - 1], p.layer_in[l - 1], p.layer_out[l - 1], min(p.layer_in[l - 1], kFMaxN), 0);
          if (l == n_layers - 1) NCF_STAMP(19);
          if (live_col) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int lrow = 4 * (lane >> 4) + r;
              // (the mask as a factor: hipcc 7.2 compiled `mask ? acc[r] : 0.f` into "acc[r] = 0" for BOTH arms once the
              // value went to LDS only -- the select's true arm was coalesced away; r06 experiments)
              const float hm = masked ? h_l[lrow * ld_h_l + col] : 1.f;
              float v = acc[r] * (hm > 0.f ? 1.f : 0.f);
              if constexpr (DROP)
                if (p.keep[l]) v = kb[r] ? v * p.keep_scale : 0.f;
              cout[lrow * ld_h_l + col] = v;
            }
          }
          if (l == n_layers - 1) NCF_STAMP(20);
          lds_barrier();
#ifdef NCF_PRINTF
          if (blockIdx.x == 0 && (tid == 0 || tid == 17) && n_off == 0)
            printf("   after l %d tid %d: cout off %d live %d cout[0..2] %g %g %g  row1: %g %g\n", l, tid, (int)(cout - lds_raw), (int)live_col,
                   cout[0], cout[1], cout[2], cout[ld_h_l], cout[ld_h_l + 1]);
#endif
        }
        NCF_STAMP(10 + (n_layers - 1 - l));
      }
    }
  }

  // ---- the block's only stores: one sweep, rows by wave, 256-byte row segments -------------------------------------
  if constexpr (TRAIN) {
    publish_partials<kFWaves>(loss_w, 0.f, gb_w, inv_batch, scratch);  // (barrier inside: s_gw is complete after it)
    lds_barrier();
    if (wave == 0 && lane < kFR && m0 + lane < batch) p.scores[m0 + lane] = y_mine;
    if (stepper) step_store_advanced(stats, step_state);
    // d affine_output.weight: the waves' sums, one row of per-block partial sums for the grouped launch's column sums
    // (or, without the work space, one atomic per block and column)
    for (int c = tid; c < nV; c += kFThreads) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < kFWaves; ++w) t += s_gw[w * (3 * kWave) + c];
      if (gw_ws) gw_ws[static_cast<int64_t>(blockIdx.x) * nV + c] = t;
      else if (t != 0.f) atomic_add_f32(p.g_out_w + c, t);
    }
  }
  NCF_STAMP(13);
  // Activations for the weight-gradient launch: rows by wave, 256-byte row segments.  (16 bytes per lane, eight pieces
  // read from LDS together and stored together, was SLOWER: 51.0 against 49.3 us, same box.)  With BWD the store wave
  // has sent them, and every dZ_l, on their way long ago (fused_store_wave).
  if constexpr (!BWD) {
#pragma unroll
    for (int j = 0; j < kFR / kFWaves; ++j) {
      const int r = wave + j * kFWaves;
      const int64_t b = m0 + r;
      if (b >= batch) continue;
      for (int c = lane; c < K0; c += kWave) p.act[0][b * K0 + c] = lds_raw[r * ld0 + c];
      int off = kFR * ld0;
      for (int l = 0; l < n_layers; ++l) {
        const int N = p.layer_out[l], ld = N + 1;
        for (int c = lane; c < N; c += kWave) p.act[l + 1][b * N + c] = lds_raw[off + r * ld + c];
        off += kFR * ld;
      }
    }
  }
  NCF_STAMP(14);
  if constexpr (BWD) {
    // The embedding rows' gradients: tower input = [user_mlp row | item_mlp row], 256-byte runs of one row; GMF rows:
    // d user_mf = dmf * item_mf and the other way round (dmf sits where the product was, the factors in s_um / s_im).
    // 5 120 float atomics per block, one L2 operation per element (~300 G/s over the chip: 4.5 us of this launch), and
    // no wave can hide them -- issued earlier, by the store wave, they hold up the weight loads behind them in the
    // CU's memory pipeline (chain layer 1: 9.4 k -> 18.6 k cycles).  What replaces them -- every sample's rows as plain
    // rows of the workspace + per-row contribution lists that the tables' sweep walks -- saves 2 us here and costs 3.7 in
    // the sweep (profiles/r06_experiments.md 75).
#pragma unroll
    for (int j = 0; j < kFR / kFWaves; ++j) {
      const int r = wave + j * kFWaves;
      const long long u = s_u[r], it = s_i[r];
      if (m0 + r >= batch || u < 0) continue;
      for (int c = lane; c < K0; c += kWave) {
        const float v = d0[r * ld0 + c];
        if (v != 0.f) {
          if (c < Dm) atomic_add_f32(p.g_user_mlp + u * Dm + c, v);
          else atomic_add_f32(p.g_item_mlp + it * Dm + (c - Dm), v);
        }
      }
      if (lane < E) {
        const float d = s_mf[r * kFLdE + lane];
        if (d != 0.f) {
          atomic_add_f32(p.g_user_mf + u * E + lane, d * s_im[r * kFLdE + lane]);
          atomic_add_f32(p.g_item_mf + it * E + lane, d * s_um[r * kFLdE + lane]);
        }
      }
    }
    NCF_STAMP(16);
  }
}

// ---- backward: dZ_L -> dZ_{L-1} -> ... -> d(tower input), embedding-row gradients scattered on the way out ------
// The input-gradient chain of the tower for 16 samples, in LDS like the forward: dZ_{l-1} = (dZ_l W_l) * [H_{l-1} > 0]
// (and the Dropout in front of Linear l).  The three grouped launches it replaces each paid 11-18 us for a dgrad that
// the next launch had to wait for; the weight / bias gradients, which only READ the dZ_l this kernel writes, follow in
// one grouped launch.  W_l ([nout][nin] row-major) is the B operand as it lies in memory: k = its rows, 16-byte
// staging loads along n.  A pass produces 128 columns (8 waves x 16); the 2 * dim_mlp columns of layer 0 take two.
// The layer-0 pass never writes d(tower input): each lane adds its four elements straight into the user / item
// embedding gradients (what ncf_scatter_kernel did in a launch of its own), and the GMF rows' gradients leave at
// the start, they depend on the head only.
template <bool DROP>
__global__ __launch_bounds__(kFThreads) void ncf_fused_dgrad_kernel(hiprec_ncf_plan p,
                                                                    const int64_t* __restrict__ users,
                                                                    const int64_t* __restrict__ items,
                                                                    int64_t batch) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  float* wide = lds_raw;                     // [kFR][kFLdIn]
  float* narrow = wide + kFR * kFLdIn;       // [kFR][kFLdN]
  float* bs = narrow + kFR * kFLdN;          // [2][kFK][kFLdB]
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kFR;
  const int L = p.n_layers, Dm = p.dim_mlp, E = p.dim_mf;
  __shared__ long long s_u[kFR], s_i[kFR];

  // first weight chunks of the last layer's dgrad, the sample ids, dZ_L: nothing here waits for anything else
  FusedGemmNN gemm;
  gemm.begin(p.fc_w[L - 1], p.layer_in[L - 1], p.layer_out[L - 1], min(p.layer_in[L - 1], kFMaxN), 0);
  if (tid < kFR) {
    const int64_t b = m0 + tid;
    long long u = -1, it = -1;
    if (b < batch) {
      u = users[b];
      it = items[b];
      if (static_cast<uint64_t>(u) >= static_cast<uint64_t>(p.n_users) ||
          static_cast<uint64_t>(it) >= static_cast<uint64_t>(p.n_items))
        u = it = -1;  // flagged by the forward
    }
    s_u[tid] = u;
    s_i[tid] = it;
  }
  // chain buffers alternate; the widest result (layer 0's, if anything of it were kept) belongs in `wide`
  float* in = (L & 1) ? narrow : wide;
  float* out = (L & 1) ? wide : narrow;
  int ld_in = (L & 1) ? kFLdN : kFLdIn, ld_out = (L & 1) ? kFLdIn : kFLdN;
  {
    const int nH = p.layer_out[L - 1];
    for (int e = tid; e < kFR * nH; e += kFThreads) {
      const int r = e / nH, c = e - r * nH;
      in[r * ld_in + c] = m0 + r < batch ? p.dact[L][(m0 + r) * nH + c] : 0.f;
    }
  }
  lds_barrier();
  // GMF rows: d user_mf = dmf * item_mf and the other way round (the head left dmf).  Operands are requested here,
  // the atomics leave after the chain: ahead of it they were on every block's critical path.
  constexpr int kGmfPer = kFR * kFMaxE / kFThreads;  // 2 elements per thread
  float g_d[kGmfPer], g_im[kGmfPer], g_um[kGmfPer];
#pragma unroll
  for (int j = 0; j < kGmfPer; ++j) {
    const int e = tid + j * kFThreads, r = e / kFMaxE, c = e % kFMaxE;
    const long long u = s_u[r], it = s_i[r];
    const bool ok = c < E && u >= 0;
    g_d[j] = ok ? p.dmf[(m0 + r) * E + c] : 0.f;
    g_im[j] = ok ? p.item_mf[it * E + c] : 0.f;
    g_um[j] = ok ? p.user_mf[u * E + c] : 0.f;
  }
  for (int l = L - 1; l >= 0; --l) {
    const int nin = p.layer_in[l], nout = p.layer_out[l];
    const bool masked = l > 0 || p.relu_input;  // [H_{l-1} > 0]; layer 0: NeuMF's ReLU on the raw embeddings (Q7)
    for (int n_off = 0; n_off < nin; n_off += kFMaxN) {
      const int n_pass = min(kFMaxN, nin - n_off);
      // this lane's four (row, column) outputs of the pass: their masks / keep bytes are requested before the GEMM
      const int col = n_off + wn * 16 + (lane & 15);
      const bool live_col = wn * 16 < n_pass;
      float h[4];
      uint8_t kb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + 4 * (lane >> 4) + r;
        const bool ok = live_col && row < batch;
        h[r] = masked && ok ? p.act[l][row * nin + col] : 1.f;
        kb[r] = 1;
        if constexpr (DROP)
          if (p.keep[l] && ok) kb[r] = p.keep[l][row * nin + col];
      }
      f32x4 acc;
      gemm.run(acc, in, ld_in, bs);
      // the next pass's first weight chunks travel under this epilogue
      if (n_off + kFMaxN < nin)
        gemm.begin(p.fc_w[l], nin, nout, min(kFMaxN, nin - n_off - kFMaxN), n_off + kFMaxN);
      else if (l > 0)
        gemm.begin(p.fc_w[l - 1], p.layer_in[l - 1], p.layer_out[l - 1], min(p.layer_in[l - 1], kFMaxN), 0);
      if (live_col) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lrow = 4 * (lane >> 4) + r;
          float v = h[r] > 0.f ? acc[r] : 0.f;
          if constexpr (DROP)
            if (p.keep[l]) v = kb[r] ? v * p.keep_scale : 0.f;
          if (l > 0) {
            out[lrow * ld_out + col] = v;
            if (m0 + lrow < batch) p.dact[l][(m0 + lrow) * nin + col] = v;
          } else if (s_u[lrow] >= 0 && v != 0.f) {  // tower input = [user_mlp row | item_mlp row]
            if (col < Dm) atomic_add_f32(p.g_user_mlp + s_u[lrow] * Dm + col, v);
            else atomic_add_f32(p.g_item_mlp + s_i[lrow] * Dm + (col - Dm), v);
          }
        }
      }
      lds_barrier();
    }
    float* t = in;
    in = out;
    out = t;
    const int tl = ld_in;
    ld_in = ld_out;
    ld_out = tl;
  }
#pragma unroll
  for (int j = 0; j < kGmfPer; ++j) {
    const int e = tid + j * kFThreads, r = e / kFMaxE, c = e % kFMaxE;
    const long long u = s_u[r], it = s_i[r];
    if (c < E && u >= 0 && g_d[j] != 0.f) {
      atomic_add_f32(p.g_user_mf + u * E + c, g_d[j] * g_im[j]);
      atomic_add_f32(p.g_item_mf + it * E + c, g_d[j] * g_um[j]);
    }
  }
}

static int head_grid(int64_t batch, int samples_per_wave) {
  const int64_t per_block = static_cast<int64_t>(kHeadWaves) * samples_per_wave;
  return static_cast<int>(std::min<int64_t>(std::max<int64_t>((batch + per_block - 1) / per_block, 1), 256));
}

static int check_plan(const hiprec_ncf_plan* p, int64_t batch, bool train) {
  HIPREC_REQUIRE(p != nullptr, "NULL plan");
  HIPREC_REQUIRE(p->n_users > 0 && p->n_items > 0, "bad table sizes");
  HIPREC_REQUIRE(p->dim_mlp >= 0 && p->dim_mf >= 0 && (p->dim_mlp > 0 || p->dim_mf > 0),
                 "plan has neither an MLP nor a GMF half");
  HIPREC_REQUIRE(batch >= 0 && batch <= p->max_batch, "batch %lld exceeds the plan's workspace (%lld)",
                 (long long)batch, (long long)p->max_batch);
  HIPREC_REQUIRE(p->out_w && p->out_b && p->scores, "NULL head pointers");
  if (p->dim_mlp > 0) {
    HIPREC_REQUIRE(p->n_layers >= 1 && p->n_layers <= HIPREC_NCF_MAX_LAYERS, "bad n_layers");
    HIPREC_REQUIRE(p->user_mlp && p->item_mlp && p->act[0], "NULL MLP pointers");
    HIPREC_REQUIRE(p->layer_in[0] == 2 * p->dim_mlp, "layer_in[0] != 2*dim_mlp");
    for (int l = 0; l < p->n_layers; ++l) {
      HIPREC_REQUIRE(p->fc_w[l] && p->fc_b[l] && p->act[l + 1], "NULL tower pointer at layer %d", l);
      if (l > 0) HIPREC_REQUIRE(p->layer_in[l] == p->layer_out[l - 1], "layer widths do not chain");
      if (train) HIPREC_REQUIRE(p->g_fc_w[l] && p->g_fc_b[l] && p->dact[l + 1], "NULL grad pointer");
    }
    if (train) HIPREC_REQUIRE(p->dact[0] && p->g_user_mlp && p->g_item_mlp, "NULL MLP grad pointers");
  }
  if (p->dim_mf > 0) {
    HIPREC_REQUIRE(p->user_mf && p->item_mf && p->mf, "NULL GMF pointers");
    if (train) HIPREC_REQUIRE(p->dmf && p->g_user_mf && p->g_item_mf, "NULL GMF grad pointers");
  }
  const int nV = (p->dim_mlp > 0 ? p->layer_out[p->n_layers - 1] : 0) + p->dim_mf;
  HIPREC_REQUIRE(nV <= 320, "affine_output wider than 320 inputs is not supported (got %d)", nV);
  if (train) HIPREC_REQUIRE(p->g_out_w && p->g_out_b, "NULL head grad pointers");
  return 0;
}

static int fused_attrs() {
  static std::atomic<uint64_t> fwd_ok{0}, bwd_ok{0};
  if (int rc = allow_dynamic_lds({reinterpret_cast<const void*>(&ncf_fused_forward_kernel<false, false>),
                                  reinterpret_cast<const void*>(&ncf_fused_forward_kernel<false, true>),
                                  reinterpret_cast<const void*>(&ncf_fused_forward_kernel<true, false>),
                                  reinterpret_cast<const void*>(&ncf_fused_forward_kernel<true, true>),
                                  reinterpret_cast<const void*>(&ncf_fused_forward_kernel<true, false, true>),
                                  reinterpret_cast<const void*>(&ncf_fused_forward_kernel<true, true, true>)},
                                 kFusedMaxLds, fwd_ok, "the fused NCF forward"))
    return rc;
  return allow_dynamic_lds({reinterpret_cast<const void*>(&ncf_fused_dgrad_kernel<false>),
                            reinterpret_cast<const void*>(&ncf_fused_dgrad_kernel<true>)},
                           kFusedBwdLdsBytes, bwd_ok, "the fused NCF input-gradient chain");
}

// Tower forward.  Returns (through *scored) whether plan->scores already holds the sigmoid outputs.
// With `ratings` (training) the fused launch also does the head's backward half; *scored then means
// "plan->scores, dact[L], dmf, g_out_w and the loss partials are all in place".
static int forward(const hiprec_ncf_plan* p, const int64_t* users, const int64_t* items,
                   int64_t batch, hiprec_stats* stats, hipStream_t st, bool* scored,
                   const float* ratings = nullptr, float inv_batch = 0.f, Scratch* scratch = nullptr,
                   bool* chained = nullptr, float* gw_ws = nullptr) {
  // chained (training only): if given and the fused launch is taken, it also runs the tower's input-gradient chain
  // and the embedding scatter (*chained = true); the caller then only owes the weight / bias gradients
  *scored = false;
  // the training launch publishes one loss partial per block: batches beyond kMaxBlocks * kFR samples (32 768) take
  // the launch-per-layer path
  const bool chain = ratings && chained && fusable(p, true);
  if (fusable(p) && (!ratings || (batch + kFR - 1) / kFR <= kMaxBlocks)) {
    if (int rc = fused_attrs()) return rc;
    const int grid = static_cast<int>((batch + kFR - 1) / kFR);
    const size_t lds = sizeof(float) * fused_lds_floats(p, chain);
    bool drop = false;
    for (int l = 0; l < p->n_layers; ++l) drop = drop || p->keep[l] != nullptr;
    if (chain) {
      if (drop)
        ncf_fused_forward_kernel<true, true, true><<<grid, kFThreads + kWave, lds, st>>>(
            *p, users, items, ratings, batch, inv_batch, stats, scratch, gw_ws);
      else
        ncf_fused_forward_kernel<true, false, true><<<grid, kFThreads + kWave, lds, st>>>(
            *p, users, items, ratings, batch, inv_batch, stats, scratch, gw_ws);
      *chained = true;
    } else if (ratings && drop)
      ncf_fused_forward_kernel<true, true><<<grid, kFThreads, lds, st>>>(
          *p, users, items, ratings, batch, inv_batch, stats, scratch, nullptr);
    else if (ratings)
      ncf_fused_forward_kernel<true, false><<<grid, kFThreads, lds, st>>>(
          *p, users, items, ratings, batch, inv_batch, stats, scratch, nullptr);
    else if (drop)   // model.train() + forward(): the reference applies dropout there too
      ncf_fused_forward_kernel<false, true><<<grid, kFThreads, lds, st>>>(
          *p, users, items, nullptr, batch, 0.f, stats, nullptr, nullptr);
    else
      ncf_fused_forward_kernel<false, false><<<grid, kFThreads, lds, st>>>(
          *p, users, items, nullptr, batch, 0.f, stats, nullptr, nullptr);
    HIPREC_TRY(hipGetLastError());
    *scored = true;
    return 0;
  }
  ncf_gather_kernel<<<grid_for_waves(batch), kBlock, 0, st>>>(*p, users, items, batch, stats);
  HIPREC_TRY(hipGetLastError());
  if (p->dim_mlp > 0) {
    for (int l = 0; l < p->n_layers; ++l) {
      GemmGroup g{};
      g.n = 1;
      g.p[0] = make_gemm(kNT, static_cast<int>(batch), p->layer_out[l], p->layer_in[l], p->act[l],
                         p->layer_in[l], p->fc_w[l], p->layer_in[l], p->act[l + 1], p->layer_out[l], p->fc_b[l],
                         /*relu=*/1, nullptr, 0, false);
      if (l + 1 < p->n_layers && p->keep[l + 1]) {  // act[l+1] is stored AFTER the next Linear's Dropout
        g.p[0].keep = p->keep[l + 1];
        g.p[0].ldk = p->layer_out[l];
        g.p[0].keep_scale = p->keep_scale;
      }
      if (int rc = launch_group(g, st)) return rc;
    }
  }
  return 0;
}

}  // namespace hiprec

using namespace hiprec;

extern "C" size_t hiprec_ncf_plan_bytes(void) { return sizeof(hiprec_ncf_plan); }

extern "C" int hiprec_gemm_f32(int mode, int M, int N, int K, const float* A, int lda,
                               const float* B, int ldb, float* C, int ldc, const float* bias,
                               int relu, const float* mask, int ldm, void* stream) {
  HIPREC_REQUIRE(M >= 0 && N >= 0 && K >= 0, "negative GEMM size");
  if (M == 0 || N == 0) return 0;
  HIPREC_REQUIRE(A && B && C, "NULL GEMM operand");
  return launch_gemm(mode, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, mask, ldm,
                     static_cast<hipStream_t>(stream));
}

extern "C" int hiprec_ncf_forward(const hiprec_ncf_plan* plan, const int64_t* users,
                                  const int64_t* items, int64_t batch, hiprec_stats* stats,
                                  void* stream) {
  if (int rc = check_plan(plan, batch, false)) return rc;
  if (batch == 0) return 0;
  HIPREC_REQUIRE(users && items && stats, "NULL pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  bool scored = false;
  if (int rc = forward(plan, users, items, batch, stats, st, &scored)) return rc;
  if (!scored) {
    ncf_head_kernel<false><<<head_grid(batch, 1), kHeadBlock, 0, st>>>(*plan, nullptr, batch, 0.f, stats,
                                                                     nullptr);
    HIPREC_TRY(hipGetLastError());
  }
  return 0;
}

// sweep (optional): an optimizer sweep of the tables' part of the flat buffers that rides in the grouped weight-
// gradient launch (the embedding gradients are complete when that launch starts); *swept tells the caller whether
// that launch existed (the fused forms) or the whole sweep is still owed.
static int ncf_grad_impl(const hiprec_ncf_plan* plan, const int64_t* users, const int64_t* items,
                         const float* ratings, int64_t batch, float inv_batch, hiprec_stats* stats, void* scratch,
                         size_t scratch_bytes, void* stream, const SweepArgs* sweep, bool* swept) {
  if (swept) *swept = false;
  if (int rc = check_plan(plan, batch, true)) return rc;
  HIPREC_REQUIRE(batch > 0, "empty batch");
  HIPREC_REQUIRE(users && items && ratings && stats && scratch, "NULL pointer");
  if (scratch_bytes < kScratchBytes) {
    set_error("scratch too small: %zu < %zu", scratch_bytes, kScratchBytes);
    return HIPREC_E_SCRATCH;
  }
  const hiprec_ncf_plan* p = plan;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int B = static_cast<int>(batch);
  // The default is forward + chain in one launch; the two older forms it falls back to for shapes outside the fused
  // limits -- "unfused" = one grouped launch per layer + scatter, "split" = the chain in its own launch after the
  // forward -- can be FORCED through HIPREC_NCF_BACKWARD only in the test build (-DHIPREC_TEST_SWITCHES,
  // libhiprec_test.so); the product library reads no environment variable.
#ifdef HIPREC_TEST_SWITCHES
  static const char* bwd_env = getenv("HIPREC_NCF_BACKWARD");
  static const bool no_fused_bwd = bwd_env && strcmp(bwd_env, "unfused") == 0;
  static const bool split_bwd = bwd_env && strcmp(bwd_env, "split") == 0;
#else
  constexpr bool no_fused_bwd = false, split_bwd = false;
#endif
  const bool group_ok = !no_fused_bwd && 2 * p->n_layers + 1 <= kMaxGroup;  // one grouped launch for all weight gradients
  const bool fuse_bwd = group_ok && (fusable(p, true) || fusable_narrow(p));
  // d affine_output.weight leaves the chained launch as one row of partial sums per 16-sample block; plan->dact[0]
  // (the tower input's gradient: that launch scatters it straight into the embedding gradients and never writes it)
  // is the work space, the grouped launch adds the rows up.  (A batch too small for its dact[0] to hold them: atomics.)
  const int nV_head = p->layer_out[p->n_layers > 0 ? p->n_layers - 1 : 0] + p->dim_mf;
  const int64_t n_tiles = (batch + kFR - 1) / kFR;
  float* gw_ws = (p->dim_mlp > 0 && n_tiles * nV_head <= batch * 2 * p->dim_mlp) ? p->dact[0] : nullptr;
  bool scored = false, chained = false;
  if (int rc = forward(p, users, items, batch, stats, st, &scored, ratings, inv_batch,
                       static_cast<Scratch*>(scratch), fuse_bwd && !split_bwd ? &chained : nullptr, gw_ws))
    return rc;
  // not chained onto the forward (batch beyond its limit, "split"): the chain's own launch has narrower limits
  const bool bwd_two_launches = fuse_bwd && (chained || fusable_narrow(p));
  if (!scored) {
    ncf_head_kernel<true><<<head_grid(batch, 4), kHeadBlock, 0, st>>>(
        *p, ratings, batch, inv_batch, stats, static_cast<Scratch*>(scratch));
    HIPREC_TRY(hipGetLastError());
  }
  if (bwd_two_launches) {
    // the input-gradient chain + the embedding scatter ride on the forward launch (or, when that took the
    // launch-per-layer path, run in ONE launch of their own), then every layer's weight and bias gradients -- which
    // only read the dZ_l the chain wrote -- in one grouped launch: 2 launches for the 5 of round 1
    if (!chained) {
      if (int rc = fused_attrs()) return rc;
      bool drop = false;
      for (int l = 0; l < p->n_layers; ++l) drop = drop || p->keep[l] != nullptr;
      const int grid = static_cast<int>((batch + kFR - 1) / kFR);
      if (drop)
        ncf_fused_dgrad_kernel<true><<<grid, kFThreads, kFusedBwdLdsBytes, st>>>(*p, users, items, batch);
      else
        ncf_fused_dgrad_kernel<false><<<grid, kFThreads, kFusedBwdLdsBytes, st>>>(*p, users, items, batch);
      HIPREC_TRY(hipGetLastError());
    }
    GemmGroup g{};
    g.n = 0;
    for (int l = 0; l < p->n_layers; ++l) {
      const int nin = p->layer_in[l], nout = p->layer_out[l];
      g.p[g.n++] = make_gemm(kTNm, nout, nin, B, p->dact[l + 1], nout, p->act[l], nin, p->g_fc_w[l], nin, nullptr, 0,
                             nullptr, 0, /*split_k=*/true);
      g.p[g.n++] = make_colsum(p->dact[l + 1], B, nout, nout, p->g_fc_b[l]);
    }
    if (chained && gw_ws)
      g.p[g.n++] = make_colsum(gw_ws, static_cast<int>(n_tiles), nV_head, nV_head, p->g_out_w);
    if (sweep && sweep->n_blocks > 0) {
      g.sweep = *sweep;
      *swept = true;
    }
    return launch_group(g, st);
  }
  if (p->dim_mlp > 0) {
    for (int l = p->n_layers - 1; l >= 0; --l) {
      const int nin = p->layer_in[l], nout = p->layer_out[l];
      // One grouped launch per layer, three problems that all read dZ_l:
      //   dW_l = dZ_l^T H_{l-1}   (split-K into the all-zero gradient buffer)
      //   db_l = column sums of dZ_l
      //   dZ_{l-1} = (dZ_l W_l) * [H_{l-1} > 0]; for l == 0 the mask is the ReLU NeuMF applies to
      //   the raw embeddings (quirk Q7) and is absent for the stand-alone MLP
      const float* mask = (l > 0 || p->relu_input) ? p->act[l] : nullptr;
      GemmGroup g{};
      g.n = 3;
      g.p[0] = make_gemm(kNN, B, nin, nout, p->dact[l + 1], nout, p->fc_w[l], nin, p->dact[l], nin, nullptr,
                         0, mask, nin, false);
      if (p->keep[l]) {  // backward of the Dropout in front of Linear l
        g.p[0].keep = p->keep[l];
        g.p[0].ldk = nin;
        g.p[0].keep_scale = p->keep_scale;
      }
      g.p[1] = make_gemm(kTNm, nout, nin, B, p->dact[l + 1], nout, p->act[l], nin, p->g_fc_w[l], nin, nullptr,
                         0, nullptr, 0, /*split_k=*/true);
      g.p[2] = make_colsum(p->dact[l + 1], B, nout, nout, p->g_fc_b[l]);
      if (int rc = launch_group(g, st)) return rc;
    }
  }
  ncf_scatter_kernel<<<grid_for_waves(batch), kBlock, 0, st>>>(*p, users, items, batch);
  HIPREC_TRY(hipGetLastError());
  return 0;
}


extern "C" int hiprec_ncf_grad(const hiprec_ncf_plan* plan, const int64_t* users,
                               const int64_t* items, const float* ratings, int64_t batch,
                               float inv_batch, hiprec_stats* stats, void* scratch,
                               size_t scratch_bytes, void* stream) {
  return ncf_grad_impl(plan, users, items, ratings, batch, inv_batch, stats, scratch, scratch_bytes, stream, nullptr,
                       nullptr);
}

// hiprec_ncf_grad + optimizer.step() (ncf.py:100-120) as ONE call.  The flat buffers hold [tables | tower | head];
// the first table_floats elements (a multiple of 4) are the embedding tables, whose gradients are complete after
// the forward + chain launch: their share of the dense sweep -- 97 % of the parameters at ncf_default.json's shape --
// runs as extra blocks of the grouped weight-gradient launch, and only the tower / head tail (tens of thousands of
// parameters) keeps a launch of its own behind it.  Round 2: 3 launches, the last one a 37 MB sweep (9.7 us of 59).
extern "C" int hiprec_ncf_step(const hiprec_ncf_plan* plan, const int64_t* users, const int64_t* items,
                               const float* ratings, int64_t batch, float inv_batch, int kind, float* w_flat,
                               float* g_flat, float* m_flat, float* v_flat, int64_t n_flat, int64_t table_floats,
                               int64_t scalar_index, double lr, double beta1, double beta2, double eps,
                               hiprec_stats* stats, void* scratch, size_t scratch_bytes, void* stream) {
  HIPREC_REQUIRE(w_flat && g_flat && stats && scratch, "NULL pointer");
  HIPREC_REQUIRE(n_flat > 0 && table_floats >= 0 && table_floats <= n_flat, "bad flat sizes");
  HIPREC_REQUIRE(kind == HIPREC_OPT_SGD || (kind == HIPREC_OPT_ADAM && m_flat && v_flat) ||
                     (kind == HIPREC_OPT_RMSPROP && v_flat),
                 "optimizer state missing / unknown optimizer");
  HIPREC_REQUIRE(scalar_index < 0 || scalar_index >= table_floats, "the deferred scalar lies behind the tables");
  SweepArgs sw{};
  const bool aligned = (table_floats & 3) == 0 && (reinterpret_cast<uintptr_t>(w_flat) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(g_flat) & 15) == 0 && (reinterpret_cast<uintptr_t>(m_flat) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(v_flat) & 15) == 0;
  if (aligned && table_floats >= (1 << 16)) {
    sw.w = w_flat;
    sw.g = g_flat;
    sw.m = m_flat;
    sw.v = v_flat;
    sw.n4 = table_floats >> 2;
    sw.kind = kind;
    sw.n_blocks = static_cast<int>(std::min<int64_t>((sw.n4 + kBlock - 1) / kBlock, HIPREC_SWEEP_CAP));  // one 16-byte vector per thread
    sw.s = OptScalars{lr, static_cast<float>(lr), static_cast<float>(beta2), static_cast<float>(1.0 - beta1),
                      static_cast<float>(1.0 - beta2), static_cast<float>(eps)};
    sw.stats = stats;
  }
  bool swept = false;
  if (int rc = ncf_grad_impl(plan, users, items, ratings, batch, inv_batch, stats, scratch, scratch_bytes, stream, &sw,
                             &swept))
    return rc;
  const int64_t done = swept ? table_floats : 0;
  return hiprec_opt_dense_step(kind, w_flat + done, g_flat + done, m_flat ? m_flat + done : nullptr,
                               v_flat ? v_flat + done : nullptr, n_flat - done, lr, beta1, beta2, eps, stats, scratch,
                               scalar_index >= 0 ? scalar_index - done : -1, stream);
}

#ifdef HIPREC_NCF_DEBUG
extern "C" int hiprec_debug_gemm_stamps(unsigned long long* out32) {
  HIPREC_TRY(hipDeviceSynchronize());
  HIPREC_TRY(hipMemcpyFromSymbol(out32, HIP_SYMBOL(hiprec::g_gemm_stamps), sizeof(unsigned long long) * 32));
  return 0;
}
extern "C" int hiprec_debug_ncf_stamps(unsigned long long* out48) {
  HIPREC_TRY(hipDeviceSynchronize());
  HIPREC_TRY(hipMemcpyFromSymbol(out48, HIP_SYMBOL(hiprec::g_ncf_stamps), sizeof(unsigned long long) * 48));
  return 0;
}
#endif
