// NGCF on the LightGCN SpMM and the NCF grouped GEMM — SURVEY.md §8f rank 4 (sibling models).
//
//   beta_rec/models/ngcf.py:48-80     forward, per hop l (ego_0 = cat(user_embedding, item_embedding)):
//                                       side = A ego_l
//                                       ego_{l+1} = dropout(lrelu(GC_l(side)) + lrelu(Bi_l(ego_l * side)))
//                                     all = cat(ego_0, normalize(ego_1), ..., normalize(ego_L))  along dim 1
//   beta_rec/models/ngcf.py:82-100    predict: <all[u], all[U + i]>
//   beta_rec/models/ngcf.py:118-149   train_single_batch: gather rows of `all`, bpr_loss, backward
//   beta_rec/models/ngcf.py:172-199   bpr_loss: -mean logsigmoid(s+ - s-) + decay * (|u|^2+|p|^2+|n|^2)/2 / batch_size
//
// Every hop is a short chain of HBM-bound passes over [N, d] activations (N = users + items, d <= 256):
//   SpMM (lightgcn.hip)  ->  bi_mul  ->  ONE grouped launch of the two Linear layers (fp32 MFMA, ncf.hip)
//   ->  act (lrelu + lrelu, dropout drawn or read, row L2 norm, write the hop's slice of `all`)
// and the backward walks it in reverse:
//   act_bwd (normalize / dropout / lrelu backward -> d_sum, d_bi)  ->  ONE grouped launch of six problems
//   (two dgrads, two wgrads, two bias column sums)  ->  bi_bwd  ->  SpMM with the transposed graph.
// The loss gathers / scatters rows of the concatenated table directly (one wave per triple).
// Nothing here is GEMM-bound: 2 x N x d x d flops per Linear (80 MFLOP at ML-1M size) against 2.5 MB
// activations; the MFMA group is used because it is the exact-fp32 GEMM the NCF tower already has.
#include <algorithm>
#include <cstdlib>

#include "common.hpp"
#include "gemm.hpp"
#include "spmm.hpp"

namespace hiprec {

constexpr int kNgcfMaxNpl = 4;  // hop widths <= 256: columns lane, lane + 64, ...
constexpr float kSlope = 0.01f; // F.leaky_relu default negative_slope
constexpr float kNormEps = 1e-12f;

__device__ __forceinline__ float lrelu(float x) { return x > 0.f ? x : x * kSlope; }

// bi_in = ego * side   (n floats, 16-B vectors + tail)
__global__ __launch_bounds__(kBlock) void ngcf_bi_mul_kernel(const float* __restrict__ ego,
                                                             const float* __restrict__ side,
                                                             float* __restrict__ bi_in, int64_t n) {
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  const int64_t n4 = n >> 2;
  for (int64_t i = tid; i < n4; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(ego)[i];
    const float4 b = reinterpret_cast<const float4*>(side)[i];
    reinterpret_cast<float4*>(bi_in)[i] = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
  }
  for (int64_t i = (n4 << 2) + tid; i < n; i += stride) bi_in[i] = ego[i] * side[i];
}

// One wave per node row: ego' = keep * scale * (lrelu(sum_pre) + lrelu(bi_pre)); nrm = |ego'|_2;
// all[row, off : off + d] = ego' / max(nrm, eps).  gen: draw the keep bytes here (and store them for the
// backward) instead of reading them -- one launch less per hop than a separate mask kernel.
struct KeepGen {
  int on;
  float keep_prob;
  uint64_t seed, step;
};

__global__ __launch_bounds__(kBlock) void ngcf_act_kernel(const float* __restrict__ sum_pre,
                                                          const float* __restrict__ bi_pre,
                                                          uint8_t* __restrict__ keep, float scale, KeepGen gen,
                                                          float* __restrict__ ego_out,
                                                          float* __restrict__ nrm_out,
                                                          float* __restrict__ all, int ld_all, int off,
                                                          int64_t n_rows, int d, SlicedOut next_src) {
  const int lane = lane_id();
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block(); r < n_rows;
       r += static_cast<int64_t>(gridDim.x) * kWavesPerBlock) {
    float x[kNgcfMaxNpl];
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < kNgcfMaxNpl; ++k) {
      const int c = lane + kWave * k;
      x[k] = 0.f;
      if (c < d) {
        const int64_t i = r * d + c;
        float v = lrelu(sum_pre[i]) + lrelu(bi_pre[i]);
        if (keep) {
          bool k;
          if (gen.on) {
            k = keep_draw(gen.seed, gen.step, i, gen.keep_prob);
            keep[i] = k ? 1 : 0;
          } else {
            k = keep[i] != 0;
          }
          v = k ? v * scale : 0.f;
        }
        x[k] = v;
        ego_out[i] = v;
        next_src.put(r, c, v);  // the next hop's sliced SpMM source
      }
      sq += x[k] * x[k];
    }
    const float nrm = sqrtf(wave_sum(sq));
    const float inv = 1.0f / fmaxf(nrm, kNormEps);
    if (lane == 0) nrm_out[r] = nrm;
#pragma unroll
    for (int k = 0; k < kNgcfMaxNpl; ++k) {
      const int c = lane + kWave * k;
      if (c < d) all[r * ld_all + off + c] = x[k] * inv;
    }
  }
}

// ---- one hop's per-node chain in ONE launch: bi = ego * side, the two Linear layers, leaky-ReLUs, dropout, row norm ----
// What ngcf_bi_mul_kernel + the grouped GEMM launch + ngcf_act_kernel do in three launches (4.9 + 12.3 + 7.3 us at
// the ML-1M graph: each a pass over [N, d] activations that starts by missing on what the previous one wrote).  A
// workgroup of 8 waves owns 16 node rows (the NCF recipe, ncf.hip): both weight matrices (d x d: 16 KB each) go to LDS
// whole, waves 0-3 multiply side by GC^T and waves 4-7 ego * side by Bi^T on 16x16x4 fp32 MFMAs (one 16-column tile
// each), the two results meet in LDS, and two rows per wave get activation, dropout, norm and all their outputs.
// Limits: input width <= 128 (a multiple of 4), output width <= 64 (a multiple of 16).
using hop_f32x4 = float __attribute__((ext_vector_type(4)));
constexpr int kHopRows = 16, kHopThreads = 512, kHopMaxIn = 128, kHopMaxOut = 64;
constexpr int kHopW4 = kHopMaxOut * kHopMaxIn / 4 / (kHopThreads / 2);  // float4 of one matrix per thread (8)

static size_t hop_lds_floats(int di, int dout) {
  return 2 * static_cast<size_t>(kHopRows) * (di + 1) + 2 * static_cast<size_t>(di) * (dout + 1) +
         2 * static_cast<size_t>(kHopRows) * (dout + 1);
}

__global__ __launch_bounds__(kHopThreads) void ngcf_hop_forward_kernel(
    const float* __restrict__ side, const float* __restrict__ ego_in, const float* __restrict__ gc_w,
    const float* __restrict__ gc_b, const float* __restrict__ bi_w, const float* __restrict__ bi_b, int di, int dout,
    int64_t n_rows, float* __restrict__ bi_in, float* __restrict__ sum_pre, float* __restrict__ bi_pre,
    uint8_t* __restrict__ keep, float scale, KeepGen gen, float* __restrict__ ego_out, float* __restrict__ nrm_out,
    float* __restrict__ all, int ld_all, int off, SlicedOut next_src) {
  extern __shared__ __attribute__((aligned(16))) float hop_lds[];
  const int lda = di + 1, ldw = dout + 1, ldx = dout + 1;
  float* a_side = hop_lds;                       // [16][di + 1]
  float* a_bi = a_side + kHopRows * lda;         // [16][di + 1]
  float* wt = a_bi + kHopRows * lda;             // [2][di][dout + 1]: GC^T, Bi^T
  float* xs = wt + 2 * di * ldw;                 // [2][16][dout + 1]: lrelu(GC(side)), lrelu(Bi(ego * side))
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wn = wave & 3;      // which Linear, which 16-column tile
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kHopRows;

  // every load first: this thread's share of one weight matrix (float4 along k), its bias, its tile elements
  const float* __restrict__ W = grp ? bi_w : gc_w;
  const int half_tid = tid & (kHopThreads / 2 - 1), k4s = di >> 2, n_w4 = dout * k4s;
  float4 wreg[kHopW4];
#pragma unroll
  for (int j = 0; j < kHopW4; ++j) {
    const int q = half_tid + j * (kHopThreads / 2);
    wreg[j] = q < n_w4 ? reinterpret_cast<const float4*>(W)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int col_e = wn * 16 + (lane & 15);
  const float bias = col_e < dout ? (grp ? bi_b : gc_b)[col_e] : 0.f;
  constexpr int kPerT = kHopRows * kHopMaxIn / kHopThreads;  // 4 tile elements per thread
  float sv[kPerT], ev[kPerT];
#pragma unroll
  for (int j = 0; j < kPerT; ++j) {
    const int e = tid + j * kHopThreads, r = e / di, c = e - r * di;
    const bool ok = e < kHopRows * di && m0 + r < n_rows;
    sv[j] = ok ? side[(m0 + r) * di + c] : 0.f;
    ev[j] = ok ? ego_in[(m0 + r) * di + c] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < kHopW4; ++j) {
    const int q = half_tid + j * (kHopThreads / 2);
    if (q < n_w4) {  // W[n][k4 .. k4 + 3] -> transposed tile [k][n]
      const int n = q / k4s, k4 = (q - n * k4s) * 4;
      float* d = wt + grp * di * ldw + k4 * ldw + n;
      d[0] = wreg[j].x;
      d[ldw] = wreg[j].y;
      d[2 * ldw] = wreg[j].z;
      d[3 * ldw] = wreg[j].w;
    }
  }
#pragma unroll
  for (int j = 0; j < kPerT; ++j) {
    const int e = tid + j * kHopThreads, r = e / di, c = e - r * di;
    if (e < kHopRows * di) {
      const float b = sv[j] * ev[j];
      a_side[r * lda + c] = sv[j];
      a_bi[r * lda + c] = b;
      if (m0 + r < n_rows) bi_in[(m0 + r) * di + c] = b;
    }
  }
  lds_barrier();
  // the wave's 16 x 16 tile of its Linear; operands of eight k-steps are read before their MFMAs
  hop_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (wn * 16 < dout) {
    const float* A = grp ? a_bi : a_side;
    const float* B = wt + grp * di * ldw;
    const int i = lane & 15, kq = lane >> 4;
    for (int k0 = 0; k0 < di; k0 += 32) {
      float a[8], b[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + 4 * j + kq;
        a[j] = k < di ? A[i * lda + k] : 0.f;
        b[j] = k < di ? B[k * ldw + wn * 16 + i] : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // C[row 4 * (lane >> 4) + r][col lane & 15]
      const int row = 4 * (lane >> 4) + r;
      const float pre = acc[r] + bias;
      if (m0 + row < n_rows) (grp ? bi_pre : sum_pre)[(m0 + row) * dout + col_e] = pre;
      xs[(grp * kHopRows + row) * ldx + col_e] = lrelu(pre);
    }
  }
  lds_barrier();
  // rows wave, wave + 8: lane = column
#pragma unroll
  for (int j = 0; j < kHopRows / (kHopThreads / kWave); ++j) {
    const int r = wave + j * (kHopThreads / kWave);
    const int64_t row = m0 + r;
    if (row >= n_rows) continue;
    float v = 0.f;
    if (lane < dout) {
      v = xs[r * ldx + lane] + xs[(kHopRows + r) * ldx + lane];
      const int64_t i = row * dout + lane;
      if (keep) {
        bool k;
        if (gen.on) {
          k = keep_draw(gen.seed, gen.step, i, gen.keep_prob);
          keep[i] = k ? 1 : 0;
        } else {
          k = keep[i] != 0;
        }
        v = k ? v * scale : 0.f;
      }
      ego_out[i] = v;
      next_src.put(row, lane, v);  // the next hop's sliced SpMM source
    }
    const float nrm = sqrtf(wave_sum(v * v));
    const float inv = 1.0f / fmaxf(nrm, kNormEps);
    if (lane == 0) nrm_out[row] = nrm;
    if (lane < dout) all[row * ld_all + off + lane] = v * inv;
  }
}

// ---- ... and the hop's backward chain in ONE launch: normalize / dropout / leaky-ReLU backward, both input gradients,
// the bilinear backward.  What ngcf_act_bwd_kernel + the two dgrad problems of the grouped launch + ngcf_bi_bwd_kernel
// do (5.6 + ~7 + 5.7 us), on the same 16 node rows: two rows per wave produce d_sum / d_bi (kept per hop for the
// weight gradients, which then all go into ONE grouped launch after the last hop), waves 0-3 multiply d_sum by GC and
// waves 4-7 d_bi by Bi (the weights as they lie in memory are the B operands), the bilinear term joins them in LDS and
// d_ego / d_side (+ the transposed SpMM's sliced source) leave from registers.
// Limits: both widths <= 64, input width a multiple of 16, output width a multiple of 4.
constexpr int kHopBwdMax = 64;

static size_t hop_bwd_lds_floats(int di, int dout) {
  return 2 * static_cast<size_t>(kHopRows) * (dout + 1) + 2 * static_cast<size_t>(dout) * (di + 4) +
         static_cast<size_t>(kHopRows) * (di + 1);
}

template <bool ACCUMULATE>
__global__ __launch_bounds__(kHopThreads) void ngcf_hop_backward_kernel(
    float* __restrict__ d_all, const float* __restrict__ all, int ld_all, int off, const float* __restrict__ nrm,
    const float* __restrict__ d_next, const uint8_t* __restrict__ keep, float scale,
    const float* __restrict__ sum_pre, const float* __restrict__ bi_pre, const float* __restrict__ gc_w,
    const float* __restrict__ bi_w, const float* __restrict__ side, const float* __restrict__ ego_in, int di, int dout,
    int64_t n_rows, float* __restrict__ d_sum, float* __restrict__ d_bi, float* __restrict__ d_ego,
    float* __restrict__ d_side, SlicedOut spmm_src) {
  extern __shared__ __attribute__((aligned(16))) float hop_lds[];
  const int ldd = dout + 1, ldw = di + 4, ldg = di + 1;
  float* wt = hop_lds;                           // [2][dout][di + 4]: GC, Bi as stored ([k = out][n = in]); 16-B rows
  float* ds = wt + 2 * dout * ldw;               // [2][16][dout + 1]: d_sum, d_bi of the tile
  float* gt = ds + 2 * kHopRows * ldd;           // [16][di + 1]: d_bi_in = d_bi Bi
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wn = wave & 3;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kHopRows;

  // loads that wait for nothing: this thread's share of one weight matrix, and (waves 0-3) side / ego of its outputs
  const float* __restrict__ W = grp ? bi_w : gc_w;
  const int half_tid = tid & (kHopThreads / 2 - 1), n4s = di >> 2, n_w4 = dout * n4s;
  constexpr int kW4 = kHopBwdMax * kHopBwdMax / 4 / (kHopThreads / 2);  // 4
  float4 wreg[kW4];
#pragma unroll
  for (int j = 0; j < kW4; ++j) {
    const int q = half_tid + j * (kHopThreads / 2);
    wreg[j] = q < n_w4 ? reinterpret_cast<const float4*>(W)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int col_e = wn * 16 + (lane & 15);
  float sv[4] = {0.f, 0.f, 0.f, 0.f}, ev[4] = {0.f, 0.f, 0.f, 0.f};
  if (grp == 0 && col_e < di) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = m0 + 4 * (lane >> 4) + r;
      if (row < n_rows) {
        sv[r] = side[row * di + col_e];
        ev[r] = ego_in[row * di + col_e];
      }
    }
  }
  // rows wave, wave + 8: normalize / dropout / leaky-ReLU backward (lane = column)
#pragma unroll
  for (int j = 0; j < kHopRows / (kHopThreads / kWave); ++j) {
    const int r = wave + j * (kHopThreads / kWave);
    const int64_t row = m0 + r;
    float a = 0.f, b = 0.f;
    if (row < n_rows) {
      const bool live = lane < dout;
      const float dy = live ? d_all[row * ld_all + off + lane] : 0.f;
      const float y = live ? all[row * ld_all + off + lane] : 0.f;
      // d_all is read exactly once per step (each hop its own columns): whoever reads it leaves it zero for the
      // next step's loss scatter -- the step had a 10 MB fill for this (5.4 us)
      if (live) d_all[row * ld_all + off + lane] = 0.f;
      const float proj = wave_sum(y * dy);
      const float n = nrm[row];
      const bool big = n >= kNormEps;  // clamp_min passes the norm's gradient only where norm >= eps
      const float inv = 1.0f / fmaxf(n, kNormEps);
      if (live) {
        const int64_t i = row * dout + lane;
        float dx = (big ? dy - y * proj : dy) * inv;
        if (d_next) dx += d_next[i];
        if (keep) dx = keep[i] ? dx * scale : 0.f;
        a = sum_pre[i] > 0.f ? dx : dx * kSlope;
        b = bi_pre[i] > 0.f ? dx : dx * kSlope;
        d_sum[i] = a;
        d_bi[i] = b;
      }
    }
    if (lane < dout) {
      ds[r * ldd + lane] = a;
      ds[(kHopRows + r) * ldd + lane] = b;
    }
  }
#pragma unroll
  for (int j = 0; j < kW4; ++j) {
    const int q = half_tid + j * (kHopThreads / 2);
    if (q < n_w4) {
      const int k = q / n4s, n4 = (q - k * n4s) * 4;
      *reinterpret_cast<float4*>(wt + (grp * dout + k) * ldw + n4) = wreg[j];
    }
  }
  lds_barrier();
  hop_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (wn * 16 < di) {
    const float* A = ds + grp * kHopRows * ldd;
    const float* B = wt + grp * dout * ldw;
    const int i = lane & 15, kq = lane >> 4;
    for (int k0 = 0; k0 < dout; k0 += 32) {
      float a[8], b[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + 4 * j + kq;
        a[j] = k < dout ? A[i * ldd + k] : 0.f;
        b[j] = k < dout ? B[k * ldw + wn * 16 + i] : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
    }
    if (grp == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) gt[(4 * (lane >> 4) + r) * ldg + col_e] = acc[r];
    }
  }
  lds_barrier();
  if (grp == 0 && col_e < di) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // d_ego (= or +=) d_bi_in * side ;  d_side = d_sum GC + d_bi_in * ego
      const int lrow = 4 * (lane >> 4) + r;
      const int64_t row = m0 + lrow;
      if (row >= n_rows) continue;
      const float g = gt[lrow * ldg + col_e];
      const int64_t i = row * di + col_e;
      if (ACCUMULATE) d_ego[i] += g * sv[r];
      else d_ego[i] = g * sv[r];
      const float dsv = acc[r] + g * ev[r];
      d_side[i] = dsv;
      spmm_src.put(row, col_e, dsv);
    }
  }
}

// One wave per triple on rows of the concatenated table (width dt): BPR loss + L2 term, gradient rows
// scattered with atomics.  Hop 0's slice (columns < d0) is not copied anywhere: it is read from the
// embedding tables e0 themselves and its gradient goes straight into their gradient g_e0; columns
// >= d0 live in `all` / `d_all` (row stride dt, their first d0 columns unused).
__global__ __launch_bounds__(kBlock) void ngcf_loss_kernel(const float* __restrict__ e0,
                                                           float* __restrict__ g_e0, int d0,
                                                           const float* __restrict__ all,
                                                           float* __restrict__ d_all, int dt,
                                                           int64_t n_users, int64_t n_items,
                                                           const int64_t* __restrict__ users,
                                                           const int64_t* __restrict__ pos,
                                                           const int64_t* __restrict__ neg, int64_t batch,
                                                           float inv_batch, float reg_coef,
                                                           hiprec_stats* stats, Scratch* scratch) {
  const int lane = lane_id();
  const bool stepper = blockIdx.x == 0 && threadIdx.x == 0;
  StepState step_state{};
  if (stepper) step_state = step_load(stats);
  float loss_acc = 0.f, reg_acc = 0.f;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block(); t < batch;
       t += static_cast<int64_t>(gridDim.x) * kWavesPerBlock) {
    const int64_t u = users[t], p = pos[t], n = neg[t];
    const bool u_ok = static_cast<uint64_t>(u) < static_cast<uint64_t>(n_users);
    const bool i_ok = static_cast<uint64_t>(p) < static_cast<uint64_t>(n_items) &&
                      static_cast<uint64_t>(n) < static_cast<uint64_t>(n_items);
    if (!(u_ok && i_ok)) {
      if (lane == 0)
        atomicOr(&stats->status,
                 (u_ok ? 0u : HIPREC_STATUS_USER_OOB) | (i_ok ? 0u : HIPREC_STATUS_ITEM_OOB));
      continue;
    }
    const int64_t ru = u, rp = n_users + p, rn = n_users + n;
    auto at = [&](int64_t r, int c) { return c < d0 ? e0[r * d0 + c] : all[r * dt + c]; };
    auto grad_at = [&](int64_t r, int c) { return c < d0 ? g_e0 + r * d0 + c : d_all + r * dt + c; };
    float dp = 0.f, dn = 0.f, sq = 0.f;
    for (int c = lane; c < dt; c += kWave) {
      const float a = at(ru, c), b = at(rp, c), e = at(rn, c);
      dp += a * b;
      dn += a * e;
      sq += a * a + b * b + e * e;
    }
    float sig;
    loss_acc += neg_logsigmoid(wave_sum(dp) - wave_sum(dn), &sig);
    reg_acc += 0.5f * sq;
    const float dx = -sig * inv_batch;
    for (int c = lane; c < dt; c += kWave) {  // second pass over rows that are in cache by now
      const float a = at(ru, c), b = at(rp, c), e = at(rn, c);
      atomic_add_f32(grad_at(ru, c), dx * (b - e) + reg_coef * a);
      atomic_add_f32(grad_at(rp, c), dx * a + reg_coef * b);
      atomic_add_f32(grad_at(rn, c), -dx * a + reg_coef * e);
    }
  }
  // stats->loss = mf_loss + emb_loss (what train_single_batch returns); partial.y keeps the L2 part
  const float reg_w = wave_sum(reg_acc);
  publish_partials<kWavesPerBlock>(loss_acc * inv_batch + reg_coef * reg_w, lane == 0 ? reg_coef * reg_w : 0.f,
                                   0.f, 1.0f, scratch);
  if (stepper) step_store_advanced(stats, step_state);
}

// One wave per node row: backward of normalize, (+ the gradient arriving from the next hop), dropout and
// the two leaky-ReLUs:  d_sum = d_x * keep * scale * lrelu'(sum_pre), d_bi likewise with bi_pre.
__global__ __launch_bounds__(kBlock) void ngcf_act_bwd_kernel(
    float* __restrict__ d_all, const float* __restrict__ all, int ld_all, int off,
    const float* __restrict__ nrm, const float* __restrict__ d_next, const uint8_t* __restrict__ keep,
    float scale, const float* __restrict__ sum_pre, const float* __restrict__ bi_pre,
    float* __restrict__ d_sum, float* __restrict__ d_bi, int64_t n_rows, int d) {
  const int lane = lane_id();
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block(); r < n_rows;
       r += static_cast<int64_t>(gridDim.x) * kWavesPerBlock) {
    float dy[kNgcfMaxNpl], y[kNgcfMaxNpl];
    float proj = 0.f;
#pragma unroll
    for (int k = 0; k < kNgcfMaxNpl; ++k) {
      const int c = lane + kWave * k;
      dy[k] = c < d ? d_all[r * ld_all + off + c] : 0.f;
      y[k] = c < d ? all[r * ld_all + off + c] : 0.f;
      if (c < d) d_all[r * ld_all + off + c] = 0.f;  // read once per step: left zero for the next step's scatter
      proj += y[k] * dy[k];
    }
    proj = wave_sum(proj);
    const float n = nrm[r];
    const bool big = n >= kNormEps;  // clamp_min passes the norm's gradient only where norm >= eps
    const float inv = 1.0f / fmaxf(n, kNormEps);
#pragma unroll
    for (int k = 0; k < kNgcfMaxNpl; ++k) {
      const int c = lane + kWave * k;
      if (c < d) {
        const int64_t i = r * d + c;
        float dx = (big ? dy[k] - y[k] * proj : dy[k]) * inv;
        if (d_next) dx += d_next[i];
        if (keep) dx = keep[i] ? dx * scale : 0.f;
        d_sum[i] = sum_pre[i] > 0.f ? dx : dx * kSlope;
        d_bi[i] = bi_pre[i] > 0.f ? dx : dx * kSlope;
      }
    }
  }
}

// d_ego (=  or +=, hop 0 adds into the embedding gradient) d_bi_in * side ;  d_side += d_bi_in * ego
template <bool ACCUMULATE>
__global__ __launch_bounds__(kBlock) void ngcf_bi_bwd_kernel(const float* __restrict__ d_bi_in,
                                                             const float* __restrict__ side,
                                                             const float* __restrict__ ego,
                                                             float* __restrict__ d_ego,
                                                             float* __restrict__ d_side, int64_t n, int d,
                                                             SlicedOut spmm_src) {
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = tid; i < n; i += stride) {
    const float g = d_bi_in[i];
    if (ACCUMULATE) d_ego[i] += g * side[i]; else d_ego[i] = g * side[i];
    const float ds = d_side[i] + g * ego[i];
    d_side[i] = ds;
    if (spmm_src.xs != nullptr) spmm_src.put(i / d, static_cast<int>(i % d), ds);  // source of the transposed SpMM
  }
}

// scores[k] = <all[u], all[U + i]>
__global__ __launch_bounds__(kBlock) void ngcf_predict_kernel(const float* __restrict__ e0, int d0,
                                                              const float* __restrict__ all, int dt,
                                                              int64_t n_users, int64_t n_items,
                                                              const int64_t* __restrict__ users,
                                                              const int64_t* __restrict__ items, int64_t n,
                                                              float* __restrict__ scores,
                                                              hiprec_stats* stats) {
  const int lane = lane_id();
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block(); t < n;
       t += static_cast<int64_t>(gridDim.x) * kWavesPerBlock) {
    const int64_t u = users[t], i = items[t];
    const bool u_ok = static_cast<uint64_t>(u) < static_cast<uint64_t>(n_users);
    const bool i_ok = static_cast<uint64_t>(i) < static_cast<uint64_t>(n_items);
    if (!(u_ok && i_ok)) {
      if (lane == 0) {
        atomicOr(&stats->status,
                 (u_ok ? 0u : HIPREC_STATUS_USER_OOB) | (i_ok ? 0u : HIPREC_STATUS_ITEM_OOB));
        scores[t] = __builtin_nanf("");
      }
      continue;
    }
    float dot = 0.f;
    const int64_t ri = n_users + i;
    for (int c = lane; c < dt; c += kWave)
      dot += c < d0 ? e0[u * d0 + c] * e0[ri * d0 + c] : all[u * dt + c] * all[ri * dt + c];
    dot = wave_sum(dot);
    if (lane == 0) scores[t] = dot;
  }
}

inline int total_width(const hiprec_ngcf_plan* p) {
  int t = 0;
  for (int l = 0; l <= p->n_layers; ++l) t += p->dim[l];
  return t;
}

inline int check_ngcf_plan(const hiprec_ngcf_plan* p, bool train) {
  HIPREC_REQUIRE(p != nullptr, "NULL plan");
  HIPREC_REQUIRE(p->n_layers >= 1 && p->n_layers <= HIPREC_NGCF_MAX_LAYERS, "n_layers %d outside 1..%d",
                 p->n_layers, HIPREC_NGCF_MAX_LAYERS);
  HIPREC_REQUIRE(p->n_users > 0 && p->n_items > 0, "bad table sizes");
  HIPREC_REQUIRE(p->a.rowptr && p->a.nnz >= 0 && (p->a.nnz == 0 || (p->a.col && p->a.val)), "bad CSR a");
  HIPREC_REQUIRE(p->a.n_rows == p->n_users + p->n_items, "graph has %lld rows, tables %lld",
                 (long long)p->a.n_rows, (long long)(p->n_users + p->n_items));
  for (int l = 0; l <= p->n_layers; ++l)
    HIPREC_REQUIRE(p->dim[l] > 0 && p->dim[l] <= kNgcfMaxNpl * kWave, "hop width %d outside 1..%d",
                   p->dim[l], kNgcfMaxNpl * kWave);
  HIPREC_REQUIRE(p->e0 && p->all, "NULL e0 / all");
  if (p->zero_ws) {  // every buffer the SpMMs / the loss scatter into must lie inside the zero-once region
    const float* lo = p->zero_ws;
    const float* hi = p->zero_ws + p->zero_ws_floats;
    auto inside = [&](const float* q) { return q >= lo && q < hi; };
    for (int l = 0; l < p->n_layers; ++l)
      HIPREC_REQUIRE(inside(p->side[l]) && (!train || inside(p->spmm_tmp[l])),
                     "layer %d: side / spmm_tmp outside zero_ws", l);
    HIPREC_REQUIRE(!train || inside(p->d_all), "d_all outside zero_ws");
  }
  for (int l = 0; l < p->n_layers; ++l) {
    HIPREC_REQUIRE(p->gc_w[l] && p->gc_b[l] && p->bi_w[l] && p->bi_b[l], "NULL layer %d weights", l);
    HIPREC_REQUIRE(p->side[l] && p->bi_in[l] && p->sum_pre[l] && p->bi_pre[l] && p->ego[l] && p->nrm[l],
                   "NULL layer %d workspace", l);
  }
  if (train) {
    HIPREC_REQUIRE(p->at.rowptr && (p->at.nnz == 0 || (p->at.col && p->at.val)), "bad CSR at");
    HIPREC_REQUIRE(p->at.n_rows == p->a.n_rows && p->at.nnz == p->a.nnz, "transposed graph differs in shape");
    HIPREC_REQUIRE(p->g_e0 && p->d_all && p->d_sum && p->d_bi && p->d_side && p->d_bi_in && p->d_ego[0] &&
                       p->d_ego[1],
                   "NULL backward workspace");
    for (int l = 0; l < p->n_layers; ++l) HIPREC_REQUIRE(p->spmm_tmp[l], "NULL spmm_tmp[%d]", l);
    for (int l = 0; l < p->n_layers; ++l)
      HIPREC_REQUIRE(p->g_gc_w[l] && p->g_gc_b[l] && p->g_bi_w[l] && p->g_bi_b[l], "NULL layer %d gradients", l);
  }
  return 0;
}

// The column-sliced SpMM (spmm_sliced.hip) serves a hop when the plan carries the sliced graphs and every hop's input
// width takes the plan's slice width.  Its source is written in the sliced layout by whoever produces it (the
// activation kernel of the previous hop, the bilinear backward kernel; one transpose launch for e0) and its result
// goes straight to the row-major buffer the GEMMs read: no atomics, no zero fills of SpMM outputs.
inline bool ngcf_sliced(const hiprec_ngcf_plan* p) {
  if (p->sliced_src == nullptr || p->slice_w <= 0) return false;
  for (int l = 0; l < p->n_layers; ++l)
    if (sliced_width(p->n_users + p->n_items, p->dim[l]) != p->slice_w) return false;
  return true;
}

inline SlicedOut ngcf_sliced_out(const hiprec_ngcf_plan* p, const hiprec_sliced_csr* graph) {
  SlicedOut o;
  o.xs = p->sliced_src;
  o.col_scale = graph->col_scale;
  o.n_rows = p->n_users + p->n_items;
  o.w_shift = p->slice_w == 4 ? 2 : 1;
  return o;
}

// NGCF.forward: fills the per-hop workspaces and `all`.  keep bytes are used only when train.
inline int ngcf_forward(const hiprec_ngcf_plan* p, bool train, hipStream_t st) {
  const int64_t N = p->n_users + p->n_items;
  const int dt = total_width(p);
  const bool sliced = ngcf_sliced(p);
  const bool zeroed = p->zero_ws != nullptr;  // ONE fill for every SpMM output (+ d_all) of the step
  // sliced SpMMs write their outputs whole: nothing to clear (d_all, which only the loss scatters into, is zero on
  // entry -- the caller allocates the workspace zeroed -- and the backward's readers leave it zero again)
  if (zeroed && !sliced) HIPREC_TRY(hipMemsetAsync(p->zero_ws, 0, sizeof(float) * p->zero_ws_floats, st));
  const float* ego = p->e0;
  int off = p->dim[0];
  if (sliced) {
    if (int rc = launch_to_sliced(p->e0, N, p->dim[0], p->slice_w, p->sa.col_scale, p->sliced_src, nullptr, st))
      return rc;
  }
  for (int l = 0; l < p->n_layers; ++l) {
    const int di = p->dim[l], dout = p->dim[l + 1];
    if (sliced) {
      SlicedFlush fl;
      fl.final_out = p->side[l];
      fl.final_set = true;
      if (int rc = launch_spmm_sliced(&p->sa, nullptr, 1.0f, p->sliced_src, nullptr, nullptr, 0, di, p->slice_w, st, fl))
        return rc;
    } else if (int rc = launch_spmm(&p->a, nullptr, 1.0f, ego, p->side[l], nullptr, di, st, zeroed)) {
      return rc;
    }
    uint8_t* keep = train ? p->keep[l] : nullptr;
    const KeepGen gen{p->keep_gen, p->keep_prob[l], p->keep_seed * 64 + static_cast<uint64_t>(l), p->keep_step};
    const SlicedOut next_src = sliced && l + 1 < p->n_layers ? ngcf_sliced_out(p, &p->sa) : SlicedOut{};
#ifdef HIPREC_TEST_SWITCHES   // A/B switch of the test build only (libhiprec_test.so); the product reads no environment
    static const bool unfused_hop = getenv("HIPREC_NGCF_UNFUSED_HOP") != nullptr;
#else
    constexpr bool unfused_hop = false;
#endif
    if (!unfused_hop && di <= kHopMaxIn && di % 4 == 0 && dout <= kHopMaxOut && dout % 16 == 0) {
      // bi = ego * side, both Linear layers, activation, dropout, norm: one launch, 16 node rows per workgroup
      const size_t lds = sizeof(float) * hop_lds_floats(di, dout);
      static std::atomic<uint64_t> lds_ok{0};
      if (int rc = allow_dynamic_lds({reinterpret_cast<const void*>(&ngcf_hop_forward_kernel)},
                                     sizeof(float) * hop_lds_floats(kHopMaxIn, kHopMaxOut), lds_ok, "the fused NGCF hop"))
        return rc;
      ngcf_hop_forward_kernel<<<static_cast<int>((N + kHopRows - 1) / kHopRows), kHopThreads, lds, st>>>(
          p->side[l], ego, p->gc_w[l], p->gc_b[l], p->bi_w[l], p->bi_b[l], di, dout, N, p->bi_in[l], p->sum_pre[l],
          p->bi_pre[l], keep, p->keep_scale[l], gen, p->ego[l], p->nrm[l], p->all, dt, off, next_src);
      HIPREC_TRY(hipGetLastError());
    } else {
      ngcf_bi_mul_kernel<<<grid_for_threads((N * di + 3) / 4), kBlock, 0, st>>>(ego, p->side[l], p->bi_in[l],
                                                                               N * di);
      HIPREC_TRY(hipGetLastError());
      GemmGroup g{};
      g.n = 2;
      g.p[0] = make_gemm(kNT, static_cast<int>(N), dout, di, p->side[l], di, p->gc_w[l], di, p->sum_pre[l], dout,
                         p->gc_b[l], 0, nullptr, 0, false);
      g.p[1] = make_gemm(kNT, static_cast<int>(N), dout, di, p->bi_in[l], di, p->bi_w[l], di, p->bi_pre[l], dout,
                         p->bi_b[l], 0, nullptr, 0, false);
      if (int rc = launch_group(g, st)) return rc;
      ngcf_act_kernel<<<grid_for_waves(N), kBlock, 0, st>>>(p->sum_pre[l], p->bi_pre[l], keep, p->keep_scale[l],
                                                           gen, p->ego[l], p->nrm[l], p->all, dt, off, N, dout,
                                                           next_src);
      HIPREC_TRY(hipGetLastError());
    }
    ego = p->ego[l];
    off += dout;
  }
  return 0;
}

}  // namespace hiprec

using namespace hiprec;

extern "C" size_t hiprec_ngcf_plan_bytes(void) { return sizeof(hiprec_ngcf_plan); }

extern "C" int hiprec_ngcf_forward(const hiprec_ngcf_plan* plan, int train, void* stream) {
  if (int rc = check_ngcf_plan(plan, false)) return rc;
  HIPREC_REQUIRE(plan->n_users + plan->n_items < (1ll << 31), "too many nodes for the 32-bit GEMM extents");
  return ngcf_forward(plan, train != 0, static_cast<hipStream_t>(stream));
}

extern "C" int hiprec_ngcf_predict(const hiprec_ngcf_plan* plan, const int64_t* users, const int64_t* items,
                                   int64_t n, float* scores, hiprec_stats* stats, void* stream) {
  if (int rc = check_ngcf_plan(plan, false)) return rc;
  HIPREC_REQUIRE(n >= 0, "negative n");
  if (n == 0) return 0;
  HIPREC_REQUIRE(users && items && scores && stats, "NULL pointer");
  ngcf_predict_kernel<<<grid_for_waves(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      plan->e0, plan->dim[0], plan->all, total_width(plan), plan->n_users, plan->n_items, users, items, n, scores,
      stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

static int ngcf_grad_impl(const hiprec_ngcf_plan* plan, const int64_t* users, const int64_t* pos, const int64_t* neg,
                          int64_t batch, float inv_batch, hiprec_stats* stats, void* scratch, size_t scratch_bytes,
                          void* stream);

extern "C" int hiprec_ngcf_grad(const hiprec_ngcf_plan* plan, const int64_t* users, const int64_t* pos,
                                const int64_t* neg, int64_t batch, float inv_batch, hiprec_stats* stats,
                                void* scratch, size_t scratch_bytes, void* stream) {
  if (int rc = check_ngcf_plan(plan, true)) return rc;   // (a plan that does not check out is not touched at all)
  const int rc = ngcf_grad_impl(plan, users, pos, neg, batch, inv_batch, stats, scratch, scratch_bytes, stream);
  // On the sliced path nothing clears d_all per step: the backward's readers leave it zero for the next step's loss
  // scatter.  A step that failed between the scatter and those readers would leak its gradient rows into the next
  // one: put the contract back before reporting the error.
  if (rc != 0)
    (void)hipMemsetAsync(plan->d_all, 0, sizeof(float) * (plan->n_users + plan->n_items) * total_width(plan),
                         static_cast<hipStream_t>(stream));
  return rc;
}

static int ngcf_grad_impl(const hiprec_ngcf_plan* plan, const int64_t* users, const int64_t* pos, const int64_t* neg,
                          int64_t batch, float inv_batch, hiprec_stats* stats, void* scratch, size_t scratch_bytes,
                          void* stream) {
  if (int rc = check_ngcf_plan(plan, true)) return rc;
  const hiprec_ngcf_plan* p = plan;
  HIPREC_REQUIRE(p->n_users + p->n_items < (1ll << 31), "too many nodes for the 32-bit GEMM extents");
  HIPREC_REQUIRE(stats && scratch, "NULL stats/scratch");
  HIPREC_REQUIRE(batch >= 0, "negative batch");
  HIPREC_REQUIRE(batch == 0 || (users && pos && neg), "NULL index arrays");
  if (scratch_bytes < kScratchBytes) {
    set_error("scratch %zu B < %zu B", scratch_bytes, kScratchBytes);
    return HIPREC_E_SCRATCH;
  }
  auto st = static_cast<hipStream_t>(stream);
  const int64_t N = p->n_users + p->n_items;
  const int dt = total_width(p);
  if (int rc = ngcf_forward(p, true, st)) return rc;
  const bool sliced = ngcf_sliced(p);
  if (!p->zero_ws) HIPREC_TRY(hipMemsetAsync(p->d_all, 0, sizeof(float) * N * dt, st));
  ngcf_loss_kernel<<<grid_for_waves(batch > 0 ? batch : 1), kBlock, 0, st>>>(
      p->e0, p->g_e0, p->dim[0], p->all, p->d_all, dt, p->n_users, p->n_items, users, pos, neg, batch, inv_batch,
      p->decay * p->inv_reg_batch, stats, static_cast<Scratch*>(scratch));
  HIPREC_TRY(hipGetLastError());

  int off = dt;
  const float* d_next = nullptr;
  // per-hop d_sum / d_bi (optional workspaces) let the hop's backward chain ride one launch and ALL weight / bias
  // gradients one grouped launch at the end
#ifdef HIPREC_TEST_SWITCHES
  static const bool unfused_bwd = getenv("HIPREC_NGCF_UNFUSED_HOP") != nullptr;
#else
  constexpr bool unfused_bwd = false;
#endif
  bool hop_bwd = !unfused_bwd && 4 * p->n_layers <= kMaxGroup;
  for (int l = 0; l < p->n_layers; ++l)
    hop_bwd = hop_bwd && p->d_sum_l[l] && p->d_bi_l[l] && p->dim[l] <= kHopBwdMax && p->dim[l] % 16 == 0 &&
              p->dim[l + 1] <= kHopBwdMax && p->dim[l + 1] % 4 == 0;
  const int n32 = static_cast<int>(N);
  for (int l = p->n_layers - 1; l >= 0; --l) {
    const int di = p->dim[l], dout = p->dim[l + 1];
    off -= dout;
    const float* ego_in = l == 0 ? p->e0 : p->ego[l - 1];
    // hop 0 hands its result to the embedding gradient itself (which already holds the loss's share)
    float* d_ego = l == 0 ? p->g_e0 : p->d_ego[l & 1];
    const SlicedOut src = sliced ? ngcf_sliced_out(p, &p->sat) : SlicedOut{};
    if (hop_bwd) {
      const size_t lds = sizeof(float) * hop_bwd_lds_floats(di, dout);
      static std::atomic<uint64_t> lds_ok{0};
      if (int rc = allow_dynamic_lds({reinterpret_cast<const void*>(&ngcf_hop_backward_kernel<true>),
                                      reinterpret_cast<const void*>(&ngcf_hop_backward_kernel<false>)},
                                     sizeof(float) * hop_bwd_lds_floats(kHopBwdMax, kHopBwdMax), lds_ok,
                                     "the fused NGCF hop (backward)"))
        return rc;
      const int grid = static_cast<int>((N + kHopRows - 1) / kHopRows);
      if (l == 0)
        ngcf_hop_backward_kernel<true><<<grid, kHopThreads, lds, st>>>(
            p->d_all, p->all, dt, off, p->nrm[l], d_next, p->keep[l], p->keep_scale[l], p->sum_pre[l], p->bi_pre[l],
            p->gc_w[l], p->bi_w[l], p->side[l], ego_in, di, dout, N, p->d_sum_l[l], p->d_bi_l[l], d_ego, p->d_side, src);
      else
        ngcf_hop_backward_kernel<false><<<grid, kHopThreads, lds, st>>>(
            p->d_all, p->all, dt, off, p->nrm[l], d_next, p->keep[l], p->keep_scale[l], p->sum_pre[l], p->bi_pre[l],
            p->gc_w[l], p->bi_w[l], p->side[l], ego_in, di, dout, N, p->d_sum_l[l], p->d_bi_l[l], d_ego, p->d_side, src);
      HIPREC_TRY(hipGetLastError());
    } else {
      ngcf_act_bwd_kernel<<<grid_for_waves(N), kBlock, 0, st>>>(
          p->d_all, p->all, dt, off, p->nrm[l], d_next, p->keep[l], p->keep_scale[l], p->sum_pre[l],
          p->bi_pre[l], p->d_sum, p->d_bi, N, dout);
      HIPREC_TRY(hipGetLastError());
      GemmGroup g{};
      g.n = 6;
      g.p[0] = make_gemm(kNN, n32, di, dout, p->d_sum, dout, p->gc_w[l], di, p->d_side, di, nullptr, 0, nullptr, 0,
                         false);
      g.p[1] = make_gemm(kNN, n32, di, dout, p->d_bi, dout, p->bi_w[l], di, p->d_bi_in, di, nullptr, 0, nullptr, 0,
                         false);
      g.p[2] = make_gemm(kTNm, dout, di, n32, p->d_sum, dout, p->side[l], di, p->g_gc_w[l], di, nullptr, 0, nullptr,
                         0, true);
      g.p[3] = make_gemm(kTNm, dout, di, n32, p->d_bi, dout, p->bi_in[l], di, p->g_bi_w[l], di, nullptr, 0, nullptr,
                         0, true);
      // bias gradients: cancelling column sums over all N nodes -- two levels in a fixed order (run-order atomics made
      // them differ from run to run by more than the reference's own fp32 error).  Workspace: the other d_ego
      // buffer, which the activation backward above has just consumed (d_next) and nobody writes before the SpMM
      float* ws = p->d_ego[(l + 1) & 1];
      const bool ws_fits = 2 * colsum_ws_floats(n32, dout) <= N * static_cast<int64_t>(di > dout ? di : dout);
      g.p[4] = make_colsum(p->d_sum, n32, dout, dout, p->g_gc_b[l], ws_fits ? ws : nullptr);
      g.p[5] = make_colsum(p->d_bi, n32, dout, dout, p->g_bi_b[l], ws_fits ? ws + colsum_ws_floats(n32, dout) : nullptr);
      if (int rc = launch_group(g, st)) return rc;
      if (int rc = launch_colsum_reduce(g, st)) return rc;
      if (l == 0)
        ngcf_bi_bwd_kernel<true><<<grid_for_threads(N * di), kBlock, 0, st>>>(p->d_bi_in, p->side[l], ego_in, d_ego,
                                                                             p->d_side, N * di, di, src);
      else
        ngcf_bi_bwd_kernel<false><<<grid_for_threads(N * di), kBlock, 0, st>>>(p->d_bi_in, p->side[l], ego_in, d_ego,
                                                                              p->d_side, N * di, di, src);
      HIPREC_TRY(hipGetLastError());
    }
    // d_ego += A^T d_side
    if (sliced) {
      SlicedFlush fl;
      fl.final_out = d_ego;
      if (int rc = launch_spmm_sliced(&p->sat, nullptr, 1.0f, p->sliced_src, nullptr, nullptr, 0, di, p->slice_w, st, fl))
        return rc;
    } else if (int rc = launch_spmm(&p->at, nullptr, 1.0f, p->d_side, p->spmm_tmp[l], d_ego, di, st,
                                    p->zero_ws != nullptr)) {
      return rc;
    }
    d_next = d_ego;
  }
  if (hop_bwd) {  // every hop's weight and bias gradients: they only read what the hop launches wrote
    // bias gradients in two levels with a fixed order (see the per-hop form above); workspace: d_bi_in, which only
    // the per-hop form uses
    GemmGroup g{};
    g.n = 0;
    int64_t ws_need = 0, ws_have = 0;
    for (int l = 0; l < p->n_layers; ++l) {
      ws_need += 2 * colsum_ws_floats(n32, p->dim[l + 1]);
      ws_have = std::max<int64_t>(ws_have, N * static_cast<int64_t>(std::max(p->dim[l], p->dim[l + 1])));
    }
    float* ws = p->d_bi_in && ws_need <= ws_have ? p->d_bi_in : nullptr;
    for (int l = 0; l < p->n_layers; ++l) {
      const int di = p->dim[l], dout = p->dim[l + 1];
      g.p[g.n++] = make_gemm(kTNm, dout, di, n32, p->d_sum_l[l], dout, p->side[l], di, p->g_gc_w[l], di, nullptr, 0,
                             nullptr, 0, true);
      g.p[g.n++] = make_gemm(kTNm, dout, di, n32, p->d_bi_l[l], dout, p->bi_in[l], di, p->g_bi_w[l], di, nullptr, 0,
                             nullptr, 0, true);
      g.p[g.n++] = make_colsum(p->d_sum_l[l], n32, dout, dout, p->g_gc_b[l], ws);
      if (ws) ws += colsum_ws_floats(n32, dout);
      g.p[g.n++] = make_colsum(p->d_bi_l[l], n32, dout, dout, p->g_bi_b[l], ws);
      if (ws) ws += colsum_ws_floats(n32, dout);
    }
    if (int rc = launch_group(g, st)) return rc;
    if (int rc = launch_colsum_reduce(g, st)) return rc;
  }
  return 0;
}
