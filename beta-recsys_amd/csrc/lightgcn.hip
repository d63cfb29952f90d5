// LightGCN training step for gfx950.
//
// Replaces beta_rec/models/lightgcn.py:46-78 (LightGCN.forward: edge dropout + n_layers x
// torch.sparse.mm + mean over layers), :119-152 / :171-191 (gather of the propagated rows, BPR
// softplus loss, L2 on the layer-0 rows) and the autograd backward (n_layers x sparse.mm with the
// transposed graph, dense gradients).
//
// SpMM: the graph is the reference's D^-1 (A + I) in CSR (int64 row pointers, int32 columns, fp32
// values); item degrees follow a popularity law (a handful of rows own thousands of edges), so the
// work is split by EDGES, not rows: every wavefront takes a contiguous slice of kEdgesPerWave edges,
// finds its first row by binary search, keeps the running row sum in registers (one lane per
// embedding column) and flushes at row boundaries with fp32 atomics.  64 (col, val) pairs are
// fetched with ONE coalesced vector load and broadcast lane by lane with v_readlane.  The flush goes
// to the layer output Y and, fused, to the running layer sum ACC (the mean over layers / the
// accumulated gradient), so no separate add kernel exists.  Edge dropout is a per-edge keep byte
// (drawn like the reference does, or by hiprec_edge_dropout_mask) applied on the fly; the
// transposed graph looks its edges up through `eid` so both directions drop the same edges.
#include "common.hpp"
#include "spmm.hpp"

namespace hiprec {

constexpr int kEdgesPerWave = 256;

__device__ __forceinline__ int64_t find_row(const int64_t* __restrict__ rowptr, int64_t n_rows,
                                            int64_t e) {
  // largest r with rowptr[r] <= e
  int64_t lo = 0, hi = n_rows;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// One thread per kEdgesPerWave-edge slice: the row its first edge belongs to (done once per graph, so
// that a wave of the SpMM does not start with a 14-step dependent binary search through L2).
__global__ __launch_bounds__(kBlock) void csr_slice_rows_kernel(const int64_t* __restrict__ rowptr,
                                                                int64_t n_rows, int64_t nnz,
                                                                int32_t* __restrict__ out,
                                                                int64_t n_slices) {
  const int64_t s = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (s < n_slices) out[s] = static_cast<int32_t>(find_row(rowptr, n_rows, min(s * kEdgesPerWave, nnz - 1)));
}

// NPL = columns per lane held in registers (dim <= 64 * NPL).
// A wave owns one slice of kEdgesPerWave consecutive edges and ALL waves of the launch are resident
// at once (~23 per CU at ML-1M size), so the launch lasts as long as ONE wave's chain of dependent
// memory round trips (~1 us each through L2 under load): the chain is kept short --
//   * the slice's first row comes from a.slice_row (one load) instead of a binary search (14 loads),
//   * the (col, val, keep) lanes of the NEXT 64-edge chunk are requested before the current chunk's
//     row gathers, so they never cost a round trip of their own,
//   * kGroup = 32 / NPL source rows are requested together and unconditionally (lanes past the slice
//     hold column 0 / value 0, dropped edges multiply by 0); row bookkeeping is scalar and runs after
//     the loads are issued.
template <int NPL>
__global__ __launch_bounds__(kBlock) void spmm_csr_kernel(hiprec_csr a,
                                                          const uint8_t* __restrict__ keep,
                                                          float scale, const float* __restrict__ x,
                                                          float* __restrict__ y,
                                                          float* __restrict__ acc_out, int dim) {
  const int lane = lane_id();
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  const int64_t e_begin = wave * kEdgesPerWave;
  if (e_begin >= a.nnz) return;
  const int64_t e_end = min(a.nnz, e_begin + kEdgesPerWave);

  auto load_chunk = [&](int64_t e0, int& col, float& val) {
    const int64_t e = e0 + lane;
    col = 0;
    val = 0.f;
    if (e < e_end) {
      col = a.col[e];
      val = a.val[e];
      if (keep) val = keep[a.eid ? a.eid[e] : e] ? val * scale : 0.f;
    }
  };
  int my_col, nxt_col = 0;
  float my_val, nxt_val = 0.f;
  load_chunk(e_begin, my_col, my_val);
  int64_t row = a.slice_row ? static_cast<int64_t>(a.slice_row[wave]) : find_row(a.rowptr, a.n_rows, e_begin);
  int64_t row_end = a.rowptr[row + 1];
  float acc[NPL];
#pragma unroll
  for (int k = 0; k < NPL; ++k) acc[k] = 0.f;

  auto flush = [&](int64_t r) {
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int c = lane + kWave * k;
      if (c < dim && acc[k] != 0.f) {
        atomic_add_f32(y + r * dim + c, acc[k]);
        if (acc_out) atomic_add_f32(acc_out + r * dim + c, acc[k]);
      }
      acc[k] = 0.f;
    }
  };

  for (int64_t e0 = e_begin; e0 < e_end; e0 += kWave) {
    if (e0 + kWave < e_end) load_chunk(e0 + kWave, nxt_col, nxt_val);
    const int n_here = static_cast<int>(min<int64_t>(kWave, e_end - e0));
    constexpr int kGroup = 32 / NPL;
    for (int j = 0; j < n_here; j += kGroup) {
      float xv[kGroup][NPL];
#pragma unroll
      for (int q = 0; q < kGroup; ++q) {
        const float* xr = x + static_cast<int64_t>(__builtin_amdgcn_readlane(my_col, j + q)) * dim;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
          const int c = lane + kWave * k;
          xv[q][k] = xr[c < dim ? c : dim - 1];
        }
      }
#pragma unroll
      for (int q = 0; q < kGroup; ++q) {
        if (j + q < n_here) {
          while (e0 + j + q >= row_end) {  // row boundary (empty rows are skipped)
            flush(row);
            ++row;
            row_end = a.rowptr[row + 1];
          }
          const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_val), j + q));
#pragma unroll
          for (int k = 0; k < NPL; ++k)
            if (lane + kWave * k < dim) acc[k] += v * xv[q][k];
        }
      }
    }
    my_col = nxt_col;
    my_val = nxt_val;
  }
  flush(row);
}

// keep[e] = (uniform(seed, step, e) < keep_prob): counter-based (splitmix64 finaliser), no state
__global__ __launch_bounds__(kBlock) void edge_dropout_kernel(uint8_t* __restrict__ keep,
                                                              int64_t nnz, float keep_prob,
                                                              uint64_t seed, uint64_t step) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < nnz; e += stride) {
    keep[e] = keep_draw(seed, step, e, keep_prob) ? 1 : 0;
  }
}

// BPR softplus loss on the propagated rows + L2 on the layer-0 rows (lightgcn.py:171-191) and
// the gradient w.r.t. the propagated embeddings.  One wave per triple.  `acc` holds the SUM over
// layers (out = acc / (L+1)).  d_out contributions go to `d` (input of the backward propagation)
// AND to `g` (its l = 0 term); the L2 gradient goes to `g` only.
// SLICED: the propagated sums are read from, and d_out is scattered into, the sliced layout [D / W][N][W] of the
// column-sliced SpMM (p.acc / p.da then point at sliced buffers): no transposes around the loss.
// NPL: columns per lane (dim <= 64 * NPL).  The nine rows of a triple (three propagated sums, three layer-0 rows, and
// for the transposed graph's column factors three scalars) are requested together and stay in registers: the gradient
// half of the wave's work starts from them, not from a second round trip (round 5: 9.0 us for 1024 triples, the second
// loop re-read all six rows because the atomics in between may alias them).
template <bool SLICED, int NPL>
__global__ __launch_bounds__(kBlock) void lightgcn_loss_kernel(
    hiprec_lightgcn_plan p, const int64_t* __restrict__ users, const int64_t* __restrict__ pos,
    const int64_t* __restrict__ neg, int64_t batch, float inv_batch, hiprec_stats* stats,
    Scratch* scratch) {
  const int lane = lane_id();
  const int D = p.dim;
  const int w_shift = p.slice_w == 4 ? 2 : 1;
  const int64_t n_rows = p.a.n_rows;
  // offset of (node row, column c) in the layout of acc / da (no division: the node index comes with the call)
  auto at = [&](int64_t row, int c) -> int64_t {
    if (!SLICED) return row * D + c;
    return ((static_cast<int64_t>(c >> w_shift) * n_rows + row) << w_shift) + (c & ((1 << w_shift) - 1));
  };
  const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const float inv_l = 1.0f / static_cast<float>(p.n_layers + 1);
  const float* __restrict__ acc = p.acc;
  const float* __restrict__ e0 = p.e0;
  // (sliced, factored transposed graph: its passes take their source scaled by the column factor)
  const float* __restrict__ cs = SLICED ? p.sat.col_scale : nullptr;
  float loss_acc = 0.f, reg_acc = 0.f;
  const bool stepper = blockIdx.x == 0 && threadIdx.x == 0;
  StepState ss{};
  if (stepper) ss = step_load(stats);
  for (int64_t t = wave0; t < batch; t += n_waves) {
    const int64_t u = users[t], i = pos[t], j = neg[t];
    const bool u_ok = static_cast<uint64_t>(u) < static_cast<uint64_t>(p.n_users);
    const bool i_ok = static_cast<uint64_t>(i) < static_cast<uint64_t>(p.n_items) &&
                      static_cast<uint64_t>(j) < static_cast<uint64_t>(p.n_items);
    if (!(u_ok && i_ok)) {
      if (lane == 0)
        atomicOr(&stats->status,
                 (u_ok ? 0u : HIPREC_STATUS_USER_OOB) | (i_ok ? 0u : HIPREC_STATUS_ITEM_OOB));
      continue;
    }
    const int64_t nu = u, np_ = p.n_users + i, nn_ = p.n_users + j;  // node rows
    const int64_t ru = nu * D, rp = np_ * D, rn = nn_ * D;
    float ue[NPL], pe[NPL], ne[NPL], u0[NPL], p0[NPL], n0[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int c = lane + kWave * k;
      const bool in = c < D;
      const int cc = in ? c : 0;
      ue[k] = in ? acc[at(nu, cc)] : 0.f;
      pe[k] = in ? acc[at(np_, cc)] : 0.f;
      ne[k] = in ? acc[at(nn_, cc)] : 0.f;
      u0[k] = in ? e0[ru + cc] : 0.f;
      p0[k] = in ? e0[rp + cc] : 0.f;
      n0[k] = in ? e0[rn + cc] : 0.f;
    }
    const float su = cs ? cs[nu] : 1.f, sp = cs ? cs[np_] : 1.f, sn = cs ? cs[nn_] : 1.f;
    float dp = 0.f, dn = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      ue[k] *= inv_l;
      pe[k] *= inv_l;
      ne[k] *= inv_l;
      dp += ue[k] * pe[k];
      dn += ue[k] * ne[k];
      reg_acc += u0[k] * u0[k] + p0[k] * p0[k] + n0[k] * n0[k];
    }
    dp = wave_sum(dp);
    dn = wave_sum(dn);
    const float xx = dn - dp;  // softplus(neg - pos)
    loss_acc += xx > 20.f ? xx : log1pf(expf(xx));
    const float dx = sigmoid_f32(xx) * inv_batch * inv_l;  // d mf / d x, pre-scaled by 1/(L+1)
    const float cr = p.decay * inv_batch;                   // d reg / d row = decay * row / B
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int c = lane + kWave * k;
      if (c < D) {
        const float gu = dx * (ne[k] - pe[k]), gp = -dx * ue[k], gn = dx * ue[k];
        atomic_add_f32(p.da + at(nu, c), gu * su);
        atomic_add_f32(p.da + at(np_, c), gp * sp);
        atomic_add_f32(p.da + at(nn_, c), gn * sn);
        atomic_add_f32(p.g + ru + c, gu + cr * u0[k]);
        atomic_add_f32(p.g + rp + c, gp + cr * p0[k]);
        atomic_add_f32(p.g + rn + c, gn + cr * n0[k]);
      }
    }
  }
  if (stepper) step_store_advanced(stats, ss);
  // loss = mean softplus + decay * 0.5 * sum(...) / B ; published as ONE number (loss slot)
  const float reg_w = wave_sum(reg_acc);
  publish_partials<kWavesPerBlock>(loss_acc + 0.5f * p.decay * reg_w, 0.f, 0.f, inv_batch, scratch);
}

template <bool SLICED>
static int launch_lightgcn_loss(const hiprec_lightgcn_plan& q, const int64_t* users, const int64_t* pos,
                                const int64_t* neg, int64_t batch, float inv_batch, hiprec_stats* stats,
                                Scratch* scratch, hipStream_t st) {
  const int grid = grid_for_waves(batch);
  if (q.dim <= 64)
    lightgcn_loss_kernel<SLICED, 1><<<grid, kBlock, 0, st>>>(q, users, pos, neg, batch, inv_batch, stats, scratch);
  else if (q.dim <= 128)
    lightgcn_loss_kernel<SLICED, 2><<<grid, kBlock, 0, st>>>(q, users, pos, neg, batch, inv_batch, stats, scratch);
  else if (q.dim <= 256)
    lightgcn_loss_kernel<SLICED, 4><<<grid, kBlock, 0, st>>>(q, users, pos, neg, batch, inv_batch, stats, scratch);
  else {
    set_error("LightGCN embedding dim %d > 256 is not supported", q.dim);
    return HIPREC_E_UNSUPPORTED;
  }
  HIPREC_TRY(hipGetLastError());
  return 0;
}

// scores[k] = sigmoid(<out[u], out[U + i]>) with out = acc / (L+1)   (LightGCN.predict :98-100)
__global__ __launch_bounds__(kBlock) void lightgcn_predict_kernel(hiprec_lightgcn_plan p,
                                                                  const int64_t* __restrict__ users,
                                                                  const int64_t* __restrict__ items,
                                                                  int64_t n, float* __restrict__ scores,
                                                                  hiprec_stats* stats) {
  const int lane = lane_id();
  const int D = p.dim;
  const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const float inv_l = 1.0f / static_cast<float>(p.n_layers + 1);
  for (int64_t t = wave0; t < n; t += n_waves) {
    const int64_t u = users[t], i = items[t];
    if (static_cast<uint64_t>(u) >= static_cast<uint64_t>(p.n_users) ||
        static_cast<uint64_t>(i) >= static_cast<uint64_t>(p.n_items)) {
      if (lane == 0) {
        atomicOr(&stats->status, HIPREC_STATUS_ROW_OOB);
        scores[t] = __builtin_nanf("");
      }
      continue;
    }
    float dot = 0.f;
    for (int c = lane; c < D; c += kWave)
      dot += (p.acc[u * D + c] * inv_l) * (p.acc[(p.n_users + i) * D + c] * inv_l);
    dot = wave_sum(dot);
    if (lane == 0) scores[t] = sigmoid_f32(dot);
  }
}

int launch_spmm(const hiprec_csr* a, const uint8_t* keep, float scale, const float* x, float* y,
                float* acc, int dim, hipStream_t st, bool y_is_zero) {
  if (!y_is_zero) HIPREC_TRY(hipMemsetAsync(y, 0, sizeof(float) * a->n_rows * dim, st));
  if (a->nnz == 0) return 0;
  const int64_t n_waves = (a->nnz + kEdgesPerWave - 1) / kEdgesPerWave;
  const int64_t blocks = (n_waves + kWavesPerBlock - 1) / kWavesPerBlock;
  HIPREC_REQUIRE(blocks < (1ll << 31), "graph too large for one launch");
  const int grid = static_cast<int>(blocks);
  if (dim <= 64) spmm_csr_kernel<1><<<grid, kBlock, 0, st>>>(*a, keep, scale, x, y, acc, dim);
  else if (dim <= 128) spmm_csr_kernel<2><<<grid, kBlock, 0, st>>>(*a, keep, scale, x, y, acc, dim);
  else if (dim <= 256) spmm_csr_kernel<4><<<grid, kBlock, 0, st>>>(*a, keep, scale, x, y, acc, dim);
  else {
    set_error("LightGCN embedding dim %d > 256 is not supported", dim);
    return HIPREC_E_UNSUPPORTED;
  }
  HIPREC_TRY(hipGetLastError());
  return 0;
}

static int check_csr(const hiprec_csr* a, const char* name) {
  HIPREC_REQUIRE(a && a->rowptr && a->n_rows > 0 && a->nnz >= 0, "bad CSR %s", name);
  HIPREC_REQUIRE(a->nnz == 0 || (a->col && a->val), "CSR %s has NULL col/val", name);
  return 0;
}

static int check_lg_plan(const hiprec_lightgcn_plan* p, bool train) {
  HIPREC_REQUIRE(p != nullptr, "NULL plan");
  if (int rc = check_csr(&p->a, "a")) return rc;
  HIPREC_REQUIRE(p->n_users > 0 && p->n_items > 0 && p->dim > 0 && p->n_layers >= 0, "bad sizes");
  HIPREC_REQUIRE(p->a.n_rows == p->n_users + p->n_items, "graph size != n_users + n_items");
  const bool ws = p->zero_ws != nullptr;
  if (ws)
    HIPREC_REQUIRE(p->zero_ws_floats >= (1 + 2 * static_cast<int64_t>(p->n_layers)) * p->a.n_rows * p->dim,
                   "zero_ws holds %lld floats, (1 + 2*n_layers) * n_rows * dim are needed",
                   (long long)p->zero_ws_floats);
  HIPREC_REQUIRE(p->e0 && p->acc && (ws || (p->xa && p->xb)), "NULL forward buffers");
  if (p->sliced_ws) {
    HIPREC_REQUIRE(p->slice_w == sliced_width(p->a.n_rows, p->dim) && p->slice_w > 0,
                   "slice_w %d is not the sliced SpMM's width for %lld rows x dim %d", p->slice_w,
                   (long long)p->a.n_rows, p->dim);
    HIPREC_REQUIRE(p->sliced_ws_floats >= 4 * p->a.n_rows * p->dim + p->sa.n_slots + p->sat.n_slots,
                   "sliced_ws holds %lld floats, 4 * n_rows * dim + sa.n_slots + sat.n_slots needed",
                   (long long)p->sliced_ws_floats);
    HIPREC_REQUIRE(p->sa.n_rows == p->a.n_rows && p->sa.n_slots >= p->a.nnz && p->sa.eid &&
                       (!train || (p->sat.n_rows == p->a.n_rows && p->sat.n_slots >= p->a.nnz && p->sat.eid)),
                   "sliced graphs do not match the CSRs");
  }
  if (train) {
    if (int rc = check_csr(&p->at, "at")) return rc;
    HIPREC_REQUIRE(p->at.n_rows == p->a.n_rows && p->at.nnz == p->a.nnz, "a / at mismatch");
    HIPREC_REQUIRE(p->g && (ws || (p->da && p->db)), "NULL backward buffers");
  }
  return 0;
}

// slice k of the contiguous zero-once workspace: 0 = d_out, 1..L = forward layer outputs,
// L+1..2L = backward layer outputs
static float* ws_slice(const hiprec_lightgcn_plan* p, int k) {
  return p->zero_ws + static_cast<int64_t>(k) * p->a.n_rows * p->dim;
}

// The sliced path (csrc/spmm_sliced.hip): four buffers of n_rows * dim floats in plan->sliced_ws --
// [0] the transposed input, [1] the running layer sum, [2], [3] ping-pong layer outputs -- then the dropped edge
// values of the step: sa.n_slots floats for the forward graph, sat.n_slots for its transpose.
static bool use_sliced(const hiprec_lightgcn_plan* p) { return p->sliced_ws != nullptr && p->slice_w > 0; }

static float* sliced_buf(const hiprec_lightgcn_plan* p, int k) {
  return p->sliced_ws + static_cast<int64_t>(k) * p->a.n_rows * p->dim;
}

// out (row-major) = [add] sum_{l = first .. L} G^l in, G = `graph`; the l = 0 term only when `with_input`.
// in NULL: the input is already in sliced buffer 0 (scaled by graph->col_scale if the graph is factored);
// out NULL: the result stays in sliced buffer 1.  clear_buf0: sliced buffer 0 is all zeros afterwards (the last pass
// clears it on the way when it is not that pass's own source).
static int propagate_sliced(const hiprec_lightgcn_plan* p, const hiprec_sliced_csr* graph, const uint8_t* keep,
                            float keep_prob, const float* in, float* out, bool with_input, bool add,
                            hipStream_t st, bool clear_buf0 = false) {
  const int64_t N = p->a.n_rows;
  const int D = p->dim, W = p->slice_w;
  const int L = p->n_layers;
  const float scale = keep ? 1.0f / keep_prob : 1.0f;
  float* xs0 = sliced_buf(p, 0);
  float* accs = sliced_buf(p, 1);
  // (a prepared step -- hiprec_lightgcn_step_values -- already laid E0 out for the forward graph's first pass)
  const bool prepared = p->dropped_ready && keep && in == p->e0 && graph == &p->sa && with_input;
  if (in != nullptr && !prepared) {
    if (int rc = launch_to_sliced(in, N, D, W, graph->col_scale, xs0, with_input ? accs : nullptr, st)) return rc;
  }
  const void* val = nullptr;
  if (keep && L > 0) {  // once per step and graph, not per pass
    float* dropped = sliced_buf(p, 4) + (graph == &p->sat ? p->sa.n_slots : 0);
    if (!p->dropped_ready) {
      if (int rc = launch_step_values(graph, nullptr, const_cast<uint8_t*>(keep), false, keep_prob, 0, 0, dropped,
                                      nullptr, st))
        return rc;
    }
    val = dropped;
  }
  // the last pass adds the finished layer sum to a row-major `out` itself (no transpose launch) when `out` is added to
  const bool direct_out = out != nullptr && add && L > 0;
  const float* cur = xs0;
  for (int l = 0; l < L; ++l) {
    float* nxt = sliced_buf(p, 2 + (l & 1));
    const int mode = (l == 0 && !with_input) ? 2 : 1;
    const bool last = l == L - 1;
    SlicedFlush fl;
    fl.zero_out = last && clear_buf0 && l > 0 ? xs0 : nullptr;
    fl.final_out = last && direct_out ? out : nullptr;
    if (int rc = launch_spmm_sliced(graph, val, scale, cur, nxt, accs, mode, D, W, st, fl)) return rc;
    cur = nxt;
  }
  if (clear_buf0 && L < 2) HIPREC_TRY(hipMemsetAsync(xs0, 0, sizeof(float) * N * D, st));
  if (out == nullptr || direct_out || (L == 0 && !with_input)) return 0;
  return launch_from_sliced(accs, N, D, W, out, add, st);
}

// acc = sum_{l=0..L} A^l E0   (out = acc / (L+1))
// `ws_zeroed`: the caller already cleared the whole zero_ws region (training step)
static int propagate(const hiprec_lightgcn_plan* p, const uint8_t* keep, float keep_prob,
                     hipStream_t st, bool ws_zeroed = false) {
  if (use_sliced(p)) return propagate_sliced(p, &p->sa, keep, keep_prob, p->e0, p->acc, true, false, st);
  const size_t bytes = sizeof(float) * p->a.n_rows * p->dim;
  HIPREC_TRY(hipMemcpyAsync(p->acc, p->e0, bytes, hipMemcpyDeviceToDevice, st));
  const bool ws = p->zero_ws != nullptr;
  if (ws && !ws_zeroed && p->n_layers > 0)
    HIPREC_TRY(hipMemsetAsync(ws_slice(p, 1), 0, bytes * p->n_layers, st));
  const float scale = keep ? 1.0f / keep_prob : 1.0f;
  const float* cur = p->e0;
  for (int l = 0; l < p->n_layers; ++l) {
    float* nxt = ws ? ws_slice(p, 1 + l) : ((l & 1) ? p->xb : p->xa);
    if (int rc = launch_spmm(&p->a, keep, scale, cur, nxt, p->acc, p->dim, st, ws)) return rc;
    cur = nxt;
  }
  return 0;
}

}  // namespace hiprec

using namespace hiprec;

extern "C" size_t hiprec_lightgcn_plan_bytes(void) { return sizeof(hiprec_lightgcn_plan); }

extern "C" int hiprec_spmm_csr(const hiprec_csr* a, const uint8_t* keep, float scale, const float* x,
                               float* y, float* acc, int32_t dim, void* stream) {
  if (int rc = check_csr(a, "a")) return rc;
  HIPREC_REQUIRE(x && y && dim > 0, "NULL x / y or bad dim");
  return launch_spmm(a, keep, keep ? scale : 1.0f, x, y, acc, dim, static_cast<hipStream_t>(stream));
}

extern "C" int64_t hiprec_csr_n_slices(int64_t nnz) { return (nnz + kEdgesPerWave - 1) / kEdgesPerWave; }

extern "C" int hiprec_csr_slice_rows(const hiprec_csr* a, int32_t* out, int64_t n_out, void* stream) {
  if (int rc = check_csr(a, "a")) return rc;
  const int64_t n = hiprec_csr_n_slices(a->nnz);
  HIPREC_REQUIRE(a->n_rows < (1ll << 31), "too many rows for 32-bit slice rows");
  HIPREC_REQUIRE(n_out >= n && (n == 0 || out), "slice_row buffer holds %lld entries, %lld needed",
                 (long long)n_out, (long long)n);
  if (n == 0) return 0;
  csr_slice_rows_kernel<<<grid_for_threads(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      a->rowptr, a->n_rows, a->nnz, out, n);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_edge_dropout_mask(uint8_t* keep, int64_t nnz, float keep_prob, uint64_t seed,
                                        uint64_t step, void* stream) {
  HIPREC_REQUIRE(nnz >= 0 && (nnz == 0 || keep), "bad mask buffer");
  if (nnz == 0) return 0;
  edge_dropout_kernel<<<grid_for_threads(nnz), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      keep, nnz, keep_prob, seed, step);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_lightgcn_step_values(const hiprec_lightgcn_plan* plan, uint8_t* keep, float keep_prob,
                                           int32_t draw, uint64_t seed, uint64_t step, void* stream) {
  if (int rc = check_lg_plan(plan, false)) return rc;
  HIPREC_REQUIRE(use_sliced(plan), "the plan has no sliced graphs");
  HIPREC_REQUIRE(plan->sat.n_rows == plan->a.n_rows && plan->sat.eid, "the plan has no transposed sliced graph");
  // the same launch lays E0 out for the first pass (sliced buffer 0, times the graph's column factor) and seeds the
  // layer sum (buffer 1): a step that sets dropped_ready starts with its passes
  SlicedInput in;
  in.x = plan->e0;
  in.n_rows = plan->a.n_rows;
  in.dim = plan->dim;
  in.W = plan->slice_w;
  in.row_scale = plan->sa.col_scale;
  in.xs = sliced_buf(plan, 0);
  in.xs_copy = sliced_buf(plan, 1);
  return launch_step_values(&plan->sa, &plan->sat, keep, draw != 0, keep_prob, seed, step, sliced_buf(plan, 4),
                            sliced_buf(plan, 4) + plan->sa.n_slots, static_cast<hipStream_t>(stream), in);
}

// optimizer.step() of step t AND the preparation of step t + 1 in one launch (sliced plans, device draw): the dense
// optimizer over the N x dim embedding matrix (plan->e0 = w) writes the fresh weights row-major and in the sliced layout
// of the next step's first pass, the rest of the grid draws the next step's dropped edge streams.  The caller starts
// step t + 1 with dropped_ready = 1 -- provided nothing touched the weights in between.
extern "C" int hiprec_lightgcn_opt_stage(const hiprec_lightgcn_plan* plan, int32_t kind, float* g, float* m, float* v,
                                         double lr, double beta1, double beta2, double eps, hiprec_stats* stats,
                                         const void* scratch, float keep_prob, uint64_t seed, uint64_t next_step,
                                         void* stream) {
  if (int rc = check_lg_plan(plan, false)) return rc;
  HIPREC_REQUIRE(use_sliced(plan) && plan->slice_w == 4, "the fused optimizer launch needs a sliced plan of width 4");
  HIPREC_REQUIRE(plan->sat.n_rows == plan->a.n_rows && plan->sat.eid, "the plan has no transposed sliced graph");
  HIPREC_REQUIRE(g && stats, "NULL g / stats");
  SlicedInput in;
  in.x = plan->e0;
  in.n_rows = plan->a.n_rows;
  in.dim = plan->dim;
  in.W = plan->slice_w;
  in.row_scale = plan->sa.col_scale;
  in.xs = sliced_buf(plan, 0);
  in.xs_copy = sliced_buf(plan, 1);
  SlicedOpt f;
  f.kind = kind;
  f.w = plan->e0;
  f.g = g;
  f.m = m;
  f.v = v;
  f.s = OptScalars{lr,
                   static_cast<float>(lr),
                   static_cast<float>(beta2),
                   static_cast<float>(1.0 - beta1),
                   static_cast<float>(1.0 - beta2),
                   static_cast<float>(eps)};
  f.stats = stats;
  f.scratch = static_cast<const Scratch*>(scratch);
  return launch_step_values(&plan->sa, &plan->sat, nullptr, true, keep_prob, seed, next_step, sliced_buf(plan, 4),
                            sliced_buf(plan, 4) + plan->sa.n_slots, static_cast<hipStream_t>(stream), in, &f);
}

extern "C" int hiprec_lightgcn_propagate(const hiprec_lightgcn_plan* plan, const uint8_t* keep,
                                         float keep_prob, void* stream) {
  if (int rc = check_lg_plan(plan, false)) return rc;
  return propagate(plan, keep, keep_prob, static_cast<hipStream_t>(stream));
}

extern "C" int hiprec_lightgcn_predict(const hiprec_lightgcn_plan* plan, const int64_t* users,
                                       const int64_t* items, int64_t n, float* scores,
                                       hiprec_stats* stats, void* stream) {
  if (int rc = check_lg_plan(plan, false)) return rc;
  if (n == 0) return 0;
  HIPREC_REQUIRE(n > 0 && users && items && scores && stats, "NULL pointer");
  lightgcn_predict_kernel<<<grid_for_waves(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      *plan, users, items, n, scores, stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_lightgcn_grad(const hiprec_lightgcn_plan* plan, const uint8_t* keep,
                                    float keep_prob, const int64_t* users, const int64_t* pos,
                                    const int64_t* neg, int64_t batch, float inv_batch,
                                    hiprec_stats* stats, void* scratch, size_t scratch_bytes,
                                    void* stream) {
  if (int rc = check_lg_plan(plan, true)) return rc;
  HIPREC_REQUIRE(batch > 0 && users && pos && neg && stats && scratch, "NULL pointer / empty batch");
  if (scratch_bytes < kScratchBytes) {
    set_error("scratch too small: %zu < %zu", scratch_bytes, kScratchBytes);
    return HIPREC_E_SCRATCH;
  }
  const hiprec_lightgcn_plan* p = plan;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t bytes = sizeof(float) * p->a.n_rows * p->dim;
  const bool ws = p->zero_ws != nullptr;
  hiprec_lightgcn_plan q = *p;  // d_out may live in the workspace
  if (use_sliced(p)) {
    // everything between E0 and g stays in the sliced layout: E0 -> sliced, L passes, the loss kernel reads the
    // sliced layer sum and scatters d_out into sliced buffer 0 (free again after the first pass), L passes of the
    // transposed graph, and one transpose adds the result to g.  No output fills: the SpMM writes every row.
    if (int rc = propagate_sliced(p, &p->sa, keep, keep_prob, p->e0, nullptr, true, false, st, /*clear_buf0=*/true))
      return rc;
    q.acc = sliced_buf(p, 1);
    q.da = sliced_buf(p, 0);
    if (int rc = launch_lightgcn_loss<true>(q, users, pos, neg, batch, inv_batch, stats, static_cast<Scratch*>(scratch), st))
      return rc;
    return propagate_sliced(p, &p->sat, keep, keep_prob, nullptr, p->g, false, true, st);
  }
  if (ws) {
    // ONE fill for d_out and every layer output of the step (it was 7 launches of ~5 us each)
    HIPREC_TRY(hipMemsetAsync(p->zero_ws, 0, bytes * (1 + 2 * static_cast<size_t>(p->n_layers)), st));
    q.da = ws_slice(p, 0);
  }
  if (int rc = propagate(p, keep, keep_prob, st, /*ws_zeroed=*/ws)) return rc;
  if (!ws) HIPREC_TRY(hipMemsetAsync(q.da, 0, bytes, st));
  if (int rc = launch_lightgcn_loss<false>(q, users, pos, neg, batch, inv_batch, stats, static_cast<Scratch*>(scratch), st))
    return rc;
  // g = sum_{l=0..L} (A^T)^l d_out : the l = 0 term is already in g
  const float scale = keep ? 1.0f / keep_prob : 1.0f;
  float* cur = q.da;
  float* nxt = p->db;
  for (int l = 0; l < p->n_layers; ++l) {
    if (ws) nxt = ws_slice(p, 1 + p->n_layers + l);
    if (int rc = launch_spmm(&p->at, keep, scale, cur, nxt, p->g, p->dim, st, ws)) return rc;
    float* t = cur;
    cur = nxt;
    nxt = t;
  }
  return 0;
}
