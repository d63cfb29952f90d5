// MF (matrix factorisation) hot path for gfx950: gather -> score -> BPR/BCE gradient -> scatter.
//
// Replaces, op for op, the PyTorch sequence of beta_rec/models/mf.py:32-55 (MF.forward, called
// twice per BPR step), beta_rec/models/torch_engine.py:92-121 (bpr_loss / bce_loss) and the
// autograd backward of mf.py:117 (8x embedding_dense_backward into dense gradients).
//
// Mapping: one 64-lane wavefront per triple, one lane per embedding column (dim 64 = exactly one
// dword per lane, so a row read is ONE coalesced 256-B global_load_dword and a row of gradient is
// ONE global_atomic_add_f32 wave instruction).  Index triples are wave-uniform -> scalar loads.
// The two dot products are reduced on the VALU with DPP (no LDS round trip); loss / regularizer
// partial sums stay in registers across the grid-stride loop and are published once per block.
#include <algorithm>
#include <type_traits>

#include "common.hpp"

namespace hiprec {

// ---- gradient kernels ---------------------------------------------------------------------------
// One 64-lane wavefront per triple, one lane per embedding column; 16 waves (= 16 consecutive
// triples of the batch) per block.  User and negative-item gradients go straight to the dense
// gradient buffer with global_atomic_add_f32.  Positive items follow a popularity (Zipf) law: in a
// batch of 4096 MovieLens-shaped triples the most popular item is hit ~450 times and same-row
// atomics serialise at ~25 ns each (measured: 15 us/launch against 6 us for uniform items).  So the
// positive-item gradients of a block are first merged in LDS: adjacent triples with the same item
// (the batcher sorts each batch by item, which does not change a sum over the batch) are added
// into the slot of the run's first wave with ds_add_f32, and only run heads issue the global
// atomic.  An unsorted batch is still handled correctly, it just merges less.
//
// NPL = embedding columns held per lane (dim <= 64*NPL).  NPL == 0: any dim, rows are re-read
// (from L1/L2) when needed instead of being held in registers.


// Triple of one wave for one trip of the grid-stride loop (all fields wave-uniform).
struct TripleIdx {
  int64_t u, p, n;
  bool valid;
};

__device__ __forceinline__ TripleIdx load_triple(const hiprec_mf_tables& w,
                                                 const int64_t* __restrict__ users,
                                                 const int64_t* __restrict__ pos,
                                                 const int64_t* __restrict__ neg,
                                                 const int64_t* __restrict__ perm, int64_t t,
                                                 int64_t batch, hiprec_stats* stats, int lane) {
  TripleIdx r{0, 0, 0, t < batch};
  if (r.valid) {
    const int64_t j = perm ? perm[t] : t;
    r.u = users[j];
    r.p = pos[j];
    r.n = neg[j];
    const bool u_ok = static_cast<uint64_t>(r.u) < static_cast<uint64_t>(w.n_users);
    const bool i_ok = static_cast<uint64_t>(r.p) < static_cast<uint64_t>(w.n_items) &&
                      static_cast<uint64_t>(r.n) < static_cast<uint64_t>(w.n_items);
    if (r.u == -1) {
      r.valid = false;  // padding slot of a fixed-capacity exchange
    } else if (!(u_ok && i_ok)) {
      if (lane == 0)
        atomicOr(&stats->status,
                 (u_ok ? 0u : HIPREC_STATUS_USER_OOB) | (i_ok ? 0u : HIPREC_STATUS_ITEM_OOB));
      r.valid = false;
    }
  }
  return r;
}

template <int NPL>
__global__ __launch_bounds__(kAggBlock) void mf_bpr_grad_kernel(
    hiprec_mf_tables w, hiprec_mf_tables g, const int64_t* __restrict__ users,
    const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
    const int64_t* __restrict__ perm, int64_t batch, float inv_batch, float reg_coef,
    hiprec_stats* stats, Scratch* scratch) {
  extern __shared__ __attribute__((aligned(16))) float s_acc[];
  __shared__ long long s_item[kAggWaves];
  const int lane = lane_id();
  const int wv = wave_in_block();
  const int D = w.dim;
  const int ld = D + 1;
  const float gb = load_scalar_param(w.global_bias);
  // mf.py:116 batch_loss = loss + reg*regularizer; user terms appear in both forward calls
  const float ru = 4.f * reg_coef * inv_batch, ri = 2.f * reg_coef * inv_batch;
  constexpr int R = NPL > 0 ? NPL : 1;

  float loss_acc = 0.f;  // wave-uniform
  float reg_acc = 0.f;   // per lane
  float gb_acc = 0.f;    // wave-uniform: d(loss)/d(global_bias)

  const bool stepper = blockIdx.x == 0 && threadIdx.x == 0;
  StepState step_state{};
  if (stepper) step_state = step_load(stats);

  const int64_t stride = static_cast<int64_t>(gridDim.x) * kAggWaves;
  const int64_t first = static_cast<int64_t>(blockIdx.x) * kAggWaves;

  // Software pipeline over the grid-stride loop (batches larger than the grid, i.e. the
  // HBM-resident regime): the rows of trip i+1 are requested right after the dot products of trip
  // i and land while trip i sits in its atomics and LDS-merge barriers.
  float cu[R], cp[R], cn[R], cbu = 0.f, cbp = 0.f, cbn = 0.f;
  auto fetch_rows = [&](const TripleIdx& tr, float (&ru_)[R], float (&rp_)[R], float (&rn_)[R],
                        float& bu_, float& bp_, float& bn_) {
    if constexpr (NPL > 0) {
      const float* ur = w.user_emb + tr.u * D;
      const float* pr = w.item_emb + tr.p * D;
      const float* nr = w.item_emb + tr.n * D;
#pragma unroll
      for (int k = 0; k < NPL; ++k) {
        const int c = lane + kWave * k;
        const bool in = tr.valid && c < D;
        ru_[k] = in ? ur[c] : 0.f;
        rp_[k] = in ? pr[c] : 0.f;
        rn_[k] = in ? nr[c] : 0.f;
      }
    }
    bu_ = tr.valid ? w.user_bias[tr.u] : 0.f;
    bp_ = tr.valid ? w.item_bias[tr.p] : 0.f;
    bn_ = tr.valid ? w.item_bias[tr.n] : 0.f;
  };

  TripleIdx cur = load_triple(w, users, pos, neg, perm, first + wv, batch, stats, lane);
  fetch_rows(cur, cu, cp, cn, cbu, cbp, cbn);
  // the stepper's loads were requested before the index loads, so they are back by now: finish the
  // update in the shadow of the row gather
  if (stepper) step_store_advanced(stats, step_state);

  for (int64_t base = first; base < batch; base += stride) {
    const bool valid = cur.valid;
    const int64_t u = cur.u, p = cur.p, n = cur.n;
    if (lane == 0) s_item[wv] = valid ? static_cast<long long>(p) : -static_cast<long long>(wv + 1);
    // indices of the next trip: scalar loads, issued before the barrier
    const bool more = base + stride < batch;
    TripleIdx nxt{0, 0, 0, false};
    if (more) nxt = load_triple(w, users, pos, neg, perm, base + stride + wv, batch, stats, lane);
    lds_barrier();
    const int head = valid ? run_head(s_item, wv, p) : wv;
    const bool is_head = head == wv;
    const int tail = (valid && is_head) ? run_end(s_item, wv, p, kAggWaves) : wv + 1;

    const float* ur = w.user_emb + u * D;
    const float* pr = w.item_emb + p * D;
    const float* nr = w.item_emb + n * D;
    float uu[R], pp[R];
    float dpos = 0.f, dneg = 0.f, bp = cbp, bu = cbu, bn = cbn;
    float dp = 0.f, dn = 0.f;
    float nn[R];
    if (valid) {
      if constexpr (NPL > 0) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
          uu[k] = cu[k];
          pp[k] = cp[k];
          nn[k] = cn[k];
          dp += uu[k] * pp[k];
          dn += uu[k] * nn[k];
          reg_acc += 2.f * uu[k] * uu[k] + pp[k] * pp[k] + nn[k] * nn[k];
        }
      } else {
        for (int c = lane; c < D; c += kWave) {
          const float a = ur[c], b = pr[c], d = nr[c];
          dp += a * b;
          dn += a * d;
          reg_acc += 2.f * a * a + b * b + d * d;
        }
      }
    }
    // request the next trip's rows now: they travel while this trip finishes
    if (more) fetch_rows(nxt, cu, cp, cn, cbu, cbp, cbn);
    if (valid) {
      float* gur = g.user_emb + u * D;
      float* gnr = g.item_emb + n * D;
      dp = wave_sum(dp);
      dn = wave_sum(dn);
      // mf.py:43-48: sigmoid(sum + u_bias + i_bias + global_bias)
      const float yp = sigmoid_f32(((dp + bu) + bp) + gb);
      const float yn = sigmoid_f32(((dn + bu) + bn) + gb);
      // torch_engine.py:104-105: -mean(logsigmoid(pos - neg))
      float sig_neg_x;
      const float nls = neg_logsigmoid(yp - yn, &sig_neg_x);
      const float delta = -sig_neg_x * inv_batch;    // dL/d(yp) ; dL/d(yn) = -delta
      dpos = delta * ((1.f - yp) * yp);              // through the sigmoid
      dneg = -delta * ((1.f - yn) * yn);

      // user row, negative-item row: no popularity skew -> straight to the dense gradient;
      // positive-item row of a run head: parked in its LDS slot.
      float* slot = s_acc + wv * ld;
      if constexpr (NPL > 0) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
          const int c = lane + kWave * k;
          if (c < D) {
            atomic_add_f32(gur + c, (dpos * pp[k] + dneg * nn[k]) + ru * uu[k]);
            atomic_add_f32(gnr + c, dneg * uu[k] + ri * nn[k]);
            slot[c] = dpos * uu[k] + ri * pp[k];
          }
        }
      } else {
        for (int c = lane; c < D; c += kWave) {
          const float a = ur[c], b = pr[c], d = nr[c];
          atomic_add_f32(gur + c, (dpos * b + dneg * d) + ru * a);
          atomic_add_f32(gnr + c, dneg * a + ri * d);
          slot[c] = dpos * a + ri * b;
        }
      }
      if (lane == 0) {
        atomic_add_f32(g.user_bias + u, (dpos + dneg) + ru * bu);
        atomic_add_f32(g.item_bias + n, dneg + ri * bn);
        slot[D] = dpos + ri * bp;
        reg_acc += 2.f * bu * bu + bp * bp + bn * bn;
      }
      loss_acc += nls;
      gb_acc += dpos + dneg;
    }
    lds_barrier();
    if (valid && is_head) {  // the head sums its run's slots (plain LDS reads: LDS float atomics cost ~3 cycles per
                             // lane, experiments 39) and issues one global atomic per element of the run
      float* gpr = g.item_emb + p * D;
      for (int c = lane; c <= D; c += kWave) {
        float t = s_acc[wv * ld + c];
        for (int j = wv + 1; j < tail; ++j) t += s_acc[j * ld + c];
        atomic_add_f32(c < D ? gpr + c : g.item_bias + p, t);
      }
    }
    cur = nxt;
  }
  // d(loss)/d(global_bias) goes out with the per-block partials (no same-address atomics)
  publish_partials<kAggWaves>(loss_acc, reg_acc, gb_acc, inv_batch, scratch);
}

// ---- fused step: ONE launch per step ---------------------------------------------------------------
// torch's SGD / Adam / RMSprop updates are elementwise functions of (w, g, m, v), so the optimizer
// kernel of step k-1 can ride inside the gradient kernel of step k without giving up the reference's
// semantics (all gradients of a batch come from the pre-step weights):
//   gather blocks   evaluate  w_eff = update(W_a, G_prev, M_a, V_a)[row]  on the fly  (= the weights
//                   after step k-1; none of the four is written by this launch) and accumulate the
//                   gradients of step k into G_cur;
//   sweep blocks    (the rest of the same grid) write W_b, M_b, V_b = update(...) for the WHOLE buffer
//                   and clear G_zero (the accumulator of step k+1); the first of them reduces step
//                   k-1's loss partials into the stats, advances the step counter and leaves step
//                   k's bias-correction scalars in this step's scratch header for launch k+1.
// W, M, V ping-pong between two buffers each, G rotates through three; the arithmetic per element is
// opt_update<KIND>, the very expression the dense sweep of the two-kernel path evaluates.
constexpr int kFusedMaxGather = 256;  // gather blocks of one fused launch (bigger batches loop)
constexpr int kFusedSlots = kFusedMaxGather / kWave;

// Everything is addressed through FLAT buffers + one shared set of offsets (all of W, G, M, V have the
// layout of hiprec_mf_tables): five table structs as kernel arguments cost 80 SGPRs and the compiler
// spilled as many (80 v_writelane + 110 v_readlane in the Adam instantiation).
struct FusedOpt {
  int64_t n_users, n_items;
  int32_t dim, _pad0;
  float* g_cur;                 // flat accumulator of THIS step
  const float* w_read;          // flat W_a
  const float* g_prev;          // flat G_prev
  const float* m_read;          // flat M_a (Adam)
  const float* v_read;          // flat V_a (Adam, RMSprop)
  float* w_write;               // flat W_b
  float* m_write;
  float* v_write;
  float* g_zero;                // flat accumulator of the NEXT step, cleared here
  float* g_zero2;               // flush only: the gradient being applied, cleared as it is consumed (else NULL)
  int64_t n_flat;
  OptScalars s;
  int n_gather_blocks;
  int n_prev_partials;          // gather blocks of step k-1, known to the host: spares the gather
                                // blocks a dependent load of scratch_prev->n_partials (~1 us miss)
  int apply_prev;               // 0 for the first launch of an epoch: nothing is pending
  const Scratch* scratch_prev;  // partials + step scalars of step k-1
};

template <int KIND>
__device__ __forceinline__ float fused_eff(float w, float g, float m, float v, const OptScalars& s,
                                           float step_size, float bc2_sqrt) {
  opt_update<KIND>(w, g, m, v, s, step_size, bc2_sqrt);
  return w;
}

// Two 16-wave blocks must fit on a CU (8 waves per SIMD, i.e. <= 64 VGPRs): the sweep blocks are the
// second half of the grid and would otherwise wait for a gather block to retire (measured: Adam at 70
// VGPRs ran 15.4 us per step instead of 10.7).  The dim > 64 variants keep their registers.
// FLUSH: the sweep-only launch that ends an epoch (no gather blocks): it also clears the gradient it
// applies (g_zero2) and marks both scratch blocks empty, and may run in place (w_write == w_read).  A
// separate instantiation so that the per-step kernel carries none of it.
template <int NPL, int KIND, bool FLUSH>
__global__ __launch_bounds__(kAggBlock) __attribute__((amdgpu_waves_per_eu(NPL == 1 ? 8 : 4, 8)))
void mf_bpr_fused_kernel(
    FusedOpt f, const int64_t* __restrict__ users, const int64_t* __restrict__ pos,
    const int64_t* __restrict__ neg, int64_t batch, float inv_batch, float reg_coef,
    hiprec_stats* stats, Scratch* scratch) {
  extern __shared__ __attribute__((aligned(16))) float s_acc[];
  __shared__ long long s_item[kAggWaves];
  const int lane = lane_id();
  const int wv = wave_in_block();
  constexpr bool kHasM = KIND == HIPREC_OPT_ADAM;
  constexpr bool kHasV = KIND != HIPREC_OPT_SGD;
  const bool apply = f.apply_prev != 0;

  if (static_cast<int>(blockIdx.x) >= f.n_gather_blocks) {
    // ------------------------------ sweep part ------------------------------
    const int sb = static_cast<int>(blockIdx.x) - f.n_gather_blocks;
    const int n_sb = static_cast<int>(gridDim.x) - f.n_gather_blocks;
    float step_size = f.s.lr, bc2_sqrt = 1.f;
    if constexpr (KIND == HIPREC_OPT_ADAM) {
      step_size = __uint_as_float(f.scratch_prev->_pad[0]);
      bc2_sqrt = __uint_as_float(f.scratch_prev->_pad[1]);
    }
    const int64_t gb_index = f.n_flat - 1;  // global_bias is the last element of the flat layout
    const int64_t n4 = f.n_flat >> 2;
    const int64_t skip4 = (gb_index < (n4 << 2)) ? (gb_index >> 2) : -1;
    const float4* wr4 = reinterpret_cast<const float4*>(f.w_read);
    const float4* gp4 = reinterpret_cast<const float4*>(f.g_prev);
    const float4* mr4 = reinterpret_cast<const float4*>(f.m_read);
    const float4* vr4 = reinterpret_cast<const float4*>(f.v_read);
    float4* ww4 = reinterpret_cast<float4*>(f.w_write);
    float4* mw4 = reinterpret_cast<float4*>(f.m_write);
    float4* vw4 = reinterpret_cast<float4*>(f.v_write);
    float4* gz4 = reinterpret_cast<float4*>(f.g_zero);
    const int64_t stride = static_cast<int64_t>(n_sb) * kAggBlock;
    for (int64_t i = static_cast<int64_t>(sb) * kAggBlock + threadIdx.x; i < n4; i += stride) {
      if (i == skip4) continue;
      float4 a = wr4[i], d = gp4[i];
      float4 mv = make_float4(0.f, 0.f, 0.f, 0.f), vv = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (kHasM) mv = mr4[i];
      if constexpr (kHasV) vv = vr4[i];
      if (apply) {
        opt_update<KIND>(a.x, d.x, mv.x, vv.x, f.s, step_size, bc2_sqrt);
        opt_update<KIND>(a.y, d.y, mv.y, vv.y, f.s, step_size, bc2_sqrt);
        opt_update<KIND>(a.z, d.z, mv.z, vv.z, f.s, step_size, bc2_sqrt);
        opt_update<KIND>(a.w, d.w, mv.w, vv.w, f.s, step_size, bc2_sqrt);
      }
      ww4[i] = a;
      if constexpr (kHasM) mw4[i] = mv;
      if constexpr (kHasV) vw4[i] = vv;
      gz4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (FLUSH) reinterpret_cast<float4*>(f.g_zero2)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto scalar_update = [&](int64_t i, float extra_g) {
      float a = f.w_read[i], d = f.g_prev[i] + extra_g, mv = 0.f, vv = 0.f;
      if constexpr (kHasM) mv = f.m_read[i];
      if constexpr (kHasV) vv = f.v_read[i];
      if (apply) opt_update<KIND>(a, d, mv, vv, f.s, step_size, bc2_sqrt);
      f.w_write[i] = a;
      if constexpr (kHasM) f.m_write[i] = mv;
      if constexpr (kHasV) f.v_write[i] = vv;
      f.g_zero[i] = 0.f;
      if constexpr (FLUSH) f.g_zero2[i] = 0.f;
    };
    for (int64_t i = (n4 << 2) + static_cast<int64_t>(sb) * kAggBlock + threadIdx.x; i < f.n_flat;
         i += stride) {
      if (i != gb_index) scalar_update(i, 0.f);
    }
    if (sb == 0) {
      if (!apply && threadIdx.x == 0) {  // first launch of an epoch: hiprec_stats_begin_epoch, folded in
        stats->loss_sum = 0.0;
        stats->reg_sum = 0.0;
      }
      const float gb_part = finalize_partials<kAggBlock>(stats, f.scratch_prev);
      if (threadIdx.x == 0) {
        if constexpr (FLUSH) {  // both scratch blocks are spent, leave them marked empty
          scratch->n_partials = 0;
          const_cast<Scratch*>(f.scratch_prev)->n_partials = 0;
        }
        const int64_t lo = skip4 >= 0 ? (skip4 << 2) : gb_index;
        for (int64_t i = lo; i <= gb_index; ++i) scalar_update(i, i == gb_index ? gb_part : 0.f);
        if (batch > 0) {
          // count step k and leave its bias-correction scalars for launch k+1 (kept off the gather
          // blocks' critical path; they read step k-1's pair from scratch_prev, never this one)
          advance_step(stats);
          float ss, bq;
          step_scalars<KIND>(f.s, stats, &ss, &bq);
          scratch->_pad[0] = __float_as_uint(ss);
          scratch->_pad[1] = __float_as_uint(bq);
        }
      }
    }
    return;
  }

  // ------------------------------ gather part ------------------------------
  if constexpr (FLUSH) return;  // a flush launch has no gather blocks
  const int D = f.dim;
  // offsets of the five tensors inside any of the flat buffers
  const int64_t o_ie = f.n_users * D, o_ub = o_ie + f.n_items * D, o_ib = o_ub + f.n_users,
                o_gb = o_ib + f.n_items;
  const float* const wf = f.w_read;
  float* const gf = f.g_cur;
  const int ld = D + 1;
  const float ru = 4.f * reg_coef * inv_batch, ri = 2.f * reg_coef * inv_batch;
  // global_bias after step k-1: its gradient lives in the previous step's partials.  The loads are
  // only REQUESTED here; they travel together with the index and row loads below and are reduced
  // at their first use (the sigmoid), so the launch pays one memory round trip for them instead
  // of two in front of everything else.
  float gbz[kFusedSlots];  // every slot of the scratch block is addressable: no branches
  {
    const float4* pv = f.scratch_prev->partials;
#pragma unroll
    for (int j = 0; j < kFusedSlots; ++j) gbz[j] = pv[lane + kWave * j].z;
  }
  // VECTOR loads on purpose: as scalar loads they would share lgkmcnt with the index loads below,
  // and all of these were written by the previous launch (full misses).
  const float gb_w = load_scalar_param(wf + o_gb), gb_g = load_scalar_param(f.g_prev + o_gb);
  float gb_m = 0.f, gb_v = 0.f, step_size = f.s.lr, bc2_sqrt = 1.f;
  if constexpr (kHasM) gb_m = load_scalar_param(f.m_read + o_gb);
  if constexpr (kHasV) gb_v = load_scalar_param(f.v_read + o_gb);
  if constexpr (KIND == HIPREC_OPT_ADAM) {
    const float* hdr = reinterpret_cast<const float*>(f.scratch_prev->_pad);
    step_size = load_scalar_param(hdr);
    bc2_sqrt = load_scalar_param(hdr + 1);
  }
  float gb = 0.f;
  bool gb_ready = false;

  float loss_acc = 0.f, reg_acc = 0.f, gb_acc = 0.f;

  for (int64_t base = static_cast<int64_t>(blockIdx.x) * kAggWaves; base < batch;
       base += static_cast<int64_t>(f.n_gather_blocks) * kAggWaves) {
    const int64_t t = base + wv;
    bool valid = t < batch;
    int64_t u = 0, p = 0, n = 0;
    if (valid) {
      u = users[t];
      p = pos[t];
      n = neg[t];
      const bool u_ok = static_cast<uint64_t>(u) < static_cast<uint64_t>(f.n_users);
      const bool i_ok = static_cast<uint64_t>(p) < static_cast<uint64_t>(f.n_items) &&
                        static_cast<uint64_t>(n) < static_cast<uint64_t>(f.n_items);
      if (!(u_ok && i_ok)) {
        if (lane == 0)
          atomicOr(&stats->status,
                   (u_ok ? 0u : HIPREC_STATUS_USER_OOB) | (i_ok ? 0u : HIPREC_STATUS_ITEM_OOB));
        valid = false;
      }
    }
    if (lane == 0) s_item[wv] = valid ? static_cast<long long>(p) : -static_cast<long long>(wv + 1);
    lds_barrier();
    const int head = valid ? run_head(s_item, wv, p) : wv;
    const bool is_head = head == wv;
    const int tail = (valid && is_head) ? run_end(s_item, wv, p, kAggWaves) : wv + 1;

    float uu[NPL], pp[NPL];
    float dpos = 0.f, bp = 0.f;
    if (valid) {
      const int64_t ou = u * D, op = o_ie + p * D, on = o_ie + n * D;  // rows inside a flat buffer
      float nn[NPL];
      float dp = 0.f, dn = 0.f;
      // everything this triple needs goes out in one burst (one round trip): biases ...
      const float wbu = wf[o_ub + u], gbu = f.g_prev[o_ub + u];
      const float wbn = wf[o_ib + n], gbn = f.g_prev[o_ib + n];
      const float wbp = wf[o_ib + p], gbp_ = f.g_prev[o_ib + p];
      float mbu = 0.f, mbn = 0.f, mbp = 0.f, vbu = 0.f, vbn = 0.f, vbp = 0.f;
      if constexpr (kHasM) {
        mbu = f.m_read[o_ub + u];
        mbn = f.m_read[o_ib + n];
        mbp = f.m_read[o_ib + p];
      }
      if constexpr (kHasV) {
        vbu = f.v_read[o_ub + u];
        vbn = f.v_read[o_ib + n];
        vbp = f.v_read[o_ib + p];
      }
      // ... and rows
      float wu[NPL], gu[NPL], wp[NPL], gpv[NPL], wn[NPL], gn[NPL];
      float mu[NPL], mpv[NPL], mn[NPL], vu[NPL], vpv[NPL], vn[NPL];
#pragma unroll
      for (int k = 0; k < NPL; ++k) {
        const int c = lane + kWave * k;
        const int cc = c < D ? c : D - 1;  // always a valid column: unconditional loads, one wait
        wu[k] = wf[ou + cc];
        gu[k] = f.g_prev[ou + cc];
        wp[k] = wf[op + cc];
        gpv[k] = f.g_prev[op + cc];
        wn[k] = wf[on + cc];
        gn[k] = f.g_prev[on + cc];
        mu[k] = mpv[k] = mn[k] = vu[k] = vpv[k] = vn[k] = 0.f;
        if constexpr (kHasM) {
          mu[k] = f.m_read[ou + cc];
          mpv[k] = f.m_read[op + cc];
          mn[k] = f.m_read[on + cc];
        }
        if constexpr (kHasV) {
          vu[k] = f.v_read[ou + cc];
          vpv[k] = f.v_read[op + cc];
          vn[k] = f.v_read[on + cc];
        }
      }
#pragma unroll
      for (int k = 0; k < NPL; ++k) {
        const bool in = lane + kWave * k < D;
        // weights after step k-1, evaluated on the fly (same expression as the sweep writes)
        if (apply) {
          wu[k] = fused_eff<KIND>(wu[k], gu[k], mu[k], vu[k], f.s, step_size, bc2_sqrt);
          wp[k] = fused_eff<KIND>(wp[k], gpv[k], mpv[k], vpv[k], f.s, step_size, bc2_sqrt);
          wn[k] = fused_eff<KIND>(wn[k], gn[k], mn[k], vn[k], f.s, step_size, bc2_sqrt);
        }
        uu[k] = in ? wu[k] : 0.f;
        pp[k] = in ? wp[k] : 0.f;
        nn[k] = in ? wn[k] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < NPL; ++k) {
        dp += uu[k] * pp[k];
        dn += uu[k] * nn[k];
        reg_acc += 2.f * uu[k] * uu[k] + pp[k] * pp[k] + nn[k] * nn[k];
      }
      dp = wave_sum(dp);
      dn = wave_sum(dn);
      if (!gb_ready) {
        float gbp = 0.f;
#pragma unroll
        for (int j = 0; j < kFusedSlots; ++j) {
          asm volatile("" : "+v"(gbz[j]));  // keep the reduction (and its vmcnt wait) down here
          gbp += lane + kWave * j < f.n_prev_partials ? gbz[j] : 0.f;
        }
        gb = apply ? fused_eff<KIND>(gb_w, gb_g + wave_sum(gbp), gb_m, gb_v, f.s, step_size, bc2_sqrt)
                   : gb_w;
        gb_ready = true;
      }
      float bu = wbu, bn = wbn;
      bp = wbp;
      if (apply) {
        bu = fused_eff<KIND>(wbu, gbu, mbu, vbu, f.s, step_size, bc2_sqrt);
        bn = fused_eff<KIND>(wbn, gbn, mbn, vbn, f.s, step_size, bc2_sqrt);
        bp = fused_eff<KIND>(wbp, gbp_, mbp, vbp, f.s, step_size, bc2_sqrt);
      }
      const float yp = sigmoid_f32(((dp + bu) + bp) + gb);
      const float yn = sigmoid_f32(((dn + bu) + bn) + gb);
      float sig_neg_x;
      const float nls = neg_logsigmoid(yp - yn, &sig_neg_x);
      const float delta = -sig_neg_x * inv_batch;
      dpos = delta * ((1.f - yp) * yp);
      const float dneg = -delta * ((1.f - yn) * yn);
      float* gur = gf + ou;
      float* gnr = gf + on;
      float* slot = s_acc + wv * ld;
#pragma unroll
      for (int k = 0; k < NPL; ++k) {
        const int c = lane + kWave * k;
        if (c < D) {
          atomic_add_f32(gur + c, (dpos * pp[k] + dneg * nn[k]) + ru * uu[k]);
          atomic_add_f32(gnr + c, dneg * uu[k] + ri * nn[k]);
          slot[c] = dpos * uu[k] + ri * pp[k];
        }
      }
      if (lane == 0) {
        atomic_add_f32(gf + o_ub + u, (dpos + dneg) + ru * bu);
        atomic_add_f32(gf + o_ib + n, dneg + ri * bn);
        slot[D] = dpos + ri * bp;
        reg_acc += 2.f * bu * bu + bp * bp + bn * bn;
      }
      loss_acc += nls;
      gb_acc += dpos + dneg;
    }
    lds_barrier();
    if (valid && is_head) {  // the head sums its run's slots: plain LDS reads instead of LDS float atomics
      float* gpr = gf + o_ie + p * D;
      for (int c = lane; c <= D; c += kWave) {
        float t = s_acc[wv * ld + c];
        for (int j = wv + 1; j < tail; ++j) t += s_acc[j * ld + c];
        atomic_add_f32(c < D ? gpr + c : gf + o_ib + p, t);
      }
    }
  }
  // publish this step's partials; n_partials = number of GATHER blocks
  __shared__ float s_loss[kAggWaves], s_reg[kAggWaves], s_gbv[kAggWaves];
  const float reg_w = wave_sum(reg_acc);
  if (lane == 0) {
    s_loss[wv] = loss_acc;
    s_reg[wv] = reg_w;
    s_gbv[wv] = gb_acc;
  }
  lds_barrier();
  if (threadIdx.x == 0) {
    float l = 0.f, r = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < kAggWaves; ++i) {
      l += s_loss[i];
      r += s_reg[i];
      b += s_gbv[i];
    }
    scratch->partials[blockIdx.x] = make_float4(l * inv_batch, r * inv_batch, b, 0.f);
    if (blockIdx.x == 0) scratch->n_partials = static_cast<uint32_t>(f.n_gather_blocks);
  }
}

template <int NPL>
__global__ __launch_bounds__(kAggBlock) void mf_bce_grad_kernel(
    hiprec_mf_tables w, hiprec_mf_tables g, const int64_t* __restrict__ users,
    const int64_t* __restrict__ items, const float* __restrict__ ratings,
    const int64_t* __restrict__ perm, int64_t batch, float inv_batch, float reg_coef,
    hiprec_stats* stats, Scratch* scratch) {
  extern __shared__ __attribute__((aligned(16))) float s_acc[];
  __shared__ long long s_item[kAggWaves];
  const int lane = lane_id();
  const int wv = wave_in_block();
  const int D = w.dim;
  const int ld = D + 1;
  const float gb = load_scalar_param(w.global_bias);
  const float rr = 2.f * reg_coef * inv_batch;

  float loss_acc = 0.f, reg_acc = 0.f, gb_acc = 0.f;
  const bool stepper = blockIdx.x == 0 && threadIdx.x == 0;
  StepState step_state{};
  if (stepper) step_state = step_load(stats);

  for (int64_t base = static_cast<int64_t>(blockIdx.x) * kAggWaves; base < batch;
       base += static_cast<int64_t>(gridDim.x) * kAggWaves) {
    const int64_t t = base + wv;
    bool valid = t < batch;
    int64_t u = 0, i = 0;
    float r = 0.f;
    if (valid) {
      const int64_t j = perm ? perm[t] : t;
      u = users[j];
      i = items[j];
      r = ratings[j];
      const bool u_ok = static_cast<uint64_t>(u) < static_cast<uint64_t>(w.n_users);
      const bool i_ok = static_cast<uint64_t>(i) < static_cast<uint64_t>(w.n_items);
      if (!(u_ok && i_ok)) {
        if (lane == 0)
          atomicOr(&stats->status,
                   (u_ok ? 0u : HIPREC_STATUS_USER_OOB) | (i_ok ? 0u : HIPREC_STATUS_ITEM_OOB));
        valid = false;
      }
    }
    if (lane == 0) s_item[wv] = valid ? static_cast<long long>(i) : -static_cast<long long>(wv + 1);
    lds_barrier();
    const int head = valid ? run_head(s_item, wv, i) : wv;
    const bool is_head = head == wv;
    const int tail = (valid && is_head) ? run_end(s_item, wv, i, kAggWaves) : wv + 1;

    const float* ur = w.user_emb + u * D;
    const float* ir = w.item_emb + i * D;
    float uu[NPL > 0 ? NPL : 1], ii[NPL > 0 ? NPL : 1];
    float ds = 0.f, bi = 0.f;
    if (valid) {
      float* gur = g.user_emb + u * D;
      float dot = 0.f;
      if constexpr (NPL > 0) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
          const int c = lane + kWave * k;
          const bool in = c < D;
          uu[k] = in ? ur[c] : 0.f;
          ii[k] = in ? ir[c] : 0.f;
          dot += uu[k] * ii[k];
          reg_acc += uu[k] * uu[k] + ii[k] * ii[k];
        }
      } else {
        for (int c = lane; c < D; c += kWave) {
          const float a = ur[c], b = ir[c];
          dot += a * b;
          reg_acc += a * a + b * b;
        }
      }
      dot = wave_sum(dot);
      const float bu = w.user_bias[u];
      bi = w.item_bias[i];
      const float y = sigmoid_f32(((dot + bu) + bi) + gb);
      // torch.nn.BCELoss (mean): -(r*max(log y,-100) + (1-r)*max(log(1-y),-100))
      const float ly = fmaxf(logf(y), -100.f);
      const float l1y = fmaxf(log1pf(-y), -100.f);
      const float loss_k = -(r * ly + (1.f - r) * l1y);
      // ATen binary_cross_entropy_backward, then sigmoid_backward
      const float gy = (y - r) / fmaxf((1.f - y) * y, 1e-12f) * inv_batch;
      ds = gy * ((1.f - y) * y);

      float* slot = s_acc + wv * ld;
      if constexpr (NPL > 0) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
          const int c = lane + kWave * k;
          if (c < D) {
            atomic_add_f32(gur + c, ds * ii[k] + rr * uu[k]);
            slot[c] = ds * uu[k] + rr * ii[k];
          }
        }
      } else {
        for (int c = lane; c < D; c += kWave) {
          const float a = ur[c], b = ir[c];
          atomic_add_f32(gur + c, ds * b + rr * a);
          slot[c] = ds * a + rr * b;
        }
      }
      if (lane == 0) {
        atomic_add_f32(g.user_bias + u, ds + rr * bu);
        slot[D] = ds + rr * bi;
        reg_acc += bu * bu + bi * bi;
      }
      loss_acc += loss_k;
      gb_acc += ds;
    }
    lds_barrier();
    if (valid && is_head) {  // the head sums its run's slots: plain LDS reads instead of LDS float atomics
      float* gir = g.item_emb + i * D;
      for (int c = lane; c <= D; c += kWave) {
        float t = s_acc[wv * ld + c];
        for (int j = wv + 1; j < tail; ++j) t += s_acc[j * ld + c];
        atomic_add_f32(c < D ? gir + c : g.item_bias + i, t);
      }
    }
  }
  publish_partials<kAggWaves>(loss_acc, reg_acc, gb_acc, inv_batch, scratch);
  if (stepper) step_store_advanced(stats, step_state);
}

// scores[k] = sigmoid(<U[u], I[i]> + bu + bi + g)   (MF.predict, mf.py:57-70)
// sq (optional): sq[k] = |U[u]|^2 + |I[i]|^2 + bu^2 + bi^2, the sample's share of MF.forward's regularizer
// (mf.py:46-53) -- the rows are in registers for the dot product anyway.
__global__ __launch_bounds__(kBlock) void mf_predict_kernel(hiprec_mf_tables w,
                                                            const int64_t* __restrict__ users,
                                                            const int64_t* __restrict__ items,
                                                            int64_t n, float* __restrict__ scores,
                                                            float* __restrict__ sq, hiprec_stats* stats) {
  const int lane = lane_id();
  const int D = w.dim;
  const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const float gb = load_scalar_param(w.global_bias);
  for (int64_t t = wave0; t < n; t += n_waves) {
    const int64_t u = users[t], i = items[t];
    const bool u_ok = static_cast<uint64_t>(u) < static_cast<uint64_t>(w.n_users);
    const bool i_ok = static_cast<uint64_t>(i) < static_cast<uint64_t>(w.n_items);
    if (!(u_ok && i_ok)) {
      if (lane == 0) {
        atomicOr(&stats->status,
                 (u_ok ? 0u : HIPREC_STATUS_USER_OOB) | (i_ok ? 0u : HIPREC_STATUS_ITEM_OOB));
        scores[t] = __builtin_nanf("");
        if (sq) sq[t] = 0.f;
      }
      continue;
    }
    const float* ur = w.user_emb + u * D;
    const float* ir = w.item_emb + i * D;
    float dot = 0.f, ss = 0.f;
    for (int c = lane; c < D; c += kWave) {
      const float a = ur[c], b = ir[c];
      dot += a * b;
      ss += a * a + b * b;
    }
    dot = wave_sum(dot);
    if (sq) ss = wave_sum(ss);
    if (lane == 0) {
      const float bu = w.user_bias[u], bi = w.item_bias[i];
      scores[t] = sigmoid_f32(((dot + bu) + bi) + gb);
      if (sq) sq[t] = (ss + bu * bu) + bi * bi;
    }
  }
}

// Exact SGD on the rows a batch touched.  One wave per triple: lanes 0..2 race (one returning
// atomicExch each, all three in flight together) for the stamps of the triple's user / item rows;
// the single winner of a row applies w -= lr*g and clears g.
// NPL = columns per lane (dim <= 64 * NPL; 0 = any dim, rows streamed).  A trip is one triple: the
// stamps of its three rows are raced first, then the gradient and weight rows of EVERY row this wave
// won are requested together and only then updated and stored, and the next trip's indices are
// already on their way -- the first version took the rows one after the other (five dependent
// memory round trips per trip, 135 us per 65 536-triple step at the configs[3] size; the gradient
// kernel of the same step takes 100).
struct RowPtrs {
  float* w;
  float* g;
  float* wb;
  float* gb;
};

template <int NPL>
__global__ __launch_bounds__(kBlock) void mf_sgd_rows_kernel(
    hiprec_mf_tables w, hiprec_mf_tables g, const int64_t* __restrict__ users,
    const int64_t* __restrict__ items_a, const int64_t* __restrict__ items_b,
    const int64_t* __restrict__ perm, int64_t batch, float lr, int32_t* user_stamp,
    int32_t* item_stamp, int32_t stamp, hiprec_stats* stats, const Scratch* scratch) {
  const int lane = lane_id();
  const int D = w.dim;
  const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  constexpr int R = NPL > 0 ? NPL : 1;

  // users / items_a may be NULL, and any index may be -1: "no row of that table for this entry" (the
  // row-sharded engine passes the user rows it received and the item rows its peers fetched as two
  // separate lists).  Any OTHER out-of-range index marks a triple the grad kernel flagged and skipped.
  auto load_idx = [&](int64_t t, int64_t& u, int64_t& a, int64_t& b) {
    u = a = b = -1;
    if (t < batch) {
      const int64_t j = perm ? perm[t] : t;
      u = users ? users[j] : -1;
      a = items_a ? items_a[j] : -1;
      b = items_b ? items_b[j] : a;
    }
  };
  int64_t u, a, b;
  load_idx(wave0, u, a, b);
  for (int64_t t = wave0; t < batch; t += n_waves) {
    int64_t nu, na, nb;
    load_idx(t + n_waves, nu, na, nb);  // next trip's indices travel during this trip
    const bool u_in = static_cast<uint64_t>(u) < static_cast<uint64_t>(w.n_users);
    const bool a_in = static_cast<uint64_t>(a) < static_cast<uint64_t>(w.n_items);
    const bool b_in = static_cast<uint64_t>(b) < static_cast<uint64_t>(w.n_items);
    // (a triple with an out-of-range id was flagged and skipped by the grad kernel too)
    const bool ok = (u_in || u == -1) && (a_in || a == -1) && (b_in || b == -1);
    int won = 0;
    if (ok) {
      if (lane == 0 && u_in) won = atomicExch(user_stamp + u, stamp) != stamp;
      else if (lane == 1 && a_in) won = atomicExch(item_stamp + a, stamp) != stamp;
      else if (lane == 2 && items_b && b_in && b != a) won = atomicExch(item_stamp + b, stamp) != stamp;
    }
    const bool won_r[3] = {__builtin_amdgcn_readlane(won, 0) != 0, __builtin_amdgcn_readlane(won, 1) != 0,
                           __builtin_amdgcn_readlane(won, 2) != 0};
    // rows this wave does not own point at row 0 (valid memory): their loads stay unconditional
    // and branch-free, only the stores are guarded
    const int64_t ru = won_r[0] ? u : 0, ra = won_r[1] ? a : 0, rb = won_r[2] ? b : 0;
    const RowPtrs rows[3] = {{w.user_emb + ru * D, g.user_emb + ru * D, w.user_bias + ru, g.user_bias + ru},
                             {w.item_emb + ra * D, g.item_emb + ra * D, w.item_bias + ra, g.item_bias + ra},
                             {w.item_emb + rb * D, g.item_emb + rb * D, w.item_bias + rb, g.item_bias + rb}};
    if constexpr (NPL > 0) {
      float wv[3][R], gv[3][R], wbv[3], gbv[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const int c = lane + kWave * k;
          const int cc = c < D ? c : D - 1;
          gv[q][k] = rows[q].g[cc];
          wv[q][k] = rows[q].w[cc];
        }
        gbv[q] = *rows[q].gb;
        wbv[q] = *rows[q].wb;
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (!won_r[q]) continue;
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const int c = lane + kWave * k;
          if (c < D) {
            rows[q].w[c] = wv[q][k] - lr * gv[q][k];
            rows[q].g[c] = 0.f;
          }
        }
        if (lane == 0) {
          *rows[q].wb = wbv[q] - lr * gbv[q];
          *rows[q].gb = 0.f;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (!won_r[q]) continue;
        for (int c = lane; c < D; c += kWave) {
          const float gvv = rows[q].g[c];
          rows[q].w[c] = rows[q].w[c] - lr * gvv;
          rows[q].g[c] = 0.f;
        }
        if (lane == 0) {
          const float gvv = *rows[q].gb;
          *rows[q].wb = *rows[q].wb - lr * gvv;
          *rows[q].gb = 0.f;
        }
      }
    }
    u = nu;
    a = na;
    b = nb;
  }
  if (blockIdx.x == 0) {
    const float gb_part = scratch ? finalize_partials(stats, scratch) : 0.f;
    if (threadIdx.x == 0) {
      const float gv = *g.global_bias + gb_part;
      *w.global_bias = *w.global_bias - lr * gv;
      *g.global_bias = 0.f;
    }
  }
}

static int check_tables(const hiprec_mf_tables* w, const char* name) {
  HIPREC_REQUIRE(w != nullptr, "%s is NULL", name);
  HIPREC_REQUIRE(w->user_emb && w->item_emb && w->user_bias && w->item_bias && w->global_bias,
                 "%s has a NULL tensor pointer", name);
  HIPREC_REQUIRE(w->n_users > 0 && w->n_items > 0 && w->dim > 0,
                 "%s has non-positive n_users/n_items/dim (%lld, %lld, %d)", name,
                 (long long)w->n_users, (long long)w->n_items, w->dim);
  return 0;
}

static int check_same_shape(const hiprec_mf_tables* w, const hiprec_mf_tables* g) {
  HIPREC_REQUIRE(w->n_users == g->n_users && w->n_items == g->n_items && w->dim == g->dim,
                 "weight and gradient tables differ in shape");
  return 0;
}

static int agg_grid(int64_t batch) {
  int64_t blocks = (batch + kAggWaves - 1) / kAggWaves;
  if (blocks < 1) blocks = 1;
  if (blocks > kAggMaxBlocks) blocks = kAggMaxBlocks;
  return static_cast<int>(blocks);
}

static size_t agg_lds_bytes(int dim) { return sizeof(float) * kAggWaves * (static_cast<size_t>(dim) + 1); }

template <typename Launch>
static int dispatch_npl(int dim, Launch&& launch) {
  if (dim <= 64) launch(std::integral_constant<int, 1>{});
  else if (dim <= 128) launch(std::integral_constant<int, 2>{});
  else if (dim <= 256) launch(std::integral_constant<int, 4>{});
  else launch(std::integral_constant<int, 0>{});
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
}

}  // namespace hiprec

using namespace hiprec;

extern "C" int hiprec_mf_bpr_grad(const hiprec_mf_tables* w, const hiprec_mf_tables* g,
                                  const int64_t* users, const int64_t* pos, const int64_t* neg,
                                  const int64_t* perm, int64_t batch, float inv_batch,
                                  float reg_coef, hiprec_stats* stats, void* scratch,
                                  size_t scratch_bytes, void* stream) {
  if (int rc = check_tables(w, "w")) return rc;
  if (int rc = check_tables(g, "g")) return rc;
  if (int rc = check_same_shape(w, g)) return rc;
  HIPREC_REQUIRE(users && pos && neg && stats && scratch, "NULL index/stats/scratch pointer");
  HIPREC_REQUIRE(batch >= 0, "negative batch");
  if (scratch_bytes < kScratchBytes) {
    set_error("scratch too small: %zu < %zu", scratch_bytes, kScratchBytes);
    return HIPREC_E_SCRATCH;
  }
  const int grid = agg_grid(batch);
  const size_t lds = agg_lds_bytes(w->dim);
  HIPREC_REQUIRE(lds <= 64 * 1024, "dim %d too large for the LDS merge slots", w->dim);
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto* sc = static_cast<Scratch*>(scratch);
  return dispatch_npl(w->dim, [&](auto npl) {
    mf_bpr_grad_kernel<decltype(npl)::value><<<grid, kAggBlock, lds, s>>>(
        *w, *g, users, pos, neg, perm, batch, inv_batch, reg_coef, stats, sc);
  });
}

extern "C" int hiprec_mf_bce_grad(const hiprec_mf_tables* w, const hiprec_mf_tables* g,
                                  const int64_t* users, const int64_t* items, const float* ratings,
                                  const int64_t* perm, int64_t batch, float inv_batch,
                                  float reg_coef, hiprec_stats* stats, void* scratch,
                                  size_t scratch_bytes, void* stream) {
  if (int rc = check_tables(w, "w")) return rc;
  if (int rc = check_tables(g, "g")) return rc;
  if (int rc = check_same_shape(w, g)) return rc;
  HIPREC_REQUIRE(users && items && ratings && stats && scratch,
                 "NULL index/ratings/stats/scratch pointer");
  HIPREC_REQUIRE(batch >= 0, "negative batch");
  if (scratch_bytes < kScratchBytes) {
    set_error("scratch too small: %zu < %zu", scratch_bytes, kScratchBytes);
    return HIPREC_E_SCRATCH;
  }
  const int grid = agg_grid(batch);
  const size_t lds = agg_lds_bytes(w->dim);
  HIPREC_REQUIRE(lds <= 64 * 1024, "dim %d too large for the LDS merge slots", w->dim);
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto* sc = static_cast<Scratch*>(scratch);
  return dispatch_npl(w->dim, [&](auto npl) {
    mf_bce_grad_kernel<decltype(npl)::value><<<grid, kAggBlock, lds, s>>>(
        *w, *g, users, items, ratings, perm, batch, inv_batch, reg_coef, stats, sc);
  });
}

extern "C" int hiprec_mf_predict(const hiprec_mf_tables* w, const int64_t* users,
                                 const int64_t* items, int64_t n, float* scores,
                                 hiprec_stats* stats, void* stream) {
  if (int rc = check_tables(w, "w")) return rc;
  HIPREC_REQUIRE(n >= 0, "negative n");
  if (n == 0) return 0;
  HIPREC_REQUIRE(users && items && scores && stats, "NULL pointer");
  mf_predict_kernel<<<grid_for_waves(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      *w, users, items, n, scores, nullptr, stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_mf_forward(const hiprec_mf_tables* w, const int64_t* users, const int64_t* items, int64_t n,
                                 float* scores, float* sq, hiprec_stats* stats, void* stream) {
  if (int rc = check_tables(w, "w")) return rc;
  HIPREC_REQUIRE(n >= 0, "negative n");
  if (n == 0) return 0;
  HIPREC_REQUIRE(users && items && scores && sq && stats, "NULL pointer");
  mf_predict_kernel<<<grid_for_waves(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      *w, users, items, n, scores, sq, stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_mf_sgd_rows(const hiprec_mf_tables* w, const hiprec_mf_tables* g,
                                  const int64_t* users, const int64_t* items_a,
                                  const int64_t* items_b, const int64_t* perm, int64_t batch,
                                  double lr, int32_t* user_stamp, int32_t* item_stamp,
                                  int32_t stamp, hiprec_stats* stats, const void* scratch,
                                  void* stream) {
  if (int rc = check_tables(w, "w")) return rc;
  if (int rc = check_tables(g, "g")) return rc;
  if (int rc = check_same_shape(w, g)) return rc;
  HIPREC_REQUIRE((users || items_a) && user_stamp && item_stamp && stats, "NULL pointer");
  HIPREC_REQUIRE(batch >= 0, "negative batch");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = grid_for_waves(batch);
  return dispatch_npl(w->dim, [&](auto npl) {
    mf_sgd_rows_kernel<decltype(npl)::value><<<grid, kBlock, 0, st>>>(
        *w, *g, users, items_a, items_b, perm, batch, static_cast<float>(lr), user_stamp, item_stamp, stamp,
        stats, static_cast<const Scratch*>(scratch));
  });
}

// One epoch of BPR-MF with ONE kernel per step (see mf_bpr_fused_kernel).
// w_flat[2] / g_flat[3] / scratch[2] (and m_flat[2] for Adam, v_flat[2] for Adam and RMSprop) are
// caller-owned; index 0 of w/m/v holds the state on entry, all g buffers and both scratch blocks are
// zero.  On return the state is in w_flat/m_flat/v_flat[*final_index] and every g buffer is zero
// again.  users/pos/neg are the epoch laid out in visiting order.
template <int KIND>
static int launch_fused(int dim, int grid, size_t lds, hipStream_t st, const FusedOpt& f, const int64_t* uu,
                        const int64_t* pp, const int64_t* nn, int64_t b, float inv_b, float reg_coef,
                        hiprec_stats* stats, Scratch* sc) {
  if (b == 0)  // flush: sweep blocks only, any NPL
    mf_bpr_fused_kernel<1, KIND, true><<<grid, kAggBlock, lds, st>>>(f, uu, pp, nn, b, inv_b, reg_coef, stats, sc);
  else if (dim <= 64)
    mf_bpr_fused_kernel<1, KIND, false><<<grid, kAggBlock, lds, st>>>(f, uu, pp, nn, b, inv_b, reg_coef, stats, sc);
  else if (dim <= 128)
    mf_bpr_fused_kernel<2, KIND, false><<<grid, kAggBlock, lds, st>>>(f, uu, pp, nn, b, inv_b, reg_coef, stats, sc);
  else
    mf_bpr_fused_kernel<4, KIND, false><<<grid, kAggBlock, lds, st>>>(f, uu, pp, nn, b, inv_b, reg_coef, stats, sc);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

// One launch of the fused sequence with explicitly named buffers (see hiprec_fused_step in hiprec.h).
extern "C" int hiprec_mf_bpr_fused_step(const hiprec_fused_step* c, const int64_t* users,
                                        const int64_t* pos, const int64_t* neg, int64_t batch,
                                        int64_t prev_batch, float inv_batch, hiprec_stats* stats,
                                        void* stream) {
  HIPREC_REQUIRE(c && stats, "NULL step / stats");
  const int kind = c->kind;
  HIPREC_REQUIRE(kind == HIPREC_OPT_SGD || kind == HIPREC_OPT_ADAM || kind == HIPREC_OPT_RMSPROP,
                 "unknown optimizer kind %d", kind);
  const bool has_m = kind == HIPREC_OPT_ADAM, has_v = kind != HIPREC_OPT_SGD;
  HIPREC_REQUIRE(c->w_read && c->w_write && c->g_prev && c->g_cur && c->g_zero && c->scratch_prev &&
                     c->scratch_cur, "NULL buffer");
  HIPREC_REQUIRE(!has_m || (c->m_read && c->m_write), "Adam needs m_read / m_write");
  HIPREC_REQUIRE(!has_v || (c->v_read && c->v_write), "Adam/RMSprop need v_read / v_write");
  HIPREC_REQUIRE(c->n_users > 0 && c->n_items > 0 && c->dim > 0 && c->dim <= 256, "fused step needs dim <= 256");
  HIPREC_REQUIRE(batch >= 0 && prev_batch >= 0, "negative batch");
  HIPREC_REQUIRE(batch == 0 || (users && pos && neg), "NULL index arrays");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t n_users = c->n_users, n_items = c->n_items;
  const int dim = c->dim;
  const int64_t n_flat = (n_users + n_items) * (static_cast<int64_t>(dim) + 1) + 1;
  FusedOpt f;
  f.n_users = n_users;
  f.n_items = n_items;
  f.dim = dim;
  f._pad0 = 0;
  f.g_cur = c->g_cur;
  f.w_read = c->w_read;
  f.g_prev = c->g_prev;
  f.m_read = has_m ? c->m_read : nullptr;
  f.v_read = has_v ? c->v_read : nullptr;
  f.w_write = c->w_write;
  f.m_write = has_m ? c->m_write : nullptr;
  f.v_write = has_v ? c->v_write : nullptr;
  f.g_zero = c->g_zero;
  f.g_zero2 = batch == 0 ? const_cast<float*>(c->g_prev) : nullptr;  // the flush clears what it applies
  f.n_flat = n_flat;
  f.s = OptScalars{c->lr,
                   static_cast<float>(c->lr),
                   static_cast<float>(c->beta2),
                   static_cast<float>(1.0 - c->beta1),
                   static_cast<float>(1.0 - c->beta2),
                   static_cast<float>(c->eps)};
  f.n_gather_blocks = batch > 0 ? std::min(agg_grid(batch), kFusedMaxGather) : 0;
  f.n_prev_partials = prev_batch > 0 ? std::min(agg_grid(prev_batch), kFusedMaxGather) : 0;
  f.apply_prev = prev_batch > 0 ? 1 : 0;
  f.scratch_prev = static_cast<const Scratch*>(c->scratch_prev);
  Scratch* sc = static_cast<Scratch*>(c->scratch_cur);
  const size_t lds = agg_lds_bytes(dim);
  const int n_sweep = static_cast<int>(std::min<int64_t>(256, (n_flat / 4 + kAggBlock - 1) / kAggBlock));
  const int grid = f.n_gather_blocks + n_sweep;
  int rc;
  if (kind == HIPREC_OPT_SGD)
    rc = launch_fused<HIPREC_OPT_SGD>(dim, grid, lds, st, f, users, pos, neg, batch, inv_batch, c->reg_coef, stats, sc);
  else if (kind == HIPREC_OPT_ADAM)
    rc = launch_fused<HIPREC_OPT_ADAM>(dim, grid, lds, st, f, users, pos, neg, batch, inv_batch, c->reg_coef, stats, sc);
  else
    rc = launch_fused<HIPREC_OPT_RMSPROP>(dim, grid, lds, st, f, users, pos, neg, batch, inv_batch, c->reg_coef, stats, sc);
  return rc;
}

extern "C" size_t hiprec_fused_step_bytes(void) { return sizeof(hiprec_fused_step); }

extern "C" int hiprec_mf_bpr_epoch_fused_range(int kind, float* const* w_flat, float* const* g_flat,
                                               float* const* m_flat, float* const* v_flat,
                                               void* const* scratch2, int64_t n_users, int64_t n_items,
                                               int32_t dim, const int64_t* users, const int64_t* pos,
                                               const int64_t* neg, int64_t n_triples, int64_t batch,
                                               int64_t step_begin, int64_t step_end,
                                               float reg_coef, double lr, double beta1, double beta2,
                                               double eps, hiprec_stats* stats, int32_t* final_index,
                                               void* stream) {
  HIPREC_REQUIRE(w_flat && g_flat && scratch2 && final_index && stats, "NULL pointer");
  HIPREC_REQUIRE(w_flat[0] && w_flat[1] && g_flat[0] && g_flat[1] && g_flat[2] && scratch2[0] &&
                     scratch2[1], "NULL buffer");
  const bool has_m = kind == HIPREC_OPT_ADAM, has_v = kind != HIPREC_OPT_SGD;
  HIPREC_REQUIRE(!has_m || (m_flat && m_flat[0] && m_flat[1]), "Adam needs m_flat[2]");
  HIPREC_REQUIRE(!has_v || (v_flat && v_flat[0] && v_flat[1]), "Adam/RMSprop need v_flat[2]");
  HIPREC_REQUIRE(n_triples >= 0 && batch > 0, "bad n_triples/batch");
  HIPREC_REQUIRE(n_triples == 0 || (users && pos && neg), "NULL index arrays");
  const int64_t n_steps = (n_triples + batch - 1) / batch;
  HIPREC_REQUIRE(0 <= step_begin && step_begin <= step_end && step_end <= n_steps,
                 "bad step range [%lld, %lld) of %lld", (long long)step_begin, (long long)step_end, (long long)n_steps);
  // steps [step_begin, step_end) of the epoch; the call that reaches the epoch's last step also enqueues
  // the sweep-only flush (launch index n_steps).  Until then the state is mid-rotation: W / M / V of step k
  // live in buffer (k & 1) with the update of step k-1 still pending in g_flat[(k + 2) % 3].
  const int64_t k_end = step_end == n_steps ? n_steps + 1 : step_end;
  hiprec_fused_step c{};
  c.kind = kind;
  c.dim = dim;
  c.n_users = n_users;
  c.n_items = n_items;
  c.lr = lr;
  c.beta1 = beta1;
  c.beta2 = beta2;
  c.eps = eps;
  c.reg_coef = reg_coef;
  for (int64_t k = step_begin; k < k_end; ++k) {  // launch n_steps is the sweep-only flush
    const int64_t off = k * batch;
    const int64_t b = k < n_steps ? std::min<int64_t>(batch, n_triples - off) : 0;
    const int64_t prev_b = k > 0 ? std::min<int64_t>(batch, n_triples - (k - 1) * batch) : 0;
    // the flush (no gather blocks, one thread per element) may update in place: it always lands
    // in buffer 0, so nothing has to be copied back whatever the parity of the epoch
    const int out = k == n_steps ? 0 : static_cast<int>((k + 1) & 1);
    c.w_read = w_flat[k & 1];
    c.w_write = w_flat[out];
    c.m_read = has_m ? m_flat[k & 1] : nullptr;
    c.m_write = has_m ? m_flat[out] : nullptr;
    c.v_read = has_v ? v_flat[k & 1] : nullptr;
    c.v_write = has_v ? v_flat[out] : nullptr;
    c.g_prev = g_flat[(k + 2) % 3];
    c.g_cur = g_flat[k % 3];
    c.g_zero = g_flat[(k + 1) % 3];
    c.scratch_prev = scratch2[(k + 1) & 1];
    c.scratch_cur = scratch2[k & 1];
    const float inv_b = b > 0 ? 1.0f / static_cast<float>(b) : 0.f;
    if (int rc = hiprec_mf_bpr_fused_step(&c, users ? users + off : nullptr, pos ? pos + off : nullptr,
                                          neg ? neg + off : nullptr, b, prev_b, inv_b, stats, stream))
      return rc;
  }
  // the flush cleared the gradient it applied and both scratch headers, and wrote into buffer 0;
  // a range that stops short of the epoch's end leaves the state in rotation (-1)
  *final_index = step_end == n_steps ? 0 : -1;
  return 0;
}

extern "C" int hiprec_mf_bpr_epoch_fused(int kind, float* const* w_flat, float* const* g_flat,
                                         float* const* m_flat, float* const* v_flat,
                                         void* const* scratch2, int64_t n_users, int64_t n_items,
                                         int32_t dim, const int64_t* users, const int64_t* pos,
                                         const int64_t* neg, int64_t n_triples, int64_t batch,
                                         float reg_coef, double lr, double beta1, double beta2,
                                         double eps, hiprec_stats* stats, int32_t* final_index,
                                         void* stream) {
  const int64_t n_steps = batch > 0 && n_triples >= 0 ? (n_triples + batch - 1) / batch : 0;
  return hiprec_mf_bpr_epoch_fused_range(kind, w_flat, g_flat, m_flat, v_flat, scratch2, n_users, n_items, dim,
                                         users, pos, neg, n_triples, batch, 0, n_steps, reg_coef, lr, beta1,
                                         beta2, eps, stats, final_index, stream);
}

// Steps [step_begin, step_end) of a DATA-PARALLEL fused epoch (beta-recsys_amd/replicated.py): every step is the
// fused launch on this rank's batch followed by ONE in-place sum all-reduce of [loss partials | gradient], both
// enqueued from here -- no host code between them (through torch.distributed the step was host-bound: 20-26 us
// at world size 1 for an 11.5 us kernel).  The collective is the caller's: the address of ncclAllReduce (RCCL) and
// a communicator; this library does not link RCCL.
using nccl_all_reduce_fn = int (*)(const void*, void*, size_t, int, int, void*, hipStream_t);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;  // ncclDataType_t / ncclRedOp_t values (nccl.h, rccl.h)

extern "C" int hiprec_mf_bpr_dp_epoch_fused_range(int kind, float* const* w_flat, float* const* bufs,
                                                  int64_t scratch_floats, float* const* m_flat,
                                                  float* const* v_flat, int64_t n_flat, int64_t n_users,
                                                  int64_t n_items, int32_t dim, const int64_t* users,
                                                  const int64_t* pos, const int64_t* neg, int64_t n_triples,
                                                  int64_t batch, int64_t step_begin, int64_t step_end,
                                                  int32_t world, float reg_coef, double lr, double beta1,
                                                  double beta2, double eps, hiprec_stats* stats,
                                                  void* all_reduce_fn, void* comm, int32_t* final_index,
                                                  void* stream) {
  HIPREC_REQUIRE(w_flat && bufs && final_index && stats && all_reduce_fn && comm, "NULL pointer");
  HIPREC_REQUIRE(w_flat[0] && w_flat[1] && bufs[0] && bufs[1] && bufs[2], "NULL buffer");
  const bool has_m = kind == HIPREC_OPT_ADAM, has_v = kind != HIPREC_OPT_SGD;
  HIPREC_REQUIRE(!has_m || (m_flat && m_flat[0] && m_flat[1]), "Adam needs m_flat[2]");
  HIPREC_REQUIRE(!has_v || (v_flat && v_flat[0] && v_flat[1]), "Adam/RMSprop need v_flat[2]");
  HIPREC_REQUIRE(n_triples >= 0 && batch > 0 && world >= 1 && scratch_floats >= 4 && n_flat > 0, "bad sizes");
  HIPREC_REQUIRE(n_triples == 0 || (users && pos && neg), "NULL index arrays");
  const int64_t n_steps = (n_triples + batch - 1) / batch;
  HIPREC_REQUIRE(0 <= step_begin && step_begin <= step_end && step_end <= n_steps,
                 "bad step range [%lld, %lld) of %lld", (long long)step_begin, (long long)step_end, (long long)n_steps);
  const auto all_reduce = reinterpret_cast<nccl_all_reduce_fn>(all_reduce_fn);
  const int64_t k_end = step_end == n_steps ? n_steps + 1 : step_end;  // launch n_steps is the sweep-only flush
  hiprec_fused_step c{};
  c.kind = kind;
  c.dim = dim;
  c.n_users = n_users;
  c.n_items = n_items;
  c.lr = lr;
  c.beta1 = beta1;
  c.beta2 = beta2;
  c.eps = eps;
  c.reg_coef = reg_coef;
  for (int64_t k = step_begin; k < k_end; ++k) {
    const int64_t off = k * batch;
    const int64_t b = k < n_steps ? std::min<int64_t>(batch, n_triples - off) : 0;
    const int64_t prev_b = k > 0 ? std::min<int64_t>(batch, n_triples - (k - 1) * batch) : 0;
    const int out = k == n_steps ? 0 : static_cast<int>((k + 1) & 1);
    float* prev = bufs[(k + 2) % 3];  // each buffer: [scratch block | gradient]
    float* cur = bufs[k % 3];
    float* nxt = bufs[(k + 1) % 3];
    c.w_read = w_flat[k & 1];
    c.w_write = w_flat[out];
    c.m_read = has_m ? m_flat[k & 1] : nullptr;
    c.m_write = has_m ? m_flat[out] : nullptr;
    c.v_read = has_v ? v_flat[k & 1] : nullptr;
    c.v_write = has_v ? v_flat[out] : nullptr;
    c.scratch_prev = prev;
    c.g_prev = prev + scratch_floats;
    c.scratch_cur = cur;
    c.g_cur = cur + scratch_floats;
    c.g_zero = nxt + scratch_floats;
    // the gradient is the GLOBAL batch's mean: every rank scales by 1 / (its batch * world)
    const float inv_b = b > 0 ? 1.0f / (static_cast<float>(b) * static_cast<float>(world)) : 0.f;
    if (int rc = hiprec_mf_bpr_fused_step(&c, b ? users + off : nullptr, b ? pos + off : nullptr,
                                          b ? neg + off : nullptr, b, prev_b, inv_b, stats, stream))
      return rc;
    if (k < n_steps) {  // everything behind the 4-float scratch header: the loss partials and the gradient
      const int rc = all_reduce(cur + 4, cur + 4, static_cast<size_t>(scratch_floats - 4 + n_flat), kNcclFloat32,
                                kNcclSum, comm, static_cast<hipStream_t>(stream));
      if (rc != 0) {
        set_error("the caller's all-reduce returned %d at step %lld", rc, (long long)k);
        return HIPREC_E_UNSUPPORTED;
      }
    }
  }
  *final_index = step_end == n_steps ? 0 : -1;
  return 0;
}

// The plain-SGD spelling of the above, kept for callers of the first ABI revision.
extern "C" int hiprec_mf_bpr_epoch_sgd_fused(float* const* w_flat, float* const* g_flat,
                                             void* const* scratch2, int64_t n_users,
                                             int64_t n_items, int32_t dim, const int64_t* users,
                                             const int64_t* pos, const int64_t* neg,
                                             int64_t n_triples, int64_t batch, float reg_coef,
                                             double lr, hiprec_stats* stats, int32_t* final_index,
                                             void* stream) {
  return hiprec_mf_bpr_epoch_fused(HIPREC_OPT_SGD, w_flat, g_flat, nullptr, nullptr, scratch2, n_users,
                                   n_items, dim, users, pos, neg, n_triples, batch, reg_coef, lr, 0.0,
                                   0.0, 0.0, stats, final_index, stream);
}
