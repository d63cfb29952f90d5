// Exact lazy Adam / RMSprop for embedding tables that live in HBM (BASELINE configs[3]; SURVEY.md "Hard parts":
// "exact lazy catch-up: replay k missed zero-grad steps per row on next touch -- needs per-row last-step stamps and a
// flush before predict / state_dict").
//
// Reference semantics (beta_rec/models/torch_engine.py:30-39): nn.Embedding is dense, so torch.optim.Adam / RMSprop
// step EVERY element EVERY step, with a zero gradient for rows the batch did not touch.  The dense sweep
// (csrc/optim.hip) does exactly that and moves 28 bytes per parameter per step: 39.7 GB per step at configs[3],
// 195 x the bytes of the rows a step touches.  A zero-gradient step of a row depends on nothing but the row's own
// (w, m, v) and the step number, so it can be postponed until the row is needed again and then replayed -- the SAME
// fp32 operations in the SAME order (opt_update<KIND> with g = 0 and the step's own bias-correction scalars), hence
// bit-identical to the sweep:
//   * stamp[row] (int32 per table row) = the step the row is current as of; -1 = never touched (m = v = 0: every
//     zero-gradient step is the identity on it);
//   * scalars[t] = (lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t) -- the v_rcp_f32 the sweep of step t takes of it; the value
//     itself in the correctly rounded build, which divides) as step t derived them from hiprec_stats: the update kernel
//     of step t records them, the replays read them;
//   * catch-up (before a step reads its rows): rows of the step that lag behind are replayed up to the last completed
//     step; only w is stored (what the step reads) and the stamp is flagged "w ahead" -- the moments are replayed
//     again (one fma + one multiply per step) by the same step's update instead of being written and re-read.  Adam
//     only: RMSprop's zero-gradient step leaves w alone (w += -lr * 0 / avg) and only decays v;
//   * update (after the step's gradients are complete in the dense gradient buffer): every row of the step takes
//     its real step -- after replaying what is still behind --, its gradient row is cleared, stamp = t;
//   * flush: every lagging row of the tables is replayed up to the clock (before predict / state_dict / checkpoint /
//     any dense sweep).
// The lists name a step's rows with duplicates (a user that occurs twice, an item several peers ask for); whoever
// changes the row's stamp first (atomicOr of the "w ahead" flag in the catch-up, atomicExch of the step number in the
// update) owns it for that launch, the others skip; an entry that repeats its predecessor is not even looked at.
// dim % 4 == 0: 64 / LPR rows per wave, LPR lanes x one float4 per row and array, the row's bias element in the row's
// first lane, row loads issued before the claim (lazy_rows_vec_kernel).  Other widths: one row per wave, lane = column
// (+ kWave * j) (lazy_rows_kernel).
#include <algorithm>
#include <cmath>

#include "common.hpp"
#include "pull.hpp"

namespace hiprec {
namespace {

constexpr int kLazyMaxNpl = 4;  // dim <= 256
// stamp bit 30: "w (only) has been caught up to the clock by this step's catch-up; m, v are as of the stamp".  The
// state lasts from a step's catch-up to the same step's update, which every caught-up row goes through.
constexpr int kWAhead = 1 << 30;

struct LazyCtx {
  float* w;
  float* g;
  float* m;
  float* v;
  int64_t n_users, n_items;
  int32_t dim;
  int32_t* stamp_u;
  int32_t* stamp_i;
  float2* scalars;
  int32_t scalars_cap;
  // Adam: from this step number on a weight that a zero-gradient step left unchanged stays unchanged for good
  // (lazy_freeze_from below); INT_MAX = never assume so
  int32_t freeze_from;
  float freeze_wmin;   // ... for weights of at least this magnitude (guard (ii) below)
};

// BOUNDED REPLAY (round 5).  A zero-gradient Adam step k of a row moves w by x_k = ss_t * m_k / (sqrt(v_k) * r_t + eps)
// with m_k = (1 - (1 - beta1)) m_(k-1), v_k = beta2 v_(k-1): |x_(k+1)| / |x_k| <= beta1 / sqrt(beta2) *
// sqrt((1 - beta2^(t+1)) / (1 - beta2^t)) (eps in the denominator only helps, ss_t falls with t): 0.93 at t = 16 for
// the default betas, 0.90 in the limit.  The x_k of an element all have the sign of its m.  So once fl(w - x_k) == w -- x_k is at most
// half the gap to w's neighbour on that side -- every later x is smaller still and leaves w alone as well: the
// remaining zero-gradient steps of that element are decays of m and v only (one fma + one multiply instead of sqrt +
// rcp + 6 more), and a catch-up, which stores nothing but w, is DONE.  With lr 0.05 that happens ~150-200 steps after
// the last gradient: the replay of a row costs at most that many full steps however long it lagged -- exactly, no
// closed form.  Guards: (i) the step number is past lazy_freeze_from(beta1, beta2) (INT_MAX for betas whose ratio above
// is not safely below one, or without an eps to bound the denominator from below); (ii) |w| >= freeze_wmin or m == 0: a
// first moment that has decayed into the smallest denormals stops shrinking (0.9 x rounds back to x for x <= 4 units)
// while the denominator still falls, so x can GROW again -- up to 1.25 lr * 8 * 1.4e-45 / eps, which is below half an ulp
// of every |w| >= 2^26 times that (2.5e-30 for lr 0.05, eps 1e-8); an element with m == 0 does not move at all.
int lazy_freeze_from(double b1, double b2, double lr, double eps, float* wmin) {
  constexpr int kNever = 0x7fffffff;
  *wmin = static_cast<float>(std::max(1e-30, std::ldexp(1.0, 26) * 1.25 * lr * 8.0 * 1.4012984643e-45 / std::max(eps, 1e-300)));
  if (!(lr > 0.0 && eps >= 1e-12 && *wmin < 1e-20f)) return kNever;
  if (!(b1 > 0.0 && b1 < 1.0 && b2 > 0.0 && b2 < 1.0)) return kNever;
  if (b1 * b1 >= 0.96 * b2) return kNever;
  double p = b2;   // beta2^t
  for (int t = 1; t < (1 << 20); ++t) {
    const double pn = p * b2;
    if (b1 * b1 * (1.0 - pn) < 0.96 * b2 * (1.0 - p)) return std::max(t, 16);
    p = pn;
  }
  return kNever;
}
constexpr int kFreezeEvery = 8;   // the test costs a compare per element and a ballot: taken every 8th step ...
constexpr int kFreezeStart = 192; // ... and not before a walk is this long (weights move for 150-200 steps at lr 0.05)

__device__ __forceinline__ float2 lazy_scalars_at(const LazyCtx& c, long long t) {
  return c.scalars[t < c.scalars_cap ? t : c.scalars_cap - 1];
}

// The model's scalar (global_bias, the last element) takes a real step every step; its gradient is in g (the row-sharded
// step's bookkeeping put it there) plus, for callers that pass the gradient kernel's scratch, the per-block partials
// (which also books the step's loss, as the dense sweep does).  Block 0 of an update launch.
template <int KIND, int NT = kBlock>
__device__ __forceinline__ void lazy_scalar_step(const LazyCtx& c, const OptScalars& s, hiprec_stats* stats,
                                                 const Scratch* scratch, long long clock, float ss_now, float bc2_now,
                                                 int bid = blockIdx.x) {
  if (bid != 0) return;
  float extra = 0.f;
  if (scratch) extra = finalize_partials<NT>(stats, scratch);
  if (threadIdx.x != 0) return;
  const int64_t i = (c.n_users + c.n_items) * (static_cast<int64_t>(c.dim) + 1);
  float wv = c.w[i], gv = c.g[i] + extra, mv = 0.f, vv = c.v[i];
  if constexpr (KIND == HIPREC_OPT_ADAM) mv = c.m[i];
  opt_update<KIND>(wv, gv, mv, vv, s, ss_now, bc2_now);
  c.w[i] = wv;
  c.g[i] = 0.f;
  if constexpr (KIND == HIPREC_OPT_ADAM) c.m[i] = mv;
  c.v[i] = vv;
  // this step's scalars for the replays to come.  Beyond the table the powers must have converged (default betas:
  // beta2^t < 2^-53 from t ~ 36 800 on): a later step with different scalars cannot be replayed.
  // (the second scalar as the replay's opt_update<KIND, true> wants it: its v_rcp_f32, or itself where that build divides)
#ifdef HIPREC_IEEE_DIV
  const float bc2_rec = bc2_now;
#else
  const float bc2_rec = __builtin_amdgcn_rcpf(bc2_now);
#endif
  if (clock < c.scalars_cap) {
    c.scalars[clock] = make_float2(ss_now, bc2_rec);
  } else {
    const float2 last = c.scalars[c.scalars_cap - 1];
    if (last.x != ss_now || last.y != bc2_rec) atomicOr(&stats->status, HIPREC_STATUS_LAZY_TABLE);
  }
}

// One entry of the step's lists (or, flush, row e of the tables): which table, which row; id < 0 = nothing.
// An entry that names the same row as the entry before it in its list is skipped (-1): the batcher groups every batch
// by positive item, so the duplicates of a Zipf-popular item are neighbours and all but the first would only load
// rows to lose the claim.
template <int MODE>
__device__ __forceinline__ int64_t lazy_entry(const LazyCtx& c, const hiprec_lazy_rows& rows, int64_t e, bool* is_item) {
  const int64_t n0 = MODE >= 2 ? c.n_users : rows.n_users, n1 = MODE >= 2 ? c.n_items : rows.n_items_a;
  const int64_t n2 = MODE >= 2 ? 0 : rows.n_items_b;
  *is_item = e >= n0;
  if constexpr (MODE >= 2) return e < n0 ? e : e - n0;
  auto at = [&](auto* list, int64_t k) -> int64_t {
    const int64_t id = list[k];
    return k > 0 && static_cast<int64_t>(list[k - 1]) == id ? -1 : id;
  };
  if (e < n0) return at(rows.users, e);
  if (e < n0 + n1) return at(rows.items_a, e - n0);
  if (e < n0 + n1 + n2) return at(rows.items_b, e - n0 - n1);
  return at(rows.items_c, e - n0 - n1 - n2);
}

// Who works on a row (one lane per row calls this): `old` = the stamp found, returns whether the caller owns the row
// for this launch.
// MODE 2 / 3 = the two launches of a flush (vector kernels, Adam): 2 takes the rows that lag by at most max_gap steps
// and runs the plain walk only, 3 takes whatever still lags afterwards with the bounded replay compiled in (its rows'
// bias elements are fetched after the claim: on most flushes it claims nothing and only reads the stamps).
template <int MODE>
__device__ __forceinline__ bool lazy_claim(int32_t* sp, int target, int* old_out, int max_gap = 0x7fffffff) {
  int old = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool go = false;
  if constexpr (MODE == 0) {   // raise the W_AHEAD flag: the first to set it owns the row
    if (old >= 0 && !(old & kWAhead) && old < target) {
      old = atomicOr(sp, kWAhead);
      go = !(old & kWAhead);
    }
  } else if constexpr (MODE == 1) {   // the first to write the step's number owns the row
    if (old != target) {
      old = atomicExch(sp, target);
      go = old != target;
    }
  } else {                     // flush: one visitor per row; a never-touched row stays -1
    go = old >= 0 && (old & ~kWAhead) < target && target - (old & ~kWAhead) <= max_gap;
    if (go) *sp = target;
  }
  *old_out = old;
  return go;
}

// dim % 4 == 0: a workgroup takes 64 entries at a time.
//   phase A (wave 0, lane = entry): the list entry, the claim on its stamp and the row's BIAS element (one element per
//     lane: its loads are issued before the claim, its replay runs with a per-lane start) -- the rows' embeddings then
//     have exactly dim floats and no fifth slot that only one lane in LPR would use;
//   the entries that won their claim are ordered by stamp, oldest first (a rank by counting, 64 v_readlane), and the
//     workgroup's four waves deal them out ROWS = 64 / LPR at a time: the rows a wave replays together lag by about the
//     same number of steps (a wave walks from the OLDEST of its rows: with rows in list order E[max of 2 gaps] is 1.5 x
//     the mean gap, of 4 gaps 2.1 x);
//   phase B (every wave): LPR lanes x one float4 per row and array, the next group's loads in flight while this one is
//     replayed (two register sets, unconditional loads so that the wait counts stay exact), the per-step scalars of up
//     to 64 steps fetched with ONE load (lane k holds step first + k) and handed out with v_readlane.
template <int KIND, int MODE, int LPR>
__device__ __forceinline__ void lazy_rows_vec_body(const LazyCtx& c, const hiprec_lazy_rows& rows, const OptScalars& s,
                                                   hiprec_stats* stats, const Scratch* scratch, int bid, int n_blocks) {
  constexpr int ROWS = kWave / LPR;
  constexpr bool kHasW = KIND == HIPREC_OPT_ADAM || MODE == 1;
  constexpr bool kAdam = KIND == HIPREC_OPT_ADAM;
  __shared__ long long s_id[kWave];
  __shared__ int s_old[kWave];
  __shared__ int s_order[kWave];   // sorted position -> entry (phase A's lane), bit 8 = item table
  __shared__ int s_active;
  const int lane = lane_id(), wv = wave_in_block(), sub = lane / LPR, sl = lane % LPR;
  const int D = c.dim;
  const long long clock = stats->step;
  float ss_now = s.lr, bc2_now = 1.f;
  if constexpr (MODE == 1) step_scalars<KIND>(s, stats, &ss_now, &bc2_now);
  const int64_t nu = c.n_users, ni = c.n_items;
  const int64_t total = MODE >= 2 ? nu + ni : rows.n_users + rows.n_items_a + rows.n_items_b + rows.n_items_c;
  const int64_t n_chunks = (total + kWave - 1) / kWave;
  const int target = static_cast<int>(clock);
  const long long last = MODE == 1 ? clock - 1 : clock;
  const int64_t bias0 = (nu + ni) * static_cast<int64_t>(D);

  // The zero-gradient steps (base, last] of this lane's N elements; `on` = this lane has any, `first` (uniform) = the
  // oldest stamp of the wave's lanes that are on, + 1: the wave walks from there in blocks of 64 steps (lane k of
  // `mine` holds the scalars of step tb + k), a lane joins when t passes its own stamp.
  // The plain block: every zero-gradient step in full (or, for rows a catch-up already moved, the moments alone).  The
  // first kFreezeStart steps of every walk take it -- the common case, a row that lags a few dozen steps, runs the
  // instruction stream it always ran.
  auto replay_block_plain = [&](auto& w, auto& m, auto& v, bool on, int base, bool w_ahead, long long tb, float2 mine)
                                __attribute__((always_inline)) {
    constexpr int N = sizeof(w) / sizeof(w[0]);
    const int cnt = static_cast<int>(last - tb + 1 < kWave ? last - tb + 1 : kWave);
    for (int k = 0; k < cnt; ++k) {
      float2 sc = make_float2(s.lr, 1.f);
      if constexpr (kAdam) {
        sc.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.x), k));
        sc.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.y), k));
      }
      if (!(on && tb + k > base)) continue;
      // w is current already: only the moments (one fma and one multiply per step).  A flush meets the flag only when
      // a step was abandoned between its catch-up and its update (an error return in between): w must not take the
      // zero-gradient steps a second time
      if (MODE != 0 && w_ahead) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
          float zero = 0.f, w_unused = 0.f;
          opt_update<KIND>(w_unused, zero, m[j], v[j], s, 1.f, 1.f);
        }
      } else {
#pragma unroll
        for (int j = 0; j < N; ++j) {
          float zero = 0.f;
          opt_update<KIND, true>(w[j], zero, m[j], v[j], s, sc.x, sc.y);
        }
      }
    }
  };
  // The checked block (walks longer than kFreezeStart steps): the plain block plus, every kFreezeEvery-th step, the
  // test of BOUNDED REPLAY.  Returns the index of the step after which no weight of the wave moves any more, -1 if
  // there is none in this block.  (Kept apart from the moments-only loop below: folded into one loop the compiler
  // predicates both paths and the frozen steps cost what the full ones do -- measured, call 9.)
  auto replay_block_checked = [&](auto& w, auto& m, auto& v, bool on, int base, bool w_ahead, long long tb, float2 mine)
                                  __attribute__((always_inline)) -> int {
    constexpr int N = sizeof(w) / sizeof(w[0]);
    const int cnt = static_cast<int>(last - tb + 1 < kWave ? last - tb + 1 : kWave);
    for (int k = 0; k < cnt; ++k) {
      const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.x), k));
      const float sy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.y), k));
      const bool act = on && tb + k > base;
      const bool check = (k % kFreezeEvery) == kFreezeEvery - 1 && tb + k >= c.freeze_from;
      bool moved = false;
      if (act) {
        if (MODE != 0 && w_ahead) {
#pragma unroll
          for (int j = 0; j < N; ++j) {
            float zero = 0.f, w_unused = 0.f;
            opt_update<KIND>(w_unused, zero, m[j], v[j], s, 1.f, 1.f);
          }
        } else {
#pragma unroll
          for (int j = 0; j < N; ++j) {
            float zero = 0.f;
            const float w0 = w[j];
            opt_update<KIND, true>(w[j], zero, m[j], v[j], s, sx, sy);
            if (check) moved |= w[j] != w0 || !(__builtin_fabsf(w0) >= c.freeze_wmin || m[j] == 0.f);
          }
        }
      }
      // a lane holds the wave back while it has yet to join the walk, or while a weight of its still moves
      if (check && __ballot(on && !(MODE != 0 && w_ahead) && (!act || moved)) == 0ull) return k;
    }
    return -1;
  };
  // n_lane zero-gradient steps of the MOMENTS alone (one fma + one multiply per element and step, no per-step scalars)
  auto decay_only = [&](auto& m, auto& v, int n_lane) __attribute__((always_inline)) {
    constexpr int N = sizeof(m) / sizeof(m[0]);
    int n_max = n_lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_max = max(n_max, __shfl_xor(n_max, o));
    n_max = __builtin_amdgcn_readfirstlane(n_max);
    for (int i = 0; i < n_max; ++i) {
      if (i >= n_lane) continue;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        float zero = 0.f, w_unused = 0.f;
        opt_update<KIND>(w_unused, zero, m[j], v[j], s, 1.f, 1.f);
      }
    }
  };
  // `pre` = the scalars of steps pre_tb + lane, fetched by the caller ahead of time.  The walk starts at pre_tb unless
  // the oldest row turned out to have nothing to replay (all its moments zero); that path and the blocks after the
  // first fetch their scalars themselves -- kept apart so that the common path has no load between its waits.
  auto replay = [&](auto& w, auto& m, auto& v, bool on, int base, bool w_ahead, int first, float2 pre, int pre_tb)
                    __attribute__((always_inline)) {
    if (first > last) return;
    if constexpr (!kAdam) {   // RMSprop's zero-gradient step leaves w alone: v decays, that is all
      decay_only(m, v, on ? static_cast<int>(last - base) : 0);
      return;
    }
    long long tb = first;
    if (first == pre_tb) {
      replay_block_plain(w, m, v, on, base, w_ahead, tb, pre);
      tb += kWave;
    }
    for (; tb <= last; tb += kWave) {
      const float2 mine = lazy_scalars_at(c, tb + lane);
      // (an Adam update launch follows a catch-up: its rows are w-ahead, their replay is the moments' anyway; the first
      // launch of a flush takes only rows whose walk is shorter than kFreezeStart)
      if (MODE == 1 || MODE == 2 || tb - first < kFreezeStart) {
        replay_block_plain(w, m, v, on, base, w_ahead, tb, mine);
        continue;
      }
      const int kz = replay_block_checked(w, m, v, on, base, w_ahead, tb, mine);
      if (kz < 0) continue;
      // no weight of the wave moves after step tb + kz, and every lane has joined the walk by then: a catch-up (which
      // stores w only) is done, the others let the moments decay over the steps that are left
      // (a w-ahead lane does not hold the wave back -- its weights are current -- so it may not have joined the walk at
      // step tb + kz: its moments then decay over the steps IT lagged, last - base, not over last - (tb + kz): ADVICE r5)
      if constexpr (MODE != 0)
        decay_only(m, v, on ? static_cast<int>(last - (tb + kz > base ? tb + kz : static_cast<long long>(base))) : 0);
      return;
    }
  };

  struct Group {
    int64_t emb;
    int base, first;   // first (uniform): the oldest stamp of the group's rows + 1
    bool col_on, w_ahead;
    float2 sc;         // the scalars of steps first + lane
    float w[4], m[4], v[4], g[4];
  };

  for (int64_t ch = bid; ch < n_chunks; ch += n_blocks) {
    // ---- phase A, part 1: entries, claims, order
    int64_t bat = bias0;
    float bw[1] = {0.f}, bm[1] = {0.f}, bv[1] = {0.f}, bg[1] = {0.f};
    int old = 0;
    bool go = false;
    if (wv == 0) {
      const int64_t e = ch * kWave + lane;
      bool is_item = true;
      int64_t id = e < total ? lazy_entry<MODE>(c, rows, e, &is_item) : -1;
      if (id >= (is_item ? ni : nu)) {
        atomicOr(&stats->status, HIPREC_STATUS_ROW_OOB);
        id = -1;
      }
      const bool ent = id >= 0;
      bat = bias0 + (is_item ? nu : 0) + (ent ? id : 0);
      // the bias element, issued before the claim (a lost claim wastes the loads, a won one has them in hand)
      if constexpr (MODE != 3) {
        if (kHasW) bw[0] = c.w[bat];
        if constexpr (kAdam) bm[0] = c.m[bat];
        bv[0] = c.v[bat];
        if constexpr (MODE == 1) bg[0] = c.g[bat];
      }
      if (ent) go = lazy_claim<MODE>((is_item ? c.stamp_i : c.stamp_u) + id, target, &old,
                                     MODE == 2 && kAdam ? kFreezeStart : 0x7fffffff);
      if (MODE != 1 && old < 0) go = false;   // never touched: m = v = 0, nothing to replay
      if constexpr (MODE == 3) {
        if (go) {
          bw[0] = c.w[bat];
          if constexpr (kAdam) bm[0] = c.m[bat];
          bv[0] = c.v[bat];
        }
      }
      const int n_act = __popcll(__ballot(go));
      if (n_act > 0) {
        const int key = !go ? 0x7fffffff : (old < 0 ? 0x7ffffffe : (old & ~kWAhead));
        int rank = 0;
        for (int j = 0; j < kWave; ++j) {
          const int kj = __builtin_amdgcn_readlane(key, j);
          rank += (kj < key || (kj == key && j < lane)) ? 1 : 0;
        }
        s_order[rank] = lane | (is_item ? 256 : 0);
        s_id[lane] = id;
        s_old[lane] = old;
      }
      if (lane == 0) s_active = n_act;
    }
    __syncthreads();
    const int n_active = s_active;
    if (n_active > 0) {
      // ---- phase A, part 2: the bias elements (wave 0; the other waves are already at their rows)
      if (wv == 0) {
        const bool w_ahead = old >= 0 && (old & kWAhead);
        const int base = old < 0 ? -1 : (old & ~kWAhead);
        const bool on = go && base >= 0 && (bm[0] != 0.f || bv[0] != 0.f);
        int first = on ? base + 1 : 0x7fffffff;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off));
        first = __builtin_amdgcn_readfirstlane(first);
        replay(bw, bm, bv, on, base, w_ahead, first, make_float2(s.lr, 1.f), -1);
        if (go) {
          if constexpr (MODE == 1) opt_update<KIND>(bw[0], bg[0], bm[0], bv[0], s, ss_now, bc2_now);
          if (kHasW) c.w[bat] = bw[0];
          if constexpr (MODE != 0) {   // catch-up stores only w: the moments are replayed by this step's update
            if constexpr (kAdam) c.m[bat] = bm[0];
            c.v[bat] = bv[0];
            if constexpr (MODE == 1) c.g[bat] = bg[0];   // opt_update left g = 0
          }
        }
      }
      // ---- phase B: the rows, ROWS at a time in stamp order, dealt out to the waves
      auto fetch = [&](Group& G, int gi) __attribute__((always_inline)) {
        const int p = gi * ROWS + sub;
        const bool valid = p < n_active;
        const int o = s_order[valid ? p : 0];
        const long long rid = s_id[o & 63];
        const int r_old = s_old[o & 63];
        G.col_on = valid && 4 * sl < D;
        G.w_ahead = r_old >= 0 && (r_old & kWAhead);
        G.base = r_old < 0 ? -1 : (r_old & ~kWAhead);
        G.emb = G.col_on ? ((o & 256) ? nu * D : 0) + rid * D + 4 * sl : 0;
        G.first = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          const int b = __builtin_amdgcn_readlane(G.base, r * LPR);
          if (gi * ROWS + r < n_active && b >= 0) G.first = min(G.first, b + 1);
        }
        G.sc = make_float2(s.lr, 1.f);
        if constexpr (kAdam) G.sc = lazy_scalars_at(c, static_cast<long long>(G.first) + lane);
        auto load4 = [&](const float* ptr, float (&dst)[4]) __attribute__((always_inline)) {
          const float4 x = *reinterpret_cast<const float4*>(ptr + G.emb);   // unconditional: lanes that are off read row 0
          dst[0] = G.col_on ? x.x : 0.f, dst[1] = G.col_on ? x.y : 0.f;
          dst[2] = G.col_on ? x.z : 0.f, dst[3] = G.col_on ? x.w : 0.f;
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) G.w[j] = G.m[j] = G.g[j] = 0.f;
        if (kHasW) load4(c.w, G.w);
        if constexpr (kAdam) load4(c.m, G.m);
        load4(c.v, G.v);
        if constexpr (MODE == 1) load4(c.g, G.g);
      };
      auto process = [&](Group& G) __attribute__((always_inline)) {
        bool moving = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) moving |= G.m[j] != 0.f || G.v[j] != 0.f;
        const bool on = G.col_on && G.base >= 0 && moving;
        const uint64_t on_mask = __ballot(on);
        int first = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          const uint64_t row = (LPR == 64 ? ~0ull : ((1ull << LPR) - 1ull)) << (r * LPR);
          const int b = __builtin_amdgcn_readlane(G.base, r * LPR);
          if (on_mask & row) first = min(first, b + 1);
        }
        replay(G.w, G.m, G.v, on, G.base, G.w_ahead, first, G.sc, G.first);
        if (!G.col_on) return;
        if constexpr (MODE == 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) opt_update<KIND>(G.w[j], G.g[j], G.m[j], G.v[j], s, ss_now, bc2_now);
        }
        auto store4 = [&](float* ptr, const float (&src)[4]) __attribute__((always_inline)) {
          *reinterpret_cast<float4*>(ptr + G.emb) = float4{src[0], src[1], src[2], src[3]};
        };
        if (kHasW) store4(c.w, G.w);
        if constexpr (MODE != 0) {
          if constexpr (kAdam) store4(c.m, G.m);
          store4(c.v, G.v);
          if constexpr (MODE == 1) store4(c.g, G.g);
        }
      };
      Group ga, gb;
      int gi = wv;
      fetch(ga, gi);
      for (;;) {
        fetch(gb, gi + kWavesPerBlock);
        if (gi * ROWS >= n_active) break;
        process(ga);
        gi += 2 * kWavesPerBlock;
        fetch(ga, gi);
        if ((gi - kWavesPerBlock) * ROWS >= n_active) break;
        process(gb);
      }
    }
    __syncthreads();   // the next chunk rewrites the shared arrays
  }
  if constexpr (MODE == 1) lazy_scalar_step<KIND>(c, s, stats, scratch, clock, ss_now, bc2_now, bid);
}

// (the catch-up is the launch every step waits for: held to the 72 VGPRs = seven waves per SIMD it had before the
// checked block was added)
template <int KIND, int MODE, int LPR>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MODE == 0 ? 7 : 1)))
void lazy_rows_vec_kernel(LazyCtx c, hiprec_lazy_rows rows, OptScalars s, hiprec_stats* stats, const Scratch* scratch) {
  lazy_rows_vec_body<KIND, MODE, LPR>(c, rows, s, stats, scratch, static_cast<int>(blockIdx.x),
                                      static_cast<int>(gridDim.x));
}

#ifdef HIPREC_TEST_SWITCHES
// Timing experiment (tools/exp_lazy_overlap.py, libhiprec_test.so only): the update of one state (memory-bound) and the
// catch-up of another (arithmetic-bound) in ONE launch, even / odd workgroups -- what merging the update of step t
// with the catch-up of step t + 1 could buy.
__global__ __launch_bounds__(kBlock) void lazy_dual_kernel(LazyCtx c1, hiprec_lazy_rows r1, OptScalars s1,
                                                           hiprec_stats* st1, LazyCtx c2, hiprec_lazy_rows r2,
                                                           OptScalars s2, hiprec_stats* st2) {
  const int half = static_cast<int>(gridDim.x) / 2, bid = static_cast<int>(blockIdx.x) >> 1;
  if (blockIdx.x & 1) lazy_rows_vec_body<HIPREC_OPT_ADAM, 0, 32>(c2, r2, s2, st2, nullptr, bid, half);
  else lazy_rows_vec_body<HIPREC_OPT_ADAM, 1, 32>(c1, r1, s1, st1, nullptr, bid, half);
}
#endif


// MODE 0 = catch-up (to the clock), 1 = update (the step the clock shows), 2 = flush (all rows, to the clock; the vector
// kernels take an Adam flush in two launches, MODE 2 then MODE 3: see lazy_claim)
// (declared before the kernels that use them)

template <int KIND, int MODE>
__global__ __launch_bounds__(kBlock) void lazy_rows_kernel(LazyCtx c, hiprec_lazy_rows rows, OptScalars s,
                                                           hiprec_stats* stats, const Scratch* scratch) {
  const int lane = lane_id();
  const int D = c.dim;
  const long long clock = stats->step;
  float ss_now = s.lr, bc2_now = 1.f;
  if constexpr (MODE == 1) step_scalars<KIND>(s, stats, &ss_now, &bc2_now);
  const int64_t nu = c.n_users, ni = c.n_items;
  const int64_t n0 = MODE == 2 ? nu : rows.n_users, n1 = MODE == 2 ? ni : rows.n_items_a;
  const int64_t n2 = MODE == 2 ? 0 : rows.n_items_b, n3 = MODE == 2 ? 0 : rows.n_items_c;
  const int64_t total = n0 + n1 + n2 + n3;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block(); e < total; e += n_waves) {
    bool is_item = true;
    const int64_t id = lazy_entry<MODE>(c, rows, e, &is_item);
    if (id < 0) continue;  // padding of a fixed-size block / an exchange's extra row
    if (id >= (is_item ? ni : nu)) {
      if (lane == 0) atomicOr(&stats->status, HIPREC_STATUS_ROW_OOB);
      continue;
    }
    int32_t* sp = (is_item ? c.stamp_i : c.stamp_u) + id;
    const int target = static_cast<int>(clock);
    // lane 0 settles who works on the row; `old` = the stamp found, `go` = this wave owns the row for this launch
    int old = 0, go = 0;
    if (lane == 0) {
      old = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if constexpr (MODE == 0) {   // raise the W_AHEAD flag: the first to set it owns the row
        if (old >= 0 && !(old & kWAhead) && old < target) {
          old = atomicOr(sp, kWAhead);
          go = !(old & kWAhead);
        }
      } else if constexpr (MODE == 1) {   // the first to write the step's number owns the row
        if (old != target) {
          old = atomicExch(sp, target);
          go = old != target;
        }
      } else {                     // flush: one visitor per row; a never-touched row stays -1
        go = old >= 0 && (old & ~kWAhead) < target;
        if (go) *sp = target;
      }
    }
    old = __builtin_amdgcn_readfirstlane(old);
    if (!__builtin_amdgcn_readfirstlane(go)) continue;
    const bool w_ahead = old >= 0 && (old & kWAhead);   // catch-up already moved w (only) up to the clock
    const int base = old < 0 ? -1 : (old & ~kWAhead);
    const int64_t emb = (is_item ? nu * D : 0) + id * D;
    const int64_t bias = (nu + ni) * static_cast<int64_t>(D) + (is_item ? nu : 0) + id;
    // this lane's elements: columns lane + 64 j, and the bias element in lane 0
    int64_t at[kLazyMaxNpl + 1];
    bool on[kLazyMaxNpl + 1];
    float w[kLazyMaxNpl + 1], m[kLazyMaxNpl + 1], v[kLazyMaxNpl + 1], g[kLazyMaxNpl + 1];
#pragma unroll
    for (int j = 0; j <= kLazyMaxNpl; ++j) {
      const int col = lane + kWave * j;
      on[j] = j < kLazyMaxNpl ? col < D : lane == 0;
      at[j] = j < kLazyMaxNpl ? emb + col : bias;
      w[j] = m[j] = v[j] = g[j] = 0.f;
      if (on[j]) {
        if (KIND == HIPREC_OPT_ADAM || MODE == 1) w[j] = c.w[at[j]];
        if constexpr (KIND == HIPREC_OPT_ADAM) m[j] = c.m[at[j]];
        v[j] = c.v[at[j]];
        if constexpr (MODE == 1) g[j] = c.g[at[j]];
      }
    }
    // the zero-gradient steps this row has missed: (base, last] with last = the last COMPLETED step.  A row whose
    // moments are all zero (marked current by a dense sweep that never gave it a gradient) is a fixed point of them.
    const long long last = MODE == 1 ? clock - 1 : clock;
    bool moving = false;
#pragma unroll
    for (int j = 0; j <= kLazyMaxNpl; ++j) moving |= on[j] && (m[j] != 0.f || v[j] != 0.f);
    const bool replay = base >= 0 && __ballot(moving) != 0ull;
    // w is current already (w_ahead; a flush meets the flag only after an abandoned step, see lazy_rows_vec_body), or
    // has stopped moving (BOUNDED REPLAY above; RMSprop's zero-gradient step never moves it): only the moments are
    // replayed -- one fma and one multiply per step, no sqrt / rcp
    bool frozen = (MODE != 0 && w_ahead) || KIND != HIPREC_OPT_ADAM;
    for (long long t = base + 1LL; replay && t <= last; ++t) {
      if (frozen) {
        if constexpr (MODE == 0) break;   // a catch-up stores w only
#pragma unroll
        for (int j = 0; j <= kLazyMaxNpl; ++j) {
          float zero = 0.f, w_unused = 0.f;
          opt_update<KIND>(w_unused, zero, m[j], v[j], s, 1.f, 1.f);
        }
        continue;
      }
      float2 sc = make_float2(s.lr, 1.f);
      if constexpr (KIND == HIPREC_OPT_ADAM) sc = lazy_scalars_at(c, t);
      const bool check = t - base >= kFreezeStart && (t - base) % kFreezeEvery == 0 && t >= c.freeze_from;
      bool moved = false;
#pragma unroll
      for (int j = 0; j <= kLazyMaxNpl; ++j) {
        float zero = 0.f;
        const float w0 = w[j];
        opt_update<KIND, true>(w[j], zero, m[j], v[j], s, sc.x, sc.y);
        if (check) moved |= on[j] && (w[j] != w0 || !(__builtin_fabsf(w0) >= c.freeze_wmin || m[j] == 0.f));
      }
      if (check && __ballot(moved) == 0ull) frozen = true;
    }
    if constexpr (MODE == 1) {
#pragma unroll
      for (int j = 0; j <= kLazyMaxNpl; ++j) opt_update<KIND>(w[j], g[j], m[j], v[j], s, ss_now, bc2_now);
    }
    if constexpr (MODE == 0) {   // only w: the moments are replayed by this step's update, which follows in any case
#pragma unroll
      for (int j = 0; j <= kLazyMaxNpl; ++j)
        if (on[j]) c.w[at[j]] = w[j];
      continue;
    }
#pragma unroll
    for (int j = 0; j <= kLazyMaxNpl; ++j) {
      if (!on[j]) continue;
      if (KIND == HIPREC_OPT_ADAM || MODE == 1) c.w[at[j]] = w[j];
      if constexpr (KIND == HIPREC_OPT_ADAM) c.m[at[j]] = m[j];
      c.v[at[j]] = v[j];
      if constexpr (MODE == 1) c.g[at[j]] = 0.f;
    }
  }
  if constexpr (MODE == 1) lazy_scalar_step<KIND>(c, s, stats, scratch, clock, ss_now, bc2_now);
}

// ---- owner pulls (round 5): the apply launch of a lazy Adam / RMSprop step ------------------------------------------
// hiprec_mf_epoch_lazy_pull: catch-up (above) -> the gradient launch of csrc/mf_owned.hip with EVERY row's gradient
// parts stored to the contribution buffer (hiprec_batch_row_contrib with min_contrib = 1; it also counts the step) ->
// this launch: one lane group per row of the batch sums the row's range, replays the MOMENTS over the steps the row
// lagged (its weights are current: the catch-up's w-ahead state, the same fma + multiply per step as the update launch
// above), takes the real step and stamps the row.  Against catch-up + gradient kernel + update (three launches, the
// gradient through a dense buffer: written with atomics where waves share a row, read and cleared by the update) a row's
// gradient crosses memory once, nobody claims anything (a row has ONE record), and the gradient buffer is not touched.
// The arithmetic of a row is the update launch's, operation for operation; what differs is the order in which the
// parts of a shared row's gradient are summed.  dim % 4 == 0.
template <int KIND, int LPR>
__global__ __launch_bounds__(kPullBlock) __attribute__((amdgpu_waves_per_eu(8)))
void lazy_pull_apply_kernel(PullApply f, LazyCtx c, OptScalars s,
                                                                     hiprec_stats* stats, const Scratch* scratch) {
  constexpr int RPW = kWave / LPR;
  constexpr int GROUPS = kPullWaves * RPW;
  constexpr int DEPTH = 2;   // (most rows of a batch have ONE part; 64 VGPRs = two workgroups per CU)
  constexpr int kLongDepth = 4;
  constexpr bool kAdam = KIND == HIPREC_OPT_ADAM;
  __shared__ float4 s_part[GROUPS][LPR];
  __shared__ float s_pb[GROUPS];
  const int lane = lane_id(), wv = wave_in_block();
  const int nb = static_cast<int>(gridDim.x) - 1, blk = static_cast<int>(blockIdx.x) - 1;
  const long long clock = stats->step;          // the step being taken: the gradient launch has counted it
  float ss_now = s.lr, bc2_now = 1.f;
  step_scalars<KIND>(s, stats, &ss_now, &bc2_now);
  if (blk < 0) {
    if (threadIdx.x == 0 && f.counts[3] != 0) atomicOr(&stats->status, HIPREC_STATUS_TABLE_FULL);   // incomplete lists
    lazy_scalar_step<KIND, kPullBlock>(c, s, stats, scratch, clock, ss_now, bc2_now, 0);
    return;
  }
  const int D = f.dim;
  const int sub = lane / LPR, sl = lane % LPR, grp = wv * RPW + sub;
  const bool col = sl * 4 < D;
  const int target = static_cast<int>(clock);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int n_short = f.counts[0], n_long = f.counts[1];

  // A row of the batch in two halves, so that everything a row needs travels together: load_row requests its stamp,
  // weights and moments (before the caller waits for the row's gradient parts), finish_row replays the moments over
  // (base, clock - 1], takes the real step, stores and stamps.  `on`: this lane group holds a row; g / gb: the row's
  // complete gradient (gb valid in every lane of the group).
  struct Row {
    int key, old;                // (addresses are derived again from the key when the row is stored: registers)
    float w[5], m[5], v[5];
  };
  auto load_row = [&](bool on, int key) __attribute__((always_inline)) {
    Row r;
    r.key = key;
    r.old = target - 1;
#pragma unroll
    for (int j = 0; j < 5; ++j) r.w[j] = r.m[j] = r.v[j] = 0.f;
    if (on) {
      int64_t ro, bo;
      pull_row_of(f, key, sl, &ro, &bo);
      r.old = *(key < f.n_users ? c.stamp_u + key : c.stamp_i + (key - f.n_users));
      if (col) {
        const float4 w4 = *reinterpret_cast<const float4*>(c.w + ro), v4 = *reinterpret_cast<const float4*>(c.v + ro);
        r.w[0] = w4.x, r.w[1] = w4.y, r.w[2] = w4.z, r.w[3] = w4.w;
        r.v[0] = v4.x, r.v[1] = v4.y, r.v[2] = v4.z, r.v[3] = v4.w;
        if constexpr (kAdam) {
          const float4 m4 = *reinterpret_cast<const float4*>(c.m + ro);
          r.m[0] = m4.x, r.m[1] = m4.y, r.m[2] = m4.z, r.m[3] = m4.w;
        }
      }
      if (sl == 0) {
        r.w[4] = c.w[bo];
        r.v[4] = c.v[bo];
        if constexpr (kAdam) r.m[4] = c.m[bo];
      }
    }
    return r;
  };
  auto finish_row = [&](Row& r, bool on, float4 g, float gb) __attribute__((always_inline)) {
    // zero-gradient steps (base, clock - 1]: the moments only (w is current: it was caught up before the gradients
    // were taken, or the row did not lag; RMSprop's zero-gradient step leaves w alone)
    const int base = r.old < 0 ? -1 : (r.old & ~kWAhead);
    int n_rep = on && base >= 0 ? target - 1 - base : 0;
    n_rep = n_rep > 0 ? n_rep : 0;
    int n_max = n_rep;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_max = max(n_max, __shfl_xor(n_max, o));
    n_max = __builtin_amdgcn_readfirstlane(n_max);
    for (int k = 0; k < n_max; ++k) {
      if (k >= n_rep) continue;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        float zero = 0.f, w_unused = 0.f;
        opt_update<KIND>(w_unused, zero, r.m[j], r.v[j], s, 1.f, 1.f);
      }
    }
    float gg[5] = {g.x, g.y, g.z, g.w, gb};
#pragma unroll
    for (int j = 0; j < 5; ++j) opt_update<KIND>(r.w[j], gg[j], r.m[j], r.v[j], s, ss_now, bc2_now);
    if (!on) return;
    int64_t ro, bo;
    pull_row_of(f, r.key, sl, &ro, &bo);
    if (col) {
      *reinterpret_cast<float4*>(c.w + ro) = make_float4(r.w[0], r.w[1], r.w[2], r.w[3]);
      *reinterpret_cast<float4*>(c.v + ro) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
      if constexpr (kAdam) *reinterpret_cast<float4*>(c.m + ro) = make_float4(r.m[0], r.m[1], r.m[2], r.m[3]);
    }
    if (sl == 0) {
      c.w[bo] = r.w[4];
      c.v[bo] = r.v[4];
      if constexpr (kAdam) c.m[bo] = r.m[4];
      *(r.key < f.n_users ? c.stamp_u + r.key : c.stamp_i + (r.key - f.n_users)) = target;
    }
  };

  // long rows first: one workgroup each
  for (int i = blk; i < n_long; i += nb) {
    const int4 rec = f.rows[f.row_cap - 1 - i];
    float4 g = zero4;
    float gb = 0.f;
    const int per_trip = kLongDepth * GROUPS;
    pull_sum_range<LPR, kLongDepth>(f, rec.y, grp, rec.z, GROUPS, (rec.z + per_trip - 1) / per_trip, sl, col, g, gb);
    s_part[grp][sl] = g;
    if (sl == 0) s_pb[grp] = gb;
    __syncthreads();
    if (wv == 0) {
      float4 tot = zero4;
      float tb = 0.f;
#pragma unroll 4
      for (int q = 0; q < GROUPS; ++q) {
        const float4 p = s_part[q][sl];
        tot.x += p.x, tot.y += p.y, tot.z += p.z, tot.w += p.w;
        tb += s_pb[q];
      }
      Row r = load_row(sub == 0, rec.x);
      finish_row(r, sub == 0, tot, tb);
    }
    __syncthreads();
  }
  // the other rows: one lane group each, the next record requested before this one's row is stepped
  const int i_first = blk * GROUPS + wv * RPW;
  int4 rec = make_int4(0, 0, 0, 0);
  if (i_first + sub < f.row_cap) rec = f.rows[i_first + sub];
  for (int i0 = i_first; __builtin_amdgcn_readfirstlane(i0) < n_short; i0 += nb * GROUPS) {
    const bool on = i0 + sub < n_short;
    if (!on) rec = make_int4(0, 0, 0, 0);
    int trips = (rec.z + DEPTH - 1) / DEPTH;
#pragma unroll
    for (int o = LPR; o < kWave; o <<= 1) trips = max(trips, __shfl_xor(trips, o));
    trips = __builtin_amdgcn_readfirstlane(trips);
    float4 g = zero4;
    float gb = 0.f;
    Row r = load_row(on, rec.x);                 // stamp, weights, moments: requested with the gradient parts
    pull_sum_range<LPR, DEPTH>(f, rec.y, 0, rec.z, 1, trips, sl, col, g, gb);
    const int nxt = i0 + nb * GROUPS + sub;
    if (nxt < n_short) rec = f.rows[nxt];
    finish_row(r, on, g, gb);
  }
}

// A dense sweep was run while lazy state exists (the caller flushed first): every row is current as of the clock.
__global__ __launch_bounds__(kBlock) void lazy_mark_current_kernel(int32_t* stamp, int64_t n, const hiprec_stats* stats) {
  const int target = static_cast<int>(stats->step);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock)
    stamp[i] = target;   // EVERY row: the sweep may have given any of them its first gradient
}

template <int MODE>
int lazy_launch(const hiprec_lazy_state* st, const hiprec_lazy_rows* rows, const void* scratch, hiprec_stats* stats,
                void* stream) {
  HIPREC_REQUIRE(st && stats, "NULL pointer");
  HIPREC_REQUIRE(st->kind == HIPREC_OPT_ADAM || st->kind == HIPREC_OPT_RMSPROP,
                 "lazy optimizer state exists for Adam and RMSprop (SGD touches only the step's rows anyway)");
  HIPREC_REQUIRE(st->w && st->g && st->v && (st->kind != HIPREC_OPT_ADAM || st->m) && st->stamp_u && st->stamp_i,
                 "NULL buffer in the lazy optimizer state");
  HIPREC_REQUIRE(st->n_users >= 0 && st->n_items >= 0 && st->dim > 0 && st->dim <= kLazyMaxNpl * kWave,
                 "lazy optimizer rows need 0 < dim <= %d", kLazyMaxNpl * kWave);
  HIPREC_REQUIRE(st->kind != HIPREC_OPT_ADAM || (st->scalars && st->scalars_cap >= 2), "Adam needs the scalars table");
  hiprec_lazy_rows r{};
  int64_t total = st->n_users + st->n_items;
  if (MODE != 2) {
    HIPREC_REQUIRE(rows, "NULL row lists");
    r = *rows;
    HIPREC_REQUIRE(r.n_users >= 0 && r.n_items_a >= 0 && r.n_items_b >= 0 && r.n_items_c >= 0, "negative list length");
    HIPREC_REQUIRE((r.n_users == 0 || r.users) && (r.n_items_a == 0 || r.items_a) && (r.n_items_b == 0 || r.items_b) &&
                       (r.n_items_c == 0 || r.items_c),
                   "NULL row list");
    total = r.n_users + r.n_items_a + r.n_items_b + r.n_items_c;
  }
  if (MODE == 0 && st->kind == HIPREC_OPT_RMSPROP) return 0;  // w does not move on a zero-gradient RMSprop step
  if (MODE != 1 && total == 0) return 0;
  LazyCtx c{st->w, st->g, st->m, st->v, st->n_users, st->n_items, st->dim, st->stamp_u, st->stamp_i,
            reinterpret_cast<float2*>(st->scalars), st->scalars_cap, 0x7fffffff, 0.f};
  c.freeze_from = lazy_freeze_from(st->beta1, st->beta2, st->lr, st->eps, &c.freeze_wmin);
  const OptScalars s{st->lr,
                     static_cast<float>(st->lr),
                     static_cast<float>(st->beta2),
                     static_cast<float>(1.0 - st->beta1),
                     static_cast<float>(1.0 - st->beta2),
                     static_cast<float>(st->eps)};
  hipStream_t stm = static_cast<hipStream_t>(stream);
  const auto* sc = static_cast<const Scratch*>(scratch);
  const bool adam = st->kind == HIPREC_OPT_ADAM;
  if (st->dim % 4 == 0) {   // 16-byte vectors, several rows per wave
#define HIPREC_LAZY_VEC(LPR)                                                                                         \
  do {                                                                                                               \
    const int grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>((total + kWave - 1) / kWave, kMaxBlocks))); \
    if (adam) lazy_rows_vec_kernel<HIPREC_OPT_ADAM, MODE, LPR><<<grid, kBlock, 0, stm>>>(c, r, s, stats, sc);        \
    else lazy_rows_vec_kernel<HIPREC_OPT_RMSPROP, MODE, LPR><<<grid, kBlock, 0, stm>>>(c, r, s, stats, sc);          \
    if (adam && MODE == 2)   /* the rows that lag by more than kFreezeStart steps: bounded replay */                 \
      lazy_rows_vec_kernel<HIPREC_OPT_ADAM, MODE == 2 ? 3 : MODE, LPR><<<grid, kBlock, 0, stm>>>(c, r, s, stats, sc); \
  } while (0)
    if (st->dim <= 64) HIPREC_LAZY_VEC(16);
    else if (st->dim <= 128) HIPREC_LAZY_VEC(32);
    else HIPREC_LAZY_VEC(64);
#undef HIPREC_LAZY_VEC
  } else {
    const int grid = grid_for_waves(total);
    if (adam) lazy_rows_kernel<HIPREC_OPT_ADAM, MODE><<<grid, kBlock, 0, stm>>>(c, r, s, stats, sc);
    else lazy_rows_kernel<HIPREC_OPT_RMSPROP, MODE><<<grid, kBlock, 0, stm>>>(c, r, s, stats, sc);
  }
  HIPREC_TRY(hipGetLastError());
  return 0;
}

}  // namespace
}  // namespace hiprec

using namespace hiprec;

extern "C" int hiprec_lazy_catchup(const hiprec_lazy_state* state, const hiprec_lazy_rows* rows, hiprec_stats* stats,
                                   void* stream) {
  return lazy_launch<0>(state, rows, nullptr, stats, stream);
}

extern "C" int hiprec_lazy_update(const hiprec_lazy_state* state, const hiprec_lazy_rows* rows, const void* scratch,
                                  hiprec_stats* stats, void* stream) {
  return lazy_launch<1>(state, rows, scratch, stats, stream);
}

extern "C" int hiprec_lazy_flush(const hiprec_lazy_state* state, hiprec_stats* stats, void* stream) {
  return lazy_launch<2>(state, nullptr, nullptr, stats, stream);
}

extern "C" int hiprec_lazy_mark_current(const hiprec_lazy_state* state, const hiprec_stats* stats, void* stream) {
  HIPREC_REQUIRE(state && stats && state->stamp_u && state->stamp_i, "NULL pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (state->n_users > 0)
    lazy_mark_current_kernel<<<grid_for_threads(state->n_users), kBlock, 0, st>>>(state->stamp_u, state->n_users, stats);
  if (state->n_items > 0)
    lazy_mark_current_kernel<<<grid_for_threads(state->n_items), kBlock, 0, st>>>(state->stamp_i, state->n_items, stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" size_t hiprec_lazy_state_bytes(void) { return sizeof(hiprec_lazy_state); }

// MFEngine.train_an_epoch (mf.py:121-139) over a staged epoch with the exact lazy optimizer: per step catch-up of the
// batch's rows -> the gradient kernel (dense gradient, csrc/mf.hip) -> update of the batch's rows, which also folds
// the loss partials and the scalar bias like the dense sweep does.  The caller flushes afterwards.
extern "C" int hiprec_mf_epoch_lazy(const hiprec_lazy_state* state, const hiprec_mf_tables* w, const hiprec_mf_tables* g,
                                    const int64_t* users, const int64_t* items_a, const void* third, int32_t loss_kind,
                                    int64_t n, int64_t batch, int32_t first_of_epoch, float reg_coef,
                                    hiprec_stats* stats, void* scratch, size_t scratch_bytes, void* stream) {
  HIPREC_REQUIRE(state && w && g && stats && scratch, "NULL pointer");
  HIPREC_REQUIRE(n >= 0 && batch > 0 && (loss_kind == 0 || loss_kind == 1), "bad n / batch / loss kind");
  HIPREC_REQUIRE(n == 0 || (users && items_a && third), "NULL batch arrays");
  if (first_of_epoch)
    if (int rc = hiprec_stats_begin_epoch(stats, stream)) return rc;
  for (int64_t off = 0; off < n; off += batch) {
    const int64_t b = std::min<int64_t>(batch, n - off);  // drop_last = False
    const float inv_b = 1.0f / static_cast<float>(b);
    const int64_t* neg = loss_kind == 0 ? static_cast<const int64_t*>(third) + off : nullptr;
    const hiprec_lazy_rows rows{users + off, b, items_a + off, b, neg, neg ? b : 0, nullptr, 0};
    if (int rc = hiprec_lazy_catchup(state, &rows, stats, stream)) return rc;
    int rc;
    if (loss_kind == 0)
      rc = hiprec_mf_bpr_grad(w, g, users + off, items_a + off, neg, nullptr, b, inv_b, reg_coef, stats, scratch,
                              scratch_bytes, stream);
    else
      rc = hiprec_mf_bce_grad(w, g, users + off, items_a + off, static_cast<const float*>(third) + off, nullptr, b, inv_b,
                              reg_coef, stats, scratch, scratch_bytes, stream);
    if (rc) return rc;
    if ((rc = hiprec_lazy_update(state, &rows, scratch, stats, stream))) return rc;
  }
  return 0;
}

// The same epoch with the owned-rows gradient kernel (csrc/mf_owned.hip, hiprec_mf_bpr_grad_owned) in place of the
// atomics of mf_bpr_grad_kernel: BPR only, needs the epoch's row-ownership arrays (hiprec_batch_row_ownership: own_*
// [n], total [n_batches][total_stride]).
extern "C" int hiprec_mf_epoch_lazy_owned(const hiprec_lazy_state* state, const int64_t* users, const int64_t* pos,
                                          const int64_t* neg, const int32_t* own_u, const int32_t* own_p,
                                          const int32_t* own_n, const int32_t* total, int64_t total_stride, int64_t n,
                                          int64_t batch, int32_t first_of_epoch, float reg_coef, hiprec_stats* stats,
                                          void* scratch, void* stream) {
  HIPREC_REQUIRE(state && stats && scratch, "NULL pointer");
  HIPREC_REQUIRE(n >= 0 && batch > 0 && total_stride >= 0, "bad n / batch / total_stride");
  HIPREC_REQUIRE(n == 0 || (users && pos && neg && own_u && own_p && own_n && total), "NULL batch / ownership arrays");
  if (first_of_epoch)
    if (int rc = hiprec_stats_begin_epoch(stats, stream)) return rc;
  for (int64_t off = 0, k = 0; off < n; off += batch, ++k) {
    const int64_t b = std::min<int64_t>(batch, n - off);  // drop_last = False
    const float inv_b = 1.0f / static_cast<float>(b);
    const hiprec_lazy_rows rows{users + off, b, pos + off, b, neg + off, b, nullptr, 0};
    if (int rc = hiprec_lazy_catchup(state, &rows, stats, stream)) return rc;
    if (int rc = hiprec_mf_bpr_grad_owned(state->w, state->g, state->n_users, state->n_items, state->dim, users + off,
                                          pos + off, neg + off, own_u + off, own_p + off, own_n + off,
                                          total + k * total_stride, b, inv_b, reg_coef, stats, scratch, stream))
      return rc;
    if (int rc = hiprec_lazy_update(state, &rows, scratch, stats, stream)) return rc;
  }
  return 0;
}

// The same epoch as owner pulls (round 5): per step catch-up -> gradient launch (every row's parts to the contribution
// buffer; counts the step) -> lazy_pull_apply_kernel.  cidx / rows / counts: hiprec_batch_row_contrib's arrays made
// with min_contrib = 1, already offset to this piece's first step (cidx_stride = the n they were made for); cbuf
// [3 * batch, dim] / cbias [3 * batch]: work space.  BPR, dim % 4 == 0.  The dense gradient buffer of `state` is not
// touched (its scalar-bias element is read: zero for this caller).
extern "C" int hiprec_mf_epoch_lazy_pull(const hiprec_lazy_state* state, const int64_t* users, const int64_t* pos,
                                         const int64_t* neg, const int32_t* cidx, int64_t cidx_stride,
                                         const int32_t* rows, int64_t row_cap, const int32_t* counts, float* cbuf,
                                         float* cbias, int64_t n, int64_t batch, int32_t first_of_epoch, float reg_coef,
                                         hiprec_stats* stats, void* scratch, void* stream) {
  HIPREC_REQUIRE(state && stats && scratch, "NULL pointer");
  HIPREC_REQUIRE(state->kind == HIPREC_OPT_ADAM || state->kind == HIPREC_OPT_RMSPROP, "lazy state is Adam's or RMSprop's");
  HIPREC_REQUIRE(state->w && state->g && state->v && (state->kind != HIPREC_OPT_ADAM || state->m) && state->stamp_u &&
                     state->stamp_i, "NULL buffer in the lazy optimizer state");
  HIPREC_REQUIRE(state->kind != HIPREC_OPT_ADAM || (state->scalars && state->scalars_cap >= 2), "Adam needs the scalars table");
  HIPREC_REQUIRE(state->dim > 0 && state->dim <= 256 && state->dim % 4 == 0, "the pull form needs dim %% 4 == 0, dim <= 256");
  HIPREC_REQUIRE(n >= 0 && batch > 0 && cidx_stride >= n, "bad n / batch / cidx_stride");
  HIPREC_REQUIRE(n == 0 || (users && pos && neg && cidx && rows && counts && cbuf && cbias), "NULL batch / contribution arrays");
  HIPREC_REQUIRE(n == 0 || row_cap >= 3 * std::min(batch, n), "row_cap too small: every row of a batch has a record");
  if (first_of_epoch)
    if (int rc = hiprec_stats_begin_epoch(stats, stream)) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int32_t dim = state->dim;
  const LazyCtx c{state->w, state->g, state->m, state->v, state->n_users, state->n_items, dim, state->stamp_u,
                  state->stamp_i, reinterpret_cast<float2*>(state->scalars), state->scalars_cap, 0x7fffffff, 0.f};
  const OptScalars s{state->lr, static_cast<float>(state->lr), static_cast<float>(state->beta2),
                     static_cast<float>(1.0 - state->beta1), static_cast<float>(1.0 - state->beta2),
                     static_cast<float>(state->eps)};
  PullApply a;
  a.w = state->w;
  a.n_users = state->n_users;
  a.o_ie = state->n_users * dim;
  a.o_ub = (state->n_users + state->n_items) * static_cast<int64_t>(dim);
  a.o_ib = a.o_ub + state->n_users;
  a.dim = dim;
  a.begin_epoch = 0;
  a.count_step = 0;
  a.cbuf = cbuf;
  a.cbias = cbias;
  a.row_cap = row_cap;
  a.gb = nullptr;
  a.lr = static_cast<float>(state->lr);
  a.slot_out = nullptr;
  a.extra_rows = nullptr;
  a.n_dest = 0;
  const bool adam = state->kind == HIPREC_OPT_ADAM;
  for (int64_t off = 0, k = 0; off < n; off += batch, ++k) {
    const int64_t b = std::min<int64_t>(batch, n - off);  // drop_last = False
    const hiprec_lazy_rows lists{users + off, b, pos + off, b, neg + off, b, nullptr, 0};
    if (int rc = hiprec_lazy_catchup(state, &lists, stats, stream)) return rc;
    if (int rc = launch_pull_grad(state->w, state->n_users, state->n_items, dim, users + off, pos + off, neg + off,
                                  cidx + off, cidx + cidx_stride + off, cidx + 2 * cidx_stride + off, cbuf, cbias, b,
                                  reg_coef, 0.f, 1, stats, scratch, st))
      return rc;
    a.rows = reinterpret_cast<const int4*>(rows) + k * row_cap;
    a.counts = counts + 4 * k;
    // one lane group per row of the batch (at most 3 b), two workgroups per CU
    const int64_t per_block = kPullWaves * (dim <= 64 ? 4 : dim <= 128 ? 2 : 1);
    const int grid = static_cast<int>(std::min<int64_t>((3 * b + per_block - 1) / per_block, 512)) + 1;
    const auto* sc = static_cast<const Scratch*>(scratch);
#define HIPREC_LAZY_PULL(LPR)                                                                                      \
  do {                                                                                                             \
    if (adam) lazy_pull_apply_kernel<HIPREC_OPT_ADAM, LPR><<<grid, kPullBlock, 0, st>>>(a, c, s, stats, sc);        \
    else lazy_pull_apply_kernel<HIPREC_OPT_RMSPROP, LPR><<<grid, kPullBlock, 0, st>>>(a, c, s, stats, sc);          \
  } while (0)
    if (dim <= 64) HIPREC_LAZY_PULL(16);
    else if (dim <= 128) HIPREC_LAZY_PULL(32);
    else HIPREC_LAZY_PULL(64);
#undef HIPREC_LAZY_PULL
    HIPREC_TRY(hipGetLastError());
  }
  return 0;
}

#ifdef HIPREC_TEST_SWITCHES
extern "C" int hiprec_debug_lazy_dual(const hiprec_lazy_state* a, const hiprec_lazy_rows* ra, hiprec_stats* sa,
                                      const hiprec_lazy_state* b, const hiprec_lazy_rows* rb, hiprec_stats* sb,
                                      void* stream) {
  HIPREC_REQUIRE(a && b && ra && rb && sa && sb && a->dim == 128 && b->dim == 128 && a->kind == HIPREC_OPT_ADAM &&
                     b->kind == HIPREC_OPT_ADAM,
                 "two Adam states of dim 128");
  auto ctx = [](const hiprec_lazy_state* st) {
    return LazyCtx{st->w, st->g, st->m, st->v, st->n_users, st->n_items, st->dim, st->stamp_u, st->stamp_i,
                   reinterpret_cast<float2*>(st->scalars), st->scalars_cap, 0x7fffffff, 0.f};
  };
  auto sc = [](const hiprec_lazy_state* st) {
    return OptScalars{st->lr, static_cast<float>(st->lr), static_cast<float>(st->beta2),
                      static_cast<float>(1.0 - st->beta1), static_cast<float>(1.0 - st->beta2),
                      static_cast<float>(st->eps)};
  };
  lazy_dual_kernel<<<2 * kMaxBlocks, kBlock, 0, static_cast<hipStream_t>(stream)>>>(ctx(a), *ra, sc(a), sa, ctx(b), *rb,
                                                                                  sc(b), sb);
  HIPREC_TRY(hipGetLastError());
  return 0;
}
#endif
