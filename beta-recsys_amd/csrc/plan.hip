// The row-sharded engine's epoch planner (beta-recsys_amd/sharded.py::plan_epoch) as HIP kernels.
//
// Which rank owns a triple's user row and which item rows a rank must fetch for a step depend on the ids only,
// so a whole epoch is routed once, ahead of its steps.  Round 2 did that with generic torch ops on int64 keys
// (five argsorts, a unique, bincounts: 5-8 ms per 50-step epoch of 65 536-triple batches, the longer of the two
// streams next to the steps it was meant to hide behind).  Here every stage is integer work on 32-bit ids with no
// sort at all:
//   route     triples -> owner(user) = user mod R, (destination, step)-ordered: a stable counting sort over
//             tiles of 1024 triples (count per tile and destination, ONE scan, ranks by wave ballots);
//   place     what arrives, (source, step)-ordered, goes into fixed-size blocks per step (binary search over
//             the R x S group starts);
//   slots     the item references of a step are de-duplicated by the row-ownership hash tables the owned-rows
//             step needs anyway (csrc/ownership.hip: the table position of a row is its accumulator slot): every
//             occupied item entry of a step's table IS one distinct item, numbered per owner in table order --
//             that number is the row's slot in the step's exchange buffer -- and the positive-occurrence counts
//             of the same build lay every step out grouped by positive item (what the gradient kernel's
//             run-merge wants) by a counting sort over table entries;
//   requests  the owner side of the same: what peers will ask for, step by step.
// Orders inside a group (equal destination / equal item) never matter: a batch's loss and gradient are sums.
// Everything here is enqueued on the caller's stream; the two host round trips of an epoch plan (exact split
// sizes of its exchanges) stay in sharded.py.
#include <algorithm>

#include "common.hpp"

namespace hiprec {
namespace {

constexpr int kPlanTile = 1024;     // triples per tile of the routing sort
constexpr int kPlanMaxDest = 64;    // ranks (one lane per destination in the slot numbering)
constexpr int kPlanThreads = 1024;
constexpr int kPlanPartBits = 14;   // = kOwnPartBits of csrc/ownership.hip: the tables' partitions

struct TripleIds {
  int64_t u, p, q;
  bool ok;
};

__device__ __forceinline__ TripleIds load_triple(const int64_t* __restrict__ users, const int64_t* __restrict__ pos,
                                                 const int64_t* __restrict__ neg, const int64_t* __restrict__ perm,
                                                 int64_t j, int64_t n_users, int64_t n_items, hiprec_stats* stats) {
  const int64_t idx = perm ? perm[j] : j;
  TripleIds t{users[idx], pos[idx], neg[idx], true};
  const bool u_ok = static_cast<uint64_t>(t.u) < static_cast<uint64_t>(n_users);
  const bool i_ok = static_cast<uint64_t>(t.p) < static_cast<uint64_t>(n_items) &&
                    static_cast<uint64_t>(t.q) < static_cast<uint64_t>(n_items);
  t.ok = u_ok && i_ok;
  if (!t.ok && stats)  // reported once per epoch plan: IndexError on the host, like nn.Embedding's
    atomicOr(&stats->status, (u_ok ? 0u : HIPREC_STATUS_USER_OOB) | (i_ok ? 0u : HIPREC_STATUS_ITEM_OOB));
  return t;
}

// ---- route: count -------------------------------------------------------------------------------------------------
// tile = 1024 consecutive triples of ONE step (a step's last tile is short); tile_cnt[d * n_tiles + tile] and
// cnt_ds[d * S + step] (zero on entry) receive the number of its triples whose user row lives on rank d.
__global__ __launch_bounds__(kBlock) void plan_route_count_kernel(
    const int64_t* __restrict__ users, const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
    const int64_t* __restrict__ perm, int64_t n, int64_t bs, int tiles_per_step, int64_t n_tiles, int R, int S,
    int64_t n_users, int64_t n_items, int32_t* __restrict__ tile_cnt, int32_t* __restrict__ cnt_ds,
    hiprec_stats* stats) {
  __shared__ int s_cnt[kPlanMaxDest];
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t s = tile / tiles_per_step, k = tile % tiles_per_step;
    if (static_cast<int>(threadIdx.x) < R) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t lo = s * bs + k * kPlanTile;
    const int64_t hi = min(min(lo + kPlanTile, (s + 1) * bs), n);
    for (int64_t j = lo + threadIdx.x; j < hi; j += kBlock) {
      const TripleIds t = load_triple(users, pos, neg, perm, j, n_users, n_items, stats);
      if (t.ok) atomicAdd(&s_cnt[static_cast<int>(static_cast<uint32_t>(t.u) % static_cast<uint32_t>(R))], 1);  // ids < 2^31
    }
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < R) {
      const int c = s_cnt[threadIdx.x];
      tile_cnt[threadIdx.x * n_tiles + tile] = c;
      if (c) atomicAdd(&cnt_ds[threadIdx.x * S + s], c);
    }
    __syncthreads();
  }
}

// In-place exclusive prefix sum of a[0, L) by ONE workgroup (L is a few thousand to a few hundred thousand counts).
__global__ __launch_bounds__(kPlanThreads) void plan_exclusive_scan_kernel(int32_t* __restrict__ a, int64_t L) {
  __shared__ int s_sum[kPlanThreads];
  const int64_t per = (L + kPlanThreads - 1) / kPlanThreads;
  const int64_t lo = min(static_cast<int64_t>(threadIdx.x) * per, L), hi = min(lo + per, L);
  int sum = 0;
  for (int64_t i = lo; i < hi; ++i) sum += a[i];
  s_sum[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x < kWave) {  // one wave scans the 1024 partial sums, 16 per lane
    int loc[kPlanThreads / kWave], tot = 0;
#pragma unroll
    for (int i = 0; i < kPlanThreads / kWave; ++i) {
      loc[i] = tot;
      tot += s_sum[threadIdx.x * (kPlanThreads / kWave) + i];
    }
    int incl = tot;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (static_cast<int>(threadIdx.x) >= d) incl += up;
    }
    const int base = incl - tot;
#pragma unroll
    for (int i = 0; i < kPlanThreads / kWave; ++i) s_sum[threadIdx.x * (kPlanThreads / kWave) + i] = base + loc[i];
  }
  __syncthreads();
  int run = s_sum[threadIdx.x];
  for (int64_t i = lo; i < hi; ++i) {
    const int t = a[i];
    a[i] = run;
    run += t;
  }
}

// ---- route: pack --------------------------------------------------------------------------------------------------
// send[pos] = (user / R, positive item, negative item) as int32 x 3, pos = tile_base[d][tile] (the scanned counts:
// (destination, step, tile)-ordered) + the triple's rank among the tile's triples with the same destination, in
// visiting order (stable): rounds of 256 triples, per round one ballot per destination and wave.
__global__ __launch_bounds__(kBlock) void plan_route_pack_kernel(
    const int64_t* __restrict__ users, const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
    const int64_t* __restrict__ perm, int64_t n, int64_t bs, int tiles_per_step, int64_t n_tiles, int R,
    int64_t n_users, int64_t n_items, const int32_t* __restrict__ tile_base, int32_t* __restrict__ send) {
  __shared__ int s_run[kPlanMaxDest];
  __shared__ int s_wcnt[kWavesPerBlock][kPlanMaxDest];
  const int lane = lane_id(), w = wave_in_block();
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t s = tile / tiles_per_step, k = tile % tiles_per_step;
    const int64_t lo = s * bs + k * kPlanTile;
    const int64_t hi = min(min(lo + kPlanTile, (s + 1) * bs), n);
    if (static_cast<int>(threadIdx.x) < R) s_run[threadIdx.x] = tile_base[threadIdx.x * n_tiles + tile];
    for (int64_t r0 = lo; r0 < hi; r0 += kBlock) {
      const int64_t j = r0 + threadIdx.x;
      TripleIds t{0, 0, 0, false};
      if (j < hi) t = load_triple(users, pos, neg, perm, j, n_users, n_items, nullptr);
      const int d = t.ok ? static_cast<int>(static_cast<uint32_t>(t.u) % static_cast<uint32_t>(R)) : -1;
      int rank = 0;
      for (int q = 0; q < R; ++q) {
        const unsigned long long m = __ballot(d == q);
        if (lane == 0) s_wcnt[w][q] = __popcll(m);
        if (d == q) rank = __popcll(m & ((1ull << lane) - 1ull));
      }
      __syncthreads();
      if (d >= 0) {
        int at = s_run[d] + rank;
        for (int ww = 0; ww < w; ++ww) at += s_wcnt[ww][d];
        int32_t* o = send + 3 * static_cast<int64_t>(at);
        o[0] = static_cast<int32_t>(static_cast<uint32_t>(t.u) / static_cast<uint32_t>(R));
        o[1] = static_cast<int32_t>(t.p);
        o[2] = static_cast<int32_t>(t.q);
      }
      __syncthreads();
      if (static_cast<int>(threadIdx.x) < R) {
        int c = 0;
#pragma unroll
        for (int ww = 0; ww < kWavesPerBlock; ++ww) c += s_wcnt[ww][threadIdx.x];
        s_run[threadIdx.x] += c;
      }
      __syncthreads();
    }
    __syncthreads();
  }
}

// ---- groups of a (source, step)-ordered receive buffer --------------------------------------------------------
// cnt[Q][S] = elements source q sent for step s.  start[q * S + s] (and start[Q * S] = the total): the group's
// first element in the receive buffer; dst[q * S + s]: where that element goes in the step-major layout -- step s
// starts at s * cap (fixed blocks: cap > 0) or at step_off[s] (packed: cap == 0; step_off[S] = total length),
// sources follow each other inside a step, each followed by `extra` spare elements whose position inside the
// step's block is extra_pos[s * Q + q].  ONE workgroup: Q <= 64, S is the number of steps of an epoch.
__global__ __launch_bounds__(kPlanThreads) void plan_scan_groups_kernel(const int32_t* __restrict__ cnt, int Q,
                                                                         int S, int64_t cap, int extra,
                                                                         int32_t* __restrict__ start,
                                                                         int32_t* __restrict__ dst,
                                                                         int32_t* __restrict__ step_off,
                                                                         int32_t* __restrict__ extra_pos,
                                                                         int32_t* __restrict__ step_fill) {
  __shared__ int s_q[kPlanMaxDest + 1];
  const int tid = static_cast<int>(threadIdx.x);
  if (tid < Q) {
    int sum = 0;
    for (int s = 0; s < S; ++s) sum += cnt[tid * S + s];
    s_q[tid] = sum;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int q = 0; q < Q; ++q) {
      const int t = s_q[q];
      s_q[q] = run;
      run += t;
    }
    s_q[Q] = run;
  }
  __syncthreads();
  if (tid < Q) {
    int run = s_q[tid];
    for (int s = 0; s < S; ++s) {
      start[tid * S + s] = run;
      run += cnt[tid * S + s];
    }
    if (tid == Q - 1) start[Q * S] = run;
  }
  if (cap == 0) {
    for (int s = tid; s < S; s += kPlanThreads) {
      int tot = 0;
      for (int q = 0; q < Q; ++q) tot += cnt[q * S + s] + extra;
      step_off[s] = tot;
    }
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int s = 0; s < S; ++s) {
        const int t = step_off[s];
        step_off[s] = run;
        run += t;
      }
      step_off[S] = run;
    }
    __syncthreads();
  }
  for (int s = tid; s < S; s += kPlanThreads) {
    const int64_t base = cap > 0 ? static_cast<int64_t>(s) * cap : step_off[s];
    int run = 0;
    for (int q = 0; q < Q; ++q) {
      dst[q * S + s] = static_cast<int32_t>(base + run);
      run += cnt[q * S + s];
      if (extra) {
        extra_pos[s * Q + q] = run;
        run += extra;
      }
    }
    if (step_fill) step_fill[s] = run;   // elements of step s: what lies behind them in a fixed-size block is padding
  }
}

// Padding of fixed-size step blocks: positions [fill[s], cap) of step s get user -1, slots / items 0 and (own != NULL)
// ownership -1 -- only those: a memset of the whole arrays was most of the plan's fill traffic (no padding at all
// when every step received exactly cap triples).
__global__ __launch_bounds__(kBlock) void plan_pad_kernel(const int32_t* __restrict__ fill, int S, int64_t cap,
                                                          int64_t* __restrict__ U, int64_t* __restrict__ P,
                                                          int64_t* __restrict__ N, int32_t* __restrict__ own) {
  const int64_t tot = static_cast<int64_t>(S) * cap;
  for (int s = blockIdx.y; s < S; s += gridDim.y) {
    const int64_t lo = fill[s];
    for (int64_t i = lo + static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < cap;
         i += static_cast<int64_t>(gridDim.x) * kBlock) {
      const int64_t at = static_cast<int64_t>(s) * cap + i;
      U[at] = -1;
      P[at] = 0;
      N[at] = 0;
      if (own) own[at] = own[tot + at] = own[2 * tot + at] = -1;
    }
  }
}

// element j of the receive buffer -> its place in the step-major layout (binary search over the group starts)
__device__ __forceinline__ int64_t place_of(int64_t j, const int32_t* __restrict__ start,
                                            const int32_t* __restrict__ dst, int n_groups) {
  int lo = 0, hi = n_groups;  // the last g with start[g] <= j (empty groups share a start: take the last)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (start[mid] <= j) lo = mid;
    else hi = mid;
  }
  return static_cast<int64_t>(dst[lo]) + (j - start[lo]);
}

__global__ __launch_bounds__(kBlock) void plan_place_triples_kernel(const int32_t* __restrict__ in, int64_t n_in,
                                                                    const int32_t* __restrict__ start,
                                                                    const int32_t* __restrict__ dst, int n_groups,
                                                                    int64_t* __restrict__ U, int64_t* __restrict__ P,
                                                                    int64_t* __restrict__ N) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; j < n_in; j += stride) {
    const int64_t at = place_of(j, start, dst, n_groups);
    U[at] = in[3 * j];
    P[at] = in[3 * j + 1];
    N[at] = in[3 * j + 2];
  }
}

__global__ __launch_bounds__(kBlock) void plan_place_values_kernel(const int32_t* __restrict__ in, int64_t n_in,
                                                                   const int32_t* __restrict__ start,
                                                                   const int32_t* __restrict__ dst, int n_groups,
                                                                   int32_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; j < n_in; j += stride)
    out[place_of(j, start, dst, n_groups)] = in[j];
}

// Which of the rows a step's peers ask for are asked for by MORE than one of them?  (Every peer's list is free of
// duplicates, so a row occurs at most once per source.)  One bit per (step, local item row): the owner's apply pass
// adds the gradients of the others with plain read-modify-writes instead of fp32 atomics.  seen / dup: [S][words],
// zero on entry.
constexpr int kDupWords = 16384;   // words of one bitmap a workgroup keeps in LDS (2 bitmaps: 128 KB)
__global__ __launch_bounds__(kPlanThreads) void plan_mark_duplicates_kernel(const int32_t* __restrict__ in_idx,
                                                                             const int32_t* __restrict__ step_off,
                                                                             int64_t words, int n_parts,
                                                                             uint32_t* __restrict__ dup) {
  // block = (step, part): the bits of rows [part * 32 kDupWords, ...) of one step, built with LDS atomics from the
  // step's list (global atomics on a 12 MB bitmap: 200 us per plan; this: one pass over an L2-resident list)
  extern __shared__ uint32_t s_bits[];   // [w] seen, then [w] dup
  const int s = static_cast<int>(blockIdx.x) / n_parts, part = static_cast<int>(blockIdx.x) % n_parts;
  const int64_t w0 = static_cast<int64_t>(part) * kDupWords;
  const int w = static_cast<int>(min<int64_t>(kDupWords, words - w0));
  for (int i = threadIdx.x; i < 2 * w; i += kPlanThreads) s_bits[i] = 0u;
  __syncthreads();
  const int64_t lo = step_off[s], hi = step_off[s + 1];
  const int64_t r0 = w0 * 32, r1 = r0 + static_cast<int64_t>(w) * 32;
  for (int64_t k = lo + threadIdx.x; k < hi; k += kPlanThreads) {
    const int64_t row = in_idx[k];
    if (row < r0 || row >= r1) continue;
    const int j = static_cast<int>((row - r0) >> 5);
    const uint32_t bit = 1u << (row & 31);
    if (atomicOr(s_bits + j, bit) & bit) atomicOr(s_bits + w + j, bit);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < w; i += kPlanThreads) dup[static_cast<int64_t>(s) * words + w0 + i] = s_bits[w + i];
}

// ---- slots ----------------------------------------------------------------------------------------------------
// Every occupied ITEM entry (key >= n_users_local) of step s's ownership table is one distinct item of the step.
// part_cnt[(s * n_parts + p) * (R + 1) + d] = item entries of partition p owned by rank d (item mod R);
// [.. + R] = positive occurrences of the partition's items.
__global__ __launch_bounds__(kPlanThreads) void plan_slot_count_kernel(const int32_t* __restrict__ tab_keys,
                                                                        const int32_t* __restrict__ pos_cnt,
                                                                        int table_bits, int R, int32_t n_users_local,
                                                                        int32_t* __restrict__ part_cnt) {
  __shared__ int s_hist[kPlanMaxDest + 1];
  const int part_bits = table_bits < kPlanPartBits ? table_bits : kPlanPartBits;
  const int64_t e0 = static_cast<int64_t>(blockIdx.x) << part_bits;  // blockIdx = s * n_parts + p
  const int lane = lane_id();
  if (static_cast<int>(threadIdx.x) <= R) s_hist[threadIdx.x] = 0;
  __syncthreads();
  int npos = 0;
  for (int i = threadIdx.x; i < (1 << part_bits); i += kPlanThreads) {
    const int32_t key = tab_keys[e0 + i];
    const int d = key >= n_users_local ? (key - n_users_local) % R : -1;
    if (d >= 0) npos += pos_cnt[e0 + i];
    for (int q = 0; q < R; ++q) {  // one LDS add per wave and destination
      const unsigned long long m = __ballot(d == q);
      if (m && lane == 0) atomicAdd(&s_hist[q], __popcll(m));
    }
  }
  for (int off = 32; off > 0; off >>= 1) npos += __shfl_down(npos, off);
  if (lane == 0 && npos) atomicAdd(&s_hist[R], npos);
  __syncthreads();
  if (static_cast<int>(threadIdx.x) <= R)
    part_cnt[static_cast<int64_t>(blockIdx.x) * (R + 1) + threadIdx.x] = s_hist[threadIdx.x];
}

// From the partition counts, per step s: req_cnt[s][d] (rows asked of owner d) and its transpose req_ds[d][s] (the
// count exchange's layout); the step's exchange buffer is, owner by owner, [rows asked of d ..., one extra row]:
// chunk_start[s][d], ex_req[s][d] (the extra row), n_slots[s]; slot_base[(s * n_parts + p) * R + d] = first slot of
// partition p's items owned by d; pos_base[s * n_parts + p] = positive occurrences in earlier partitions; and
// send_base[d * S + s] = where step s's requests to owner d start in the (destination, step)-ordered request
// list (send_base[R * S] = its length).  ONE workgroup.
__global__ __launch_bounds__(kPlanThreads) void plan_slot_scan_kernel(
    const int32_t* __restrict__ part_cnt, int S, int n_parts, int R, int32_t* __restrict__ req_cnt,
    int32_t* __restrict__ req_ds, int32_t* __restrict__ chunk_start, int32_t* __restrict__ ex_req,
    int32_t* __restrict__ n_slots, int32_t* __restrict__ slot_base, int32_t* __restrict__ pos_base,
    int32_t* __restrict__ send_base, uint8_t* __restrict__ slot_shared, int64_t slot_stride) {
  __shared__ int s_q[kPlanMaxDest + 1];
  const int tid = static_cast<int>(threadIdx.x);
  for (int s = tid; s < S; s += kPlanThreads) {
    int run = 0;
    for (int d = 0; d < R; ++d) {
      int c = 0;
      for (int p = 0; p < n_parts; ++p) {
        slot_base[(static_cast<int64_t>(s) * n_parts + p) * R + d] = run + c;
        c += part_cnt[(static_cast<int64_t>(s) * n_parts + p) * (R + 1) + d];
      }
      req_cnt[s * R + d] = c;
      req_ds[d * S + s] = c;
      chunk_start[s * R + d] = run;
      run += c + 1;
      ex_req[s * R + d] = run - 1;
      if (slot_shared) slot_shared[static_cast<int64_t>(s) * slot_stride + run - 1] = 1;   // the extra row: cleared too
    }
    n_slots[s] = run;
    int acc = 0;
    for (int p = 0; p < n_parts; ++p) {
      pos_base[static_cast<int64_t>(s) * n_parts + p] = acc;
      acc += part_cnt[(static_cast<int64_t>(s) * n_parts + p) * (R + 1) + R];
    }
  }
  __syncthreads();
  if (tid < R) {
    int sum = 0;
    for (int s = 0; s < S; ++s) sum += req_ds[tid * S + s];
    s_q[tid] = sum;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int q = 0; q < R; ++q) {
      const int t = s_q[q];
      s_q[q] = run;
      run += t;
    }
    s_q[R] = run;
  }
  __syncthreads();
  if (tid < R) {
    int run = s_q[tid];
    for (int s = 0; s < S; ++s) {
      send_base[tid * S + s] = run;
      run += req_ds[tid * S + s];
    }
    if (tid == R - 1) send_base[R * S] = run;
  }
}

// Number the item entries: slot_of[e] = the item's slot in the step's exchange buffer (-1 for user rows and empty
// entries), req_send[...] = its row at the owner (item / R), and pos_cnt[e] becomes the START of the item's
// positive occurrences inside the step's block (exclusive prefix over the step's table).  Block = (step,
// partition), every wave a contiguous stretch of entries; lane d carries the running slot of destination d.
__global__ __launch_bounds__(kPlanThreads) void plan_slot_assign_kernel(
    const int32_t* __restrict__ tab_keys, int32_t* __restrict__ pos_cnt, int table_bits, int n_parts, int R, int S,
    int32_t n_users_local, const int32_t* __restrict__ slot_base, const int32_t* __restrict__ pos_base,
    const int32_t* __restrict__ chunk_start, const int32_t* __restrict__ send_base, int32_t* __restrict__ slot_of,
    int32_t* __restrict__ req_send, const int32_t* __restrict__ total, uint8_t* __restrict__ slot_shared,
    int64_t slot_stride) {
  constexpr int NW = kPlanThreads / kWave;
  __shared__ int s_wcnt[NW][kPlanMaxDest];
  __shared__ int s_wpos[NW];
  const int part_bits = table_bits < kPlanPartBits ? table_bits : kPlanPartBits;
  const int part_size = 1 << part_bits;
  const int s = static_cast<int>(blockIdx.x / n_parts);
  const int64_t e0 = static_cast<int64_t>(blockIdx.x) << part_bits;
  const int lane = lane_id(), w = wave_in_block();
  const int per_wave = ((part_size + NW - 1) / NW + kWave - 1) / kWave * kWave;  // whole groups of 64 entries
  const int w_lo = min(w * per_wave, part_size), w_hi = min(w_lo + per_wave, part_size);
  for (int i = threadIdx.x; i < NW * kPlanMaxDest; i += kPlanThreads) (&s_wcnt[0][0])[i] = 0;
  __syncthreads();
  // pass A: this wave's item entries per destination and its positive occurrences
  int npos = 0;
  for (int g = w_lo; g < w_hi; g += kWave) {
    const int i = g + lane;
    const int32_t key = i < w_hi ? tab_keys[e0 + i] : -1;
    const int d = key >= n_users_local ? (key - n_users_local) % R : -1;
    if (d >= 0) npos += pos_cnt[e0 + i];
    for (int q = 0; q < R; ++q) {
      const unsigned long long m = __ballot(d == q);
      if (m && lane == 0) s_wcnt[w][q] += __popcll(m);
    }
  }
  for (int off = 32; off > 0; off >>= 1) npos += __shfl_down(npos, off);
  if (lane == 0) s_wpos[w] = npos;
  __syncthreads();
  int run = 0;  // lane d: next slot of destination d for this wave
  if (lane < R) {
    run = slot_base[static_cast<int64_t>(blockIdx.x) * R + lane];
    for (int ww = 0; ww < w; ++ww) run += s_wcnt[ww][lane];
  }
  int pos_run = pos_base[blockIdx.x];
  for (int ww = 0; ww < w; ++ww) pos_run += s_wpos[ww];
  // pass B
  for (int g = w_lo; g < w_hi; g += kWave) {
    const int i = g + lane;
    const int32_t key = i < w_hi ? tab_keys[e0 + i] : -1;
    const int d = key >= n_users_local ? (key - n_users_local) % R : -1;
    int slot = -1;
    unsigned long long left = __ballot(d >= 0);
    while (left) {
      const int lead = __builtin_ctzll(left);
      const int dl = __builtin_amdgcn_readlane(d, lead);
      const unsigned long long m = __ballot(d == dl);
      const int base = __builtin_amdgcn_readlane(run, dl);
      if (d == dl) slot = base + __popcll(m & ((1ull << lane) - 1ull));
      if (lane == dl) run += __popcll(m);
      left &= ~m;
    }
    // exclusive prefix of the positive counts across the 64 entries of the group
    const int c = d >= 0 ? pos_cnt[e0 + i] : 0;
    int incl = c;
#pragma unroll
    for (int k = 1; k < kWave; k <<= 1) {
      const int up = __shfl_up(incl, k);
      if (lane >= k) incl += up;
    }
    if (i < w_hi) {
      slot_of[e0 + i] = slot;
      pos_cnt[e0 + i] = pos_run + incl - c;
      if (d >= 0) {
        const int item = key - n_users_local;
        req_send[send_base[d * S + s] + slot - chunk_start[s * R + d]] = item / R;
        // a slot several triples reference receives atomic adds: only those have to be zero when the step starts
        if (slot_shared) slot_shared[static_cast<int64_t>(s) * slot_stride + slot] = total[e0 + i] > 1 ? 1 : 0;
      }
    }
    pos_run += __builtin_amdgcn_readlane(incl, kWave - 1);
  }
}

// The step-major blocks, re-laid grouped by positive item: live triple t of step s goes to s * cap +
// pos_start[own_p[t]] + occ_p[t] (its rank among the item's positive occurrences); its item ids become slots.
// Outputs are pre-filled with padding (U -1, slots 0, own -1).
__global__ __launch_bounds__(kBlock) void plan_finalize_kernel(
    const int64_t* __restrict__ U, int64_t n_tot, int64_t cap, int table_bits, const int32_t* __restrict__ own,
    const int32_t* __restrict__ occ_p, const int32_t* __restrict__ pos_start, const int32_t* __restrict__ slot_of,
    int64_t* __restrict__ U2, int64_t* __restrict__ SP, int64_t* __restrict__ SN, int32_t* __restrict__ own2) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < n_tot; t += stride) {
    const int64_t u = U[t];
    if (u < 0) continue;
    const int64_t s = static_cast<uint32_t>(t) / static_cast<uint32_t>(cap), tab = s << table_bits;  // n_tot < 2^31
    const int32_t ou = own[t], op = own[n_tot + t], on = own[2 * n_tot + t];
    if (op < 0 || on < 0) continue;  // cannot happen for a live triple: its rows are in the table
    const int64_t at = s * cap + pos_start[tab + op] + occ_p[t];
    U2[at] = u;
    SP[at] = slot_of[tab + op];
    SN[at] = slot_of[tab + on];
    own2[at] = ou;
    own2[n_tot + at] = op;
    own2[2 * n_tot + at] = on;
  }
}

}  // namespace
}  // namespace hiprec

using namespace hiprec;

static inline int tiles_per_step(int64_t bs) { return static_cast<int>((bs + kPlanTile - 1) / kPlanTile); }

extern "C" int64_t hiprec_plan_route_tiles(int64_t n, int64_t batch) {
  if (n <= 0 || batch <= 0) return 0;
  return (n + batch - 1) / batch * tiles_per_step(batch);
}

extern "C" int hiprec_plan_route_triples(const int64_t* users, const int64_t* pos, const int64_t* neg,
                                         const int64_t* perm, int64_t n, int64_t batch, int32_t world,
                                         int64_t n_users, int64_t n_items, int32_t* tile_ws, int32_t* cnt_ds,
                                         int32_t* send, hiprec_stats* stats, void* stream) {
  HIPREC_REQUIRE(n >= 0 && batch > 0 && world > 0 && world <= kPlanMaxDest, "bad sizes (1 <= world <= 64)");
  HIPREC_REQUIRE(n_users > 0 && n_items > 0 && n_users < (1ll << 31) && n_items < (1ll << 31) && n < (1ll << 31),
                 "the epoch planner works on 32-bit ids and positions");
  HIPREC_REQUIRE(cnt_ds && stats, "NULL pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t S = (n + batch - 1) / batch;
  HIPREC_TRY(hipMemsetAsync(cnt_ds, 0, sizeof(int32_t) * world * std::max<int64_t>(S, 1), st));
  if (n == 0) return 0;
  HIPREC_REQUIRE(users && pos && neg && tile_ws && send, "NULL pointer");
  const int tps = tiles_per_step(batch);
  const int64_t n_tiles = S * tps;
  HIPREC_REQUIRE(n_tiles * world < (1ll << 31), "too many routing tiles");
  const int grid = static_cast<int>(std::min<int64_t>(n_tiles, 1 << 20));
  plan_route_count_kernel<<<grid, kBlock, 0, st>>>(users, pos, neg, perm, n, batch, tps, n_tiles, world,
                                                   static_cast<int>(S), n_users, n_items, tile_ws, cnt_ds, stats);
  plan_exclusive_scan_kernel<<<1, kPlanThreads, 0, st>>>(tile_ws, n_tiles * world);
  plan_route_pack_kernel<<<grid, kBlock, 0, st>>>(users, pos, neg, perm, n, batch, tps, n_tiles, world, n_users,
                                                  n_items, tile_ws, send);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_plan_place_triples(const int32_t* recv, int64_t n_recv, const int32_t* recv_cnt, int32_t world,
                                         int64_t n_steps, int64_t cap, int32_t* group_ws, int64_t* users,
                                         int64_t* pos, int64_t* neg, void* stream) {
  HIPREC_REQUIRE(n_recv >= 0 && world > 0 && world <= kPlanMaxDest && n_steps > 0 && cap > 0, "bad sizes");
  HIPREC_REQUIRE(n_steps * cap < (1ll << 31) && world * n_steps < (1ll << 30), "plan too large for 32-bit positions");
  HIPREC_REQUIRE(recv_cnt && group_ws && users && pos && neg && (n_recv == 0 || recv), "NULL pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int G = static_cast<int>(world * n_steps);
  int32_t* start = group_ws;          // [G + 1]
  int32_t* dst = group_ws + G + 1;    // [G]
  int32_t* fill = dst + G;            // [n_steps]: triples per step; behind them: padding (user -1, items 0)
  plan_scan_groups_kernel<<<1, kPlanThreads, 0, st>>>(recv_cnt, world, static_cast<int>(n_steps), cap, 0, start, dst,
                                                      nullptr, nullptr, fill);
  plan_pad_kernel<<<dim3(4, static_cast<unsigned>(std::min<int64_t>(n_steps, 1024))), kBlock, 0, st>>>(
      fill, static_cast<int>(n_steps), cap, users, pos, neg, nullptr);
  if (n_recv > 0)
    plan_place_triples_kernel<<<grid_for_threads(n_recv), kBlock, 0, st>>>(recv, n_recv, start, dst, G, users, pos, neg);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_plan_place_requests(const int32_t* incoming, int64_t n_in, const int32_t* in_cnt, int32_t world,
                                          int64_t n_steps, int32_t* group_ws, int32_t* in_idx, int32_t* step_off,
                                          int32_t* extra_pos, int64_t n_rows_local, uint32_t* dup_ws, void* stream) {
  HIPREC_REQUIRE(n_in >= 0 && world > 0 && world <= kPlanMaxDest && n_steps > 0, "bad sizes");
  HIPREC_REQUIRE(n_in + world * n_steps < (1ll << 31), "plan too large for 32-bit positions");
  HIPREC_REQUIRE(in_cnt && group_ws && in_idx && step_off && extra_pos && (n_in == 0 || incoming), "NULL pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIPREC_TRY(hipMemsetAsync(in_idx, 0xFF, sizeof(int32_t) * (n_in + world * n_steps), st));  // extra rows stay -1
  const int G = static_cast<int>(world * n_steps);
  int32_t* start = group_ws;
  int32_t* dst = group_ws + G + 1;
  plan_scan_groups_kernel<<<1, kPlanThreads, 0, st>>>(in_cnt, world, static_cast<int>(n_steps), 0, 1, start, dst,
                                                      step_off, extra_pos, nullptr);
  if (n_in > 0)
    plan_place_values_kernel<<<grid_for_threads(n_in), kBlock, 0, st>>>(incoming, n_in, start, dst, G, in_idx);
  if (dup_ws != nullptr) {   // [n_steps][words]: the duplicate bits the step driver reads (every word is written)
    HIPREC_REQUIRE(n_rows_local >= 0, "bad n_rows_local");
    const int64_t words = (n_rows_local + 31) / 32;
    if (words > 0) {
      const int n_parts = static_cast<int>((words + kDupWords - 1) / kDupWords);
      const size_t lds = sizeof(uint32_t) * 2 * static_cast<size_t>(std::min<int64_t>(words, kDupWords));
      static std::atomic<uint64_t> lds_ok{0};
      if (int rc = allow_dynamic_lds({reinterpret_cast<const void*>(plan_mark_duplicates_kernel)},
                                     sizeof(uint32_t) * 2 * kDupWords, lds_ok, "the planner's duplicate marks"))
        return rc;
      plan_mark_duplicates_kernel<<<static_cast<int>(n_steps) * n_parts, kPlanThreads, lds, st>>>(in_idx, step_off, words,
                                                                                                n_parts, dup_ws);
    }
  }
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int64_t hiprec_plan_slot_ws_ints(int64_t n_steps, int32_t table_bits, int32_t world) {
  const int part_bits = std::min<int>(table_bits, kPlanPartBits);
  const int64_t n_parts = 1ll << (table_bits - part_bits);
  // part_cnt [S * n_parts * (R + 1)] | slot_base [S * n_parts * R] | pos_base [S * n_parts] | chunk_start [S * R]
  return n_steps * n_parts * (world + 1) + n_steps * n_parts * world + n_steps * n_parts + n_steps * world;
}

extern "C" int hiprec_plan_item_slots(const int64_t* users, int64_t n_steps, int64_t cap, int32_t world,
                                      int64_t n_users_local, int32_t table_bits, const int32_t* own,
                                      const int32_t* occ, const int32_t* tab_keys, int32_t* pos_cnt, int32_t* ws,
                                      int32_t* slot_of, int32_t* req_cnt, int32_t* req_ds, int32_t* ex_req,
                                      int32_t* n_slots, int32_t* send_base, int32_t* req_send, int64_t* users_out,
                                      int64_t* pos_slot, int64_t* neg_slot, int32_t* own_out, const int32_t* total,
                                      uint8_t* slot_shared, int64_t slot_stride, const int32_t* step_fill,
                                      void* stream) {
  HIPREC_REQUIRE(n_steps > 0 && cap > 0 && world > 0 && world <= kPlanMaxDest && n_users_local >= 0, "bad sizes");
  HIPREC_REQUIRE(table_bits >= 2 && table_bits <= 30 && (n_steps << table_bits) < (1ll << 31) &&
                     n_steps * cap < (1ll << 31),
                 "plan too large for 32-bit positions");
  HIPREC_REQUIRE(users && own && occ && tab_keys && pos_cnt && ws && slot_of && req_cnt && req_ds && ex_req &&
                     n_slots && send_base && req_send && users_out && pos_slot && neg_slot && own_out,
                 "NULL pointer");
  HIPREC_REQUIRE(slot_shared == nullptr || (total != nullptr && slot_stride >= 2 * cap + world),
                 "slot_shared needs total and a stride of at least 2 * cap + world");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int part_bits = std::min<int>(table_bits, kPlanPartBits);
  const int n_parts = 1 << (table_bits - part_bits);
  const int S = static_cast<int>(n_steps), R = world;
  int32_t* part_cnt = ws;
  int32_t* slot_base = part_cnt + static_cast<int64_t>(S) * n_parts * (R + 1);
  int32_t* pos_base = slot_base + static_cast<int64_t>(S) * n_parts * R;
  int32_t* chunk_start = pos_base + static_cast<int64_t>(S) * n_parts;
  const int64_t tot = n_steps * cap;
  if (step_fill != nullptr) {   // the live triples of a step end up in [0, step_fill[s]): pad only what lies behind
    plan_pad_kernel<<<dim3(4, static_cast<unsigned>(std::min<int64_t>(n_steps, 1024))), kBlock, 0, st>>>(
        step_fill, S, cap, users_out, pos_slot, neg_slot, own_out);
  } else {
    HIPREC_TRY(hipMemsetAsync(users_out, 0xFF, sizeof(int64_t) * tot, st));
    HIPREC_TRY(hipMemsetAsync(pos_slot, 0, sizeof(int64_t) * tot, st));
    HIPREC_TRY(hipMemsetAsync(neg_slot, 0, sizeof(int64_t) * tot, st));
    HIPREC_TRY(hipMemsetAsync(own_out, 0xFF, sizeof(int32_t) * 3 * tot, st));
  }
  const int grid = S * n_parts;
  plan_slot_count_kernel<<<grid, kPlanThreads, 0, st>>>(tab_keys, pos_cnt, table_bits, R,
                                                        static_cast<int32_t>(n_users_local), part_cnt);
  plan_slot_scan_kernel<<<1, kPlanThreads, 0, st>>>(part_cnt, S, n_parts, R, req_cnt, req_ds, chunk_start, ex_req,
                                                    n_slots, slot_base, pos_base, send_base, slot_shared, slot_stride);
  plan_slot_assign_kernel<<<grid, kPlanThreads, 0, st>>>(tab_keys, pos_cnt, table_bits, n_parts, R, S,
                                                         static_cast<int32_t>(n_users_local), slot_base, pos_base,
                                                         chunk_start, send_base, slot_of, req_send, total, slot_shared,
                                                         slot_stride);
  plan_finalize_kernel<<<grid_for_threads(tot), kBlock, 0, st>>>(users, tot, cap, table_bits, own, occ + tot, pos_cnt,
                                                                 slot_of, users_out, pos_slot, neg_slot, own_out);
  HIPREC_TRY(hipGetLastError());
  return 0;
}
