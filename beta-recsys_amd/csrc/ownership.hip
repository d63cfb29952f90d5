// Row ownership of a staged epoch, for the owned-rows SGD step (csrc/mf_owned.hip): which rows occur ONCE in
// their batch (their update is a plain store by the wave that read them) and which several times (their
// contributions meet in a slot).  Part of the device batcher's staging: integer work, independent of the weights.
//
// One open-addressing hash table per batch, 2^table_bits entries (>= 4 x batch: at most 3 x batch distinct rows
// go in); the table position of a row IS its slot id -- no compaction, no sort.  The table is cut into
// partitions of 2^14 entries by the top bits of the hash.  A pre-pass (one workgroup per batch) makes the 32-bit
// keys of the batch's 3 x batch row occurrences (key = user row, or n_users + item row) and buckets them by
// partition; one workgroup then builds one (batch, partition) in LDS from ITS bucket: LDS compare-and-swap +
// linear probing (inside the partition), counts, and every occurrence gets its slot; the partition's counts go to
// total[] in one coalesced write.  A row that occurs once has total[slot] == 1: the step kernel treats it
// exactly like own = -1.  History: tables in global memory, 20 M device atomics per 50-batch epoch, 1.5 ms; LDS
// tables with every partition's workgroup scanning all of its batch's keys, 0.37-0.41 ms; bucketed (round 4).
#include <algorithm>

#include <atomic>

#include "common.hpp"

namespace hiprec {

constexpr int kOwnPartBits = 14;              // 16 384 entries x (key + count) = 128 KB of LDS
constexpr int kOwnThreads = 1024;

__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

constexpr int kOwnMaxParts = 1024;            // table_bits <= 24: batches of up to 4 M triples
constexpr int kBucketThreads = 256;
constexpr int kBucketSlice = 4096;            // triples per workgroup of the pre-pass

__device__ __forceinline__ uint32_t own_part(int32_t key, int table_bits, int part_bits) {
  return table_bits == part_bits ? 0u : hash_u32(static_cast<uint32_t>(key)) >> (32 - (table_bits - part_bits));
}

// Pre-pass (round 4): the batch's 3 x cnt row occurrences become 32-bit keys (user row, or n_users + item row)
// BUCKETED by (phase, partition of the hash): phase 0 = user and positive rows, phase 1 = negative rows.  Occurrence i
// of the batch (role-major inside the batch: [0,cnt) users, [cnt,2cnt) positives, [2cnt,3cnt) negatives) lands in
// bkey / bidx at 3 * t0 + position; boff[b][phase * n_parts + part] = start of a bucket, boff[b][2 * n_parts] = number
// of valid occurrences.  Triples with an out-of-range id get own = -1 and are not bucketed.  (Round 3: every (batch,
// partition) workgroup scanned ALL keys of its batch and kept one in n_parts -- 16 x the key traffic and hash work at
// configs[3].)  Two launches over slices of 4096 triples: COUNT (per-workgroup histogram in LDS -> one global add per
// bucket) and SCATTER (histogram again, one global add per bucket reserves the workgroup's range, LDS cursors inside
// it).  The order inside a bucket is arbitrary: the ownership kernel only needs every key once.
struct BucketTriple {
  int32_t key[3];
  int bucket[3];
  bool ok;
  bool pos_counts;   // false: a positive occurrence that rides in the run of equal items its chunk neighbour heads
};

__device__ __forceinline__ bool triple_in_range(const int64_t* __restrict__ users, const int64_t* __restrict__ pos,
                                                const int64_t* __restrict__ neg, int64_t t, int64_t n_users,
                                                int64_t n_items) {
  return static_cast<uint64_t>(users[t]) < static_cast<uint64_t>(n_users) &&
         static_cast<uint64_t>(pos[t]) < static_cast<uint64_t>(n_items) &&
         static_cast<uint64_t>(neg[t]) < static_cast<uint64_t>(n_items);
}

// chunk > 0 (the contribution lists of the owner-pulls step): the step kernel hands `chunk` consecutive triples of a
// batch to one wave, which sums the run of equal positive items itself -- only the run's HEAD contributes to the row.
// j = the triple's position inside its batch.  A triple with an out-of-range id ends a run (the kernel skips it).
__device__ __forceinline__ BucketTriple bucket_triple(const int64_t* __restrict__ users, const int64_t* __restrict__ pos,
                                                      const int64_t* __restrict__ neg, int64_t t, int64_t n_users,
                                                      int64_t n_items, int table_bits, int part_bits, int n_parts,
                                                      int chunk = 0, int64_t j = 0) {
  BucketTriple r;
  const int64_t u = users[t], p = pos[t], q = neg[t];
  r.ok = static_cast<uint64_t>(u) < static_cast<uint64_t>(n_users) && static_cast<uint64_t>(p) < static_cast<uint64_t>(n_items) &&
         static_cast<uint64_t>(q) < static_cast<uint64_t>(n_items);
  r.pos_counts = true;
  if (chunk > 0 && r.ok && j % chunk != 0)
    r.pos_counts = pos[t - 1] != p || !triple_in_range(users, pos, neg, t - 1, n_users, n_items);
  r.key[0] = static_cast<int32_t>(u);
  r.key[1] = static_cast<int32_t>(n_users + p);
  r.key[2] = static_cast<int32_t>(n_users + q);
  r.bucket[0] = static_cast<int>(own_part(r.key[0], table_bits, part_bits));
  r.bucket[1] = static_cast<int>(own_part(r.key[1], table_bits, part_bits));
  r.bucket[2] = n_parts + static_cast<int>(own_part(r.key[2], table_bits, part_bits));
  return r;
}

// SCATTER = false: counts[b][bucket] += this slice's occurrences (counts zero on entry), own = -1 for bad triples.
// SCATTER = true: counts holds the batch's totals; cursor[b][bucket] (zero on entry) hands out ranges.
template <bool SCATTER>
__global__ __launch_bounds__(kBucketThreads) void ownership_bucket_kernel(
    const int64_t* __restrict__ users, const int64_t* __restrict__ pos, const int64_t* __restrict__ neg, int64_t n,
    int64_t batch, int64_t n_users, int64_t n_items, int table_bits, int slices, int32_t* __restrict__ counts,
    int32_t* __restrict__ cursor, int32_t* __restrict__ bkey, uint32_t* __restrict__ bidx, int32_t* __restrict__ boff,
    int32_t* __restrict__ own, int chunk) {
  __shared__ int32_t s_hist[2 * kOwnMaxParts];
  __shared__ int32_t s_base[2 * kOwnMaxParts];
  const int part_bits = table_bits < kOwnPartBits ? table_bits : kOwnPartBits;
  const int n_parts = 1 << (table_bits - part_bits), nb = 2 * n_parts;
  const int64_t b = blockIdx.x / slices, t0 = b * batch, cnt = min<int64_t>(batch, n - t0);
  const int64_t j0 = static_cast<int64_t>(blockIdx.x % slices) * kBucketSlice, j1 = min<int64_t>(cnt, j0 + kBucketSlice);
  if (j0 >= cnt && !(SCATTER && blockIdx.x % slices == 0)) return;
  for (int i = threadIdx.x; i < nb; i += kBucketThreads) s_hist[i] = 0;
  __syncthreads();
  for (int64_t j = j0 + threadIdx.x; j < j1; j += kBucketThreads) {
    const BucketTriple r = bucket_triple(users, pos, neg, t0 + j, n_users, n_items, table_bits, part_bits, n_parts, chunk, j);
    if (!r.ok) {
      if (!SCATTER) own[t0 + j] = own[n + t0 + j] = own[2 * n + t0 + j] = -1;
      continue;
    }
    if (!SCATTER && !r.pos_counts) own[n + t0 + j] = -2;   // rides in its chunk neighbour's run
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k != 1 || r.pos_counts) atomicAdd(&s_hist[r.bucket[k]], 1);
  }
  __syncthreads();
  int32_t* cb = counts + b * static_cast<int64_t>(nb);
  if constexpr (!SCATTER) {
    for (int i = threadIdx.x; i < nb; i += kBucketThreads)
      if (s_hist[i]) atomicAdd(cb + i, s_hist[i]);
    return;
  } else {
    // bucket starts of the batch (every workgroup scans the <= 2048 totals itself), this slice's range inside each
    if (threadIdx.x == 0) {
      int32_t run = 0;
      for (int i = 0; i < nb; ++i) {
        const int32_t c = cb[i];
        s_base[i] = run;
        run += c;
      }
      if (blockIdx.x % slices == 0) {
        int32_t* bo = boff + b * (static_cast<int64_t>(nb) + 1);
        for (int i = 0; i < nb; ++i) bo[i] = s_base[i];
        bo[nb] = run;
      }
    }
    __syncthreads();
    int32_t* cur = cursor + b * static_cast<int64_t>(nb);
    for (int i = threadIdx.x; i < nb; i += kBucketThreads) {
      const int32_t h = s_hist[i];
      s_base[i] += h ? atomicAdd(cur + i, h) : 0;
      s_hist[i] = 0;   // now the cursor inside the slice's range
    }
    __syncthreads();
    int32_t* k_out = bkey + 3 * t0;
    uint32_t* i_out = bidx + 3 * t0;
    for (int64_t j = j0 + threadIdx.x; j < j1; j += kBucketThreads) {
      const BucketTriple r = bucket_triple(users, pos, neg, t0 + j, n_users, n_items, table_bits, part_bits, n_parts, chunk, j);
      if (!r.ok) continue;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (k == 1 && !r.pos_counts) continue;
        const int at = s_base[r.bucket[k]] + atomicAdd(&s_hist[r.bucket[k]], 1);
        k_out[at] = r.key[k];
        i_out[at] = static_cast<uint32_t>(k * cnt + j);
      }
    }
  }
}

// tab_keys / pos_cnt / occ (all optional) serve the row-sharded engine's epoch planner (csrc/plan.hip): tab_keys =
// the table itself (the row key that sits in every entry, -1 = empty); the user and positive-item occurrences are
// inserted BEFORE the negative ones, so that an entry's count at that point -- pos_cnt -- is the number of POSITIVE
// occurrences of an item row, and occ[at] (the entry's count when occurrence `at` arrived) is, for a positive
// occurrence, its rank among them: the planner lays every batch out grouped by positive item from these two,
// without a sort.
// One workgroup per (batch, partition): it reads ITS bucket of the pre-pass, every lane inserts one key per trip (the
// probe loop is a chain of LDS compare-and-swap round trips whose length is the longest chain among the active lanes).
__global__ __launch_bounds__(kOwnThreads) void ownership_kernel(const int32_t* __restrict__ bkey,
                                                                const uint32_t* __restrict__ bidx,
                                                                const int32_t* __restrict__ boff, int64_t n,
                                                                int64_t batch, int table_bits,
                                                                int32_t* __restrict__ total,
                                                                int32_t* __restrict__ own,
                                                                int32_t* __restrict__ tab_keys,
                                                                int32_t* __restrict__ pos_cnt,
                                                                int32_t* __restrict__ occ) {
  extern __shared__ int32_t s_tab[];          // [part_size] keys, then [part_size] counts
  const int part_bits = table_bits < kOwnPartBits ? table_bits : kOwnPartBits;
  const uint32_t part_size = 1u << part_bits, part_mask = part_size - 1u;
  const int n_parts = 1 << (table_bits - part_bits);
  const int64_t b = blockIdx.x / n_parts;
  const uint32_t part = static_cast<uint32_t>(blockIdx.x % n_parts);
  int32_t* s_key = s_tab;
  int32_t* s_cnt = s_tab + part_size;
  for (uint32_t i = threadIdx.x; i < part_size; i += kOwnThreads) {
    s_key[i] = -1;
    s_cnt[i] = 0;
  }
  __syncthreads();
  const int64_t t0 = b * batch;
  const int64_t cnt = min<int64_t>(batch, n - t0);
  const int64_t tab0 = (b << table_bits) + (static_cast<int64_t>(part) << part_bits);
  const int32_t* off = boff + b * (2 * static_cast<int64_t>(n_parts) + 1);
  const int32_t* keys = bkey + 3 * t0;
  const uint32_t* idx = bidx + 3 * t0;
  // phase 0: user and positive rows, phase 1: negative rows
  for (int phase = 0; phase < 2; ++phase) {
    const int lo = off[phase * n_parts + part], hi = off[phase * n_parts + part + 1];
    for (int j = lo + static_cast<int>(threadIdx.x); j < hi; j += kOwnThreads) {
      const int32_t key = keys[j];
      const uint32_t i = idx[j];
      const int role = i < cnt ? 0 : i < 2 * cnt ? 1 : 2;
      const int64_t at = role * n + t0 + (static_cast<int64_t>(i) - role * cnt);
      uint32_t h = hash_u32(static_cast<uint32_t>(key)) & part_mask;
      for (uint32_t probes = 0;; ++probes) {
        const int32_t prev = atomicCAS(s_key + h, -1, key);
        if (prev == -1 || prev == key) break;
        h = (h + 1u) & part_mask;
        if (probes > part_size) {  // a full partition (cannot happen at >= 4 x batch entries): give up, never spin
          h = part_size;
          break;
        }
      }
      if (h == part_size) {
        own[at] = -1;
        continue;
      }
      const int32_t before = atomicAdd(s_cnt + h, 1);
      own[at] = static_cast<int32_t>((part << part_bits) | h);
      if (occ) occ[at] = before;
    }
    __syncthreads();
    if (phase == 0 && pos_cnt) {
      for (uint32_t i = threadIdx.x; i < part_size; i += kOwnThreads) pos_cnt[tab0 + i] = s_cnt[i];
      // the snapshot must be complete before any wave's phase 1 adds negative occurrences to s_cnt
      __syncthreads();
    }
  }
  for (uint32_t i = threadIdx.x; i < part_size; i += kOwnThreads) {
    total[tab0 + i] = s_cnt[i];
    if (tab_keys) tab_keys[tab0 + i] = s_key[i];
  }
}

// ---- contribution lists for the owner-pulls step (csrc/mf_owned.hip, hiprec_mf_bpr_epoch_pull) ------------------------
// The same tables, counting CONTRIBUTIONS instead of occurrences (a wave's run of equal positive items is one
// contribution, made by its head).  A row with one contribution is updated in place by its contributor (cidx = -1).
// A row with several gets a contiguous range of the step's contribution buffer -- cidx = range start + arrival rank:
// the contributor stores its part of the gradient THERE with plain stores -- and one record {row key, range start,
// contributions} in the batch's row list; after the gradient launch one wave per record sums the range and writes
// w - lr * g: no float atomics, no arrival counters, nothing to clear.  min_c = 1 (the lazy Adam / RMSprop form,
// csrc/lazy_opt.hip): EVERY row gets a record and a range, a row with one contribution included -- its optimizer step
// needs the moments, which the gradient launch does not carry.  Records of rows with more than
// kContribLongRow contributions (a Zipf head item at configs[3]: ~650 runs) are listed from the END of the list and
// taken by a whole workgroup each.  counts[b] = {short rows, long rows, contributions, -}.
constexpr int kContribLongRow = 32;

__global__ __launch_bounds__(kOwnThreads) void contrib_kernel(int32_t* __restrict__ bkey,
                                                              const uint32_t* __restrict__ bidx,
                                                              const int32_t* __restrict__ boff, int64_t n,
                                                              int64_t batch, int table_bits,
                                                              int32_t* __restrict__ cidx, int4* __restrict__ rows,
                                                              int64_t row_cap, int32_t* __restrict__ counts,
                                                              int min_c) {
  extern __shared__ int32_t s_tab[];          // [part_size] keys, then [part_size] counts -> range starts
  __shared__ int32_t s_wave[3][kOwnThreads / kWave];
  __shared__ int32_t s_base[3];
  const int part_bits = table_bits < kOwnPartBits ? table_bits : kOwnPartBits;
  const uint32_t part_size = 1u << part_bits, part_mask = part_size - 1u;
  const int n_parts = 1 << (table_bits - part_bits);
  const int64_t b = blockIdx.x / n_parts;
  const uint32_t part = static_cast<uint32_t>(blockIdx.x % n_parts);
  int32_t* s_key = s_tab;
  int32_t* s_cnt = s_tab + part_size;
  for (uint32_t i = threadIdx.x; i < part_size; i += kOwnThreads) {
    s_key[i] = -1;
    s_cnt[i] = 0;
  }
  __syncthreads();
  const int64_t t0 = b * batch;
  const int64_t cnt = min<int64_t>(batch, n - t0);
  const int32_t* off = boff + b * (2 * static_cast<int64_t>(n_parts) + 1);
  int32_t* keys = bkey + 3 * t0;
  const uint32_t* idx = bidx + 3 * t0;
  auto where = [&](uint32_t i) {
    const int role = i < cnt ? 0 : i < 2 * cnt ? 1 : 2;
    return role * n + t0 + (static_cast<int64_t>(i) - role * cnt);
  };
  // pass 1: insert every contribution; its table entry replaces its key in the bucket, its arrival rank waits in cidx
  for (int phase = 0; phase < 2; ++phase) {
    const int lo = off[phase * n_parts + part], hi = off[phase * n_parts + part + 1];
    for (int j = lo + static_cast<int>(threadIdx.x); j < hi; j += kOwnThreads) {
      const int32_t key = keys[j];
      uint32_t h = hash_u32(static_cast<uint32_t>(key)) & part_mask;
      for (uint32_t probes = 0;; ++probes) {
        const int32_t prev = atomicCAS(s_key + h, -1, key);
        if (prev == -1 || prev == key) break;
        h = (h + 1u) & part_mask;
        if (probes > part_size) {  // a full partition (cannot happen at >= 4 x batch entries): give up, never spin --
          h = part_size;           // and say so: the step that consumes this batch's lists raises HIPREC_STATUS_TABLE_FULL
          atomicOr(counts + 4 * b + 3, 1);
          break;
        }
      }
      keys[j] = static_cast<int32_t>(h);
      if (h != part_size) cidx[where(idx[j])] = atomicAdd(s_cnt + h, 1);
    }
  }
  __syncthreads();
  // scan: every thread owns kContribPerThread consecutive entries (fewer for small tables)
  const uint32_t per = part_size >= kOwnThreads ? part_size / kOwnThreads : 1;
  const uint32_t e0 = threadIdx.x * per;
  int32_t my_c = 0, my_s = 0, my_l = 0;
  if (e0 < part_size)
    for (uint32_t e = e0; e < e0 + per; ++e) {
      const int32_t c = s_cnt[e];
      if (c >= min_c) {
        my_c += c;
        if (c > kContribLongRow) ++my_l;
        else ++my_s;
      }
    }
  int32_t in_c = my_c, in_s = my_s, in_l = my_l;
  const int lane = lane_id(), wv = wave_in_block();
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int32_t uc = __shfl_up(in_c, o), us = __shfl_up(in_s, o), ul = __shfl_up(in_l, o);
    if (lane >= o) {
      in_c += uc;
      in_s += us;
      in_l += ul;
    }
  }
  if (lane == kWave - 1) {
    s_wave[0][wv] = in_c;
    s_wave[1][wv] = in_s;
    s_wave[2][wv] = in_l;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t tc = 0, ts = 0, tl = 0;
    for (int w = 0; w < kOwnThreads / kWave; ++w) {
      const int32_t c = s_wave[0][w], s_ = s_wave[1][w], l = s_wave[2][w];
      s_wave[0][w] = tc;
      s_wave[1][w] = ts;
      s_wave[2][w] = tl;
      tc += c;
      ts += s_;
      tl += l;
    }
    int32_t* cb = counts + 4 * b;
    s_base[0] = tc ? atomicAdd(cb + 2, tc) : 0;
    s_base[1] = ts ? atomicAdd(cb + 0, ts) : 0;
    s_base[2] = tl ? atomicAdd(cb + 1, tl) : 0;
  }
  __syncthreads();
  if (e0 < part_size) {
    int32_t at_c = s_base[0] + s_wave[0][wv] + in_c - my_c;
    int32_t at_s = s_base[1] + s_wave[1][wv] + in_s - my_s;
    int32_t at_l = s_base[2] + s_wave[2][wv] + in_l - my_l;
    int4* rb = rows + b * row_cap;
    for (uint32_t e = e0; e < e0 + per; ++e) {
      const int32_t c = s_cnt[e];
      if (c >= min_c) {
        const int4 rec = make_int4(s_key[e], at_c, c, 0);
        if (c > kContribLongRow) rb[row_cap - 1 - at_l++] = rec;
        else rb[at_s++] = rec;
        s_cnt[e] = at_c;
        at_c += c;
      } else {
        s_cnt[e] = -1;
      }
    }
  }
  __syncthreads();
  // pass 2: range start + arrival rank
  for (int phase = 0; phase < 2; ++phase) {
    const int lo = off[phase * n_parts + part], hi = off[phase * n_parts + part + 1];
    for (int j = lo + static_cast<int>(threadIdx.x); j < hi; j += kOwnThreads) {
      const uint32_t h = static_cast<uint32_t>(keys[j]);
      const int64_t at = where(idx[j]);
      if (h == part_size) {
        cidx[at] = -1;
        continue;
      }
      const int32_t start = s_cnt[h];
      cidx[at] = start < 0 ? -1 : start + cidx[at];
    }
  }
}

// ---- an epoch laid out with every batch GROUPED BY POSITIVE ITEM, without a sort -----------------------------------------
// (What torch.argsort + three index gathers did for batches too large for the staging kernel's LDS sort: the gradient
// kernels sum adjacent equal positive items before they touch memory.)  The ownership tables already hold the counting
// sort: after the first phase an ITEM entry's count is its number of positive occurrences (pos_cnt) and a positive
// occurrence's arrival rank (occ) is its place inside the item's group.  An exclusive scan of pos_cnt over a batch's
// table gives every item's first position; one scatter moves the triples and their ownership slots there.

// position j of the epoch's visiting order -> index into the loader's arrays: perm[j] / P_seed(j) / j
__device__ __forceinline__ int64_t visit_index(int64_t j, const int64_t* __restrict__ perm, int shuffle, int half_bits,
                                               uint64_t seed, int64_t n) {
  if (perm) return perm[j];
  if (shuffle) return static_cast<int64_t>(feistel_permute(static_cast<uint64_t>(j), static_cast<uint64_t>(n), half_bits, seed));
  return j;
}

// the epoch laid out: out[j] = in[visit(order ? order[j] : j)] for the three arrays, one launch
__global__ __launch_bounds__(kBlock) void gather_epoch_kernel(const int64_t* __restrict__ users,
                                                              const int64_t* __restrict__ pos,
                                                              const int64_t* __restrict__ neg,
                                                              const int64_t* __restrict__ perm, int shuffle,
                                                              int half_bits, uint64_t seed,
                                                              const int64_t* __restrict__ order, int64_t n,
                                                              int64_t* __restrict__ ou, int64_t* __restrict__ op,
                                                              int64_t* __restrict__ on) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; j < n; j += stride) {
    const int64_t i = visit_index(order ? order[j] : j, perm, shuffle, half_bits, seed, n);
    ou[j] = users[i];
    op[j] = pos[i];
    on[j] = neg[i];
  }
}

// sort key of visiting position j: (its batch, its positive item) -- a stable sort by it groups every batch by item in
// ascending row order (neighbouring groups are neighbouring table rows)
template <class K>
__global__ __launch_bounds__(kBlock) void stage_sort_keys_kernel(const int64_t* __restrict__ items,
                                                                 const int64_t* __restrict__ perm, int shuffle,
                                                                 int half_bits, uint64_t seed, int64_t n, int64_t batch,
                                                                 int64_t n_items, K* __restrict__ keys) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; j < n; j += stride) {
    int64_t it = items[visit_index(j, perm, shuffle, half_bits, seed, n)];
    it = it < 0 ? 0 : it >= n_items ? n_items - 1 : it;   // out-of-range ids are flagged by the step, not here
    keys[j] = static_cast<K>((j / batch) * n_items + it);
  }
}

// ---- batches beyond the staging kernel's LDS sort, grouped by positive item WITHOUT a device sort (round 5) ---------
// What hiprec_stage_sort_keys + torch.sort (rocprim onesweep: five launches) + hiprec_gather_epoch did, as a two-level
// counting sort in three launches of this library: (1) histogram of the batch's positive items over RANGES of
// kStageRange consecutive item ids, (2) the occurrences scattered into their range's bucket (position inside the batch
// + item), (3) one workgroup per (batch, range) counts the range's items in LDS, scans the 4096 counters and writes
// every triple of the bucket to its place -- reading users / pos / neg through the visiting order (perm[], the Feistel
// shuffle evaluated on the fly, or sequential), i.e. the gather is folded in.  Result: every batch sorted by positive
// item (ascending; the order INSIDE a group of equal items is whatever the LDS atomics make it: a batch's loss and
// gradients are sums over its triples).  Out-of-range items are clamped for the grouping only (the step flags them).
constexpr int kStageRange = 4096;             // item ids per range = LDS counters of the third launch
constexpr int kStageMaxRanges = 2048;         // n_items <= 8 M (beyond: the caller keeps the sort)
constexpr int kStageThreads = 1024;

__device__ __forceinline__ int32_t stage_item(const int64_t* __restrict__ items, int64_t i, int64_t n_items) {
  const int64_t it = items[i];
  return static_cast<int32_t>(it < 0 ? 0 : it >= n_items ? n_items - 1 : it);
}

// SCATTER = false: counts[b][range] += this slice's occurrences.  SCATTER = true: counts holds the batch's totals; the
// slice reserves its part of every bucket (cursor, zero on entry) and writes (position in the batch, item).
template <bool SCATTER>
__global__ __launch_bounds__(kBucketThreads) void stage_range_kernel(
    const int64_t* __restrict__ pos, const int64_t* __restrict__ perm, int shuffle, int half_bits, uint64_t seed, int64_t n,
    int64_t batch, int64_t n_items, int n_ranges, int slices, int32_t* __restrict__ counts, int32_t* __restrict__ cursor,
    int32_t* __restrict__ boff, int32_t* __restrict__ bitem, uint32_t* __restrict__ bidx) {
  __shared__ int32_t s_hist[kStageMaxRanges];
  __shared__ int32_t s_base[kStageMaxRanges];
  const int64_t b = blockIdx.x / slices, t0 = b * batch, cnt = min<int64_t>(batch, n - t0);
  const int64_t j0 = static_cast<int64_t>(blockIdx.x % slices) * kBucketSlice, j1 = min<int64_t>(cnt, j0 + kBucketSlice);
  if (j0 >= cnt && !(SCATTER && blockIdx.x % slices == 0)) return;
  for (int i = threadIdx.x; i < n_ranges; i += kBucketThreads) s_hist[i] = 0;
  __syncthreads();
  for (int64_t j = j0 + threadIdx.x; j < j1; j += kBucketThreads) {
    const int32_t it = stage_item(pos, visit_index(t0 + j, perm, shuffle, half_bits, seed, n), n_items);
    atomicAdd(&s_hist[it / kStageRange], 1);
  }
  __syncthreads();
  int32_t* cb = counts + b * static_cast<int64_t>(n_ranges);
  if constexpr (!SCATTER) {
    for (int i = threadIdx.x; i < n_ranges; i += kBucketThreads)
      if (s_hist[i]) atomicAdd(cb + i, s_hist[i]);
    return;
  } else {
    if (threadIdx.x == 0) {
      int32_t run = 0;
      for (int i = 0; i < n_ranges; ++i) {
        const int32_t c = cb[i];
        s_base[i] = run;
        run += c;
      }
      if (blockIdx.x % slices == 0) {
        int32_t* bo = boff + b * (static_cast<int64_t>(n_ranges) + 1);
        for (int i = 0; i < n_ranges; ++i) bo[i] = s_base[i];
        bo[n_ranges] = run;
      }
    }
    __syncthreads();
    int32_t* cur = cursor + b * static_cast<int64_t>(n_ranges);
    for (int i = threadIdx.x; i < n_ranges; i += kBucketThreads) {
      const int32_t h = s_hist[i];
      s_base[i] += h ? atomicAdd(cur + i, h) : 0;
      s_hist[i] = 0;
    }
    __syncthreads();
    for (int64_t j = j0 + threadIdx.x; j < j1; j += kBucketThreads) {
      const int32_t it = stage_item(pos, visit_index(t0 + j, perm, shuffle, half_bits, seed, n), n_items);
      const int r = it / kStageRange;
      const int at = s_base[r] + atomicAdd(&s_hist[r], 1);
      bitem[t0 + at] = it;
      bidx[t0 + at] = static_cast<uint32_t>(j);
    }
  }
}

// one workgroup per (batch, range): counting sort of the bucket by item, the triples written to their places
__global__ __launch_bounds__(kStageThreads) void stage_place_kernel(
    const int64_t* __restrict__ users, const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
    const int64_t* __restrict__ perm, int shuffle, int half_bits, uint64_t seed, int64_t n, int64_t batch, int n_ranges,
    const int32_t* __restrict__ boff, const int32_t* __restrict__ bitem, const uint32_t* __restrict__ bidx,
    int64_t* __restrict__ ou, int64_t* __restrict__ op, int64_t* __restrict__ on) {
  __shared__ int32_t s_cnt[kStageRange];      // occurrences per item of the range -> running cursor
  __shared__ int32_t s_wave[kStageThreads / kWave];
  const int64_t b = blockIdx.x / n_ranges;
  const int r = static_cast<int>(blockIdx.x % n_ranges);
  const int64_t t0 = b * batch;
  const int32_t* off = boff + b * (static_cast<int64_t>(n_ranges) + 1);
  const int lo = off[r], hi = off[r + 1];
  if (lo == hi) return;
  for (int i = threadIdx.x; i < kStageRange; i += kStageThreads) s_cnt[i] = 0;
  __syncthreads();
  const int32_t first = r * kStageRange;
  for (int e = lo + static_cast<int>(threadIdx.x); e < hi; e += kStageThreads) atomicAdd(&s_cnt[bitem[t0 + e] - first], 1);
  __syncthreads();
  // exclusive scan of the 4096 counters: 4 per thread, waves, workgroup
  constexpr int PER = kStageRange / kStageThreads;
  const int lane = lane_id(), wv = wave_in_block();
  int32_t c[PER], mine = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    c[k] = s_cnt[threadIdx.x * PER + k];
    mine += c[k];
  }
  int32_t incl = mine;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int32_t up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == kWave - 1) s_wave[wv] = incl;
  __syncthreads();
  int32_t carry = 0;
  for (int w = 0; w < wv; ++w) carry += s_wave[w];
  int32_t run = lo + carry + incl - mine;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    s_cnt[threadIdx.x * PER + k] = run;     // where the item's group starts inside the batch
    run += c[k];
  }
  __syncthreads();
  for (int e = lo + static_cast<int>(threadIdx.x); e < hi; e += kStageThreads) {
    const int at = atomicAdd(&s_cnt[bitem[t0 + e] - first], 1);
    const int64_t i = visit_index(t0 + bidx[t0 + e], perm, shuffle, half_bits, seed, n);
    ou[t0 + at] = users[i];
    op[t0 + at] = pos[i];
    on[t0 + at] = neg[i];
  }
}

constexpr int kGroupThreads = 1024;
constexpr int kGroupWaves = kGroupThreads / kWave;

// one workgroup per batch: pos_cnt[entry] <- first position of the entry's item inside the batch (entries of user
// rows and empty entries contribute nothing).  Every wave owns a contiguous share of the table: totals first, then the
// prefix with the carry of the shares before it.
__global__ __launch_bounds__(kGroupThreads) void group_scan_kernel(const int32_t* __restrict__ tab_keys,
                                                                   int32_t* __restrict__ pos_cnt, int table_bits,
                                                                   int32_t n_users) {
  __shared__ int32_t s_tot[kGroupWaves];
  const int64_t T = 1ll << table_bits, base = static_cast<int64_t>(blockIdx.x) << table_bits;
  const int lane = lane_id(), wv = wave_in_block();
  const int64_t share = (T + kGroupWaves - 1) / kGroupWaves, lo = wv * share, hi = min<int64_t>(T, lo + share);
  auto count = [&](int64_t i) { return i < hi && tab_keys[base + i] >= n_users ? pos_cnt[base + i] : 0; };
  int32_t tot = 0;
  for (int64_t i = lo + lane; i < hi; i += kWave) tot += count(i);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
  if (lane == 0) s_tot[wv] = tot;
  __syncthreads();
  int32_t carry = 0;
  for (int w = 0; w < wv; ++w) carry += s_tot[w];
  for (int64_t i0 = lo; i0 < hi; i0 += kWave) {
    const int64_t i = i0 + lane;
    const int32_t c = count(i);
    int32_t incl = c;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int32_t up = __shfl_up(incl, off);
      if (lane >= off) incl += up;
    }
    if (i < hi) pos_cnt[base + i] = carry + incl - c;
    carry += __shfl(incl, kWave - 1);
  }
}

__global__ __launch_bounds__(kBlock) void group_scatter_kernel(
    const int64_t* __restrict__ users, const int64_t* __restrict__ pos, const int64_t* __restrict__ neg, int64_t n,
    int64_t batch, int table_bits, const int32_t* __restrict__ own, const int32_t* __restrict__ occ,
    const int32_t* __restrict__ pos_start, int32_t* __restrict__ invalid, int64_t* __restrict__ ou,
    int64_t* __restrict__ op, int64_t* __restrict__ on, int32_t* __restrict__ own_out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < n; t += stride) {
    const int64_t b = t / batch, t0 = b * batch, cnt = min<int64_t>(batch, n - t0);
    const int32_t slot = own[n + t];   // the table entry of the triple's positive item; -1: an out-of-range id
    int64_t at;
    if (slot >= 0) at = t0 + pos_start[(b << table_bits) + slot] + occ[n + t];
    else at = t0 + cnt - 1 - atomicAdd(invalid + b, 1);   // skipped by the step kernel anyway: parked at the batch's end
    ou[at] = users[t];
    op[at] = pos[t];
    on[at] = neg[t];
    own_out[at] = own[t];
    own_out[n + at] = slot;
    own_out[2 * n + at] = own[2 * n + t];
  }
}

}  // namespace hiprec

using namespace hiprec;

extern "C" int32_t hiprec_ownership_table_bits(int64_t batch) {
  int bits = 6;
  while ((1ll << bits) < 4 * batch && bits < 24) ++bits;   // (batches beyond 4 M triples are refused by the kernels)
  return bits;
}

extern "C" int64_t hiprec_ownership_ws_ints(int64_t n, int64_t batch, int32_t table_bits) {
  if (n <= 0 || batch <= 0 || table_bits < 2 || table_bits > 24) return 0;
  const int part_bits = std::min<int>(table_bits, kOwnPartBits);
  const int64_t n_batches = (n + batch - 1) / batch, n_parts = 1ll << (table_bits - part_bits);
  // bucketed keys, their occurrence indices, bucket offsets, bucket totals + cursors
  return 6 * n + n_batches * (2 * n_parts + 1) + n_batches * 4 * n_parts;
}

static int ownership_impl(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t n, int64_t batch,
                          int64_t n_users, int64_t n_items, int32_t table_bits, int32_t* ws, int32_t* total,
                          int32_t* own, int32_t* tab_keys, int32_t* pos_cnt, int32_t* occ, void* stream) {
  HIPREC_REQUIRE(n >= 0 && batch > 0 && n_users > 0 && n_items > 0, "bad sizes");
  HIPREC_REQUIRE(n_users + n_items < (1ll << 31), "row keys need n_users + n_items < 2^31");
  HIPREC_REQUIRE(table_bits >= 2 && table_bits <= 24 && (1ll << table_bits) >= 4 * std::min<int64_t>(batch, n > 0 ? n : 1),
                 "table of 2^%d entries does not fit batches of %lld (at least 4 x the batch, at most 2^24 entries)",
                 table_bits, (long long)batch);
  HIPREC_REQUIRE(3 * batch < (1ll << 31), "batch too large for 32-bit occurrence indices");
  if (n == 0) return 0;
  HIPREC_REQUIRE(users && pos && neg && ws && total && own, "NULL pointer");
  const int64_t n_batches = (n + batch - 1) / batch;
  const int part_bits = std::min<int>(table_bits, kOwnPartBits);
  const int64_t n_parts = 1ll << (table_bits - part_bits);
  const int64_t grid = n_batches * n_parts;
  HIPREC_REQUIRE(grid < (1ll << 31), "too many (batch, partition) pairs");
  const size_t lds = sizeof(int32_t) * 2 * (static_cast<size_t>(1) << part_bits);
  static std::atomic<uint64_t> lds_ok{0};   // 128 KB of dynamic LDS (gfx950 has 160 KB per workgroup)
  if (int rc = allow_dynamic_lds({reinterpret_cast<const void*>(ownership_kernel)}, 2 * sizeof(int32_t) << kOwnPartBits,
                                 lds_ok, "the ownership tables"))
    return rc;
  int32_t* bkey = ws;
  uint32_t* bidx = reinterpret_cast<uint32_t*>(ws + 3 * n);
  int32_t* boff = ws + 6 * n;
  int32_t* counts = boff + n_batches * (2 * n_parts + 1);
  int32_t* cursor = counts + n_batches * 2 * n_parts;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t slices = (std::min<int64_t>(batch, n) + kBucketSlice - 1) / kBucketSlice;
  HIPREC_REQUIRE(n_batches * slices < (1ll << 31), "too many pre-pass workgroups");
  HIPREC_TRY(hipMemsetAsync(counts, 0, sizeof(int32_t) * n_batches * 4 * n_parts, st));
  const int pre = static_cast<int>(n_batches * slices);
  ownership_bucket_kernel<false><<<pre, kBucketThreads, 0, st>>>(users, pos, neg, n, batch, n_users, n_items, table_bits,
                                                                 static_cast<int>(slices), counts, cursor, bkey, bidx, boff, own, 0);
  ownership_bucket_kernel<true><<<pre, kBucketThreads, 0, st>>>(users, pos, neg, n, batch, n_users, n_items, table_bits,
                                                                static_cast<int>(slices), counts, cursor, bkey, bidx, boff, own, 0);
  ownership_kernel<<<static_cast<int>(grid), kOwnThreads, lds, st>>>(bkey, bidx, boff, n, batch, table_bits, total, own,
                                                                     tab_keys, pos_cnt, occ);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_batch_row_ownership(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t n,
                                          int64_t batch, int64_t n_users, int64_t n_items, int32_t table_bits,
                                          int32_t* keys, int32_t* total, int32_t* own, void* stream) {
  return ownership_impl(users, pos, neg, n, batch, n_users, n_items, table_bits, keys, total, own, nullptr, nullptr,
                        nullptr, stream);
}

extern "C" int hiprec_batch_row_ownership_tables(const int64_t* users, const int64_t* pos, const int64_t* neg,
                                                 int64_t n, int64_t batch, int64_t n_users, int64_t n_items,
                                                 int32_t table_bits, int32_t* keys, int32_t* total, int32_t* own,
                                                 int32_t* tab_keys, int32_t* pos_cnt, int32_t* occ, void* stream) {
  HIPREC_REQUIRE(tab_keys && pos_cnt && occ, "NULL pointer");
  return ownership_impl(users, pos, neg, n, batch, n_users, n_items, table_bits, keys, total, own, tab_keys, pos_cnt,
                        occ, stream);
}

extern "C" int64_t hiprec_contrib_row_cap(int64_t batch, int32_t min_contrib) {
  return batch <= 0 ? 0 : min_contrib <= 1 ? 3 * batch : (3 * batch + 1) / 2;
}

extern "C" int hiprec_batch_row_contrib(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t n,
                                        int64_t batch, int64_t n_users, int64_t n_items, int32_t table_bits,
                                        int32_t chunk, int32_t min_contrib, int32_t* ws, int32_t* cidx, int32_t* rows,
                                        int64_t row_cap, int32_t* counts, void* stream) {
  HIPREC_REQUIRE(n >= 0 && batch > 0 && n_users > 0 && n_items > 0, "bad sizes");
  HIPREC_REQUIRE(chunk > 0, "chunk = the triples one wave of the step kernel takes (hiprec_mf_pull_chunk)");
  HIPREC_REQUIRE(min_contrib == 1 || min_contrib == 2, "min_contrib: 2 = records for shared rows only, 1 = for every row");
  HIPREC_REQUIRE(n_users + n_items < (1ll << 31), "row keys need n_users + n_items < 2^31");
  HIPREC_REQUIRE(table_bits >= 2 && table_bits <= 24 && (1ll << table_bits) >= 4 * std::min<int64_t>(batch, n > 0 ? n : 1),
                 "table of 2^%d entries does not fit batches of %lld (at least 4 x the batch, at most 2^24 entries)",
                 table_bits, (long long)batch);
  HIPREC_REQUIRE(3 * batch < (1ll << 31), "batch too large for 32-bit occurrence indices");
  HIPREC_REQUIRE(row_cap >= hiprec_contrib_row_cap(std::min<int64_t>(batch, n > 0 ? n : 1), min_contrib),
                 "row_cap %lld: a batch can need %lld records", (long long)row_cap,
                 (long long)hiprec_contrib_row_cap(batch, min_contrib));
  if (n == 0) return 0;
  HIPREC_REQUIRE(users && pos && neg && ws && cidx && rows && counts, "NULL pointer");
  const int64_t n_batches = (n + batch - 1) / batch;
  const int part_bits = std::min<int>(table_bits, kOwnPartBits);
  const int64_t n_parts = 1ll << (table_bits - part_bits);
  const int64_t grid = n_batches * n_parts;
  HIPREC_REQUIRE(grid < (1ll << 31), "too many (batch, partition) pairs");
  const size_t lds = sizeof(int32_t) * 2 * (static_cast<size_t>(1) << part_bits);
  static std::atomic<uint64_t> lds_ok{0};
  if (int rc = allow_dynamic_lds({reinterpret_cast<const void*>(contrib_kernel)}, 2 * sizeof(int32_t) << kOwnPartBits,
                                 lds_ok, "the contribution tables"))
    return rc;
  int32_t* bkey = ws;
  uint32_t* bidx = reinterpret_cast<uint32_t*>(ws + 3 * n);
  int32_t* boff = ws + 6 * n;
  int32_t* bcounts = boff + n_batches * (2 * n_parts + 1);
  int32_t* cursor = bcounts + n_batches * 2 * n_parts;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t slices = (std::min<int64_t>(batch, n) + kBucketSlice - 1) / kBucketSlice;
  HIPREC_REQUIRE(n_batches * slices < (1ll << 31), "too many pre-pass workgroups");
  HIPREC_TRY(hipMemsetAsync(bcounts, 0, sizeof(int32_t) * n_batches * 4 * n_parts, st));
  HIPREC_TRY(hipMemsetAsync(counts, 0, sizeof(int32_t) * 4 * n_batches, st));
  const int pre = static_cast<int>(n_batches * slices);
  ownership_bucket_kernel<false><<<pre, kBucketThreads, 0, st>>>(users, pos, neg, n, batch, n_users, n_items, table_bits,
                                                                 static_cast<int>(slices), bcounts, cursor, bkey, bidx, boff,
                                                                 cidx, chunk);
  ownership_bucket_kernel<true><<<pre, kBucketThreads, 0, st>>>(users, pos, neg, n, batch, n_users, n_items, table_bits,
                                                                static_cast<int>(slices), bcounts, cursor, bkey, bidx, boff,
                                                                cidx, chunk);
  contrib_kernel<<<static_cast<int>(grid), kOwnThreads, lds, st>>>(bkey, bidx, boff, n, batch, table_bits, cidx,
                                                                   reinterpret_cast<int4*>(rows), row_cap, counts,
                                                                   min_contrib);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_gather_epoch(const int64_t* users, const int64_t* pos, const int64_t* neg, const int64_t* perm,
                                   int32_t shuffle, uint64_t seed, const int64_t* order, int64_t n, int64_t* users_out,
                                   int64_t* pos_out, int64_t* neg_out, void* stream) {
  HIPREC_REQUIRE(n >= 0, "negative n");
  if (n == 0) return 0;
  HIPREC_REQUIRE(users && pos && neg && users_out && pos_out && neg_out, "NULL pointer");
  int half_bits = 0;
  if (!perm && shuffle) {
    half_bits = feistel_half_bits(static_cast<uint64_t>(n));
    HIPREC_REQUIRE(half_bits <= 31, "n too large for the 32-bit Feistel halves");
  }
  gather_epoch_kernel<<<grid_for_threads(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      users, pos, neg, perm, shuffle, half_bits, seed, order, n, users_out, pos_out, neg_out);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_stage_sort_keys(const int64_t* items, const int64_t* perm, int32_t shuffle, uint64_t seed, int64_t n,
                                      int64_t batch, int64_t n_items, int32_t key_bytes, void* keys, void* stream) {
  HIPREC_REQUIRE(n >= 0 && batch > 0 && n_items > 0 && (key_bytes == 4 || key_bytes == 8), "bad sizes");
  if (n == 0) return 0;
  HIPREC_REQUIRE(items && keys, "NULL pointer");
  const int64_t n_batches = (n + batch - 1) / batch;
  HIPREC_REQUIRE(key_bytes == 8 || n_batches * n_items < (1ll << 31), "32-bit keys need n_batches * n_items < 2^31");
  int half_bits = 0;
  if (!perm && shuffle) {
    half_bits = feistel_half_bits(static_cast<uint64_t>(n));
    HIPREC_REQUIRE(half_bits <= 31, "n too large for the 32-bit Feistel halves");
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (key_bytes == 4)
    stage_sort_keys_kernel<int32_t><<<grid_for_threads(n), kBlock, 0, st>>>(items, perm, shuffle, half_bits, seed, n, batch,
                                                                            n_items, static_cast<int32_t*>(keys));
  else
    stage_sort_keys_kernel<int64_t><<<grid_for_threads(n), kBlock, 0, st>>>(items, perm, shuffle, half_bits, seed, n, batch,
                                                                            n_items, static_cast<int64_t*>(keys));
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int64_t hiprec_stage_grouped_ws_ints(int64_t n, int64_t batch, int64_t n_items) {
  if (n <= 0 || batch <= 0 || n_items <= 0) return 0;
  const int64_t n_ranges = (n_items + kStageRange - 1) / kStageRange;
  if (n_ranges > kStageMaxRanges) return 0;     // not supported at this size: the caller keeps its device sort
  const int64_t n_batches = (n + batch - 1) / batch;
  return 2 * n + n_batches * (3 * n_ranges + 1);
}

extern "C" int hiprec_stage_epoch_grouped(const int64_t* users, const int64_t* pos, const int64_t* neg,
                                          const int64_t* perm, int32_t shuffle, uint64_t seed, int64_t n, int64_t batch,
                                          int64_t n_items, int32_t* ws, int64_t* users_out, int64_t* pos_out,
                                          int64_t* neg_out, void* stream) {
  HIPREC_REQUIRE(n >= 0 && batch > 0 && n_items > 0, "bad sizes");
  if (n == 0) return 0;
  HIPREC_REQUIRE(users && pos && neg && ws && users_out && pos_out && neg_out, "NULL pointer");
  HIPREC_REQUIRE(users_out != users && pos_out != pos && neg_out != neg, "the layout is not made in place");
  const int64_t n_ranges = (n_items + kStageRange - 1) / kStageRange;
  HIPREC_REQUIRE(n_ranges <= kStageMaxRanges, "n_items %lld: more than %d ranges of %d items", (long long)n_items,
                 kStageMaxRanges, kStageRange);
  HIPREC_REQUIRE(batch < (1ll << 31), "batch too large for 32-bit positions");
  int half_bits = 0;
  if (!perm && shuffle) {
    half_bits = feistel_half_bits(static_cast<uint64_t>(n));
    HIPREC_REQUIRE(half_bits <= 31, "n too large for the 32-bit Feistel halves");
  }
  const int64_t n_batches = (n + batch - 1) / batch;
  const int64_t slices = (std::min<int64_t>(batch, n) + kBucketSlice - 1) / kBucketSlice;
  HIPREC_REQUIRE(n_batches * slices < (1ll << 31) && n_batches * n_ranges < (1ll << 31), "too many workgroups");
  int32_t* bitem = ws;
  uint32_t* bidx = reinterpret_cast<uint32_t*>(ws + n);
  int32_t* boff = ws + 2 * n;
  int32_t* counts = boff + n_batches * (n_ranges + 1);
  int32_t* cursor = counts + n_batches * n_ranges;
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIPREC_TRY(hipMemsetAsync(counts, 0, sizeof(int32_t) * n_batches * 2 * n_ranges, st));
  const int pre = static_cast<int>(n_batches * slices), nr = static_cast<int>(n_ranges);
  stage_range_kernel<false><<<pre, kBucketThreads, 0, st>>>(pos, perm, shuffle, half_bits, seed, n, batch, n_items, nr,
                                                            static_cast<int>(slices), counts, cursor, boff, bitem, bidx);
  stage_range_kernel<true><<<pre, kBucketThreads, 0, st>>>(pos, perm, shuffle, half_bits, seed, n, batch, n_items, nr,
                                                           static_cast<int>(slices), counts, cursor, boff, bitem, bidx);
  stage_place_kernel<<<static_cast<int>(n_batches * n_ranges), kStageThreads, 0, st>>>(
      users, pos, neg, perm, shuffle, half_bits, seed, n, batch, nr, boff, bitem, bidx, users_out, pos_out, neg_out);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_group_epoch_by_item(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t n,
                                          int64_t batch, int64_t n_users, int32_t table_bits, const int32_t* own,
                                          const int32_t* occ, const int32_t* tab_keys, int32_t* pos_cnt,
                                          int32_t* invalid_cnt, int64_t* users_out, int64_t* pos_out,
                                          int64_t* neg_out, int32_t* own_out, void* stream) {
  HIPREC_REQUIRE(n >= 0 && batch > 0 && n_users > 0 && table_bits >= 2 && table_bits <= 30, "bad sizes");
  if (n == 0) return 0;
  HIPREC_REQUIRE(users && pos && neg && own && occ && tab_keys && pos_cnt && invalid_cnt && users_out && pos_out &&
                     neg_out && own_out,
                 "NULL pointer");
  HIPREC_REQUIRE(users_out != users && pos_out != pos && neg_out != neg && own_out != own, "the scatter is not in place");
  const int64_t n_batches = (n + batch - 1) / batch;
  HIPREC_REQUIRE(n_batches < (1ll << 31) && n_users < (1ll << 31), "too many batches / users");
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIPREC_TRY(hipMemsetAsync(invalid_cnt, 0, sizeof(int32_t) * n_batches, st));
  group_scan_kernel<<<static_cast<int>(n_batches), kGroupThreads, 0, st>>>(tab_keys, pos_cnt, table_bits,
                                                                          static_cast<int32_t>(n_users));
  group_scatter_kernel<<<grid_for_threads(n), kBlock, 0, st>>>(users, pos, neg, n, batch, table_bits, own, occ, pos_cnt,
                                                               invalid_cnt, users_out, pos_out, neg_out, own_out);
  HIPREC_TRY(hipGetLastError());
  return 0;
}
