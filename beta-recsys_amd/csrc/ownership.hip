// Row ownership of a staged epoch, for the owned-rows SGD step (csrc/mf_owned.hip): which rows occur ONCE in
// their batch (their update is a plain store by the wave that read them) and which several times (their
// contributions meet in a slot).  Part of the device batcher's staging: integer work, independent of the weights.
//
// One open-addressing hash table per batch, 2^table_bits entries (>= 4 x batch: at most 3 x batch distinct rows
// go in); the table position of a row IS its slot id -- no compaction, no sort.  The table is cut into
// partitions of 2^14 entries by the top bits of the hash, and one workgroup builds one (batch, partition) in
// LDS: it scans the 32-bit keys of the batch's 3 x batch row occurrences (key = user row, or n_users + item row,
// made once by a streaming pre-pass), inserts the ones that hash into its partition with LDS compare-and-swap +
// linear probing (inside the partition), counts them and hands every occurrence its slot; the partition's counts
// go to total[] in one coalesced write.  A row that occurs once has total[slot] == 1: the step kernel treats it
// exactly like own = -1.  A first version kept the tables in global memory: 20 M device atomics per 50-batch
// epoch, 1.5 ms; LDS atomics make it one pass over L2-resident keys per partition.
#include <algorithm>

#include <atomic>

#include "common.hpp"

namespace hiprec {

constexpr int kOwnPartBits = 14;              // 16 384 entries x (key + count) = 128 KB of LDS
constexpr int kOwnThreads = 1024;
constexpr int kOwnUnroll = 8;                // keys in flight per thread

__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

// key of every row occurrence, role-major ([0,n) user rows, [n,2n) positive, [2n,3n) negative item rows):
// user row, or n_users + item row; -1 when the triple has an out-of-range id (the step kernel skips it whole)
__global__ __launch_bounds__(kBlock) void ownership_keys_kernel(const int64_t* __restrict__ users,
                                                                const int64_t* __restrict__ pos,
                                                                const int64_t* __restrict__ neg, int64_t n,
                                                                int64_t n_users, int64_t n_items,
                                                                int32_t* __restrict__ keys) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < n; t += stride) {
    const int64_t u = users[t], p = pos[t], q = neg[t];
    const bool ok = static_cast<uint64_t>(u) < static_cast<uint64_t>(n_users) &&
                    static_cast<uint64_t>(p) < static_cast<uint64_t>(n_items) &&
                    static_cast<uint64_t>(q) < static_cast<uint64_t>(n_items);
    keys[t] = ok ? static_cast<int32_t>(u) : -1;
    keys[n + t] = ok ? static_cast<int32_t>(n_users + p) : -1;
    keys[2 * n + t] = ok ? static_cast<int32_t>(n_users + q) : -1;
  }
}

// tab_keys / pos_cnt / occ (all optional) serve the row-sharded engine's epoch planner (csrc/plan.hip): tab_keys =
// the table itself (the row key that sits in every entry, -1 = empty); the user and positive-item occurrences are
// inserted BEFORE the negative ones, so that an entry's count at that point -- pos_cnt -- is the number of POSITIVE
// occurrences of an item row, and occ[at] (the entry's count when occurrence `at` arrived) is, for a positive
// occurrence, its rank among them: the planner lays every batch out grouped by positive item from these two,
// without a sort and without a second pass of atomics.
__global__ __launch_bounds__(kOwnThreads) void ownership_kernel(const int32_t* __restrict__ keys, int64_t n,
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
  __shared__ int32_t s_qkey[kOwnThreads / kWave][2 * kWave];   // per wave: ring of filtered keys ...
  __shared__ uint32_t s_qidx[kOwnThreads / kWave][2 * kWave];  // ... and their occurrence index inside the batch
  const int lane = lane_id();
  int32_t* q_key = s_qkey[wave_in_block()];
  uint32_t* q_idx = s_qidx[wave_in_block()];
  for (uint32_t i = threadIdx.x; i < part_size; i += kOwnThreads) {
    s_key[i] = -1;
    s_cnt[i] = 0;
  }
  __syncthreads();
  const int64_t t0 = b * batch;
  const int64_t cnt = min<int64_t>(batch, n - t0);
  const int64_t tab0 = (b << table_bits) + (static_cast<int64_t>(part) << part_bits);
  // phase 0: user and positive rows ([0, 2 cnt)), phase 1: negative rows
  for (int phase = 0; phase < 2; ++phase) {
    const int64_t lo = phase == 0 ? 0 : 2 * cnt, hi = phase == 0 ? 2 * cnt : 3 * cnt;
    // Two stages per wave.  (1) Filter: kOwnUnroll keys per lane are requested together and the ones that hash into
    // this partition (one in n_parts) are appended to the wave's queue in LDS (ballot + prefix: order of arrival).
    // (2) Insert: as soon as 64 are queued, every lane takes one -- the probe loop is a chain of LDS compare-and-swap
    // round trips whose length is the longest chain among the ACTIVE lanes; run straight on the filtered stream it
    // had ~4 active lanes per wave instruction (560-760 us per 50 x 65 536-triple epoch).
    uint32_t q_head = 0, q_tail = 0;  // wave-uniform; the ring holds q_tail - q_head <= 2 * kWave entries
    auto drain = [&](uint32_t n_take) {
      const bool on = static_cast<uint32_t>(lane) < n_take;
      const uint32_t slot = (q_head + lane) & (2 * kWave - 1);
      const int32_t key = on ? q_key[slot] : -1;
      const uint32_t i = on ? q_idx[slot] : 0u;
      q_head += n_take;
      if (!on) return;
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
        return;
      }
      const int32_t before = atomicAdd(s_cnt + h, 1);
      own[at] = static_cast<int32_t>((part << part_bits) | h);
      if (occ) occ[at] = before;
    };
    for (int64_t i0 = lo + (threadIdx.x & ~(kWave - 1)); i0 < hi; i0 += static_cast<int64_t>(kOwnThreads) * kOwnUnroll) {
      int32_t key[kOwnUnroll];
#pragma unroll
      for (int j = 0; j < kOwnUnroll; ++j) {
        const int64_t i = i0 + lane + static_cast<int64_t>(j) * kOwnThreads;
        const int role = i < cnt ? 0 : i < 2 * cnt ? 1 : 2;   // (no 64-bit division here)
        key[j] = i < hi ? keys[role * n + t0 + (i - role * cnt)] : -2;
      }
#pragma unroll
      for (int j = 0; j < kOwnUnroll; ++j) {
        const int64_t i = i0 + lane + static_cast<int64_t>(j) * kOwnThreads;
        if (key[j] == -1 && part == 0) {  // a triple with an out-of-range id: no slot
          const int role = i < cnt ? 0 : i < 2 * cnt ? 1 : 2;
          own[role * n + t0 + (i - role * cnt)] = -1;
        }
        const bool pass = key[j] >= 0 && (n_parts == 1 || (hash_u32(static_cast<uint32_t>(key[j])) >>
                                                            (32 - (table_bits - part_bits))) == part);
        const unsigned long long m = __ballot(pass);
        if (m == 0) continue;
        if (pass) {
          const uint32_t slot = (q_tail + __popcll(m & ((1ull << lane) - 1ull))) & (2 * kWave - 1);
          q_key[slot] = key[j];
          q_idx[slot] = static_cast<uint32_t>(i);
        }
        q_tail += __popcll(m);
        if (q_tail - q_head >= kWave) drain(kWave);
      }
    }
    if (q_tail != q_head) drain(q_tail - q_head);
    __syncthreads();
    if (phase == 0 && pos_cnt) {
      for (uint32_t i = threadIdx.x; i < part_size; i += kOwnThreads) pos_cnt[tab0 + i] = s_cnt[i];
      // the snapshot must be complete before any wave's phase 1 adds negative occurrences to s_cnt (a
      // wave that finished its stride early raced the slower ones' reads)
      __syncthreads();
    }
  }
  for (uint32_t i = threadIdx.x; i < part_size; i += kOwnThreads) {
    total[tab0 + i] = s_cnt[i];
    if (tab_keys) tab_keys[tab0 + i] = s_key[i];
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
  while ((1ll << bits) < 4 * batch && bits < 30) ++bits;
  return bits;
}

static int ownership_impl(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t n, int64_t batch,
                          int64_t n_users, int64_t n_items, int32_t table_bits, int32_t* keys, int32_t* total,
                          int32_t* own, int32_t* tab_keys, int32_t* pos_cnt, int32_t* occ, void* stream) {
  HIPREC_REQUIRE(n >= 0 && batch > 0 && n_users > 0 && n_items > 0, "bad sizes");
  HIPREC_REQUIRE(n_users + n_items < (1ll << 31), "row keys need n_users + n_items < 2^31");
  HIPREC_REQUIRE(table_bits >= 2 && table_bits <= 30 && (1ll << table_bits) >= 4 * std::min<int64_t>(batch, n > 0 ? n : 1),
                 "table of 2^%d entries is too small for batches of %lld", table_bits, (long long)batch);
  if (n == 0) return 0;
  HIPREC_REQUIRE(users && pos && neg && keys && total && own, "NULL pointer");
  const int64_t n_batches = (n + batch - 1) / batch;
  const int part_bits = std::min<int>(table_bits, kOwnPartBits);
  const int64_t grid = n_batches << (table_bits - part_bits);
  HIPREC_REQUIRE(grid < (1ll << 31), "too many (batch, partition) pairs");
  const size_t lds = sizeof(int32_t) * 2 * (static_cast<size_t>(1) << part_bits);
  static std::atomic<uint64_t> lds_ok{0};   // 128 KB of dynamic LDS (gfx950 has 160 KB per workgroup)
  if (int rc = allow_dynamic_lds({reinterpret_cast<const void*>(ownership_kernel)}, 2 * sizeof(int32_t) << kOwnPartBits,
                                 lds_ok, "the ownership tables"))
    return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ownership_keys_kernel<<<grid_for_threads(n), kBlock, 0, st>>>(users, pos, neg, n, n_users, n_items, keys);
  ownership_kernel<<<static_cast<int>(grid), kOwnThreads, lds, st>>>(keys, n, batch, table_bits, total, own, tab_keys,
                                                                     pos_cnt, occ);
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
