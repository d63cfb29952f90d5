// Row ownership of a staged epoch, for the owned-rows SGD step (csrc/mf_owned.hip): which rows occur ONCE in
// their batch (their update is a plain store by the wave that read them) and which several times (their
// contributions meet in a slot).  Part of the device batcher's staging: integer work, independent of the weights.
//
// One open-addressing hash table per batch, 2^table_bits entries (>= 4 x batch: at most 3 x batch distinct rows
// go in).  Pass 1 inserts every row occurrence (key = user row, or n_users + item row) with atomicCAS + linear
// probing and counts it in total[]; the table position IS the row's slot id -- no compaction, no sort.  Pass 2
// turns the positions of rows that were counted once into -1.
#include "common.hpp"

namespace hiprec {

__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(kBlock) void ownership_insert_kernel(
    const int64_t* __restrict__ users, const int64_t* __restrict__ pos, const int64_t* __restrict__ neg, int64_t n,
    int64_t batch, int64_t n_users, int64_t n_items, int table_bits, int32_t* __restrict__ keys,
    int32_t* __restrict__ total, int32_t* __restrict__ own_u, int32_t* __restrict__ own_p,
    int32_t* __restrict__ own_n) {
  const uint32_t mask = (1u << table_bits) - 1u;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < 3 * n; i += stride) {
    const int role = static_cast<int>(i / n);
    const int64_t t = i - role * n;
    const int64_t u = users[t], p = pos[t], q = neg[t];
    int32_t* out = role == 0 ? own_u : role == 1 ? own_p : own_n;
    const bool ok = static_cast<uint64_t>(u) < static_cast<uint64_t>(n_users) &&
                    static_cast<uint64_t>(p) < static_cast<uint64_t>(n_items) &&
                    static_cast<uint64_t>(q) < static_cast<uint64_t>(n_items);
    if (!ok) {  // the step kernel skips (and flags) the whole triple
      out[t] = -1;
      continue;
    }
    const int32_t key = static_cast<int32_t>(role == 0 ? u : n_users + (role == 1 ? p : q));
    const int64_t base = (t / batch) << table_bits;
    uint32_t h = hash_u32(static_cast<uint32_t>(key)) & mask;
    for (;;) {
      const int32_t prev = atomicCAS(keys + base + h, -1, key);
      if (prev == -1 || prev == key) break;
      h = (h + 1u) & mask;
    }
    atomicAdd(total + base + h, 1);
    out[t] = static_cast<int32_t>(h);
  }
}

__global__ __launch_bounds__(kBlock) void ownership_resolve_kernel(int64_t n, int64_t batch, int table_bits,
                                                                   const int32_t* __restrict__ total,
                                                                   int32_t* __restrict__ own_u,
                                                                   int32_t* __restrict__ own_p,
                                                                   int32_t* __restrict__ own_n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < 3 * n; i += stride) {
    const int role = static_cast<int>(i / n);
    const int64_t t = i - role * n;
    int32_t* out = role == 0 ? own_u : role == 1 ? own_p : own_n;
    const int32_t h = out[t];
    if (h >= 0 && total[((t / batch) << table_bits) + h] < 2) out[t] = -1;
  }
}

}  // namespace hiprec

using namespace hiprec;

extern "C" int32_t hiprec_ownership_table_bits(int64_t batch) {
  int bits = 6;
  while ((1ll << bits) < 4 * batch && bits < 30) ++bits;
  return bits;
}

extern "C" int hiprec_batch_row_ownership(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t n,
                                          int64_t batch, int64_t n_users, int64_t n_items, int32_t table_bits,
                                          int32_t* keys, int32_t* total, int32_t* own_u, int32_t* own_p,
                                          int32_t* own_n, void* stream) {
  HIPREC_REQUIRE(n >= 0 && batch > 0 && n_users > 0 && n_items > 0, "bad sizes");
  HIPREC_REQUIRE(n_users + n_items < (1ll << 31), "row keys need n_users + n_items < 2^31");
  HIPREC_REQUIRE(table_bits >= 2 && table_bits <= 30 && (1ll << table_bits) >= 4 * std::min<int64_t>(batch, n > 0 ? n : 1),
                 "table of 2^%d entries is too small for batches of %lld", table_bits, (long long)batch);
  if (n == 0) return 0;
  HIPREC_REQUIRE(users && pos && neg && keys && total && own_u && own_p && own_n, "NULL pointer");
  const int64_t n_batches = (n + batch - 1) / batch;
  const size_t bytes = static_cast<size_t>(n_batches) << table_bits << 2;
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIPREC_TRY(hipMemsetAsync(keys, 0xFF, bytes, st));
  HIPREC_TRY(hipMemsetAsync(total, 0, bytes, st));
  const int grid = grid_for_threads(3 * n);
  ownership_insert_kernel<<<grid, kBlock, 0, st>>>(users, pos, neg, n, batch, n_users, n_items, table_bits, keys,
                                                   total, own_u, own_p, own_n);
  ownership_resolve_kernel<<<grid, kBlock, 0, st>>>(n, batch, table_bits, total, own_u, own_p, own_n);
  HIPREC_TRY(hipGetLastError());
  return 0;
}
