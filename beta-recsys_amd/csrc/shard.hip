// Routing / packing kernels of the row-sharded engine: owner(row) = row mod R.
//
// One training step of the sharded engine moves (A2A-1) triples to the rank that owns the user row,
// (A2A-2) item ids to the ranks that own the item rows and the rows (+bias) back, (A2A-3) the item-row
// gradients back to their owners.  Every exchange is a fixed-capacity all-to-all: R buckets of `cap`
// slots, padding = -1.  The first implementation built the send buffers with ~45 small torch ops per
// step (stack / full / where / index_put / cat / div ...), which made the step host-bound (0.53 ms at
// the 4096-triple size).  The kernels below do the bucketing and the packing / unpacking in one launch
// each; the engine's step is then ~14 launches and 5 collectives.
//
// Bucketing is the ballot scheme of route_bucket_kernel (util.hip): one atomic per (wave,
// destination).  A full bucket sets HIPREC_STATUS_ROUTE_OVERFLOW and drops the entry (slot -1).
#include "common.hpp"

namespace hiprec {
namespace {

// slot of `key` in bucket (key mod n_dest), or -1 (padding key / overflow); wave-cooperative:
// every lane of the wave must call it
__device__ __forceinline__ int64_t bucket_slot(int64_t key, int n_dest, int64_t cap,
                                               int32_t* __restrict__ counts, hiprec_stats* stats) {
  const int lane = lane_id();
  const int d = key >= 0 ? static_cast<int>(key % n_dest) : -1;
  int64_t slot = -1;
  for (int q = 0; q < n_dest; ++q) {
    const unsigned long long m = __ballot(d == q);
    if (m == 0) continue;
    int start = 0;
    if (lane == 0) start = atomicAdd(counts + q, __popcll(m));
    start = __builtin_amdgcn_readfirstlane(start);
    if (d == q) {
      const int pos = start + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < cap) slot = static_cast<int64_t>(q) * cap + pos;
      else atomicOr(&stats->status, HIPREC_STATUS_ROUTE_OVERFLOW);
    }
  }
  return slot;
}

// A2A-1 send buffer: send[slot] = (u, p, n) with slot bucketed by owner(u); `send` is pre-filled with -1
__global__ __launch_bounds__(kBlock) void shard_route_triples_kernel(
    const int64_t* __restrict__ users, const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
    int64_t n, int n_dest, int64_t cap, int32_t* __restrict__ counts, int64_t* __restrict__ send,
    hiprec_stats* stats) {
  const int lane = lane_id();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * kBlock + (threadIdx.x & ~63); base < n;
       base += stride) {
    const int64_t i = base + lane;
    const int64_t u = i < n ? users[i] : -1;
    const int64_t slot = bucket_slot(u, n_dest, cap, counts, stats);
    if (slot >= 0) {
      send[3 * slot + 0] = u;
      send[3 * slot + 1] = pos[i];
      send[3 * slot + 2] = neg[i];
    }
  }
}

// A2A-2 request buffer from the received triples: req[slot] = item id bucketed by owner(item) (pre-filled
// with -1); slot_pos / slot_neg = where each triple's rows will come back; u_loc = local user row
__global__ __launch_bounds__(kBlock) void shard_route_items_kernel(
    const int64_t* __restrict__ recv, int64_t n_slots, int n_dest, int64_t cap,
    int32_t* __restrict__ counts, int64_t* __restrict__ req, int64_t* __restrict__ slot_pos,
    int64_t* __restrict__ slot_neg, int64_t* __restrict__ u_loc, hiprec_stats* stats) {
  const int lane = lane_id();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * kBlock + (threadIdx.x & ~63); base < n_slots;
       base += stride) {
    const int64_t i = base + lane;
    int64_t u = -1, p = -1, nn = -1;
    if (i < n_slots) {
      u = recv[3 * i];
      if (u >= 0) {
        p = recv[3 * i + 1];
        nn = recv[3 * i + 2];
      }
    }
    const int64_t sp = bucket_slot(p, n_dest, cap, counts, stats);
    const int64_t sn = bucket_slot(nn, n_dest, cap, counts, stats);
    if (sp >= 0) req[sp] = p;
    if (sn >= 0) req[sn] = nn;
    if (i < n_slots) {
      // a triple whose item request overflowed cannot be trained: it becomes padding (the overflow bit
      // is already raised and turns into an error on the host)
      const bool live = u >= 0 && sp >= 0 && sn >= 0;
      slot_pos[i] = live ? sp : 0;
      slot_neg[i] = live ? sn : 0;
      u_loc[i] = live ? u / n_dest : -1;
    }
  }
}

// owner side of A2A-2: payload[k] = [item_emb[local(incoming[k])] | item_bias[...]] (zeros for padding),
// local_idx[k] = incoming[k] / n_dest or -1.  One wave per slot.
__global__ __launch_bounds__(kBlock) void shard_gather_payload_kernel(
    const float* __restrict__ item_emb, const float* __restrict__ item_bias, int64_t n_rows, int dim,
    const int64_t* __restrict__ incoming, int64_t n, int n_dest, float* __restrict__ payload,
    int64_t* __restrict__ local_idx, hiprec_stats* stats) {
  const int lane = lane_id();
  const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int ld = dim + 1;
  for (int64_t k = wave0; k < n; k += n_waves) {
    const int64_t id = incoming[k];
    int64_t r = id >= 0 ? id / n_dest : -1;
    if (r >= n_rows) {
      if (lane == 0) atomicOr(&stats->status, HIPREC_STATUS_ROW_OOB);
      r = -1;
    }
    float* out = payload + k * ld;
    for (int c = lane; c < ld; c += kWave) {
      float v = 0.f;
      if (r >= 0) v = c < dim ? item_emb[r * dim + c] : item_bias[r];
      out[c] = v;
    }
    if (lane == 0) local_idx[k] = r;
  }
}

// [n, dim+1] <-> ([n, dim], [n]): the gradient kernel wants dense dim-strided rows
__global__ __launch_bounds__(kBlock) void shard_split_kernel(const float* __restrict__ src, int64_t n,
                                                             int dim, float* __restrict__ emb,
                                                             float* __restrict__ bias) {
  const int64_t total = n * (dim + 1);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / (dim + 1);
    const int c = static_cast<int>(e - r * (dim + 1));
    if (c < dim) emb[r * dim + c] = src[e];
    else bias[r] = src[e];
  }
}

__global__ __launch_bounds__(kBlock) void shard_join_kernel(const float* __restrict__ emb,
                                                            const float* __restrict__ bias, int64_t n,
                                                            int dim, float* __restrict__ dst) {
  const int64_t total = n * (dim + 1);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / (dim + 1);
    const int c = static_cast<int>(e - r * (dim + 1));
    dst[e] = c < dim ? emb[r * dim + c] : bias[r];
  }
}

// ---- the epoch-planned sharded step (sharded.py::ShardedMFEngine.train_an_epoch, plain SGD) --------------------
// Each destination's chunk of the gradient exchange ends with ONE extra row that carries this rank's
// [loss, reg, d loss / d scalar bias] of the step: the 3-float all-reduce rides in the all-to-all.
__global__ __launch_bounds__(kBlock) void shard_publish_partials_kernel(const Scratch* __restrict__ scratch,
                                                                        float* __restrict__ g_send, int ld,
                                                                        const int64_t* __restrict__ extra_rows,
                                                                        int n_dest) {
  __shared__ double s_l[kBlock], s_r[kBlock], s_b[kBlock];
  const uint32_t n = scratch->n_partials;
  double l = 0.0, r = 0.0, b = 0.0;
  for (uint32_t i = threadIdx.x; i < n; i += kBlock) {
    const float4 p = scratch->partials[i];
    l += p.x;
    r += p.y;
    b += p.z;
  }
  s_l[threadIdx.x] = l;
  s_r[threadIdx.x] = r;
  s_b[threadIdx.x] = b;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s) {
      s_l[threadIdx.x] += s_l[threadIdx.x + s];
      s_r[threadIdx.x] += s_r[threadIdx.x + s];
      s_b[threadIdx.x] += s_b[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (static_cast<int>(threadIdx.x) < n_dest) {
    float* row = g_send + extra_rows[threadIdx.x] * ld;
    row[0] = static_cast<float>(s_l[0]);
    row[1] = static_cast<float>(s_r[0]);
    row[2] = static_cast<float>(s_b[0]);
  }
}

// Owner side of the gradient exchange: item row idx[k] (and its bias) -= lr * g_recv[k]; rows with idx -1 (the
// extra rows, padding) are skipped.  Several peers may return gradients of one item: fp32 atomics.  Plain SGD on
// the received rows IS the whole item-side optimizer step: no dense gradient buffer, no second pass.
__global__ __launch_bounds__(kBlock) void shard_apply_rows_kernel(float* __restrict__ item_emb,
                                                                  float* __restrict__ item_bias, int64_t n_rows,
                                                                  int dim, const int64_t* __restrict__ idx,
                                                                  const float* __restrict__ g_recv, int64_t n,
                                                                  float lr, hiprec_stats* stats) {
  const int lane = lane_id();
  const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int ld = dim + 1;
  for (int64_t k = wave0; k < n; k += n_waves) {
    const int64_t r = idx[k];
    if (r < 0) continue;
    if (r >= n_rows) {
      if (lane == 0) atomicOr(&stats->status, HIPREC_STATUS_ROW_OOB);
      continue;
    }
    const float* g = g_recv + k * ld;
    for (int c = lane; c < dim; c += kWave) atomic_add_f32(item_emb + r * dim + c, -lr * g[c]);
    if (lane == 0) atomic_add_f32(item_bias + r, -lr * g[dim]);
  }
}

// After the gradient exchange: sum the extra rows of every peer (the global loss, regularizer and scalar-bias
// gradient of the step), book them in hiprec_stats, apply the scalar bias update, count the step.
__global__ void shard_finish_step_kernel(const float* __restrict__ g_recv, int ld,
                                         const int64_t* __restrict__ extra_rows, int n_src, float* global_bias,
                                         float lr, int first_of_epoch, hiprec_stats* stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float l = 0.f, r = 0.f, b = 0.f;
  for (int q = 0; q < n_src; ++q) {
    const float* row = g_recv + extra_rows[q] * ld;
    l += row[0];
    r += row[1];
    b += row[2];
  }
  if (first_of_epoch) {
    stats->loss_sum = 0.0;
    stats->reg_sum = 0.0;
  }
  stats->loss = l;
  stats->reg = r;
  stats->loss_sum += static_cast<double>(l);
  stats->reg_sum += static_cast<double>(r);
  *global_bias = *global_bias - lr * b;
  advance_step(stats);
}

}  // namespace
}  // namespace hiprec

using namespace hiprec;

extern "C" int hiprec_shard_publish_partials(const void* scratch, float* g_send, int32_t dim,
                                             const int64_t* extra_rows, int32_t n_dest, void* stream) {
  HIPREC_REQUIRE(scratch && g_send && extra_rows && dim > 0 && n_dest > 0 && n_dest <= kBlock, "bad arguments");
  shard_publish_partials_kernel<<<1, kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const Scratch*>(scratch), g_send, dim + 1, extra_rows, n_dest);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_shard_apply_rows(float* item_emb, float* item_bias, int64_t n_rows, int32_t dim,
                                       const int64_t* idx, const float* g_recv, int64_t n, double lr,
                                       hiprec_stats* stats, void* stream) {
  HIPREC_REQUIRE(n >= 0 && n_rows >= 0 && dim > 0, "bad sizes");
  if (n == 0) return 0;
  HIPREC_REQUIRE(item_emb && item_bias && idx && g_recv && stats, "NULL pointer");
  shard_apply_rows_kernel<<<grid_for_waves(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      item_emb, item_bias, n_rows, dim, idx, g_recv, n, static_cast<float>(lr), stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_shard_finish_step(const float* g_recv, int32_t dim, const int64_t* extra_rows, int32_t n_src,
                                        float* global_bias, double lr, int32_t first_of_epoch, hiprec_stats* stats,
                                        void* stream) {
  HIPREC_REQUIRE(g_recv && extra_rows && global_bias && stats && dim > 0 && n_src > 0, "bad arguments");
  shard_finish_step_kernel<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(g_recv, dim + 1, extra_rows, n_src,
                                                                       global_bias, static_cast<float>(lr),
                                                                       first_of_epoch, stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_shard_route_triples(const int64_t* users, const int64_t* pos, const int64_t* neg,
                                          int64_t n, int32_t n_dest, int64_t cap, int32_t* counts,
                                          int64_t* send, hiprec_stats* stats, void* stream) {
  HIPREC_REQUIRE(n >= 0 && n_dest > 0 && n_dest <= 64 && cap > 0, "bad routing sizes");
  HIPREC_REQUIRE(counts && send && stats && (n == 0 || (users && pos && neg)), "NULL pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIPREC_TRY(hipMemsetAsync(counts, 0, sizeof(int32_t) * n_dest, st));
  HIPREC_TRY(hipMemsetAsync(send, 0xFF, sizeof(int64_t) * 3 * n_dest * cap, st));  // every slot = -1
  if (n == 0) return 0;
  shard_route_triples_kernel<<<grid_for_threads(n), kBlock, 0, st>>>(users, pos, neg, n, n_dest, cap, counts,
                                                                     send, stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_shard_route_items(const int64_t* recv, int64_t n_slots, int32_t n_dest, int64_t cap,
                                        int32_t* counts, int64_t* req, int64_t* slot_pos, int64_t* slot_neg,
                                        int64_t* u_loc, hiprec_stats* stats, void* stream) {
  HIPREC_REQUIRE(n_slots >= 0 && n_dest > 0 && n_dest <= 64 && cap > 0, "bad routing sizes");
  HIPREC_REQUIRE(counts && req && stats && (n_slots == 0 || (recv && slot_pos && slot_neg && u_loc)),
                 "NULL pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIPREC_TRY(hipMemsetAsync(counts, 0, sizeof(int32_t) * n_dest, st));
  HIPREC_TRY(hipMemsetAsync(req, 0xFF, sizeof(int64_t) * n_dest * cap, st));
  if (n_slots == 0) return 0;
  shard_route_items_kernel<<<grid_for_threads(n_slots), kBlock, 0, st>>>(recv, n_slots, n_dest, cap, counts, req,
                                                                         slot_pos, slot_neg, u_loc, stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_shard_gather_payload(const float* item_emb, const float* item_bias, int64_t n_rows,
                                           int32_t dim, const int64_t* incoming, int64_t n, int32_t n_dest,
                                           float* payload, int64_t* local_idx, hiprec_stats* stats,
                                           void* stream) {
  HIPREC_REQUIRE(n >= 0 && n_rows > 0 && dim > 0 && n_dest > 0, "bad sizes");
  if (n == 0) return 0;
  HIPREC_REQUIRE(item_emb && item_bias && incoming && payload && local_idx && stats, "NULL pointer");
  shard_gather_payload_kernel<<<grid_for_waves(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      item_emb, item_bias, n_rows, dim, incoming, n, n_dest, payload, local_idx, stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_shard_split_rows(const float* src, int64_t n, int32_t dim, float* emb, float* bias,
                                       void* stream) {
  HIPREC_REQUIRE(n >= 0 && dim > 0, "bad sizes");
  if (n == 0) return 0;
  HIPREC_REQUIRE(src && emb && bias, "NULL pointer");
  shard_split_kernel<<<grid_for_threads(n * (dim + 1)), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      src, n, dim, emb, bias);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_shard_join_rows(const float* emb, const float* bias, int64_t n, int32_t dim, float* dst,
                                      void* stream) {
  HIPREC_REQUIRE(n >= 0 && dim > 0, "bad sizes");
  if (n == 0) return 0;
  HIPREC_REQUIRE(emb && bias && dst, "NULL pointer");
  shard_join_kernel<<<grid_for_threads(n * (dim + 1)), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      emb, bias, n, dim, dst);
  HIPREC_TRY(hipGetLastError());
  return 0;
}
