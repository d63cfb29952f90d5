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
#include <algorithm>

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
                                                                        const int32_t* __restrict__ extra_rows32,
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
    float* row = g_send + (extra_rows ? extra_rows[threadIdx.x] : extra_rows32[threadIdx.x]) * ld;
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


// ---- the planned step, fused (round 3): two launches around the gradient kernel instead of five --------------------
// Row k of a step's incoming block: rows [self_lo, self_hi) are the ones this rank asked of ITSELF -- they never
// travel: the payload goes straight into the fetched buffer, the gradient is read straight from the send buffer.
__device__ __forceinline__ float* seg_row(float* other, float* self_base, int64_t k, int64_t self_lo, int64_t self_hi,
                                          int ld) {
  return (k >= self_lo && k < self_hi) ? self_base + (k - self_lo) * ld : other + k * ld;
}

// payload[k] = [item_emb[idx[k]] | item_bias[idx[k]]] (zeros for idx -1: the extra rows) for the rows peers asked
// for, AND -- in the blocks behind the row blocks -- the clear of the gradient exchange buffer g_send.
__global__ __launch_bounds__(kBlock) void shard_payload_zero_kernel(
    const float* __restrict__ item_emb, const float* __restrict__ item_bias, int64_t n_rows, int dim,
    const int32_t* __restrict__ idx, int64_t n, int64_t self_lo, int64_t self_hi, float* __restrict__ payload,
    float* __restrict__ self_dst, float* __restrict__ zero, int64_t zero_floats, const uint8_t* __restrict__ shared,
    int n_row_blocks, hiprec_stats* stats) {
  if (static_cast<int>(blockIdx.x) >= n_row_blocks) {
    const int64_t nb = gridDim.x - n_row_blocks;
    if (shared != nullptr) {  // only the slots several triples add into (and the extra rows) have to start from zero
      const int ldz = dim + 1;
      const int64_t n_slots = zero_floats / ldz;
      const int64_t w0 = static_cast<int64_t>(blockIdx.x - n_row_blocks) * kWavesPerBlock + wave_in_block();
      for (int64_t slot = w0; slot < n_slots; slot += nb * kWavesPerBlock) {
        if (!shared[slot]) continue;
        for (int c = lane_id(); c < ldz; c += kWave) zero[slot * ldz + c] = 0.f;
      }
      return;
    }
    const int64_t tid = static_cast<int64_t>(blockIdx.x - n_row_blocks) * kBlock + threadIdx.x;
    float4* z4 = reinterpret_cast<float4*>(zero);
    const int64_t n4 = zero_floats >> 2;
    for (int64_t i = tid; i < n4; i += nb * kBlock) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (n4 << 2) + tid; i < zero_floats; i += nb * kBlock) zero[i] = 0.f;
    return;
  }
  const int lane = lane_id();
  const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  const int64_t n_waves = static_cast<int64_t>(n_row_blocks) * kWavesPerBlock;
  const int ld = dim + 1;
  for (int64_t k = wave0; k < n; k += n_waves) {
    int64_t r = idx[k];
    if (r >= n_rows) {
      if (lane == 0) atomicOr(&stats->status, HIPREC_STATUS_ROW_OOB);
      r = -1;
    }
    float* out = seg_row(payload, self_dst, k, self_lo, self_hi, ld);
    for (int c = lane; c < ld; c += kWave) {
      float v = 0.f;
      if (r >= 0) v = c < dim ? item_emb[r * dim + c] : item_bias[r];
      out[c] = v;
    }
  }
}

// After the gradient exchange, ONE launch: target row idx[k] (+ its bias) += coef * g_recv[k] with fp32 atomics
// (several peers may return gradients of one item) -- plain SGD: target = the item table, coef = -lr, which IS the
// item-side optimizer step; Adam / RMSprop: target = the dense gradient, coef = 1 -- and, in the last block, the
// step's bookkeeping: the peers' extra rows (loss, reg, d loss / d scalar bias) summed in rank order ->
// hiprec_stats, *scalar_target += scalar_coef * (its gradient), t <- t + 1.
__global__ __launch_bounds__(kBlock) void shard_apply_finish_kernel(
    float* __restrict__ t_emb, float* __restrict__ t_bias, int64_t n_rows, int dim, const int32_t* __restrict__ idx,
    const float* __restrict__ g_recv, int64_t n, int64_t self_lo, int64_t self_hi, const float* __restrict__ g_self,
    float coef, const int32_t* __restrict__ extra_pos, int n_src, float* scalar_target, float scalar_coef,
    int first_of_epoch, const uint32_t* __restrict__ dup_bits, int n_row_blocks, hiprec_stats* stats) {
  const int ld = dim + 1;
  if (static_cast<int>(blockIdx.x) >= n_row_blocks) {
    if (threadIdx.x != 0) return;
    float l = 0.f, r = 0.f, b = 0.f;
    for (int q = 0; q < n_src; ++q) {
      const float* row = seg_row(const_cast<float*>(g_recv), const_cast<float*>(g_self), extra_pos[q], self_lo,
                                 self_hi, ld);
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
    *scalar_target = *scalar_target + scalar_coef * b;
    advance_step(stats);
    return;
  }
  const int lane = lane_id();
  const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  const int64_t n_waves = static_cast<int64_t>(n_row_blocks) * kWavesPerBlock;
  for (int64_t k = wave0; k < n; k += n_waves) {
    const int64_t r = idx[k];
    if (r < 0) continue;
    if (r >= n_rows) {
      if (lane == 0) atomicOr(&stats->status, HIPREC_STATUS_ROW_OOB);
      continue;
    }
    const float* g = seg_row(const_cast<float*>(g_recv), const_cast<float*>(g_self), k, self_lo, self_hi, ld);
    // a row only ONE peer returns a gradient for (the plan's duplicate bits say so) has a single writer: plain RMW
    const bool contended = dup_bits == nullptr || ((dup_bits[r >> 5] >> (r & 31)) & 1u);
    if (contended) {
      for (int c = lane; c < dim; c += kWave) atomic_add_f32(t_emb + r * dim + c, coef * g[c]);
      if (lane == 0) atomic_add_f32(t_bias + r, coef * g[dim]);
    } else {
      for (int c = lane; c < dim; c += kWave) t_emb[r * dim + c] += coef * g[c];
      if (lane == 0) t_bias[r] += coef * g[dim];
    }
  }
}

}  // namespace
}  // namespace hiprec

using namespace hiprec;

extern "C" int hiprec_shard_publish_partials(const void* scratch, float* g_send, int32_t dim,
                                             const int64_t* extra_rows, int32_t n_dest, void* stream) {
  HIPREC_REQUIRE(scratch && g_send && extra_rows && n_dest > 0 && n_dest <= kBlock, "bad arguments");
  HIPREC_REQUIRE(dim >= 2, "an extra row carries 3 floats: the planned sharded step needs emb_dim >= 2");
  shard_publish_partials_kernel<<<1, kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const Scratch*>(scratch), g_send, dim + 1, extra_rows, nullptr, n_dest);
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
  HIPREC_REQUIRE(g_recv && extra_rows && global_bias && stats && n_src > 0, "bad arguments");
  HIPREC_REQUIRE(dim >= 2, "an extra row carries 3 floats: the planned sharded step needs emb_dim >= 2");
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

// ---- the planned step through the fused launches, and a whole range of steps enqueued from C -----------------------
extern "C" int hiprec_shard_payload_zero(const float* item_emb, const float* item_bias, int64_t n_rows, int32_t dim,
                                         const int32_t* idx, int64_t n, int64_t self_lo, int64_t self_hi,
                                         float* payload, float* self_dst, float* zero, int64_t zero_floats,
                                         const uint8_t* shared, hiprec_stats* stats, void* stream) {
  HIPREC_REQUIRE(n >= 0 && n_rows >= 0 && dim > 0 && zero_floats >= 0 && self_lo >= 0 && self_lo <= self_hi, "bad sizes");
  HIPREC_REQUIRE(stats && (n == 0 || (idx && payload)) && (zero_floats == 0 || zero) && (self_lo == self_hi || self_dst),
                 "NULL pointer");
  HIPREC_REQUIRE(n == 0 || n_rows == 0 || (item_emb && item_bias), "NULL pointer");
  if (n == 0 && zero_floats == 0) return 0;
  const int rb = n > 0 ? grid_for_waves(n) : 0;
  const int zb = zero_floats > 0 ? grid_for_threads((zero_floats + 15) / 16) : 0;   // 4 x float4 per thread
  shard_payload_zero_kernel<<<rb + zb, kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      item_emb, item_bias, n_rows, dim, idx, n, self_lo, self_hi, payload, self_dst, zero, zero_floats, shared, rb,
      stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_shard_apply_finish(float* t_emb, float* t_bias, int64_t n_rows, int32_t dim, const int32_t* idx,
                                         const float* g_recv, int64_t n, int64_t self_lo, int64_t self_hi,
                                         const float* g_self, double coef, const int32_t* extra_pos, int32_t n_src,
                                         float* scalar_target, double scalar_coef, int32_t first_of_epoch,
                                         const uint32_t* dup_bits, hiprec_stats* stats, void* stream) {
  HIPREC_REQUIRE(n >= 0 && n_rows >= 0 && n_src > 0 && self_lo >= 0 && self_lo <= self_hi, "bad sizes");
  HIPREC_REQUIRE(dim >= 2, "an extra row carries 3 floats: the planned sharded step needs emb_dim >= 2");
  HIPREC_REQUIRE(stats && extra_pos && scalar_target && idx && g_recv && (self_lo == self_hi || g_self), "NULL pointer");
  HIPREC_REQUIRE(n_rows == 0 || (t_emb && t_bias), "NULL pointer");
  const int rb = n > 0 ? grid_for_waves(n) : 0;
  shard_apply_finish_kernel<<<rb + 1, kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      t_emb, t_bias, n_rows, dim, idx, g_recv, n, self_lo, self_hi, g_self, static_cast<float>(coef), extra_pos, n_src,
      scalar_target, static_cast<float>(scalar_coef), first_of_epoch, dup_bits, rb, stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

// RCCL entry points as the caller hands them over (beta-recsys_amd/_rccl.py binds the librccl PyTorch loaded; this
// library does not link it): ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd.
using nccl_send_fn = int (*)(const void*, size_t, int, int, void*, hipStream_t);
using nccl_recv_fn = int (*)(void*, size_t, int, int, void*, hipStream_t);
using nccl_group_fn = int (*)();
constexpr int kNcclFloat32 = 7;  // ncclDataType_t (nccl.h / rccl.h)

extern "C" int hiprec_mf_bpr_owned_remote_step(float*, int64_t, int64_t, int32_t, const float*, float*, int64_t,
                                               const int64_t*, const int64_t*, const int64_t*, const int32_t*,
                                               const int32_t*, const int32_t*, const int32_t*, int32_t*, float*,
                                               int64_t, float, float, double, hiprec_stats*, void*, void*);
extern "C" int hiprec_mf_bpr_pull_remote_step(float*, int64_t, int64_t, int32_t, const float*, float*, int64_t,
                                              const int64_t*, const int64_t*, const int64_t*, const int32_t*,
                                              const int32_t*, const int32_t*, const int32_t*, int64_t, const int32_t*,
                                              float*, float*, const int32_t*, int32_t, int64_t, float, float, double,
                                              hiprec_stats*, void*, void*);
extern "C" int hiprec_mf_bpr_grad_remote_step(const float*, float*, int64_t, int64_t, int32_t, const float*, float*,
                                              int64_t, const int64_t*, const int64_t*, const int64_t*, const int32_t*,
                                              const int32_t*, const int32_t*, const int32_t*, int64_t, float, float,
                                              hiprec_stats*, void*, void*);

extern "C" size_t hiprec_shard_plan_bytes(void) { return sizeof(hiprec_shard_plan); }
extern "C" size_t hiprec_shard_bufs_bytes(void) { return sizeof(hiprec_shard_bufs); }

extern "C" int hiprec_shard_planned_steps_ex(const hiprec_shard_plan* plan, const hiprec_shard_bufs* bufs,
                                             int64_t step_begin, int64_t step_end, int32_t kind, float reg_coef,
                                             double lr, double beta1, double beta2, double eps,
                                             const hiprec_nccl_fns* nccl, void* comm, uint32_t flags,
                                             hiprec_stats* stats, void* stream) {
  HIPREC_REQUIRE(plan && bufs && stats, "NULL pointer");
  const int R = plan->world, me = plan->rank;
  HIPREC_REQUIRE(R >= 1 && R <= 64 && me >= 0 && me < R, "bad world / rank");
  HIPREC_REQUIRE((flags & ~static_cast<uint32_t>(HIPREC_SHARD_EXCHANGE_SELF)) == 0, "unknown flag bits %#x", flags);
  // HIPREC_SHARD_EXCHANGE_SELF: this rank's OWN segment of both exchanges travels through the communicator as well (a
  // grouped send to + recv from itself) instead of being written in place by the payload / read in place by the apply
  // launch.  Same results bit for bit (the exchange is a copy); it is how a single GPU executes the ncclSend / ncclRecv
  // path of this driver against the real library, and a self-test of the binding on any rank of a larger world.
  const bool self_x = (flags & HIPREC_SHARD_EXCHANGE_SELF) != 0;
  const bool exchange = R > 1 || self_x;
  HIPREC_REQUIRE(!exchange || (nccl && comm && nccl->send && nccl->recv && nccl->group_start && nccl->group_end),
                 "a world of %d ranks%s needs the RCCL entry points and a communicator", R,
                 self_x ? " exchanging with itself" : "");
  HIPREC_REQUIRE(0 <= step_begin && step_begin <= step_end && step_end <= plan->n_steps, "bad step range");
  HIPREC_REQUIRE(plan->users && plan->pos_slot && plan->neg_slot && plan->own && plan->total && plan->in_idx &&
                     plan->ex_req && plan->ex_in && plan->in_off_host && plan->n_slots_host && plan->req_cnt_host &&
                     plan->in_cnt_host,
                 "incomplete plan");
  HIPREC_REQUIRE(bufs->w_flat && bufs->payload && bufs->g_recv && bufs->fetched && bufs->g_send && bufs->scratch,
                 "incomplete step buffers");
  const bool dense = kind != HIPREC_OPT_SGD;
  HIPREC_REQUIRE(kind == HIPREC_OPT_SGD || kind == HIPREC_OPT_ADAM || kind == HIPREC_OPT_RMSPROP, "unknown optimizer");
  // plain SGD as owner pulls (round 5): the plan carries the step blocks' contribution lists
  const bool pull = !dense && plan->cidx != nullptr && bufs->dim % 4 == 0;
  HIPREC_REQUIRE(!pull || (plan->rows && plan->counts && plan->row_cap >= (3 * plan->cap + 1) / 2 && bufs->cbuf && bufs->cbias),
                 "incomplete contribution lists / buffers of the owner-pulls step");
  HIPREC_REQUIRE(dense ? (bufs->g_flat != nullptr) : (pull || (bufs->arrived && bufs->acc)), "incomplete step buffers");
  // exact lazy Adam / RMSprop (csrc/lazy_opt.hip): the step's rows are caught up before they are read and stepped
  // after their gradients are complete -- the dense sweep of the whole shard goes
  const bool lazy = dense && bufs->stamp_u != nullptr;
  HIPREC_REQUIRE(!lazy || (bufs->stamp_i && bufs->v_flat && (kind != HIPREC_OPT_ADAM || (bufs->m_flat && bufs->lazy_scalars))),
                 "incomplete lazy optimizer state");
  const int D = bufs->dim, ld = D + 1;
  HIPREC_REQUIRE(D >= 2 && D <= 256, "the planned sharded step needs 2 <= emb_dim <= 256");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t nu = bufs->n_users_local, ni = bufs->n_items_local, cap = plan->cap, tot = plan->n_steps * cap;
  float* w = bufs->w_flat;
  float* item_emb = w + nu * D;
  float* item_bias = w + (nu + ni) * D + nu;
  float* gbias = w + (nu + ni) * static_cast<int64_t>(ld);
  float* g = bufs->g_flat;
  const int64_t n_flat = (nu + ni) * static_cast<int64_t>(ld) + 1;
  const auto send = nccl ? reinterpret_cast<nccl_send_fn>(nccl->send) : nullptr;
  const auto recv = nccl ? reinterpret_cast<nccl_recv_fn>(nccl->recv) : nullptr;
  const auto g_start = nccl ? reinterpret_cast<nccl_group_fn>(nccl->group_start) : nullptr;
  const auto g_end = nccl ? reinterpret_cast<nccl_group_fn>(nccl->group_end) : nullptr;
  hiprec_lazy_state lz{};
  if (lazy) {
    lz.w = w, lz.g = g, lz.m = bufs->m_flat, lz.v = bufs->v_flat;
    lz.n_users = nu, lz.n_items = ni, lz.dim = D, lz.kind = kind;
    lz.stamp_u = bufs->stamp_u, lz.stamp_i = bufs->stamp_i;
    lz.scalars = bufs->lazy_scalars, lz.scalars_cap = static_cast<int32_t>(std::min<int64_t>(bufs->lazy_scalars_cap, 1 << 30));
    lz.lr = lr, lz.beta1 = beta1, lz.beta2 = beta2, lz.eps = eps;
  }
  // Every rank must post every exchange of the range: a rank that returned half-way would leave its peers waiting in
  // theirs.  Whatever can be checked is therefore checked BEFORE the first launch.
  for (int64_t s = step_begin; s < step_end; ++s) {
    const int64_t* req = plan->req_cnt_host + s * R;
    const int64_t* inc = plan->in_cnt_host + s * R;
    int64_t in_rows = 0, req_rows = 0;
    for (int q = 0; q < R; ++q) {
      HIPREC_REQUIRE(req[q] >= 0 && inc[q] >= 0, "inconsistent plan: negative row count in step %lld", (long long)s);
      in_rows += inc[q] + 1;
      req_rows += req[q] + 1;
    }
    HIPREC_REQUIRE(inc[me] == req[me], "inconsistent plan: a rank asks itself for %lld rows and expects %lld",
                   (long long)req[me], (long long)inc[me]);
    HIPREC_REQUIRE(in_rows == plan->in_off_host[s + 1] - plan->in_off_host[s] && req_rows == plan->n_slots_host[s],
                   "inconsistent plan: the per-peer row counts of step %lld do not add up to its block sizes",
                   (long long)s);
  }
  for (int64_t s = step_begin; s < step_end; ++s) {
    const int64_t* req = plan->req_cnt_host + s * R;
    const int64_t* inc = plan->in_cnt_host + s * R;
    const int64_t il = plan->in_off_host[s + 1] - plan->in_off_host[s], sl = plan->n_slots_host[s];
    const int32_t* idx = plan->in_idx + plan->in_off_host[s];
    int64_t in_lo = 0, req_lo = 0;   // this rank's own segment in the incoming / the fetched block
    for (int q = 0; q < me; ++q) {
      in_lo += inc[q] + 1;
      req_lo += req[q] + 1;
    }
    if (self_x) in_lo = 0;   // no segment is "self": [0, 0)
    const int64_t in_hi = self_x ? 0 : in_lo + inc[me] + 1;
    float* self_fetched = bufs->fetched + req_lo * ld;
    float* self_g = bufs->g_send + req_lo * ld;
    const uint8_t* shared = plan->slot_shared ? plan->slot_shared + s * plan->slot_stride : nullptr;
    const uint32_t* dup = plan->dup_bits ? plan->dup_bits + s * plan->dup_words : nullptr;
    // the rows this step touches here: the local users of its triples and the item rows its peers ask for
    const hiprec_lazy_rows touched{plan->users + s * cap, cap, nullptr, 0, nullptr, 0, idx, il};
    if (lazy)
      if (int rc = hiprec_lazy_catchup(&lz, &touched, stats, stream)) return rc;
    // (the owner-pulls step writes every slot of g_send exactly once: nothing to clear)
    if (int rc = hiprec_shard_payload_zero(item_emb, item_bias, ni, D, idx, il, in_lo, in_hi, bufs->payload,
                                           self_fetched, bufs->g_send, pull ? 0 : sl * ld, shared, stats, stream))
      return rc;
    if (exchange) {
      if (g_start()) {
        set_error("ncclGroupStart failed before the row exchange of step %lld", (long long)s);
        return HIPREC_E_UNSUPPORTED;
      }
      int64_t io = 0, ro = 0;
      for (int q = 0; q < R; ++q) {
        if (q != me || self_x) {
          if (send(bufs->payload + io * ld, static_cast<size_t>((inc[q] + 1) * ld), kNcclFloat32, q, comm, st) ||
              recv(bufs->fetched + ro * ld, static_cast<size_t>((req[q] + 1) * ld), kNcclFloat32, q, comm, st)) {
            g_end();
            set_error("ncclSend / ncclRecv failed in the row exchange of step %lld", (long long)s);
            return HIPREC_E_UNSUPPORTED;
          }
        }
        io += inc[q] + 1;
        ro += req[q] + 1;
      }
      if (g_end()) {
        set_error("ncclGroupEnd failed in the row exchange of step %lld (peers disagree about its sizes?)", (long long)s);
        return HIPREC_E_UNSUPPORTED;
      }
    }
    const int64_t off = s * cap;
    const int64_t b_local = std::min<int64_t>(plan->local_batch, plan->n_local - s * plan->local_batch);
    const float inv_b = 1.0f / static_cast<float>(static_cast<int64_t>(R) * b_local);
    int rc;
    if (pull)
      rc = hiprec_mf_bpr_pull_remote_step(w, nu, ni, D, bufs->fetched, bufs->g_send, sl, plan->users + off,
                                          plan->pos_slot + off, plan->neg_slot + off, plan->cidx + off,
                                          plan->cidx + tot + off, plan->cidx + 2 * tot + off,
                                          plan->rows + s * plan->row_cap * 4, plan->row_cap, plan->counts + 4 * s,
                                          bufs->cbuf, bufs->cbias, plan->ex_req + s * R, R, cap, inv_b, reg_coef, lr,
                                          stats, bufs->scratch, stream);
    else if (dense)
      rc = hiprec_mf_bpr_grad_remote_step(w, g, nu, ni, D, bufs->fetched, bufs->g_send, sl, plan->users + off,
                                          plan->pos_slot + off, plan->neg_slot + off, plan->own + off,
                                          plan->own + tot + off, plan->own + 2 * tot + off,
                                          plan->total + s * plan->total_stride, cap, inv_b, reg_coef, stats,
                                          bufs->scratch, stream);
    else
      rc = hiprec_mf_bpr_owned_remote_step(w, nu, ni, D, bufs->fetched, bufs->g_send, sl, plan->users + off,
                                           plan->pos_slot + off, plan->neg_slot + off, plan->own + off,
                                           plan->own + tot + off, plan->own + 2 * tot + off,
                                           plan->total + s * plan->total_stride, bufs->arrived, bufs->acc, cap, inv_b,
                                           reg_coef, lr, stats, bufs->scratch, stream);
    if (rc) return rc;
    if (!pull)   // (the pull launch has published the partials itself)
      shard_publish_partials_kernel<<<1, kBlock, 0, st>>>(static_cast<const Scratch*>(bufs->scratch), bufs->g_send, ld,
                                                          nullptr, plan->ex_req + s * R, R);
    if (exchange) {
      if (g_start()) {
        set_error("ncclGroupStart failed before the gradient exchange of step %lld", (long long)s);
        return HIPREC_E_UNSUPPORTED;
      }
      int64_t io = 0, ro = 0;
      for (int q = 0; q < R; ++q) {
        if (q != me || self_x) {
          if (send(bufs->g_send + ro * ld, static_cast<size_t>((req[q] + 1) * ld), kNcclFloat32, q, comm, st) ||
              recv(bufs->g_recv + io * ld, static_cast<size_t>((inc[q] + 1) * ld), kNcclFloat32, q, comm, st)) {
            g_end();
            set_error("ncclSend / ncclRecv failed in the gradient exchange of step %lld", (long long)s);
            return HIPREC_E_UNSUPPORTED;
          }
        }
        io += inc[q] + 1;
        ro += req[q] + 1;
      }
      if (g_end()) {
        set_error("ncclGroupEnd failed in the gradient exchange of step %lld (peers disagree about its sizes?)",
                  (long long)s);
        return HIPREC_E_UNSUPPORTED;
      }
    }
    float* t_emb = dense ? g + nu * D : item_emb;
    float* t_bias = dense ? g + (nu + ni) * D + nu : item_bias;
    float* scalar = dense ? g + (nu + ni) * static_cast<int64_t>(ld) : gbias;
    if ((rc = hiprec_shard_apply_finish(t_emb, t_bias, ni, D, idx, bufs->g_recv, il, in_lo, in_hi, self_g,
                                        dense ? 1.0 : -lr, plan->ex_in + s * R, R, scalar, dense ? 1.0 : -lr,
                                        s == 0 ? 1 : 0, dup, stats, stream)))
      return rc;
    if (lazy) {
      if ((rc = hiprec_lazy_update(&lz, &touched, nullptr, stats, stream))) return rc;
    } else if (dense && (rc = hiprec_opt_dense_step(kind, w, g, bufs->m_flat, bufs->v_flat, n_flat, lr, beta1, beta2,
                                                    eps, stats, nullptr, -1, stream))) {
      return rc;
    }
  }
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_shard_planned_steps(const hiprec_shard_plan* plan, const hiprec_shard_bufs* bufs,
                                          int64_t step_begin, int64_t step_end, int32_t kind, float reg_coef,
                                          double lr, double beta1, double beta2, double eps,
                                          const hiprec_nccl_fns* nccl, void* comm, hiprec_stats* stats, void* stream) {
  return hiprec_shard_planned_steps_ex(plan, bufs, step_begin, step_end, kind, reg_coef, lr, beta1, beta2, eps, nccl,
                                       comm, 0u, stats, stream);
}
