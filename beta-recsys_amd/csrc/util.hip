// Library plumbing: error strings, device stats helpers, bit-exact row gather and the
// epoch-level driver that enqueues a whole MFEngine.train_an_epoch (beta_rec/models/mf.py:121-139)
// without returning to the host between batches.
#include <cstring>

#include <cmath>

#include "common.hpp"

namespace hiprec {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) in %s", static_cast<int>(e), hipGetErrorString(e), what);
  return static_cast<int>(e);
}

int allow_dynamic_lds(std::initializer_list<const void*> kernels, size_t bytes, std::atomic<uint64_t>& done,
                      const char* what) {
  int dev = 0, lds_max = 0;
  HIPREC_TRY(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return 0;
  HIPREC_TRY(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
  if (static_cast<size_t>(lds_max) < bytes) {
    set_error("%s needs %zu bytes of LDS per workgroup, this device offers %d (gfx950: 160 KB)", what, bytes, lds_max);
    return HIPREC_E_UNSUPPORTED;
  }
  for (const void* k : kernels)
    HIPREC_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)));
  done.fetch_or(bit, std::memory_order_release);
  return 0;
}

__global__ void stats_reset_kernel(hiprec_stats* s, double b1, double b2) {
  s->loss = 0.f;
  s->reg = 0.f;
  s->loss_sum = 0.0;
  s->reg_sum = 0.0;
  s->step = 0;
  s->beta1 = b1;
  s->beta2 = b2;
  s->beta1_pow = 1.0;
  s->beta2_pow = 1.0;
  s->status = 0u;
  s->_pad = 0u;
}

__global__ void stats_set_step_kernel(hiprec_stats* s, long long step, double b1, double b2, double b1p, double b2p) {
  s->step = step;
  s->beta1 = b1;
  s->beta2 = b2;
  s->beta1_pow = b1p;
  s->beta2_pow = b2p;
}

__global__ void stats_begin_epoch_kernel(hiprec_stats* s) {
  s->loss_sum = 0.0;
  s->reg_sum = 0.0;
}

__global__ void stats_advance_kernel(hiprec_stats* s) { advance_step(s); }

__global__ __launch_bounds__(kBlock) void finalize_kernel(hiprec_stats* s, const Scratch* sc,
                                                          float* g_scalar, float* loss_reg_out) {
  const float gb_part = finalize_partials(s, sc);
  if (threadIdx.x == 0) {
    if (g_scalar) *g_scalar += gb_part;
    if (loss_reg_out) {
      loss_reg_out[0] = s->loss;
      loss_reg_out[1] = s->reg;
    }
  }
}

// out[k, :] = table[idx[k], :] — a pure copy, hence bit-exact.  VEC floats per thread.
template <int VEC>
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const float* __restrict__ table,
                                                             int64_t n_rows, int dim,
                                                             const int64_t* __restrict__ idx,
                                                             int64_t n, float* __restrict__ out,
                                                             hiprec_stats* stats) {
  const int per_row = dim / VEC;
  const int64_t total = n * per_row;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < total;
       e += stride) {
    const int64_t k = e / per_row;
    const int c = static_cast<int>(e - k * per_row);
    const int64_t r = idx[k];
    if (r == -1) {  // padding slot of a fixed-capacity exchange: a zero row, no error
      if constexpr (VEC == 4) reinterpret_cast<float4*>(out)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      else out[e] = 0.f;
      continue;
    }
    if (static_cast<uint64_t>(r) >= static_cast<uint64_t>(n_rows)) {
      if (c == 0) atomicOr(&stats->status, HIPREC_STATUS_ROW_OOB);
      continue;
    }
    if constexpr (VEC == 4) {
      reinterpret_cast<float4*>(out)[e] =
          reinterpret_cast<const float4*>(table + r * dim)[c];
    } else {
      out[e] = table[r * dim + c];
    }
  }
}

// Bucketing for the fixed-capacity all-to-all of the row-sharded engine: key k goes to destination
// rank d = key mod n_dest and receives slot d*cap + (its arrival position in bucket d).  One wave
// aggregates per destination with a ballot, so only one atomic per (wave, destination) reaches the
// n_dest counters.  Negative keys are padding (slot -1); a full bucket raises ROUTE_OVERFLOW.
__global__ __launch_bounds__(kBlock) void route_bucket_kernel(const int64_t* __restrict__ keys,
                                                              int64_t n, int n_dest, int64_t cap,
                                                              int32_t* __restrict__ counts,
                                                              int64_t* __restrict__ slot_out,
                                                              hiprec_stats* stats) {
  const int lane = lane_id();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * kBlock + (threadIdx.x & ~63); base < n;
       base += stride) {
    const int64_t i = base + lane;
    const int64_t key = i < n ? keys[i] : -1;
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
    if (i < n) slot_out[i] = slot;
  }
}

// table[idx[k], :] += src[k, :]  (and nothing when idx[k] is out of range): the owner-side
// accumulation of gradient rows returned by the all-to-all in the row-sharded engine.  One wave per
// source row, lanes over columns, hardware fp32 atomics (rows of different peers may coincide).
__global__ __launch_bounds__(kBlock) void scatter_add_rows_kernel(float* __restrict__ table,
                                                                  int64_t n_rows, int dim,
                                                                  const int64_t* __restrict__ idx,
                                                                  const float* __restrict__ src,
                                                                  int64_t src_stride, int64_t n,
                                                                  hiprec_stats* stats) {
  const int lane = lane_id();
  const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  for (int64_t k = wave0; k < n; k += n_waves) {
    const int64_t r = idx[k];
    if (r == -1) continue;  // padding slot of a fixed-capacity exchange
    if (static_cast<uint64_t>(r) >= static_cast<uint64_t>(n_rows)) {
      if (lane == 0) atomicOr(&stats->status, HIPREC_STATUS_ROW_OOB);
      continue;
    }
    const float* s = src + k * src_stride;
    float* d = table + r * dim;
    for (int c = lane; c < dim; c += kWave) atomic_add_f32(d + c, s[c]);
  }
}

// ---- device-side shuffle: out[i] = P_seed(i), a bijection of [0, n) ---------------------------------
// A 6-round Feistel network over the smallest even-width power of two >= n with cycle walking
// (values that land outside [0, n) are encrypted again): O(1) state, no sort — torch.randperm on the
// device costs ~2 ms for 8 M elements, this ~0.05 ms.
__global__ __launch_bounds__(kBlock) void random_permutation_kernel(int64_t* __restrict__ out,
                                                                    int64_t n, int half_bits,
                                                                    uint64_t seed) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    const uint64_t x = feistel_permute(static_cast<uint64_t>(i), static_cast<uint64_t>(n), half_bits, seed);
    out[i] = static_cast<int64_t>(x);
  }
}

// ---- device-side batcher: stage one epoch -----------------------------------------------------------
// Replaces DataLoader(PairwiseNegativeDataset / RatingDataset, shuffle=True) + default_collate
// (data/base_data.py:247-253, data/data_loaders.py): block b takes batch b of the epoch's visiting
// order (perm[], the Feistel bijection P_seed evaluated on the fly, or sequential), sorts it by item id
// in LDS (bitonic sort of (item, position) pairs; a sum over the batch does not depend on order, and
// the gradient kernels merge adjacent equal items) and writes the three arrays of the batch
// contiguously, so that the training kernels read their batches as plain slices.
//
// Each wave owns a contiguous chunk of C = NPAD / n_waves keys.  A bitonic stage with partner distance
// j < C never leaves the chunk, so it needs no workgroup barrier: a wave's LDS operations execute in
// order, the stage only has to keep the compiler from reordering them.  Of the 78 stages of a
// 4096-key sort only the 10 with j >= 256 synchronise the 16 waves (85 -> ~25 us per batch).
// Bitonic sort of NPAD keys in LDS by a block of NT threads.  Each wave owns a contiguous chunk of
// C = NPAD / n_waves keys; a stage with partner distance j < C never leaves the chunk, so it needs no
// workgroup barrier: a wave's LDS operations execute in order, the stage only has to keep the compiler
// from reordering them.  Of the 78 stages of a 4096-key sort only the 10 with j >= 256 synchronise
// the 16 waves.  `dirty` says whether other waves may have written this wave's chunk last.
template <typename K, int NPAD, int NT>
__device__ __forceinline__ void bitonic_sort_lds(K* s_kv, bool dirty) {
  constexpr int kWaves = NT / kWave;
  constexpr int C = NPAD / kWaves;          // keys per wave-local chunk (power of two, >= 64)
  static_assert(C >= kWave && (C & (C - 1)) == 0, "chunk must be a power of two >= the wave size");
  const int lane = lane_id();
  const int chunk0 = wave_in_block() * C;
  for (int k = 2; k <= NPAD; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= C) {
        __syncthreads();
        for (int p = threadIdx.x; p < NPAD / 2; p += NT) {
          const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i | j;
          const K a = s_kv[i], b = s_kv[l];
          if ((a > b) == ((i & k) == 0)) {
            s_kv[i] = b;
            s_kv[l] = a;
          }
        }
        dirty = true;
      } else {
        if (dirty) {
          __syncthreads();
          dirty = false;
        }
#pragma unroll
        for (int p = lane; p < C / 2; p += kWave) {
          const int i = chunk0 + (((p & ~(j - 1)) << 1) | (p & (j - 1))), l = i | j;
          const K a = s_kv[i], b = s_kv[l];
          if ((a > b) == ((i & k) == 0)) {
            s_kv[i] = b;
            s_kv[l] = a;
          }
        }
        // same-wave LDS operations execute in order; this only pins the compiler's order
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __syncthreads();
}

// Keys are (item id, position in the batch).  When every item id of the batch is below 2^19 (any catalogue
// up to 524 288 items) the pair fits 32 bits -- half the LDS traffic of the 64-bit keys the general path
// sorts; which path a batch takes is decided by the block itself.
constexpr int kStagePosBits = 13;  // NPAD <= 8192

template <int NPAD, int NT>
__global__ __launch_bounds__(NT) void stage_epoch_kernel(const int64_t* __restrict__ users,
                                                         const int64_t* __restrict__ items,
                                                         const void* __restrict__ third,
                                                         int third_bytes,
                                                         const int64_t* __restrict__ perm,
                                                         uint64_t seed, int half_bits,
                                                         int64_t n, int64_t batch,
                                                         int64_t* __restrict__ out_u,
                                                         int64_t* __restrict__ out_i,
                                                         void* __restrict__ out_third) {
  static_assert(NPAD <= (1 << kStagePosBits), "position bits");
  __shared__ unsigned long long s_kv[NPAD];
  uint32_t* s_k32 = reinterpret_cast<uint32_t*>(s_kv);
  const int64_t off = static_cast<int64_t>(blockIdx.x) * batch;
  const int cnt = static_cast<int>(min<int64_t>(batch, n - off));
  auto source = [&](int i) -> int64_t {  // where slot i of this batch's visiting order comes from
    const int64_t j = off + i;
    if (perm) return perm[j];
    if (half_bits)
      return static_cast<int64_t>(feistel_permute(static_cast<uint64_t>(j), static_cast<uint64_t>(n), half_bits, seed));
    return j;
  };
  // pass 1: this thread's item ids (kept in registers for pass 2), and whether they all fit the short key
  constexpr int kPer = (NPAD + NT - 1) / NT;
  uint32_t mine[kPer];
  bool fits = true;
#pragma unroll
  for (int t = 0; t < kPer; ++t) {
    const int i = threadIdx.x + t * NT;
    mine[t] = 0xFFFFFFFFu;
    if (i < cnt) {
      mine[t] = static_cast<uint32_t>(items[source(i)]);
      fits = fits && (mine[t] >> (32 - kStagePosBits)) == 0;
    }
  }
  const bool small = __syncthreads_and(fits) != 0;
  int src_of[kPer];
  if (small) {
#pragma unroll
    for (int t = 0; t < kPer; ++t) {
      const int i = threadIdx.x + t * NT;
      if (i < NPAD) s_k32[i] = i < cnt ? (mine[t] << kStagePosBits) | static_cast<uint32_t>(i) : 0xFFFFFFFFu;
    }
    bitonic_sort_lds<uint32_t, NPAD, NT>(s_k32, true);
#pragma unroll
    for (int t = 0; t < kPer; ++t) {
      const int i = threadIdx.x + t * NT;
      src_of[t] = i < cnt ? static_cast<int>(s_k32[i] & ((1u << kStagePosBits) - 1u)) : 0;
    }
  } else {
#pragma unroll
    for (int t = 0; t < kPer; ++t) {
      const int i = threadIdx.x + t * NT;
      if (i < NPAD)
        s_kv[i] = i < cnt ? (static_cast<unsigned long long>(mine[t]) << 32) | static_cast<uint32_t>(i) : ~0ull;
    }
    bitonic_sort_lds<unsigned long long, NPAD, NT>(s_kv, true);
#pragma unroll
    for (int t = 0; t < kPer; ++t) {
      const int i = threadIdx.x + t * NT;
      src_of[t] = i < cnt ? static_cast<int>(s_kv[i] & 0xFFFFFFFFull) : 0;
    }
  }
#pragma unroll
  for (int t = 0; t < kPer; ++t) {
    const int i = threadIdx.x + t * NT;
    if (i >= cnt) continue;
    const int64_t j = source(src_of[t]);
    out_u[off + i] = users[j];
    out_i[off + i] = items[j];
    if (third_bytes == 8)
      static_cast<int64_t*>(out_third)[off + i] = static_cast<const int64_t*>(third)[j];
    else
      static_cast<float*>(out_third)[off + i] = static_cast<const float*>(third)[j];
  }
}

}  // namespace hiprec

using namespace hiprec;

extern "C" int hiprec_random_permutation(int64_t* out, int64_t n, uint64_t seed, void* stream) {
  HIPREC_REQUIRE(n >= 0 && n < (1ll << 62) && (n == 0 || out), "bad permutation request");
  if (n == 0) return 0;
  const int half_bits = feistel_half_bits(static_cast<uint64_t>(n));  // domain 2^(2*half_bits) >= n, at most 4n
  HIPREC_REQUIRE(half_bits <= 31, "n too large for the 32-bit Feistel halves");
  random_permutation_kernel<<<grid_for_threads(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      out, n, half_bits, seed);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

static int stage_epoch_impl(const int64_t* users, const int64_t* items, const void* third, int32_t third_bytes,
                            const int64_t* perm, uint64_t seed, bool shuffle, int64_t n, int64_t batch,
                            int64_t* out_users, int64_t* out_items, void* out_third, void* stream) {
  HIPREC_REQUIRE(n >= 0 && batch > 0, "bad n / batch");
  HIPREC_REQUIRE(third_bytes == 4 || third_bytes == 8, "third array must be fp32 or int64");
  if (n == 0) return 0;
  HIPREC_REQUIRE(users && items && third && out_users && out_items && out_third, "NULL pointer");
  if (batch > 8192) {
    set_error("hiprec_stage_epoch sorts a batch in LDS: batch %lld > 8192 is not supported",
              (long long)batch);
    return HIPREC_E_UNSUPPORTED;
  }
  const int64_t n_batches = (n + batch - 1) / batch;
  HIPREC_REQUIRE(n_batches < (1ll << 31), "too many batches");
  int half_bits = 0;
  if (shuffle) {
    half_bits = feistel_half_bits(static_cast<uint64_t>(n));
    HIPREC_REQUIRE(half_bits <= 31, "n too large for the 32-bit Feistel halves");
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = static_cast<int>(n_batches);
#define HIPREC_STAGE(NP, T)                                                                              \
  stage_epoch_kernel<NP, T><<<grid, T, 0, st>>>(users, items, third, third_bytes, perm, seed, half_bits, \
                                                n, batch, out_users, out_items, out_third)
  if (batch <= 64) HIPREC_STAGE(64, 64);
  else if (batch <= 256) HIPREC_STAGE(256, 256);
  else if (batch <= 1024) HIPREC_STAGE(1024, 512);
  else if (batch <= 2048) HIPREC_STAGE(2048, 1024);
  else if (batch <= 4096) HIPREC_STAGE(4096, 1024);
  else HIPREC_STAGE(8192, 1024);
#undef HIPREC_STAGE
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_stage_epoch(const int64_t* users, const int64_t* items, const void* third,
                                  int32_t third_bytes, const int64_t* perm, int64_t n, int64_t batch,
                                  int64_t* out_users, int64_t* out_items, void* out_third,
                                  void* stream) {
  return stage_epoch_impl(users, items, third, third_bytes, perm, 0, false, n, batch, out_users, out_items,
                          out_third, stream);
}

extern "C" int hiprec_stage_epoch_shuffled(const int64_t* users, const int64_t* items, const void* third,
                                           int32_t third_bytes, uint64_t seed, int64_t n, int64_t batch,
                                           int64_t* out_users, int64_t* out_items, void* out_third,
                                           void* stream) {
  return stage_epoch_impl(users, items, third, third_bytes, nullptr, seed, true, n, batch, out_users,
                          out_items, out_third, stream);
}

extern "C" int hiprec_scatter_add_rows(float* table, int64_t n_rows, int32_t dim,
                                       const int64_t* idx, const float* src, int64_t src_stride,
                                       int64_t n, hiprec_stats* stats, void* stream) {
  HIPREC_REQUIRE(n >= 0 && n_rows > 0 && dim > 0 && src_stride >= dim, "bad sizes");
  if (n == 0) return 0;
  HIPREC_REQUIRE(table && idx && src && stats, "NULL pointer");
  scatter_add_rows_kernel<<<grid_for_waves(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      table, n_rows, dim, idx, src, src_stride, n, stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_version(void) { return HIPREC_VERSION; }

// sha256 over the sources this library was built from (csrc/*.hip, csrc/*.hpp, include/hiprec.h, in sorted order;
// __graft_entry__.build() passes it).  bench.py compares it with the stamp of the committed profiles: counters taken
// with another build of the kernels are reported as stale instead of being attached to this build's timings.
#ifndef HIPREC_SOURCE_HASH
#define HIPREC_SOURCE_HASH "unknown"
#endif
extern "C" const char* hiprec_source_hash(void) { return HIPREC_SOURCE_HASH; }
extern "C" const char* hiprec_last_error(void) { return g_err; }
extern "C" size_t hiprec_stats_bytes(void) { return sizeof(hiprec_stats); }
extern "C" size_t hiprec_scratch_bytes(int64_t) { return kScratchBytes; }

extern "C" int hiprec_stats_reset(hiprec_stats* stats, double beta1, double beta2, void* stream) {
  HIPREC_REQUIRE(stats, "NULL stats");
  stats_reset_kernel<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(stats, beta1, beta2);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_stats_set_step(hiprec_stats* stats, int64_t step, double beta1, double beta2, void* stream) {
  HIPREC_REQUIRE(stats && step >= 0, "NULL stats / negative step");
  // torch.optim computes beta ** step with the host's pow() on python doubles: do exactly that
  stats_set_step_kernel<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(
      stats, static_cast<long long>(step), beta1, beta2, pow(beta1, static_cast<double>(step)),
      pow(beta2, static_cast<double>(step)));
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_stats_begin_epoch(hiprec_stats* stats, void* stream) {
  HIPREC_REQUIRE(stats, "NULL stats");
  stats_begin_epoch_kernel<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_stats_advance_step(hiprec_stats* stats, void* stream) {
  HIPREC_REQUIRE(stats, "NULL stats");
  stats_advance_kernel<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_finalize_stats(hiprec_stats* stats, const void* scratch, float* g_scalar,
                                     float* loss_reg_out, void* stream) {
  HIPREC_REQUIRE(stats && scratch, "NULL stats/scratch");
  finalize_kernel<<<1, kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      stats, static_cast<const Scratch*>(scratch), g_scalar, loss_reg_out);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_gather_rows(const float* table, int64_t n_rows, int32_t dim,
                                  const int64_t* idx, int64_t n, float* out, hiprec_stats* stats,
                                  void* stream) {
  HIPREC_REQUIRE(n >= 0 && n_rows > 0 && dim > 0, "bad sizes");
  if (n == 0) return 0;
  HIPREC_REQUIRE(table && idx && out && stats, "NULL pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec4 = (dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(table) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  if (vec4) {
    gather_rows_kernel<4><<<grid_for_threads(n * (dim / 4)), kBlock, 0, st>>>(table, n_rows, dim,
                                                                             idx, n, out, stats);
  } else {
    gather_rows_kernel<1><<<grid_for_threads(n * dim), kBlock, 0, st>>>(table, n_rows, dim, idx,
                                                                       n, out, stats);
  }
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_route_bucket(const int64_t* keys, int64_t n, int32_t n_dest, int64_t cap,
                                   int32_t* counts, int64_t* slot_out, hiprec_stats* stats,
                                   void* stream) {
  HIPREC_REQUIRE(n >= 0 && n_dest > 0 && n_dest <= 64 && cap > 0, "bad routing sizes");
  HIPREC_REQUIRE(counts && stats && (n == 0 || (keys && slot_out)), "NULL pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIPREC_TRY(hipMemsetAsync(counts, 0, sizeof(int32_t) * n_dest, st));
  if (n == 0) return 0;
  route_bucket_kernel<<<grid_for_threads(n), kBlock, 0, st>>>(keys, n, n_dest, cap, counts, slot_out,
                                                              stats);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

extern "C" int hiprec_mf_bpr_epoch(const hiprec_mf_tables* w, const hiprec_mf_tables* g,
                                   const int64_t* users, const int64_t* pos, const int64_t* neg,
                                   const int64_t* perm, int64_t n_triples, int64_t batch,
                                   float reg_coef, int kind,
                                   double lr, double beta1, double beta2, double eps,
                                   float* flat_w, float* flat_g, float* flat_m, float* flat_v,
                                   int64_t n_flat, int32_t* user_stamp, int32_t* item_stamp,
                                   int32_t first_stamp, hiprec_stats* stats, void* scratch,
                                   size_t scratch_bytes, void* stream) {
  HIPREC_REQUIRE(n_triples >= 0 && batch > 0, "bad n_triples/batch");
  const bool rows_sgd = (kind == HIPREC_OPT_SGD) && user_stamp && item_stamp;
  if (!rows_sgd) HIPREC_REQUIRE(flat_w && flat_g && n_flat > 0, "dense optimizer needs flat buffers");
  if (int rc = hiprec_stats_begin_epoch(stats, stream)) return rc;
  int32_t stamp = first_stamp;
  for (int64_t off = 0; off < n_triples; off += batch, ++stamp) {
    const int64_t b = (n_triples - off < batch) ? (n_triples - off) : batch;  // drop_last=False
    const float inv_b = 1.0f / static_cast<float>(b);
    const int64_t* pm = perm ? perm + off : nullptr;
    const int64_t* uu = perm ? users : users + off;
    const int64_t* pp = perm ? pos : pos + off;
    const int64_t* nn = perm ? neg : neg + off;
    if (int rc = hiprec_mf_bpr_grad(w, g, uu, pp, nn, pm, b, inv_b, reg_coef, stats, scratch,
                                    scratch_bytes, stream))
      return rc;
    int rc;
    if (rows_sgd)
      rc = hiprec_mf_sgd_rows(w, g, uu, pp, nn, pm, b, lr, user_stamp, item_stamp, stamp, stats,
                              scratch, stream);
    else
      rc = hiprec_opt_dense_step(kind, flat_w, flat_g, flat_m, flat_v, n_flat, lr, beta1, beta2,
                                 eps, stats, scratch, w->global_bias - flat_w, stream);
    if (rc) return rc;
  }
  return 0;
}

// MFEngine.train_an_epoch with loss == "bce" (mf.py:108-111,121-139) over resident
// (user, item, rating) arrays — the RatingDataset loader of data/base_data.py:182-216 replaced by
// perm[] slices / a staged layout, exactly like hiprec_mf_bpr_epoch.
extern "C" int hiprec_mf_bce_epoch(const hiprec_mf_tables* w, const hiprec_mf_tables* g,
                                   const int64_t* users, const int64_t* items, const float* ratings,
                                   const int64_t* perm, int64_t n_samples, int64_t batch,
                                   float reg_coef, int kind, double lr, double beta1, double beta2,
                                   double eps, float* flat_w, float* flat_g, float* flat_m,
                                   float* flat_v, int64_t n_flat, int32_t* user_stamp,
                                   int32_t* item_stamp, int32_t first_stamp, hiprec_stats* stats,
                                   void* scratch, size_t scratch_bytes, void* stream) {
  HIPREC_REQUIRE(n_samples >= 0 && batch > 0, "bad n_samples/batch");
  const bool rows_sgd = (kind == HIPREC_OPT_SGD) && user_stamp && item_stamp;
  if (!rows_sgd) HIPREC_REQUIRE(flat_w && flat_g && n_flat > 0, "dense optimizer needs flat buffers");
  if (int rc = hiprec_stats_begin_epoch(stats, stream)) return rc;
  int32_t stamp = first_stamp;
  for (int64_t off = 0; off < n_samples; off += batch, ++stamp) {
    const int64_t b = (n_samples - off < batch) ? (n_samples - off) : batch;
    const float inv_b = 1.0f / static_cast<float>(b);
    const int64_t* pm = perm ? perm + off : nullptr;
    const int64_t* uu = perm ? users : users + off;
    const int64_t* ii = perm ? items : items + off;
    const float* rr = perm ? ratings : ratings + off;
    if (int rc = hiprec_mf_bce_grad(w, g, uu, ii, rr, pm, b, inv_b, reg_coef, stats, scratch,
                                    scratch_bytes, stream))
      return rc;
    int rc;
    if (rows_sgd)
      rc = hiprec_mf_sgd_rows(w, g, uu, ii, nullptr, pm, b, lr, user_stamp, item_stamp, stamp, stats,
                              scratch, stream);
    else
      rc = hiprec_opt_dense_step(kind, flat_w, flat_g, flat_m, flat_v, n_flat, lr, beta1, beta2,
                                 eps, stats, scratch, w->global_bias - flat_w, stream);
    if (rc) return rc;
  }
  return 0;
}

extern "C" size_t hiprec_dp_step_bytes(void) { return sizeof(hiprec_dp_step); }

extern "C" int hiprec_mf_dp_step_begin(const hiprec_dp_step* c, const int64_t* users,
                                       const int64_t* items_a, const void* third, int64_t batch,
                                       float inv_batch_global, void* stream) {
  HIPREC_REQUIRE(c != nullptr, "NULL step context");
  int rc;
  if (c->loss_kind == 0)
    rc = hiprec_mf_bpr_grad(&c->w, &c->g, users, items_a, static_cast<const int64_t*>(third), nullptr, batch,
                            inv_batch_global, c->reg_coef, c->stats, c->scratch, c->scratch_bytes, stream);
  else
    rc = hiprec_mf_bce_grad(&c->w, &c->g, users, items_a, static_cast<const float*>(third), nullptr, batch,
                            inv_batch_global, c->reg_coef, c->stats, c->scratch, c->scratch_bytes, stream);
  if (rc) return rc;
  return hiprec_finalize_stats(c->stats, c->scratch, c->g.global_bias, c->loss_reg_out, stream);
}

extern "C" int hiprec_mf_dp_step_end(const hiprec_dp_step* c, void* stream) {
  HIPREC_REQUIRE(c != nullptr, "NULL step context");
  return hiprec_opt_dense_step(c->opt_kind, c->w_flat, c->g_flat, c->m_flat, c->v_flat, c->n_flat, c->lr,
                               c->beta1, c->beta2, c->eps, c->stats, nullptr, -1, stream);
}
