// Shared device/host helpers for libhiprec (gfx950 only: wave64, DPP, hardware fp32 atomics).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <initializer_list>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/hiprec.h"

namespace hiprec {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // 4 waves = one per SIMD
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxBlocks = 2048;   // 256 CUs x 8 blocks; grid-stride beyond that
constexpr int kAggWaves = 16;      // waves per block of the gradient kernels (one triple per wave)
constexpr int kAggBlock = kAggWaves * kWave;
constexpr int kAggMaxBlocks = 512; // 2 x 16-wave blocks per CU
constexpr size_t kScratchBytes = 16 + sizeof(float) * 4 * kMaxBlocks;

// scratch block layout: header + per-block {loss, reg, d(loss)/d(scalar bias), -} partial sums of
// the last *_grad call.  The scalar-bias gradient travels here instead of through one atomicAdd per
// block: a thousand atomics on ONE address serialise at ~25 ns each (measured 15 us per launch).
struct Scratch {
  uint32_t n_partials;
  uint32_t _pad[3];
  float4 partials[kMaxBlocks];
};
static_assert(sizeof(Scratch) == kScratchBytes, "scratch layout");

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

// Kernels that want more dynamic LDS than HIP's 64 KB default: the limit is an attribute per (function, DEVICE), so
// `done` keeps one bit per HIP device (thread-safe; ADVICE r3: a process-wide `static bool` was neither).  Checks what
// the current device offers first: a part with less LDS gets HIPREC_E_UNSUPPORTED and a message, not a raw HIP error.
int allow_dynamic_lds(std::initializer_list<const void*> kernels, size_t bytes, std::atomic<uint64_t>& done,
                      const char* what);

// hiprec_opt_dense_step (csrc/optim.hip) with an optional gradient-norm clip riding in the sweep: clip_ws = the
// workspace clip_sumsq_kernel filled (n_clip per-block sums behind two result slots), nullptr = no clip.
int opt_dense_step_impl(int kind, float* w, float* g, float* m, float* v, int64_t n, double lr, double beta1,
                        double beta2, double eps, hiprec_stats* stats, const void* scratch, int64_t scalar_index,
                        double* clip_ws, int n_clip, float max_norm, void* stream);

#define HIPREC_TRY(expr)                                   \
  do {                                                     \
    hipError_t _e = (expr);                                \
    if (_e != hipSuccess) return hiprec::hip_fail(_e, #expr); \
  } while (0)

#define HIPREC_REQUIRE(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      hiprec::set_error(__VA_ARGS__);    \
      return HIPREC_E_BADARG;            \
    }                                    \
  } while (0)

inline int grid_for_waves(int64_t n_waves) {
  int64_t blocks = (n_waves + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks < 1) blocks = 1;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  return static_cast<int>(blocks);
}

inline int grid_for_threads(int64_t n_threads) {
  int64_t blocks = (n_threads + kBlock - 1) / kBlock;
  if (blocks < 1) blocks = 1;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  return static_cast<int>(blocks);
}

// Width of one Feistel half for a bijection of [0, n): the smallest even-width power of two >= n.
__host__ __device__ inline int feistel_half_bits(uint64_t n) {
  int bits = 2;
  while (bits < 62 && (1ull << bits) < n) ++bits;
  return (bits + 1) / 2;
}

#if defined(__HIPCC__)

// ---- counter-based randomness (stateless, replayable, restated bit for bit in oracle/sampler_numpy.py)
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// keep/drop draw of element e at (seed, step): uniform 24-bit fraction < keep_prob.  Counter-based and
// stateless: LightGCN's edge dropout and NGCF's message dropout use it.
__device__ __forceinline__ bool keep_draw(uint64_t seed, uint64_t step, int64_t e, float keep_prob) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (static_cast<uint64_t>(e) + 1) + 0xD1B54A32D192ED03ull * (step + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return static_cast<float>(z >> 40) * (1.0f / 16777216.0f) < keep_prob;  // 24 bits -> [0, 1)
}

__device__ __forceinline__ uint32_t feistel_round(uint32_t x, uint32_t key) {
  x = (x ^ key) * 0x9E3779B1u;
  x ^= x >> 15;
  x *= 0x85EBCA77u;
  x ^= x >> 13;
  return x;
}

// P_seed(i): a bijection of [0, n).  6-round Feistel network over 2^(2*half_bits) >= n with cycle
// walking (values that land outside [0, n) are encrypted again).
__device__ __forceinline__ uint64_t feistel_permute(uint64_t i, uint64_t n, int half_bits, uint64_t seed) {
  const uint64_t mask = (1ull << half_bits) - 1ull;
  uint64_t x = i;
  do {
    uint32_t l = static_cast<uint32_t>(x >> half_bits), r = static_cast<uint32_t>(x & mask);
#pragma unroll
    for (int round = 0; round < 6; ++round) {
      const uint32_t k = static_cast<uint32_t>(seed >> (8 * (round & 3))) + 0x632BE5ABu * (round + 1) +
                         static_cast<uint32_t>(seed >> 32);
      const uint32_t f = feistel_round(r, k) & static_cast<uint32_t>(mask);
      const uint32_t nl = r;
      r = l ^ f;
      l = nl;
    }
    x = (static_cast<uint64_t>(l) << half_bits) | r;
  } while (x >= n);
  return x;
}

// ---- wave64 all-reduce (sum) on the VALU: 4 DPP steps inside each 16-lane row, then the four row
// totals are read back through SGPRs.  No LDS traffic, result uniform across the wave.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false);
  return v + __builtin_bit_cast(float, moved);
}

__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror -> every lane of a row holds the row total
  int iv = __builtin_bit_cast(int, v);
  float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
  float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
  float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ int wave_in_block() {
  return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
}

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & 63; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a release/acquire fence for
// GLOBAL memory too: hipcc emits s_waitcnt vmcnt(0) in front of it, i.e. every wave would sit out
// the full latency of its fire-and-forget gradient atomics (1-2 us under load) at each barrier of
// the LDS merge.  Nothing in the block ever reads those atomics' targets, so only LDS (and scalar)
// operations have to be complete before the barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ void lds_add_f32(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_f32
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  // lowers to global_atomic_add_f32 (no return) under -munsafe-fp-atomics
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Find the first wave of the run of equal items that `wv` belongs to (s_item: one item id per wave of the block).
__device__ __forceinline__ int run_head(const long long* s_item, int wv, long long item) {
  int head = wv;
  while (head > 0 && s_item[head - 1] == item) --head;
  return head;
}

// ... and one past its last wave (n_waves = waves of the block)
__device__ __forceinline__ int run_end(const long long* s_item, int wv, long long item, int n_waves) {
  int end = wv + 1;
  while (end < n_waves && s_item[end] == item) ++end;
  return end;
}

__device__ __forceinline__ float sigmoid_f32(float s) { return 1.0f / (1.0f + expf(-s)); }

// -logsigmoid(x) and sigmoid(-x) the way ATen computes them (min(x,0) - log1p(exp(-|x|)))
__device__ __forceinline__ float neg_logsigmoid(float x, float* sig_neg_x) {
  float z = expf(-fabsf(x));
  float frac = z / (1.0f + z);
  *sig_neg_x = x < 0.0f ? 1.0f - frac : frac;
  return log1pf(z) - fminf(x, 0.0f);
}

// Block-level reduction of per-wave values and publication of this block's partial sums.
// `loss_w` and `gb_w` (already scaled by 1/B) are wave-uniform; `reg_lane` is a per-lane partial.
template <int NW>
__device__ __forceinline__ void publish_partials(float loss_w, float reg_lane, float gb_w,
                                                 float inv_batch, Scratch* scratch) {
  __shared__ float s_loss[NW];
  __shared__ float s_reg[NW];
  __shared__ float s_gb[NW];
  float reg_w = wave_sum(reg_lane);
  const int w = wave_in_block();
  if (lane_id() == 0) {
    s_loss[w] = loss_w;
    s_reg[w] = reg_w;
    s_gb[w] = gb_w;
  }
  lds_barrier();
  if (threadIdx.x == 0) {
    float l = 0.f, r = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      l += s_loss[i];
      r += s_reg[i];
      b += s_gb[i];
    }
    scratch->partials[blockIdx.x] = make_float4(l * inv_batch, r * inv_batch, b, 0.f);
    if (blockIdx.x == 0) scratch->n_partials = gridDim.x;
  }
}

// Run by block 0 (all kBlock threads) of the kernel that FOLLOWS a *_grad kernel: deterministic
// reduction of the per-block partials into the device stats.  Returns (valid in thread 0) the
// gradient of the scalar bias, which the caller adds to that parameter's gradient.
// NT = threads of the calling block.  A scratch block that holds no partials (n_partials == 0:
// nothing was computed yet) leaves the stats untouched.
template <int NT = kBlock>
__device__ __forceinline__ float finalize_partials(hiprec_stats* stats, const Scratch* scratch) {
  __shared__ double s_l[NT];
  __shared__ double s_r[NT];
  __shared__ double s_b[NT];
  const uint32_t n = scratch->n_partials;
  double l = 0.0, r = 0.0, b = 0.0;
  for (uint32_t i = threadIdx.x; i < n; i += NT) {
    float4 p = scratch->partials[i];
    l += p.x;
    r += p.y;
    b += p.z;
  }
  s_l[threadIdx.x] = l;
  s_r[threadIdx.x] = r;
  s_b[threadIdx.x] = b;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s) {
      s_l[threadIdx.x] += s_l[threadIdx.x + s];
      s_r[threadIdx.x] += s_r[threadIdx.x + s];
      s_b[threadIdx.x] += s_b[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && n > 0) {
    stats->loss = static_cast<float>(s_l[0]);
    stats->reg = static_cast<float>(s_r[0]);
    stats->loss_sum += static_cast<double>(static_cast<float>(s_l[0]));
    stats->reg_sum += static_cast<double>(static_cast<float>(s_r[0]));
  }
  return static_cast<float>(s_b[0]);
}

// One thread per *_grad launch: t <- t+1 and the running beta powers used by Adam's bias correction
// (kept on the device so that a captured graph of steps can be replayed).
__device__ __forceinline__ void advance_step(hiprec_stats* stats) {
  stats->step += 1;
  stats->beta1_pow *= stats->beta1;
  stats->beta2_pow *= stats->beta2;
}

// ---- optimizer arithmetic shared by the dense sweep (optim.hip) and the fused step (mf.hip) ----
// Scalars exactly as the reference's python doubles become fp32 inside the ATen ops:
// (float)lr, (float)beta2, (float)(1 - beta1), (float)(1 - beta2), (float)eps.
struct OptScalars {
  double lr_d;
  float lr, beta2, omb1, omb2, eps;
};

// Division and square root of the Adam / RMSprop denominators.  The default uses the hardware
// v_rcp_f32 / v_sqrt_f32 (1 ulp each): the update then differs from ATen's correctly rounded
// sqrt/div by a few ulp of the UPDATE (<= 1e-6 * lr in absolute terms, two orders below the 1e-5
// parity bar), and costs ~8 VALU issue slots instead of ~35 per element -- the IEEE sequences made
// the fused Adam step VALU-bound (13.6 us per step, 2 us of it in these two functions).
// Build with -DHIPREC_IEEE_DIV for the op-for-op ATen arithmetic.
//
// Every multiply / add / fused multiply-add below is spelled out and contraction is OFF inside the function: which of
// two products of `a * b + c * d` the compiler fuses depends on the code around the call, so two kernels inlining
// the same expression differed in the last bit (the dense sweep against csrc/lazy_opt.hip's replay of it, round 4).
// With the operations pinned, every kernel that steps an element -- dense sweep, fused MF step, lazy replay --
// produces the same bits from the same inputs.  The fused forms are the ones the compiler chose for the dense sweep.
// PRE_RCP (the lazy replay, csrc/lazy_opt.hip): `bc2_sqrt` already is what the denominator multiplies by -- the
// v_rcp_f32 of the step's sqrt(1 - beta2^t), taken once when the step was recorded instead of once per replayed
// step and lane (the correctly rounded build divides, and is handed the value itself).
template <int KIND, bool PRE_RCP = false>
__device__ __forceinline__ void opt_update(float& w, float& g, float& m, float& v,
                                           const OptScalars s, float step_size, float bc2_sqrt) {
#pragma clang fp contract(off)
  if constexpr (KIND == HIPREC_OPT_SGD) {
    w = __builtin_fmaf(-s.lr, g, w);  // param.add_(grad, alpha=-lr)
  } else {
    // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)   /   square_avg.mul_(alpha).addcmul_(g, g, 1 - alpha)
    v = __builtin_fmaf(s.omb2 * g, g, v * s.beta2);
    if constexpr (KIND == HIPREC_OPT_ADAM) {
      m = __builtin_fmaf(s.omb1, g - m, m);  // exp_avg.lerp_(grad, 1 - beta1)
      const float num = -step_size * m;       // param.addcdiv_(m, denom, value=-step_size)
#ifdef HIPREC_IEEE_DIV
      const float denom = sqrtf(v) / bc2_sqrt + s.eps;  // (sqrt(v) / sqrt(bc2)).add_(eps)
      w = w + num / denom;
#else
      const float r_bc2 = PRE_RCP ? bc2_sqrt : __builtin_amdgcn_rcpf(bc2_sqrt);
      const float denom = __builtin_fmaf(__builtin_amdgcn_sqrtf(v), r_bc2, s.eps);
      w = __builtin_fmaf(num, __builtin_amdgcn_rcpf(denom), w);
#endif
    } else {
      const float num = -s.lr * g;  // param.addcdiv_(grad, avg, value=-lr)
#ifdef HIPREC_IEEE_DIV
      const float avg = sqrtf(v) + s.eps;  // square_avg.sqrt().add_(eps)
      w = w + num / avg;
#else
      const float avg = __builtin_amdgcn_sqrtf(v) + s.eps;
      w = __builtin_fmaf(num, __builtin_amdgcn_rcpf(avg), w);
#endif
    }
  }
  g = 0.f;
}

// Step-dependent scalars of the update that follows the step counted last in `stats`.
template <int KIND>
__device__ __forceinline__ void step_scalars(const OptScalars& s, const hiprec_stats* stats,
                                             float* step_size, float* bc2_sqrt) {
  *step_size = s.lr;
  *bc2_sqrt = 1.f;
  if constexpr (KIND == HIPREC_OPT_ADAM) {
    // bias_correction1 = 1 - beta1**t ; step_size = lr / bc1 ; bc2_sqrt = sqrt(1 - beta2**t)
    // (python doubles in torch, then rounded to fp32 when they enter the tensor ops)
    const double bc1 = 1.0 - stats->beta1_pow;
    const double bc2 = 1.0 - stats->beta2_pow;
    *step_size = static_cast<float>(s.lr_d / bc1);
    *bc2_sqrt = static_cast<float>(sqrt(bc2));
  }
}

// The same update split in two so that its loads travel with the caller's own first loads and its
// arithmetic + stores happen at the end of the kernel: called back to back (as advance_step does)
// the one thread that runs it, and with it its whole block, starts a memory round trip late.
struct StepState {
  long long step;
  double b1, b2, b1p, b2p;
};

__device__ __forceinline__ StepState step_load(const hiprec_stats* stats) {
  return StepState{stats->step, stats->beta1, stats->beta2, stats->beta1_pow, stats->beta2_pow};
}

__device__ __forceinline__ void step_store_advanced(hiprec_stats* stats, const StepState& s) {
  stats->step = s.step + 1;
  stats->beta1_pow = s.b1p * s.b1;
  stats->beta2_pow = s.b2p * s.b2;
}

// *p as a VECTOR load (the zero offset is opaque to the compiler).  A scalar load of a value the
// previous launch wrote is a full miss, and because scalar loads share lgkmcnt the wave's index
// loads would have to queue behind it.
__device__ __forceinline__ float load_scalar_param(const float* p) {
  int vzero = 0;
  asm volatile("" : "+v"(vzero));
  return p[vzero];
}

#endif  // __HIPCC__

}  // namespace hiprec
