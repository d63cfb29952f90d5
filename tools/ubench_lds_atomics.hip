// Microbenchmark: what do LDS atomics cost on gfx950?  One 1024-thread workgroup per CU; every lane issues batches of
// one LDS operation at a pseudo-random (or fixed-pattern) word of a 64 KB region.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_lds_atomics.hip -o tools/ubench_lds_atomics && tools/ubench_lds_atomics
// OP 0: ds_write_b32   1: ds_add_f32 (no return)   2: ds_add_u32 (no return)   3: ds_add_rtn_u32   4: ds_cmpswap_rtn_b32
// ADDR 0: every lane its own random word   1: the 64 lanes of a wave hit 4 words   2: lane-linear (conflict-free)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)
constexpr int kWords = 16384, kThreads = 1024;

template <int OP, int ADDR>
__global__ __launch_bounds__(kThreads) void k(unsigned* __restrict__ out, int iters) {
  __shared__ unsigned s[kWords];
  for (int i = threadIdx.x; i < kWords; i += kThreads) s[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned r = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u, acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      r = r * 1664525u + 1013904223u;
      unsigned w = (r >> 8) % kWords;
      if (ADDR == 1) w = ((it * 8 + j) * 64 + (lane & 3)) % kWords;
      if (ADDR == 2) w = ((it * 8 + j) * 64 + lane) % kWords;
      if (OP == 0) s[w] = r;
      if (OP == 1) __hip_atomic_fetch_add(reinterpret_cast<float*>(s) + w, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == 2) __hip_atomic_fetch_add(s + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == 3) acc += __hip_atomic_fetch_add(s + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == 4) acc += atomicCAS(s + w, 0u, r | 1u);
    }
  }
  __syncthreads();
  out[blockIdx.x * kThreads + threadIdx.x] = acc + s[threadIdx.x];
}

template <int OP, int ADDR> void run(unsigned* out, const char* what) {
  const int iters = 64, blocks = 256;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k<OP, ADDR><<<blocks, kThreads>>>(out, iters);
  CK(hipEventRecord(a));
  for (int i = 0; i < 10; ++i) k<OP, ADDR><<<blocks, kThreads>>>(out, iters);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms / 10 * 1e3;
  k<OP, ADDR><<<blocks, kThreads>>>(out, 0);
  CK(hipEventRecord(a));
  for (int i = 0; i < 10; ++i) k<OP, ADDR><<<blocks, kThreads>>>(out, 0);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  CK(hipEventElapsedTime(&ms, a, b));
  const double us0 = ms / 10 * 1e3;
  const double instr_per_cu = 8.0 * iters * (kThreads / 64);
  printf("%-52s %7.1f us -> %6.1f cycles per wave instruction at 2.1 GHz (%4.2f per lane)\n", what, us - us0,
         (us - us0) * 1e3 / instr_per_cu * 2.1, (us - us0) * 1e3 / instr_per_cu * 2.1 / 64);
}

int main() {
  unsigned* out;
  CK(hipMalloc(&out, 256 * kThreads * 4));
  run<0, 0>(out, "ds_write_b32, random words");
  run<0, 2>(out, "ds_write_b32, lane-linear");
  run<1, 0>(out, "ds_add_f32, random words");
  run<1, 2>(out, "ds_add_f32, lane-linear");
  run<1, 1>(out, "ds_add_f32, 64 lanes on 4 words");
  run<2, 0>(out, "ds_add_u32, random words");
  run<2, 2>(out, "ds_add_u32, lane-linear");
  run<2, 1>(out, "ds_add_u32, 64 lanes on 4 words");
  run<3, 0>(out, "ds_add_rtn_u32, random words");
  run<3, 1>(out, "ds_add_rtn_u32, 64 lanes on 4 words");
  run<4, 0>(out, "ds_cmpswap_rtn_b32, random words");
  return 0;
}
