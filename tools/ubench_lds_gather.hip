// Microbenchmark: what does a random 16-byte row read out of a 156 KB LDS slice cost on gfx950?  The floor of the
// column-sliced SpMM's inner loop (csrc/spmm_sliced.hip): one workgroup of 1024 threads per CU, every lane issues
// batches of ds_read_b128 at pseudo-random row addresses and adds the results.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_lds_gather.hip -o tools/ubench_lds_gather && tools/ubench_lds_gather
// MODE 0: random rows;  1: every lane the same row (broadcast);  2: conflict-free (the 16 lanes of a lane group cover
// the 16 bank quads);  3: random rows read as 2 x ds_read_b64;  4: random, 16 reads in flight (one wait per 16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)
constexpr int kRows = 9746, kThreads = 1024;
using f4 = float __attribute__((ext_vector_type(4)));
using f2 = float __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(kThreads) void k(const float* __restrict__ x, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float s[];
  for (int i = threadIdx.x; i < kRows * 4; i += kThreads) s[i] = x[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t r = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    uint32_t row[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      r = r * 1664525u + 1013904223u;
      uint32_t v = (r >> 8) % kRows;
      if (MODE == 1) v = (it * 16 + j) % kRows;
      if (MODE == 2) v = ((v >> 4) << 4) + (lane & 15) < kRows ? ((v >> 4) << 4) + (lane & 15) : (lane & 15);
      row[j] = v;
    }
    if (MODE == 4) {
      f4 t[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) t[j] = *reinterpret_cast<const f4*>(s + row[j] * 4);
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += t[j];
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (MODE == 3) {
            const f2 lo = *reinterpret_cast<const f2*>(s + row[4 * b + j] * 4), hi = *reinterpret_cast<const f2*>(s + row[4 * b + j] * 4 + 2);
            t[j] = f4{lo.x, lo.y, hi.x, hi.y};
          } else {
            t[j] = *reinterpret_cast<const f4*>(s + row[4 * b + j] * 4);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += t[j];
      }
    }
  }
  out[blockIdx.x * kThreads + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE> void run(const float* x, float* out, const char* what) {
  const int iters = 64, blocks = 256;
  const size_t lds = kRows * 16 + 64;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k<MODE><<<blocks, kThreads, lds>>>(x, out, iters);
  CK(hipEventRecord(a));
  for (int i = 0; i < 10; ++i) k<MODE><<<blocks, kThreads, lds>>>(x, out, iters);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms / 10 * 1e3;
  k<MODE><<<blocks, kThreads, lds>>>(x, out, 0);
  CK(hipEventRecord(a));
  for (int i = 0; i < 10; ++i) k<MODE><<<blocks, kThreads, lds>>>(x, out, 0);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  CK(hipEventElapsedTime(&ms, a, b));
  const double us0 = ms / 10 * 1e3;
  const double instr_per_cu = 16.0 * iters * (kThreads / 64);   // wave-instructions of 16-byte reads per CU
  printf("%-34s %7.1f us (fill + launch alone %5.1f) -> %5.2f ns per wave-read = %5.1f cycles at 2.1 GHz; %5.1f TB/s chip\n", what, us,
         us0, (us - us0) * 1e3 / instr_per_cu, (us - us0) * 1e3 / instr_per_cu * 2.1, 256 * instr_per_cu * 1024 / ((us - us0) * 1e-6) / 1e12);
}

int main() {
  float *x, *out;
  CK(hipMalloc(&x, kRows * 16)); CK(hipMemset(x, 0, kRows * 16)); CK(hipMalloc(&out, 256 * kThreads * 4));
  run<0>(x, out, "random rows, 8 in flight");
  run<1>(x, out, "one address (broadcast)");
  run<2>(x, out, "conflict-free lane groups");
  run<3>(x, out, "random rows, 2 x ds_read_b64");
  run<4>(x, out, "random rows, 16 in flight");
  return 0;
}
