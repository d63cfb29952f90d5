// Microbenchmark: random gather (and write-back) of 512-B rows from tables beyond the caches -- the floor of
// the owned-rows SGD step at BASELINE configs[3] shard size (1.25M user rows + 125k item rows of 128 floats,
// 65536 triples: 3 rows read, 3 rows written per triple).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)
constexpr int D = 128;

// MODE 0: wave per triple, 2 dword loads per row.  MODE 1: wave per triple, one 8-B load per row.
// MODE 2: wave per 8 consecutive triples, all 24 rows requested at once (8-B loads).  WRITE: rows written back.
template <int MODE, bool WRITE>
__global__ __launch_bounds__(256) void k(float* __restrict__ U, float* __restrict__ I, const int* __restrict__ idx, int B, float* out) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (MODE < 2) {
    if (wave >= B) return;
    const size_t u = (size_t)idx[3 * wave] * D, p = (size_t)idx[3 * wave + 1] * D, n = (size_t)idx[3 * wave + 2] * D;
    float s;
    if (MODE == 0) {
      float a0 = U[u + lane], a1 = U[u + 64 + lane], b0 = I[p + lane], b1 = I[p + 64 + lane], c0 = I[n + lane], c1 = I[n + 64 + lane];
      s = a0 * b0 + a1 * b1 + a0 * c0 + a1 * c1;
      if (WRITE) { U[u + lane] = a0 + 1.f; U[u + 64 + lane] = a1 + 1.f; I[p + lane] = b0 + 1.f; I[p + 64 + lane] = b1 + 1.f; I[n + lane] = c0 + 1.f; I[n + 64 + lane] = c1 + 1.f; }
    } else {
      float2 a = reinterpret_cast<float2*>(U + u)[lane], b = reinterpret_cast<float2*>(I + p)[lane], c = reinterpret_cast<float2*>(I + n)[lane];
      s = a.x * b.x + a.y * b.y + a.x * c.x + a.y * c.y;
      if (WRITE) { reinterpret_cast<float2*>(U + u)[lane] = make_float2(a.x + 1.f, a.y + 1.f); reinterpret_cast<float2*>(I + p)[lane] = make_float2(b.x + 1.f, b.y + 1.f); reinterpret_cast<float2*>(I + n)[lane] = make_float2(c.x + 1.f, c.y + 1.f); }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[wave] = s;
  } else {
    const int t0 = wave * 8;
    if (t0 >= B) return;
    float2 a[8], b[8], c[8];
    size_t u[8], p[8], n[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { u[i] = (size_t)idx[3 * (t0 + i)] * D; p[i] = (size_t)idx[3 * (t0 + i) + 1] * D; n[i] = (size_t)idx[3 * (t0 + i) + 2] * D; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = reinterpret_cast<float2*>(U + u[i])[lane]; b[i] = reinterpret_cast<float2*>(I + p[i])[lane]; c[i] = reinterpret_cast<float2*>(I + n[i])[lane]; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s += a[i].x * b[i].x + a[i].y * b[i].y + a[i].x * c[i].x + a[i].y * c[i].y;
      if (WRITE) { reinterpret_cast<float2*>(U + u[i])[lane] = make_float2(a[i].x + 1.f, a[i].y + 1.f); reinterpret_cast<float2*>(I + p[i])[lane] = make_float2(b[i].x + 1.f, b[i].y + 1.f); reinterpret_cast<float2*>(I + n[i])[lane] = make_float2(c[i].x + 1.f, c[i].y + 1.f); }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[wave] = s;
  }
}
template <int MODE, bool WRITE> float run(float* U, float* I, const int* idx, int B, float* out, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int waves = MODE < 2 ? B : (B + 7) / 8;
  for (int i = 0; i < 3; i++) k<MODE, WRITE><<<(waves + 3) / 4, 256>>>(U, I, idx, B, out);
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; i++) k<MODE, WRITE><<<(waves + 3) / 4, 256>>>(U, I, idx, B, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters * 1e3f;
}
int main() {
  const int B = 65536;
  for (int full = 0; full < 2; ++full) {
    const size_t RU = full ? 10000000 : 1250000, RI = full ? 1000000 : 125000;
    float *U, *I, *out; int* idx;
    CK(hipMalloc(&U, RU * D * 4)); CK(hipMalloc(&I, RI * D * 4)); CK(hipMalloc(&out, B * 4)); CK(hipMalloc(&idx, B * 12));
    CK(hipMemset(U, 0, RU * D * 4)); CK(hipMemset(I, 0, RI * D * 4));
    std::vector<int> h(3 * B); srand(1);
    for (int t = 0; t < B; ++t) { h[3 * t] = (int)((((unsigned long long)rand() << 16) ^ rand()) % RU); h[3 * t + 1] = (int)((((unsigned long long)rand() << 16) ^ rand()) % RI); h[3 * t + 2] = (int)((((unsigned long long)rand() << 16) ^ rand()) % RI); }
    CK(hipMemcpy(idx, h.data(), B * 12, hipMemcpyHostToDevice));
    printf("users %zu items %zu rows x 512 B, %d triples (100.7 MB read, +100.7 MB written):\n", RU, RI, B);
    printf("  read only : wave/triple dword %.1f us | wave/triple 8-B %.1f us | wave/8 triples bulk %.1f us\n", run<0, false>(U, I, idx, B, out, 30), run<1, false>(U, I, idx, B, out, 30), run<2, false>(U, I, idx, B, out, 30));
    printf("  read+write: wave/triple dword %.1f us | wave/triple 8-B %.1f us | wave/8 triples bulk %.1f us\n", run<0, true>(U, I, idx, B, out, 30), run<1, true>(U, I, idx, B, out, 30), run<2, true>(U, I, idx, B, out, 30));
    hipFree(U); hipFree(I); hipFree(out); hipFree(idx);
  }
}
