// Microbenchmark: what does a row-granular scatter cost on gfx950?
//   mode 0: gather only (read 3 rows / triple, reduce, write 1 float)
//   mode 1: + 3 row atomicAdd (agent scope, hardware global_atomic_add_f32)
//   mode 2: + 3 row plain stores
//   mode 3: + 3 row read-modify-write (non-atomic load+store)
//   mode 4: atomics with workgroup scope
// one wave per "triple", 64 lanes = 64 columns, rows chosen pseudo-randomly in an R-row table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

template<int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ W, float* __restrict__ G, const int* __restrict__ idx, int B, int R, float* out){
  int lane = threadIdx.x & 63; int wave = blockIdx.x*4 + (threadIdx.x>>6);
  if (wave >= B) return;
  int u = idx[3*wave], p = idx[3*wave+1], n = idx[3*wave+2];
  float a = W[(size_t)u*64+lane], b = W[(size_t)p*64+lane], c = W[(size_t)n*64+lane];
  float s = a*b + a*c;
  for (int o=32;o>0;o>>=1) s += __shfl_xor(s,o);
  float d = 1.f/(1.f+__expf(-s));
  if (MODE==0){ if(lane==0) out[wave]=d; }
  if (MODE==1){ __hip_atomic_fetch_add(&G[(size_t)u*64+lane], d*b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&G[(size_t)p*64+lane], d*a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&G[(size_t)n*64+lane], d*c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);}
  if (MODE==2){ G[(size_t)u*64+lane]=d*b; G[(size_t)p*64+lane]=d*a; G[(size_t)n*64+lane]=d*c; }
  if (MODE==3){ G[(size_t)u*64+lane]+=d*b; G[(size_t)p*64+lane]+=d*a; G[(size_t)n*64+lane]+=d*c; }
  if (MODE==4){ __hip_atomic_fetch_add(&G[(size_t)u*64+lane], d*b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&G[(size_t)p*64+lane], d*a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&G[(size_t)n*64+lane], d*c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);}
}
template<int MODE> float run(const float*W,float*G,const int*idx,int B,int R,float*out,int iters){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for(int i=0;i<5;i++) k<MODE><<<(B+3)/4,256>>>(W,G,idx,B,R,out);
  CK(hipEventRecord(a));
  for(int i=0;i<iters;i++) k<MODE><<<(B+3)/4,256>>>(W,G,idx,B,R,out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/iters*1e3f;
}
int main(){
  for (int R : {9746, 1<<20}) for (int B : {4096, 65536}) {
    float *W,*G,*out; int* idx;
    CK(hipMalloc(&W,(size_t)R*256)); CK(hipMalloc(&G,(size_t)R*256)); CK(hipMalloc(&out,B*4)); CK(hipMalloc(&idx,B*12));
    CK(hipMemset(W,0,(size_t)R*256)); CK(hipMemset(G,0,(size_t)R*256));
    std::vector<int> h(3*B); srand(1); for(auto&x:h) x = (int)(((unsigned)rand()*2654435761u)%R);
    CK(hipMemcpy(idx,h.data(),B*12,hipMemcpyHostToDevice));
    int it = B>4096?50:200;
    printf("R=%7d B=%6d  gather %.2f  atomic-agent %.2f  store %.2f  rmw %.2f  atomic-wg %.2f  (us)\n",R,B,
      run<0>(W,G,idx,B,R,out,it),run<1>(W,G,idx,B,R,out,it),run<2>(W,G,idx,B,R,out,it),run<3>(W,G,idx,B,R,out,it),run<4>(W,G,idx,B,R,out,it));
    hipFree(W);hipFree(G);hipFree(out);hipFree(idx);
  }
}
