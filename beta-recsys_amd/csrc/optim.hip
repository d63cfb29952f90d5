// Dense optimizer sweeps: torch.optim.{SGD,Adam,RMSprop}.step() as configured by
// beta_rec/models/torch_engine.py:23-39 (only `lr` given => Adam betas (0.9, 0.999), eps 1e-8,
// RMSprop alpha 0.99 eps 1e-8, no momentum, no weight decay, no amsgrad).
//
// nn.Embedding is non-sparse in the reference, so autograd produces DENSE gradients and Adam /
// RMSprop move every element whose moments are non-zero on every step, touched by the batch or
// not.  Parity therefore needs a sweep over the whole parameter buffer; it is fused with the next
// step's zero_grad (g is cleared as it is consumed): 16-B vectors, grid-stride, pure HBM stream of
// 28 B/element (Adam: read w,m,v,g + write w,m,v,g=0 -> 32 B with the clear).
//
// The arithmetic follows torch/optim/adam.py::_single_tensor_adam and rmsprop.py op by op in fp32;
// step-dependent scalars (bias corrections) are derived on the device from hiprec_stats so that a
// captured hipGraph of steps stays replayable.
#include "common.hpp"

namespace hiprec {

// CLIP: torch.nn.utils.clip_grad_norm_'s scaling rides in the sweep -- clip_ws[2 ..] holds the per-block sums of
// squares hiprec's clip_sumsq_kernel left (csrc/pgmf.hip); every block reduces them in clip_scale_kernel's order
// (the same total_norm / coef bits), multiplies its gradients by coef before stepping them (the same two fp32
// operations as scaling in memory first) and block 0 publishes (total_norm, coef) in clip_ws[0 .. 1].  One launch
// and one pass over g less per step than clip_scale_kernel + the sweep.
template <int KIND, bool CLIP = false>
__global__ __launch_bounds__(kBlock) void opt_dense_kernel(float* __restrict__ w,
                                                           float* __restrict__ g,
                                                           float* __restrict__ m,
                                                           float* __restrict__ v, int64_t n,
                                                           OptScalars s, hiprec_stats* stats,
                                                           const Scratch* scratch,
                                                           int64_t scalar_index, double* clip_ws = nullptr,
                                                           int n_clip = 0, float max_norm = 0.f) {
  float step_size, bc2_sqrt;
  step_scalars<KIND>(s, stats, &step_size, &bc2_sqrt);
  float coef = 1.f;
  if constexpr (CLIP) {
    __shared__ double s_p[kBlock];
    double t = 0.0;
    for (int i = threadIdx.x; i < n_clip; i += kBlock) t += clip_ws[2 + i];
    s_p[threadIdx.x] = t;
    __syncthreads();
    for (int r = kBlock / 2; r > 0; r >>= 1) {
      if (static_cast<int>(threadIdx.x) < r) s_p[threadIdx.x] += s_p[threadIdx.x + r];
      __syncthreads();
    }
    const float total_norm = static_cast<float>(sqrt(s_p[0]));
    const float raw = max_norm / (total_norm + 1e-6f);
    coef = raw > 1.0f ? 1.0f : raw;  // torch.clamp(max=1.0): a NaN stays a NaN
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      clip_ws[0] = static_cast<double>(total_norm);
      clip_ws[1] = static_cast<double>(coef);
    }
  }
  const bool scale = CLIP && !(coef >= 1.0f);  // g * 1.0f is g; a NaN norm poisons g like torch
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  const int64_t n4 = n >> 2;
  // The element at scalar_index (MF's global_bias) receives its gradient through the scratch
  // partials of the preceding *_grad call.  Its 16-B vector (or tail element) is left out of the
  // sweep and updated by thread 0 of block 0 once the partials are reduced.
  const bool deferred = scratch != nullptr && scalar_index >= 0 && scalar_index < n;
  const int64_t skip4 = (deferred && scalar_index < (n4 << 2)) ? (scalar_index >> 2) : -1;
  const int64_t skip1 = (deferred && scalar_index >= (n4 << 2)) ? scalar_index : -1;
  float4* w4 = reinterpret_cast<float4*>(w);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (int64_t i = tid; i < n4; i += stride) {
    if (i == skip4) continue;
    float4 wv = w4[i], gv = g4[i];
    float4 mv = make_float4(0, 0, 0, 0), vv = make_float4(0, 0, 0, 0);
    if constexpr (KIND == HIPREC_OPT_ADAM) mv = m4[i];
    if constexpr (KIND != HIPREC_OPT_SGD) vv = v4[i];
    if (scale) {
      gv.x *= coef;
      gv.y *= coef;
      gv.z *= coef;
      gv.w *= coef;
    }
    opt_update<KIND>(wv.x, gv.x, mv.x, vv.x, s, step_size, bc2_sqrt);
    opt_update<KIND>(wv.y, gv.y, mv.y, vv.y, s, step_size, bc2_sqrt);
    opt_update<KIND>(wv.z, gv.z, mv.z, vv.z, s, step_size, bc2_sqrt);
    opt_update<KIND>(wv.w, gv.w, mv.w, vv.w, s, step_size, bc2_sqrt);
    w4[i] = wv;
    g4[i] = gv;
    if constexpr (KIND == HIPREC_OPT_ADAM) m4[i] = mv;
    if constexpr (KIND != HIPREC_OPT_SGD) v4[i] = vv;
  }
  auto scalar_update = [&](int64_t i, float extra_g) {
    float wv = w[i], gv = g[i], mv = 0.f, vv = 0.f;
    if (scale) gv *= coef;
    gv += extra_g;
    if constexpr (KIND == HIPREC_OPT_ADAM) mv = m[i];
    if constexpr (KIND != HIPREC_OPT_SGD) vv = v[i];
    opt_update<KIND>(wv, gv, mv, vv, s, step_size, bc2_sqrt);
    w[i] = wv;
    g[i] = gv;
    if constexpr (KIND == HIPREC_OPT_ADAM) m[i] = mv;
    if constexpr (KIND != HIPREC_OPT_SGD) v[i] = vv;
  };
  for (int64_t i = (n4 << 2) + tid; i < n; i += stride) {  // scalar tail (< 4 elements)
    if (i != skip1) scalar_update(i, 0.f);
  }
  if (blockIdx.x == 0 && scratch) {
    const float gb_part = finalize_partials(stats, scratch);
    if (threadIdx.x == 0 && deferred) {
      if (skip4 >= 0) {
        for (int64_t i = skip4 << 2; i < (skip4 << 2) + 4; ++i)
          scalar_update(i, i == scalar_index ? gb_part : 0.f);
      } else {
        scalar_update(skip1, gb_part);
      }
    }
  }
}

}  // namespace hiprec

using namespace hiprec;

namespace hiprec {
int opt_dense_step_impl(int kind, float* w, float* g, float* m, float* v, int64_t n, double lr, double beta1,
                        double beta2, double eps, hiprec_stats* stats, const void* scratch, int64_t scalar_index,
                        double* clip_ws, int n_clip, float max_norm, void* stream) {
  HIPREC_REQUIRE(w && g && stats, "NULL w/g/stats");
  HIPREC_REQUIRE(n >= 0, "negative n");
  HIPREC_REQUIRE((reinterpret_cast<uintptr_t>(w) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(g) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(m) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(v) & 15) == 0,
                 "flat buffers must be 16-byte aligned");
  const OptScalars s{lr,
                     static_cast<float>(lr),
                     static_cast<float>(beta2),
                     static_cast<float>(1.0 - beta1),
                     static_cast<float>(1.0 - beta2),
                     static_cast<float>(eps)};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = grid_for_threads((n + 3) / 4);
  const auto* sc = static_cast<const Scratch*>(scratch);
#define HIPREC_OPT_LAUNCH(KIND)                                                                                     \
  do {                                                                                                             \
    if (clip_ws)                                                                                                   \
      opt_dense_kernel<KIND, true><<<grid, kBlock, 0, st>>>(w, g, m, v, n, s, stats, sc, scalar_index, clip_ws,    \
                                                            n_clip, max_norm);                                    \
    else                                                                                                           \
      opt_dense_kernel<KIND><<<grid, kBlock, 0, st>>>(w, g, m, v, n, s, stats, sc, scalar_index);                  \
  } while (0)
  switch (kind) {
    case HIPREC_OPT_SGD:
      HIPREC_OPT_LAUNCH(HIPREC_OPT_SGD);
      break;
    case HIPREC_OPT_ADAM:
      HIPREC_REQUIRE(m && v, "adam needs exp_avg / exp_avg_sq buffers");
      HIPREC_OPT_LAUNCH(HIPREC_OPT_ADAM);
      break;
    case HIPREC_OPT_RMSPROP:
      HIPREC_REQUIRE(v, "rmsprop needs a square_avg buffer");
      HIPREC_OPT_LAUNCH(HIPREC_OPT_RMSPROP);
      break;
    default:
      set_error("unknown optimizer kind %d", kind);
      return HIPREC_E_UNSUPPORTED;
  }
#undef HIPREC_OPT_LAUNCH
  HIPREC_TRY(hipGetLastError());
  return 0;
}
}  // namespace hiprec

extern "C" int hiprec_opt_dense_step(int kind, float* w, float* g, float* m, float* v, int64_t n,
                                     double lr, double beta1, double beta2, double eps,
                                     hiprec_stats* stats, const void* scratch,
                                     int64_t scalar_index, void* stream) {
  return opt_dense_step_impl(kind, w, g, m, v, n, lr, beta1, beta2, eps, stats, scratch, scalar_index, nullptr, 0,
                             0.f, stream);
}
