// adam.hip — torch.optim.Adam(lr, betas=(.9,.999), eps=1e-8, weight_decay=l2) with coupled L2
// (main_frame_val_text_missing.py:317) as ONE launch over the flat live-parameter bucket.
// Dead parameters (never receive a gradient) live outside the bucket and are
// never touched, exactly like torch.optim.Adam skips grad-None parameters.
//
// hyper (device, 4 floats): [0] lr (host-written, LambdaLR value)   [1] step count t (kernel-incremented)
//                           [2] lr / (1 - beta1^t)                  [3] sqrt(1 - beta2^t)
// The bias corrections are derived on the device (in double, like torch's Python-side doubles) so a
// captured hipGraph replays with the right step count.
#include "common.h"

namespace {

__global__ void adam_hyper_kernel(float* hyper, double beta1, double beta2) {
  const double t = (double)hyper[1] + 1.0;
  hyper[1] = (float)t;
  hyper[2] = (float)((double)hyper[0] / (1.0 - pow(beta1, t)));
  hyper[3] = (float)sqrt(1.0 - pow(beta2, t));
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                                   int64_t n, const float* hyper, float beta1, float beta2, float eps,
                                                   float wd, float gscale, uint32_t* rng, uint32_t rng_inc,
                                                   const sdumc_total_loss tl) {
  const float step_size = hyper[2], bc2_sqrt = hyper[3];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // A clustered utterance-level kernel whose spin ran into its cap finished with wrong data: the gradients of this step are
  // garbage, so NOTHING is applied -- parameters and moments stay as they were (every thread reads the word; it is sticky
  // until sdumc_chain_cluster_reset_error)
  const bool poisoned = tl.chain_err != nullptr && *tl.chain_err != 0;
  if (i < n4 && !poisoned) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = gg[e] * gscale + wd * pp[e];
      mm[e] = mm[e] + (ge - mm[e]) * (1.f - beta1);           // exp_avg.lerp_(grad, 1-beta1)
      vv[e] = vv[e] * beta2 + (1.f - beta2) * ge * ge;        // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
      pp[e] = pp[e] - step_size * (mm[e] / (sqrtf(vv[e]) / bc2_sqrt + eps));
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  // scalar tail (+ the dropout call counter: every reader of this step is ordered before this launch)
  if (i == 0) {
    if (rng) rng[2] += rng_inc;
    if (tl.losses) {     // the step's weighted total (main :149), for the caller's read-back: the six terms were final long ago
      float* L = tl.losses;
      L[0] = tl.w[0] * L[1] + tl.w[1] * L[2] + tl.w[2] * L[3] + tl.w[3] * L[4] + tl.w[4] * L[5] + tl.w[5] * L[6];
      L[7] = 0.f;
      // fail loudly: a clustered utterance-level kernel whose spin ran into its cap finished with wrong data
      if (poisoned) { L[0] = __int_as_float(0x7fc00000); L[7] = 1.f; }
    }
    for (int64_t t = n4 * 4; t < n && !poisoned; ++t) {
      const float ge = g[t] * gscale + wd * p[t];
      m[t] = m[t] + (ge - m[t]) * (1.f - beta1);
      v[t] = v[t] * beta2 + (1.f - beta2) * ge * ge;
      p[t] = p[t] - step_size * (m[t] / (sqrtf(v[t]) / bc2_sqrt + eps));
    }
  }
}

}  // namespace

extern "C" int sdumc_adam_hyper_(float* hyper, float beta1, float beta2, void* stream) {
  if (!hyper) return SDUMC_EINVAL;
  hipLaunchKernelGGL(adam_hyper_kernel, dim3(1), dim3(1), 0, as_stream(stream), hyper, (double)beta1, (double)beta2);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

extern "C" int sdumc_adam_apply_(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 const float* hyper, float beta1, float beta2, float eps, float weight_decay,
                                 float grad_scale, uint32_t* rng_state, uint32_t rng_inc, const sdumc_total_loss* total,
                                 void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !hyper || n <= 0) return SDUMC_EINVAL;
  sdumc_total_loss tl;
  tl.losses = nullptr;
  tl.chain_err = nullptr;
  for (int i = 0; i < 6; ++i) tl.w[i] = 0.f;
  if (total) tl = *total;
  // (callers without a loss record -- the data-parallel step, sdumc_adam_step -- are guarded by the device's error word too)
  if (!tl.chain_err) tl.chain_err = sdumc_chain_cluster_err_ptr_();
  if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
       reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
    return SDUMC_EINVAL;
  const int64_t n4 = n / 4;
  const int64_t threads = n4 > 0 ? n4 : 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream), param, grad,
                     exp_avg, exp_avg_sq, n4, n, hyper, beta1, beta2, eps, weight_decay, grad_scale, rng_state, rng_inc, tl);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

extern "C" int sdumc_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                               float* hyper, float beta1, float beta2, float eps, float weight_decay,
                               float grad_scale, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !hyper || n <= 0) return SDUMC_EINVAL;
  int rc = sdumc_adam_hyper_(hyper, beta1, beta2, stream);
  if (rc) return rc;
  return sdumc_adam_apply_(param, grad, exp_avg, exp_avg_sq, n, hyper, beta1, beta2, eps, weight_decay, grad_scale, nullptr,
                           0u, nullptr, stream);
}

// One empty kernel per source file = per gfx950 code object: sdumc_preload_() asks for its attributes, which makes the HIP runtime load
// this file's code object NOW (outside any timed or latency-sensitive region) instead of at the first launch of one of its kernels --
// with deferred loading that first launch stalls the host for tens of milliseconds (seen as a 36-59 ms gap in the middle of an epoch,
// at the first batch whose shape took a fallback path: profiles/README.md, round 6).
__global__ void sdumc_preload_adam_kernel() {}
extern "C" int sdumc_preload_adam_(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&sdumc_preload_adam_kernel)) == hipSuccess ? SDUMC_OK : SDUMC_ELAUNCH;
}
