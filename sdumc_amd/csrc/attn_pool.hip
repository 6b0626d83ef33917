// attn_pool.hip — softmax-over-time attention pooling, forward and backward.
//
// The body shared by FRA2UTT_new.forward (model :56-68, one shared query) and
// Cross_Attention.forward (model :79-95, the 7 view queries) once the tanh key
// projection exists (produced by gemm_f32.hip with the input dropout fused):
//     S[v,t,i] = K[v,t,:] . Q[v,i,:]          (torch.bmm,            model :61 / :88)
//     A        = softmax_t(0.3 * S)           (F.softmax dim=1,      model :63 / :90)
//     O[v,i,:] = sum_t A[v,t,i] * xd[v,t,:]   (7 mul + sum + cat,    model :64-66 / :91-93)
//     out      = dropout(O)                   (model :67 / :94)
// xd = dropout(x) is never materialised: the Philox mask is recomputed from
// (seed, call, site, sample, t, channel) wherever xd is needed.
//
// These are HBM/L2-streaming kernels (4*T*D*nq flops against a T*D tile): rows are
// read as whole 1-KiB lines (one 16-B load per lane), dot products reduce with
// wave64 shuffles, the softmax statistics live in LDS.
#include "common.h"

namespace {

constexpr int D = SDUMC_D;       // 256 channels = 64 lanes x 4
constexpr int MAXQ = 8;
constexpr int ROWS_PER_WG = 64;  // rows of one v handled by a 4-wave workgroup in the row kernels

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

// ---- forward pass 1: scaled scores S -> attn buffer ------------------------------------------
__global__ __launch_bounds__(256) void scores_kernel(const sdumc_attnpool p) {
  const int v = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 q[MAXQ];
#pragma unroll
  for (int i = 0; i < MAXQ; ++i)
    if (i < p.nq) q[i] = ld4(p.q + (size_t)v * p.q_stride + (size_t)i * D + 4 * lane);
  const int t0 = blockIdx.x * ROWS_PER_WG;
  for (int r = wave; r < ROWS_PER_WG; r += 4) {
    const int t = t0 + r;
    if (t >= p.T) break;
    const f32x4 k = ld4(p.keys + ((size_t)v * p.T + t) * D + 4 * lane);
    float mine = 0.f;
#pragma unroll
    for (int i = 0; i < MAXQ; ++i)
      if (i < p.nq) {
        const float s = wave_sum(dot4(k, q[i]));
        if (lane == i) mine = s;
      }
    if (lane < p.nq) p.attn[((size_t)v * p.T + t) * p.nq + lane] = p.scale * mine;
  }
}

// ---- forward pass 2: softmax over T (per query) + pooling + output dropout ---------------------
// one workgroup per virtual sample; dynamic LDS: attn [T*nq] floats + reduction scratch
__global__ __launch_bounds__(256) void softmax_pool_kernel(const sdumc_attnpool p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* a_s = smem;                        // [T][nq]
  float* red = smem + (size_t)p.T * p.nq;   // [4][MAXQ][256] for the cross-wave pooling reduce
  const int v = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = p.T, nq = p.nq;
  float* attn = p.attn + (size_t)v * T * nq;
  for (int e = tid; e < T * nq; e += 256) a_s[e] = attn[e];
  __syncthreads();
  // softmax statistics: wave w owns queries w, w+4
  for (int i = wave; i < nq; i += 4) {
    float m = -INFINITY;
    for (int t = lane; t < T; t += 64) m = fmaxf(m, a_s[t * nq + i]);
    m = wave_max(m);
    float l = 0.f;
    for (int t = lane; t < T; t += 64) {
      const float e = expf(a_s[t * nq + i] - m);
      a_s[t * nq + i] = e;
      l += e;
    }
    l = wave_sum(l);
    const float inv = 1.f / l;
    for (int t = lane; t < T; t += 64) a_s[t * nq + i] *= inv;
  }
  __syncthreads();
  for (int e = tid; e < T * nq; e += 256) attn[e] = a_s[e];

  // pooling: wave w takes rows t = w, w+4, ...; lane owns channels 4*lane..4*lane+3
  const DropRT xd = drop_resolve(p.x_drop);
  const int vx = v % p.x_samples;
  const float* xrow = p.x + (size_t)vx * T * D + 4 * lane;
  f32x4 acc[MAXQ];
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int t = wave; t < T; t += 4) {
    f32x4 x = ld4(xrow + (size_t)t * D);
    if (xd.enabled) x *= drop_mask4(xd, (uint32_t)(v * T + t), (uint32_t)lane);
#pragma unroll
    for (int i = 0; i < MAXQ; ++i)
      if (i < nq) acc[i] += x * a_s[t * nq + i];
  }
#pragma unroll
  for (int i = 0; i < MAXQ; ++i)
    if (i < nq) st4(red + ((size_t)(wave * MAXQ + i)) * D + 4 * lane, acc[i]);
  __syncthreads();
  const DropRT od = drop_resolve(p.out_drop);
  for (int e = tid; e < nq * (D / 4); e += 256) {
    const int i = e / (D / 4), cq = e - i * (D / 4);
    f32x4 s = ld4(red + (size_t)i * D + 4 * cq);
#pragma unroll
    for (int w = 1; w < 4; ++w) s += ld4(red + ((size_t)(w * MAXQ + i)) * D + 4 * cq);
    const size_t o = ((size_t)v * nq + i) * D + 4 * cq;
    st4(p.pooled + o, s);
    if (od.enabled) s *= drop_mask4(od, (uint32_t)(v * nq + i), (uint32_t)cq);
    st4(p.out + o, s);
  }
}

// ---- backward: one pass over the rows of (v, T-chunk) ----------------------------------------
// dO = dout * out_mask ; delta_i = dO_i . O_i ; per row t:
//   dA_i = dO_i . xd_t ; dS_i = 0.3 A_ti (dA_i - delta_i)
//   dK_t = sum_i dS_i Q_i ; dz_t = dK_t (1 - K_t^2)            -> dz
//   dxd_t (pool path) = sum_i A_ti dO_i                         -> dxd
//   dQ_i += dS_i K_t                                            -> per-chunk partials (deterministic)
__global__ __launch_bounds__(256) void attnpool_bwd_kernel(const sdumc_attnpool_bwd_t b, float* dq_part,
                                                           const int nchunk) {
  __shared__ __attribute__((aligned(16))) float dO_s[MAXQ * D];
  __shared__ __attribute__((aligned(16))) float red[4 * MAXQ * D];
  __shared__ float delta_s[MAXQ];
  const sdumc_attnpool& p = b.f;
  const int v = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = p.T, nq = p.nq;
  const DropRT od = drop_resolve(p.out_drop);
  for (int e = tid; e < nq * (D / 4); e += 256) {
    const int i = e / (D / 4), cq = e - i * (D / 4);
    f32x4 g = ld4(b.dout + ((size_t)v * nq + i) * D + 4 * cq);
    if (od.enabled) g *= drop_mask4(od, (uint32_t)(v * nq + i), (uint32_t)cq);
    st4(dO_s + i * D + 4 * cq, g);
  }
  __syncthreads();
  for (int i = wave; i < nq; i += 4) {
    const float d = wave_sum(dot4(ld4(dO_s + i * D + 4 * lane), ld4(p.pooled + ((size_t)v * nq + i) * D + 4 * lane)));
    if (lane == 0) delta_s[i] = d;
  }
  __syncthreads();

  f32x4 q[MAXQ], dqa[MAXQ];
  float delta[MAXQ];
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    dqa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < nq) {
      q[i] = ld4(p.q + (size_t)v * p.q_stride + (size_t)i * D + 4 * lane);
      delta[i] = delta_s[i];
    }
  }
  const DropRT xd = drop_resolve(p.x_drop);
  const int vx = v % p.x_samples;
  const int t0 = chunk * ROWS_PER_WG;
  for (int r = wave; r < ROWS_PER_WG; r += 4) {
    const int t = t0 + r;
    if (t >= T) break;
    const size_t row = (size_t)v * T + t;
    f32x4 x = ld4(p.x + ((size_t)vx * T + t) * D + 4 * lane);
    if (xd.enabled) x *= drop_mask4(xd, (uint32_t)row, (uint32_t)lane);
    const f32x4 k = ld4(p.keys + row * D + 4 * lane);
    f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MAXQ; ++i)
      if (i < nq) {
        const f32x4 g = ld4(dO_s + i * D + 4 * lane);
        const float a = p.attn[row * nq + i];
        const float dA = wave_sum(dot4(g, x));
        const float dS = p.scale * a * (dA - delta[i]);
        dk += q[i] * dS;
        dx += g * a;
        dqa[i] += k * dS;
      }
    const f32x4 one = {1.f, 1.f, 1.f, 1.f};
    st4(b.dz + row * D + 4 * lane, dk * (one - k * k));
    st4(b.dxd + row * D + 4 * lane, dx);
  }
  // cross-wave reduce of the dQ partials, then one deterministic slab per (v, chunk)
#pragma unroll
  for (int i = 0; i < MAXQ; ++i)
    if (i < nq) st4(red + (wave * MAXQ + i) * D + 4 * lane, dqa[i]);
  __syncthreads();
  for (int e = tid; e < nq * (D / 4); e += 256) {
    const int i = e / (D / 4), cq = e - i * (D / 4);
    f32x4 s = ld4(red + i * D + 4 * cq);
#pragma unroll
    for (int w = 1; w < 4; ++w) s += ld4(red + (w * MAXQ + i) * D + 4 * cq);
    st4(dq_part + (((size_t)v * nchunk + chunk) * nq + i) * D + 4 * cq, s);
  }
}

__global__ __launch_bounds__(256) void dq_reduce_kernel(const float* part, float* dq, const int nchunk,
                                                        const int per_v /* nq*256 */, const size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const size_t v = idx / per_v, e = idx - v * per_v;
  const float* s = part + v * nchunk * per_v + e;
  float a = 0.f;
  for (int c = 0; c < nchunk; ++c) a += s[(size_t)c * per_v];
  dq[idx] = a;
}

int check(const sdumc_attnpool& p) {
  if (p.V <= 0 || p.T <= 0 || p.nq < 1 || p.nq > MAXQ || p.x_samples <= 0) return SDUMC_EINVAL;
  if (!p.x || !p.keys || !p.q || !p.attn || !p.pooled || !p.out) return SDUMC_EINVAL;
  if ((size_t)p.T * p.nq * 4 + 4 * MAXQ * D * 4 > 160 * 1024) return SDUMC_EINVAL;  // LDS budget
  return SDUMC_OK;
}

}  // namespace

extern "C" int sdumc_attnpool_fwd(const sdumc_attnpool* pp, void* stream) {
  if (!pp) return SDUMC_EINVAL;
  const sdumc_attnpool& p = *pp;
  int rc = check(p);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(scores_kernel, dim3((p.T + ROWS_PER_WG - 1) / ROWS_PER_WG, p.V), dim3(256), 0, st, p);
  SDUMC_CHECK_LAUNCH();
  const size_t lds = ((size_t)p.T * p.nq + 4 * MAXQ * D) * sizeof(float);
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)softmax_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return SDUMC_ELAUNCH;
  }
  hipLaunchKernelGGL(softmax_pool_kernel, dim3(p.V), dim3(256), lds, st, p);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

extern "C" size_t sdumc_attnpool_bwd_workspace_bytes(int32_t V, int32_t T, int32_t nq) {
  const size_t nchunk = (size_t)(T + ROWS_PER_WG - 1) / ROWS_PER_WG;
  return (size_t)V * nchunk * nq * D * sizeof(float);
}

extern "C" int sdumc_attnpool_bwd(const sdumc_attnpool_bwd_t* bp, void* stream) {
  if (!bp) return SDUMC_EINVAL;
  const sdumc_attnpool_bwd_t& b = *bp;
  int rc = check(b.f);
  if (rc) return rc;
  if (!b.dout || !b.dz || !b.dxd || !b.dq || !b.workspace) return SDUMC_EINVAL;
  const sdumc_attnpool& p = b.f;
  const int nchunk = (p.T + ROWS_PER_WG - 1) / ROWS_PER_WG;
  if (b.workspace_bytes < sdumc_attnpool_bwd_workspace_bytes(p.V, p.T, p.nq)) return SDUMC_ENOMEM;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(attnpool_bwd_kernel, dim3(nchunk, p.V), dim3(256), 0, st, b, b.workspace, nchunk);
  SDUMC_CHECK_LAUNCH();
  const size_t total = (size_t)p.V * p.nq * D;
  hipLaunchKernelGGL(dq_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, b.workspace, b.dq,
                     nchunk, p.nq * D, total);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}
