// transformer.hip — generic multi-head attention, LayerNorm and the residual/dropout glue of the pre-LN
// Transformer encoder.
//
// Replaces toolkit/models/modules/transformers_encoder/{multihead_attention.py:48-131, transformer.py:137-176,
// :201-203, position_embedding.py:8-26} -- an orphan module of the reference (nothing on the SDUMC step imports
// it), built here because north_star names "multi-head QKV projection, scaled dot-product attention, softmax,
// output projection, LayerNorm" literally.
//
// Design (MI355X-first):
//   * Every product runs on the fp32 MFMA GEMM of gemm_f32.hip.  The per-(sample, head) QK^T / PV products are
//     ONE strided-batched launch each: with [T, B, H*d_h] activations the head slice of (b, h) starts at
//     (b*H + h)*d_h, so a single stride walks samples and heads and no transpose/reshape copy
//     (multihead_attention.py:87-91 `.contiguous().view().transpose()`) ever exists.
//   * The probabilities must be materialised anyway (the reference returns their head average, :128-130, and the
//     backward needs them), so the softmax is a separate HBM-bound kernel rather than a flash-style fusion:
//     one wave64 per (sample, query) row walks the H heads, keeps a row in registers (<= 2048 keys), reduces
//     max / sum with wavefront shuffles, applies scale + additive mask + dropout and accumulates the head mean
//     in registers: S is read once, P written once, the [B, Tq, Tk] weights written once.
//   * LayerNorm: one wave per row, row cached in registers, two-pass variance (no E[x^2]-E[x]^2 cancellation).
//   * No float atomics anywhere: parameter-gradient reductions are two-stage and ordered.
#include <algorithm>

#include "common.h"

namespace {

template <int W>
__device__ __forceinline__ void ldu(float (&v)[W], const float* p) {
  if constexpr (W == 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else {
    v[0] = *p;
  }
}
template <int W>
__device__ __forceinline__ void stu(float* p, const float (&v)[W]) {
  if constexpr (W == 4) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  } else {
    *p = v[0];
  }
}

// Visit the units (W consecutive floats) of one row that belong to this lane: u = lane + 64 i.
// MAXU > 0: at most MAXU units per lane, compile-time trip count (the row lives in registers, slot i);
// MAXU = 0: any length, the body re-reads memory (slot 0).
template <int MAXU, typename F>
__device__ __forceinline__ void for_units(int nu, int lane, F&& f) {
  if constexpr (MAXU > 0) {
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
      const int u = lane + 64 * i;
      if (u < nu) f(i, u);
    }
  } else {
    for (int u = lane; u < nu; u += 64) f(0, u);
  }
}

template <int W>
__device__ __forceinline__ void drop_unit(float (&m)[W], const DropRT& d, uint32_t vrow, int u) {
  if constexpr (W == 4) {
    const f32x4 t = drop_mask4(d, vrow, (uint32_t)u);
    m[0] = t[0]; m[1] = t[1]; m[2] = t[2]; m[3] = t[3];
  } else {
    m[0] = drop_mask1(d, vrow, (uint32_t)u);
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm
// ------------------------------------------------------------------------------------------------
template <int W, int MAXU>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int64_t rows, int width, float eps) {
  constexpr int NC = MAXU > 0 ? MAXU : 1;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * width;
  float* yr = y + row * width;
  const int nu = width / W;
  float v[NC][W];
  float s = 0.f;
  for_units<MAXU>(nu, lane, [&](int i, int u) {
    ldu<W>(v[i], xr + u * W);
#pragma unroll
    for (int e = 0; e < W; ++e) s += v[i][e];
  });
  const float mu = wave_sum(s) / (float)width;
  float ss = 0.f;
  for_units<MAXU>(nu, lane, [&](int i, int u) {
    if constexpr (MAXU == 0) ldu<W>(v[0], xr + u * W);
#pragma unroll
    for (int e = 0; e < W; ++e) {
      const float d = v[i][e] - mu;
      ss += d * d;
    }
  });
  const float rs = 1.f / sqrtf(wave_sum(ss) / (float)width + eps);
  for_units<MAXU>(nu, lane, [&](int i, int u) {
    if constexpr (MAXU == 0) ldu<W>(v[0], xr + u * W);
    float g[W], b[W], o[W];
    ldu<W>(g, gamma + u * W);
    ldu<W>(b, beta + u * W);
#pragma unroll
    for (int e = 0; e < W; ++e) o[e] = (v[i][e] - mu) * rs * g[e] + b[e];
    stu<W>(yr + u * W, o);
  });
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat))
template <int W, int MAXU>
__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* dx,
                                                               const float* dx_add, int64_t rows, int width) {
  constexpr int NC = MAXU > 0 ? MAXU : 1;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * width;
  const float* dr = dy + row * width;
  float* or_ = dx + row * width;
  const float* ar = dx_add ? dx_add + row * width : nullptr;
  const int nu = width / W;
  const float mu = mean[row], rs = rstd[row];
  float xh[NC][W], dh[NC][W];   // xhat and g*dy
  float s1 = 0.f, s2 = 0.f;
  auto fetch = [&](int i, int u) {
    float g[W];
    ldu<W>(xh[i], xr + u * W);
    ldu<W>(dh[i], dr + u * W);
    ldu<W>(g, gamma + u * W);
#pragma unroll
    for (int e = 0; e < W; ++e) {
      xh[i][e] = (xh[i][e] - mu) * rs;
      dh[i][e] *= g[e];
    }
  };
  for_units<MAXU>(nu, lane, [&](int i, int u) {
    fetch(i, u);
#pragma unroll
    for (int e = 0; e < W; ++e) {
      s1 += dh[i][e];
      s2 += dh[i][e] * xh[i][e];
    }
  });
  const float m1 = wave_sum(s1) / (float)width, m2 = wave_sum(s2) / (float)width;
  for_units<MAXU>(nu, lane, [&](int i, int u) {
    if constexpr (MAXU == 0) fetch(0, u);
    float o[W];
    if (ar) ldu<W>(o, ar + u * W);
#pragma unroll
    for (int e = 0; e < W; ++e) {
      const float d = rs * (dh[i][e] - m1 - xh[i][e] * m2);
      o[e] = ar ? o[e] + d : d;
    }
    stu<W>(or_ + u * W, o);
  });
}

// dgamma / dbeta partials: thread = column, workgroup = (256 columns) x (one chunk of rows)
__global__ __launch_bounds__(256) void layernorm_bwd_gb_stage1(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, int64_t rows, int width,
                                                               int rows_per_chunk, float* __restrict__ part) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  if (c >= width) return;
  float dg = 0.f, db = 0.f;
#pragma unroll 4
  for (int64_t r = r0; r < r1; ++r) {
    const float d = dy[r * width + c];
    dg += d * (x[r * width + c] - mean[r]) * rstd[r];
    db += d;
  }
  part[((size_t)blockIdx.y * 2) * width + c] = dg;
  part[((size_t)blockIdx.y * 2 + 1) * width + c] = db;
}

__global__ __launch_bounds__(256) void layernorm_bwd_gb_stage2(const float* __restrict__ part, int nchunk, int width,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= width) return;
  float dg = 0.f, db = 0.f;
  for (int k = 0; k < nchunk; ++k) {
    dg += part[((size_t)k * 2) * width + c];
    db += part[((size_t)k * 2 + 1) * width + c];
  }
  dgamma[c] = dg;
  dbeta[c] = db;
}

int ln_chunks(int64_t rows, int* rows_per_chunk) {
  // <= 128 row chunks: stage 2 walks them serially per column (512 chunks cost 124 us at 16k rows x 1024; 128: ~30 us)
  int rpc = (int)std::max<int64_t>(64, (rows + 127) / 128);
  *rows_per_chunk = rpc;
  return (int)((rows + rpc - 1) / rpc);
}

// ------------------------------------------------------------------------------------------------
// Softmax over keys + additive mask + dropout + head mean
// ------------------------------------------------------------------------------------------------
template <int W, int MAXU>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const sdumc_softmax p) {
  constexpr int NC = MAXU > 0 ? MAXU : 1;
  const int lane = threadIdx.x & 63;
  const int64_t rid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (b, tq)
  if (rid >= (int64_t)p.batch * p.tq) return;
  const int b = (int)(rid / p.tq), tq = (int)(rid - (int64_t)b * p.tq);
  const int tk = p.tk, nu = tk / W;
  const float* mrow = p.mask ? p.mask + (size_t)tq * tk : nullptr;
  float* wrow = p.weights ? p.weights + (size_t)rid * tk : nullptr;
  const DropRT dr = drop_resolve(p.drop);
  const float invh = 1.f / (float)p.heads;
  float v[NC][W], acc[NC][W];
  if constexpr (MAXU > 0) {
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int e = 0; e < W; ++e) acc[i][e] = 0.f;
  }
  for (int h = 0; h < p.heads; ++h) {
    const uint32_t vrow = (uint32_t)((b * p.heads + h) * p.tq + tq);
    float* sr = p.scores + (size_t)vrow * tk;
    float* pd = dr.enabled ? p.probs_drop + (size_t)vrow * tk : nullptr;
    auto logits = [&](int i, int u) {   // scale * S + mask
      ldu<W>(v[i], sr + u * W);
      float mk[W];
      if (mrow) ldu<W>(mk, mrow + u * W);
#pragma unroll
      for (int e = 0; e < W; ++e) v[i][e] = mrow ? v[i][e] * p.scale + mk[e] : v[i][e] * p.scale;
    };
    float mx = -INFINITY;
    for_units<MAXU>(nu, lane, [&](int i, int u) {
      logits(i, u);
#pragma unroll
      for (int e = 0; e < W; ++e) mx = fmaxf(mx, v[i][e]);
    });
    mx = wave_max(mx);
    float sum = 0.f;
    for_units<MAXU>(nu, lane, [&](int i, int u) {
      if constexpr (MAXU == 0) logits(0, u);
#pragma unroll
      for (int e = 0; e < W; ++e) {
        v[i][e] = expf(v[i][e] - mx);
        sum += v[i][e];
      }
    });
    sum = wave_sum(sum);
    for_units<MAXU>(nu, lane, [&](int i, int u) {
      if constexpr (MAXU == 0) {
        logits(0, u);
#pragma unroll
        for (int e = 0; e < W; ++e) v[0][e] = expf(v[0][e] - mx);
      }
      float o[W];
#pragma unroll
      for (int e = 0; e < W; ++e) o[e] = v[i][e] / sum;
      stu<W>(sr + u * W, o);
      if (dr.enabled) {
        float m[W];
        drop_unit<W>(m, dr, vrow, u);
#pragma unroll
        for (int e = 0; e < W; ++e) o[e] *= m[e];
        stu<W>(pd + u * W, o);
      }
      if (wrow) {
        if constexpr (MAXU > 0) {
#pragma unroll
          for (int e = 0; e < W; ++e) acc[i][e] += o[e];
        } else {   // long rows: the running head sum lives in the output row (this wave is its only writer)
          float a[W];
          if (h > 0) ldu<W>(a, wrow + u * W);
#pragma unroll
          for (int e = 0; e < W; ++e) {
            a[e] = h > 0 ? a[e] + o[e] : o[e];
            if (h == p.heads - 1) a[e] *= invh;
          }
          stu<W>(wrow + u * W, a);
        }
      }
    });
  }
  if constexpr (MAXU > 0) {
    if (wrow) {
      for_units<MAXU>(nu, lane, [&](int i, int u) {
        float o[W];
#pragma unroll
        for (int e = 0; e < W; ++e) o[e] = acc[i][e] * invh;
        stu<W>(wrow + u * W, o);
      });
    }
  }
}

// dS = scale * P * (dP - sum_k dP_k P_k), dP = dropout mask * incoming
template <int W, int MAXU>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const sdumc_softmax p, float* __restrict__ dscores) {
  constexpr int NC = MAXU > 0 ? MAXU : 1;
  const int lane = threadIdx.x & 63;
  const int64_t vrow = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (vrow >= (int64_t)p.batch * p.heads * p.tq) return;
  const int tk = p.tk, nu = tk / W;
  const float* pr = p.scores + (size_t)vrow * tk;
  float* dr_ = dscores + (size_t)vrow * tk;
  const DropRT dr = drop_resolve(p.drop);
  float pv[NC][W], dv[NC][W];
  auto fetch = [&](int i, int u) {
    ldu<W>(pv[i], pr + u * W);
    ldu<W>(dv[i], dr_ + u * W);
    if (dr.enabled) {
      float m[W];
      drop_unit<W>(m, dr, (uint32_t)vrow, u);
#pragma unroll
      for (int e = 0; e < W; ++e) dv[i][e] *= m[e];
    }
  };
  float dot = 0.f;
  for_units<MAXU>(nu, lane, [&](int i, int u) {
    fetch(i, u);
#pragma unroll
    for (int e = 0; e < W; ++e) dot += dv[i][e] * pv[i][e];
  });
  dot = wave_sum(dot);
  for_units<MAXU>(nu, lane, [&](int i, int u) {
    if constexpr (MAXU == 0) fetch(0, u);
    float o[W];
#pragma unroll
    for (int e = 0; e < W; ++e) o[e] = p.scale * pv[i][e] * (dv[i][e] - dot);
    stu<W>(dr_ + u * W, o);
  });
}

// ------------------------------------------------------------------------------------------------
// y = drop(alpha * x + pos) + residual
// ------------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(256) void dropadd_kernel(const sdumc_dropadd p) {
  const int nu = p.width / W;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)p.samples * p.rows * nu;
  if (idx >= total) return;
  const int64_t vrow = idx / nu;
  const int u = (int)(idx - vrow * nu);
  const size_t off = (size_t)vrow * p.width + (size_t)u * W;
  float v[W];
  ldu<W>(v, p.x + off);
#pragma unroll
  for (int e = 0; e < W; ++e) v[e] *= p.alpha;
  if (p.pos_table) {
    const int t = (int)(vrow / p.rows);
    const int pos = p.pos_src[(size_t)vrow * p.width] != 0.f ? t + 1 : 0;
    float pe[W];
    ldu<W>(pe, p.pos_table + (size_t)pos * p.width + (size_t)u * W);
#pragma unroll
    for (int e = 0; e < W; ++e) v[e] += pe[e];
  }
  if (p.drop.enabled) {
    const DropRT dr = drop_resolve(p.drop);
    float m[W];
    drop_unit<W>(m, dr, (uint32_t)vrow, u);
#pragma unroll
    for (int e = 0; e < W; ++e) v[e] *= m[e];
  }
  if (p.residual) {
    float r[W];
    ldu<W>(r, p.residual + off);
#pragma unroll
    for (int e = 0; e < W; ++e) v[e] += r[e];
  }
  stu<W>(p.y + off, v);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// row kernels: pick the unit width (16-byte units need width % 4 == 0 and aligned bases) and the register budget
template <typename L4a, typename L4b, typename L4c, typename L1a, typename L1b>
void dispatch_row(bool vec, int width, L4a&& v_small, L4b&& v_big, L4c&& v_any, L1a&& s_small, L1b&& s_any) {
  if (vec) {
    if (width <= 64 * 4 * 2) v_small();
    else if (width <= 64 * 4 * 8) v_big();
    else v_any();
  } else {
    if (width <= 64 * 8) s_small();
    else s_any();
  }
}

// ---- MHA composition helpers -----------------------------------------------------------------------
void gemm_init(sdumc_gemm& g, int layout, int M, int N, int K) {
  g = sdumc_gemm{};
  g.layout = layout;
  g.M = M;
  g.N = N;
  g.K = K;
  g.groups = 1;
  g.splitk = 0;
}

// k[tk] = bias_k, v[tk] = bias_v for every sample (add_bias_kv), then one zero row in both (add_zero_attn)
__global__ __launch_bounds__(256) void mha_extra_rows_kernel(float* k, float* v, const float* bias_k, const float* bias_v, int tk,
                                                             int nb, int nz, int B, int E) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, n = (int64_t)B * E;
  if (i >= n) return;
  const int e = (int)(i % E);
  if (nb) {
    k[(int64_t)tk * n + i] = bias_k[e];
    v[(int64_t)tk * n + i] = bias_v[e];
  }
  if (nz) {
    k[(int64_t)(tk + nb) * n + i] = 0.f;
    v[(int64_t)(tk + nb) * n + i] = 0.f;
  }
}

struct MhaPlan {
  int E, H, dh, B, tq, tk, BH;
  int nb, nz, ts;         // add_bias_kv / add_zero_attn rows (multihead_attention.py:86-104); ts = tk + nb + nz = the source length
  size_t n_q, n_k, n_p;   // floats in a [tq,B,E], a [ts,B,E] and a [BH,tq,ts] tensor
};

int mha_plan(const sdumc_mha& m, MhaPlan* p) {
  if (m.tq <= 0 || m.tk <= 0 || m.batch <= 0 || m.embed <= 0 || m.heads <= 0) return SDUMC_EINVAL;
  if (m.embed % m.heads) return SDUMC_EINVAL;   // multihead_attention.py:20 assert
  p->E = m.embed;
  p->H = m.heads;
  p->dh = m.embed / m.heads;
  p->B = m.batch;
  p->tq = m.tq;
  p->tk = m.tk;
  p->BH = m.batch * m.heads;
  if ((long)p->BH > 65535) return SDUMC_EINVAL;
  if ((m.bias_k != nullptr) != (m.bias_v != nullptr)) return SDUMC_EINVAL;   // :86 asserts both
  p->nb = m.bias_k ? 1 : 0;
  p->nz = m.add_zero_attn ? 1 : 0;
  p->ts = m.tk + p->nb + p->nz;
  p->n_q = (size_t)m.tq * m.batch * m.embed;
  p->n_k = (size_t)p->ts * m.batch * m.embed;
  p->n_p = (size_t)p->BH * m.tq * p->ts;
  return SDUMC_OK;
}

// in-projection GEMMs (multihead_attention.py:133-154): parts first..first+n-1 of (q, k, v), all over T*B rows
void inproj_desc(sdumc_gemm& g, const sdumc_mha& m, const MhaPlan& p, int first, int n, int T) {
  gemm_init(g, SDUMC_NT, T * p.B, p.E, p.E);
  g.groups = n;
  const float* in[3] = {m.query, m.key, m.value};
  float* dst[3] = {m.q, m.k, m.v};
  for (int i = 0; i < n; ++i) {
    const int part = first + i;
    g.A[i] = in[part];
    g.B[i] = m.in_proj_weight + (size_t)part * p.E * p.E;
    g.C[i] = dst[part];
    g.bias[i] = m.in_proj_bias ? m.in_proj_bias + (size_t)part * p.E : nullptr;
  }
  g.lda = g.ldb = g.ldc = p.E;
}

void batched_desc(sdumc_gemm& g, const MhaPlan& p, int layout, int M, int N, int K, const float* A, int lda, long sa,
                  const float* B, int ldb, long sb, float* C, int ldc, long sc) {
  gemm_init(g, layout, M, N, K);
  g.A[0] = A;
  g.B[0] = B;
  g.C[0] = C;
  g.lda = lda;
  g.ldb = ldb;
  g.ldc = ldc;
  g.batch = p.BH;
  g.stride_a = sa;
  g.stride_b = sb;
  g.stride_c = sc;
}

// dW_in / db_in (TN over the T*B rows) for parts first..first+n-1
void dw_in_desc(sdumc_gemm& g, const sdumc_mha& m, const sdumc_mha_grads& gr, const MhaPlan& p, float* const* dqkv,
                int first, int n, int T) {
  gemm_init(g, SDUMC_TN, p.E, p.E, T * p.B);
  g.groups = n;
  const float* in[3] = {m.query, m.key, m.value};
  for (int i = 0; i < n; ++i) {
    const int part = first + i;
    g.A[i] = dqkv[part];
    g.B[i] = in[part];
    g.C[i] = gr.d_in_proj_weight + (size_t)part * p.E * p.E;
    g.colsum_a[i] = gr.d_in_proj_bias ? gr.d_in_proj_bias + (size_t)part * p.E : nullptr;
  }
  g.lda = g.ldb = g.ldc = p.E;
}

size_t align_up(size_t n) { return (n + 63) & ~(size_t)63; }

// bf16-operand products need every extent that is a leading dimension, a stride or a channel count to be a multiple of 4
int mha_bf16(const sdumc_mha& m, const MhaPlan& p) {
  return m.bf16 && (p.E % 4 == 0) && (p.dh % 4 == 0) && (p.tq % 4 == 0) && (p.tk % 4 == 0) && (p.ts % 4 == 0) ? 1 : 0;
}

struct Runner {   // launches descriptors with the shared split-K scratch
  float* ws;
  size_t ws_bytes;
  hipStream_t st;
  int bf16 = 0;
  int rc = SDUMC_OK;
  void run(sdumc_gemm& g) {
    if (rc != SDUMC_OK) return;
    g.workspace = ws;
    g.workspace_bytes = ws_bytes;
    g.bf16 = bf16;
    rc = sdumc_gemm_f32(&g, (void*)st);
  }
};

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int sdumc_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                   float* rstd, int64_t rows, int32_t width, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows < 0 || width <= 0) return SDUMC_EINVAL;
  if (rows == 0) return SDUMC_OK;
  hipStream_t st = as_stream(stream);
  const bool vec = (width & 3) == 0 && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta);
  const dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
#define LN_FWD(W, U) hipLaunchKernelGGL((layernorm_fwd_kernel<W, U>), grid, blk, 0, st, x, gamma, beta, y, mean, rstd, rows, width, eps)
  dispatch_row(vec, width, [&] { LN_FWD(4, 2); }, [&] { LN_FWD(4, 8); }, [&] { LN_FWD(4, 0); }, [&] { LN_FWD(1, 8); },
               [&] { LN_FWD(1, 0); });
#undef LN_FWD
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

extern "C" size_t sdumc_layernorm_bwd_workspace_bytes(int64_t rows, int32_t width) {
  if (rows <= 0 || width <= 0) return 0;
  int rpc;
  const int nchunk = ln_chunks(rows, &rpc);
  return (size_t)nchunk * 2 * width * sizeof(float);
}

extern "C" int sdumc_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                   const float* rstd, float* dx, float* dgamma, float* dbeta, const float* dx_add,
                                   int64_t rows, int32_t width, float* workspace, size_t workspace_bytes, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || rows <= 0 || width <= 0) return SDUMC_EINVAL;
  hipStream_t st = as_stream(stream);
  if (dx) {
    const bool vec = (width & 3) == 0 && aligned16(x) && aligned16(dy) && aligned16(gamma) && aligned16(dx) && aligned16(dx_add);
    const dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
#define LN_BWD(W, U) hipLaunchKernelGGL((layernorm_bwd_dx_kernel<W, U>), grid, blk, 0, st, dy, x, gamma, mean, rstd, dx, dx_add, rows, width)
    dispatch_row(vec, width, [&] { LN_BWD(4, 2); }, [&] { LN_BWD(4, 8); }, [&] { LN_BWD(4, 0); }, [&] { LN_BWD(1, 8); },
                 [&] { LN_BWD(1, 0); });
#undef LN_BWD
    SDUMC_CHECK_LAUNCH();
  }
  if (dgamma || dbeta) {
    if (!dgamma || !dbeta || !workspace) return SDUMC_EINVAL;
    if (workspace_bytes < sdumc_layernorm_bwd_workspace_bytes(rows, width)) return SDUMC_ENOMEM;
    int rpc;
    const int nchunk = ln_chunks(rows, &rpc);
    hipLaunchKernelGGL(layernorm_bwd_gb_stage1, dim3((width + 255) / 256, nchunk), dim3(256), 0, st, dy, x, mean, rstd,
                       rows, width, rpc, workspace);
    SDUMC_CHECK_LAUNCH();
    hipLaunchKernelGGL(layernorm_bwd_gb_stage2, dim3((width + 255) / 256), dim3(256), 0, st, workspace, nchunk, width,
                       dgamma, dbeta);
    SDUMC_CHECK_LAUNCH();
  }
  return SDUMC_OK;
}

static int softmax_check(const sdumc_softmax* s) {
  if (!s || !s->scores || s->batch <= 0 || s->heads <= 0 || s->tq <= 0 || s->tk <= 0) return SDUMC_EINVAL;
  if (s->drop.enabled && (!s->probs_drop || (int)s->drop.width != s->tk || (int)s->drop.rows != s->tq)) return SDUMC_EINVAL;
  if ((int64_t)s->batch * s->heads * s->tq > 0x7fffffffLL) return SDUMC_EINVAL;
  return SDUMC_OK;
}

extern "C" int sdumc_softmax_fwd(const sdumc_softmax* s, void* stream) {
  if (int rc = softmax_check(s)) return rc;
  hipStream_t st = as_stream(stream);
  const bool vec = (s->tk & 3) == 0 && aligned16(s->scores) && (!s->mask || aligned16(s->mask)) &&
                   (!s->weights || aligned16(s->weights)) && (!s->drop.enabled || aligned16(s->probs_drop));
  const int64_t nrow = (int64_t)s->batch * s->tq;
  const dim3 grid((unsigned)((nrow + 3) / 4)), blk(256);
#define SM_FWD(W, U) hipLaunchKernelGGL((softmax_fwd_kernel<W, U>), grid, blk, 0, st, *s)
  dispatch_row(vec, s->tk, [&] { SM_FWD(4, 2); }, [&] { SM_FWD(4, 8); }, [&] { SM_FWD(4, 0); }, [&] { SM_FWD(1, 8); },
               [&] { SM_FWD(1, 0); });
#undef SM_FWD
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

extern "C" int sdumc_softmax_bwd(const sdumc_softmax* s, float* dscores, void* stream) {
  if (int rc = softmax_check(s)) return rc;
  if (!dscores) return SDUMC_EINVAL;
  hipStream_t st = as_stream(stream);
  const bool vec = (s->tk & 3) == 0 && aligned16(s->scores) && aligned16(dscores);
  const int64_t nrow = (int64_t)s->batch * s->heads * s->tq;
  const dim3 grid((unsigned)((nrow + 3) / 4)), blk(256);
#define SM_BWD(W, U) hipLaunchKernelGGL((softmax_bwd_kernel<W, U>), grid, blk, 0, st, *s, dscores)
  dispatch_row(vec, s->tk, [&] { SM_BWD(4, 2); }, [&] { SM_BWD(4, 8); }, [&] { SM_BWD(4, 0); }, [&] { SM_BWD(1, 8); },
               [&] { SM_BWD(1, 0); });
#undef SM_BWD
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

extern "C" int sdumc_drop_add(const sdumc_dropadd* d, void* stream) {
  if (!d || !d->x || !d->y || d->samples < 0 || d->rows <= 0 || d->width <= 0) return SDUMC_EINVAL;
  if (d->pos_table && !d->pos_src) return SDUMC_EINVAL;
  if (d->drop.enabled && ((int)d->drop.width != d->width || (int)d->drop.rows != d->rows)) return SDUMC_EINVAL;
  if (d->samples == 0) return SDUMC_OK;
  hipStream_t st = as_stream(stream);
  const bool vec = (d->width & 3) == 0 && aligned16(d->x) && aligned16(d->y) && (!d->residual || aligned16(d->residual)) &&
                   (!d->pos_table || aligned16(d->pos_table));
  const int64_t total = (int64_t)d->samples * d->rows * (vec ? d->width / 4 : d->width);
  const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
  if (vec) hipLaunchKernelGGL(dropadd_kernel<4>, grid, blk, 0, st, *d);
  else hipLaunchKernelGGL(dropadd_kernel<1>, grid, blk, 0, st, *d);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

extern "C" size_t sdumc_mha_workspace_bytes(const sdumc_mha* mp, int32_t backward) {
  MhaPlan p;
  if (!mp || mha_plan(*mp, &p) != SDUMC_OK) return 0;
  const sdumc_mha& m = *mp;
  sdumc_gemm g;
  size_t gws = 0;
  auto need = [&](const sdumc_gemm& d) { gws = std::max(gws, sdumc_gemm_workspace_bytes(&d)); };
  // forward: in-projections + out-projection (the batched products never split K)
  inproj_desc(g, m, p, 0, 3, std::max(p.tq, p.tk));
  need(g);
  gemm_init(g, SDUMC_NT, p.tq * p.B, p.E, p.E);
  need(g);
  // add_bias_kv / add_zero_attn with a mask: the mask grows zero columns (:90, :104); it is rebuilt in the workspace
  if (p.ts > p.tk && m.attn_mask) gws = std::max(gws, (size_t)p.tq * p.ts * sizeof(float));
  if (p.nb) gws = std::max(gws, sdumc_colsum_workspace_bytes(p.B, p.E));
  if (!backward) return align_up(gws);
  sdumc_mha_grads gr{};
  float* none[3] = {nullptr, nullptr, nullptr};
  dw_in_desc(g, m, gr, p, none, 0, 3, std::max(p.tq, p.tk));
  for (int i = 0; i < 3; ++i) g.colsum_a[i] = reinterpret_cast<float*>(16);   // worst case: with bias gradients
  need(g);
  gemm_init(g, SDUMC_TN, p.E, p.E, p.tq * p.B);
  g.colsum_a[0] = reinterpret_cast<float*>(16);
  need(g);
  gemm_init(g, SDUMC_NN, std::max(p.tq, p.tk) * p.B, p.E, p.E);
  need(g);
  // dctx + dq [tq,B,E], dk + dv [tk,B,E], dP [BH,tq,tk]
  return align_up(gws) + (align_up(2 * p.n_q) + align_up(2 * p.n_k) + align_up(p.n_p)) * sizeof(float);
}

extern "C" int sdumc_mha_forward(const sdumc_mha* mp, void* stream) {
  MhaPlan p;
  if (!mp) return SDUMC_EINVAL;
  if (int rc = mha_plan(*mp, &p)) return rc;
  const sdumc_mha& m = *mp;
  if (!m.query || !m.key || !m.value || !m.in_proj_weight || !m.out_proj_weight || !m.out || !m.q || !m.k || !m.v ||
      !m.probs || !m.ctx)
    return SDUMC_EINVAL;
  if (m.attn_drop.enabled && !m.probs_drop) return SDUMC_EINVAL;
  const size_t need = sdumc_mha_workspace_bytes(mp, 0);
  if (need && (!m.workspace || m.workspace_bytes < need)) return SDUMC_ENOMEM;
  Runner r{m.workspace, need, as_stream(stream), mha_bf16(m, p)};
  sdumc_gemm g;
  // q, k, v = in_proj(query | key | value)   (:64-82)
  if (p.tq == p.tk) {
    inproj_desc(g, m, p, 0, 3, p.tq);
    r.run(g);
  } else {
    inproj_desc(g, m, p, 0, 1, p.tq);
    r.run(g);
    inproj_desc(g, m, p, 1, 2, p.tk);
    r.run(g);
  }
  // S[z] = q_z k_z^T for every (sample, head) z   (:103)
  const int ldx = p.B * p.E;
  // add_bias_kv: k, v grow the row (bias_k | bias_v) for every sample (:86-90); add_zero_attn: a zero row (:100-104)
  if (p.ts > p.tk) {
    const int64_t n = (int64_t)p.B * p.E;
    hipLaunchKernelGGL(mha_extra_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, r.st, m.k, m.v, m.bias_k, m.bias_v,
                       p.tk, p.nb, p.nz, p.B, p.E);
    SDUMC_CHECK_LAUNCH();
  }
  batched_desc(g, p, SDUMC_NT, p.tq, p.ts, p.dh, m.q, ldx, p.dh, m.k, ldx, p.dh, m.probs, p.ts, (long)p.tq * p.ts);
  r.run(g);
  if (r.rc != SDUMC_OK) return r.rc;
  const float* mask = m.attn_mask;
  if (mask && p.ts > p.tk) {   // zero columns for the extra rows: rebuilt in the (now idle) GEMM scratch
    if (int rc = sdumc_fill(m.workspace, 0.f, (int64_t)p.tq * p.ts, stream)) return rc;
    if (int rc = sdumc_copy2d(m.attn_mask, p.tk, m.workspace, p.ts, p.tq, p.tk, stream)) return rc;
    mask = m.workspace;
  }
  // P = softmax(d_h^-0.5 S + mask), dropout, head mean   (:84, :104-117, :128-130)
  sdumc_softmax s{};
  s.batch = p.B;
  s.heads = p.H;
  s.tq = p.tq;
  s.tk = p.ts;
  s.scale = 1.0f / sqrtf((float)p.dh);
  s.mask = mask;
  s.scores = m.probs;
  s.probs_drop = m.probs_drop;
  s.weights = m.weights;
  s.drop = m.attn_drop;
  if (int rc = sdumc_softmax_fwd(&s, stream)) return rc;
  // ctx_z = P_z v_z   (:119), written straight into the [tq, B, E] layout of :122
  const float* pv = m.attn_drop.enabled ? m.probs_drop : m.probs;
  batched_desc(g, p, SDUMC_NN, p.tq, p.dh, p.ts, pv, p.ts, (long)p.tq * p.ts, m.v, ldx, p.dh, m.ctx, ldx, p.dh);
  r.run(g);
  // out = out_proj(ctx)   (:123)
  gemm_init(g, SDUMC_NT, p.tq * p.B, p.E, p.E);
  g.A[0] = m.ctx;
  g.B[0] = m.out_proj_weight;
  g.C[0] = m.out;
  g.bias[0] = m.out_proj_bias;
  g.lda = g.ldb = g.ldc = p.E;
  r.run(g);
  return r.rc;
}

extern "C" int sdumc_mha_backward(const sdumc_mha* mp, const sdumc_mha_grads* gp, void* stream) {
  MhaPlan p;
  if (!mp || !gp) return SDUMC_EINVAL;
  if (int rc = mha_plan(*mp, &p)) return rc;
  const sdumc_mha& m = *mp;
  const sdumc_mha_grads& gr = *gp;
  if (!gr.dout || !gr.dquery || !gr.dkey || !gr.dvalue || !gr.d_in_proj_weight || !gr.d_out_proj_weight)
    return SDUMC_EINVAL;
  if ((m.in_proj_bias != nullptr) != (gr.d_in_proj_bias != nullptr) ||
      (m.out_proj_bias != nullptr) != (gr.d_out_proj_bias != nullptr))
    return SDUMC_EINVAL;
  const size_t need = sdumc_mha_workspace_bytes(mp, 1);
  if (!m.workspace || m.workspace_bytes < need) return SDUMC_ENOMEM;
  const size_t gws = need - (align_up(2 * p.n_q) + align_up(2 * p.n_k) + align_up(p.n_p)) * sizeof(float);
  float* base = m.workspace + gws / sizeof(float);
  float* dctx = base;
  float* dq = dctx + p.n_q;
  float* dk = base + align_up(2 * p.n_q);
  float* dv = dk + p.n_k;
  float* dP = dk + align_up(2 * p.n_k);
  Runner r{m.workspace, gws, as_stream(stream), mha_bf16(m, p)};
  sdumc_gemm g;
  const int ldx = p.B * p.E;
  const float* pv = m.attn_drop.enabled ? m.probs_drop : m.probs;

  // out_proj: dctx = dout W_o ; dW_o = dout^T ctx ; db_o = colsum(dout)
  gemm_init(g, SDUMC_NN, p.tq * p.B, p.E, p.E);
  g.A[0] = gr.dout;
  g.B[0] = m.out_proj_weight;
  g.C[0] = dctx;
  g.lda = g.ldb = g.ldc = p.E;
  r.run(g);
  gemm_init(g, SDUMC_TN, p.E, p.E, p.tq * p.B);
  g.A[0] = gr.dout;
  g.B[0] = m.ctx;
  g.C[0] = gr.d_out_proj_weight;
  g.colsum_a[0] = gr.d_out_proj_bias;
  g.lda = g.ldb = g.ldc = p.E;
  r.run(g);
  // dP_z = dctx_z v_z^T ; dv_z = P_z^T dctx_z
  batched_desc(g, p, SDUMC_NT, p.tq, p.ts, p.dh, dctx, ldx, p.dh, m.v, ldx, p.dh, dP, p.ts, (long)p.tq * p.ts);
  r.run(g);
  batched_desc(g, p, SDUMC_TN, p.ts, p.dh, p.tq, pv, p.ts, (long)p.tq * p.ts, dctx, ldx, p.dh, dv, ldx, p.dh);
  r.run(g);
  if (r.rc != SDUMC_OK) return r.rc;
  // dS = softmax backward (dropout mask recomputed)
  sdumc_softmax s{};
  s.batch = p.B;
  s.heads = p.H;
  s.tq = p.tq;
  s.tk = p.ts;
  s.scale = 1.0f / sqrtf((float)p.dh);
  s.scores = m.probs;
  s.probs_drop = m.probs_drop;
  s.drop = m.attn_drop;
  if (int rc = sdumc_softmax_bwd(&s, dP, stream)) return rc;
  // dq_z = dS_z k_z ; dk_z = dS_z^T q_z   (the d_h^-0.5 of :84 is inside dS)
  batched_desc(g, p, SDUMC_NN, p.tq, p.dh, p.ts, dP, p.ts, (long)p.tq * p.ts, m.k, ldx, p.dh, dq, ldx, p.dh);
  r.run(g);
  batched_desc(g, p, SDUMC_TN, p.ts, p.dh, p.tq, dP, p.ts, (long)p.tq * p.ts, m.q, ldx, p.dh, dk, ldx, p.dh);
  r.run(g);
  if (p.nb) {   // d bias_k / d bias_v = the gradient of row tk summed over the samples (bias.repeat(1, bsz, 1), :88-89)
    if (!gr.d_bias_k || !gr.d_bias_v) return SDUMC_EINVAL;
    if (r.rc != SDUMC_OK) return r.rc;
    if (gws < sdumc_colsum_workspace_bytes(p.B, p.E)) return SDUMC_ENOMEM;
    if (int rc = sdumc_colsum(dk + (size_t)p.tk * p.B * p.E, p.B, p.E, p.E, gr.d_bias_k, 0, m.workspace, stream)) return rc;
    if (int rc = sdumc_colsum(dv + (size_t)p.tk * p.B * p.E, p.B, p.E, p.E, gr.d_bias_v, 0, m.workspace, stream)) return rc;
  }
  // in_proj parameter gradients
  float* dqkv[3] = {dq, dk, dv};
  if (p.tq == p.tk) {
    dw_in_desc(g, m, gr, p, dqkv, 0, 3, p.tq);
    r.run(g);
  } else {
    dw_in_desc(g, m, gr, p, dqkv, 0, 1, p.tq);
    r.run(g);
    dw_in_desc(g, m, gr, p, dqkv, 1, 2, p.tk);
    r.run(g);
  }
  // input gradients: d(input_part) (+)= d(part) W_part ; aliased destinations accumulate
  float* dst[3] = {gr.dquery, gr.dkey, gr.dvalue};
  const int T[3] = {p.tq, p.tk, p.tk};
  for (int part = 0; part < 3; ++part) {
    bool seen = false;
    for (int j = 0; j < part; ++j) seen |= dst[j] == dst[part];
    gemm_init(g, SDUMC_NN, T[part] * p.B, p.E, p.E);
    g.A[0] = dqkv[part];
    g.B[0] = m.in_proj_weight + (size_t)part * p.E * p.E;
    g.C[0] = dst[part];
    g.lda = g.ldb = g.ldc = p.E;
    g.accumulate = seen ? 1 : 0;
    r.run(g);
  }
  return r.rc;
}

// One empty kernel per source file = per gfx950 code object: sdumc_preload_() asks for its attributes, which makes the HIP runtime load
// this file's code object NOW (outside any timed or latency-sensitive region) instead of at the first launch of one of its kernels --
// with deferred loading that first launch stalls the host for tens of milliseconds (seen as a 36-59 ms gap in the middle of an epoch,
// at the first batch whose shape took a fallback path: profiles/README.md, round 6).
__global__ void sdumc_preload_transformer_kernel() {}
extern "C" int sdumc_preload_transformer_(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&sdumc_preload_transformer_kernel)) == hipSuccess ? SDUMC_OK : SDUMC_ELAUNCH;
}
