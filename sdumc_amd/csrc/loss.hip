// loss.hip — the loss reductions of the self-distillation step, value + gradient.
//   MSELoss   toolkit/utils/loss.py:19-33   sum((p-t)^2)/len(p)
//   RMSELoss  toolkit/utils/loss.py:37-51   sqrt(mean((a-b)^2))   (split: ssd, then sqrt -> DP-exact)
//   RnCLoss   toolkit/utils/loss.py:271-315 Rank-N-Contrast; the boolean neg_mask (loss.py:303) is
//             evaluated with the same fp32 operations as the reference, so membership is bit-exact.
// All sums run in a fixed order (no float atomics): results are bitwise reproducible.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red /*[4]*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void mse_kernel(const float* pred, const float* target, int rows, float denom,
                                                  float weight, float* loss_out, float* dpred) {
  __shared__ float red[4];
  float acc = 0.f;
  const float inv = 1.f / denom;
  for (int i = threadIdx.x; i < rows; i += 256) {
    const float d = pred[i] - target[i];
    acc += d * d;
    if (dpred) dpred[i] = weight * 2.f * d * inv;
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) *loss_out = s * inv;
}

constexpr int SSD_CHUNK = 8192;
__global__ __launch_bounds__(256) void ssd_stage1(const float* a, const float* b, int64_t n, float* part) {
  __shared__ float red[4];
  const int64_t i0 = (int64_t)blockIdx.x * SSD_CHUNK, i1 = min(n, i0 + SSD_CHUNK);
  float acc = 0.f;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const float d = a[i] - b[i];
    acc += d * d;
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void ssd_stage2(const float* part, int nchunk, float* out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nchunk; i += 256) acc += part[i];
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) *out = s;
}

__global__ void rmse_bwd_kernel(const float* a, const float* b, int64_t n, const float* ssd, float inv_numel,
                                float weight, float* loss_out, float* da, int da_acc, float* db, int db_acc) {
  const float rmse = sqrtf(*ssd * inv_numel);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && loss_out) *loss_out = rmse;
  if (i >= n) return;
  // d sqrt(mean)/da = (a-b) / (numel * rmse)   (0/0 -> NaN exactly like torch's sqrt backward)
  const float g = weight * (a[i] - b[i]) * inv_numel / rmse;
  if (da) da[i] = da_acc ? da[i] + g : g;
  if (db) db[i] = db_acc ? db[i] - g : -g;
}

// ---------------------------------------------------------------------------------------------
// Rank-N-Contrast.  Workspace layout (floats): dist[n*n] e[n*n] ldiff[n*n] invD[n*n] G[n*n]
// rowmax[n] rowloss[n]
// ---------------------------------------------------------------------------------------------
struct RncWs {
  float *dist, *e, *ldiff, *invD, *G, *rowmax, *rowloss;
};
__host__ __device__ inline RncWs rnc_ws(float* w, int n) {
  RncWs r;
  const size_t nn = (size_t)n * n;
  r.dist = w;
  r.e = w + nn;
  r.ldiff = w + 2 * nn;
  r.invD = w + 3 * nn;
  r.G = w + 4 * nn;
  r.rowmax = w + 5 * nn;
  r.rowloss = r.rowmax + n;
  return r;
}

__device__ __forceinline__ void rnc_loss_body(int n, RncWs w, float* loss_out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += w.rowloss[i];
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) *loss_out = -s / ((float)n * (float)(n - 1));
}
__global__ __launch_bounds__(256) void rnc_loss_kernel(int n, RncWs w, float* loss_out) { rnc_loss_body(n, w, loss_out); }

// One workgroup per anchor i, everything that depends on row i only: distances, label differences, exp,
// the masked denominators D_ik, the row's loss and G_i* = dLoss/dlogit_i*.  labels: y[j] = labels[j % label_mod]
// when label_mod > 0 (the reference's labels.repeat(2, 1), loss.py:283, without materialising it).
__device__ __forceinline__ void rnc_row_body(const float* f, const float* labels, int label_mod, int n, int dim, float inv_t,
                                             RncWs w, int want_grad, const int i, float* sm /* fi[dim] | ld[n] | ee[n] | dd[n] | thr[n] | iD[n] */) {
  __shared__ float red[4];
  float* fi = sm;
  float* ld = fi + ((dim + 3) & ~3);
  float* ee = ld + n;
  float* dd = ee + n;
  float* thr = dd + n;
  float* iD = thr + n;
  for (int c = threadIdx.x; c < dim; c += 256) fi[c] = f[(size_t)i * dim + c];
  __syncthreads();
  const float yi = labels[label_mod > 0 ? i % label_mod : i];
  float mx = -INFINITY;
  const bool vec = (dim & 3) == 0 && (reinterpret_cast<uintptr_t>(f) & 15) == 0;
  for (int j = threadIdx.x; j < n; j += 256) {
    float s = 0.f;
    if (vec) {
      // 16-byte loads, 8 in flight: the scalar loop below is one dependent L2 round trip per channel (64 of them at the
      // model's width: ~20 of this kernel's 26 us).  Same summation order as the scalar loop.
      const f32x4* fj = reinterpret_cast<const f32x4*>(f + (size_t)j * dim);
      for (int c0 = 0; c0 < dim / 4; c0 += 8) {
        f32x4 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = c0 + u < dim / 4 ? fj[c0 + u] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (c0 + u < dim / 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float d = fi[4 * (c0 + u) + e] - r[u][e];
              s += d * d;
            }
          }
      }
    } else {
      for (int c = 0; c < dim; ++c) {
        const float d = fi[c] - f[(size_t)j * dim + c];
        s += d * d;
      }
    }
    const float dist = sqrtf(s);
    dd[j] = dist;
    w.dist[(size_t)i * n + j] = dist;
    const float l = fabsf(yi - labels[label_mod > 0 ? j % label_mod : j]);
    ld[j] = l;
    thr[j] = __fsub_rn(l, 0.0001f);
    mx = fmaxf(mx, -dist * inv_t);
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  for (int j = threadIdx.x; j < n; j += 256) ee[j] = expf(-dd[j] * inv_t - mx);
  __syncthreads();
  float acc = 0.f;
  for (int k = threadIdx.x; k < n; k += 256) {
    if (k == i) {
      iD[k] = 0.f;
      continue;
    }
    const float t = thr[k];
    float dsum = 0.f;
#pragma unroll 8
    for (int j = 0; j < n; ++j)      // unrolled: the LDS reads of 8 iterations are in flight together (same summation order)
      dsum += (j != i && ld[j] >= t) ? ee[j] : 0.f;
    iD[k] = 1.f / dsum;
    acc += (-dd[k] * inv_t - mx) - logf(dsum);
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) w.rowloss[i] = s;
  if (!want_grad) return;
  __syncthreads();
  const float c = 1.f / ((float)n * (float)(n - 1));
  for (int j = threadIdx.x; j < n; j += 256) {
    float g = 0.f;
    if (j != i) {
      const float lj = ld[j];
      float sum = 0.f;
#pragma unroll 8
      for (int k = 0; k < n; ++k)
        sum += (k != i && lj >= thr[k]) ? iD[k] : 0.f;
      g = -c * (1.f - ee[j] * sum);
    }
    w.G[(size_t)i * n + j] = g;
  }
}
__global__ __launch_bounds__(256) void rnc_row_kernel(const float* f, const float* labels, int label_mod, int n,
                                                      int dim, float inv_t, RncWs w, int want_grad) {
  extern __shared__ float sm[];
  rnc_row_body(f, labels, label_mod, n, dim, inv_t, w, want_grad, blockIdx.x, sm);
}

// ---- O(n^2 log n) formulation (n <= 2048) ---------------------------------------------------------------------
// For anchor i the reference's neg_mask row for k, {j : |y_i - y_j| >= |y_i - y_k| - 1e-4} (loss.py:303), is the
// complement of a window around i in the order of the labels: with the labels sorted ONCE (rnc_sort_kernel) it is a
// prefix [0, L) plus a suffix [U, n) of the sorted order, and the gradient's {k : |y_i - y_k| - 1e-4 <= |y_i - y_j|}
// is a window [A, B] around i.  Prefix / suffix sums of exp(logit) and window sums of 1/D then replace the two
// O(n^2) loops per anchor of rnc_row_kernel; L, U, A, B come from binary searches that evaluate the SAME fp32
// predicate as the reference on the probed elements (fabsf(y_i - y_j) >= fsub(|y_i - y_k|, 1e-4f): monotone on either
// side of i), so set membership stays bit-exact.  All partial sums are sums of non-negative terms taken outward from
// the ends / from i (no prefix differences -> no cancellation), in a fixed order.
// At n = 128 (one GPU, B = 64) this is on par with the direct loops; at n = 1024 (B_global = 512 under data
// parallelism, where EVERY rank evaluates the full loss) it is 0.67 ms -> tens of microseconds per step.
// sorted order of the labels by counting: rank[j] = #{k : y_k < y_j or (y_k == y_j and k < j)} (ties by index: a
// total, deterministic order), perm = rank^-1.  n threads x n comparisons from LDS: a few microseconds at n = 1024.
__global__ __launch_bounds__(256) void rnc_sort_kernel(const float* labels, int label_mod, int n, int* perm, int* rank) {
  extern __shared__ float ysm[];   // [n]
  for (int t = threadIdx.x; t < n; t += 256) ysm[t] = labels[label_mod > 0 ? t % label_mod : t];
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float yj = ysm[j];
  int r = 0;
#pragma unroll 8
  for (int k = 0; k < n; ++k) {
    const float yk = ysm[k];
    r += (yk < yj || (yk == yj && k < j)) ? 1 : 0;
  }
  rank[j] = r;
  perm[r] = j;
}

// inclusive scan of a[0..n) in LDS with 256 threads (reverse: suffix sums), fixed order; tmp: >= 4 floats
__device__ __forceinline__ void block_scan(float* a, int n, bool reverse, float* tmp) {
  const int per = (n + 255) / 256;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lo = t * per, hi = min(n, lo + per);
  float s = 0.f;
  for (int p = lo; p < hi; ++p) s += a[reverse ? n - 1 - p : p];
  // exclusive scan of the 256 per-thread partials: shuffles inside a wave, then the four wave totals
  float incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  __syncthreads();                 // tmp may still be read by a previous scan
  if (lane == 63) tmp[wave] = incl;
  __syncthreads();
  float run = incl - s;
  for (int w2 = 0; w2 < wave; ++w2) run += tmp[w2];
  for (int p = lo; p < hi; ++p) {
    const int q = reverse ? n - 1 - p : p;
    run += a[q];
    a[q] = run;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void rnc_row_sorted_kernel(const float* f, const float* labels, int label_mod, int n, int dim,
                                                             float inv_t, RncWs w, const int* perm, const int* rank,
                                                             int want_grad) {
  extern __shared__ float sm[];   // fi[dim4] | ys[n] | dd[n] | ee[n] | pre[n] | suf[n] | iD[n] | oL[n] | oR[n] | tmp[8]
  __shared__ float red[4];
  const int dim4 = (dim + 3) & ~3;
  float* fi = sm;
  float* ys = fi + dim4;
  float* dd = ys + n;
  float* ee = dd + n;
  float* pre = ee + n;
  float* suf = pre + n;
  float* iD = suf + n;
  float* oL = iD + n;
  float* oR = oL + n;
  float* tmp = oR + n;
  const int i = blockIdx.x, tid = threadIdx.x;
  const int pos_i = rank[i];
  for (int c = tid; c < dim; c += 256) fi[c] = f[(size_t)i * dim + c];
  for (int r = tid; r < n; r += 256) {
    const int j = perm[r];
    ys[r] = labels[label_mod > 0 ? j % label_mod : j];
  }
  __syncthreads();
  const float yi = ys[pos_i];
  const bool vec = (dim & 3) == 0 && (reinterpret_cast<uintptr_t>(f) & 15) == 0;
  float mx = -INFINITY;
  for (int r = tid; r < n; r += 256) {
    const int j = perm[r];
    float s = 0.f;
    if (vec) {
      const f32x4* fj = reinterpret_cast<const f32x4*>(f + (size_t)j * dim);
      for (int c0 = 0; c0 < dim / 4; c0 += 8) {
        f32x4 rr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) rr[u] = c0 + u < dim / 4 ? fj[c0 + u] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (c0 + u < dim / 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float d = fi[4 * (c0 + u) + e] - rr[u][e];
              s += d * d;
            }
          }
      }
    } else {
      for (int c = 0; c < dim; ++c) {
        const float d = fi[c] - f[(size_t)j * dim + c];
        s += d * d;
      }
    }
    const float dist = sqrtf(s);
    dd[r] = dist;
    w.dist[(size_t)i * n + j] = dist;
    mx = fmaxf(mx, -dist * inv_t);
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  for (int r = tid; r < n; r += 256) {
    const float e = r == pos_i ? 0.f : expf(-dd[r] * inv_t - mx);   // the anchor itself is excluded (diagonal removal, loss.py:294-296)
    ee[r] = e;
    pre[r] = e;
    suf[r] = e;
  }
  __syncthreads();
  block_scan(pre, n, false, tmp);   // pre[r] = sum_{p <= r} ee[p]
  block_scan(suf, n, true, tmp);    // suf[r] = sum_{p >= r} ee[p]
  float acc = 0.f;
  for (int r = tid; r < n; r += 256) {
    float inv = 0.f;
    if (r != pos_i) {
      const float t = __fsub_rn(fabsf(yi - ys[r]), 0.0001f);
      // L = number of p in [0, pos_i] with fabsf(yi - ys[p]) >= t (true ... true false ... false)
      int lo = 0, hi = pos_i + 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (fabsf(yi - ys[mid]) >= t) lo = mid + 1; else hi = mid;
      }
      const int L = lo;
      // U = first p in [pos_i, n) with fabsf(yi - ys[p]) >= t (false ... false true ... true)
      lo = pos_i;
      hi = n;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (fabsf(yi - ys[mid]) >= t) hi = mid; else lo = mid + 1;
      }
      const int U = lo;
      const float dsum = (L > 0 ? pre[L - 1] : 0.f) + (U < n ? suf[U] : 0.f);
      inv = 1.f / dsum;
      acc += (-dd[r] * inv_t - mx) - logf(dsum);
    }
    iD[r] = inv;
  }
  const float sl = block_sum_256(acc, red);
  if (tid == 0) w.rowloss[i] = sl;
  if (!want_grad) return;
  __syncthreads();
  for (int r = tid; r < n; r += 256) {
    oL[r] = r <= pos_i ? iD[r] : 0.f;
    oR[r] = r >= pos_i ? iD[r] : 0.f;
  }
  __syncthreads();
  block_scan(oL, n, true, tmp);     // oL[p] = sum_{q = p .. pos_i} iD[q]   (p <= pos_i)
  block_scan(oR, n, false, tmp);    // oR[p] = sum_{q = pos_i .. p} iD[q]   (p >= pos_i)
  const float c = 1.f / ((float)n * (float)(n - 1));
  for (int r = tid; r < n; r += 256) {
    const int j = perm[r];
    float g = 0.f;
    if (r != pos_i) {
      const float lj = fabsf(yi - ys[r]);
      // A = first p in [0, pos_i] with thr(p) <= lj  (thr non-increasing towards pos_i: false ... false true ... true)
      int lo = 0, hi = pos_i;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (lj >= __fsub_rn(fabsf(yi - ys[mid]), 0.0001f)) hi = mid; else lo = mid + 1;
      }
      const int A = lo;
      // B = last p in [pos_i, n) with thr(p) <= lj  (true ... true false ... false)
      lo = pos_i;
      hi = n - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (lj >= __fsub_rn(fabsf(yi - ys[mid]), 0.0001f)) lo = mid; else hi = mid - 1;
      }
      const int B = lo;
      g = -c * (1.f - ee[r] * (oL[A] + oR[B]));   // iD[pos_i] = 0: counting it on both sides is harmless
    }
    w.G[(size_t)i * n + j] = g;
  }
}

// df_i = -(1/t) sum_j (G_ij + G_ji) (f_i - f_j) / dist_ij ; one workgroup per local row
__device__ __forceinline__ void rnc_dfeat_body(const float* f, int n, int dim, float inv_t, float weight, int row0, RncWs w, float* df,
                                               const int bx, float* sm /* coef[n], then red[4][64] */) {
  float* coef = sm;
  float* red = sm + n;
  const int i = row0 + bx;
  for (int j = threadIdx.x; j < n; j += 256) {
    const float d = w.dist[(size_t)i * n + j];
    coef[j] = (j != i && d > 0.f) ? (w.G[(size_t)i * n + j] + w.G[(size_t)j * n + i]) / d : 0.f;
  }
  __syncthreads();
  const int part = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int c0 = 0; c0 < dim; c0 += 64) {
    const int c = c0 + lane;
    float acc = 0.f;
    if (c < dim) {
      const float fic = f[(size_t)i * dim + c];
      int j = part;
      for (; j + 28 < n; j += 32) {   // 8 rows of loads in flight (same summation order as the plain loop)
        float fj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) fj[u] = f[(size_t)(j + 4 * u) * dim + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += coef[j + 4 * u] * (fic - fj[u]);
      }
      for (; j < n; j += 4) acc += coef[j] * (fic - f[(size_t)j * dim + c]);
    }
    red[part * 64 + lane] = acc;
    __syncthreads();
    if (part == 0 && c < dim)
      df[(size_t)bx * dim + c] = -weight * inv_t * (red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane]);
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void rnc_dfeat_kernel(const float* f, int n, int dim, float inv_t, float weight,
                                                        int row0, RncWs w, float* df) {
  extern __shared__ float sm[];
  rnc_dfeat_body(f, n, dim, inv_t, weight, row0, w, df, blockIdx.x, sm);
}

__global__ void rnc_mask_kernel(const float* y, int n, uint8_t* mask) {
  const int64_t total = (int64_t)n * (n - 1) * (n - 1);
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int i = (int)(e / ((int64_t)(n - 1) * (n - 1)));
  const int64_t r = e - (int64_t)i * (n - 1) * (n - 1);
  int k = (int)(r / (n - 1)), j = (int)(r - (int64_t)k * (n - 1));
  k += (k >= i);
  j += (j >= i);
  const float dij = fabsf(y[i] - y[j]), dik = fabsf(y[i] - y[k]);
  mask[e] = dij >= __fsub_rn(dik, 0.0001f) ? 1 : 0;
}


// ---- the five "small" distillation terms of main :137-148 in two launches -------------------------------
// pass 1: per-block partial sums of squared differences of the three RMSE pairs (+ both MSE terms, block 0)
// pass 2: every block re-reduces the (few) partials in a fixed order, then writes the gradients of its chunk
struct DistillArgs {
  int B;            // local samples per stream
  float denom;      // B_global
  const float *vals, *labels, *th, *ct, *z;   // network outputs [2B, ...]
  float w[5];       // full_mse, missing_mse, text_feat, text_query_feat, features
  const float* ssd_global;                      // [3] or nullptr
  float *d_vals, *d_th, *d_ct, *d_z;
  float* losses;    // [8]: writes 1..5
  float* part;      // [3][nblk]
  int nblk[3];
};
constexpr int DCH = 4096;   // elements per block
__device__ __forceinline__ void distill_pair(const DistillArgs& a, int p, const float*& s1, const float*& s0, float*& g,
                                             int64_t& n) {
  const int64_t per = p == 0 ? SDUMC_D : (p == 1 ? SDUMC_NQ * SDUMC_H : SDUMC_H);
  n = (int64_t)a.B * per;
  const float* base = p == 0 ? a.th : (p == 1 ? a.ct : a.z);
  s0 = base;
  s1 = base + n;
  g = p == 0 ? a.d_th : (p == 1 ? a.d_ct : a.d_z);
}
__device__ __forceinline__ void distill_partials_body(const DistillArgs a, const int bx, const int p) {
  __shared__ float red[4];
  if (bx >= a.nblk[p]) return;
  const float *s1, *s0;
  float* g;
  int64_t n;
  distill_pair(a, p, s1, s0, g, n);
  const int64_t i0 = (int64_t)bx * DCH, i1 = min(n, i0 + DCH);
  float acc = 0.f;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const float d = s1[i] - s0[i];
    acc += d * d;
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) a.part[p * a.nblk[1] + bx] = s;
}
__global__ __launch_bounds__(256) void distill_partials_kernel(const DistillArgs a) { distill_partials_body(a, blockIdx.x, blockIdx.y); }
__device__ __forceinline__ void distill_apply_body(const DistillArgs a, const int bx, const int p) {
  __shared__ float red[4];
  if (p == 3) {   // MSELoss on both streams (loss.py:19-33): value + gradient
    if (bx > 1) return;
    const int sidx = bx;
    const float inv = 1.f / a.denom;
    float acc = 0.f;
    for (int i = threadIdx.x; i < a.B; i += 256) {
      const float d = a.vals[sidx * a.B + i] - a.labels[i];
      acc += d * d;
      a.d_vals[sidx * a.B + i] = a.w[sidx] * 2.f * d * inv;
    }
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) a.losses[1 + sidx] = s * inv;
    return;
  }
  if (bx >= a.nblk[p]) return;
  const float *s1, *s0;
  float* g;
  int64_t n;
  distill_pair(a, p, s1, s0, g, n);
  float ssd;
  if (a.ssd_global) {
    ssd = a.ssd_global[p];
  } else {   // ordered re-reduction of the partials (identical in every block)
    float acc = 0.f;
    for (int i = threadIdx.x; i < a.nblk[p]; i += 256) acc += a.part[p * a.nblk[1] + i];
    ssd = block_sum_256(acc, red);
  }
  const float per = p == 0 ? SDUMC_D : (p == 1 ? SDUMC_NQ * SDUMC_H : SDUMC_H);
  const float inv_numel = 1.f / (a.denom * per);
  const float rmse = sqrtf(ssd * inv_numel);
  if (bx == 0 && threadIdx.x == 0) a.losses[3 + p] = rmse;
  const float k = a.w[2 + p] * inv_numel / rmse;   // 0/0 -> NaN exactly like torch's sqrt backward
  const int64_t i0 = (int64_t)bx * DCH, i1 = min(n, i0 + DCH);
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const float gr = k * (s1[i] - s0[i]);
    g[n + i] = gr;                        // stream 1 (student)
    g[i] = p == 2 ? -gr : 0.f;           // stream 0: detached for text_feat / text_query_feat (main :148), not for features
  }
}
__global__ __launch_bounds__(256) void distill_apply_kernel(const DistillArgs a) { distill_apply_body(a, blockIdx.x, blockIdx.y); }

// ---- the six loss launches of a single-GPU step as TWO (+ the one-thread total): the distillation terms and Rank-N-Contrast
// are independent, so their first passes share a launch (workgroups [0, n): one RnC anchor each; the rest: partial sums of
// squares) and so do their second passes (RnC feature gradients | distillation values + gradients | the RnC mean).  No
// atomics, no change in any summation order: same bits as the separate launches.
struct RncArgs {
  const float* f;
  const float* labels;
  int label_mod, n, dim;
  float inv_t, weight;
  RncWs w;
  float* loss_out;
  float* df;
};
__global__ __launch_bounds__(256) void loss_stage1_kernel(const DistillArgs a, const RncArgs r, const int mx) {
  extern __shared__ float sm[];
  const int bid = blockIdx.x;
  if (bid < r.n) {
    rnc_row_body(r.f, r.labels, r.label_mod, r.n, r.dim, r.inv_t, r.w, 1, bid, sm);
  } else {
    const int q = bid - r.n;
    distill_partials_body(a, q % mx, q / mx);
  }
}
__global__ __launch_bounds__(256) void loss_stage2_kernel(const DistillArgs a, const RncArgs r, const int amx, float* hyper,
                                                          const double beta1, const double beta2) {
  extern __shared__ float sm[];
  const int bid = blockIdx.x;
  if (bid < r.n) {
    rnc_dfeat_body(r.f, r.n, r.dim, r.inv_t, r.weight, 0, r.w, r.df, bid, sm);
  } else if (bid < r.n + 4 * amx) {
    const int q = bid - r.n;
    distill_apply_body(a, q % amx, q / amx);
  } else {
    rnc_loss_body(r.n, r.w, r.loss_out);
    if (hyper && threadIdx.x == 0) {   // Adam bias correction of this step (adam.hip's adam_hyper_kernel, same double arithmetic)
      const double t = (double)hyper[1] + 1.0;
      hyper[1] = (float)t;
      hyper[2] = (float)((double)hyper[0] / (1.0 - pow(beta1, t)));
      hyper[3] = (float)sqrt(1.0 - pow(beta2, t));
    }
  }
}
}  // namespace

namespace {
// The whole data-parallel exchange record of one rank in ONE launch: blockIdx.y = 0..2 -> partial sums of squares of the
// three RMSE pairs (DCH elements per block), blockIdx.y = 3 -> copies of the RnC features and the labels.  The last
// block to finish (device-scope counter) adds the partials of each pair in index order -- the same bits whatever the
// scheduling -- writes the three sums behind the labels and re-arms the counter.
struct DpRecordArgs {
  int B, rd;
  const float *th, *ct, *z, *rnc, *labels;
  float* rec;        // [2*B*rd | B | 3]
  float* part;       // [3][nblk_max]
  unsigned* counter; // zero before the first launch; left zero by every launch
  int nblk[3], nblk_max, ncopy;
};
__global__ __launch_bounds__(256) void dp_record_kernel(const DpRecordArgs a) {
  __shared__ float red[4];
  __shared__ bool last;
  const int p = blockIdx.y;
  const int nf = 2 * a.B * a.rd;
  if (p == 3) {
    if ((int)blockIdx.x < a.ncopy) {
      const int i0 = blockIdx.x * DCH, i1 = min(nf + a.B, i0 + DCH);
      for (int i = i0 + threadIdx.x; i < i1; i += 256) a.rec[i] = i < nf ? a.rnc[i] : a.labels[i - nf];
    }
  } else if ((int)blockIdx.x < a.nblk[p]) {
    const int64_t per = p == 0 ? SDUMC_D : (p == 1 ? SDUMC_NQ * SDUMC_H : SDUMC_H);
    const int64_t n = (int64_t)a.B * per;
    const float* s0 = p == 0 ? a.th : (p == 1 ? a.ct : a.z);
    const float* s1 = s0 + n;
    const int64_t i0 = (int64_t)blockIdx.x * DCH, i1 = min(n, i0 + DCH);
    float acc = 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
      const float d = s1[i] - s0[i];
      acc += d * d;
    }
    const float sum = block_sum_256(acc, red);
    if (threadIdx.x == 0) a.part[p * a.nblk_max + blockIdx.x] = sum;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(a.counter, 1u) == gridDim.x * gridDim.y - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  for (int q = 0; q < 3; ++q) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < a.nblk[q]; i += 256) acc += __builtin_nontemporal_load(a.part + q * a.nblk_max + i);
    const float sum = block_sum_256(acc, red);
    if (threadIdx.x == 0) a.rec[nf + a.B + q] = sum;
    __syncthreads();
  }
  if (threadIdx.x == 0) *a.counter = 0u;
}
}  // namespace

extern "C" size_t sdumc_dp_record_workspace_bytes(int32_t B) {
  const int64_t nb = ((int64_t)B * SDUMC_NQ * SDUMC_H + DCH - 1) / DCH;
  return (size_t)(3 * nb + 16) * sizeof(float);
}

extern "C" int sdumc_dp_record(int32_t B, int32_t rd, const float* text_hidden, const float* cross_text, const float* fused,
                               const float* rnc, const float* labels, float* record, void* workspace, void* stream) {
  if (B <= 0 || rd <= 0 || !text_hidden || !cross_text || !fused || !rnc || !labels || !record || !workspace)
    return SDUMC_EINVAL;
  DpRecordArgs a;
  a.B = B; a.rd = rd;
  a.th = text_hidden; a.ct = cross_text; a.z = fused; a.rnc = rnc; a.labels = labels;
  a.rec = record;
  const int64_t per[3] = {SDUMC_D, SDUMC_NQ * SDUMC_H, SDUMC_H};
  int mx = 1;
  for (int p = 0; p < 3; ++p) {
    a.nblk[p] = (int)(((int64_t)B * per[p] + DCH - 1) / DCH);
    mx = a.nblk[p] > mx ? a.nblk[p] : mx;
  }
  a.nblk_max = a.nblk[1];
  a.ncopy = (2 * B * rd + B + DCH - 1) / DCH;
  mx = a.ncopy > mx ? a.ncopy : mx;
  a.counter = reinterpret_cast<unsigned*>(workspace);
  a.part = static_cast<float*>(workspace) + 16;
  hipLaunchKernelGGL(dp_record_kernel, dim3(mx, 4), dim3(256), 0, as_stream(stream), a);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

extern "C" int sdumc_mse_fwd_bwd(const float* pred, const float* target, int32_t rows, float denom, float weight,
                                 float* loss_out, float* dpred, void* stream) {
  if (!pred || !target || !loss_out || rows <= 0 || denom <= 0.f) return SDUMC_EINVAL;
  hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(256), 0, as_stream(stream), pred, target, rows, denom, weight, loss_out, dpred);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

extern "C" size_t sdumc_ssd_workspace_bytes(int64_t n) { return (size_t)((n + SSD_CHUNK - 1) / SSD_CHUNK) * sizeof(float); }

extern "C" int sdumc_ssd(const float* a, const float* b, int64_t n, float* ssd_out, float* workspace, void* stream) {
  if (!a || !b || !ssd_out || !workspace || n <= 0) return SDUMC_EINVAL;
  const int nchunk = (int)((n + SSD_CHUNK - 1) / SSD_CHUNK);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(ssd_stage1, dim3(nchunk), dim3(256), 0, st, a, b, n, workspace);
  SDUMC_CHECK_LAUNCH();
  hipLaunchKernelGGL(ssd_stage2, dim3(1), dim3(256), 0, st, workspace, nchunk, ssd_out);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

extern "C" int sdumc_rmse_bwd(const float* a, const float* b, int64_t n_local, const float* ssd_global,
                              double numel_global, float weight, float* loss_out, float* da, int32_t da_accumulate,
                              float* db, int32_t db_accumulate, void* stream) {
  if (!a || !b || !ssd_global || n_local <= 0 || numel_global <= 0) return SDUMC_EINVAL;
  hipLaunchKernelGGL(rmse_bwd_kernel, dim3((unsigned)((n_local + 255) / 256)), dim3(256), 0, as_stream(stream), a, b,
                     n_local, ssd_global, (float)(1.0 / numel_global), weight, loss_out, da, da_accumulate, db,
                     db_accumulate);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

extern "C" size_t sdumc_rnc_workspace_bytes(int32_t n) { return ((size_t)5 * n * n + 2 * (size_t)n) * sizeof(float); }

static int rnc_impl(const float* feats, const float* labels, int label_mod, int32_t n, int32_t dim, float temperature,
                    float weight, int32_t row0, int32_t rows_local, float* loss_out, float* dfeats, float* workspace,
                    void* stream) {
  if (!feats || !labels || !loss_out || !workspace || n < 2 || dim <= 0 || temperature <= 0.f) return SDUMC_EINVAL;
  if (row0 < 0 || rows_local < 0 || row0 + rows_local > n) return SDUMC_EINVAL;
  hipStream_t st = as_stream(stream);
  const RncWs w = rnc_ws(workspace, n);
  const float inv_t = 1.f / temperature;
  const int want_grad = dfeats && rows_local > 0;
  if (n > 256 && n <= 2048) {
    // sorted formulation; perm / rank live in the (otherwise unused) `e` region of the workspace
    int* perm = reinterpret_cast<int*>(w.e);
    int* rank = perm + n;
    hipLaunchKernelGGL(rnc_sort_kernel, dim3((n + 255) / 256), dim3(256), (size_t)n * sizeof(float), st, labels, label_mod, n,
                       perm, rank);
    SDUMC_CHECK_LAUNCH();
    const size_t lds = ((((size_t)dim + 3) & ~(size_t)3) + 8 * (size_t)n + 8) * sizeof(float);
    if (lds > 160 * 1024) return SDUMC_EINVAL;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(rnc_row_sorted_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return SDUMC_ELAUNCH;
    hipLaunchKernelGGL(rnc_row_sorted_kernel, dim3(n), dim3(256), lds, st, feats, labels, label_mod, n, dim, inv_t, w, perm,
                       rank, want_grad);
    SDUMC_CHECK_LAUNCH();
  } else {
    const size_t lds = (((size_t)dim + 3) & ~(size_t)3) * sizeof(float) + 5 * (size_t)n * sizeof(float);
    if (lds > 64 * 1024) return SDUMC_EINVAL;
    hipLaunchKernelGGL(rnc_row_kernel, dim3(n), dim3(256), lds, st, feats, labels, label_mod, n, dim, inv_t, w, want_grad);
    SDUMC_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(rnc_loss_kernel, dim3(1), dim3(256), 0, st, n, w, loss_out);
  SDUMC_CHECK_LAUNCH();
  if (want_grad) {
    hipLaunchKernelGGL(rnc_dfeat_kernel, dim3(rows_local), dim3(256), (n + 256) * sizeof(float), st, feats, n, dim,
                       inv_t, weight, row0, w, dfeats);
    SDUMC_CHECK_LAUNCH();
  }
  return SDUMC_OK;
}

extern "C" int sdumc_rnc_fwd_bwd(const float* feats, const float* labels, int32_t n, int32_t dim, float temperature,
                                 float weight, int32_t row0, int32_t rows_local, float* loss_out, float* dfeats,
                                 float* workspace, void* stream) {
  return rnc_impl(feats, labels, 0, n, dim, temperature, weight, row0, rows_local, loss_out, dfeats, workspace, stream);
}

// labels given once ([n/2]) and read as labels[j % (n/2)]: the reference's labels.repeat(2, 1) (loss.py:283)
extern "C" int sdumc_rnc_fwd_bwd_rep(const float* feats, const float* labels_half, int32_t n, int32_t dim,
                                     float temperature, float weight, int32_t row0, int32_t rows_local, float* loss_out,
                                     float* dfeats, float* workspace, void* stream) {
  if (n & 1) return SDUMC_EINVAL;
  return rnc_impl(feats, labels_half, n / 2, n, dim, temperature, weight, row0, rows_local, loss_out, dfeats, workspace,
                  stream);
}

extern "C" size_t sdumc_distill_workspace_bytes(int32_t B) {
  const int64_t nb = ((int64_t)B * SDUMC_NQ * SDUMC_H + DCH - 1) / DCH;
  return (size_t)(3 * nb + 64) * sizeof(float);
}

// MSE x2 + RMSE x3 of main :137-148 (value + gradients w.r.t. the network outputs) in two launches
extern "C" int sdumc_distill_fwd_bwd(int32_t B, float denom, const float* vals, const float* labels, const float* th,
                                     const float* ct, const float* z, const float* weights5, const float* ssd_global,
                                     float* d_vals, float* d_th, float* d_ct, float* d_z, float* losses,
                                     float* workspace, void* stream) {
  if (B <= 0 || denom <= 0.f || !vals || !labels || !th || !ct || !z || !weights5 || !d_vals || !d_th || !d_ct || !d_z ||
      !losses || !workspace)
    return SDUMC_EINVAL;
  DistillArgs a;
  a.B = B;
  a.denom = denom;
  a.vals = vals; a.labels = labels; a.th = th; a.ct = ct; a.z = z;
  for (int i = 0; i < 5; ++i) a.w[i] = weights5[i];
  a.ssd_global = ssd_global;
  a.d_vals = d_vals; a.d_th = d_th; a.d_ct = d_ct; a.d_z = d_z;
  a.losses = losses;
  a.part = workspace;
  const int64_t per[3] = {SDUMC_D, SDUMC_NQ * SDUMC_H, SDUMC_H};
  int mx = 1;
  for (int p = 0; p < 3; ++p) {
    a.nblk[p] = (int)(((int64_t)B * per[p] + DCH - 1) / DCH);
    mx = a.nblk[p] > mx ? a.nblk[p] : mx;
  }
  hipStream_t st = as_stream(stream);
  if (!ssd_global) {
    hipLaunchKernelGGL(distill_partials_kernel, dim3(mx, 3), dim3(256), 0, st, a);
    SDUMC_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(distill_apply_kernel, dim3(mx > 2 ? mx : 2, 4), dim3(256), 0, st, a);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

// MSE x2 + RMSE x3 + RnC over cat(stream 0, stream 1) with the labels repeated (main :134-148): values and every gradient
// w.r.t. the network outputs in two launches.  Returns 1 when the shape is not one it takes (n = 2B > 256: the sorted RnC form).
extern "C" int sdumc_losses_fused_(int32_t B, const float* vals, const float* labels, const float* th, const float* ct,
                                   const float* z, const float* rnc_feats, int32_t rd, float temperature, const float* weights6,
                                   float* d_vals, float* d_th, float* d_ct, float* d_z, float* d_rnc, float* losses,
                                   float* distill_ws, float* rnc_workspace, float* hyper, double beta1, double beta2,
                                   void* stream) {
  const int n = 2 * B;
  if (B <= 0 || n > 256 || rd <= 0 || temperature <= 0.f) return 1;
  const size_t lds1 = (((size_t)rd + 3) & ~(size_t)3) * sizeof(float) + 5 * (size_t)n * sizeof(float);
  const size_t lds2 = ((size_t)n + 256) * sizeof(float);
  if (lds1 > 64 * 1024) return 1;
  DistillArgs a;
  a.B = B;
  a.denom = (float)B;
  a.vals = vals; a.labels = labels; a.th = th; a.ct = ct; a.z = z;
  for (int i = 0; i < 5; ++i) a.w[i] = weights6[i];
  a.ssd_global = nullptr;
  a.d_vals = d_vals; a.d_th = d_th; a.d_ct = d_ct; a.d_z = d_z;
  a.losses = losses;
  a.part = distill_ws;
  const int64_t per[3] = {SDUMC_D, SDUMC_NQ * SDUMC_H, SDUMC_H};
  int mx = 1;
  for (int p = 0; p < 3; ++p) {
    a.nblk[p] = (int)(((int64_t)B * per[p] + DCH - 1) / DCH);
    mx = a.nblk[p] > mx ? a.nblk[p] : mx;
  }
  const int amx = mx > 2 ? mx : 2;
  RncArgs r;
  r.f = rnc_feats; r.labels = labels; r.label_mod = B; r.n = n; r.dim = rd;
  r.inv_t = 1.f / temperature; r.weight = weights6[5];
  r.w = rnc_ws(rnc_workspace, n);
  r.loss_out = losses + 6;
  r.df = d_rnc;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(loss_stage1_kernel, dim3(n + 3 * mx), dim3(256), lds1, st, a, r, mx);
  SDUMC_CHECK_LAUNCH();
  hipLaunchKernelGGL(loss_stage2_kernel, dim3(n + 4 * amx + 1), dim3(256), lds2, st, a, r, amx, hyper, beta1, beta2);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

extern "C" int sdumc_rnc_dfeat_rows(const float* feats, int32_t n, int32_t dim, float temperature, float weight,
                                    int32_t row0, int32_t rows, float* dfeats, float* workspace, void* stream) {
  if (!feats || !dfeats || !workspace || n < 2 || dim <= 0 || row0 < 0 || rows <= 0 || row0 + rows > n) return SDUMC_EINVAL;
  const RncWs w = rnc_ws(workspace, n);
  hipLaunchKernelGGL(rnc_dfeat_kernel, dim3(rows), dim3(256), (n + 256) * sizeof(float), as_stream(stream), feats, n,
                     dim, 1.f / temperature, weight, row0, w, dfeats);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

extern "C" int sdumc_rnc_mask(const float* labels, int32_t n, uint8_t* mask, void* stream) {
  if (!labels || !mask || n < 2) return SDUMC_EINVAL;
  const int64_t total = (int64_t)n * (n - 1) * (n - 1);
  hipLaunchKernelGGL(rnc_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), labels, n, mask);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

// One empty kernel per source file = per gfx950 code object: sdumc_preload_() asks for its attributes, which makes the HIP runtime load
// this file's code object NOW (outside any timed or latency-sensitive region) instead of at the first launch of one of its kernels --
// with deferred loading that first launch stalls the host for tens of milliseconds (seen as a 36-59 ms gap in the middle of an epoch,
// at the first batch whose shape took a fallback path: profiles/README.md, round 6).
__global__ void sdumc_preload_loss_kernel() {}
extern "C" int sdumc_preload_loss_(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&sdumc_preload_loss_kernel)) == hipSuccess ? SDUMC_OK : SDUMC_ELAUNCH;
}
