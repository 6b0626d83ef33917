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
// Structure (MI355X-first): the work is a stream over [V*T] rows of 1 KiB, so the grid is
// (64-row chunk, v) -- hundreds of workgroups instead of one per sample -- and the softmax is
// flash-style: every chunk produces (max, sum, unnormalised pooled partial), a tiny per-v kernel
// combines the chunks and normalises the stored weights.  The row x query products (scores in the
// forward, dA = xd . dO^T in the backward) are [rows,256] x [256,<=8] contractions: they run on the
// matrix cores as v_mfma_f32_16x16x4_f32 (queries padded to 16 columns) instead of 64-lane shuffle
// reductions; the [rows] x [channels] parts stay on the VALU with one 16-B access per lane per row.
// No float atomics: partials are reduced in a fixed order (bitwise reproducible).
#include <type_traits>

#include "common.h"
#include "p3_loop.h"
#include <cstring>

namespace {

constexpr int D = SDUMC_D;       // one channel block = 256 channels = 64 lanes x 4; a row holds C blocks (dim = 256 C)
constexpr int MAXQ = 8;
constexpr int CH = 64;           // rows of one v handled by a 4-wave workgroup (16 per wave)

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// agent-scope (write-through / L2-bypassing) accesses for data that another workgroup of the SAME launch reads after a ticket
// hand-shake (descriptor field `tickets`): the chunks of one sample may run on different XCDs, whose L2s are not coherent
__device__ __forceinline__ void stf_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ldf_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st4_dev(float* p, f32x4 v) {
#pragma unroll
  for (int j = 0; j < 4; ++j) stf_dev(p + j, v[j]);
}
__device__ __forceinline__ f32x4 ld4_dev(const float* p) {
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = ldf_dev(p + j);
  return v;
}
// true in exactly one workgroup per ticket: the last of `n` to arrive (which also re-arms the ticket).  Every thread calls it,
// after the workgroup's agent-scope stores; ends with a barrier.
__device__ __forceinline__ bool last_arrival(uint32_t* ticket, uint32_t n, int* s_flag) {
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = t == n - 1u;
    if (t == n - 1u) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return *s_flag != 0;
}
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
// Frame tensors (x, keys, dz, dxd) are fp32, or bf16 in the engine's bf16-storage mode (sdumc_attnpool.bf16): `base` is the
// tensor's address as float*, `off` an ELEMENT offset; 4 consecutive channels either way (16 or 8 bytes).
template <bool HF>
__device__ __forceinline__ f32x4 ldx(const float* base, size_t off) {
  if constexpr (HF) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off);
    return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                 __uint_as_float(u.y & 0xffff0000u)};
  } else {
    return ld4(base + off);
  }
}
template <bool HF>
__device__ __forceinline__ void stx(float* base, size_t off, f32x4 v) {
  if constexpr (HF) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};      // round to nearest even
    *reinterpret_cast<bf16x4*>(reinterpret_cast<unsigned short*>(base) + off) = h;
  } else {
    st4(base + off, v);
  }
}

// workspace layout of the forward: part [V][nchunk][nq][256], then stats [V][nchunk][2][MAXQ]
struct FwdWs {
  float* part;
  float* stats;
};
__host__ __device__ inline FwdWs fwd_ws(float* w, int V, int nchunk, int nq, int dd) {
  FwdWs r;
  r.part = w;
  r.stats = w + (size_t)V * nchunk * nq * dd;
  return r;
}
__host__ __device__ inline int row_dim(const sdumc_attnpool& p) { return p.dim > 0 ? p.dim : D; }

// [16 rows of this wave] x [16 query columns] = sum over 256 channels, on the matrix cores.
// A operand: lane (r = lane&15, kk = lane>>4) supplies rows[r][16 j + 4 kk + e]; B operand: the same
// channel of column r.  `rowp` = this lane's row pointer (or nullptr), bq[j] = this lane's B fragments.
// The B side lives in LDS as [MAXQ][LDQ] (LDQ = 272: rows 16 banks apart -> at most 2-way conflicts).
template <bool DROP, int C, bool HF = false>
__device__ __forceinline__ f32x4 rows_times_cols(const float* rowp, const float* b_lds, const DropRT& d,
                                                 uint32_t vrow, int r16, int kk, size_t rowoff = 0) {
  constexpr int LDQ = D * C + 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // two batches of 8 channel groups: a batch's row loads (and keep-bits bytes) are all issued before its first MFMA
  const bool bits = DROP && d.bits != nullptr;
#pragma unroll
  for (int jb = 0; jb < 16 * C; jb += 8) {
    f32x4 a[8];
    uint32_t mb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = jb + u;
      a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      mb[u] = 0;
      if (rowp) {
        a[u] = ldx<HF>(rowp, rowoff + 16 * j + 4 * kk);
        if (bits) mb[u] = d.bits[(size_t)vrow * d.qwidth + 4 * j + kk];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = jb + u;
      if (DROP) {
        if (bits) {
          const uint32_t m = mb[u];
          a[u][0] = (m & 1u) ? a[u][0] * d.scale : 0.f;
          a[u][1] = (m & 2u) ? a[u][1] * d.scale : 0.f;
          a[u][2] = (m & 4u) ? a[u][2] * d.scale : 0.f;
          a[u][3] = (m & 8u) ? a[u][3] * d.scale : 0.f;
        } else if (rowp) {
          a[u] *= drop_mask4(d, vrow, (uint32_t)(4 * j + kk));
        }
      }
      f32x4 bq = {0.f, 0.f, 0.f, 0.f};
      if (r16 < MAXQ) bq = ld4(b_lds + r16 * LDQ + 16 * j + 4 * kk);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], bq[e], acc, 0, 0, 0);
    }
  }
  return acc;
}

// ---- forward, pass 2: combine the chunks of one v, normalise the weights, output dropout ---------
// DEV: called by the last chunk workgroup of v inside the partial kernel: the other chunks' results are read at agent scope
template <bool DEV>
__device__ __forceinline__ void attn_fwd_combine_body(const sdumc_attnpool& p, const float* ws, const int nchunk, const int v,
                                                      float* fac /* LDS [nchunk][MAXQ]: exp(m_c - m) / l */) {
  const int tid = threadIdx.x;
  const int T = p.T, nq = p.nq, DD = row_dim(p);
  const FwdWs w = fwd_ws(const_cast<float*>(ws), p.V, nchunk, nq, DD);
  const float* st = w.stats + (size_t)v * nchunk * 2 * MAXQ;
  if (tid < nq) {
    float m = -INFINITY;
    auto ldst = [&](int k) { return DEV ? ldf_dev(st + k) : st[k]; };
    for (int c = 0; c < nchunk; ++c) m = fmaxf(m, ldst(c * 2 * MAXQ + tid));
    float l = 0.f;
    for (int c = 0; c < nchunk; ++c) l += ldst(c * 2 * MAXQ + MAXQ + tid) * expf(ldst(c * 2 * MAXQ + tid) - m);
    const float inv = 1.f / l;
    for (int c = 0; c < nchunk; ++c) fac[c * MAXQ + tid] = expf(ldst(c * 2 * MAXQ + tid) - m) * inv;
  }
  __syncthreads();
  const DropRT od = drop_resolve(p.out_drop);
  for (int e = tid; e < nq * (DD / 4); e += 256) {
    const int i = e / (DD / 4), cq = e - i * (DD / 4);
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nchunk; ++c) {
      const float* src = w.part + (((size_t)v * nchunk + c) * nq + i) * DD + 4 * cq;
      sum += (DEV ? ld4_dev(src) : ld4(src)) * fac[c * MAXQ + i];
    }
    const size_t o = ((size_t)v * nq + i) * DD + 4 * cq;
    st4(p.pooled + o, sum);
    if (od.enabled) sum *= drop_mask4(od, (uint32_t)(v * nq + i), (uint32_t)cq);
    st4(p.out + o, sum);
  }
  float* attn = p.attn + (size_t)v * T * nq;
  for (int e = tid; e < T * nq; e += 256) {
    const int t = e / nq, i = e - t * nq;
    attn[e] = (DEV ? ldf_dev(attn + e) : attn[e]) * fac[(t / CH) * MAXQ + i];
  }
}

// ---- forward, pass 1: one workgroup per (chunk of 64 rows, v) -----------------------------------
// PHILOX: the input dropout mask is recomputed per row (no precomputed keep-bits attached: tests, one-off calls)
// KLDS (the fused UMCA kernel, umca_fwd_kernel below): the chunk's tanh keys are NOT read from p.keys but from the LDS tile
// k_lds [CH][UMCA_LDK] the key projection has just left there; q_ext / red_ext are the caller's LDS buffers for the query tile
// and for the pooling reduce (the latter may overlay k_lds: it is first written after the last score has been computed).
constexpr int UMCA_LDK = D + 4;
template <bool PHILOX, int C, bool HF = false, bool KLDS = false>
__device__ __forceinline__ void attn_fwd_partial_body(const sdumc_attnpool p, float* ws, const int nchunk, const int chunk, const int v,
                                                      const float* k_lds = nullptr, float* q_ext = nullptr, float* red_ext = nullptr) {
  constexpr int DD = D * C, LDQ = DD + 16;
  constexpr int RED = 4 * MAXQ * D > MAXQ * LDQ ? 4 * MAXQ * D : MAXQ * LDQ;
  __shared__ __attribute__((aligned(16))) float P_s[CH * MAXQ];
  float* red;                                                // first the query tile, later the pooling reduce (per channel block)
  if constexpr (KLDS) {
    red = red_ext;
  } else {
    __shared__ __attribute__((aligned(16))) float red_s[RED];
    red = red_s;
  }
  __shared__ float wstat[4][16];
  __shared__ float cstat[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kk = lane >> 4;
  const int T = p.T, nq = p.nq;
  const int t0 = chunk * CH;
  const FwdWs w = fwd_ws(ws, p.V, nchunk, nq, DD);
  const bool fuse = p.tickets != nullptr;     // the last chunk of v to finish also combines (no second launch)
  // key-padding extension: frames at or beyond Tv get weight 0 (a chunk wholly beyond Tv reports max -inf, sum 0)
  const int Tv = p.lengths ? min(T, max(1, p.lengths[v])) : T;

  // scores of this wave's 16 rows against the (<= 8) queries
  float* q_s = KLDS ? q_ext : red;
  for (int e = tid; e < MAXQ * (DD / 4); e += 256) {
    const int i = e / (DD / 4), cq = e - i * (DD / 4);
    st4(q_s + i * LDQ + 4 * cq, i < nq ? ld4(p.q + (size_t)v * p.q_stride + (size_t)i * DD + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f});
  }
  __syncthreads();
  // The pooling below streams this wave's 16 frame rows (one 16-byte access per lane and row).  Rows 0-7 are requested HERE,
  // before the scores: their HBM round trip then runs under the key-row loads, the MFMAs and the two chunk reductions instead
  // of being exposed behind them; rows 8-15 are requested while rows 0-7 are consumed.  (Round 3: the forward pooling of the
  // three Cross_Attention sites 40.6 -> see profiles/README.md.)
  constexpr int RB = 4;
  const DropRT xd = drop_resolve(p.x_drop);
  const int vx = v % p.x_samples;
  const bool masked = !PHILOX && xd.enabled != 0;
  f32x4 xa[RB], xb[RB];
  uint32_t ma[RB], mb2[RB];
  auto load_rows = [&](int rb, f32x4* xr, uint32_t* mb) {       // channel block 0
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int t = min(t0 + 16 * wave + rb + j, T - 1);   // rows beyond T re-read the last row; their P_s is 0
      xr[j] = ldx<HF>(p.x, ((size_t)vx * T + t) * DD + 4 * lane);
      mb[j] = masked ? xd.bits[(size_t)(v * T + t) * xd.qwidth + lane] : 0xfu;
    }
  };
  if constexpr (!PHILOX) {
    load_rows(0, xa, ma);
    load_rows(RB, xb, mb2);
  }
  const int myrow = t0 + 16 * wave + r16;
  const DropRT nodrop = {};
  const f32x4 s4 = KLDS ? rows_times_cols<false, C, false>(k_lds, q_s, nodrop, 0u, r16, kk, (size_t)(16 * wave + r16) * UMCA_LDK)
                        : rows_times_cols<false, C, HF>(myrow < T ? p.keys : nullptr, q_s, nodrop, 0u, r16, kk, ((size_t)v * T + myrow) * DD);
  // C layout: column (query) = lane & 15, rows = 16 wave + 4 (lane >> 4) + e
  float s[4], mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int t = t0 + 16 * wave + 4 * kk + e;
    s[e] = t < Tv ? p.scale * s4[e] : -INFINITY;
    mx = fmaxf(mx, s[e]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  if (lane < 16) wstat[wave][lane] = mx;
  __syncthreads();
  if (tid < 16) cstat[tid] = fmaxf(fmaxf(wstat[0][tid], wstat[1][tid]), fmaxf(wstat[2][tid], wstat[3][tid]));
  __syncthreads();
  const float mc = cstat[r16];          // chunk max of my query column (-inf only for a chunk wholly beyond Tv)
  float lsum = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int rl = 16 * wave + 4 * kk + e, t = t0 + rl;
    const float pe = t < Tv ? expf(s[e] - mc) : 0.f;
    lsum += pe;
    if (r16 < MAXQ) P_s[rl * MAXQ + r16] = pe;
    if (r16 < nq && t < T) {                                               // normalised by the combine step
      if (fuse) stf_dev(p.attn + ((size_t)v * T + t) * nq + r16, pe);
      else p.attn[((size_t)v * T + t) * nq + r16] = pe;
    }
  }
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  __syncthreads();                      // cstat reads done, P_s complete
  if (lane < 16) wstat[wave][lane] = lsum;
  __syncthreads();
  if (tid < nq) {
    float* st = w.stats + ((size_t)v * nchunk + chunk) * 2 * MAXQ;
    const float l = wstat[0][tid] + wstat[1][tid] + wstat[2][tid] + wstat[3][tid];
    if (fuse) { stf_dev(st + tid, cstat[tid]); stf_dev(st + MAXQ + tid, l); }
    else { st[tid] = cstat[tid]; st[MAXQ + tid] = l; }
  }
  // unnormalised pooling of this chunk, one 256-channel block at a time: lane owns channels 4*lane..4*lane+3 of the
  // block, wave owns 16 rows
  // (a generic lambda called once per block, not a loop over blocks: ANY enclosing loop -- even one of trip count 1 --
  // makes hipcc (ROCm 7.2) unroll the row loops inside it despite their `unroll 1` and spill ~4600 VGPRs)
  auto pool_block = [&](auto cbc) {
    constexpr int cb = decltype(cbc)::value;
    const int ch = D * cb + 4 * lane;           // this lane's first channel
    f32x4 acc[MAXQ];
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PHILOX) {
      // row at a time
#pragma unroll 1
      for (int r = 0; r < 16; ++r) {
        const int rl = 16 * wave + r, t = t0 + rl;
        if (t >= T) break;
        f32x4 x = ldx<HF>(p.x, ((size_t)vx * T + t) * DD + ch);
        x *= drop_mask4(xd, (uint32_t)(v * T + t), (uint32_t)(ch >> 2));
#pragma unroll
        for (int i = 0; i < MAXQ; ++i)
          if (i < nq) acc[i] += x * P_s[rl * MAXQ + i];
      }
    } else if constexpr (cb == 0) {
      // rows in batches of RB, two batches in flight (requested before the scores, see above)
      const float mscale = masked ? xd.scale : 1.f;
      auto consume = [&](int rb, const f32x4* xr, const uint32_t* mb) {
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const int rl = 16 * wave + rb + j;
          f32x4 x = xr[j];
          const uint32_t b = mb[j];
          x[0] = (b & 1u) ? x[0] * mscale : 0.f;
          x[1] = (b & 2u) ? x[1] * mscale : 0.f;
          x[2] = (b & 4u) ? x[2] * mscale : 0.f;
          x[3] = (b & 8u) ? x[3] * mscale : 0.f;
#pragma unroll
          for (int i = 0; i < MAXQ; ++i)
            if (i < nq) acc[i] += x * P_s[rl * MAXQ + i];   // rows beyond T: P_s = 0
        }
      };
      consume(0, xa, ma);
      load_rows(2 * RB, xa, ma);
      consume(RB, xb, mb2);
      load_rows(3 * RB, xb, mb2);
      consume(2 * RB, xa, ma);
      consume(3 * RB, xb, mb2);
    } else {
      // (channel blocks 1..3 of the blocks' stand-alone 1024-channel use: batches of RB, one at a time)
      const float mscale = masked ? xd.scale : 1.f;
#pragma unroll 1
      for (int rb = 0; rb < 16; rb += RB) {
        f32x4 xr[RB];
        uint32_t mb[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const int t = min(t0 + 16 * wave + rb + j, T - 1);   // rows beyond T re-read the last row; their P_s is 0
          xr[j] = ldx<HF>(p.x, ((size_t)vx * T + t) * DD + ch);
          mb[j] = masked ? xd.bits[(size_t)(v * T + t) * xd.qwidth + (ch >> 2)] : 0xfu;
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const int rl = 16 * wave + rb + j;
          f32x4 x = xr[j];
          {
            const uint32_t b = mb[j];
            x[0] = (b & 1u) ? x[0] * mscale : 0.f;
            x[1] = (b & 2u) ? x[1] * mscale : 0.f;
            x[2] = (b & 4u) ? x[2] * mscale : 0.f;
            x[3] = (b & 8u) ? x[3] * mscale : 0.f;
          }
#pragma unroll
          for (int i = 0; i < MAXQ; ++i)
            if (i < nq) acc[i] += x * P_s[rl * MAXQ + i];   // rows beyond T: x = 0 and P_s = 0
        }
      }
    }
    if (cb > 0) __syncthreads();              // the previous block's reduce has read `red`
#pragma unroll
    for (int i = 0; i < MAXQ; ++i)
      if (i < nq) st4(red + (wave * MAXQ + i) * D + 4 * lane, acc[i]);
    __syncthreads();
    for (int e = tid; e < nq * (D / 4); e += 256) {
      const int i = e / (D / 4), cq = e - i * (D / 4);
      f32x4 sum = ld4(red + i * D + 4 * cq);
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) sum += ld4(red + (ww * MAXQ + i) * D + 4 * cq);
      float* dst = w.part + (((size_t)v * nchunk + chunk) * nq + i) * DD + D * cb + 4 * cq;
      if (fuse) st4_dev(dst, sum);
      else st4(dst, sum);
    }
  };
  pool_block(std::integral_constant<int, 0>{});
  if constexpr (C > 1) pool_block(std::integral_constant<int, 1>{});
  if constexpr (C > 2) pool_block(std::integral_constant<int, 2>{});
  if constexpr (C > 3) pool_block(std::integral_constant<int, 3>{});
  if (fuse) {
    __shared__ int s_last;
    if (last_arrival(p.tickets + v, (uint32_t)nchunk, &s_last)) attn_fwd_combine_body<true>(p, ws, nchunk, v, red);
  }
}

template <bool PHILOX, int C, bool HF = false>
__global__ __launch_bounds__(256, C == 1 ? 4 : 2) void attn_fwd_partial_kernel(const sdumc_attnpool p, float* ws, const int nchunk) {
  attn_fwd_partial_body<PHILOX, C, HF>(p, ws, nchunk, blockIdx.x, blockIdx.y);
}

// Several pooling sites in ONE launch (the three Cross_Attention blocks of a step: sdumc_attnpool_fwd_multi): a flat grid, the
// heaviest site first; workgroup -> (site, chunk, v).  256-channel rows with keep-bits (or no mask) only.
constexpr int MAXSITES = 4;
struct MultiFwd {
  sdumc_attnpool p[MAXSITES];
  int32_t wg_end[MAXSITES];     // exclusive prefix ends of the partial kernel's flat grid
  int32_t v_end[MAXSITES];      // same for the combine kernel (one workgroup per v)
  int32_t nchunk[MAXSITES];
};
struct MultiBwd {
  sdumc_attnpool_bwd_t b[MAXSITES];
  int32_t wg_end[MAXSITES];
  int32_t nchunk[MAXSITES];
  unsigned long long dq_end[MAXSITES];   // prefix ends of the dq reduce (elements)
};
__device__ __forceinline__ int site_of(const int32_t* ends, const int bid) {
  int s = 0;
#pragma unroll
  for (int i = 0; i < MAXSITES - 1; ++i) s += bid >= ends[i] ? 1 : 0;
  return s;
}

template <bool HF>
__global__ __launch_bounds__(256, 4) void attn_fwd_partial_multi_kernel(const MultiFwd m) {
  const int s = site_of(m.wg_end, blockIdx.x);
  const int local = blockIdx.x - (s ? m.wg_end[s - 1] : 0);
  const int nchunk = m.nchunk[s];
  sdumc_attnpool p = m.p[0];   // uniform select, by value (see attnpool_bwd_multi_kernel)
  if (s == 1) p = m.p[1];
  if (s == 2) p = m.p[2];
  if (s == 3) p = m.p[3];
  attn_fwd_partial_body<false, 1, HF>(p, static_cast<float*>(p.workspace), nchunk, local % nchunk, local / nchunk);
}

__global__ __launch_bounds__(256) void attn_fwd_combine_kernel(const sdumc_attnpool p, const float* ws, const int nchunk) {
  extern __shared__ float fac[];
  attn_fwd_combine_body<false>(p, ws, nchunk, blockIdx.x, fac);
}
__global__ __launch_bounds__(256) void attn_fwd_combine_multi_kernel(const MultiFwd m) {
  extern __shared__ float fac[];
  const int s = site_of(m.v_end, blockIdx.x);
  const sdumc_attnpool& p = m.p[s];
  attn_fwd_combine_body<false>(p, static_cast<const float*>(p.workspace), m.nchunk[s], blockIdx.x - (s ? m.v_end[s - 1] : 0), fac);
}

// ---- backward: one workgroup per (chunk, v) ----------------------------------------------------
// dO = dout * out_mask ; delta_i = dO_i . O_i ; per row t:
//   dA_i = xd_t . dO_i (matrix cores) ; dS_i = 0.3 A_ti (dA_i - delta_i)
//   dK_t = sum_i dS_i Q_i ; dz_t = dK_t (1 - K_t^2)            -> dz
//   dxd_t (pool path) = sum_i A_ti dO_i                         -> dxd
//   dQ_i += dS_i K_t                                            -> per-chunk slabs (deterministic)
template <int C, bool HF = false>
__device__ __forceinline__ void attnpool_bwd_body(const sdumc_attnpool_bwd_t b, float* dq_part, const int nchunk, const int chunk,
                                                  const int v) {
  constexpr int DD = D * C, LDQ = DD + 16;
  __shared__ __attribute__((aligned(16))) float dO_s[MAXQ * LDQ];
  __shared__ __attribute__((aligned(16))) float red[4 * MAXQ * D];
  __shared__ __attribute__((aligned(16))) float dS_s[CH * MAXQ];
  __shared__ __attribute__((aligned(16))) float A_s[CH * MAXQ];
  __shared__ float delta_s[16];
  const sdumc_attnpool& p = b.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kk = lane >> 4;
  const int T = p.T, nq = p.nq;
  const int t0 = chunk * CH;
  const bool fuse = p.tickets != nullptr;
  const DropRT od = drop_resolve(p.out_drop);
  // this wave's first four key rows are requested before anything else: the round trip runs under the dO staging, the
  // delta dot products and the dA MFMAs
  constexpr int RB = 4;
  f32x4 kr[2][RB];
  auto load_batch = [&](int rb, f32x4* dst, int ch) {
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int t = t0 + 16 * wave + rb + j;
      dst[j] = t < T ? ldx<HF>(p.keys, ((size_t)v * T + t) * DD + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  load_batch(0, kr[0], 4 * lane);
  for (int e = tid; e < MAXQ * (DD / 4); e += 256) {
    const int i = e / (DD / 4), cq = e - i * (DD / 4);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (i < nq) {
      g = ld4(b.dout + ((size_t)v * nq + i) * DD + 4 * cq);
      if (od.enabled) g *= drop_mask4(od, (uint32_t)(v * nq + i), (uint32_t)cq);
    }
    st4(dO_s + i * LDQ + 4 * cq, g);
  }
  if (tid < 16) delta_s[tid] = 0.f;
  __syncthreads();
  for (int i = wave; i < nq; i += 4) {
    float d = 0.f;
#pragma unroll
    for (int cb = 0; cb < C; ++cb)
      d += dot4(ld4(dO_s + i * LDQ + D * cb + 4 * lane), ld4(p.pooled + ((size_t)v * nq + i) * DD + D * cb + 4 * lane));
    d = wave_sum(d);
    if (lane == 0) delta_s[i] = d;
  }
  __syncthreads();

  // dA for this wave's 16 rows on the matrix cores: A = xd rows, B = dO columns
  const DropRT xd = drop_resolve(p.x_drop);
  const int vx = v % p.x_samples;
  {
    const int myrow = t0 + 16 * wave + r16;
    const float* rowp = myrow < T ? p.x : nullptr;
    const size_t rowoff = ((size_t)vx * T + myrow) * DD;
    const uint32_t vrow = (uint32_t)(v * T + myrow);
    const f32x4 dA = xd.enabled ? rows_times_cols<true, C, HF>(rowp, dO_s, xd, vrow, r16, kk, rowoff)
                                : rows_times_cols<false, C, HF>(rowp, dO_s, xd, vrow, r16, kk, rowoff);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rl = 16 * wave + 4 * kk + e, t = t0 + rl;
      if (r16 < MAXQ) {
        float a = 0.f, dS = 0.f;
        if (r16 < nq && t < T) {
          a = p.attn[((size_t)v * T + t) * nq + r16];
          dS = p.scale * a * (dA[e] - delta_s[r16]);
        }
        A_s[rl * MAXQ + r16] = a;
        dS_s[rl * MAXQ + r16] = dS;
      }
    }
  }
  __syncthreads();

  // one 256-channel block at a time
#pragma unroll 1
  for (int cb = 0; cb < C; ++cb) {
    const int ch = D * cb + 4 * lane;
    f32x4 q[MAXQ], g[MAXQ], dqa[MAXQ];
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
      dqa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < nq) {
        q[i] = ld4(p.q + (size_t)v * p.q_stride + (size_t)i * DD + ch);
        g[i] = ld4(dO_s + i * LDQ + ch);
      }
    }
    // rows in batches of RB: the batch's key-row loads are all in flight before the first row is processed
    // (bf16 rows: 8 per batch measured SLOWER -- 1.156 vs 1.128 ms per bf16 step -- although they are half as long)
    // (round 3: the NEXT batch's loads are issued before this batch's stores -- the compiler does not move a load above a
    //  store that may alias it, so every batch used to start with an exposed HBM round trip)
    if (cb > 0) load_batch(0, kr[0], ch);
#pragma unroll
    for (int rb = 0; rb < 16; rb += RB) {
      if (rb + RB < 16) load_batch(rb + RB, kr[((rb / RB) + 1) & 1], ch);
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int rl = 16 * wave + rb + j, t = t0 + rl;
        const size_t row = (size_t)v * T + t;
        const f32x4 k = kr[(rb / RB) & 1][j];
        f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MAXQ; ++i)
          if (i < nq) {
            const float dS = dS_s[rl * MAXQ + i], a = A_s[rl * MAXQ + i];   // both 0 for rows beyond T
            dk += q[i] * dS;
            dx += g[i] * a;
            dqa[i] += k * dS;
          }
        if (t < T) {
          const f32x4 one = {1.f, 1.f, 1.f, 1.f};
          stx<HF>(b.dz, row * DD + ch, dk * (one - k * k));
          stx<HF>(b.dxd, row * DD + ch, dx);
        }
      }
    }
    if (cb > 0) __syncthreads();              // the previous block's reduce has read `red`
#pragma unroll
    for (int i = 0; i < MAXQ; ++i)
      if (i < nq) st4(red + (wave * MAXQ + i) * D + 4 * lane, dqa[i]);
    __syncthreads();
    for (int e = tid; e < nq * (D / 4); e += 256) {
      const int i = e / (D / 4), cq = e - i * (D / 4);
      f32x4 s = ld4(red + i * D + 4 * cq);
#pragma unroll
      for (int w = 1; w < 4; ++w) s += ld4(red + (w * MAXQ + i) * D + 4 * cq);
      float* dst = dq_part + (((size_t)v * nchunk + chunk) * nq + i) * DD + D * cb + 4 * cq;
      if (fuse) st4_dev(dst, s);
      else st4(dst, s);
    }
  }
  if (fuse) {   // the last chunk of v to finish sums the per-chunk slabs (fixed order: deterministic)
    __shared__ int s_last;
    if (last_arrival(p.tickets + p.V + v, (uint32_t)nchunk, &s_last)) {
      const int per_v = nq * DD;
      for (int e = tid; e < per_v / 4; e += 256) {
        const float* src = dq_part + (size_t)v * nchunk * per_v + 4 * e;
        f32x4 a = ld4_dev(src);
        for (int c = 1; c < nchunk; ++c) a += ld4_dev(src + (size_t)c * per_v);
        st4(b.dq + (size_t)v * per_v + 4 * e, a);
      }
    }
  }
}

template <int C, bool HF = false>
__global__ __launch_bounds__(256, 2) void attnpool_bwd_kernel(const sdumc_attnpool_bwd_t b, float* dq_part,
                                                           const int nchunk) {
  attnpool_bwd_body<C, HF>(b, dq_part, nchunk, blockIdx.x, blockIdx.y);
}
template <bool HF>
__global__ __launch_bounds__(256, 2) void attnpool_bwd_multi_kernel(const MultiBwd m) {
  const int s = site_of(m.wg_end, blockIdx.x);
  const int local = blockIdx.x - (s ? m.wg_end[s - 1] : 0);
  const int nchunk = m.nchunk[s];
  // (the descriptor reaches the body BY VALUE through a uniform select: handed over as a reference into the kernel argument --
  //  indexed or not -- hipcc (ROCm 7.2) compiled the same body to 256 VGPRs + 6500 spilled dwords instead of 136 VGPRs)
  sdumc_attnpool_bwd_t b = m.b[0];
  if (s == 1) b = m.b[1];
  if (s == 2) b = m.b[2];
  if (s == 3) b = m.b[3];
  attnpool_bwd_body<1, HF>(b, static_cast<float*>(b.workspace), nchunk, local % nchunk, local / nchunk);
}

// ====================================================================================================================
// Round 4: the network path's pooling kernels re-cut for memory-level parallelism ("v2": 256-channel rows, keep-bits or
// pre-masked bf16 frames, two-pass combine -- everything the engine launches; the kernels above stay for 512..1024-channel rows,
// recomputed Philox masks and the ticket option).
// What bounded the kernels above (profiles/r3x: 4.2 TB/s = 0.52 of the HBM peak backward, 4.3 forward): a wave never had more
// than 8-12 KiB of frame rows in flight (4-row register batches behind a prologue of three dependent round trips), and the
// grids' last dispatch round was 40-80 % full.  Here
//   * the global loads of a workgroup are issued in its first instructions, smallest first (loads return in order: the prologue's
//     few dwords must not queue behind the rows): per wave the 16 rows of the MFMA operand tile at once, and the 16 rows of the
//     stream tile right behind them (bf16 rows) or as soon as the MFMA phase has released the operand tile's registers (fp32
//     rows: occupancy -- three / four waves per SIMD -- measured worth more than the second 16 KiB in flight);
//   * the per-row softmax factors reach the row x channel FMAs through v_readlane from the MFMA result layout (SGPR operands:
//     no LDS round trip, no barrier between the two phases);
//   * workgroup -> (stream, sample, chunk) is XCD-aware: the two streams' workgroups that read the SAME frame rows x[b, t0..]
//     (audio / video: one x for both streams) are dispatched 8 apart, i.e. to the same XCD at about the same time, so the second
//     read is an L2 hit instead of a second HBM fetch.
// Same arithmetic and summation order as the kernels above (bit-identical results: tests/test_gpu_ops.py).
// ====================================================================================================================
template <bool HF> struct RawRow { typedef f32x4 type; };
template <> struct RawRow<true> { typedef uint2 type; };
template <bool HF>
__device__ __forceinline__ typename RawRow<HF>::type ldraw(const float* base, size_t off) {
  if constexpr (HF) return *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off);
  else return ld4(base + off);
}
template <bool HF>
__device__ __forceinline__ f32x4 cvtraw(const typename RawRow<HF>::type& u) {
  if constexpr (HF) return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                                 __uint_as_float(u.y & 0xffff0000u)};
  else return u;
}
__device__ __forceinline__ float lane_bcast(float v, int srclane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srclane));
}
// keep-bits nibble -> the 4 channels of a quad: kept ones pass, dropped ones become +0 (the 1/(1-p) scale is applied by the caller)
__device__ __forceinline__ f32x4 keep4(f32x4 x, uint32_t b) {
  f32x4 r;
  r[0] = __uint_as_float(__float_as_uint(x[0]) & (uint32_t)__builtin_amdgcn_sbfe(b, 0, 1));
  r[1] = __uint_as_float(__float_as_uint(x[1]) & (uint32_t)__builtin_amdgcn_sbfe(b, 1, 1));
  r[2] = __uint_as_float(__float_as_uint(x[2]) & (uint32_t)__builtin_amdgcn_sbfe(b, 2, 1));
  r[3] = __uint_as_float(__float_as_uint(x[3]) & (uint32_t)__builtin_amdgcn_sbfe(b, 3, 1));
  return r;
}
// workgroup id inside a site -> (v, chunk): see the header comment (pairs of streams 8 workgroups apart when they share x)
__device__ __forceinline__ void v2_unit(const sdumc_attnpool& p, int local, int nchunk, int& v, int& chunk) {
  if (p.x_samples * 2 == p.V) {
    const int U = p.x_samples * nchunk;
    const int blk = local >> 4, base = blk << 3;
    const int rem = min(8, U - base), w = local - (blk << 4);
    const int s = w / rem, unit = base + (w - s * rem);
    v = s * p.x_samples + unit / nchunk;
    chunk = unit % nchunk;
  } else {
    v = local / nchunk;
    chunk = local - v * nchunk;
  }
}

// ---- backward v2 --------------------------------------------------------------------------------------------------
template <bool HF, int NQT>
__device__ __forceinline__ void attnpool_bwd_v2(const sdumc_attnpool_bwd_t b, float* dq_part, const int nchunk, const int chunk, const int v) {
  // fp32 rows: the 16 key rows are requested when the MFMA phase has released the frame tile's 64 registers -- 157 instead of 216
  // VGPRs, three waves per SIMD instead of two (measured, tools/attn_bench.py: 78.6 vs 83.4 us for the three Cross_Attention
  // sites; early keys at three waves spill: 99 us).  bf16 rows are half as wide: everything up front, three waves either way.
  constexpr bool LATE_KEYS = !HF;
  constexpr int LDQ = D + 16;
  typedef typename RawRow<HF>::type raw_t;
  __shared__ __attribute__((aligned(16))) float dO_s[MAXQ * LDQ];
  __shared__ __attribute__((aligned(16))) float red[4 * NQT * D];
  __shared__ float delta_s[16];
  const sdumc_attnpool& p = b.f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kk = lane >> 4;
  const int T = p.T;
  const int t0 = chunk * CH, tw = t0 + 16 * wave;       // this wave's first row
  const DropRT od = drop_resolve(p.out_drop);
  const DropRT xd = drop_resolve(p.x_drop);
  const bool masked = xd.enabled != 0;                  // (keep-bits attached: checked by the launcher)
  const int vx = v % p.x_samples;
  // ---- 1. every global load, smallest first -----------------------------------------------------------------------
  // dout quads of this thread (<= 2), pooled rows for delta (<= 2 per wave), the weights of the wave's rows, the queries
  f32x4 gq[2], po[2], qv[NQT];
  float av[4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 256 * u, i = e >> 6, cq = e & 63;
    gq[u] = (NQT * 64 > 256 * u && i < NQT) ? ld4(b.dout + ((size_t)v * NQT + i) * D + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = wave + 4 * u;
    po[u] = i < NQT ? ld4(p.pooled + ((size_t)v * NQT + i) * D + 4 * lane) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int t = tw + 4 * kk + e;
    av[e] = (r16 < NQT && t < T) ? p.attn[((size_t)v * T + t) * NQT + r16] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < NQT; ++i) qv[i] = ld4(p.q + (size_t)v * p.q_stride + (size_t)i * D + 4 * lane);
  // the wave's 16 frame rows as the MFMA A operand: lane (r16, kk) takes x[row r16][16 j + 4 kk ..], j = 0..15; their keep-bits
  // (64 bytes per row) as four 16-byte loads; then the 16 key rows, lane = 4 channels
  raw_t xa[16], kr[16];
  uint4 xb[4] = {};
  {
    const int t = min(tw + r16, T - 1);                 // rows beyond T re-read the last row; their weights are 0
    const size_t ro = ((size_t)vx * T + t) * D + 4 * kk;
#pragma unroll
    for (int j = 0; j < 16; ++j) xa[j] = ldraw<HF>(p.x, ro + 16 * j);
    if (!HF && masked) {
      const uint4* bp = reinterpret_cast<const uint4*>(xd.bits + ((size_t)v * T + t) * (D / 4));
#pragma unroll
      for (int j = 0; j < 4; ++j) xb[j] = bp[j];
    }
  }
  if constexpr (!LATE_KEYS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) kr[r] = ldraw<HF>(p.keys, ((size_t)v * T + min(tw + r, T - 1)) * D + 4 * lane);
  }
  // ---- 2. dO = dout * out_mask -> LDS; delta_i = dO_i . O_i ----------------------------------------------------------
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 256 * u, i = e >> 6, cq = e & 63;
    if (i < MAXQ) {
      f32x4 g = gq[u];
      if (i < NQT && od.enabled) g *= drop_mask4(od, (uint32_t)(v * NQT + i), (uint32_t)cq);
      st4(dO_s + i * LDQ + 4 * cq, g);
      if (b.dout_masked && chunk == 0 && i < NQT) st4(b.dout_masked + ((size_t)v * NQT + i) * D + 4 * cq, g);
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = wave + 4 * u;
    if (i < NQT) {
      const float d = wave_sum(dot4(ld4(dO_s + i * LDQ + 4 * lane), po[u]));
      if (lane == 0) delta_s[i] = d;
    }
  }
  __syncthreads();
  // ---- 3. dA = xd . dO^T on the matrix cores; dS = 0.3 A (dA - delta) -------------------------------------------------
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  {
    const uint32_t sh = 8u * (uint32_t)kk;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      f32x4 a = cvtraw<HF>(xa[j]);
      if (!HF && masked) {
        const uint4 w4 = xb[j >> 2];
        const uint32_t w = (j & 3) == 0 ? w4.x : (j & 3) == 1 ? w4.y : (j & 3) == 2 ? w4.z : w4.w;
        a = keep4(a, w >> sh);
      }
      f32x4 bq = {0.f, 0.f, 0.f, 0.f};
      if (r16 < MAXQ) bq = ld4(dO_s + r16 * LDQ + 16 * j + 4 * kk);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], bq[e], acc, 0, 0, 0);
    }
  }
  if constexpr (LATE_KEYS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) kr[r] = ldraw<HF>(p.keys, ((size_t)v * T + min(tw + r, T - 1)) * D + 4 * lane);
  }
  const float xscale = (!HF && masked) ? xd.scale : 1.f;
  float dSv[4];
  {
    const float dl = r16 < NQT ? delta_s[r16] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) dSv[e] = p.scale * av[e] * (acc[e] * xscale - dl);      // rows beyond T, columns beyond nq: av = 0
  }
  // ---- 4. the row x channel parts: lane = 4 channels, the row's factors by v_readlane ----------------------------------
  f32x4 g[NQT], dqa[NQT];
#pragma unroll
  for (int i = 0; i < NQT; ++i) {
    g[i] = ld4(dO_s + i * LDQ + 4 * lane);
    dqa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const f32x4 one = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const f32x4 k = cvtraw<HF>(kr[r]);
    f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NQT; ++i) {
      const float dS = lane_bcast(dSv[r & 3], 16 * (r >> 2) + i), a = lane_bcast(av[r & 3], 16 * (r >> 2) + i);
      dk += qv[i] * dS;
      dx += g[i] * a;
      dqa[i] += k * dS;
    }
    if (tw + r < T) {
      const size_t row = (size_t)v * T + tw + r;
      stx<HF>(b.dz, row * D + 4 * lane, dk * (one - k * k));
      if (b.dxd) stx<HF>(b.dxd, row * D + 4 * lane, dx);
    }
  }
  // ---- 5. dQ partial of the chunk: four waves summed in a fixed order ----------------------------------------------------
#pragma unroll
  for (int i = 0; i < NQT; ++i) st4(red + (wave * NQT + i) * D + 4 * lane, dqa[i]);
  __syncthreads();
  for (int e = tid; e < NQT * (D / 4); e += 256) {
    const int i = e / (D / 4), cq = e - i * (D / 4);
    f32x4 s = ld4(red + i * D + 4 * cq);
#pragma unroll
    for (int w = 1; w < 4; ++w) s += ld4(red + (w * NQT + i) * D + 4 * cq);
    st4(dq_part + (((size_t)v * nchunk + chunk) * NQT + i) * D + 4 * cq, s);
  }
}

constexpr int bwd_v2_occ(bool, int nq) { return nq == 1 ? 4 : 3; }
template <bool HF, int NQT>
__global__ __launch_bounds__(256, bwd_v2_occ(HF, NQT)) void attnpool_bwd_v2_kernel(const sdumc_attnpool_bwd_t b, float* dq_part, const int nchunk) {
  int v, chunk;
  v2_unit(b.f, blockIdx.x, nchunk, v, chunk);
  attnpool_bwd_v2<HF, NQT>(b, dq_part, nchunk, chunk, v);
}
template <bool HF, int NQT>
__global__ __launch_bounds__(256, bwd_v2_occ(HF, NQT)) void attnpool_bwd_v2_multi_kernel(const MultiBwd m) {
  const int s = site_of(m.wg_end, blockIdx.x);
  const int local = blockIdx.x - (s ? m.wg_end[s - 1] : 0);
  const int nchunk = m.nchunk[s];
  sdumc_attnpool_bwd_t b = m.b[0];     // uniform select, by value (see attnpool_bwd_multi_kernel)
  if (s == 1) b = m.b[1];
  if (s == 2) b = m.b[2];
  if (s == 3) b = m.b[3];
  int v, chunk;
  v2_unit(b.f, local, nchunk, v, chunk);
  attnpool_bwd_v2<HF, NQT>(b, static_cast<float*>(b.workspace), nchunk, chunk, v);
}

// ---- forward partial v2 ---------------------------------------------------------------------------------------------
template <bool HF, int NQT>
__device__ __forceinline__ void attn_fwd_partial_v2(const sdumc_attnpool p, float* ws, const int nchunk, const int chunk, const int v) {
  // fp32 rows: the 16 frame rows are requested when the score MFMAs have released the key tile's registers -- 104 instead of 168
  // VGPRs, four waves per SIMD instead of three (measured: 41.9 vs 45.0 us for the three Cross_Attention sites)
  constexpr bool LATE_X = !HF;
  constexpr int LDQ = D + 16;
  typedef typename RawRow<HF>::type raw_t;
  __shared__ __attribute__((aligned(16))) float q_s[MAXQ * LDQ];
  __shared__ __attribute__((aligned(16))) float red[4 * NQT * D];
  __shared__ float wstat[4][16];
  __shared__ float cstat[16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kk = lane >> 4;
  const int T = p.T;
  const int t0 = chunk * CH, tw = t0 + 16 * wave;
  const FwdWs w = fwd_ws(ws, p.V, nchunk, NQT, D);
  const int Tv = p.lengths ? min(T, max(1, p.lengths[v])) : T;
  const DropRT xd = drop_resolve(p.x_drop);
  const bool masked = xd.enabled != 0;
  const int vx = v % p.x_samples;
  // ---- 1. every global load: the queries (<= 2 quads per thread), the 16 key rows as the MFMA A operand, the 16 frame rows
  f32x4 qq[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 256 * u, i = e >> 6, cq = e & 63;
    qq[u] = (NQT * 64 > 256 * u && i < NQT) ? ld4(p.q + (size_t)v * p.q_stride + (size_t)i * D + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  raw_t ka[16], xr[16];
  uint32_t xm[16];
  {
    const size_t ro = ((size_t)v * T + min(tw + r16, T - 1)) * D + 4 * kk;
#pragma unroll
    for (int j = 0; j < 16; ++j) ka[j] = ldraw<HF>(p.keys, ro + 16 * j);
  }
  auto load_x = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = min(tw + r, T - 1);                  // rows beyond T re-read the last row; their weight is 0
      xr[r] = ldraw<HF>(p.x, ((size_t)vx * T + t) * D + 4 * lane);
      xm[r] = (!HF && masked) ? xd.bits[((size_t)v * T + t) * (D / 4) + lane] : 0xfu;
    }
  };
  if constexpr (!LATE_X) load_x();
  // ---- 2. scores of the wave's 16 rows against the queries -----------------------------------------------------------------
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 256 * u, i = e >> 6, cq = e & 63;
    if (i < MAXQ) st4(q_s + i * LDQ + 4 * cq, qq[u]);
  }
  __syncthreads();
  f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const f32x4 a = cvtraw<HF>(ka[j]);
    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
    if (r16 < MAXQ) bq = ld4(q_s + r16 * LDQ + 16 * j + 4 * kk);
#pragma unroll
    for (int e = 0; e < 4; ++e) s4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], bq[e], s4, 0, 0, 0);
  }
  if constexpr (LATE_X) load_x();
  // C layout: column (query) = lane & 15, rows = 16 wave + 4 (lane >> 4) + e
  float s[4], mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int t = tw + 4 * kk + e;
    s[e] = t < Tv ? p.scale * s4[e] : -INFINITY;
    mx = fmaxf(mx, s[e]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  if (lane < 16) wstat[wave][lane] = mx;
  __syncthreads();
  if (tid < 16) cstat[tid] = fmaxf(fmaxf(wstat[0][tid], wstat[1][tid]), fmaxf(wstat[2][tid], wstat[3][tid]));
  __syncthreads();
  const float mc = cstat[r16];          // chunk max of my query column (-inf only for a chunk wholly beyond Tv)
  float pe[4], lsum = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int t = tw + 4 * kk + e;
    pe[e] = t < Tv ? expf(s[e] - mc) : 0.f;
    lsum += pe[e];
    if (r16 < NQT && t < T) p.attn[((size_t)v * T + t) * NQT + r16] = pe[e];      // normalised by the combine step
  }
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  __syncthreads();                      // cstat reads done
  if (lane < 16) wstat[wave][lane] = lsum;
  __syncthreads();
  if (tid < NQT) {
    float* st = w.stats + ((size_t)v * nchunk + chunk) * 2 * MAXQ;
    st[tid] = cstat[tid];
    st[MAXQ + tid] = wstat[0][tid] + wstat[1][tid] + wstat[2][tid] + wstat[3][tid];
  }
  // ---- 3. unnormalised pooling of the chunk: lane = 4 channels, the row's weights by v_readlane ---------------------------
  f32x4 acc[NQT];
#pragma unroll
  for (int i = 0; i < NQT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float mscale = (!HF && masked) ? xd.scale : 1.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    f32x4 x = cvtraw<HF>(xr[r]);
    if (!HF && masked) {
      const uint32_t bb = xm[r];
      x[0] = (bb & 1u) ? x[0] * mscale : 0.f;
      x[1] = (bb & 2u) ? x[1] * mscale : 0.f;
      x[2] = (bb & 4u) ? x[2] * mscale : 0.f;
      x[3] = (bb & 8u) ? x[3] * mscale : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NQT; ++i) acc[i] += x * lane_bcast(pe[r & 3], 16 * (r >> 2) + i);     // rows beyond T: weight 0
  }
#pragma unroll
  for (int i = 0; i < NQT; ++i) st4(red + (wave * NQT + i) * D + 4 * lane, acc[i]);
  __syncthreads();
  for (int e = tid; e < NQT * (D / 4); e += 256) {
    const int i = e / (D / 4), cq = e - i * (D / 4);
    f32x4 sum = ld4(red + i * D + 4 * cq);
#pragma unroll
    for (int ww = 1; ww < 4; ++ww) sum += ld4(red + (ww * NQT + i) * D + 4 * cq);
    st4(w.part + (((size_t)v * nchunk + chunk) * NQT + i) * D + 4 * cq, sum);
  }
}

template <bool HF, int NQT>
__global__ __launch_bounds__(256, 4) void attn_fwd_partial_v2_kernel(const sdumc_attnpool p, float* ws, const int nchunk) {
  int v, chunk;
  v2_unit(p, blockIdx.x, nchunk, v, chunk);
  attn_fwd_partial_v2<HF, NQT>(p, ws, nchunk, chunk, v);
}
template <bool HF, int NQT>
__global__ __launch_bounds__(256, 4) void attn_fwd_partial_v2_multi_kernel(const MultiFwd m) {
  const int s = site_of(m.wg_end, blockIdx.x);
  const int local = blockIdx.x - (s ? m.wg_end[s - 1] : 0);
  const int nchunk = m.nchunk[s];
  sdumc_attnpool p = m.p[0];
  if (s == 1) p = m.p[1];
  if (s == 2) p = m.p[2];
  if (s == 3) p = m.p[3];
  int v, chunk;
  v2_unit(p, local, nchunk, v, chunk);
  attn_fwd_partial_v2<HF, NQT>(p, static_cast<float*>(p.workspace), nchunk, chunk, v);
}

// the v2 kernels take: 256-channel rows, nq = 1 or 7, keep-bits (or no input mask), the two-pass combine; everything else (wider
// rows, Philox masks, tickets) runs on the round-3 kernels.  sdumc_attnpool_set_v2_(0): the bit-for-bit comparison of the tests
int g_v2 = 1;
bool v2_on() { return g_v2 != 0; }
bool v2_takes(const sdumc_attnpool& p) {
  return v2_on() && row_dim(p) == D && (p.nq == 1 || p.nq == 7) && !p.tickets && !(p.x_drop.enabled && !p.x_drop.bits) &&
         (!p.bf16 || !p.x_drop.enabled);
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

// shared query (FRA2UTT_new's context vector): sum of the per-chunk slabs of EVERY sample, one workgroup per 64 quads of the
// [nq, dim] gradient: 16 row groups x 64 quads, a group walks rows g, g + 16, .. eight loads at a time, the groups meet in LDS
// in ascending order -- one launch where dq_reduce + two column-sum launches (a 4-workgroup first stage of 25-38 us inside the
// step, profiles/r3x_timeline_fp32.txt) used to sit on every modality's lane in front of its input-gradient launch
__global__ __launch_bounds__(1024) void dq_sum_kernel(const float* part, float* dq_sum, const int rows /* V * nchunk */,
                                                      const int per_v /* nq * dim */) {
  __shared__ __attribute__((aligned(16))) float red[16][256];
  const int cq = threadIdx.x & 63, g = threadIdx.x >> 6;
  const size_t col = (size_t)blockIdx.x * 256 + 4 * cq;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int r = g;
  for (; r + 7 * 16 < rows; r += 8 * 16) {
    f32x4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = ld4(part + (size_t)(r + 16 * u) * per_v + col);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += t[u];
  }
  for (; r < rows; r += 16) acc += ld4(part + (size_t)r * per_v + col);
  st4(&red[g][4 * cq], acc);
  __syncthreads();
  if (g == 0) {
    f32x4 s = ld4(&red[0][4 * cq]);
#pragma unroll
    for (int k = 1; k < 16; ++k) s += ld4(&red[k][4 * cq]);
    st4(dq_sum + col, s);
  }
}

__global__ __launch_bounds__(256) void dq_reduce_multi_kernel(const MultiBwd m) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  int s = 0;
#pragma unroll
  for (int i = 0; i < MAXSITES - 1; ++i) s += idx >= m.dq_end[i] ? 1 : 0;
  if (idx >= m.dq_end[MAXSITES - 1]) return;
  const sdumc_attnpool_bwd_t& b = m.b[s];
  const size_t local = idx - (s ? m.dq_end[s - 1] : 0);
  const int nchunk = m.nchunk[s], per_v = b.f.nq * D;
  const size_t v = local / per_v, e = local - v * per_v;
  const float* src = static_cast<const float*>(b.workspace) + v * nchunk * per_v + e;
  float a = 0.f;
  for (int c = 0; c < nchunk; ++c) a += src[(size_t)c * per_v];
  b.dq[local] = a;
}

int check(const sdumc_attnpool& p) {
  if (p.V <= 0 || p.T <= 0 || p.nq < 1 || p.nq > MAXQ || p.x_samples <= 0) return SDUMC_EINVAL;
  if (p.dim != 0 && p.dim != 256 && p.dim != 512 && p.dim != 768 && p.dim != 1024) return SDUMC_EINVAL;
  if (!p.x || !p.keys || !p.q || !p.attn || !p.pooled || !p.out) return SDUMC_EINVAL;
  if ((p.T + CH - 1) / CH > 4096) return SDUMC_EINVAL;
  if (p.partial_only && p.tickets) return SDUMC_EINVAL;
  return SDUMC_OK;
}

}  // namespace

extern "C" int sdumc_attnpool_set_v2_(int on) {
  g_v2 = on ? 1 : 0;
  return SDUMC_OK;
}

extern "C" size_t sdumc_attnpool_fwd_workspace_bytes_dim(int32_t V, int32_t T, int32_t nq, int32_t dim) {
  const size_t nchunk = (size_t)(T + CH - 1) / CH;
  return ((size_t)V * nchunk * nq * (dim > 0 ? dim : D) + (size_t)V * nchunk * 2 * MAXQ) * sizeof(float);
}
extern "C" size_t sdumc_attnpool_fwd_workspace_bytes(int32_t V, int32_t T, int32_t nq) {
  return sdumc_attnpool_fwd_workspace_bytes_dim(V, T, nq, D);
}

extern "C" int sdumc_attnpool_fwd(const sdumc_attnpool* pp, void* stream) {
  if (!pp) return SDUMC_EINVAL;
  const sdumc_attnpool& p = *pp;
  int rc = check(p);
  if (rc) return rc;
  if (!p.workspace || p.workspace_bytes < sdumc_attnpool_fwd_workspace_bytes_dim(p.V, p.T, p.nq, row_dim(p))) return SDUMC_ENOMEM;
  hipStream_t st = as_stream(stream);
  const int nchunk = (p.T + CH - 1) / CH;
  const bool philox = p.x_drop.enabled && !p.x_drop.bits;
  const dim3 grid(nchunk, p.V), blk(256);
  if (v2_takes(p)) {
    const dim3 g1(nchunk * p.V);
    if (p.bf16) {
      if (p.nq == 1) hipLaunchKernelGGL((attn_fwd_partial_v2_kernel<true, 1>), g1, blk, 0, st, p, static_cast<float*>(p.workspace), nchunk);
      else hipLaunchKernelGGL((attn_fwd_partial_v2_kernel<true, 7>), g1, blk, 0, st, p, static_cast<float*>(p.workspace), nchunk);
    } else {
      if (p.nq == 1) hipLaunchKernelGGL((attn_fwd_partial_v2_kernel<false, 1>), g1, blk, 0, st, p, static_cast<float*>(p.workspace), nchunk);
      else hipLaunchKernelGGL((attn_fwd_partial_v2_kernel<false, 7>), g1, blk, 0, st, p, static_cast<float*>(p.workspace), nchunk);
    }
    SDUMC_CHECK_LAUNCH();
    if (p.partial_only) return SDUMC_OK;      // the caller combines the chunks
    hipLaunchKernelGGL(attn_fwd_combine_kernel, dim3(p.V), dim3(256), (size_t)nchunk * MAXQ * sizeof(float), st, p,
                       p.workspace, nchunk);
    SDUMC_CHECK_LAUNCH();
    return SDUMC_OK;
  }
#define FWD_PARTIAL(PH, CC) hipLaunchKernelGGL((attn_fwd_partial_kernel<PH, CC>), grid, blk, 0, st, p, p.workspace, nchunk)
  if (p.bf16) {      // bf16 frames: the engine's bf16-storage mode (256 channels, masks pre-applied or keep-bits attached)
    if (row_dim(p) != D || philox) return SDUMC_EINVAL;
    hipLaunchKernelGGL((attn_fwd_partial_kernel<false, 1, true>), grid, blk, 0, st, p, p.workspace, nchunk);
  } else
  switch (row_dim(p) / D) {
    case 1: if (philox) FWD_PARTIAL(true, 1); else FWD_PARTIAL(false, 1); break;
    case 2: if (philox) FWD_PARTIAL(true, 2); else FWD_PARTIAL(false, 2); break;
    case 3: if (philox) FWD_PARTIAL(true, 3); else FWD_PARTIAL(false, 3); break;
    default: if (philox) FWD_PARTIAL(true, 4); else FWD_PARTIAL(false, 4); break;
  }
#undef FWD_PARTIAL
  SDUMC_CHECK_LAUNCH();
  if (p.tickets || p.partial_only) return SDUMC_OK;          // the combine ran inside the partial kernel / is the caller's
  hipLaunchKernelGGL(attn_fwd_combine_kernel, dim3(p.V), dim3(256), (size_t)nchunk * MAXQ * sizeof(float), st, p,
                     p.workspace, nchunk);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

extern "C" size_t sdumc_attnpool_bwd_workspace_bytes_dim(int32_t V, int32_t T, int32_t nq, int32_t dim) {
  const size_t nchunk = (size_t)(T + CH - 1) / CH;
  return (size_t)V * nchunk * nq * (dim > 0 ? dim : D) * sizeof(float);
}
extern "C" size_t sdumc_attnpool_bwd_workspace_bytes(int32_t V, int32_t T, int32_t nq) {
  return sdumc_attnpool_bwd_workspace_bytes_dim(V, T, nq, D);
}

extern "C" int sdumc_attnpool_bwd(const sdumc_attnpool_bwd_t* bp, void* stream) {
  if (!bp) return SDUMC_EINVAL;
  const sdumc_attnpool_bwd_t& b = *bp;
  int rc = check(b.f);
  if (rc) return rc;
  if (!b.dout || !b.dz || (!b.dq && !b.dq_sum) || !b.workspace) return SDUMC_EINVAL;
  const sdumc_attnpool& p = b.f;
  if ((!b.dxd || b.dout_masked) && !v2_takes(p)) return SDUMC_EINVAL;      // (dxd left to the consumer: the wavefront-tiled kernels only)
  if (b.dq_sum && (p.q_stride != 0 || p.tickets)) return SDUMC_EINVAL;      // the sum over samples is the gradient of a SHARED query
  const int nchunk = (p.T + CH - 1) / CH;
  const int DD = row_dim(p);
  if (b.workspace_bytes < sdumc_attnpool_bwd_workspace_bytes_dim(p.V, p.T, p.nq, DD)) return SDUMC_ENOMEM;
  hipStream_t st = as_stream(stream);
  const dim3 grid(nchunk, p.V), blk(256);
  if (v2_takes(p)) {
    const dim3 g1(nchunk * p.V);
    float* wsf = static_cast<float*>(b.workspace);
    if (p.bf16) {
      if (p.nq == 1) hipLaunchKernelGGL((attnpool_bwd_v2_kernel<true, 1>), g1, blk, 0, st, b, wsf, nchunk);
      else hipLaunchKernelGGL((attnpool_bwd_v2_kernel<true, 7>), g1, blk, 0, st, b, wsf, nchunk);
    } else {
      if (p.nq == 1) hipLaunchKernelGGL((attnpool_bwd_v2_kernel<false, 1>), g1, blk, 0, st, b, wsf, nchunk);
      else hipLaunchKernelGGL((attnpool_bwd_v2_kernel<false, 7>), g1, blk, 0, st, b, wsf, nchunk);
    }
  } else if (p.bf16) {
    if (DD != D || (p.x_drop.enabled && !p.x_drop.bits)) return SDUMC_EINVAL;
    hipLaunchKernelGGL((attnpool_bwd_kernel<1, true>), grid, blk, 0, st, b, b.workspace, nchunk);
  } else
  switch (DD / D) {
    case 1: hipLaunchKernelGGL(attnpool_bwd_kernel<1>, grid, blk, 0, st, b, b.workspace, nchunk); break;
    case 2: hipLaunchKernelGGL(attnpool_bwd_kernel<2>, grid, blk, 0, st, b, b.workspace, nchunk); break;
    case 3: hipLaunchKernelGGL(attnpool_bwd_kernel<3>, grid, blk, 0, st, b, b.workspace, nchunk); break;
    default: hipLaunchKernelGGL(attnpool_bwd_kernel<4>, grid, blk, 0, st, b, b.workspace, nchunk); break;
  }
  SDUMC_CHECK_LAUNCH();
  if (b.dq_sum) {                          // shared query: every sample's slabs summed in one launch
    hipLaunchKernelGGL(dq_sum_kernel, dim3((unsigned)(p.nq * DD / 256)), dim3(1024), 0, st, b.workspace, b.dq_sum, p.V * nchunk, p.nq * DD);
    SDUMC_CHECK_LAUNCH();
    return SDUMC_OK;
  }
  if (p.tickets) return SDUMC_OK;          // the slabs were summed inside the kernel
  const size_t total = (size_t)p.V * p.nq * DD;
  hipLaunchKernelGGL(dq_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, b.workspace, b.dq,
                     nchunk, p.nq * DD, total);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

// ---- several sites in one launch -----------------------------------------------------------------
// The three Cross_Attention blocks of a step are independent, small (17-42 us each at C2) and sit between two utterance-level
// launches on the caller's stream: forking them onto side streams costs ~20 us of cross-queue event latency on the way out and
// ~12 us on the way back (tools/step_marks.py), one grouped launch costs neither.
extern "C" int sdumc_attnpool_fwd_multi(const sdumc_attnpool* ps, int32_t n, void* stream) {
  if (!ps || n < 1 || n > MAXSITES) return SDUMC_EINVAL;
  MultiFwd m;
  memset(&m, 0, sizeof(m));
  int wg = 0, vs = 0, max_chunk = 0;
  for (int i = 0; i < MAXSITES; ++i) {
    if (i < n) {
      const sdumc_attnpool& p = ps[i];
      int rc = check(p);
      if (rc) return rc;
      if (row_dim(p) != D || (p.x_drop.enabled && !p.x_drop.bits) || p.bf16 != ps[0].bf16 || p.partial_only != ps[0].partial_only) return SDUMC_EINVAL;
      if ((p.tickets != nullptr) != (ps[0].tickets != nullptr)) return SDUMC_EINVAL;
      if (!p.workspace || p.workspace_bytes < sdumc_attnpool_fwd_workspace_bytes_dim(p.V, p.T, p.nq, D)) return SDUMC_ENOMEM;
      m.p[i] = p;
      m.nchunk[i] = (p.T + CH - 1) / CH;
      wg += m.nchunk[i] * p.V;
      vs += p.V;
      max_chunk = m.nchunk[i] > max_chunk ? m.nchunk[i] : max_chunk;
    }
    m.wg_end[i] = wg;
    m.v_end[i] = vs;
  }
  hipStream_t st = as_stream(stream);
  bool v2 = true;
  for (int i = 0; i < n; ++i) v2 = v2 && v2_takes(ps[i]) && ps[i].nq == ps[0].nq;
  if (v2) {
    if (ps[0].bf16) {
      if (ps[0].nq == 1) hipLaunchKernelGGL((attn_fwd_partial_v2_multi_kernel<true, 1>), dim3(wg), dim3(256), 0, st, m);
      else hipLaunchKernelGGL((attn_fwd_partial_v2_multi_kernel<true, 7>), dim3(wg), dim3(256), 0, st, m);
    } else {
      if (ps[0].nq == 1) hipLaunchKernelGGL((attn_fwd_partial_v2_multi_kernel<false, 1>), dim3(wg), dim3(256), 0, st, m);
      else hipLaunchKernelGGL((attn_fwd_partial_v2_multi_kernel<false, 7>), dim3(wg), dim3(256), 0, st, m);
    }
  } else if (ps[0].bf16) hipLaunchKernelGGL(attn_fwd_partial_multi_kernel<true>, dim3(wg), dim3(256), 0, st, m);
  else hipLaunchKernelGGL(attn_fwd_partial_multi_kernel<false>, dim3(wg), dim3(256), 0, st, m);
  SDUMC_CHECK_LAUNCH();
  if (ps[0].tickets || ps[0].partial_only) return SDUMC_OK;
  hipLaunchKernelGGL(attn_fwd_combine_multi_kernel, dim3(vs), dim3(256), (size_t)max_chunk * MAXQ * sizeof(float), st, m);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

extern "C" int sdumc_attnpool_bwd_multi(const sdumc_attnpool_bwd_t* bs, int32_t n, void* stream) {
  if (!bs || n < 1 || n > MAXSITES) return SDUMC_EINVAL;
  MultiBwd m;
  memset(&m, 0, sizeof(m));
  int wg = 0;
  unsigned long long dq = 0;
  for (int i = 0; i < MAXSITES; ++i) {
    if (i < n) {
      const sdumc_attnpool_bwd_t& b = bs[i];
      const sdumc_attnpool& p = b.f;
      int rc = check(p);
      if (rc) return rc;
      if (!b.dout || !b.dz || !b.dq || !b.workspace || b.dq_sum || p.partial_only != bs[0].f.partial_only) return SDUMC_EINVAL;
      if (row_dim(p) != D || (p.x_drop.enabled && !p.x_drop.bits) || p.bf16 != bs[0].f.bf16) return SDUMC_EINVAL;
      if ((p.tickets != nullptr) != (bs[0].f.tickets != nullptr)) return SDUMC_EINVAL;
      if (b.workspace_bytes < sdumc_attnpool_bwd_workspace_bytes_dim(p.V, p.T, p.nq, D)) return SDUMC_ENOMEM;
      m.b[i] = b;
      m.nchunk[i] = (p.T + CH - 1) / CH;
      wg += m.nchunk[i] * p.V;
      dq += (unsigned long long)p.V * p.nq * D;
    }
    m.wg_end[i] = wg;
    m.dq_end[i] = dq;
  }
  hipStream_t st = as_stream(stream);
  bool v2 = true;
  for (int i = 0; i < n; ++i) v2 = v2 && v2_takes(bs[i].f) && bs[i].f.nq == bs[0].f.nq;
  for (int i = 0; i < n && !v2; ++i)
    if (!bs[i].dxd || bs[i].dout_masked) return SDUMC_EINVAL;      // (dxd left to the consumer: the wavefront-tiled kernels only)
  if (v2) {
    if (bs[0].f.bf16) {
      if (bs[0].f.nq == 1) hipLaunchKernelGGL((attnpool_bwd_v2_multi_kernel<true, 1>), dim3(wg), dim3(256), 0, st, m);
      else hipLaunchKernelGGL((attnpool_bwd_v2_multi_kernel<true, 7>), dim3(wg), dim3(256), 0, st, m);
    } else {
      if (bs[0].f.nq == 1) hipLaunchKernelGGL((attnpool_bwd_v2_multi_kernel<false, 1>), dim3(wg), dim3(256), 0, st, m);
      else hipLaunchKernelGGL((attnpool_bwd_v2_multi_kernel<false, 7>), dim3(wg), dim3(256), 0, st, m);
    }
  } else if (bs[0].f.bf16) hipLaunchKernelGGL(attnpool_bwd_multi_kernel<true>, dim3(wg), dim3(256), 0, st, m);
  else hipLaunchKernelGGL(attnpool_bwd_multi_kernel<false>, dim3(wg), dim3(256), 0, st, m);
  SDUMC_CHECK_LAUNCH();
  if (bs[0].f.tickets || bs[0].f.partial_only) return SDUMC_OK;      // (partial_only: the caller sums the per-chunk dq slabs of `workspace`)
  hipLaunchKernelGGL(dq_reduce_multi_kernel, dim3((unsigned)((dq + 255) / 256)), dim3(256), 0, st, m);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

// ====================================================================================================================
// K3 -- the UMCA forward as ONE kernel per (64-frame chunk, virtual sample)   (Cross_Attention, model :79-95;
// FRA2UTT_new, model :56-68, is the nq = 1 case):
//     xd = drop(x)                       the frames, masked by the keep-bits, staged ONCE per k-tile through LDS (LDS-DMA ring)
//     K  = tanh(xd W_in^T + b)           fp32 MFMA 32x32x2, the whole 64 x 256 key tile of the chunk stays on chip
//     S  = 0.3 K Q'^T, softmax partials  MFMA 16x16x4 on the LDS key tile, chunk max / sum (flash-style partials)
//     O_c = P^T xd                       the chunk's unnormalised pooled rows; sdumc's per-sample combine finishes the softmax
// The keys never travel to HBM and back between the projection and the scores; they are WRITTEN (p.a.keys) only when the caller
// wants them for the backward (training), and not at all in inference.
// Main loop = gemm_wide.hip's NT loop at its 64 x 256 x 16 configuration (3-deep LDS-DMA ring, XOR-swizzled 16-byte chunks,
// counted vmcnt, one raw barrier per k-tile); epilogue = attn_fwd_partial_body on the LDS key tile.
// LDS: max(ring 62 208 B, key tile 66 560 B) + query tile 8 704 B + 2.4 KB of statistics = 77.7 KB -> two workgroups per CU.
// ====================================================================================================================
namespace sdumc_k3 {

typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int waitcnt_vm(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0xF << 8); }
constexpr int BM = 64, BN = 256, BK = 16, NST = 3, NW = 4;
constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 4, BITS_LDS = 256;
constexpr int A_CH = A_BYTES / 1024, B_CH = B_BYTES / 1024;
[[maybe_unused]] constexpr int NI = (A_CH + B_CH) / NW;
constexpr int STAGE_BYTES = A_BYTES + B_BYTES + BITS_LDS, RING_BYTES = NST * STAGE_BYTES;
constexpr int K_BYTES = BM * UMCA_LDK * 4;
constexpr int TOP_BYTES = K_BYTES > RING_BYTES ? K_BYTES : RING_BYTES;
constexpr int LDQ = D + 16, Q_BYTES = MAXQ * LDQ * 4;
constexpr int LDS_BYTES = TOP_BYTES + Q_BYTES;
static_assert(4 * MAXQ * D * 4 <= TOP_BYTES, "the pooling reduce overlays the key tile");

__device__ __forceinline__ int swz16(int row) { return (row >> 2) & 3; }   // gemm_wide.hip swz<16>
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }   // = gemm_wide.hip

// SPLIT: the key projection's fp32 products on the bf16 matrix pipe from exactly split operands (gemm_group.hip, "fp32 products
// on the bf16 pipe", has the arithmetic; sdumc_hip.h: sdumc_set_split_).
template <bool MASK, bool SPLIT>
__device__ __forceinline__ void umca_fwd_body(const sdumc_umca& u, const int nchunk, char* lds) {
#if defined(__HIP_DEVICE_COMPILE__)
  const sdumc_attnpool& p = u.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wn0 = wave * 64;
  const int chunk = blockIdx.x, v = blockIdx.y;
  const int T = p.T, t0 = chunk * CH, vx = v % p.x_samples;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0,
      (int)min((size_t)p.x_samples * T * D * 4, (size_t)0xFFFFFFF0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(u.w_in), 0, D * D * 4, 0x00020000);
  uint32_t voff[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int piece = wave + i * NW;
    const int q = ((piece < A_CH ? piece : piece - A_CH) << 6) + lane;       // 16-byte chunk index inside the tile
    const int row = q >> 2, cp = q & 3, c = cp ^ swz16(row);
    if (piece < A_CH) voff[i] = ((uint32_t)(vx * T + min(t0 + row, T - 1)) * (uint32_t)D + (uint32_t)(4 * c)) * 4u;   // rows beyond T: the last row
    else voff[i] = ((uint32_t)row * (uint32_t)D + (uint32_t)(4 * c)) * 4u;
  }
  const uint8_t* bitsp = MASK ? p.x_drop.bits : nullptr;
  const uint32_t qw = (p.x_drop.width + 3u) >> 2;
  const float mscale = MASK ? p.x_drop.scale : 1.f;
  const __amdgpu_buffer_rsrc_t rbits = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(MASK ? bitsp : (const uint8_t*)p.x), 0,
      MASK ? (int)min((size_t)p.V * T * qw, (size_t)0xFFFFFFF0u) : 0, 0x00020000);
  uint32_t bvoff = 0;
  if constexpr (MASK) bvoff = (uint32_t)(v * T + min(t0 + lane, T - 1)) * qw;     // one dword (16 channels) per row and k-tile

  auto issue = [&](int buf) {
    char* base = lds + buf * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int piece = wave + i * NW;
      const bool isA = piece < A_CH;
      char* dst = isA ? base + piece * 1024 : base + A_BYTES + (piece - A_CH) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? ra : rb, (lds_void_t*)dst, 16, voff[i], 0, 0, 0);
      voff[i] += BK * 4;
    }
    if constexpr (MASK) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rbits, (lds_void_t*)(base + A_BYTES + B_BYTES), 4, bvoff, 0, 0, 0);
      bvoff += BK / 4;
    }
  };
  constexpr int PER = NI + (MASK ? 1 : 0);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto read_a = [&](const char* base, int gq, f32x4 (&af)[2]) {
    const float* As = reinterpret_cast<const float*>(base);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 32 * i + li;
      af[i] = *reinterpret_cast<const f32x4*>(As + row * BK + 4 * ((2 * gq + lh) ^ swz16(row)));
      if constexpr (MASK) {
        const uint32_t b = reinterpret_cast<const uint8_t*>(base + A_BYTES + B_BYTES)[row * (BK / 4) + 2 * gq + lh];
        af[i][0] = (b & 1u) ? af[i][0] * mscale : 0.f;
        af[i][1] = (b & 2u) ? af[i][1] * mscale : 0.f;
        af[i][2] = (b & 4u) ? af[i][2] * mscale : 0.f;
        af[i][3] = (b & 8u) ? af[i][3] * mscale : 0.f;
      }
    }
  };
  auto read_b = [&](const char* base, int gq, f32x4 (&bf)[2]) {
    const float* Bs = reinterpret_cast<const float*>(base + A_BYTES);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wn0 + 32 * j + li;
      bf[j] = *reinterpret_cast<const f32x4*>(Bs + row * BK + 4 * ((2 * gq + lh) ^ swz16(row)));
    }
  };
  auto compute = [&](const char* base) {
    f32x4 af[2][2], bf[2][2];
    read_a(base, 0, af[0]);
    read_b(base, 0, bf[0]);
#pragma unroll
    for (int gq = 0; gq < BK / 8; ++gq) {
      const int cur = gq & 1, nxt = cur ^ 1;
      if (gq + 1 < BK / 8) {
        read_a(base, gq + 1, af[nxt]);
        read_b(base, gq + 1, bf[nxt]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][s], bf[cur][j][s], acc[i][j], 0, 0, 0);
    }
  };

  // the k-tile (16 k) as ONE bf16 MFMA depth: a lane's eight k of an operand are its two fp32 fragments side by side
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2s __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  auto pk = [](float x, float y) -> uint32_t {       // v_cvt_pk_bf16_f32 (round to nearest even), low half = x
    const f32x2s v = {x, y};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  };
  auto split8 = [&](const f32x4 lo, const f32x4 hi, u32x4* pl) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float x = d < 2 ? lo[2 * d] : hi[2 * d - 4], y = d < 2 ? lo[2 * d + 1] : hi[2 * d - 3];
      const uint32_t p0 = pk(x, y);
      const float x1 = x - __uint_as_float(p0 << 16), y1 = y - __uint_as_float(p0 & 0xFFFF0000u);       // exact
      const uint32_t p1 = pk(x1, y1);
      const float x2 = x1 - __uint_as_float(p1 << 16), y2 = y1 - __uint_as_float(p1 & 0xFFFF0000u);     // exact
      pl[0][d] = p0;
      pl[1][d] = p1;
      pl[2][d] = pk(x2, y2);
    }
  };
  auto compute_split = [&](const char* base) {
    f32x4 af[2][2], bf[2][2];
    read_a(base, 0, af[0]);
    read_a(base, 1, af[1]);
    read_b(base, 0, bf[0]);
    read_b(base, 1, bf[1]);
    u32x4 pa[2][3], pb[2][3];
    auto op = [](const u32x4& v) { return __builtin_bit_cast(bf16x8, v); };
#pragma unroll
    for (int j = 0; j < 2; ++j) split8(bf[0][j], bf[1][j], pb[j]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      split8(af[0][i], af[1][i], pa[i]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {   // smallest terms first
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(pa[i][2]), op(pb[j][0]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(pa[i][0]), op(pb[j][2]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(pa[i][1]), op(pb[j][1]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(pa[i][1]), op(pb[j][0]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(pa[i][0]), op(pb[j][1]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(pa[i][0]), op(pb[j][0]), acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- key projection: 16 k-tiles through the ring ----
  constexpr int nk = D / BK;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) issue(s);
  int buf = 0, ibuf = NST - 1;
  for (int t = 0; t < nk; ++t) {
    if (t + NST - 2 < nk) __builtin_amdgcn_s_waitcnt(waitcnt_vm((NST - 2) * PER));
    else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    __builtin_amdgcn_s_barrier();
    if (t + NST - 1 < nk) issue(ibuf);
    if constexpr (SPLIT) compute_split(lds + buf * STAGE_BYTES);
    else compute(lds + buf * STAGE_BYTES);
    buf = buf + 1 == NST ? 0 : buf + 1;
    ibuf = ibuf + 1 == NST ? 0 : ibuf + 1;
  }
  __syncthreads();      // every wave has read its last fragments: the ring's memory becomes the key tile

  // ---- K = tanh(acc + b): to the LDS key tile (and to HBM when the caller keeps the keys for the backward) ----
  float* K_s = reinterpret_cast<float*>(lds);
  float* q_s = reinterpret_cast<float*>(lds + TOP_BYTES);
  float* keys_out = const_cast<float*>(p.keys);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wn0 + 32 * j + li;
      const float bv = u.b_in[col];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const float kv = fast_tanh(acc[i][j][e] + bv);
        K_s[row * UMCA_LDK + col] = kv;
        if (keys_out && t0 + row < T) keys_out[((size_t)v * T + t0 + row) * D + col] = kv;
      }
    }
  __syncthreads();
  attn_fwd_partial_body<false, 1, false, true>(p, static_cast<float*>(p.workspace), nchunk, chunk, v, K_s, q_s, K_s);
#endif
}
template <bool MASK>
__global__ __launch_bounds__(256, 2) void umca_fwd_kernel(const sdumc_umca u, const int nchunk) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  umca_fwd_body<MASK, false>(u, nchunk, lds);
}
template <bool MASK>
__global__ __launch_bounds__(256, 2) void umca_fwd_split_kernel(const sdumc_umca u, const int nchunk) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  umca_fwd_body<MASK, true>(u, nchunk, lds);
}

// The same kernel with the key projection on operands split ONCE PER TENSOR (gemm_p3.hip; round 5): A = the projected frames' bf16
// planes (written by the frame projection's epilogue), B = the fragment-major planes of W_in, straight into registers -- the k-loop
// is p3_mainloop (p3_loop.h) on a 64 x 256 tile, 256 threads, two workgroups per CU as before; the pooling part is unchanged and
// still reads the fp32 frames.  No split of x or W per workgroup per k-tile: the in-kernel form spent ~10 VALU instructions per MFMA.
using K3P3Cfg_m = sdumc_p3::PCfg<64, 4, true, 4>;
using K3P3Cfg_n = sdumc_p3::PCfg<64, 4, false, 4>;
static_assert(K3P3Cfg_m::LDS_BYTES <= TOP_BYTES, "the P3 ring lies inside the key tile's memory");
template <bool MASK>
__global__ __launch_bounds__(256, 2) void umca_fwd_p3_kernel(const sdumc_umca u, const int nchunk) {
#if defined(__HIP_DEVICE_COMPILE__)
  using CF = std::conditional_t<MASK, K3P3Cfg_m, K3P3Cfg_n>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const sdumc_attnpool& p = u.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wn0 = wave * 64;
  const int chunk = blockIdx.x, v = blockIdx.y;
  const int T = p.T, t0 = chunk * CH, vx = v % p.x_samples;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(u.x_p3), 0,
      (int)min((size_t)p.x_samples * T * D * 6, (size_t)0xFFFFFFF0u), 0x00020000);
  const uint32_t qw = (p.x_drop.width + 3u) >> 2;
  const __amdgpu_buffer_rsrc_t rbits = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(MASK ? p.x_drop.bits : static_cast<const uint8_t*>(u.x_p3)), 0,
      MASK ? (int)min((size_t)p.V * T * qw, (size_t)0xFFFFFFF0u) : 0, 0x00020000);
  f32x16 acc[2][2];
  sdumc_p3::p3_mainloop<CF>(lds, ra, (int64_t)D * 6, [&](int row) { return vx * T + min(t0 + row, T - 1); },      // rows beyond T: the last row
                            rbits, (int)qw, [&](int row) { return v * T + min(t0 + row, T - 1); },
                            static_cast<const char*>(u.w_in_p3f) + (size_t)(wave * 2) * (size_t)((D / 16) * sdumc_p3::FRAG_KT), (size_t)((D / 16) * sdumc_p3::FRAG_KT),
                            0, D / sdumc_p3::BK, acc);
  __syncthreads();      // every wave has read its last fragments: the ring's memory becomes the key tile
  const float mscale = MASK ? p.x_drop.scale : 1.f;
  float* K_s = reinterpret_cast<float*>(lds);
  float* q_s = reinterpret_cast<float*>(lds + TOP_BYTES);
  float* keys_out = const_cast<float*>(p.keys);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wn0 + 32 * j + li;
      const float bv = u.b_in[col];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const float kv = fast_tanh(acc[i][j][e] * mscale + bv);
        K_s[row * UMCA_LDK + col] = kv;
        if (keys_out && t0 + row < T) keys_out[((size_t)v * T + t0 + row) * D + col] = kv;
      }
    }
  __syncthreads();
  attn_fwd_partial_body<false, 1, false, true>(p, static_cast<float*>(p.workspace), nchunk, chunk, v, K_s, q_s, K_s);
#endif
}

}  // namespace sdumc_k3

static inline uint32_t qw_of(const sdumc_dropout& d) { return (d.width + 3u) >> 2; }
extern "C" int sdumc_umca_fwd(const sdumc_umca* up, void* stream) {
  if (!up) return SDUMC_EINVAL;
  sdumc_umca u = *up;
  const sdumc_attnpool& p = u.a;
  if (p.V <= 0 || p.T <= 0 || p.nq < 1 || p.nq > MAXQ || p.x_samples <= 0) return SDUMC_EINVAL;
  if ((p.dim != 0 && p.dim != D) || p.bf16 || p.tickets) return SDUMC_EINVAL;       // 256-channel fp32 rows, two-pass combine
  if (!p.x || !p.q || !p.attn || !p.pooled || !p.out || !u.w_in || !u.b_in) return SDUMC_EINVAL;
  if (p.x_drop.enabled && !p.x_drop.bits) return SDUMC_EINVAL;                       // the input mask comes as keep-bits
  if ((p.T + CH - 1) / CH > 4096) return SDUMC_EINVAL;
  if (!p.workspace || p.workspace_bytes < sdumc_attnpool_fwd_workspace_bytes_dim(p.V, p.T, p.nq, D)) return SDUMC_ENOMEM;
  static sdumc_dev_once attr;
  if (sdumc_once_per_device(attr, [] {
        return sdumc_set_dyn_lds(&sdumc_k3::umca_fwd_kernel<true>, sdumc_k3::LDS_BYTES) && sdumc_set_dyn_lds(&sdumc_k3::umca_fwd_kernel<false>, sdumc_k3::LDS_BYTES) &&
               sdumc_set_dyn_lds(&sdumc_k3::umca_fwd_split_kernel<true>, sdumc_k3::LDS_BYTES) && sdumc_set_dyn_lds(&sdumc_k3::umca_fwd_split_kernel<false>, sdumc_k3::LDS_BYTES);
      }) != SDUMC_OK)
    return SDUMC_ELAUNCH;
  hipStream_t st = as_stream(stream);
  const int nchunk = (p.T + CH - 1) / CH;
  const dim3 grid(nchunk, p.V), blk(256);
  if (u.x_p3 && u.w_in_p3f) {      // the projection on operands split once per tensor
    if (!sdumc_split_on_(SDUMC_SPLIT_UMCA)) return SDUMC_EINVAL;                      // planes ARE the split arithmetic
    if ((reinterpret_cast<uintptr_t>(u.x_p3) | reinterpret_cast<uintptr_t>(u.w_in_p3f)) & 15) return SDUMC_EINVAL;
    if (p.x_drop.enabled && (qw_of(p.x_drop) & 3)) return SDUMC_EINVAL;
    static sdumc_dev_once attr_p3;
    if (sdumc_once_per_device(attr_p3, [] {
          return sdumc_set_dyn_lds(&sdumc_k3::umca_fwd_p3_kernel<true>, sdumc_k3::LDS_BYTES) && sdumc_set_dyn_lds(&sdumc_k3::umca_fwd_p3_kernel<false>, sdumc_k3::LDS_BYTES);
        }) != SDUMC_OK)
      return SDUMC_ELAUNCH;
    if (p.x_drop.enabled) hipLaunchKernelGGL(sdumc_k3::umca_fwd_p3_kernel<true>, grid, blk, sdumc_k3::LDS_BYTES, st, u, nchunk);
    else hipLaunchKernelGGL(sdumc_k3::umca_fwd_p3_kernel<false>, grid, blk, sdumc_k3::LDS_BYTES, st, u, nchunk);
  } else if (sdumc_split_on_(SDUMC_SPLIT_UMCA)) {
    if (p.x_drop.enabled) hipLaunchKernelGGL(sdumc_k3::umca_fwd_split_kernel<true>, grid, blk, sdumc_k3::LDS_BYTES, st, u, nchunk);
    else hipLaunchKernelGGL(sdumc_k3::umca_fwd_split_kernel<false>, grid, blk, sdumc_k3::LDS_BYTES, st, u, nchunk);
  } else if (p.x_drop.enabled) hipLaunchKernelGGL(sdumc_k3::umca_fwd_kernel<true>, grid, blk, sdumc_k3::LDS_BYTES, st, u, nchunk);
  else hipLaunchKernelGGL(sdumc_k3::umca_fwd_kernel<false>, grid, blk, sdumc_k3::LDS_BYTES, st, u, nchunk);
  SDUMC_CHECK_LAUNCH();
  if (p.partial_only) return SDUMC_OK;
  hipLaunchKernelGGL(attn_fwd_combine_kernel, dim3(p.V), dim3(256), (size_t)nchunk * MAXQ * sizeof(float), st, p,
                     static_cast<const float*>(p.workspace), nchunk);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

// One empty kernel per source file = per gfx950 code object: sdumc_preload_() asks for its attributes, which makes the HIP runtime load
// this file's code object NOW (outside any timed or latency-sensitive region) instead of at the first launch of one of its kernels --
// with deferred loading that first launch stalls the host for tens of milliseconds (seen as a 36-59 ms gap in the middle of an epoch,
// at the first batch whose shape took a fallback path: profiles/README.md, round 6).
__global__ void sdumc_preload_attn_pool_kernel() {}
extern "C" int sdumc_preload_attn_pool_(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&sdumc_preload_attn_pool_kernel)) == hipSuccess ? SDUMC_OK : SDUMC_ELAUNCH;
}
