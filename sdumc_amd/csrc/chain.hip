// chain.hip — the utterance-level network of WengnetMOSEIMultViewsTextMissing as FOUR launches instead of ~77.
//
//   stage A forward   model :293-332  audio/text/video_mlp -> attention_mlp -> fc_att -> fusion algebra -> the 7 query MLPs
//                                      -> query_proj of the three Cross_Attention blocks (model :85)
//   stage B forward   model :338-368  cross_{audio,text,video}_mlp -> modality-weighted sum -> cross_attention_mlp ->
//                                      cross_fc_att -> cross_fused_feat -> fc_out_v, orgin_linear_change
//   stage B backward / stage A backward: the dX mirror of both (loss.backward(), main :149); every pre-activation gradient
//   is written to HBM, so the weight gradients stay grouped MFMA GEMMs (dW = dz^T x) on a side lane, off this chain.
//
// Why one launch per stage: every layer here is per-sample (row-independent), M = 2B or 14B rows, 0.03-0.6 GFLOP -- the
// ~77 launches they used to be cost 6-13 us each whatever they computed (0.65 ms of a 2.1 ms step with the chip nearly
// idle).  Here a 512-thread workgroup owns R virtual samples and walks the whole stage with the activations in LDS; the
// only thing streamed is the weights (5.1 MB in stage A, 2.2 MB in stage B, L2-resident across the grid).
//
// The per-layer product is out[r][:] = sum_i in[r][i] * M[i][:] with M row-major [I][O]:
//   forward  M = W^T (the engine keeps transposed copies of the utterance-level weights, refreshed once per forward call)
//   backward M = W as stored ([out][in]): dX = dz W
// so that a lane owns 4 consecutive OUTPUT columns, a wave-load is one fully coalesced KiB of M, the input value is a
// wave-uniform LDS broadcast, and the inner loop is pure FMA (no cross-lane reduction per output).  The 8 waves split the
// I range; their partial rows meet in LDS once per layer, in a fixed order (bitwise reproducible, no atomics).
// fp32 throughout: with M <= 14 rows per workgroup the matrix cores would run at 1/16 .. 1/2 occupancy of their 16/32-row
// tiles, and fp32 MFMA has the VALU's rate anyway (MI355X_MICROARCH.md) -- the bound here is the weight stream per CU.
#include "chain_common.h"


namespace {

// ------------------------------------------------------------------------------------------------------------------
// stage A forward (model :293-332 + :85)
// ------------------------------------------------------------------------------------------------------------------
template <int R, class WT>
__device__ __forceinline__ void chain_fwd_a_body(const sdumc_chain_args a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* part = sm;                              // PART_FLOATS
  float* s_hpre = part + PART_FLOATS;            // [3][R][256]
  float* s_u1 = s_hpre + 3 * R * D;              // [3][R][256]
  float* s_u = s_u1 + 3 * R * D;                 // [R][768]
  float* s_att1 = s_u + 3 * R * D;               // [R][256]
  float* s_att2 = s_att1 + R * D;                // [R][256]
  float* s_alpha = s_att2 + R * D;               // [R][4]
  float* s_qin = s_alpha + R * 4;                // [7][R][256]
  float* s_q = s_qin + 7 * R * D;                // [R][7][256]
  float* s_bias = s_q + 7 * R * D;               // [18][256]: every bias of the stage (an epilogue then waits on no global load)
  const int V = a.V, v0 = blockIdx.x * R, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t VD = (int64_t)V * D;
  {
    const float* bsrc[18] = {a.umlp0_b[0], a.umlp0_b[1], a.umlp0_b[2], a.umlp3_b[0], a.umlp3_b[1], a.umlp3_b[2], a.att0_b, a.att3_b,
                             a.query_b[0], a.query_b[1], a.query_b[2], a.query_b[3], a.query_b[4], a.query_b[5], a.query_b[6],
                             a.caq_b[0], a.caq_b[1], a.caq_b[2]};
    for (int u = tid; u < 18 * (D / 4); u += NTHR) st4(s_bias + 4 * u, ld4(bsrc[u / (D / 4)] + 4 * (u % (D / 4))));
  }

  const DropRT dbase = drop_resolve(a.drop);
  RingOf<D, D, WT> ring;          // every layer of this stage streams through the same ring type (NB = 1, 4 deep)
  rxm_prefetch<D, D, WT>(ring, a.umlp0_w[0], D);
  for (int m = 0; m < 3; ++m) load_rows<R>(s_hpre + m * R * D, a.hpre + m * VD, D, D, v0, V);
  __syncthreads();
  // audio / text / video_mlp (model :293-295)
#pragma unroll 1
  for (int m = 0; m < 3; ++m) {
    FwdEpi e{s_bias + m * D, s_u1 + m * R * D, D, a.u1 + m * VD + (int64_t)v0 * D, D, true,
             mkdrop_rt(dbase, 6 + 2 * m, 1, D), (uint32_t)v0, 1u};
    auto epi = [&](int r, int col, f32x4 v) { if (v0 + r < V) e(r, col, v); };
    rxm_run<R, D, D, WT>(ring, s_hpre + m * R * D, D, a.umlp0_w[m], D, part, epi,
                     [&] { rxm_prefetch<D, D, WT>(ring, m < 2 ? a.umlp0_w[m + 1] : a.umlp3_w[0], D); });
  }
#pragma unroll 1
  for (int m = 0; m < 3; ++m) {
    FwdEpi e{s_bias + (3 + m) * D, s_u + m * D, 3 * D, a.u + (int64_t)v0 * 3 * D + m * D, 3 * D, true,
             mkdrop_rt(dbase, 7 + 2 * m, 1, D), (uint32_t)v0, 1u};
    auto epi = [&](int r, int col, f32x4 v) { if (v0 + r < V) e(r, col, v); else st4(s_u + r * 3 * D + m * D + col, f32x4{0.f, 0.f, 0.f, 0.f}); };
    rxm_run<R, D, D, WT>(ring, s_u1 + m * R * D, D, a.umlp3_w[m], D, part, epi, [&] {
      if (m < 2) rxm_prefetch<D, D, WT>(ring, a.umlp3_w[m + 1], D);
      else rxm_prefetch<3 * D, D, WT>(ring, a.att0_w, D);
    });
  }
  // attention_mlp + fc_att (model :301-303)
  {
    FwdEpi e{s_bias + 6 * D, s_att1, D, a.att1 + (int64_t)v0 * D, D, true, mkdrop_rt(dbase, 12, 1, D), (uint32_t)v0, 1u};
    auto epi = [&](int r, int col, f32x4 v) { if (v0 + r < V) e(r, col, v); else st4(s_att1 + r * D + col, f32x4{0.f, 0.f, 0.f, 0.f}); };
    rxm_run<R, 3 * D, D, WT>(ring, s_u, 3 * D, a.att0_w, D, part, epi, [&] { rxm_prefetch<D, D, WT>(ring, a.att3_w, D); });
  }
  {
    FwdEpi e{s_bias + 7 * D, s_att2, D, a.att2 + (int64_t)v0 * D, D, true, mkdrop_rt(dbase, 13, 1, D), (uint32_t)v0, 1u};
    auto epi = [&](int r, int col, f32x4 v) { if (v0 + r < V) e(r, col, v); else st4(s_att2 + r * D + col, f32x4{0.f, 0.f, 0.f, 0.f}); };
    rxm_run<R, D, D, WT>(ring, s_att1, D, a.att3_w, D, part, epi, [&] { rxm_prefetch<D, D, WT>(ring, a.query_w[0], D); });
  }
  for (int p = wave; p < 3 * R; p += NWV) {          // alpha[r][j] = att2[r] . W[j] + b[j]
    const int r = p / 3, j = p - 3 * r;
    const float s = wave_sum(dot4(ld4(s_att2 + r * D + 4 * lane), ld4(a.fc_att_w + j * D + 4 * lane)));
    if (lane == 0) {
      const float al = s + a.fc_att_b[j];
      s_alpha[r * 4 + j] = al;
      if (v0 + r < V) a.alpha[(int64_t)(v0 + r) * 3 + j] = al;
    }
  }
  __syncthreads();
  // fusion algebra (model :305-320): fused, a+t, t+v, a+v, a, t, v
  for (int u = tid; u < R * (D / 4); u += NTHR) {
    const int r = u / (D / 4), c = 4 * (u - r * (D / 4));
    const f32x4 ua = ld4(s_u + r * 3 * D + c), ut = ld4(s_u + r * 3 * D + D + c), uv = ld4(s_u + r * 3 * D + 2 * D + c);
    const float aa = s_alpha[r * 4], at = s_alpha[r * 4 + 1], av = s_alpha[r * 4 + 2];
    f32x4 o[7] = {ua * aa + ut * at + uv * av, ua * aa + ut * at, ut * at + uv * av, ua * aa + uv * av, ua, ut, uv};
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      st4(s_qin + (i * R + r) * D + c, o[i]);
      if (v0 + r < V) st4(a.qin + i * VD + (int64_t)(v0 + r) * D + c, o[i]);
    }
  }
  __syncthreads();
  // the 7 query MLPs -> multi_query [V, 7, 256] (model :324-332); text_hidden = query 5 (model :329, :370)
#pragma unroll 1
  for (int i = 0; i < 7; ++i) {
    FwdEpi e{s_bias + (8 + i) * D, s_q + i * D, NQ * D, a.q + (int64_t)v0 * NQ * D + i * D, (int64_t)NQ * D, true,
             mkdrop_rt(dbase, 14 + i, 1, D), (uint32_t)v0, 1u};
    float* th = (i == 5 && a.o_text_hidden) ? a.o_text_hidden + (int64_t)v0 * D : nullptr;
    auto epi = [&](int r, int col, f32x4 v) {
      if (v0 + r < V) {
        e(r, col, v);
        if (th) st4(th + (int64_t)r * D + col, ld4(s_q + r * NQ * D + i * D + col));
      } else {
        st4(s_q + r * NQ * D + i * D + col, f32x4{0.f, 0.f, 0.f, 0.f});
      }
    };
    rxm_run<R, D, D, WT>(ring, s_qin + i * R * D, D, a.query_w[i], D, part, epi,
                     [&] { rxm_prefetch<D, D, WT>(ring, i < 6 ? a.query_w[i + 1] : a.caq_w[0], D); });
  }
  // query_proj of the three Cross_Attention blocks (model :85): rows = (sample, query)
#pragma unroll 1
  for (int m = 0; m < 3; ++m) {
    FwdEpi e{s_bias + (15 + m) * D, nullptr, 0, a.qp + ((int64_t)m * V + v0) * NQ * D, D, false, DropRT{}, 0u, 0u};
    auto epi = [&](int r, int col, f32x4 v) { if (v0 + r / NQ < V) e(r, col, v); };
    rxm_run<NQ * R, D, D, WT>(ring, s_q, D, a.caq_w[m], D, part, epi, [&] { if (m < 2) rxm_prefetch<D, D, WT>(ring, a.caq_w[m + 1], D); });
  }
}
template <int R, class WT>
__global__ __launch_bounds__(NTHR) void chain_fwd_a_kernel(const sdumc_chain_args a) { chain_fwd_a_body<R, WT>(a); }

// ------------------------------------------------------------------------------------------------------------------
// stage B forward (model :338-368)
// ------------------------------------------------------------------------------------------------------------------
template <int R, class WT>
__device__ __forceinline__ void chain_fwd_b_body(const sdumc_chain_args a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* part = sm;
  float* s_x = part + PART_FLOATS;               // [7R][256]  ca_out_m, one modality at a time
  float* s_c1 = s_x + NQ * R * D;                // [7R][256]
  float* s_c = s_c1 + NQ * R * D;                // [3][7R][128]
  float* s_h = s_c + 3 * NQ * R * H;             // [R][896]
  float* s_e1 = s_h + R * NQ * H;                // [R][256]
  float* s_e2 = s_e1 + R * D;                    // [R][128]
  float* s_z = s_e2 + R * H;                     // [R][128]
  float* s_r1 = s_z + R * H;                     // [R][64]
  float* s_small = s_r1 + R * RD;                // alpha [R][4], beta [R][8]
  float* s_bias = s_small + 12 * R;              // cmlp0 [3][256], cmlp3 [3][128], catt0 [256], catt3 [128], rnc0 [64], rnc2 [64]
  const int V = a.V, v0 = blockIdx.x * R, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t VQ = (int64_t)V * NQ;
  {
    for (int u = tid; u < 3 * (D / 4); u += NTHR) st4(s_bias + 4 * u, ld4(a.cmlp0_b[u / (D / 4)] + 4 * (u % (D / 4))));
    for (int u = tid; u < 3 * (H / 4); u += NTHR) st4(s_bias + 3 * D + 4 * u, ld4(a.cmlp3_b[u / (H / 4)] + 4 * (u % (H / 4))));
    for (int u = tid; u < D / 4; u += NTHR) st4(s_bias + 3 * D + 3 * H + 4 * u, ld4(a.catt0_b + 4 * u));
    for (int u = tid; u < H / 4; u += NTHR) st4(s_bias + 4 * D + 3 * H + 4 * u, ld4(a.catt3_b + 4 * u));
    for (int u = tid; u < RD / 4; u += NTHR) {
      st4(s_bias + 4 * D + 4 * H + 4 * u, ld4(a.rnc0_b + 4 * u));
      st4(s_bias + 4 * D + 4 * H + RD + 4 * u, ld4(a.rnc2_b + 4 * u));
    }
  }

  const DropRT dbase = drop_resolve(a.drop);
  RingOf<D, D, WT> ring;
  rxm_prefetch<D, D, WT>(ring, a.cmlp0_w[0], D);
  for (int u = tid; u < R * 3; u += NTHR) {
    const int r = u / 3, j = u - 3 * r;
    s_small[r * 4 + j] = v0 + r < V ? a.alpha[(int64_t)(v0 + r) * 3 + j] : 0.f;
  }
  // cross_{audio,text,video}_mlp (model :338-340), rows = (sample, query)
#pragma unroll 1
  for (int m = 0; m < 3; ++m) {
    __syncthreads();
    load_rows<NQ * R>(s_x, a.ca_out + (int64_t)m * VQ * D, D, D, v0 * NQ, V * NQ);
    __syncthreads();
    {
      FwdEpi e{s_bias + m * D, s_c1, D, a.c1 + ((int64_t)m * VQ + (int64_t)v0 * NQ) * D, D, true,
               mkdrop_rt(dbase, 27 + 2 * m, NQ, D), (uint32_t)(v0 * NQ), 1u};
      auto epi = [&](int r, int col, f32x4 v) { if (v0 * NQ + r < V * NQ) e(r, col, v); else st4(s_c1 + r * D + col, f32x4{0.f, 0.f, 0.f, 0.f}); };
      rxm_run<NQ * R, D, D, WT>(ring, s_x, D, a.cmlp0_w[m], D, part, epi, [&] { rxm_prefetch<D, H, WT>(ring, a.cmlp3_w[m], H); });
    }
    {
      FwdEpi e{s_bias + 3 * D + m * H, s_c + m * NQ * R * H, H, a.c + ((int64_t)m * VQ + (int64_t)v0 * NQ) * H, H, true,
               mkdrop_rt(dbase, 28 + 2 * m, NQ, H), (uint32_t)(v0 * NQ), 1u};
      float* ct = (m == 1 && a.o_cross_text) ? a.o_cross_text + (int64_t)v0 * NQ * H : nullptr;
      auto epi = [&](int r, int col, f32x4 v) {
        if (v0 * NQ + r < V * NQ) {
          e(r, col, v);
          if (ct) st4(ct + (int64_t)r * H + col, ld4(s_c + (m * NQ * R + r) * H + col));
        } else {
          st4(s_c + (m * NQ * R + r) * H + col, f32x4{0.f, 0.f, 0.f, 0.f});
        }
      };
      rxm_run<NQ * R, D, H, WT>(ring, s_c1, D, a.cmlp3_w[m], H, part, epi, [&] {
        if (m < 2) rxm_prefetch<D, D, WT>(ring, a.cmlp0_w[m + 1], D);
        else rxm_prefetch<NQ * H, D, WT>(ring, a.catt0_w, D);
      });
    }
  }
  // modality-weighted sum (model :346-349): h[r][i][:] = sum_m alpha[r][m] c_m[r][i][:]
  for (int u = tid; u < R * NQ * (H / 4); u += NTHR) {
    const int ri = u / (H / 4), c = 4 * (u - ri * (H / 4)), r = ri / NQ;
    const f32x4 hv = ld4(s_c + ri * H + c) * s_small[r * 4] + ld4(s_c + (NQ * R + ri) * H + c) * s_small[r * 4 + 1] +
                     ld4(s_c + (2 * NQ * R + ri) * H + c) * s_small[r * 4 + 2];
    st4(s_h + ri * H + c, hv);
    if (v0 + r < V) st4(a.h + ((int64_t)v0 * NQ + ri) * H + c, hv);
  }
  __syncthreads();
  // cross_attention_mlp + cross_fc_att (model :352-354)
  {
    FwdEpi e{s_bias + 3 * D + 3 * H, s_e1, D, a.e1 + (int64_t)v0 * D, D, true, mkdrop_rt(dbase, 33, 1, D), (uint32_t)v0, 1u};
    auto epi = [&](int r, int col, f32x4 v) { if (v0 + r < V) e(r, col, v); else st4(s_e1 + r * D + col, f32x4{0.f, 0.f, 0.f, 0.f}); };
    rxm_run<R, NQ * H, D, WT>(ring, s_h, NQ * H, a.catt0_w, D, part, epi, [&] { rxm_prefetch<D, H, WT>(ring, a.catt3_w, H); });
  }
  {
    FwdEpi e{s_bias + 4 * D + 3 * H, s_e2, H, a.e2 + (int64_t)v0 * H, H, true, mkdrop_rt(dbase, 34, 1, H), (uint32_t)v0, 1u};
    auto epi = [&](int r, int col, f32x4 v) { if (v0 + r < V) e(r, col, v); else st4(s_e2 + r * H + col, f32x4{0.f, 0.f, 0.f, 0.f}); };
    rxm_run<R, D, H, WT>(ring, s_e1, D, a.catt3_w, H, part, epi, [] {});
  }
  for (int p = wave; p < NQ * R; p += NWV) {          // beta[r][i] = e2[r] . W[i] + b[i]
    const int r = p / NQ, i = p - NQ * r;
    float s = lane < H / 4 ? dot4(ld4(s_e2 + r * H + 4 * lane), ld4(a.cfa_w + i * H + 4 * lane)) : 0.f;
    s = wave_sum(s);
    if (lane == 0) {
      const float bt = s + a.cfa_b[i];
      s_small[4 * R + r * 8 + i] = bt;
      if (v0 + r < V) a.beta[(int64_t)(v0 + r) * NQ + i] = bt;
    }
  }
  __syncthreads();
  // cross_fused_feat (model :356-358), fc_out_v (model :364)
  for (int u = tid; u < R * (H / 4); u += NTHR) {
    const int r = u / (H / 4), c = 4 * (u - r * (H / 4));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NQ; ++i) acc += ld4(s_h + (r * NQ + i) * H + c) * s_small[4 * R + r * 8 + i];
    st4(s_z + r * H + c, acc);
    if (v0 + r < V) {
      st4(a.z + (int64_t)(v0 + r) * H + c, acc);
      if (a.o_fused) st4(a.o_fused + (int64_t)(v0 + r) * H + c, acc);
    }
  }
  __syncthreads();
  for (int r = wave; r < R; r += NWV) {
    float s = lane < H / 4 ? dot4(ld4(s_z + r * H + 4 * lane), ld4(a.fcv_w + 4 * lane)) : 0.f;
    s = wave_sum(s);
    if (lane == 0 && v0 + r < V) {
      const float y = s + a.fcv_b[0];
      a.vals[v0 + r] = y;
      if (a.o_vals) a.o_vals[v0 + r] = y;
    }
  }
  // orgin_linear_change (model :246-250, :368): Linear -> ReLU -> Linear
  {
    FwdEpi e{s_bias + 4 * D + 4 * H, s_r1, RD, a.r1 + (int64_t)v0 * RD, RD, true, DropRT{}, 0u, 0u};
    auto epi = [&](int r, int col, f32x4 v) { if (v0 + r < V) e(r, col, v); else st4(s_r1 + r * RD + col, f32x4{0.f, 0.f, 0.f, 0.f}); };
    rows_x_matrix<R, H, RD>(s_z, H, a.rnc0_w, RD, part, epi);
  }
  {
    FwdEpi e{s_bias + 4 * D + 4 * H + RD, nullptr, 0, a.r + (int64_t)v0 * RD, RD, false, DropRT{}, 0u, 0u};
    float* ro = a.o_rnc ? a.o_rnc + (int64_t)v0 * RD : nullptr;
    const float* b2 = s_bias + 4 * D + 4 * H + RD;
    auto epi = [&](int r, int col, f32x4 v) {
      if (v0 + r < V) {
        e(r, col, v);
        if (ro) st4(ro + (int64_t)r * RD + col, v + ld4(b2 + col));
      }
    };
    rows_x_matrix<R, RD, RD>(s_r1, RD, a.rnc2_w, RD, part, epi);
  }
}
template <int R, class WT>
__global__ __launch_bounds__(NTHR) void chain_fwd_b_kernel(const sdumc_chain_args a) { chain_fwd_b_body<R, WT>(a); }

// ------------------------------------------------------------------------------------------------------------------
// stage B backward: d(vals, fused, rnc, cross_text) -> d_ca_out [3][V,7,256], d_alpha (second-level part), and every
// pre-activation gradient the dW GEMMs need (d_rnc is the caller's; d_r1, d_z, d_beta, d_e2, d_e1, d_c, d_c1)
// ------------------------------------------------------------------------------------------------------------------
template <int R, class WT>
__device__ __forceinline__ void chain_bwd_b_body(const sdumc_chain_args a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* part = sm;
  float* s_g = part + PART_FLOATS;               // [R][64]   d_rnc
  float* s_r1 = s_g + R * RD;                    // [R][64]   saved r1, then d_r1
  float* s_dz = s_r1 + R * RD;                   // [R][128]
  float* s_h = s_dz + R * H;                     // [R][896]  saved h
  float* s_dh = s_h + R * NQ * H;                // [R][896]
  float* s_e2 = s_dh + R * NQ * H;               // [R][128]  saved e2, then d_e2
  float* s_e1 = s_e2 + R * H;                    // [R][256]  saved e1, then d_e1
  float* s_dc = s_e1 + R * D;                    // [7R][128] d_c of one modality
  float* s_c1 = s_dc + NQ * R * H;               // [7R][256] saved c1, then d_c1
  float* s_small = s_c1 + NQ * R * D;            // beta [R][8], d_beta [R][8], alpha [R][4], d_vals [R], dalpha partial
  const int V = a.V, v0 = blockIdx.x * R, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t VQ = (int64_t)V * NQ;
  const float sc = a.relu_scale;
  float* s_beta = s_small;
  float* s_dbeta = s_small + 8 * R;
  float* s_alpha = s_small + 16 * R;
  float* s_dvals = s_small + 20 * R;

  RingOf<H, D, WT> ring;           // catt3 / cmlp3 / cmlp0 backward: NB = 1, 4 deep
  RingOf<D, NQ * H, WT> ring4;     // catt0 backward: 896 output columns = 4 blocks, 2 deep
  rxm_prefetch<H, D, WT>(ring, a.catt3_w, D);
  if (a.g_rnc) load_rows<R>(s_g, a.g_rnc, RD, RD, v0, V);
  else for (int u = tid; u < R * RD; u += NTHR) s_g[u] = 0.f;
  load_rows<R>(s_r1, a.r1, RD, RD, v0, V);
  load_rows<R>(s_h, a.h, NQ * H, NQ * H, v0, V);
  load_rows<R>(s_e2, a.e2, H, H, v0, V);
  load_rows<R>(s_e1, a.e1, D, D, v0, V);
  for (int u = tid; u < R * 8; u += NTHR) {
    const int r = u >> 3, i = u & 7;
    s_beta[u] = (i < NQ && v0 + r < V) ? a.beta[(int64_t)(v0 + r) * NQ + i] : 0.f;
    if (i < 3) s_alpha[r * 4 + i] = v0 + r < V ? a.alpha[(int64_t)(v0 + r) * 3 + i] : 0.f;
    if (i == 0) s_dvals[r] = (a.g_vals && v0 + r < V) ? a.g_vals[v0 + r] : 0.f;
  }
  __syncthreads();
  // 12'. orgin_linear_change backward: d_r1 = (d_rnc W2) [r1 > 0] ; d_z = d_r1 W0
  {
    BwdEpi e{nullptr, 0, s_r1, RD, 1.0f, s_r1, RD, a.d_r1 + (int64_t)v0 * RD, RD};
    auto epi = [&](int r, int col, f32x4 v) { if (v0 + r < V) e(r, col, v); else st4(s_r1 + r * RD + col, f32x4{0.f, 0.f, 0.f, 0.f}); };
    rows_x_matrix<R, RD, RD>(s_g, RD, a.rnc2_w, RD, part, epi);
  }
  {
    // d_z = d_r1 W0 + d_vals * w_v + d_fused
    const float* gf = a.g_fused;
    const float* wv = a.fcv_w;
    float* dzg = a.d_z;
    auto epi = [&](int r, int col, f32x4 v) {
      v += ld4(wv + col) * s_dvals[r];
      if (gf && v0 + r < V) v += ld4(gf + (int64_t)(v0 + r) * H + col);
      if (v0 + r >= V) v = f32x4{0.f, 0.f, 0.f, 0.f};
      st4(s_dz + r * H + col, v);
      if (v0 + r < V) st4(dzg + (int64_t)(v0 + r) * H + col, v);
    };
    rows_x_matrix<R, RD, H>(s_r1, RD, a.rnc0_w, H, part, epi);
  }
  // zpool backward: d_h[i] = beta_i d_z ; d_beta_i = <d_z, h_i>
  for (int u = tid; u < R * NQ * (H / 4); u += NTHR) {
    const int ri = u / (H / 4), c = 4 * (u - ri * (H / 4)), r = ri / NQ, i = ri - r * NQ;
    st4(s_dh + ri * H + c, ld4(s_dz + r * H + c) * s_beta[r * 8 + i]);
  }
  for (int p = wave; p < NQ * R; p += NWV) {
    const int r = p / NQ, i = p - NQ * r;
    float s = lane < H / 4 ? dot4(ld4(s_dz + r * H + 4 * lane), ld4(s_h + (r * NQ + i) * H + 4 * lane)) : 0.f;
    s = wave_sum(s);
    if (lane == 0) {
      s_dbeta[r * 8 + i] = s;
      if (v0 + r < V) a.d_beta[(int64_t)(v0 + r) * NQ + i] = s;
    }
  }
  __syncthreads();
  // 11'. cross_fc_att: d_e2 = (d_beta W_cfa) [e2 > 0] s ; cross_attention_mlp.3: d_e1 = (d_e2 W) [e1 > 0] s ; .0: d_h += d_e1 W
  for (int u = tid; u < R * (H / 4); u += NTHR) {
    const int r = u / (H / 4), c = 4 * (u - r * (H / 4));
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NQ; ++i) g += ld4(a.cfa_w + i * H + c) * s_dbeta[r * 8 + i];
    const f32x4 y = ld4(s_e2 + r * H + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = y[j] > 0.f ? g[j] * sc : 0.f;
    if (v0 + r < V) st4(a.d_e2 + (int64_t)(v0 + r) * H + c, g);
    st4(s_dz + r * H + c, g);        // d_z is no longer needed in LDS: the slot now holds d_e2
  }
  __syncthreads();
  {
    BwdEpi e{nullptr, 0, s_e1, D, sc, s_e1, D, a.d_e1 + (int64_t)v0 * D, D};
    auto epi = [&](int r, int col, f32x4 v) { if (v0 + r < V) e(r, col, v); else st4(s_e1 + r * D + col, f32x4{0.f, 0.f, 0.f, 0.f}); };
    rxm_run<R, H, D, WT>(ring, s_dz, H, a.catt3_w, D, part, epi, [&] { rxm_prefetch<D, NQ * H, WT>(ring4, a.catt0_w, NQ * H); });
  }
  {
    BwdEpi e{s_dh, NQ * H, nullptr, 0, 1.f, s_dh, NQ * H, nullptr, 0};
    auto epi = [&](int r, int col, f32x4 v) { e(r, col, v); };
    rxm_run<R, D, NQ * H, WT>(ring4, s_e1, D, a.catt0_w, NQ * H, part, epi, [&] { rxm_prefetch<H, D, WT>(ring, a.cmlp3_w[0], D); });
  }
  // 10'. modality-weighted sum backward, then 9'. cross_*_mlp, one modality at a time
#pragma unroll 1
  for (int m = 0; m < 3; ++m) {
    // d_c_m = (alpha_m d_h (+ d_cross_text on m = 1)) [c_m > 0] s ; d_alpha_m = <d_h, c_m> (per-quad partials through LDS,
    // summed per sample in a fixed order)
    for (int u = tid; u < R * NQ * (H / 4); u += NTHR) {
      const int ri = u / (H / 4), c = 4 * (u - ri * (H / 4)), r = ri / NQ;
      const bool live = v0 + r < V;
      f32x4 cm = {0.f, 0.f, 0.f, 0.f}, g = ld4(s_dh + ri * H + c);
      if (live) cm = ld4(a.c + ((int64_t)m * VQ + (int64_t)v0 * NQ + ri) * H + c);
      part[u] = dot4(g, cm);
      g = g * s_alpha[r * 4 + m];
      if (m == 1 && a.g_cross_text && live) g += ld4(a.g_cross_text + ((int64_t)v0 * NQ + ri) * H + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = cm[j] > 0.f ? g[j] * sc : 0.f;
      st4(s_dc + ri * H + c, g);
      if (live) st4(a.d_c + ((int64_t)m * VQ + (int64_t)v0 * NQ + ri) * H + c, g);
    }
    __syncthreads();
    for (int r = wave; r < R; r += NWV) {
      float s = 0.f;
      for (int k = lane; k < NQ * (H / 4); k += 64) s += part[r * NQ * (H / 4) + k];
      s = wave_sum(s);
      if (lane == 0 && v0 + r < V) a.d_alpha[(int64_t)(v0 + r) * 3 + m] = s;
    }
    load_rows<NQ * R>(s_c1, a.c1 + (int64_t)m * VQ * D, D, D, v0 * NQ, V * NQ);
    __syncthreads();
    {
      BwdEpi e{nullptr, 0, s_c1, D, sc, s_c1, D, a.d_c1 + ((int64_t)m * VQ + (int64_t)v0 * NQ) * D, D};
      auto epi = [&](int r, int col, f32x4 v) { if (v0 * NQ + r < V * NQ) e(r, col, v); else st4(s_c1 + r * D + col, f32x4{0.f, 0.f, 0.f, 0.f}); };
      rxm_run<NQ * R, H, D, WT>(ring, s_dc, H, a.cmlp3_w[m], D, part, epi, [&] { rxm_prefetch<D, D, WT>(ring, a.cmlp0_w[m], D); });
    }
    {
      BwdEpi e{nullptr, 0, nullptr, 0, 1.f, nullptr, 0, a.d_ca_out + ((int64_t)m * VQ + (int64_t)v0 * NQ) * D, D};
      auto epi = [&](int r, int col, f32x4 v) { if (v0 * NQ + r < V * NQ) e(r, col, v); };
      rxm_run<NQ * R, D, D, WT>(ring, s_c1, D, a.cmlp0_w[m], D, part, epi, [&] { if (m < 2) rxm_prefetch<H, D, WT>(ring, a.cmlp3_w[m + 1], D); });
    }
  }
}
template <int R, class WT>
__global__ __launch_bounds__(NTHR) void chain_bwd_b_kernel(const sdumc_chain_args a) { chain_bwd_b_body<R, WT>(a); }

// ------------------------------------------------------------------------------------------------------------------
// stage A backward: d_qp [3][V,7,256] (+ d text_hidden, + d_alpha from stage B) -> d_hpre [3][V,256] and the
// pre-activation gradients d_q, d_qin (as d of the query MLP inputs), d_att2, d_att1, d_u, d_u1
// ------------------------------------------------------------------------------------------------------------------
template <int R, class WT>
__device__ __forceinline__ void chain_bwd_a_body(const sdumc_chain_args a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* part = sm;
  float* s_x = part + PART_FLOATS;               // [7R][256]  d_qp_m, one modality at a time
  float* s_dq = s_x + NQ * R * D;                // [R][7][256]
  float* s_dqin = s_dq + NQ * R * D;             // [7][R][256]
  float* s_u = s_dqin + NQ * R * D;              // [R][768]  saved u
  float* s_du = s_u + 3 * R * D;                 // [R][768]
  float* s_a2 = s_du + 3 * R * D;                // [R][256]  saved att2, then d_att2
  float* s_a1 = s_a2 + R * D;                    // [R][256]  saved att1, then d_att1
  float* s_u1 = s_a1 + R * D;                    // [3][R][256] saved u1, then d_u1
  float* s_small = s_u1 + 3 * R * D;             // alpha [R][4], d_alpha [R][4]
  const int V = a.V, v0 = blockIdx.x * R, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t VD = (int64_t)V * D, VQ = (int64_t)V * NQ;
  const float sc = a.relu_scale;
  float* s_alpha = s_small;
  float* s_dalpha = s_small + 4 * R;

  RingOf<D, D, WT> ring;
  RingOf<D, 3 * D, WT> ring3;      // att0 backward: 768 output columns = 3 blocks, 2 deep
  rxm_prefetch<D, D, WT>(ring, a.caq_w[0], D);
  load_rows<R>(s_u, a.u, 3 * D, 3 * D, v0, V);
  load_rows<R>(s_a2, a.att2, D, D, v0, V);
  load_rows<R>(s_a1, a.att1, D, D, v0, V);
  for (int m = 0; m < 3; ++m) load_rows<R>(s_u1 + m * R * D, a.u1 + m * VD, D, D, v0, V);
  for (int u = tid; u < R * 3; u += NTHR) {
    const int r = u / 3, j = u - 3 * r;
    s_alpha[r * 4 + j] = v0 + r < V ? a.alpha[(int64_t)(v0 + r) * 3 + j] : 0.f;
    s_dalpha[r * 4 + j] = v0 + r < V ? a.d_alpha[(int64_t)(v0 + r) * 3 + j] : 0.f;
  }
  // 7'. query_proj: d_q = sum_m d_qp_m W_q[m]  (accumulated in LDS over the three modalities)
#pragma unroll 1
  for (int m = 0; m < 3; ++m) {
    __syncthreads();
    load_rows<NQ * R>(s_x, a.d_qp + (int64_t)m * VQ * D, D, D, v0 * NQ, V * NQ);
    __syncthreads();
    BwdEpi e{m > 0 ? s_dq : nullptr, D, nullptr, 0, 1.f, s_dq, D, nullptr, 0};
    auto epi = [&](int r, int col, f32x4 v) { e(r, col, v); };
    rxm_run<NQ * R, D, D, WT>(ring, s_x, D, a.caq_w[m], D, part, epi,
                          [&] { rxm_prefetch<D, D, WT>(ring, m < 2 ? a.caq_w[m + 1] : a.query_w[0], D); });
  }
  // 6'. + the external gradient of text_hidden (= query 5), ReLU/dropout mask of q -> d_q (pre-activation); query MLPs
  for (int u = tid; u < R * NQ * (D / 4); u += NTHR) {
    const int ri = u / (D / 4), c = 4 * (u - ri * (D / 4)), r = ri / NQ, i = ri - r * NQ;
    f32x4 g = ld4(s_dq + ri * D + c);
    const bool live = v0 + r < V;
    if (i == 5 && a.g_text_hidden && live) g += ld4(a.g_text_hidden + (int64_t)(v0 + r) * D + c);
    f32x4 y = {0.f, 0.f, 0.f, 0.f};
    if (live) y = ld4(a.q + ((int64_t)(v0 + r) * NQ + i) * D + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = y[j] > 0.f ? g[j] * sc : 0.f;
    st4(s_dq + ri * D + c, g);
    if (live) st4(a.d_q + ((int64_t)(v0 + r) * NQ + i) * D + c, g);
  }
  __syncthreads();
#pragma unroll 1
  for (int i = 0; i < 7; ++i) {
    BwdEpi e{nullptr, 0, nullptr, 0, 1.f, s_dqin + i * R * D, D, a.d_qin + i * VD + (int64_t)v0 * D, D};
    auto epi = [&](int r, int col, f32x4 v) { if (v0 + r < V) e(r, col, v); else st4(s_dqin + (i * R + r) * D + col, f32x4{0.f, 0.f, 0.f, 0.f}); };
    rxm_run<R, D, D, WT>(ring, s_dq + i * D, NQ * D, a.query_w[i], D, part, epi,
                     [&] { rxm_prefetch<D, D, WT>(ring, i < 6 ? a.query_w[i + 1] : a.att3_w, D); });
  }
  // 5'. fusion algebra backward: d_u (fusion part), d_alpha += <g_m, u_m>
  for (int u = tid; u < R * (D / 4); u += NTHR) {
    const int r = u / (D / 4), c = 4 * (u - r * (D / 4));
    const f32x4 df = ld4(s_dqin + (0 * R + r) * D + c), dfat = ld4(s_dqin + (1 * R + r) * D + c),
                dftv = ld4(s_dqin + (2 * R + r) * D + c), dfav = ld4(s_dqin + (3 * R + r) * D + c);
    const f32x4 ga = df + dfat + dfav, gt = df + dfat + dftv, gv = df + dftv + dfav;
    const float aa = s_alpha[r * 4], at = s_alpha[r * 4 + 1], av = s_alpha[r * 4 + 2];
    st4(s_du + r * 3 * D + c, ga * aa + ld4(s_dqin + (4 * R + r) * D + c));
    st4(s_du + r * 3 * D + D + c, gt * at + ld4(s_dqin + (5 * R + r) * D + c));
    st4(s_du + r * 3 * D + 2 * D + c, gv * av + ld4(s_dqin + (6 * R + r) * D + c));
    part[(r * 3 + 0) * (D / 4) + (c >> 2)] = dot4(ga, ld4(s_u + r * 3 * D + c));
    part[(r * 3 + 1) * (D / 4) + (c >> 2)] = dot4(gt, ld4(s_u + r * 3 * D + D + c));
    part[(r * 3 + 2) * (D / 4) + (c >> 2)] = dot4(gv, ld4(s_u + r * 3 * D + 2 * D + c));
  }
  __syncthreads();
  for (int p = wave; p < 3 * R; p += NWV) {
    const float s = wave_sum(part[p * (D / 4) + lane]);
    if (lane == 0) {
      const int r = p / 3, j = p - 3 * r;
      const float da = s_dalpha[r * 4 + j] + s;
      s_dalpha[r * 4 + j] = da;
      if (v0 + r < V) a.d_alpha[(int64_t)(v0 + r) * 3 + j] = da;      // final d_alpha: what the fc_att dW GEMM reads
    }
  }
  __syncthreads();
  // 4'. fc_att: d_att2 = (d_alpha W_fc) [att2 > 0] s ; attention_mlp.3: d_att1 = (d_att2 W) [att1 > 0] s ;
  //     .0: d_u = (d_u + d_att1 W) [u > 0] s
  for (int u = tid; u < R * (D / 4); u += NTHR) {
    const int r = u / (D / 4), c = 4 * (u - r * (D / 4));
    f32x4 g = ld4(a.fc_att_w + c) * s_dalpha[r * 4] + ld4(a.fc_att_w + D + c) * s_dalpha[r * 4 + 1] +
              ld4(a.fc_att_w + 2 * D + c) * s_dalpha[r * 4 + 2];
    const f32x4 y = ld4(s_a2 + r * D + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = y[j] > 0.f ? g[j] * sc : 0.f;
    st4(s_a2 + r * D + c, g);
    if (v0 + r < V) st4(a.d_att2 + (int64_t)(v0 + r) * D + c, g);
  }
  __syncthreads();
  {
    BwdEpi e{nullptr, 0, s_a1, D, sc, s_a1, D, a.d_att1 + (int64_t)v0 * D, D};
    auto epi = [&](int r, int col, f32x4 v) { if (v0 + r < V) e(r, col, v); else st4(s_a1 + r * D + col, f32x4{0.f, 0.f, 0.f, 0.f}); };
    rxm_run<R, D, D, WT>(ring, s_a2, D, a.att3_w, D, part, epi, [&] { rxm_prefetch<D, 3 * D, WT>(ring3, a.att0_w, 3 * D); });
  }
  {
    BwdEpi e{s_du, 3 * D, s_u, 3 * D, sc, s_du, 3 * D, a.d_u + (int64_t)v0 * 3 * D, 3 * D};
    auto epi = [&](int r, int col, f32x4 v) { if (v0 + r < V) e(r, col, v); else st4(s_du + r * 3 * D + col, f32x4{0.f, 0.f, 0.f, 0.f}); };
    rxm_run<R, D, 3 * D, WT>(ring3, s_a1, D, a.att0_w, 3 * D, part, epi, [&] { rxm_prefetch<D, D, WT>(ring, a.umlp3_w[0], D); });
  }
  // 3'. audio / text / video_mlp: d_u1 = (d_u_m W3) [u1 > 0] s ; d_hpre = d_u1 W0
#pragma unroll 1
  for (int m = 0; m < 3; ++m) {
    {
      BwdEpi e{nullptr, 0, s_u1 + m * R * D, D, sc, s_u1 + m * R * D, D, a.d_u1 + m * VD + (int64_t)v0 * D, D};
      auto epi = [&](int r, int col, f32x4 v) { if (v0 + r < V) e(r, col, v); else st4(s_u1 + (m * R + r) * D + col, f32x4{0.f, 0.f, 0.f, 0.f}); };
      rxm_run<R, D, D, WT>(ring, s_du + m * D, 3 * D, a.umlp3_w[m], D, part, epi, [&] { rxm_prefetch<D, D, WT>(ring, a.umlp0_w[m], D); });
    }
    {
      BwdEpi e{nullptr, 0, nullptr, 0, 1.f, nullptr, 0, a.d_hpre + m * VD + (int64_t)v0 * D, D};
      auto epi = [&](int r, int col, f32x4 v) { if (v0 + r < V) e(r, col, v); };
      rxm_run<R, D, D, WT>(ring, s_u1 + m * R * D, D, a.umlp0_w[m], D, part, epi, [&] { if (m < 2) rxm_prefetch<D, D, WT>(ring, a.umlp3_w[m + 1], D); });
    }
  }
}
template <int R, class WT>
__global__ __launch_bounds__(NTHR) void chain_bwd_a_kernel(const sdumc_chain_args a) { chain_bwd_a_body<R, WT>(a); }

// transposed mirror of the utterance-level weights: dst[i][o] = src[o][i] for up to 40 matrices in one launch
struct TransposeList {
  int n;
  struct { int64_t off; int32_t out, in; } e[40];
};
__global__ __launch_bounds__(256) void transpose_params_kernel(const float* __restrict__ src, float* __restrict__ dst, const TransposeList tl) {
  __shared__ float t[32][33];
  const auto& e = tl.e[blockIdx.z];
  const int o0 = blockIdx.y * 32, i0 = blockIdx.x * 32;
  if (o0 >= e.out || i0 >= e.in) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8)
    if (o0 + k < e.out && i0 + tx < e.in) t[k][tx] = src[e.off + (int64_t)(o0 + k) * e.in + i0 + tx];
  __syncthreads();
  for (int k = ty; k < 32; k += 8)
    if (i0 + k < e.in && o0 + tx < e.out) dst[e.off + (int64_t)(i0 + k) * e.out + o0 + tx] = t[tx][k];
}

template <int R> constexpr size_t smem_fwd_a() { return sizeof(float) * (PART_FLOATS + 3 * R * D * 3 + 2 * R * D + 4 * R + 14 * R * D + 18 * D); }
template <int R> constexpr size_t smem_fwd_b() { return sizeof(float) * (PART_FLOATS + 2 * NQ * R * D + 3 * NQ * R * H + R * NQ * H + R * D + 2 * R * H + R * RD + 12 * R + 4 * D + 4 * H + 2 * RD); }
template <int R> constexpr size_t smem_bwd_b() { return sizeof(float) * (PART_FLOATS + 2 * R * RD + R * H + 2 * R * NQ * H + R * H + R * D + NQ * R * H + NQ * R * D + 24 * R); }
template <int R> constexpr size_t smem_bwd_a() { return sizeof(float) * (PART_FLOATS + 3 * NQ * R * D + 6 * R * D + 2 * R * D + 3 * R * D + 8 * R); }

template <class K>
int set_smem(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess
             ? SDUMC_OK : SDUMC_ELAUNCH;
}

}  // namespace

// which: 0 = stage A forward, 1 = stage B forward, 2 = stage B backward, 3 = stage A backward
extern "C" int sdumc_chain_launch_(const sdumc_chain_args* a, int which, void* stream) {
  if (!a || a->V <= 0 || which < 0 || which > 3) return SDUMC_EINVAL;
  constexpr int R = 2;
  typedef unsigned short bf;
  static sdumc_dev_once attr;
  if (sdumc_once_per_device(attr, [] {
        return !(set_smem(chain_fwd_a_kernel<R, float>, smem_fwd_a<R>()) || set_smem(chain_fwd_b_kernel<R, float>, smem_fwd_b<R>()) ||
                 set_smem(chain_bwd_b_kernel<R, float>, smem_bwd_b<R>()) || set_smem(chain_bwd_a_kernel<R, float>, smem_bwd_a<R>()) ||
                 set_smem(chain_fwd_a_kernel<R, bf>, smem_fwd_a<R>()) || set_smem(chain_fwd_b_kernel<R, bf>, smem_fwd_b<R>()) ||
                 set_smem(chain_bwd_b_kernel<R, bf>, smem_bwd_b<R>()) || set_smem(chain_bwd_a_kernel<R, bf>, smem_bwd_a<R>()));
      }) != SDUMC_OK)
    return SDUMC_ELAUNCH;
  const dim3 grid((a->V + R - 1) / R), blk(NTHR);
  hipStream_t st = as_stream(stream);
  // (one entry point per stage and weight type: the whole device build carries no packed fp32 operations -- Makefile, NOPACK --,
  //  so the "_np_" twins of round 4 compiled to the same code and are gone)
  if (a->w_bf16) {
    switch (which) {
      case 0: hipLaunchKernelGGL((chain_fwd_a_kernel<R, bf>), grid, blk, smem_fwd_a<R>(), st, *a); break;
      case 1: hipLaunchKernelGGL((chain_fwd_b_kernel<R, bf>), grid, blk, smem_fwd_b<R>(), st, *a); break;
      case 2: hipLaunchKernelGGL((chain_bwd_b_kernel<R, bf>), grid, blk, smem_bwd_b<R>(), st, *a); break;
      default: hipLaunchKernelGGL((chain_bwd_a_kernel<R, bf>), grid, blk, smem_bwd_a<R>(), st, *a); break;
    }
  } else {
    switch (which) {
      case 0: hipLaunchKernelGGL((chain_fwd_a_kernel<R, float>), grid, blk, smem_fwd_a<R>(), st, *a); break;
      case 1: hipLaunchKernelGGL((chain_fwd_b_kernel<R, float>), grid, blk, smem_fwd_b<R>(), st, *a); break;
      case 2: hipLaunchKernelGGL((chain_bwd_b_kernel<R, float>), grid, blk, smem_bwd_b<R>(), st, *a); break;
      default: hipLaunchKernelGGL((chain_bwd_a_kernel<R, float>), grid, blk, smem_bwd_a<R>(), st, *a); break;
    }
  }
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

// dst[off .. ] = transpose of the n listed [out][in] matrices of src (same offsets): the forward chain's weight layout
extern "C" int sdumc_chain_transpose_(const float* src, float* dst, const int64_t* offs, const int32_t* outs, const int32_t* ins, int n,
                                      void* stream) {
  if (!src || !dst || n <= 0 || n > 40) return SDUMC_EINVAL;
  TransposeList tl;
  tl.n = n;
  int mo = 0, mi = 0;
  for (int i = 0; i < n; ++i) {
    tl.e[i].off = offs[i];
    tl.e[i].out = outs[i];
    tl.e[i].in = ins[i];
    mo = outs[i] > mo ? outs[i] : mo;
    mi = ins[i] > mi ? ins[i] : mi;
  }
  hipLaunchKernelGGL(transpose_params_kernel, dim3((mi + 31) / 32, (mo + 31) / 32, n), dim3(256), 0, as_stream(stream), src, dst, tl);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

// One empty kernel per source file = per gfx950 code object: sdumc_preload_() asks for its attributes, which makes the HIP runtime load
// this file's code object NOW (outside any timed or latency-sensitive region) instead of at the first launch of one of its kernels --
// with deferred loading that first launch stalls the host for tens of milliseconds (seen as a 36-59 ms gap in the middle of an epoch,
// at the first batch whose shape took a fallback path: profiles/README.md, round 6).
__global__ void sdumc_preload_chain_kernel() {}
extern "C" int sdumc_preload_chain_(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&sdumc_preload_chain_kernel)) == hipSuccess ? SDUMC_OK : SDUMC_ELAUNCH;
}
