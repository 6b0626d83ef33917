// chain_common.h -- the rows-x-matrix engine shared by chain.hip (one workgroup per R samples) and chain_cluster.hip (the same
// stages with every layer's output columns split over a cluster of workgroups).  See chain.hip for the design notes.
#pragma once

// Packed FP32 VALU instructions (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) and bf16 MFMAs on one CU.
// Measured on MI355X (round 4: tools/fwd_determinism_probe.py, profiles/README.md): when a workgroup of these kernels shares a CU
// with a workgroup of a bf16 MFMA GEMM (v_mfma_f32_32x32x16_bf16 -- the Cross_Attention key projections that run beside stage A),
// the LOW half of some v_pk_fma_f32 results comes out wrong: 26-56 of 299 forwards differed from the fp64 truth in a few
// even-numbered output columns of a layer; 0 of 3 x 299 without packed ops, 0 of 2 x 299 when the two kernels cannot share a CU
// (LDS padding), 0 of 199 with fp32-MFMA neighbours.  Inputs were verified intact inside the kernel (weights in registers against
// agent-scope reloads, the LDS copy of the input rows against global memory).
// The whole library is now built without packed fp32 operations (Makefile, NOPACK: since the fp32 GEMMs compute their products on
// the bf16 matrix pipe every kernel has such neighbours, and the pooling kernels showed the same symptom).  The attribute below
// and the "_np_" entry points it marks (selected by sdumc_chain_args.no_packed_fp32) predate that flag; under it both entry points
// of a stage compile to the same code.
#if defined(__HIP_DEVICE_COMPILE__)
#define SDUMC_NO_PACKED_FP32 __attribute__((target("no-packed-fp32-ops")))
#else
#define SDUMC_NO_PACKED_FP32      /* (the host pass does not know the feature) */
#endif
#include <type_traits>

#include "common.h"

namespace {

constexpr int D = SDUMC_D, H = SDUMC_H, NQ = SDUMC_NQ, RD = SDUMC_RNC_DIM;
constexpr int NTHR = 512, NWV = 8;
constexpr int PART_FLOATS = NWV * 7 * D;   // partial-sum area: 8 waves x up to 7 rows x 256 columns (or 2 rows x 896)

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

__device__ __forceinline__ DropRT mkdrop_rt(const DropRT& base, uint32_t site, uint32_t rows, uint32_t width) {
  DropRT d = base;     // resolved once per kernel (seed / call counter come from device memory)
  d.site = site;
  d.rows = rows;
  d.qwidth = width >> 2;
  d.bits = nullptr;
  return d;
}

// ------------------------------------------------------------------------------------------------------------------
// rows x matrix.  in_lds: [ROWS][ld_in] floats in LDS; M: global row-major [I][ldm]; result rows are handed, 4 columns at
// a time, to epi(r, col, f32x4) (every (r, col quad) exactly once, by some thread).  `part` is PART_FLOATS of LDS scratch.
// All 512 threads must call it (barriers inside); it ends with a barrier, so LDS written by epi is visible afterwards.
//
// The weight rows stream through a register ring DEP iterations deep (one iteration = 4 rows of M per lane group, 16-byte
// loads): DEP - 1 iterations stay in flight under the FMAs.  The ring is an argument: rxm_prefetch() issues a layer's first
// DEP - 1 iterations, rxm_run() consumes them -- and calls `hook` between its k-loop and its reduction, where the caller
// prefetches the NEXT layer's first rows, so that their latency (and the L2 miss of a cold weight matrix) passes under this
// layer's reduction, epilogue and barriers instead of at the head of the next one: the chain is a sequence of ~20 dependent
// layers per stage and that head latency was half of its time.
// ------------------------------------------------------------------------------------------------------------------
// WT = float, or unsigned short for bf16 weights (the engine's bf16-storage mode streams bf16 copies of the utterance-level
// weights: half the bytes of the stream this chain is bound by; accumulation stays fp32)
template <int I, int O, class WT>
struct RxmGeom {
  static constexpr int EPL = 16 / (int)sizeof(WT);      // weight elements (output columns) per lane per 16-byte load
  static constexpr int QPL = EPL / 4;                   // f32x4 accumulators per lane per block
  static constexpr int OG = O / EPL;                    // lane-groups of output columns
  static constexpr int OGW = OG >= 64 ? 64 : OG;        // lanes that cover one row of M
  static constexpr int S = 64 / OGW;                    // rows of M covered by one wave-load (1, 2, 4 or 8)
  static constexpr int NB = (OG + 63) / 64;             // column blocks of 64 lanes
  static constexpr int BW = 64 * EPL;                   // columns per block
  static constexpr int UNITS = I / (4 * S);             // 4-row groups of M per lane group
  // 8 waves, or 7 when that makes the per-wave trip count even where 8 leaves it odd (I = 896 at S = 4: 7 x 8 instead of 8 x 7,
  // so that the register ring can run 4 deep)
  static constexpr int WAVES = UNITS >= NWV ? ((UNITS % 8 == 0 && (UNITS / 8) % 2 == 1 && UNITS % 7 == 0 && (UNITS / 7) % 2 == 0) ? 7 : NWV)
                                            : UNITS;
  static constexpr int IW = I / WAVES;                  // rows of M per wave
  static constexpr int ITER = IW / (4 * S);
  // ring depth: 4 for fp32 single-block layers; bf16 lanes carry two accumulator quads per row, so their ring stays 2 deep
  // (deeper spilled at 14 rows)
  static constexpr int DEP = (ITER % 4 == 0 && NB == 1 && EPL == 4) ? 4 : ((ITER % 2 == 0) ? 2 : 1);
  // PF1 (opt-in, chain_cluster.hip): a single-iteration layer prefetches that one iteration too
  static constexpr bool ONE = ITER == 1;
  static_assert(I % (4 * S * WAVES) == 0 && ITER >= 1, "k range must split evenly");
  static_assert(S <= 8 && O % EPL == 0, "at least 8 lane-groups of output columns");
};
template <int NB>
struct WRing {
  uint4 w[4][4][NB];          // raw 16-byte loads (4 fp32 or 8 bf16), expanded when they are multiplied; a layer uses the first
                              // DEP (1, 2 or 4) slots, so that layers of different depth can hand one ring on to each other
};
template <int I, int O, class WT>
using RingOf = WRing<RxmGeom<I, O, WT>::NB>;

template <int I, int O, class WT, int NBv>
__device__ __forceinline__ void rxm_load(uint4 (&dst)[4][NBv], const WT* mp, int j, int ldm, int cg) {
  using G = RxmGeom<I, O, WT>;
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int b = 0; b < G::NB; ++b) {
      if (b * 64 + cg < G::OG) dst[e][b] = *reinterpret_cast<const uint4*>(mp + (size_t)(j * 4 * G::S + e) * ldm + b * G::BW);
      else dst[e][b] = uint4{0u, 0u, 0u, 0u};
    }
}

template <int I, int O, class WT, bool PF1 = false>
__device__ __forceinline__ void rxm_prefetch(RingOf<I, O, WT>& ring, const void* __restrict__ Mv, int ldm) {
  using G = RxmGeom<I, O, WT>;
  if constexpr (PF1 && G::ONE) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < G::WAVES) {
      const int sq = lane / G::OGW, cg = lane % G::OGW;
      const WT* mp = static_cast<const WT*>(Mv) + (size_t)(wave * G::IW + 4 * sq) * ldm + G::EPL * cg;
      rxm_load<I, O, WT, G::NB>(ring.w[0], mp, 0, ldm, cg);
    }
  } else if constexpr (G::DEP > 1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < G::WAVES) {
      const int sq = lane / G::OGW, cg = lane % G::OGW;
      const WT* mp = static_cast<const WT*>(Mv) + (size_t)(wave * G::IW + 4 * sq) * ldm + G::EPL * cg;
#pragma unroll
      for (int d = 0; d < G::DEP - 1; ++d) rxm_load<I, O, WT, G::NB>(ring.w[d], mp, d, ldm, cg);
    }
  }
}

template <int ROWS, int I, int O, class WT, bool PF1 = false, class Epi, class Hook>
__device__ __forceinline__ void rxm_run(RingOf<I, O, WT>& ring, const float* in_lds, int ld_in, const void* __restrict__ Mv, int ldm,
                                        float* part, Epi&& epi, Hook&& hook) {
  using G = RxmGeom<I, O, WT>;
  constexpr int OG = G::OG, OGW = G::OGW, S = G::S, NB = G::NB, WAVES = G::WAVES, IW = G::IW, ITER = G::ITER, DEP = G::DEP;
  constexpr int EPL = G::EPL, QPL = G::QPL, BW = G::BW;
  constexpr int RC = ROWS > 7 ? 7 : ROWS;        // rows per reduction round
  static_assert(NWV * RC * O <= PART_FLOATS, "partial-sum area too small");
  static_assert(ROWS % RC == 0, "rows per round");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sq = lane / OGW, cg = lane % OGW;

  f32x4 acc[ROWS][NB][QPL];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int q = 0; q < QPL; ++q) acc[r][b][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (wave < WAVES) {
    const int i0 = wave * IW + 4 * sq;
    const WT* mp = static_cast<const WT*>(Mv) + (size_t)i0 * ldm + EPL * cg;
    auto fma = [&](int j, const uint4 (&ws)[4][NB]) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const f32x4 x = ld4(in_lds + r * ld_in + i0 + j * 4 * S);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const uint4 u = ws[e][b];
            if constexpr (QPL == 1) {
              acc[r][b][0] += f32x4{__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)} * x[e];
            } else {        // 8 bf16: low halves are the even columns
              acc[r][b][0] += f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                                    __uint_as_float(u.y & 0xffff0000u)} * x[e];
              acc[r][b][1] += f32x4{__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16),
                                    __uint_as_float(u.w & 0xffff0000u)} * x[e];
            }
          }
        // (keeps the scheduler from hoisting every row's LDS read of several iterations to the top: with 14 rows that
        // alone is 56 live registers per iteration in flight, and the kernel spilled)
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (PF1 && G::ONE) {
      fma(0, ring.w[0]);                       // already loaded by rxm_prefetch<.., true>
    } else if constexpr (DEP == 1) {
#pragma unroll
      for (int j = 0; j < ITER; ++j) {
        rxm_load<I, O, WT, NB>(ring.w[0], mp, j, ldm, cg);
        fma(j, ring.w[0]);
      }
    } else {
#pragma unroll 1
      for (int jb = 0; jb < ITER; jb += DEP) {
#pragma unroll
        for (int d = 0; d < DEP; ++d) {
          const int j = jb + d;
          const int jn = j + DEP - 1 < ITER ? j + DEP - 1 : ITER - 1;     // past the end: a harmless re-load, no branch
          rxm_load<I, O, WT, NB>(ring.w[(d + DEP - 1) % DEP], mp, jn, ldm, cg);
          fma(j, ring.w[d]);
        }
      }
    }
    if constexpr (S >= 2) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int q = 0; q < QPL; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              acc[r][b][q][c] += __shfl_xor(acc[r][b][q][c], 32, 64);
              if constexpr (S >= 4) acc[r][b][q][c] += __shfl_xor(acc[r][b][q][c], 16, 64);
              if constexpr (S == 8) acc[r][b][q][c] += __shfl_xor(acc[r][b][q][c], 8, 64);
            }
    }
  }
  hook();        // the next layer's first weight rows start moving here
#pragma unroll
  for (int r0 = 0; r0 < ROWS; r0 += RC) {
    if (wave < WAVES && sq == 0) {
#pragma unroll
      for (int r = 0; r < RC; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          if (b * 64 + cg < OG) {
#pragma unroll
            for (int q = 0; q < QPL; ++q) st4(part + (wave * RC + r) * O + b * BW + EPL * cg + 4 * q, acc[r0 + r][b][q]);
          }
    }
    __syncthreads();
    for (int u = tid; u < RC * (O / 4); u += NTHR) {
      const int r = u / (O / 4), cq = u - r * (O / 4);
      f32x4 v = ld4(part + r * O + 4 * cq);
#pragma unroll
      for (int ww = 1; ww < WAVES; ++ww) v += ld4(part + (ww * RC + r) * O + 4 * cq);
      epi(r0 + r, 4 * cq, v);
    }
    __syncthreads();
  }
}

// stand-alone form (fp32 weights): prefetch + run, nothing chained behind it
template <int ROWS, int I, int O, class Epi>
__device__ __forceinline__ void rows_x_matrix(const float* in_lds, int ld_in, const float* __restrict__ M, int ldm, float* part,
                                              Epi&& epi) {
  RingOf<I, O, float> ring;
  rxm_prefetch<I, O, float>(ring, M, ldm);
  rxm_run<ROWS, I, O, float>(ring, in_lds, ld_in, M, ldm, part, epi, [] {});
}

// y = drop(relu(v + bias)) -> LDS and HBM (forward layer epilogue)
struct FwdEpi {
  const float* bias;
  float* out_lds;  int ld_lds;          // may be nullptr
  float* out_g;    int64_t ld_g;        // row r of this workgroup -> out_g + r * ld_g (already offset to the first row)
  bool relu;
  DropRT drop;                           // drop.enabled == 0: none
  uint32_t vrow0, vrow_stride;           // dropout row of local row r = vrow0 + r * vrow_stride ... see call sites
  __device__ __forceinline__ void operator()(int r, int col, f32x4 v) const {
    v += ld4(bias + col);
    if (relu) {
      v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    if (drop.enabled) v *= drop_mask4(drop, vrow0 + (uint32_t)r * vrow_stride, (uint32_t)(col >> 2));
    if (out_lds) st4(out_lds + r * ld_lds + col, v);
    if (out_g) st4(out_g + (int64_t)r * ld_g + col, v);
  }
};

// dx = v (+ add) masked by the saved post-dropout output y of the layer below: [y > 0] * scale (backward epilogue)
struct BwdEpi {
  const float* add_lds;  int ld_add;     // optional term added before the mask (LDS), may be nullptr
  const float* y_lds;    int ld_y;       // optional mask source (LDS), may be nullptr (= plain)
  float scale;
  float* out_lds;  int ld_lds;
  float* out_g;    int64_t ld_g;
  __device__ __forceinline__ void operator()(int r, int col, f32x4 v) const {
    if (add_lds) v += ld4(add_lds + r * ld_add + col);
    if (y_lds) {
      const f32x4 y = ld4(y_lds + r * ld_y + col);
      v[0] = y[0] > 0.f ? v[0] * scale : 0.f;
      v[1] = y[1] > 0.f ? v[1] * scale : 0.f;
      v[2] = y[2] > 0.f ? v[2] * scale : 0.f;
      v[3] = y[3] > 0.f ? v[3] * scale : 0.f;
    }
    if (out_lds) st4(out_lds + r * ld_lds + col, v);
    if (out_g) st4(out_g + (int64_t)r * ld_g + col, v);
  }
};

// rows [v0, v0 + R) of a global [.., width] tensor -> LDS [R][width] (rows beyond V are zero-filled)
template <int R>
__device__ __forceinline__ void load_rows(float* dst_lds, const float* src, int64_t ld, int width, int v0, int V) {
  const int q = width >> 2;
  for (int u = threadIdx.x; u < R * q; u += NTHR) {
    const int r = u / q, c = u - r * q;
    st4(dst_lds + r * width + 4 * c, v0 + r < V ? ld4(src + (int64_t)(v0 + r) * ld + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f});
  }
}

}  // namespace
