// gemm_wide.hip — the big fp32 GEMMs of the SDUMC step on a wide-tile, LDS-DMA-fed MFMA loop (gfx950).
//
// What runs here (C2 shapes): the frame projections frame_dim_reshape_{0,1,2} (model :282-284; M = B*T up to 24000,
// N = 256, K = 1024 / 4096), the key projections input_proj of FRA2UTT_new / Cross_Attention (model :60, :82;
// M = 2*B*T up to 48000, N = K = 256, input dropout + bias + tanh fused), their dX (the same kernel on the transposed
// weight copies the engine keeps) and their dW (TN, split-K) -- 121 of the step's 127 GFLOP.
//
// Structure (why it differs from gemm_f32.hip's 64x64 register-staged loop):
//   * operands go global -> LDS with buffer_load_dwordx4 ... lds (LDS-DMA): no staging VGPRs, no ds_write pass, the k-tiles
//     of a 3-deep LDS ring stay in flight across the ONE barrier per k-tile (counted s_waitcnt vmcnt, raw s_barrier);
//   * a wave owns a 64x64 (or 32x64) block of the output = 4 (2) independent 32x32x2 fp32 MFMA accumulator chains, so a
//     fragment read from LDS feeds two MFMAs and the matrix pipe never waits on one dependent chain;
//   * k-contiguous operands ([rows][K]: activations, weights) land in LDS as [row][BK] with the 16-byte chunk index XORed by
//     row bits (the DMA writes lane-linearly, so the permutation is applied to the per-lane SOURCE address and again at the
//     ds_read_b128 fragment read: conflict-free); row-contiguous operands ([K][rows]: both operands of dW = dz^T x) land as
//     [k][row] and are read with conflict-free ds_read_b32;
//   * the input-dropout keep-bits (one byte per 4 elements) ride the same ring as their own small tile and are applied to
//     the fragments after the LDS read; the bias gradient (column sums of dz) is accumulated from the A fragments.
// Every fusion gemm_f32.hip offers on these shapes is kept; anything else (ragged K, unaligned operands, epilogue dropout,
// strided batches, bf16) stays on gemm_f32.hip -- sdumc_gemm_wide_ returns 1 ("not mine") and the caller falls through.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "common.h"

// (a named namespace: hipFuncSetAttribute takes the kernels' addresses, which needs external linkage on the host side)
namespace sdumc_wide {

typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int waitcnt_vm(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0xF << 8); }

__device__ __forceinline__ float fast_tanh(float x) {
  // 1 - 2 / (1 + e^{2x}): v_exp_f32 + v_rcp_f32, absolute error ~1e-7 (saturates correctly at +-1, NaN propagates)
  return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x));
}

template <int BM_, int BN_, int WGM_, int WGN_, int BK_, int NST_, bool A_KC_, bool B_KC_, int OCC_>
struct WideCfg {
  static constexpr int OCC = OCC_;          // waves per SIMD the register allocation must leave room for
  static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_, BK = BK_, NST = NST_;
  static constexpr bool A_KC = A_KC_, B_KC = B_KC_;
  static constexpr int NW = WGM * WGN, NTHR = 64 * NW;
  static constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
  static constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 4;
  static constexpr int A_CH = A_BYTES / 1024, B_CH = B_BYTES / 1024;     // 1-KiB pieces = one wave-instruction each
  static constexpr int NCH = A_CH + B_CH;
  static constexpr int NI = NCH / NW;                                    // data pieces per wave per stage
  // keep-bits tile: NT (mask on the k-contiguous A): [BM][BK/4] bytes; TN (mask on the row-contiguous B): [BK][BN/4] bytes
  static constexpr int BITS_BYTES = A_KC ? BM * (BK / 4) : BK * (BN / 4);
  static constexpr int BITS_CH = (BITS_BYTES + 255) / 256;               // 256-byte pieces (4 bytes per lane)
  static constexpr int NBI = (BITS_CH + NW - 1) / NW;                    // bits pieces per wave per stage (duplicates allowed)
  static constexpr int BITS_LDS = BITS_CH * 256;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES + BITS_LDS;
  static constexpr int LDS_BYTES = NST * STAGE_BYTES;
  static_assert(NCH % NW == 0, "pieces must divide evenly over the waves (vmcnt bookkeeping)");
  static_assert(WM % 32 == 0 && WN % 32 == 0 && BK % 8 == 0, "tile shape");
};

// swizzle of the 16-byte chunk index inside a [row][BK] tile: rows that share a 256-byte bank row get distinct slots
template <int BK>
__device__ __forceinline__ int swz(int row) {
  constexpr int CPR = BK / 4;                 // chunks per row
  constexpr int SH = BK == 32 ? 1 : (BK == 16 ? 2 : 3);
  return (row >> SH) & (CPR - 1);
}

// SPLIT (NT configurations): every fp32 product on the bf16 matrix pipe from operands split exactly into three bf16 parts -- six
// v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16 block instead of eight v_mfma_f32_32x32x2_f32, at the fp32 kernel's accuracy
// (gemm_group.hip, "fp32 products on the bf16 pipe", has the arithmetic).
template <class CF, bool MASK, bool CS, bool SPLIT>
__device__ __forceinline__ void gemm_wide_body(const sdumc_gemm& g, const int nsplit, const int kchunk, char* lds) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub: the body uses gfx950-only types (__amdgpu_buffer_rsrc_t)
  constexpr int BM = CF::BM, BN = CF::BN, BK = CF::BK, NST = CF::NST, NW = CF::NW, TM = CF::TM, TN = CF::TN;
  constexpr bool A_KC = CF::A_KC, B_KC = CF::B_KC;
  static_assert(!SPLIT || (A_KC && B_KC && BK == 16 && !CS), "the split form is written for the NT configurations");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm0 = (wave / CF::WGN) * CF::WM, wn0 = (wave % CF::WGN) * CF::WN;

  // work order: the n-tiles of an m panel (they read the same A rows) are consecutive tiles of ONE XCD (xcd_tile): with
  // blockIdx.x = n fastest they alternated between XCDs and every L2 fetched the panel again (FETCH_SIZE of the frame
  // projections: 163 MB per launch against 79 MB of operands)
  const int tiles_n_ = gridDim.x;
  const int tlin = xcd_tile(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int tile_m = tlin / tiles_n_, tile_n = tlin - tile_m * tiles_n_;
  const int gz = blockIdx.z / nsplit, ks = blockIdx.z - gz * nsplit;
  const int grp = gz;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = ks * kchunk, kend = min(g.K, kbeg + kchunk);
  const int nk = (kend - kbeg + BK - 1) / BK;

  const float* Ap = g.A[grp];
  const float* Bp = g.B[grp];
  // rows of the A / B matrices in memory (for the descriptors' range): KC [R][K], RC [K][R]
  const int a_rows = A_KC ? (g.a_row_mod > 0 ? g.a_row_mod : g.M) : (g.K);
  const int b_rows = B_KC ? g.N : (g.b_row_mod > 0 ? g.b_row_mod : g.K);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ap), 0, (int)min((size_t)a_rows * g.lda * 4, (size_t)0xFFFFFFF0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bp), 0, (int)min((size_t)b_rows * g.ldb * 4, (size_t)0xFFFFFFF0u), 0x00020000);

  // ---- per-lane source offsets (bytes) of this wave's pieces; pieces 0..A_CH-1 belong to A, the rest to B ----
  uint32_t voff[CF::NI];
  int srck[CF::NI];          // row-contiguous operands with a row modulo: current source k row of this lane's piece
#pragma unroll
  for (int i = 0; i < CF::NI; ++i) {
    const int piece = wave + i * NW;
    const int q = ((piece < CF::A_CH ? piece : piece - CF::A_CH) << 6) + lane;     // 16-byte chunk index inside the tile
    srck[i] = 0;
    if (piece < CF::A_CH) {
      if constexpr (A_KC) {
        constexpr int CPR = BK / 4;
        const int row = q / CPR, cp = q % CPR, c = cp ^ swz<BK>(row);
        int r = min(m0 + row, g.M - 1);
        if (g.a_row_mod > 0) r %= g.a_row_mod;
        voff[i] = ((uint32_t)r * (uint32_t)g.lda + (uint32_t)(kbeg + 4 * c)) * 4u;
      } else {
        constexpr int RPC = BM / 4;
        const int krow = q / RPC, c = q % RPC;
        const int col = min(m0 + 4 * c, g.M - 4);
        srck[i] = min(kbeg + krow, g.K - 1);
        voff[i] = ((uint32_t)srck[i] * (uint32_t)g.lda + (uint32_t)col) * 4u;
      }
    } else {
      if constexpr (B_KC) {
        constexpr int CPR = BK / 4;
        const int row = q / CPR, cp = q % CPR, c = cp ^ swz<BK>(row);
        const int r = min(n0 + row, g.N - 1);
        voff[i] = ((uint32_t)r * (uint32_t)g.ldb + (uint32_t)(kbeg + 4 * c)) * 4u;
      } else {
        constexpr int RPC = BN / 4;
        const int krow = q / RPC, c = q % RPC;
        const int col = min(n0 + 4 * c, g.N - 4);
        int kr = min(kbeg + krow, g.K - 1);
        if (g.b_row_mod > 0) kr %= g.b_row_mod;
        srck[i] = kr;
        voff[i] = ((uint32_t)kr * (uint32_t)g.ldb + (uint32_t)col) * 4u;
      }
    }
  }
  // keep-bits pieces (4 bytes per lane)
  const uint8_t* bitsp = nullptr;
  uint32_t qw = 0;
  float mscale = 1.f;
  if constexpr (MASK) {
    const sdumc_dropout& dd = A_KC ? g.a_drop : g.b_drop;
    bitsp = g.ab_drop_bits[grp] ? g.ab_drop_bits[grp] : dd.bits;
    qw = (dd.width + 3u) >> 2;
    mscale = dd.scale;
  }
  const int bits_rows = A_KC ? g.M : g.K;
  const __amdgpu_buffer_rsrc_t rbits = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(MASK ? bitsp : (const uint8_t*)Ap), 0,
                                                                        MASK ? (int)min((size_t)bits_rows * qw, (size_t)0xFFFFFFF0u) : 0, 0x00020000);
  uint32_t bvoff[CF::NBI];
  if constexpr (MASK) {
#pragma unroll
    for (int i = 0; i < CF::NBI; ++i) {
      const int piece = (wave + i * NW) % CF::BITS_CH;
      const int idx = (piece << 6) + lane;                  // dword index inside the bits tile
      if constexpr (A_KC) {
        constexpr int DPR = BK / 16;                        // dwords per row
        const int row = idx / DPR, dw = idx % DPR;
        bvoff[i] = (uint32_t)min(m0 + row, g.M - 1) * qw + (uint32_t)(kbeg >> 2) + 4u * dw;
      } else {
        constexpr int DPK = BN / 16;                        // dwords per k row
        const int krow = idx / DPK, dw = idx % DPK;
        bvoff[i] = (uint32_t)min(kbeg + krow, g.K - 1) * qw + (uint32_t)(n0 >> 2) + 4u * dw;
      }
    }
  }

  auto issue = [&](int buf) {
    char* base = lds + buf * CF::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < CF::NI; ++i) {
      const int piece = wave + i * NW;
      const bool isA = piece < CF::A_CH;
      char* dst = isA ? base + piece * 1024 : base + CF::A_BYTES + (piece - CF::A_CH) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? ra : rb, (lds_void_t*)dst, 16, voff[i], 0, 0, 0);
      // advance to the next k-tile
      if (isA ? A_KC : B_KC) {
        voff[i] += BK * 4;
      } else {
        const uint32_t ld = isA ? (uint32_t)g.lda : (uint32_t)g.ldb;
        const int mod = isA ? 0 : g.b_row_mod;
        const int lim = g.K - 1;
        int nk_ = srck[i] + BK;
        if (mod > 0) {
          voff[i] += (uint32_t)BK * ld * 4u;
          if (nk_ >= mod) { nk_ -= mod; voff[i] -= (uint32_t)mod * ld * 4u; }
          srck[i] = nk_;
        } else {
          // beyond the last row: stay on it (the tail iteration zeroes what it reads from there)
          const int clamped = min(nk_, lim);
          voff[i] += (uint32_t)(clamped - srck[i]) * ld * 4u;
          srck[i] = clamped;
        }
      }
    }
    if constexpr (MASK) {
#pragma unroll
      for (int i = 0; i < CF::NBI; ++i) {
        const int piece = (wave + i * NW) % CF::BITS_CH;
        char* dst = base + CF::A_BYTES + CF::B_BYTES + piece * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rbits, (lds_void_t*)dst, 4, bvoff[i], 0, 0, 0);
        if constexpr (A_KC) bvoff[i] += BK / 4;
        else bvoff[i] += (uint32_t)BK * qw;      // (rows beyond K re-read in-range bytes or get 0 from the range check: masked out anyway)
      }
    }
  };
  constexpr int PER = CF::NI + (MASK ? CF::NBI : 0);      // vector-memory operations per wave per stage

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float csum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) csum[i] = 0.f;
  const bool do_cs = CS && wn0 == 0 && tile_n == 0;

  // fragments of MFMA group gq (8 k): element s of a fragment is k = 8 gq + 4 lh + s
  auto read_a = [&](const char* base, int gq, f32x4 (&af)[TM]) {
    const float* As = reinterpret_cast<const float*>(base);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm0 + 32 * i + li;
      if constexpr (A_KC) {
        af[i] = *reinterpret_cast<const f32x4*>(As + row * BK + 4 * ((2 * gq + lh) ^ swz<BK>(row)));
        if constexpr (MASK) {
          const uint32_t b = reinterpret_cast<const uint8_t*>(base + CF::A_BYTES + CF::B_BYTES)[row * (BK / 4) + 2 * gq + lh];
          af[i][0] = (b & 1u) ? af[i][0] * mscale : 0.f;
          af[i][1] = (b & 2u) ? af[i][1] * mscale : 0.f;
          af[i][2] = (b & 4u) ? af[i][2] * mscale : 0.f;
          af[i][3] = (b & 8u) ? af[i][3] * mscale : 0.f;
        }
      } else {
        const float* p = As + (8 * gq + 4 * lh) * BM + row;
        af[i][0] = p[0];
        af[i][1] = p[BM];
        af[i][2] = p[2 * BM];
        af[i][3] = p[3 * BM];
      }
    }
  };
  auto read_b = [&](const char* base, int gq, f32x4 (&bf)[TN]) {
    const float* Bs = reinterpret_cast<const float*>(base + CF::A_BYTES);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wn0 + 32 * j + li;
      if constexpr (B_KC) {
        bf[j] = *reinterpret_cast<const f32x4*>(Bs + row * BK + 4 * ((2 * gq + lh) ^ swz<BK>(row)));
      } else {
        const float* p = Bs + (8 * gq + 4 * lh) * BN + row;
        bf[j][0] = p[0];
        bf[j][1] = p[BN];
        bf[j][2] = p[2 * BN];
        bf[j][3] = p[3 * BN];
        if constexpr (MASK && !A_KC) {
          const uint8_t* bt = reinterpret_cast<const uint8_t*>(base + CF::A_BYTES + CF::B_BYTES) + (8 * gq + 4 * lh) * (BN / 4) + (row >> 2);
          const uint32_t bit = 1u << (row & 3);
#pragma unroll
          for (int s = 0; s < 4; ++s) bf[j][s] = (bt[s * (BN / 4)] & bit) ? bf[j][s] * mscale : 0.f;
        }
      }
    }
  };
  // TAIL: the k-tile runs past kend (row-contiguous operands only): operands at k >= kend count as zero
  auto compute = [&](const char* base, int k0, auto tail_c) {
    constexpr bool TAIL = decltype(tail_c)::value;
    f32x4 af[2][TM], bf[2][TN];
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
      for (int s = 0; s < 4; ++s) {
        if constexpr (TAIL) {
          const bool live = k0 + 8 * gq + 4 * lh + s < kend;
#pragma unroll
          for (int i = 0; i < TM; ++i) af[cur][i][s] = live ? af[cur][i][s] : 0.f;
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[cur][j][s] = live ? bf[cur][j][s] : 0.f;
        }
        if constexpr (CS) {
          if (do_cs) {
#pragma unroll
            for (int i = 0; i < TM; ++i) csum[i] += af[cur][i][s];
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][s], bf[cur][j][s], acc[i][j], 0, 0, 0);
      }
    }
  };

  // the k-tile (16 k) as ONE bf16 MFMA depth: a lane's eight k of an operand are its two fp32 fragments side by side
  // (k = 4 lh + s and 8 + 4 lh + s), the same assignment for A and B
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
    f32x4 af[2][TM], bf[2][TN];
    read_a(base, 0, af[0]);
    read_a(base, 1, af[1]);
    read_b(base, 0, bf[0]);
    read_b(base, 1, bf[1]);
    u32x4 pa[TM][3], pb[TN][3];
    auto op = [](const u32x4& v) { return __builtin_bit_cast(bf16x8, v); };
#pragma unroll
    for (int j = 0; j < TN; ++j) split8(bf[0][j], bf[1][j], pb[j]);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      split8(af[0][i], af[1][i], pa[i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) {   // smallest terms first
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(pa[i][2]), op(pb[j][0]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(pa[i][0]), op(pb[j][2]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(pa[i][1]), op(pb[j][1]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(pa[i][1]), op(pb[j][0]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(pa[i][0]), op(pb[j][1]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(pa[i][0]), op(pb[j][0]), acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- the ring: stage t lives in buffer t % NST; NST - 1 stages are in flight ahead of the one being multiplied ----
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue(s);
  int buf = 0, ibuf = NST - 1;
  for (int t = 0; t < nk; ++t) {
    if (t + NST - 2 < nk) __builtin_amdgcn_s_waitcnt(waitcnt_vm((NST - 2) * PER));
    else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    __builtin_amdgcn_s_barrier();
    if (t + NST - 1 < nk) issue(ibuf);
    const int k0 = kbeg + t * BK;
    if constexpr (SPLIT) compute_split(lds + buf * CF::STAGE_BYTES);
    else if ((!A_KC || !B_KC) && k0 + BK > kend) compute(lds + buf * CF::STAGE_BYTES, k0, std::true_type{});
    else compute(lds + buf * CF::STAGE_BYTES, k0, std::false_type{});
    buf = buf + 1 == NST ? 0 : buf + 1;
    ibuf = ibuf + 1 == NST ? 0 : ibuf + 1;
  }

  const bool to_slab = nsplit > 1;
  // ---- fused column sums of A (TN: the bias gradient) ----
  if constexpr (CS) {
    if (do_cs && g.colsum_a[grp] != nullptr) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float v = csum[i] + __shfl_xor(csum[i], 32, 64);
        const int m = m0 + wm0 + 32 * i + li;
        if (lh == 0 && m < g.M) {
          if (to_slab) {
            g.workspace[(size_t)g.groups * nsplit * g.M * g.N + ((size_t)grp * nsplit + ks) * g.M + m] = v;
          } else {
            float* dst = g.colsum_a[grp] + m;
            *dst = g.accumulate ? *dst + v : v;
          }
        }
      }
    }
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) ----
  float* C = to_slab ? g.workspace + ((size_t)grp * nsplit + ks) * (size_t)g.M * g.N : g.C[grp];
  const int ldc = to_slab ? g.N : g.ldc;
  const float* bias = to_slab ? nullptr : g.bias[grp];
  const int act = to_slab ? SDUMC_ACT_NONE : g.act;
  const bool accum = !to_slab && g.accumulate;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn0 + 32 * j + li;
      if (col >= g.N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row >= g.M) continue;
        float v = acc[i][j][e] + bv;
        if (act == SDUMC_ACT_TANH) v = fast_tanh(v);
        else if (act == SDUMC_ACT_RELU) v = fmaxf(v, 0.f);
        float* dst = C + (size_t)row * ldc + col;
        if (accum) v += *dst;
        *dst = v;
      }
    }
#endif
}
template <class CF, bool MASK, bool CS>
__global__ __launch_bounds__(CF::NTHR, CF::OCC) void gemm_wide_kernel(const sdumc_gemm g, const int nsplit, const int kchunk) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  gemm_wide_body<CF, MASK, CS, false>(g, nsplit, kchunk, lds);
}
template <class CF, bool MASK>
__global__ __launch_bounds__(CF::NTHR, CF::OCC) void gemm_wide_split_kernel(const sdumc_gemm g, const int nsplit, const int kchunk) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  gemm_wide_body<CF, MASK, false, true>(g, nsplit, kchunk, lds);
}

bool split_products() { return sdumc_split_on_(SDUMC_SPLIT_WIDE) != 0; }

// (A persistent NT variant -- a workgroup walking a range of output tiles with the LDS ring running on across tile boundaries,
//  DMA issue interleaved between the MFMAs, staggered starts -- was built and measured in round 2: within +-2 % of this
//  kernel on every shape of the step; removed in round 3, the numbers are in profiles/README.md.)
template <class CF>
int launch_cfg(const sdumc_gemm& g, int nsplit, int kchunk, bool mask, bool cs, hipStream_t st) {
  const dim3 grid((g.N + CF::BN - 1) / CF::BN, (g.M + CF::BM - 1) / CF::BM, g.groups * nsplit);
  const dim3 blk(CF::NTHR);
  const size_t shm = CF::LDS_BYTES;
#define SDUMC_WIDE_LAUNCH(MK, CSV)                                                                                           \
  do {                                                                                                                       \
    static sdumc_dev_once attr_set;                                                                                          \
    if (sdumc_once_per_device(attr_set, [&] { return sdumc_set_dyn_lds(&gemm_wide_kernel<CF, MK, CSV>, shm); }) != SDUMC_OK) \
      return SDUMC_ELAUNCH;                                                                                                  \
    hipLaunchKernelGGL((gemm_wide_kernel<CF, MK, CSV>), grid, blk, shm, st, g, nsplit, kchunk);                              \
  } while (0)
#define SDUMC_WIDE_LAUNCH_SPLIT(MK)                                                                                           \
  do {                                                                                                                       \
    static sdumc_dev_once attr_set;                                                                                          \
    if (sdumc_once_per_device(attr_set, [&] { return sdumc_set_dyn_lds(&gemm_wide_split_kernel<CF, MK>, shm); }) != SDUMC_OK) \
      return SDUMC_ELAUNCH;                                                                                                  \
    hipLaunchKernelGGL((gemm_wide_split_kernel<CF, MK>), grid, blk, shm, st, g, nsplit, kchunk);                             \
  } while (0)
  if constexpr (CF::A_KC) {       // NT: optional mask on A
    if (split_products()) {
      if (mask) SDUMC_WIDE_LAUNCH_SPLIT(true);
      else SDUMC_WIDE_LAUNCH_SPLIT(false);
    } else {
      if (mask) SDUMC_WIDE_LAUNCH(true, false);
      else SDUMC_WIDE_LAUNCH(false, false);
    }
  } else {                        // TN: optional mask on B, optional column sums of A
    if (mask) { if (cs) SDUMC_WIDE_LAUNCH(true, true); else SDUMC_WIDE_LAUNCH(true, false); }
    else { if (cs) SDUMC_WIDE_LAUNCH(false, true); else SDUMC_WIDE_LAUNCH(false, false); }
  }
#undef SDUMC_WIDE_LAUNCH
#undef SDUMC_WIDE_LAUNCH_SPLIT
  return SDUMC_OK;
}

// configurations: <BM, BN, waves along M, waves along N, BK, ring depth, A k-contiguous, B k-contiguous>
using NT_64x256 = WideCfg<64, 256, 1, 4, 16, 3, true, true, 2>;      // 60 KiB LDS: 2 workgroups per CU
using NT_128x256 = WideCfg<128, 256, 2, 4, 16, 3, true, true, 4>;    // 8 waves, 72 KiB: 2 per CU
using NT_128x128 = WideCfg<128, 128, 2, 2, 16, 3, true, true, 3>;    // 48 KiB: 3 per CU
using NT_64x128 = WideCfg<64, 128, 1, 4, 16, 3, true, true, 4>;      // waves 64x32, 36 KiB: 4 per CU
using TN_128x128 = WideCfg<128, 128, 2, 2, 16, 3, false, false, 2>;
using TN_64x128 = WideCfg<64, 128, 1, 4, 16, 3, false, false, 4>;
using TN_64x256 = WideCfg<64, 256, 1, 4, 16, 3, false, false, 2>;

}  // namespace sdumc_wide
using namespace sdumc_wide;

// cfg: 1 = 64x256, 2 = 128x256, 3 = 128x128, 4 = 64x128.  Returns SDUMC_OK when launched, 1 when the problem is not one this
// kernel takes (the caller then uses gemm_f32.hip's kernels), < 0 on errors.
extern "C" int sdumc_gemm_wide_(const sdumc_gemm* gp, int cfg, int nsplit, int kchunk, void* stream) {
  const sdumc_gemm& g = *gp;
  if (g.layout != SDUMC_NT && g.layout != SDUMC_TN) return 1;
  if (g.bf16 || g.batch > 1 || g.c_drop.enabled) return 1;
  for (int i = 0; i < g.groups; ++i)
    if (g.c_mask_y[i]) return 1;
  const bool nt = g.layout == SDUMC_NT;
  if (cfg < 1 || cfg > 4) return 1;
  const int bn = (cfg == 1 || cfg == 2) ? 256 : 128;
  const int bm = (cfg == 2 || cfg == 3) ? 128 : 64;
  if (g.N % bn || g.M < 4 || g.N < 4) return 1;
  if ((g.lda & 3) || (g.ldb & 3)) return 1;
  if (nt) {
    if (g.K % 16 || kchunk % 16 || g.b_drop.enabled || g.b_row_mod) return 1;
    for (int i = 0; i < g.groups; ++i)
      if (g.colsum_a[i]) return 1;
  } else {
    if ((g.M & 3) || (g.N & 3) || g.M % bm || g.a_drop.enabled || g.a_row_mod || kchunk % 16) return 1;
    if (g.b_row_mod > 0 && g.b_row_mod < 16) return 1;
  }
  const sdumc_dropout& dd = nt ? g.a_drop : g.b_drop;
  bool mask = dd.enabled != 0;
  if (mask) {
    if (dd.width & 15) return 1;
    for (int i = 0; i < g.groups; ++i) {
      const uint8_t* b = g.ab_drop_bits[i] ? g.ab_drop_bits[i] : dd.bits;
      if (!b || (reinterpret_cast<uintptr_t>(b) & 3)) return 1;     // Philox-in-the-loader calls stay on the generic kernel
    }
  }
  bool cs = false;
  for (int i = 0; i < g.groups; ++i) {
    if ((reinterpret_cast<uintptr_t>(g.A[i]) | reinterpret_cast<uintptr_t>(g.B[i])) & 15) return 1;
    cs |= g.colsum_a[i] != nullptr;
  }
  // 32-bit byte offsets inside one operand
  const size_t a_bytes = (size_t)(nt ? (g.a_row_mod > 0 ? g.a_row_mod : g.M) : g.K) * g.lda * 4;
  const size_t b_bytes = (size_t)(nt ? g.N : (g.b_row_mod > 0 ? g.b_row_mod : g.K)) * g.ldb * 4;
  if (a_bytes >= 0xFFFFFFF0u || b_bytes >= 0xFFFFFFF0u) return 1;
  hipStream_t st = as_stream(stream);
  int rc;
  if (nt) {
    switch (cfg) {
      case 1: rc = launch_cfg<NT_64x256>(g, nsplit, kchunk, mask, false, st); break;
      case 2: rc = launch_cfg<NT_128x256>(g, nsplit, kchunk, mask, false, st); break;
      case 3: rc = launch_cfg<NT_128x128>(g, nsplit, kchunk, mask, false, st); break;
      default: rc = launch_cfg<NT_64x128>(g, nsplit, kchunk, mask, false, st); break;
    }
  } else {
    switch (cfg) {
      case 1: rc = launch_cfg<TN_64x256>(g, nsplit, kchunk, mask, cs, st); break;
      case 3: rc = launch_cfg<TN_128x128>(g, nsplit, kchunk, mask, cs, st); break;
      case 4: rc = launch_cfg<TN_64x128>(g, nsplit, kchunk, mask, cs, st); break;
      default: return 1;
    }
  }
  if (rc != SDUMC_OK) return rc;
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

// One empty kernel per source file = per gfx950 code object: sdumc_preload_() asks for its attributes, which makes the HIP runtime load
// this file's code object NOW (outside any timed or latency-sensitive region) instead of at the first launch of one of its kernels --
// with deferred loading that first launch stalls the host for tens of milliseconds (seen as a 36-59 ms gap in the middle of an epoch,
// at the first batch whose shape took a fallback path: profiles/README.md, round 6).
__global__ void sdumc_preload_gemm_wide_kernel() {}
extern "C" int sdumc_preload_gemm_wide_(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&sdumc_preload_gemm_wide_kernel)) == hipSuccess ? SDUMC_OK : SDUMC_ELAUNCH;
}
