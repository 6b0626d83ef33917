// gemm_bf16.hip — the frame-level GEMMs of the SDUMC step on bf16 STORAGE (BASELINE configs[2] / configs[4]).
//
// In bf16 mode (sdumc_net_dims.bf16 = 2) the pre-extracted features, the projected frames, the masked frames xd, the tanh
// keys and the frame-level gradients dz / dxd / dx live in HBM as bf16; products accumulate in fp32 on
// v_mfma_f32_32x32x16_bf16, bias / tanh / the softmax and pooling / every loss / Adam stay fp32.  What runs here:
//   NT  C[M,N] = A[M,K] . B[N,K]^T : frame_dim_reshape (model :282-284; A = features), input_proj of FRA2UTT_new /
//       Cross_Attention (model :60, :82; A = xd, + bias + tanh), and their dX = dz . W (B = the transposed weight copy,
//       accumulated onto the pooling-path gradient)
//   TN  C[M,N] = A[K,M]^T . B[K,N] : the weight gradients dW = dz^T xd and dW_frame = dx^T features (split-K, fp32 slabs,
//       bias gradient = column sums of dz fused)
// No dropout logic in these kernels: in bf16 mode the engine materialises xd = drop(x) once per (site, stream)
// (sdumc_mask_apply_bf16), so every consumer reads plain bf16 rows.
//
// Structure = gemm_wide.hip's: operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds) through a ring of k-tiles,
// one raw barrier per k-tile with a counted vmcnt.  bf16 specifics:
//   * k-contiguous operands (NT): LDS rows of 64 bf16 = 128 bytes, the 16-byte chunk index XOR-swizzled by row bits on the
//     DMA's per-lane SOURCE address and again at the ds_read_b128 that fetches a lane's 8 consecutive k (one MFMA operand);
//   * row-contiguous operands (TN; k is the row index of both dz and x): the tile lands as [k][row] and the MFMA operand --
//     8 consecutive k of one row -- comes out of two ds_read_b64_tr_b16 (the gfx950 transposing LDS read: a 16-lane group
//     reads a 4 x 16 block and each lane receives one column).
// These kernels are bound by the operand stream (at bf16 MFMA rates a 128x128 tile needs ~8x the bytes per cycle of its fp32
// twin), so the tiles are the largest that keep two workgroups per CU in flight and the ring is what hides the latency.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace sdumc_bf16 {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int waitcnt_vm(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0xF << 8); }

__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 h = (__bf16)f;      // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  return *reinterpret_cast<unsigned short*>(&h);
}

template <int BM_, int BN_, int WGM_, int WGN_, int BK_, int NST_, bool KC_, int OCC_>
struct HCfg {
  static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_, BK = BK_, NST = NST_, OCC = OCC_;
  static constexpr bool KC = KC_;                       // true: NT (both operands k-contiguous); false: TN (both row-contiguous)
  static constexpr int NW = WGM * WGN, NTHR = 64 * NW;
  static constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  static constexpr int A_CH = A_BYTES / 1024, B_CH = B_BYTES / 1024;
  static constexpr int NCH = A_CH + B_CH, NI = NCH / NW;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = NST * STAGE_BYTES;
  static constexpr int NKS = BK / 16;                   // MFMA k-steps per k-tile
  static_assert(NCH % NW == 0, "pieces must divide evenly over the waves");
  static_assert(!KC || BK == 64, "k-contiguous tiles are built for 128-byte rows");
};

}  // namespace sdumc_bf16

// arguments (also the C ABI of sdumc_gemm_bf16_run, include/sdumc_hip.h)
using sdumc_bf16::HCfg;

namespace sdumc_bf16 {

// chunk swizzle of a [row][64 bf16] tile (8 chunks of 16 bytes per 128-byte row; two rows share a 256-byte bank row)
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <class CF, bool CS>
__global__ __launch_bounds__(CF::NTHR, CF::OCC) void gemm_bf16_kernel(const sdumc_gemm_bf16 g, const int nsplit, const int kchunk) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = CF::BM, BN = CF::BN, BK = CF::BK, NST = CF::NST, NW = CF::NW, TM = CF::TM, TN = CF::TN, NKS = CF::NKS;
  constexpr bool KC = CF::KC;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm0 = (wave / CF::WGN) * CF::WM, wn0 = (wave % CF::WGN) * CF::WN;
  // (tile order: see xcd_tile in common.h -- the n-tiles of an m panel behind one XCD's L2)
  const int tlin = xcd_tile(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int tile_m = tlin / (int)gridDim.x, tile_n = tlin - tile_m * (int)gridDim.x;
  const int grp = blockIdx.z / nsplit, ks_ = blockIdx.z - grp * nsplit;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = ks_ * kchunk, kend = min(g.K, kbeg + kchunk);
  const int nk = (kend - kbeg + BK - 1) / BK;

  const int a_rows = KC ? (g.a_row_mod > 0 ? g.a_row_mod : g.M) : g.K;
  const int b_rows = KC ? g.N : (g.b_row_mod > 0 ? g.b_row_mod : g.K);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A[grp]), 0, (int)min((size_t)a_rows * g.lda * 2, (size_t)0xFFFFFFF0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B[grp]), 0, (int)min((size_t)b_rows * g.ldb * 2, (size_t)0xFFFFFFF0u), 0x00020000);

  // per-lane byte offsets of this wave's 1-KiB pieces (pieces 0 .. A_CH-1 belong to A)
  uint32_t voff[CF::NI];
  int srck[CF::NI];
#pragma unroll
  for (int i = 0; i < CF::NI; ++i) {
    const int piece = wave + i * NW;
    const bool isA = piece < CF::A_CH;
    const int q = ((isA ? piece : piece - CF::A_CH) << 6) + lane;          // 16-byte chunk (8 bf16) inside the tile
    srck[i] = 0;
    if constexpr (KC) {
      const int row = q >> 3, cp = q & 7, c = cp ^ swz(row);
      int r = isA ? min(m0 + row, g.M - 1) : min(n0 + row, g.N - 1);
      if (isA && g.a_row_mod > 0) r %= g.a_row_mod;
      voff[i] = ((uint32_t)r * (uint32_t)(isA ? g.lda : g.ldb) + (uint32_t)(kbeg + 8 * c)) * 2u;
    } else {
      const int BR = isA ? BM : BN;
      const int RPC = BR / 8;                                                 // chunks per k row
      const int krow = q / RPC, c = q % RPC;
      const int col = isA ? min(m0 + 8 * c, g.M - 8) : min(n0 + 8 * c, g.N - 8);
      int kr = min(kbeg + krow, g.K - 1);
      if (!isA && g.b_row_mod > 0) kr %= g.b_row_mod;
      srck[i] = kr;
      voff[i] = ((uint32_t)kr * (uint32_t)(isA ? g.lda : g.ldb) + (uint32_t)col) * 2u;
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
      if constexpr (KC) {
        voff[i] += BK * 2;
      } else {
        const uint32_t ld = isA ? (uint32_t)g.lda : (uint32_t)g.ldb;
        const int mod = isA ? 0 : g.b_row_mod;
        int nxt = srck[i] + BK;
        if (mod > 0) {
          voff[i] += (uint32_t)BK * ld * 2u;
          if (nxt >= mod) { nxt -= mod; voff[i] -= (uint32_t)mod * ld * 2u; }
          srck[i] = nxt;
        } else {
          const int clamped = min(nxt, g.K - 1);      // beyond the last row: stay on it (the tail iteration zeroes those k)
          voff[i] += (uint32_t)(clamped - srck[i]) * ld * 2u;
          srck[i] = clamped;
        }
      }
    }
  };
  constexpr int PER = CF::NI;

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

  // one MFMA operand = 8 consecutive k of one row: lane (row = li, k half = lh) of k-step s
  auto frag = [&](const char* tile, int rowbase, int s, int BR) -> bf16x8 {
    if constexpr (KC) {
      const int row = rowbase + li;
      return *reinterpret_cast<const bf16x8*>(tile + row * 128 + 16 * ((2 * s + lh) ^ swz(row)));
    } else {
      // tile is [k][BR] bf16; two transposing reads of 4 k each.  16-lane group: its lanes 4q + p address row (k0 + q),
      // columns 4p .. 4p + 3 of the 4 x 16 block; lane i of the group receives column i, the 4 rows in its 4 elements.
      const int g16 = (lane >> 4) & 1, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
      const int k0 = 16 * s + 8 * lh;
      const char* a0 = tile + ((k0 + q) * BR + rowbase + 16 * g16 + 4 * p) * 2;
      const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
      const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * BR * 2));
      s16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      return *reinterpret_cast<bf16x8*>(&v);
    }
  };
  auto compute = [&](const char* base, int k0, auto tail_c) {
    constexpr bool TAIL = decltype(tail_c)::value;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = frag(base, wm0 + 32 * i, s, BM);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = frag(base + CF::A_BYTES, wn0 + 32 * j, s, BN);
      if constexpr (TAIL) {          // k >= kend counts as zero (row-contiguous operands: the last k-tile of a K range)
        const int kk = k0 + 16 * s + 8 * lh;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (kk + e >= kend) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i][e] = (__bf16)0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j][e] = (__bf16)0.f;
          }
        }
      }
      if constexpr (CS) {
        if (do_cs) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) csum[i] += (float)af[i][e];
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

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
    if (!KC && k0 + BK > kend) compute(lds + buf * CF::STAGE_BYTES, k0, std::true_type{});
    else compute(lds + buf * CF::STAGE_BYTES, k0, std::false_type{});
    buf = buf + 1 == NST ? 0 : buf + 1;
    ibuf = ibuf + 1 == NST ? 0 : ibuf + 1;
  }

  const bool to_slab = nsplit > 1;
  if constexpr (CS) {
    if (do_cs && g.colsum_a[grp] != nullptr) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float v = csum[i] + __shfl_xor(csum[i], 32, 64);
        const int m = m0 + wm0 + 32 * i + li;
        if (lh == 0 && m < g.M) {
          if (to_slab) g.workspace[(size_t)g.groups * nsplit * g.M * g.N + ((size_t)grp * nsplit + ks_) * g.M + m] = v;
          else g.colsum_a[grp][m] = g.accumulate ? g.colsum_a[grp][m] + v : v;
        }
      }
    }
  }
  // epilogue: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  const float* bias = to_slab ? nullptr : g.bias[grp];
  const int act = to_slab ? SDUMC_ACT_NONE : g.act;
  const bool accum = !to_slab && g.accumulate;
  const bool cbf = !to_slab && g.c_bf16;
  float* Cf = to_slab ? g.workspace + ((size_t)grp * nsplit + ks_) * (size_t)g.M * g.N : static_cast<float*>(g.C[grp]);
  unsigned short* Ch = static_cast<unsigned short*>(g.C[grp]);
  const int ldc = to_slab ? g.N : g.ldc;
  if (cbf && (ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(Ch) & 15) == 0) {
    // bf16 output: a lane owns one COLUMN of the accumulator tile, i.e. 2-byte stores 64 bytes apart per instruction.  Turn the
    // tile through LDS (32 rows at a time, the k-loop's ring is free by now) so that a lane stores 8 consecutive columns = 16 bytes.
    constexpr int LDT = CF::WN + 4;                         // floats per transposed row (keeps the b128 reads 16-byte aligned)
    static_assert(NW * 32 * LDT * 4 <= CF::LDS_BYTES, "epilogue staging must fit the ring");
    float* tw = reinterpret_cast<float*>(lds) + wave * 32 * LDT;
    __syncthreads();                                        // every wave is done reading the last k-tile
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) tw[((e & 3) + 8 * (e >> 2) + 4 * lh) * LDT + 32 * j + li] = acc[i][j][e];
      __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0): this wave's own LDS writes (no other wave reads them)
      constexpr int CPRW = CF::WN / 8;                      // 16-byte output chunks per row of the wave tile
#pragma unroll
      for (int u = lane; u < 32 * CPRW; u += 64) {
        const int r = u / CPRW, cq = u - r * CPRW;
        const int row = m0 + wm0 + 32 * i + r, col = n0 + wn0 + 8 * cq;
        if (row < g.M && col < g.N) {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(tw + r * LDT + 8 * cq), a1 = *reinterpret_cast<const f32x4*>(tw + r * LDT + 8 * cq + 4);
          float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          unsigned short* dst = Ch + (size_t)row * ldc + col;
          uint4 old4 = {0u, 0u, 0u, 0u};
          if (accum) old4 = *reinterpret_cast<const uint4*>(dst);
          const unsigned ow[4] = {old4.x, old4.y, old4.z, old4.w};
          unsigned pk[4];
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            float x = v[c] + (bias ? bias[col + c] : 0.f);
            if (act == SDUMC_ACT_TANH) x = fast_tanh(x);
            else if (act == SDUMC_ACT_RELU) x = fmaxf(x, 0.f);
            if (accum) x += bf2f((unsigned short)(ow[c >> 1] >> (16 * (c & 1))));
            const unsigned h = f2bf(x);
            if (c & 1) pk[c >> 1] |= h << 16; else pk[c >> 1] = h;
          }
          *reinterpret_cast<uint4*>(dst) = uint4{pk[0], pk[1], pk[2], pk[3]};
        }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);                   // reads done before the next 32 rows overwrite the staging
    }
  } else {
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
          const size_t o = (size_t)row * ldc + col;
          if (cbf) {
            if (accum) v += bf2f(Ch[o]);
            Ch[o] = f2bf(v);
          } else {
            if (accum) v += Cf[o];
            Cf[o] = v;
          }
        }
      }
  }
#endif
}

// ordered reduction of the split-K slabs (fp32 C only; dW)
__global__ __launch_bounds__(256) void bf16_splitk_reduce_kernel(const sdumc_gemm_bf16 g, const int nsplit) {
  const int grp = blockIdx.y;
  const size_t mn = (size_t)g.M * g.N;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= mn) {
    const size_t m = idx - mn;
    if (g.colsum_a[grp] != nullptr && m < (size_t)g.M) {
      const float* cs = g.workspace + (size_t)g.groups * nsplit * mn + (size_t)grp * nsplit * g.M + m;
      float v = 0.f;
      for (int z = 0; z < nsplit; ++z) v += cs[(size_t)z * g.M];
      float* dst = g.colsum_a[grp] + m;
      *dst = g.accumulate ? *dst + v : v;
    }
    return;
  }
  const float* s = g.workspace + (size_t)grp * nsplit * mn + idx;
  float v = 0.f;
  for (int z = 0; z < nsplit; ++z) v += s[(size_t)z * mn];
  const int row = (int)(idx / g.N), col = (int)(idx - (size_t)row * g.N);
  if (g.bias[grp]) v += g.bias[grp][col];
  if (g.act == SDUMC_ACT_TANH) v = fast_tanh(v);
  else if (g.act == SDUMC_ACT_RELU) v = fmaxf(v, 0.f);
  if (g.c_bf16) {
    unsigned short* dst = static_cast<unsigned short*>(g.C[grp]) + (size_t)row * g.ldc + col;
    if (g.accumulate) v += bf2f(*dst);
    *dst = f2bf(v);
  } else {
    float* dst = static_cast<float*>(g.C[grp]) + (size_t)row * g.ldc + col;
    if (g.accumulate) v += *dst;
    *dst = v;
  }
}

using NT128 = HCfg<128, 128, 2, 2, 64, 2, true, 2>;     // 64 KiB LDS: two workgroups per CU
using NT64 = HCfg<64, 128, 1, 4, 64, 3, true, 2>;       // 72 KiB: for M below ~16k rows (more tiles)
using TN128w = HCfg<128, 128, 2, 4, 64, 2, false, 4>;   // 8 waves (wave tile 64x32): two waves per SIMD from one workgroup

struct Plan {
  int nsplit, kchunk;
};
inline size_t ws_bytes(const sdumc_gemm_bf16& g, int nsplit) {
  if (nsplit <= 1) return 0;
  bool cs = false;
  for (int i = 0; i < g.groups; ++i) cs |= g.colsum_a[i] != nullptr;
  return ((size_t)nsplit * g.groups * (size_t)g.M * g.N + (cs ? (size_t)nsplit * g.groups * g.M : 0)) * sizeof(float);
}
inline bool nt_small(const sdumc_gemm_bf16& g) { return g.M < 16384; }
inline Plan plan(const sdumc_gemm_bf16& g, size_t have) {
  // Split K over workgroups (fp32 slabs + an ordered reduce) when the output has too few tiles to fill the chip:
  // always for TN (dW: M = 256 rows, K = tens of thousands), for NT only on few-row / long-K problems (the text-slot
  // frame projections: 2048 x 256 x 4096).  Slab traffic is capped at 16 MB per launch: beyond that the reduce costs
  // more than the extra parallelism gives (keys dW at 187 slices: 93 us; at 64: see profiles/README.md).
  const int bk = 64;
  const int bm = g.layout == SDUMC_TN ? 128 : (nt_small(g) ? 64 : 128);
  Plan p{1, ((g.K + bk - 1) / bk) * bk};
  const long tiles = (long)((g.M + bm - 1) / bm) * ((g.N + 127) / 128) * g.groups;
  const int kt = (g.K + bk - 1) / bk;
  int s;
  if (g.splitk >= 1) s = std::min(g.splitk, kt);
  else {
    const long by_fill = 512 / std::max<long>(1, tiles);
    const long by_slab = (16L << 20) / std::max<long>(1, (long)g.M * g.N * 4 * g.groups);
    s = (int)std::max<long>(1, std::min<long>(std::min<long>(kt / 8, by_fill), by_slab));
    if (g.layout == SDUMC_NT && tiles >= 128) s = 1;
  }
  while (s > 1 && ws_bytes(g, s) > have) --s;
  p.kchunk = ((kt + s - 1) / s) * bk;
  p.nsplit = (kt * bk + p.kchunk - 1) / p.kchunk;
  if (p.nsplit < 1) p.nsplit = 1;
  return p;
}

template <class CF>
int launch(const sdumc_gemm_bf16& g, const Plan& p, bool cs, hipStream_t st) {
  const dim3 grid((g.N + CF::BN - 1) / CF::BN, (g.M + CF::BM - 1) / CF::BM, g.groups * p.nsplit);
  const size_t shm = CF::LDS_BYTES;
#define SDUMC_H_LAUNCH(CSV)                                                                                                  \
  do {                                                                                                                       \
    static sdumc_dev_once attr_set;                                                                                          \
    if (sdumc_once_per_device(attr_set, [&] { return sdumc_set_dyn_lds(&gemm_bf16_kernel<CF, CSV>, shm); }) != SDUMC_OK)     \
      return SDUMC_ELAUNCH;                                                                                                  \
    hipLaunchKernelGGL((gemm_bf16_kernel<CF, CSV>), grid, dim3(CF::NTHR), shm, st, g, p.nsplit, p.kchunk);                   \
  } while (0)
  if (cs) SDUMC_H_LAUNCH(true);
  else SDUMC_H_LAUNCH(false);
#undef SDUMC_H_LAUNCH
  return SDUMC_OK;
}

}  // namespace sdumc_bf16

extern "C" int sdumc_prof_begin_(int variant, double flops, void* stream);     // gemm_f32.hip: bench.py's per-launch HIP events
extern "C" void sdumc_prof_end_(int token, void* stream);

extern "C" size_t sdumc_gemm_bf16_workspace_bytes(const sdumc_gemm_bf16* g) {
  if (!g || g->M <= 0 || g->N <= 0 || g->K <= 0 || g->groups < 1) return 0;
  return sdumc_bf16::ws_bytes(*g, sdumc_bf16::plan(*g, (size_t)-1).nsplit);
}

extern "C" int sdumc_gemm_bf16_run(const sdumc_gemm_bf16* gp, void* stream) {
  using namespace sdumc_bf16;
  if (!gp) return SDUMC_EINVAL;
  const sdumc_gemm_bf16& g = *gp;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.groups < 1 || g.groups > SDUMC_MAX_GROUPS) return SDUMC_EINVAL;
  if (g.layout != SDUMC_NT && g.layout != SDUMC_TN) return SDUMC_EINVAL;
  const bool nt = g.layout == SDUMC_NT;
  // 16-byte rows everywhere: leading dimensions and the contiguous extents are multiples of 8 bf16
  if ((g.lda & 7) || (g.ldb & 7) || (g.N & 7) || (nt ? (g.K & 7) : (g.M & 7))) return SDUMC_EINVAL;
  if (nt && (g.K % 64)) return SDUMC_EINVAL;                   // whole k-tiles (K = 256 / 1024 / 4096 on this path)
  if (!nt && (g.c_bf16 || g.act != SDUMC_ACT_NONE || g.a_row_mod)) return SDUMC_EINVAL;
  if (nt && g.b_row_mod) return SDUMC_EINVAL;
  if (g.b_row_mod > 0 && g.b_row_mod < 32) return SDUMC_EINVAL;
  bool cs = false;
  for (int i = 0; i < g.groups; ++i) {
    if (!g.A[i] || !g.B[i] || !g.C[i]) return SDUMC_EINVAL;
    if ((reinterpret_cast<uintptr_t>(g.A[i]) | reinterpret_cast<uintptr_t>(g.B[i])) & 15) return SDUMC_EINVAL;
    if (g.colsum_a[i]) {
      if (nt) return SDUMC_EINVAL;
      cs = true;
    }
  }
  const size_t a_bytes = (size_t)(nt ? (g.a_row_mod > 0 ? g.a_row_mod : g.M) : g.K) * g.lda * 2;
  const size_t b_bytes = (size_t)(nt ? g.N : (g.b_row_mod > 0 ? g.b_row_mod : g.K)) * g.ldb * 2;
  if (a_bytes >= 0xFFFFFFF0u || b_bytes >= 0xFFFFFFF0u) return SDUMC_EINVAL;
  const Plan p = plan(g, g.workspace ? g.workspace_bytes : 0);
  if (p.nsplit > 1 && (!g.workspace || g.workspace_bytes < ws_bytes(g, p.nsplit))) return SDUMC_ENOMEM;
  hipStream_t st = as_stream(stream);
  const int tok = sdumc_prof_begin_(nt ? 17 : 18, 2.0 * g.M * (double)g.N * g.K * g.groups, stream);
  int rc;
  if (nt) rc = nt_small(g) ? launch<NT64>(g, p, false, st) : launch<NT128>(g, p, false, st);
  else {
    // measured on MI355X (tools/gemm_bf16_check.py; frame dW 256 x 1024 x 24000 / keys dW 256 x 256 x 48000 / 4096^3):
    //   4 waves, BK 32, 3 stages: 70 / 58 us / 482 TF;  4 waves, BK 64, 2 stages: 62 / 53 us / 610 TF;
    //   8 waves (wave tile 64x32), BK 64, 2 stages: 55 / 47 us / 646 TF  <- the one kept
    rc = launch<TN128w>(g, p, cs, st);
  }
  if (rc != SDUMC_OK) return rc;
  SDUMC_CHECK_LAUNCH();
  sdumc_prof_end_(tok, stream);
  if (p.nsplit > 1) {
    const size_t mn = (size_t)g.M * g.N + (size_t)g.M;
    hipLaunchKernelGGL(bf16_splitk_reduce_kernel, dim3((unsigned)((mn + 255) / 256), g.groups), dim3(256), 0, st, g, p.nsplit);
    SDUMC_CHECK_LAUNCH();
  }
  return SDUMC_OK;
}

// One empty kernel per source file = per gfx950 code object: sdumc_preload_() asks for its attributes, which makes the HIP runtime load
// this file's code object NOW (outside any timed or latency-sensitive region) instead of at the first launch of one of its kernels --
// with deferred loading that first launch stalls the host for tens of milliseconds (seen as a 36-59 ms gap in the middle of an epoch,
// at the first batch whose shape took a fallback path: profiles/README.md, round 6).
__global__ void sdumc_preload_gemm_bf16_kernel() {}
extern "C" int sdumc_preload_gemm_bf16_(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&sdumc_preload_gemm_bf16_kernel)) == hipSuccess ? SDUMC_OK : SDUMC_ELAUNCH;
}
