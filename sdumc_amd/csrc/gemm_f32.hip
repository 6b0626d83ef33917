// gemm_f32.hip — grouped fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces, on the SDUMC hot path, every torch Linear / bmm and its autograd:
//   frame_dim_reshape_{0,1,2}          model :193-195,282-284   (NT, M = B*T up to 24k, K = 1024/4096)
//   input_proj of FRA2UTT_new / Cross_Attention (model :60,:82)  (NT + fused input dropout + tanh)
//   every utterance-level Linear (model :264-273, :293-368)      (NT + bias + ReLU + dropout)
//   dX = dY.W (NN), dW = dY^T.X (TN, split-K) of all of the above (main :149 loss.backward())
//
// Design (MI355X-first):
//   * exact-fp32 MFMA 32x32x2: 64 FLOP/clk/SIMD = the fp32 roofline of the chip (155 TF measured).
//   * 256 threads = 4 waves (2x2); block tile 128x128 (wave 64x64 = 2x2 MFMA tiles, 64 accumulator
//     VGPRs) or 64x64 for the launch-bound utterance-level layers; BK = 32.
//   * global -> registers (16-B loads, prefetch of tile t+1 in flight during the MFMAs of tile t)
//     -> LDS -> fragments.  k-contiguous operands are staged as [row][BK+4]: the +4 pad makes the
//     ds_read_b128 fragment reads (4 k-steps per read) conflict-free; the k index inside an MFMA
//     group is permuted identically for A and B (lane half h supplies k = 8g + 4h + s), which any
//     product sum is invariant to.  row-contiguous operands (NN's B, TN's A and B) are staged as
//     [k][row] and read with conflict-free ds_read_b32.
//   * fusions: dropout (Philox, recomputed, never stored) on the staged A (NT/NN) or B (TN) operand,
//     source-row modulo (the two streams share x_audio/x_video), bias + ReLU/tanh + dropout epilogue,
//     accumulate, deterministic split-K (slabs + ordered reduce; no float atomics).
#include <algorithm>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"

namespace {

// ---- optional per-launch timing with HIP events on the launch stream (bench.py roofline leg) -----
struct ProfRec {
  hipEvent_t a, b;
  int variant;  // layout * 2 + (tile == 64x64)
  double flops;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::mutex g_prof_mu;
constexpr int kNumVariants = 29;
const char* const kVariantName[kNumVariants] = {"gemm_nt_128x128", "gemm_nt_64x64",  "gemm_nn_128x128",
                                                "gemm_nn_64x64",   "gemm_tn_128x128", "gemm_tn_64x64",
                                                "gemm_small_nt",   "gemm_small_nn",   "gemm_small_tn",
                                                "gemm_bf16_nt_128x128", "gemm_bf16_nt_64x64",
                                                "gemm_bf16_nn_128x128", "gemm_bf16_nn_64x64",
                                                "gemm_bf16_tn_128x128", "gemm_bf16_tn_64x64",
                                                "gemm_wide_nt", "gemm_wide_tn",
                                                "gemm_bf16s_nt", "gemm_bf16s_tn",    // gemm_bf16.hip (bf16 storage)
                                                "gemm_group_tn", "gemm_group_tn_bf16",    // gemm_group.hip (the persistent kernel alone, without its reduce)
                                                "gemm_rows256", "gemm_rows256_bf16",      // gemm_rows.hip
                                                // fp32 operands, products on the bf16 matrix pipe from exactly split operands (sdumc_set_split_)
                                                "gemm_wide_nt_bf16x3", "gemm_group_tn_bf16x3", "gemm_rows256_bf16x3",
                                                "gemm_p3_nt_bf16x3", "gemm_p3_nt_masked_bf16x3",
                                                "gemm_b1_nt_bf16"};     // gemm_p3.hip: operands pre-split into bf16 planes (P3 tensors); the two
                                                // instantiations rocprof lists too: plain (frame projections), keep-bits on A (key projections)

constexpr int BK = 32;
constexpr int LDK = BK + 4;
#ifndef SDUMC_GEMM_WPE128
#define SDUMC_GEMM_WPE128 2
#endif
#ifndef SDUMC_GEMM_WPE64
#define SDUMC_GEMM_WPE64 5   // waves per SIMD the 64x64 variants are register-allocated for (96 VGPRs, 1-2 spilled outside the k-loop): 2.197 vs 2.214 ms per step against 4
#endif

struct TileLoadCtx {
  const float* p;
  int ld;
  int row_mod;
  bool vec;
  DropRT drop;
};

// Load one [BR x BK] operand tile into registers.
// KC = true : matrix stored [R][K] (k contiguous): thread -> (row = tid/8 + 32 j, k = 4 (tid%8))
// KC = false: matrix stored [K][R] (R contiguous): thread -> (k = tid/(BR/4) + (1024/BR) j, r = 4 (tid % (BR/4)))
// "row" in the dropout / row_mod sense is the non-channel index: R index for KC, K index otherwise.
template <int BR, bool KC>
__device__ __forceinline__ void load_tile(f32x4 (&reg)[BR / 32], const TileLoadCtx& c, int r0, int R, int k0,
                                          int kend, int tid) {
  constexpr int P = BR / 32;
#pragma unroll
  for (int j = 0; j < P; ++j) {
    int row, ch, row_lim, ch_lim;
    if (KC) {
      row = r0 + (tid >> 3) + 32 * j;
      ch = k0 + 4 * (tid & 7);
      row_lim = R;
      ch_lim = kend;
    } else {
      constexpr int Q = BR / 4;
      constexpr int RP = 256 / Q;
      row = k0 + tid / Q + RP * j;
      ch = r0 + 4 * (tid % Q);
      row_lim = kend;
      ch_lim = R;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < row_lim && ch < ch_lim) {
      const int srow = c.row_mod > 0 ? row % c.row_mod : row;
      const float* src = c.p + (size_t)srow * c.ld + ch;
      if (c.vec) {
        v = *reinterpret_cast<const f32x4*>(src);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (ch + e < ch_lim) v[e] = src[e];
      }
      if (c.drop.enabled) {
        const f32x4 m = drop_mask4(c.drop, (uint32_t)row, (uint32_t)(ch >> 2));
        v *= m;
      }
    }
    reg[j] = v;
  }
}

// Fast staging path (aligned operands): everything that does not depend on k is computed ONCE per thread --
// element offsets relative to a wave-uniform base, keep-bits offsets, range flags -- and merely advanced by a
// constant per k-tile.  The generic load_tile above costs ~25 VALU instructions per 16-byte load (64-bit
// multiply-adds, bounds, an integer modulo for the shared-x mapping); with 4-5 waves per SIMD that filled the
// vector-issue slots beside the MFMAs (SQ counters: 6.4 VALU per MFMA).
// PAIRK (row-contiguous operands only): the thread's P loads are k-consecutive rows (k = P (tid / Q) + j) instead of RP
// apart, so that the bf16 variant can store k-pairs / k-quads of one row with one LDS write
template <int BR, bool KC, bool PAIRK = false>
struct Stager {
  static constexpr int P = BR / 32;
  static constexpr int Q = BR / 4, RP = 256 / Q;
  static constexpr int KSTEP = PAIRK ? 1 : RP;   // k distance between consecutive loads of one thread
  uint32_t off[P];    // element offset of this thread's next 16 bytes
  uint32_t boff[P];   // byte offset of the matching keep-bits
  int src[P];         // !KC with row_mod: current source row (to wrap)
  int pos;            // KC: this thread's current k;  !KC: current k row of j = 0
  uint32_t ok;        // bit j: the k-invariant coordinate is in range
  int vrow0;          // KC: this thread's first row / !KC: this thread's column
  uint32_t step, wrap;

  __device__ __forceinline__ void init(const TileLoadCtx& c, int r0, int R, int kbeg, int tid) {
    ok = 0;
    if (KC) {
      pos = kbeg + 4 * (tid & 7);
      vrow0 = r0 + (tid >> 3);
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int row = vrow0 + 32 * j;
        if (row < R) ok |= 1u << j;
        const int srow = c.row_mod > 0 ? row % c.row_mod : row;
        off[j] = (uint32_t)srow * (uint32_t)c.ld + (uint32_t)pos;
        boff[j] = (uint32_t)row * c.drop.qwidth + (uint32_t)(pos >> 2);
        src[j] = 0;
      }
      step = BK;
      wrap = 0;
    } else {
      pos = kbeg + (PAIRK ? P * (tid / Q) : tid / Q);
      vrow0 = r0 + 4 * (tid % Q);
      if (vrow0 < R) ok = (1u << P) - 1;
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int row = pos + KSTEP * j;
        src[j] = c.row_mod > 0 ? row % c.row_mod : row;
        off[j] = (uint32_t)src[j] * (uint32_t)c.ld + (uint32_t)vrow0;
        boff[j] = (uint32_t)row * c.drop.qwidth + (uint32_t)(vrow0 >> 2);
      }
      step = (uint32_t)BK * (uint32_t)c.ld;
      wrap = (uint32_t)c.row_mod * (uint32_t)c.ld;
    }
  }

  // Issues the 16-byte loads (and, with MASK, the keep-bits byte loads) of the next k-tile and returns WITHOUT touching
  // the loaded values: anything that reads them here puts an s_waitcnt vmcnt right behind the load and the prefetch
  // stops overlapping the MFMAs of the current tile.  The dropout mask is applied by apply(), just before the tile
  // is written to LDS one iteration later.
  // FULL: every tile of the problem is a full tile (M, N multiples of the block tile, K of BK): no range predicates,
  // i.e. no exec-mask save/restore around the loads
  template <bool MASK, bool FULL>
  __device__ __forceinline__ void load(f32x4 (&reg)[P], uint32_t (&mb)[P], const TileLoadCtx& c, int kend) {
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int kk = KC ? pos : pos + KSTEP * j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      uint32_t b = 0;
      if (FULL || (((ok >> j) & 1u) && kk < kend)) {
        v = *reinterpret_cast<const f32x4*>(c.p + off[j]);
        if (MASK) b = c.drop.bits[boff[j]];
      }
      reg[j] = v;
      mb[j] = b;
      off[j] += step;
      if (KC) {
        boff[j] += BK / 4;
      } else {
        boff[j] += (uint32_t)BK * c.drop.qwidth;
        if (c.row_mod > 0) {
          src[j] += BK;
          if (src[j] >= c.row_mod) {
            src[j] -= c.row_mod;
            off[j] -= wrap;
          }
        }
      }
    }
    pos += BK;
  }

  // keep-bits -> the four multiplicative mask values (out-of-range elements were loaded as 0 and stay 0)
  __device__ __forceinline__ static void apply(f32x4 (&reg)[P], const uint32_t (&mb)[P], float scale) {
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const uint32_t b = mb[j];
      reg[j][0] = (b & 1u) ? reg[j][0] * scale : 0.f;
      reg[j][1] = (b & 2u) ? reg[j][1] * scale : 0.f;
      reg[j][2] = (b & 4u) ? reg[j][2] * scale : 0.f;
      reg[j][3] = (b & 8u) ? reg[j][3] * scale : 0.f;
    }
  }
};

template <int BR, bool KC>
__device__ __forceinline__ void store_tile(float* lds, const f32x4 (&reg)[BR / 32], int tid) {
  constexpr int P = BR / 32;
#pragma unroll
  for (int j = 0; j < P; ++j) {
    if (KC) {
      const int row = (tid >> 3) + 32 * j;
      *reinterpret_cast<f32x4*>(lds + row * LDK + 4 * (tid & 7)) = reg[j];
    } else {
      constexpr int Q = BR / 4;
      constexpr int RP = 256 / Q;
      const int k = tid / Q + RP * j;
      *reinterpret_cast<f32x4*>(lds + k * BR + 4 * (tid % Q)) = reg[j];
    }
  }
}

// fragment of MFMA group g (8 consecutive k) for the 32 rows starting at `base`:
// element s (0..3) is operand k = 8g + 4h + s of row base + i  (i = lane&31, h = lane>>5)
template <int BR, bool KC>
__device__ __forceinline__ f32x4 read_frag(const float* lds, int base, int g, int li, int lh) {
  if (KC) {
    return *reinterpret_cast<const f32x4*>(lds + (base + li) * LDK + 8 * g + 4 * lh);
  } else {
    f32x4 v;
    const float* p = lds + (8 * g + 4 * lh) * BR + base + li;
    v[0] = p[0];
    v[1] = p[BR];
    v[2] = p[2 * BR];
    v[3] = p[3 * BR];
    return v;
  }
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == SDUMC_ACT_RELU) return fmaxf(v, 0.f);
  if (act == SDUMC_ACT_TANH) return tanhf(v);
  return v;
}

// BF16 = true: the staged operands are rounded to bf16 (v_cvt_pk_bf16_f32, RNE) on their way into LDS -- always as
// [row][k], so row-contiguous operands (NN's B, TN's A and B) are transposed by their LDS stores -- and multiplied on
// v_mfma_f32_32x32x16_bf16 -- 16x the fp32 MFMA rate -- with fp32 accumulation and the same
// fp32 epilogue.  This is the "bf16 compute" mode of BASELINE configs[2]; the default path is exact fp32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int LDH = BK + 8;   // bf16 elements per LDS row (80 B): the 16-B fragment reads are conflict-free

// registers -> bf16 LDS tile [row][LDH].  KC: the thread's four values are k-consecutive (one 8-byte store).
// !KC (the transpose): PAIRK -- the thread holds P consecutive k of four rows: one 4-byte (P = 2) or 8-byte (P = 4) store
// per row; otherwise (generic loader) four 2-byte stores per load
template <int BR, bool KC, bool PAIRK>
__device__ __forceinline__ void store_half(__bf16* H, const f32x4 (&reg)[BR / 32], int tid) {
  constexpr int P = BR / 32;
  if constexpr (KC) {
#pragma unroll
    for (int j = 0; j < P; ++j) {
      bf16x4 h = {(__bf16)reg[j][0], (__bf16)reg[j][1], (__bf16)reg[j][2], (__bf16)reg[j][3]};
      *reinterpret_cast<bf16x4*>(H + ((tid >> 3) + 32 * j) * LDH + 4 * (tid & 7)) = h;
    }
  } else {
    constexpr int Q = BR / 4, RP = 256 / Q;
    const int r = 4 * (tid % Q);
    if constexpr (PAIRK) {
      const int k0 = P * (tid / Q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (P == 2) {
          typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
          bf16x2 h = {(__bf16)reg[0][e], (__bf16)reg[1][e]};
          *reinterpret_cast<bf16x2*>(H + (r + e) * LDH + k0) = h;
        } else {
          bf16x4 h = {(__bf16)reg[0][e], (__bf16)reg[1][e], (__bf16)reg[2][e], (__bf16)reg[3][e]};
          *reinterpret_cast<bf16x4*>(H + (r + e) * LDH + k0) = h;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int k = tid / Q + RP * j;
#pragma unroll
        for (int e = 0; e < 4; ++e) H[(r + e) * LDH + k] = (__bf16)reg[j][e];
      }
    }
  }
}

template <int BM, int BN, bool A_K, bool B_K, bool BF16 = false>
__global__ __launch_bounds__(256, (BM == 64 ? SDUMC_GEMM_WPE64 : SDUMC_GEMM_WPE128)) void gemm_kernel(const sdumc_gemm g, const int nsplit, const int kchunk) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int A_ELEMS = A_K ? BM * LDK : BK * BM;
  constexpr int B_ELEMS = B_K ? BN * LDK : BK * BN;
  __shared__ __attribute__((aligned(16))) float lds[A_ELEMS + B_ELEMS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  // XCD-aware work order.  The dispatcher deals workgroups round-robin over the 8 XCDs (launch-order id % 8), each
  // with its own L2.  In launch order the workgroups that share an operand panel -- the N/BN tiles of one A row
  // panel, the M/BM x N/BN tiles of one split-K slice or of one (sample, head) batch entry -- land on 8 different
  // XCDs and each pulls the panel through the fabric again (rocprofv3 FETCH_SIZE: 3.6x the algorithmic bytes on the
  // split-K dW GEMMs).  Renumber: XCD x works on the x-th contiguous run of (slice-major, n-fastest) tiles, so that
  // sharers sit behind one L2 and run back to back.  Bijective for any grid (cdna_hip_programming.md, T1).
  int tile_m = blockIdx.y, tile_n = blockIdx.x, slice = blockIdx.z;
  {
    const int plane = gridDim.x * gridDim.y, total = plane * gridDim.z;
    if (total >= 16) {
      const int id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      const int xcd = id & 7, j = id >> 3, q = total >> 3, r = total & 7;
      const int t = xcd * q + min(xcd, r) + j;
      slice = t / plane;
      const int rem = t - slice * plane;
      tile_m = rem / gridDim.x;
      tile_n = rem - tile_m * gridDim.x;
    }
  }
  // slice = ((group * batch) + batch entry) * nsplit + k-split
  const int gz = slice / nsplit, ks = slice - gz * nsplit;
  const int nb = g.batch > 1 ? g.batch : 1;
  const int grp = gz / nb, bz = gz - grp * nb;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = ks * kchunk;
  const int kend = min(g.K, kbeg + kchunk);

  TileLoadCtx ca, cb;
  ca.p = g.A[grp] + (size_t)bz * g.stride_a;
  ca.ld = g.lda;
  ca.row_mod = g.a_row_mod;
  cb.p = g.B[grp] + (size_t)bz * g.stride_b;
  cb.ld = g.ldb;
  cb.row_mod = (!A_K && !B_K) ? g.b_row_mod : 0;
  // 16-byte vector loads need aligned rows and a channel extent that is a multiple of 4
  {
    const int a_ch = A_K ? g.K : g.M;
    const int b_ch = B_K ? g.K : g.N;
    ca.vec = ((g.lda & 3) == 0) && ((a_ch & 3) == 0) && ((reinterpret_cast<uintptr_t>(ca.p) & 15) == 0);
    cb.vec = ((g.ldb & 3) == 0) && ((b_ch & 3) == 0) && ((reinterpret_cast<uintptr_t>(cb.p) & 15) == 0);
  }
  ca.drop = drop_resolve(g.a_drop);
  cb.drop = drop_resolve(g.b_drop);
  if (!A_K) ca.drop.enabled = 0;          // a_drop is defined for row-major [M][K] A only
  if (A_K || B_K) cb.drop.enabled = 0;    // b_drop is defined for TN only
  ca.drop.site += (uint32_t)(grp * g.ab_drop_group_stride);   // grouped launches: per-group site / keep-bits
  cb.drop.site += (uint32_t)(grp * g.ab_drop_group_stride);
  if (g.ab_drop_bits[grp]) ca.drop.bits = cb.drop.bits = g.ab_drop_bits[grp];

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // TN only: column sums of A (= the bias gradient when A is dz) ride along with the staging loads of
  // the first n-tile: every A element passes through exactly one thread of those workgroups.
  const bool do_cs = !A_K && g.colsum_a[grp] != nullptr && tile_n == 0;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};

  // 32-bit element offsets: fine for every operand below 2^32 floats (16 GiB).  A fused dropout without precomputed
  // keep-bits (Philox recomputed in the loader: tests, one-off calls) takes the generic path.
  const bool mask_a = ca.drop.enabled != 0, mask_b = cb.drop.enabled != 0;
  const bool fast = ca.vec && cb.vec && BK < (ca.row_mod > 0 ? ca.row_mod : BK + 1) &&
                    BK < (cb.row_mod > 0 ? cb.row_mod : BK + 1) && (!mask_a || ca.drop.bits) && (!mask_b || cb.drop.bits);

  // One k-tile of MFMAs from the LDS stage.
  auto compute = [&](const float* As, const float* Bs) {
    if constexpr (BF16) {
      // lane (r = lane&31, h = lane>>5) holds A[row r][k = 16 ks + 8 h .. +7] and the same k range of B's row
      const __bf16* Ah = reinterpret_cast<const __bf16*>(As);
      const __bf16* Bh = reinterpret_cast<const __bf16*>(Bs);
      bf16x8 ah[BK / 16][TM], bh[BK / 16][TN];
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i) ah[ks][i] = *reinterpret_cast<const bf16x8*>(Ah + (wm0 + 32 * i + li) * LDH + 16 * ks + 8 * lh);
#pragma unroll
        for (int j = 0; j < TN; ++j) bh[ks][j] = *reinterpret_cast<const bf16x8*>(Bh + (wn0 + 32 * j + li) * LDH + 16 * ks + 8 * lh);
      }
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
    } else {
      // fragment reads are software-pipelined over two register sets: group g+1 is read from LDS while the
      // MFMAs of group g issue (with one set the compiler emitted read -> lgkmcnt(0) -> 4 MFMA, four times,
      // exposing the LDS latency of every group)
      f32x4 af[2][TM], bf[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[0][i] = read_frag<BM, A_K>(As, wm0 + 32 * i, 0, li, lh);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[0][j] = read_frag<BN, B_K>(Bs, wn0 + 32 * j, 0, li, lh);
#pragma unroll
      for (int gq = 0; gq < BK / 8; ++gq) {
        const int cur = gq & 1, nxt = cur ^ 1;
        if (gq + 1 < BK / 8) {
#pragma unroll
          for (int i = 0; i < TM; ++i) af[nxt][i] = read_frag<BM, A_K>(As, wm0 + 32 * i, gq + 1, li, lh);
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[nxt][j] = read_frag<BN, B_K>(Bs, wn0 + 32 * j, gq + 1, li, lh);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the reads above ahead of the MFMAs below (hipcc sinks them otherwise)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][s], bf[cur][j][s], acc[i][j], 0, 0, 0);
      }
    }
  };
  auto store_stage = [&](float* As, float* Bs, const f32x4 (&ra)[BM / 32], const f32x4 (&rb)[BN / 32], auto pair_c) {
    if constexpr (BF16) {
      constexpr bool PAIRK = decltype(pair_c)::value;
      store_half<BM, A_K, PAIRK>(reinterpret_cast<__bf16*>(As), ra, tid);
      store_half<BN, B_K, PAIRK>(reinterpret_cast<__bf16*>(Bs), rb, tid);
    } else {
      store_tile<BM, A_K>(As, ra, tid);
      store_tile<BN, B_K>(Bs, rb, tid);
    }
  };

  // The k-loop, instantiated once per combination of the block-uniform run-time switches
  //   FAST (aligned operands, hoisted addressing), MA / MB (keep-bits mask on the A / B operand), CS (fused column sums)
  // so that its body carries no scalar branches on them (the one loop that tested them per k-tile spent ~10 taken/not
  // taken s_cbranch per iteration and kept every variant's registers live).
  // Structure: global -> registers (prefetch of tile t+1 in flight during the MFMAs of tile t) -> LDS -> fragments.
  float* const As = lds;
  float* const Bs = lds + A_ELEMS;
  auto k_loop = [&](auto fast_c, auto ma_c, auto mb_c, auto cs_c) {
    constexpr int FASTV = decltype(fast_c)::value;   // 0 generic loader, 1 hoisted addressing, 2 the same on full tiles only
    constexpr bool FAST = FASTV > 0, FULL = FASTV == 2, MA = decltype(ma_c)::value, MB = decltype(mb_c)::value,
                   CS = decltype(cs_c)::value;
    Stager<BM, A_K, BF16 && !A_K> sa;     // bf16: row-contiguous operands are loaded as k-pairs (see store_half)
    Stager<BN, B_K, BF16 && !B_K> sb;
    if constexpr (FAST) {
      sa.init(ca, m0, g.M, kbeg, tid);
      sb.init(cb, n0, g.N, kbeg, tid);
    }
    f32x4 ra[BM / 32], rb[BN / 32];
    uint32_t ba[BM / 32], bb[BN / 32];
    // nothing reads the registers here: the loads stay in flight
    auto prefetch = [&](int k) {
      if constexpr (FAST) {
        sa.template load<MA, FULL>(ra, ba, ca, kend);
        sb.template load<MB, FULL>(rb, bb, cb, kend);
      } else {
        load_tile<BM, A_K>(ra, ca, m0, g.M, k, kend, tid);
        load_tile<BN, B_K>(rb, cb, n0, g.N, k, kend, tid);
      }
    };
    if (kbeg < kend) prefetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      // first use of the prefetched registers: dropout masks and the fused column sums are applied here, not at
      // load time
      if constexpr (FAST && MA) Stager<BM, A_K>::apply(ra, ba, ca.drop.scale);
      if constexpr (FAST && MB) Stager<BN, B_K>::apply(rb, bb, cb.drop.scale);
      constexpr bool PAIR = FAST && BF16;   // the generic loader keeps the RP-apart mapping
      if constexpr (CS) {
#pragma unroll
        for (int j = 0; j < BM / 32; ++j) csum += ra[j];
      }
      __syncthreads();  // everyone is done reading the previous tile
      store_stage(As, Bs, ra, rb, std::integral_constant<bool, PAIR>{});
      __syncthreads();
      if (k0 + BK < kend) prefetch(k0 + BK);   // in flight during the MFMAs below
      compute(As, Bs);
    }
  };
  {
    using T = std::true_type;
    using F = std::false_type;
    using G0 = std::integral_constant<int, 0>;
    using F1 = std::integral_constant<int, 1>;
    using F2 = std::integral_constant<int, 2>;
    const bool full = (g.M % BM) == 0 && (g.N % BN) == 0 && (g.K % BK) == 0;   // then every k-range is whole k-tiles too
    if constexpr (A_K) {          // NT / NN: optional mask on A
      if (!fast) k_loop(G0{}, F{}, F{}, F{});
      else if (mask_a) { if (full) k_loop(F2{}, T{}, F{}, F{}); else k_loop(F1{}, T{}, F{}, F{}); }
      else { if (full) k_loop(F2{}, F{}, F{}, F{}); else k_loop(F1{}, F{}, F{}, F{}); }
    } else {                      // TN: optional mask on B, optional column sums of A
      if (!fast) {
        if (do_cs) k_loop(G0{}, F{}, F{}, T{});
        else k_loop(G0{}, F{}, F{}, F{});
      } else if (full) {
        if (mask_b) { if (do_cs) k_loop(F2{}, F{}, T{}, T{}); else k_loop(F2{}, F{}, T{}, F{}); }
        else { if (do_cs) k_loop(F2{}, F{}, F{}, T{}); else k_loop(F2{}, F{}, F{}, F{}); }
      } else {
        if (mask_b) { if (do_cs) k_loop(F1{}, F{}, T{}, T{}); else k_loop(F1{}, F{}, T{}, F{}); }
        else { if (do_cs) k_loop(F1{}, F{}, F{}, T{}); else k_loop(F1{}, F{}, F{}, F{}); }
      }
    }
  }

  const bool to_slab = nsplit > 1;
  if (!A_K && g.colsum_a[grp] != nullptr && tile_n == 0) {   // block-uniform
    constexpr int Q = BM / 4, RP = 256 / Q;
    __syncthreads();                                  // the last tile's fragment reads are done
    *reinterpret_cast<f32x4*>(lds + (tid / Q) * BM + 4 * (tid % Q)) = csum;
    __syncthreads();
    if (tid < BM && m0 + tid < g.M) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < RP; ++r) sum += lds[r * BM + tid];
      if (to_slab) {
        g.workspace[(size_t)g.groups * nsplit * g.M * g.N + (size_t)slice * g.M + m0 + tid] = sum;
      } else {
        float* dst = g.colsum_a[grp] + m0 + tid;
        *dst = g.accumulate ? *dst + sum : sum;
      }
    }
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  float* C = to_slab ? g.workspace + ((size_t)slice) * (size_t)g.M * g.N : g.C[grp] + (size_t)bz * g.stride_c;
  const int ldc = to_slab ? g.N : g.ldc;
  const float* bias = to_slab ? nullptr : g.bias[grp];
  DropRT cd = drop_resolve(g.c_drop);
  cd.site += (uint32_t)(grp * g.c_drop_group_stride);
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
        float v = acc[i][j][e];
        float* dst = C + (size_t)row * ldc + col;
        if (!to_slab) {
          v = apply_act(v + bv, g.act);
          if (cd.enabled) v *= drop_mask1(cd, (uint32_t)row, (uint32_t)col);
          if (g.accumulate) v += *dst;
        }
        *dst = v;
      }
    }
}

// ------------------------------------------------------------------------------------------------
// Small-problem kernel (the utterance-level layers: M = 2B or 14B rows, N, K <= 896).  These GEMMs are
// latency-bound, not MFMA-bound: what matters is the length of the dependent chain inside one workgroup.
// One workgroup = one 32x32 output tile; its four waves split K between them (intra-workgroup split-K);
// each wave loads its operand slices straight from global memory into registers in MFMA fragment layout
// (no LDS staging, no barriers in the k-loop, several k-groups of loads in flight at once), then the four
// partial tiles are summed through LDS in a fixed order and the usual fused epilogue runs.
// ------------------------------------------------------------------------------------------------
template <bool A_K, bool B_K, int NW>
__device__ __forceinline__ void gemm_small_body(const sdumc_gemm& g, const int kq /* k per wave, multiple of 8 */, const int grp,
                                                const int by, const int bx) {
  __shared__ float part[NW][32 * 33];
  __shared__ float cs_s[NW][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = by * 32, n0 = bx * 32;
  const float* A = g.A[grp];
  const float* B = g.B[grp];
  const int K = g.K;
  const int kbeg = wave * kq, kend = min(K, kbeg + kq);
  const bool a_ok = m0 + li < g.M, b_ok = n0 + li < g.N;
  const bool a_vec = A_K && ((g.lda & 3) == 0) && ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  const bool b_vec = B_K && ((g.ldb & 3) == 0) && ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  // k-contiguous operand ([R][K]): this lane's row; row-contiguous operand ([K][R]): this lane's column
  const float* ap = A_K ? A + (size_t)(m0 + li) * g.lda : A + m0 + li;
  const float* bp = B_K ? B + (size_t)(n0 + li) * g.ldb : B + n0 + li;
  const bool do_cs = !A_K && g.colsum_a[grp] != nullptr && bx == 0;
  float csum = 0.f;

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;

  // one 8-k group: this lane supplies k .. k+3 (same k for A and B)
  auto load_group = [&](int k, f32x4& a, f32x4& b) {
    a = f32x4{0.f, 0.f, 0.f, 0.f};
    b = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a_ok) {
      if (A_K) {
        if (a_vec && k + 3 < kend) a = *reinterpret_cast<const f32x4*>(ap + k);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (k + e < kend) a[e] = ap[k + e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (k + e < kend) a[e] = ap[(size_t)(k + e) * g.lda];
      }
    }
    if (b_ok) {
      if (B_K) {
        if (b_vec && k + 3 < kend) b = *reinterpret_cast<const f32x4*>(bp + k);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (k + e < kend) b[e] = bp[k + e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (k + e < kend) b[e] = bp[(size_t)(k + e) * g.ldb];
      }
    }
  };
  constexpr int UB = 8;   // k-groups per batch: all of a batch's loads are in flight before its first MFMA
  for (int k0 = kbeg; k0 < kend; k0 += 8 * UB) {
    f32x4 a[UB], b[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) load_group(k0 + 8 * u + 4 * lh, a[u], b[u]);
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (do_cs) csum += (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], b[u][e], acc, 0, 0, 0);
    }
  }

  // partial tiles -> LDS; C/D layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
  for (int e = 0; e < 16; ++e) part[wave][((e & 3) + 8 * (e >> 2) + 4 * lh) * 33 + li] = acc[e];
  if (do_cs) {
    csum += __shfl_xor(csum, 32, 64);
    if (lane < 32) cs_s[wave][lane] = csum;
  }
  __syncthreads();
  if (do_cs && tid < 32 && m0 + tid < g.M) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += cs_s[w][tid];
    float* dst = g.colsum_a[grp] + m0 + tid;
    *dst = g.accumulate ? *dst + sum : sum;
  }
  float* C = g.C[grp];
  const float* bias = g.bias[grp];
  const float* cy = g.c_mask_y[grp];   // optional: C = C * [Y > 0] * scale, applied last (after accumulate)
  DropRT cd = drop_resolve(g.c_drop);
  cd.site += (uint32_t)(grp * g.c_drop_group_stride);
#pragma unroll
  for (int q = 0; q < 16 / NW; ++q) {
    const int o = tid + 64 * NW * q, r = o >> 5, c = o & 31;
    const int row = m0 + r, col = n0 + c;
    if (row >= g.M || col >= g.N) continue;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += part[w][r * 33 + c];   // fixed order: deterministic
    v = apply_act(v + (bias ? bias[col] : 0.f), g.act);
    if (cd.enabled) v *= drop_mask1(cd, (uint32_t)row, (uint32_t)col);
    float* dst = C + (size_t)row * g.ldc + col;
    if (g.accumulate) v += *dst;
    if (cy) v = cy[(size_t)row * g.ldc + col] > 0.f ? v * g.c_mask_scale : 0.f;
    *dst = v;
  }
}

template <bool A_K, bool B_K, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_small_kernel(const sdumc_gemm g, const int kq) {
  gemm_small_body<A_K, B_K, NW>(g, kq, blockIdx.z, blockIdx.y, blockIdx.x);
}
// up to three TN problems of DIFFERENT shapes in one launch (the weight gradients whose outputs have 3 / 7 / 1 rows: fc_att,
// cross_fc_att, fc_out_v -- three dependent 10-us launches in front of the grouped dW launch otherwise): workgroup -> (problem, tile)
struct SmallMulti {
  sdumc_gemm g[3];
  int32_t kq[3], blk_end[3], tiles_n[3];
};
__global__ __launch_bounds__(256) void gemm_small_tn_multi_kernel(const SmallMulti m) {
  sdumc_gemm g = m.g[0];      // uniform select, by value (a dynamically indexed kernel argument goes through scratch)
  int i = 0;
  if ((int)blockIdx.x >= m.blk_end[0]) { g = m.g[1]; i = 1; }
  if ((int)blockIdx.x >= m.blk_end[1]) { g = m.g[2]; i = 2; }
  const int kq = i == 0 ? m.kq[0] : (i == 1 ? m.kq[1] : m.kq[2]);
  const int tn = i == 0 ? m.tiles_n[0] : (i == 1 ? m.tiles_n[1] : m.tiles_n[2]);
  const int local = (int)blockIdx.x - (i == 0 ? 0 : (i == 1 ? m.blk_end[0] : m.blk_end[1]));
  gemm_small_body<false, false, 4>(g, kq, 0, local / tn, local - (local / tn) * tn);
}

// ordered (deterministic) reduction of the split-K slabs + the epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const sdumc_gemm g, const int nsplit) {
  const int grp = blockIdx.y;
  const size_t mn = (size_t)g.M * g.N;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= mn) {
    const size_t m = idx - mn;                       // tail threads: the fused column sums
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
  v = apply_act(v, g.act);
  if (g.c_drop.enabled) {
    DropRT cd = drop_resolve(g.c_drop);
    cd.site += (uint32_t)(grp * g.c_drop_group_stride);
    v *= drop_mask1(cd, (uint32_t)row, (uint32_t)col);
  }
  float* dst = g.C[grp] + (size_t)row * g.ldc + col;
  if (g.accumulate) v += *dst;
  *dst = v;
}

// small-problem kernel: 8 waves split K when each still gets >= 16 k (one batch of loads per wave up to K = 512)
#ifndef SDUMC_SMALL_WAVES_MAX
#define SDUMC_SMALL_WAVES_MAX 4   // 8 waves splitting K measured 1.5 % slower per step (2.289 vs 2.255 ms): more partial tiles to meet in LDS
#endif
inline int small_waves(int K) { return (SDUMC_SMALL_WAVES_MAX == 8 && K >= 128) ? 8 : 4; }

struct GemmPlan {
  int tile;    // 1 = 128x128, 2 = 64x64, 3 = small-problem kernel (32x32 tile, intra-workgroup split-K),
               // 11..14 = gemm_wide.hip's LDS-DMA kernels (64x256, 128x256, 128x128, 64x128)
  int nsplit;
  int kchunk;
  int waves = 4;   // tile 3 only
};

size_t plan_ws_bytes(const sdumc_gemm& g, int nsplit) {
  if (nsplit <= 1) return 0;
  bool cs = false;
  for (int i = 0; i < g.groups; ++i) cs |= g.colsum_a[i] != nullptr;
  return ((size_t)nsplit * g.groups * (size_t)g.M * g.N + (cs ? (size_t)nsplit * g.groups * g.M : 0)) * sizeof(float);
}

// Tile and split-K choice (measured on MI355X, tools/gemm_bench.py, profiles/README.md):
// the kernel core reaches ~110 TF at 4096^3 with either tile, but this path's shapes are skinny
// (N = 256; M*N is at most 376 tiles of 128x128) and there the 64x64 tile wins clearly
// (M=48000,K=256: 85 vs 56 TF; M=24000,K=1024: 91 vs 77 TF): four resident workgroups per CU hide each
// other's prologue/epilogue and the dynamic dispatch balances the tail.  So: 64x64 unless the problem has
// >= 8192 such tiles; split K until ~1024 workgroups (4 per CU) exist, but keep >= 6 k-tiles (K >= 192)
// per workgroup -- below that the extra reduce launch costs more than the shorter k-loop saves.
GemmPlan plan_gemm(const sdumc_gemm& g, size_t ws_bytes) {
  GemmPlan p;
  if (g.tile >= 11 && g.tile <= 18) {     // wide kernels (15..18: persistent NT variants): NT unsplit, TN split K over ~2 workgroups per CU
    p.tile = g.tile;
    const int wt = g.tile > 14 ? g.tile - 4 : g.tile;
    const int bm = (wt == 12 || wt == 13) ? 128 : 64, bn = (wt == 11 || wt == 12) ? 256 : 128;
    const long tiles = (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * g.groups;
    const int kt = (g.K + 15) / 16;
    int s = 1;
    if (g.splitk >= 1) s = std::min(g.splitk, kt);
    else if (g.layout == SDUMC_TN) s = (int)std::max<long>(1, std::min<long>(kt / 8, 512 / std::max<long>(1, tiles)));
    while (s > 1 && plan_ws_bytes(g, s) > ws_bytes) --s;
    p.nsplit = s;
    p.kchunk = ((kt + s - 1) / s) * 16;
    p.nsplit = (kt * 16 + p.kchunk - 1) / p.kchunk;
    if (p.nsplit <= 1) { p.nsplit = 1; p.kchunk = ((g.K + 15) / 16) * 16; }
    return p;
  }
  // Automatic use of gemm_wide.hip's LDS-DMA kernels where they measured faster than the 64x64 register-staged loop
  // (tools/gemm_wide_check.py on MI355X, profiles/README.md r2): skinny NT problems with N a multiple of 128 --
  // the frame projections (K >= 512: 64x128 tiles, 118 vs 95 TF at M = 24000, 104 vs 96 at M = 14400) and the key projections
  // of the long modality (K = 256, M >= 32768: 128x128 tiles, 88 vs 76 TF; 93 vs 82 grouped).  sdumc_gemm_wide_ itself
  // declines what it cannot take (ragged K, unaligned operands, epilogue dropout ...) and the call falls back to the plan below.
  if (g.tile == 0 && g.layout == SDUMC_NT && !g.bf16 && g.batch <= 1 && g.splitk <= 1 && (g.N % 128) == 0 && (g.K % 16) == 0 &&
      g.N <= 1024 && !g.c_drop.enabled) {
    int wide = 0;
    if (g.K >= 512 && g.M >= 8192) wide = 14;
    else if (g.K >= 128 && g.K < 512 && (long)g.M * g.groups >= 32768) wide = 13;
    if (wide) {
      sdumc_gemm g2 = g;
      g2.tile = wide;
      return plan_gemm(g2, ws_bytes);
    }
  }
  const int ktiles = (g.K + BK - 1) / BK;
  const long big = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * g.groups;
  const long small = (long)((g.M + 63) / 64) * ((g.N + 63) / 64) * g.groups;
  // wide problems (both M and N >= 1024, >= 4 M outputs: the generic transformer's projections and FFN) run 2-10 % faster on
  // 128x128 tiles (tools/gemm_tile_choice.py: 103 vs 93, 128 vs 117, 124 vs 114 TF); this path's N = 256 shapes stay on 64x64
  const bool wide = std::min(g.M, g.N) >= 1024 && (long)g.M * g.N >= 4096L * 1024L;
  p.tile = g.tile ? g.tile : ((small >= 8192 || wide) ? 1 : 2);
  // (128x64 tiles, tile = 4, are +4-6 % on the K >= 1024 frame projections in isolation but cost 2 % of the step under
  // the three-lane schedule: 2.199 vs 2.157 ms -- not selected automatically)
  {   // launch-bound problems: at most 768 tiles of 32x32, K <= 1024, no operand-side fusions, no explicit split
    const long t32 = (long)((g.M + 31) / 32) * ((g.N + 31) / 32) * g.groups;
    const bool plain = !g.a_drop.enabled && !g.b_drop.enabled && g.a_row_mod == 0 && g.b_row_mod == 0;
    bool masked = false;
    for (int i = 0; i < g.groups; ++i) masked |= g.c_mask_y[i] != nullptr;   // only the small kernel implements it
    if (!g.bf16 && g.batch <= 1 && ((g.tile == 0 && g.splitk <= 1 && t32 <= 768 && g.K <= 1024 && plain) || g.tile == 3 || masked)) {
      p.tile = 3;
      p.nsplit = 1;
      p.waves = small_waves(g.K);                               // waves per workgroup that split K
      p.kchunk = ((((g.K + p.waves - 1) / p.waves) + 7) / 8) * 8;   // k per wave
      return p;
    }
  }
  int s;
  if (g.batch > 1) {                                     // strided-batched: the batch dimension fills the chip
    s = 1;
  } else if (g.splitk >= 1) {                            // explicit
    s = std::min(g.splitk, ktiles);
  } else {                                               // auto
    const long tiles = p.tile == 1 ? big : p.tile == 4 ? (long)((g.M + 127) / 128) * ((g.N + 63) / 64) * g.groups : small;
    const long want = p.tile == 1 ? 512 : 1024;
    const long smax = std::max(1, ktiles / 6);
    s = (int)std::min<long>(smax, want / tiles);   // floor: 900 tiles stay unsplit, 128 tiles split 8 ways
    if (s < 1) s = 1;
    while (s > 1 && plan_ws_bytes(g, s) > ws_bytes) --s;
  }
  p.nsplit = std::max(1, s);
  p.kchunk = g.K;
  if (p.nsplit > 1) {
    p.kchunk = ((ktiles + p.nsplit - 1) / p.nsplit) * BK;
    p.nsplit = (ktiles * BK + p.kchunk - 1) / p.kchunk;   // drop empty trailing splits
  }
  return p;
}

template <int BM, int BN>
int launch(const sdumc_gemm& g, int nsplit, int kchunk, hipStream_t st) {
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.groups * (g.batch > 1 ? g.batch : 1) * nsplit);
  if (g.bf16) {
    switch (g.layout) {
      case SDUMC_NT: hipLaunchKernelGGL((gemm_kernel<BM, BN, true, true, true>), grid, dim3(256), 0, st, g, nsplit, kchunk); break;
      case SDUMC_NN: hipLaunchKernelGGL((gemm_kernel<BM, BN, true, false, true>), grid, dim3(256), 0, st, g, nsplit, kchunk); break;
      case SDUMC_TN: hipLaunchKernelGGL((gemm_kernel<BM, BN, false, false, true>), grid, dim3(256), 0, st, g, nsplit, kchunk); break;
      default: return SDUMC_EINVAL;
    }
    return SDUMC_OK;
  }
  switch (g.layout) {
    case SDUMC_NT: hipLaunchKernelGGL((gemm_kernel<BM, BN, true, true>), grid, dim3(256), 0, st, g, nsplit, kchunk); break;
    case SDUMC_NN: hipLaunchKernelGGL((gemm_kernel<BM, BN, true, false>), grid, dim3(256), 0, st, g, nsplit, kchunk); break;
    case SDUMC_TN: hipLaunchKernelGGL((gemm_kernel<BM, BN, false, false>), grid, dim3(256), 0, st, g, nsplit, kchunk); break;
    default: return SDUMC_EINVAL;
  }
  return SDUMC_OK;
}

}  // namespace

// per-launch timing hooks for the other GEMM files (gemm_bf16.hip): begin returns a token (or -1 when profiling is off)
extern "C" int sdumc_prof_begin_(int variant, double flops, void* stream) {
  if (!g_prof_on) return -1;
  ProfRec r;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
  r.variant = variant;
  r.flops = flops;
  (void)hipEventRecord(r.a, as_stream(stream));
  std::lock_guard<std::mutex> lock(g_prof_mu);
  g_prof.push_back(r);
  return (int)g_prof.size() - 1;
}
extern "C" void sdumc_prof_end_(int token, void* stream) {
  if (token < 0) return;
  std::lock_guard<std::mutex> lock(g_prof_mu);
  if (token < (int)g_prof.size()) (void)hipEventRecord(g_prof[token].b, as_stream(stream));
}

extern "C" int sdumc_gemm_wide_(const sdumc_gemm* gp, int cfg, int nsplit, int kchunk, void* stream);   // gemm_wide.hip

extern "C" size_t sdumc_gemm_workspace_bytes(const sdumc_gemm* g) {
  if (!g || g->M <= 0 || g->N <= 0 || g->K <= 0 || g->groups < 1) return 0;
  return plan_ws_bytes(*g, plan_gemm(*g, (size_t)-1).nsplit);
}

// n = 2 or 3 single-group TN problems that sdumc_gemm_f32 would each hand to the small-problem kernel: one launch for all of them.
// SDUMC_EINVAL = not such a set (the caller launches them one by one).
extern "C" int sdumc_gemm_small_tn_multi_(const sdumc_gemm* gs, int n, void* stream) {
  if (!gs || n < 2 || n > 3) return SDUMC_EINVAL;
  SmallMulti m;
  memset(&m, 0, sizeof(m));
  int blocks = 0;
  double flops = 0.0;
  for (int i = 0; i < 3; ++i) {
    if (i >= n) { m.blk_end[i] = blocks; m.tiles_n[i] = 1; continue; }
    const sdumc_gemm& g = gs[i];
    if (g.layout != SDUMC_TN || g.groups != 1 || g.batch > 1 || g.bf16 || g.M <= 0 || g.N <= 0 || g.K <= 0 || !g.A[0] || !g.B[0] || !g.C[0]) return SDUMC_EINVAL;
    if (g.accumulate && (g.act != SDUMC_ACT_NONE || g.c_drop.enabled)) return SDUMC_EINVAL;
    if (g.c_drop.enabled && (g.c_drop.width & 3)) return SDUMC_EINVAL;
    const GemmPlan pl = plan_gemm(g, 0);
    if (pl.tile != 3 || pl.waves != 4) return SDUMC_EINVAL;
    m.g[i] = g;
    m.kq[i] = pl.kchunk;
    m.tiles_n[i] = (g.N + 31) / 32;
    blocks += m.tiles_n[i] * ((g.M + 31) / 32);
    m.blk_end[i] = blocks;
    flops += 2.0 * g.M * (double)g.N * g.K;
  }
  const int tok = sdumc_prof_begin_(8, flops, stream);      // (gemm_small_tn)
  hipLaunchKernelGGL(gemm_small_tn_multi_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), m);
  SDUMC_CHECK_LAUNCH();
  sdumc_prof_end_(tok, stream);
  return SDUMC_OK;
}

extern "C" int sdumc_gemm_f32(const sdumc_gemm* gp, void* stream) {
  if (!gp) return SDUMC_EINVAL;
  const sdumc_gemm& g = *gp;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.groups < 1 || g.groups > SDUMC_MAX_GROUPS) return SDUMC_EINVAL;
  if (g.layout < 0 || g.layout > 2) return SDUMC_EINVAL;
  for (int i = 0; i < g.groups; ++i)
    if (!g.A[i] || !g.B[i] || !g.C[i]) return SDUMC_EINVAL;
  if (g.accumulate && (g.act != SDUMC_ACT_NONE || g.c_drop.enabled)) return SDUMC_EINVAL;
  if ((g.a_drop.enabled && (g.a_drop.width & 3)) || (g.b_drop.enabled && (g.b_drop.width & 3)) ||
      (g.c_drop.enabled && (g.c_drop.width & 3)))
    return SDUMC_EINVAL;
  for (int i = 0; i < g.groups; ++i)
    if (g.colsum_a[i] && g.layout != SDUMC_TN) return SDUMC_EINVAL;
  if (g.batch > 1) {   // strided-batched mode: plain products only (the attention-core GEMMs of transformer.hip)
    if (g.tile == 3 || g.splitk > 1 || g.a_drop.enabled || g.b_drop.enabled || g.c_drop.enabled || g.a_row_mod ||
        g.b_row_mod || (long)g.groups * g.batch > 65535)
      return SDUMC_EINVAL;
    for (int i = 0; i < g.groups; ++i)
      if (g.colsum_a[i] || g.c_mask_y[i] || g.bias[i]) return SDUMC_EINVAL;
  }
  if (g.bf16) {   // bf16 operands: 16-byte aligned rows, channel extents multiples of 4, 64x64 / 128x128 tiles
    if ((g.lda & 3) || (g.ldb & 3) || (g.K & 3) || (g.layout != SDUMC_NT && ((g.N & 3) || (g.layout == SDUMC_TN && (g.M & 3)))) ||
        g.tile == 3 || g.tile == 4)
      return SDUMC_EINVAL;
    for (int i = 0; i < g.groups; ++i)
      if ((reinterpret_cast<uintptr_t>(g.A[i]) | reinterpret_cast<uintptr_t>(g.B[i])) & 15) return SDUMC_EINVAL;
  }
  GemmPlan pl = plan_gemm(g, g.workspace ? g.workspace_bytes : 0);
  hipStream_t st = as_stream(stream);
  if (pl.tile >= 11) {
    if (pl.nsplit > 1 && (!g.workspace || g.workspace_bytes < plan_ws_bytes(g, pl.nsplit))) return SDUMC_ENOMEM;
    ProfRec wrec;
    const bool wprof = g_prof_on;
    if (wprof) {
      if (hipEventCreate(&wrec.a) != hipSuccess || hipEventCreate(&wrec.b) != hipSuccess) return SDUMC_ELAUNCH;
      wrec.variant = g.layout == SDUMC_TN ? 16 : (sdumc_split_on_(SDUMC_SPLIT_WIDE) ? 23 : 15);
      wrec.flops = 2.0 * g.M * (double)g.N * g.K * g.groups;
      (void)hipEventRecord(wrec.a, st);
    }
    const int wrc = sdumc_gemm_wide_(&g, pl.tile - 10, pl.nsplit, pl.kchunk, stream);
    if (wrc < 0) return wrc;
    if (wrc == SDUMC_OK) {
      if (wprof) {
        (void)hipEventRecord(wrec.b, st);
        std::lock_guard<std::mutex> lock(g_prof_mu);
        g_prof.push_back(wrec);
      }
      if (pl.nsplit > 1) {
        const size_t mn = (size_t)g.M * g.N + (size_t)g.M;
        dim3 grid((unsigned)((mn + 255) / 256), g.groups);
        hipLaunchKernelGGL(splitk_reduce_kernel, grid, dim3(256), 0, st, g, pl.nsplit);
        SDUMC_CHECK_LAUNCH();
      }
      return SDUMC_OK;
    }
    if (wprof) { (void)hipEventDestroy(wrec.a); (void)hipEventDestroy(wrec.b); }
    sdumc_gemm g2 = g;           // not a problem the wide kernels take: the generic kernels
    g2.tile = 2;
    g2.splitk = 0;
    return sdumc_gemm_f32(&g2, stream);
  }
  const int nsplit = pl.nsplit, kchunk = pl.kchunk, tile = pl.tile;
  const int small_nw = pl.waves;
  if (nsplit > 1 && (!g.workspace || g.workspace_bytes < plan_ws_bytes(g, nsplit))) return SDUMC_ENOMEM;
  ProfRec rec;
  const bool prof = g_prof_on;
  if (prof) {
    if (hipEventCreate(&rec.a) != hipSuccess || hipEventCreate(&rec.b) != hipSuccess) return SDUMC_ELAUNCH;
    rec.variant = g.bf16 ? 9 + g.layout * 2 + (tile == 1 ? 0 : 1) : tile == 3 ? 6 + g.layout : g.layout * 2 + (tile == 1 ? 0 : 1);
    rec.flops = 2.0 * g.M * (double)g.N * g.K * g.groups * (g.batch > 1 ? g.batch : 1);
    (void)hipEventRecord(rec.a, st);
  }
  int rc = SDUMC_OK;
  if (tile == 3) {
    if (g.tile == 3 && (g.a_drop.enabled || g.b_drop.enabled || g.a_row_mod || g.b_row_mod)) return SDUMC_EINVAL;
    dim3 grid((g.N + 31) / 32, (g.M + 31) / 32, g.groups);
    if (small_nw == 8) {
      switch (g.layout) {
        case SDUMC_NT: hipLaunchKernelGGL((gemm_small_kernel<true, true, 8>), grid, dim3(512), 0, st, g, kchunk); break;
        case SDUMC_NN: hipLaunchKernelGGL((gemm_small_kernel<true, false, 8>), grid, dim3(512), 0, st, g, kchunk); break;
        default: hipLaunchKernelGGL((gemm_small_kernel<false, false, 8>), grid, dim3(512), 0, st, g, kchunk); break;
      }
    } else {
      switch (g.layout) {
        case SDUMC_NT: hipLaunchKernelGGL((gemm_small_kernel<true, true, 4>), grid, dim3(256), 0, st, g, kchunk); break;
        case SDUMC_NN: hipLaunchKernelGGL((gemm_small_kernel<true, false, 4>), grid, dim3(256), 0, st, g, kchunk); break;
        default: hipLaunchKernelGGL((gemm_small_kernel<false, false, 4>), grid, dim3(256), 0, st, g, kchunk); break;
      }
    }
  } else {
    rc = tile == 1 ? launch<128, 128>(g, nsplit, kchunk, st)
         : tile == 4 ? launch<128, 64>(g, nsplit, kchunk, st) : launch<64, 64>(g, nsplit, kchunk, st);
  }
  if (rc != SDUMC_OK) return rc;
  SDUMC_CHECK_LAUNCH();
  if (prof) {
    (void)hipEventRecord(rec.b, st);
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof.push_back(rec);
  }
  if (nsplit > 1) {
    const size_t mn = (size_t)g.M * g.N + (size_t)g.M;   // + M tail threads for the fused column sums
    dim3 grid((unsigned)((mn + 255) / 256), g.groups);
    hipLaunchKernelGGL(splitk_reduce_kernel, grid, dim3(256), 0, st, g, nsplit);
    SDUMC_CHECK_LAUNCH();
  }
  return SDUMC_OK;
}

// ---- profiling hooks ------------------------------------------------------------------------
extern "C" int sdumc_profile_enable(int on) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  for (ProfRec& r : g_prof) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof.clear();
  g_prof_on = on != 0;
  return SDUMC_OK;
}

extern "C" int sdumc_profile_report(sdumc_prof_entry* out, int max_entries) {
  if (!out || max_entries < kNumVariants) return SDUMC_EINVAL;
  std::lock_guard<std::mutex> lock(g_prof_mu);
  for (int v = 0; v < kNumVariants; ++v) {
    out[v].name = kVariantName[v];
    out[v].launches = 0;
    out[v].total_ms = 0.0;
    out[v].total_flops = 0.0;
  }
  for (ProfRec& r : g_prof) {
    if (hipEventSynchronize(r.b) != hipSuccess) return SDUMC_ELAUNCH;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return SDUMC_ELAUNCH;
    out[r.variant].launches += 1;
    out[r.variant].total_ms += ms;
    out[r.variant].total_flops += r.flops;
  }
  return kNumVariants;
}

// One empty kernel per source file = per gfx950 code object: sdumc_preload_() asks for its attributes, which makes the HIP runtime load
// this file's code object NOW (outside any timed or latency-sensitive region) instead of at the first launch of one of its kernels --
// with deferred loading that first launch stalls the host for tens of milliseconds (seen as a 36-59 ms gap in the middle of an epoch,
// at the first batch whose shape took a fallback path: profiles/README.md, round 6).
__global__ void sdumc_preload_gemm_f32_kernel() {}
extern "C" int sdumc_preload_gemm_f32_(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&sdumc_preload_gemm_f32_kernel)) == hipSuccess ? SDUMC_OK : SDUMC_ELAUNCH;
}
