// gemm_p3.hip — fp32 GEMMs on operands that were split ONCE PER TENSOR into three bf16 planes ("P3" layout), gfx950.
//
// Round 4 moved the fp32 products onto the bf16 matrix pipe: an fp32 value is the exact sum of three bf16 values, the six largest
// of the nine part products are accumulated in fp32 (gemm_group.hip has the arithmetic and its error bound).  Those kernels split
// their operands per workgroup per k-tile: a 256 x 1024 weight was re-split by every m-tile of every launch, the features by
// every consumer -- ~10 VALU instructions per MFMA, and the VALU, not the matrix pipe, set the rate (0.25 of the six-term
// ceiling).  Here the split happens once, where a tensor is produced (sdumc_p3_split for the features when a batch is installed
// and for the weights at the head of a forward; the GEMM epilogue below for the projected frames), and the inner loop is a plain
// LDS-DMA bf16 loop with six v_mfma_f32_32x32x16_bf16 per loaded (A, B) fragment pair.
//
// P3 layout of a row-major matrix X[rows][K] (K a multiple of 8): row r is 6 K bytes at r * ld (bytes), made of K / 8 chunks of
// 48 bytes; chunk c holds k = 8 c .. 8 c + 7 as [plane 0: 8 bf16][plane 1: 8 bf16][plane 2: 8 bf16], plane 0 = bf16(x) (round to
// nearest even), plane 1 = bf16(x - p0), plane 2 = bf16(x - p0 - p1): x == p0 + p1 + p2 exactly (sdumc_hip.h states the edges).
// The three planes of a k-tile of a row are ONE contiguous run (96 bytes per 16 k), so the operand streams of the NT product
// below -- and of a TN product that contracts over the rows -- are sequential in HBM.
//
// What runs here (C2 shapes): NT  C[M, 256] = act(mask(A)[M, K] . B[256, K]^T * scale + bias)
//   * the frame projections frame_dim_reshape_{0,1,2} (model :193-195, :282-284): A = feature planes, B = weight planes,
//     K = 1024 / 4096, M = B T up to 24 000 (the text slot: K split over workgroups, fp32 slabs, ordered reduce);
//   * the key projections input_proj of FRA2UTT_new / Cross_Attention (model :60, :82): A = the projected frames' planes with the
//     keep-bits of the fused input dropout applied to all three planes, bias + tanh in the epilogue.
// Tile: BM x 256 (all of N: A crosses HBM once) per 512-thread workgroup, wave w owns columns [32 w, 32 w + 32) of every row --
// TM = BM / 32 accumulator tiles, 3 (TM + 1) ds_read_b128 per 6 TM MFMAs, far below what the LDS array delivers.  Operands go
// global -> LDS by LDS-DMA through a ring of NST k-tiles of 16 k (rows of 96 bytes; the two 48-byte k-halves of a row swapped by
// bit 3 of the row: conflict-free b128 reads for every lane group), ONE raw barrier per k-tile, counted vmcnt.  BM is chosen per
// shape so that the tiles fill 256 CUs in whole rounds (24 000 rows: 250 tiles of 96).
#include <algorithm>
#include <cstdlib>

#include "p3_loop.h"

namespace sdumc_p3 {

struct Args {
  sdumc_gemm_p3 g;
  int nsplit, kchunk;
};

// MAPPED: the rows of A are named by a row map (sdumc_gemm_p3.a_map / a2_map: A is a resident store's packed tensor, the batch is read
// in place).  1: the packed tensor is below 4 GiB (a_map_rows says so): the map entry takes the row index's place in the descriptor
// offset; 2: any size, rows fetched by 64-bit address (global_load_lds; measured 8-13 % slower on the frame projections than the
// descriptor form: tools/map_bench.py)
template <class CF, int MAPPED = 0>
__global__ __launch_bounds__(CF::NTHR, CF::NW == 4 ? 2 : 1) void gemm_p3_nt_kernel(const Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = CF::BM, TM = CF::TM, TN = CF::TN;
  constexpr bool MASK = CF::MASK;
  const sdumc_gemm_p3& g = a.g;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_n = g.N / BN;
  const int tlin = blockIdx.x;
  const int tile_m = tlin / tiles_n, tile_n = tlin - tile_m * tiles_n;
  const int ks = blockIdx.y;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = ks * a.kchunk, kend = min(g.K, kbeg + a.kchunk);
  const int nk = (kend - kbeg) / BK;                      // a multiple of 4 (the launcher checks K and the split)

#if defined(SDUMC_P3_DBG) && (SDUMC_P3_DBG & 8)
  const uint64_t dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // rows [a2_row0, M) of A live in a second tensor (the two streams' text features): a tile lies on one side (a2_row0 % BM == 0)
  const bool second = g.A2 != nullptr && m0 >= g.a2_row0;
  const int arow0 = second ? g.a2_row0 : 0;                                  // tile rows are m - arow0 inside the chosen tensor
  const int a_rows = g.a_row_mod > 0 ? g.a_row_mod : (g.A2 ? (second ? g.M - g.a2_row0 : g.a2_row0) : g.M);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(second ? g.A2 : g.A), 0, (int)min((size_t)a_rows * (size_t)g.lda, (size_t)0xFFFFFFF0u), 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rbits = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(MASK ? g.a_bits : static_cast<const uint8_t*>(g.A)), 0,
                                                                        MASK ? (int)min((size_t)g.M * (size_t)g.bits_qw, (size_t)0xFFFFFFF0u) : 0, 0x00020000);

  f32x16 acc[TM][TN];
  uint64_t dbg_r1 = 0;
  const char* bw = static_cast<const char*>(g.B) + (size_t)(tile_n * 8 + wave * TN) * (size_t)g.ldb;
  if constexpr (MAPPED == 2) {
    const char* abase = static_cast<const char*>(second ? g.A2 : g.A);
    const int32_t* amap = second ? g.a2_map : g.a_map;
    p3_mainloop<CF>(lds, ra, g.lda,
                    [&](int row) -> const char* { return abase + (size_t)amap[min(m0 + row, g.M - 1) - arow0] * (size_t)g.lda; },
                    rbits, g.bits_qw, [&](int row) { return min(m0 + row, g.M - 1); }, bw, (size_t)g.ldb, kbeg, nk, acc, &dbg_r1);
  } else if constexpr (MAPPED == 1) {
    const int32_t* amap = second ? g.a2_map : g.a_map;
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(second ? g.A2 : g.A), 0, (int)min((size_t)g.a_map_rows * (size_t)g.lda, (size_t)0xFFFFFFF0u), 0x00020000);
    p3_mainloop<CF>(lds, rm, g.lda, [&](int row) { return amap[min(m0 + row, g.M - 1) - arow0]; },
                    rbits, g.bits_qw, [&](int row) { return min(m0 + row, g.M - 1); }, bw, (size_t)g.ldb, kbeg, nk, acc, &dbg_r1);
  } else {
    p3_mainloop<CF>(lds, ra, g.lda,
                    [&](int row) { int r = min(m0 + row, g.M - 1) - arow0; if (g.a_row_mod > 0) r %= g.a_row_mod; return r; },
                    rbits, g.bits_qw, [&](int row) { return min(m0 + row, g.M - 1); }, bw, (size_t)g.ldb, kbeg, nk, acc, &dbg_r1);
  }
  (void)dbg_r1;
#if defined(SDUMC_P3_DBG) && (SDUMC_P3_DBG & 8)
  const uint64_t dbg_r2 = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).  The tile turns
  //      through LDS (the ring is free) so that a lane owns 8 consecutive columns of a row: 32-byte fp32 stores, and the P3 copy
  //      of the same values as one 48-byte chunk ----
  const bool to_slab = a.nsplit > 1;
  const float mscale = MASK ? g.a_scale : 1.f;
  constexpr int LDT = CF::LDT, WCOLS = 32 * TN, CPW = WCOLS / 8;      // a wave's columns, its 8-column chunks per row
  float* tw = reinterpret_cast<float*>(lds) + wave * 32 * LDT;
  const int colw = n0 + WCOLS * wave;
  float bv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) bv[c] = (!to_slab && g.bias) ? g.bias[colw + 8 * (lane & (CPW - 1)) + c] : 0.f;
  __syncthreads();                                          // every wave is done reading the last k-tile
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) tw[((e & 3) + 8 * (e >> 2) + 4 * lh) * LDT + 32 * j + li] = acc[i][j][e] * mscale;
    __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): this wave's own LDS writes (no other wave reads them)
#pragma unroll
    for (int u = lane; u < 32 * CPW; u += 64) {
      const int r = u / CPW, cq = u % CPW;
      const int row = m0 + 32 * i + r, col = colw + 8 * cq;
      if (row < g.M) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(tw + r * LDT + 8 * cq), a1 = *reinterpret_cast<const f32x4*>(tw + r * LDT + 8 * cq + 4);
        float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        if (to_slab) {
          float* dst = g.workspace + ((size_t)ks * g.M + row) * g.N + col;
          *reinterpret_cast<f32x4*>(dst) = a0;
          *reinterpret_cast<f32x4*>(dst + 4) = a1;
        } else {
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            float x = v[c] + bv[c];
            if (g.act == SDUMC_ACT_TANH) x = fast_tanh(x);
            else if (g.act == SDUMC_ACT_RELU) x = fmaxf(x, 0.f);
            v[c] = x;
          }
          if (g.C) {
            float* dst = g.C + (size_t)row * g.ldc + col;
            *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
          }
          if (g.C_p3) {
            u32x4 pl[3];
            split8(v, pl);
            char* dst = static_cast<char*>(g.C_p3) + (size_t)row * (size_t)g.ldc_p3 + (size_t)(col >> 3) * 48;
            *reinterpret_cast<u32x4*>(dst) = pl[0];
            *reinterpret_cast<u32x4*>(dst + 16) = pl[1];
            *reinterpret_cast<u32x4*>(dst + 32) = pl[2];
          }
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                     // reads done before the next 32 rows overwrite the staging
  }
#if defined(SDUMC_P3_DBG) && (SDUMC_P3_DBG & 8)
  __syncthreads();
  if (tid == 0 && g.C) {      // shader-clock and 100 MHz stamps of this workgroup, over its tile's first four outputs
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t* o = reinterpret_cast<uint32_t*>(g.C + (size_t)m0 * g.ldc + n0);
    o[0] = (uint32_t)(t1 - dbg_t0);
    o[1] = (uint32_t)(r1 - dbg_r0);
    o[2] = (uint32_t)(dbg_r1 - dbg_r0);      // prologue: start -> stage 0 landed
    o[3] = (uint32_t)(r1 - dbg_r2);          // epilogue
  }
#endif
#endif
}

// ordered reduction of the split-K slabs + the epilogue the tiles skipped (bias, activation, fp32 and / or P3 output)
__global__ __launch_bounds__(256) void p3_splitk_reduce_kernel(const sdumc_gemm_p3 g, const int nsplit) {
  const size_t u = (size_t)blockIdx.x * 256 + threadIdx.x;       // one 8-column chunk of one row
  const int cpr = g.N >> 3;
  if (u >= (size_t)g.M * cpr) return;
  const int row = (int)(u / cpr), col = (int)(u - (size_t)row * cpr) * 8;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* s = g.workspace + (size_t)row * g.N + col;
  const size_t slab = (size_t)g.M * g.N;
  int z = 0;
  for (; z + 4 <= nsplit; z += 4) {      // four slabs in flight; summed in ascending order
    f32x4 a[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q][0] = *reinterpret_cast<const f32x4*>(s + (size_t)(z + q) * slab);
      a[q][1] = *reinterpret_cast<const f32x4*>(s + (size_t)(z + q) * slab + 4);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) { v[c] += a[q][0][c]; v[4 + c] += a[q][1][c]; }
  }
  for (; z < nsplit; ++z) {
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(s + (size_t)z * slab), a1 = *reinterpret_cast<const f32x4*>(s + (size_t)z * slab + 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) { v[c] += a0[c]; v[4 + c] += a1[c]; }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float x = v[c] + (g.bias ? g.bias[col + c] : 0.f);
    if (g.act == SDUMC_ACT_TANH) x = sdumc_p3::fast_tanh(x);
    else if (g.act == SDUMC_ACT_RELU) x = fmaxf(x, 0.f);
    v[c] = x;
  }
  if (g.C) {
    float* dst = g.C + (size_t)row * g.ldc + col;
    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
  if (g.C_p3) {
    u32x4 pl[3];
    split8(v, pl);
    char* dst = static_cast<char*>(g.C_p3) + (size_t)row * (size_t)g.ldc_p3 + (size_t)(col >> 3) * 48;
    *reinterpret_cast<u32x4*>(dst) = pl[0];
    *reinterpret_cast<u32x4*>(dst + 16) = pl[1];
    *reinterpret_cast<u32x4*>(dst + 32) = pl[2];
  }
}

// fp32 [rows][cols] (row stride ld floats) -> P3 (row stride ldp bytes); one thread per 8-column chunk
__global__ __launch_bounds__(256) void p3_split_kernel(const float* __restrict__ src, int64_t ld, char* __restrict__ dst, int64_t ldp, int64_t rows,
                                                       int cols) {
  const int cpr = cols >> 3;
  const int64_t total = rows * cpr;
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
    const int64_t row = u / cpr;
    const int c = (int)(u - row * cpr);
    const float* s = src + row * ld + 8 * c;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(s), a1 = *reinterpret_cast<const f32x4*>(s + 4);
    const float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    u32x4 pl[3];
    split8(v, pl);
    char* d = dst + row * ldp + (int64_t)c * 48;
    *reinterpret_cast<u32x4*>(d) = pl[0];
    *reinterpret_cast<u32x4*>(d + 16) = pl[1];
    *reinterpret_cast<u32x4*>(d + 32) = pl[2];
  }
}
// fp32 weight [rows][cols] (rows % 32 == 0, cols % 16 == 0) -> fragment-major P3: [rows / 32][cols / 16][3 planes][64 lanes][16 bytes],
// lane (li = lane & 31, lh = lane >> 5) of block (rb, kt) holds W[32 rb + li][16 kt + 8 lh .. + 7] -- a wave's MFMA B operand of a
// k-tile is three contiguous KiB.  One thread per (row, 8-column chunk).
__global__ __launch_bounds__(256) void p3_split_frag_kernel(const float* __restrict__ src, int64_t ld, char* __restrict__ dst, int rows, int cols) {
  const int cpr = cols >> 3;
  const int64_t total = (int64_t)rows * cpr;
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
    const int row = (int)(u / cpr), c = (int)(u - (int64_t)row * cpr);
    const float* sp = src + (int64_t)row * ld + 8 * c;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(sp), a1 = *reinterpret_cast<const f32x4*>(sp + 4);
    const float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    u32x4 pl[3];
    split8(v, pl);
    const int rb = row >> 5, li = row & 31, kt = c >> 1, lh = c & 1;
    char* d = dst + ((int64_t)rb * (cols >> 4) + kt) * FRAG_KT + (lh * 32 + li) * 16;
    *reinterpret_cast<u32x4*>(d) = pl[0];
    *reinterpret_cast<u32x4*>(d + 1024) = pl[1];
    *reinterpret_cast<u32x4*>(d + 2048) = pl[2];
  }
}

// up to 12 weights of the flat parameter buffer in one launch (the engine's head-of-forward refresh): blockIdx.y = tensor
struct FragList {
  int64_t src_off[12], dst_off[12];      // floats into the parameter buffer / bytes into the destination
  int32_t rows[12], cols[12];
};
__global__ __launch_bounds__(256) void p3_split_frag_multi_kernel(const float* __restrict__ P, char* __restrict__ dst, const FragList L) {
  const int i = blockIdx.y;
  const int rows = L.rows[i], cols = L.cols[i], cpr = cols >> 3;
  const float* src = P + L.src_off[i];
  char* out = dst + L.dst_off[i];
  const int64_t total = (int64_t)rows * cpr;
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
    const int row = (int)(u / cpr), c = (int)(u - (int64_t)row * cpr);
    const float* sp = src + (int64_t)row * cols + 8 * c;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(sp), a1 = *reinterpret_cast<const f32x4*>(sp + 4);
    const float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    u32x4 pl[3];
    split8(v, pl);
    const int rb = row >> 5, li = row & 31, kt = c >> 1, lh = c & 1;
    char* d = out + ((int64_t)rb * (cols >> 4) + kt) * FRAG_KT + (lh * 32 + li) * 16;
    *reinterpret_cast<u32x4*>(d) = pl[0];
    *reinterpret_cast<u32x4*>(d + 1024) = pl[1];
    *reinterpret_cast<u32x4*>(d + 2048) = pl[2];
  }
}
// P3 -> fp32: (p0 + p1) + p2, exact (the parts' bits do not overlap)
__global__ __launch_bounds__(256) void p3_join_kernel(const char* __restrict__ src, int64_t ldp, float* __restrict__ dst, int64_t ld, int64_t rows,
                                                      int cols) {
  const int cpr = cols >> 3;
  const int64_t total = rows * cpr;
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
    const int64_t row = u / cpr;
    const int c = (int)(u - row * cpr);
    const char* s = src + row * ldp + (int64_t)c * 48;
    const u32x4 p0 = *reinterpret_cast<const u32x4*>(s), p1 = *reinterpret_cast<const u32x4*>(s + 16), p2 = *reinterpret_cast<const u32x4*>(s + 32);
    float v[8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      v[2 * d] = (__uint_as_float(p0[d] << 16) + __uint_as_float(p1[d] << 16)) + __uint_as_float(p2[d] << 16);
      v[2 * d + 1] = (__uint_as_float(p0[d] & 0xFFFF0000u) + __uint_as_float(p1[d] & 0xFFFF0000u)) + __uint_as_float(p2[d] & 0xFFFF0000u);
    }
    float* o = dst + row * ld + 8 * c;
    *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
}

struct Plan {
  int bm, nsplit, kchunk;
};
// BM: the tile height whose tile count fills 256 CUs in whole rounds with the least idle share; split K (fp32 slabs + an ordered
// reduce) only when even 64-row tiles leave most of the chip idle and K is long.  k-tiles come in groups of four (the unrolled loop).
inline Plan plan(const sdumc_gemm_p3& g, size_t have) {
  const int forced = g.tile_m ? g.tile_m : 64;
  const int tiles_n = g.N / BN;
  Plan best{64, 1, g.K};
  double best_cost = 1e30;
  const int cands[3] = {128, 96, 64};
  for (int ci = 0; ci < 3; ++ci) {
    const int bm = cands[ci];
    if (forced && forced != bm) continue;
    if (g.A2 && g.a2_row0 % bm) continue;                 // a tile must not straddle the two A tensors
    const long tiles = (long)((g.M + bm - 1) / bm) * tiles_n;
    const long rounds = (tiles + 255) / 256;
    // time ~ rounds * (rows per tile + a fixed prologue / epilogue share); smaller tiles pay more B traffic per row
    const double cost = (double)rounds * (bm + 24.0) * (1.0 + 8.0 / bm);
    if (cost < best_cost) { best_cost = cost; best.bm = bm; }
  }
  const int kq = g.K / (4 * BK);                       // groups of four k-tiles
  int s = 1;
  if (g.splitk >= 1) s = std::min(g.splitk, kq);
  else {
    const long tiles = (long)((g.M + best.bm - 1) / best.bm) * tiles_n;
    if (tiles <= 64 && kq >= 16) s = (int)std::min<long>(256 / std::max<long>(1, tiles), kq / 4);
    if (s < 1) s = 1;
  }
  while (s > 1 && (size_t)s * g.M * g.N * sizeof(float) > have) --s;
  best.kchunk = ((kq + s - 1) / s) * 4 * BK;
  best.nsplit = (g.K + best.kchunk - 1) / best.kchunk;
  return best;
}

template <class CF, int MAPPED = 0>
int launch(const sdumc_gemm_p3& g, const Plan& p, hipStream_t st) {
  static sdumc_dev_once attr_set;
  if (sdumc_once_per_device(attr_set, [] { return sdumc_set_dyn_lds(&gemm_p3_nt_kernel<CF, MAPPED>, CF::LDS_BYTES); }) != SDUMC_OK) return SDUMC_ELAUNCH;
  Args a{g, p.nsplit, p.kchunk};
  const dim3 grid((unsigned)(((g.M + CF::BM - 1) / CF::BM) * (g.N / BN)), (unsigned)p.nsplit);
  hipLaunchKernelGGL((gemm_p3_nt_kernel<CF, MAPPED>), grid, dim3(CF::NTHR), CF::LDS_BYTES, st, a);
  return SDUMC_OK;
}

}  // namespace sdumc_p3

extern "C" int sdumc_prof_begin_(int variant, double flops, void* stream);     // gemm_f32.hip: bench.py's per-launch HIP events
extern "C" void sdumc_prof_end_(int token, void* stream);

extern "C" size_t sdumc_gemm_p3_workspace_bytes(const sdumc_gemm_p3* g) {
  if (!g || g->M <= 0 || g->N <= 0 || g->K <= 0 || (g->K % (4 * sdumc_p3::BK))) return 0;
  const sdumc_p3::Plan p = sdumc_p3::plan(*g, (size_t)-1);
  return p.nsplit > 1 ? (size_t)p.nsplit * g->M * g->N * sizeof(float) : 0;
}

extern "C" int sdumc_gemm_p3_nt(const sdumc_gemm_p3* gp, void* stream) {
  using namespace sdumc_p3;
  if (!gp) return SDUMC_EINVAL;
  const sdumc_gemm_p3& g = *gp;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.N % BN) || (g.K % (4 * BK))) return SDUMC_EINVAL;
  if (!g.A || !g.B || (!g.C && !g.C_p3)) return SDUMC_EINVAL;
  if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.B) | reinterpret_cast<uintptr_t>(g.C) | reinterpret_cast<uintptr_t>(g.C_p3)) & 15) return SDUMC_EINVAL;
  if (g.lda < 6 * (int64_t)g.K || (g.lda & 15)) return SDUMC_EINVAL;
  if (g.ldb < (int64_t)(g.K / BK) * FRAG_KT || (g.ldb & 15)) return SDUMC_EINVAL;
  if (g.C && ((g.ldc & 3) || g.ldc < g.N)) return SDUMC_EINVAL;
  if (g.C_p3 && ((g.ldc_p3 & 15) || g.ldc_p3 < 6 * (int64_t)g.N)) return SDUMC_EINVAL;
  if (g.tile_m != 0 && g.tile_m != 64 && g.tile_m != 96 && g.tile_m != 128) return SDUMC_EINVAL;
  if (g.A2 && (g.a_row_mod || g.a2_row0 <= 0 || g.a2_row0 >= g.M || (g.a2_row0 % 64) || (g.tile_m && g.a2_row0 % g.tile_m) || (reinterpret_cast<uintptr_t>(g.A2) & 15))) return SDUMC_EINVAL;
  if (g.act != SDUMC_ACT_NONE && g.act != SDUMC_ACT_TANH && g.act != SDUMC_ACT_RELU) return SDUMC_EINVAL;
  const bool mapped = g.a_map != nullptr;
  if (mapped && (g.a_row_mod || g.a_bits || (g.A2 != nullptr) != (g.a2_map != nullptr))) return SDUMC_EINVAL;   // (the frame projections' form)
  if (!mapped && g.a2_map) return SDUMC_EINVAL;
  const size_t a_bytes = (size_t)(g.a_row_mod > 0 ? g.a_row_mod : (g.A2 ? std::max(g.a2_row0, g.M - g.a2_row0) : g.M)) * (size_t)g.lda;
  if (!mapped && a_bytes >= 0xFFFFFFF0u) return SDUMC_EINVAL;      // (mapped rows are fetched by 64-bit address: no such limit)
  const bool mask = g.a_bits != nullptr;
  if (mask && (g.bits_qw < g.K / 4 || (g.bits_qw & 3) || (reinterpret_cast<uintptr_t>(g.a_bits) & 3) || (size_t)g.M * g.bits_qw >= 0xFFFFFFF0u)) return SDUMC_EINVAL;
  const Plan p = plan(g, g.workspace ? g.workspace_bytes : 0);
  if (p.nsplit > 1 && (!g.workspace || (reinterpret_cast<uintptr_t>(g.workspace) & 15))) return SDUMC_ENOMEM;
  hipStream_t st = as_stream(stream);
  const int tok = sdumc_prof_begin_(mask ? 27 : 26, 2.0 * g.M * (double)g.N * g.K, stream);
  int rc;
  // tile_m = 0: 64-row tiles on 256-thread workgroups, two per CU -- slower alone than the tall 512-thread forms on a shape that fills
  // the chip in one round (audio frame projection: 76 against 62 us at 96 rows), faster inside the step, where three lanes' kernels
  // share the chip (fp32 C2 step 1.370 against 1.386-1.391 ms, two alternations) -- a 512-thread workgroup holds its CU alone.
  // An explicit tile_m: the 512-thread forms.
  // (a map over a packed tensor below 4 GiB keeps the descriptor form; else rows by 64-bit address)
  const bool map32 = mapped && g.a_map_rows > 0 && (size_t)g.a_map_rows * (size_t)g.lda < 0xFFFFFFF0u;
  if (mapped && g.tile_m) return SDUMC_EINVAL;      // (row maps: the step's form, 64-row tiles on 256-thread workgroups)
  if (map32) rc = launch<PCfg<64, 4, false, 4>, 1>(g, p, st);
  else if (mapped) rc = launch<PCfg<64, 4, false, 4>, 2>(g, p, st);
  else if (!g.tile_m) {
    rc = mask ? launch<PCfg<64, 4, true, 4>>(g, p, st) : launch<PCfg<64, 4, false, 4>>(g, p, st);
  } else if (mask) {
    rc = p.bm == 128 ? launch<PCfg<128, 4, true>>(g, p, st) : p.bm == 96 ? launch<PCfg<96, 4, true>>(g, p, st) : launch<PCfg<64, 4, true>>(g, p, st);
  } else {
    rc = p.bm == 128 ? launch<PCfg<128, 4, false>>(g, p, st) : p.bm == 96 ? launch<PCfg<96, 4, false>>(g, p, st) : launch<PCfg<64, 4, false>>(g, p, st);
  }
  if (rc != SDUMC_OK) return rc;
  SDUMC_CHECK_LAUNCH();
  sdumc_prof_end_(tok, stream);
  if (p.nsplit > 1) {
    const size_t units = (size_t)g.M * (g.N >> 3);
    hipLaunchKernelGGL(p3_splitk_reduce_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st, g, p.nsplit);
    SDUMC_CHECK_LAUNCH();
  }
  return SDUMC_OK;
}

extern "C" int sdumc_p3_split(const float* src, int64_t ld, void* dst, int64_t ld_bytes, int64_t rows, int32_t cols, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || (cols & 7) || (ld & 3) || ld < cols || (ld_bytes & 15) || ld_bytes < 6 * (int64_t)cols) return SDUMC_EINVAL;
  if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) return SDUMC_EINVAL;
  const int64_t units = rows * (cols >> 3);
  const unsigned blocks = (unsigned)std::min<int64_t>((units + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(sdumc_p3::p3_split_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src, ld, static_cast<char*>(dst), ld_bytes, rows, (int)cols);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}
extern "C" int sdumc_p3_split_frag(const float* src, int64_t ld, void* dst, int32_t rows, int32_t cols, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || (rows & 31) || (cols & 15) || (ld & 3) || ld < cols) return SDUMC_EINVAL;
  if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) return SDUMC_EINVAL;
  const int64_t units = (int64_t)rows * (cols >> 3);
  const unsigned blocks = (unsigned)std::min<int64_t>((units + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(sdumc_p3::p3_split_frag_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src, ld, static_cast<char*>(dst), (int)rows, (int)cols);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}
// n <= 12 weights [rows_i][cols_i] at P + src_off[i] (row stride cols_i) -> fragment-major P3 at dst + dst_off[i] (bytes)
extern "C" int sdumc_p3_split_frag_multi_(const float* P, void* dst, const int64_t* src_off, const int64_t* dst_off, const int32_t* rows,
                                          const int32_t* cols, int n, void* stream) {
  if (!P || !dst || n < 1 || n > 12) return SDUMC_EINVAL;
  sdumc_p3::FragList L;
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    if ((rows[i] & 31) || (cols[i] & 15) || (src_off[i] & 3) || (dst_off[i] & 15)) return SDUMC_EINVAL;
    L.src_off[i] = src_off[i]; L.dst_off[i] = dst_off[i]; L.rows[i] = rows[i]; L.cols[i] = cols[i];
    most = std::max<int64_t>(most, (int64_t)rows[i] * (cols[i] >> 3));
  }
  const unsigned bx = (unsigned)std::min<int64_t>((most + 255) / 256, 128);
  hipLaunchKernelGGL(sdumc_p3::p3_split_frag_multi_kernel, dim3(bx, n), dim3(256), 0, as_stream(stream), P, static_cast<char*>(dst), L);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}
extern "C" int sdumc_p3_join(const void* src, int64_t ld_bytes, float* dst, int64_t ld, int64_t rows, int32_t cols, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || (cols & 7) || (ld & 3) || ld < cols || (ld_bytes & 15) || ld_bytes < 6 * (int64_t)cols) return SDUMC_EINVAL;
  if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) return SDUMC_EINVAL;
  const int64_t units = rows * (cols >> 3);
  const unsigned blocks = (unsigned)std::min<int64_t>((units + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(sdumc_p3::p3_join_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), static_cast<const char*>(src), ld_bytes, dst, ld, rows, (int)cols);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

// One empty kernel per source file = per gfx950 code object: sdumc_preload_() asks for its attributes, which makes the HIP runtime load
// this file's code object NOW (outside any timed or latency-sensitive region) instead of at the first launch of one of its kernels --
// with deferred loading that first launch stalls the host for tens of milliseconds (seen as a 36-59 ms gap in the middle of an epoch,
// at the first batch whose shape took a fallback path: profiles/README.md, round 6).
__global__ void sdumc_preload_gemm_p3_kernel() {}
extern "C" int sdumc_preload_gemm_p3_(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&sdumc_preload_gemm_p3_kernel)) == hipSuccess ? SDUMC_OK : SDUMC_ELAUNCH;
}
