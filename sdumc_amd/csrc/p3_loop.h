// p3_loop.h -- the k-loop of the fp32 NT products on operands pre-split into three bf16 planes (gemm_p3.hip has the story; the
// fused UMCA forward kernel of attn_pool.hip runs the same loop in front of its pooling epilogue).  gfx950 only.
#pragma once
#include <type_traits>

#include "common.h"

namespace sdumc_p3 {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef const __attribute__((address_space(4))) int32_t const_i32_t;      // loads through it with a wave-uniform index are scalar loads
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int waitcnt_vm(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0xF << 8); }

__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }

__device__ __forceinline__ uint32_t pk(float x, float y) {       // v_cvt_pk_bf16_f32 (round to nearest even), low half = x
  const f32x2s v = {x, y};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
// eight fp32 values -> their three bf16 planes (4 dwords each)
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&pl)[3]) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const float x = v[2 * d], y = v[2 * d + 1];
    const uint32_t p0 = pk(x, y);
    const float x1 = x - __uint_as_float(p0 << 16), y1 = y - __uint_as_float(p0 & 0xFFFF0000u);       // exact
    const uint32_t p1 = pk(x1, y1);
    const float x2 = x1 - __uint_as_float(p1 << 16), y2 = y1 - __uint_as_float(p1 & 0xFFFF0000u);     // exact
    pl[0][d] = p0;
    pl[1][d] = p1;
    pl[2][d] = pk(x2, y2);
  }
}

constexpr int BN = 256, BK = 16, ROWB = 96;      // ROWB: bytes of one row of a k-tile (16 k x 3 planes x 2)
[[maybe_unused]] constexpr int NBS = 4, DB = 3;      // register sets of B fragments / how many k-tiles ahead B is requested
constexpr int FRAG_KT = 3 * 1024;   // bytes of one (32-row block, k-tile) of a fragment-major tensor: 3 planes x 64 lanes x 16 bytes

// NW_ = 8: wave w owns columns [32 w, 32 w + 32) (TN = 1), one workgroup per CU (tall tiles: least B traffic per row).
// NW_ = 4: wave w owns 64 columns (TN = 2); a 256-thread workgroup takes half a CU's registers, so TWO are resident per CU -- of one
//          launch or of two: the prologue / epilogue bursts of one overlap the k-loop of the other, and kernels of different lanes
//          share a CU the way the step's schedule expects (a 512-thread workgroup holds its CU alone).
template <int BM_, int NST_, bool MASK_, int NW_ = 8>
struct PCfg {
  static constexpr int BM = BM_, NST = NST_, TM = BM_ / 32, NW = NW_, TN = BN / (32 * NW_), NTHR = 64 * NW_;
  static constexpr bool MASK = MASK_;
  static constexpr int A_BYTES = BM * ROWB;
  static constexpr int A_P = A_BYTES / 1024;                                 // 1-KiB DMA pieces of A per stage: wave w takes w and w + 8
  static constexpr int NA = (A_P + NW - 1) / NW;
  static constexpr int BITS_BYTES = BM * 4;                                  // keep-bits of a k-tile: [BM][4 bytes] (low nibbles)
  static constexpr int BITS_P = (BITS_BYTES + 255) / 256;                    // 256-byte pieces (4 bytes per lane): wave w < BITS_P takes piece w
  static constexpr int BITS_LDS = MASK ? BITS_P * 256 : 0;
  static constexpr int STAGE = A_BYTES + BITS_LDS;
  static constexpr int LDT = 32 * TN + 4;                                    // floats per staged row of the epilogue (keeps b128 reads aligned)
  static constexpr int EPI_BYTES = NW * 32 * LDT * 4;                        // epilogue staging (overlays the ring)
  static constexpr int LDS_BYTES = NST * STAGE > EPI_BYTES ? NST * STAGE : EPI_BYTES;
  static_assert(BM % 32 == 0 && A_BYTES % 1024 == 0 && NA <= 2 && (NW == 4 || NW == 8), "tile shape");
  static_assert(BITS_P <= NW, "one keep-bits piece per wave");
  static_assert(NST >= DB + 1, "the A stage a step needs must be older in the queue than the B fragments it needs");
  static_assert(3 * (3 * TN + NA + 1) < 64, "vmcnt is a 6-bit counter");
};

// The loop (one 512-thread workgroup per BM x 256 tile; wave w owns columns [32 w, 32 w + 32) of every row):
//   * A (features / projected frames, P3 rows) is what the eight waves SHARE: it goes global -> LDS by LDS-DMA through a ring of NST
//     k-tiles (rows of 96 bytes, the two 48-byte k-halves swapped by bit 3 of the row: conflict-free ds_read_b128 for every lane
//     group), one raw barrier per k-tile, counted vmcnt;
//   * B (the weight, fragment-major) is PRIVATE to a wave -- nobody else multiplies its 32 columns -- so it never touches LDS: three
//     coalesced 1-KiB global loads per k-tile bring a wave's MFMA operands straight into registers, DB k-tiles ahead, through four
//     register sets.  With B in the LDS ring too (first form of this kernel) a k-tile moved 34 KB through LDS-DMA per CU and the
//     fill path -- ~70 GB/s per CU whatever the source -- set the time: 0.70 us per k-tile against 0.48 of MFMA work;
//   * A fragments are double-buffered in registers: while the six MFMA terms of k-tile t run, the fragments of k-tile t + 1 are read
//     from LDS into the other set -- the LDS latency sits in the gaps between MFMAs instead of in front of them.
// A wave's vector-memory queue is in issue order: step s issues [A pieces of stage s + NST, B fragments of k-tile s + DB].  Step t
// needs stage t + 1 in LDS and B(t) in registers; both belong to step t - DB's group or older (NST >= DB + 1), so the wait is
// "at most DB - 1 groups outstanding".
// p3_mainloop: acc[TM][TN] += A_tile[BM, 16 nk] . B_wave[32 TN, 16 nk]^T for this wave (wave w = columns [32 TN w, 32 TN (w + 1)) of the
// tile).  ra: buffer descriptor of the A tensor (P3 rows, row stride lda bytes); src_row(r): source row of tile row r < BM (clamped
// by the caller) -- or, when it returns a POINTER, the address of that row's first byte: the rows of a tile then lie anywhere in memory
// (a batch read in place from a resident feature store through a row map: tens of GB, beyond a descriptor's 32-bit offsets) and A is
// fetched with global_load_lds from per-lane 64-bit addresses instead of buffer_load ... lds; rbits / bits_qw / bits_row(r): the keep-bits (MASK); bwave: this wave's first 32-row block of the fragment-major
// weight (bblk = bytes between two blocks); k-tiles [kbeg / 16, kbeg / 16 + nk), nk a multiple of 4.  `lds`: CF::LDS_BYTES of ring.
// On return every wave has issued its last LDS reads (a workgroup barrier is still needed before the ring's memory is reused).
template <class CF, class SrcRow, class BitsRow>
__device__ __forceinline__ void p3_mainloop(char* lds, const __amdgpu_buffer_rsrc_t ra, const int64_t lda, SrcRow src_row,
                                            const __amdgpu_buffer_rsrc_t rbits, const int bits_qw, BitsRow bits_row, const char* bwave,
                                            const size_t bblk, const int kbeg, const int nk, f32x16 (&acc)[CF::TM][CF::TN],
                                            uint64_t* stamp = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NST = CF::NST, TM = CF::TM, TN = CF::TN, NW = CF::NW;
  constexpr bool MASK = CF::MASK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  // ---- this wave's DMA pieces of a stage of A: pieces wave and wave + 8 (where they exist), its keep-bits piece ----
  constexpr bool GA = std::is_pointer<decltype(src_row(0))>::value;      // A rows by address (global_load_lds) instead of by row index
  [[maybe_unused]] uint32_t voff[CF::NA];
  [[maybe_unused]] const char* gaddr[CF::NA];
  bool pvalid[CF::NA];
#pragma unroll
  for (int i = 0; i < CF::NA; ++i) {
    const int piece = wave + i * NW;
    pvalid[i] = piece < CF::A_P;
    const int q = (min(piece, CF::A_P - 1) << 6) + lane;                 // 16-byte slot inside the tile: [row][k-half'][plane]
    const int row = q / 6, sl = q - 6 * row;
    const int hp = sl >= 3 ? 1 : 0, p = sl - 3 * hp;
    const int h = hp ^ ((row >> 3) & 1);                                 // the k-half this slot holds (swizzle by bit 3 of the row)
    if constexpr (GA) gaddr[i] = reinterpret_cast<const char*>(src_row(row)) + (size_t)((kbeg >> 3) + h) * 48u + (size_t)p * 16u;
    else voff[i] = (uint32_t)src_row(row) * (uint32_t)lda + (uint32_t)((kbeg >> 3) + h) * 48u + (uint32_t)p * 16u;
  }
  const bool has_bits = MASK && wave < CF::BITS_P;
  uint32_t bvoff = 0;
  if constexpr (MASK) bvoff = (uint32_t)bits_row((wave << 6) + lane) * (uint32_t)bits_qw + (uint32_t)(kbeg >> 2);
  int per = 3 * TN;                                       // vector-memory operations of this wave per step (wave-uniform)
#pragma unroll
  for (int i = 0; i < CF::NA; ++i) per += pvalid[i] ? 1 : 0;
  per += has_bits ? 1 : 0;
  auto issue_a = [&](int buf) {
    char* base = lds + buf * CF::STAGE;
#pragma unroll
    for (int i = 0; i < CF::NA; ++i) {
#ifndef SDUMC_P3_DBG
#define SDUMC_P3_DBG 0      /* measurement builds only: bit 0 = no DMA of A, bit 1 = no loads of B, bit 2 = no MFMAs */
#endif
      if (pvalid[i] && !(SDUMC_P3_DBG & 1)) {
        if constexpr (GA) {
          __builtin_amdgcn_global_load_lds((gbl_void_t*)gaddr[i], (lds_void_t*)(base + (wave + i * NW) * 1024), 16, 0, 0);
          gaddr[i] += ROWB;
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(base + (wave + i * NW) * 1024), 16, voff[i], 0, 0, 0);
          voff[i] += ROWB;
        }
      }
    }
    if constexpr (MASK) {
      if (has_bits) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rbits, (lds_void_t*)(base + CF::A_BYTES + wave * 256), 4, bvoff, 0, 0, 0);
        bvoff += BK / 4;
      }
    }
  };
  // at most `groups` of this wave's per-step groups outstanding (per = 3 TN B fragments + 0 .. 2 pieces of A + keep-bits)
  auto wait_groups = [&](auto groups_c) {
    constexpr int G = decltype(groups_c)::value, B0 = 3 * TN;
    if (per == B0) __builtin_amdgcn_s_waitcnt(waitcnt_vm(G * B0));
    else if (per == B0 + 1) __builtin_amdgcn_s_waitcnt(waitcnt_vm(G * (B0 + 1)));
    else if (per == B0 + 2) __builtin_amdgcn_s_waitcnt(waitcnt_vm(G * (B0 + 2)));
    else __builtin_amdgcn_s_waitcnt(waitcnt_vm(G * (B0 + 3)));
  };

  // ---- B: fragment-major [N / 32][K / 16][3 planes][64 lanes][16 bytes]; this wave's block, its lane's 16 bytes ----
  const char* bptr = bwave + (size_t)(kbeg >> 4) * FRAG_KT + lane * 16;      // this wave's block, its lane's 16 bytes
  u32x4 pb[NBS][TN][3];
  auto load_b = [&](auto set_c) {
    constexpr int S = decltype(set_c)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        if (!(SDUMC_P3_DBG & 2)) pb[S][j][p] = *reinterpret_cast<const u32x4*>(bptr + j * bblk + p * 1024);
    bptr += FRAG_KT;
  };

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // a lane's MFMA operand of a 32-row block of A: 8 consecutive k (k-half lh) of row li, the three planes side by side (48 bytes)
  const int lane_off = li * ROWB + ((lh ^ ((li >> 3) & 1)) * 48);
  auto op = [](const u32x4& v) { return __builtin_bit_cast(bf16x8, v); };
  u32x4 pa[2][TM][3];
  uint32_t kb[2][TM];
  auto load_a = [&](const char* base, auto par) {
    constexpr int P = decltype(par)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) pa[P][i][2] = *reinterpret_cast<const u32x4*>(base + i * 32 * ROWB + lane_off + 32);
#pragma unroll
    for (int i = 0; i < TM; ++i) pa[P][i][0] = *reinterpret_cast<const u32x4*>(base + i * 32 * ROWB + lane_off);
#pragma unroll
    for (int i = 0; i < TM; ++i) pa[P][i][1] = *reinterpret_cast<const u32x4*>(base + i * 32 * ROWB + lane_off + 16);
    if constexpr (MASK) {      // keep-bits of (row, k-half): two bytes, low nibbles = elements 0..3 and 4..7
#pragma unroll
      for (int i = 0; i < TM; ++i) kb[P][i] = *reinterpret_cast<const uint16_t*>(base + CF::A_BYTES + (32 * i + li) * 4 + 2 * lh);
    }
  };
  auto mfmas = [&](auto par, auto set_c) {
    constexpr int P = decltype(par)::value, S = decltype(set_c)::value;
    if constexpr (MASK) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const uint32_t u = kb[P][i];
        const uint32_t b8 = (u & 0xFu) | ((u >> 4) & 0xF0u);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const uint32_t lo = 0u - ((b8 >> (2 * d)) & 1u), hi = 0u - ((b8 >> (2 * d + 1)) & 1u);
          const uint32_t m = (lo & 0xFFFFu) | (hi << 16);
          pa[P][i][0][d] &= m;
          pa[P][i][1][d] &= m;
          pa[P][i][2][d] &= m;
        }
      }
    }
    // smallest terms first: (a2 b0), (a0 b2), (a1 b1), (a1 b0), (a0 b1), (a0 b0); term-major over the TM accumulator tiles, so
    // consecutive MFMAs never depend on each other
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (SDUMC_P3_DBG & 4) acc[i][j][t] += __uint_as_float(pa[P][i][TA[t]][0] ^ pb[S][j][TB[t]][1]);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(pa[P][i][TA[t]]), op(pb[S][j][TB[t]]), acc[i][j], 0, 0, 0);
        }
  };
  auto interleave = [&]() {      // one memory instruction in the shadow of every MFMA as long as there are any
#pragma unroll
    for (int u = 0; u < 6 * TM * TN; ++u) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
    }
  };

  // ---- prologue: the queue as the steady state leaves it -- A0 .. A(NST-4), [A(NST-3), B0], [A(NST-2), B1], [A(NST-1), B2] ----
  static_assert(NST >= 4 && DB == 3 && NBS == 4, "the prologue and the unrolled steps are written for these depths");
#pragma unroll
  for (int s0 = 0; s0 <= NST - 4; ++s0) issue_a(s0);
  issue_a(NST - 3);
  load_b(std::integral_constant<int, 0>{});
  issue_a(NST - 2);
  load_b(std::integral_constant<int, 1>{});
  issue_a(NST - 1);
  load_b(std::integral_constant<int, 2>{});
  if constexpr (NST > 4) __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
  wait_groups(std::integral_constant<int, 3>{});          // stage 0 has landed
  __builtin_amdgcn_s_barrier();
  if (stamp) *stamp = __builtin_amdgcn_s_memrealtime();
  load_a(lds, std::integral_constant<int, 0>{});
  int nbuf = 1;                                           // buffer of stage t + 1; stage t's (refilled with t + NST) is the one before it
  // step t: STEADY = stage t + NST and k-tile t + DB exist (every step issues, waits with the same count, has no tail logic)
  auto step = [&](int t, auto par, auto set_c, auto steady_c) {
    constexpr int P = decltype(par)::value, S = decltype(set_c)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    const bool more = STEADY || t + 1 < nk;
    if (STEADY) wait_groups(std::integral_constant<int, DB - 1>{});
    else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    if (more) {
      __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0): this wave's reads of stage t's buffer are done (it is refilled below)
      __builtin_amdgcn_s_barrier();
      if (STEADY || t + NST < nk) issue_a(nbuf == 0 ? NST - 1 : nbuf - 1);
      if (STEADY || t + DB < nk) load_b(std::integral_constant<int, (S + DB) % NBS>{});
      load_a(lds + nbuf * CF::STAGE, std::integral_constant<int, P ^ 1>{});
      nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
    }
    mfmas(par, set_c);
    if (STEADY) interleave();
  };
  int t = 0;
  for (; t + 3 + NST < nk; t += 4) {
    step(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::true_type{});
    step(t + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, std::true_type{});
    step(t + 2, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, std::true_type{});
    step(t + 3, std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{}, std::true_type{});
  }
  for (; t < nk; t += 4) {
    step(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::false_type{});
    step(t + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, std::false_type{});
    step(t + 2, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, std::false_type{});
    step(t + 3, std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{}, std::false_type{});
  }

#endif
}

}  // namespace sdumc_p3
