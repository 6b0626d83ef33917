// gemm_group.hip — every weight gradient of a backward phase in ONE persistent launch plus one ordered reduce (gfx950).
//
// What runs here (C2 shapes): dW = dz^T x (+ db = column sums of dz) of frame_dim_reshape_{0,1,2} (model :282-284; 256 x 1024
// / 4096 outputs over K = B*T up to 24000 rows), of the six input_proj layers of FRA2UTT_new / Cross_Attention (model :60, :82;
// 256 x 256 outputs over K = 2*B*T up to 48000 rows, input dropout fused on x) and of the utterance-level Linear layers
// (model :293-368; K = 128 or 896 rows): 51 of the step's 127 GFLOP, all of it autograd of main :149.  Nothing but Adam (and a
// data-parallel all-reduce) reads these results, so they are off the step's dependency chain; what matters is that they cost
// as little matrix-core time as possible.
//
// Why not one split-K GEMM per layer (round 2: 8 + 15 launches, 10 reduce launches, 6 column-sum launches): a weight gradient
// is a small output (16-64 tiles of 128 x 128) over a very long K, so each launch had to split K 16-32 ways to fill 256 CUs,
// every workgroup paid a prologue, a slab and a share of the reduce for < 50 k-tiles of work, ~16 us per launch were fixed,
// and the last dispatch round of every launch ran half empty.
//
// Structure (stream-K): the k-tiles (16 rows of K) of all output tiles (256 x 128) of all problems form ONE line.  The launch
// has one 512-thread workgroup per CU; workgroup w multiplies the contiguous range [w L / n, (w + 1) L / n) of the line --
// every CU the same number of k-tiles (+-1) whatever the shapes.  A range covers the tail of one tile, some whole tiles and the
// head of another: a piece that is a whole tile is written straight to C; any other piece goes to an fp32 slab slot
// (slot = workgroup + global unit index: unique, no table), and the reduce launch -- one workgroup per (tile, sixteenth of
// the tile) -- sums a tile's slots in ascending k order.  No atomics, no flags, no spinning: the kernel boundary publishes the
// slabs, and the sum order is a function of the shapes only (bit-identical from run to run).
// Inner loop: gemm_wide.hip's LDS-DMA ring (buffer_load ... lds, 4 stages, one raw s_barrier per k-tile, counted vmcnt), both
// operands row-contiguous ([k][row] tiles, conflict-free ds_read_b32 fragments), a wave owns 64 x 64 of the 256 x 128 tile =
// 4 independent 32x32x2 fp32 MFMA chains; the keep-bits of the fused input dropout ride the ring as their own 512-byte tile;
// the bias gradient is accumulated from the A fragments on the VALU in the MFMAs' shadow.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "common.h"

namespace sdumc_gg {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(4))) int32_t const_i32_t;      // (a load through it with a wave-uniform index is a scalar load)
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) i32x4 const_i32x4_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

constexpr int waitcnt_vm(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0xF << 8); }

constexpr int BM = 256, BN = 128, BK = 16;
constexpr int NW = 8, NTHR = 64 * NW, WGN = 2;       // 4 x 2 waves, each 64 x 64
constexpr int WM = 64, WN = 64, TM = 2, TN = 2;
constexpr int A_BYTES = BK * BM * 4, B_BYTES = BK * BN * 4, BITS_BYTES = BK * (BN / 4);
constexpr int A_CH = A_BYTES / 1024, B_CH = B_BYTES / 1024;   // 1-KiB pieces = one wave-instruction each
constexpr int NI = (A_CH + B_CH) / NW;                         // data pieces per wave per stage
[[maybe_unused]] constexpr int BITS_CH = BITS_BYTES / 256;
constexpr int STAGE_BYTES = A_BYTES + B_BYTES + BITS_BYTES;
constexpr int SLOT_FLOATS = BM * BN + BM;                      // a partial tile in register order + its column sums of A
static_assert((A_CH + B_CH) % NW == 0 && A_CH % NW == 0, "pieces divide evenly over the waves");
static_assert(NI == 3, "issue() below is written for 2 A rows + 1 B piece per wave");

constexpr int MAXP = SDUMC_GG_MAX_PROBLEMS;

struct Launch {
  sdumc_gg_problem p[MAXP];
  int32_t xcd;               // > 0: the workgroups' ranges follow an XCD-major order of the physical workgroup ids (see wg_logical)
  int32_t line0[MAXP + 1];   // first line position (k-tiles) of problem i; line0[n] = L
  int32_t unit0[MAXP + 1];   // first global unit index of problem i
  int32_t nchunk[MAXP];      // K of a problem's tiles cut into this many chunks, line order (chunk, tile, k): tiles of one chunk
                             // (which read the same rows of A) are neighbours on the line.  1 = plain (tile, k) order
  int32_t n;
  int32_t nwg;
  int32_t bk;                // rows of K per k-tile: 16 (fp32 operands) or 64 (bf16 operands)
  int32_t hf;                // 1: A and B are bf16 tensors (gg_tn_bf16_kernel); the accumulator tiles then keep their natural row order
  int32_t bn;                // width of an output tile: 128, or 256 (gg_tn_split2_kernel)
  int32_t nat;               // 1: the accumulator tiles of the slabs are in natural row / column order (gg_tn_split2_kernel, bf16)
  int32_t ovh;               // line positions charged to every unit for its fixed costs (ring start-up latency, epilogue): no k-tiles behind them
  float* slab;
};

struct Geo {       // derived shape of one problem
  int ntn, ntiles, nk0, nk;
};
__host__ __device__ inline Geo geo_of(const sdumc_gg_problem& p, int bk, int bn) {
  Geo g;
  g.ntn = (p.N + bn - 1) / bn;
  g.ntiles = g.ntn * ((p.M + BM - 1) / BM);
  g.nk0 = (p.K[0] + bk - 1) / bk;
  g.nk = g.nk0 + (p.K[1] > 0 ? (p.K[1] + bk - 1) / bk : 0);
  return g;
}
// (32-bit products: the host refuses lines with L * nwg >= 2^31)
__host__ __device__ inline int chunk_k(int c, int nk, int nchunk) { return (int)(((uint32_t)c * (uint32_t)nk) / (uint32_t)nchunk); }
__host__ __device__ inline int range_begin(int w, int L, int nwg) { return (int)(((uint32_t)w * (uint32_t)L) / (uint32_t)nwg); }
// the workgroup whose range holds line position x
__host__ __device__ inline int wg_of(int x, int L, int nwg) { return (int)((((uint32_t)x + 1u) * (uint32_t)nwg - 1u) / (uint32_t)L); }

struct Where {     // a line position resolved
  int p, chunk, tile, kt;      // problem, chunk, tile inside the problem, k-tile inside the problem's K
  int unit_end;                // line position one past this unit
  int kt_end;                  // k-tile one past this unit's k range
  int unit;                    // global unit index
  int ko;                      // position inside the unit (the first L.ovh positions carry no k-tiles)
};
__device__ __forceinline__ Where locate(const Launch& L, int x) {
  Where w;
  int p = 0;
  while (p + 1 < L.n && L.line0[p + 1] <= x) ++p;
  const Geo g = geo_of(L.p[p], L.bk, L.bn);
  const int nc = L.nchunk[p], ovh = L.ovh;
  const int xr = x - L.line0[p];
  // chunk c starts at line offset ntiles * (chunk_k(c) + c * ovh)
  int c = nc == 1 ? 0 : (int)(((uint32_t)xr * (uint32_t)nc) / ((uint32_t)g.ntiles * (uint32_t)(g.nk + nc * ovh)));
  if (c > nc - 1) c = nc - 1;
  while (c + 1 < nc && g.ntiles * (chunk_k(c + 1, g.nk, nc) + (c + 1) * ovh) <= xr) ++c;
  while (c > 0 && g.ntiles * (chunk_k(c, g.nk, nc) + c * ovh) > xr) --c;
  const int k0 = chunk_k(c, g.nk, nc), k1 = chunk_k(c + 1, g.nk, nc), len = k1 - k0 + ovh;
  const int y = xr - g.ntiles * (k0 + c * ovh);
  const int t = y / len, ko = y - t * len;
  w.p = p;
  w.chunk = c;
  w.tile = t;
  w.kt = k0 + max(ko - ovh, 0);
  w.kt_end = k1;
  w.unit_end = x - ko + len;
  w.unit = L.unit0[p] + c * g.ntiles + t;
  w.ko = ko;
  return w;
}

// Workgroup ids are dealt round-robin over the 8 XCDs (each with its own L2): physical id w runs on XCD w % 8.  With L.xcd the ranges
// of the line are handed out XCD-major -- the workgroups of ONE XCD hold CONSECUTIVE ranges -- so that the units of one k-chunk (the
// n-tiles that re-read the same rows of A) run side by side under one L2.
__device__ __forceinline__ int wg_logical(const Launch& L, int w) {
  if (L.xcd <= 0) return w;
  const int per = L.nwg / L.xcd, full = per * L.xcd;      // (ids beyond the last whole round keep their place)
  return w < full ? (w % L.xcd) * per + w / L.xcd : w;
}

// what a wave holds of one MFMA group (8 k) of a k-tile: element s of a fragment is k = 8 gq + 4 lh + s
struct Frag {
  f32x4 a[TM], b[TN];
  uint32_t mb[4];         // keep-bits bytes of the B fragment (masked problems)
};

// (no packed fp32 VALU operations anywhere in the library: Makefile, NOPACK)
template <int NST>
__device__ __forceinline__ void gg_tn_body(const Launch& L, char* lds) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int LL = L.line0[L.n];
  const int wg = wg_logical(L, (int)blockIdx.x);
  int x = range_begin(wg, LL, L.nwg);
  const int x_end = range_begin(wg + 1, LL, L.nwg);
  constexpr int PER = NI + 1;      // vector-memory operations per wave per stage: 2 A rows, 1 B piece, 1 keep-bits piece (issued
                                   // for unmasked problems too, against an empty descriptor: the vmcnt bookkeeping stays one)
  static_assert(NST >= 3 && BK == 16, "the loop below is written for two MFMA groups per k-tile");

  f32x16 acc[TM][TN];
  float csum[TM];
  // Fragment reads.  The operands lie [k][row] in LDS, so a lane's four k values of one 32-row MFMA tile are four ds_read_b32 a
  // KiB apart.  Reading 8 bytes instead gives the lane rows 2 li and 2 li + 1: the wave's two MFMA tiles along M (and along N)
  // take the EVEN and the ODD rows of its 64-row block instead of the lower and the upper half -- half the LDS instructions
  // and LDS cycles, conflict-free (a lane group covers 256 contiguous bytes).  The permutation is undone where the accumulators
  // leave the registers: row (i, r) of the block is 2 r + i, column (j, c) is 2 c + j.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto read_frag = [&](const char* base, int gq, Frag& f, auto mask_c) {
    constexpr bool MASK = decltype(mask_c)::value;
    const float* As = reinterpret_cast<const float*>(base) + (8 * gq + 4 * lh) * BM + wm0 + 2 * li;
    const float* Bs = reinterpret_cast<const float*>(base + A_BYTES) + (8 * gq + 4 * lh) * BN + wn0 + 2 * li;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f32x2 va = *reinterpret_cast<const f32x2*>(As + s * BM);
      const f32x2 vb = *reinterpret_cast<const f32x2*>(Bs + s * BN);
      f.a[0][s] = va[0];
      f.a[1][s] = va[1];
      f.b[0][s] = vb[0];
      f.b[1][s] = vb[1];
    }
    if constexpr (MASK) {   // columns 2 li, 2 li + 1 share one keep-bits byte (bits 2 (li & 1) and 2 (li & 1) + 1)
      const uint8_t* bt = reinterpret_cast<const uint8_t*>(base + A_BYTES + B_BYTES) + (8 * gq + 4 * lh) * (BN / 4) + ((wn0 + 2 * li) >> 2);
#pragma unroll
      for (int s = 0; s < 4; ++s) f.mb[s] = bt[s * (BN / 4)];
    }
  };
  // the 16 MFMAs of one group (+ the keep-bits applied to the B fragment, + the column sums of A on the VALU)
  auto mma = [&](Frag& f, auto mask_c) {
    constexpr bool MASK = decltype(mask_c)::value;
    if constexpr (MASK) {   // keep-bit -> 0 / ~0 (v_bfe_i32), AND on the float's bits; the scale 1 / (1 - p) multiplies the tile once, at its end
      const uint32_t pos = 2u * (li & 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)f.mb[s], pos, 1u);
        const uint32_t m1 = (uint32_t)__builtin_amdgcn_sbfe((int)f.mb[s], pos + 1u, 1u);
        f.b[0][s] = __uint_as_float(__float_as_uint(f.b[0][s]) & m0);
        f.b[1][s] = __uint_as_float(__float_as_uint(f.b[1][s]) & m1);
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < TM; ++i) csum[i] += f.a[i][s];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][s], f.b[j][s], acc[i][j], 0, 0, 0);
    }
  };
  // all but the `stages` youngest stages of this wave's LDS-DMA have landed
  auto wait_stages = [&](int stages) {
    if (stages >= 2) __builtin_amdgcn_s_waitcnt(waitcnt_vm(2 * PER));
    else if (stages == 1) __builtin_amdgcn_s_waitcnt(waitcnt_vm(PER));
    else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
  };

  while (x < x_end) {
    const Where w = locate(L, x);
    const sdumc_gg_problem& pr = L.p[w.p];
    const Geo g = geo_of(pr, L.bk, L.bn);
    const int px_end = min(x_end, w.unit_end);
    // k-tiles [ka, kb) of the problem's concatenated K (a range that ends inside the unit's overhead positions holds none)
    const int ka = w.kt, kb = w.kt_end - (w.unit_end - px_end);
    if (kb <= ka) {
      x = px_end;
      continue;
    }
    const int tile_m = w.tile / g.ntn, tile_n = w.tile - tile_m * g.ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      csum[i] = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    bool seg_masked = false;
#pragma nounroll
    for (int seg = 0; seg < 2; ++seg) {
      // k-tiles [t0, t0 + nk) of K-segment `seg`
      int t0, nk;
      if (seg == 0) {
        if (ka >= g.nk0) continue;
        t0 = ka;
        nk = min(kb, g.nk0) - ka;
      } else {
        if (kb <= g.nk0) continue;
        t0 = max(ka, g.nk0) - g.nk0;
        nk = kb - g.nk0 - t0;
      }
      const int kbeg = t0 * BK;
      const int segK = pr.K[seg], seg_mod = pr.b_row_mod[seg];
      const uint32_t lda4 = (uint32_t)pr.lda * 4u, ldb4 = (uint32_t)pr.ldb * 4u, qw = (uint32_t)pr.bits_qw;
      const bool masked = pr.b_bits[seg] != nullptr;
      // Rows at and beyond K are outside the descriptors' ranges: the hardware returns zeros for them, so a ragged last k-tile
      // needs no special case (with a row modulo B stays in range there, against zeros of A).
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pr.A[seg]), 0, (int)((uint32_t)segK * lda4), 0x00020000);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pr.B[seg]), 0, (int)((uint32_t)(seg_mod > 0 ? seg_mod : segK) * ldb4), 0x00020000);
      const __amdgpu_buffer_rsrc_t rbits = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(masked ? pr.b_bits[seg] : (const uint8_t*)pr.A[seg]), 0,
                                                                             masked ? (int)((uint32_t)segK * qw) : 0, 0x00020000);
      // pieces of this wave: A k-rows `wave` and `wave + 8` (64 lanes x 4 columns = the 256 columns of the tile), B piece `wave`
      // = k-rows 2 wave, 2 wave + 1 (32 lanes x 4 columns each), keep-bits piece wave % 2
      uint32_t voff[NI], bvoff;
      int srck = 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) voff[i] = (uint32_t)(kbeg + wave + 8 * i) * lda4 + (uint32_t)min(m0 + 4 * lane, pr.M - 4) * 4u;
      {
        int kr = kbeg + 2 * wave + (lane >> 5);
        if (seg_mod > 0) kr %= seg_mod;
        srck = kr;
        voff[2] = (uint32_t)kr * ldb4 + (uint32_t)min(n0 + 4 * (lane & 31), pr.N - 4) * 4u;
      }
      {
        const int idx = ((wave % BITS_CH) << 6) + lane;         // dword index inside the [16][8 dwords] bits tile
        bvoff = (uint32_t)(kbeg + (idx >> 3)) * qw + (uint32_t)(n0 >> 2) + 4u * (idx & 7);
      }
      auto issue = [&](int buf) {
        char* base = lds + buf * STAGE_BYTES;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(base + wave * 1024), 16, voff[0], 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(base + (wave + 8) * 1024), 16, voff[1], 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t*)(base + A_BYTES + wave * 1024), 16, voff[2], 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rbits, (lds_void_t*)(base + A_BYTES + B_BYTES + (wave % BITS_CH) * 256), 4, bvoff, 0, 0, 0);
        voff[0] += (uint32_t)BK * lda4;
        voff[1] += (uint32_t)BK * lda4;
        voff[2] += (uint32_t)BK * ldb4;
        bvoff += (uint32_t)BK * qw;
        if (seg_mod > 0) {
          srck += BK;
          if (srck >= seg_mod) { srck -= seg_mod; voff[2] -= (uint32_t)seg_mod * ldb4; }
        }
      };
      // Software pipeline over the k-tiles of the sub-piece.  A k-tile is two MFMA groups g0, g1 of 16 MFMAs.  While g0 of tile
      // t multiplies, the fragments of g1 are read; then -- in the MIDDLE of the tile, with the matrix pipe still draining g0 --
      // the wave waits for its own pieces of stage t + 1, meets the others at the barrier, issues the LDS-DMA of stage
      // t + NST - 1 and reads the g0 fragments of tile t + 1, all under the 16 MFMAs of g1.  So the barrier, the DMA issue (60-180
      // cycles per instruction) and the LDS latency of the next tile's first fragments are covered by matrix work of the same
      // wave, not exposed at the head of every tile with all eight waves in lock-step.
      // (Tried on top of this: waves 4-7 taking the barrier at the head of the tile, half a tile behind waves 0-3, so that the two
      //  waves of a SIMD never sit at the barrier together -- 114.0 vs 114.8 TF on the frame problems, no gain: dropped.)
      auto ring = [&](auto mask_c) {
        Frag f0, f1;
#pragma unroll
        for (int s = 0; s < NST - 1; ++s)
          if (s < nk) issue(s);
        wait_stages(min(nk, NST - 1) - 1);
        __builtin_amdgcn_s_barrier();
        read_frag(lds, 0, f0, mask_c);
        int buf = 0, nbuf = 1, ibuf = NST - 1;
#pragma nounroll
        for (int t = 0; t < nk; ++t) {
          read_frag(lds + buf * STAGE_BYTES, 1, f1, mask_c);
          __builtin_amdgcn_sched_barrier(0);
          mma(f0, mask_c);
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < nk) {
            wait_stages(min(nk - t - 2, NST - 3));
            __builtin_amdgcn_s_barrier();
            if (t + NST - 1 < nk) issue(ibuf);
            read_frag(lds + nbuf * STAGE_BYTES, 0, f0, mask_c);
          }
          __builtin_amdgcn_sched_barrier(0);
          mma(f1, mask_c);
          __builtin_amdgcn_sched_barrier(0);
          buf = nbuf;
          nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
          ibuf = ibuf + 1 == NST ? 0 : ibuf + 1;
        }
      };
      // ---- fp32 products on the bf16 pipe (the arithmetic of gg_tn_split2_kernel below, of gemm_wide / gemm_rows / gemm_p3 / K3) ------------------------------------------------------------------------
      // v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate (157 TF against 2.5 PF dense).  An fp32 value is the EXACT sum of
      // three bf16 values -- a0 = a truncated to its top 16 bits, a1 = (a - a0) truncated, a2 = a - a0 - a1 (8 + 8 + 8
      // significand bits; both subtractions are exact) -- so a b = sum of nine bf16 x bf16 products, each exact in fp32.  The six
      // largest (a0 b0, a0 b1, a1 b0, a0 b2, a1 b1, a2 b0) are accumulated in fp32 by six v_mfma_f32_32x32x16_bf16; the three
      // dropped ones are below 2^-22 |a b| together (|a1| < 2^-7 |a|, |a2| < 2^-15 |a|), the size of the rounding an fp32 FMA chain
      // commits on every one of its K steps.  Six 32-cycle MFMAs replace eight 64-cycle ones per 32 x 32 x 16 block (2.67x the
      // matrix rate); the split costs 5.5 VALU operations per fragment value (and, sub, and, sub + three half v_perm_b32 that
      // pack two values' top halves into one operand dword), which run beside the MFMAs of the previous k-tile.
      // A lane's eight k of one operand are k = 4 lh + s and 8 + 4 lh + s (s < 4) -- the two fp32 fragments of the k-tile side
      // by side; A and B use the same assignment, which is all a contraction needs.
      // (the in-fragment form of this kernel -- every wave splitting the fragments it multiplies, 136-147 TF -- was superseded by
      //  gg_tn_split2_kernel in round 4 and removed in round 5; this body is the fp32-MFMA form, sdumc_set_split_ bit 0 off)
      if (masked) ring(std::true_type{});
      else ring(std::false_type{});
      seg_masked = masked;
      __builtin_amdgcn_s_barrier();          // every wave is done reading the ring before the next sub-piece refills it
    }
    if (seg_masked) {   // (both segments of a problem are masked or neither: checked on the host)
      const float sc = pr.b_scale;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] *= sc;
    }
    const bool direct = L.nchunk[w.p] == 1 && ka == 0 && kb == g.nk;
    const bool do_cs = pr.colsum_a != nullptr && tile_n == 0 && wn0 == 0;
    if (direct) {
      // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
      // (with the column permutation a lane's j = 0 / 1 values are NEIGHBOURS in C, columns 2 li and 2 li + 1)
      const bool accum = pr.accumulate != 0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = n0 + wn0 + 2 * li + j;
          if (col >= pr.N) continue;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = m0 + wm0 + 2 * ((e & 3) + 8 * (e >> 2) + 4 * lh) + i;
            if (row >= pr.M) continue;
            float* dst = pr.C + ((uint32_t)row * (uint32_t)pr.ldc + (uint32_t)col);
            float v = acc[i][j][e];
            if (accum) v += *dst;
            *dst = v;
          }
        }
      if (do_cs) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float v = csum[i] + __shfl_xor(csum[i], 32, 64);
          const int m = m0 + wm0 + 2 * li + i;
          if (lh == 0 && m < pr.M) {
            float* dst = pr.colsum_a + m;
            *dst = accum ? *dst + v : v;
          }
        }
      }
    } else {
      // slab slot in register order: [wave][i][j][e / 4][lane][4] -- every store instruction of a wave is 1 KiB contiguous
      float* slot = L.slab + (size_t)(wg + w.unit) * SLOT_FLOATS;
      float* mine = slot + (uint32_t)(wave * 16 * 64 + lane) * 4u;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *reinterpret_cast<f32x4*>(mine + ((i * TN + j) * 4 + q) * 256) = v;
          }
      if (do_cs) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float v = csum[i] + __shfl_xor(csum[i], 32, 64);
          if (lh == 0) slot[BM * BN + wm0 + 2 * li + i] = v;
        }
      }
    }
    x = px_end;
  }
#endif
}
template <int NST, int OCC>
__global__ __launch_bounds__(NTHR, OCC) void gg_tn_kernel(const Launch L) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  gg_tn_body<NST>(L, lds);
}

// ------------------------------------------------------------------------------------------------------------------------
// The split form with the operands split ONCE per workgroup (gg_tn_split2_kernel).  In the in-fragment form of round 4 (removed) every
// wave split the fragments it multiplied -- a 64-row block of A is split by two waves,
// a 64-column block of B by four -- and the ~230 VALU operations per wave and k-tile, not the 24 MFMAs, bound the loop
// (9.7 VALU instructions per MFMA in the counters; 136-147 TF).  Here, per k-tile:
//   1. the raw fp32 stage (the same LDS-DMA ring) is converted by all 512 threads: a thread takes 8 k of one A row and 4 k of one
//      B column (column reads of the [k][row] tiles: conflict-free ds_read_b32), applies the keep-bits, adds its A values to its
//      column-sum register, splits, and writes the parts k-contiguous into three bf16 planes ([row][16 k] per plane; the two 16-byte
//      halves of a row swapped by bit 3 of the row: conflict-free ds_read_b128 / ds_write_b128);   66 VALU operations per thread
//   2. barrier; every wave reads its operands -- one ds_read_b128 per plane and 32-row block -- and issues the 24 MFMAs;
//   3. barrier (the planes are single-buffered: 36 KB beside the 5-stage raw ring = 158.5 KB of LDS), LDS-DMA issue of the stage
//      that goes into the buffer just converted.
// The accumulator tiles keep their natural row / column order (Launch.nat: the reduce kernel's index arithmetic).
// ------------------------------------------------------------------------------------------------------------------------
#ifndef SDUMC_GG_DBG
#define SDUMC_GG_DBG 0      /* measurement builds only (wrong results): bit 0 = no DMA, bit 1 = no conversion, bit 2 = no MFMAs */
#endif
namespace s2 {
[[maybe_unused]] constexpr int TM2 = 4;
constexpr int BN2 = 256, WGN2 = 4, WM2 = 128, WN2 = 64, TN2 = 2;      // 2 x 4 waves, each 128 x 64
constexpr int B2_BYTES = BK * BN2 * 4, BITS2_BYTES = BK * (BN2 / 4), STAGE2 = A_BYTES + B2_BYTES + BITS2_BYTES;   // 33 KB
constexpr int PA_BYTES = BM * BK * 2, PB_BYTES = BN2 * BK * 2;        // one plane of A / B of a k-tile: 8 KB each
constexpr int PLANES_BYTES = 3 * (PA_BYTES + PB_BYTES);               // 48 KB
constexpr int NST2 = 3;
constexpr int LDS2 = NST2 * STAGE2 + PLANES_BYTES;        // 147 KB (the column-sum exchange at a unit's end uses the planes' memory)
constexpr int SLOT2_FLOATS = BM * BN2 + BM;
static_assert(LDS2 <= 160 * 1024, "LDS");
}  // namespace s2

__global__ __launch_bounds__(NTHR, 2) void gg_tn_split2_kernel(const Launch L) {
#if defined(__HIP_DEVICE_COMPILE__)
  using namespace s2;
  constexpr int NST = NST2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // (the planes first: their addresses then fit the 16-bit offset field of the LDS instructions, plane by plane, from one
  //  register per operand -- behind the ring every (operand, plane) pair would need a register of its own)
  char* const planes = lds;
  char* const ring0 = lds + PLANES_BYTES;
  float* const cs_x = reinterpret_cast<float*>(planes);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm0 = (wave / WGN2) * WM2, wn0 = (wave % WGN2) * WN2;
  const int LL = L.line0[L.n];
  const int wg = wg_logical(L, (int)blockIdx.x);
  int x = range_begin(wg, LL, L.nwg);
  const int x_end = range_begin(wg + 1, LL, L.nwg);
  constexpr int PER = 5;      // 2 A rows, 2 B rows, 1 keep-bits piece per wave and stage

  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2s __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  auto pk = [](float a, float b) -> uint32_t {       // v_cvt_pk_bf16_f32 (round to nearest even), low half = a
    const f32x2s v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  };
  auto split2 = [&](float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = pk(a, b);
    const float a1 = a - __uint_as_float(p0 << 16), b1 = b - __uint_as_float(p0 & 0xFFFF0000u);       // exact
    p1 = pk(a1, b1);
    const float a2 = a1 - __uint_as_float(p1 << 16), b2 = b1 - __uint_as_float(p1 & 0xFFFF0000u);     // exact
    p2 = pk(a2, b2);
  };
  auto op = [](const u32x4& v) { return __builtin_bit_cast(bf16x8, v); };

  // conversion roles: row ar of A and column ar of B, k 8 ag .. + 7 of both
  const int ar = tid & (BM - 1), ag = tid >> 8;
  const uint32_t a_rd = (uint32_t)((8 * ag) * BM + ar) * 4u;                                  // + s * BM * 4
  const uint32_t a_wr = (uint32_t)(ar * (BK * 2) + ((ag ^ ((ar >> 3) & 1)) << 4));
  const uint32_t b_rd = (uint32_t)(A_BYTES + ((8 * ag) * BN2 + ar) * 4);                      // + s * BN2 * 4
  const uint32_t b_wr = (uint32_t)(3 * PA_BYTES) + a_wr;
  const uint32_t bit_rd = (uint32_t)(A_BYTES + B2_BYTES + (8 * ag) * (BN2 / 4) + (ar >> 2));  // + s * (BN2 / 4)
  const uint32_t bit_pos = (uint32_t)(ar & 3);
  // operand reads: rows (columns) 32 i + li of the wave's block, k group lh
  uint32_t a_op[TM2], b_op[TN2];
#pragma unroll
  for (int i = 0; i < TM2; ++i) {
    const int r = wm0 + 32 * i + li;
    a_op[i] = (uint32_t)(r * (BK * 2) + ((lh ^ ((r >> 3) & 1)) << 4));
  }
#pragma unroll
  for (int j = 0; j < TN2; ++j) {
    const int c = wn0 + 32 * j + li;
    b_op[j] = (uint32_t)(3 * PA_BYTES + c * (BK * 2) + ((lh ^ ((c >> 3) & 1)) << 4));
  }

  f32x16 acc[TM2][TN2];
  auto wait_n = [&](int n) {
    if (n >= 4) __builtin_amdgcn_s_waitcnt(waitcnt_vm(4 * PER));
    else if (n == 3) __builtin_amdgcn_s_waitcnt(waitcnt_vm(3 * PER));
    else if (n == 2) __builtin_amdgcn_s_waitcnt(waitcnt_vm(2 * PER));
    else if (n == 1) __builtin_amdgcn_s_waitcnt(waitcnt_vm(PER));
    else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
  };

  while (x < x_end) {
    const Where w = locate(L, x);
    const sdumc_gg_problem& pr = L.p[w.p];
    const Geo g = geo_of(pr, L.bk, L.bn);
    const int px_end = min(x_end, w.unit_end);
    const int ka = w.kt, kb = w.kt_end - (w.unit_end - px_end);
    if (kb <= ka) {
      x = px_end;
      continue;
    }
    const int tile_m = w.tile / g.ntn, tile_n = w.tile - tile_m * g.ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN2;
    float csum = 0.f;
#pragma unroll
    for (int i = 0; i < TM2; ++i)
#pragma unroll
      for (int j = 0; j < TN2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bool seg_masked = false;
#pragma nounroll
    for (int seg = 0; seg < 2; ++seg) {
      int t0, nk;
      if (seg == 0) {
        if (ka >= g.nk0) continue;
        t0 = ka;
        nk = min(kb, g.nk0) - ka;
      } else {
        if (kb <= g.nk0) continue;
        t0 = max(ka, g.nk0) - g.nk0;
        nk = kb - g.nk0 - t0;
      }
      const int kbeg = t0 * BK;
      const int segK = pr.K[seg], seg_mod = pr.b_row_mod[seg];
      const uint32_t lda4 = (uint32_t)pr.lda * 4u, ldb4 = (uint32_t)pr.ldb * 4u, qw = (uint32_t)pr.bits_qw;
      const bool masked = pr.b_bits[seg] != nullptr;
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pr.A[seg]), 0, (int)((uint32_t)segK * lda4), 0x00020000);
      const __amdgpu_buffer_rsrc_t rbits = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(masked ? pr.b_bits[seg] : (const uint8_t*)pr.A[seg]), 0,
                                                                             masked ? (int)((uint32_t)segK * qw) : 0, 0x00020000);
      uint32_t voff[3], bvoff;
      // A wave's piece of B is one whole k-row (1 KiB = 256 columns), so the row's address is wave-uniform: a descriptor per row, built
      // on the scalar unit -- base = B + source row * ldb, one row of records, ZERO records for rows at and beyond K (zeros, as rows
      // outside a whole-tensor descriptor's range would be).  Source row of k-row r: r, or r % b_row_mod (the streams share x_audio /
      // x_video), or b_map[r] (B is a resident store's packed tensor, of any size: the batch is read in place; the entry is fetched
      // with a scalar load one k-tile ahead).
      const_i32_t* const bmap = (const_i32_t*)(uintptr_t)pr.b_map[seg];
      const bool mapped = pr.b_map[seg] != nullptr;
      int kcur[2], srow[2];      // k-row of this wave's two B pieces at the next issue / its source row
#pragma unroll
      for (int i = 0; i < 2; ++i) voff[i] = (uint32_t)(kbeg + wave + 8 * i) * lda4 + (uint32_t)min(m0 + 4 * lane, pr.M - 4) * 4u;
      voff[2] = (uint32_t)min(n0 + 4 * lane, pr.N - 4) * 4u;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        kcur[i] = kbeg + wave + 8 * i;
        srow[i] = mapped ? bmap[min(kcur[i], segK - 1)] : (seg_mod > 0 ? kcur[i] % seg_mod : kcur[i]);
      }
      {
        const int idx = ((wave & 3) << 6) + lane;               // dword index inside the [16][16 dwords] bits tile
        bvoff = (uint32_t)(kbeg + (idx >> 4)) * qw + (uint32_t)(n0 >> 2) + 4u * (idx & 15);
      }
      auto issue = [&](int buf) {
        char* base = ring0 + buf * STAGE2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(base + wave * 1024), 16, voff[0], 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(base + (wave + 8) * 1024), 16, voff[1], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const __amdgpu_buffer_rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pr.B[seg]) + (size_t)srow[i] * (size_t)pr.ldb, 0,
                                                                                kcur[i] < segK ? (int)ldb4 : 0, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rrow, (lds_void_t*)(base + A_BYTES + (wave + 8 * i) * 1024), 16, voff[2], 0, 0, 0);
          kcur[i] += BK;
          if (mapped) srow[i] = bmap[min(kcur[i], segK - 1)];      // (the next k-tile's row: back long before the next issue)
          else {
            srow[i] += BK;
            if (seg_mod > 0 && srow[i] >= seg_mod) srow[i] -= seg_mod;
          }
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rbits, (lds_void_t*)(base + A_BYTES + B2_BYTES + (wave & 3) * 256), 4, bvoff, 0, 0, 0);
        voff[0] += (uint32_t)BK * lda4;
        voff[1] += (uint32_t)BK * lda4;
        bvoff += (uint32_t)BK * qw;
      };
      // (`masked` is a runtime branch around eight byte reads here, not a second instantiation of the loop: with two copies
      //  of the k loop the register allocator kept five of the eight accumulator tiles in scratch)
      auto convert = [&](const char* base) {
        float av[8], bv[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) av[s] = *reinterpret_cast<const float*>(base + a_rd + s * (BM * 4));
#pragma unroll
        for (int s = 0; s < 8; ++s) bv[s] = *reinterpret_cast<const float*>(base + b_rd + s * (BN2 * 4));
        if (masked) {
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            const uint32_t byte = *reinterpret_cast<const uint8_t*>(base + bit_rd + s * (BN2 / 4));
            bv[s] = __uint_as_float(__float_as_uint(bv[s]) & (uint32_t)__builtin_amdgcn_sbfe((int)byte, bit_pos, 1u));
          }
        }
        csum += ((av[0] + av[1]) + (av[2] + av[3])) + ((av[4] + av[5]) + (av[6] + av[7]));
        uint32_t q[3][4];
#pragma unroll
        for (int d = 0; d < 4; ++d) split2(av[2 * d], av[2 * d + 1], q[0][d], q[1][d], q[2][d]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(planes + pl * PA_BYTES + a_wr) = u32x4{q[pl][0], q[pl][1], q[pl][2], q[pl][3]};
#pragma unroll
        for (int d = 0; d < 4; ++d) split2(bv[2 * d], bv[2 * d + 1], q[0][d], q[1][d], q[2][d]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(planes + pl * PB_BYTES + b_wr) = u32x4{q[pl][0], q[pl][1], q[pl][2], q[pl][3]};
      };
      auto multiply = [&]() {
        u32x4 pb[TN2][3], pa[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) pa[0][pl] = *reinterpret_cast<const u32x4*>(planes + pl * PA_BYTES + a_op[0]);
#pragma unroll
        for (int j = 0; j < TN2; ++j)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) pb[j][pl] = *reinterpret_cast<const u32x4*>(planes + pl * PB_BYTES + b_op[j]);
#pragma unroll
        for (int i = 0; i < TM2; ++i) {
          // (one block of rows ahead, not all four: 128 accumulator registers leave room for little else)
          if (i + 1 < TM2) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) pa[(i + 1) & 1][pl] = *reinterpret_cast<const u32x4*>(planes + pl * PA_BYTES + a_op[i + 1]);
          }
          const u32x4* A_ = pa[i & 1];
#pragma unroll
          for (int j = 0; j < TN2; ++j) {   // smallest terms first
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(A_[2]), op(pb[j][0]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(A_[0]), op(pb[j][2]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(A_[1]), op(pb[j][1]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(A_[1]), op(pb[j][0]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(A_[0]), op(pb[j][1]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(A_[0]), op(pb[j][0]), acc[i][j], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      {
#pragma unroll
        for (int s = 0; s < NST - 1; ++s)
          if (s < nk && !(SDUMC_GG_DBG & 1)) issue(s);
        int buf = 0, ibuf = NST - 1;
#pragma nounroll
        for (int t = 0; t < nk; ++t) {
          // stage t has landed (this wave's pieces) / everyone is done with the planes of k-tile t - 1 and has converted stage t - 1
          wait_n(min(nk - t - 1, NST - 2));
          __builtin_amdgcn_s_barrier();
          if (t + NST - 1 < nk && !(SDUMC_GG_DBG & 1)) issue(ibuf);
          if (!(SDUMC_GG_DBG & 2)) convert(ring0 + buf * STAGE2);
          __builtin_amdgcn_s_barrier();
          if (!(SDUMC_GG_DBG & 4)) multiply();
          buf = buf + 1 == NST ? 0 : buf + 1;
          ibuf = ibuf + 1 == NST ? 0 : ibuf + 1;
        }
      }
      seg_masked = masked;
      __builtin_amdgcn_s_barrier();          // every wave is done with the planes before the next sub-piece converts into them
    }
    if (seg_masked) {
      const float sc = pr.b_scale;
#pragma unroll
      for (int i = 0; i < TM2; ++i)
#pragma unroll
        for (int j = 0; j < TN2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] *= sc;
    }
    const bool direct = L.nchunk[w.p] == 1 && ka == 0 && kb == g.nk;
    const bool do_cs = pr.colsum_a != nullptr && tile_n == 0;
    if (do_cs) {      // the two k-group halves of a row meet in LDS (all 512 threads hold a partial)
      cs_x[tid] = csum;
      __builtin_amdgcn_s_barrier();
      if (tid < BM) csum = cs_x[tid] + cs_x[tid + BM];
    }
    if (direct) {
      // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
      const bool accum = pr.accumulate != 0;
#pragma unroll
      for (int i = 0; i < TM2; ++i)
#pragma unroll
        for (int j = 0; j < TN2; ++j) {
          const int col = n0 + wn0 + 32 * j + li;
          if (col >= pr.N) continue;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (row >= pr.M) continue;
            float* dst = pr.C + ((uint32_t)row * (uint32_t)pr.ldc + (uint32_t)col);
            float v = acc[i][j][e];
            if (accum) v += *dst;
            *dst = v;
          }
        }
      if (do_cs && tid < BM && m0 + tid < pr.M) {
        float* dst = pr.colsum_a + m0 + tid;
        *dst = accum ? *dst + csum : csum;
      }
    } else {
      // slab slot in register order: [wave][i][j][e / 4][lane][4]
      float* slot = L.slab + (size_t)(wg + w.unit) * SLOT2_FLOATS;
      float* mine = slot + (uint32_t)(wave * 32 * 64 + lane) * 4u;
#pragma unroll
      for (int i = 0; i < TM2; ++i)
#pragma unroll
        for (int j = 0; j < TN2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *reinterpret_cast<f32x4*>(mine + ((i * TN2 + j) * 4 + q) * 256) = v;
          }
      if (do_cs && tid < BM) slot[BM * BN2 + tid] = csum;
    }
    x = px_end;
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// The same launch on bf16 STORAGE (the engine's bf16 mode, BASELINE configs[2] / [4]): dz / dx and the (masked) frames / features
// are bf16 tensors, products accumulate in fp32 on v_mfma_f32_32x32x16_bf16, C and the slabs are fp32.  Tile 256 x 128 x 64 k
// (48 KiB per stage, 3 stages).  Operands are row-contiguous ([k][row] tiles); an MFMA operand -- 8 consecutive k of one row --
// comes out of two transposing reads (ds_read_b64_tr_b16: a 16-lane group reads a 4 k x 16 row block).  The 16-byte chunks of
// a k-row are XOR-swizzled by 4 (k & 3) (applied to the DMA's source address and again at the read): the four k-rows of a
// transposing read are a multiple of 256 bytes apart and would otherwise share their banks.
// At bf16 MFMA rates this kernel is bound by the operand stream (48 KiB per 1024 matrix-pipe cycles and CU), not by the
// matrix cores; what the grouping buys is the same as in fp32 -- every CU streams for the same time, one launch, one reduce.
// ------------------------------------------------------------------------------------------------------------------------
namespace hf {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int HBK = 64, HNST = 3;
constexpr int HA_BYTES = HBK * BM * 2, HB_BYTES = HBK * BN * 2, HSTAGE = HA_BYTES + HB_BYTES;
constexpr int HA_CH = HA_BYTES / 1024, HB_CH = HB_BYTES / 1024, HNI = (HA_CH + HB_CH) / NW;     // 32 + 16 pieces, 6 per wave
static_assert(HNI == 6 && HA_CH % NW == 0, "pieces per wave");
}  // namespace hf

__global__ __launch_bounds__(NTHR, 2) void gg_tn_bf16_kernel(const Launch L) {
#if defined(__HIP_DEVICE_COMPILE__)
  using namespace hf;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int LL = L.line0[L.n];
  const int wg = wg_logical(L, (int)blockIdx.x);
  int x = range_begin(wg, LL, L.nwg);
  const int x_end = range_begin(wg + 1, LL, L.nwg);
  constexpr int PER = HNI;

  f32x16 acc[TM][TN];
  float csum[TM];

  // one MFMA operand: rows rowbase + li, k = 16 s + 8 lh .. + 7 of a [k][BR] tile
  auto frag = [&](const char* tile, int rowbase, int s, int BR) -> bf16x8 {
    const int g16 = (lane >> 4) & 1, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int k0 = 16 * s + 8 * lh + q;                    // this lane's k row of the first read (the second: k0 + 4, same k & 3)
    const int col = rowbase + 16 * g16 + 4 * p;            // first of the 4 columns this lane addresses
    const int chunk = (col >> 3) ^ ((k0 & 3) << 2);
    const char* a0 = tile + (k0 * BR + chunk * 8 + (col & 7)) * 2;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * BR * 2));
    s16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return *reinterpret_cast<bf16x8*>(&v);
  };

  while (x < x_end) {
    const Where w = locate(L, x);
    const sdumc_gg_problem& pr = L.p[w.p];
    const Geo g = geo_of(pr, L.bk, L.bn);
    const int px_end = min(x_end, w.unit_end);
    const int ka = w.kt, kb = w.kt_end - (w.unit_end - px_end);
    if (kb <= ka) {
      x = px_end;
      continue;
    }
    const int tile_m = w.tile / g.ntn, tile_n = w.tile - tile_m * g.ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const bool do_cs = pr.colsum_a != nullptr && tile_n == 0 && wn0 == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      csum[i] = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
#pragma nounroll
    for (int seg = 0; seg < 2; ++seg) {
      int t0, nk;
      if (seg == 0) {
        if (ka >= g.nk0) continue;
        t0 = ka;
        nk = min(kb, g.nk0) - ka;
      } else {
        if (kb <= g.nk0) continue;
        t0 = max(ka, g.nk0) - g.nk0;
        nk = kb - g.nk0 - t0;
      }
      const int kbeg = t0 * HBK;
      const int segK = pr.K[seg], seg_mod = pr.b_row_mod[seg];
      const uint32_t lda2 = (uint32_t)pr.lda * 2u, ldb2 = (uint32_t)pr.ldb * 2u;
      // (rows at and beyond K lie outside the descriptors' ranges: zeros)
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pr.A[seg]), 0, (int)((uint32_t)segK * lda2), 0x00020000);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pr.B[seg]), 0, (int)((uint32_t)(seg_mod > 0 ? seg_mod : segK) * ldb2), 0x00020000);
      // pieces of this wave: A pieces wave + 8 i (i < 4): k-rows 2 piece, 2 piece + 1 of the 64 (a k-row of 256 bf16 = 32 chunks);
      // B pieces wave + 8 i (i < 2): k-rows 4 piece .. 4 piece + 3 (a k-row of 128 bf16 = 16 chunks).  Lane l of a piece writes LDS
      // chunk position (piece * 64 + l); the column chunk it FETCHES is that position's chunk index ^ 4 (k & 3).
      uint32_t voff[HNI];
      int srck = 0;      // B rows with a modulo: source row of B piece 0 of this wave's lane (pieces advance by a fixed 32 rows)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = ((wave + 8 * i) << 6) + lane, krow = q >> 5, cpos = q & 31;
        const int c = cpos ^ ((krow & 3) << 2);
        voff[i] = (uint32_t)(kbeg + krow) * lda2 + (uint32_t)min(m0 + 8 * c, pr.M - 8) * 2u;
      }
      int bk_row[2];
      // b_map: B is a resident store's packed bf16 tensor (any size) and k-row r of the problem is ITS row b_map[r].  A B piece holds
      // four CONSECUTIVE k-rows (16 lanes each), so a piece's four map entries are one scalar x4 load, fetched a stage ahead; the lane
      // picks its row's entry and fetches by 64-bit address (global_load_lds).  Rows at and beyond K: any readable row -- the
      // matching rows of A are zeros (outside A's descriptor), so they add nothing.
      const bool mapped = pr.b_map[seg] != nullptr;
      const int jsel = lane >> 4;                                                     // which of its piece's four k-rows this lane fetches
      const uint32_t colb = (uint32_t)min(n0 + 8 * ((lane & 15) ^ ((jsel & 3) << 2)), pr.N - 8) * 2u;      // (the same for both pieces: 4 | piece's first k-row)
      const int kmap_last = max((segK - 1) & ~3, 0);                                  // first k-row of the last group of four that has a valid row
      int kpiece[2] = {kbeg + 4 * wave, kbeg + 4 * (wave + 8)};                       // first k-row of this wave's two pieces at the next issue
      i32x4 mnext[2];
      auto map4 = [&](int k) -> i32x4 {
        return *(const_i32x4_t*)(uintptr_t)(pr.b_map[seg] + min(k, kmap_last));
      };
      if (mapped) { mnext[0] = map4(kpiece[0]); mnext[1] = map4(kpiece[1]); }
      else { mnext[0] = i32x4{0, 0, 0, 0}; mnext[1] = mnext[0]; }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = ((wave + 8 * i) << 6) + lane, krow = q >> 4, cpos = q & 15;
        const int c = cpos ^ ((krow & 3) << 2);
        int kr = kbeg + krow;
        if (seg_mod > 0) kr %= seg_mod;
        bk_row[i] = kr;
        voff[4 + i] = (uint32_t)kr * ldb2 + (uint32_t)min(n0 + 8 * c, pr.N - 8) * 2u;
      }
      (void)srck;
      auto issue = [&](int buf) {
        char* base = lds + buf * HSTAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(base + (wave + 8 * i) * 1024), 16, voff[i], 0, 0, 0);
          voff[i] += (uint32_t)HBK * lda2;
        }
        if (mapped) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const i32x4 m = mnext[i];
            const int row = jsel == 0 ? m[0] : (jsel == 1 ? m[1] : (jsel == 2 ? m[2] : m[3]));
            const char* src = reinterpret_cast<const char*>(pr.B[seg]) + (size_t)(uint32_t)row * (size_t)ldb2 + colb;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(base + HA_BYTES + (wave + 8 * i) * 1024), 16, 0, 0);
            kpiece[i] += HBK;
            mnext[i] = map4(kpiece[i]);      // (the next stage's rows: back long before the next issue)
          }
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t*)(base + HA_BYTES + (wave + 8 * i) * 1024), 16, voff[4 + i], 0, 0, 0);
            voff[4 + i] += (uint32_t)HBK * ldb2;
            if (seg_mod > 0) {
              bk_row[i] += HBK;
              while (bk_row[i] >= seg_mod) { bk_row[i] -= seg_mod; voff[4 + i] -= (uint32_t)seg_mod * ldb2; }
            }
          }
        }
      };
      auto compute = [&](const char* base) {
#pragma unroll
        for (int s = 0; s < HBK / 16; ++s) {
          bf16x8 af[TM], bfr[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) af[i] = frag(base, wm0 + 32 * i, s, BM);
#pragma unroll
          for (int j = 0; j < TN; ++j) bfr[j] = frag(base + HA_BYTES, wn0 + 32 * j, s, BN);
          if (do_cs) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int e = 0; e < 8; ++e) csum[i] += (float)af[i][e];
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
      };
#pragma unroll
      for (int s = 0; s < HNST - 1; ++s)
        if (s < nk) issue(s);
      int buf = 0, ibuf = HNST - 1;
#pragma nounroll
      for (int t = 0; t < nk; ++t) {
        if (t + HNST - 2 < nk) __builtin_amdgcn_s_waitcnt(waitcnt_vm((HNST - 2) * PER));
        else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
        __builtin_amdgcn_s_barrier();
        if (t + HNST - 1 < nk) issue(ibuf);
        compute(lds + buf * HSTAGE);
        buf = buf + 1 == HNST ? 0 : buf + 1;
        ibuf = ibuf + 1 == HNST ? 0 : ibuf + 1;
      }
      __builtin_amdgcn_s_barrier();
    }
    const bool direct = L.nchunk[w.p] == 1 && ka == 0 && kb == g.nk;
    if (direct) {
      const bool accum = pr.accumulate != 0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = n0 + wn0 + 32 * j + li;
          if (col >= pr.N) continue;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (row >= pr.M) continue;
            float* dst = pr.C + ((uint32_t)row * (uint32_t)pr.ldc + (uint32_t)col);
            float v = acc[i][j][e];
            if (accum) v += *dst;
            *dst = v;
          }
        }
      if (do_cs) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float v = csum[i] + __shfl_xor(csum[i], 32, 64);
          const int m = m0 + wm0 + 32 * i + li;
          if (lh == 0 && m < pr.M) {
            float* dst = pr.colsum_a + m;
            *dst = accum ? *dst + v : v;
          }
        }
      }
    } else {
      float* slot = L.slab + (size_t)(wg + w.unit) * SLOT_FLOATS;
      float* mine = slot + (uint32_t)(wave * 16 * 64 + lane) * 4u;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *reinterpret_cast<f32x4*>(mine + ((i * TN + j) * 4 + q) * 256) = v;
          }
      if (do_cs) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float v = csum[i] + __shfl_xor(csum[i], 32, 64);
          if (lh == 0) slot[BM * BN + wm0 + 32 * i + li] = v;
        }
      }
    }
    x = px_end;
  }
#endif
}

// One workgroup per (tile, part): the float4 chunks a lane holds of its wave's block (16 for the 64 x 64 blocks of the 256 x 128
// tiles, 32 for the 128 x 64 blocks of gg_tn_split2_kernel's 256 x 256 tiles), last part = the column sums of A.  Sums the tile's
// slab slots in ascending k order and writes C (the GEMM kernel's epilogue arithmetic).
__global__ __launch_bounds__(NTHR) void gg_reduce_kernel(const Launch L, const int tiles_total) {
  const bool wide = L.bn == s2::BN2;
  const int nparts = wide ? 32 : 16, slot_floats = BM * L.bn + BM;
  const int wgn = wide ? s2::WGN2 : WGN, wmx = wide ? s2::WM2 : WM, wnx = wide ? s2::WN2 : WN, tnx = wide ? s2::TN2 : TN;
  const int part = blockIdx.y;
  int tg = blockIdx.x;
  int p = 0;
  Geo g = geo_of(L.p[0], L.bk, L.bn);
  while (p + 1 < L.n && tg >= g.ntiles) {
    tg -= g.ntiles;
    ++p;
    g = geo_of(L.p[p], L.bk, L.bn);
  }
  const sdumc_gg_problem& pr = L.p[p];
  const int nc = L.nchunk[p];
  const int LL = L.line0[L.n];
  const int tile_m = tg / g.ntn, tile_n = tg - tile_m * g.ntn;
  const int m0 = tile_m * BM, n0 = tile_n * L.bn;
  const int ovh = L.ovh;
  if (nc == 1) {   // written directly by the one workgroup that held all its k-tiles?
    const int us = L.line0[p] + tg * (g.nk + ovh);
    if (wg_of(us + ovh, LL, L.nwg) == wg_of(us + ovh + g.nk - 1, LL, L.nwg)) return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (part == nparts) {
    if (!pr.colsum_a || tile_n != 0 || tid >= BM) return;
    float s = 0.f;
    for (int c = 0; c < nc; ++c) {
      const int k0 = chunk_k(c, g.nk, nc), len = chunk_k(c + 1, g.nk, nc) - k0 + ovh;
      const int us = L.line0[p] + g.ntiles * (k0 + c * ovh) + tg * len;
      const int unit = L.unit0[p] + c * g.ntiles + tg;
      const int wl = wg_of(us + len - 1, LL, L.nwg);
      for (int w = wg_of(us + ovh, LL, L.nwg); w <= wl; ++w) s += L.slab[(size_t)(w + unit) * slot_floats + BM * L.bn + tid];
    }
    const int m = m0 + tid;
    if (m < pr.M) pr.colsum_a[m] = pr.accumulate ? pr.colsum_a[m] + s : s;
    return;
  }
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const size_t off = ((size_t)((wave * nparts + part) * 64 + lane)) * 4;
  {
    // The tile's slots in ascending k order = (chunk c, workgroup w) in ascending (c, w): the walk below is wave-uniform (scalar unit);
    // eight slots' loads are requested before the first is added (one load per add, as first written, left a wave with 1 KB in flight
    // and the launch at 2.8 TB/s on slabs the persistent kernel had just written: latency, not bandwidth).  Same sums, same order.
    int c = 0, w = 0, wl = -1, unit = 0;
    auto next = [&]() -> const float* {
      while (w > wl) {
        if (c == nc) return nullptr;
        const int k0 = chunk_k(c, g.nk, nc), len = chunk_k(c + 1, g.nk, nc) - k0 + ovh;
        const int us = L.line0[p] + g.ntiles * (k0 + c * ovh) + tg * len;
        unit = L.unit0[p] + c * g.ntiles + tg;
        w = wg_of(us + ovh, LL, L.nwg);
        wl = wg_of(us + len - 1, LL, L.nwg);
        ++c;
      }
      const float* q = L.slab + (size_t)(w + unit) * slot_floats + off;
      ++w;
      return q;
    };
    constexpr int NB = 8;
    for (;;) {
      const float* src[NB];
      int n = 0;
      for (; n < NB; ++n) {
        src[n] = next();
        if (!src[n]) break;
      }
      f32x4 v[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j)
        if (j < n) v[j] = *reinterpret_cast<const f32x4*>(src[j]);
#pragma unroll
      for (int j = 0; j < NB; ++j)
        if (j < n) s += v[j];
      if (n < NB) break;
    }
  }
  const int li = lane & 31, lh = lane >> 5;
  const int ij = part >> 2, q = part & 3, i = ij / tnx, j = ij - i * tnx;
  const bool nat = L.hf || L.nat;
  const int col = n0 + (wave % wgn) * wnx + (nat ? 32 * j + li : 2 * li + j);
  if (col >= pr.N) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int rr = e + 8 * q + 4 * lh;
    const int row = m0 + (wave / wgn) * wmx + (nat ? 32 * i + rr : 2 * rr + i);
    if (row >= pr.M) continue;
    float* dst = pr.C + (size_t)row * pr.ldc + col;
    *dst = pr.accumulate ? *dst + s[e] : s[e];
  }
}

int cu_count() {
  static std::mutex mu;
  static int per_device[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  std::lock_guard<std::mutex> lock(mu);
  if (!per_device[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    per_device[dev] = n;
  }
  return per_device[dev];
}

bool set_lds_attr() {   // the dynamic-LDS limit is a per-device function attribute
  static std::mutex mu;
  static bool done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lock(mu);
  if (!done[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gg_tn_kernel<5, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * STAGE_BYTES) != hipSuccess)
      return false;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gg_tn_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, hf::HNST * hf::HSTAGE) != hipSuccess)
      return false;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gg_tn_split2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, s2::LDS2) != hipSuccess)
      return false;
    done[dev] = true;
  }
  return true;
}

bool valid(const sdumc_gg_problem& p) {
  if (!p.A[0] || !p.B[0] || !p.C || p.M < 4 || p.N < 4 || (p.M & 3) || (p.N & 3) || (p.lda & 3) || (p.ldb & 3) || p.K[0] <= 0 || p.K[1] < 0)
    return false;
  if (p.lda < p.M || p.ldb < p.N || p.ldc < p.N || p.M > BM * 255) return false;
  if (p.K[1] > 0 && ((p.b_bits[0] != nullptr) != (p.b_bits[1] != nullptr))) return false;
  for (int s = 0; s < 2; ++s) {
    if (p.K[s] == 0) continue;
    if (!p.A[s] || !p.B[s]) return false;
    if ((reinterpret_cast<uintptr_t>(p.A[s]) | reinterpret_cast<uintptr_t>(p.B[s])) & 15) return false;
    if (p.b_bits[s] && ((reinterpret_cast<uintptr_t>(p.b_bits[s]) & 3) || (p.bits_qw & 3) || p.bits_qw * 4 < p.N)) return false;
    if (p.b_row_mod[s] < 0 || (p.b_row_mod[s] > 0 && p.b_row_mod[s] < BK)) return false;
    if (p.b_map[s] && (p.b_row_mod[s] || (reinterpret_cast<uintptr_t>(p.b_map[s]) & 3))) return false;
    const size_t brows = p.b_map[s] ? 1 : p.b_row_mod[s] > 0 ? (size_t)p.b_row_mod[s] : (size_t)p.K[s];      // (mapped: a descriptor per row)
    // (32-bit byte offsets that run up to a ring of k-tiles past the last row; descriptor ranges below 2 GiB)
    const size_t pad = 8 * BK;
    if (((size_t)p.K[s] + pad) * p.lda * 4 >= 0x7FFFFFF0u || (brows + pad) * p.ldb * 4 >= 0x7FFFFFF0u) return false;
    if (p.b_bits[s] && ((size_t)p.K[s] + pad) * p.bits_qw >= 0x7FFFFFF0u) return false;
  }
  return true;
}

// fills L for problems [first, first + n); returns the number of tiles
int plan(const sdumc_gg_problem* probs, int n, int nwg_max, bool hf, bool wide, Launch& L, int& units) {
  memset(&L, 0, sizeof(L));
  L.n = n;
  L.hf = hf ? 1 : 0;
  L.bk = hf ? 64 : BK;
  L.bn = wide ? s2::BN2 : BN;
  L.nat = wide ? 1 : 0;
  // (measured on MI355X, tools/gg_bench.py, all 37 problems of a C2 backward in one launch: 623 us with 0, 515 with 2, 490 with 3-4,
  //  494-500 with 6-12: a unit's ring start-up -- the HBM latency of its first stage -- and its epilogue are worth ~3 k-tiles)
  L.ovh = hf ? 5 : 3;      // (bf16: 2, 5, 8, 12 measured 110 / 111 / 115 / 118 us on the frame-level problems of a C2 backward)
  // (the 256 x 256 form: 1, 2, 3, 5, 8 measured 400 / 364 / 339 / 323 / 325 us on all problems of a C2 backward in one launch; no
  //  difference on the step, which issues them as two launches: 1.456-1.469 ms at 2, 3 and 5)
  if (wide) L.ovh = 4;
  int line = 0, tiles = 0;
  units = 0;
  for (int i = 0; i < n; ++i) {
    L.p[i] = probs[i];
    const Geo g = geo_of(probs[i], L.bk, L.bn);
    // (a second problem form in this kernel -- the input gradients dxd += dz . W of the input_proj layers, A k-contiguous, M the
    //  long dimension, K = 256 -- was built and measured in round 3: 237 us for the five sites of a C2 backward against 210 us
    //  for the per-layer 64x64 NN kernel: units of 16 k-tiles pay a ring start-up and a 256 KB read-modify-write epilogue each;
    //  removed again)
    L.nchunk[i] = 1;
    L.line0[i] = line;
    L.unit0[i] = units;
    line += g.ntiles * (g.nk + L.nchunk[i] * L.ovh);
    units += g.ntiles * L.nchunk[i];
    tiles += g.ntiles;
  }
  // (Measured and dropped again, round 6 -- round 3 had tried it on the fp32-MFMA form: K cut into chunks of 1 / 2 / 4 workgroup
  //  shares, line order (chunk, tile, k), ranges handed out XCD-major (wg_logical) so that the n-tiles that re-read the same rows of A
  //  run side by side under one L2: frame launch 195 / 185 / 180 us against 176-177 us, one audio problem 131 / 112 / 104 against 97.
  //  The in-kernel ablations say why traffic is not the bound: DMA alone 70 us, MFMAs + fragment reads alone 106 us, conversion
  //  +33 us and ring stalls +23 us on top, nearly additive -- profiles/README.md, round 6.)
  L.xcd = 0;
  L.line0[n] = line;
  L.unit0[n] = units;
  L.nwg = std::min(nwg_max, line);
  return tiles;
}

}  // namespace sdumc_gg
using namespace sdumc_gg;

// engine.hip: slab bytes that cover any problem list with at most `tiles` output tiles (256 x 128) on the current device; also
// the place where the kernels' per-device attributes are set outside any stream capture
bool valid_bf16(const sdumc_gg_problem& p) {
  if (!p.A[0] || !p.B[0] || !p.C || p.M < 8 || p.N < 8 || (p.M & 7) || (p.N & 7) || (p.lda & 7) || (p.ldb & 7) || p.K[0] <= 0 || p.K[1] < 0)
    return false;
  if (p.lda < p.M || p.ldb < p.N || p.ldc < p.N || p.M > BM * 255) return false;
  for (int s = 0; s < 2; ++s) {
    if (p.K[s] == 0) continue;
    if (!p.A[s] || !p.B[s] || p.b_bits[s]) return false;      // no fused dropout on bf16 storage: the engine materialises the masked frames
    if (p.b_map[s] && (p.b_row_mod[s] || (reinterpret_cast<uintptr_t>(p.b_map[s]) & 15))) return false;      // (x4 scalar loads of the map)
    if ((reinterpret_cast<uintptr_t>(p.A[s]) | reinterpret_cast<uintptr_t>(p.B[s])) & 15) return false;
    if (p.b_row_mod[s] < 0 || (p.b_row_mod[s] > 0 && p.b_row_mod[s] < 64)) return false;
    const size_t brows = p.b_map[s] ? 1 : p.b_row_mod[s] > 0 ? (size_t)p.b_row_mod[s] : (size_t)p.K[s];      // (mapped: 64-bit addresses)
    const size_t pad = 4 * 64;
    if (((size_t)p.K[s] + pad) * p.lda * 2 >= 0x7FFFFFF0u || (brows + pad) * p.ldb * 2 >= 0x7FFFFFF0u) return false;
  }
  return true;
}

extern "C" size_t sdumc_gg_slab_bytes_(int tiles) {
  (void)set_lds_attr();
  // (`tiles` counts 256 x 128 tiles; the 256 x 256 form has at most as many, of twice the size)
  return (size_t)(cu_count() + tiles) * s2::SLOT2_FLOATS * sizeof(float);
}

extern "C" int sdumc_prof_begin_(int variant, double flops, void* stream);     // gemm_f32.hip: bench.py's per-launch HIP events
extern "C" void sdumc_prof_end_(int token, void* stream);

namespace {
bool split_products() { return sdumc_split_on_(SDUMC_SPLIT_GROUP) != 0; }
size_t gg_workspace_bytes(const sdumc_gg_problem* probs, int32_t n, bool hf) {
  if (!probs || n <= 0) return 0;
  const int nwg = cu_count();
  size_t need = 0;
  for (int first = 0; first < n; first += MAXP) {
    const int cnt = std::min(MAXP, n - first);
    Launch L;
    int units = 0;
    plan(probs + first, cnt, nwg, hf, false, L, units);
    need = std::max(need, (size_t)(L.nwg + units) * SLOT_FLOATS * sizeof(float));
    if (!hf) {      // whichever form the process-wide switch selects at launch time fits
      plan(probs + first, cnt, nwg, hf, true, L, units);
      need = std::max(need, (size_t)(L.nwg + units) * s2::SLOT2_FLOATS * sizeof(float));
    }
  }
  return need;
}

int gg_run(const sdumc_gg_problem* probs, int32_t n, void* workspace, size_t workspace_bytes, bool hf, void* stream) {
  if (!probs || n <= 0 || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 15)) return SDUMC_EINVAL;
  for (int i = 0; i < n; ++i)
    if (!(hf ? valid_bf16(probs[i]) : valid(probs[i]))) return SDUMC_EINVAL;
  if (workspace_bytes < gg_workspace_bytes(probs, n, hf)) return SDUMC_ENOMEM;
  if (!set_lds_attr()) return SDUMC_ELAUNCH;
  const int nwg = cu_count();      // one workgroup per CU (two on a 3-stage ring spilled registers and ran at 2/3 of the rate)
  hipStream_t st = as_stream(stream);
  for (int first = 0; first < n; first += MAXP) {
    const int cnt = std::min(MAXP, n - first);
    Launch L;
    int units = 0;
    const bool wide = !hf && split_products();
    if (!wide && !hf)      // row maps are read by gg_tn_split2_kernel and gg_tn_bf16_kernel
      for (int i = 0; i < cnt; ++i)
        if (probs[first + i].b_map[0] || probs[first + i].b_map[1]) return SDUMC_EINVAL;
    const int tiles = plan(probs + first, cnt, nwg, hf, wide, L, units);
    if ((long long)L.line0[cnt] * (L.nwg + 1) >= (1LL << 31)) return SDUMC_EINVAL;   // 32-bit index arithmetic in the kernels
    L.slab = static_cast<float*>(workspace);
    double flops = 0.0;
    for (int i = 0; i < cnt; ++i) flops += 2.0 * probs[first + i].M * (double)probs[first + i].N * ((double)probs[first + i].K[0] + probs[first + i].K[1]);
    const int tok = sdumc_prof_begin_(hf ? 20 : (split_products() ? 24 : 19), flops, stream);
    if (hf) hipLaunchKernelGGL(gg_tn_bf16_kernel, dim3(L.nwg), dim3(NTHR), hf::HNST * hf::HSTAGE, st, L);
    else if (wide) {
      // (round 6: a form that converts k-tile t + 1 while it multiplies k-tile t -- double-buffered planes, 8-k raw halves, the two
      //  waves of a SIMD in opposite convert / multiply order -- measured 178-180 us against this kernel's 170 us on the frame launch,
      //  its ablations as additive as this one's: tools/experiments/gg_tn_split3.inc, profiles/r6_gg_split3.txt)
      // (a third form, the raw fp32 ring replaced by global -> register loads one k-tile ahead and double-buffered planes with ONE
      //  barrier per k-tile: 185-187 us against 185-186 on the same box; alone its multiply loop takes 88 us (this kernel's: 106) and its
      //  loads + conversion 80, together 161: tools/experiments/gg_tn_split4.inc, profiles/r6_gg_split4.txt)
      hipLaunchKernelGGL(gg_tn_split2_kernel, dim3(L.nwg), dim3(NTHR), s2::LDS2, st, L);
    } else hipLaunchKernelGGL((gg_tn_kernel<5, 2>), dim3(L.nwg), dim3(NTHR), 5 * STAGE_BYTES, st, L);
    sdumc_prof_end_(tok, stream);
    SDUMC_CHECK_LAUNCH();
    hipLaunchKernelGGL(gg_reduce_kernel, dim3(tiles, wide ? 33 : 17), dim3(NTHR), 0, st, L, tiles);
    SDUMC_CHECK_LAUNCH();
  }
  return SDUMC_OK;
}
}  // namespace

extern "C" size_t sdumc_gemm_group_workspace_bytes(const sdumc_gg_problem* probs, int32_t n) { return gg_workspace_bytes(probs, n, false); }
extern "C" int sdumc_gemm_group_tn(const sdumc_gg_problem* probs, int32_t n, void* workspace, size_t workspace_bytes, void* stream) {
  return gg_run(probs, n, workspace, workspace_bytes, false, stream);
}
// A, B: bf16 tensors (lda / ldb in elements; M, N, lda, ldb multiples of 8; no fused dropout), C and colsum_a fp32
extern "C" size_t sdumc_gemm_group_bf16_workspace_bytes(const sdumc_gg_problem* probs, int32_t n) { return gg_workspace_bytes(probs, n, true); }
extern "C" int sdumc_gemm_group_tn_bf16(const sdumc_gg_problem* probs, int32_t n, void* workspace, size_t workspace_bytes, void* stream) {
  return gg_run(probs, n, workspace, workspace_bytes, true, stream);
}
// ---- which fp32 GEMM kernels compute their products on the bf16 matrix pipe (SDUMC_SPLIT_* bits; environment SDUMC_SPLIT) ----
namespace {
std::atomic<int> g_split_mask{-1};
// the mask of the network-level call this host thread is inside (sdumc_ctx_set_option(SDUMC_OPT_SPLIT): engine.hip opens a scope
// around forward / backward / step), -1 outside: every launcher of such a call then sees the CONTEXT's arithmetic, two contexts
// can hold different ones, and nothing process-wide changes
thread_local int tl_split_scope = -1;
}
extern "C" int sdumc_get_split_(void) {
  if (tl_split_scope >= 0) return tl_split_scope;
  int v = g_split_mask.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("SDUMC_SPLIT");
    v = e ? (atoi(e) & SDUMC_SPLIT_ALL) : SDUMC_SPLIT_ALL;
    g_split_mask.store(v, std::memory_order_relaxed);
  }
  return v;
}
extern "C" int sdumc_split_on_(int bit) { return (sdumc_get_split_() & bit) != 0; }
extern "C" void sdumc_set_split_(int mask) { g_split_mask.store(mask & SDUMC_SPLIT_ALL, std::memory_order_relaxed); }
extern "C" int sdumc_split_scope_(int mask) {      // returns the previous scope value (restore with it)
  const int prev = tl_split_scope;
  tl_split_scope = mask < 0 ? -1 : (mask & SDUMC_SPLIT_ALL);
  return prev;
}

// One empty kernel per source file = per gfx950 code object: sdumc_preload_() asks for its attributes, which makes the HIP runtime load
// this file's code object NOW (outside any timed or latency-sensitive region) instead of at the first launch of one of its kernels --
// with deferred loading that first launch stalls the host for tens of milliseconds (seen as a 36-59 ms gap in the middle of an epoch,
// at the first batch whose shape took a fallback path: profiles/README.md, round 6).
__global__ void sdumc_preload_gemm_group_kernel() {}
extern "C" int sdumc_preload_gemm_group_(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&sdumc_preload_gemm_group_kernel)) == hipSuccess ? SDUMC_OK : SDUMC_ELAUNCH;
}
