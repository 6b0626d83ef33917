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
#include <cstring>
#include <mutex>
#include <type_traits>

#include "common.h"

namespace sdumc_gg {

typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int waitcnt_vm(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0xF << 8); }

constexpr int BM = 256, BN = 128, BK = 16;
constexpr int NW = 8, NTHR = 64 * NW, WGN = 2;       // 4 x 2 waves, each 64 x 64
constexpr int WM = 64, WN = 64, TM = 2, TN = 2;
constexpr int A_BYTES = BK * BM * 4, B_BYTES = BK * BN * 4, BITS_BYTES = BK * (BN / 4);
constexpr int A_CH = A_BYTES / 1024, B_CH = B_BYTES / 1024;   // 1-KiB pieces = one wave-instruction each
constexpr int NI = (A_CH + B_CH) / NW;                         // data pieces per wave per stage
constexpr int BITS_CH = BITS_BYTES / 256;
constexpr int STAGE_BYTES = A_BYTES + B_BYTES + BITS_BYTES;
constexpr int NST = 4;
constexpr int LDS_BYTES = NST * STAGE_BYTES;
constexpr int SLOT_FLOATS = BM * BN + BM;                      // a partial tile in register order + its column sums of A
static_assert((A_CH + B_CH) % NW == 0 && A_CH % NW == 0, "pieces divide evenly over the waves");
static_assert(NI == 3, "issue() below is written for 2 A rows + 1 B piece per wave");

constexpr int MAXP = SDUMC_GG_MAX_PROBLEMS;

struct Launch {
  sdumc_gg_problem p[MAXP];
  int32_t line0[MAXP + 1];   // first line position (k-tiles) of problem i; line0[n] = L
  int32_t unit0[MAXP + 1];   // first global unit index of problem i
  int32_t nchunk[MAXP];      // K of a problem's tiles cut into this many chunks, line order (chunk, tile, k): tiles of one chunk
                             // (which read the same rows of A) are neighbours on the line.  1 = plain (tile, k) order
  int32_t n;
  int32_t nwg;
  float* slab;
};

struct Geo {       // derived shape of one problem
  int ntn, ntiles, nk0, nk;
};
__host__ __device__ inline Geo geo_of(const sdumc_gg_problem& p) {
  Geo g;
  g.ntn = (p.N + BN - 1) / BN;
  g.ntiles = g.ntn * ((p.M + BM - 1) / BM);
  g.nk0 = (p.K[0] + BK - 1) / BK;
  g.nk = g.nk0 + (p.K[1] > 0 ? (p.K[1] + BK - 1) / BK : 0);
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
};
__device__ __forceinline__ Where locate(const Launch& L, int x) {
  Where w;
  int p = 0;
  while (p + 1 < L.n && L.line0[p + 1] <= x) ++p;
  const Geo g = geo_of(L.p[p]);
  const int nc = L.nchunk[p];
  const int xr = x - L.line0[p];
  int c = nc == 1 ? 0 : (int)(((uint32_t)xr * (uint32_t)nc) / ((uint32_t)g.ntiles * (uint32_t)g.nk));
  if (c > nc - 1) c = nc - 1;
  while (c + 1 < nc && g.ntiles * chunk_k(c + 1, g.nk, nc) <= xr) ++c;
  while (c > 0 && g.ntiles * chunk_k(c, g.nk, nc) > xr) --c;
  const int k0 = chunk_k(c, g.nk, nc), k1 = chunk_k(c + 1, g.nk, nc), len = k1 - k0;
  const int y = xr - g.ntiles * k0;
  const int t = y / len, ko = y - t * len;
  w.p = p;
  w.chunk = c;
  w.tile = t;
  w.kt = k0 + ko;
  w.kt_end = k1;
  w.unit_end = x - ko + len;
  w.unit = L.unit0[p] + c * g.ntiles + t;
  return w;
}

__global__ __launch_bounds__(NTHR, 2) void gg_tn_kernel(const Launch L) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int LL = L.line0[L.n];
  const int wg = blockIdx.x;
  int x = range_begin(wg, LL, L.nwg);
  const int x_end = range_begin(wg + 1, LL, L.nwg);

  f32x16 acc[TM][TN];
  float csum[TM];

  // ---- state of the current K-segment (set by seg_begin) ----
  __amdgpu_buffer_rsrc_t ra, rb, rbits;
  uint32_t voff[NI], bvoff = 0;
  int srck[NI];
  int segK = 0, seg_mod = 0, seg_lda = 0, seg_ldb = 0, seg_qw = 0;
  float mscale = 1.f;

  // sub-piece = rows [kbeg, kend) of one K-segment of problem pr for the tile at (m0, n0)
  auto run_ring = [&](const sdumc_gg_problem& pr, int seg, int m0, int n0, int kbeg, int kend, auto mask_c) {
    constexpr bool MASK = decltype(mask_c)::value;
    constexpr int PER = NI + (MASK ? 1 : 0);
    segK = pr.K[seg];
    seg_mod = pr.b_row_mod[seg];
    seg_lda = pr.lda;
    seg_ldb = pr.ldb;
    seg_qw = pr.bits_qw;
    mscale = pr.b_scale;
    const int b_rows = seg_mod > 0 ? seg_mod : segK;
    ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pr.A[seg]), 0, (int)min((size_t)segK * seg_lda * 4, (size_t)0xFFFFFFF0u), 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pr.B[seg]), 0, (int)min((size_t)b_rows * seg_ldb * 4, (size_t)0xFFFFFFF0u), 0x00020000);
    rbits = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(MASK ? pr.b_bits[seg] : (const uint8_t*)pr.A[seg]), 0,
                                              MASK ? (int)min((size_t)segK * seg_qw, (size_t)0xFFFFFFF0u) : 0, 0x00020000);
    // pieces of this wave: A k-rows `wave` and `wave + 8` (64 lanes x 4 columns = the 256 columns of the tile), B piece `wave`
    // = k-rows 2 wave, 2 wave + 1 (32 lanes x 4 columns each)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = min(m0 + 4 * lane, pr.M - 4);
      srck[i] = min(kbeg + wave + 8 * i, segK - 1);
      voff[i] = ((uint32_t)srck[i] * (uint32_t)seg_lda + (uint32_t)col) * 4u;
    }
    {
      const int col = min(n0 + 4 * (lane & 31), pr.N - 4);
      int kr = min(kbeg + 2 * wave + (lane >> 5), segK - 1);
      if (seg_mod > 0) kr %= seg_mod;
      srck[2] = kr;
      voff[2] = ((uint32_t)kr * (uint32_t)seg_ldb + (uint32_t)col) * 4u;
    }
    if constexpr (MASK) {
      const int idx = ((wave % BITS_CH) << 6) + lane;         // dword index inside the [16][8 dwords] bits tile
      const int krow = idx >> 3, dw = idx & 7;
      bvoff = (uint32_t)min(kbeg + krow, segK - 1) * (uint32_t)seg_qw + (uint32_t)(n0 >> 2) + 4u * dw;
    }
    auto issue = [&](int buf) {
      char* base = lds + buf * STAGE_BYTES;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(base + (wave + 8 * i) * 1024), 16, voff[i], 0, 0, 0);
        const int nxt = min(srck[i] + BK, segK - 1);     // beyond the last row: stay on it (the tail iteration zeroes it)
        voff[i] += (uint32_t)(nxt - srck[i]) * (uint32_t)seg_lda * 4u;
        srck[i] = nxt;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t*)(base + A_BYTES + wave * 1024), 16, voff[2], 0, 0, 0);
      if (seg_mod > 0) {
        int nxt = srck[2] + BK;
        voff[2] += (uint32_t)BK * (uint32_t)seg_ldb * 4u;
        if (nxt >= seg_mod) { nxt -= seg_mod; voff[2] -= (uint32_t)seg_mod * (uint32_t)seg_ldb * 4u; }
        srck[2] = nxt;
      } else {
        const int nxt = min(srck[2] + BK, segK - 1);
        voff[2] += (uint32_t)(nxt - srck[2]) * (uint32_t)seg_ldb * 4u;
        srck[2] = nxt;
      }
      if constexpr (MASK) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rbits, (lds_void_t*)(base + A_BYTES + B_BYTES + (wave % BITS_CH) * 256), 4, bvoff, 0, 0, 0);
        bvoff += (uint32_t)BK * (uint32_t)seg_qw;      // (rows beyond K read in-range bytes or 0: masked out by the tail anyway)
      }
    };
    // fragments of MFMA group gq (8 k): element s of a fragment is k = 8 gq + 4 lh + s
    auto read_a = [&](const char* base, int gq, f32x4 (&af)[TM]) {
      const float* As = reinterpret_cast<const float*>(base);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float* q = As + (8 * gq + 4 * lh) * BM + wm0 + 32 * i + li;
        af[i][0] = q[0];
        af[i][1] = q[BM];
        af[i][2] = q[2 * BM];
        af[i][3] = q[3 * BM];
      }
    };
    auto read_b = [&](const char* base, int gq, f32x4 (&bf)[TN]) {
      const float* Bs = reinterpret_cast<const float*>(base + A_BYTES);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn0 + 32 * j + li;
        const float* q = Bs + (8 * gq + 4 * lh) * BN + row;
        bf[j][0] = q[0];
        bf[j][1] = q[BN];
        bf[j][2] = q[2 * BN];
        bf[j][3] = q[3 * BN];
        if constexpr (MASK) {
          const uint8_t* bt = reinterpret_cast<const uint8_t*>(base + A_BYTES + B_BYTES) + (8 * gq + 4 * lh) * (BN / 4) + (row >> 2);
          const uint32_t bit = 1u << (row & 3);
#pragma unroll
          for (int s = 0; s < 4; ++s) bf[j][s] *= (bt[s * (BN / 4)] & bit) ? mscale : 0.f;
        }
      }
    };
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
#pragma unroll
          for (int i = 0; i < TM; ++i) csum[i] += af[cur][i][s];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][s], bf[cur][j][s], acc[i][j], 0, 0, 0);
        }
      }
    };
    const int nk = (kend - kbeg + BK - 1) / BK;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < nk) issue(s);
    int buf = 0, ibuf = NST - 1;
    for (int t = 0; t < nk; ++t) {
      const int ahead = nk - 1 - t;        // stages issued beyond the one multiplied now (capped by the ring)
      if (ahead >= NST - 2) __builtin_amdgcn_s_waitcnt(waitcnt_vm((NST - 2) * PER));
      else if (ahead == 1 && NST > 3) __builtin_amdgcn_s_waitcnt(waitcnt_vm(PER));
      else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
      __builtin_amdgcn_s_barrier();
      if (t + NST - 1 < nk) issue(ibuf);
      const int k0 = kbeg + t * BK;
      if (k0 + BK > kend) compute(lds + buf * STAGE_BYTES, k0, std::true_type{});
      else compute(lds + buf * STAGE_BYTES, k0, std::false_type{});
      buf = buf + 1 == NST ? 0 : buf + 1;
      ibuf = ibuf + 1 == NST ? 0 : ibuf + 1;
    }
    __builtin_amdgcn_s_barrier();          // every wave is done reading the ring before the next sub-piece refills it
  };

  while (x < x_end) {
    const Where w = locate(L, x);
    const sdumc_gg_problem& pr = L.p[w.p];
    const Geo g = geo_of(pr);
    const int px_end = min(x_end, w.unit_end);
    const int ka = w.kt, kb = w.kt + (px_end - x);          // k-tiles [ka, kb) of the problem's concatenated K
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
    if (ka < g.nk0) {
      const int kbeg = ka * BK, kend = min(min(kb, g.nk0) * BK, pr.K[0]);
      if (pr.b_bits[0]) run_ring(pr, 0, m0, n0, kbeg, kend, std::true_type{});
      else run_ring(pr, 0, m0, n0, kbeg, kend, std::false_type{});
    }
    if (kb > g.nk0) {
      const int kbeg = (max(ka, g.nk0) - g.nk0) * BK, kend = min((kb - g.nk0) * BK, pr.K[1]);
      if (pr.b_bits[1]) run_ring(pr, 1, m0, n0, kbeg, kend, std::true_type{});
      else run_ring(pr, 1, m0, n0, kbeg, kend, std::false_type{});
    }
    const bool direct = L.nchunk[w.p] == 1 && ka == 0 && kb == g.nk;
    const bool do_cs = pr.colsum_a != nullptr && tile_n == 0 && wn0 == 0;
    if (direct) {
      // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
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
            float* dst = pr.C + (size_t)row * pr.ldc + col;
            float v = acc[i][j][e];
            if (pr.accumulate) v += *dst;
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
            *dst = pr.accumulate ? *dst + v : v;
          }
        }
      }
    } else {
      // slab slot in register order: [wave][i][j][e / 4][lane][4] -- every store instruction of a wave is 1 KiB contiguous
      float* slot = L.slab + (size_t)(wg + w.unit) * SLOT_FLOATS;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *reinterpret_cast<f32x4*>(slot + ((size_t)((wave * 16 + (i * TN + j) * 4 + q) * 64 + lane)) * 4) = v;
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

// One workgroup per (tile, part): parts 0..15 = the sixteen float4 chunks a lane holds of its wave's 64 x 64 block, part 16 =
// the column sums of A.  Sums the tile's slab slots in ascending k order and writes C (the GEMM kernel's epilogue arithmetic).
__global__ __launch_bounds__(NTHR) void gg_reduce_kernel(const Launch L, const int tiles_total) {
  const int part = blockIdx.y;
  int tg = blockIdx.x;
  int p = 0;
  Geo g = geo_of(L.p[0]);
  while (p + 1 < L.n && tg >= g.ntiles) {
    tg -= g.ntiles;
    ++p;
    g = geo_of(L.p[p]);
  }
  const sdumc_gg_problem& pr = L.p[p];
  const int nc = L.nchunk[p];
  const int LL = L.line0[L.n];
  const int tile_m = tg / g.ntn, tile_n = tg - tile_m * g.ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (nc == 1) {   // written directly by the one workgroup that held the whole tile?
    const int us = L.line0[p] + tg * g.nk;
    if (wg_of(us, LL, L.nwg) == wg_of(us + g.nk - 1, LL, L.nwg)) return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (part == 16) {
    if (!pr.colsum_a || tile_n != 0 || tid >= BM) return;
    float s = 0.f;
    for (int c = 0; c < nc; ++c) {
      const int k0 = chunk_k(c, g.nk, nc), len = chunk_k(c + 1, g.nk, nc) - k0;
      const int us = L.line0[p] + g.ntiles * k0 + tg * len;
      const int unit = L.unit0[p] + c * g.ntiles + tg;
      const int wl = wg_of(us + len - 1, LL, L.nwg);
      for (int w = wg_of(us, LL, L.nwg); w <= wl; ++w) s += L.slab[(size_t)(w + unit) * SLOT_FLOATS + BM * BN + tid];
    }
    const int m = m0 + tid;
    if (m < pr.M) pr.colsum_a[m] = pr.accumulate ? pr.colsum_a[m] + s : s;
    return;
  }
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const size_t off = ((size_t)((wave * 16 + part) * 64 + lane)) * 4;
  for (int c = 0; c < nc; ++c) {
    const int k0 = chunk_k(c, g.nk, nc), len = chunk_k(c + 1, g.nk, nc) - k0;
    const int us = L.line0[p] + g.ntiles * k0 + tg * len;
    const int unit = L.unit0[p] + c * g.ntiles + tg;
    const int wl = wg_of(us + len - 1, LL, L.nwg);
    for (int w = wg_of(us, LL, L.nwg); w <= wl; ++w) s += *reinterpret_cast<const f32x4*>(L.slab + (size_t)(w + unit) * SLOT_FLOATS + off);
  }
  const int li = lane & 31, lh = lane >> 5;
  const int ij = part >> 2, q = part & 3, i = ij / TN, j = ij - i * TN;
  const int col = n0 + (wave % WGN) * WN + 32 * j + li;
  if (col >= pr.N) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = m0 + (wave / WGN) * WM + 32 * i + e + 8 * q + 4 * lh;
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
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gg_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
      return false;
    done[dev] = true;
  }
  return true;
}

bool valid(const sdumc_gg_problem& p) {
  if (!p.A[0] || !p.B[0] || !p.C || p.M < 4 || p.N < 4 || (p.M & 3) || (p.N & 3) || (p.lda & 3) || (p.ldb & 3) || p.K[0] <= 0 || p.K[1] < 0)
    return false;
  if (p.lda < p.M || p.ldb < p.N || p.ldc < p.N || p.M > BM * 255) return false;
  for (int s = 0; s < 2; ++s) {
    if (p.K[s] == 0) continue;
    if (!p.A[s] || !p.B[s]) return false;
    if ((reinterpret_cast<uintptr_t>(p.A[s]) | reinterpret_cast<uintptr_t>(p.B[s])) & 15) return false;
    if (p.b_bits[s] && ((reinterpret_cast<uintptr_t>(p.b_bits[s]) & 3) || (p.bits_qw & 3) || p.bits_qw * 4 < p.N)) return false;
    if (p.b_row_mod[s] < 0 || (p.b_row_mod[s] > 0 && p.b_row_mod[s] < BK)) return false;
    const size_t brows = p.b_row_mod[s] > 0 ? (size_t)p.b_row_mod[s] : (size_t)p.K[s];
    if ((size_t)p.K[s] * p.lda * 4 >= 0xFFFFFFF0u || brows * p.ldb * 4 >= 0xFFFFFFF0u) return false;
    if (p.b_bits[s] && (size_t)p.K[s] * p.bits_qw >= 0xFFFFFFF0u) return false;
  }
  return true;
}

// fills L for problems [first, first + n); returns the number of tiles
int plan(const sdumc_gg_problem* probs, int n, int nwg_max, Launch& L, int& units) {
  memset(&L, 0, sizeof(L));
  L.n = n;
  int line = 0, tiles = 0;
  units = 0;
  for (int i = 0; i < n; ++i) {
    L.p[i] = probs[i];
    const Geo g = geo_of(probs[i]);
    L.nchunk[i] = 1;
    L.line0[i] = line;
    L.unit0[i] = units;
    line += g.ntiles * g.nk;
    units += g.ntiles * L.nchunk[i];
    tiles += g.ntiles;
  }
  L.line0[n] = line;
  L.unit0[n] = units;
  L.nwg = std::min(nwg_max, line);
  return tiles;
}

}  // namespace sdumc_gg
using namespace sdumc_gg;

extern "C" size_t sdumc_gemm_group_workspace_bytes(const sdumc_gg_problem* probs, int32_t n) {
  if (!probs || n <= 0) return 0;
  const int nwg = cu_count();
  size_t need = 0;
  for (int first = 0; first < n; first += MAXP) {
    const int cnt = std::min(MAXP, n - first);
    Launch L;
    int units = 0;
    plan(probs + first, cnt, nwg, L, units);
    need = std::max(need, (size_t)(L.nwg + units) * SLOT_FLOATS * sizeof(float));
  }
  return need;
}

extern "C" int sdumc_gemm_group_tn(const sdumc_gg_problem* probs, int32_t n, void* workspace, size_t workspace_bytes, void* stream) {
  if (!probs || n <= 0 || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 15)) return SDUMC_EINVAL;
  for (int i = 0; i < n; ++i)
    if (!valid(probs[i])) return SDUMC_EINVAL;
  if (workspace_bytes < sdumc_gemm_group_workspace_bytes(probs, n)) return SDUMC_ENOMEM;
  if (!set_lds_attr()) return SDUMC_ELAUNCH;
  const int nwg = cu_count();
  hipStream_t st = as_stream(stream);
  for (int first = 0; first < n; first += MAXP) {
    const int cnt = std::min(MAXP, n - first);
    Launch L;
    int units = 0;
    const int tiles = plan(probs + first, cnt, nwg, L, units);
    if ((long long)L.line0[cnt] * (L.nwg + 1) >= (1LL << 31)) return SDUMC_EINVAL;   // 32-bit index arithmetic in the kernels
    L.slab = static_cast<float*>(workspace);
    hipLaunchKernelGGL(gg_tn_kernel, dim3(L.nwg), dim3(NTHR), LDS_BYTES, st, L);
    SDUMC_CHECK_LAUNCH();
    hipLaunchKernelGGL(gg_reduce_kernel, dim3(tiles, 17), dim3(NTHR), 0, st, L, tiles);
    SDUMC_CHECK_LAUNCH();
  }
  return SDUMC_OK;
}
