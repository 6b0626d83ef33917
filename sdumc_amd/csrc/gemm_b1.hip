// gemm_b1.hip — NT GEMMs of the bf16-STORAGE step (BASELINE configs[2] / [4]) with the weight operand in registers, gfx950.
//
//   C[M, 256 n] = act(A[M, K] . W[256 n, K]^T + bias),   A, W bf16, fp32 accumulation, C bf16 or fp32
// the frame projections frame_dim_reshape_{0,1,2} (model :193-195, :282-284; A = bf16 features) and the key projections input_proj of
// FRA2UTT_new / Cross_Attention (model :60, :82; A = the masked bf16 frames the engine materialises) of sdumc_net_dims.bf16 = 2.
//
// Why not gemm_bf16.hip's 128 x 128 LDS-staged tiles here: at bf16 MFMA rates a tile needs its operands 6x faster than the fp32
// planes kernels do, and the LDS-DMA fill path (~32 B/clk per CU: gemm_p3.hip, round 5) is what bounds it -- 437 TF on the frame
// shape, 183 TF on the K = 256 key projections, 90 TF on the text slot's 2048 x 256 x 4096 with its split-K reduce.  This kernel
// applies what the planes kernels taught:
//   * the weight is PRIVATE to a wave (wave w owns columns [64 w, 64 w + 64) of every row), so it is stored fragment-major
//     ([N / 32][K / 16][64 lanes][16 bytes]: a wave's MFMA B operand of a k-tile is one contiguous KiB) and loaded straight into
//     registers, six k-tiles ahead, through eight rotating register sets -- it never touches LDS;
//   * only A (shared by the four waves) goes through LDS: an LDS-DMA ring of 64-k stages (rows of 128 bytes, the 16-byte units of a row
//     XORed with (row >> 1) & 7: conflict-free ds_read_b128), one barrier per STAGE (16 MFMAs per wave), counted vmcnt; the fragments
//     of stage s + 1 are read into a second register set while stage s multiplies;
//   * 64 x 256 tiles on 256-thread workgroups, two resident per CU: the prologue / epilogue bursts of one overlap the k-loop of the other.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace sdumc_b1 {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int waitcnt_vm(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0xF << 8); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }
__device__ __forceinline__ float bf2f(uint32_t h16) { return __uint_as_float(h16 << 16); }
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {       // two fp32 -> one dword of two bf16 (round to nearest even)
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2s __attribute__((ext_vector_type(2)));
  const f32x2s v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

constexpr int BN = 256, BM = 64, NW = 4, NTHR = 256, TM = 2, TN = 2;
constexpr int SK = 64, KS = 4;                 // k per stage, MFMA k-steps (16 k) per stage
constexpr int NSTA = 3;                        // stages of A in the LDS ring
constexpr int A_STAGE = BM * SK * 2;           // 8 KB: two 1-KiB DMA pieces per wave
constexpr int NBS = 8, DBK = 6;                // register sets of B k-tiles / how many k-tiles ahead B is requested
constexpr int FRAG_KT = 1024;                  // bytes of one (32-row block, k-tile) of the fragment-major weight
constexpr int LDT = 68;                        // floats per staged row of the epilogue
constexpr int EPI_BYTES = NW * 32 * LDT * 4;
constexpr int LDS_BYTES = NSTA * A_STAGE > EPI_BYTES ? NSTA * A_STAGE : EPI_BYTES;
static_assert(A_STAGE == 2 * NW * 1024, "two pieces per wave");
static_assert(TM * 32 == BM && TN * 32 * NW == BN && KS * 16 == SK, "tile shape");
static_assert(DBK == 6 && NBS == 8 && NSTA == 3 && KS == 4, "the prologue and the unrolled stages are written for these depths");

struct Args {
  sdumc_gemm_b1 g;
  int nsplit, kchunk;
};

// MAPPED: the rows of A are named by a row map (sdumc_gemm_b1.a_map: A is a resident store's packed bf16 tensor, the batch is read in
// place): 1 = the tensor is below 4 GiB, the map entry replaces the row index in the descriptor offset; 2 = rows by 64-bit address
template <int MAPPED>
__global__ __launch_bounds__(NTHR, 2) void gemm_b1_nt_kernel(const Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
  const sdumc_gemm_b1& g = a.g;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_n = g.N / BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  const int ks = blockIdx.y;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = ks * a.kchunk, kend = min(g.K, kbeg + a.kchunk);
  const int nst = (kend - kbeg) / SK;                    // stages; an even number >= 2 (the launcher checks K and the split)
  const int nkt = nst * KS;

  const bool second = g.A2 != nullptr && m0 >= g.a2_row0;      // rows [a2_row0, M) live in a second tensor (the text slot's two streams)
  const int arow0 = second ? g.a2_row0 : 0;
  const int a_rows = g.a_row_mod > 0 ? g.a_row_mod : (g.A2 ? (second ? g.M - g.a2_row0 : g.a2_row0) : g.M);
  const size_t ra_rows = MAPPED == 1 ? (size_t)g.a_map_rows : (size_t)a_rows;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(second ? g.A2 : g.A), 0,
                                                                     (int)min(ra_rows * (size_t)g.lda * 2, (size_t)0xFFFFFFF0u), 0x00020000);
  // ---- A: this wave's two DMA pieces of a stage (pieces wave and wave + 4: 8 rows of 128 bytes each) ----
  [[maybe_unused]] uint32_t voff[2];
  [[maybe_unused]] const char* gaddr[2];
  [[maybe_unused]] const int32_t* amap = second ? g.a2_map : g.a_map;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int W = ((wave + 4 * i) << 6) + lane, row = W >> 3, up = W & 7, u = up ^ ((row >> 1) & 7);
    int r = min(m0 + row, g.M - 1) - arow0;
    if constexpr (MAPPED) r = amap[r];
    else if (g.a_row_mod > 0) r %= g.a_row_mod;
    if constexpr (MAPPED == 2) gaddr[i] = static_cast<const char*>(second ? g.A2 : g.A) + ((size_t)r * (size_t)g.lda + (size_t)(kbeg + 8 * u)) * 2u;
    else voff[i] = ((uint32_t)r * (uint32_t)g.lda + (uint32_t)(kbeg + 8 * u)) * 2u;
  }
  auto issue_a = [&](int buf) {
    char* base = lds + buf * A_STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (MAPPED == 2) {
        __builtin_amdgcn_global_load_lds((gbl_void_t*)gaddr[i], (lds_void_t*)(base + (wave + 4 * i) * 1024), 16, 0, 0);
        gaddr[i] += SK * 2;
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(base + (wave + 4 * i) * 1024), 16, voff[i], 0, 0, 0);
        voff[i] += SK * 2;
      }
    }
  };
  // fragment reads: rows 32 i + li, k-step j: unit (2 j + lh) ^ ((row >> 1) & 7)  ((32 i >> 1) & 7 == 0: the swizzle is the lane's)
  uint32_t aoff[KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) aoff[j] = (uint32_t)(li * 128 + (((2 * j + lh) ^ ((li >> 1) & 7)) << 4));

  // ---- B: fragment-major, this wave's two 32-column blocks, its lane's 16 bytes; k-tile q lives in register set q % NBS ----
  const char* bptr = static_cast<const char*>(g.B) + (size_t)(tile_n * 8 + wave * TN) * (size_t)g.ldb + (size_t)(kbeg >> 4) * FRAG_KT + lane * 16;
  const size_t bblk = (size_t)g.ldb;
  u32x4 pb[NBS][TN];
  auto load_b = [&](auto set_c) {
    constexpr int S = decltype(set_c)::value;
#pragma unroll
    for (int n = 0; n < TN; ++n) pb[S][n] = *reinterpret_cast<const u32x4*>(bptr + n * bblk);
    bptr += FRAG_KT;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][n][e] = 0.f;
  u32x4 fa[2][TM][KS];
  auto load_a = [&](const char* base, auto par) {
    constexpr int P = decltype(par)::value;
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[P][i][j] = *reinterpret_cast<const u32x4*>(base + i * 32 * 128 + aoff[j]);
  };
  auto op = [](const u32x4& v) { return __builtin_bit_cast(bf16x8, v); };

  // ---- prologue: A0 .. A(NSTA-1), then B k-tiles 0 .. DBK-1 ----
#pragma unroll
  for (int s = 0; s < NSTA; ++s)
    if (s < nst) issue_a(s);
  load_b(std::integral_constant<int, 0>{});
  load_b(std::integral_constant<int, 1>{});
  load_b(std::integral_constant<int, 2>{});
  load_b(std::integral_constant<int, 3>{});
  load_b(std::integral_constant<int, 4>{});
  load_b(std::integral_constant<int, 5>{});
  __builtin_amdgcn_s_waitcnt(waitcnt_vm(2 * (NSTA - 2) + DBK * TN));      // stages 0 and 1 have landed (what is younger: A2, the six B k-tiles)
  __builtin_amdgcn_s_barrier();
  load_a(lds, std::integral_constant<int, 0>{});
  int nbuf = 1;                                            // buffer of stage s + 1
  // stage s (parity P): STEADY = stage s + NSTA and k-tile 4 s + 3 + DBK exist.  The queue between the issue of stage s + 1 (top of
  // stage s - 2) and this point: 8 B loads of stage s - 2, 2 A + 8 B of stage s - 1 -> at most 18 younger operations.
  auto stage = [&](int s, auto par, auto steady_c) {
    constexpr int P = decltype(par)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    const bool more = STEADY || s + 1 < nst;
    if (more) {
      if (STEADY) __builtin_amdgcn_s_waitcnt(waitcnt_vm(18));
      else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
      __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): this wave's reads of stage s's buffer are done (refilled below)
      __builtin_amdgcn_s_barrier();
      if (STEADY || s + NSTA < nst) issue_a(nbuf == 0 ? NSTA - 1 : nbuf - 1);
      load_a(lds + nbuf * A_STAGE, std::integral_constant<int, P ^ 1>{});
      nbuf = nbuf + 1 == NSTA ? 0 : nbuf + 1;
    }
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      // (k-tile 4 s + j multiplies from set (4 P + j) % 8; the tile requested now, DBK ahead, goes into the set that was used DBK - NBS = 2 k-steps ago)
      if (STEADY || 4 * s + j + DBK < nkt) {
        if (j == 0) load_b(std::integral_constant<int, (4 * P + 0 + DBK) % NBS>{});
        else if (j == 1) load_b(std::integral_constant<int, (4 * P + 1 + DBK) % NBS>{});
        else if (j == 2) load_b(std::integral_constant<int, (4 * P + 2 + DBK) % NBS>{});
        else load_b(std::integral_constant<int, (4 * P + 3 + DBK) % NBS>{});
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(fa[P][i][j]), op(pb[(4 * P + j) % NBS][n]), acc[i][n], 0, 0, 0);
    }
  };
  int s = 0;
  for (; s + 1 + NSTA < nst; s += 2) {
    stage(s, std::integral_constant<int, 0>{}, std::true_type{});
    stage(s + 1, std::integral_constant<int, 1>{}, std::true_type{});
  }
  for (; s < nst; s += 2) {
    stage(s, std::integral_constant<int, 0>{}, std::false_type{});
    stage(s + 1, std::integral_constant<int, 1>{}, std::false_type{});
  }

  // ---- epilogue: the tile turns through LDS so that a lane owns 8 consecutive columns of a row (16-byte bf16 / 32-byte fp32 stores) ----
  const bool to_slab = a.nsplit > 1;
  float* tw = reinterpret_cast<float*>(lds) + wave * 32 * LDT;
  const int colw = n0 + 64 * wave;
  float bv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) bv[c] = (!to_slab && g.bias) ? g.bias[colw + 8 * (lane & 7) + c] : 0.f;
  __syncthreads();                                          // every wave is done reading the last stage
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) tw[((e & 3) + 8 * (e >> 2) + 4 * lh) * LDT + 32 * n + li] = acc[i][n][e];
    __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): this wave's own LDS writes (no other wave reads them)
#pragma unroll
    for (int u = lane; u < 256; u += 64) {
      const int r = u >> 3, cq = u & 7;
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
          if (g.c_bf16) {
            uint32_t* dst = reinterpret_cast<uint32_t*>(static_cast<unsigned short*>(g.C) + (size_t)row * g.ldc + col);
            *reinterpret_cast<u32x4*>(dst) = u32x4{f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7])};
          } else {
            float* dst = static_cast<float*>(g.C) + (size_t)row * g.ldc + col;
            *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
          }
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                     // reads done before the next 32 rows overwrite the staging
  }
#endif
}

// ordered reduction of the split-K slabs + the epilogue the tiles skipped
__global__ __launch_bounds__(256) void b1_splitk_reduce_kernel(const sdumc_gemm_b1 g, const int nsplit) {
  const size_t u = (size_t)blockIdx.x * 256 + threadIdx.x;       // one 8-column chunk of one row
  const int cpr = g.N >> 3;
  if (u >= (size_t)g.M * cpr) return;
  const int row = (int)(u / cpr), col = (int)(u - (size_t)row * cpr) * 8;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* s = g.workspace + (size_t)row * g.N + col;
  const size_t slab = (size_t)g.M * g.N;
  int z = 0;
  for (; z + 4 <= nsplit; z += 4) {      // four slabs in flight; summed in ascending order
    f32x4 q[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      q[k][0] = *reinterpret_cast<const f32x4*>(s + (size_t)(z + k) * slab);
      q[k][1] = *reinterpret_cast<const f32x4*>(s + (size_t)(z + k) * slab + 4);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) { v[c] += q[k][0][c]; v[4 + c] += q[k][1][c]; }
  }
  for (; z < nsplit; ++z) {
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(s + (size_t)z * slab), a1 = *reinterpret_cast<const f32x4*>(s + (size_t)z * slab + 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) { v[c] += a0[c]; v[4 + c] += a1[c]; }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float x = v[c] + (g.bias ? g.bias[col + c] : 0.f);
    if (g.act == SDUMC_ACT_TANH) x = fast_tanh(x);
    else if (g.act == SDUMC_ACT_RELU) x = fmaxf(x, 0.f);
    v[c] = x;
  }
  if (g.c_bf16) {
    uint32_t* dst = reinterpret_cast<uint32_t*>(static_cast<unsigned short*>(g.C) + (size_t)row * g.ldc + col);
    *reinterpret_cast<u32x4*>(dst) = u32x4{f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7])};
  } else {
    float* dst = static_cast<float*>(g.C) + (size_t)row * g.ldc + col;
    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
}

// up to 12 fp32 weights [rows_i][cols_i] of the flat parameter buffer -> fragment-major bf16 ([rows / 32][cols / 16][64 lanes][16 bytes],
// lane (li, lh) of block (rb, kt) holds W[32 rb + li][16 kt + 8 lh .. + 7], rounded to nearest even); blockIdx.y = tensor
struct FragList {
  int64_t src_off[12], dst_off[12];      // floats into the parameter buffer / bytes into the destination
  int32_t rows[12], cols[12];
};
__global__ __launch_bounds__(256) void b1_frag_multi_kernel(const float* __restrict__ P, char* __restrict__ dst, const FragList L) {
  const int i = blockIdx.y;
  const int rows = L.rows[i], cols = L.cols[i], cpr = cols >> 3;
  const float* src = P + L.src_off[i];
  char* out = dst + L.dst_off[i];
  const int64_t total = (int64_t)rows * cpr;
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
    const int row = (int)(u / cpr), c = (int)(u - (int64_t)row * cpr);
    const float* sp = src + (int64_t)row * cols + 8 * c;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(sp), a1 = *reinterpret_cast<const f32x4*>(sp + 4);
    const int rb = row >> 5, li = row & 31, kt = c >> 1, lh = c & 1;
    char* d = out + ((int64_t)rb * (cols >> 4) + kt) * FRAG_KT + (lh * 32 + li) * 16;
    *reinterpret_cast<u32x4*>(d) = u32x4{f2bf2(a0[0], a0[1]), f2bf2(a0[2], a0[3]), f2bf2(a1[0], a1[1]), f2bf2(a1[2], a1[3])};
  }
}

struct Plan {
  int nsplit, kchunk;
};
// k comes in pairs of 64-k stages (the unrolled loop); K split over workgroups (fp32 slabs + an ordered reduce) only when the 64-row
// tiles leave most of the chip idle and K is long (the text slot: 4096 rows x 4096)
inline Plan plan(const sdumc_gemm_b1& g, size_t have) {
  const int kp = g.K / (2 * SK);                       // pairs of stages
  const long tiles = (long)((g.M + BM - 1) / BM) * (g.N / BN);
  int s = 1;
  if (g.splitk >= 1) s = std::min(g.splitk, kp);
  else if (tiles <= 128 && kp >= 8) s = (int)std::min<long>(512 / std::max<long>(1, tiles), kp / 2);
  if (s < 1) s = 1;
  while (s > 1 && (size_t)s * g.M * g.N * sizeof(float) > have) --s;
  Plan p;
  p.kchunk = ((kp + s - 1) / s) * 2 * SK;
  p.nsplit = (g.K + p.kchunk - 1) / p.kchunk;
  return p;
}

}  // namespace sdumc_b1

extern "C" int sdumc_prof_begin_(int variant, double flops, void* stream);     // gemm_f32.hip: bench.py's per-launch HIP events
extern "C" void sdumc_prof_end_(int token, void* stream);

extern "C" size_t sdumc_gemm_b1_workspace_bytes(const sdumc_gemm_b1* g) {
  if (!g || g->M <= 0 || g->N <= 0 || g->K <= 0 || (g->K % (2 * sdumc_b1::SK))) return 0;
  const sdumc_b1::Plan p = sdumc_b1::plan(*g, (size_t)-1);
  return p.nsplit > 1 ? (size_t)p.nsplit * g->M * g->N * sizeof(float) : 0;
}

extern "C" int sdumc_gemm_b1_nt(const sdumc_gemm_b1* gp, void* stream) {
  using namespace sdumc_b1;
  if (!gp) return SDUMC_EINVAL;
  const sdumc_gemm_b1& g = *gp;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.N % BN) || (g.K % (2 * SK))) return SDUMC_EINVAL;
  if (!g.A || !g.B || !g.C) return SDUMC_EINVAL;
  if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.B) | reinterpret_cast<uintptr_t>(g.C) | reinterpret_cast<uintptr_t>(g.A2)) & 15) return SDUMC_EINVAL;
  if (g.lda < g.K || (g.lda & 7) || g.ldc < g.N || (g.ldc & 7)) return SDUMC_EINVAL;
  if (g.ldb < (int64_t)(g.K / 16) * FRAG_KT || (g.ldb & 15)) return SDUMC_EINVAL;
  if (g.act != SDUMC_ACT_NONE && g.act != SDUMC_ACT_TANH && g.act != SDUMC_ACT_RELU) return SDUMC_EINVAL;
  if (g.A2 && (g.a_row_mod || g.a2_row0 <= 0 || g.a2_row0 >= g.M || (g.a2_row0 % BM))) return SDUMC_EINVAL;
  const bool mapped = g.a_map != nullptr;
  if (mapped && (g.a_row_mod || (g.A2 != nullptr) != (g.a2_map != nullptr) || ((reinterpret_cast<uintptr_t>(g.a_map) | reinterpret_cast<uintptr_t>(g.a2_map)) & 3))) return SDUMC_EINVAL;
  if (!mapped && g.a2_map) return SDUMC_EINVAL;
  const size_t a_bytes = (size_t)(g.a_row_mod > 0 ? g.a_row_mod : (g.A2 ? std::max(g.a2_row0, g.M - g.a2_row0) : g.M)) * (size_t)g.lda * 2;
  if (!mapped && a_bytes >= 0xFFFFFFF0u) return SDUMC_EINVAL;      // (mapped rows beyond 4 GiB are fetched by 64-bit address)
  const int mode = !mapped ? 0 : (g.a_map_rows > 0 && (size_t)g.a_map_rows * (size_t)g.lda * 2 < 0xFFFFFFF0u) ? 1 : 2;
  const Plan p = plan(g, g.workspace ? g.workspace_bytes : 0);
  if (p.nsplit > 1 && (!g.workspace || (reinterpret_cast<uintptr_t>(g.workspace) & 15))) return SDUMC_ENOMEM;
  static sdumc_dev_once attr_set;
  if (sdumc_once_per_device(attr_set, [] {
        return sdumc_set_dyn_lds(&gemm_b1_nt_kernel<0>, LDS_BYTES) && sdumc_set_dyn_lds(&gemm_b1_nt_kernel<1>, LDS_BYTES) && sdumc_set_dyn_lds(&gemm_b1_nt_kernel<2>, LDS_BYTES);
      }) != SDUMC_OK)
    return SDUMC_ELAUNCH;
  hipStream_t st = as_stream(stream);
  const int tok = sdumc_prof_begin_(28, 2.0 * g.M * (double)g.N * g.K, stream);
  Args a{g, p.nsplit, p.kchunk};
  const dim3 grid((unsigned)(((g.M + BM - 1) / BM) * (g.N / BN)), (unsigned)p.nsplit);
  if (mode == 0) hipLaunchKernelGGL(gemm_b1_nt_kernel<0>, grid, dim3(NTHR), LDS_BYTES, st, a);
  else if (mode == 1) hipLaunchKernelGGL(gemm_b1_nt_kernel<1>, grid, dim3(NTHR), LDS_BYTES, st, a);
  else hipLaunchKernelGGL(gemm_b1_nt_kernel<2>, grid, dim3(NTHR), LDS_BYTES, st, a);
  SDUMC_CHECK_LAUNCH();
  sdumc_prof_end_(tok, stream);
  if (p.nsplit > 1) {
    const size_t units = (size_t)g.M * (g.N >> 3);
    hipLaunchKernelGGL(b1_splitk_reduce_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st, g, p.nsplit);
    SDUMC_CHECK_LAUNCH();
  }
  return SDUMC_OK;
}

// n <= 12 weights [rows_i][cols_i] at P + src_off[i] (row stride cols_i; rows % 32 == 0, cols % 16 == 0) -> fragment-major bf16 at
// dst + dst_off[i] (bytes): the B operand of sdumc_gemm_b1_nt
extern "C" int sdumc_b1_frag_multi(const float* P, void* dst, const int64_t* src_off, const int64_t* dst_off, const int32_t* rows,
                                   const int32_t* cols, int n, void* stream) {
  if (!P || !dst || n < 1 || n > 12 || (reinterpret_cast<uintptr_t>(P) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) return SDUMC_EINVAL;
  sdumc_b1::FragList L;
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    if ((rows[i] & 31) || (cols[i] & 15) || (src_off[i] & 3) || (dst_off[i] & 15) || rows[i] <= 0 || cols[i] <= 0) return SDUMC_EINVAL;
    L.src_off[i] = src_off[i]; L.dst_off[i] = dst_off[i]; L.rows[i] = rows[i]; L.cols[i] = cols[i];
    most = std::max<int64_t>(most, (int64_t)rows[i] * (cols[i] >> 3));
  }
  const unsigned bx = (unsigned)std::min<int64_t>((most + 255) / 256, 128);
  hipLaunchKernelGGL(sdumc_b1::b1_frag_multi_kernel, dim3(bx, n), dim3(256), 0, as_stream(stream), P, static_cast<char*>(dst), L);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}

// One empty kernel per source file = per gfx950 code object: sdumc_preload_() asks for its attributes, which makes the HIP runtime load
// this file's code object NOW (outside any timed or latency-sensitive region) instead of at the first launch of one of its kernels --
// with deferred loading that first launch stalls the host for tens of milliseconds (seen as a 36-59 ms gap in the middle of an epoch,
// at the first batch whose shape took a fallback path: profiles/README.md, round 6).
__global__ void sdumc_preload_gemm_b1_kernel() {}
extern "C" int sdumc_preload_gemm_b1_(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&sdumc_preload_gemm_b1_kernel)) == hipSuccess ? SDUMC_OK : SDUMC_ELAUNCH;
}
