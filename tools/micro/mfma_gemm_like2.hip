// Synthetic probe #3: structural variants of the fp32 64x64-tile GEMM loop (trivial addressing, 4 WGs share an A panel).
//   BKT   : k-tile depth (32 or 64)
//   PRIO  : s_setprio(1) around the MFMA block
//   GLDS  : 0 = register staging (global_load -> ds_write), 1 = LDS-DMA (global_load_lds_dwordx4) into 2 LDS buffers,
//           XOR-swizzled source addresses, one barrier per k-tile
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_gemm_like2.hip -o tools/micro/mfma_gemm_like2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BKT, bool PRIO>
__global__ __launch_bounds__(256) void probe_reg(const float* __restrict__ A, const float* __restrict__ B, float* out, int ktiles,
                                                int lda) {
  constexpr int LDK = BKT + 4, NV = BKT / 16;   // f32x4 per thread per operand
  __shared__ __attribute__((aligned(16))) float lds[2 * 64 * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  constexpr int TPR = BKT / 4;            // threads per row
  constexpr int RPP = 256 / TPR;          // rows per pass
  const int panel = (blockIdx.x >> 5) * 8 + (blockIdx.x & 7);
  const float* ap = A + ((size_t)panel * 64 + tid / TPR) * lda + 4 * (tid % TPR);
  const float* bp = B + ((size_t)((blockIdx.x >> 3) & 3) * 64 + tid / TPR) * lda + 4 * (tid % TPR);
  f32x4 ra[NV], rb[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    ra[j] = *reinterpret_cast<const f32x4*>(ap + (size_t)j * RPP * lda);
    rb[j] = *reinterpret_cast<const f32x4*>(bp + (size_t)j * RPP * lda);
  }
  float* As = lds;
  float* Bs = lds + 64 * LDK;
  for (int t = 0; t < ktiles; ++t) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      *reinterpret_cast<f32x4*>(As + (tid / TPR + j * RPP) * LDK + 4 * (tid % TPR)) = ra[j];
      *reinterpret_cast<f32x4*>(Bs + (tid / TPR + j * RPP) * LDK + 4 * (tid % TPR)) = rb[j];
    }
    __syncthreads();
    ap += BKT;
    bp += BKT;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      ra[j] = *reinterpret_cast<const f32x4*>(ap + (size_t)j * RPP * lda);
      rb[j] = *reinterpret_cast<const f32x4*>(bp + (size_t)j * RPP * lda);
    }
    f32x4 fa[2], fb[2];
    const float* arow = As + (wm0 + li) * LDK + 4 * lh;
    const float* brow = Bs + (wn0 + li) * LDK + 4 * lh;
    fa[0] = *reinterpret_cast<const f32x4*>(arow);
    fb[0] = *reinterpret_cast<const f32x4*>(brow);
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int g = 0; g < BKT / 8; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g + 1 < BKT / 8) {
        fa[nxt] = *reinterpret_cast<const f32x4*>(arow + 8 * (g + 1));
        fb[nxt] = *reinterpret_cast<const f32x4*>(brow + 8 * (g + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s], fb[cur][s], acc, 0, 0, 0);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  }
  float s = 0.f;
  for (int j = 0; j < NV; ++j) s += ra[j][0] + rb[j][3];
  for (int e = 0; e < 16; ++e) s += acc[e];
  out[blockIdx.x * 256 + tid] = s;
}

// LDS-DMA variant: unpadded [64][32] tiles (128-B rows), 16-B chunk c of row r stored at chunk slot c ^ ((r >> 1) & 7);
// NBUF LDS buffers, tile t+NBUF-1 is put in flight right after the barrier of iteration t.
typedef __attribute__((address_space(3))) void lds_void;
template <int NBUF, bool PRIO>
__global__ __launch_bounds__(256) void probe_glds(const float* __restrict__ A, const float* __restrict__ B, float* out, int ktiles,
                                                 int lda) {
  constexpr int TILE = 64 * 32;                                   // floats per operand tile
  __shared__ __attribute__((aligned(1024))) float lds[NBUF * 2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  const int panel = (blockIdx.x >> 5) * 8 + (blockIdx.x & 7);
  // wave w fills rows [16w, 16w+16) of A and of B: two 1-KiB pieces (8 rows x 128 B) per operand.
  // lane -> (row r = 16w + 8p + lane/8, slot = lane%8) ; source chunk = slot ^ ((r >> 1) & 7)
  const float* asrc[2];
  const float* bsrc[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int r = 16 * wave + 8 * p + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    asrc[p] = A + ((size_t)panel * 64 + r) * lda + 4 * c;
    bsrc[p] = B + ((size_t)((blockIdx.x >> 3) & 3) * 64 + r) * lda + 4 * c;
  }
  auto issue = [&](int buf) {
    float* base = lds + buf * 2 * TILE;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      __builtin_amdgcn_global_load_lds(asrc[p], (lds_void*)(base + (16 * wave + 8 * p) * 32), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(bsrc[p], (lds_void*)(base + TILE + (16 * wave + 8 * p) * 32), 16, 0, 0);
      asrc[p] += 32;
      bsrc[p] += 32;
    }
  };
#pragma unroll
  for (int b = 0; b < NBUF - 1; ++b) issue(b);
  // fragment addresses: row wm0+li, logical chunk 2g+lh -> slot (2g+lh) ^ ((row >> 1) & 7)
  const int arow = wm0 + li, brow = wn0 + li;
  const int asw = (arow >> 1) & 7, bsw = (brow >> 1) & 7;
  int buf = 0;
  for (int t = 0; t < ktiles; ++t) {
    // tile t must have landed: all but the (NBUF-2) most recent groups of 4 DMAs
    if (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {   // refill the buffer everyone has just finished reading (tile t-1's) with tile t+NBUF-1
      int nb = buf + NBUF - 1;
      if (nb >= NBUF) nb -= NBUF;
      issue(nb);
    }
    const float* As = lds + buf * 2 * TILE;
    const float* Bs = As + TILE;
    f32x4 fa[2], fb[2];
    fa[0] = *reinterpret_cast<const f32x4*>(As + arow * 32 + 4 * ((0 + lh) ^ asw));
    fb[0] = *reinterpret_cast<const f32x4*>(Bs + brow * 32 + 4 * ((0 + lh) ^ bsw));
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g + 1 < 4) {
        fa[nxt] = *reinterpret_cast<const f32x4*>(As + arow * 32 + 4 * ((2 * (g + 1) + lh) ^ asw));
        fb[nxt] = *reinterpret_cast<const f32x4*>(Bs + brow * 32 + 4 * ((2 * (g + 1) + lh) ^ bsw));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s], fb[cur][s], acc, 0, 0, 0);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (++buf == NBUF) buf = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc[e];
  out[blockIdx.x * 256 + tid] = s;
}

template <typename K>
void run(const char* name, K kern, const float* A, const float* B, float* out, int wgs, int ktiles, int bk, int lda) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  kern<<<wgs, 256>>>(A, B, out, ktiles, lda);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  kern<<<wgs, 256>>>(A, B, out, ktiles, lda);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 64 * 64 * bk * (double)ktiles * wgs;
  printf("%-46s : %7.3f ms  %6.1f TF\n", name, ms, flops / ms / 1e9);
}

template <typename K>
void sustain(const char* name, K kern, const float* A, const float* B, float* out, int wgs, int ktiles, int bk, int lda, double seconds) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int n = (int)(seconds / 1.1e-3);
  printf("SUSTAIN %s start\n", name);
  fflush(stdout);
  (void)hipEventRecord(e0);
  for (int i = 0; i < n; ++i) kern<<<wgs, 256>>>(A, B, out, ktiles, lda);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("SUSTAIN %s : %d launches, %.3f ms each, %.1f TF\n", name, n, ms / n, 2.0 * 64 * 64 * bk * (double)ktiles * wgs * n / ms / 1e9);
  fflush(stdout);
}

__global__ void fill_random(float* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    unsigned x = (unsigned)i * 2654435761u ^ seed;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    p[i] = (float)(x & 0xffff) / 32768.0f - 1.0f;
  }
}

int main(int argc, char** argv) {
  const int wgs = 4096, K = 4096, lda = 4096 + 64;
  float *A, *B, *out;
  (void)hipMalloc(&A, (size_t)(wgs / 4 + 8) * 64 * lda * sizeof(float));
  (void)hipMalloc(&B, (size_t)320 * lda * sizeof(float));
  (void)hipMalloc(&out, (size_t)wgs * 256 * sizeof(float));
  (void)hipMemset(A, 0, (size_t)(wgs / 4 + 8) * 64 * lda * sizeof(float));
  (void)hipMemset(B, 0, (size_t)320 * lda * sizeof(float));
  if (argc > 1) {   // sustained runs for the clock/power poller: zero-filled, then random operands
    sustain("reg BK=32 zeros", probe_reg<32, false>, A, B, out, wgs, K / 32, 32, lda, 4.0);
    const size_t na = (size_t)(wgs / 4 + 8) * 64 * lda, nb = (size_t)320 * lda;
    fill_random<<<(unsigned)((na + 255) / 256), 256>>>(A, na, 1u);
    fill_random<<<(unsigned)((nb + 255) / 256), 256>>>(B, nb, 2u);
    (void)hipDeviceSynchronize();
    sustain("reg BK=32 random", probe_reg<32, false>, A, B, out, wgs, K / 32, 32, lda, 4.0);
    sustain("glds3 BK=32 random", probe_glds<3, false>, A, B, out, wgs, K / 32, 32, lda, 4.0);
    return 0;
  }
  for (int rep = 0; rep < 2; ++rep) {
    run("reg staging BK=32", probe_reg<32, false>, A, B, out, wgs, K / 32, 32, lda);
    run("reg staging BK=32 + setprio", probe_reg<32, true>, A, B, out, wgs, K / 32, 32, lda);
    run("reg staging BK=64", probe_reg<64, false>, A, B, out, wgs, K / 64, 64, lda);
    run("reg staging BK=64 + setprio", probe_reg<64, true>, A, B, out, wgs, K / 64, 64, lda);
    run("glds 2 buffers BK=32", probe_glds<2, false>, A, B, out, wgs, K / 32, 32, lda);
    run("glds 3 buffers BK=32", probe_glds<3, false>, A, B, out, wgs, K / 32, 32, lda);
    run("glds 3 buffers BK=32 + setprio", probe_glds<3, true>, A, B, out, wgs, K / 32, 32, lda);
  }
  return 0;
}
