// Synthetic probe #4: "weight-panel resident" GEMM for the K = 256 key projections (C[M,256] = A[M,256] . W[256,256]^T):
// a persistent workgroup keeps one 64-column panel of W for all of K in LDS (or in registers) and streams 64-row
// tiles of A through a register-staged LDS tile; the k-loop runs on into the next m-tile, so no workgroup-level
// prologue/epilogue convoy exists.  Compared with the library-style one-tile-per-workgroup loop on the same problem.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/panel_probe.hip -o tools/micro/panel_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 256, N = 256, BK = 32, LDA = BK + 4, LDB = K + 4;

// BREG = false: W panel in LDS [64][K+4];  BREG = true: each wave keeps its 32 columns x K of W in 128 VGPRs
template <bool BREG>
__global__ __launch_bounds__(256, 2) void panel_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                      float* __restrict__ C, int M, int wgs_per_panel) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                    // [64][LDA]
  float* Bs = lds + 64 * LDA;         // [64][LDB] (unused with BREG)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  const int panel = blockIdx.x & 3, slot = blockIdx.x >> 2;
  const int n0 = panel * 64;
  // ---- W panel, once
  f32x4 breg[BREG ? K / 8 : 1];
  if (BREG) {
    // fragment of k-group g for this lane: W[n0 + wn0 + li][8g + 4lh .. +3]
#pragma unroll
    for (int g = 0; g < K / 8; ++g)
      breg[g] = *reinterpret_cast<const f32x4*>(W + (size_t)(n0 + wn0 + li) * K + 8 * g + 4 * lh);
  } else {
    for (int i = tid; i < 64 * (K / 4); i += 256) {
      const int r = i / (K / 4), c = i % (K / 4);
      *reinterpret_cast<f32x4*>(Bs + r * LDB + 4 * c) = *reinterpret_cast<const f32x4*>(W + (size_t)(n0 + r) * K + 4 * c);
    }
  }
  const int mtiles = (M + 63) / 64;
  // A staging: thread -> (row = tid/8 + 32 j, k = 4 (tid%8))
  f32x4 ra[2];
  auto loadA = [&](int mt, int kt) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = mt * 64 + (tid >> 3) + 32 * j;
      ra[j] = row < M ? *reinterpret_cast<const f32x4*>(A + (size_t)row * K + kt * BK + 4 * (tid & 7)) : f32x4{0, 0, 0, 0};
    }
  };
  int mt = slot;
  if (mt < mtiles) loadA(mt, 0);
  while (mt < mtiles) {
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll 1
    for (int kt = 0; kt < K / BK; ++kt) {
      __syncthreads();
      *reinterpret_cast<f32x4*>(As + (tid >> 3) * LDA + 4 * (tid & 7)) = ra[0];
      *reinterpret_cast<f32x4*>(As + ((tid >> 3) + 32) * LDA + 4 * (tid & 7)) = ra[1];
      __syncthreads();
      // next A tile: the next k-tile of this m-tile, or the first k-tile of the next m-tile
      if (kt + 1 < K / BK) loadA(mt, kt + 1);
      else if (mt + wgs_per_panel < mtiles) loadA(mt + wgs_per_panel, 0);
      f32x4 fa[2], fb[2];
      const float* arow = As + (wm0 + li) * LDA + 4 * lh;
      const float* brow = Bs + (wn0 + li) * LDB + kt * BK + 4 * lh;
      fa[0] = *reinterpret_cast<const f32x4*>(arow);
      if (!BREG) fb[0] = *reinterpret_cast<const f32x4*>(brow);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cur = g & 1, nxt = cur ^ 1;
        if (g + 1 < 4) {
          fa[nxt] = *reinterpret_cast<const f32x4*>(arow + 8 * (g + 1));
          if (!BREG) fb[nxt] = *reinterpret_cast<const f32x4*>(brow + 8 * (g + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 b = BREG ? breg[(kt * 4 + g) % (BREG ? K / 8 : 1)] : fb[cur];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s], b[s], acc, 0, 0, 0);
      }
    }
    // epilogue (the loads of the next m-tile's first k-tile are in flight)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = mt * 64 + wm0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      if (row < M) C[(size_t)row * N + n0 + wn0 + li] = tanhf(acc[e]);
    }
    mt += wgs_per_panel;
  }
}

// reference structure: one 64x64 tile per workgroup, A and B staged per k-tile (the library kernel's loop)
__global__ __launch_bounds__(256, 4) void tile_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                     float* __restrict__ C, int M) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 64 * LDA];
  float* As = lds;
  float* Bs = lds + 64 * LDA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  const int n0 = (blockIdx.x & 3) * 64, m0 = (blockIdx.x >> 2) * 64;
  f32x4 ra[2], rb[2];
  auto load = [&](int kt) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = m0 + (tid >> 3) + 32 * j;
      ra[j] = row < M ? *reinterpret_cast<const f32x4*>(A + (size_t)row * K + kt * BK + 4 * (tid & 7)) : f32x4{0, 0, 0, 0};
      rb[j] = *reinterpret_cast<const f32x4*>(W + (size_t)(n0 + (tid >> 3) + 32 * j) * K + kt * BK + 4 * (tid & 7));
    }
  };
  load(0);
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll 1
  for (int kt = 0; kt < K / BK; ++kt) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      *reinterpret_cast<f32x4*>(As + ((tid >> 3) + 32 * j) * LDA + 4 * (tid & 7)) = ra[j];
      *reinterpret_cast<f32x4*>(Bs + ((tid >> 3) + 32 * j) * LDA + 4 * (tid & 7)) = rb[j];
    }
    __syncthreads();
    if (kt + 1 < K / BK) load(kt + 1);
    f32x4 fa[2], fb[2];
    const float* arow = As + (wm0 + li) * LDA + 4 * lh;
    const float* brow = Bs + (wn0 + li) * LDA + 4 * lh;
    fa[0] = *reinterpret_cast<const f32x4*>(arow);
    fb[0] = *reinterpret_cast<const f32x4*>(brow);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g + 1 < 4) {
        fa[nxt] = *reinterpret_cast<const f32x4*>(arow + 8 * (g + 1));
        fb[nxt] = *reinterpret_cast<const f32x4*>(brow + 8 * (g + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s], fb[cur][s], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = m0 + wm0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
    if (row < M) C[(size_t)row * N + n0 + wn0 + li] = tanhf(acc[e]);
  }
}

__global__ void fill_random(float* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    unsigned x = (unsigned)i * 2654435761u ^ seed;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    p[i] = ((float)(x & 0xffff) / 32768.0f - 1.0f) * 0.1f;
  }
}

template <typename F>
float timeit(F&& f) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) f();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 20;
}

int main() {
  for (int M : {48000, 28800, 96000}) {
    float *A, *W, *C, *C2;
    (void)hipMalloc(&A, (size_t)M * K * 4);
    (void)hipMalloc(&W, (size_t)N * K * 4);
    (void)hipMalloc(&C, (size_t)M * N * 4);
    (void)hipMalloc(&C2, (size_t)M * N * 4);
    fill_random<<<(unsigned)(((size_t)M * K + 255) / 256), 256>>>(A, (size_t)M * K, 1);
    fill_random<<<(N * K + 255) / 256, 256>>>(W, (size_t)N * K, 2);
    const double flops = 2.0 * M * N * K;
    const int mt = (M + 63) / 64;
    float ms = timeit([&] { tile_kernel<<<mt * 4, 256>>>(A, W, C2, M); });
    printf("M=%6d tile-per-workgroup (library structure) : %7.1f us %6.1f TF\n", M, ms * 1e3, flops / ms / 1e9);
    for (int per : {128, 256}) {   // persistent workgroups per panel (x4 panels)
      const size_t sh = (64 * LDA + 64 * LDB) * 4;
      (void)hipFuncSetAttribute((const void*)panel_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      ms = timeit([&] { panel_kernel<false><<<per * 4, 256, sh>>>(A, W, C, M, per); });
      printf("M=%6d W panel in LDS, %4d persistent WGs      : %7.1f us %6.1f TF\n", M, per * 4, ms * 1e3, flops / ms / 1e9);
      const size_t sh2 = 64 * LDA * 4;
      ms = timeit([&] { panel_kernel<true><<<per * 4, 256, sh2>>>(A, W, C, M, per); });
      printf("M=%6d W panel in VGPRs, %4d persistent WGs    : %7.1f us %6.1f TF\n", M, per * 4, ms * 1e3, flops / ms / 1e9);
    }
    // correctness of the two panel variants against the tile kernel
    tile_kernel<<<mt * 4, 256>>>(A, W, C2, M);
    panel_kernel<true><<<512, 256, 64 * LDA * 4>>>(A, W, C, M, 128);
    (void)hipDeviceSynchronize();
    float* h1 = (float*)malloc((size_t)M * N * 4);
    float* h2 = (float*)malloc((size_t)M * N * 4);
    (void)hipMemcpy(h1, C, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h2, C2, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    double md = 0;
    for (size_t i = 0; i < (size_t)M * N; ++i) md = fmax(md, fabs((double)h1[i] - h2[i]));
    printf("M=%6d max |panel - tile| = %.3g\n", M, md);
    free(h1); free(h2);
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(C); (void)hipFree(C2);
  }
  return 0;
}
