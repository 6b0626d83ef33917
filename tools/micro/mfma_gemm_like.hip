// Synthetic probe #2: the 64x64-tile GEMM loop skeleton with trivial addressing, built up step by step from the
// pure MFMA chain: + global loads (A, B k-tiles, streaming), + LDS stores, + barriers, + real fragment reads.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_gemm_like.hip -o tools/micro/mfma_gemm_like
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LDK = 36;

// MODE bit0: global loads, bit1: LDS stores of the loaded tile, bit2: barriers, bit3: fragment reads from LDS
template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ A, const float* __restrict__ B, float* out, int ktiles,
                                            int lda) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 64 * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 2 * 64 * LDK; i += 256) lds[i] = 0.001f * (i & 15);
  __syncthreads();
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  // thread -> (row = tid/8 + 32 j, k = 4 (tid%8)) of a [64][32] tile
  // 4 workgroups (same XCD under round-robin placement: equal blockIdx % 8, consecutive blockIdx / 8) share one A row panel, like the N = 256 GEMMs
  const int panel = (blockIdx.x >> 5) * 8 + (blockIdx.x & 7);
  const float* ap = A + ((size_t)panel * 64 + (tid >> 3)) * lda + 4 * (tid & 7);
  const float* bp = B + ((size_t)((blockIdx.x >> 3) & 3) * 64 + (tid >> 3)) * lda + 4 * (tid & 7);
  f32x4 ra[2], rb[2];
  ra[0] = ra[1] = rb[0] = rb[1] = f32x4{1.f, 1.f, 1.f, 1.f};
  if (MODE & 1) {
    ra[0] = *reinterpret_cast<const f32x4*>(ap);
    ra[1] = *reinterpret_cast<const f32x4*>(ap + 32 * lda);
    rb[0] = *reinterpret_cast<const f32x4*>(bp);
    rb[1] = *reinterpret_cast<const f32x4*>(bp + 32 * lda);
  }
  float* As = lds;
  float* Bs = lds + 64 * LDK;
  for (int t = 0; t < ktiles; ++t) {
    if (MODE & 4) __syncthreads();
    if (MODE & 2) {
      *reinterpret_cast<f32x4*>(As + (tid >> 3) * LDK + 4 * (tid & 7)) = ra[0];
      *reinterpret_cast<f32x4*>(As + ((tid >> 3) + 32) * LDK + 4 * (tid & 7)) = ra[1];
      *reinterpret_cast<f32x4*>(Bs + (tid >> 3) * LDK + 4 * (tid & 7)) = rb[0];
      *reinterpret_cast<f32x4*>(Bs + ((tid >> 3) + 32) * LDK + 4 * (tid & 7)) = rb[1];
    }
    if (MODE & 4) __syncthreads();
    if (MODE & 1) {
      ap += 32;
      bp += 32;
      ra[0] = *reinterpret_cast<const f32x4*>(ap);
      ra[1] = *reinterpret_cast<const f32x4*>(ap + 32 * lda);
      rb[0] = *reinterpret_cast<const f32x4*>(bp);
      rb[1] = *reinterpret_cast<const f32x4*>(bp + 32 * lda);
    }
    f32x4 fa[2], fb[2];
    const float* arow = As + (wm0 + li) * LDK + 4 * lh;
    const float* brow = Bs + (wn0 + li) * LDK + 4 * lh;
    if (MODE & 8) {
      fa[0] = *reinterpret_cast<const f32x4*>(arow);
      fb[0] = *reinterpret_cast<const f32x4*>(brow);
    } else {
      fa[0] = fb[0] = f32x4{1.f, 0.5f, 0.25f, 2.f};
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g + 1 < 4) {
        if (MODE & 8) {
          fa[nxt] = *reinterpret_cast<const f32x4*>(arow + 8 * (g + 1));
          fb[nxt] = *reinterpret_cast<const f32x4*>(brow + 8 * (g + 1));
        } else {
          fa[nxt] = fa[cur];
          fb[nxt] = fb[cur];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s], fb[cur][s], acc, 0, 0, 0);
    }
  }
  float s = ra[0][0] + rb[1][3] + ra[1][1] + rb[0][2];
  for (int e = 0; e < 16; ++e) s += acc[e];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, const float* A, const float* B, float* out, int wgs, int ktiles, int lda) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  probe<MODE><<<wgs, 256>>>(A, B, out, ktiles, lda);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<MODE><<<wgs, 256>>>(A, B, out, ktiles, lda);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 64 * 64 * 32 * (double)ktiles * wgs;
  printf("%-44s wgs %5d ktiles %4d : %7.3f ms  %6.1f TF\n", name, wgs, ktiles, ms, flops / ms / 1e9);
}

int main() {
  const int wgs = 4096, ktiles = 128, lda = 4096;   // A: 4096*64 rows x 4096
  float *A, *B, *out;
  (void)hipMalloc(&A, (size_t)wgs * 64 * lda * sizeof(float) + 65536);
  (void)hipMalloc(&B, (size_t)256 * lda * sizeof(float) + 65536);
  (void)hipMalloc(&out, (size_t)wgs * 256 * sizeof(float));
  (void)hipMemset(A, 0, (size_t)wgs * 64 * lda * sizeof(float));
  (void)hipMemset(B, 0, (size_t)256 * lda * sizeof(float));
  run<0>("mfma only", A, B, out, wgs, ktiles, lda);
  run<8>("+ frag reads", A, B, out, wgs, ktiles, lda);
  run<8 | 4>("+ frag reads + barriers", A, B, out, wgs, ktiles, lda);
  run<8 | 4 | 2>("+ frag reads + barriers + lds stores", A, B, out, wgs, ktiles, lda);
  run<1>("mfma + global loads (unused)", A, B, out, wgs, ktiles, lda);
  run<8 | 4 | 2 | 1>("full skeleton (loads+stores+barriers+frags)", A, B, out, wgs, ktiles, lda);
  run<8 | 2 | 1>("full without barriers (racy, timing only)", A, B, out, wgs, ktiles, lda);
  return 0;
}
