// Synthetic probe: how busy can the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32) be kept by W waves per SIMD that each
// run a chain of NACC independent accumulators, with and without the LDS fragment reads of the GEMM kernel?
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_chain.hip -o tools/micro/mfma_chain ; run: ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS_READS, bool BARRIER>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 64 * 36];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2 * 64 * 36; i += 256) lds[i] = 0.001f * (i & 15);
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float av = 1.0f + lane * 1e-3f, bv = 0.5f;
  const float* ap = lds + (lane & 31) * 36 + 4 * (lane >> 5);
  for (int it = 0; it < iters; ++it) {
    // one "k-tile": 16 MFMAs per accumulator-set step, like the 64x64 GEMM (NACC = 1) or the 128x128 one (NACC = 4)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 fa = {av, av, av, av}, fb = {bv, bv, bv, bv};
      if (LDS_READS) {
        fa = *reinterpret_cast<const f32x4*>(ap + 8 * g);
        fb = *reinterpret_cast<const f32x4*>(ap + 64 * 36 + 8 * g);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], acc[a], 0, 0, 0);
    }
    if (BARRIER) {
      __syncthreads();
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool L, bool B>
void run(const char* name, int wg_per_cu, float* out) {
  const int iters = 4000 / NACC;
  dim3 grid(256 * wg_per_cu), blk(256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<NACC, L, B><<<grid, blk>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NACC, L, B><<<grid, blk>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 32 * 32 * 2 * 16.0 * NACC * iters * 4 /*waves*/ * grid.x;
  printf("%-28s waves/SIMD %d  acc/wave %d : %7.3f ms  %6.1f TF\n", name, wg_per_cu, NACC, ms, flops / ms / 1e9);
}

int main(int argc, char** argv) {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  if (argc > 1) {   // sustained MFMA-only run for the clock/power poller
    for (int i = 0; i < 600; ++i) probe<1, false, false><<<dim3(256 * 4), dim3(256)>>>(out, 4000);
    hipDeviceSynchronize();
    printf("SUSTAIN mfma-only done\n");
    return 0;
  }
  for (int w : {1, 2, 3, 4, 6, 8}) run<1, false, false>("chain, no lds", w, out);
  for (int w : {1, 2, 4}) run<4, false, false>("4 acc, no lds", w, out);
  for (int w : {1, 2, 3, 4, 6, 8}) run<1, true, false>("chain + lds frag reads", w, out);
  for (int w : {2, 4}) run<4, true, false>("4 acc + lds frag reads", w, out);
  for (int w : {2, 3, 4, 6, 8}) run<1, true, true>("chain + lds + 2 barriers", w, out);
  for (int w : {1, 2}) run<4, true, true>("4 acc + lds + 2 barriers", w, out);
  return 0;
}
