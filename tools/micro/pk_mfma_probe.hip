// pk_mfma_probe.hip -- do packed FP32 VALU results change when a bf16 MFMA workgroup shares the CU?
// Round-4 finding inside the product: the clustered utterance-level kernels produced wrong LOW halves of
// v_pk_fma_f32 results in 9-19 % of bf16-storage forwards, only while a workgroup of the bf16 GEMM (v_mfma_f32_32x32x16_bf16) was
// resident on the same CU.  This probe isolates the two ingredients:
//   fma kernel  : 512-thread workgroups (84 KB of LDS, like the stage-A kernel), every lane runs dependent chains of f32x4 FMAs on
//                 operands from LDS and from global memory; built twice: with packed FP32 instructions and without
//   mfma kernel : 256-thread workgroups with 64 KB of LDS (they fit beside an fma workgroup on a CU) that issue
//                 v_mfma_f32_32x32x16_bf16 (or, MODE f32, v_mfma_f32_32x32x2_f32) back to back
// The fma kernel runs alone (reference), then REPS times with the mfma kernel on a second stream; results are compared bit for bit.
// Build: hipcc -O3 --offload-arch=gfx950 -o pk_mfma_probe pk_mfma_probe.hip ;  run: ./pk_mfma_probe [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int KD = 256;          // length of each dot product
constexpr int NL = 24;           // "layers" per launch

template <bool DUMMY>
__device__ __forceinline__ void fma_body(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* xs = sm;                                     // [2][KD] input rows
  for (int u = tid; u < 2 * KD; u += 512) xs[u] = X[(size_t)blockIdx.x * 2 * KD + u];
  __syncthreads();
  const int sq = lane >> 4, cg = lane & 15;           // as in chain_cluster.hip: 4 k-rows per wave-load, 16 column quads
  f32x4 res[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int l = 0; l < NL; ++l) {
    const float* M = W + (size_t)l * KD * 64 + (size_t)(wave * 32 + 4 * sq) * 64 + 4 * cg;
    f32x4 w[8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) w[j * 4 + e] = *reinterpret_cast<const f32x4*>(M + (size_t)(j * 16 + e) * 64);
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(xs + r * KD + wave * 32 + 4 * sq + j * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[r] += w[j * 4 + e] * x[e];
      }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[r][c] += __shfl_xor(acc[r][c], 32, 64);
        acc[r][c] += __shfl_xor(acc[r][c], 16, 64);
      }
    res[0] += acc[0];
    res[1] += acc[1];
  }
  if (sq == 0) {
    *reinterpret_cast<f32x4*>(out + ((size_t)blockIdx.x * 8 + wave) * 128 + 4 * cg) = res[0];
    *reinterpret_cast<f32x4*>(out + ((size_t)blockIdx.x * 8 + wave) * 128 + 64 + 4 * cg) = res[1];
  }
}
__global__ __launch_bounds__(512) void fma_packed_kernel(const float* W, const float* X, float* out) { fma_body<false>(W, X, out); }
#if defined(__HIP_DEVICE_COMPILE__)
#define NOPK __attribute__((target("no-packed-fp32-ops")))
#else
#define NOPK
#endif
__global__ NOPK __launch_bounds__(512) void fma_plain_kernel(const float* W, const float* X, float* out) { fma_body<true>(W, X, out); }

template <bool BF>
__global__ __launch_bounds__(256) void mfma_kernel(float* sink, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const float s = 1.0f + threadIdx.x * 1e-3f;
  bf16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(s + e); b[e] = (__bf16)(s - e); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s + i, acc[i], 0, 0, 0);
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) t += acc[i][e];
  if (t == 12345.678f) sink[0] = t + sm[threadIdx.x];
#endif
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 300;
  const int NWG = 256;
  std::vector<float> hW((size_t)NL * KD * 64), hX((size_t)NWG * 2 * KD);
  srand(1);
  for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 0.125f;
  for (auto& v : hX) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  float *W, *X, *out, *ref, *sink;
  const size_t nout = (size_t)NWG * 8 * 128;
  CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&X, hX.size() * 4)); CK(hipMalloc(&out, nout * 4)); CK(hipMalloc(&ref, nout * 4));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
  const size_t lds_fma = 84000, lds_mfma = 65536;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fma_packed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fma));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fma_plain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fma));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mfma));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mfma));
  hipStream_t s0, s1;
  CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  std::vector<float> hr(nout), ho(nout);
  for (int packed = 1; packed >= 0; --packed) {
    auto fma = [&](float* dst) {
      if (packed) hipLaunchKernelGGL(fma_packed_kernel, dim3(NWG), dim3(512), lds_fma, s0, W, X, dst);
      else hipLaunchKernelGGL(fma_plain_kernel, dim3(NWG), dim3(512), lds_fma, s0, W, X, dst);
    };
    fma(ref);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hr.data(), ref, nout * 4, hipMemcpyDeviceToHost));
    for (int mode = 0; mode < 3; ++mode) {       // 0 alone, 1 beside fp32 MFMA, 2 beside bf16 MFMA
      int bad_runs = 0;
      long bad_words = 0, bad_even = 0;
      for (int r = 0; r < reps; ++r) {
        if (mode == 1) hipLaunchKernelGGL(mfma_kernel<false>, dim3(1024), dim3(256), lds_mfma, s1, sink, 3000);
        if (mode == 2) hipLaunchKernelGGL(mfma_kernel<true>, dim3(1024), dim3(256), lds_mfma, s1, sink, 3000);
        fma(out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ho.data(), out, nout * 4, hipMemcpyDeviceToHost));
        long nb = 0;
        for (size_t i = 0; i < nout; ++i)
          if (memcmp(&ho[i], &hr[i], 4)) { ++nb; if ((i & 1) == 0) ++bad_even; }
        if (nb) { ++bad_runs; bad_words += nb; }
      }
      printf("%s FMAs, %s: %d of %d launches differ from the launch that ran alone (%ld words, %ld of them in even columns)\n",
             packed ? "packed  " : "unpacked", mode == 0 ? "alone          " : mode == 1 ? "beside fp32 MFMA" : "beside bf16 MFMA", bad_runs, reps,
             bad_words, bad_even);
    }
  }
  return 0;
}
