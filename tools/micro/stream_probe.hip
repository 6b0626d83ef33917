// stream_probe.hip -- what the attention-pooling backward's ACCESS PATTERN can reach on MI355X, without its arithmetic.
// Four frame tensors of R rows x 1 KiB (keys, x read; dz, dxd written), tiles of 64 rows per 256-thread workgroup:
//   copy      : plain grid-stride float4 copy keys -> dz, x -> dxd (the ceiling: same bytes, no structure)
//   tile      : one workgroup per tile; every load issued up front (16 x-rows in the MFMA operand pattern + 16 key rows per wave),
//               then the stores                                                              (= attnpool_bwd_v2's skeleton)
//   tile_occ3 : the same with 8-row half tiles per wave (half the registers: 3-4 waves per SIMD)
//   persist   : 2 workgroups per CU walk tiles with a stride; the next tile's loads are issued BEFORE this tile's stores
// Build: hipcc -O3 --offload-arch=gfx950 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int D = 256;

__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ k, const f32x4* __restrict__ x, f32x4* __restrict__ dz,
                                                    f32x4* __restrict__ dxd, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 a = k[i], b = x[i];
    dz[i] = a;
    dxd[i] = b;
  }
}

template <int RW>   // rows per wave (16 or 8)
__device__ __forceinline__ void tile_body(const float* __restrict__ k, const float* __restrict__ x, float* __restrict__ dz,
                                          float* __restrict__ dxd, size_t row0, f32x4 (&xa)[RW], f32x4 (&kr)[RW]) {
  const int lane = threadIdx.x & 63, r16 = lane & 15, kk = lane >> 4;
  // MFMA-operand pattern: lane (r16, kk) takes 16 bytes of row (r16 % RW) per instruction
#pragma unroll
  for (int j = 0; j < RW; ++j) xa[j] = *reinterpret_cast<const f32x4*>(x + (row0 + (r16 % RW)) * D + 16 * (j * (16 / RW) + (r16 / RW)) + 4 * kk);
#pragma unroll
  for (int r = 0; r < RW; ++r) kr[r] = *reinterpret_cast<const f32x4*>(k + (row0 + r) * D + 4 * lane);
  __builtin_amdgcn_sched_barrier(0);      // every load is issued before anything below
}
template <int RW>
__device__ __forceinline__ void tile_store(float* __restrict__ dz, float* __restrict__ dxd, size_t row0, const f32x4 (&xa)[RW], const f32x4 (&kr)[RW]) {
  const int lane = threadIdx.x & 63;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < RW; ++j) s += xa[j];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    *reinterpret_cast<f32x4*>(dz + (row0 + r) * D + 4 * lane) = kr[r] * 0.5f;
    *reinterpret_cast<f32x4*>(dxd + (row0 + r) * D + 4 * lane) = kr[r] + s;
  }
}

template <int RW, int OCC>
__global__ __launch_bounds__(256, OCC) void tile_kernel(const float* __restrict__ k, const float* __restrict__ x, float* __restrict__ dz,
                                                        float* __restrict__ dxd, size_t rows) {
  const int wave = threadIdx.x >> 6;
  const size_t row0 = ((size_t)blockIdx.x * 4 + wave) * RW;
  if (row0 + RW > rows) return;
  f32x4 xa[RW], kr[RW];
  tile_body<RW>(k, x, dz, dxd, row0, xa, kr);
  tile_store<RW>(dz, dxd, row0, xa, kr);
}

template <int RW>
__global__ __launch_bounds__(256, 2) void persist_kernel(const float* __restrict__ k, const float* __restrict__ x, float* __restrict__ dz,
                                                         float* __restrict__ dxd, size_t rows) {
  const int wave = threadIdx.x >> 6;
  const size_t ntile = rows / (4 * RW);
  size_t t = blockIdx.x;
  if (t >= ntile) return;
  f32x4 xa[RW], kr[RW], xb[RW], kb[RW];
  tile_body<RW>(k, x, dz, dxd, (t * 4 + wave) * RW, xa, kr);
  for (;;) {
    const size_t tn = t + gridDim.x;
    if (tn < ntile) tile_body<RW>(k, x, dz, dxd, (tn * 4 + wave) * RW, xb, kb);
    tile_store<RW>(dz, dxd, (t * 4 + wave) * RW, xa, kr);
    if (tn >= ntile) break;
    t = tn;
    const size_t tn2 = t + gridDim.x;
    if (tn2 < ntile) tile_body<RW>(k, x, dz, dxd, (tn2 * 4 + wave) * RW, xa, kr);
    tile_store<RW>(dz, dxd, (t * 4 + wave) * RW, xb, kb);
    if (tn2 >= ntile) break;
    t = tn2;
  }
}

int main() {
  const size_t rows = 80896;                 // C2: 128 x (375 + 32 + 225)
  const size_t n = rows * D;
  const int SETS = 2;
  float* buf[SETS][4];
  for (int s = 0; s < SETS; ++s)
    for (int b = 0; b < 4; ++b) { CK(hipMalloc(&buf[s][b], n * 4)); CK(hipMemset(buf[s][b], b < 2 ? 0x3c : 0, n * 4)); }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = 4.0 * n * 4;
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 5; ++i) launch(i % SETS);
    CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 20; ++i) launch(i % SETS);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms / 20 < best ? ms / 20 : best;
    }
    printf("%-34s %7.1f us  %5.2f TB/s (%.0f MB)\n", name, best * 1e3, bytes / best / 1e9, bytes / 1e6);
  };
#define ARGS(s) buf[s][0], buf[s][1], buf[s][2], buf[s][3]
  for (int g : {1024, 2048, 4096, 8192})
    run((std::string("copy grid ") + std::to_string(g)).c_str(), [&](int s) {
      hipLaunchKernelGGL(copy_kernel, dim3(g), dim3(256), 0, 0, (const f32x4*)buf[s][0], (const f32x4*)buf[s][1], (f32x4*)buf[s][2], (f32x4*)buf[s][3], n / 4); });
  run("tile 16 rows/wave, 2 WG/CU", [&](int s) { hipLaunchKernelGGL((tile_kernel<16, 2>), dim3(rows / 64), dim3(256), 0, 0, ARGS(s), rows); });
  run("tile 16 rows/wave, occ hint 3", [&](int s) { hipLaunchKernelGGL((tile_kernel<16, 3>), dim3(rows / 64), dim3(256), 0, 0, ARGS(s), rows); });
  run("tile 8 rows/wave, occ hint 4", [&](int s) { hipLaunchKernelGGL((tile_kernel<8, 4>), dim3(rows / 32), dim3(256), 0, 0, ARGS(s), rows); });
  run("tile 8 rows/wave, occ hint 6", [&](int s) { hipLaunchKernelGGL((tile_kernel<8, 6>), dim3(rows / 32), dim3(256), 0, 0, ARGS(s), rows); });
  for (int g : {256, 512, 768})
    run((std::string("persist 8 rows/wave grid ") + std::to_string(g)).c_str(), [&](int s) { hipLaunchKernelGGL((persist_kernel<8>), dim3(g), dim3(256), 0, 0, ARGS(s), rows); });
  for (int g : {256, 512})
    run((std::string("persist 16 rows/wave grid ") + std::to_string(g)).c_str(), [&](int s) { hipLaunchKernelGGL((persist_kernel<16>), dim3(g), dim3(256), 0, 0, ARGS(s), rows); });
  return 0;
}
