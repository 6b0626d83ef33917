// Cost of one activation exchange between the workgroups of a "cluster" (c workgroups that split the columns of a layer and
// swap their slices through global memory + a flag) on MI355X: same-XCD clusters (members blockIdx = k + j * nclusters, equal
// modulo 8) against adjacent-block clusters (members spread over XCDs).  Decides whether chain.hip's layers can be column-split
// over several CUs.  build: hipcc --offload-arch=gfx950 -O3 -o cluster_sync_probe cluster_sync_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ROWS = 2, COLS = 256;

template <bool SAME_XCD>
__global__ __launch_bounds__(512) void probe(float* xch, unsigned* flags, int* err, float* out, int iters, int c, int nclusters) {
  const int bid = blockIdx.x, tid = threadIdx.x;
  const int cluster = SAME_XCD ? bid % nclusters : bid / c;
  const int member = SAME_XCD ? bid / nclusters : bid % c;
  const int slice = COLS / c;
  float* buf = xch + (size_t)cluster * 2 * ROWS * COLS;   // double-buffered by iteration parity
  unsigned* flag = flags + cluster * 32;                   // own 128-byte line
  float acc = (float)tid;
  __shared__ int bail;
  if (tid == 0) bail = 0;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    float* b = buf + (it & 1) * ROWS * COLS;
    if (tid < ROWS * slice) {
      const int r = tid / slice, cc = tid - r * slice;
      // value = f(iteration, cluster, position): the consumer checks every element it reads (stale lines of an earlier
      // iteration or of another XCD's L2 would show up as mismatches)
      __hip_atomic_store(b + r * COLS + member * slice + cc, (float)(it * 7 + cluster) + 0.001f * (float)(r * COLS + member * slice + cc),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0);     // the write-through stores have been acknowledged
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)c * (it + 1);
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > 2000000) { bail = 1; *err = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (bail) return;
    const float v = __hip_atomic_load(b + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // 512 threads = ROWS * COLS
    if (v != (float)(it * 7 + cluster) + 0.001f * (float)tid) atomicAdd(err + 1, 1);
    acc = acc * 0.999f + v;
  }
  out[(size_t)bid * 512 + tid] = acc;
}

__global__ void dirty_l2(float* p, size_t n, int reps) {
  for (int k = 0; k < reps; ++k)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + k;
}

int main() {
  const int nwg = 256, iters = 200;
  float *xch, *out; unsigned* flags; int* err;
  CK(hipMalloc(&xch, (size_t)nwg * 2 * ROWS * COLS * sizeof(float)));
  CK(hipMalloc(&out, (size_t)nwg * 512 * sizeof(float)));
  CK(hipMalloc(&flags, nwg * 32 * sizeof(unsigned)));
  CK(hipMalloc(&err, 2 * sizeof(int)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* big;
  const size_t nbig = 64u << 20;
  CK(hipMalloc(&big, nbig * sizeof(float)));
  CK(hipMemset(big, 0, nbig * sizeof(float)));
  hipStream_t side;
  CK(hipStreamCreate(&side));
  for (int load = 0; load < 2; ++load)
  for (int c : {1, 2, 4, 8}) {
    if (load) { printf("-- with a read-modify-write sweep over 256 MB running on another stream (64 workgroups) --\n");
                hipLaunchKernelGGL(dirty_l2, dim3(64), dim3(256), 0, side, big, nbig, 40); }
    for (int same = 1; same >= 0; --same) {
      float best = 1e9f;
      int herr = 0, bad = 0;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(flags, 0, nwg * 32 * sizeof(unsigned)));
        CK(hipMemset(err, 0, 2 * sizeof(int)));
        CK(hipEventRecord(e0));
        if (same) hipLaunchKernelGGL(probe<true>, dim3(nwg), dim3(512), 0, 0, xch, flags, err, out, iters, c, nwg / c);
        else hipLaunchKernelGGL(probe<false>, dim3(nwg), dim3(512), 0, 0, xch, flags, err, out, iters, c, nwg / c);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        int h2[2];
        CK(hipMemcpy(h2, err, 2 * sizeof(int), hipMemcpyDeviceToHost));
        herr |= h2[0];
        bad += h2[1];
      }
      printf("cluster of %d, %s: %.2f us per exchange%s\n", c, same ? "same XCD (stride layout)" : "adjacent blocks", best * 1e3f / iters,
             herr ? "  [SPIN CAP HIT]" : "");
      printf("    stale / wrong elements read: %d of %d\n", bad, 5 * nwg * 512 * iters);
    }
  }
  CK(hipDeviceSynchronize());
  return 0;
}
