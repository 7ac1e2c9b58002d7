// microbenchmark: LDS atomic add throughput on gfx950 for f32 / u32 / u64 / f64, random addresses in a 4096-entry window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <typename T> __device__ __forceinline__ void lds_add(T* p, T v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <typename T, int SPREAD>
__global__ __launch_bounds__(256) void k(const uint32_t* idx, int iters, T* out) {
  __shared__ T win[4352];
  for (int p = threadIdx.x; p < 4352; p += 256) win[p] = 0;
  __syncthreads();
  const uint32_t* my = idx + (size_t)blockIdx.x * iters * 256;
  for (int i = 0; i < iters; i++) {
    uint32_t a = my[i * 256 + threadIdx.x];
    if (SPREAD == 0) a = a & 4095; else if (SPREAD == 1) a = (a & 63) * 64 + ((a >> 6) & 1); /* vertical edge */ else a = (a & 31);
    lds_add(&win[a], (T)1); lds_add(&win[a + 1], (T)2); lds_add(&win[a + 67], (T)3); lds_add(&win[a + 68], (T)4);
  }
  __syncthreads();
  T s = 0; for (int p = threadIdx.x; p < 4352; p += 256) s += win[p];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename T, int SPREAD> float run(const uint32_t* d_idx, int blocks, int iters, void* d_out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<T, SPREAD>), dim3(blocks), dim3(256), 0, 0, d_idx, iters, (T*)d_out);
  hipEventRecord(a); for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<T, SPREAD>), dim3(blocks), dim3(256), 0, 0, d_idx, iters, (T*)d_out);
  hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms / 5 * 1e3f;
}
int main() {
  const int blocks = 1024, iters = 16; size_t n = (size_t)blocks * iters * 256;
  std::vector<uint32_t> h(n); uint32_t s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 8; }
  uint32_t* d_idx; hipMalloc(&d_idx, n * 4); hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice);
  void* d_out; hipMalloc(&d_out, blocks * 256 * 8);
  double lane_ops = (double)n * 4;
  printf("lane-atomics per launch: %.1f M\n", lane_ops / 1e6);
#define R(T, S, name) { float us = run<T, S>(d_idx, blocks, iters, d_out); printf("%-28s %8.1f us  %7.1f G lane-atomics/s\n", name, us, lane_ops / us / 1e3); }
  R(float, 0, "f32 random"); R(unsigned, 0, "u32 random"); R(unsigned long long, 0, "u64 random"); R(double, 0, "f64 random");
  R(float, 1, "f32 vertical-edge"); R(unsigned, 1, "u32 vertical-edge"); R(unsigned long long, 1, "u64 vertical-edge");
  R(float, 2, "f32 32-addresses"); R(unsigned, 2, "u32 32-addresses"); R(unsigned long long, 2, "u64 32-addresses");
  return 0;
}
