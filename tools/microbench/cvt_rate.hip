// cvt_rate.hip -- issue rate of v_cvt_f64_f32 against v_fma_f64 / v_fma_f32 / ds_read_b64 on gfx950 (one workgroup of 8 waves per CU,
// the shape of the fused image pass).  Prints clocks per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o cvt_rate cvt_rate.hip && ./cvt_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(512) void k(float *out, float seed, unsigned long long *clk) {
  __shared__ double lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i * 0.5;
  __syncthreads();
  float a[8];
  double d[8];
  for (int q = 0; q < 8; q++) { a[q] = seed + q + threadIdx.x; d[q] = seed * q; }
  const unsigned long long t0 = clock64();
  for (int it = 0; it < 256; it++) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (OP == 0) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[q]) : "v"(a[q])); }
      if (OP == 1) { asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d[q]) : "v"(d[(q + 1) & 7])); }
      if (OP == 2) { asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[q]) : "v"(a[(q + 1) & 7])); }
      if (OP == 3) { asm volatile("ds_read_b64 %0, %1" : "=v"(d[q]) : "v"((threadIdx.x * 8 + q * 512 * 8) & 32767)); }
      if (OP == 4) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[q]) : "v"(d[q])); }
    }
    if (OP == 3) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const unsigned long long t1 = clock64();
  float s = 0;
  for (int q = 0; q < 8; q++) s += a[q] + (float)d[q];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[OP] = t1 - t0;
}

int main() {
  float *out;
  unsigned long long *clk, h[5];
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&clk, 5 * 8);
  hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, 1.f, clk);
  hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, 1.f, clk);
  hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, 1.f, clk);
  hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, 1.f, clk);
  hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, out, 1.f, clk);
  hipDeviceSynchronize();
  hipMemcpy(h, clk, 40, hipMemcpyDeviceToHost);
  const char *names[5] = {"v_cvt_f64_f32", "v_fma_f64", "v_fma_f32", "ds_read_b64", "v_cvt_f32_f64"};
  // per SIMD: 2 waves x 256 iterations x 8 instructions
  for (int i = 0; i < 5; i++) printf("%-14s %6.2f clocks per wave-instruction per SIMD (2 waves per SIMD)\n", names[i], (double)h[i] / (2 * 256 * 8));
  return 0;
}
