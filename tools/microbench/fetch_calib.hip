// fetch_calib.hip -- known-byte-count streaming reads in the access widths the per-event kernels use, for calibrating
// rocprofv3's FETCH_SIZE on gfx950 (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
// streaming read; other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   read_u32   : 4 B / lane, coalesced            (packed events, batch indices)
//   read_f64   : 8 B / lane                        (dt stream)
//   read_16B   : 16 B / lane                       (bearing stream)
//   read_mix_be_splat : 4 + 4 + 16 B / lane        (the three streams of be_splat_lds: sxy, sbatch, sb)
//   read_mix_fe       : 16 + 8 B / lane            (the two streams of fe_splat_lds / fe_gather: sb, sdt)
// Each kernel reads N elements once (grid-stride over a buffer far larger than the 256 MB Infinity Cache is NOT used: the
// buffers are 20-80 MB and cold for the first launch only -- run every kernel several times and read min / max).
// build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ;  run under rocprofv3 --pmc FETCH_SIZE
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void read_u32(const uint32_t *a, int n, uint32_t *out) {
  uint32_t s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s ^= a[i];
  if (s == 0x12345678u) out[0] = s;
}
__global__ void read_f64(const double *a, int n, double *out) {
  double s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += a[i];
  if (s == 1.2345e300) out[0] = s;
}
__global__ void read_16B(const double2 *a, int n, double *out) {
  double s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const double2 v = a[i]; s += v.x + v.y; }
  if (s == 1.2345e300) out[0] = s;
}
__global__ void read_mix_be_splat(const uint32_t *xy, const uint32_t *bi, const double2 *b, int n, double *out) {
  double s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double2 v = b[i];
    s += v.x + v.y + (double)(xy[i] ^ bi[i]);
  }
  if (s == 1.2345e300) out[0] = s;
}
__global__ void read_mix_fe(const double2 *b, const double *dt, int n, double *out) {
  double s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const double2 v = b[i]; s += v.x + v.y + dt[i]; }
  if (s == 1.2345e300) out[0] = s;
}

int main() {
  const int n = 5000000;
  uint32_t *xy, *bi;
  double *d8, *out;
  double2 *d16;
  hipMalloc(&xy, (size_t)n * 4); hipMalloc(&bi, (size_t)n * 4); hipMalloc(&d8, (size_t)n * 8); hipMalloc(&d16, (size_t)n * 16);
  hipMalloc(&out, 64);
  hipMemset(xy, 1, (size_t)n * 4); hipMemset(bi, 2, (size_t)n * 4); hipMemset(d8, 0, (size_t)n * 8); hipMemset(d16, 0, (size_t)n * 16);
  const dim3 g(4096), b(256);
  for (int rep = 0; rep < 5; rep++) {
    hipLaunchKernelGGL(read_u32, g, b, 0, 0, xy, n, (uint32_t *)out);
    hipLaunchKernelGGL(read_f64, g, b, 0, 0, d8, n, out);
    hipLaunchKernelGGL(read_16B, g, b, 0, 0, d16, n, out);
    hipLaunchKernelGGL(read_mix_be_splat, g, b, 0, 0, xy, bi, d16, n, out);
    hipLaunchKernelGGL(read_mix_fe, g, b, 0, 0, d16, d8, n, out);
  }
  hipDeviceSynchronize();
  printf("known bytes per launch (n = %d): read_u32 %.1f KiB, read_f64 %.1f KiB, read_16B %.1f KiB, read_mix_be_splat %.1f KiB, read_mix_fe %.1f KiB\n",
         n, n * 4 / 1024.0, n * 8 / 1024.0, n * 16 / 1024.0, n * 24 / 1024.0, n * 24 / 1024.0);
  return 0;
}
