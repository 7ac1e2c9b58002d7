// graph_vs_stream.hip -- does a hipGraph shorten a chain of four dependent ~10 us kernels followed by a host wait
// (the shape of one front-end evaluation)?  Build: hipcc --offload-arch=gfx950 -O2 -o graph_vs_stream graph_vs_stream.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void work(float *p, int n, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = p[i];
  for (int k = 0; k < iters; k++) v = v * 1.0001f + 0.5f;
  p[i] = v;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const int n = 300 * 1024, iters = 350, reps = 4000;
  float *d;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemset(d, 0, n * sizeof(float)));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto chain = [&]() {
    for (int k = 0; k < 4; k++) hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, s, d, n, iters);
  };
  for (int w = 0; w < 50; w++) { chain(); CK(hipStreamSynchronize(s)); }
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) { chain(); CK(hipStreamSynchronize(s)); }
  const double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;

  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  chain();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 50; w++) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
  const double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;

  // one kernel alone, for the per-kernel time
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) { hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, s, d, n, iters); CK(hipStreamSynchronize(s)); }
  const double us_one = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  printf("one kernel + sync: %.1f us; chain of 4 + sync: stream launches %.1f us, hipGraphLaunch %.1f us\n", us_one, us_stream, us_graph);
  return 0;
}
