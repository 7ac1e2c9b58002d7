// kernarg_preload.hip -- does preloading kernel arguments into SGPRs (gfx950: amdhsa_user_sgpr_kernarg_preload_length, clang
// -mllvm -amdgpu-kernarg-preload-count=N) shorten a latency-bound launch?  Three forms of the same tiny kernel (one dependent
// load per lane, one store), ~1000 workgroups like the evaluator's per-event launches:
//   A: arguments in one struct passed by value (how cmx_kernels.hip passes them: a byref kernarg, never preloaded)
//   B: the same values as separate scalar / pointer arguments (preloaded when the file is built with the option)
// build both ways:  hipcc --offload-arch=gfx950 -O3 kernarg_preload.hip -o kp_off
//                   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=8 kernarg_preload.hip -o kp_on
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
struct Args { const double *in; double *out; int n; int pad; double scale; };
__global__ __launch_bounds__(256) void k_struct(Args a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.n) a.out[i] = a.in[i] * a.scale;
}
__global__ __launch_bounds__(256) void k_scalar(const double *in, double *out, int n, double scale) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i] * scale;
}
int main() {
  const int n = 1000 * 256;
  double *in, *out;
  hipMalloc(&in, n * 8); hipMalloc(&out, n * 8); hipMemset(in, 0, n * 8);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  Args a{in, out, n, 0, 2.0};
  for (int form = 0; form < 2; form++) {
    for (int rep = 0; rep < 3; rep++) {
      // (1) chain of 2000 dependent launches: time per launch = boundary + duration
      for (int w = 0; w < 100; w++) { if (form == 0) hipLaunchKernelGGL(k_struct, dim3(1000), dim3(256), 0, s, a); else hipLaunchKernelGGL(k_scalar, dim3(1000), dim3(256), 0, s, in, out, n, 2.0); }
      hipStreamSynchronize(s);
      hipEventRecord(e0, s);
      for (int w = 0; w < 2000; w++) { if (form == 0) hipLaunchKernelGGL(k_struct, dim3(1000), dim3(256), 0, s, a); else hipLaunchKernelGGL(k_scalar, dim3(1000), dim3(256), 0, s, in, out, n, 2.0); }
      hipEventRecord(e1, s); hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // (2) kernel-exact duration of single launches
      double dur = 0; int cnt = 0;
      for (int w = 0; w < 200; w++) {
        if (form == 0) hipExtLaunchKernelGGL(k_struct, dim3(1000), dim3(256), 0, s, e0, e1, 0, a); else hipExtLaunchKernelGGL(k_scalar, dim3(1000), dim3(256), 0, s, e0, e1, 0, in, out, n, 2.0);
        hipStreamSynchronize(s);
        float d; if (hipEventElapsedTime(&d, e0, e1) == hipSuccess) { dur += d; cnt++; }
      }
      printf("%s  chain %.3f us per launch   single-launch duration %.3f us\n", form == 0 ? "struct by value" : "scalar arguments", ms * 1e3 / 2000, dur * 1e3 / cnt);
    }
  }
  return 0;
}
