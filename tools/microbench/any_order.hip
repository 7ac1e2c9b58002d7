// any_order.hip -- does hipExtAnyOrderLaunch let the SECOND kernel of a stream start while the first one still runs on gfx950?
// K1 (one workgroup) spins until K2 raises a flag (or 20 ms pass); K2 is launched behind it on the same stream, once with flags = 0
// and once with hipExtAnyOrderLaunch.  If the second launch overlaps, K1 sees the flag after microseconds; if not, it times out.
//   hipcc --offload-arch=gfx950 -O3 -o any_order any_order.hip && ./any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void k1(unsigned *flag, unsigned long long *out) {
  const unsigned long long t0 = wall_clock64();
  unsigned long long t = t0;
  unsigned seen = 0;
  while ((t = wall_clock64()) - t0 < 2000000ull) {  // 20 ms of the 100 MHz clock
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { seen = 1; break; }
    __builtin_amdgcn_s_sleep(8);
  }
  out[0] = t - t0;
  out[1] = seen;
}
__global__ void k2(unsigned *flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

int main() {
  unsigned *flag;
  unsigned long long *out, h[2];
  hipMalloc(&flag, 4);
  hipMalloc(&out, 16);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int pass = 0; pass < 4; pass++) {
    const unsigned fl = (pass & 1) ? hipExtAnyOrderLaunch : 0u;
    hipMemsetAsync(flag, 0, 4, s);
    hipStreamSynchronize(s);
    hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, s, flag, out);
    hipExtLaunchKernelGGL(k2, dim3(1), dim3(64), 0, s, nullptr, nullptr, fl, flag);
    hipStreamSynchronize(s);
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("second launch flags=%u: first kernel ran %.1f us, saw the flag: %llu  -> %s\n", fl, h[0] / 100.0, h[1],
           h[1] ? "the two kernels OVERLAPPED" : "serialised (first kernel timed out)");
  }
  return 0;
}
