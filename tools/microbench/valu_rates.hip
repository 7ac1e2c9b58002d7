// valu_rates.hip -- issue cost of the VALU instruction classes the SQ counters distinguish (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 / _F32,
// _INT32, _INT64, _CVT, the rest), in units of v_fma_f32's (the 4-clock full-rate instruction), on gfx950.  bench.py's back-end
// roofline prices the counted instruction mix of a launch with these weights ("replace all VALU at the fp64 rate").
// One 512-thread workgroup per CU (2 waves per SIMD), eight independent chains per thread: issue-bound, not latency-bound.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define OPS(X)                                                                          \
  X(0, "v_fma_f32", "v_fma_f32 %0, %1, %1, %0", "+v"(a[q]), "v"(a[(q + 1) & 7]))            \
  X(1, "v_add_f32", "v_add_f32 %0, %1, %0", "+v"(a[q]), "v"(a[(q + 1) & 7]))                \
  X(2, "v_mul_f32", "v_mul_f32 %0, %1, %0", "+v"(a[q]), "v"(a[(q + 1) & 7]))                \
  X(3, "v_rcp_f32 (TRANS_F32)", "v_rcp_f32 %0, %1", "=v"(a[q]), "v"(a[(q + 1) & 7]))          \
  X(4, "v_fma_f64", "v_fma_f64 %0, %1, %1, %0", "+v"(d[q]), "v"(d[(q + 1) & 7]))            \
  X(5, "v_add_f64", "v_add_f64 %0, %1, %0", "+v"(d[q]), "v"(d[(q + 1) & 7]))                \
  X(6, "v_mul_f64", "v_mul_f64 %0, %1, %0", "+v"(d[q]), "v"(d[(q + 1) & 7]))                \
  X(7, "v_rcp_f64 (TRANS_F64)", "v_rcp_f64 %0, %1", "=v"(d[q]), "v"(d[(q + 1) & 7]))          \
  X(8, "v_rsq_f64 (TRANS_F64)", "v_rsq_f64 %0, %1", "=v"(d[q]), "v"(d[(q + 1) & 7]))          \
  X(9, "v_cvt_f64_f32 (CVT)", "v_cvt_f64_f32 %0, %1", "=v"(d[q]), "v"(a[q]))                   \
  X(10, "v_cvt_f32_f64 (CVT)", "v_cvt_f32_f64 %0, %1", "=v"(a[q]), "v"(d[q]))                  \
  X(11, "v_cvt_i32_f64 (CVT)", "v_cvt_i32_f64 %0, %1", "=v"(i[q]), "v"(d[q]))                  \
  X(12, "v_add_u32 (INT32)", "v_add_u32 %0, %1, %0", "+v"(i[q]), "v"(i[(q + 1) & 7]))          \
  X(13, "v_mul_lo_u32 (INT32)", "v_mul_lo_u32 %0, %1, %0", "+v"(i[q]), "v"(i[(q + 1) & 7]))    \
  X(14, "v_lshlrev_b64 (INT64)", "v_lshlrev_b64 %0, 3, %1", "=v"(l[q]), "v"(l[(q + 1) & 7]))   \
  X(15, "v_mov_b32 (other)", "v_mov_b32 %0, %1", "=v"(i[q]), "v"(i[(q + 1) & 7]))              \
  X(16, "v_cndmask_b32 (other)", "v_cndmask_b32 %0, %1, %0, vcc", "+v"(i[q]), "v"(i[(q + 1) & 7])) \
  X(17, "v_cmp_lt_f64 (other)", "v_cmp_lt_f64 vcc, %0, %1", "+v"(d[q]), "v"(d[(q + 1) & 7]))

constexpr int kOps = 18;

template <int OP>
__global__ __launch_bounds__(512) void k(float *out, float seed, unsigned long long *clk) {
  float a[8];
  double d[8];
  int i[8];
  long long l[8];
  for (int q = 0; q < 8; q++) { a[q] = seed + q + threadIdx.x; d[q] = seed * q + 1.0; i[q] = q + threadIdx.x; l[q] = q; }
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int it = 0; it < 512; it++) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
#define X(ID, NAME, ASM, O, I) \
      if (OP == ID) { asm volatile(ASM : O : I : "vcc"); }
      OPS(X)
#undef X
    }
  }
  const unsigned long long t1 = clock64();
  float s = 0;
  for (int q = 0; q < 8; q++) s += a[q] + (float)d[q] + (float)i[q] + (float)l[q];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[OP] = t1 - t0;
}

template <int OP>
void run_all(float *out, unsigned long long *clk) {
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(512), 0, 0, out, 1.f, clk);
  if constexpr (OP + 1 < kOps) run_all<OP + 1>(out, clk);
}

int main() {
  float *out;
  unsigned long long *clk, h[kOps];
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&clk, kOps * 8);
  run_all<0>(out, clk);  // warm
  run_all<0>(out, clk);
  hipDeviceSynchronize();
  hipMemcpy(h, clk, kOps * 8, hipMemcpyDeviceToHost);
  const char *names[kOps] = {
#define X(ID, NAME, ASM, O, I) NAME,
      OPS(X)
#undef X
  };
  const double unit = (double)h[0];
  for (int n = 0; n < kOps; n++)
    printf("%-26s %6.2f counter ticks per wave-instruction per SIMD   = %5.2f x v_fma_f32 = %5.1f clocks\n", names[n],
           (double)h[n] / (2 * 512 * 8), (double)h[n] / unit, 4.0 * (double)h[n] / unit);
  return 0;
}
