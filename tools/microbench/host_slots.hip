// host_slots.hip -- can the last kernel of an evaluation hand its per-workgroup partial sums to the HOST instead of running the
// last-arriver protocol + finalize on the device (round 5 question)?  Per launch of G workgroups x 256 threads, each doing the same
// small amount of streaming work (`work` 24-byte records per thread) and producing 8 doubles:
//   (a) tail   : what fe_gather_kernel does today -- fp64 atomics into 8 accumulator rows, write-through drain, sharded arrival tickets,
//                the last arriver loads the rows, sums them and stores 8 doubles + checksum + ticket to mapped host memory;
//                the host spins on the ticket.
//   (b) slots  : every workgroup stores ONE stamped, checksummed 128-byte record (16 lanes x 8 B, contiguous) to its own slot in
//                mapped host memory; the host walks the slots in index order, adding each as soon as its stamp and checksum verify
//                (a fixed summation order whatever the arrival order) -- no device-side dependency between workgroups at all.
// Reported per G: host wall time launch -> result on the host (mean of `reps`, launches back to back, each waited for) and the
// kernel's own duration (hipExt start / stop events on a separate pass).
// build: hipcc --offload-arch=gfx950 -O3 -o host_slots host_slots.hip       run: ./host_slots [reps] [work]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr unsigned long long kMix = 0x9E3779B97F4A7C15ull;
constexpr int kShards = 8, kStride = 32;

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double ld_sc1(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the work: every thread streams `work` records of 3 doubles and folds them into 8 sums
__device__ __forceinline__ void do_work(const double *src, size_t n, int work, double acc[8]) {
  const size_t per_block = (size_t)work * 256;
  const size_t beg = (size_t)blockIdx.x * per_block;
  for (int k = 0; k < work; k++) {
    const size_t i = (beg + (size_t)k * 256 + threadIdx.x) % n;
    const double a = src[3 * i], b = src[3 * i + 1], c = src[3 * i + 2];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] += a * (j + 1) + b * c;
  }
}
__device__ __forceinline__ void block_reduce(double acc[8], double red[4][8], double out[8]) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const double w = wave_sum(acc[j]);
    if (lane == 0) red[wave][j] = w;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; j++) out[j] = red[0][j] + red[1][j] + red[2][j] + red[3][j];
}

__global__ __launch_bounds__(256) void tail_kernel(const double *src, size_t n, int work, double *gacc, unsigned *counters, double *h_result,
                                                   unsigned long long ticket) {
  __shared__ double red[4][8];
  __shared__ int is_last;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, out[8];
  do_work(src, n, work, acc);
  block_reduce(acc, red, out);
  if (threadIdx.x < 8)
    __hip_atomic_fetch_add(gacc + (size_t)(blockIdx.x % kShards) * 16 + threadIdx.x, out[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nblocks = gridDim.x, shard = blockIdx.x % kShards;
    const int nshards = nblocks < kShards ? nblocks : kShards;
    const unsigned shard_size = (unsigned)((nblocks - shard + kShards - 1) / kShards);
    unsigned *cs = counters + shard * kStride, *ct = counters + kShards * kStride;
    int last = 0;
    if (atomicAdd(cs, 1u) == shard_size - 1u) {
      __hip_atomic_store(cs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (atomicAdd(ct, 1u) == (unsigned)nshards - 1u) {
        __hip_atomic_store(ct, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1;
      }
    }
    is_last = last;
  }
  __syncthreads();
  if (!is_last) return;
  __shared__ unsigned long long chk;
  if (threadIdx.x == 0) chk = 0ull;
  __syncthreads();
  if (threadIdx.x < 8) {
    double w = 0;
    for (int q = 0; q < kShards; q++) w += ld_sc1(gacc + (size_t)q * 16 + threadIdx.x);
    for (int q = 0; q < kShards; q++) st_sc1(gacc + (size_t)q * 16 + threadIdx.x, 0.0);
    h_result[threadIdx.x] = w;
    atomicXor(&chk, (unsigned long long)__double_as_longlong(w));
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long *s = reinterpret_cast<unsigned long long *>(h_result);
    s[8] = chk ^ (ticket * kMix);
    s[9] = ticket;
  }
}

__global__ __launch_bounds__(256) void slots_kernel(const double *src, size_t n, int work, unsigned long long *h_slots, unsigned long long ticket) {
  __shared__ double red[4][8];
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, out[8];
  do_work(src, n, work, acc);
  block_reduce(acc, red, out);
  if (threadIdx.x < 16) {  // ONE contiguous 128-byte store by 16 lanes of wave 0: 8 values, stamp, checksum, padding
    unsigned long long w = 0ull;
    if (threadIdx.x < 8) w = (unsigned long long)__double_as_longlong(out[threadIdx.x]);
    unsigned long long x = w;
    for (int o = 8; o > 0; o >>= 1) x ^= __shfl_xor(x, o, 16);  // xor of the 16 lanes' words (lanes 8..15 hold 0)
    if (threadIdx.x == 8) w = ticket;
    if (threadIdx.x == 9) w = x ^ (ticket * kMix);
    h_slots[(size_t)blockIdx.x * 16 + threadIdx.x] = w;
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 2000, work = argc > 2 ? atoi(argv[2]) : 4;
  const size_t n = 1u << 20;
  double *d_src, *d_gacc, *h_result;
  unsigned *d_counters;
  unsigned long long *h_slots;
  CK(hipMalloc(&d_src, n * 3 * sizeof(double)));
  {
    double *h = (double *)malloc(n * 3 * sizeof(double));
    for (size_t i = 0; i < 3 * n; i++) h[i] = (double)((i * 2654435761u) % 1000) * 1e-3;
    CK(hipMemcpy(d_src, h, n * 3 * sizeof(double), hipMemcpyHostToDevice));
    free(h);
  }
  CK(hipMalloc(&d_gacc, kShards * 16 * sizeof(double)));
  CK(hipMemset(d_gacc, 0, kShards * 16 * sizeof(double)));
  CK(hipMalloc(&d_counters, (kShards + 1) * kStride * sizeof(unsigned)));
  CK(hipMemset(d_counters, 0, (kShards + 1) * kStride * sizeof(unsigned)));
  CK(hipHostMalloc((void **)&h_result, 4096, hipHostMallocMapped));
  CK(hipHostMalloc((void **)&h_slots, 4096 * 128, hipHostMallocMapped));
  memset(h_result, 0, 4096);
  memset(h_slots, 0, 4096 * 128);
  double *dh_result;
  unsigned long long *dh_slots;
  CK(hipHostGetDevicePointer((void **)&dh_result, h_result, 0));
  CK(hipHostGetDevicePointer((void **)&dh_slots, h_slots, 0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("reps %d, work %d records per thread (G x 256 threads); us: host wall launch->result | kernel duration\n", reps, work);
  unsigned long long ticket = 0;
  for (int G : {245, 489, 977, 1954}) {
    double sums_tail[8] = {0}, sums_slots[8] = {0};
    for (int mode = 0; mode < 2; mode++) {
      double wall = 0, kern = 0;
      for (int pass = 0; pass < 2; pass++) {  // pass 0: wall clock, untimed launches; pass 1: kernel duration through its own events
        const int R = pass == 0 ? reps : reps / 10;
        for (int it = -20; it < R; it++) {
          ++ticket;
          const double t0 = now_us();
          if (mode == 0) {
            if (pass == 0) hipLaunchKernelGGL(tail_kernel, dim3(G), dim3(256), 0, s, d_src, n, work, d_gacc, d_counters, dh_result, ticket);
            else hipExtLaunchKernelGGL(tail_kernel, dim3(G), dim3(256), 0, s, e0, e1, 0, d_src, n, work, d_gacc, d_counters, dh_result, ticket);
            const volatile unsigned long long *w = reinterpret_cast<const volatile unsigned long long *>(h_result);
            for (;;) {
              if (w[9] == ticket) {
                unsigned long long x = 0;
                for (int j = 0; j < 8; j++) x ^= w[j];
                if ((x ^ (ticket * kMix)) == w[8]) break;
              }
              __builtin_ia32_pause();
            }
            for (int j = 0; j < 8; j++) sums_tail[j] = h_result[j];
          } else {
            if (pass == 0) hipLaunchKernelGGL(slots_kernel, dim3(G), dim3(256), 0, s, d_src, n, work, dh_slots, ticket);
            else hipExtLaunchKernelGGL(slots_kernel, dim3(G), dim3(256), 0, s, e0, e1, 0, d_src, n, work, dh_slots, ticket);
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const volatile unsigned long long *w = h_slots;
            for (int b = 0; b < G;) {  // index order: a fixed summation order whatever the arrival order
              const volatile unsigned long long *q = w + (size_t)b * 16;
              if (q[8] == ticket) {
                unsigned long long v[8], x = 0;
                for (int j = 0; j < 8; j++) { v[j] = q[j]; x ^= v[j]; }
                if ((x ^ (ticket * kMix)) == q[9]) {
                  for (int j = 0; j < 8; j++) { double d; memcpy(&d, &v[j], 8); acc[j] += d; }
                  b++;
                  continue;
                }
              }
              __builtin_ia32_pause();
            }
            for (int j = 0; j < 8; j++) sums_slots[j] = acc[j];
          }
          const double t1 = now_us();
          if (it >= 0 && pass == 0) wall += t1 - t0;
          if (pass == 1) {
            CK(hipStreamSynchronize(s));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 0) kern += ms * 1e3;
          }
        }
      }
      printf("G %4d %-5s: wall %7.2f | kernel %7.2f\n", G, mode == 0 ? "tail" : "slots", wall / reps, kern / (reps / 10));
    }
    double worst = 0;
    for (int j = 0; j < 8; j++) {
      const double d = sums_tail[j] - sums_slots[j], r = d / (sums_tail[j] != 0 ? sums_tail[j] : 1);
      worst = (r < 0 ? -r : r) > worst ? (r < 0 ? -r : r) : worst;
    }
    printf("        sums agree to %.1e relative\n", worst);
  }
  return 0;
}
