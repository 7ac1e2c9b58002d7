// grid_barrier.hip -- what one evaluation would pay for being ONE cooperative launch (splat -> grid barrier -> image pass ->
// grid barrier -> gather) instead of three launches.  Three phases of the evaluation's shape:
//   phase 1: every workgroup streams its share of `n` 24-byte records and adds to a 1.2 MB plane with atomics (the splat)
//   phase 2: the first 300 workgroups read the plane and write a second plane (the image pass)
//   phase 3: every workgroup streams its share of the records again and gathers from the second plane (the gather)
// run (a) as three kernels on one stream, (b) as one kernel with two grid barriers (per-XCD-sharded arrival counters, one
// top counter, generation words per shard polled with sc1 loads -- the hierarchical form MI355X_MICROARCH.md prices as
// "barrier-xcd"; a first version in which every workgroup polled ONE generation word took 7.5 us per barrier at 64 workgroups
// and ~95 us at 813: the pollers queue on one memory-side address).
// Every spin is bounded: a barrier that does not complete within ~2 ms sets an abort flag and falls through (no hang).
// build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

struct Bar { unsigned shard[8 * 32]; unsigned top[32]; unsigned gen_top[32]; unsigned gen_shard[8 * 32]; unsigned abort_flag[32]; };

__device__ __forceinline__ unsigned ld_sc1(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool poll(const unsigned *p, unsigned want, Bar *b) {
  unsigned spins = 0;
  while (ld_sc1(p) != want) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > 2000000u) { b->abort_flag[0] = 1u; return false; }
  }
  return true;
}

// hierarchical: arrivals on 8 shard counters (blockIdx % 8 = the XCD of a workgroup), the last of a shard arrives on the top
// counter; the overall last publishes the generation, shard leaders poll THAT word (8 pollers) and re-publish it on their
// shard's own word, which the shard's other workgroups poll (~nblocks / 8 pollers per word)
__device__ __forceinline__ void grid_barrier(Bar *b, unsigned nblocks, unsigned &my_gen) {
  // every wave drains its stores, ONE lane issues the release fence (an L2 write-back: with a fence in every wave the barrier
  // cost 0.12 us PER WORKGROUP -- 97 us at 813 workgroups), every wave acquires afterwards
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned shard = blockIdx.x % 8, nshards = nblocks < 8 ? nblocks : 8;
    const unsigned shard_size = (nblocks - shard + 7) / 8;
    const unsigned want = my_gen + 1;
    if (atomicAdd(&b->shard[shard * 32], 1u) == shard_size - 1u) {
      __hip_atomic_store(&b->shard[shard * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (atomicAdd(&b->top[0], 1u) == nshards - 1u) {
        __hip_atomic_store(&b->top[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&b->gen_top[0], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        poll(&b->gen_top[0], want, b);
      }
      __hip_atomic_store(&b->gen_shard[shard * 32], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      poll(&b->gen_shard[shard * 32], want, b);
    }
    my_gen = want;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// Variant B (round 3, later): no release fence and no returning atomics.  What a phase hands to the next one is written with
// memory-side atomics or write-through (sc1) stores, so there is nothing in an L2 to write back; a workgroup ARRIVES with one
// fire-and-forget add on its shard's monotonic counter (never reset: no reset races), workgroup 0 alone polls the 8 counters
// until they add up, then publishes the generation on 8 words the others poll.
struct BarB { unsigned long long count[8 * 16]; unsigned gen[8 * 32]; unsigned abort_flag[32]; };
__device__ __forceinline__ void grid_barrier_b(BarB *b, unsigned nblocks, unsigned &my_gen) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < 64) {
    const unsigned want = my_gen + 1;
    if (threadIdx.x == 0)
      (void)__hip_atomic_fetch_add(&b->count[(blockIdx.x % 8) * 16], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0) {
      unsigned spins = 0;
      for (;;) {  // lanes 0..7 read the 8 counters, the wave adds them up
        unsigned long long v = threadIdx.x < 8 ? __hip_atomic_load(&b->count[threadIdx.x * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        for (int o = 4; o; o >>= 1) v += __shfl_down(v, o, 64);
        v = __shfl(v, 0, 64);
        if (v >= (unsigned long long)nblocks * want) break;
        if (++spins > 2000000u) { b->abort_flag[0] = 1u; break; }
      }
      if (threadIdx.x < 8) __hip_atomic_store(&b->gen[threadIdx.x * 32], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (threadIdx.x == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(&b->gen[(blockIdx.x % 8) * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000u) { b->abort_flag[0] = 1u; break; }
      }
    }
    my_gen = want;
  }
  __syncthreads();
}

__device__ __forceinline__ void phase_splat(const double *rec, int n, float *plane, int npix) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double a = rec[3 * (size_t)i], b = rec[3 * (size_t)i + 1], c = rec[3 * (size_t)i + 2];
    const int cell = (int)((unsigned)(i * 2654435761u) % (unsigned)npix);
    atomicAdd(&plane[cell], (float)(a + b + c));
  }
}
__device__ __forceinline__ void phase_image(const float *plane, float *out, int npix) {
  if (blockIdx.x >= 300) return;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += 300 * blockDim.x) {
    const float v = __builtin_nontemporal_load(plane + p);
    out[p] = 0.25f * v + 1.0f;
  }
}
__device__ __forceinline__ void phase_gather(const double *rec, int n, const float *img, int npix, double *acc) {
  double s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double a = rec[3 * (size_t)i], b = rec[3 * (size_t)i + 1], c = rec[3 * (size_t)i + 2];
    const int cell = (int)((unsigned)(i * 2654435761u) % (unsigned)npix);
    s += (a + b + c) * (double)img[cell];
  }
  if (s == 1.2345e300) acc[0] = s;
}

__device__ __forceinline__ void phase_image_sc1(const float *plane, float *out, int npix) {
  if (blockIdx.x >= 300) return;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += 300 * blockDim.x) {
    const float v = __hip_atomic_load(plane + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(out + p, 0.25f * v + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void phase_gather_sc1(const double *rec, int n, const float *img, int npix, double *acc) {
  double s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double a = rec[3 * (size_t)i], b = rec[3 * (size_t)i + 1], c = rec[3 * (size_t)i + 2];
    const int cell = (int)((unsigned)(i * 2654435761u) % (unsigned)npix);
    s += (a + b + c) * (double)__hip_atomic_load(img + cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (s == 1.2345e300) acc[0] = s;
}
__global__ __launch_bounds__(256) void k_fused_b(const double *rec, int n, float *plane, float *out, int npix, double *acc, BarB *bar, unsigned gen0) {
  unsigned g = gen0;
  phase_splat(rec, n, plane, npix);
  grid_barrier_b(bar, gridDim.x, g);
  phase_image_sc1(plane, out, npix);
  grid_barrier_b(bar, gridDim.x, g);
  phase_gather_sc1(rec, n, out, npix, acc);
}
__global__ __launch_bounds__(256) void k_splat(const double *rec, int n, float *plane, int npix) { phase_splat(rec, n, plane, npix); }
__global__ __launch_bounds__(256) void k_image(const float *plane, float *out, int npix) { phase_image(plane, out, npix); }
__global__ __launch_bounds__(256) void k_gather(const double *rec, int n, const float *img, int npix, double *acc) { phase_gather(rec, n, img, npix, acc); }
__global__ __launch_bounds__(256) void k_fused(const double *rec, int n, float *plane, float *out, int npix, double *acc, Bar *bar, unsigned gen0) {
  unsigned g = gen0;
  phase_splat(rec, n, plane, npix);
  grid_barrier(bar, gridDim.x, g);
  phase_image(plane, out, npix);
  grid_barrier(bar, gridDim.x, g);
  phase_gather(rec, n, out, npix, acc);
}

int main() {
  const int npix = 640 * 480;
  double *rec, *acc;
  float *plane, *out;
  Bar *bar;
  BarB *barb;
  hipMalloc(&rec, (size_t)1000000 * 24); hipMemset(rec, 0, (size_t)1000000 * 24);
  hipMalloc(&plane, npix * 4); hipMalloc(&out, npix * 4); hipMalloc(&acc, 64); hipMalloc(&bar, sizeof(Bar)); hipMalloc(&barb, sizeof(BarB)); hipMemset(barb, 0, sizeof(BarB));
  hipMemset(plane, 0, npix * 4); hipMemset(bar, 0, sizeof(Bar));
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 300;
  printf("%10s %6s | %12s %12s %12s | per evaluation, us (events around %d back-to-back evaluations)\n", "events", "WGs", "3 launches", "1 launch+2bar", "variant B", reps);
  const int cases[][2] = {{60000, 64}, {100000, 128}, {1000000, 256}, {1000000, 512}, {1000000, 813}, {1000000, 1024}};
  for (auto &cs : cases) {
    const int n = cs[0], wg = cs[1];
    float ms3 = 0, ms1 = 0, msb = 0;
    for (int pass = 0; pass < 2; pass++) {
      hipEventRecord(e0, s);
      for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL(k_splat, dim3(wg), dim3(256), 0, s, rec, n, plane, npix);
        hipLaunchKernelGGL(k_image, dim3(wg), dim3(256), 0, s, plane, out, npix);
        hipLaunchKernelGGL(k_gather, dim3(wg), dim3(256), 0, s, rec, n, out, npix, acc);
      }
      hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms3, e0, e1);
      hipEventRecord(e0, s);
      for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL(k_fused, dim3(wg), dim3(256), 0, s, rec, n, plane, out, npix, acc, bar, (unsigned)(r * 2));
      hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms1, e0, e1);
      hipEventRecord(e0, s);
      for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL(k_fused_b, dim3(wg), dim3(256), 0, s, rec, n, plane, out, npix, acc, barb, (unsigned)(r * 2));
      hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&msb, e0, e1);
      hipMemsetAsync(barb, 0, sizeof(BarB), s);
      // the generation counter must start where the kernel argument says: reset between cases
      hipMemsetAsync(bar, 0, sizeof(Bar), s);
    }
    unsigned ab = 0;
    hipMemcpy(&ab, &bar->abort_flag[0], 4, hipMemcpyDeviceToHost);
    unsigned abb = 0;
    hipMemcpy(&abb, &barb->abort_flag[0], 4, hipMemcpyDeviceToHost);
    printf("%10d %6d | %12.2f %12.2f %12.2f %s\n", n, wg, ms3 * 1e3 / reps, ms1 * 1e3 / reps, msb * 1e3 / reps, (ab || abb) ? "(a barrier timed out)" : "");
  }
  return 0;
}
