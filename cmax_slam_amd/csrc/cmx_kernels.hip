// cmx_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for cmax_slam's event-warping hot path.
//
//   fe_splat        K1  front-end warp + bilinear splat      (reference local_image_warped_events.cpp:94-169)
//   be_pose_table   K0  per-batch spline pose + Jacobian     (reference event_pano_warper.cpp:239-256, so3_spline.h:218-274)
//   be_splat        K2  back-end rotate + equirect + splat   (reference event_pano_warper.cpp:262-335,
//                                                             equirectangular_camera.h:18-45)
//   image_moments   K3+K5+K6  compose I = IL + alpha*IGp, separable Gaussian (REFLECT_101), moment reduction
//                                                            (reference event_pano_warper.cpp:199-230,
//                                                             local_focus_funcs.cpp:9-44, global_focus_funcs.cpp:11-47)
//   alpha_*         K4  event-density ratio alpha            (reference event_pano_warper.cpp:134-165)
//   reduce/finalize     partial moments -> contrast, gradient (fp64)
//
// Numerics: geometry in fp64 exactly as the reference, weights/accumulators fp32.  This file is compiled with
// -ffp-contract=off so the fp64 warp, the fp32 weights and the fp32 blur are bit-identical to the CPU path for
// identical inputs; the only reordering is the fp32 atomic accumulation.
#include "cmx_internal.hpp"

namespace cmx {

// fire-and-forget fp32 atomic add (global_atomic_add_f32, no return value); needs -munsafe-fp-atomics
__device__ __forceinline__ void atomic_add_f32(float *p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------- K1
template <bool DERIV>
__global__ __launch_bounds__(256) void fe_splat_kernel(FeSplatArgs a) {
  const int W = a.W, H = a.H;
  const size_t np = (size_t)W * H;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n; i += gridDim.x * 256) {
    const uint32_t e = a.xy[i];
    const int ex = e & 0xffff, ey = (e >> 16) & 0x7fff;
    const double dt = a.batch_dt[i / a.per_batch];
    const double *b = a.lut + 3 * ((size_t)ey * W + ex);
    const double px = b[0], py = b[1], pz = b[2];
    // p' = p + (omega*dt) x p   (first-order rotation)
    const double drx = a.wx * dt, dry = a.wy * dt, drz = a.wz * dt;
    const double rx = px + (dry * pz - drz * py);
    const double ry = py + (drz * px - drx * pz);
    const double rz = pz + (drx * py - dry * px);
    const double iz = 1.0 / rz;
    const double cxn = rx * iz, cyn = ry * iz;
    const double u = a.fx * cxn + a.cx;
    const double v = a.fy * cyn + a.cy;
    const int xx = (int)u, yy = (int)v;
    if (1 <= xx && xx < W - 2 && 1 <= yy && yy < H - 2) {
      const float dx = (float)(u - xx), dy = (float)(v - yy);
      float *q = a.planes + (size_t)yy * W + xx;
      atomic_add_f32(q, (1.f - dx) * (1.f - dy));
      atomic_add_f32(q + 1, dx * (1.f - dy));
      atomic_add_f32(q + W, (1.f - dx) * dy);
      atomic_add_f32(q + W + 1, dx * dy);
      if (DERIV) {
        // J = diag(fx,fy) * J_proj(2x3) * [(-dt) p]_x   evaluated in the reference's operation order
        const double vx = (-dt) * px, vy = (-dt) * py, vz = (-dt) * pz;
        const double a02 = -cxn * iz, a12 = -cyn * iz;
        double c[6];
        c[0] = a02 * (-vy);
        c[1] = iz * (-vz) + a02 * vx;
        c[2] = iz * vy;
        c[3] = iz * vz + a12 * (-vy);
        c[4] = a12 * vx;
        c[5] = iz * (-vx);
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const float r0 = (float)(a.fx * c[k]), r1 = (float)(a.fy * c[3 + k]);
          float *d = q + (size_t)(1 + k) * np;
          atomic_add_f32(d, r0 * (-(1.f - dy)) + r1 * (-(1.f - dx)));
          atomic_add_f32(d + 1, r0 * (1.f - dy) + r1 * (-dx));
          atomic_add_f32(d + W, r0 * (-dy) + r1 * (1.f - dx));
          atomic_add_f32(d + W + 1, r0 * dy + r1 * dx);
        }
      }
    }
  }
}

static int splat_grid(int n) {
  int blocks = (n + 255) / 256;
  const int cap = 256 * 8;  // 256 CUs x 8 blocks; grid-stride beyond
  return blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
}

void launch_fe_splat(const FeSplatArgs &a, bool deriv, hipStream_t s) {
  if (a.n <= 0) return;
  if (deriv) hipLaunchKernelGGL(fe_splat_kernel<true>, dim3(splat_grid(a.n)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(fe_splat_kernel<false>, dim3(splat_grid(a.n)), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------------- K0
template <int N, bool WANT_J>
__global__ __launch_bounds__(64) void be_pose_table_kernel(const SplineArgs *sp, const long long *batch_t, int nb,
                                                           PoseEntry *out) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= nb) return;
  Mat3 R, J[N];
  int idx;
  spline_eval<N, WANT_J>(*sp, batch_t[b], R, J, idx);
  PoseEntry &o = out[b];
#pragma unroll
  for (int i = 0; i < 9; i++) o.R[i] = R.m[i];
  o.idx_cp_beg = idx;
  if (WANT_J) {
    // 3 x 3N fp32, block k at columns 3k..3k+2  (Trajectory::evaluate copies d_val_d_knot[k] as float)
#pragma unroll
    for (int k = 0; k < N; k++)
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) o.Jcp[r * (3 * N) + 3 * k + c] = (float)J[k].m[r * 3 + c];
  }
}

void launch_be_pose_table(const SplineArgs *d_spline, const long long *d_batch_t, int nb, int order, bool want_j,
                          PoseEntry *out, hipStream_t s) {
  if (nb <= 0) return;
  const dim3 g((nb + 63) / 64), b(64);
  if (order == 2) {
    if (want_j) hipLaunchKernelGGL((be_pose_table_kernel<2, true>), g, b, 0, s, d_spline, d_batch_t, nb, out);
    else hipLaunchKernelGGL((be_pose_table_kernel<2, false>), g, b, 0, s, d_spline, d_batch_t, nb, out);
  } else {
    if (want_j) hipLaunchKernelGGL((be_pose_table_kernel<4, true>), g, b, 0, s, d_spline, d_batch_t, nb, out);
    else hipLaunchKernelGGL((be_pose_table_kernel<4, false>), g, b, 0, s, d_spline, d_batch_t, nb, out);
  }
}

// ---------------------------------------------------------------------------------------------- K2
template <int N, bool DERIV>
__global__ __launch_bounds__(256) void be_splat_kernel(BeSplatArgs a) {
  const int Wp = a.Wp, Hp = a.Hp;
  const size_t np = (size_t)Wp * Hp;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n; i += gridDim.x * 256) {
    const uint32_t e = a.xy[i];
    const int ex = e & 0xffff, ey = (e >> 16) & 0x7fff;
    const bool is_old = (e >> 31) != 0;
    const PoseEntry &pe = a.poses[i / a.per_batch];
    const double *b = a.lut + 3 * ((size_t)ey * a.W + ex);
    const double b0 = b[0], b1 = b[1], b2 = b[2];
    // e_ray_w = R * bearing
    const double x = pe.R[0] * b0 + pe.R[1] * b1 + pe.R[2] * b2;
    const double y = pe.R[3] * b0 + pe.R[4] * b1 + pe.R[5] * b2;
    const double z = pe.R[6] * b0 + pe.R[7] * b1 + pe.R[8] * b2;
    // equirectangular projection
    const double phi = atan2(x, z);
    const double rho = sqrt(x * x + y * y + z * z);
    const double theta = asin(y / rho);
    const double pxm = a.cxp + phi * a.fx;
    const double pym = a.cyp + theta * a.fy;
    const int xx = (int)pxm, yy = (int)pym;
    if (1 <= xx && xx < Wp - 2 && 1 <= yy && yy < Hp - 2) {
      const float dx = (float)(pxm - xx), dy = (float)(pym - yy);
      const size_t off = (size_t)yy * Wp + xx;
      float *q = a.planes + (is_old ? 0 : np) + off;
      atomic_add_f32(q, (1.f - dx) * (1.f - dy));
      atomic_add_f32(q + 1, dx * (1.f - dy));
      atomic_add_f32(q + Wp, (1.f - dx) * dy);
      atomic_add_f32(q + Wp + 1, dx * dy);
      if (DERIV) {
        const double Ydivrho = y / rho;
        const double XdivZ = x / z;
        const double tmp1 = a.fx / ((1 + XdivZ * XdivZ) * z);
        const double tmp2 = -a.fy / sqrt(1 - Ydivrho * Ydivrho);
        const double tmp3 = Ydivrho / (rho * rho);
        const float d00 = (float)tmp1, d02 = (float)(-tmp1 * XdivZ);
        const float d10 = (float)(tmp2 * tmp3 * x), d11 = (float)(tmp2 * (tmp3 * y - 1 / rho)),
                    d12 = (float)(tmp2 * tmp3 * z);
        const float rbx = (float)x, rby = (float)y, rbz = (float)z;
        // dpm_ddrot = dpm_drb(2x3) * (-[rb]x)(3x3), fp32 accumulation in k order (d01 == 0)
        float m[6];
        m[0] = 0.f * (-rbz) + d02 * rby;  // d00*0 + d01*(-rb.z) + d02*rb.y
        m[1] = d00 * rbz + d02 * (-rbx);
        m[2] = d00 * (-rby) + 0.f * rbx;
        m[3] = d11 * (-rbz) + d12 * rby;
        m[4] = d10 * rbz + d12 * (-rbx);
        m[5] = d10 * (-rby) + d11 * rbx;
        const int jbase = 3 * (pe.idx_cp_beg - a.num_fixed);
        const float w00a = -(1.f - dy), w00b = -(1.f - dx);
#pragma unroll
        for (int c = 0; c < 3 * N; c++) {
          const int j = jbase + c;
          if (j >= 0) {
            // jac = dpm_ddrot(2x3) * ddrot_ddrot_cp(3x3N): fp64 accumulation, fp32 result
            const double j0 = (double)pe.Jcp[c], j1 = (double)pe.Jcp[3 * N + c], j2 = (double)pe.Jcp[6 * N + c];
            const float r0 = (float)((double)m[0] * j0 + (double)m[1] * j1 + (double)m[2] * j2);
            const float r1 = (float)((double)m[3] * j0 + (double)m[4] * j1 + (double)m[5] * j2);
            float *d = a.planes + (size_t)(2 + j) * np + off;
            atomic_add_f32(d, r0 * w00a + r1 * w00b);
            atomic_add_f32(d + 1, r0 * (1.f - dy) + r1 * (-dx));
            atomic_add_f32(d + Wp, r0 * (-dy) + r1 * (1.f - dx));
            atomic_add_f32(d + Wp + 1, r0 * dy + r1 * dx);
          }
        }
      }
    }
  }
}

void launch_be_splat(const BeSplatArgs &a, bool deriv, hipStream_t s) {
  if (a.n <= 0) return;
  const dim3 g(splat_grid(a.n)), b(256);
  if (a.order == 2) {
    if (deriv) hipLaunchKernelGGL((be_splat_kernel<2, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((be_splat_kernel<2, false>), g, b, 0, s, a);
  } else {
    if (deriv) hipLaunchKernelGGL((be_splat_kernel<4, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((be_splat_kernel<4, false>), g, b, 0, s, a);
  }
}

// ---------------------------------------------------------------------------------------------- K3+K5+K6
__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * (len - 1) - p;
  return p;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide sum of a per-thread double; result valid in thread 0.  red: 4 doubles of LDS per call site.
__device__ __forceinline__ double block_sum(double v, double *red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

size_t image_lds_bytes(int r) {
  const int rawW = kTileX + 2 * r, rawH = kTileY + 2 * r;
  return sizeof(float) * ((size_t)rawW * rawH + (size_t)kTileX * rawH) + sizeof(double) * 4;
}

// One workgroup = one 64x16 output tile x one group of <= kPlaneGroup derivative planes (blockIdx.z).
// LDS: raw tile with halo -> row-blurred tile -> column pass in registers -> fp64 moments.
// Row pass:  s = k[0]*S[x-r]; s += k[j]*S[x-r+j]           (generic row filter order)
// Col pass:  s = k[r]*T[y];   s += k[r+j]*(T[y+j]+T[y-j])  (symmetric column filter order)
__global__ __launch_bounds__(kImgThreads) void image_moments_kernel(ImgArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int r = a.r, W = a.W, H = a.H;
  const int rawW = kTileX + 2 * r, rawH = kTileY + 2 * r;
  double *red = reinterpret_cast<double *>(smem_raw);
  float *raw = reinterpret_cast<float *>(smem_raw + 4 * sizeof(double));
  float *rowb = raw + rawW * rawH;
  const int tid = threadIdx.x;
  const int tile = blockIdx.x;
  const int x0 = (tile % a.tiles_x) * kTileX, y0 = (tile / a.tiles_x) * kTileY;
  const int tx = tid & 63, tq = tid >> 6;  // output column, row quad
  const size_t np = (size_t)W * H;
  const float alpha = a.alpha ? (float)(*a.alpha) : 0.f;
  const int g = blockIdx.z;                // plane group; group 0 also owns the I moments
  const int k_beg = g * kPlaneGroup;
  const int k_end = min(a.P, k_beg + kPlaneGroup);

  float I[4] = {0.f, 0.f, 0.f, 0.f};
  bool valid[4];
#pragma unroll
  for (int j = 0; j < 4; j++) valid[j] = (x0 + tx < W) && (y0 + tq * 4 + j < H);

  // pass over plane 0 (k == -1) then the group's derivative planes
  for (int k = -1; k < k_end; k = (k < 0 ? k_beg : k + 1)) {
    __syncthreads();
    for (int idx = tid; idx < rawW * rawH; idx += kImgThreads) {
      const int ly = idx / rawW, lx = idx - ly * rawW;
      const int gx = reflect101(x0 + lx - r, W), gy = reflect101(y0 + ly - r, H);
      const size_t off = (size_t)gy * W + gx;
      float v;
      if (k < 0) {
        v = a.src_a[off];
        if (a.src_b) v = v + a.src_b[off];        // IL = IL_old + IL_new
        if (a.igp) v = a.igp[off] * alpha + v;     // I = IGp*alpha + IL
      } else {
        v = a.dplanes[(size_t)k * np + off];
      }
      raw[idx] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < kTileX * rawH; idx += kImgThreads) {
      const int ly = idx >> 6, lx = idx & 63;
      const float *S = raw + ly * rawW + lx;
      float s = a.taps[0] * S[0];
      for (int j = 1; j <= 2 * r; j++) s += a.taps[j] * S[j];
      rowb[idx] = s;
    }
    __syncthreads();
    double sD = 0, sID = 0, sI = 0, sII = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int ly = tq * 4 + j + r;
      const float *T = rowb + ly * kTileX + tx;
      float s = a.taps[r] * T[0];
      for (int t = 1; t <= r; t++) s += a.taps[r + t] * (T[t * kTileX] + T[-t * kTileX]);
      if (valid[j]) {
        const size_t o = (size_t)(y0 + tq * 4 + j) * W + (x0 + tx);
        if (k < 0) {
          I[j] = s;
          sI += (double)s;
          sII += (double)s * (double)s;
          if (a.out_blur0 && g == 0) a.out_blur0[o] = s;
        } else {
          sD += (double)s;
          sID += (double)I[j] * (double)s;
          if (a.out_blurd) a.out_blurd[(size_t)k * np + o] = s;
        }
      }
    }
    if (k < 0) {
      if (g == 0) {
        const double t0 = block_sum(sI, red), t1 = block_sum(sII, red);
        if (tid == 0) {
          a.partials[(size_t)0 * a.nblk + tile] = t0;
          a.partials[(size_t)1 * a.nblk + tile] = t1;
        }
      }
    } else {
      const double t0 = block_sum(sD, red), t1 = block_sum(sID, red);
      if (tid == 0) {
        a.partials[(size_t)(2 + 2 * k) * a.nblk + tile] = t0;
        a.partials[(size_t)(3 + 2 * k) * a.nblk + tile] = t1;
      }
    }
  }
}

void launch_image_moments(const ImgArgs &a, hipStream_t s) {
  const int groups = a.P > 0 ? (a.P + kPlaneGroup - 1) / kPlaneGroup : 1;
  hipLaunchKernelGGL(image_moments_kernel, dim3(a.nblk, 1, groups), dim3(kImgThreads), image_lds_bytes(a.r), s, a);
}

// ---------------------------------------------------------------------------------------------- reduce + finalize
__global__ __launch_bounds__(256) void reduce_partials_kernel(FinalizeArgs a) {
  __shared__ double red[4];
  const int q = blockIdx.x;
  double s = 0;
  for (int i = threadIdx.x; i < a.nblk; i += 256) s += a.partials[(size_t)q * a.nblk + i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) a.sums[q] = s;
}

// contrast / gradient from the moments (fp64):
//   variance:     contrast = (sqrt(max(E[I^2]-mu^2,0)))^2 ; grad_k = 2*(E[I D_k] - mu*E[D_k])
//   mean square:  contrast = E[I^2]                        ; grad_k = 2*E[I D_k]
__global__ void finalize_kernel(FinalizeArgs a) {
  const int t = threadIdx.x;
  const double N = a.npix;
  const double mu = a.sums[0] / N;
  if (t == 0) {
    double c;
    if (a.measure == 1) {
      c = a.sums[1] / N;
    } else {
      double var = a.sums[1] / N - mu * mu;
      if (var < 0) var = 0;
      const double sd = sqrt(var);
      c = sd * sd;
    }
    a.result[0] = c;
    a.result[1] = mu;
  }
  for (int k = t; k < a.P; k += blockDim.x) {
    const double eD = a.sums[2 + 2 * k] / N, eID = a.sums[3 + 2 * k] / N;
    a.result[2 + k] = (a.measure == 1) ? 2.0 * eID : 2.0 * (eID - mu * eD);
  }
}

void launch_finalize(const FinalizeArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(2 + 2 * a.P), dim3(256), 0, s, a);
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, s, a);
}

// ---------------------------------------------------------------------------------------------- K4 alpha
__global__ __launch_bounds__(256) void alpha_partials_kernel(AlphaArgs a) {
  __shared__ double red[4];
  double area_g = 0, num_g = 0, area_l = 0, num_l = 0, nz = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.npix; i += gridDim.x * 256) {
    const float gq = a.igp[i];
    const float il = a.il_old[i] + a.il_new[i];
    area_g += (double)(1.f - expf(-1.0f * gq));
    num_g += (double)gq;
    area_l += (double)(1.f - expf(-1.0f * il));
    num_l += (double)il;
    nz += (gq != 0.f) ? 1.0 : 0.0;
  }
  const double v[5] = {area_g, num_g, area_l, num_l, nz};
  for (int q = 0; q < 5; q++) {
    const double t = block_sum(v[q], red);
    if (threadIdx.x == 0) a.partials[(size_t)q * a.nblk + blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(256) void alpha_finalize_kernel(AlphaArgs a) {
  __shared__ double red[4];
  double tot[5];
  for (int q = 0; q < 5; q++) {
    double s = 0;
    for (int i = threadIdx.x; i < a.nblk; i += 256) s += a.partials[(size_t)q * a.nblk + i];
    tot[q] = block_sum(s, red);
  }
  if (threadIdx.x == 0) {
    double alpha = 0;
    if (tot[4] >= 1) alpha = (tot[3] / tot[2]) / (tot[1] / tot[0]);
    *a.alpha = alpha;
    *a.result_alpha = alpha;
  }
}
void launch_alpha(const AlphaArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(alpha_partials_kernel, dim3(a.nblk), dim3(256), 0, s, a);
  hipLaunchKernelGGL(alpha_finalize_kernel, dim3(1), dim3(256), 0, s, a);
}

// planar [3][npix] -> interleaved [npix][3]  (CV_32FC3 layout of the reference's derivative image)
__global__ void interleave3_kernel(const float *planes, float *out, int npix) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
    out[3 * (size_t)i] = planes[i];
    out[3 * (size_t)i + 1] = planes[(size_t)npix + i];
    out[3 * (size_t)i + 2] = planes[2 * (size_t)npix + i];
  }
}
void launch_interleave3(const float *planes, float *out, int npix, hipStream_t s) {
  hipLaunchKernelGGL(interleave3_kernel, dim3(1024), dim3(256), 0, s, planes, out, npix);
}

}  // namespace cmx
