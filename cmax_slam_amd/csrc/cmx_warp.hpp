// cmx_warp.hpp -- the per-event warps (device inline), shared by the splat, gather and binning kernels.
//   front end: first-order rotation + pinhole      (reference local_image_warped_events.cpp:94-145)
//   back end : R*b + equirectangular projection    (reference event_pano_warper.cpp:262-296,
//                                                   equirectangular_camera.h:18-45)
// fp64 geometry exactly as the reference (compile with -ffp-contract=off), fp32 bilinear offsets.
#pragma once
#include "cmx_internal.hpp"
#include "cmx_trig.hpp"

namespace cmx {

// fire-and-forget fp32 atomic add (global_atomic_add_f32, no return value); needs -munsafe-fp-atomics
__device__ __forceinline__ void atomic_add_f32(float *p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// per-event front-end warp: fp64 first-order rotation + pinhole; fp32 bilinear offsets; optional 2x3 Jacobian rows
// bearing vector of sensor pixel (ex, ey): one 16-byte load when the table's z column is all ones
template <typename Args>
__device__ __forceinline__ void load_bearing(const Args &a, int ex, int ey, double &b0, double &b1, double &b2) {
  const size_t i = (size_t)ey * a.W + ex;
  if (a.lut2) {
    const double2 v = *reinterpret_cast<const double2 *>(a.lut2 + 2 * i);
    b0 = v.x; b1 = v.y; b2 = 1.0;
  } else {
    const double *l = a.lut + 3 * i;
    b0 = l[0]; b1 = l[1]; b2 = l[2];
  }
}

// device-driven solve: omega lives in device memory (uniform address: scalar loads)
__device__ __forceinline__ void fe_resolve_omega(FeSplatArgs &a) {
  if (a.w_dev) { a.wx = a.w_dev[0]; a.wy = a.w_dev[1]; a.wz = a.w_dev[2]; }
}

// device-driven solve: "finished" (FeSplatArgs::skip / ImgArgs::skip).  The word is set by the finalize step of an earlier launch --
// or by the HOST, on the null stream, in the middle of whatever launch is running when it takes a solve over (cmx_chain.cpp).  The
// decision must therefore be ONE read per workgroup: with one read per wave, some waves of a workgroup leave and the others stay, and
// the ones that stay find LDS the leavers were to clear (a splat window full of a previous workgroup's data, flushed as votes -- also
// to rows below the image's last one, i.e. into whatever allocation follows the plane).
__device__ __forceinline__ bool wg_stop_requested(const int *skip) {
  if (!skip) return false;  // (kernel argument: uniform)
  __shared__ int stop_sh;
  if (threadIdx.x == 0) stop_sh = __hip_atomic_load(skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  return stop_sh != 0;
}

struct FeWarp {
  int xx, yy;
  float dx, dy;
  bool ok;
  float r0[3], r1[3];
};

// the arithmetic of the front-end warp on an already-loaded bearing vector (lets callers batch the loads)
template <bool DERIV>
__device__ __forceinline__ FeWarp fe_warp_math(const FeSplatArgs &a, double px, double py, double pz, double dt) {
  FeWarp w;
  // p' = p + (omega*dt) x p   (first-order rotation)
  const double drx = a.wx * dt, dry = a.wy * dt, drz = a.wz * dt;
  const double rx = px + (dry * pz - drz * py);
  const double ry = py + (drz * px - drx * pz);
  const double rz = pz + (drx * py - dry * px);
  const double iz = 1.0 / rz;
  const double cxn = rx * iz, cyn = ry * iz;
  const double u = a.fx * cxn + a.cx;
  const double v = a.fy * cyn + a.cy;
  w.xx = (int)u;
  w.yy = (int)v;
  w.ok = (1 <= w.xx && w.xx < a.W - 2 && 1 <= w.yy && w.yy < a.H - 2);
  w.dx = (float)(u - w.xx);
  w.dy = (float)(v - w.yy);
  if (DERIV) {
    // J = diag(fx,fy) * J_proj(2x3) * [(-dt) p]_x   evaluated in the reference's operation order
    const double vx = (-dt) * px, vy = (-dt) * py, vz = (-dt) * pz;
    const double a02 = -cxn * iz, a12 = -cyn * iz;
    w.r0[0] = (float)(a.fx * (a02 * (-vy)));
    w.r0[1] = (float)(a.fx * (iz * (-vz) + a02 * vx));
    w.r0[2] = (float)(a.fx * (iz * vy));
    w.r1[0] = (float)(a.fy * (iz * vz + a12 * (-vy)));
    w.r1[1] = (float)(a.fy * (a12 * vx));
    w.r1[2] = (float)(a.fy * (iz * (-vx)));
  }
  return w;
}

template <bool DERIV>
__device__ __forceinline__ FeWarp fe_warp_core(const FeSplatArgs &a, uint32_t e, double dt) {
  const int ex = e & 0xffff, ey = (e >> 16) & 0x7fff;
  double b0, b1, b2;
  load_bearing(a, ex, ey, b0, b1, b2);
  return fe_warp_math<DERIV>(a, b0, b1, b2, dt);
}

template <bool DERIV>
__device__ __forceinline__ FeWarp fe_warp_event(const FeSplatArgs &a, int i) {
  return fe_warp_core<DERIV>(a, a.xy[i], a.batch_dt[i / a.per_batch]);
}

// per-event back-end warp: R*b, equirectangular projection (fp64), optional fp32 d(pixel)/d(rotation) 2x3
struct BeWarp {
  int xx, yy;
  float dx, dy;
  bool ok, is_old;
  int batch;
  float m[6];
};

// Back-end arithmetic and FMA contraction (round 5).  The library is built with -ffp-contract=off so that the FRONT end's fp64 warp is
// the reference's x86-64 (no FMA) arithmetic bit for bit.  The back end never was bit-identical -- its atan2 / asin are polynomial forms
// 1-3 ulp from glibc's (cmx_trig.hpp, themselves FMA chains) -- so the rotation R b, |R b|^2, the Newton steps and the pixel
// coordinate are contracted too: a difference of ~1e-16 relative in the ray, far below the trig forms' own, for 15 fp64 instructions
// less per event in kernels whose VALU issue is what binds them (profiles/r05_be_valu.txt).  Every back-end kernel (tile keys, splat,
// gather, reference-shaped splat) goes through these two functions, so they agree with one another on every vote cell bit for bit.
// Z1: the bearing's z is the constant 1 (tile- / time-ordered bearing streams hold (x, y) only): R[2], R[5], R[8] are added, not multiplied.
template <bool Z1>
__device__ __forceinline__ void be_rotate(const double *R, double b0, double b1, double b2, double &x, double &y, double &z) {
  // explicit FMAs in ONE association for both forms: with b2 == 1.0 the general form rounds R2 * b2 exactly, so
  // fma(R1, b1, R2 * b2) is the Z1 form's fma(R1, b1, R2) bit for bit -- whatever the compiler's contraction pass would have chosen
  if (Z1) {
    x = __builtin_fma(R[0], b0, __builtin_fma(R[1], b1, R[2]));
    y = __builtin_fma(R[3], b0, __builtin_fma(R[4], b1, R[5]));
    z = __builtin_fma(R[6], b0, __builtin_fma(R[7], b1, R[8]));
  } else {
    x = __builtin_fma(R[0], b0, __builtin_fma(R[1], b1, R[2] * b2));
    y = __builtin_fma(R[3], b0, __builtin_fma(R[4], b1, R[5] * b2));
    z = __builtin_fma(R[6], b0, __builtin_fma(R[7], b1, R[8] * b2));
  }
}

// the arithmetic of the back-end warp behind the rotation: e_ray_w = (x, y, z) = R * bearing.
// DERIV = 1: d(pixel)/d(rotation) with the projection Jacobian evaluated in fp64 and cast to fp32 entry by entry,
//            exactly the reference's Matx23f (equirectangular_camera.h:33-44) -- the derivative-plane (faithful) mode.
// DERIV = 2: the same five entries evaluated in fp32 from the fp64 ray (the reference multiplies them in fp32 anyway,
//            event_pano_warper.cpp:281-285); relative difference ~1e-7 per entry, saves five fp64 divisions and a
//            square root per event in the ALU-bound gather pass of the adjoint mode.
template <int DERIV>
__device__ __forceinline__ BeWarp be_project(const BeSplatArgs &a, uint32_t e, int batch, double x, double y, double z) {
  BeWarp w;
  w.is_old = (e >> 31) != 0;
  w.batch = batch;
  double pxm, pym, rho;
  {
#pragma clang fp contract(fast)  // (this block only: the Jacobian entries below keep the reference's operation-by-operation rounding)
    // equirectangular projection
    const double phi = lean_atan2(x, z);  // (cmx_trig.hpp: the two transcendental calls were three quarters of this function)
    // rho = |R b|: the reference's sqrt and division (y / rho) as ONE reciprocal square root -- v_rsq_f64 and two Newton steps
    // (~14 instructions against ~45; y * (1/rho) is within 2 ulp of y / rho, i.e. 1e-16 of the pixel coordinate)
    const double rho2 = x * x + (y * y + z * z);
    double inv_rho = __builtin_amdgcn_rsq(rho2);
    const double h = 0.5 * rho2;
    inv_rho = inv_rho * (1.5 - (h * inv_rho) * inv_rho);
    inv_rho = inv_rho * (1.5 - (h * inv_rho) * inv_rho);
    rho = rho2 * inv_rho;
    const double theta = lean_asin(y * inv_rho);
    pxm = a.cxp + phi * a.fx;
    pym = a.cyp + theta * a.fy;
  }
  w.xx = (int)pxm;
  w.yy = (int)pym;
  w.ok = (1 <= w.xx && w.xx < a.Wp - 2 && 1 <= w.yy && w.yy < a.Hp - 2);
  w.dx = (float)(pxm - w.xx);
  w.dy = (float)(pym - w.yy);
  if (DERIV == 2) {
    const float xf = (float)x, yf = (float)y, zf = (float)z, rhof = (float)rho;
    const float fxf = (float)a.fx, fyf = (float)a.fy;
    // hardware reciprocal / reciprocal square root (1 ulp) instead of IEEE fp32 divisions and a square root: these five
    // entries are fp32 approximations of fp64 quantities to begin with, and four correctly rounded divisions were ~40 of
    // the gather's ~380 VALU instructions per event
    const float inv_rho = __builtin_amdgcn_rcpf(rhof), inv_z = __builtin_amdgcn_rcpf(zf);
    const float Ydivrho = yf * inv_rho, XdivZ = xf * inv_z;
    const float tmp1 = fxf * inv_z * __builtin_amdgcn_rcpf(1.f + XdivZ * XdivZ);
    const float tmp2 = -fyf * __builtin_amdgcn_rsqf(1.f - Ydivrho * Ydivrho);
    const float tmp3 = Ydivrho * inv_rho * inv_rho;
    const float d00 = tmp1, d02 = -tmp1 * XdivZ;
    const float d10 = tmp2 * tmp3 * xf, d11 = tmp2 * (tmp3 * yf - inv_rho), d12 = tmp2 * tmp3 * zf;
    w.m[0] = d02 * yf;
    w.m[1] = d00 * zf + d02 * (-xf);
    w.m[2] = d00 * (-yf);
    w.m[3] = d11 * (-zf) + d12 * yf;
    w.m[4] = d10 * zf + d12 * (-xf);
    w.m[5] = d10 * (-yf) + d11 * xf;
  } else if (DERIV == 1) {
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
    w.m[0] = 0.f * (-rbz) + d02 * rby;
    w.m[1] = d00 * rbz + d02 * (-rbx);
    w.m[2] = d00 * (-rby) + 0.f * rbx;
    w.m[3] = d11 * (-rbz) + d12 * rby;
    w.m[4] = d10 * rbz + d12 * (-rbx);
    w.m[5] = d10 * (-rby) + d11 * rbx;
  }
  return w;
}

// rotation + projection on already-loaded operands: bearing (b0, b1, b2) and the batch rotation R[9]
template <int DERIV>
__device__ __forceinline__ BeWarp be_warp_math(const BeSplatArgs &a, uint32_t e, int batch, double b0, double b1,
                                              double b2, const double *R) {
  double x, y, z;
  be_rotate<false>(R, b0, b1, b2, x, y, z);
  return be_project<DERIV>(a, e, batch, x, y, z);
}

template <int DERIV>
__device__ __forceinline__ BeWarp be_warp_core(const BeSplatArgs &a, uint32_t e, int batch) {
  const int ex = e & 0xffff, ey = (e >> 16) & 0x7fff;
  double b0, b1, b2;
  load_bearing(a, ex, ey, b0, b1, b2);
  double R[9];
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = a.poseR[batch].R[k];
  return be_warp_math<DERIV>(a, e, batch, b0, b1, b2, R);
}

template <int DERIV>
__device__ __forceinline__ BeWarp be_warp_event(const BeSplatArgs &a, int i) {
  return be_warp_core<DERIV>(a, a.xy[i], i / a.per_batch);
}

}  // namespace cmx
