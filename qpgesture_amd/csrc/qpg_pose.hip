// Post-decode pose conversion: de-normalise the decoder's output, optionally smooth it in time,
// and turn every joint's 3x3 rotation matrix into intrinsic Z-X-Y Euler angles in degrees - the array the reference
// hands to its (third-party) BVH writer.
//   VisualizeCodebook.py:148-149       out_poses = poses * clip(std, 0.01) + mean          (float64)
//   process_bvh.py:62-68               savgol_filter(column, 15, 2)                          (optional)
//   process_bvh.py:70-76               R.from_matrix(3x3).as_euler('ZXY', degrees=True)
// scipy's from_matrix orthogonalises a non-orthogonal input (the decoder's matrices are only approximately rotations)
// by the orthogonal-Procrustes solution U V^T of its SVD and rejects non-positive determinants; U V^T is the orthogonal
// polar factor, computed here by the Newton iteration X <- (X + X^-T)/2 (quadratically convergent, f64).  The Euler
// angles are the closed form of R = Rz(a) Rx(b) Ry(c); at gimbal lock (|b| = 90 deg) the third angle is set to 0 like
// scipy does.  One thread per (frame, joint); everything in f64.
#include "qpg_common.h"

__global__ __launch_bounds__(256) void pose_to_euler_kernel(const float* __restrict__ poses, int64_t T, int J,
                                                            const double* __restrict__ mean,
                                                            const double* __restrict__ stdc,
                                                            const double* __restrict__ sg_mid,    // [W] or null
                                                            const double* __restrict__ sg_head,   // [W/2][W]
                                                            const double* __restrict__ sg_tail,   // [W/2][W]
                                                            int W, double* __restrict__ euler,
                                                            int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * J) return;
  const int64_t t = i / J;
  const int j = (int)(i - t * J);
  const int C = J * 9;
  double m[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    const int c = j * 9 + e;
    if (!sg_mid) {
      m[e] = (double)poses[t * C + c] * stdc[c] + mean[c];
    } else {
      // Savitzky-Golay, window W, polynomial order 2, scipy mode='interp': interior frames = symmetric convolution,
      // the first / last W/2 frames = the fitted polynomial of the first / last W frames evaluated there
      const int h = W / 2;
      const double* wrow;
      int64_t t0;
      if (t < h) {
        wrow = sg_head + (int64_t)t * W;
        t0 = 0;
      } else if (t >= T - h) {
        wrow = sg_tail + (int64_t)(t - (T - h)) * W;
        t0 = T - W;
      } else {
        wrow = sg_mid;
        t0 = t - h;
      }
      double s = 0.0;
      for (int k = 0; k < W; ++k) s += wrow[k] * ((double)poses[(t0 + k) * C + c] * stdc[c] + mean[c]);
      m[e] = s;
    }
  }
  auto det3 = [](const double* a) {
    return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
  };
  if (!(det3(m) > 0.0)) {
    atomicMax(status, 1);             // scipy: ValueError("Non-positive determinant (left-handed or null ...)")
    euler[i * 3 + 0] = euler[i * 3 + 1] = euler[i * 3 + 2] = 0.0;
    return;
  }
  double x[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) x[e] = m[e];
  for (int it = 0; it < 30; ++it) {
    const double d = det3(x);
    // inverse transpose = cofactor matrix / det
    double cf[9];
    cf[0] = x[4] * x[8] - x[5] * x[7];
    cf[1] = x[5] * x[6] - x[3] * x[8];
    cf[2] = x[3] * x[7] - x[4] * x[6];
    cf[3] = x[2] * x[7] - x[1] * x[8];
    cf[4] = x[0] * x[8] - x[2] * x[6];
    cf[5] = x[1] * x[6] - x[0] * x[7];
    cf[6] = x[1] * x[5] - x[2] * x[4];
    cf[7] = x[2] * x[3] - x[0] * x[5];
    cf[8] = x[0] * x[4] - x[1] * x[3];
    double delta = 0.0;
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      const double nx = 0.5 * (x[e] + cf[e] / d);
      delta = fmax(delta, fabs(nx - x[e]));
      x[e] = nx;
    }
    if (delta < 1e-15) break;
  }
  // R = Rz(a) Rx(b) Ry(c):  R21 = sin b,  R01 = -sin a cos b,  R11 = cos a cos b,  R20 = -cos b sin c,  R22 = cos b cos c
  const double RAD = 57.29577951308232;
  double sb = x[7];
  sb = sb > 1.0 ? 1.0 : (sb < -1.0 ? -1.0 : sb);
  const double b = asin(sb);
  double a, c;
  if (fabs(sb) < 1.0 - 1e-14) {
    a = atan2(-x[1], x[4]);
    c = atan2(-x[6], x[8]);
  } else {                                 // gimbal lock: only a +- c is defined; scipy sets the third angle to 0
    c = 0.0;
    a = atan2(x[3], x[0]);
  }
  euler[i * 3 + 0] = a * RAD;
  euler[i * 3 + 1] = b * RAD;
  euler[i * 3 + 2] = c * RAD;
}

extern "C" int qpg_pose_to_euler_f64(qpg_ctx* ctx, void* stream, const float* poses, int64_t T, int J, const double* mean,
                                     const double* stdc, const double* sg_mid, const double* sg_head,
                                     const double* sg_tail, int W, double* euler, int32_t* status) {
  QPG_REQUIRE(ctx && poses && mean && stdc && euler && status && T >= 0 && J > 0, "qpg_pose_to_euler_f64: bad argument");
  QPG_REQUIRE(!sg_mid || (sg_head && sg_tail && W >= 3 && (W & 1) && T >= W),
              "qpg_pose_to_euler_f64: smoothing needs an odd window W <= T and the three coefficient tables");
  if (T == 0) return QPG_OK;
  const int64_t n = T * J;
  hipLaunchKernelGGL(pose_to_euler_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, qpg_stream(stream), poses, T, J,
                     mean, stdc, sg_mid, sg_head, sg_tail, W, euler, status);
  QPG_LAUNCH_CHECK("pose_to_euler_kernel");
  return QPG_OK;
}
