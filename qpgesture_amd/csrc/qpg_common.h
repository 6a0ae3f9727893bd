// Shared host-side plumbing for libqpg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/qpg.h"

struct qpg_ctx {
  int device;
  int n_cu;
  float* zeros;   // 256 B of device zeros: out-of-range tile loads are redirected here instead of being selected to 0
  bool select_lds_raised;   // percode_select_mixed_f64_kernel's dynamic-LDS limit has been raised on this device
};

void qpg_set_error(const char* fmt, ...);

#define QPG_REQUIRE(cond, ...)      \
  do {                              \
    if (!(cond)) {                  \
      qpg_set_error(__VA_ARGS__);   \
      return QPG_EINVAL;            \
    }                               \
  } while (0)

#define QPG_LAUNCH_CHECK(name)                                                \
  do {                                                                        \
    hipError_t e_ = hipGetLastError();                                        \
    if (e_ != hipSuccess) {                                                   \
      qpg_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));    \
      return QPG_EHIP;                                                        \
    }                                                                         \
  } while (0)

static inline hipStream_t qpg_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// IEEE single-rounding float ops: the text and phase-gate distances must reproduce
// NumPy/scikit-learn float32 arithmetic bit for bit (separate multiply and add, correctly
// rounded sqrt and divide).  hipcc's __fmul_rn/__fsqrt_rn are plain `*` / native sqrt unless
// OCML_BASIC_ROUNDED_OPERATIONS is set, so exactness comes from the build flags instead:
// every file is compiled with -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt, and
// the pragma below repeats it where it matters.
#pragma clang fp contract(off)
__device__ __forceinline__ float f_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float f_add(float a, float b) { return a + b; }
__device__ __forceinline__ float f_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float f_div(float a, float b) { return a / b; }
__device__ __forceinline__ float f_sqrt(float a) { return __builtin_sqrtf(a); }
// the same for float64 (the near-tie guard re-evaluates audio distances in the reference's f64 arithmetic)
__device__ __forceinline__ double f_mul(double a, double b) { return a * b; }
__device__ __forceinline__ double f_add(double a, double b) { return a + b; }
__device__ __forceinline__ double f_sub(double a, double b) { return a - b; }
__device__ __forceinline__ double f_div(double a, double b) { return a / b; }

// sklearn semantics for degenerate rows: a row whose norm is < 10*eps is left unscaled by
// normalize(); for an all-zero row that gives 0.5*|other unit vector|^2 = 0.5 (0 if both are zero).
__device__ __forceinline__ double cosine_from_dot(double dot, double qn2, double cn2) {
  const double tiny = 10.0 * 2.220446049250313e-16;
  double nq = sqrt(qn2), nc = sqrt(cn2);
  bool zq = nq < tiny, zc = nc < tiny;
  if (zq || zc) {
    // unscaled row contributes its own squared norm; exact only for all-zero rows, which is
    // the case that occurs (zero padding); both-degenerate -> 0.5*(qn2 + cn2 - 2 dot)
    double a = zq ? qn2 : 1.0, b = zc ? cn2 : 1.0;
    double cross = dot / ((zq ? 1.0 : nq) * (zc ? 1.0 : nc));
    return 0.5 * (a + b - 2.0 * cross);
  }
  return 1.0 - dot / (nq * nc);
}

// A-priori error bound of the mixed-precision audio sweep (qpg_audio_cosine_mx, derivation in qpg_audio.hip):
// |D_mx[q][c] - D_f64[q][c]| <= gamma_32 (f32 FMA chains of 32 products) + f64 noise, for every pair.
// (the value is QPG_AUDIO_MX_ERR of include/qpg.h)
