// Context, error reporting and the one-off database preparation kernels.
#include <stdarg.h>

#include "qpg_common.h"
#include <stdlib.h>

static thread_local char g_err[512] = "";

void qpg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int qpg_version(void) { return 107; }          // 1.07: round 6 (qpg_build_id, qpg_ctx_set_option; the qpg_debug_* setters left the product: include/qpg.h)

// The SHA-256 (first 16 hex digits) of the sources this library was compiled from - csrc/*.hip, csrc/*.h, include/qpg.h in
// name order, as qpgesture_amd/build.py computes it and passes it to THIS file's compile (-DQPG_BUILD_ID).  _lib.load()
// compares it with the tree it finds itself in and rebuilds on a mismatch; tests/test_host_cpu.py asserts the equality.
#ifndef QPG_BUILD_ID
#define QPG_BUILD_ID "unstamped"
#endif
static const char g_build_id[] = "QPG_BUILD_ID=" QPG_BUILD_ID;       // (build.py reads the marker out of the file, no dlopen)
extern "C" const char* qpg_build_id(void) { return g_build_id + 13; }

// Is HIP_FORCE_DEV_KERNARG=1 in this process's environment? (see qpg_ctx_create in include/qpg.h)
extern "C" int qpg_dev_kernarg(void) {
  const char* e = getenv("HIP_FORCE_DEV_KERNARG");
  return (e && e[0] == '1') ? 1 : 0;
}

extern "C" int qpg_last_error(char* buf, size_t n) {
  if (!buf || n == 0) return QPG_EINVAL;
  strncpy(buf, g_err, n - 1);
  buf[n - 1] = 0;
  return QPG_OK;
}

extern "C" int qpg_ctx_create(int device, qpg_ctx** out) {
  QPG_REQUIRE(out != nullptr, "qpg_ctx_create: out is null");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
    qpg_set_error("qpg_ctx_create: no HIP device %d (count %d)", device, n);
    return QPG_EHIP;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) {
    qpg_set_error("qpg_ctx_create: hipGetDeviceProperties failed");
    return QPG_EHIP;
  }
  qpg_ctx* c = new qpg_ctx;
  c->device = device;
  c->n_cu = p.multiProcessorCount;
  c->zeros = nullptr;
  c->select_lds_raised = false;
  for (int i = 0; i < QPG_OPT_COUNT; ++i) c->opt[i] = 0;
  c->opt[QPG_OPT_GATE_DEDUP_FROM_CHAINS] = 1;
  int prev = 0;
  (void)hipGetDevice(&prev);
  const bool ok = hipSetDevice(device) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&c->zeros), 4096) == hipSuccess &&
                  hipMemset(c->zeros, 0, 4096) == hipSuccess;
  (void)hipSetDevice(prev);
  if (!ok) {
    qpg_set_error("qpg_ctx_create: could not allocate the context's zero page on device %d", device);
    delete c;
    return QPG_EHIP;
  }
  *out = c;
  return QPG_OK;
}

extern "C" int qpg_ctx_set_option(qpg_ctx* ctx, int option, int value) {
  QPG_REQUIRE(ctx != nullptr, "qpg_ctx_set_option: null context");
  QPG_REQUIRE(option >= 0 && option < QPG_OPT_COUNT, "qpg_ctx_set_option: unknown option %d", option);
  if (option == QPG_OPT_GATE_DEDUP_FROM_CHAINS) QPG_REQUIRE(value >= 0, "qpg_ctx_set_option: gate_dedup_from_chains >= 0");
  ctx->opt[option] = value;
  return QPG_OK;
}

extern "C" int qpg_ctx_get_option(qpg_ctx* ctx, int option, int* value) {
  QPG_REQUIRE(ctx != nullptr && value != nullptr, "qpg_ctx_get_option: null pointer");
  QPG_REQUIRE(option >= 0 && option < QPG_OPT_COUNT, "qpg_ctx_get_option: unknown option %d", option);
  *value = ctx->opt[option];
  return QPG_OK;
}

extern "C" int qpg_ctx_destroy(qpg_ctx* ctx) {
  if (ctx && ctx->zeros) (void)hipFree(ctx->zeros);
  delete ctx;
  return QPG_OK;
}

// ---------------------------------------------------------------------------------------------
// A stream-ordered signal to the host (round 6, GraphPipeline): one thread stores `value` to *dst - pinned host memory
// the device can reach - at system scope, behind everything enqueued on the stream before it.  What the host polls to learn
// that a replay's SWEEP is over (the moment the other lane's sweep may start) without waiting for the replay's tail.
// ---------------------------------------------------------------------------------------------
__global__ void signal_i32_kernel(int32_t* dst, int32_t value) {
  __hip_atomic_store(dst, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int qpg_signal_i32(qpg_ctx* ctx, void* stream, int32_t* dst, int32_t value) {
  QPG_REQUIRE(ctx && dst && (reinterpret_cast<uintptr_t>(dst) % 4) == 0, "qpg_signal_i32: null or misaligned destination");
  hipLaunchKernelGGL(signal_i32_kernel, dim3(1), dim3(1), 0, qpg_stream(stream), dst, value);
  QPG_LAUNCH_CHECK("signal_i32_kernel");
  return QPG_OK;
}

// ---------------------------------------------------------------------------------------------
// The doorbell of a PRE-LAUNCHED replay (round 6, ClipGraph(doorbell=True)).  A serial matching step pays ~17 us of host time
// for hipGraphLaunch plus the command processor's start-up before its first kernel runs; the host can enqueue the NEXT replay
// while the current one still executes if that replay's first node waits for the host's go.  One thread: takes this replay's
// sequence number from a device counter (seq = ++*counter: immune to the host racing ahead) and waits until the host has
// stored go >= seq into pinned memory (system-scope loads, s_sleep between polls).  The host writes the replay's seed block
// and result sentinels BEFORE it stores go, so the GPU work of a step still starts only when its inputs are final.
// The wait is BOUNDED (timeout_ms of the 100 MHz wall clock): a host that never rings cannot hang the device - the replay then
// runs on whatever the seed block holds and its results are discarded by the host side (ClipGraph.drain).
// ---------------------------------------------------------------------------------------------
__global__ void doorbell_wait_kernel(int32_t* __restrict__ counter, const int32_t* __restrict__ go, int32_t timeout_ms) {
  const int32_t seq = atomicAdd(counter, 1) + 1;
  const long long t0 = wall_clock64();
  const long long limit = (long long)timeout_ms * 100000ll;                       // 100 MHz
  while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq < 0) {   // (wrap-safe comparison)
    if (wall_clock64() - t0 > limit) break;
    __builtin_amdgcn_s_sleep(16);
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);                                         // the seed block written before go
}

extern "C" int qpg_doorbell_wait(qpg_ctx* ctx, void* stream, int32_t* counter, const int32_t* go, int32_t timeout_ms) {
  QPG_REQUIRE(ctx && counter && go && timeout_ms > 0 && timeout_ms <= 60000, "qpg_doorbell_wait: bad argument (timeout 1..60000 ms)");
  hipLaunchKernelGGL(doorbell_wait_kernel, dim3(1), dim3(1), 0, qpg_stream(stream), counter, go, timeout_ms);
  QPG_LAUNCH_CHECK("doorbell_wait_kernel");
  return QPG_OK;
}

// ---------------------------------------------------------------------------------------------
// per-frame squared norm in f64: one wave per row, 16 B loads, wave64 shuffle reduce.
// HBM-bound: reads rows*F*4 bytes once.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frame_norm2_kernel(const float* __restrict__ x, int64_t rows, int F,
                                                          double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = x + row * F;
  double s = 0.0;
  if ((F & 3) == 0) {
    for (int e = lane * 4; e < F; e += 256) {
      f32x4 v = *reinterpret_cast<const f32x4*>(p + e);
      s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  } else {
    for (int e = lane; e < F; e += 64) s += (double)p[e] * p[e];
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) out[row] = s;
}

extern "C" int qpg_frame_norm2_f64(qpg_ctx* ctx, void* stream, const float* x, int64_t rows, int F, double* out) {
  QPG_REQUIRE(ctx && x && out && rows >= 0 && F > 0, "qpg_frame_norm2_f64: bad argument");
  if (rows == 0) return QPG_OK;
  hipLaunchKernelGGL(frame_norm2_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, qpg_stream(stream), x, rows,
                     F, out);
  QPG_LAUNCH_CHECK("frame_norm2_kernel");
  return QPG_OK;
}

__global__ void audio_cand_norm2_kernel(const double* __restrict__ fn2, int N, int T, const int32_t* __restrict__ cand_t,
                                        int G, int n_taps, int tap_stride, double* __restrict__ cn2) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * G) return;
  int j = (int)(i / G), g = (int)(i % G);
  int t0 = cand_t[g];
  double s = 0.0;
  for (int k = 0; k < n_taps; ++k) {
    int t = t0 + k * tap_stride;
    if (t < T) s += fn2[(int64_t)j * T + t];
  }
  cn2[i] = s;
}

extern "C" int qpg_audio_cand_norm2(qpg_ctx* ctx, void* stream, const double* fn2, int N, int T,
                                    const int32_t* cand_t, int G, int n_taps, int tap_stride, double* cn2) {
  QPG_REQUIRE(ctx && fn2 && cand_t && cn2 && N >= 0 && T > 0 && G > 0 && n_taps > 0 && tap_stride > 0,
              "qpg_audio_cand_norm2: bad argument");
  int64_t n = (int64_t)N * G;
  if (n == 0) return QPG_OK;
  hipLaunchKernelGGL(audio_cand_norm2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, qpg_stream(stream),
                     fn2, N, T, cand_t, G, n_taps, tap_stride, cn2);
  QPG_LAUNCH_CHECK("audio_cand_norm2_kernel");
  return QPG_OK;
}

// ---------------------------------------------------------------------------------------------
// scikit-learn-exact float32 row normalisation.  NumPy's einsum keeps 4 lane accumulators
// (lane = e & 3), visits 16-element groups as u = 3,2,1,0, finishes the tail in 4-wide zero-filled
// steps and combines (l0+l1)+(l2+l3) — see oracle/knn_oracle.py.  The four lane chains are
// independent, so a row is handled by 4 adjacent threads (one per einsum lane) and the horizontal
// sum is two shuffles; every thread then divides its quarter of the row.
// ---------------------------------------------------------------------------------------------
// GATHER: row r is read from x[(win[r]*R + row[r])*D] (the text queries of a clip: clip_context[int(i/n*30)] of
// window win[r], GestureKNN.py:549-551) instead of x[r*D].
template <bool GATHER>
__global__ __launch_bounds__(256) void l2_normalize_rows_kernel(const float* __restrict__ x, int64_t rows, int D,
                                                                float* __restrict__ out,
                                                                const int32_t* __restrict__ win,
                                                                const int32_t* __restrict__ row, int R) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t r = t >> 2;
  const int l = (int)(t & 3);
  const bool live = r < rows;
  if (!live) r = rows - 1;          // keep the whole aligned group of 4 in the shuffles
  const float* p = GATHER ? x + ((int64_t)win[r] * R + row[r]) * D : x + r * D;
  float a = 0.f;
  const int nfull = D >> 4;
  int g = 0;
  // eight 16-element groups per trip: the 32 loads of a lane are in flight together, the additions keep their order
  // (one group per trip waited a load latency per group: 20 us for 48 rows of 384, the first kernel of a clip's text side)
  for (; g + 8 <= nfull; g += 8) {
    float v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = p[(g + (j >> 2)) * 16 + (j & 3) * 4 + l];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int u = 3; u >= 0; --u) a = f_add(f_mul(v[j * 4 + u], v[j * 4 + u]), a);
    }
  }
  for (; g < nfull; ++g) {
#pragma unroll
    for (int u = 3; u >= 0; --u) {
      const float v = p[g * 16 + u * 4 + l];
      a = f_add(f_mul(v, v), a);
    }
  }
  for (int i = nfull * 16; i < D; i += 4) {
    const float v = (i + l < D) ? p[i + l] : 0.f;
    a = f_add(f_mul(v, v), a);
  }
  const float o1 = __shfl_xor(a, 1, 64);
  const float pair = f_add(a, o1);                    // (l0+l1) on lanes 0,1 ; (l2+l3) on lanes 2,3
  const float o2 = __shfl_xor(pair, 2, 64);
  float n = f_sqrt(f_add(pair, o2));                  // IEEE add is commutative: (l0+l1)+(l2+l3) on all 4
  if (n < 10.f * 1.1920928955078125e-07f) n = 1.f;    // sklearn _handle_zeros_in_scale
  if (live) {
    float* o = out + r * D;
    int e = l;
    for (; e + 28 < D; e += 32) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[e + 4 * j];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[e + 4 * j] = f_div(v[j], n);
    }
    for (; e < D; e += 4) o[e] = f_div(p[e], n);
  }
}

extern "C" int qpg_l2_normalize_rows_f32(qpg_ctx* ctx, void* stream, const float* x, int64_t rows, int D, float* out) {
  QPG_REQUIRE(ctx && x && out && rows >= 0 && D > 0, "qpg_l2_normalize_rows_f32: bad argument");
  if (rows == 0) return QPG_OK;
  hipLaunchKernelGGL(l2_normalize_rows_kernel<false>, dim3((unsigned)((rows * 4 + 255) / 256)), dim3(256), 0,
                     qpg_stream(stream), x, rows, D, out, (const int32_t*)nullptr, (const int32_t*)nullptr, 0);
  QPG_LAUNCH_CHECK("l2_normalize_rows_kernel");
  return QPG_OK;
}

extern "C" int qpg_text_pack_queries_f32(qpg_ctx* ctx, void* stream, const float* x, int M, int R, int D,
                                         const int32_t* q_win, const int32_t* q_row, int Q, float* out) {
  QPG_REQUIRE(ctx && x && q_win && q_row && out && M > 0 && R > 0 && D > 0 && Q >= 0,
              "qpg_text_pack_queries_f32: bad argument");
  if (Q == 0) return QPG_OK;
  hipLaunchKernelGGL(l2_normalize_rows_kernel<true>, dim3((unsigned)(((int64_t)Q * 4 + 255) / 256)), dim3(256), 0,
                     qpg_stream(stream), x, (int64_t)Q, D, out, q_win, q_row, R);
  QPG_LAUNCH_CHECK("l2_normalize_rows_kernel<gather>");
  return QPG_OK;
}

// ---------------------------------------------------------------------------------------------
// WavLM track resampling, 199 -> 180 frames: torch.nn.functional.interpolate(mode='linear',
// align_corners=True) in f32 (data_processing.py:258-261), bit-exact with torch's CPU kernel:
//   src = scale * t (f32), i0 = floor(src), l1 = src - i0, l0 = 1 - l1, out = fma(l0, x[i0], l1 * x[i1])
// (the l1*x1 product is rounded on its own, then fused with l0*x0 — verified against torch 2.10 on CPU,
// tests/test_gpu_matching.py::test_device_resample_bitexact).  HBM-bound: reads N*Tin*F*4 B once.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wavlm_resample_kernel(const float* __restrict__ x, int64_t N, int Tin, int F,
                                                             int Tout, float scale, float* __restrict__ out) {
  const int64_t row = blockIdx.x;                   // (n, t_out)
  const int64_t n = row / Tout;
  const int t = (int)(row - n * Tout);
  const float src = f_mul(scale, (float)t);
  int i0 = (int)src;                                 // src >= 0: truncation == floor
  if (i0 > Tin - 1) i0 = Tin - 1;
  const int i1 = i0 + 1 < Tin ? i0 + 1 : Tin - 1;
  const float l1 = f_sub(src, (float)i0), l0 = f_sub(1.f, l1);
  const float* p0 = x + (n * Tin + i0) * F;
  const float* p1 = x + (n * Tin + i1) * F;
  float* o = out + row * F;
  for (int e = threadIdx.x; e < F; e += blockDim.x) o[e] = fmaf(l0, p0[e], f_mul(l1, p1[e]));
}

extern "C" int qpg_wavlm_resample_f32(qpg_ctx* ctx, void* stream, const float* x, int64_t N, int Tin, int F, int Tout,
                                      float* out) {
  QPG_REQUIRE(ctx && x && out && N >= 0 && Tin > 0 && F > 0 && Tout > 0, "qpg_wavlm_resample_f32: bad argument");
  QPG_REQUIRE(N * Tout < 0x7fffffffll, "qpg_wavlm_resample_f32: too many rows for one launch");
  if (N == 0) return QPG_OK;
  // torch: area_pixel_compute_scale<float>(in, out, align_corners=true) = (in-1)/(out-1) in float (0 if out == 1)
  const float scale = Tout > 1 ? (float)(Tin - 1) / (float)(Tout - 1) : 0.f;
  hipLaunchKernelGGL(wavlm_resample_kernel, dim3((unsigned)(N * Tout)), dim3(256), 0, qpg_stream(stream), x, N, Tin, F,
                     Tout, scale, out);
  QPG_LAUNCH_CHECK("wavlm_resample_kernel");
  return QPG_OK;
}
