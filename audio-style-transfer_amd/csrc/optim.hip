// Optimiser step over flat f32 parameter/gradient buffers (HBM-bound streaming):
// global gradient-norm clipping + Adam in two launches for the whole model.
#include "ast_common.h"
#include "../../include/ast_hip.h"

namespace {
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
  __shared__ float red[17];
  float q = 0.f;
  const size_t n4 = n / 4;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  // four 16-byte loads in flight per thread (the straight loop waited for each: 3 TB/s on 124 MB)
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float q1 = 0.f, q2 = 0.f, q3 = 0.f;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
    q += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
    q1 += b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
    q2 += c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3];
    q3 += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
  }
  for (; i < n4; i += stride) {
    const f32x4 v = x4[i];
    q += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  q += (q1 + q2) + q3;
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = x[n4 * 4 + threadIdx.x]; q += v * v; }
  q = block_sum(q, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(out, q);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd,
                                                    const int64_t* __restrict__ d_step, const float* __restrict__ gnorm_sq, float max_norm,
                                                    const float* __restrict__ d_hyper) {
  if (d_hyper) { lr = d_hyper[0]; max_norm = d_hyper[1]; }         // [lr, max_norm] read at run time: a replayed graph follows a schedule
  const float step = (float)(*d_step);
  const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
  float clip = 1.f;
  if (gnorm_sq && max_norm > 0.f) clip = fminf(1.f, max_norm / (sqrtf(gnorm_sq[0]) + 1e-6f));
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i] * clip;
    if (wd != 0.f) gi += wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
  }
}
__global__ void counter_incr_kernel(int64_t* c) { if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += 1; }

// ---- step scalars that live on the device (LR schedule, clip norm, ramped loss weights under hipGraph replay) -------------
struct HostVals { float v[AST_MAX_STEP_SCALARS]; };
__global__ void set_values_kernel(float* __restrict__ dst, const HostVals hv, const int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = hv.v[threadIdx.x];
}
struct WsumArgs { const float* term[AST_MAX_STEP_SCALARS]; int widx[AST_MAX_STEP_SCALARS]; int n; };
// out[0] = sum_i w[widx_i] * term_i[0]   (terms added in argument order: bit-reproducible)
__global__ void weighted_sum_kernel(const WsumArgs a, const float* __restrict__ w, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  float s = 0.f;
  for (int i = 0; i < a.n; ++i) s += (a.widx[i] >= 0 ? w[a.widx[i]] : 1.f) * a.term[i][0];
  out[0] = s;
}
// grads[i] = g[0] * w[widx_i]
__global__ void weighted_sum_bwd_kernel(const float* __restrict__ g, const WsumArgs a, const float* __restrict__ w, float* __restrict__ grads) {
  if ((int)threadIdx.x < a.n) grads[threadIdx.x] = g[0] * (a.widx[threadIdx.x] >= 0 ? w[a.widx[threadIdx.x]] : 1.f);
}
}  // namespace

extern "C" int ast_set_values(float* dst, const float* host_vals, int n, void* stream) {
  if (!dst || !host_vals || n < 1 || n > AST_MAX_STEP_SCALARS) AST_FAIL("ast_set_values: 1..%d values", AST_MAX_STEP_SCALARS);
  HostVals hv;
  for (int i = 0; i < AST_MAX_STEP_SCALARS; ++i) hv.v[i] = i < n ? host_vals[i] : 0.f;
  hipLaunchKernelGGL(set_values_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, hv, n);
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_weighted_sum(const float* const* terms, const int* widx, int n, const float* weights, float* out, void* stream) {
  if (!terms || !widx || !weights || !out || n < 1 || n > AST_MAX_STEP_SCALARS) AST_FAIL("ast_weighted_sum: 1..%d terms", AST_MAX_STEP_SCALARS);
  WsumArgs a;
  a.n = n;
  for (int i = 0; i < AST_MAX_STEP_SCALARS; ++i) {
    a.term[i] = i < n ? terms[i] : nullptr; a.widx[i] = i < n ? widx[i] : -1;
    if (i < n && (!terms[i] || widx[i] >= AST_MAX_STEP_SCALARS)) AST_FAIL("ast_weighted_sum: term %d: null pointer or weight index out of range", i);
  }
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, weights, out);
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_weighted_sum_bwd(const float* g, const int* widx, int n, const float* weights, float* grads, void* stream) {
  if (!g || !widx || !weights || !grads || n < 1 || n > AST_MAX_STEP_SCALARS) AST_FAIL("ast_weighted_sum_bwd: 1..%d terms", AST_MAX_STEP_SCALARS);
  WsumArgs a;
  a.n = n;
  for (int i = 0; i < AST_MAX_STEP_SCALARS; ++i) {
    a.term[i] = nullptr; a.widx[i] = i < n ? widx[i] : -1;
    if (i < n && widx[i] >= AST_MAX_STEP_SCALARS) AST_FAIL("ast_weighted_sum_bwd: weight index out of range");
  }
  hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g, a, weights, grads);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_sumsq(const float* x, int64_t n, float* out, void* stream) {
  if (!x || !out || n < 0) AST_FAIL("ast_sumsq: bad args");
  if (n == 0) return 0;
  if (((uintptr_t)x) & 15) AST_FAIL("ast_sumsq: x must be 16-byte aligned");
  // one same-address f32 atomic per workgroup, ~11 ns each once they queue up: 2048 workgroups 38 us, 1024 28 us,
  // 512 23 us, 256 21 us (5.9 TB/s) for the 31 M-parameter gradient (tools/sumsq_time.py)
  static const int max_blocks = getenv("AST_SUMSQ_BLOCKS") ? atoi(getenv("AST_SUMSQ_BLOCKS")) : 256;
  const int grid = (int)std::min<size_t>(((size_t)n / 4 + 255) / 256 + 1, (size_t)std::max(1, max_blocks));
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, out);
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                        const int64_t* d_step, const float* gnorm_sq, float max_norm, void* stream) {
  if (!p || !g || !m || !v || !d_step || n < 0) AST_FAIL("ast_adam: bad args");
  if (n == 0) return 0;
  const int grid = (int)std::min<size_t>(((size_t)n + 255) / 256, 4096);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (size_t)n, lr, b1, b2, eps, wd, d_step,
                     gnorm_sq, max_norm, (const float*)nullptr);
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* d_hyper, float b1, float b2, float eps, float wd,
                            const int64_t* d_step, const float* gnorm_sq, void* stream) {
  if (!p || !g || !m || !v || !d_step || !d_hyper || n < 0) AST_FAIL("ast_adam_dev: bad args");
  if (n == 0) return 0;
  const int grid = (int)std::min<size_t>(((size_t)n + 255) / 256, 4096);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (size_t)n, 0.f, b1, b2, eps, wd, d_step,
                     gnorm_sq, 0.f, d_hyper);
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_counter_incr(int64_t* c, void* stream) {
  if (!c) AST_FAIL("ast_counter_incr: null");
  hipLaunchKernelGGL(counter_incr_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, c);
  AST_CHECK_LAUNCH();
  return 0;
}
