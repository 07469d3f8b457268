// Token-sized GEMMs (M <= 64 rows: transformer / projection / discriminator
// linears at B*(S+1) <= 40 rows).  An MFMA tile would be >75 % padding and the
// launch is latency-bound, so these run as wave-per-output-column dot products:
// the weight row streams once (coalesced 16-B loads), the <=64 activation rows
// come from L1/L2.  The weight gradient + bias gradient are one launch that adds
// straight into the parameter gradient (no packed staging: plain linears have no
// spectral norm).
#include "ast_common.h"
#include "../../include/ast_hip.h"

namespace {
constexpr int RG = 8;   // activation rows per register group

// y[m][n] = act(sum_k x[m][k] * w[n][k] + b[n]);  w row pitch = ldw.  One wave per column n.
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int M, int N,
                                                           int K, int ldw, int ldy, int relu) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* wr = w + (size_t)n * ldw;
  const float bn = bias ? bias[n] : 0.f;
  for (int m0 = 0; m0 < M; m0 += RG) {
    float acc[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) acc[r] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
#pragma unroll
      for (int r = 0; r < RG; ++r) {
        if (m0 + r < M) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)(m0 + r) * K + k);
          acc[r] += wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      const float s = wave_sum(acc[r]) + bn;
      if (lane == 0 && m0 + r < M) y[(size_t)(m0 + r) * ldy + n] = relu ? fmaxf(s, 0.f) : s;
    }
  }
}

// dW[n][k] += sum_m dy[m][n] x[m][k] ; db[n] += sum_m dy[m][n].  block = (one n, 256 k's)
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            float* __restrict__ dW, float* __restrict__ db, int M, int N, int K,
                                                            int lddy, int ldw) {
  const int n = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  float a = 0.f, bsum = 0.f;
  for (int m = 0; m < M; ++m) {
    const float g = dy[(size_t)m * lddy + n];
    bsum += g;
    if (k < K) a += g * x[(size_t)m * K + k];
  }
  if (k < K) dW[(size_t)n * ldw + k] += a;
  if (db && k == 0) db[n] += bsum;
}
}  // namespace

extern "C" int ast_skinny_gemm(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int ldw, int ldy,
                               int relu, void* stream) {
  if (!x || !w || !y || M < 1 || M > 64 || N < 1 || K < 4 || (K & 3) || (ldw & 3)) AST_FAIL("ast_skinny_gemm: bad args M=%d N=%d K=%d", M, N, K);
  if ((((uintptr_t)x) | ((uintptr_t)w)) & 15) AST_FAIL("ast_skinny_gemm: operands must be 16-byte aligned");
  hipLaunchKernelGGL(skinny_gemm_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, M, N, K, ldw, ldy, relu);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_linear_wgrad(const float* dy, const float* x, float* dW, float* db, int M, int N, int K, int lddy, int ldw,
                                void* stream) {
  if (!dy || !x || !dW || M < 1 || N < 1 || K < 1) AST_FAIL("ast_linear_wgrad: bad args");
  hipLaunchKernelGGL(linear_wgrad_kernel, dim3((K + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, dy, x, dW, db, M, N, K, lddy, ldw);
  AST_CHECK_LAUNCH();
  return 0;
}
