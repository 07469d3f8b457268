// Spectral normalisation (one power iteration per training forward, torch
// nn/utils/spectral_norm.py:92-114) fused with packing of W/sigma into the two
// GEMM layouts, batched over ALL weights of a model in three launches
// (launch boundaries act as the grid-wide syncs between v, u and sigma), and
// the matching backward:  dW_orig = (dW - <dW, W/sigma> u v^T) / sigma.
#include "ast_common.h"
#include "../../include/ast_hip.h"

namespace {

__device__ __forceinline__ size_t woff(const ast_weight_desc_t& d, int co, int ci, int tap) {
  return (size_t)co * d.s_co + (size_t)ci * d.s_ci + tap;
}

// t[j] = sum_co W(co, j) u[co],  j = ci*KK + tap  -> scratch[Co + j]
// rows are split over grid.z (RZ chunks) and summed with atomics; scratch[Co..] is zeroed by sn_pack_kernel
// of the previous forward (and at allocation), so the launch needs no memset.
constexpr int RZ = 8;
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const ast_weight_desc_t* __restrict__ descs) {
  const ast_weight_desc_t d = descs[blockIdx.y];
  if (!d.u || !d.power_iter) return;
  const int ncols = d.Ci * d.KK;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= ncols) return;
  const int rows_per = (d.Co + RZ - 1) / RZ;
  const int c0 = blockIdx.z * rows_per, c1 = min(d.Co, c0 + rows_per);
  if (c0 >= c1) return;
  const int ci = j / d.KK, tap = j - ci * d.KK;
  float acc = 0.f;
  for (int co = c0; co < c1; ++co) acc += d.w[woff(d, co, ci, tap)] * d.u[co];
  unsafeAtomicAdd(d.scratch + d.Co + j, acc);
}

// v = t/|t| (training) ; s[co] = sum_j W(co,j) v[j] -> scratch[co]; one wave per row
__global__ __launch_bounds__(256) void sn_w_v_kernel(const ast_weight_desc_t* __restrict__ descs) {
  __shared__ float red[17];
  const ast_weight_desc_t d = descs[blockIdx.y];
  if (!d.u) return;
  const int ncols = d.Ci * d.KK;
  const int row0 = blockIdx.x * 4;
  if (row0 >= d.Co) return;                         // uniform per block
  float inv = 1.f;
  const float* vec = d.v;
  if (d.power_iter) {
    vec = d.scratch + d.Co;
    float q = 0.f;
    for (int j = threadIdx.x; j < ncols; j += 256) q += vec[j] * vec[j];
    inv = 1.f / fmaxf(sqrtf(block_sum(q, red)), 1e-12f);
    if (blockIdx.x == 0)
      for (int j = threadIdx.x; j < ncols; j += 256) d.v[j] = vec[j] * inv;
  }
  const int row = row0 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row < d.Co) {
    float acc = 0.f;
    for (int j = lane; j < ncols; j += 64) {
      const int ci = j / d.KK, tap = j - ci * d.KK;
      acc += d.w[woff(d, row, ci, tap)] * vec[j];
    }
    acc = wave_sum(acc) * inv;
    if (lane == 0) d.scratch[row] = acc;
  }
}

// sigma, u, then pack W/sigma into wf [Cop][KK][Cip] and wb [Cip][KK][Cop]
template <typename T>
__device__ void pack_body(const ast_weight_desc_t& d, float inv_sigma) {
  const size_t nf = (size_t)d.Cop * d.KK * d.Cip;
  const size_t stride = (size_t)gridDim.x * 256;
  if (d.wf) {
    T* wf = (T*)d.wf;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nf; i += stride) {
      const int ci = (int)(i % d.Cip);
      const size_t t = i / d.Cip;
      const int tap = (int)(t % d.KK), co = (int)(t / d.KK);
      wf[i] = (T)((co < d.Co && ci < d.Ci) ? d.w[woff(d, co, ci, tap)] * inv_sigma : 0.f);
    }
  }
  if (d.wb) {
    T* wb = (T*)d.wb;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nf; i += stride) {
      const int co = (int)(i % d.Cop);
      const size_t t = i / d.Cop;
      const int tap = (int)(t % d.KK), ci = (int)(t / d.KK);
      wb[i] = (T)((co < d.Co && ci < d.Ci) ? d.w[woff(d, co, ci, tap)] * inv_sigma : 0.f);
    }
  }
}

__global__ __launch_bounds__(256) void sn_pack_kernel(const ast_weight_desc_t* __restrict__ descs, const int* __restrict__ dtypes) {
  __shared__ float red[17];
  const ast_weight_desc_t d = descs[blockIdx.y];
  float inv_sigma = 1.f;
  if (d.u) {
    float sigma;
    if (d.power_iter) {
      float q = 0.f;
      for (int i = threadIdx.x; i < d.Co; i += 256) q += d.scratch[i] * d.scratch[i];
      const float nrm = sqrtf(block_sum(q, red));
      const float inv = 1.f / fmaxf(nrm, 1e-12f);
      sigma = nrm * nrm * inv;                      // u_new . (W v) with u_new = s / max(|s|, eps)
      if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < d.Co; i += 256) d.u[i] = d.scratch[i] * inv;
    } else {
      float q = 0.f;
      for (int i = threadIdx.x; i < d.Co; i += 256) q += d.scratch[i] * d.u[i];
      sigma = block_sum(q, red);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) d.sigma[0] = sigma;
    inv_sigma = 1.f / sigma;
  }
  __syncthreads();
  if (d.u && blockIdx.x == 0)                       // leave t = W^T u zeroed for the next forward's atomics
    for (int j = threadIdx.x; j < d.Ci * d.KK; j += 256) d.scratch[d.Co + j] = 0.f;
  if (dtypes[blockIdx.y] == AST_BF16) pack_body<bf16_t>(d, inv_sigma); else pack_body<float>(d, inv_sigma);
  if (d.dwp && d.power_iter) {                      // training forward: fresh gradient staging for this step
    const size_t nf = (size_t)d.Cop * d.KK * d.Cip;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nf; i += (size_t)gridDim.x * 256) d.dwp[i] = 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && d.inner) d.inner[0] = 0.f;
  }
}

// batched backward: inner[w] = <dWp, W>/sigma for spectral-normalised weights
__global__ __launch_bounds__(256) void flush_inner_kernel(const ast_weight_desc_t* __restrict__ descs) {
  __shared__ float red[17];
  const ast_weight_desc_t d = descs[blockIdx.y];
  if (!d.dwp || !d.u) return;
  const size_t n = (size_t)d.Co * d.Ci * d.KK;
  float q = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int tap = (int)(i % d.KK);
    const size_t t = i / d.KK;
    const int ci = (int)(t % d.Ci), co = (int)(t / d.Ci);
    const size_t pi = d.dwp_from_wb ? ((size_t)ci * d.KK + tap) * d.Cop + co : ((size_t)co * d.KK + tap) * d.Cip + ci;
    q += d.dwp[pi] * d.w[woff(d, co, ci, tap)];
  }
  q = block_sum(q, red);
  if (threadIdx.x == 0 && q != 0.f) unsafeAtomicAdd(d.inner, q / d.sigma[0]);
}

__global__ __launch_bounds__(256) void flush_unpack_kernel(const ast_weight_desc_t* __restrict__ descs) {
  const ast_weight_desc_t d = descs[blockIdx.y];
  if (!d.dwp || !d.grad) return;
  const size_t n = (size_t)d.Co * d.Ci * d.KK;
  const float inner = d.u ? d.inner[0] : 0.f;
  const float inv_sigma = d.u ? 1.f / d.sigma[0] : 1.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int tap = (int)(i % d.KK);
    const size_t t = i / d.KK;
    const int ci = (int)(t % d.Ci), co = (int)(t / d.Ci);
    const size_t pi = d.dwp_from_wb ? ((size_t)ci * d.KK + tap) * d.Cop + co : ((size_t)co * d.KK + tap) * d.Cip + ci;
    float gv = d.dwp[pi];
    if (d.u) gv = (gv - inner * d.u[co] * d.v[ci * d.KK + tap]) * inv_sigma;
    d.grad[woff(d, co, ci, tap)] += gv;
  }
}

// ---- backward --------------------------------------------------------------------
// inner = <dWp, W>/sigma  -> scratch[0] (zeroed by host)
__global__ __launch_bounds__(256) void wgrad_inner_kernel(const float* __restrict__ dwp, int from_wb, const float* __restrict__ w,
                                                           const float* __restrict__ sigma, float* scratch, int Co, int Ci, int KK,
                                                           int s_co, int s_ci, int Cop, int Cip) {
  __shared__ float red[17];
  const size_t n = (size_t)Co * Ci * KK;
  float q = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int tap = (int)(i % KK);
    const size_t t = i / KK;
    const int ci = (int)(t % Ci), co = (int)(t / Ci);
    const size_t pi = from_wb ? ((size_t)ci * KK + tap) * Cop + co : ((size_t)co * KK + tap) * Cip + ci;
    q += dwp[pi] * w[(size_t)co * s_co + (size_t)ci * s_ci + tap];
  }
  q = block_sum(q, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(scratch, q / sigma[0]);
}

__global__ __launch_bounds__(256) void wgrad_unpack_kernel(const float* __restrict__ dwp, int from_wb, const float* __restrict__ u,
                                                            const float* __restrict__ v, const float* __restrict__ sigma,
                                                            const float* __restrict__ scratch, float* __restrict__ g_orig, int Co, int Ci,
                                                            int KK, int s_co, int s_ci, int Cop, int Cip) {
  const size_t n = (size_t)Co * Ci * KK;
  const float inner = u ? scratch[0] : 0.f;
  const float inv_sigma = u ? 1.f / sigma[0] : 1.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int tap = (int)(i % KK);
    const size_t t = i / KK;
    const int ci = (int)(t % Ci), co = (int)(t / Ci);
    const size_t pi = from_wb ? ((size_t)ci * KK + tap) * Cop + co : ((size_t)co * KK + tap) * Cip + ci;
    float gv = dwp[pi];
    if (u) gv = (gv - inner * u[co] * v[ci * KK + tap]) * inv_sigma;
    g_orig[(size_t)co * s_co + (size_t)ci * s_ci + tap] += gv;
  }
}

}  // namespace

// dtypes: device int array [n] giving the packed dtype of each descriptor.
extern "C" int ast_weights_prepare_v(const ast_weight_desc_t* descs, const int* dtypes, int n, int max_co, int max_cols,
                                     long max_packed, void* stream) {
  if (!descs || !dtypes || n <= 0 || max_co <= 0 || max_cols <= 0) AST_FAIL("ast_weights_prepare: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sn_wt_u_kernel, dim3((max_cols + 255) / 256, n, RZ), dim3(256), 0, s, descs);
  hipLaunchKernelGGL(sn_w_v_kernel, dim3((max_co + 3) / 4, n), dim3(256), 0, s, descs);
  const int nb = (int)std::max(1L, std::min(64L, (max_packed + 2047) / 2048));
  hipLaunchKernelGGL(sn_pack_kernel, dim3(nb, n), dim3(256), 0, s, descs, dtypes);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_weight_grads_flush_v(const ast_weight_desc_t* descs, int n, long max_elems, void* stream) {
  if (!descs || n <= 0) AST_FAIL("ast_weight_grads_flush_v: bad args");
  hipStream_t s = (hipStream_t)stream;
  const int nb = (int)std::max(1L, std::min(64L, (max_elems + 4095) / 4096));
  hipLaunchKernelGGL(flush_inner_kernel, dim3(nb, n), dim3(256), 0, s, descs);
  hipLaunchKernelGGL(flush_unpack_kernel, dim3(nb, n), dim3(256), 0, s, descs);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_weight_grad_unpack(const float* dwp, int from_wb, const float* w, const float* u, const float* v,
                                      const float* sigma, float* g_orig, int Co, int Ci, int KK, int s_co, int s_ci, int Cop,
                                      int Cip, float* scratch, void* stream) {
  if (!dwp || !w || !g_orig || (u && (!v || !sigma || !scratch))) AST_FAIL("ast_weight_grad_unpack: bad args");
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)Co * Ci * KK;
  const int nb = (int)std::max<size_t>(1, std::min<size_t>(512, (n + 1023) / 1024));
  if (u) {
    AST_HIP(hipMemsetAsync(scratch, 0, sizeof(float), s));
    hipLaunchKernelGGL(wgrad_inner_kernel, dim3(nb), dim3(256), 0, s, dwp, from_wb, w, sigma, scratch, Co, Ci, KK, s_co, s_ci, Cop, Cip);
  }
  hipLaunchKernelGGL(wgrad_unpack_kernel, dim3(nb), dim3(256), 0, s, dwp, from_wb, u, v, sigma, scratch, g_orig, Co, Ci, KK, s_co,
                     s_ci, Cop, Cip);
  AST_CHECK_LAUNCH();
  return 0;
}
