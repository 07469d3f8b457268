// Layout conversion at the module boundary, pooling / resampling with the exact
// torch index rules, the tiny-sequence attention core and elementwise helpers.
#include "ast_common.h"
#include "../../include/ast_hip.h"

namespace {

int grid_for(size_t n, int block = 256) { return (int)std::min<size_t>((n + block - 1) / block, 256 * 16); }

// ---- NCHW f32 (strided) -> NHWC T, channels zero padded to Cp -------------------
// one thread per output pixel: reads C planes (coalesced along w), writes Cp/8 units
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int C, int H, int W, int64_t sn,
                                    int64_t sc, int64_t sh, int Cp, size_t npix) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const size_t t = i / W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    const float* px = x + n * sn + h * sh + w;
    for (int u = 0; u < Cp; u += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (u + k) < C ? px[(u + k) * sc] : 0.f;
      U8<T>::store(y + i * Cp + u, v);
    }
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int C, int H, int W, int Cp, size_t npix) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    const size_t hw = (size_t)H * W;
    const size_t n = i / hw, p = i % hw;
    for (int c = 0; c < C; ++c) y[(n * C + c) * hw + p] = (float)x[i * Cp + c];
  }
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (TO)(float)x[i];
}

// ---- adaptive average pooling (NHWC) --------------------------------------------
__device__ __forceinline__ int bin_start(int i, int in, int out) { return (i * in) / out; }
__device__ __forceinline__ int bin_end(int i, int in, int out) { return ((i + 1) * in + out - 1) / out; }

template <typename T>
__global__ void adaptive_pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C, int Ho, int Wo, size_t units) {
  const int U = C >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < units; i += (size_t)gridDim.x * blockDim.x) {
    const int u = (int)(i % U);
    size_t t = i / U;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const int h0 = bin_start(ho, H, Ho), h1 = bin_end(ho, H, Ho), w0 = bin_start(wo, W, Wo), w1 = bin_end(wo, W, Wo);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int h = h0; h < h1; ++h)
      for (int w = w0; w < w1; ++w) {
        float v[8];
        U8<T>::load(x + (((size_t)n * H + h) * W + w) * C + u * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += v[k];
      }
    const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] *= inv;
    U8<T>::store(y + i * 8, s);
  }
}

template <typename T>
__global__ void adaptive_pool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int H, int W, int C, int Ho, int Wo, size_t units) {
  const int U = C >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < units; i += (size_t)gridDim.x * blockDim.x) {
    const int u = (int)(i % U);
    size_t t = i / U;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int ilo = max(0, (h * Ho) / H - 1), ihi = min(Ho - 1, ((h + 1) * Ho + H - 1) / H);
    const int jlo = max(0, (w * Wo) / W - 1), jhi = min(Wo - 1, ((w + 1) * Wo + W - 1) / W);
    for (int io = ilo; io <= ihi; ++io) {
      const int h0 = bin_start(io, H, Ho), h1 = bin_end(io, H, Ho);
      if (h < h0 || h >= h1) continue;
      for (int jo = jlo; jo <= jhi; ++jo) {
        const int w0 = bin_start(jo, W, Wo), w1 = bin_end(jo, W, Wo);
        if (w < w0 || w >= w1) continue;
        float v[8];
        U8<T>::load(dy + (((size_t)n * Ho + io) * Wo + jo) * C + u * 8, v);
        const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += v[k] * inv;
      }
    }
    U8<T>::store(dx + i * 8, s);
  }
}

// ---- bilinear, align_corners=False: src = max((o+0.5)*in/out - 0.5, 0) ----------
__device__ __forceinline__ void bil_src(int o, float scale, int in, int& i0, int& i1, float& lam) {
  const float s = fmaxf((o + 0.5f) * scale - 0.5f, 0.f);
  i0 = min((int)s, in - 1);
  i1 = min(i0 + 1, in - 1);
  lam = s - (float)i0;
}

template <typename T>
__global__ void bilinear_fwd_kernel(const T* __restrict__ x, float* __restrict__ y, int C, int Cp, int H, int W, int Ho, int Wo, size_t npix) {
  const float sh = (float)H / Ho, sw = (float)W / Wo;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    const int wo = (int)(i % Wo);
    const size_t t = i / Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    int h0, h1, w0, w1; float lh, lw;
    bil_src(ho, sh, H, h0, h1, lh);
    bil_src(wo, sw, W, w0, w1, lw);
    const T* b = x + (size_t)n * H * W * Cp;
    for (int c = 0; c < C; ++c) {
      const float v00 = (float)b[((size_t)h0 * W + w0) * Cp + c], v01 = (float)b[((size_t)h0 * W + w1) * Cp + c];
      const float v10 = (float)b[((size_t)h1 * W + w0) * Cp + c], v11 = (float)b[((size_t)h1 * W + w1) * Cp + c];
      y[((size_t)n * C + c) * Ho * Wo + (size_t)ho * Wo + wo] =
          (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11);
    }
  }
}

// gather form of the backward: every input pixel sums the output pixels that tap it
template <typename T>
__global__ void bilinear_bwd_kernel(const float* __restrict__ dy, T* __restrict__ dx, int C, int Cp, int H, int W, int Ho, int Wo, size_t npix) {
  const float sh = (float)H / Ho, sw = (float)W / Wo;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const size_t t = i / W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    const int olo_h = max(0, (int)floorf((h - 1 + 0.5f) / sh - 0.5f) - 1), ohi_h = min(Ho - 1, (int)ceilf((h + 1 + 0.5f) / sh - 0.5f) + 1);
    const int olo_w = max(0, (int)floorf((w - 1 + 0.5f) / sw - 0.5f) - 1), ohi_w = min(Wo - 1, (int)ceilf((w + 1 + 0.5f) / sw - 0.5f) + 1);
    float a0 = 0.f, a1 = 0.f;   // C <= 2 (real / imaginary planes)
    // the column weights do not depend on the output row: computed once per pixel (up to 10 candidate columns in registers; the
    // 5 x 5 candidate window used to evaluate bil_src 30 times per pixel -- 32 us on the step's chain for a 1:1 resize)
    constexpr int MAXC = 10;
    const int ncw = ohi_w - olo_w + 1;
    float cwv[MAXC];
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      cwv[k] = 0.f;
      if (k < ncw) {
        int w0, w1; float lw;
        bil_src(olo_w + k, sw, W, w0, w1, lw);
        cwv[k] = (w0 == w ? 1.f - lw : 0.f) + (w1 == w ? lw : 0.f);
      }
    }
    for (int oh = olo_h; oh <= ohi_h; ++oh) {
      int h0, h1; float lh;
      bil_src(oh, sh, H, h0, h1, lh);
      const float ch = (h0 == h ? 1.f - lh : 0.f) + (h1 == h ? lh : 0.f);
      if (ch == 0.f) continue;
      const float* r0 = dy + ((size_t)n * C + 0) * Ho * Wo + (size_t)oh * Wo + olo_w;
      const float* r1 = r0 + (size_t)Ho * Wo;
      if (ncw <= MAXC) {
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
          if (k < ncw && cwv[k] != 0.f) {
            a0 += ch * cwv[k] * r0[k];
            if (C > 1) a1 += ch * cwv[k] * r1[k];
          }
        }
      } else {
        for (int ow = olo_w; ow <= ohi_w; ++ow) {
          int w0, w1; float lw;
          bil_src(ow, sw, W, w0, w1, lw);
          const float cw = (w0 == w ? 1.f - lw : 0.f) + (w1 == w ? lw : 0.f);
          if (cw == 0.f) continue;
          a0 += ch * cw * r0[ow - olo_w];
          if (C > 1) a1 += ch * cw * r1[ow - olo_w];
        }
      }
    }
    for (int u = 0; u < Cp; u += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = 0.f;
      if (u == 0) { v[0] = a0; if (C > 1) v[1] = a1; }
      U8<T>::store(dx + i * Cp + u, v);
    }
  }
}

// ---- attention core for L <= 16 tokens: one wave per (batch, head), lane = feature
// ML = the compile-time key-count bound (4, 8 or 16): every per-key array is indexed by a fully unrolled loop under a
// `j < Lk` predicate, so it lives in registers (runtime-indexed arrays went to scratch memory).
constexpr int MAXL = 16;
template <int ML>
__global__ __launch_bounds__(64) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                       float* __restrict__ o, float* __restrict__ probs, int H, int Lq, int Lk, int dh,
                                                       int ldq, int ldk, int ldo, int causal, const float* __restrict__ drop,
                                                       float pdrop, uint64_t seed, const int64_t* __restrict__ d_offset) {
  // attention-probability dropout: either a precomputed mask (`drop`) or, with pdrop > 0, the mask value is DRAWN here
  // from (seed, step counter, element index) -- the backward kernel draws the same values again, so no mask tensor
  // and no mask kernel exist
  const uint64_t dbase = pdrop > 0.f ? mix64(seed ^ mix64((uint64_t)(d_offset ? *d_offset : 0))) : 0;
  const float dkeep = 1.f / (1.f - pdrop);
  const int b = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
  const float scale = rsqrtf((float)dh);
  float kv[ML], vv[ML];
#pragma unroll
  for (int j = 0; j < ML; ++j) {
    if (j >= Lk) continue;
    kv[j] = lane < dh ? k[((size_t)b * Lk + j) * ldk + h * dh + lane] : 0.f;
    vv[j] = lane < dh ? v[((size_t)b * Lk + j) * ldk + h * dh + lane] : 0.f;
  }
  for (int i = 0; i < Lq; ++i) {
    const float qi = lane < dh ? q[((size_t)b * Lq + i) * ldq + h * dh + lane] * scale : 0.f;
    float s[ML];
    float mx = -INFINITY;
  #pragma unroll
    for (int j = 0; j < ML; ++j) {
      if (j >= Lk) continue;
      s[j] = wave_sum(qi * kv[j]);
      if (causal && j > i) s[j] = -INFINITY;
      mx = fmaxf(mx, s[j]);
    }
    float den = 0.f;
  #pragma unroll
    for (int j = 0; j < ML; ++j) if (j < Lk) { s[j] = __expf(s[j] - mx); den += s[j]; }
    float acc = 0.f;
    const size_t pbase = (((size_t)b * H + h) * Lq + i) * Lk;
  #pragma unroll
    for (int j = 0; j < ML; ++j) {
      if (j >= Lk) continue;
      const float p = s[j] / den;
      if (lane == 0) probs[pbase + j] = p;
      acc += (drop ? p * drop[pbase + j] : (pdrop > 0.f ? p * dropout_keep(dbase, pbase + j, pdrop, dkeep) : p)) * vv[j];
    }
    if (lane < dh) o[((size_t)b * Lq + i) * ldo + h * dh + lane] = acc;
  }
}

template <int ML>
__global__ __launch_bounds__(64) void attn_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, const float* __restrict__ probs, float* __restrict__ dq,
                                                       float* __restrict__ dk, float* __restrict__ dv, int H, int Lq, int Lk, int dh, int ldq,
                                                       int ldk, int ldo, const float* __restrict__ drop, float pdrop, uint64_t seed,
                                                       const int64_t* __restrict__ d_offset) {
  const uint64_t dbase = pdrop > 0.f ? mix64(seed ^ mix64((uint64_t)(d_offset ? *d_offset : 0))) : 0;
  const float dkeep = 1.f / (1.f - pdrop);
  const int b = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
  const float scale = rsqrtf((float)dh);
  float kv[ML], vv[ML], dkv[ML], dvv[ML];
#pragma unroll
  for (int j = 0; j < ML; ++j) {
    if (j >= Lk) continue;
    kv[j] = lane < dh ? k[((size_t)b * Lk + j) * ldk + h * dh + lane] : 0.f;
    vv[j] = lane < dh ? v[((size_t)b * Lk + j) * ldk + h * dh + lane] : 0.f;
    dkv[j] = 0.f; dvv[j] = 0.f;
  }
  for (int i = 0; i < Lq; ++i) {
    const float qi = lane < dh ? q[((size_t)b * Lq + i) * ldq + h * dh + lane] : 0.f;
    const float doi = lane < dh ? dout[((size_t)b * Lq + i) * ldo + h * dh + lane] : 0.f;
    const size_t pbase = (((size_t)b * H + h) * Lq + i) * Lk;
    float dp[ML], p[ML];
    float dot = 0.f;
  #pragma unroll
    for (int j = 0; j < ML; ++j) {
      if (j >= Lk) continue;
      p[j] = probs[pbase + j];
      const float m = drop ? drop[pbase + j] : (pdrop > 0.f ? dropout_keep(dbase, pbase + j, pdrop, dkeep) : 1.f);
      dvv[j] += p[j] * m * doi;
      dp[j] = wave_sum(doi * vv[j]) * m;
      dot += dp[j] * p[j];
    }
    float dqi = 0.f;
  #pragma unroll
    for (int j = 0; j < ML; ++j) {
      if (j >= Lk) continue;
      const float ds = p[j] * (dp[j] - dot) * scale;
      dqi += ds * kv[j];
      dkv[j] += ds * qi;
    }
    if (lane < dh) dq[((size_t)b * Lq + i) * ldq + h * dh + lane] = dqi;
  }
  if (lane < dh) {
#pragma unroll
    for (int j = 0; j < ML; ++j) {
      if (j >= Lk) continue;
      dk[((size_t)b * Lk + j) * ldk + h * dh + lane] = dkv[j];
      dv[((size_t)b * Lk + j) * ldk + h * dh + lane] = dvv[j];
    }
  }
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (T)((float)a[i] + (float)b[i]);
}
template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dx[i] = (float)y[i] > 0.f ? dy[i] : (T)0.f;
}
template <typename T>
__global__ void mul_kernel(const T* __restrict__ a, const float* __restrict__ m, T* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (T)((float)a[i] * m[i]);
}
// out[c] += sum_r x[r][c]  (bias gradients); one thread per column per row chunk
template <typename T>
__global__ void colsum_acc_kernel(const T* __restrict__ x, size_t rows, int C, int Creal, float* __restrict__ out, size_t rows_per_blk) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Creal) return;
  const size_t r0 = (size_t)blockIdx.y * rows_per_blk, r1 = r0 + rows_per_blk < rows ? r0 + rows_per_blk : rows;
  float a = 0.f;
  for (size_t r = r0; r < r1; ++r) a += (float)x[r * C + c];
  unsafeAtomicAdd(out + c, a);
}
// small C (<= 64, a multiple of 8): the tensor is read as a flat stream of 8-element chunks, four 16-byte loads in flight
// per thread; a thread always meets the same column octet (the grid stride is a multiple of C/8), keeps 8 column sums,
// and the workgroup combines them through LDS before ONE atomic per column.  (The first version read one 2-byte
// element per thread per step: 31 us for the 67 MB of a 2 M-pixel x 16-channel bias gradient.)
template <typename T>
__global__ __launch_bounds__(256) void colsum_small_kernel(const T* __restrict__ x, size_t n8, int C, int Creal, float* __restrict__ out) {
  __shared__ float red[256][9];
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n8; i += 4 * stride) {
    float a[8], b[8], c[8], d[8];
    U8<T>::load(x + i * 8, a); U8<T>::load(x + (i + stride) * 8, b);
    U8<T>::load(x + (i + 2 * stride) * 8, c); U8<T>::load(x + (i + 3 * stride) * 8, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += (a[e] + b[e]) + (c[e] + d[e]);
  }
  for (; i < n8; i += stride) {
    float a[8];
    U8<T>::load(x + i * 8, a);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += a[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
  __syncthreads();
  if ((int)threadIdx.x < Creal) {
    const int no = C >> 3, oct = threadIdx.x >> 3, e = threadIdx.x & 7;
    float t = 0.f;
    for (int q = oct; q < 256; q += no) t += red[q][e];
    unsafeAtomicAdd(out + threadIdx.x, t);
  }
}
__global__ void dropout_mask_kernel(float* __restrict__ mask, size_t n, float p, uint64_t seed, const int64_t* __restrict__ d_offset) {
  const uint64_t base = mix64(seed ^ mix64((uint64_t)(d_offset ? *d_offset : 0)));
  const float keep = 1.f / (1.f - p);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    mask[i] = dropout_keep(base, i, p, keep);
  }
}

__global__ void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ mask, size_t n, float p,
                                   uint64_t seed, const int64_t* __restrict__ d_offset) {
  const uint64_t base = mix64(seed ^ mix64((uint64_t)(d_offset ? *d_offset : 0)));
  const float keep = 1.f / (1.f - p);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float m = dropout_keep(base, i, p, keep);
    mask[i] = m;
    y[i] = x[i] * m;
  }
}

// Y[o][d] = sum_i A[o][i] X[i][d] for a small dense coefficient matrix A (R_in <= 8192): per-class means of embeddings
// (style_encoder.py:243-253), the per-row class-prototype gather (new_decoder.py:214-223 callers), means over the
// section axis (losses.py:88, 142) and their backward passes (A transposed).  One workgroup per output row.
__global__ __launch_bounds__(256) void rowmix_kernel(const float* __restrict__ A, const float* __restrict__ X, float* __restrict__ Y,
                                                     int R_in, int D) {
  const int o = blockIdx.x;
  extern __shared__ float a[];                                  // R_in coefficients of this output row
  for (int i = threadIdx.x; i < R_in; i += 256) a[i] = A[(size_t)o * R_in + i];
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
    for (int i = 0; i < R_in; ++i) {
      const float w = a[i];
      if (w != 0.f) acc += w * X[(size_t)i * D + d];
    }
    Y[(size_t)o * D + d] = acc;
  }
}

}  // namespace

extern "C" int ast_dropout_fwd(const float* x, float* y, float* mask, int64_t n, float p, uint64_t seed, const int64_t* d_offset,
                               void* stream) {
  if (!x || !y || !mask || n < 0 || p < 0.f || p >= 1.f) AST_FAIL("ast_dropout_fwd: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, x, y, mask, (size_t)n, p, seed,
                     d_offset);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_nchw_to_nhwc(const float* x, void* y, int N, int C, int H, int W, int64_t sn, int64_t sc, int64_t sh, int Cp,
                                int dtype, void* stream) {
  if (!x || !y || C > Cp || (Cp & 7) || N <= 0 || H <= 0 || W <= 0) AST_FAIL("ast_nchw_to_nhwc: bad args");
  const size_t npix = (size_t)N * H * W;
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, x,
                                            (T*)y, C, H, W, sn, sc, sh, Cp, npix));
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_nhwc_to_nchw(const void* x, float* y, int N, int C, int H, int W, int Cp, int dtype, void* stream) {
  if (!x || !y || C > Cp) AST_FAIL("ast_nhwc_to_nchw: bad args");
  const size_t npix = (size_t)N * H * W;
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream,
                                            (const T*)x, y, C, H, W, Cp, npix));
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_cast(const void* x, int dtype_in, void* y, int dtype_out, int64_t n, void* stream) {
  if (!x || !y || n < 0) AST_FAIL("ast_cast: bad args");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  dim3 g(grid_for((size_t)n)), b(256);
  if (dtype_in == AST_F32 && dtype_out == AST_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, b, 0, s, (const float*)x, (bf16_t*)y, (size_t)n);
  else if (dtype_in == AST_BF16 && dtype_out == AST_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)x, (float*)y, (size_t)n);
  else if (dtype_in == AST_F32 && dtype_out == AST_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, s, (const float*)x, (float*)y, (size_t)n);
  else if (dtype_in == AST_BF16 && dtype_out == AST_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)x, (bf16_t*)y, (size_t)n);
  else AST_FAIL("ast_cast: bad dtypes %d -> %d", dtype_in, dtype_out);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_adaptive_pool_fwd(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int dtype, void* stream) {
  if (!x || !y || (C & 7) || Ho <= 0 || Wo <= 0) AST_FAIL("ast_adaptive_pool_fwd: bad args");
  const size_t units = (size_t)N * Ho * Wo * (C >> 3);
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((adaptive_pool_fwd_kernel<T>), dim3(grid_for(units)), dim3(256), 0, (hipStream_t)stream,
                                            (const T*)x, (T*)y, H, W, C, Ho, Wo, units));
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_adaptive_pool_bwd(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, int dtype, void* stream) {
  if (!dy || !dx || (C & 7)) AST_FAIL("ast_adaptive_pool_bwd: bad args");
  const size_t units = (size_t)N * H * W * (C >> 3);
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((adaptive_pool_bwd_kernel<T>), dim3(grid_for(units)), dim3(256), 0, (hipStream_t)stream,
                                            (const T*)dy, (T*)dx, H, W, C, Ho, Wo, units));
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_bilinear_fwd(const void* x, float* y, int N, int C, int Cp, int H, int W, int Ho, int Wo, int dtype, void* stream) {
  if (!x || !y || C > Cp || C > 8) AST_FAIL("ast_bilinear_fwd: bad args");
  const size_t npix = (size_t)N * Ho * Wo;
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((bilinear_fwd_kernel<T>), dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream,
                                            (const T*)x, y, C, Cp, H, W, Ho, Wo, npix));
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_bilinear_bwd(const float* dy, void* dx, int N, int C, int Cp, int H, int W, int Ho, int Wo, int dtype, void* stream) {
  if (!dy || !dx || C > Cp || C > 2) AST_FAIL("ast_bilinear_bwd: bad args (C<=2)");
  const size_t npix = (size_t)N * H * W;
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((bilinear_bwd_kernel<T>), dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, dy,
                                            (T*)dx, C, Cp, H, W, Ho, Wo, npix));
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_attn_fwd_p(const float* q, const float* k, const float* v, float* o, float* probs, int B, int H, int Lq, int Lk,
                              int dh, int ldq, int ldk, int ldo, int causal, const float* drop_mask, float p, uint64_t seed,
                              const int64_t* d_offset, void* stream) {
  if (!q || !k || !v || !o || !probs || p < 0.f || p >= 1.f) AST_FAIL("ast_attn_fwd: bad args");
  if (Lq < 1 || Lk < 1 || Lq > MAXL || Lk > MAXL || dh < 1 || dh > 64) AST_FAIL("ast_attn_fwd: needs 1<=L<=%d and dh<=64 (Lq=%d Lk=%d dh=%d)", MAXL, Lq, Lk, dh);
#define AST_ATTN_FWD(ML_)                                                                                                          \
  hipLaunchKernelGGL(attn_fwd_kernel<ML_>, dim3(B * H), dim3(64), 0, (hipStream_t)stream, q, k, v, o, probs, H, Lq, Lk, dh, ldq, ldk, \
                     ldo, causal, drop_mask, p, seed, d_offset)
  if (Lk <= 4) AST_ATTN_FWD(4); else if (Lk <= 8) AST_ATTN_FWD(8); else AST_ATTN_FWD(16);
#undef AST_ATTN_FWD
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_attn_fwd(const float* q, const float* k, const float* v, float* o, float* probs, int B, int H, int Lq, int Lk,
                            int dh, int ldq, int ldk, int ldo, int causal, const float* drop_mask, void* stream) {
  return ast_attn_fwd_p(q, k, v, o, probs, B, H, Lq, Lk, dh, ldq, ldk, ldo, causal, drop_mask, 0.f, 0, nullptr, stream);
}
extern "C" int ast_attn_bwd_p(const float* dout, const float* q, const float* k, const float* v, const float* probs, float* dq, float* dk,
                              float* dv, int B, int H, int Lq, int Lk, int dh, int ldq, int ldk, int ldo, const float* drop_mask,
                              float p, uint64_t seed, const int64_t* d_offset, void* stream) {
  if (!dout || !q || !k || !v || !probs || !dq || !dk || !dv || p < 0.f || p >= 1.f) AST_FAIL("ast_attn_bwd: bad args");
  if (Lq < 1 || Lk < 1 || Lq > MAXL || Lk > MAXL || dh < 1 || dh > 64) AST_FAIL("ast_attn_bwd: needs 1<=L<=%d and dh<=64", MAXL);
#define AST_ATTN_BWD(ML_)                                                                                                           \
  hipLaunchKernelGGL(attn_bwd_kernel<ML_>, dim3(B * H), dim3(64), 0, (hipStream_t)stream, dout, q, k, v, probs, dq, dk, dv, H, Lq, Lk, \
                     dh, ldq, ldk, ldo, drop_mask, p, seed, d_offset)
  if (Lk <= 4) AST_ATTN_BWD(4); else if (Lk <= 8) AST_ATTN_BWD(8); else AST_ATTN_BWD(16);
#undef AST_ATTN_BWD
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_attn_bwd(const float* dout, const float* q, const float* k, const float* v, const float* probs, float* dq, float* dk,
                            float* dv, int B, int H, int Lq, int Lk, int dh, int ldq, int ldk, int ldo, const float* drop_mask,
                            void* stream) {
  return ast_attn_bwd_p(dout, q, k, v, probs, dq, dk, dv, B, H, Lq, Lk, dh, ldq, ldk, ldo, drop_mask, 0.f, 0, nullptr, stream);
}

extern "C" int ast_add(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream) {
  if (!a || !b || !y || n < 0) AST_FAIL("ast_add: bad args");
  if (n == 0) return 0;
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((add_kernel<T>), dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, (const T*)a,
                                            (const T*)b, (T*)y, (size_t)n));
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, int dtype, void* stream) {
  if (!dy || !y || !dx || n < 0) AST_FAIL("ast_relu_bwd: bad args");
  if (n == 0) return 0;
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((relu_bwd_kernel<T>), dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream,
                                            (const T*)dy, (const T*)y, (T*)dx, (size_t)n));
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_mul(const void* a, const float* mask, void* y, int64_t n, int dtype, void* stream) {
  if (!a || !mask || !y || n < 0) AST_FAIL("ast_mul: bad args");
  if (n == 0) return 0;
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((mul_kernel<T>), dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, (const T*)a,
                                            mask, (T*)y, (size_t)n));
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_colsum_acc(const void* x, int64_t rows, int C, int Creal, float* out, int dtype, void* stream) {
  if (!x || !out || rows < 0 || C <= 0 || Creal > C) AST_FAIL("ast_colsum_acc: bad args");
  if (rows == 0) return 0;
  if (C <= 64 && !(C & 7) && !(C & (C - 1)) && !((uintptr_t)x & 15)) {      // 8, 16, 32, 64 channels: C/8 divides the grid stride
    const size_t n8 = (size_t)rows * C / 8;
    const int nb = (int)std::min<size_t>(256, (n8 + 1023) / 1024);                 // few workgroups: one atomic per column each
    AST_DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_small_kernel<T>), dim3(std::max(nb, 1)), dim3(256), 0, (hipStream_t)stream, (const T*)x, n8,
                                              C, Creal, out));
    AST_CHECK_LAUNCH();
    return 0;
  }
  const int bx = C >= 256 ? 256 : 64;
  const int nchunk = (int)std::min<size_t>(1024, ((size_t)rows + 63) / 64);
  const size_t rpb = ((size_t)rows + nchunk - 1) / nchunk;
  dim3 grid((Creal + bx - 1) / bx, (unsigned)(((size_t)rows + rpb - 1) / rpb));
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_acc_kernel<T>), grid, dim3(bx), 0, (hipStream_t)stream, (const T*)x, (size_t)rows, C,
                                            Creal, out, rpb));
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_dropout_mask(float* mask, int64_t n, float p, uint64_t seed, const int64_t* d_offset, void* stream) {
  if (!mask || n < 0 || p < 0.f || p >= 1.f) AST_FAIL("ast_dropout_mask: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, mask, (size_t)n, p, seed, d_offset);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_rowmix(const float* A, const float* X, float* Y, int R_out, int R_in, int D, void* stream) {
  if (!A || !X || !Y || R_out < 1 || R_in < 1 || R_in > 8192 || D < 1) AST_FAIL("ast_rowmix: bad args R_out=%d R_in=%d D=%d", R_out, R_in, D);
  hipLaunchKernelGGL(rowmix_kernel, dim3(R_out), dim3(256), R_in * sizeof(float), (hipStream_t)stream, A, X, Y, R_in, D);
  AST_CHECK_LAUNCH();
  return 0;
}
