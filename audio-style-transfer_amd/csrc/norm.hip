// Per-channel statistics and fused normalise/activation passes over NHWC
// activations (HBM-bound streaming kernels; 16-32 B per lane, f32 math).
//   BatchNorm2d (training): conv -> chan_stats -> norm_finalize -> affine_act
//   ResBlock tail: relu(BN(c2) + IN(ds)) is ONE affine_act pass over two inputs
//   backward: norm_bwd_sums -> norm_bwd_finalize -> norm_bwd_apply (one reduce
//   pass + one elementwise pass for both branches together)
#include "ast_common.h"
#include "../../include/ast_hip.h"

namespace {

// ---- reductions over pixels: sums[n][c][K] -------------------------------------
// MODE 0: {x, x^2}           (K=2)   inputs: a=x
// MODE 1: {dz, dz*x, dz*r}   (K=3)   inputs: a=dy, b=y (relu mask), c=x, d=r (may be null)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                           const T* __restrict__ c, const T* __restrict__ d,
                                                           float* __restrict__ sums, int HW, int C, int ppb, int relu,
                                                           const float* __restrict__ sc1, const float* __restrict__ sf1,
                                                           const float* __restrict__ sc2, const float* __restrict__ sf2) {
  // MODE 1 with sc1 != null: the ReLU mask is recomputed from the pre-activation fma(x, sc1, sf1) [+ fma(r, sc2[n], sf2[n])]
  // exactly as affine_act_kernel formed it, so the activation output `b` is not read at all (one third of the traffic)
  constexpr int K = MODE == 0 ? 2 : 3;
  __shared__ float red[256 * 8];
  const int U = C >> 3;
  const int PL = 256 / U;                 // pixel lanes per block (host guarantees U <= 256)
  const int tid = threadIdx.x;
  const int u = tid % U, pl = tid / U;
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * ppb, p1 = min(HW, p0 + ppb);
  float s[K][8];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) s[k][i] = 0.f;
  // per-channel / per-(n, channel) coefficients of the pre-activation: fixed for this thread (u, n), loaded once
  float a1[8], b1[8], a2[8], b2[8];
  if (MODE == 1 && relu && sc1 && pl < PL) {
    U8<float>::load(sc1 + u * 8, a1);
    U8<float>::load(sf1 + u * 8, b1);
    if (d) {
      U8<float>::load(sc2 + (size_t)n * C + u * 8, a2);
      U8<float>::load(sf2 + (size_t)n * C + u * 8, b2);
    }
  }
  if (pl < PL) {
    for (int p = p0 + pl; p < p1; p += PL) {
      const size_t off = ((size_t)n * HW + p) * C + u * 8;
      float va[8];
      U8<T>::load(a + off, va);
      if constexpr (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { s[0][i] += va[i]; s[1][i] += va[i] * va[i]; }
      } else {
        float vx[8], vr[8];
        U8<T>::load(c + off, vx);
        if (d) U8<T>::load(d + off, vr);
        if (relu) {
          float vy[8];
          if (sc1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) vy[i] = __builtin_fmaf(vx[i], a1[i], b1[i]);
            if (d) {
#pragma unroll
              for (int i = 0; i < 8; ++i) vy[i] += __builtin_fmaf(vr[i], a2[i], b2[i]);
            }
          } else {
            U8<T>::load(b + off, vy);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) va[i] = vy[i] > 0.f ? va[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s[0][i] += va[i]; s[1][i] += va[i] * vx[i]; }
        if (d) {
#pragma unroll
          for (int i = 0; i < 8; ++i) s[2][i] += va[i] * vr[i];
        }
      }
    }
  }
  // reduce over pixel lanes through LDS, one quantity at a time
#pragma unroll
  for (int k = 0; k < K; ++k) {
    __syncthreads();
    if (pl < PL) {
#pragma unroll
      for (int i = 0; i < 8; ++i) red[(pl * U + u) * 8 + i] = s[k][i];
    }
    __syncthreads();
    for (int ch = tid; ch < U * 8; ch += 256) {
      float t = 0.f;
      for (int q = 0; q < PL; ++q) t += red[q * U * 8 + ch];
      unsafeAtomicAdd(sums + ((size_t)n * C + ch) * K + k, t);
    }
  }
}

// one wave per output statistic (channel for batch norm, (n,c) for instance norm); lanes stride over images
__global__ __launch_bounds__(256) void norm_finalize_kernel(float* __restrict__ sums, int zero_sums, int64_t* nbt, int N, int HW, int C, int Creal, int instance,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float* running_mean, float* running_var, int eval_mode, float eps,
                                     float* __restrict__ mean, float* __restrict__ rstd,
                                     float* __restrict__ scale, float* __restrict__ shift, long count) {
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int total = instance ? N * C : C;
  if (idx >= total) return;
  const int c = idx % C;
  float m, var;
  // per-channel parameters are fetched up front so they overlap the statistics loads instead of trailing them
  const bool real_c = c < Creal;
  const float gmm = real_c ? gamma[c] : 0.f, bt = real_c ? beta[c] : 0.f;
  const bool upd = !instance && !eval_mode && running_mean && real_c;
  const float rm_old = (upd || (eval_mode && real_c)) ? running_mean[c] : 0.f;
  const float rv_old = (upd || (eval_mode && real_c)) ? running_var[c] : 1.f;
  if (nbt && idx == 0 && lane == 0) nbt[0] += 1;       // BatchNorm2d.num_batches_tracked
  if (instance) {
    const float cnt = (float)HW;
    m = sums[(size_t)idx * 2] / cnt;
    var = fmaxf(sums[(size_t)idx * 2 + 1] / cnt - m * m, 0.f);
    if (zero_sums && lane == 0) { sums[(size_t)idx * 2] = 0.f; sums[(size_t)idx * 2 + 1] = 0.f; }
  } else if (eval_mode) {
    m = rm_old;
    var = rv_old;
  } else {
    float s0 = 0.f, s1 = 0.f;
    for (int n = lane; n < N; n += 64) {
      float* q = sums + ((size_t)n * C + c) * 2;
      s0 += q[0]; s1 += q[1];
      if (zero_sums) { q[0] = 0.f; q[1] = 0.f; }          // leave the shared statistics scratch clean for the next caller
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    const double cnt = count > 0 ? (double)count : (double)N * HW;   // count > 0: the N rows are partial-sum slots, not images
    const double md = (double)s0 / cnt;
    const double vd = fmax((double)s1 / cnt - md * md, 0.0);
    m = (float)md; var = (float)vd;
    if (lane == 0 && upd) {
      running_mean[c] = 0.9f * rm_old + 0.1f * m;
      running_var[c] = 0.9f * rv_old + 0.1f * (float)(vd * cnt / fmax(cnt - 1.0, 1.0));
    }
  }
  if (lane != 0) return;
  const float r = rsqrtf(var + eps);
  mean[idx] = m; rstd[idx] = r;
  scale[idx] = gmm * r;
  shift[idx] = bt - m * gmm * r;
}

// grid = (blocks per image, N): a thread keeps one 8-channel unit u of one image n for its whole pixel loop (the
// stride is a multiple of U whenever U divides 256), so every per-channel coefficient is loaded ONCE; with the
// coefficient loads inside the loop these passes issued ~10 loads per 16 bytes of payload.
template <typename T>
__global__ __launch_bounds__(256) void affine_act_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const T* __restrict__ r,
                                                          const float* __restrict__ scale2, const float* __restrict__ shift2,
                                                          T* __restrict__ y, int HW, int C, size_t units, int s1_per_n, int relu) {
  const int U = C >> 3;
  const int n = blockIdx.y;
  const size_t per_img = (size_t)HW * U, base = (size_t)n * per_img;
  const size_t stride = (size_t)gridDim.x * 256;
  const bool hoist = (256 % U) == 0;
  float sc[8], sf[8], sc2[8], sf2[8];
  auto coef = [&](int u) __attribute__((always_inline)) {
    const size_t cidx = (s1_per_n ? (size_t)n * C : 0) + u * 8;
    U8<float>::load(scale + cidx, sc);
    U8<float>::load(shift + cidx, sf);
    if (r) {
      U8<float>::load(scale2 + (size_t)n * C + u * 8, sc2);
      U8<float>::load(shift2 + (size_t)n * C + u * 8, sf2);
    }
  };
  const size_t j0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (hoist) coef((int)(j0 % U));
  for (size_t j = j0; j < per_img; j += stride) {
    if (!hoist) coef((int)(j % U));
    const size_t i = base + j;
    float v[8];
    U8<T>::load(x + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], sc[k], sf[k]);
    if (r) {
      float w[8];
      U8<T>::load(r + i * 8, w);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += __builtin_fmaf(w[k], sc2[k], sf2[k]);
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    U8<T>::store(y + i * 8, v);
  }
}

// k1[c][3] for the batch branch, k2[n][c][3] for the instance branch; one wave per channel, lanes over images
__global__ __launch_bounds__(256) void norm_bwd_finalize_kernel(float* __restrict__ sums3, int zero_sums, int N, int HW, int C, int Creal,
                                         const float* gamma1, const float* mean1, const float* rstd1,
                                         float* dgamma1, float* dbeta1, float* k1,
                                         const float* gamma2, const float* mean2, const float* rstd2,
                                         float* dgamma2, float* dbeta2, float* k2, long count) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  const bool real = c < Creal;
  float s0 = 0.f, s1 = 0.f, dg = 0.f;
  for (int n = lane; n < N; n += 64) {
    const size_t i = (size_t)n * C + c;
    const float T0 = sums3[i * 3], T1 = sums3[i * 3 + 1];
    const float T2raw = sums3[i * 3 + 2];
    if (zero_sums) { sums3[i * 3] = 0.f; sums3[i * 3 + 1] = 0.f; sums3[i * 3 + 2] = 0.f; }
    s0 += T0; s1 += T1;
    if (k2) {
      if (real) {
        const double T2 = T2raw;
        const double P = (double)HW, g = gamma2[c], m = mean2[i], r = rstd2[i];
        const double Q = r * (T2 - m * (double)T0);
        dg += (float)Q;
        k2[i * 3 + 0] = (float)(g * r);
        k2[i * 3 + 1] = (float)(-g * r * r * Q / P);
        k2[i * 3 + 2] = (float)(-g * r * (double)T0 / P + g * r * r * m * Q / P);
      } else { k2[i * 3] = k2[i * 3 + 1] = k2[i * 3 + 2] = 0.f; }
    }
  }
  const double S0 = wave_sum(s0), S1 = wave_sum(s1);
  dg = wave_sum(dg);
  if (lane != 0) return;
  if (k1) {
    if (real) {
      const double P = count > 0 ? (double)count : (double)N * HW, g = gamma1[c], m = mean1[c], r = rstd1[c];   // count > 0: rows are slots
      const double Q = r * (S1 - m * S0);
      if (dgamma1) dgamma1[c] += (float)Q;
      if (dbeta1) dbeta1[c] += (float)S0;
      k1[c * 3 + 0] = (float)(g * r);
      k1[c * 3 + 1] = (float)(-g * r * r * Q / P);
      k1[c * 3 + 2] = (float)(-g * r * S0 / P + g * r * r * m * Q / P);
    } else { k1[c * 3] = k1[c * 3 + 1] = k1[c * 3 + 2] = 0.f; }
  }
  if (k2 && real) { if (dgamma2) dgamma2[c] += dg; if (dbeta2) dbeta2[c] += (float)S0; }
}

template <typename T>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                              const T* __restrict__ x, const T* __restrict__ r,
                                                              const float* __restrict__ k1, const float* __restrict__ k2,
                                                              T* __restrict__ dx, T* __restrict__ dr, int HW, int C,
                                                              size_t units, int relu, const float* __restrict__ sc1,
                                                              const float* __restrict__ sf1, const float* __restrict__ sc2,
                                                              const float* __restrict__ sf2) {
  const int U = C >> 3;
  const int n = blockIdx.y;                       // grid = (blocks per image, N): see affine_act_kernel
  const size_t per_img = (size_t)HW * U, base = (size_t)n * per_img;
  const size_t stride = (size_t)gridDim.x * 256;
  const bool hoist = (256 % U) == 0;
  const bool remask = relu && sc1;               // mask from the pre-activation (see chan_reduce_kernel): y is not read
  float a1[8], b1[8], a2[8], b2[8], ka[8][3], kb[8][3];
  auto coef = [&](int u) __attribute__((always_inline)) {
    if (remask) {
      U8<float>::load(sc1 + u * 8, a1);
      U8<float>::load(sf1 + u * 8, b1);
      if (r) {
        U8<float>::load(sc2 + (size_t)n * C + u * 8, a2);
        U8<float>::load(sf2 + (size_t)n * C + u * 8, b2);
      }
    }
    if (dx && k1) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int q = 0; q < 3; ++q) ka[k][q] = k1[(u * 8 + k) * 3 + q];
    }
    if (dr) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int q = 0; q < 3; ++q) kb[k][q] = k2[((size_t)n * C + u * 8 + k) * 3 + q];
    }
  };
  const size_t j0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (hoist) coef((int)(j0 % U));
  for (size_t j = j0; j < per_img; j += stride) {
    if (!hoist) coef((int)(j % U));
    const size_t i = base + j;
    float dz[8], v[8], o[8], vr[8];
    U8<T>::load(dy + i * 8, dz);
    if (remask || (dx && k1)) U8<T>::load(x + i * 8, v);
    if (dr || (remask && r)) U8<T>::load(r + i * 8, vr);
    if (relu) {
      float pre[8];
      if (remask) {
#pragma unroll
        for (int k = 0; k < 8; ++k) pre[k] = __builtin_fmaf(v[k], a1[k], b1[k]);
        if (r) {
#pragma unroll
          for (int k = 0; k < 8; ++k) pre[k] += __builtin_fmaf(vr[k], a2[k], b2[k]);
        }
      } else {
        U8<T>::load(y + i * 8, pre);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) dz[k] = pre[k] > 0.f ? dz[k] : 0.f;
    }
    if (dx) {
      if (k1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = ka[k][0] * dz[k] + ka[k][1] * v[k] + ka[k][2];
        U8<T>::store(dx + i * 8, o);
      } else {
        U8<T>::store(dx + i * 8, dz);   // plain ReLU backward
      }
    }
    if (dr) {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = kb[k][0] * dz[k] + kb[k][1] * vr[k] + kb[k][2];
      U8<T>::store(dr + i * 8, o);
    }
  }
}

// ---- statistics finalize folded into the apply passes ---------------------------------------------------------------------
// The per-channel coefficients of a BatchNorm / InstanceNorm layer used to come from their own launches (norm_finalize,
// norm_bwd_finalize): 38 + 40 single-wave kernels of ~5 us per step that do nothing but sit between a GEMM and the pass that
// needs their 4 x C floats -- pure dependency latency on the step's longest chain (0.7 ms of the step is node
// dispatch).  Here every workgroup of the APPLY pass reduces the statistics table itself (C x slots x 2-3 floats from L2, at
// most 32 KB: the host keeps C x slots <= 4096) into LDS before its pixel loop; workgroup (0, 0) also writes the layer's
// outputs (mean / rstd / scale / shift for the backward pass, running statistics, gamma / beta gradients).  Nobody may clear
// the table (other workgroups are still reading it): the tables live in a per-step arena that ONE memset clears (ops.stat_table).
//   coef (LDS): [0,C) scale1, [C,2C) shift1, [2C,3C) scale2 of image n, [3C,4C) shift2 of image n, then reduction scratch
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const T* __restrict__ x, const T* __restrict__ r, T* __restrict__ y,
                                                            const float* __restrict__ tab1, const int rows1, const float count1,
                                                            const float* __restrict__ tab2, const float* __restrict__ gamma1,
                                                            const float* __restrict__ beta1, float* __restrict__ running_mean,
                                                            float* __restrict__ running_var, int64_t* __restrict__ nbt, const float eps1,
                                                            const float* __restrict__ gamma2, const float* __restrict__ beta2, const float eps2,
                                                            float* __restrict__ out1, float* __restrict__ out2, const int N, const int HW,
                                                            const int C, const int Creal, const int relu) {
  extern __shared__ float coef[];
  const int tid = threadIdx.x, n = blockIdx.y;
  const bool lead = blockIdx.x == 0 && n == 0;
  const int nparts = max(1, min(rows1, 256 / min(C, 256)));          // threads per channel in the slot reduction
  float* part = coef + 4 * C;
  for (int idx = tid; idx < C * nparts; idx += 256) {
    const int c = idx % C, pt = idx / C;
    float s0 = 0.f, s1 = 0.f;
    for (int q = pt; q < rows1; q += nparts) {
      const f32x2 v = *reinterpret_cast<const f32x2*>(tab1 + ((size_t)q * C + c) * 2);
      s0 += v[0]; s1 += v[1];
    }
    part[(pt * C + c) * 2] = s0; part[(pt * C + c) * 2 + 1] = s1;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float s0 = 0.f, s1 = 0.f;
    for (int pt = 0; pt < nparts; ++pt) { s0 += part[(pt * C + c) * 2]; s1 += part[(pt * C + c) * 2 + 1]; }
    const double cnt = (double)count1;
    const double md = (double)s0 / cnt, vd = fmax((double)s1 / cnt - md * md, 0.0);
    const float m = (float)md, rs = rsqrtf((float)vd + eps1);
    const bool real = c < Creal;
    const float g = real ? gamma1[c] : 0.f, b = real ? beta1[c] : 0.f;
    const float sc = g * rs, sf = b - m * g * rs;
    coef[c] = sc; coef[C + c] = sf;
    if (lead) {
      out1[c] = m; out1[C + c] = rs; out1[2 * C + c] = sc; out1[3 * C + c] = sf;
      if (running_mean && real) {                                      // momentum 0.1, unbiased variance (nn.BatchNorm2d)
        running_mean[c] = 0.9f * running_mean[c] + 0.1f * m;
        running_var[c] = 0.9f * running_var[c] + 0.1f * (float)(vd * cnt / fmax(cnt - 1.0, 1.0));
      }
    }
    if (r) {                                                           // InstanceNorm2d of the second input, image n
      const float cntf = (float)HW;
      const float m2 = tab2[((size_t)n * C + c) * 2] / cntf;
      const float v2 = fmaxf(tab2[((size_t)n * C + c) * 2 + 1] / cntf - m2 * m2, 0.f);
      const float r2 = rsqrtf(v2 + eps2);
      const float g2 = real ? gamma2[c] : 0.f, b2 = real ? beta2[c] : 0.f;
      coef[2 * C + c] = g2 * r2; coef[3 * C + c] = b2 - m2 * g2 * r2;
      if (blockIdx.x == 0) {
        const size_t o = (size_t)n * C + c, NC = (size_t)N * C;
        out2[o] = m2; out2[NC + o] = r2; out2[2 * NC + o] = g2 * r2; out2[3 * NC + o] = b2 - m2 * g2 * r2;
      }
    }
  }
  if (lead && tid == 0 && nbt) nbt[0] += 1;                            // BatchNorm2d.num_batches_tracked
  __syncthreads();

  const int U = C >> 3;
  const size_t per_img = (size_t)HW * U, base = (size_t)n * per_img;
  const size_t stride = (size_t)gridDim.x * 256;
  const bool hoist = (256 % U) == 0;
  float sc[8], sf[8], sc2[8], sf2[8];
  auto load_coef = [&](int u) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = coef[u * 8 + k]; sf[k] = coef[C + u * 8 + k]; }
    if (r) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { sc2[k] = coef[2 * C + u * 8 + k]; sf2[k] = coef[3 * C + u * 8 + k]; }
    }
  };
  const size_t j0 = (size_t)blockIdx.x * 256 + tid;
  if (hoist) load_coef((int)(j0 % U));
  for (size_t j = j0; j < per_img; j += stride) {
    if (!hoist) load_coef((int)(j % U));
    const size_t i = base + j;
    float v[8];
    U8<T>::load(x + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], sc[k], sf[k]);
    if (r) {
      float w[8];
      U8<T>::load(r + i * 8, w);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += __builtin_fmaf(w[k], sc2[k], sf2[k]);
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    U8<T>::store(y + i * 8, v);
  }
}

// Backward twin: k1[c][3] (batch branch, from all rows of tab3) and k2[c][3] (instance branch of image n, from row n) are
// formed in LDS by every workgroup; workgroup (0, 0) adds the gamma / beta gradients.  tab3[row][c] = {sum dz, sum dz x, sum dz r};
// rows = statistics slots (filled by the data-gradient GEMM's epilogue, count = pixels) or images (a reduction pass).
//   kc (LDS): [0,3C) k1, [3C,6C) k2, then reduction scratch
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ r,
                                                            T* __restrict__ dx, T* __restrict__ dr, const float* __restrict__ tab3,
                                                            const int rows, const float count, const float* __restrict__ gamma1,
                                                            const float* __restrict__ mean1, const float* __restrict__ rstd1,
                                                            float* __restrict__ dgamma1, float* __restrict__ dbeta1,
                                                            const float* __restrict__ gamma2, const float* __restrict__ mean2,
                                                            const float* __restrict__ rstd2, float* __restrict__ dgamma2,
                                                            float* __restrict__ dbeta2, const float* __restrict__ sc1, const float* __restrict__ sf1,
                                                            const float* __restrict__ sc2, const float* __restrict__ sf2, const int N, const int HW,
                                                            const int C, const int Creal, const int relu) {
  extern __shared__ float kc[];
  const int tid = threadIdx.x, n = blockIdx.y;
  const bool lead = blockIdx.x == 0 && n == 0;
  const int nparts = max(1, min(rows, 256 / min(C, 256)));
  float* part = kc + 6 * C;
  for (int idx = tid; idx < C * nparts; idx += 256) {
    const int c = idx % C, pt = idx / C;
    float s0 = 0.f, s1 = 0.f;
    for (int q = pt; q < rows; q += nparts) {
      const float* t3 = tab3 + ((size_t)q * C + c) * 3;
      s0 += t3[0]; s1 += t3[1];
    }
    part[(pt * C + c) * 2] = s0; part[(pt * C + c) * 2 + 1] = s1;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float s0 = 0.f, s1 = 0.f;
    for (int pt = 0; pt < nparts; ++pt) { s0 += part[(pt * C + c) * 2]; s1 += part[(pt * C + c) * 2 + 1]; }
    const bool real = c < Creal;
    if (real) {
      const double P = (double)count, g = gamma1[c], m = mean1[c], rs = rstd1[c], S0 = s0, S1 = s1;
      const double Q = rs * (S1 - m * S0);
      kc[c * 3 + 0] = (float)(g * rs);
      kc[c * 3 + 1] = (float)(-g * rs * rs * Q / P);
      kc[c * 3 + 2] = (float)(-g * rs * S0 / P + g * rs * rs * m * Q / P);
      if (lead) { if (dgamma1) dgamma1[c] += (float)Q; if (dbeta1) dbeta1[c] += (float)S0; }
    } else { kc[c * 3] = kc[c * 3 + 1] = kc[c * 3 + 2] = 0.f; }
    if (dr) {                                                          // instance branch: image n (tab3 rows are images here)
      if (real) {
        const size_t i = (size_t)n * C + c;
        const double P = (double)HW, g = gamma2[c], m = mean2[i], rs = rstd2[i], T0 = tab3[i * 3], T2 = tab3[i * 3 + 2];
        const double Q = rs * (T2 - m * T0);
        kc[3 * C + c * 3 + 0] = (float)(g * rs);
        kc[3 * C + c * 3 + 1] = (float)(-g * rs * rs * Q / P);
        kc[3 * C + c * 3 + 2] = (float)(-g * rs * T0 / P + g * rs * rs * m * Q / P);
        if (lead) {                                                    // gamma2 / beta2 gradients: sums over the images, in a fixed order
          float dg = 0.f, db = 0.f;
          for (int q = 0; q < N; ++q) {
            const size_t iq = (size_t)q * C + c;
            const float t0 = tab3[iq * 3], t2 = tab3[iq * 3 + 2];
            dg += (float)((double)rstd2[iq] * ((double)t2 - (double)mean2[iq] * (double)t0));
            db += t0;
          }
          if (dgamma2) dgamma2[c] += dg;
          if (dbeta2) dbeta2[c] += db;
        }
      } else { kc[3 * C + c * 3] = kc[3 * C + c * 3 + 1] = kc[3 * C + c * 3 + 2] = 0.f; }
    }
  }
  __syncthreads();

  const int U = C >> 3;
  const size_t per_img = (size_t)HW * U, base = (size_t)n * per_img;
  const size_t stride = (size_t)gridDim.x * 256;
  const bool hoist = (256 % U) == 0;
  const bool remask = relu != 0;                   // mask recomputed from the pre-activation fma(x, sc1, sf1) [+ fma(r, sc2[n], sf2[n])]
  // the 8 x 3 coefficients of the thread's channel unit as six 16-byte LDS reads per branch, indexed by compile-time constants only
  // (as [8][3] arrays filled element by element they went to scratch memory: 208 B per lane, and the pixel loop ran 3x slower)
  float a1[8], b1[8], a2[8], b2[8];
  f32x4 kaq[6], kbq[6];
  auto load_coef = [&](int u) __attribute__((always_inline)) {
    if (remask) {
      U8<float>::load(sc1 + u * 8, a1);
      U8<float>::load(sf1 + u * 8, b1);
      if (r) {
        U8<float>::load(sc2 + (size_t)n * C + u * 8, a2);
        U8<float>::load(sf2 + (size_t)n * C + u * 8, b2);
      }
    }
    const f32x4* ka4 = reinterpret_cast<const f32x4*>(kc + u * 24);
    const f32x4* kb4 = reinterpret_cast<const f32x4*>(kc + 3 * C + u * 24);
#pragma unroll
    for (int q = 0; q < 6; ++q) { kaq[q] = ka4[q]; kbq[q] = dr ? kb4[q] : f32x4{0.f, 0.f, 0.f, 0.f}; }
  };
#define KA(k, q) kaq[((k) * 3 + (q)) >> 2][((k) * 3 + (q)) & 3]
#define KB(k, q) kbq[((k) * 3 + (q)) >> 2][((k) * 3 + (q)) & 3]
  const size_t j0 = (size_t)blockIdx.x * 256 + tid;
  if (hoist) load_coef((int)(j0 % U));
  for (size_t j = j0; j < per_img; j += stride) {
    if (!hoist) load_coef((int)(j % U));
    const size_t i = base + j;
    float dz[8], v[8], o[8], vr[8];
    U8<T>::load(dy + i * 8, dz);
    U8<T>::load(x + i * 8, v);
    if (r) U8<T>::load(r + i * 8, vr);
    if (remask) {
      float pre[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) pre[k] = __builtin_fmaf(v[k], a1[k], b1[k]);
      if (r) {
#pragma unroll
        for (int k = 0; k < 8; ++k) pre[k] += __builtin_fmaf(vr[k], a2[k], b2[k]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) dz[k] = pre[k] > 0.f ? dz[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = KA(k, 0) * dz[k] + KA(k, 1) * v[k] + KA(k, 2);
    U8<T>::store(dx + i * 8, o);
    if (dr) {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = KB(k, 0) * dz[k] + KB(k, 1) * vr[k] + KB(k, 2);
      U8<T>::store(dr + i * 8, o);
    }
  }
#undef KA
#undef KB
}

// ---- LayerNorm over the last dim (rows x D), one wave per row, f32 --------------
__global__ void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ mean,
                                     float* __restrict__ rstd, int rows, int D, float eps) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xr[i];
  const float m = wave_sum(s) / D;
  float q = 0.f;
  for (int i = lane; i < D; i += 64) { const float d = xr[i] - m; q += d * d; }
  const float r = rsqrtf(wave_sum(q) / D + eps);
  for (int i = lane; i < D; i += 64) y[(size_t)row * D + i] = (xr[i] - m) * r * gamma[i] + beta[i];
  if (lane == 0) { mean[row] = m; rstd[row] = r; }
}

__global__ void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, float* __restrict__ dx, float* dgamma, float* dbeta,
                                     int rows, int D) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float m = mean[row], r = rstd[row];
  const float* xr = x + (size_t)row * D;
  const float* dr = dy + (size_t)row * D;
  float a = 0.f, b = 0.f;
  for (int i = lane; i < D; i += 64) {
    const float xh = (xr[i] - m) * r, g = dr[i] * gamma[i];
    a += g; b += g * xh;
  }
  a = wave_sum(a) / D; b = wave_sum(b) / D;
  for (int i = lane; i < D; i += 64) {
    const float xh = (xr[i] - m) * r, g = dr[i] * gamma[i];
    dx[(size_t)row * D + i] = r * (g - a - xh * b);
    if (dgamma) unsafeAtomicAdd(dgamma + i, dr[i] * xh);
    if (dbeta) unsafeAtomicAdd(dbeta + i, dr[i]);
  }
}


// ---- fused residual + dropout + LayerNorm (token rows, f32), one wave per row -------------------------------
//   s = x + dropout(sub)        (x may be null: s = dropout(sub);  p == 0: no mask is written)
//   y = LayerNorm(s)            (gamma null: no normalisation, s is the only output)
// s is always written: it is the residual stream the pre-norm decoder carries on and the LN input its backward needs.
__global__ void add_drop_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ sub, float* __restrict__ mask,
                                       float* __restrict__ s_out, const float* __restrict__ gamma, const float* __restrict__ beta,
                                       float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int rows, int D,
                                       float eps, float p, uint64_t seed, const int64_t* __restrict__ d_offset) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const size_t o = (size_t)row * D;
  const uint64_t base = p > 0.f ? mix64(seed ^ mix64((uint64_t)(d_offset ? *d_offset : 0))) : 0;
  const float keep = 1.f / (1.f - p);
  float acc = 0.f;
  for (int i = lane; i < D; i += 64) {
    float v = sub[o + i];
    if (p > 0.f) { const float m = dropout_keep(base, o + i, p, keep); mask[o + i] = m; v *= m; }
    if (x) v += x[o + i];
    s_out[o + i] = v;
    acc += v;
  }
  if (!gamma) return;
  const float m = wave_sum(acc) / D;
  float q = 0.f;
  for (int i = lane; i < D; i += 64) { const float d = s_out[o + i] - m; q += d * d; }      // own stores: visible to this lane
  const float r = rsqrtf(wave_sum(q) / D + eps);
  for (int i = lane; i < D; i += 64) y[o + i] = (s_out[o + i] - m) * r * gamma[i] + beta[i];
  if (lane == 0) { mean[row] = m; rstd[row] = r; }
}

//   ds = ds_ext + LayerNorm_bwd(dy; s)     (either term may be absent)
//   dx = ds,  dsub = ds * mask             (dx may alias nothing; both are plain stores)
__global__ void add_drop_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ds_ext, const float* __restrict__ s,
                                       const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                                       const float* __restrict__ mask, float* __restrict__ dx, float* __restrict__ dsub,
                                       float* dgamma, float* dbeta, int rows, int D) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const size_t o = (size_t)row * D;
  float a = 0.f, b = 0.f, m = 0.f, r = 0.f;
  if (dy) {
    m = mean[row]; r = rstd[row];
    for (int i = lane; i < D; i += 64) {
      const float xh = (s[o + i] - m) * r, g = dy[o + i] * gamma[i];
      a += g; b += g * xh;
    }
    a = wave_sum(a) / D; b = wave_sum(b) / D;
  }
  for (int i = lane; i < D; i += 64) {
    float d = ds_ext ? ds_ext[o + i] : 0.f;
    if (dy) {
      const float xh = (s[o + i] - m) * r, g = dy[o + i] * gamma[i];
      d += r * (g - a - xh * b);
      if (dgamma) unsafeAtomicAdd(dgamma + i, dy[o + i] * xh);
      if (dbeta) unsafeAtomicAdd(dbeta + i, dy[o + i]);
    }
    if (dx) dx[o + i] = d;
    dsub[o + i] = mask ? d * mask[o + i] : d;
  }
}

// D == 256 forms: a lane owns 4 consecutive elements, every operand is one 16-byte load issued up front (one memory
// round trip instead of three or four dependent ones), and the gamma/beta gradients of the workgroup's 4 rows are
// summed in LDS before ONE pair of atomics per element (an atomic costs its issue slot).
__global__ __launch_bounds__(256) void add_drop_ln_fwd256_kernel(const float* __restrict__ x, const float* __restrict__ sub,
                                                                 float* __restrict__ mask, float* __restrict__ s_out,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                                 int rows, float eps, float p, uint64_t seed,
                                                                 const int64_t* __restrict__ d_offset) {
  constexpr int D = 256;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const size_t o = (size_t)row * D + lane * 4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 sv = *reinterpret_cast<const f32x4*>(sub + o);
  const f32x4 xv = x ? *reinterpret_cast<const f32x4*>(x + o) : z;
  const f32x4 g4 = gamma ? *reinterpret_cast<const f32x4*>(gamma + lane * 4) : z;
  const f32x4 b4 = gamma ? *reinterpret_cast<const f32x4*>(beta + lane * 4) : z;
  const uint64_t base = p > 0.f ? mix64(seed ^ mix64((uint64_t)(d_offset ? *d_offset : 0))) : 0;
  __builtin_amdgcn_sched_barrier(0);
  const float keep = 1.f / (1.f - p);
  f32x4 sv2 = sv;
  if (p > 0.f) {
    f32x4 m4;
#pragma unroll
    for (int e = 0; e < 4; ++e) { m4[e] = dropout_keep(base, o + e, p, keep); sv2[e] *= m4[e]; }
    *reinterpret_cast<f32x4*>(mask + o) = m4;
  }
  f32x4 s4;
#pragma unroll
  for (int e = 0; e < 4; ++e) s4[e] = x ? sv2[e] + xv[e] : sv2[e];
  *reinterpret_cast<f32x4*>(s_out + o) = s4;
  if (!gamma) return;
  const float m = wave_sum(s4[0] + s4[1] + s4[2] + s4[3]) / D;
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { const float d = s4[e] - m; q += d * d; }
  const float r = rsqrtf(wave_sum(q) / D + eps);
  f32x4 y4;
#pragma unroll
  for (int e = 0; e < 4; ++e) y4[e] = (s4[e] - m) * r * g4[e] + b4[e];
  *reinterpret_cast<f32x4*>(y + o) = y4;
  if (lane == 0) { mean[row] = m; rstd[row] = r; }
}

__global__ __launch_bounds__(256) void add_drop_ln_bwd256_kernel(const float* __restrict__ dy, const float* __restrict__ ds_ext,
                                                                 const float* __restrict__ s, const float* __restrict__ gamma,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ mask, float* __restrict__ dx,
                                                                 float* __restrict__ dsub, float* dgamma, float* dbeta, int rows) {
  constexpr int D = 256;
  __shared__ float red[2][4][D];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  const bool valid = row < rows;
  const size_t o = (size_t)(valid ? row : 0) * D + lane * 4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 dy4 = dy ? *reinterpret_cast<const f32x4*>(dy + o) : z;
  const f32x4 s4 = dy ? *reinterpret_cast<const f32x4*>(s + o) : z;
  const f32x4 g4 = dy ? *reinterpret_cast<const f32x4*>(gamma + lane * 4) : z;
  const f32x4 e4 = ds_ext ? *reinterpret_cast<const f32x4*>(ds_ext + o) : z;
  const f32x4 k4 = mask ? *reinterpret_cast<const f32x4*>(mask + o) : f32x4{1.f, 1.f, 1.f, 1.f};
  const float m = dy ? mean[valid ? row : 0] : 0.f, r = dy ? rstd[valid ? row : 0] : 0.f;
  __builtin_amdgcn_sched_barrier(0);
  f32x4 d4 = e4, xh4 = z;
  if (dy) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { xh4[e] = (s4[e] - m) * r; const float g = dy4[e] * g4[e]; a += g; b += g * xh4[e]; }
    a = wave_sum(a) / D; b = wave_sum(b) / D;
#pragma unroll
    for (int e = 0; e < 4; ++e) d4[e] += r * (dy4[e] * g4[e] - a - xh4[e] * b);
  }
  if (valid) {
    if (dx) *reinterpret_cast<f32x4*>(dx + o) = d4;
    f32x4 ds4;
#pragma unroll
    for (int e = 0; e < 4; ++e) ds4[e] = d4[e] * k4[e];
    *reinterpret_cast<f32x4*>(dsub + o) = ds4;
  }
  if (dy && (dgamma || dbeta)) {              // uniform per launch: every wave reaches the barrier
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][wave][lane * 4 + e] = valid ? dy4[e] * xh4[e] : 0.f;
      red[1][wave][lane * 4 + e] = valid ? dy4[e] : 0.f;
    }
    __syncthreads();
    const int i = threadIdx.x;
    if (dgamma) unsafeAtomicAdd(dgamma + i, red[0][0][i] + red[0][1][i] + red[0][2][i] + red[0][3][i]);
    if (dbeta) unsafeAtomicAdd(dbeta + i, red[1][0][i] + red[1][1][i] + red[1][2][i] + red[1][3][i]);
  }
}

// workgroups per statistics pass: every workgroup ends with C*K same-address f32 atomics, so FEWER is faster up to the
// point where the chip runs dry (A/B on the whole step: 8192 -> 8.9 ms, 4096 -> 8.35, 2048 -> 7.9, 1024 -> 7.5, 256 -> 7.55)
// (round 2, per-kernel time in the replayed step, tools/knob_ab.sh: 1024 -> 16.4-16.6 us, 512 -> 14.9, 384 -> 14.8, 256 -> 17.1)
int reduce_blocks() { static const int v = getenv("AST_REDUCE_BLOCKS") ? atoi(getenv("AST_REDUCE_BLOCKS")) : 512; return v; }

int elem_blocks() { static const int v = getenv("AST_ELEM_BLOCKS") ? atoi(getenv("AST_ELEM_BLOCKS")) : 2048; return v; }

int grid_for(size_t n, int block = 256) { return (int)std::min<size_t>((n + block - 1) / block, 256 * 16); }

}  // namespace

extern "C" int ast_chan_stats(const void* x, float* sums, int N, int HW, int C, int dtype, int assume_zeroed, void* stream) {
  if (!x || !sums || N <= 0 || HW <= 0 || C <= 0 || (C & 7) || C > 2048) AST_FAIL("ast_chan_stats: bad args N=%d HW=%d C=%d", N, HW, C);
  hipStream_t s = (hipStream_t)stream;
  if (!assume_zeroed) AST_HIP(hipMemsetAsync(sums, 0, sizeof(float) * (size_t)N * C * 2, s));
  const int PL = 256 / (C >> 3);
  const int nblk = max(1, min((HW + PL - 1) / PL, max(1, reduce_blocks() / N)));
  const int ppb = (HW + nblk - 1) / nblk;
  dim3 grid((HW + ppb - 1) / ppb, N);
  const float* nf = nullptr;
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((chan_reduce_kernel<T, 0>), grid, dim3(256), 0, s, (const T*)x, (const T*)nullptr,
                                            (const T*)nullptr, (const T*)nullptr, sums, HW, C, ppb, 0, nf, nf, nf, nf));
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_norm_finalize(float* sums, int zero_sums, int64_t* num_batches_tracked, int N, int HW, int C, int Creal, int instance,
                                 const float* gamma, const float* beta, float* running_mean, float* running_var, int eval_mode, float eps,
                                 float* mean, float* rstd, float* scale, float* shift, long count, void* stream) {
  if (!gamma || !beta || !mean || !rstd || !scale || !shift) AST_FAIL("ast_norm_finalize: null pointer");
  if (!eval_mode && !sums) AST_FAIL("ast_norm_finalize: sums required in training mode");
  if (eval_mode && (instance || !running_mean || !running_var)) AST_FAIL("ast_norm_finalize: eval mode needs running stats");
  const int total = instance ? N * C : C;
  hipLaunchKernelGGL(norm_finalize_kernel, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, sums, zero_sums,
                     num_batches_tracked, N, HW, C, Creal, instance, gamma, beta, running_mean, running_var, eval_mode, eps, mean, rstd,
                     scale, shift, count);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_affine_act(const void* x, const float* scale, const float* shift, const void* r, const float* scale2,
                              const float* shift2, void* y, int N, int HW, int C, int relu, int dtype, void* stream) {
  if (!x || !scale || !shift || !y || (C & 7) || (r && (!scale2 || !shift2))) AST_FAIL("ast_affine_act: bad args");
  const size_t units = (size_t)N * HW * (C >> 3);
  // scale given per channel ([C]); the instance form ([N][C]) is requested with relu bit 2
  const int s1_per_n = (relu >> 1) & 1;
  const dim3 agrid(std::max(1, std::min<int>((int)(((size_t)HW * (C >> 3) + 255) / 256), std::max(1, elem_blocks() / N))), N);
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((affine_act_kernel<T>), agrid, dim3(256), 0, (hipStream_t)stream,
                                            (const T*)x, scale, shift, (const T*)r, scale2, shift2, (T*)y, HW, C, units,
                                            s1_per_n, relu & 1));
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_norm_bwd_sums_pre(const void* dy, const void* y, const void* x, const void* r, float* sums3, int N, int HW,
                                     int C, int relu, int dtype, int assume_zeroed, const float* scale1, const float* shift1,
                                     const float* scale2, const float* shift2, void* stream) {
  if (!dy || !x || !sums3 || (relu && !y && !scale1) || (C & 7) || C > 2048) AST_FAIL("ast_norm_bwd_sums: bad args");
  if (scale1 && (!shift1 || (r && (!scale2 || !shift2)))) AST_FAIL("ast_norm_bwd_sums: incomplete pre-activation coefficients");
  hipStream_t s = (hipStream_t)stream;
  if (!assume_zeroed) AST_HIP(hipMemsetAsync(sums3, 0, sizeof(float) * (size_t)N * C * 3, s));
  const int PL = 256 / (C >> 3);
  const int nblk = max(1, min((HW + PL - 1) / PL, max(1, reduce_blocks() / N)));
  const int ppb = (HW + nblk - 1) / nblk;
  dim3 grid((HW + ppb - 1) / ppb, N);
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((chan_reduce_kernel<T, 1>), grid, dim3(256), 0, s, (const T*)dy, (const T*)y,
                                            (const T*)x, (const T*)r, sums3, HW, C, ppb, relu, scale1, shift1, scale2, shift2));
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_norm_bwd_sums(const void* dy, const void* y, const void* x, const void* r, float* sums3, int N, int HW,
                                 int C, int relu, int dtype, int assume_zeroed, void* stream) {
  return ast_norm_bwd_sums_pre(dy, y, x, r, sums3, N, HW, C, relu, dtype, assume_zeroed, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int ast_norm_bwd_finalize_n(float* sums3, int zero_sums, int N, int HW, int C, int Creal, const float* gamma1,
                                     const float* mean1, const float* rstd1, float* dgamma1, float* dbeta1, float* k1,
                                     const float* gamma2, const float* mean2, const float* rstd2, float* dgamma2,
                                     float* dbeta2, float* k2, long count, void* stream) {
  if (!sums3 || (k1 && (!gamma1 || !mean1 || !rstd1)) || (k2 && (!gamma2 || !mean2 || !rstd2))) AST_FAIL("ast_norm_bwd_finalize: bad args");
  hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, sums3, zero_sums, N, HW, C, Creal,
                     gamma1, mean1, rstd1, dgamma1, dbeta1, k1, gamma2, mean2, rstd2, dgamma2, dbeta2, k2, count);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_norm_bwd_finalize(float* sums3, int zero_sums, int N, int HW, int C, int Creal, const float* gamma1,
                                     const float* mean1, const float* rstd1, float* dgamma1, float* dbeta1, float* k1,
                                     const float* gamma2, const float* mean2, const float* rstd2, float* dgamma2,
                                     float* dbeta2, float* k2, void* stream) {
  return ast_norm_bwd_finalize_n(sums3, zero_sums, N, HW, C, Creal, gamma1, mean1, rstd1, dgamma1, dbeta1, k1, gamma2, mean2, rstd2,
                                 dgamma2, dbeta2, k2, 0, stream);
}

extern "C" int ast_norm_bwd_apply_pre(const void* dy, const void* y, const void* x, const void* r, const float* k1,
                                      const float* k2, void* dx, void* dr, int N, int HW, int C, int relu, int dtype,
                                      const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                                      void* stream) {
  if (!dy || (relu && !y && !scale1) || (dx && k1 && !x) || (dr && (!r || !k2)) || (C & 7)) AST_FAIL("ast_norm_bwd_apply: bad args");
  if (scale1 && (!x || !shift1 || (r && (!scale2 || !shift2)))) AST_FAIL("ast_norm_bwd_apply: incomplete pre-activation coefficients");
  const size_t units = (size_t)N * HW * (C >> 3);
  const dim3 bgrid(std::max(1, std::min<int>((int)(((size_t)HW * (C >> 3) + 255) / 256), std::max(1, elem_blocks() / N))), N);
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((norm_bwd_apply_kernel<T>), bgrid, dim3(256), 0,
                                            (hipStream_t)stream, (const T*)dy, (const T*)y, (const T*)x, (const T*)r, k1, k2,
                                            (T*)dx, (T*)dr, HW, C, units, relu, scale1, shift1, scale2, shift2));
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_norm_bwd_apply(const void* dy, const void* y, const void* x, const void* r, const float* k1,
                                  const float* k2, void* dx, void* dr, int N, int HW, int C, int relu, int dtype,
                                  void* stream) {
  return ast_norm_bwd_apply_pre(dy, y, x, r, k1, k2, dx, dr, N, HW, C, relu, dtype, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int ast_bn_apply_fwd(const void* x, const void* r, void* y, const float* tab1, int rows1, long count1, const float* tab2,
                               const float* gamma1, const float* beta1, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                               float eps1, const float* gamma2, const float* beta2, float eps2, float* out1, float* out2, int N, int HW, int C,
                               int Creal, int relu, int dtype, void* stream) {
  if (!x || !y || !tab1 || !gamma1 || !beta1 || !out1 || rows1 < 1 || count1 < 1 || N < 1 || HW < 1 || C < 8 || (C & 7) || C > 2048 || Creal > C)
    AST_FAIL("ast_bn_apply_fwd: bad args (rows %d count %ld N %d HW %d C %d)", rows1, count1, N, HW, C);
  if (r && (!tab2 || !gamma2 || !beta2 || !out2)) AST_FAIL("ast_bn_apply_fwd: the second (InstanceNorm) input needs its table, gamma / beta and out2");
  if ((long)rows1 * C > 65536) AST_FAIL("ast_bn_apply_fwd: statistics table too large for the in-kernel reduction (rows %d x C %d)", rows1, C);
  const int nparts = std::max(1, std::min(rows1, 256 / std::min(C, 256)));
  const size_t lds = sizeof(float) * (size_t)(4 * C + 2 * C * nparts);
  const dim3 grid(std::max(1, std::min<int>((int)(((size_t)HW * (C >> 3) + 255) / 256), std::max(1, elem_blocks() / N))), N);
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((bn_apply_fwd_kernel<T>), grid, dim3(256), lds, (hipStream_t)stream, (const T*)x, (const T*)r, (T*)y, tab1,
                                            rows1, (float)count1, tab2, gamma1, beta1, running_mean, running_var, num_batches_tracked, eps1, gamma2,
                                            beta2, eps2, out1, out2, N, HW, C, Creal, relu));
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_bn_apply_bwd(const void* dy, const void* x, const void* r, void* dx, void* dr, const float* tab3, int rows, long count,
                               const float* gamma1, const float* mean1, const float* rstd1, float* dgamma1, float* dbeta1,
                               const float* gamma2, const float* mean2, const float* rstd2, float* dgamma2, float* dbeta2,
                               const float* scale1, const float* shift1, const float* scale2, const float* shift2, int N, int HW, int C,
                               int Creal, int relu, int dtype, void* stream) {
  if (!dy || !x || !dx || !tab3 || !gamma1 || !mean1 || !rstd1 || rows < 1 || count < 1 || N < 1 || HW < 1 || C < 8 || (C & 7) || C > 2048 || Creal > C)
    AST_FAIL("ast_bn_apply_bwd: bad args (rows %d count %ld N %d HW %d C %d)", rows, count, N, HW, C);
  if (dr && (!r || !gamma2 || !mean2 || !rstd2 || rows != N)) AST_FAIL("ast_bn_apply_bwd: the instance branch needs r, gamma2 / mean2 / rstd2 and a per-image table (rows == N)");
  if (relu && (!scale1 || !shift1 || (r && (!scale2 || !shift2)))) AST_FAIL("ast_bn_apply_bwd: the ReLU mask is recomputed from the pre-activation: scale / shift required");
  if ((long)rows * C > 65536) AST_FAIL("ast_bn_apply_bwd: statistics table too large for the in-kernel reduction (rows %d x C %d)", rows, C);
  const int nparts = std::max(1, std::min(rows, 256 / std::min(C, 256)));
  const size_t lds = sizeof(float) * (size_t)(6 * C + 2 * C * nparts);
  const dim3 grid(std::max(1, std::min<int>((int)(((size_t)HW * (C >> 3) + 255) / 256), std::max(1, elem_blocks() / N))), N);
  AST_DISPATCH_T(dtype, hipLaunchKernelGGL((bn_apply_bwd_kernel<T>), grid, dim3(256), lds, (hipStream_t)stream, (const T*)dy, (const T*)x, (const T*)r,
                                            (T*)dx, (T*)dr, tab3, rows, (float)count, gamma1, mean1, rstd1, dgamma1, dbeta1, gamma2, mean2, rstd2,
                                            dgamma2, dbeta2, scale1, shift1, scale2, shift2, N, HW, C, Creal, relu));
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                 int rows, int D, float eps, int dtype, void* stream) {
  if (dtype != AST_F32) AST_FAIL("ast_layernorm_fwd: f32 only (token tensors are kept in f32)");
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || D <= 0) AST_FAIL("ast_layernorm_fwd: bad args");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)x, gamma,
                     beta, (float*)y, mean, rstd, rows, D, eps);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                 void* dx, float* dgamma, float* dbeta, int rows, int D, int dtype, void* stream) {
  if (dtype != AST_F32) AST_FAIL("ast_layernorm_bwd: f32 only");
  if (!dy || !x || !gamma || !mean || !rstd || !dx || rows <= 0 || D <= 0) AST_FAIL("ast_layernorm_bwd: bad args");
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)dy,
                     (const float*)x, gamma, mean, rstd, (float*)dx, dgamma, dbeta, rows, D);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_add_drop_ln_fwd(const float* x, const float* sub, float* mask, float* s_out, const float* gamma, const float* beta,
                                   float* y, float* mean, float* rstd, int rows, int D, float eps, float p, uint64_t seed,
                                   const int64_t* d_offset, void* stream) {
  if (!sub || !s_out || rows <= 0 || D <= 0 || p < 0.f || p >= 1.f) AST_FAIL("ast_add_drop_ln_fwd: bad args");
  if (p > 0.f && !mask) AST_FAIL("ast_add_drop_ln_fwd: dropout needs a mask buffer");
  if (gamma && (!beta || !y || !mean || !rstd)) AST_FAIL("ast_add_drop_ln_fwd: LayerNorm outputs missing");
  const bool al = ((((uintptr_t)x) | ((uintptr_t)sub) | ((uintptr_t)mask) | ((uintptr_t)s_out) | ((uintptr_t)gamma) | ((uintptr_t)beta) |
                    ((uintptr_t)y)) & 15) == 0;
  if (D == 256 && al)
    hipLaunchKernelGGL(add_drop_ln_fwd256_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, sub, mask, s_out, gamma,
                       beta, y, mean, rstd, rows, eps, p, seed, d_offset);
  else
    hipLaunchKernelGGL(add_drop_ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, sub, mask, s_out, gamma, beta,
                       y, mean, rstd, rows, D, eps, p, seed, d_offset);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_add_drop_ln_bwd(const float* dy, const float* ds_ext, const float* s, const float* gamma, const float* mean,
                                   const float* rstd, const float* mask, float* dx, float* dsub, float* dgamma, float* dbeta, int rows,
                                   int D, void* stream) {
  if ((!dy && !ds_ext) || !dsub || rows <= 0 || D <= 0) AST_FAIL("ast_add_drop_ln_bwd: bad args");
  if (dy && (!s || !gamma || !mean || !rstd)) AST_FAIL("ast_add_drop_ln_bwd: LayerNorm state missing");
  const bool al = ((((uintptr_t)dy) | ((uintptr_t)ds_ext) | ((uintptr_t)s) | ((uintptr_t)gamma) | ((uintptr_t)mask) | ((uintptr_t)dx) |
                    ((uintptr_t)dsub)) & 15) == 0;
  if (D == 256 && al)
    hipLaunchKernelGGL(add_drop_ln_bwd256_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, ds_ext, s, gamma, mean, rstd,
                       mask, dx, dsub, dgamma, dbeta, rows);
  else
    hipLaunchKernelGGL(add_drop_ln_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, ds_ext, s, gamma, mean, rstd,
                       mask, dx, dsub, dgamma, dbeta, rows, D);
  AST_CHECK_LAUNCH();
  return 0;
}
