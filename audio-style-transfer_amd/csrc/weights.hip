// Spectral normalisation (one power iteration per training forward, torch
// nn/utils/spectral_norm.py:92-114) fused with packing of W/sigma into the two
// GEMM layouts, batched over ALL weights of a model in three launches
// (launch boundaries act as the grid-wide syncs between v, u and sigma), and
// the matching backward:  dW_orig = (dW - <dW, W/sigma> u v^T) / sigma.
#include <type_traits>
#include "ast_common.h"
#include "../../include/ast_hip.h"

namespace {

__device__ __forceinline__ size_t woff(const ast_weight_desc_t& d, int co, int ci, int tap) {
  return (size_t)co * d.s_co + (size_t)ci * d.s_ci + tap;
}

// t[j] = sum_co W(co, j) u[co],  j = ci*KK + tap  -> scratch[Co + j]
// The rows are split over grid.z (RZ chunks) for parallelism.  Every chunk STORES its partial sum into its own slab
// scratch[Co + ncols*(1 + z) ..] and sn_t_sum_kernel adds the slabs in a fixed order: u, v and sigma are then
// bit-reproducible (they are never communicated between data-parallel replicas, so every replica must compute the same
// bits from the same weights -- SURVEY 5.8; f32 atomics, used here before, made them depend on arrival order).
constexpr int RZ = 32;
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const ast_weight_desc_t* __restrict__ descs) {
  const ast_weight_desc_t d = descs[blockIdx.y];
  if (!d.u || !d.power_iter) return;
  const int ncols = d.Ci * d.KK;
  const int rows_per = (d.Co + RZ - 1) / RZ;
  const int c0 = blockIdx.z * rows_per, c1 = min(d.Co, c0 + rows_per);
  if (c0 >= c1) return;
  float* part = d.scratch + d.Co + (size_t)ncols * (1 + blockIdx.z);
  if (d.s_ci == d.KK && (ncols & 3) == 0 && (d.s_co & 3) == 0 && (((uintptr_t)d.w) & 15) == 0) {
    // Conv2d / Linear layout: rows are contiguous in j -> each thread owns 4 columns (16-byte loads); the first
    // quarter of the grid covers all columns, the rest exits
    const int j4 = blockIdx.x * 256 + threadIdx.x;
    if (j4 >= (ncols >> 2)) return;
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    for (int co = c0; co < c1; ++co) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(d.w + (size_t)co * d.s_co + j4 * 4);
      const float uc = d.u[co];
      acc4[0] += w4[0] * uc; acc4[1] += w4[1] * uc; acc4[2] += w4[2] * uc; acc4[3] += w4[3] * uc;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[j4 * 4 + e] = acc4[e];
    return;
  }
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= ncols) return;
  const int ci = j / d.KK, tap = j - ci * d.KK;
  float acc = 0.f;
  for (int co = c0; co < c1; ++co) acc += d.w[woff(d, co, ci, tap)] * d.u[co];
  part[j] = acc;
}

// t[j] = sum over the row chunks, in chunk order
__global__ __launch_bounds__(256) void sn_t_sum_kernel(const ast_weight_desc_t* __restrict__ descs) {
  const ast_weight_desc_t d = descs[blockIdx.y];
  if (!d.u || !d.power_iter) return;
  const int ncols = d.Ci * d.KK;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= ncols) return;
  const int rows_per = (d.Co + RZ - 1) / RZ;
  const int nz = (d.Co + rows_per - 1) / rows_per;           // chunks that hold rows
  const float* part = d.scratch + d.Co + ncols;
  float t = 0.f;
  for (int z = 0; z < nz; ++z) t += part[(size_t)z * ncols + j];
  d.scratch[d.Co + j] = t;
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
    if (d.s_ci == d.KK && (ncols & 3) == 0 && (d.s_co & 3) == 0 && ((((uintptr_t)vec) | ((uintptr_t)d.w)) & 15) == 0) {
      // Conv2d / Linear layout [co][ci][tap]: the row is contiguous in j -- 16-byte loads, no (ci, tap) split
      // (the generic loop spends a run-time division per element)
      const f32x4* wr = reinterpret_cast<const f32x4*>(d.w + (size_t)row * d.s_co);
      const f32x4* vr = reinterpret_cast<const f32x4*>(vec);
      for (int j4 = lane; j4 < (ncols >> 2); j4 += 64) {
        const f32x4 a = wr[j4], b = vr[j4];
        acc += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
      }
    } else {
      for (int j = lane; j < ncols; j += 64) {
        const int ci = j / d.KK, tap = j - ci * d.KK;
        acc += d.w[woff(d, row, ci, tap)] * vec[j];
      }
    }
    acc = wave_sum(acc) * inv;
    if (lane == 0) d.scratch[row] = acc;
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


// ---------------------------------------------------------------------------
// LDS-tiled forms of pack / flush.  A tile = 32 out-channels x 32 in-channels x all KK taps of one
// weight (<= 9216 floats).  Global accesses are contiguous runs in every layout:
//   master w / grad : runs of 32*KK floats along the master's inner channel dimension
//   wf / dW (wf layout) [co][tap][ci]: runs of 32 ci;   wb / dW (wb layout) [ci][tap][co]: runs of 32 co
// The permutation between them happens in LDS.  Tiles of all weights of a model are listed in a
// device array {weight index, co0, ci0} built once by the host.
// ---------------------------------------------------------------------------
struct WTile { int w, co0, ci0, pad; };
constexpr int TL = 32;                       // tile edge (channels)
constexpr int LP = TL + 1;                   // LDS pitch of the [tap][co][ci] image (floats)

__device__ __forceinline__ bool tile_is_full(const ast_weight_desc_t& d, int co0, int ci0) {
  return co0 + TL <= d.Co && ci0 + TL <= d.Ci;
}
// the inner (channel, tap) run of the master layout is contiguous and 16-byte aligned for every outer index
__device__ __forceinline__ bool master_vec_ok(const ast_weight_desc_t& d, const float* base) {
  const int inner = d.s_co >= d.s_ci ? d.s_ci : d.s_co, outer = d.s_co >= d.s_ci ? d.s_co : d.s_ci;
  return inner == d.KK && (outer & 3) == 0 && (((uintptr_t)base) & 15) == 0;
}

// L[tap][co_l][ci_l] <- master w (coalesced along the master's contiguous dimension)
// KKC = compile-time tap count (1 and 9 cover every weight of the model: the index arithmetic of these loops is a
// division by KK per element, ~40 VALU each with a run-time divisor); KKC = 0 keeps the run-time form.
template <int KKC>
__device__ __forceinline__ void tile_load_master(const ast_weight_desc_t& d, const float* __restrict__ base, int co0, int ci0,
                                                 float* __restrict__ L, float scale) {
  const int KK = KKC > 0 ? KKC : d.KK;
  const bool co_outer = d.s_co >= d.s_ci;    // conv: [co][ci][tap];  convT: [ci][co][tap]
  const int run = TL * KK;                   // contiguous floats per outer index (inner channel x tap)
  if (KKC > 0 && tile_is_full(d, co0, ci0) && master_vec_ok(d, base)) {
    // full tile: every outer index owns one 16-byte-aligned run of 32*KK floats -> float4 loads
    const int run4 = run >> 2;
    for (int idx = threadIdx.x; idx < TL * run4; idx += 256) {
      const int o = idx / run4, r4 = idx - o * run4;
      const size_t off = co_outer ? (size_t)(co0 + o) * d.s_co + (size_t)ci0 * d.s_ci : (size_t)(ci0 + o) * d.s_ci + (size_t)co0 * d.s_co;
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(base + off + r4 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = r4 * 4 + e;
        const int in = r / KK, tap = r - in * KK;
        const int co_l = co_outer ? o : in, ci_l = co_outer ? in : o;
        L[(tap * TL + co_l) * LP + ci_l] = v4[e] * scale;
      }
    }
    return;
  }
  for (int idx = threadIdx.x; idx < TL * run; idx += 256) {
    const int o = idx / run, r = idx - o * run;
    const int in = r / KK, tap = r - in * KK;
    const int co = co_outer ? co0 + o : co0 + in, ci = co_outer ? ci0 + in : ci0 + o;
    const float v = (co < d.Co && ci < d.Ci) ? base[(size_t)co * d.s_co + (size_t)ci * d.s_ci + tap] * scale : 0.f;
    L[(tap * TL + (co - co0)) * LP + (ci - ci0)] = v;
  }
}

template <typename T, int KKC>
__device__ __forceinline__ void tile_store_packed(const ast_weight_desc_t& d, int co0, int ci0, const float* __restrict__ L) {
  const int KK = KKC > 0 ? KKC : d.KK;
  if (KKC > 0 && tile_is_full(d, co0, ci0) && (d.Cip & 7) == 0 && (d.Cop & 7) == 0) {
    // full tile: each (channel, tap) row of 32 packed values is written as four 8-element vectors
    for (int idx = threadIdx.x; idx < TL * KK * 4; idx += 256) {
      const int q = idx & 3, row = idx >> 2;
      const int out_l = row / KK, tap = row - out_l * KK;
      if (d.wf) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = L[(tap * TL + out_l) * LP + q * 8 + e];
        U8<T>::store((T*)d.wf + ((size_t)(co0 + out_l) * KK + tap) * d.Cip + ci0 + q * 8, v);
      }
      if (d.wb) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = L[(tap * TL + q * 8 + e) * LP + out_l];
        U8<T>::store((T*)d.wb + ((size_t)(ci0 + out_l) * KK + tap) * d.Cop + co0 + q * 8, v);
      }
    }
    return;
  }
  // wf[(co*KK + tap)*Cip + ci]: rows (co_l, tap), 32 ci each
  if (d.wf) {
    T* wf = (T*)d.wf;
    for (int idx = threadIdx.x; idx < TL * KK * TL; idx += 256) {
      const int ci_l = idx & (TL - 1), row = idx >> 5;
      const int co_l = row / KK, tap = row - co_l * KK;
      const int co = co0 + co_l, ci = ci0 + ci_l;
      if (co < d.Cop && ci < d.Cip) wf[((size_t)co * KK + tap) * d.Cip + ci] = (T)L[(tap * TL + co_l) * LP + ci_l];
    }
  }
  if (d.wb) {
    T* wb = (T*)d.wb;
    for (int idx = threadIdx.x; idx < TL * KK * TL; idx += 256) {
      const int co_l = idx & (TL - 1), row = idx >> 5;
      const int ci_l = row / KK, tap = row - ci_l * KK;
      const int co = co0 + co_l, ci = ci0 + ci_l;
      if (co < d.Cop && ci < d.Cip) wb[((size_t)ci * KK + tap) * d.Cop + co] = (T)L[(tap * TL + co_l) * LP + ci_l];
    }
  }
}

// sigma / u for every spectral-normalised weight (after sn_w_v_kernel); one block per weight
__global__ __launch_bounds__(256) void sn_sigma_kernel(const ast_weight_desc_t* __restrict__ descs) {
  __shared__ float red[17];
  const ast_weight_desc_t d = descs[blockIdx.x];
  if (d.inner && d.power_iter && threadIdx.x == 0) d.inner[0] = 0.f;
  if (!d.u) { if (threadIdx.x == 0 && d.sigma) d.sigma[0] = 1.f; return; }
  float sigma;
  if (d.power_iter) {
    float q = 0.f;
    for (int i = threadIdx.x; i < d.Co; i += 256) q += d.scratch[i] * d.scratch[i];
    const float nrm = sqrtf(block_sum(q, red));
    const float inv = 1.f / fmaxf(nrm, 1e-12f);
    sigma = nrm * nrm * inv;
    for (int i = threadIdx.x; i < d.Co; i += 256) d.u[i] = d.scratch[i] * inv;
  } else {
    float q = 0.f;
    for (int i = threadIdx.x; i < d.Co; i += 256) q += d.scratch[i] * d.u[i];
    sigma = block_sum(q, red);
  }
  if (threadIdx.x == 0) d.sigma[0] = sigma;
  __syncthreads();
}

template <int KKC>
__device__ __forceinline__ void pack_tile_body(const ast_weight_desc_t& d, const WTile& tl, int dtype, float* __restrict__ L) {
  const int KK = KKC > 0 ? KKC : d.KK;
  tile_load_master<KKC>(d, d.w, tl.co0, tl.ci0, L, 1.f / d.sigma[0]);
  __syncthreads();
  if (dtype == AST_BF16) tile_store_packed<bf16_t, KKC>(d, tl.co0, tl.ci0, L); else tile_store_packed<float, KKC>(d, tl.co0, tl.ci0, L);
  const int nrep = d.dwp_replicas > 1 ? d.dwp_replicas : 1;           // gradient replicas (ast_wgrad_rep): all of them start at zero
  const size_t rstride = (size_t)d.Cop * KK * d.Cip;
  if (d.dwp && d.power_iter && KKC > 0 && tile_is_full(d, tl.co0, tl.ci0) && (d.Cip & 3) == 0 && (d.Cop & 3) == 0) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int idx = threadIdx.x; idx < TL * KK * 8; idx += 256) {
      const int q = idx & 7, row = idx >> 3;
      const int out_l = row / KK, tap = row - out_l * KK;
      float* p = d.dwp_from_wb ? d.dwp + ((size_t)(tl.ci0 + out_l) * KK + tap) * d.Cop + tl.co0 : d.dwp + ((size_t)(tl.co0 + out_l) * KK + tap) * d.Cip + tl.ci0;
      for (int r = 0; r < nrep; ++r) *reinterpret_cast<f32x4*>(p + r * rstride + q * 4) = z;
    }
  } else if (d.dwp && d.power_iter) {         // fresh gradient staging for this step (this tile's slice, both layouts cover it once)
    for (int idx = threadIdx.x; idx < TL * KK * TL; idx += 256) {
      const int in_l = idx & (TL - 1), row = idx >> 5;
      const int out_l = row / KK, tap = row - out_l * KK;
      for (int r = 0; r < nrep; ++r) {
        float* base = d.dwp + r * rstride;
        if (d.dwp_from_wb) { const int ci = tl.ci0 + out_l, co = tl.co0 + in_l; if (ci < d.Cip && co < d.Cop) base[((size_t)ci * KK + tap) * d.Cop + co] = 0.f; }
        else { const int co = tl.co0 + out_l, ci = tl.ci0 + in_l; if (co < d.Cop && ci < d.Cip) base[((size_t)co * KK + tap) * d.Cip + ci] = 0.f; }
      }
    }
  }
}

__global__ __launch_bounds__(256) void pack_tiles_kernel(const ast_weight_desc_t* __restrict__ descs, const int* __restrict__ dtypes,
                                                          const WTile* __restrict__ tiles) {
  __shared__ float L[9 * TL * LP];
  const WTile tl = tiles[blockIdx.x];
  const ast_weight_desc_t d = descs[tl.w];
  const int dt = dtypes[tl.w];
  if (d.KK == 9) pack_tile_body<9>(d, tl, dt, L);
  else if (d.KK == 1) pack_tile_body<1>(d, tl, dt, L);
  else pack_tile_body<0>(d, tl, dt, L);
}

// G[tap][co_l][ci_l] <- packed gradient staging (coalesced along its contiguous channel dimension)
template <int KKC>
__device__ __forceinline__ void tile_load_dwp(const ast_weight_desc_t& d, int co0, int ci0, float* __restrict__ G) {
  const int KK = KKC > 0 ? KKC : d.KK;
  const int nrep = d.dwp_replicas > 1 ? d.dwp_replicas : 1;           // gradient replicas (ast_wgrad_rep) are summed here
  const size_t rstride = (size_t)d.Cop * KK * d.Cip;
  if (KKC > 0 && tile_is_full(d, co0, ci0) && (d.Cip & 3) == 0 && (d.Cop & 3) == 0) {
    for (int idx = threadIdx.x; idx < TL * KK * 8; idx += 256) {
      const int q = idx & 7, row = idx >> 3;
      const int out_l = row / KK, tap = row - out_l * KK;
      const float* p = d.dwp_from_wb ? d.dwp + ((size_t)(ci0 + out_l) * KK + tap) * d.Cop + co0 : d.dwp + ((size_t)(co0 + out_l) * KK + tap) * d.Cip + ci0;
      f32x4 v4 = *reinterpret_cast<const f32x4*>(p + q * 4);
      if (nrep == 8) {                              // the configured count: seven independent loads in flight, fixed summation order
        f32x4 t[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) t[r] = *reinterpret_cast<const f32x4*>(p + (r + 1) * rstride + q * 4);
        v4 = ((v4 + t[0]) + (t[1] + t[2])) + ((t[3] + t[4]) + (t[5] + t[6]));
      } else {
        for (int r = 1; r < nrep; ++r) v4 += *reinterpret_cast<const f32x4*>(p + r * rstride + q * 4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int in_l = q * 4 + e;
        const int co_l = d.dwp_from_wb ? in_l : out_l, ci_l = d.dwp_from_wb ? out_l : in_l;
        G[(tap * TL + co_l) * LP + ci_l] = v4[e];
      }
    }
    return;
  }
  for (int idx = threadIdx.x; idx < TL * KK * TL; idx += 256) {
    const int in_l = idx & (TL - 1), row = idx >> 5;
    const int out_l = row / KK, tap = row - out_l * KK;
    float v = 0.f;
    int co_l, ci_l;
    bool ok;
    size_t off;
    if (d.dwp_from_wb) { ci_l = out_l; co_l = in_l; const int ci = ci0 + ci_l, co = co0 + co_l; ok = ci < d.Ci && co < d.Co; off = ((size_t)ci * KK + tap) * d.Cop + co; }
    else { co_l = out_l; ci_l = in_l; const int co = co0 + co_l, ci = ci0 + ci_l; ok = co < d.Co && ci < d.Ci; off = ((size_t)co * KK + tap) * d.Cip + ci; }
    if (ok) {
      if (nrep == 8) {                              // eight independent loads in flight (a run-time loop serialised them: 100 us per tile)
        float t[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) t[r] = d.dwp[r * rstride + off];
        v = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
      } else {
        for (int r = 0; r < nrep; ++r) v += d.dwp[r * rstride + off];
      }
    }
    G[(tap * TL + co_l) * LP + ci_l] = v;
  }
}

__global__ __launch_bounds__(256) void flush_inner_tiles_kernel(const ast_weight_desc_t* __restrict__ descs, const WTile* __restrict__ tiles) {
  __shared__ float G[9 * TL * LP];
  __shared__ float red[17];
  const WTile tl = tiles[blockIdx.x];
  const ast_weight_desc_t d = descs[tl.w];
  if (!d.dwp || !d.u) return;
  float q = 0.f;
  auto body = [&](auto kkc) __attribute__((always_inline)) {
    constexpr int KKC = decltype(kkc)::value;
    const int KK = KKC > 0 ? KKC : d.KK;
    tile_load_dwp<KKC>(d, tl.co0, tl.ci0, G);
    __syncthreads();
    // walk the master in ITS contiguous order and pick the matching staged gradient from LDS
    const bool co_outer = d.s_co >= d.s_ci;
    const int run = TL * KK;
    if (KKC > 0 && tile_is_full(d, tl.co0, tl.ci0) && master_vec_ok(d, d.w)) {
      const int run4 = run >> 2;
      for (int idx = threadIdx.x; idx < TL * run4; idx += 256) {
        const int o = idx / run4, r4 = idx - o * run4;
        const size_t off = co_outer ? (size_t)(tl.co0 + o) * d.s_co + (size_t)tl.ci0 * d.s_ci : (size_t)(tl.ci0 + o) * d.s_ci + (size_t)tl.co0 * d.s_co;
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(d.w + off + r4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = r4 * 4 + e;
          const int in = r / KK, tap = r - in * KK;
          const int co_l = co_outer ? o : in, ci_l = co_outer ? in : o;
          q += G[(tap * TL + co_l) * LP + ci_l] * w4[e];
        }
      }
      return;
    }
    for (int idx = threadIdx.x; idx < TL * run; idx += 256) {
      const int o = idx / run, r = idx - o * run;
      const int in = r / KK, tap = r - in * KK;
      const int co_l = co_outer ? o : in, ci_l = co_outer ? in : o;
      const int co = tl.co0 + co_l, ci = tl.ci0 + ci_l;
      if (co < d.Co && ci < d.Ci) q += G[(tap * TL + co_l) * LP + ci_l] * d.w[(size_t)co * d.s_co + (size_t)ci * d.s_ci + tap];
    }
  };
  if (d.KK == 9) body(std::integral_constant<int, 9>{});
  else if (d.KK == 1) body(std::integral_constant<int, 1>{});
  else body(std::integral_constant<int, 0>{});
  q = block_sum(q, red);
  if (threadIdx.x == 0 && q != 0.f) unsafeAtomicAdd(d.inner, q / d.sigma[0]);
}

__global__ __launch_bounds__(256) void flush_unpack_tiles_kernel(const ast_weight_desc_t* __restrict__ descs, const WTile* __restrict__ tiles) {
  __shared__ float G[9 * TL * LP];
  const WTile tl = tiles[blockIdx.x];
  const ast_weight_desc_t d = descs[tl.w];
  if (!d.dwp || !d.grad) return;
  const float inner = d.u ? d.inner[0] : 0.f;
  const float inv_sigma = d.u ? 1.f / d.sigma[0] : 1.f;
  auto body = [&](auto kkc) __attribute__((always_inline)) {
    constexpr int KKC = decltype(kkc)::value;
    const int KK = KKC > 0 ? KKC : d.KK;
    tile_load_dwp<KKC>(d, tl.co0, tl.ci0, G);
    __syncthreads();
    const bool co_outer = d.s_co >= d.s_ci;
    const int run = TL * KK;
    if (KKC > 0 && tile_is_full(d, tl.co0, tl.ci0) && master_vec_ok(d, d.grad)) {
      const int run4 = run >> 2;
      for (int idx = threadIdx.x; idx < TL * run4; idx += 256) {
        const int o = idx / run4, r4 = idx - o * run4;
        const size_t off = co_outer ? (size_t)(tl.co0 + o) * d.s_co + (size_t)tl.ci0 * d.s_ci : (size_t)(tl.ci0 + o) * d.s_ci + (size_t)tl.co0 * d.s_co;
        f32x4* gp = reinterpret_cast<f32x4*>(d.grad + off + r4 * 4);
        f32x4 g4 = *gp;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = r4 * 4 + e;
          const int in = r / KK, tap = r - in * KK;
          const int co_l = co_outer ? o : in, ci_l = co_outer ? in : o;
          float gv = G[(tap * TL + co_l) * LP + ci_l];
          if (d.u) gv = (gv - inner * d.u[tl.co0 + co_l] * d.v[(tl.ci0 + ci_l) * KK + tap]) * inv_sigma;
          g4[e] += gv;
        }
        *gp = g4;
      }
      return;
    }
    for (int idx = threadIdx.x; idx < TL * run; idx += 256) {
      const int o = idx / run, r = idx - o * run;
      const int in = r / KK, tap = r - in * KK;
      const int co_l = co_outer ? o : in, ci_l = co_outer ? in : o;
      const int co = tl.co0 + co_l, ci = tl.ci0 + ci_l;
      if (co < d.Co && ci < d.Ci) {
        float gv = G[(tap * TL + co_l) * LP + ci_l];
        if (d.u) gv = (gv - inner * d.u[co] * d.v[ci * KK + tap]) * inv_sigma;
        d.grad[(size_t)co * d.s_co + (size_t)ci * d.s_ci + tap] += gv;
      }
    }
  };
  if (d.KK == 9) body(std::integral_constant<int, 9>{});
  else if (d.KK == 1) body(std::integral_constant<int, 1>{});
  else body(std::integral_constant<int, 0>{});
}
}  // namespace

extern "C" int ast_weights_prepare_t(const ast_weight_desc_t* descs, const int* dtypes, int n, int max_co, int max_cols,
                                     const void* tiles, int ntiles, void* stream) {
  if (!descs || !dtypes || !tiles || n <= 0 || ntiles <= 0 || max_co <= 0 || max_cols <= 0) AST_FAIL("ast_weights_prepare_t: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sn_wt_u_kernel, dim3((max_cols + 255) / 256, n, RZ), dim3(256), 0, s, descs);
  hipLaunchKernelGGL(sn_t_sum_kernel, dim3((max_cols + 255) / 256, n), dim3(256), 0, s, descs);
  hipLaunchKernelGGL(sn_w_v_kernel, dim3((max_co + 3) / 4, n), dim3(256), 0, s, descs);
  hipLaunchKernelGGL(sn_sigma_kernel, dim3(n), dim3(256), 0, s, descs);
  hipLaunchKernelGGL(pack_tiles_kernel, dim3(ntiles), dim3(256), 0, s, descs, dtypes, (const WTile*)tiles);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_weight_grads_flush_t(const ast_weight_desc_t* descs, const void* tiles, int ntiles, void* stream) {
  if (!descs || !tiles || ntiles <= 0) AST_FAIL("ast_weight_grads_flush_t: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(flush_inner_tiles_kernel, dim3(ntiles), dim3(256), 0, s, descs, (const WTile*)tiles);
  hipLaunchKernelGGL(flush_unpack_tiles_kernel, dim3(ntiles), dim3(256), 0, s, descs, (const WTile*)tiles);
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

extern "C" long ast_sn_scratch_floats(int Co, int ncols) {
  if (Co < 1 || ncols < 1) return -1;
  return (long)Co + (long)ncols * (1 + RZ);          // s = W v, t = W^T u, and the RZ row-chunk partial sums of t
}
