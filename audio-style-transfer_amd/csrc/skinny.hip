// Token-sized GEMMs (M <= 64 rows: transformer / projection / discriminator
// linears at B*(S+1) <= 40 rows).  An MFMA tile would be >75 % padding and the
// launch is latency-bound, so one workgroup computes one 16x16 f32 MFMA tile with its
// four waves splitting K (LDS reduce): short dependency chains, every weight byte read
// once.  The weight gradient + bias gradient are one launch that adds
// straight into the parameter gradient (no packed staging: plain linears have no
// spectral norm).
#include "ast_common.h"
#include "../../include/ast_hip.h"

namespace {

// y[m][n] = act(sum_k x[m][k] * w[n][k] + b[n]).  One workgroup = one 16(n) x 16(m) output tile; its 4 waves
// split K and are reduced through LDS.  v_mfma_f32_16x16x4_f32 (exact f32): the weight tile is the A operand
// so each lane ends up with 4 consecutive n of one token row m -> one 16-byte store.  Lane (i = l&15, g = l>>4)
// loads 16 B of row i at k = kb + 16 s + 4 g: the four MFMAs of a step contract k = 4 g + e over g (e = 0..3),
// the same k permutation on both operands.
// NW = waves per workgroup that split K (4; 8 for K = 1024, the FFN's second linear and the data gradient of its first:
// 16 dependent-free loads + 64 MFMAs per wave made those launches 8.9 us against 4.8 us for the K = 256 ones).
template <int NS, int NW = 4>
__global__ __launch_bounds__(64 * NW) void skinny_gemm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int M, int N,
                                                           int K, int ldx, int ldw, int ldy, int relu,
                                                           const float* __restrict__ mul_mask, float* __restrict__ drop_mask,
                                                           float p, uint64_t seed, const int64_t* __restrict__ d_offset) {
  __shared__ f32x4 part[NW][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  constexpr int kslice = NS * 16;                     // k per wave (host: NW * kslice >= K)
  const int kb = wave * kslice;
  const bool nv = n0 + i < N, mv = m0 + i < M;
  const float* wr = w + (size_t)(nv ? n0 + i : 0) * ldw;
  const float* xr = x + (size_t)(mv ? m0 + i : 0) * ldx;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // the epilogue's bias values (lane: n = n0 + 4 g .. +3) are fetched with the first loads, not after the reduction
  float bq[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int n = n0 + 4 * g + q; bq[q] = bias[n < N ? n : 0]; }
  }
  f32x4 wl[NS], xl[NS];                               // every load of the wave's K slice is issued before the first MFMA
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int k = kb + s * 16 + 4 * g;
    const bool kv = k < K;
    wl[s] = *reinterpret_cast<const f32x4*>(wr + (kv ? k : 0));
    xl[s] = *reinterpret_cast<const f32x4*>(xr + (kv ? k : 0));
  }
  // Without this fence the scheduler interleaves load -> s_waitcnt vmcnt(0) -> MFMA per step: 2*NS dependent round trips
  // (13.5 us per launch at K = 1024 instead of ~4).  The masking selects sit on the far side so they cannot pull
  // the waits forward.
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const bool kv = kb + s * 16 + 4 * g < K;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 a = (kv && nv) ? wl[s] : z, b = (kv && mv) ? xl[s] : z;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave != 0) return;
  f32x4 r = part[0][lane];
#pragma unroll
  for (int q = 1; q < NW; ++q) { const f32x4 t = part[q][lane]; r[0] += t[0]; r[1] += t[1]; r[2] += t[2]; r[3] += t[3]; }
  const int m = m0 + i, nb = n0 + 4 * g;              // D[row = 4 g + r (n)][col = i (m)]
  if (m >= M) return;
  // optional epilogues of the fused FFN: drop_mask != null draws the dropout mask here and stores the COMBINED
  // mask (0 where ReLU or dropout zeroed the unit, else 1/(1-p)) for the backward pass; mul_mask != null applies
  // such a mask to the result (the backward's dh = (dy W2) * mask)
  const uint64_t base = drop_mask ? mix64(seed ^ mix64((uint64_t)(d_offset ? *d_offset : 0))) : 0;
  const float keep = 1.f / (1.f - p);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = nb + q;
    if (n < N) {
      const size_t o = (size_t)m * ldy + n;
      float v = r[q] + bq[q];
      if (relu) v = fmaxf(v, 0.f);
      if (drop_mask) {
        const float km = dropout_keep(base, o, p, keep);
        drop_mask[o] = (relu && v <= 0.f) ? 0.f : km;
        v *= km;
      }
      if (mul_mask) v *= mul_mask[o];
      y[o] = v;
    }
  }
}

// dW[n][k] += sum_m dy[m][n] x[m][k] ; db[n] += sum_m dy[m][n]   (contraction over the <= 64 token rows).
// One wave = 16 n x 64 k of dW on v_mfma_f32_16x16x4_f32: A[i][g] = dy[4s+g][n0+i], B[g][j] = x[4s+g][k0+j]
// (16 lanes read 16 consecutive floats), the A fragment reused by 4 k-tiles.
template <int MS>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            float* __restrict__ dW, float* __restrict__ db, int M, int N, int K,
                                                            int lddy, int ldw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int n0 = (blockIdx.y * 4 + wave) * 16, k0 = blockIdx.x * 64;
  if (n0 >= N) return;
  const bool nv = n0 + i < N;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  float av[MS], bv[MS][4];                            // all <= 64 token rows are loaded before the first MFMA
#pragma unroll
  for (int st = 0; st < MS; ++st) {
    const int m = st * 4 + g;
    const bool mv = m < M;
    av[st] = dy[(size_t)(mv ? m : 0) * lddy + (nv ? n0 + i : 0)];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = k0 + t * 16 + i;
      const bool kv = mv && k < K;
      bv[st][t] = x[(size_t)(kv ? m : 0) * K + (kv ? k : 0)];
    }
  }
  __builtin_amdgcn_sched_barrier(0);                       // all loads in flight before the first MFMA (see skinny_gemm_kernel)
#pragma unroll
  for (int st = 0; st < MS; ++st) {
    const bool mv = st * 4 + g < M;
    const float a = (mv && nv) ? av[st] : 0.f;
    bsum += a;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float b = (mv && k0 + t * 16 + i < K) ? bv[st][t] : 0.f;
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
  }
  // D[row = n0 + 4 g + r][col = k0 + 16 t + i]; the 16 old values are fetched together, then added and stored
  float old[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int k = k0 + t * 16 + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * g + r;
      old[t][r] = (k < K && n < N) ? dW[(size_t)n * ldw + k] : 0.f;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int k = k0 + t * 16 + i;
    if (k >= K) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * g + r;
      if (n < N) dW[(size_t)n * ldw + k] = old[t][r] + acc[t][r];
    }
  }
  if (db && blockIdx.x == 0) {
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (g == 0 && nv) db[n0 + i] += bsum;
  }
}
// ---- the two 294 462 x 256 linears of SimpleDecoder_TransformerOnly.py:16-17 on <= 64 token rows ---------------
// Both are streams over a 301 MB weight matrix with a few token rows: HBM-bound, f32 MFMA 16x16x4.
//
// Y[m][n] += sum_k X[m][k] W[n][k]   (stft_to_embedding: N = 256, K = 2*287*513).  K is even but not a multiple of
// 4, so rows are only 8-byte aligned: lane (i, g) loads 2 floats at k = kb + 8 s + 2 g and the two MFMAs of a step
// contract k = 2 g + e over g.  grid = (N/16, M/16, K chunks of 1024); a workgroup's 4 waves split its chunk, reduce
// through LDS and add the 16x16 partial tile into Y with f32 atomics (Y is pre-zeroed; chunk 0 adds the bias).
__global__ __launch_bounds__(256) void bigk_gemm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int M, int N, int K,
                                                         int ldy) {
  __shared__ f32x4 part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int kb = blockIdx.z * 1024 + wave * 256;
  const bool nv = n0 + i < N, mv = m0 + i < M;
  const float* wr = w + (size_t)(nv ? n0 + i : 0) * K;
  const float* xr = x + (size_t)(mv ? m0 + i : 0) * K;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int sb = 0; sb < 32; sb += 8) {
    f32x2 wl[8], xl[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int k = kb + (sb + s) * 8 + 2 * g;
      const bool kv = k < K;                                   // K even: k and k+1 are valid together
      wl[s] = *reinterpret_cast<const f32x2*>(wr + (kv ? k : 0));
      xl[s] = *reinterpret_cast<const f32x2*>(xr + (kv ? k : 0));
    }
    __builtin_amdgcn_sched_barrier(0);                         // 16 loads in flight together (see skinny_gemm_kernel)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const bool kv = kb + (sb + s) * 8 + 2 * g < K;
      const f32x2 z = {0.f, 0.f};
      const f32x2 a = (kv && nv) ? wl[s] : z, b = (kv && mv) ? xl[s] : z;
#pragma unroll
      for (int e = 0; e < 2; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave != 0) return;
  f32x4 r = part[0][lane];
#pragma unroll
  for (int q = 1; q < 4; ++q) { const f32x4 t = part[q][lane]; r[0] += t[0]; r[1] += t[1]; r[2] += t[2]; r[3] += t[3]; }
  const int m = m0 + i, nb = n0 + 4 * g;                      // D[row = 4 g + r (n)][col = i (m)]
  if (m >= M) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = nb + q;
    if (n < N) unsafeAtomicAdd(y + (size_t)m * ldy + n, r[q] + ((bias && blockIdx.z == 0) ? bias[n] : 0.f));
  }
}

// dX[m][k] += sum_n dY[m][n] W[n][k]   (data gradient of embedding_to_stft: N = 2*287*513 contracted, K = 256 kept).
// A workgroup owns a chunk of 512 n for ALL k: wave w keeps the k tiles 4w..4w+3 (D[k][m] = sum_n W[n][k] dY[m][n];
// lane (i, g): A = W[nb + 4 s + g][k0 + i], 16 lanes = 64 contiguous bytes of a weight row; B = dY[m0 + i][nb + 4 s + g]).
__global__ __launch_bounds__(256) void bign_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                          float* __restrict__ dx, int M, int N, int K, int lddy) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int nb0 = blockIdx.x * 512, m0 = blockIdx.y * 16;
  const int kt0 = wave * 4;                                    // first of this wave's four 16-wide k tiles (K <= 256)
  const bool mv = m0 + i < M;
  const float* dr = dy + (size_t)(mv ? m0 + i : 0) * lddy;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int sb = 0; sb < 128; sb += 8) {
    float a[8][4], b[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int n = nb0 + (sb + s) * 4 + g;
      const bool nvv = n < N;
      b[s] = dr[nvv ? n : 0];
      const float* wrow = w + (size_t)(nvv ? n : 0) * K;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k = (kt0 + t) * 16 + i;
        a[s][t] = wrow[k < K ? k : 0];
      }
    }
    __builtin_amdgcn_sched_barrier(0);                         // 40 loads in flight together (see skinny_gemm_kernel)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const bool nvv = nb0 + (sb + s) * 4 + g < N;
      const float bq = (nvv && mv) ? b[s] : 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float aq = (nvv && (kt0 + t) * 16 + i < K) ? a[s][t] : 0.f;
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bq, acc[t], 0, 0, 0);
      }
    }
  }
  const int m = m0 + i;                                        // D[row = 4 g + r (k)][col = i (m)]
  if (m >= M) return;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = (kt0 + t) * 16 + 4 * g + r;
      if (k < K) unsafeAtomicAdd(dx + (size_t)m * K + k, acc[t][r]);
    }
}

// Batched form: one launch for every deferred linear weight gradient of a model.
// table[e] = {dy, x, dW, db, M, N, K, lddy, ldw}; blockIdx.y = entry, blockIdx.x = (k tile, n tile-of-64) of that entry.
struct LinWg { const float* dy; const float* x; float* dW; float* db; int M, N, K, lddy, ldw, pad0, pad1, pad2; };

// The records travel as a by-value kernel argument (<= 56 x 64 B, under the 4 KB kernarg limit): no device table, no
// host-to-device copy node in the captured step (each such memcpy node cost a 20-75 us bubble in the replay).
constexpr int LINWG_MAX = 56;
struct LinWgArgs { LinWg rec[LINWG_MAX]; };

__device__ __forceinline__ void linear_wgrad_entry(const LinWg& e);

__global__ __launch_bounds__(256) void linear_wgrad_batched_args_kernel(const LinWgArgs tab) {
  linear_wgrad_entry(tab.rec[blockIdx.y]);         // uniform index: scalar loads from the kernarg segment
}

__global__ __launch_bounds__(256) void linear_wgrad_batched_kernel(const LinWg* __restrict__ table) {
  linear_wgrad_entry(table[blockIdx.y]);
}

__device__ __forceinline__ void linear_wgrad_entry(const LinWg& e) {
  const int ktiles = (e.K + 63) / 64, ntiles = (e.N + 63) / 64;
  if ((int)blockIdx.x >= ktiles * ntiles) return;
  const int kt = blockIdx.x % ktiles, nt = blockIdx.x / ktiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int n0 = (nt * 4 + wave) * 16, k0 = kt * 64;
  if (n0 >= e.N) return;
  const bool nv = n0 + i < e.N;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  for (int st0 = 0; st0 * 4 < e.M; st0 += 4) {          // 4 MFMA k-steps (16 token rows) per batch of loads
    float av[4], bv[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                          // raw loads first (clamped addresses) ...
      const int m = (st0 + u) * 4 + g;
      const bool mv = m < e.M;
      av[u] = e.dy[(size_t)(mv ? m : 0) * e.lddy + (nv ? n0 + i : 0)];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k = k0 + t * 16 + i;
        const bool kv = mv && k < e.K;
        bv[u][t] = e.x[(size_t)(kv ? m : 0) * e.K + (kv ? k : 0)];
      }
    }
    __builtin_amdgcn_sched_barrier(0);                     // ... so that the 20 loads are in flight together (see skinny_gemm_kernel)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = (st0 + u) * 4 + g;
      const bool mv = m < e.M;
      const float a = (mv && nv) ? av[u] : 0.f;
      bsum += a;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float b = (mv && k0 + t * 16 + i < e.K) ? bv[u][t] : 0.f;
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
      }
    }
  }
  float old[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int k = k0 + t * 16 + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * g + r;
      old[t][r] = (k < e.K && n < e.N) ? e.dW[(size_t)n * e.ldw + k] : 0.f;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int k = k0 + t * 16 + i;
    if (k >= e.K) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * g + r;
      if (n < e.N) e.dW[(size_t)n * e.ldw + k] = old[t][r] + acc[t][r];
    }
  }
  if (e.db && kt == 0) {
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (g == 0 && nv) e.db[n0 + i] += bsum;
  }
}
}  // namespace

extern "C" int ast_linear_wgrad_batched_host(const void* host_table, int count, int max_tiles, void* stream) {
  if (!host_table || count <= 0 || max_tiles <= 0) AST_FAIL("ast_linear_wgrad_batched_host: bad args");
  const LinWg* recs = (const LinWg*)host_table;
  for (int c0 = 0; c0 < count; c0 += LINWG_MAX) {
    const int n = std::min(LINWG_MAX, count - c0);
    LinWgArgs args;
    memset(&args, 0, sizeof(args));
    memcpy(args.rec, recs + c0, sizeof(LinWg) * n);
    hipLaunchKernelGGL(linear_wgrad_batched_args_kernel, dim3(max_tiles, n), dim3(256), 0, (hipStream_t)stream, args);
  }
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_linear_wgrad_batched(const void* table, int count, int max_tiles, void* stream) {
  if (!table || count <= 0 || max_tiles <= 0) AST_FAIL("ast_linear_wgrad_batched: bad args");
  hipLaunchKernelGGL(linear_wgrad_batched_kernel, dim3(max_tiles, count), dim3(256), 0, (hipStream_t)stream, (const LinWg*)table);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_skinny_gemm_ex(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int ldw, int ldy,
                                  int relu, const float* mul_mask, float* drop_mask, float p, uint64_t seed, const int64_t* d_offset,
                                  void* stream) {
  if (!x || !w || !y || M < 1 || M > 64 || N < 1 || K < 4 || (K & 3) || (ldw & 3)) AST_FAIL("ast_skinny_gemm: bad args M=%d N=%d K=%d", M, N, K);
  if ((((uintptr_t)x) | ((uintptr_t)w)) & 15) AST_FAIL("ast_skinny_gemm: operands must be 16-byte aligned");
  if (drop_mask && (p <= 0.f || p >= 1.f)) AST_FAIL("ast_skinny_gemm: dropout epilogue needs 0 < p < 1");
  const int steps = (K + 63) / 64;                    // 16-wide K steps per wave (4 waves split K)
  dim3 grid((N + 15) / 16, (M + 15) / 16);
  hipStream_t s = (hipStream_t)stream;
#define AST_SK(NS_) hipLaunchKernelGGL(skinny_gemm_kernel<NS_>, grid, dim3(256), 0, s, x, w, bias, y, M, N, K, K, ldw, ldy, relu, mul_mask, \
                                       drop_mask, p, seed, d_offset)
  if (steps <= 2) AST_SK(2);
  else if (steps <= 4) AST_SK(4);
  else if (steps <= 8) AST_SK(8);
  else if (steps <= 16) {                             // eight waves, 8 steps each
    hipLaunchKernelGGL((skinny_gemm_kernel<8, 8>), grid, dim3(512), 0, s, x, w, bias, y, M, N, K, K, ldw, ldy, relu, mul_mask, drop_mask, p,
                       seed, d_offset);
  }
  else AST_FAIL("ast_skinny_gemm: K=%d too large for the token path (<= 1024)", K);
#undef AST_SK
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_skinny_gemm(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int ldw, int ldy,
                               int relu, void* stream) {
  return ast_skinny_gemm_ex(x, w, bias, y, M, N, K, ldw, ldy, relu, nullptr, nullptr, 0.f, 0, nullptr, stream);
}

extern "C" int ast_linear_wgrad(const float* dy, const float* x, float* dW, float* db, int M, int N, int K, int lddy, int ldw,
                                void* stream) {
  if (!dy || !x || !dW || M < 1 || N < 1 || K < 1) AST_FAIL("ast_linear_wgrad: bad args");
  if (M > 64) AST_FAIL("ast_linear_wgrad: M=%d > 64 token rows", M);
  dim3 grid((K + 63) / 64, (N + 63) / 64);
  hipStream_t s = (hipStream_t)stream;
  if (M <= 16) hipLaunchKernelGGL(linear_wgrad_kernel<4>, grid, dim3(256), 0, s, dy, x, dW, db, M, N, K, lddy, ldw);
  else if (M <= 32) hipLaunchKernelGGL(linear_wgrad_kernel<8>, grid, dim3(256), 0, s, dy, x, dW, db, M, N, K, lddy, ldw);
  else hipLaunchKernelGGL(linear_wgrad_kernel<16>, grid, dim3(256), 0, s, dy, x, dW, db, M, N, K, lddy, ldw);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_bigk_gemm(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int ldy, void* stream) {
  if (!x || !w || !y || M < 1 || M > 64 || N < 1 || K < 2 || (K & 1)) AST_FAIL("ast_bigk_gemm: bad args M=%d N=%d K=%d (K must be even)", M, N, K);
  if ((((uintptr_t)x) | ((uintptr_t)w)) & 7) AST_FAIL("ast_bigk_gemm: operands must be 8-byte aligned");
  dim3 grid((N + 15) / 16, (M + 15) / 16, (K + 1023) / 1024);
  AST_HIP(hipMemsetAsync(y, 0, sizeof(float) * (size_t)M * ldy, (hipStream_t)stream));
  hipLaunchKernelGGL(bigk_gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, M, N, K, ldy);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_bign_dgrad(const float* dy, const float* w, float* dx, int M, int N, int K, int lddy, void* stream) {
  if (!dy || !w || !dx || M < 1 || M > 64 || N < 1 || K < 1 || K > 256) AST_FAIL("ast_bign_dgrad: bad args M=%d N=%d K=%d (K <= 256)", M, N, K);
  dim3 grid((N + 511) / 512, (M + 15) / 16);
  AST_HIP(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)M * K, (hipStream_t)stream));
  hipLaunchKernelGGL(bign_dgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, w, dx, M, N, K, lddy);
  AST_CHECK_LAUNCH();
  return 0;
}
