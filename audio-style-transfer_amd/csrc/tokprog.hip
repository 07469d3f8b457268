// Token programs: a whole transformer layer (or several) in ONE launch.
//
// The token path of the step (style_encoder.py:181-191, content_encoder.py, new_decoder.py:111-119,252-266) is ~330
// launches of 2-9 us kernels on <= 64 token rows, every one a dependent node of the replayed graph: measured
// (tools/micro/graph_node_cost.py) a dependent node costs 1.7 us of dispatch on top of a ~3 us launch-latency-bound body,
// and forked streams do not overlap such nodes.  Here the host hands the kernel a PROGRAM -- a list of ops (GEMM with
// fused epilogues, attention core forward / backward, residual + dropout + LayerNorm forward / backward) -- and G
// workgroups walk it together, separated by a grid barrier (0.9 us for G = 16, tools/micro/gridbar.hip) instead of a
// kernel boundary.
//
// Placement: the G workgroups are the ones with blockIdx % 8 == xcd of an 8 G grid -- workgroups are dealt round-robin
// over the 8 XCDs, so they share one L2 -- and the others exit at once.  This is a speed matter only: every value one
// op writes and a later op reads goes through agent-scope relaxed atomics (global_load / global_store with sc1:
// coherent at the device level, no L1 hit, no fence, no cache write-back), which is correct wherever the workgroups
// land.  (Alternatives measured: __threadfence barriers 3.6-11 us, buffer_inv sc1 4-16 us, both because they write back
// or invalidate an L2 that concurrent convolution kernels are filling.)  Weights, biases and LayerNorm parameters are
// read-only during a launch and use plain loads.
//
// Barrier: one monotonically increasing counter per launch (op k waits for G * (k + 1) arrivals); the last workgroup
// to finish resets it, so replays of the captured graph start from zero again.  A wait that exceeds ~2^21 polls sets
// *status and the workgroup stops waiting for the rest of the launch (results are then garbage, but the grid drains;
// the host checks the flag after its warm-up steps).
#include "ast_common.h"
#include "../../include/ast_hip.h"

namespace {

constexpr int TOK_MAXOPS = 26;             // 16 + 26 * 152 bytes of kernel arguments (< 4 KB with the three pointers)
struct TokProgram { int nops, G, xcd, pad; ast_tok_op_t op[TOK_MAXOPS]; };

// ---- device-coherent accesses for values exchanged between workgroups inside a launch --------------------------------
__device__ __forceinline__ float ldc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ f32x2 ldc2(const float* p) {
  const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_bit_cast(f32x2, u);
}
__device__ __forceinline__ void stc2(float* p, f32x2 v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x4 ldc4(const float* p) {
  const f32x2 a = ldc2(p), b = ldc2(p + 2);
  return f32x4{a.x, a.y, b.x, b.y};
}
__device__ __forceinline__ void stc4(float* p, f32x4 v) { stc2(p, f32x2{v.x, v.y}); stc2(p + 2, f32x2{v.z, v.w}); }

struct Bar { unsigned* ctr; int* status; int G; unsigned phase; bool dead; };

__device__ __forceinline__ void grid_barrier(Bar& b) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's stores have been issued and acknowledged (s_waitcnt)
  __syncthreads();
  ++b.phase;
  if (threadIdx.x == 0 && !b.dead) {
    __hip_atomic_fetch_add(b.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = b.phase * (unsigned)b.G;
    int polls = 0;
    while (__hip_atomic_load(b.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++polls > (1 << 21)) { __hip_atomic_store(b.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); b.dead = true; break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- GEMM on <= 64 token rows: y[m][n] = epi(sum_k x[m][k] w[n][k]) ------------------------------------------------------
// v_mfma_f32_16x16x4_f32 (exact f32), the arrangement of skinny_gemm_kernel (skinny.hip): the weight tile is the A
// operand, lane (i = l & 15, g = l >> 4) loads 16 B of weight row n0 + i and of token row mt * 16 + i at
// k = kc + 16 s + 4 g, and ends up with n = n0 + 4 g .. + 3 of token row i.  A task is one 16-wide n tile for ALL token
// rows (MT m tiles share the weight fragment).  Few n tiles (<= G): a workgroup takes the tile and its four waves split K
// (partials summed through LDS); many: every wave takes its own tile with the whole K.
// epilogue: + bias, ReLU, dropout draw (the COMBINED ReLU & dropout mask is stored for the backward pass), * mask,
// + addend (the residual gradient a data gradient is merged with), in the order of skinny_gemm_kernel.
template <int MT>
__device__ __forceinline__ void tok_gemm(const ast_tok_op_t& op, const int wg, const int G, f32x4 (*part)[4][64], const int64_t* d_offset) {
  constexpr int CS = MT <= 2 ? 8 : 4;                 // k steps (16 wide) whose loads are in flight together
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  const int M = op.i[0], N = op.i[1], K = op.i[2], ldx = op.i[3], ldw = op.i[4], ldy = op.i[5];
  const float* x = op.in[0]; const float* w = op.in[1]; const float* bias = op.in[2]; const float* mul_mask = op.in[3];
  const float* addend = op.in[4];
  float* y = op.out[0]; float* drop_mask = op.out[1];
  const bool relu = op.flags & 1;
  const int NT = (N + 15) >> 4;
  const bool split = NT <= G;
  const int kq = split ? ((((K + 3) >> 2) + 15) & ~15) : K;        // k per wave
  const int kbeg = split ? wave * kq : 0, kend = min(K, kbeg + kq);
  const int unit = split ? wg : wg * 4 + wave, nunits = split ? G : G * 4;
  const uint64_t dbase = drop_mask ? mix64(op.seed ^ mix64((uint64_t)(d_offset ? *d_offset : 0))) : 0;
  const float keep = 1.f / (1.f - op.p);
  for (int t = unit; t < NT; t += nunits) {
    const int n0 = t << 4;
    const bool nv = n0 + i < N;
    const float* wr = w + (size_t)(nv ? n0 + i : 0) * ldw;
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kc = kbeg; kc < kend; kc += CS * 16) {
      f32x4 wl[CS], xl[MT][CS];
#pragma unroll
      for (int s = 0; s < CS; ++s) {
        const int k = kc + s * 16 + 4 * g;
        const bool kv = k < kend;
        wl[s] = *reinterpret_cast<const f32x4*>(wr + (kv ? k : 0));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int m = mt * 16 + i;
          xl[mt][s] = ldc4(x + (size_t)(m < M ? m : 0) * ldx + (kv ? k : 0));
        }
      }
      __builtin_amdgcn_sched_barrier(0);              // all loads of the chunk in flight before the first MFMA (see skinny.hip)
#pragma unroll
      for (int s = 0; s < CS; ++s) {
        const bool kv = kc + s * 16 + 4 * g < kend;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 a = (kv && nv) ? wl[s] : z;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 b = (kv && mt * 16 + i < M) ? xl[mt][s] : z;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc[mt], 0, 0, 0);
        }
      }
    }
    if (split) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) part[wave][mt][lane] = acc[mt];
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          f32x4 r = part[0][mt][lane];
#pragma unroll
          for (int q = 1; q < 4; ++q) { const f32x4 u = part[q][mt][lane]; r += u; }
          acc[mt] = r;
        }
      }
    }
    if (!split || wave == 0) {
      const int nb = n0 + 4 * g;                      // lane: n = nb .. nb + 3 of token row mt * 16 + i
      if (nb < N) {                                   // N % 4 == 0 (host)
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 b4 = bias ? *reinterpret_cast<const f32x4*>(bias + nb) : z;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int m = mt * 16 + i;
          if (m >= M) continue;
          const size_t o = (size_t)m * ldy + nb;
          f32x4 v = acc[mt] + b4;
          if (relu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
          }
          if (drop_mask) {
            f32x4 km;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float kk = dropout_keep(dbase, o + q, op.p, keep);
              km[q] = (relu && v[q] <= 0.f) ? 0.f : kk;
              v[q] *= kk;
            }
            stc4(drop_mask + o, km);
          }
          if (mul_mask) v *= ldc4(mul_mask + o);
          if (addend) v += ldc4(addend + o);
          stc4(y + o, v);
        }
      }
    }
    if (split) __syncthreads();                       // `part` is free for the next tile
  }
}

// ---- attention core for <= 16 tokens: one wave per (batch, head), lane = feature (misc.hip: attn_fwd_kernel / attn_bwd_kernel)
constexpr int TOK_MAXL = 16;
__device__ __forceinline__ void tok_attn_fwd(const ast_tok_op_t& op, const int wg, const int G, const int64_t* d_offset) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int B = op.i[0], H = op.i[1], Lq = op.i[2], Lk = op.i[3], dh = op.i[4], ldq = op.i[5], ldk = op.i[6], ldo = op.i[7];
  const bool causal = op.flags & 4;
  const float* q = op.in[0]; const float* k = op.in[1]; const float* v = op.in[2];
  float* o = op.out[0]; float* probs = op.out[1];
  const float pdrop = op.p;
  const uint64_t dbase = pdrop > 0.f ? mix64(op.seed ^ mix64((uint64_t)(d_offset ? *d_offset : 0))) : 0;
  const float dkeep = 1.f / (1.f - pdrop);
  const float scale = rsqrtf((float)dh);
  for (int t = wg * 4 + wave; t < B * H; t += G * 4) {
    const int b = t / H, h = t % H;
    float kv[TOK_MAXL], vv[TOK_MAXL];
#pragma unroll
    for (int j = 0; j < TOK_MAXL; ++j) {
      kv[j] = (j < Lk && lane < dh) ? ldc1(k + ((size_t)b * Lk + j) * ldk + h * dh + lane) : 0.f;
      vv[j] = (j < Lk && lane < dh) ? ldc1(v + ((size_t)b * Lk + j) * ldk + h * dh + lane) : 0.f;
    }
    for (int iq = 0; iq < Lq; ++iq) {
      const float qi = lane < dh ? ldc1(q + ((size_t)b * Lq + iq) * ldq + h * dh + lane) * scale : 0.f;
      float s[TOK_MAXL];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < TOK_MAXL; ++j) {
        s[j] = -INFINITY;
        if (j < Lk) {
          s[j] = wave_sum(qi * kv[j]);
          if (causal && j > iq) s[j] = -INFINITY;
          mx = fmaxf(mx, s[j]);
        }
      }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j < TOK_MAXL; ++j) { s[j] = j < Lk ? __expf(s[j] - mx) : 0.f; den += s[j]; }
      float acc = 0.f;
      const size_t pbase = (((size_t)b * H + h) * Lq + iq) * Lk;
#pragma unroll
      for (int j = 0; j < TOK_MAXL; ++j) {
        if (j < Lk) {
          const float p = s[j] / den;
          if (lane == 0) stc1(probs + pbase + j, p);
          acc += (pdrop > 0.f ? p * dropout_keep(dbase, pbase + j, pdrop, dkeep) : p) * vv[j];
        }
      }
      if (lane < dh) stc1(o + ((size_t)b * Lq + iq) * ldo + h * dh + lane, acc);
    }
  }
}

__device__ __forceinline__ void tok_attn_bwd(const ast_tok_op_t& op, const int wg, const int G, const int64_t* d_offset) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int B = op.i[0], H = op.i[1], Lq = op.i[2], Lk = op.i[3], dh = op.i[4], ldq = op.i[5], ldk = op.i[6], ldo = op.i[7];
  const float* dout = op.in[0]; const float* q = op.in[1]; const float* k = op.in[2]; const float* v = op.in[3];
  const float* probs = op.in[4];
  float* dq = op.out[0]; float* dk = op.out[1]; float* dv = op.out[2];
  const float pdrop = op.p;
  const uint64_t dbase = pdrop > 0.f ? mix64(op.seed ^ mix64((uint64_t)(d_offset ? *d_offset : 0))) : 0;
  const float dkeep = 1.f / (1.f - pdrop);
  const float scale = rsqrtf((float)dh);
  for (int t = wg * 4 + wave; t < B * H; t += G * 4) {
    const int b = t / H, h = t % H;
    float kv[TOK_MAXL], vv[TOK_MAXL], dkv[TOK_MAXL], dvv[TOK_MAXL];
#pragma unroll
    for (int j = 0; j < TOK_MAXL; ++j) {
      kv[j] = (j < Lk && lane < dh) ? ldc1(k + ((size_t)b * Lk + j) * ldk + h * dh + lane) : 0.f;
      vv[j] = (j < Lk && lane < dh) ? ldc1(v + ((size_t)b * Lk + j) * ldk + h * dh + lane) : 0.f;
      dkv[j] = 0.f; dvv[j] = 0.f;
    }
    for (int iq = 0; iq < Lq; ++iq) {
      const float qi = lane < dh ? ldc1(q + ((size_t)b * Lq + iq) * ldq + h * dh + lane) : 0.f;
      const float doi = lane < dh ? ldc1(dout + ((size_t)b * Lq + iq) * ldo + h * dh + lane) : 0.f;
      const size_t pbase = (((size_t)b * H + h) * Lq + iq) * Lk;
      float dp[TOK_MAXL], p[TOK_MAXL];
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < TOK_MAXL; ++j) {
        p[j] = 0.f; dp[j] = 0.f;
        if (j < Lk) {
          p[j] = ldc1(probs + pbase + j);
          const float m = pdrop > 0.f ? dropout_keep(dbase, pbase + j, pdrop, dkeep) : 1.f;
          dvv[j] += p[j] * m * doi;
          dp[j] = wave_sum(doi * vv[j]) * m;
          dot += dp[j] * p[j];
        }
      }
      float dqi = 0.f;
#pragma unroll
      for (int j = 0; j < TOK_MAXL; ++j) {
        if (j < Lk) {
          const float ds = p[j] * (dp[j] - dot) * scale;
          dqi += ds * kv[j];
          dkv[j] += ds * qi;
        }
      }
      if (lane < dh) stc1(dq + ((size_t)b * Lq + iq) * ldq + h * dh + lane, dqi);
    }
    if (lane < dh) {
#pragma unroll
      for (int j = 0; j < TOK_MAXL; ++j)
        if (j < Lk) {
          stc1(dk + ((size_t)b * Lk + j) * ldk + h * dh + lane, dkv[j]);
          stc1(dv + ((size_t)b * Lk + j) * ldk + h * dh + lane, dvv[j]);
        }
    }
  }
}

// ---- residual + dropout + LayerNorm on 256-wide token rows, one wave per row (norm.hip: add_drop_ln_fwd256 / bwd256) -----
//   s = x + dropout(sub)   (x may be null);   y = LayerNorm(s)   (gamma null: s only)
__device__ __forceinline__ void tok_adln_fwd(const ast_tok_op_t& op, const int wg, const int G, const int64_t* d_offset) {
  constexpr int D = 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows = op.i[0];
  const float* x = op.in[0]; const float* sub = op.in[1]; const float* gamma = op.in[2]; const float* beta = op.in[3];
  float* mask = op.out[0]; float* s_out = op.out[1]; float* y = op.out[2]; float* mean = op.out[3]; float* rstd = op.out[4];
  const float p = op.p, eps = op.eps;
  const uint64_t base = p > 0.f ? mix64(op.seed ^ mix64((uint64_t)(d_offset ? *d_offset : 0))) : 0;
  const float keep = 1.f / (1.f - p);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 g4 = gamma ? *reinterpret_cast<const f32x4*>(gamma + lane * 4) : z;
  const f32x4 b4 = gamma ? *reinterpret_cast<const f32x4*>(beta + lane * 4) : z;
  for (int row = wg * 4 + wave; row < rows; row += G * 4) {
    const size_t o = (size_t)row * D + lane * 4;
    f32x4 sv = ldc4(sub + o);
    const f32x4 xv = x ? ldc4(x + o) : z;
    if (p > 0.f) {
      f32x4 m4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { m4[e] = dropout_keep(base, o + e, p, keep); sv[e] *= m4[e]; }
      stc4(mask + o, m4);
    }
    const f32x4 s4 = x ? sv + xv : sv;
    if (s_out) stc4(s_out + o, s4);
    if (!gamma) continue;
    const float m = wave_sum(s4[0] + s4[1] + s4[2] + s4[3]) / D;
    float qq = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = s4[e] - m; qq += d * d; }
    const float r = rsqrtf(wave_sum(qq) / D + eps);
    f32x4 y4;
#pragma unroll
    for (int e = 0; e < 4; ++e) y4[e] = (s4[e] - m) * r * g4[e] + b4[e];
    stc4(y + o, y4);
    if (lane == 0) { stc1(mean + row, m); stc1(rstd + row, r); }
  }
}

//   ds = ds_ext + LayerNorm_bwd(dy; s)   (either term may be absent);   dx = ds;   dsub = ds * mask;   dgamma, dbeta += ...
__device__ __forceinline__ void tok_adln_bwd(const ast_tok_op_t& op, const int wg, const int G) {
  constexpr int D = 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows = op.i[0];
  const float* dy = op.in[0]; const float* ds_ext = op.in[1]; const float* s = op.in[2]; const float* gamma = op.in[3];
  const float* mean = op.in[4]; const float* rstd = op.in[5]; const float* mask = op.in[6];
  float* dx = op.out[0]; float* dsub = op.out[1]; float* dgamma = op.out[2]; float* dbeta = op.out[3];
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 g4 = dy ? *reinterpret_cast<const f32x4*>(gamma + lane * 4) : z;
  f32x4 accg = z, accb = z;                           // this wave's rows: one atomic per element at the end
  for (int row = wg * 4 + wave; row < rows; row += G * 4) {
    const size_t o = (size_t)row * D + lane * 4;
    const f32x4 dy4 = dy ? ldc4(dy + o) : z;
    const f32x4 s4 = dy ? ldc4(s + o) : z;
    const f32x4 e4 = ds_ext ? ldc4(ds_ext + o) : z;
    const f32x4 k4 = mask ? ldc4(mask + o) : f32x4{1.f, 1.f, 1.f, 1.f};
    const float m = dy ? ldc1(mean + row) : 0.f, r = dy ? ldc1(rstd + row) : 0.f;
    f32x4 d4 = e4, xh4 = z;
    if (dy) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { xh4[e] = (s4[e] - m) * r; const float gg = dy4[e] * g4[e]; a += gg; b += gg * xh4[e]; }
      a = wave_sum(a) / D; b = wave_sum(b) / D;
#pragma unroll
      for (int e = 0; e < 4; ++e) d4[e] += r * (dy4[e] * g4[e] - a - xh4[e] * b);
      accg += dy4 * xh4; accb += dy4;
    }
    if (dx) stc4(dx + o, d4);
    if (dsub) stc4(dsub + o, d4 * k4);
  }
  if (dy && dgamma) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsafeAtomicAdd(dgamma + lane * 4 + e, accg[e]); unsafeAtomicAdd(dbeta + lane * 4 + e, accb[e]); }
  }
}

__global__ __launch_bounds__(256) void tok_program_kernel(const TokProgram prog, unsigned* __restrict__ sync, int* __restrict__ status,
                                                          const int64_t* __restrict__ d_offset) {
  if ((int)(blockIdx.x & 7) != prog.xcd) return;
  __shared__ f32x4 part[4][4][64];
  const int wg = blockIdx.x >> 3, G = prog.G;
  Bar bar{sync, status, G, 0u, false};
  for (int k = 0; k < prog.nops; ++k) {
    const ast_tok_op_t& op = prog.op[k];
    switch (op.type) {
      case AST_TOK_GEMM: {
        const int M = op.i[0];
        if (M <= 16) tok_gemm<1>(op, wg, G, part, d_offset);
        else if (M <= 32) tok_gemm<2>(op, wg, G, part, d_offset);
        else tok_gemm<4>(op, wg, G, part, d_offset);
      } break;
      case AST_TOK_ATTN_FWD: tok_attn_fwd(op, wg, G, d_offset); break;
      case AST_TOK_ATTN_BWD: tok_attn_bwd(op, wg, G, d_offset); break;
      case AST_TOK_ADLN_FWD: tok_adln_fwd(op, wg, G, d_offset); break;
      case AST_TOK_ADLN_BWD: tok_adln_bwd(op, wg, G); break;
      default: break;
    }
    if (!(op.flags & AST_TOK_NO_BARRIER) && k + 1 < prog.nops) grid_barrier(bar);
  }
  // leave the counters at zero for the next launch (a replay of the same graph node): the last workgroup to get here
  // knows that every other one has left its last barrier
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (unsigned)G - 1) {
      __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

extern "C" int ast_tok_max_ops(void) { return TOK_MAXOPS; }

extern "C" int ast_tok_program(const ast_tok_op_t* ops, int nops, int G, int xcd, void* sync, int* status, const int64_t* d_offset,
                               void* stream) {
  if (!ops || nops <= 0 || nops > TOK_MAXOPS) AST_FAIL("ast_tok_program: 1..%d ops per launch (got %d)", TOK_MAXOPS, nops);
  if (G <= 0 || G > 32 || xcd < 0 || xcd > 7 || !sync || !status) AST_FAIL("ast_tok_program: bad G / xcd / sync / status");
  TokProgram prog;
  prog.nops = nops; prog.G = G; prog.xcd = xcd; prog.pad = 0;
  for (int k = 0; k < nops; ++k) {
    const ast_tok_op_t& op = ops[k];
    switch (op.type) {
      case AST_TOK_GEMM:
        if (op.i[0] <= 0 || op.i[0] > 64 || op.i[1] % 16 || op.i[2] % 4 || op.i[3] % 4 || op.i[4] % 4 || op.i[5] % 4 || !op.in[0] || !op.in[1] || !op.out[0])
          AST_FAIL("ast_tok_program: op %d: GEMM needs 1..64 rows, N %% 16 == 0, K and the leading dimensions %% 4 == 0 (M %d N %d K %d)", k, op.i[0], op.i[1], op.i[2]);
        if (op.out[1] && !(op.p >= 0.f && op.p < 1.f)) AST_FAIL("ast_tok_program: op %d: mask output needs 0 <= p < 1", k);
        break;
      case AST_TOK_ATTN_FWD: case AST_TOK_ATTN_BWD:
        if (op.i[2] <= 0 || op.i[3] <= 0 || op.i[2] > TOK_MAXL || op.i[3] > TOK_MAXL || op.i[4] > 64 || op.i[4] <= 0)
          AST_FAIL("ast_tok_program: op %d: attention core needs 1..%d tokens and head_dim <= 64", k, TOK_MAXL);
        break;
      case AST_TOK_ADLN_FWD: case AST_TOK_ADLN_BWD:
        if (op.i[0] <= 0 || op.i[1] != 256) AST_FAIL("ast_tok_program: op %d: residual + LayerNorm ops are 256 wide", k);
        break;
      default: AST_FAIL("ast_tok_program: op %d: unknown type %d", k, op.type);
    }
    prog.op[k] = op;
  }
  hipLaunchKernelGGL(tok_program_kernel, dim3(8 * G), dim3(256), 0, (hipStream_t)stream, prog, (unsigned*)sync, status, d_offset);
  AST_CHECK_LAUNCH();
  return 0;
}
