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
// Placement and coherence: the G workgroups are the first G of a 16 G grid that land on XCD `xcd` (HW_REG_XCC_ID + a
// ticket), so that they share one L2 and one set of CUs.  Values one op writes and a later op reads are stored and loaded
// at agent scope (sc1): coherent wherever the workgroups run, at the price of a slow path -- sc1 loads are served from
// beyond the L2.  What was measured on the way (tools/micro/gridbar.hip, gridbar2.hip, tools/tok_op_cost.py):
//   * `sc0` loads do NOT bypass the CU's L1 in this mode: correct on large buffers (the L1 thrashes), stale values on
//     the small ones (<= 8 token rows failed parity);
//   * buffer_inv sc0 drops nothing, buffer_inv sc1 / __threadfence cost 4-16 us per barrier next to convolution kernels
//     (they write back / invalidate the whole L2);
//   * L2-level atomics (no sc1) are not seen by pollers on other CUs; agent-scope ones make a 0.9 (G = 16) - 1.7 us
//     (G = 32) barrier;
//   * blockIdx % 8 is the XCD only for a dispatch that starts the round-robin at 0 -- not under graph replay.
// Weights, biases and LayerNorm parameters are read-only during a launch and use plain loads.
//
// Barrier: one monotonically increasing counter per launch (op k waits for G * (k + 1) arrivals), zeroed again by the last
// workgroup of the grid to exit, so replays of the captured graph start from zero.  A wait that exceeds ~2^21 polls sets
// *status and the workgroup stops waiting for the rest of the launch (results are then garbage, but the grid drains;
// the host checks the flag after its warm-up steps).
#include "ast_common.h"
#include "../../include/ast_hip.h"

namespace {

constexpr int TOK_MAXOPS = 26;             // 16 + 26 * 152 bytes of kernel arguments (< 4 KB with the three pointers)
struct TokProgram { int nops, G, xcd, pad; ast_tok_op_t op[TOK_MAXOPS]; };

// ---- agent-scope (sc1) loads for values exchanged between workgroups inside a launch ---------------------------------------
// The loads are inline asm (16-byte agent-scope loads have no builtin), ISSUED without a wait so that a batch is in
// flight together; TL<COH>::wait() is the s_waitcnt and ties the loaded registers to it ("+v"), so no use can be scheduled
// ahead of it.  The compiler's own vmcnt counting does not see these loads: it can only over-wait (memory ops return in
// order), never under-wait.
// COH = true: inside a persistent program (agent-scope, see above).  COH = false: one launch per op -- the kernel boundary
// orders everything and these are plain loads and stores.
template <bool COH> struct TL {
  static __device__ __forceinline__ void issue(f32x4& v, const float* p) {
    if constexpr (COH) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v) : "v"(p) : "memory");
    else v = *reinterpret_cast<const f32x4*>(p);
  }
  static __device__ __forceinline__ void issue(float& v, const float* p) {
    if constexpr (COH) asm volatile("global_load_dword %0, %1, off sc1" : "=&v"(v) : "v"(p) : "memory");
    else v = *p;
  }
  template <typename T> static __device__ __forceinline__ void wait(T& a) { if constexpr (COH) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a) : : "memory"); }
  template <typename T> static __device__ __forceinline__ void wait(T& a, T& b) { if constexpr (COH) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : : "memory"); }
  template <typename T> static __device__ __forceinline__ void wait(T& a, T& b, T& c, T& d) {
    if constexpr (COH) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
  }
  template <typename T, int N> static __device__ __forceinline__ void wait_all(T (&v)[N]) {
    static_assert(N % 4 == 0, "batches of four");
#pragma unroll
    for (int k = 0; k < N; k += 4) wait(v[k], v[k + 1], v[k + 2], v[k + 3]);
  }
  // Stores are agent-scope (sc1, acknowledged from beyond the L2): with plain stores s_waitcnt vmcnt(0) returned before
  // the data was visible to the other workgroups' sc1 loads.
  static __device__ __forceinline__ void st1(float* p, float v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
  }
  static __device__ __forceinline__ void st4(float* p, f32x4 v) {
    if constexpr (COH) {
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, f32x2{v.x, v.y}), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(p + 2), __builtin_bit_cast(unsigned long long, f32x2{v.z, v.w}), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      *reinterpret_cast<f32x4*>(p) = v;
    }
  }
};

struct Bar { unsigned* ctr; int* status; int G; unsigned phase; bool dead; };

// The counter is an agent-scope atomic (performed beyond the L2).  L2-level arrivals (global_atomic_add without sc1,
// polled with sc0 loads) never became visible to the pollers of other CUs (tools/micro/gridbar2.hip): not an option.
__device__ __forceinline__ void grid_barrier(Bar& b) {
  // every store of this wave has been acknowledged by the L2 before the arrival is published (a workgroup-scope fence
  // alone does not wait for global stores: the waves of a workgroup share their L1)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  ++b.phase;
  if (threadIdx.x == 0 && !b.dead) {
    __hip_atomic_fetch_add(b.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = b.phase * (unsigned)b.G;
    int polls = 0;
    while (__hip_atomic_load(b.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++polls > (1 << 21)) { __hip_atomic_store(b.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); b.dead = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- GEMM on <= 64 token rows: y[m][n] = epi(sum_k x[m][k] w[n][k]) ------------------------------------------------------
// v_mfma_f32_16x16x4_f32 (exact f32).  A task is one 16-wide n tile for ALL token rows (MT m tiles); the four waves of the
// workgroup split K, their partial tiles are summed through LDS and wave 0 runs the epilogue.
// Operands go through LDS in 256-float K chunks, loaded as WHOLE ROWS (a wave-instruction = one contiguous KB): the first
// version loaded MFMA fragments straight from global memory (16 rows x 64 B per instruction, the arrangement of
// skinny_gemm_kernel), which the texture path serves at 16 B/clk/CU -- with only G CUs at work that was 2.6-3.8 us of an
// 8.5 us op, the rest being the 4-long dependent MFMA chains (one accumulator per m tile: ~70 cycles per MFMA).  Here the
// activation chunk is staged once per op (K <= 256) or once per task and chunk, the weight tile per task, fragments come
// from LDS (ds_read_b128, row pitch 1040 B), and every k sub-step e has its own accumulator (4 MT independent chains).
// epilogue: + bias, ReLU, dropout draw (the COMBINED ReLU & dropout mask is stored for the backward pass), * mask,
// + addend (the residual gradient a data gradient is merged with), in the order of skinny_gemm_kernel.
constexpr int TOK_KC = 256;                      // K chunk (floats)
constexpr int TOK_PITCH = TOK_KC * 4 + 16;        // LDS row pitch in bytes
__host__ __device__ constexpr int tok_lds_bytes(int MT) { return (MT * 16 + 16) * TOK_PITCH + 4 * MT * 64 * 16; }

template <bool COH, int MT>
__device__ __forceinline__ void tok_gemm(const ast_tok_op_t& op, const int wg, const int G, unsigned char* lds, const int64_t* d_offset) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  const int M = op.i[0], N = op.i[1], K = op.i[2], ldx = op.i[3], ldw = op.i[4], ldy = op.i[5];
  const float* x = op.in[0]; const float* w = op.in[1]; const float* bias = op.in[2]; const float* mul_mask = op.in[3];
  const float* addend = op.in[4];
  float* y = op.out[0]; float* drop_mask = op.out[1];
  const bool relu = op.flags & 1;
  unsigned char* xs = lds;                                   // [MT * 16][TOK_PITCH]
  unsigned char* ws = lds + MT * 16 * TOK_PITCH;             // [16][TOK_PITCH]
  f32x4* part = reinterpret_cast<f32x4*>(ws + 16 * TOK_PITCH);      // [4][MT][64]
  const int NT = (N + 15) >> 4;
  const int nkc = (K + TOK_KC - 1) / TOK_KC;
  const uint64_t dbase = drop_mask ? mix64(op.seed ^ mix64((uint64_t)(d_offset ? *d_offset : 0))) : 0;
  const float keep = 1.f / (1.f - op.p);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  // Loader: rows r = wave, wave + 4, ...; lane l holds floats 4 l .. 4 l + 3 of the chunk.  The loads of stage n + 1 (next K
  // chunk, or the next task's weight tile and bias) are ISSUED before the MFMAs of stage n and land in registers while
  // they run: without that every task was three dependent round trips (activations, weights, bias) around 0.4 us of math.
  f32x4 xv[MT * 4], wv[4];
  auto issue_x = [&](int kc0) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < MT * 4; ++r) {
      const int m = r * 4 + wave;
      const bool ok = m < M && kc0 + lane * 4 < K;           // (clamped address + select: the asm load writes every lane)
      TL<COH>::issue(xv[r], x + (ok ? (size_t)m * ldx + kc0 + lane * 4 : 0));
    }
  };
  auto put_x = [&](int kc0) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < MT * 4; ++r) {
      const bool ok = r * 4 + wave < M && kc0 + lane * 4 < K;
      *reinterpret_cast<f32x4*>(xs + (r * 4 + wave) * TOK_PITCH + lane * 16) = ok ? xv[r] : z;
    }
  };
  auto issue_w = [&](int n0, int kc0) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + r * 4 + wave;
      wv[r] = (n < N && kc0 + lane * 4 < K) ? *reinterpret_cast<const f32x4*>(w + (size_t)n * ldw + kc0 + lane * 4) : z;
    }
  };
  auto put_w = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(ws + (r * 4 + wave) * TOK_PITCH + lane * 16) = wv[r];
  };
  auto load_bias = [&](int n0) __attribute__((always_inline)) -> f32x4 {
    const int nb = n0 + 4 * g;
    return (bias && nb < N) ? *reinterpret_cast<const f32x4*>(bias + nb) : z;
  };
  int t = wg, c = 0;
  if (t >= NT) return;                                       // (uniform per workgroup)
  issue_x(0);
  issue_w(t << 4, 0);
  f32x4 bnext = load_bias(t << 4), bcur = z;
  TL<COH>::wait_all(xv);
  bool first = true, have = true;
  f32x4 acc[MT][4];
  while (have) {
    if (nkc > 1 || first) put_x(c * TOK_KC);
    put_w();
    if (c == 0) {
      bcur = bnext;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mt][e] = z;
    }
    __syncthreads();
    int tn = t, cn = c + 1;
    if (cn == nkc) { cn = 0; tn = t + G; }
    const bool more = tn < NT;
    if (more) {
      if (nkc > 1) issue_x(cn * TOK_KC);
      issue_w(tn << 4, cn * TOK_KC);
      if (cn == 0) bnext = load_bias(tn << 4);
    }
    const int kw = wave * (TOK_KC / 4);                      // this wave's quarter of the chunk: 4 steps of 16
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int kb = (kw + s4 * 16 + 4 * g) * 4;
      const f32x4 a = *reinterpret_cast<const f32x4*>(ws + i * TOK_PITCH + kb);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(xs + (mt * 16 + i) * TOK_PITCH + kb);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc[mt][e], 0, 0, 0);
      }
    }
    if (more && nkc > 1) TL<COH>::wait_all(xv);                    // landed during the MFMAs; complete before the registers cross the back edge
    if (c == nkc - 1) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) part[(wave * MT + mt) * 64 + lane] = (acc[mt][0] + acc[mt][1]) + (acc[mt][2] + acc[mt][3]);
      __syncthreads();                                       // partials visible; every fragment read of ws / xs is done
      if (wave == 0) {
        const int nb = (t << 4) + 4 * g;                     // lane: n = nb .. nb + 3 of token row mt * 16 + i
        if (nb < N) {                                        // N % 4 == 0 (host)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int m = mt * 16 + i;
            if (m >= M) continue;
            const size_t o = (size_t)m * ldy + nb;
            f32x4 v = (part[mt * 64 + lane] + part[(MT + mt) * 64 + lane]) + (part[(2 * MT + mt) * 64 + lane] + part[(3 * MT + mt) * 64 + lane]) + bcur;
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
              TL<COH>::st4(drop_mask + o, km);
            }
            if (mul_mask || addend) {
              f32x4 mm = f32x4{1.f, 1.f, 1.f, 1.f}, ad = z;
              if (mul_mask) TL<COH>::issue(mm, mul_mask + o);
              if (addend) TL<COH>::issue(ad, addend + o);
              TL<COH>::wait(mm, ad);
              v = v * mm + ad;
            }
            TL<COH>::st4(y + o, v);
          }
        }
      }
    } else {
      __syncthreads();                                       // the chunk's fragment reads are done before the next one is staged
    }
    t = tn; c = cn; have = more; first = false;
  }
}

// ---- attention core for <= 8 tokens: one wave per (batch, head), lane = feature (misc.hip: attn_fwd_kernel / attn_bwd_kernel)
constexpr int TOK_MAXL = 8;                      // (a 16-token variant triples the code of this kernel; the model has <= 5 / <= 8)
// ML = compile-time bound on the token counts (4 or 8): the row loops are fully unrolled over it
template <bool COH, int ML>
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
  const int ln = lane < dh ? lane : 0;
  for (int t = wg * 4 + wave; t < B * H; t += G * 4) {
    const int b = t / H, h = t % H;
    float kv[ML], vv[ML], qv[ML];
#pragma unroll
    for (int j = 0; j < ML; ++j) {            // every row of K, V and Q in flight together (clamped addresses, masked below)
      const int jk = j < Lk ? j : 0, jq = j < Lq ? j : 0;
      TL<COH>::issue(kv[j], k + ((size_t)b * Lk + jk) * ldk + h * dh + ln);
      TL<COH>::issue(vv[j], v + ((size_t)b * Lk + jk) * ldk + h * dh + ln);
      TL<COH>::issue(qv[j], q + ((size_t)b * Lq + jq) * ldq + h * dh + ln);
    }
    TL<COH>::wait_all(kv); TL<COH>::wait_all(vv); TL<COH>::wait_all(qv);
#pragma unroll
    for (int j = 0; j < ML; ++j) {
      kv[j] = (j < Lk && lane < dh) ? kv[j] : 0.f;
      vv[j] = (j < Lk && lane < dh) ? vv[j] : 0.f;
    }
#pragma unroll
    for (int iq = 0; iq < ML; ++iq) {
      if (iq >= Lq) break;
      const float qi = lane < dh ? qv[iq] * scale : 0.f;
      float s[ML];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < ML; ++j) {
        s[j] = -INFINITY;
        if (j < Lk) {
          s[j] = wave_sum(qi * kv[j]);
          if (causal && j > iq) s[j] = -INFINITY;
          mx = fmaxf(mx, s[j]);
        }
      }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j < ML; ++j) { s[j] = j < Lk ? __expf(s[j] - mx) : 0.f; den += s[j]; }
      float acc = 0.f;
      const size_t pbase = (((size_t)b * H + h) * Lq + iq) * Lk;
#pragma unroll
      for (int j = 0; j < ML; ++j) {
        if (j < Lk) {
          const float p = s[j] / den;
          if (lane == 0) TL<COH>::st1(probs + pbase + j, p);
          acc += (pdrop > 0.f ? p * dropout_keep(dbase, pbase + j, pdrop, dkeep) : p) * vv[j];
        }
      }
      if (lane < dh) TL<COH>::st1(o + ((size_t)b * Lq + iq) * ldo + h * dh + lane, acc);
    }
  }
}

template <bool COH, int ML>
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
  const int ln = lane < dh ? lane : 0;
  for (int t = wg * 4 + wave; t < B * H; t += G * 4) {
    const int b = t / H, h = t % H;
    float kv[ML], vv[ML], qv[ML], dov[ML], dkv[ML], dvv[ML];
#pragma unroll
    for (int j = 0; j < ML; ++j) {
      const int jk = j < Lk ? j : 0, jq = j < Lq ? j : 0;
      TL<COH>::issue(kv[j], k + ((size_t)b * Lk + jk) * ldk + h * dh + ln);
      TL<COH>::issue(vv[j], v + ((size_t)b * Lk + jk) * ldk + h * dh + ln);
      TL<COH>::issue(qv[j], q + ((size_t)b * Lq + jq) * ldq + h * dh + ln);
      TL<COH>::issue(dov[j], dout + ((size_t)b * Lq + jq) * ldo + h * dh + ln);
    }
    TL<COH>::wait_all(kv); TL<COH>::wait_all(vv); TL<COH>::wait_all(qv); TL<COH>::wait_all(dov);
#pragma unroll
    for (int j = 0; j < ML; ++j) {
      kv[j] = (j < Lk && lane < dh) ? kv[j] : 0.f;
      vv[j] = (j < Lk && lane < dh) ? vv[j] : 0.f;
      dkv[j] = 0.f; dvv[j] = 0.f;
    }
#pragma unroll
    for (int iq = 0; iq < ML; ++iq) {
      if (iq >= Lq) break;
      const float qi = lane < dh ? qv[iq] : 0.f;
      const float doi = lane < dh ? dov[iq] : 0.f;
      const size_t pbase = (((size_t)b * H + h) * Lq + iq) * Lk;
      float dp[ML], p[ML];
#pragma unroll
      for (int j = 0; j < ML; ++j) TL<COH>::issue(p[j], probs + pbase + (j < Lk ? j : 0));
      TL<COH>::wait_all(p);
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < ML; ++j) {
        dp[j] = 0.f;
        if (j < Lk) {
          const float m = pdrop > 0.f ? dropout_keep(dbase, pbase + j, pdrop, dkeep) : 1.f;
          dvv[j] += p[j] * m * doi;
          dp[j] = wave_sum(doi * vv[j]) * m;
          dot += dp[j] * p[j];
        } else {
          p[j] = 0.f;
        }
      }
      float dqi = 0.f;
#pragma unroll
      for (int j = 0; j < ML; ++j) {
        if (j < Lk) {
          const float ds = p[j] * (dp[j] - dot) * scale;
          dqi += ds * kv[j];
          dkv[j] += ds * qi;
        }
      }
      if (lane < dh) TL<COH>::st1(dq + ((size_t)b * Lq + iq) * ldq + h * dh + lane, dqi);
    }
    if (lane < dh) {
#pragma unroll
      for (int j = 0; j < ML; ++j)
        if (j < Lk) {
          TL<COH>::st1(dk + ((size_t)b * Lk + j) * ldk + h * dh + lane, dkv[j]);
          TL<COH>::st1(dv + ((size_t)b * Lk + j) * ldk + h * dh + lane, dvv[j]);
        }
    }
  }
}

// ---- residual + dropout + LayerNorm on 256-wide token rows, one wave per row (norm.hip: add_drop_ln_fwd256 / bwd256) -----
//   s = x + dropout(sub)   (x may be null);   y = LayerNorm(s)   (gamma null: s only)
template <bool COH>
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
    f32x4 sv, xv = z;
    TL<COH>::issue(sv, sub + o);
    if (x) TL<COH>::issue(xv, x + o);
    TL<COH>::wait(sv, xv);
    if (p > 0.f) {
      f32x4 m4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { m4[e] = dropout_keep(base, o + e, p, keep); sv[e] *= m4[e]; }
      TL<COH>::st4(mask + o, m4);
    }
    const f32x4 s4 = x ? sv + xv : sv;
    if (s_out) TL<COH>::st4(s_out + o, s4);
    if (!gamma) continue;
    const float m = wave_sum(s4[0] + s4[1] + s4[2] + s4[3]) / D;
    float qq = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = s4[e] - m; qq += d * d; }
    const float r = rsqrtf(wave_sum(qq) / D + eps);
    f32x4 y4;
#pragma unroll
    for (int e = 0; e < 4; ++e) y4[e] = (s4[e] - m) * r * g4[e] + b4[e];
    TL<COH>::st4(y + o, y4);
    if (lane == 0) { TL<COH>::st1(mean + row, m); TL<COH>::st1(rstd + row, r); }
  }
}

//   ds = ds_ext + LayerNorm_bwd(dy; s)   (either term may be absent);   dx = ds;   dsub = ds * mask;   dgamma, dbeta += ...
template <bool COH>
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
    f32x4 dy4 = z, s4 = z, e4 = z, k4 = f32x4{1.f, 1.f, 1.f, 1.f};
    float m = 0.f, r = 0.f;
    if (dy) { TL<COH>::issue(dy4, dy + o); TL<COH>::issue(s4, s + o); TL<COH>::issue(m, mean + row); TL<COH>::issue(r, rstd + row); }
    if (ds_ext) TL<COH>::issue(e4, ds_ext + o);
    if (mask) TL<COH>::issue(k4, mask + o);
    TL<COH>::wait(dy4, s4, e4, k4); TL<COH>::wait(m, r);
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
    if (dx) TL<COH>::st4(dx + o, d4);
    if (dsub) TL<COH>::st4(dsub + o, d4 * k4);
  }
  if (dy && dgamma) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsafeAtomicAdd(dgamma + lane * 4 + e, accg[e]); unsafeAtomicAdd(dbeta + lane * 4 + e, accb[e]); }
  }
}

template <bool COH>
__device__ __forceinline__ void tok_run_op(const ast_tok_op_t& op, const int wg, const int G, unsigned char* lds, const int64_t* d_offset) {
  switch (op.type) {
    case AST_TOK_GEMM: {
      const int M = op.i[0];
      if (M <= 16) tok_gemm<COH, 1>(op, wg, G, lds, d_offset);
      else if (M <= 32) tok_gemm<COH, 2>(op, wg, G, lds, d_offset);
      else tok_gemm<COH, 4>(op, wg, G, lds, d_offset);
    } break;
    case AST_TOK_ATTN_FWD: {
      const int ml = max(op.i[2], op.i[3]);
      if (ml <= 4) tok_attn_fwd<COH, 4>(op, wg, G, d_offset); else tok_attn_fwd<COH, 8>(op, wg, G, d_offset);
    } break;
    case AST_TOK_ATTN_BWD: {
      const int ml = max(op.i[2], op.i[3]);
      if (ml <= 4) tok_attn_bwd<COH, 4>(op, wg, G, d_offset); else tok_attn_bwd<COH, 8>(op, wg, G, d_offset);
    } break;
    case AST_TOK_ADLN_FWD: tok_adln_fwd<COH>(op, wg, G, d_offset); break;
    case AST_TOK_ADLN_BWD: tok_adln_bwd<COH>(op, wg, G); break;
    default: break;
  }
}

// One launch per op (G = the op's task count): the same op bodies behind ordinary kernel boundaries.
__global__ __launch_bounds__(256) void tok_op_kernel(const ast_tok_op_t op, const int64_t* __restrict__ d_offset) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tok_lds[];
  tok_run_op<false>(op, blockIdx.x, gridDim.x, tok_lds, d_offset);
}

#ifdef AST_STAMPS
__device__ unsigned long long tok_stamps[32 * 64 * 3];      // [workgroup][op][start, computed, barrier passed] wall clock (100 MHz)
#define TOK_STAMP(j) do { if (threadIdx.x == 0 && wg < 32 && k < 64) tok_stamps[(wg * 64 + k) * 3 + (j)] = (j) == 2 ? __builtin_readcyclecounter() : wall_clock64(); } while (0)
#else
#define TOK_STAMP(j) do { } while (0)
#endif

__global__ __launch_bounds__(256) void tok_program_kernel(const TokProgram prog, unsigned* __restrict__ sync, int* __restrict__ status,
                                                          const int64_t* __restrict__ d_offset) {
  // Participants = the first G workgroups that LAND on XCD prog.xcd (HW_REG_XCC_ID) and draw a ticket.  Workgroups are
  // dealt over the 8 XCDs roughly round-robin, but neither the starting XCD nor the exact share is fixed (blockIdx % 8
  // was wrong under graph replay, and an 8 G grid did not always put G on the target), so the grid is 16 G and the surplus
  // exits.  `sync` (agent-scope atomics throughout): [0] barrier arrivals, [1] finished participants, and in another cache
  // line [16] exited workgroups, [17] tickets.
  // The last workgroup of the grid to exit zeroes all of them: the next launch (a replay of the same graph node) starts
  // from zero whatever happened in this one.
  extern __shared__ __attribute__((aligned(16))) unsigned char tok_lds[];
  const int G = prog.G;
  auto leave = [&]() __attribute__((always_inline)) {
    if (threadIdx.x == 0) {
      const unsigned gone = __hip_atomic_fetch_add(sync + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (gone == gridDim.x - 1) {
        // The LAST workgroup of the grid to exit resets every counter unconditionally.  (The barrier counters used to be
        // zeroed only when exactly G participants finished: with fewer than G workgroups landing on the target XCD the
        // barriers time out, the finished count never reaches G, and the stale arrivals would let every later launch or
        // graph replay pass its barriers early -- silent data races.)  Fewer than G tickets: status 3.
        if (__hip_atomic_load(sync + 17, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G)
          __hip_atomic_store(status, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 17, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  if ((int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15) != prog.xcd) { leave(); return; }
  if (threadIdx.x == 0) *reinterpret_cast<unsigned*>(tok_lds) = __hip_atomic_fetch_add(sync + 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int wg = (int)*reinterpret_cast<volatile unsigned*>(tok_lds);
  __syncthreads();
  if (wg >= G) { leave(); return; }
  Bar bar{sync, status, G, 0u, false};
  for (int k = 0; k < prog.nops; ++k) {
    const ast_tok_op_t& op = prog.op[k];
    TOK_STAMP(0);
    tok_run_op<true>(op, wg, G, tok_lds, d_offset);
    TOK_STAMP(1);
    if (!(op.flags & AST_TOK_NO_BARRIER) && k + 1 < prog.nops) grid_barrier(bar);
    TOK_STAMP(2);
  }
  __syncthreads();
  leave();                                        // (the last workgroup of the grid to exit zeroes all counters)
}

}  // namespace

extern "C" int ast_tok_max_ops(void) { return TOK_MAXOPS; }
#ifdef AST_STAMPS
extern "C" int ast_debug_read_tok_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tok_stamps), sizeof(unsigned long long) * 32 * 64 * 3);
}
#endif

extern "C" int ast_tok_program(const ast_tok_op_t* ops, int nops, int G, int xcd, void* sync, int* status, const int64_t* d_offset,
                               void* stream) {
  if (!ops || nops <= 0 || nops > TOK_MAXOPS) AST_FAIL("ast_tok_program: 1..%d ops per launch (got %d)", TOK_MAXOPS, nops);
  const bool per_op = G <= 0;                   // G <= 0: one ordinary launch per op
  if (!per_op && (G > 32 || xcd < 0 || xcd > 7 || !sync || !status)) AST_FAIL("ast_tok_program: bad G / xcd / sync / status");
  TokProgram prog;
  prog.nops = nops; prog.G = G; prog.xcd = xcd; prog.pad = 0;
  for (int k = 0; k < nops; ++k) {
    const ast_tok_op_t& op = ops[k];
    switch (op.type) {
      case AST_TOK_GEMM:
        if (op.i[0] <= 0 || op.i[0] > 64 || op.i[1] % 16 || op.i[2] % 64 || op.i[3] % 4 || op.i[4] % 4 || op.i[5] % 4 || !op.in[0] || !op.in[1] || !op.out[0])
          AST_FAIL("ast_tok_program: op %d: GEMM needs 1..64 rows, N %% 16 == 0, K %% 64 == 0, leading dimensions %% 4 == 0 (M %d N %d K %d)", k, op.i[0], op.i[1], op.i[2]);
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
  int mt = 1;
  for (int k = 0; k < nops; ++k)
    if (ops[k].type == AST_TOK_GEMM) mt = std::max(mt, ops[k].i[0] <= 16 ? 1 : (ops[k].i[0] <= 32 ? 2 : 4));
  const int lds = tok_lds_bytes(mt);
  static bool attr_set = false;
  if (!attr_set) {
    AST_HIP(hipFuncSetAttribute((const void*)tok_program_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, tok_lds_bytes(4)));
    attr_set = true;
  }
  if (per_op) {
    static bool attr1 = false;
    if (!attr1) {
      AST_HIP(hipFuncSetAttribute((const void*)tok_op_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, tok_lds_bytes(4)));
      attr1 = true;
    }
    for (int k = 0; k < nops; ++k) {
      const ast_tok_op_t& op = ops[k];
      int grid = 1, l = 0;
      if (op.type == AST_TOK_GEMM) { grid = op.i[1] / 16; l = tok_lds_bytes(op.i[0] <= 16 ? 1 : (op.i[0] <= 32 ? 2 : 4)); }
      else if (op.type == AST_TOK_ATTN_FWD || op.type == AST_TOK_ATTN_BWD) grid = (op.i[0] * op.i[1] + 3) / 4;
      else grid = (op.i[0] + 3) / 4;
      hipLaunchKernelGGL(tok_op_kernel, dim3(grid), dim3(256), l, (hipStream_t)stream, op, d_offset);
      AST_CHECK_LAUNCH();
    }
    return 0;
  }
  hipLaunchKernelGGL(tok_program_kernel, dim3(16 * G), dim3(256), lds, (hipStream_t)stream, prog, (unsigned*)sync, status, d_offset);
  AST_CHECK_LAUNCH();
  return 0;
}
