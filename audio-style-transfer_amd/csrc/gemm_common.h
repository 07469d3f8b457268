// Pieces shared by the GEMM translation units (igemm.hip: forward / data-gradient kernels; wgrad.hip: weight gradients).
#pragma once
#include <type_traits>
#include "ast_common.h"
#include "../../include/ast_hip.h"

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  using frag = f32x4;
  // lane (r, g) holds k = 4g+e (e = 0..3) of its row; step e contracts over g.
  static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
    return c;
  }
};

struct RowPix { int off, hs0, ws0; bool valid; };

__device__ __forceinline__ void decode_tap(int tp, int& dh, int& dw, int& wt) {
  dh = (tp & 255) - 64; dw = ((tp >> 8) & 255) - 64; wt = tp >> 16;
}

// Epilogue helper: lane owns 4 consecutive channels of one destination pixel.
template <typename T>
__device__ __forceinline__ void store4(T* p, float (&v)[4], bool accumulate, bool relu) {
  if constexpr (sizeof(T) == 4) {
    f32x4* q = reinterpret_cast<f32x4*>(p);
    if (accumulate) { const f32x4 o = *q; for (int r = 0; r < 4; ++r) v[r] += o[r]; }
    if (relu) for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    *q = f32x4{v[0], v[1], v[2], v[3]};
  } else {
    bf16x4* q = reinterpret_cast<bf16x4*>(p);
    if (accumulate) { const bf16x4 o = *q; for (int r = 0; r < 4; ++r) v[r] += (float)o[r]; }
    if (relu) for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    *q = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  }
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// floor(n / d) for 0 <= n < 2^31 through a float reciprocal (+-1 fix-up); integer division costs ~40 VALU.
__device__ __forceinline__ int fdiv(int n, int d, float rcp) {
  if (d == 1) return n;
  int q = (int)((float)n * rcp);
  int r = n - q * d;
  if (r < 0) { --q; r += d; }
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; r -= d; }
  if (r >= d) ++q;
  return q;
}

inline int check_gather(const ast_gather_t* g, const char* who) {
  if (!g) AST_FAIL("%s: null geometry", who);
  if (g->Cs <= 0 || g->Cd <= 0 || (g->Cs & 7) || (g->Cd & 7)) AST_FAIL("%s: channels must be positive multiples of 8 (Cs=%d Cd=%d)", who, g->Cs, g->Cd);
  if (g->ntaps < 0 || g->ntaps > AST_MAX_TAPS || g->wtaps < 1 || g->wtaps > AST_MAX_TAPS) AST_FAIL("%s: bad tap counts %d/%d", who, g->ntaps, g->wtaps);
  if (g->N <= 0 || g->Hm <= 0 || g->Wm <= 0 || g->Hs <= 0 || g->Ws <= 0 || g->Hd <= 0 || g->Wd <= 0) AST_FAIL("%s: empty tensor", who);
  for (int t = 0; t < g->ntaps; ++t) if ((g->tap[t] >> 16) >= g->wtaps) AST_FAIL("%s: tap %d weight slice out of range", who, t);
  // destination pixels must stay inside the tensor (a fault here can reset the GPU)
  const long hmax = (long)(g->Hm - 1) * g->dsh + g->doh, wmax = (long)(g->Wm - 1) * g->dsw + g->dow;
  if (g->doh < 0 || g->dow < 0 || hmax >= g->Hd || wmax >= g->Wd) AST_FAIL("%s: destination grid exceeds tensor (%ld,%ld) vs (%d,%d)", who, hmax, wmax, g->Hd, g->Wd);
  if ((long)g->N * g->Hs * g->Ws * g->Cs * 4 >= (1L << 31) || (long)g->N * g->Hm * g->Wm >= (1L << 31) ||
      (long)g->Cd * g->wtaps * g->Cs * 4 >= (1L << 31)) AST_FAIL("%s: tensor exceeds the 2 GiB buffer-addressing range", who);
  return 0;
}

}  // namespace
