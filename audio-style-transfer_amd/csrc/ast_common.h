// Shared device/host helpers for libast_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum { AST_F32 = 0, AST_BF16 = 1 };

// ---- error plumbing: C-ABI returns 0 / negative, message via ast_last_error()
void ast_set_error(const char* fmt, ...);
#define AST_FAIL(...) do { ast_set_error(__VA_ARGS__); return -1; } while (0)
#define AST_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) { ast_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return -2; } } while (0)
#define AST_HIP(x) do { hipError_t e_ = (x); \
    if (e_ != hipSuccess) { ast_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return -3; } } while (0)

// ---- 8-channel unit load/store (NHWC tensors keep C a multiple of 8)
template <typename T> struct U8;
template <> struct U8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
  }
};
template <> struct U8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
  }
};

// Wave-wide sum without the LDS crossbar: __shfl_xor is ds_bpermute (an LDS-pipe instruction, 6 dependent ones per
// sum); here 4 DPP adds give every lane its 16-lane row's sum (quad swaps, then row rotations by 4 and 8) and 4
// v_readlane pick up the four rows.  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// sum over each aligned group of 16 lanes (a DPP row); every lane of the row gets it
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);                 // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);                 // quad_perm [2,3,0,1]
  v += dpp_mov<0x124>(v);                // row_ror:4
  v += dpp_mov<0x128>(v);                // row_ror:8
  return v;
}
// Sixteen values per lane -> lane l of every 16-lane row gets the row-wide sum of value l (a transposed reduction:
// 8 + 4 + 2 + 1 exchange steps instead of 16 x 4).  Partners are l^1, l^2 (quad permutes), l^4 (row rotations by 4 and
// 12, selected by bit 2), l^8 (rotation by 8).
__device__ __forceinline__ float row16_transpose_sum(const float (&v)[16], const int l) {
  float w[8], x[4], y[2];
  const bool b0 = l & 1, b1 = l & 2, b2 = l & 4, b3 = l & 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = (b0 ? v[2 * k + 1] : v[2 * k]) + dpp_mov<0xB1>(b0 ? v[2 * k] : v[2 * k + 1]);
#pragma unroll
  for (int k = 0; k < 4; ++k) x[k] = (b1 ? w[2 * k + 1] : w[2 * k]) + dpp_mov<0x4E>(b1 ? w[2 * k] : w[2 * k + 1]);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float send = b2 ? x[2 * k] : x[2 * k + 1];
    const float lo = dpp_mov<0x124>(send), hi = dpp_mov<0x12C>(send);      // row_ror:4 = from lane l-4, row_ror:12 = from lane l+4
    y[k] = (b2 ? x[2 * k + 1] : x[2 * k]) + (b2 ? lo : hi);
  }
  return (b3 ? y[1] : y[0]) + dpp_mov<0x128>(b3 ? y[0] : y[1]);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = row16_sum(v);
  const int i = __builtin_bit_cast(int, v);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48)));
}
#ifdef AST_WAVE_SUM_SHFL
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
#else
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }
#endif
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// splitmix64 finaliser: the dropout masks are mix64(base + element index), base = f(seed, device step counter)
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__device__ __forceinline__ float dropout_keep(uint64_t base, uint64_t i, float p, float keep) {
  const uint32_t r = (uint32_t)(mix64(base + i) >> 40);             // 24 random bits
  return ((float)r * (1.f / 16777216.f)) >= p ? keep : 0.f;
}

// block-wide sum for blockDim.x = multiple of 64 (<= 1024); all threads get the result
__device__ __forceinline__ float block_sum(float v, float* red /* >= 17 floats of LDS */) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) { float s = 0.f; for (int i = 0; i < nw; ++i) s += red[i]; red[16] = s; }
  __syncthreads();
  return red[16];
}

#define AST_DISPATCH_T(dtype, ...) \
  do { if ((dtype) == AST_F32) { using T = float; __VA_ARGS__; } \
       else if ((dtype) == AST_BF16) { using T = bf16_t; __VA_ARGS__; } \
       else AST_FAIL("%s: bad dtype %d", __func__, (int)(dtype)); } while (0)
