// Micro-benchmark of the igemm K-loop skeleton without global loads: per-iteration cycles of
//  mode 0: 4 ds_write_b128 + barrier;  1: + 8 ds_read_b128;  2: + 8 MFMA 16x16x32 bf16;  3: reads+MFMA, no writes/barrier
// at 1, 2 and 4 workgroups (of 4 waves) per CU.  Cycles from s_memtime (100 MHz constant clock? -> we print both
// wall_clock64 ticks and event time).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int MODE>
__global__ __launch_bounds__(256) void loop_kernel(float* out, int iters, int wr_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  u32x4 v = {(unsigned)tid, 1, 2, 3};
  f32x4 acc[4] = {};
  const int woff = (tid >> 3) * 64 + (tid & 3) * 16 + (tid & 4 ? wr_stride : 0);
  const int roffa = (lane & 15) * 64 + (lane >> 4) * 16, roffb = 4096 + roffa;
  for (int it = 0; it < iters; ++it) {
    unsigned char* base = lds + (it & 1) * 16384;
    if (MODE != 3) {
      if (wr_stride == 1) {                     // same bytes as 8 ds_write_b64, linear
        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x2*>(base + i * 2048 + tid * 8) = u32x2{v[0], v[1] + i};
      } else if (wr_stride == 2) {              // 16 ds_write_b32
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<unsigned*>(base + i * 1024 + tid * 4) = v[0] + i;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(base + woff + i * 2048) = v;
      }
      __syncthreads();
    }
    if (MODE >= 1) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = *reinterpret_cast<const bf16x8*>(base + ks * 8192 + roffa + i * 1024);
          b[i] = *reinterpret_cast<const bf16x8*>(base + ks * 8192 + roffb + i * 1024);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (MODE >= 2) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
            else acc[i * 2 + j][0] += __builtin_bit_cast(f32x4, a[i])[0] + __builtin_bit_cast(f32x4, b[j])[1];
          }
      }
    }
    v[0] += 1;
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.f) out[0] = s;
}

template <int MODE>
void run(const char* name, float* out, int blocks, int iters, int wr_stride) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)loop_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 + 8192 + 256);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((loop_kernel<MODE>), dim3(blocks), dim3(256), 32768 + 8192 + 256, 0, out, iters, wr_stride);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((loop_kernel<MODE>), dim3(blocks), dim3(256), 32768 + 8192 + 256, 0, out, iters, wr_stride);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 100.0;
  const double per_cu_blocks = blocks / 256.0;
  printf("%-40s blocks %5d wr_stride %5d: %7.1f us -> %6.0f ns per iteration (%5.0f clk @2.4GHz) per resident set\n", name, blocks, wr_stride, us,
         us * 1e3 / iters, us * 1e3 / iters * 2.4);
  (void)per_cu_blocks;
}

int main() {
  float* out; (void)hipMalloc(&out, 64);
  const int iters = 2000;
  for (int blocks : {256, 512, 1024}) {
    for (int ws : {8192, 1, 2}) {
      run<0>("write4 + barrier", out, blocks, iters, ws);
      run<1>("write4 + barrier + read8", out, blocks, iters, ws);
      run<2>("write4 + barrier + read8 + mfma8", out, blocks, iters, ws);
    }
    run<3>("read8 + mfma8 only", out, blocks, iters, 8192);
  }
  return 0;
}
