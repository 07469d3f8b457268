// Micro-benchmark: per-CU ingest rate of the igemm operand-load patterns (no MFMA, no LDS unless asked).
// Each workgroup of 256 threads streams `iters` K tiles of ROWS rows x KCH 16-byte chunks from an L2-resident matrix
// whose rows are `pitch` bytes apart.  mode 0: row-gather (lane -> row tid/KCH, chunk tid%KCH), as igemm does;
// mode 1: the same bytes but every wave reads 1 KB contiguous (rows of 1 KB); mode 2: mode 0 + ds_write to LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int KCH, int NLOAD, int MODE>
__global__ __launch_bounds__(256) void ld_kernel(const unsigned char* __restrict__ src, unsigned* __restrict__ out, int rows_total,
                                                 int pitch, int iters, unsigned bytes) {
  __shared__ u32x4 lds[256 * NLOAD * 2];
  const __amdgpu_buffer_rsrc_t R = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  const int tid = threadIdx.x;
  const int cc = tid % KCH, r0 = tid / KCH;
  constexpr int RPP = 256 / KCH;
  const int row_base = (blockIdx.x * 64) % (rows_total - RPP * NLOAD);
  u32x4 acc = {0, 0, 0, 0};
  unsigned off[NLOAD];
  for (int i = 0; i < NLOAD; ++i) {
    if (MODE == 1) off[i] = (unsigned)((row_base + (tid >> 6) + 4 * i) * pitch + (tid & 63) * 16);
    else off[i] = (unsigned)((row_base + r0 + RPP * i) * pitch + cc * 16);
  }
  const int kstep = MODE == 1 ? 1024 : KCH * 16;
  const int kwrap = pitch / kstep;
  for (int it = 0; it < iters; ++it) {
    const unsigned d = (unsigned)((it % kwrap) * kstep);
    u32x4 v[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(R, off[i] + d, 0, 0);
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) lds[(it & 1) * 256 * NLOAD + i * 256 + tid] = v[i];
      __syncthreads();
      acc ^= lds[(it & 1) * 256 * NLOAD + ((tid * 7) & 255)];
    } else {
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) acc ^= v[i];
    }
  }
  if (acc[0] == 0x12345 && acc[1] == 7) out[0] = acc[2] + acc[3];
}

template <int KCH, int NLOAD, int MODE>
void run(const char* name, const unsigned char* d, unsigned* out, int rows, int pitch, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned bytes = (unsigned)((size_t)rows * pitch);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((ld_kernel<KCH, NLOAD, MODE>), dim3(blocks), dim3(256), 0, 0, d, out, rows, pitch, iters, bytes);
  hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((ld_kernel<KCH, NLOAD, MODE>), dim3(blocks), dim3(256), 0, 0, d, out, rows, pitch, iters, bytes);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 100.0;
  const double gb = (double)blocks * iters * 256 * NLOAD * 16 / 1e9;
  printf("%-34s blocks %5d iters %4d: %7.1f us  %6.2f TB/s  (%5.1f B/clk/CU at 2.4GHz over min(blocks,256) CUs)\n", name, blocks, iters, us,
         gb / us * 1e6 / 1e3, gb * 1e9 / (us * 1e-6) / 2.4e9 / (blocks < 256 ? blocks : 256));
}

int main() {
  const int rows = 4096, pitch = 1024;            // 4 MB: L2/MALL resident
  unsigned char* d; unsigned* out;
  hipMalloc(&d, (size_t)rows * pitch); hipMemset(d, 1, (size_t)rows * pitch); hipMalloc(&out, 64);
  for (int blocks : {256, 512, 1024, 2048}) {
    run<8, 4, 0>("gather kch8 4 loads", d, out, rows, pitch, blocks, 288);
    run<4, 4, 0>("gather kch4 4 loads", d, out, rows, pitch, blocks, 288);
    run<8, 4, 1>("wave-contiguous 1KB 4 loads", d, out, rows, pitch, blocks, 288);
    run<8, 8, 0>("gather kch8 8 loads", d, out, rows, pitch, blocks, 144);
    run<8, 4, 2>("gather kch8 4 loads + LDS", d, out, rows, pitch, blocks, 288);
  }
  return 0;
}
