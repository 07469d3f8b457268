// Micro-benchmark: per-CU L1 ingest of MFMA-fragment-shaped loads straight from an L2-resident NHWC image.
// mode F: lane (fr = l & 15, fq = l >> 4) reads the 16-byte chunk fq of pixel fr  (the 16x16x32 B-operand layout);
// mode G: lane l reads chunk l & 3 of pixel l >> 2                                 (4 lanes per pixel: the staged kernel's gather);
// mode H: lane l reads chunk l & 7 of pixel l >> 3 (8 lanes per pixel, 128 B)
// `pitch` = bytes between consecutive pixels (Cs * 2).  NL loads (pixel tiles) per iteration per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE, int NL>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ src, unsigned* __restrict__ out, int pitch, int iters, unsigned bytes, int npix) {
  const __amdgpu_buffer_rsrc_t R = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int PPL = MODE == 0 ? 16 : (MODE == 1 ? 16 : 8);       // pixels per load
  const int pix = MODE == 0 ? (l & 15) : (MODE == 1 ? (l >> 2) : (l >> 3));
  const int ch = MODE == 0 ? (l >> 4) : (MODE == 1 ? (l & 3) : (l & 7));
  const int kstep = MODE == 2 ? 128 : 64;
  const int base_pix = ((blockIdx.x * 4 + w) * NL * PPL * 7) % (npix - NL * PPL - 8);
  unsigned off[NL];
  for (int i = 0; i < NL; ++i) off[i] = (unsigned)((base_pix + i * PPL + pix) * pitch + ch * 16);
  u32x4 acc = {0, 0, 0, 0};
  const int kwrap = pitch / kstep;
  for (int it = 0; it < iters; ++it) {
    // walk the channel chunks of the pixel (K steps inside a tap), then shift one pixel (next tap)
    const unsigned d = (unsigned)((it % kwrap) * kstep + ((it / kwrap) % 3) * pitch);
    u32x4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(R, off[i] + d, 0, 0);
#pragma unroll
    for (int i = 0; i < NL; ++i) acc ^= v[i];
  }
  if (acc[0] == 0x12345 && acc[1] == 7) out[0] = acc[2] + acc[3];
}

template <int MODE, int NL>
void run(const char* name, const unsigned char* d, unsigned* out, int pitch, int blocks, int iters, size_t total) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int npix = (int)(total / pitch);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE, NL>), dim3(blocks), dim3(256), 0, 0, d, out, pitch, iters, (unsigned)total, npix);
  hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((k<MODE, NL>), dim3(blocks), dim3(256), 0, 0, d, out, pitch, iters, (unsigned)total, npix);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 100.0, gb = (double)blocks * iters * 256 * NL * 16 / 1e9;
  printf("%-28s pitch %4d NL %d blocks %5d: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU\n", name, pitch, NL, blocks, us, gb / us * 1e3,
         gb * 1e9 / (us * 1e-6) / 2.4e9 / 256);
}

int main() {
  const size_t total = 8u << 20;            // 8 MB: L2/MALL resident
  unsigned char* d; unsigned* out;
  hipMalloc(&d, total); hipMemset(d, 1, total); hipMalloc(&out, 64);
  for (int pitch : {64, 128, 256, 512})
    for (int blocks : {512, 2048}) {
      run<0, 4>("F fragment (16 px x 4 ch)", d, out, pitch, blocks, 288, total);
      run<1, 4>("G 4 lanes per pixel (64 B)", d, out, pitch, blocks, 288, total);
      if (pitch >= 128) run<2, 4>("H 8 lanes per pixel (128 B)", d, out, pitch, blocks, 288, total);
    }
  return 0;
}
