// Reduction of per-workgroup partial tiles: (a) agent-scope f32 atomics straight into the result (what wgrad_kernel does);
// (b) two levels: workgroup-scope (L2-level) atomics into a partial buffer per XCD (indexed by HW_REG_XCC_ID, so every adder
// of an address sits behind the same L2), an L2-level arrival counter, and the last workgroup of each XCD adds that XCD's
// partial into the result with agent-scope atomics (8 deep instead of W deep) and leaves the partial zeroed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int TILE = 64 * 192;            // floats per tile (48 KB)
__global__ __launch_bounds__(256) void direct_kernel(float* out, int ntiles) {
  const int tile = blockIdx.x % ntiles;
  for (int i = threadIdx.x; i < TILE; i += 256) unsafeAtomicAdd(out + (size_t)tile * TILE + i, 1.0f);
}
// the same atomics with every 64-byte line of the result placed at a pitch of `pitch` floats (spread over more channels)
__global__ __launch_bounds__(256) void spread_kernel(float* out, int ntiles, int pitch) {
  const int tile = blockIdx.x % ntiles;
  for (int i = threadIdx.x; i < TILE; i += 256) {
    const size_t line = ((size_t)tile * TILE + i) >> 4;
    unsafeAtomicAdd(out + line * pitch + (i & 15), 1.0f);
  }
}
__global__ __launch_bounds__(256) void twolevel_kernel(float* out, float* part, unsigned* cnt, int ntiles, int* per_xcd_expected) {
  __shared__ unsigned last;
  const int tile = blockIdx.x % ntiles;
  const int xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15;
  float* p = part + ((size_t)xcc * ntiles + tile) * TILE;
  for (int i = threadIdx.x; i < TILE; i += 256) __hip_atomic_fetch_add(p + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    // how many workgroups of this (xcd, tile) exist: blocks b with b % ntiles == tile that land on this XCD = those with b % 8 == c
    // for the launch's rotation; counted on the host for the test (expected[xcc * ntiles + tile]) -- here passed in
    const unsigned old = __hip_atomic_fetch_add(cnt + xcc * ntiles + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    last = (old + 1 == (unsigned)per_xcd_expected[xcc * ntiles + tile]);
  }
  __syncthreads();
  if (!last) return;
  for (int i = threadIdx.x; i < TILE; i += 256) {
    const float v = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsafeAtomicAdd(out + (size_t)tile * TILE + i, v);
    __hip_atomic_store(p + i, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (threadIdx.x == 0) __hip_atomic_store(cnt + xcc * ntiles + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ void count_kernel(int* expected, int ntiles) {      // same grid: count the workgroups per (xcd, tile) of THIS rotation
  if (threadIdx.x == 0) atomicAdd(expected + (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15) * ntiles + blockIdx.x % ntiles, 1);
}
int main() {
  const int ntiles = 3, W = 256;            // 256 workgroups, 85 per tile
  float *out, *part; unsigned* cnt; int* expected;
  CK(hipMalloc(&out, ntiles * TILE * 4)); CK(hipMalloc(&part, 8 * ntiles * TILE * 4)); CK(hipMalloc(&cnt, 8 * ntiles * 4)); CK(hipMalloc(&expected, 8 * ntiles * 4));
  CK(hipMemset(part, 0, 8 * ntiles * TILE * 4)); CK(hipMemset(cnt, 0, 8 * ntiles * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> h(ntiles * TILE);
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e9f; int bad = 0;
    for (int it = 0; it < 6; ++it) {
      CK(hipMemset(out, 0, ntiles * TILE * 4));
      if (mode == 1) { CK(hipMemset(expected, 0, 8 * ntiles * 4)); hipLaunchKernelGGL(count_kernel, dim3(W), dim3(256), 0, 0, expected, ntiles); }
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(direct_kernel, dim3(W), dim3(256), 0, 0, out, ntiles);
      else hipLaunchKernelGGL(twolevel_kernel, dim3(W), dim3(256), 0, 0, out, part, cnt, ntiles, expected);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
      CK(hipMemcpy(h.data(), out, ntiles * TILE * 4, hipMemcpyDeviceToHost));
      for (int t = 0; t < ntiles; ++t) { const float want = (float)((W - t + ntiles - 1) / ntiles); for (int i = 0; i < TILE; ++i) bad += (h[t * TILE + i] != want); }
    }
    printf("%-44s %.1f us   wrong values: %d\n", mode == 0 ? "agent-scope atomics into the result" : "L2-level partial per XCD + last arriver", best * 1e3f, bad);
  }
  float* big; CK(hipMalloc(&big, (size_t)ntiles * TILE / 16 * 1024 * 4));
  for (int pitch : {16, 32, 64, 256, 1024}) {
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      CK(hipMemset(big, 0, (size_t)ntiles * TILE / 16 * pitch * 4)); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(spread_kernel, dim3(W), dim3(256), 0, 0, big, ntiles, pitch);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    printf("agent-scope atomics, 64-byte lines at a pitch of %4d bytes: %.1f us\n", pitch * 4, best * 1e3f);
  }
  return 0;
}
