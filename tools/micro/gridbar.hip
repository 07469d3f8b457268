// Microbenchmark: (1) which XCD a workgroup lands on (HW_REG_XCC_ID vs blockIdx % 8); (2) cost of a grid barrier among G
// workgroups that all sit on ONE XCD (stores write through to the shared L2; waiters invalidate their L1 with buffer_inv sc0)
// vs an agent-scope barrier among G workgroups spread over all XCDs (__threadfence: L2 write-back + invalidate).
// hipcc --offload-arch=gfx950 -O3 tools/micro/gridbar.hip -o /tmp/gridbar && /tmp/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void xcc_kernel(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15;   // HW_REG_XCC_ID[3:0]
}

// one-XCD barrier: participants = blocks with blockIdx % 8 == xcd
__global__ void bar1_kernel(unsigned* ctr, float* buf, int G, int rounds, int xcd, long long* cycles, int* err) {
  if ((blockIdx.x & 7) != xcd) return;
  const int wg = blockIdx.x >> 3;
  long long t0 = wall_clock64();
  for (int r = 1; r <= rounds; ++r) {
    // produce: every workgroup writes its slot, then reads its neighbour's slot of the previous round
    buf[wg * 64 + (threadIdx.x & 63)] = (float)r;
    __builtin_amdgcn_s_waitcnt(0);           // stores have reached L2 (L1 is write-through)
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(G * r)) { if (++spins > (1 << 22)) { *err = 1; break; } __builtin_amdgcn_s_sleep(1); }
    }
    __syncthreads();
    asm volatile("buffer_inv sc0" ::: "memory");   // drop this CU's L1 lines
    const float v = buf[((wg + 1) % G) * 64 + (threadIdx.x & 63)];
    if (v < (float)r) *err = 2;
  }
  if (threadIdx.x == 0 && wg == 0) *cycles = wall_clock64() - t0;
}

// one XCD, plain stores, device-scope invalidate (buffer_inv sc1) after the barrier
__global__ void bar1b_kernel(unsigned* ctr, float* buf, int G, int rounds, int xcd, long long* cycles, int* err) {
  if ((blockIdx.x & 7) != xcd) return;
  const int wg = blockIdx.x >> 3;
  long long t0 = wall_clock64();
  for (int r = 1; r <= rounds; ++r) {
    buf[wg * 64 + (threadIdx.x & 63)] = (float)r;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(G * r)) { if (++spins > (1 << 22)) { *err = 1; break; } __builtin_amdgcn_s_sleep(1); }
    }
    __syncthreads();
    asm volatile("buffer_inv sc1" ::: "memory");
    const float v = buf[((wg + 1) % G) * 64 + (threadIdx.x & 63)];
    if (v < (float)r) *err = 2;
  }
  if (threadIdx.x == 0 && wg == 0) *cycles = wall_clock64() - t0;
}
// one XCD, agent-scope relaxed atomics (sc1) for the data, no invalidate
__global__ void bar1c_kernel(unsigned* ctr, float* buf, int G, int rounds, int xcd, long long* cycles, int* err) {
  if ((blockIdx.x & 7) != xcd) return;
  const int wg = blockIdx.x >> 3;
  long long t0 = wall_clock64();
  for (int r = 1; r <= rounds; ++r) {
    __hip_atomic_store(buf + wg * 64 + (threadIdx.x & 63), (float)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(G * r)) { if (++spins > (1 << 22)) { *err = 1; break; } __builtin_amdgcn_s_sleep(1); }
    }
    __syncthreads();
    const float v = __hip_atomic_load(buf + ((wg + 1) % G) * 64 + (threadIdx.x & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v < (float)r) *err = 2;
  }
  if (threadIdx.x == 0 && wg == 0) *cycles = wall_clock64() - t0;
}

// one XCD, plain stores, loads that bypass L1 (sc0) through inline asm; also reads a 96 KB block per round (bandwidth check)
__global__ void bar1d_kernel(unsigned* ctr, float* buf, float* big, int G, int rounds, int xcd, long long* cycles, int* err, int mode) {
  if ((blockIdx.x & 7) != xcd) return;
  const int wg = blockIdx.x >> 3;
  long long t0 = wall_clock64();
  float acc = 0.f;
  for (int r = 1; r <= rounds; ++r) {
    buf[wg * 64 + (threadIdx.x & 63)] = (float)r;
    // every workgroup rewrites its slice of the big block, then all read all of it (a GEMM's activation operand)
    for (int i = threadIdx.x; i < 24576 / G; i += 256) big[wg * (24576 / G) + i] = (float)r;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(G * r)) { if (++spins > (1 << 22)) { *err = 1; break; } __builtin_amdgcn_s_sleep(1); }
    }
    __syncthreads();
    float v;
    const float* p = buf + ((wg + 1) % G) * 64 + (threadIdx.x & 63);
    if (mode == 0) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v < (float)r) *err = 2;
    for (int i = threadIdx.x * 4; i < 24576; i += 1024) {
      f32x4_t q;
      const float* pb = big + i;
      if (mode == 0) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(pb) : "memory");
      else { q.x = __hip_atomic_load(pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); q.y = __hip_atomic_load(pb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
             q.z = __hip_atomic_load(pb + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); q.w = __hip_atomic_load(pb + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      if (q.x < (float)r || q.w < (float)r) *err = 3;
      acc += q.y;
    }
    __syncthreads();
  }
  if (acc == 12345.f) *err = 4;
  if (threadIdx.x == 0 && wg == 0) *cycles = wall_clock64() - t0;
}

// agent-scope barrier: all blocks participate
__global__ void bar8_kernel(unsigned* ctr, float* buf, int G, int rounds, long long* cycles, int* err) {
  const int wg = blockIdx.x;
  long long t0 = wall_clock64();
  for (int r = 1; r <= rounds; ++r) {
    buf[wg * 64 + (threadIdx.x & 63)] = (float)r;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(G * r)) { if (++spins > (1 << 22)) { *err = 1; break; } __builtin_amdgcn_s_sleep(1); }
    }
    __syncthreads();
    __threadfence();
    const float v = buf[((wg + 1) % G) * 64 + (threadIdx.x & 63)];
    if (v < (float)r) *err = 2;
  }
  if (threadIdx.x == 0 && wg == 0) *cycles = wall_clock64() - t0;
}

// sc1 data path: agent-scope relaxed atomics for the data, no fences
__global__ void bar8sc1_kernel(unsigned* ctr, float* buf, int G, int rounds, long long* cycles, int* err) {
  const int wg = blockIdx.x;
  long long t0 = wall_clock64();
  for (int r = 1; r <= rounds; ++r) {
    __hip_atomic_store(buf + wg * 64 + (threadIdx.x & 63), (float)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(G * r)) { if (++spins > (1 << 22)) { *err = 1; break; } __builtin_amdgcn_s_sleep(1); }
    }
    __syncthreads();
    const float v = __hip_atomic_load(buf + ((wg + 1) % G) * 64 + (threadIdx.x & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v < (float)r) *err = 2;
  }
  if (threadIdx.x == 0 && wg == 0) *cycles = wall_clock64() - t0;
}

int main() {
  int* d_x; CK(hipMalloc(&d_x, 2048 * 4));
  hipLaunchKernelGGL(xcc_kernel, dim3(2048), dim3(64), 0, 0, d_x);
  std::vector<int> h(2048); CK(hipMemcpy(h.data(), d_x, 2048 * 4, hipMemcpyDeviceToHost));
  int mism = 0; for (int b = 0; b < 2048; ++b) mism += (h[b] != (b & 7));
  printf("xcc ids of blocks 0..15:"); for (int b = 0; b < 16; ++b) printf(" %d", h[b]); printf("   blocks with xcc != blockIdx %% 8: %d of 2048\n", mism);
  unsigned* ctr; float* buf; float* big; long long* cyc; int* err;
  CK(hipMalloc(&big, 24576 * 4));
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&buf, 1024 * 64 * 4)); CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&err, 4));
  const int rounds = 200;
  for (int G : {16, 32, 64}) {
    for (int mode = 0; mode < 7; ++mode) {
      CK(hipMemset(ctr, 0, 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(buf, 0, 1024 * 64 * 4));
      if (mode == 0) hipLaunchKernelGGL(bar1_kernel, dim3(8 * G), dim3(256), 0, 0, ctr, buf, G, rounds, 3, cyc, err);
      if (mode == 1) hipLaunchKernelGGL(bar8_kernel, dim3(G), dim3(256), 0, 0, ctr, buf, G, rounds, cyc, err);
      if (mode == 3) hipLaunchKernelGGL(bar1b_kernel, dim3(8 * G), dim3(256), 0, 0, ctr, buf, G, rounds, 3, cyc, err);
      if (mode == 4) hipLaunchKernelGGL(bar1c_kernel, dim3(8 * G), dim3(256), 0, 0, ctr, buf, G, rounds, 3, cyc, err);
      if (mode == 5) hipLaunchKernelGGL(bar1d_kernel, dim3(8 * G), dim3(256), 0, 0, ctr, buf, big, G, rounds, 3, cyc, err, 0);
      if (mode == 6) hipLaunchKernelGGL(bar1d_kernel, dim3(8 * G), dim3(256), 0, 0, ctr, buf, big, G, rounds, 3, cyc, err, 1);
      if (mode == 2) hipLaunchKernelGGL(bar8sc1_kernel, dim3(G), dim3(256), 0, 0, ctr, buf, G, rounds, cyc, err);
      CK(hipDeviceSynchronize());
      long long c; int ev; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ev, err, 4, hipMemcpyDeviceToHost));
      printf("G=%2d %-28s %.2f us per (write, barrier, read) round   err=%d\n", G, mode == 0 ? "one XCD (buffer_inv sc0)" : mode == 1 ? "all XCDs (__threadfence)" : mode == 2 ? "all XCDs (sc1 data, no fence)" : mode == 3 ? "one XCD (buffer_inv sc1)" : mode == 4 ? "one XCD (sc1 data, no inv)" : mode == 5 ? "one XCD sc0 loads + 96KB all-read" : "one XCD sc1 loads + 96KB all-read", c / 100.0 / rounds, ev);
    }
  }
  return 0;
}
