// Which atomic / poll combination makes a working barrier among workgroups of ONE XCD (selected by HW_REG_XCC_ID + ticket)?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ unsigned ld_sc0(const unsigned* p) { unsigned v; asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void inc_plain(unsigned* p) { const unsigned one = 1u; asm volatile("global_atomic_add %0, %1, off" : : "v"(p), "v"(one) : "memory"); }
__device__ __forceinline__ void inc_sc0(unsigned* p) { const unsigned one = 1u; unsigned old; asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(old) : "v"(p), "v"(one) : "memory"); }
__global__ void k(unsigned* sync, int G, int rounds, int xcd, int mode, long long* cycles, int* err, int* landed) {
  __shared__ unsigned tk;
  if ((int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15) != xcd) return;
  if (threadIdx.x == 0) tk = __hip_atomic_fetch_add(sync + 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int wg = tk;
  if (threadIdx.x == 0) atomicAdd(landed, 1);
  if (wg >= G) return;
  long long t0 = wall_clock64();
  for (int r = 1; r <= rounds; ++r) {
    if (wg == (r % G)) for (volatile int d = 0; d < 2000; ++d) {}    // one straggler per round
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (mode == 0) inc_plain(sync); else if (mode == 1) inc_sc0(sync); else __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while ((mode == 2 ? __hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ld_sc0(sync)) < (unsigned)(G * r)) { if (++spins > (1 << 18)) { *err = 1; r = rounds; break; } __builtin_amdgcn_s_sleep(1); }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && wg == 0) *cycles = wall_clock64() - t0;
}
int main() {
  unsigned* sync; long long* cyc; int* err; int* landed;
  CK(hipMalloc(&sync, 128)); CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&landed, 4));
  for (int G : {16, 32})
    for (int mode = 0; mode < 3; ++mode)
      for (int mult : {8, 16}) {
        CK(hipMemset(sync, 0, 128)); CK(hipMemset(err, 0, 4)); CK(hipMemset(landed, 0, 4));
        hipLaunchKernelGGL(k, dim3(mult * G), dim3(256), 0, 0, sync, G, 200, 5, mode, cyc, err, landed);
        CK(hipDeviceSynchronize());
        long long c; int ev, ld; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ev, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ld, landed, 4, hipMemcpyDeviceToHost));
        printf("G=%2d grid=%3d  %-34s landed on the XCD: %3d   %.2f us per round  err=%d\n", G, mult * G, mode == 0 ? "atomic (no bits) + sc0 poll" : mode == 1 ? "atomic rtn sc0 + sc0 poll" : "agent-scope atomics", ld, c / 100.0 / 200, ev);
      }
  return 0;
}
