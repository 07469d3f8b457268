// Gathered (implicit-im2col) GEMM on MFMA for gfx950 -- the dominant kernels of
// the train step: Conv2d / ConvTranspose2d / Linear forward and data gradient
// (igemm_kernel) and their weight gradients (wgrad_kernel).
//
// Data layout: activations NHWC, C % 8 == 0; weights packed [rows][wtaps][Cs]
// so that the GEMM K index is (tap, channel) with channels contiguous: every
// 16-byte chunk a lane moves (8 bf16 / 4 f32) lies inside one tap.
//
// One K-step stages 64 B per row (4 chunks) of both operands through LDS and
// feeds  v_mfma_f32_16x16x32_bf16  (bf16 storage) or 4x v_mfma_f32_16x16x4_f32
// (f32 storage, exact) -- the LDS image and fragment reads are identical for
// both, only the MFMA differs.  Weights are the MFMA "A" operand, so D[row][col]
// has the output CHANNEL on the register index: each lane owns 4 consecutive
// channels of one pixel and stores them with one 8/16-byte store into NHWC.
#include "ast_common.h"
#include "../../include/ast_hip.h"

namespace {

constexpr int ROWB = 80;  // LDS row pitch: 64 B of data + 16 B pad (bank spread for ds_read_b128)

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

struct RowPix { int pixbase, hs0, ws0; bool valid; };

__device__ __forceinline__ void decode_tap(int tp, int& dh, int& dw, int& wt) {
  dh = (tp & 255) - 64; dw = ((tp >> 8) & 255) - 64; wt = tp >> 16;
}

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_kernel(const T* __restrict__ src, const T* __restrict__ wgt,
                                                     const float* __restrict__ bias, T* __restrict__ dst,
                                                     const ast_gather_t g, const int M, const int flags) {
  constexpr int E = 16 / sizeof(T);
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int AI = BM / 64, BI = (BN + 63) / 64;
  static_assert(WM * WN == 4 && BM % 64 == 0 && WTM % 16 == 0 && WTN % 16 == 0, "tile");
  using frag = typename Mma<T>::frag;

  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (BM + BN) * ROWB];
  __shared__ int taptab[AST_MAX_TAPS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int cc = tid & 3, r0 = tid >> 2;
  const int bm0 = blockIdx.x * BM, bn0 = blockIdx.y * BN;
  const int cpc = g.Cs / E;
  const int nchunks = g.ntaps * cpc;
  const int KT = (nchunks + 3) >> 2;
  const int HWm = g.Hm * g.Wm;

  if (tid < AST_MAX_TAPS) taptab[tid] = g.tap[tid];

  RowPix rp[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = bm0 + r0 + 64 * i;
    rp[i].valid = m < M;
    const int mm = rp[i].valid ? m : 0;
    const int n = mm / HWm, rem = mm - n * HWm;
    const int hm = rem / g.Wm, wq = rem - hm * g.Wm;
    rp[i].pixbase = n * g.Hs * g.Ws;
    rp[i].hs0 = hm * g.sh + g.oh;
    rp[i].ws0 = wq * g.sw + g.ow;
  }
  __syncthreads();

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 areg[AI], breg[BI];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);

  auto load_tile = [&](int kt) {
    const int kc = kt * 4 + cc;
    const bool kval = kc < nchunks;
    const int t = kval ? kc / cpc : 0;
    const int c0 = (kc - t * cpc) * E;
    int dh, dw, wt;
    decode_tap(taptab[t], dh, dw, wt);
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int hs = rp[i].hs0 + dh, ws = rp[i].ws0 + dw;
      const bool ok = kval && rp[i].valid && (unsigned)hs < (unsigned)g.Hs && (unsigned)ws < (unsigned)g.Ws;
      areg[i] = ok ? *reinterpret_cast<const uint4*>(src + ((size_t)(rp[i].pixbase + hs * g.Ws + ws) * g.Cs + c0)) : zero4;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int row = r0 + 64 * i, co = bn0 + row;
      const bool ok = kval && row < BN && co < g.Cd;
      breg[i] = ok ? *reinterpret_cast<const uint4*>(wgt + (((size_t)co * g.wtaps + wt) * g.Cs + c0)) : zero4;
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* As = lds + buf * (BM + BN) * ROWB;
    unsigned char* Bs = As + BM * ROWB;
#pragma unroll
    for (int i = 0; i < AI; ++i) *reinterpret_cast<uint4*>(As + (r0 + 64 * i) * ROWB + cc * 16) = areg[i];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int row = r0 + 64 * i;
      if (row < BN) *reinterpret_cast<uint4*>(Bs + row * ROWB + cc * 16) = breg[i];
    }
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) load_tile(kt + 1);
    const unsigned char* As = lds + cur * (BM + BN) * ROWB;
    const unsigned char* Bs = As + BM * ROWB;
    frag wf[TN], xf[TM];
#pragma unroll
    for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const frag*>(Bs + (wn * WTN + i * 16 + fr) * ROWB + fq * 16);
#pragma unroll
    for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const frag*>(As + (wm * WTM + j * 16 + fr) * ROWB + fq * 16);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = Mma<T>::run(wf[i], xf[j], acc[i][j]);
    if (kt + 1 < KT) store_tile(cur ^ 1);
    __syncthreads();
  }

  // epilogue: lane owns pixel (col) fr of tile j and channels fq*4..fq*4+3 (rows) of tile i
  const bool accumulate = flags & 1, relu = flags & 2;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = bm0 + wm * WTM + j * 16 + fr;
    if (m >= M) continue;
    const int n = m / HWm, rem = m - n * HWm;
    const int hm = rem / g.Wm, wq = rem - hm * g.Wm;
    const size_t pix = (size_t)(n * g.Hd + hm * g.dsh + g.doh) * g.Wd + (wq * g.dsw + g.dow);
    T* drow = dst + pix * g.Cd;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int co = bn0 + wn * WTN + i * 16 + fq * 4;
      if (co >= g.Cd) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
      if (bias) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + co);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += b4[r];
      }
      if constexpr (sizeof(T) == 4) {
        f32x4* p = reinterpret_cast<f32x4*>(drow + co);
        if (accumulate) { const f32x4 o = *p; for (int r = 0; r < 4; ++r) v[r] += o[r]; }
        if (relu) for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        *p = f32x4{v[0], v[1], v[2], v[3]};
      } else {
        bf16x4* p = reinterpret_cast<bf16x4*>(drow + co);
        if (accumulate) { const bf16x4 o = *p; for (int r = 0; r < 4; ++r) v[r] += (float)o[r]; }
        if (relu) for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        *p = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      }
    }
  }
}

// ---------------------------------------------------------------------------
// weight gradient: dw[cd][wtap][c] += sum_pix dy[pix][cd] * src[gather(pix,tap)][c]
// Block tile 64 (cd) x 64 (columns of the (tap,c) space), K = pixels, split over
// grid.z; partial tiles are added with f32 atomics (dw is zeroed by the caller).
// LDS holds [pixel][channel] images as loaded (channels contiguous); the MFMA
// operands need [channel][pixel], which bf16 gets from ds_read_b64_tr_b16 (hardware
// transpose read) and f32 from plain ds_read_b32 (one element per lane per MFMA).
// ---------------------------------------------------------------------------
template <typename T> struct WgradCfg;
template <> struct WgradCfg<bf16_t> { static constexpr int BKP = 32, PITCH = 64 + 8; };   // elements
template <> struct WgradCfg<float> { static constexpr int BKP = 16, PITCH = 64 + 16; };

template <typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ src,
                                                     float* __restrict__ dw, const ast_gather_t g,
                                                     const int P, const int pps) {
  constexpr int E = 16 / sizeof(T);
  constexpr int BKP = WgradCfg<T>::BKP, PITCH = WgradCfg<T>::PITCH;
  constexpr int CPR = 64 / E;            // 16-byte chunks per 64-channel row
  constexpr int RPP = 256 / CPR;         // rows loaded per pass
  constexpr int NP = BKP / RPP;          // passes per tile (== 1 for both dtypes)
  static_assert(NP == 1, "one pass per K tile");
  __shared__ __attribute__((aligned(16))) T Ys[2][BKP * PITCH];
  __shared__ __attribute__((aligned(16))) T Xs[2][BKP * PITCH];
  __shared__ int taptab[AST_MAX_TAPS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;          // 2x2 waves, 32x32 each
  const int cd0 = blockIdx.x * 64, col0 = blockIdx.y * 64;
  const int ncols = g.ntaps * g.Cs;
  const int HWm = g.Hm * g.Wm;
  const int p_begin = blockIdx.z * pps, p_end = min(P, p_begin + pps);
  if (tid < AST_MAX_TAPS) taptab[tid] = g.tap[tid];
  __syncthreads();

  // loader role: row = pixel within tile, chunk = 16-byte column chunk
  const int lrow = tid / CPR, lchunk = tid % CPR;
  const int ycd = cd0 + lchunk * E;                  // dy channel of this thread's chunk
  const bool yok = ycd < g.Cd;
  const int xcol = col0 + lchunk * E;                // column in (tap, c) space
  const bool xok = xcol < ncols;
  int dh = 0, dw_ = 0, wt = 0, xc0 = 0;
  if (xok) { const int t = xcol / g.Cs; xc0 = xcol - t * g.Cs; decode_tap(taptab[t], dh, dw_, wt); }

  uint4 yreg, xreg;
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  auto load_tile = [&](int p0) {
    const int p = p0 + lrow;
    const bool pv = p < p_end;
    yreg = (pv && yok) ? *reinterpret_cast<const uint4*>(dy + (size_t)p * g.Cd + ycd) : zero4;
    xreg = zero4;
    if (pv && xok) {
      const int n = p / HWm, rem = p - n * HWm;
      const int hm = rem / g.Wm, wq = rem - hm * g.Wm;
      const int hs = hm * g.sh + g.oh + dh, ws = wq * g.sw + g.ow + dw_;
      if ((unsigned)hs < (unsigned)g.Hs && (unsigned)ws < (unsigned)g.Ws)
        xreg = *reinterpret_cast<const uint4*>(src + ((size_t)((n * g.Hs + hs) * g.Ws + ws) * g.Cs + xc0));
    }
  };
  auto store_tile = [&](int buf) {
    *reinterpret_cast<uint4*>(&Ys[buf][lrow * PITCH + lchunk * E]) = yreg;
    *reinterpret_cast<uint4*>(&Xs[buf][lrow * PITCH + lchunk * E]) = xreg;
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, gq = lane >> 4;
  const int nk = (p_end - p_begin + BKP - 1) / BKP;
  if (nk > 0) { load_tile(p_begin); store_tile(0); }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(p_begin + (kt + 1) * BKP);
    const T* Yb = Ys[cur];
    const T* Xb = Xs[cur];
    if constexpr (sizeof(T) == 2) {
      // group gq reads rows 8gq+q (+4), lane 4q+p supplies row q, columns 4p..4p+3
      const int q = li >> 2, pcol = (li & 3) * 4;
      bf16x8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int cbase = wr * 32 + i * 16 + pcol;
        typedef __attribute__((address_space(3))) bf16x4 lds_b4;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Yb + (8 * gq + q) * PITCH + cbase));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Yb + (8 * gq + 4 + q) * PITCH + cbase));
        af[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const int xbase = wc * 32 + i * 16 + pcol;
        const bf16x4 xl = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Xb + (8 * gq + q) * PITCH + xbase));
        const bf16x4 xh = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Xb + (8 * gq + 4 + q) * PITCH + xbase));
        bf[i] = bf16x8{xl[0], xl[1], xl[2], xl[3], xh[0], xh[1], xh[2], xh[3]};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int s = 0; s < BKP / 4; ++s) {
        float af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          af[i] = Yb[(4 * s + gq) * PITCH + wr * 32 + i * 16 + li];
          bf[i] = Xb[(4 * s + gq) * PITCH + wc * 32 + i * 16 + li];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  // D[row = cd (gq*4+r)][col = column li]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + wc * 32 + j * 16 + li;
    if (col >= ncols) continue;
    const int t = col / g.Cs, c = col - t * g.Cs;
    int a, b, wtc;
    decode_tap(taptab[t], a, b, wtc);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cd = cd0 + wr * 32 + i * 16 + gq * 4 + r;
        if (cd < g.Cd) unsafeAtomicAdd(dw + ((size_t)cd * g.wtaps + wtc) * g.Cs + c, acc[i][j][r]);
      }
  }
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_igemm(const void* src, const void* wgt, const float* bias, void* dst, const ast_gather_t& g,
                 int M, int flags, hipStream_t s) {
  dim3 grid((M + BM - 1) / BM, (g.Cd + BN - 1) / BN);
  hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WM, WN>), grid, dim3(256), 0, s, (const T*)src, (const T*)wgt, bias,
                     (T*)dst, g, M, flags);
  AST_CHECK_LAUNCH();
  return 0;
}

int check_gather(const ast_gather_t* g, const char* who) {
  if (!g) AST_FAIL("%s: null geometry", who);
  if (g->Cs <= 0 || g->Cd <= 0 || (g->Cs & 7) || (g->Cd & 7)) AST_FAIL("%s: channels must be positive multiples of 8 (Cs=%d Cd=%d)", who, g->Cs, g->Cd);
  if (g->ntaps < 0 || g->ntaps > AST_MAX_TAPS || g->wtaps < 1 || g->wtaps > AST_MAX_TAPS) AST_FAIL("%s: bad tap counts %d/%d", who, g->ntaps, g->wtaps);
  if (g->N <= 0 || g->Hm <= 0 || g->Wm <= 0 || g->Hs <= 0 || g->Ws <= 0 || g->Hd <= 0 || g->Wd <= 0) AST_FAIL("%s: empty tensor", who);
  for (int t = 0; t < g->ntaps; ++t) if ((g->tap[t] >> 16) >= g->wtaps) AST_FAIL("%s: tap %d weight slice out of range", who, t);
  // destination pixels must stay inside the tensor (a fault here can reset the GPU)
  const long hmax = (long)(g->Hm - 1) * g->dsh + g->doh, wmax = (long)(g->Wm - 1) * g->dsw + g->dow;
  if (g->doh < 0 || g->dow < 0 || hmax >= g->Hd || wmax >= g->Wd) AST_FAIL("%s: destination grid exceeds tensor (%ld,%ld) vs (%d,%d)", who, hmax, wmax, g->Hd, g->Wd);
  if ((long)g->N * g->Hs * g->Ws >= (1L << 31) / 1 || (long)g->N * g->Hm * g->Wm >= (1L << 31)) AST_FAIL("%s: pixel count overflows int32", who);
  return 0;
}

}  // namespace

extern "C" int ast_igemm(const void* src, const void* wgt, const float* bias, void* dst, const ast_gather_t* gp,
                         int dtype, int flags, void* stream) {
  if (int rc = check_gather(gp, "ast_igemm")) return rc;
  if (!src || !wgt || !dst) AST_FAIL("ast_igemm: null pointer");
  const ast_gather_t g = *gp;
  const int M = g.N * g.Hm * g.Wm;
  hipStream_t s = (hipStream_t)stream;
  const long tiles128 = (long)((M + 127) / 128) * ((g.Cd + 127) / 128);
  AST_DISPATCH_T(dtype, {
    if (g.Cd > 64) {
      if (tiles128 >= 384) return launch_igemm<T, 128, 128, 2, 2>(src, wgt, bias, dst, g, M, flags, s);
      return launch_igemm<T, 64, 64, 2, 2>(src, wgt, bias, dst, g, M, flags, s);
    } else if (g.Cd > 32) {
      if (M >= 128 * 512) return launch_igemm<T, 128, 64, 2, 2>(src, wgt, bias, dst, g, M, flags, s);
      return launch_igemm<T, 64, 64, 2, 2>(src, wgt, bias, dst, g, M, flags, s);
    } else if (g.Cd > 16) {
      if (M >= 256 * 512) return launch_igemm<T, 256, 32, 4, 1>(src, wgt, bias, dst, g, M, flags, s);
      return launch_igemm<T, 64, 32, 4, 1>(src, wgt, bias, dst, g, M, flags, s);
    } else {
      if (M >= 256 * 512) return launch_igemm<T, 256, 16, 4, 1>(src, wgt, bias, dst, g, M, flags, s);
      return launch_igemm<T, 64, 16, 4, 1>(src, wgt, bias, dst, g, M, flags, s);
    }
  });
  return 0;
}

extern "C" int ast_wgrad(const void* dy, const void* src, float* dw, const ast_gather_t* gp, int dtype, void* stream) {
  if (int rc = check_gather(gp, "ast_wgrad")) return rc;
  if (!dy || !src || !dw) AST_FAIL("ast_wgrad: null pointer");
  const ast_gather_t g = *gp;
  if (g.ntaps == 0) return 0;
  const int P = g.N * g.Hm * g.Wm;
  const int tiles = ((g.Cd + 63) / 64) * ((g.ntaps * g.Cs + 63) / 64);
  const int bkp = dtype == AST_BF16 ? 32 : 16;
  int nsplit = max(1, min((P + 8 * bkp - 1) / (8 * bkp), (1024 + tiles - 1) / tiles));
  int pps = (P + nsplit - 1) / nsplit;
  pps = (pps + bkp - 1) / bkp * bkp;
  nsplit = (P + pps - 1) / pps;
  dim3 grid((g.Cd + 63) / 64, (g.ntaps * g.Cs + 63) / 64, nsplit);
  hipStream_t s = (hipStream_t)stream;
  AST_DISPATCH_T(dtype, {
    hipLaunchKernelGGL((wgrad_kernel<T>), grid, dim3(256), 0, s, (const T*)dy, (const T*)src, dw, g, P, pps);
  });
  AST_CHECK_LAUNCH();
  return 0;
}
