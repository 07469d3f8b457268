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

// LDS image: unpadded 64-byte rows (4 chunks of 16 B); chunk c of row r lives in slot (c + 2*((r>>3)&1)) & 3.
// With ds_read_b128's 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (lane = row + 16*chunk) this
// rotation puts the 16 lanes of every group on 16 distinct 16-byte slots of the 256-byte bank row
// (conflict-free), and 8 consecutive store lanes (2 rows x 4 chunks) still cover 128 contiguous bytes.
__device__ __forceinline__ int swz(int row, int c) { return ((c + (((row >> 3) & 1) << 1)) & 3) << 4; }

// ---- epilogue shared by igemm_kernel and igemm_direct_kernel -----------------------------------------------------------
template <typename T>
struct EpiCtx {
  T* dst; const float* bias; float* ws; const T* bn_x; const float* bn_scale; const float* bn_shift;
  int HWm; float rcp_hw, rcp_w; bool accumulate, relu, stats, bstats, bn_relu;
};

// The same for a known destination pixel index `pix` (pixels of the dst tensor, not bytes).
// MODE (compile time): 0 plain, 1 fused BatchNorm forward statistics, 2 fused BatchNorm backward sums.  The callers branch ONCE
// (wave-uniform) into the specialisation: with the three variants behind run-time flags in one body the compiler
// if-converted parts of them and a wave issued ~60 VALU per 4-value block whatever the flags (1 015 VALU per 144 MFMAs on
// the 64-channel patch kernel without any statistics requested).
template <typename T, int TN, int MODE>
__device__ __forceinline__ void epi_store_m(const EpiCtx<T>& ec, const ast_gather_t& g, const f32x4 (&col)[TN], const size_t pix, const int co0,
                                            float (&st1)[TN][4], float (&st2)[TN][4]) {
  T* drow = ec.dst + pix * g.Cd;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int co = co0 + i * 16;
    if (co >= g.Cd) continue;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = col[i][r];
    if (ec.bias) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(ec.bias + co);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += b4[r];
    }
    if constexpr (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float q = (float)(T)v[r];                      // the value as stored
        st1[i][r] += q; st2[i][r] += q * q;
      }
    }
    if constexpr (MODE == 2) {
      float xv[4];
      if constexpr (sizeof(T) == 2) {
        const bf16x4 xq = *reinterpret_cast<const bf16x4*>(ec.bn_x + pix * g.Cd + co);
#pragma unroll
        for (int r = 0; r < 4; ++r) xv[r] = (float)xq[r];
      } else {
        const f32x4 xq = *reinterpret_cast<const f32x4*>(ec.bn_x + pix * g.Cd + co);
#pragma unroll
        for (int r = 0; r < 4; ++r) xv[r] = xq[r];
      }
      // (loading the coefficients once per channel tile ahead of the pixel loop measured slower: register pressure)
      const f32x4 sc4 = *reinterpret_cast<const f32x4*>(ec.bn_scale + co), sf4 = *reinterpret_cast<const f32x4*>(ec.bn_shift + co);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float q = (float)(T)v[r];
        const float dz = (!ec.bn_relu || __builtin_fmaf(xv[r], sc4[r], sf4[r]) > 0.f) ? q : 0.f;
        st1[i][r] += dz; st2[i][r] += dz * xv[r];
      }
    }
    store4<T>(drow + co, v, ec.accumulate, ec.relu);
  }
}
template <typename T, int TN>
__device__ __forceinline__ void epi_store(const EpiCtx<T>& ec, const ast_gather_t& g, const f32x4 (&col)[TN], const size_t pix, const int co0,
                                          float (&st1)[TN][4], float (&st2)[TN][4]) {
  if (ec.stats) epi_store_m<T, TN, 1>(ec, g, col, pix, co0, st1, st2);
  else if (ec.bstats) epi_store_m<T, TN, 2>(ec, g, col, pix, co0, st1, st2);
  else epi_store_m<T, TN, 0>(ec, g, col, pix, co0, st1, st2);
}

// One destination pixel m: the lane owns channels co0 + i*16 .. +3 of channel tile i (col[i] = their accumulators).
// Adds bias, accumulates the fused BatchNorm forward / backward sums of the values as stored, stores.
template <typename T, int TN>
__device__ __forceinline__ void epi_pixel(const EpiCtx<T>& ec, const ast_gather_t& g, const f32x4 (&col)[TN], const int m, const int co0,
                                          float (&st1)[TN][4], float (&st2)[TN][4]) {
  const int n = fdiv(m, ec.HWm, ec.rcp_hw), rem = m - n * ec.HWm;
  const int hm = fdiv(rem, g.Wm, ec.rcp_w), wq = rem - hm * g.Wm;
  epi_store<T, TN>(ec, g, col, (size_t)(n * g.Hd + hm * g.dsh + g.doh) * g.Wd + (wq * g.dsw + g.dow), co0, st1, st2);
}

// The tile's fused sums -> the slot table.  Reduce over the 16 pixels of the lane group, then spread the (tile, channel,
// sum|sumsq) values over the 16 lanes so that ONE atomic instruction per pair of channel tiles carries them all (an
// atomic costs its issue slot whatever the number of active lanes: 4-lane atomics per value made this slower than the
// separate pass).  cbase = first channel of the wave's channel tiles.
// red != nullptr: the workgroup's FOUR waves hold the same channels (they split the pixels): their values are summed through
// `red` (LDS, (TN+1)/2 * 256 floats, free at this point up to a barrier) and wave 0 alone issues the atomics.  Same-address
// f32 atomics serialise, and a 0.7 M-pixel layer has 5 382 tiles: per-wave atomics put 336 adds behind each other on every
// address of the 64-slot table, 1 345 on a per-image table.  Every thread of the workgroup must arrive.
template <int TN>
__device__ __forceinline__ void epi_flush(float* ws, const bool bstats, const int Cd, float (&st1)[TN][4], float (&st2)[TN][4], const int tix,
                                          const int cbase, const int fr, const int fq, float* red = nullptr) {
  const int KS = bstats ? 3 : 2;                               // floats per channel in the slot table
  float* slot = ws + (size_t)(tix < 0 ? ~tix : (tix & 63)) * Cd * KS;      // tix < 0: ~tix is the slot itself (per-image tables)
  constexpr int NP = (TN + 1) / 2;
  float val[NP];
#pragma unroll
  for (int i0 = 0; i0 < TN; i0 += 2) {
    // value q = ii*8 + r*2 + which (which: 0 = st1, 1 = st2) ends, summed over the 16 pixels, in lane fr = q
    float v[16];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        constexpr int NT = TN;                                  // an odd TN leaves the second half of the last pair empty
        const int it = i0 + ii < NT ? i0 + ii : 0;
        v[ii * 8 + r * 2] = i0 + ii < NT ? st1[it][r] : 0.f;
        v[ii * 8 + r * 2 + 1] = i0 + ii < NT ? st2[it][r] : 0.f;
      }
    val[i0 >> 1] = row16_transpose_sum(v, fr);
  }
  if (red) {
    const int tid = threadIdx.x, lane = tid & 63;
    __syncthreads();                                            // the other waves may still read the last operand tiles
#pragma unroll
    for (int p = 0; p < NP; ++p) red[p * 256 + tid] = val[p];
    __syncthreads();
    if (tid >= 64) return;
#pragma unroll
    for (int p = 0; p < NP; ++p) val[p] = (red[p * 256 + lane] + red[p * 256 + 64 + lane]) + (red[p * 256 + 128 + lane] + red[p * 256 + 192 + lane]);
  }
#pragma unroll
  for (int i0 = 0; i0 < TN; i0 += 2) {
    const int ii = fr >> 3, r = (fr >> 1) & 3, which = fr & 1;
    const int co = cbase + (i0 + ii) * 16 + fq * 4 + r;
    if (i0 + ii < TN && co < Cd) unsafeAtomicAdd(slot + (size_t)co * KS + which, val[i0 >> 1]);
  }
}
template <typename T, int TN>
__device__ __forceinline__ void epi_flush(const EpiCtx<T>& ec, const ast_gather_t& g, float (&st1)[TN][4], float (&st2)[TN][4], const int tix,
                                          const int cbase, const int fr, const int fq, float* red = nullptr) {
  epi_flush<TN>(ec.ws, ec.bstats, g.Cd, st1, st2, tix, cbase, fr, fq, red);
}

// KCH = 16-byte chunks per row staged per barrier (4 or 8; 8 = two 64-byte sub-tiles).
// grid.z = split-K slices; with more than one slice the f32 partial tiles are added into `ws`
// ([M][Cd]) with atomics and splitk_finish_kernel applies bias / accumulate / ReLU / cast.
// Operands are read with buffer_load_dwordx4: masked lanes get an out-of-range offset and the
// hardware returns zeros (no select, no 64-bit address arithmetic).
// UT ("uniform tap"): Cs/E is a multiple of KCH, so all KCH chunks of a K tile belong to one tap.  The tap decode and
// the source / weight byte deltas are then wave-uniform and live in SGPRs; per load a lane only tests one bit of a
// per-row tap-validity mask built once, and adds the scalar delta.  (The general path costs ~7 VALU per MFMA, which
// is what bounds the kernel: the deep layers ran at the same speed for every tile shape and K split.)
template <typename T, int BM, int BN, int WM, int WN, int KCH, int D, int KG, bool UT>
__global__ __launch_bounds__(256 * KG) void igemm_kernel(const T* __restrict__ src, const T* __restrict__ wgt,
                                                     const float* __restrict__ bias, T* __restrict__ dst,
                                                     const ast_gather_t g, const int M, const int flags,
                                                     float* __restrict__ ws, const int kt_per_split, const int cpc_shift,
                                                     const unsigned src_bytes, const unsigned wgt_bytes,
                                                     const float rcp_hw, const float rcp_w,
                                                     const T* __restrict__ bn_x, const float* __restrict__ bn_scale,
                                                     const float* __restrict__ bn_shift) {
  constexpr int E = 16 / sizeof(T);
  constexpr int ES = sizeof(T);
  constexpr int NSUB = KCH / 4;
  constexpr int RPP = 256 / KCH;             // rows staged per pass
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int AI = BM / RPP, BI = (BN + RPP - 1) / RPP;
  constexpr int SUBB = (BM + BN) * 64;       // bytes of one sub-tile (A rows then B rows)
  constexpr unsigned OOB = 0x80000000u, OOBH = 0x40000000u;
  static_assert(WM * WN == 4 && BM % RPP == 0 && WTM % 16 == 0 && WTN % 16 == 0 && D % 2 == 0, "tile");
  using frag = typename Mma<T>::frag;

  // KG > 1: the workgroup has KG groups of 4 waves; group kg streams K tiles kt0+kg, kt0+kg+KG, ... through its own
  // LDS buffers and the partial accumulators are summed through LDS at the end.  For the deep layers (M of a few
  // thousand rows, K up to 4608) this puts KG times more loads in flight per CU and cuts the serial K loop by KG,
  // without the atomics + finish pass of a grid-level split.
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];  // KG * (2 * NSUB * SUBB), then 16 ints
  int* taptab = reinterpret_cast<int*>(lds_all + KG * 2 * NSUB * SUBB);
  const int kg = KG > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;   // wave-uniform
  unsigned char* lds = lds_all + kg * (2 * NSUB * SUBB);

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int cc = tid % KCH, r0 = tid / KCH;
  const int csub = cc >> 2, cch = cc & 3;
  // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2), so XCD x takes
  // the x-th contiguous eighth of the tiles in N-major order: its L2 then holds one slice of the weights and one
  // contiguous band of source rows (with its 3x3 halo) instead of a sample of everything.  Speed only; any
  // placement gives the same result.
  // flags bit 6: image-aligned tiles (no tile straddles two images), so that the fused statistics can go to per-IMAGE
  // slots (InstanceNorm2d: style_encoder.py:69); M_end = end of the tile's image, else M
  const int HWm_ = g.Hm * g.Wm;
  const bool per_image = flags & 64;
  const int MTI = (HWm_ + BM - 1) / BM;
  const int MT = per_image ? g.N * MTI : (M + BM - 1) / BM, NT = (g.Cd + BN - 1) / BN;
  const int chunk = gridDim.x >> 3;
  const int tix = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (tix >= MT * NT) return;
  const int ntile = tix / MT;
  const int mtile = tix - ntile * MT;
  const int img = per_image ? mtile / MTI : 0;
  const int bm0 = per_image ? img * HWm_ + (mtile - img * MTI) * BM : mtile * BM, bn0 = ntile * BN;
  const int M_end = per_image ? (img + 1) * HWm_ : M;
  const int cpc = g.Cs / E;
  const int nchunks = g.ntaps * cpc;
  const int KT = (nchunks + KCH - 1) / KCH;
  const int kt0 = blockIdx.z * kt_per_split, kt1 = min(KT, kt0 + kt_per_split);
  const int HWm = g.Hm * g.Wm;
  const __amdgpu_buffer_rsrc_t srcR = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wgtR = __builtin_amdgcn_make_buffer_rsrc((void*)wgt, 0, wgt_bytes, 0x00020000);

#pragma unroll
  for (int t = 0; t < AST_MAX_TAPS; ++t)
    if ((int)threadIdx.x == t) taptab[t] = g.tap[t];   // static index: a dynamic one would spill the by-value struct to scratch

  int roff[AI], rhs0[AI], rws0[AI];              // source byte offset of the row's base pixel (negative for halo rows)
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = bm0 + r0 + RPP * i;
    const bool valid = m < M_end;
    const int mm = valid ? m : 0;
    const int n = fdiv(mm, HWm, rcp_hw), rem = mm - n * HWm;
    const int hm = fdiv(rem, g.Wm, rcp_w), wq = rem - hm * g.Wm;
    rhs0[i] = valid ? hm * g.sh + g.oh : -(1 << 20);        // an invalid row fails every bounds test below
    rws0[i] = wq * g.sw + g.ow;
    roff[i] = (((n * g.Hs + rhs0[i]) * g.Ws + rws0[i]) * g.Cs) * ES;
  }
  unsigned boff[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int row = r0 + RPP * i, co = bn0 + row;
    boff[i] = (row < BN && co < g.Cd) ? (unsigned)(co * g.wtaps * g.Cs * ES) : OOB;
  }
  __syncthreads();
  unsigned rmask[AI];                            // UT: bit t = tap t of this row lies inside the source image
  if constexpr (UT) {
#pragma unroll
    for (int i = 0; i < AI; ++i) { rmask[i] = 0; roff[i] += cc * 16; }
#pragma unroll
    for (int i = 0; i < BI; ++i) boff[i] = boff[i] == OOB ? OOBH : boff[i] + cc * 16;
#pragma unroll
    for (int t = 0; t < AST_MAX_TAPS; ++t) {
      if (t < g.ntaps) {
        int dh, dw, wt;
        decode_tap(__builtin_amdgcn_readfirstlane(taptab[t]), dh, dw, wt);
#pragma unroll
        for (int i = 0; i < AI; ++i)
          if ((unsigned)(rhs0[i] + dh) < (unsigned)g.Hs && (unsigned)(rws0[i] + dw) < (unsigned)g.Ws) rmask[i] |= 1u << t;
      }
    }
    // keep the per-row bases as opaque registers: otherwise the optimiser re-derives them from (n, h, w) inside the
    // K loop to save VGPRs, which puts two quarter-rate v_mul_lo_u32 per load back into it
#pragma unroll
    for (int i = 0; i < AI; ++i) asm volatile("" : "+v"(roff[i]), "+v"(rmask[i]));
#pragma unroll
    for (int i = 0; i < BI; ++i) asm volatile("" : "+v"(boff[i]));
  }

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // D-stage register prefetch: the loads of K tile k+D are issued before tile k is consumed, so up to D
  // tiles are in flight per workgroup (the deep layers run ~1 workgroup per CU and are otherwise
  // latency-bound).  The K loop is unrolled by D so every stage index is a compile-time constant.
  u32x4 areg[D][AI], breg[D][BI];

  const int nkt = kt1 - kt0;
  auto load_tile = [&](int j, u32x4 (&ar)[AI], u32x4 (&br)[BI]) __attribute__((always_inline)) {
    const bool jval = j < nkt;
    const int kt = kt0 + j;
    if constexpr (UT) {
      const int kc0 = kt * KCH;                              // everything down to the per-row select is scalar
      const bool kval = jval;
      const int t = kval ? (cpc_shift >= 0 ? (kc0 >> cpc_shift) : kc0 / cpc) : 0;
      const int c0b = (kc0 - t * cpc) * 16;
      int dh, dw, wt;
      decode_tap(__builtin_amdgcn_readfirstlane(taptab[t]), dh, dw, wt);
      const int sdelta = (dh * g.Ws + dw) * g.Cs * ES + c0b;
      const unsigned tbit = kval ? (1u << t) : 0u;
#pragma unroll
      for (int i = 0; i < AI; ++i)
        ar[i] = __builtin_amdgcn_raw_buffer_load_b128(srcR, (rmask[i] & tbit) ? (unsigned)(roff[i] + sdelta) : OOB, 0, 0);
      const unsigned swoff = kval ? (unsigned)(wt * g.Cs * ES + c0b) : OOBH;
#pragma unroll
      for (int i = 0; i < BI; ++i) br[i] = __builtin_amdgcn_raw_buffer_load_b128(wgtR, boff[i] + swoff, 0, 0);
      return;
    }
    const int kc = kt * KCH + cc;
    const bool kval = kc < nchunks && jval;
    const int t = kval ? (cpc_shift >= 0 ? (kc >> cpc_shift) : kc / cpc) : 0;
    const int c0 = (kc - t * cpc) * E;
    int dh, dw, wt;
    decode_tap(taptab[t], dh, dw, wt);
    const int delta = ((dh * g.Ws + dw) * g.Cs + c0) * ES;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const bool ok = kval && (unsigned)(rhs0[i] + dh) < (unsigned)g.Hs && (unsigned)(rws0[i] + dw) < (unsigned)g.Ws;
      ar[i] = __builtin_amdgcn_raw_buffer_load_b128(srcR, ok ? (unsigned)(roff[i] + delta) : OOB, 0, 0);
    }
    const unsigned woff = (unsigned)((wt * g.Cs + c0) * ES);
#pragma unroll
    for (int i = 0; i < BI; ++i)
      br[i] = __builtin_amdgcn_raw_buffer_load_b128(wgtR, (kval && boff[i] != OOB) ? boff[i] + woff : OOB, 0, 0);
  };
  auto store_tile = [&](int buf, const u32x4 (&ar)[AI], const u32x4 (&br)[BI]) __attribute__((always_inline)) {
    unsigned char* base = lds + (buf * NSUB + csub) * SUBB;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int row = r0 + RPP * i;
      *reinterpret_cast<u32x4*>(base + row * 64 + swz(row, cch)) = ar[i];
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int row = r0 + RPP * i;
      if (row < BN) *reinterpret_cast<u32x4*>(base + (BM + row) * 64 + swz(row, cch)) = br[i];
    }
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int fsw = swz(fr, fq);                             // fragment rows are 16-aligned: (row>>3)&1 == (fr>>3)&1
  int aoffs[TM], boffs[TN];
#pragma unroll
  for (int j = 0; j < TM; ++j) aoffs[j] = (wm * WTM + j * 16 + fr) * 64 + fsw;
#pragma unroll
  for (int i = 0; i < TN; ++i) boffs[i] = (BM + wn * WTN + i * 16 + fr) * 64 + fsw;

#pragma unroll
  for (int st = 0; st < D; ++st) load_tile(kg + st * KG, areg[st], breg[st]);
  // every group runs the same number of iterations (its tail tiles are masked to zeros) so the
  // workgroup-wide barrier below is reached uniformly
  const int niter = (kt1 - kt0 + KG - 1) / KG;
  for (int itb = 0; itb < niter; itb += D) {
#pragma unroll
    for (int st = 0; st < D; ++st) {
      const int it = itb + st;
      if (it < niter) {                                      // uniform per workgroup
        const int j = kg + it * KG;
        const int cur = st & 1;                               // itb is a multiple of D (even): it & 1 == st & 1, a constant
        store_tile(cur, areg[st], breg[st]);                 // waits (counted vmcnt) for this stage's loads only
        load_tile(j + D * KG, areg[st], breg[st]);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < NSUB; ++ks) {
          const unsigned char* base = lds + (cur * NSUB + ks) * SUBB;
          frag wf[TN], xf[TM];
#pragma unroll
          for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const frag*>(base + boffs[i]);
#pragma unroll
          for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const frag*>(base + aoffs[j]);
#pragma unroll
          for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = Mma<T>::run(wf[i], xf[j], acc[i][j]);
        }
      }
    }
  }

  if constexpr (KG > 1) {                                     // sum the K-groups' partial tiles through LDS
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(lds_all);
    if (kg > 0) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) red[(((kg - 1) * TN + i) * TM + j) * 256 + tid] = acc[i][j];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int q = 0; q < KG - 1; ++q)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const f32x4 t = red[((q * TN + i) * TM + j) * 256 + tid];
          acc[i][j][0] += t[0]; acc[i][j][1] += t[1]; acc[i][j][2] += t[2]; acc[i][j][3] += t[3];
        }
  }
  // epilogue: lane owns pixel (col) fr of tile j and channels fq*4..fq*4+3 (rows) of tile i
  const bool accumulate = flags & 1, relu = flags & 2;
  const bool split = gridDim.z > 1;
  // flags bit 3: `ws` is a [64][Cd][2] table of per-channel (sum, sum of squares) slots and the tile adds the
  // statistics of the values it STORES (BatchNorm2d's batch statistics without a second pass over the output);
  // slot = tile index mod 64 keeps the f32 atomics off a single address per channel
  const bool stats = (flags & 8) && !split;
  // flags bit 4: this launch is the data gradient that produces dy of a BatchNorm(+ReLU) layer whose input x (bn_x, same
  // geometry as dst) and forward coefficients are given: the tile adds the layer's backward sums (sum dz, sum dz*x, with
  // dz = dy * [fma(x, scale, shift) > 0]) of the values it stores into the [64][Cd][3] slot table `ws` -- the separate
  // pass over dy and x (ast_norm_bwd_sums) disappears.  bit 5: the layer has no ReLU (mask = 1).
  const bool bstats = (flags & 16) && !split;
  const bool bn_relu = !(flags & 32);
  EpiCtx<T> ec{dst, bias, ws, bn_x, bn_scale, bn_shift, HWm, rcp_hw, rcp_w, accumulate, relu, stats, bstats, bn_relu};
  float st1[TN][4], st2[TN][4];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { st1[i][r] = 0.f; st2[i][r] = 0.f; }
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = bm0 + wm * WTM + j * 16 + fr;
    if (m >= M_end) continue;
    if (split) {
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int co = bn0 + wn * WTN + i * 16 + fq * 4;
        if (co >= g.Cd) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(ws + (size_t)m * g.Cd + co + r, acc[i][j][r]);
      }
      continue;
    }
    f32x4 col[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) col[i] = acc[i][j];
    epi_pixel<T, TN>(ec, g, col, m, bn0 + wn * WTN + fq * 4, st1, st2);
  }
  // WN == 1: the four waves split the tile's pixels and share its channels -> one atomic per value per WORKGROUP
  float* const red = (KG == 1 && WN == 1) ? reinterpret_cast<float*>(lds_all) : nullptr;
  if (stats || bstats) epi_flush<T, TN>(ec, g, st1, st2, per_image ? ~img : tix, bn0 + wn * WTN, fr, fq, red);
}

// ---- narrow layers: operands straight from L1/L2 into MFMA fragments, no LDS ----------------------------------------
// For few output channels and a short K (the first encoder / last decoder layers: 2..16 channels on up to 2.4 M pixels) the
// LDS-staged kernel spends its time in staging and barriers for 3..9 MFMA steps per tile (92 us for a launch whose
// operands are 113 MB).  The MFMA fragment layout is exactly the gather: lane (fr, fq) of a pixel tile needs the 16-byte
// chunk kc = 4*ks + fq of pixel fr -- one buffer load.  So: the weights of the wave's channel tiles for ALL K steps live
// in registers (TN * NKS fragments, loaded once), every wave walks `jt` pixel tiles of 16, and per tile issues its NKS
// chunk loads (next tile's are in flight while this tile's MFMAs run), TN * NKS MFMAs, and the shared epilogue.
// Per-lane tap decode (the 4 chunks of a K step may sit in different taps) is done once, outside the pixel loop.
template <typename T, int TN, int NKS>
__global__ __launch_bounds__(256) void igemm_direct_kernel(const T* __restrict__ src, const T* __restrict__ wgt,
                                                           const float* __restrict__ bias, T* __restrict__ dst,
                                                           const ast_gather_t g, const int M, const int flags,
                                                           float* __restrict__ ws, const int cpc_shift,
                                                           const unsigned src_bytes, const unsigned wgt_bytes,
                                                           const float rcp_hw, const float rcp_w,
                                                           const T* __restrict__ bn_x, const float* __restrict__ bn_scale,
                                                           const float* __restrict__ bn_shift, const int jt) {
  constexpr int E = 16 / sizeof(T), ES = sizeof(T);
  constexpr unsigned OOB = 0x80000000u;
  using frag = typename Mma<T>::frag;
  __shared__ int taptab[AST_MAX_TAPS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int BMW = jt * 16, BM = 4 * BMW;                     // pixels per wave / per workgroup
  const int MT = (M + BM - 1) / BM, NT = (g.Cd + TN * 16 - 1) / (TN * 16);
  const int chunk = gridDim.x >> 3;                          // XCD-aware tile order, as igemm_kernel
  const int tix = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (tix >= MT * NT) return;
  const int ntile = tix / MT;
  const int bm0 = (tix - ntile * MT) * BM + wave * BMW, bn0 = ntile * TN * 16;
#pragma unroll
  for (int t = 0; t < AST_MAX_TAPS; ++t)
    if ((int)threadIdx.x == t) taptab[t] = g.tap[t];
  __syncthreads();
  const int cpc = g.Cs / E, nchunks = g.ntaps * cpc;
  const int nks = (nchunks + 3) >> 2;
  const int HWm = g.Hm * g.Wm;
  const __amdgpu_buffer_rsrc_t srcR = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wgtR = __builtin_amdgcn_make_buffer_rsrc((void*)wgt, 0, wgt_bytes, 0x00020000);

  int sdelta[NKS], tdh[NKS], tdw[NKS];
  unsigned woff[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int kc = ks * 4 + fq;
    const bool kval = kc < nchunks;
    const int t = kval ? (cpc_shift >= 0 ? (kc >> cpc_shift) : kc / cpc) : 0;
    const int c0b = (kc - t * cpc) * 16;
    int dh, dw, wt;
    decode_tap(taptab[t], dh, dw, wt);
    sdelta[ks] = (dh * g.Ws + dw) * g.Cs * ES + c0b;
    tdh[ks] = kval ? dh : (1 << 20);                         // an invalid chunk fails every bounds test
    tdw[ks] = dw;
    woff[ks] = kval ? (unsigned)(wt * g.Cs * ES + c0b) : OOB;
  }
  frag wreg[TN][NKS];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int co = bn0 + i * 16 + fr;
    const unsigned rowoff = co < g.Cd ? (unsigned)(co * g.wtaps * g.Cs * ES) : OOB;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      wreg[i][ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(wgtR, (rowoff != OOB && woff[ks] != OOB) ? rowoff + woff[ks] : OOB, 0, 0));
  }

  const bool accumulate = flags & 1, relu = flags & 2, stats = flags & 8, bstats = flags & 16, bn_relu = !(flags & 32);
  EpiCtx<T> ec{dst, bias, ws, bn_x, bn_scale, bn_shift, HWm, rcp_hw, rcp_w, accumulate, relu, stats, bstats, bn_relu};
  float st1[TN][4], st2[TN][4];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { st1[i][r] = 0.f; st2[i][r] = 0.f; }

  auto load_pix = [&](int j, u32x4 (&xr)[NKS], int& dpix) __attribute__((always_inline)) {
    const int m = bm0 + j * 16 + fr;
    const bool valid = j < jt && m < M;
    const int mm = valid ? m : 0;
    const int n = fdiv(mm, HWm, rcp_hw), rem = mm - n * HWm;
    const int hm = fdiv(rem, g.Wm, rcp_w), wq = rem - hm * g.Wm;
    const int hs0 = valid ? hm * g.sh + g.oh : -(1 << 20);
    const int ws0 = wq * g.sw + g.ow;
    const int roff = (((n * g.Hs + (valid ? hs0 : 0)) * g.Ws + ws0) * g.Cs) * ES;
    dpix = (n * g.Hd + hm * g.dsh + g.doh) * g.Wd + (wq * g.dsw + g.dow);   // the tile's destination pixel, decoded once
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const bool ok = (unsigned)(hs0 + tdh[ks]) < (unsigned)g.Hs && (unsigned)(ws0 + tdw[ks]) < (unsigned)g.Ws;
      xr[ks] = __builtin_amdgcn_raw_buffer_load_b128(srcR, ok ? (unsigned)(roff + sdelta[ks]) : OOB, 0, 0);
    }
  };
  auto compute = [&](int j, const u32x4 (&xr)[NKS], const int dpix) __attribute__((always_inline)) {
    f32x4 acc[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if (ks < nks) {                                        // uniform
        const frag xf = __builtin_bit_cast(frag, xr[ks]);
#pragma unroll
        for (int i = 0; i < TN; ++i) acc[i] = Mma<T>::run(wreg[i][ks], xf, acc[i]);
      }
    }
    const int m = bm0 + j * 16 + fr;
    if (m < M) epi_store<T, TN>(ec, g, acc, (size_t)dpix, bn0 + fq * 4, st1, st2);
  };

  u32x4 xa[NKS], xb[NKS];
  int pa, pb;
  load_pix(0, xa, pa);
  for (int j = 0; j < jt; j += 2) {                          // uniform trip count
    load_pix(j + 1, xb, pb);
    compute(j, xa, pa);
    load_pix(j + 2, xa, pa);
    if (j + 1 < jt) compute(j + 1, xb, pb);
  }
  __shared__ float red[(TN + 1) / 2 * 256];                  // the four waves split the pixels: one atomic per value per workgroup
  if (stats || bstats) epi_flush<T, TN>(ec, g, st1, st2, tix, bn0, fr, fq, red);
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(float* __restrict__ ws, const float* __restrict__ bias,
                                                            T* __restrict__ dst, const ast_gather_t g, const int M, const int flags) {
  const int c4 = g.Cd >> 2;
  const int HWm = g.Hm * g.Wm;
  const size_t total = (size_t)M * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / c4), co = (int)(i % c4) * 4;
    const int n = m / HWm, rem = m - n * HWm;
    const int hm = rem / g.Wm, wq = rem - hm * g.Wm;
    const size_t pix = (size_t)(n * g.Hd + hm * g.dsh + g.doh) * g.Wd + (wq * g.dsw + g.dow);
    f32x4* wp = reinterpret_cast<f32x4*>(ws + (size_t)m * g.Cd + co);
    const f32x4 a = *wp;
    *wp = f32x4{0.f, 0.f, 0.f, 0.f};                  // the workspace is handed back zeroed (no memset per launch)
    float v[4] = {a[0], a[1], a[2], a[3]};
    if (bias) for (int r = 0; r < 4; ++r) v[r] += bias[co + r];
    store4<T>(dst + pix * g.Cd + co, v, flags & 1, flags & 2);
  }
}

// ---------------------------------------------------------------------------
// weight gradient: dw[cd][wtap][c] += sum_pix dy[pix][cd] * src[gather(pix,tap)][c]
// A workgroup owns BMW output channels x NCT*16 columns of the (tap, channel) space -- ALL columns
// when they fit (<= 320), so dy is read once instead of once per 64-column tile -- and a slice of the
// pixels (grid.z); partial tiles are added with f32 atomics (dw is zeroed by the caller).
// Per K tile (BKP pixels) a thread decodes ONE pixel row and fetches its strided set of 16-byte
// chunks with buffer loads (hardware zero-fill outside the image).  LDS holds [pixel][channel]
// images; the MFMA operands need [channel][pixel]: bf16 through ds_read_b64_tr_b16 (hardware
// transpose), f32 through ds_read_b32 (one element per lane per MFMA).
// ---------------------------------------------------------------------------
#ifdef AST_STAMPS
__device__ unsigned long long ast_wg_stamps[4096 * 8];      // wgrad_kernel phase stamps (tools/wgrad_stamps.py)
__device__ long long ast_wg_phase[4096 * 4];                // K-loop sub-phases: load issue, reads + MFMA, second barrier, trips
#define WG_STAMP(k) do { if (threadIdx.x == 0 && tix < 4096) ast_wg_stamps[tix * 8 + (k)] = (k) >= 6 ? wall_clock64() : __builtin_readcyclecounter(); } while (0)
#else
#define WG_STAMP(k) do { } while (0)
#endif
// Gradient replicas (ast_wgrad_rep): workgroups of pixel slice z add their tile into copy z % nrep of dw (copies nrep_stride
// floats apart); whoever reads dw sums the copies (ast_weight_grads_flush_t).  Same-address f32 atomics serialise at ~155 ns
// each wherever the address lives (tools/micro/l2atomic.hip: spreading the lines over channels or doing them at L2 level
// changes nothing), so the flush of 85 workgroups per tile took 14 us of a 40 us launch (tools/wgrad_stamps.py); with 8
// copies the chains are 11 deep.
static thread_local int g_wg_nrep = 1;
static thread_local long g_wg_rep_stride = 0;
template <typename T> struct WgradCfg;
template <> struct WgradCfg<bf16_t> { static constexpr int BKP = 64, PAD = 16; };   // elements: row pitch = 32 B x odd for rows that are multiples of 64 B (see SWZ)
template <> struct WgradCfg<float> { static constexpr int BKP = 32, PAD = 16; };

// PG pixel groups of 4 waves per workgroup, as in wgrad_halo_kernel: group pg takes every PG-th K tile of the workgroup's pixel
// slice through its own LDS staging, partial tiles are summed through LDS, one atomic flush per workgroup.
template <typename T, int BMW, int NCT, int PG>
__global__ __launch_bounds__(256 * PG) void wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ src,
                                                     float* __restrict__ dw, const ast_gather_t g,
                                                     const int P, const int pps, const unsigned dy_bytes,
                                                     const unsigned src_bytes, const float rcp_hw, const float rcp_w,
                                                     const int gx, const int gy, const int gz, const int nrep, const long rep_stride) {
  constexpr int E = 16 / sizeof(T), ES = sizeof(T);
  constexpr int BKP = WgradCfg<T>::BKP, PAD = WgradCfg<T>::PAD;
  constexpr int BNW = NCT * 16;
  constexpr int TPR = 256 / BKP;                    // threads per pixel row
  constexpr int CPX = BNW / E, CPY = BMW / E;       // 16-byte chunks per row
  constexpr int NXI = (CPX + TPR - 1) / TPR, NYI = (CPY + TPR - 1) / TPR;
  constexpr int PX = BNW + PAD, PY = BMW + PAD;     // LDS pitches (elements)
  constexpr bool SWZ = sizeof(T) == 2 && (CPX % 2 == 0) && (CPY % 2 == 0);
  constexpr int RT = BMW / 16;                      // row (cd) tiles, all handled by every wave
  constexpr int CTW = (NCT + 3) / 4;                // column tiles per wave
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char wl_all[];
  constexpr int GROUP_BYTES = (int)sizeof(T) * BKP * (PY + PX);
  constexpr int RED = PG > 1 ? RT * CTW * 256 * 16 : 0;                        // one group's partial tile in the final LDS reduction
  constexpr int TAP_OFF = (PG * GROUP_BYTES > RED ? PG * GROUP_BYTES : RED);
  const int pg = PG > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;
  unsigned char* wl = wl_all + pg * GROUP_BYTES;
  T* Ys = reinterpret_cast<T*>(wl);
  T* Xs = Ys + BKP * PY;
  int* taptab = reinterpret_cast<int*>(wl_all + TAP_OFF);

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  // XCD-aware order (workgroups b and b+8 share an L2): XCD x takes the x-th contiguous eighth of the tiles in
  // (row tile, pixel slice, column tile) order, so an L2 holds ONE channel slice of dy (deep layers: Cd/64 >= 8 row
  // tiles) or ONE band of pixels (shallow layers: many pixel slices) instead of a sample of the whole layer.  Measured
  // before: 64 MB of fabric reads per launch for 25 MB algorithmic (profiles/r01/d_pmc_traffic.json).
  const int chunk = gridDim.x >> 3;
  const int tix = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (tix >= gx * gy * gz) return;
  WG_STAMP(0); WG_STAMP(7);
  const int bx = tix / (gz * gy), bz = (tix / gy) % gz, by = tix % gy;
  const int cd0 = bx * BMW, col0 = by * BNW;
  const int ncols = g.ntaps * g.Cs;
  const int HWm = g.Hm * g.Wm;
  const int p_begin = bz * pps, p_end = min(P, p_begin + pps);
  const __amdgpu_buffer_rsrc_t dyR = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srcR = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
#pragma unroll
  for (int t = 0; t < AST_MAX_TAPS; ++t)
    if ((int)threadIdx.x == t) taptab[t] = g.tap[t];
  __syncthreads();

  // loader role: one pixel row per thread, chunks tq, tq + TPR, ...
  const int lrow = tid / TPR, tq = tid % TPR;
  int xdelta[NXI], xdh[NXI], xdw[NXI];              // per chunk slot: byte delta of (tap, channel), tap offsets; dh = 1<<20 marks "no column"
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    const int ch = tq + TPR * i, col = col0 + ch * E;
    xdelta[i] = 0; xdh[i] = 1 << 20; xdw[i] = 0;
    if (ch < CPX && col < ncols) {
      const int t = col / g.Cs, c = col - t * g.Cs;
      int dh, dw_, wt;
      decode_tap(taptab[t], dh, dw_, wt);
      xdh[i] = dh; xdw[i] = dw_;
      xdelta[i] = ((dh * g.Ws + dw_) * g.Cs + c) * ES;
    }
  }
  u32x4 yreg[NYI], xreg[NXI];
  auto load_tile = [&](int p0) __attribute__((always_inline)) {
    const int p = p0 + lrow;
    const bool pv = p < p_end;
    const int pp = pv ? p : 0;
#pragma unroll
    for (int i = 0; i < NYI; ++i) {
      const int ch = tq + TPR * i, cd = cd0 + ch * E;
      const bool ok = pv && ch < CPY && cd < g.Cd;
      yreg[i] = __builtin_amdgcn_raw_buffer_load_b128(dyR, ok ? (unsigned)((pp * g.Cd + cd) * ES) : OOB, 0, 0);
    }
    const int n = fdiv(pp, HWm, rcp_hw), rem = pp - n * HWm;
    const int hm = fdiv(rem, g.Wm, rcp_w), wq = rem - hm * g.Wm;
    const int hs0 = pv ? hm * g.sh + g.oh : -(1 << 21), ws0 = wq * g.sw + g.ow;
    const int base = (((n * g.Hs + hs0) * g.Ws + ws0) * g.Cs) * ES;
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const bool ok = (unsigned)(hs0 + xdh[i]) < (unsigned)g.Hs && (unsigned)(ws0 + xdw[i]) < (unsigned)g.Ws;
      xreg[i] = __builtin_amdgcn_raw_buffer_load_b128(srcR, ok ? (unsigned)(base + xdelta[i]) : OOB, 0, 0);
    }
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
    // bf16: 32-byte column segments are XOR-swizzled by bit 3 of the pixel row (SWZ): the transposed reads of a half-wave touch
    // rows r..r+3 and r+8..r+11, which a pitch of 32 B x odd alone leaves on the same banks.  Measured: 128-channel layer
    // 50.3 -> 48.4 us, 256-channel 47.1 -> 45.1, 512-channel 41.6 -> 39.0 (the pitch change alone: nothing); the
    // SQ_LDS_BANK_CONFLICT count of the launch did not move (3.89 M, five per MFMA), so that counter is not what it measures
    const int sw = SWZ ? ((lrow >> 3) & 1) << 1 : 0;             // in 16-byte chunks
#pragma unroll
    for (int i = 0; i < NYI; ++i) {
      const int ch = tq + TPR * i;
      if (ch < CPY) *reinterpret_cast<u32x4*>(Ys + lrow * PY + (ch ^ sw) * E) = yreg[i];
    }
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int ch = tq + TPR * i;
      if (ch < CPX) *reinterpret_cast<u32x4*>(Xs + lrow * PX + (ch ^ sw) * E) = xreg[i];
    }
  };

  f32x4 acc[RT][CTW];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, gq = lane >> 4;
  const int nk_all = (p_end - p_begin + BKP - 1) / BKP;
  const int nk = (nk_all + PG - 1) / PG;                         // trips: every group runs all of them (workgroup-wide barriers);
  if (nk > 0) load_tile(p_begin + pg * BKP);                    // a group past the slice's end loads zeros (p >= p_end)
  WG_STAMP(1);
#ifdef AST_STAMPS
  long long ph[5] = {0, 0, 0, 0, 0};              // cycles of thread 0 in: LDS store (incl. the wait for the loads), barrier, load issue, reads + MFMA, barrier
#define WG_PH(i) do { const long long t_ = __builtin_readcyclecounter(); ph[i] += t_ - tph; tph = t_; } while (0)
  long long tph = __builtin_readcyclecounter();
#else
#define WG_PH(i) do { } while (0)
#endif
  for (int kt = 0; kt < nk; ++kt) {
    store_tile();
    WG_PH(0);
    __syncthreads();
    WG_PH(1);
    if (kt + 1 < nk) load_tile(p_begin + ((kt + 1) * PG + pg) * BKP);       // in flight while this tile is consumed
    WG_PH(2);
    if constexpr (sizeof(T) == 2) {
      typedef __attribute__((address_space(3))) bf16x4 lds_b4;
      const int q = li >> 2, pcol = (li & 3) * 4;                // lane 4q+p supplies row q, columns 4p..4p+3 of its 16-lane group
#pragma unroll
      for (int ks = 0; ks < BKP / 32; ++ks) {
        const int r_lo = ks * 32 + 8 * gq + q;
        const int sx = SWZ ? (gq & 1) << 4 : 0;                 // rows r_lo and r_lo + 4 share bit 3 = gq & 1 (elements)
        bf16x8 af[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Ys + r_lo * PY + ((i * 16) ^ sx) + pcol));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Ys + (r_lo + 4) * PY + ((i * 16) ^ sx) + pcol));
          af[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int j = 0; j < CTW; ++j) {
          const int ct = wave + 4 * j;                          // uniform per wave
          if (ct < NCT) {
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Xs + r_lo * PX + ((ct * 16) ^ sx) + pcol));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Xs + (r_lo + 4) * PX + ((ct * 16) ^ sx) + pcol));
            const bf16x8 bf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[i][j], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll
      for (int s4 = 0; s4 < BKP / 4; ++s4) {
        float af[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) af[i] = Ys[(4 * s4 + gq) * PY + i * 16 + li];
#pragma unroll
        for (int j = 0; j < CTW; ++j) {
          const int ct = wave + 4 * j;
          if (ct < NCT) {
            const float bf = Xs[(4 * s4 + gq) * PX + ct * 16 + li];
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf, acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    WG_PH(3);
    __syncthreads();                                             // operand reads done before the next store
    WG_PH(4);
  }
#ifdef AST_STAMPS
  if (threadIdx.x == 0 && tix < 4096) { ast_wg_stamps[tix * 8 + 5] = (unsigned long long)((ph[0] << 32) | (ph[1] & 0xffffffffll)); ast_wg_phase[tix * 4 + 0] = ph[2]; ast_wg_phase[tix * 4 + 1] = ph[3]; ast_wg_phase[tix * 4 + 2] = ph[4]; ast_wg_phase[tix * 4 + 3] = nk; }
#endif

  WG_STAMP(2);
  if constexpr (PG > 1) {                           // sum the groups' partial tiles through LDS (the staging is free now)
    f32x4* red = reinterpret_cast<f32x4*>(wl_all);
#pragma unroll
    for (int src_g = 1; src_g < PG; ++src_g) {
      __syncthreads();
      if (pg == src_g) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < CTW; ++j) red[(i * CTW + j) * 256 + tid] = acc[i][j];
      }
      __syncthreads();
      if (pg == 0) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < CTW; ++j) {
            const f32x4 t = red[(i * CTW + j) * 256 + tid];
            acc[i][j][0] += t[0]; acc[i][j][1] += t[1]; acc[i][j][2] += t[2]; acc[i][j][3] += t[3];
          }
      }
    }
    if (pg > 0) return;
  }
  WG_STAMP(3);
  dw += (size_t)(bz % nrep) * rep_stride;          // this pixel slice's gradient replica
  // D[row = cd (gq*4+r)][col = column li]
#pragma unroll
  for (int j = 0; j < CTW; ++j) {
    const int ct = wave + 4 * j;
    const int col = col0 + ct * 16 + li;
    if (ct >= NCT || col >= ncols) continue;
    const int t = col / g.Cs, c = col - t * g.Cs;
    int a, b, wtc;
    decode_tap(taptab[t], a, b, wtc);
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cd = cd0 + i * 16 + gq * 4 + r;
        if (cd < g.Cd) unsafeAtomicAdd(dw + ((size_t)cd * g.wtaps + wtc) * g.Cs + c, acc[i][j][r]);
      }
  }
#ifdef AST_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the atomics have been acknowledged
#endif
  WG_STAMP(4); WG_STAMP(6);
}
#ifdef AST_STAMPS
extern "C" int ast_debug_read_wg_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ast_wg_stamps), (size_t)n * 8 * sizeof(unsigned long long));
}
extern "C" int ast_debug_read_wg_phases(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ast_wg_phase), (size_t)n * 4 * sizeof(long long));
}
#endif

template <typename T, int BMW, int NCT, int PG>
int launch_wgrad_pg(const void* dy, const void* src, float* dw, const ast_gather_t& g, int P, hipStream_t s) {
  constexpr int BKP = WgradCfg<T>::BKP, PAD = WgradCfg<T>::PAD;
  constexpr int GROUP = (int)sizeof(T) * BKP * ((BMW + PAD) + (NCT * 16 + PAD));
  constexpr int RED = PG > 1 ? (BMW / 16) * ((NCT + 3) / 4) * 256 * 16 : 0;
  constexpr int LDS = (PG * GROUP > RED ? PG * GROUP : RED) + 64;
  static bool attr_set = false;
  if (!attr_set) {
    AST_HIP(hipFuncSetAttribute((const void*)wgrad_kernel<T, BMW, NCT, PG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  const int gx = (g.Cd + BMW - 1) / BMW, gy = (g.ntaps * g.Cs + NCT * 16 - 1) / (NCT * 16);
  const int tiles = gx * gy;
  const char* wte = getenv("AST_WGRAD_WG_TARGET");
  // see launch_wgrad_halo; the single-group 64 x 192 tiles (23 launches of a step) take 384: 29.5-29.8 -> 28.1-28.3 us on
  // average in the replayed step (tools/knob_ab.sh; 320: 29.1, 448: 30.5), the two-group and the narrow ones do not (32.7 -> 43.3)
  const int wg_target = wte ? atoi(wte) : (P >= 1500000 ? 768 : (PG == 1 && NCT >= 12 ? 384 : 256));
  int nsplit = std::max(1, std::min((P + 4 * BKP - 1) / (4 * BKP), (wg_target + tiles - 1) / tiles));
  int pps = (P + nsplit - 1) / nsplit;
  pps = (pps + BKP - 1) / BKP * BKP;
  nsplit = (P + pps - 1) / pps;
  const unsigned dy_bytes = (unsigned)((size_t)P * g.Cd * sizeof(T));
  const unsigned src_bytes = (unsigned)((size_t)g.N * g.Hs * g.Ws * g.Cs * sizeof(T));
  const int total = gx * gy * nsplit;
  hipLaunchKernelGGL((wgrad_kernel<T, BMW, NCT, PG>), dim3((total + 7) / 8 * 8), dim3(256 * PG), LDS, s, (const T*)dy, (const T*)src, dw, g, P, pps,
                     dy_bytes, src_bytes, 1.0f / (float)(g.Hm * g.Wm), 1.0f / (float)g.Wm, gx, gy, nsplit, g_wg_nrep, g_wg_rep_stride);
  AST_CHECK_LAUNCH();
  return 0;
}

template <typename T, int BMW, int NCT>
int launch_wgrad(const void* dy, const void* src, float* dw, const ast_gather_t& g, int P, hipStream_t s) {
  // two pixel groups for the pixel-rich layers only: measured 172 800 pixels -11 % (51 -> 45 us), 43 200 pixels +20 %
  // (54 -> 65 us: their slices are a few K tiles long, halving them leaves the groups idle at the barriers)
  const char* pe = getenv("AST_WGRAD_PG");
  const char* mp = getenv("AST_WGRAD_PG_MINP");
  const int pg = pe ? atoi(pe) : 2;
  if (pg >= 2 && P >= (mp ? atoi(mp) : 100000)) return launch_wgrad_pg<T, BMW, NCT, 2>(dy, src, dw, g, P, s);
  return launch_wgrad_pg<T, BMW, NCT, 1>(dy, src, dw, g, P, s);
}

// ---------------------------------------------------------------------------
// Halo-tile weight gradient for the small-channel layers (Cd <= 64, all (tap, channel) columns
// <= 320 per workgroup).  The gathered kernel above fetches every source pixel once per tap
// (9x for 3x3) and spends ~8 VALU per 16-byte chunk on addressing: ~1.5 VALU cycles per MFMA
// cycle on these layers.  Here a workgroup walks 8x16-pixel tiles of dy: per tile it stages dy
// (128 x Cd) and the source patch (tile + halo, each pixel ONCE) in LDS, and every tap's B operand
// is a transposed read of the patch at a shifted address.  Accumulators stay in registers across
// all tiles of the workgroup; one atomic flush at the end.
// ---------------------------------------------------------------------------
constexpr int WH_TH = 8, WH_TW = 16, WH_MAXPL = 10;
struct WHaloPlan { int PH, PW, dhmin, dwmin, tiles_h, tiles_w, ntiles, lds; };

// PG: pixel groups.  The workgroup has PG groups of 4 waves; group pg streams tiles blockIdx.z*PG + pg, + gridDim.z*PG, ...
// through its OWN LDS staging and accumulators, and the groups' partial tiles are summed through LDS before ONE atomic flush per
// workgroup.  Same-address f32 atomics serialise (~38 ns per workgroup per address on the 16x72 gradient of the 2.4 M-pixel
// layer: 768 -> 3072 workgroups took 77 -> 164 us), so parallelism has to come from waves per workgroup, not from workgroups.
template <typename T, int BMW, int NCT, int PG>
__global__ __launch_bounds__(256 * PG) void wgrad_halo_kernel(const T* __restrict__ dy, const T* __restrict__ src,
                                                          float* __restrict__ dw, const ast_gather_t g, const WHaloPlan hp,
                                                          const unsigned dy_bytes, const unsigned src_bytes, const int nrep, const long rep_stride) {
  constexpr int E = 16 / sizeof(T), ES = sizeof(T);
  constexpr int MT = WH_TH * WH_TW;                 // 128 pixels per tile
  constexpr int PADY = sizeof(T) == 2 ? 8 : 16;
  constexpr int PY = BMW + PADY;                    // dy tile pitch (elements)
  constexpr int CPY = BMW / E;                      // dy chunks per pixel
  constexpr int NYI = (MT * CPY + 255) / 256;
  constexpr int RT = BMW / 16, CTW = (NCT + 3) / 4;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char wl_all[];
  const int pg = PG > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;      // wave-uniform
  const int PPX = g.Cs + (sizeof(T) == 2 ? 8 : 4);  // patch pixel pitch (elements): breaks the power-of-two stride
  const int group_bytes = ((int)sizeof(T) * (MT * PY + hp.PH * hp.PW * PPX) + 15) & ~15;
  unsigned char* wl = wl_all + pg * group_bytes;
  T* Ys = reinterpret_cast<T*>(wl);
  T* Xp = Ys + MT * PY;                             // patch [PH*PW][Cs + pad]
  constexpr int RED = PG > 1 ? (BMW / 16) * ((NCT + 3) / 4) * 256 * 16 : 0;     // bytes of one group's partial tile in the final LDS reduction
  int* taptab = reinterpret_cast<int*>(wl_all + max(PG * group_bytes, RED));   // behind both uses of the staging area

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int cd0 = blockIdx.x * BMW, col0 = blockIdx.y * NCT * 16;
  const int ncols = g.ntaps * g.Cs;
  const int UP = g.Cs / E;                          // 16-byte chunks per patch pixel
  const int PH = hp.PH, PW = hp.PW;
  const __amdgpu_buffer_rsrc_t dyR = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srcR = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
#pragma unroll
  for (int t = 0; t < AST_MAX_TAPS; ++t)
    if ((int)threadIdx.x == t) {
      int dh, dwv, wt;
      decode_tap(g.tap[t], dh, dwv, wt);
      taptab[t] = ((dh - hp.dhmin) * PW + (dwv - hp.dwmin)) * PPX;   // patch element offset of the tap
      taptab[16 + t] = wt;
    }
  __syncthreads();

  // ---- loader descriptors: patch slots (py, px, part) and dy slots (pixel, chunk)
  int ppy[WH_MAXPL], ppx[WH_MAXPL], pgo[WH_MAXPL], plo[WH_MAXPL];
  const int npatch = PH * PW * UP;
#pragma unroll
  for (int i = 0; i < WH_MAXPL; ++i) {
    const int idx = tid + 256 * i;
    ppy[i] = -(1 << 20); ppx[i] = 0; pgo[i] = 0; plo[i] = -1;
    if (idx < npatch) {
      const int pix = idx / UP, part = idx - pix * UP;
      ppy[i] = pix / PW; ppx[i] = pix - ppy[i] * PW;
      pgo[i] = ((ppy[i] * g.Ws + ppx[i]) * g.Cs + part * E) * ES;
      plo[i] = pix * PPX + part * E;
    }
  }
  int ypix[NYI], ych[NYI];
#pragma unroll
  for (int i = 0; i < NYI; ++i) {
    const int idx = tid + 256 * i;
    ypix[i] = idx < MT * CPY ? idx / CPY : -1;
    ych[i] = idx < MT * CPY ? idx % CPY : 0;
  }
  // ---- per-lane operand offsets
  const int li = lane & 15, gq = lane >> 4;
  int coloff[CTW];                                  // patch element offset of this lane's column (tap, channel) per column tile
  bool colok[CTW];
#pragma unroll
  for (int j = 0; j < CTW; ++j) {
    const int ct = wave + 4 * j;
    const int col = col0 + ct * 16 + (sizeof(T) == 2 ? (li & 3) * 4 : li);
    colok[j] = ct < NCT && col < ncols;
    const int t = colok[j] ? col / g.Cs : 0;
    coloff[j] = taptab[t] + (colok[j] ? col - t * g.Cs : 0);
  }

  f32x4 acc[RT][CTW];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 preg[WH_MAXPL], yreg[NYI];
  auto load_tile = [&](int tile_in) __attribute__((always_inline)) {
    const bool tvalid = tile_in < hp.ntiles;          // a group past its last tile loads zeros (every offset out of range)
    const int tile = tvalid ? tile_in : 0;
    const int per_img = hp.tiles_h * hp.tiles_w;
    const int n = tile / per_img, r = tile - n * per_img;
    const int th = r / hp.tiles_w, tw = r - th * hp.tiles_w;
    const int hm0 = th * WH_TH, wm0 = tw * WH_TW;
    const int hs_org = hm0 * g.sh + g.oh + hp.dhmin, ws_org = wm0 * g.sw + g.ow + hp.dwmin;
    const int base = (((n * g.Hs + hs_org) * g.Ws + ws_org) * g.Cs) * ES;
#pragma unroll
    for (int i = 0; i < WH_MAXPL; ++i) {
      const bool ok = tvalid && plo[i] >= 0 && (unsigned)(hs_org + ppy[i]) < (unsigned)g.Hs && (unsigned)(ws_org + ppx[i]) < (unsigned)g.Ws;
      preg[i] = __builtin_amdgcn_raw_buffer_load_b128(srcR, ok ? (unsigned)(base + pgo[i]) : OOB, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NYI; ++i) {
      const int ty = ypix[i] >> 4, tx = ypix[i] & 15;
      const int hm = hm0 + ty, wq = wm0 + tx;
      const int cd = cd0 + ych[i] * E;
      const bool ok = tvalid && ypix[i] >= 0 && hm < g.Hm && wq < g.Wm && cd < g.Cd;
      yreg[i] = __builtin_amdgcn_raw_buffer_load_b128(dyR, ok ? (unsigned)((((n * g.Hm + hm) * g.Wm + wq) * g.Cd + cd) * ES) : OOB, 0, 0);
    }
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WH_MAXPL; ++i)
      if (plo[i] >= 0) *reinterpret_cast<u32x4*>(Xp + plo[i]) = preg[i];
#pragma unroll
    for (int i = 0; i < NYI; ++i)
      if (ypix[i] >= 0) *reinterpret_cast<u32x4*>(Ys + ypix[i] * PY + ych[i] * E) = yreg[i];
  };

  // every group runs the workgroup's trip count (the barriers are workgroup-wide); a group without a tile works on zeros
  const int tstride = gridDim.z * PG;
  const int first = blockIdx.z * PG;
  const int ntrips = first < hp.ntiles ? (hp.ntiles - first + tstride - 1) / tstride : 0;
  int tile = first + pg;
  if (ntrips > 0) load_tile(tile);
  for (int trip = 0; trip < ntrips; ++trip, tile += tstride) {
    store_tile();
    __syncthreads();
    if (trip + 1 < ntrips) load_tile(tile + tstride);                        // next tile in flight during the MFMAs
    if constexpr (sizeof(T) == 2) {
      typedef __attribute__((address_space(3))) bf16x4 lds_b4;
      const int q = li >> 2, pcol = (li & 3) * 4;
#pragma unroll
      for (int ks = 0; ks < MT / 32; ++ks) {
        // rows (pixels) of this lane's two 4-row blocks: p = 32 ks + 8 gq + q (+4); tile row = p >> 4, column = p & 15
        const int p_lo = ks * 32 + 8 * gq + q, p_hi = p_lo + 4;
        const int x_lo = (((p_lo >> 4) * g.sh) * PW + (p_lo & 15) * g.sw) * PPX;
        const int x_hi = (((p_hi >> 4) * g.sh) * PW + (p_hi & 15) * g.sw) * PPX;
        bf16x8 af[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Ys + p_lo * PY + i * 16 + pcol));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Ys + p_hi * PY + i * 16 + pcol));
          af[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int j = 0; j < CTW; ++j) {
          if (wave + 4 * j < NCT) {                                         // uniform per wave; masked columns read a valid address
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Xp + x_lo + coloff[j]));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Xp + x_hi + coloff[j]));
            const bf16x8 bf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[i][j], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll 4
      for (int s4 = 0; s4 < MT / 4; ++s4) {
        const int p = 4 * s4 + gq;
        const int xo = (((p >> 4) * g.sh) * PW + (p & 15) * g.sw) * PPX;
        float af[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) af[i] = Ys[p * PY + i * 16 + li];
#pragma unroll
        for (int j = 0; j < CTW; ++j) {
          if (wave + 4 * j < NCT) {
            const float bf = colok[j] ? Xp[xo + coloff[j]] : 0.f;
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf, acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }

  if constexpr (PG > 1) {                           // sum the groups' partial tiles through LDS (the staging is free now)
    f32x4* red = reinterpret_cast<f32x4*>(wl_all);
#pragma unroll
    for (int src_g = 1; src_g < PG; ++src_g) {
      __syncthreads();
      if (pg == src_g) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < CTW; ++j) red[(i * CTW + j) * 256 + tid] = acc[i][j];
      }
      __syncthreads();
      if (pg == 0) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < CTW; ++j) {
            const f32x4 t = red[(i * CTW + j) * 256 + tid];
            acc[i][j][0] += t[0]; acc[i][j][1] += t[1]; acc[i][j][2] += t[2]; acc[i][j][3] += t[3];
          }
      }
    }
    if (pg > 0) return;
  }
  // D[row = cd (gq*4+r)][col = column li]
  dw += (size_t)(blockIdx.z % nrep) * rep_stride;      // this slice's gradient replica (see g_wg_nrep)
#pragma unroll
  for (int j = 0; j < CTW; ++j) {
    const int ct = wave + 4 * j;
    const int col = col0 + ct * 16 + li;
    if (ct >= NCT || col >= ncols) continue;
    const int t = col / g.Cs, c = col - t * g.Cs;
    const int wtc = taptab[16 + t];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cd = cd0 + i * 16 + gq * 4 + r;
        if (cd < g.Cd) unsafeAtomicAdd(dw + ((size_t)cd * g.wtaps + wtc) * g.Cs + c, acc[i][j][r]);
      }
  }
}

bool plan_wgrad_halo(const ast_gather_t& g, int dtype, int nct, int bmw, WHaloPlan& hp) {
  const char* mc = getenv("AST_WGRAD_HALO_MAXCD");
  if (g.ntaps < 2 || g.Cd > (mc ? atoi(mc) : 32)) return false;          // measured: wins for <= 32 output channels, loses at 64
  const int E = dtype == AST_BF16 ? 8 : 4, ES = dtype == AST_BF16 ? 2 : 4;
  int dhmin = 64, dhmax = -64, dwmin = 64, dwmax = -64;
  for (int t = 0; t < g.ntaps; ++t) {
    const int dh = (g.tap[t] & 255) - 64, dw = ((g.tap[t] >> 8) & 255) - 64;
    dhmin = std::min(dhmin, dh); dhmax = std::max(dhmax, dh); dwmin = std::min(dwmin, dw); dwmax = std::max(dwmax, dw);
  }
  hp.dhmin = dhmin; hp.dwmin = dwmin;
  hp.PH = (WH_TH - 1) * g.sh + (dhmax - dhmin) + 1;
  hp.PW = (WH_TW - 1) * g.sw + (dwmax - dwmin) + 1;
  if (hp.PH * hp.PW * (g.Cs / E) > 256 * WH_MAXPL) return false;
  hp.tiles_h = (g.Hm + WH_TH - 1) / WH_TH; hp.tiles_w = (g.Wm + WH_TW - 1) / WH_TW;
  hp.ntiles = g.N * hp.tiles_h * hp.tiles_w;
  const int ppx = g.Cs + (ES == 2 ? 8 : 4), pady = ES == 2 ? 8 : 16;
  hp.lds = ((ES * (WH_TH * WH_TW * (bmw + pady) + hp.PH * hp.PW * ppx) + 15) & ~15);      // one group's staging
  if (hp.lds > 96 * 1024) return false;
  // tile quantisation: skip when the 8x16 tiling wastes most of the work (tiny images go to the gathered kernel)
  const double eff = (double)g.Hm * g.Wm / ((double)hp.tiles_h * hp.tiles_w * WH_TH * WH_TW);
  return eff >= 0.6 && hp.ntiles >= 256;
}

template <typename T, int BMW, int NCT, int PG>
int launch_wgrad_halo_pg(const void* dy, const void* src, float* dw, const ast_gather_t& g, const WHaloPlan& hp, hipStream_t s) {
  constexpr int RED = (BMW / 16) * ((NCT + 3) / 4) * 256 * 16;          // bytes of one group's partial tile in the LDS reduction
  const int lds = std::max(PG * hp.lds, PG > 1 ? RED : 0) + 160;
  static int attr_lds = 0;
  if (lds > attr_lds) {
    AST_HIP(hipFuncSetAttribute((const void*)wgrad_halo_kernel<T, BMW, NCT, PG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_lds = 160 * 1024;
  }
  const int gx = (g.Cd + BMW - 1) / BMW, gy = (g.ntaps * g.Cs + NCT * 16 - 1) / (NCT * 16);
  // every workgroup adds its whole dW tile into the same few KB: same-address atomics serialise, so the workgroup count
  // stays at about one per CU and the waves come from the pixel groups (sweep in profiles/r01 and r02)
  const char* wt = getenv("AST_WGRAD_WG_TARGET");
  const int wg_target = wt ? atoi(wt) : (PG > 1 ? 256 : ((long)g.N * g.Hm * g.Wm >= 1500000 ? 768 : 256));
  const int gz = std::max(1, std::min((hp.ntiles + PG - 1) / PG, wg_target / (gx * gy)));
  const unsigned dy_bytes = (unsigned)((size_t)g.N * g.Hm * g.Wm * g.Cd * sizeof(T));
  const unsigned src_bytes = (unsigned)((size_t)g.N * g.Hs * g.Ws * g.Cs * sizeof(T));
  hipLaunchKernelGGL((wgrad_halo_kernel<T, BMW, NCT, PG>), dim3(gx, gy, gz), dim3(256 * PG), lds, s, (const T*)dy, (const T*)src, dw, g, hp,
                     dy_bytes, src_bytes, g_wg_nrep, g_wg_rep_stride);
  AST_CHECK_LAUNCH();
  return 0;
}

template <typename T, int BMW, int NCT>
int launch_wgrad_halo(const void* dy, const void* src, float* dw, const ast_gather_t& g, const WHaloPlan& hp, hipStream_t s) {
  // pixel groups: as many as the LDS holds (<= 4), when every group gets several tiles
  // (four groups need <= 128 VGPRs per thread; the kernel uses 130-200 and spills: 50 -> 105 us on the 32-channel layer)
  const char* pe = getenv("AST_WGRAD_PG");
  int pg = pe ? atoi(pe) : 2;
  while (pg > 1 && (pg * hp.lds > 150 * 1024 || hp.ntiles < 256 * pg * 2)) pg >>= 1;
  if (pg >= 4) return launch_wgrad_halo_pg<T, BMW, NCT, 4>(dy, src, dw, g, hp, s);
  if (pg == 2) return launch_wgrad_halo_pg<T, BMW, NCT, 2>(dy, src, dw, g, hp, s);
  return launch_wgrad_halo_pg<T, BMW, NCT, 1>(dy, src, dw, g, hp, s);
}

// ---------------------------------------------------------------------------
// Patch-staged convolution GEMM ("pconv") for multi-tap, stride-1 gathers (3x3 convolutions forward and data
// gradient, the output-parity classes of stride-2 data gradients / transposed convolutions).
//
// Why: the gathered kernel above fetches every source pixel once per TAP through the vector-memory path (im2col in the
// TA) and writes it to LDS once per tap.  Measured (tools/micro/fragbench.hip, ldbench.hip): that path delivers 57 B/clk/CU
// only for wave-contiguous kilobytes and 26 B/clk/CU for 64-byte row pieces at a pitch >= 128 B, and ds_write moves
// 64 B/clk/CU -- at 9 taps both cost more cycles than the MFMAs they feed.  Here a workgroup owns a 2-D tile of output
// pixels (128 = 8 fragments of 16 consecutive pixels of one row) x BN channels and stages the source PATCH (tile + halo)
// once per 64-channel slab: every pixel crosses L1 and the LDS write port once, in whole contiguous rows.  A tap is then a
// constant pixel offset into the patch: the B fragment of (tap, pixel fragment) is one ds_read_b128 at a shifted
// address.  Only the weights (BN x 128 B per tap and slab) stream through a double-buffered LDS stage, one barrier per
// tap.  Waves split the tile's fragments (TM each) and every wave covers all TN channel tiles: TM + TN fragment reads
// per TM*TN MFMAs.
//
// LDS images: pixel (or weight row) p at p*SLB, its 16-byte chunk c stored at chunk c ^ H(p), H(p) = p & 7 for
// 128-byte pixels, (p >> 1) & 3 for 64-byte pixels: with ds_read_b128's 16-lane groups this is conflict-free for ANY
// fragment base pixel (brute-forced over all bases), which the per-tap shifts need.
// ---------------------------------------------------------------------------
constexpr int PC_MAXPL = 8;          // patch chunks (16 B) per thread

// In-kernel phase stamps (debug build only: `make stamps` -> libast_hip_stamps.so, read by tools/pconv_stamps.py): thread 0 of
// every workgroup records the shader clock at the phase boundaries of the patch kernel.
#ifdef AST_STAMPS
__device__ unsigned long long ast_stamps[16384 * 8];
#define PC_STAMP(k) do { if (threadIdx.x == 0 && tix < 16384) ast_stamps[tix * 8 + (k)] = (k) >= 6 ? wall_clock64() : __builtin_readcyclecounter(); } while (0)
#else
#define PC_STAMP(k) do { } while (0)
#endif
struct PconvPlan { int TH, TWF, PH, PW, dhmin, dwmin, tiles_h, tiles_w, nct, lds, rows, tm; unsigned m_nct, m_per_img, m_tiles_w, m_pw20, m_twf, m_twf20; unsigned long long tapq[3]; };
// Division by a run-time constant d through m = ceil(2^32 / d): floor(n / d) = umulhi(n, m), exact while n * d < 2^32
// (plan_pconv checks).  On wave-uniform operands it is ONE scalar instruction (s_mul_hi_u32); the float-reciprocal fdiv the
// gathered kernels use is ~12 VALU even for scalars, and the four waves of a SIMD all run this prologue at the same time.
__host__ __device__ __forceinline__ unsigned pc_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }
__device__ __forceinline__ int pc_div(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }
// per-lane operands below 2^11 by divisors below 2^9: full-rate 24-bit multiply, m20 = ceil(2^20 / d)
__device__ __forceinline__ int pc_div20(int n, unsigned m20) { return (int)(__umul24((unsigned)n, m20) >> 20); }

template <int SLB> __device__ __forceinline__ int pc_h(int p) { return SLB == 128 ? (p & 7) : ((p >> 1) & 3); }

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
__device__ __forceinline__ f32x4 pc_as_f32x4(u32x4 u) {
  return __builtin_bit_cast(f32x4, u);            // (whole-vector casts: __builtin_bit_cast of a vector ELEMENT reads element 0 with this compiler)
}
// Four consecutive channels of one pixel at byte offset voff (+ compile-time imm) of a tensor of T, as f32 (zeros when out of range).
template <typename T>
__device__ __forceinline__ f32x4 pc_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, int imm) {
  if constexpr (sizeof(T) == 4) {
    return pc_as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(r, voff + (unsigned)imm, 0, 0));
  } else {
    const u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(r, voff + (unsigned)imm, 0, 0);
    const u32x4 w{u.x << 16, u.x & 0xffff0000u, u.y << 16, u.y & 0xffff0000u};
    return __builtin_bit_cast(f32x4, w);
  }
}
template <typename T>
__device__ __forceinline__ void pc_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, int imm, f32x4 v) {
  if constexpr (sizeof(T) == 4) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff + (unsigned)imm, 0, 0);
  } else {
    const bf16x4 q{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, q), r, voff + (unsigned)imm, 0, 0);
  }
}

template <typename T, int SLB, int TM, int TN>
__global__ __launch_bounds__(256, (TM * TN <= 8 ? 4 : 2)) void pconv_kernel(const T* __restrict__ src, const T* __restrict__ wgt, const float* __restrict__ bias,
                                                    T* __restrict__ dst, const ast_gather_t g, const PconvPlan pp, const int flags,
                                                    float* __restrict__ ws, const unsigned src_bytes, const unsigned wgt_bytes,
                                                    const T* __restrict__ bn_x, const float* __restrict__ bn_scale,
                                                    const float* __restrict__ bn_shift) {
  constexpr int ES = sizeof(T);
  constexpr int CPP = SLB / 16;                  // 16-byte chunks per pixel (and per weight row) per slab
  constexpr int BN = TN * 16;
  constexpr int NWL = (BN * CPP + 255) / 256;    // weight chunks per thread per stage
  constexpr int KS = SLB / 64;                   // MFMA K steps (64 B of a row) per (tap, slab)
  constexpr unsigned OOB = 0x80000000u;
  using frag = typename Mma<T>::frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char pl[];
  const int patch_bytes = pp.PH * pp.PW * SLB;
  unsigned char* wbuf = pl + patch_bytes;        // two stages of BN x SLB

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int chunk = gridDim.x >> 3;              // XCD-aware order, as igemm_kernel
  const int tix = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  const int per_img = pp.tiles_h * pp.tiles_w;
  if (tix >= g.N * per_img * pp.nct) return;
  PC_STAMP(0); PC_STAMP(7);
  const int st = pc_div(tix, pp.nct, pp.m_nct), ct = tix - st * pp.nct;      // channel tiles of one spatial tile are neighbours (same patch in L2)
  const int n = pc_div(st, per_img, pp.m_per_img), r = st - n * per_img;
  const int th = pc_div(r, pp.tiles_w, pp.m_tiles_w), tw = r - th * pp.tiles_w;
  const int TWP = pp.rows ? g.Wm : pp.TWF * 16;
  const int hm0 = th * pp.TH, wm0 = tw * TWP;
  const int bn0 = ct * BN;
  const int hs_org = hm0 + g.oh + pp.dhmin, ws_org = wm0 + g.ow + pp.dwmin;      // stride-1 gathers only (plan_pconv)
  const __amdgpu_buffer_rsrc_t srcR = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wgtR = __builtin_amdgcn_make_buffer_rsrc((void*)wgt, 0, wgt_bytes, 0x00020000);

  // Tap t: patch pixel offset (12 bits) and weight-slice index (4 bits), packed four to a 64-bit kernel argument by
  // plan_pconv: scalar shifts instead of an LDS table (whose set-up was nine branchy blocks, each behind its own s_load,
  // plus a barrier before the first load could issue).
  auto tap_entry = [&](int t) __attribute__((always_inline)) -> unsigned {
    const unsigned long long q = t < 4 ? pp.tapq[0] : (t < 8 ? pp.tapq[1] : pp.tapq[2]);
    return (unsigned)(q >> ((t & 3) * 16)) & 0xffffu;
  };
  const unsigned wslice = (unsigned)(g.Cs * ES);  // bytes of one tap's slice of a weight row

  // ---- loader descriptors (this workgroup's tile: fixed for the whole kernel).  Chunk i of thread t is chunk t % CPP of
  // patch pixel t / CPP + i * PSTEP.  PSTEP is a multiple of 8, so the swizzle term H(pixel) is the same for all of a
  // thread's chunks: their LDS addresses are l0 + i * PSTEP * SLB (compile-time immediates).
  constexpr int PSTEP = 256 / CPP;               // patch pixels between a thread's consecutive chunks
  const int npix = pp.PH * pp.PW;
  const int p0 = tid / CPP;
  const int l0 = p0 * SLB + (((tid % CPP) ^ pc_h<SLB>(p0)) << 4);
  unsigned goff[PC_MAXPL];
  {
    const int rowb = g.Cs * ES;
    const unsigned gbase = (unsigned)(((n * g.Hs + hs_org) * g.Ws + ws_org) * rowb + (tid % CPP) * 16);
#pragma unroll
    for (int i = 0; i < PC_MAXPL; ++i) {
      const int pix = p0 + i * PSTEP;
      const int py = pc_div20(pix, pp.m_pw20), px = pix - (int)__umul24(py, pp.PW);
      const bool ok = pix < npix && (unsigned)(hs_org + py) < (unsigned)g.Hs && (unsigned)(ws_org + px) < (unsigned)g.Ws;
      goff[i] = ok ? gbase + __umul24(__umul24(py, g.Ws) + px, rowb) : OOB;
    }
  }
  unsigned woff[NWL];
  int wl[NWL];
#pragma unroll
  for (int k = 0; k < NWL; ++k) {
    const int idx = tid + 256 * k;
    const int c = idx % CPP, row = idx / CPP;
    const bool ok = row < BN && bn0 + row < g.Cd;
    woff[k] = ok ? (unsigned)(((bn0 + row) * g.wtaps * g.Cs) * ES + c * 16) : OOB;
    wl[k] = row < BN ? row * SLB + ((c ^ pc_h<SLB>(row)) << 4) : -1;
  }
  // ---- fragment addresses
  int aoff[TN];                                   // weight rows i*16 + fr: H(row) = H(fr)
#pragma unroll
  for (int i = 0; i < TN; ++i) aoff[i] = (i * 16 + fr) * SLB + ((fq ^ pc_h<SLB>(fr)) << 4);
  // Lane -> output pixel of fragment f.  2-D tiles: fragment = 16 consecutive pixels of tile row f / TWF.  Row-block tiles
  // (pp.rows: TH full-width rows of a narrow image): the tile's pixels in raster order, 16 per fragment (a fragment may
  // straddle a row end; its patch pixels then jump by the halo width: a 2-way LDS conflict on that read, no more).
  auto lane_pixel = [&](int j, int& ty, int& tx) __attribute__((always_inline)) {
    const int f = __builtin_amdgcn_readfirstlane(wave) * TM + j;      // wave-uniform: scalar arithmetic
    if (pp.rows) {
      const int q = f * 16 + fr;
      ty = pc_div20(q, pp.m_twf20); tx = q - ty * g.Wm;               // m_twf20 = ceil(2^20 / Wm) in this mode
    } else {
      ty = pc_div(f, pp.TWF, pp.m_twf); tx = (f - ty * pp.TWF) * 16 + fr;
    }
  };
  int pb[TM];                                     // patch pixel of this lane's output pixel (before the tap offset)
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    int ty, tx;
    lane_pixel(j, ty, tx);
    pb[j] = ty < pp.TH ? ty * pp.PW + tx : 0;     // lanes past the tile read a valid address; their results are dropped
  }

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // destination byte offset of (pixel of fragment j, first channel of the lane), or OOB; bias of the lane's channels
  const unsigned dst_bytes = (unsigned)(g.N * g.Hd * g.Wd * g.Cd) * (unsigned)ES;            // < 2^31 (plan_pconv)
  unsigned dofs[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    int ty, tx;
    lane_pixel(j, ty, tx);
    const int hm = hm0 + ty, wq = wm0 + tx;
    const bool ok = ty < pp.TH && hm < g.Hm && wq < g.Wm;
    dofs[j] = ok ? (unsigned)((((n * g.Hd + hm * g.dsh + g.doh) * g.Wd + (wq * g.dsw + g.dow)) * g.Cd + bn0 + fq * 4) * ES) : OOB;
  }

  PC_STAMP(1);
  const int nslab = (g.Cs * ES) / SLB;
  u32x4 wr[NWL];
  for (int s = 0; s < nslab; ++s) {
    const unsigned sb = (unsigned)(s * SLB);
    {
      u32x4 pr[PC_MAXPL];
#pragma unroll
      for (int i = 0; i < PC_MAXPL; ++i) pr[i] = __builtin_amdgcn_raw_buffer_load_b128(srcR, goff[i] + sb, 0, 0);
      const unsigned w0 = (tap_entry(0) >> 12) * wslice + sb;
#pragma unroll
      for (int k = 0; k < NWL; ++k) wr[k] = __builtin_amdgcn_raw_buffer_load_b128(wgtR, woff[k] + w0, 0, 0);
      // (the previous slab's last tap ended with a barrier: every read of the patch and of stage 0 is done)
#pragma unroll
      for (int i = 0; i < PC_MAXPL; ++i)
        if (p0 + i * PSTEP < npix) *reinterpret_cast<u32x4*>(pl + l0 + i * (PSTEP * SLB)) = pr[i];
#pragma unroll
      for (int k = 0; k < NWL; ++k)
        if (wl[k] >= 0) *reinterpret_cast<u32x4*>(wbuf + wl[k]) = wr[k];
    }
    __syncthreads();
    if (s == 0) PC_STAMP(2);
    for (int t = 0; t < g.ntaps; ++t) {
      const int cur = t & 1;
      const bool more = t + 1 < g.ntaps;
      if (more) {
        const unsigned wn = (tap_entry(t + 1) >> 12) * wslice + sb;
#pragma unroll
        for (int k = 0; k < NWL; ++k) wr[k] = __builtin_amdgcn_raw_buffer_load_b128(wgtR, woff[k] + wn, 0, 0);
      }
      const int toff = (int)(tap_entry(t) & 0xfffu);
      const unsigned char* wcur = wbuf + cur * (BN * SLB);
      int xa[TM];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int p = pb[j] + toff;
        xa[j] = p * SLB + ((fq ^ pc_h<SLB>(p)) << 4);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        frag wf[TN], xf[TM];
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const frag*>(wcur + (aoff[i] ^ (ks << 6)));
#pragma unroll
        for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const frag*>(pl + (xa[j] ^ (ks << 6)));
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[i][j] = Mma<T>::run(wf[i], xf[j], acc[i][j]);
      }
      if (more) {
        unsigned char* wnext = wbuf + (cur ^ 1) * (BN * SLB);
#pragma unroll
        for (int k = 0; k < NWL; ++k)
          if (wl[k] >= 0) *reinterpret_cast<u32x4*>(wnext + wl[k]) = wr[k];
      }
      __syncthreads();
    }
  }

  PC_STAMP(3);
  // ---- epilogue: lane owns pixel fr of fragment j, channels fq*4.. of channel tile i.  Same arithmetic as epi_store_m
  // (igemm_kernel's), written for this kernel's fixed tile: destination offsets (dofs) were computed before the tap loop, the bias is
  // fetched once, loads / stores are buffer instructions whose masked lanes carry an out-of-range offset (no exec-mask branches, no
  // 64-bit address arithmetic), and the mode (plain / BatchNorm forward sums / backward sums) is a wave-uniform branch
  // around three straight-line bodies.  (The shared epilogue cost this kernel ~70 VALU per 4-value block and a dependent
  // bias load per block: 1 015 VALU per 144 MFMAs on the 64-channel layers.)
  const bool accumulate = flags & 1, relu = flags & 2, stats = flags & 8, bstats = flags & 16, bn_relu = !(flags & 32);
  const __amdgpu_buffer_rsrc_t dstR = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, dst_bytes, 0x00020000);
  f32x4 bv[TN];
  {
    const __amdgpu_buffer_rsrc_t biasR = __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, bias ? (unsigned)g.Cd * 4u : 0u, 0x00020000);
#pragma unroll
    for (int i = 0; i < TN; ++i) bv[i] = pc_as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(biasR, (unsigned)(bn0 + fq * 4) * 4u, i * 64, 0));
  }
  float st1[TN][4], st2[TN][4];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { st1[i][q] = 0.f; st2[i][q] = 0.f; }
  auto run = [&](auto mode_tag) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_tag)::value;
    const __amdgpu_buffer_rsrc_t bnxR = __builtin_amdgcn_make_buffer_rsrc((void*)bn_x, 0, MODE == 2 ? dst_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t scR = __builtin_amdgcn_make_buffer_rsrc((void*)bn_scale, 0, MODE == 2 ? (unsigned)g.Cd * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t sfR = __builtin_amdgcn_make_buffer_rsrc((void*)bn_shift, 0, MODE == 2 ? (unsigned)g.Cd * 4u : 0u, 0x00020000);
    // channel tile outermost: the coefficient / old-value / BN-input loads of one channel tile (all TM pixels) are issued
    // together and waited for once, and only one tile's worth of them is live
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      if (bn0 + i * 16 >= g.Cd) continue;           // wave-uniform (Cd % 16 == 0: plan_pconv)
      f32x4 sc4, sf4, old[TM], xq[TM];
      if constexpr (MODE == 2) {
        sc4 = pc_as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(scR, (unsigned)(bn0 + fq * 4) * 4u, i * 64, 0));
        sf4 = pc_as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(sfR, (unsigned)(bn0 + fq * 4) * 4u, i * 64, 0));
#pragma unroll
        for (int j = 0; j < TM; ++j) xq[j] = pc_load4<T>(bnxR, dofs[j], i * 16 * ES);
      }
      if (accumulate) {
#pragma unroll
        for (int j = 0; j < TM; ++j) old[j] = pc_load4<T>(dstR, dofs[j], i * 16 * ES);
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const bool live = dofs[j] != OOB;           // lanes past the tile / image computed on a valid patch address: drop them
        f32x4 v = acc[i][j] + bv[i];
        if constexpr (MODE == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float q = live ? (float)(T)v[r] : 0.f;                     // the value as stored
            st1[i][r] += q; st2[i][r] = __builtin_fmaf(q, q, st2[i][r]);
          }
        }
        if constexpr (MODE == 2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float q = (float)(T)v[r];
            const float dz = (live && (!bn_relu || __builtin_fmaf(xq[j][r], sc4[r], sf4[r]) > 0.f)) ? q : 0.f;
            st1[i][r] += dz; st2[i][r] = __builtin_fmaf(dz, xq[j][r], st2[i][r]);
          }
        }
        if (accumulate) v += old[j];
        if (relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        pc_store4<T>(dstR, dofs[j], i * 16 * ES, v);
      }
    }
  };
  if (stats) run(std::integral_constant<int, 1>{});
  else if (bstats) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 0>{});
  PC_STAMP(4);
  if (stats || bstats) epi_flush<TN>(ws, bstats, g.Cd, st1, st2, (flags & 64) ? ~n : tix, bn0, fr, fq,
                                       (flags & 64) ? reinterpret_cast<float*>(pl) : nullptr);   // per-image tables only: with 64 slots the per-wave atomics are as fast (and one barrier pair cheaper)      // bit 6: per-image slots (tiles never straddle images)
  PC_STAMP(5); PC_STAMP(6);
}
#ifdef AST_STAMPS
extern "C" int ast_debug_read_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ast_stamps), (size_t)n * 8 * sizeof(unsigned long long));
}
#endif

// Tile plan of the patch kernel, or false when the geometry should stay on the gathered kernel.
bool plan_pconv(const ast_gather_t& g, int dtype, PconvPlan& pp, int& slb, int& tn) {
  const char* en = getenv("AST_PCONV");                      // read per call (host side only): tests toggle it at run time
  if ((en && atoi(en) == 0) || g.ntaps < 2 || g.sh != 1 || g.sw != 1) return false;
  const int ES = dtype == AST_BF16 ? 2 : 4;
  const int rowb = g.Cs * ES;
  if (rowb % 64) return false;
  slb = (rowb % 128 == 0) ? 128 : 64;
  if (g.Cd < 32 || g.Cd % 16) return false;
  if ((double)g.N * g.Hd * g.Wd * g.Cd * ES >= 2147483648.0) return false;      // 32-bit destination offsets, OOB sentinel 2^31
  tn = g.Cd >= 64 ? 4 : 2;
  int dhmin = 64, dhmax = -64, dwmin = 64, dwmax = -64;
  for (int t = 0; t < g.ntaps; ++t) {
    const int dh = (g.tap[t] & 255) - 64, dw = ((g.tap[t] >> 8) & 255) - 64;
    dhmin = std::min(dhmin, dh); dhmax = std::max(dhmax, dh); dwmin = std::min(dwmin, dw); dwmax = std::max(dwmax, dw);
  }
  const int cpp = slb / 16;
  const int max_px = 256 * PC_MAXPL / cpp;                        // patch pixels a workgroup's loader covers
  double best = 0.0;
  pp.rows = 0; pp.tm = 2;
  // 2-D tiles of 8 fragments (8x16, 4x32, 2x64, 1x128 pixels); thin layers (64-byte pixels: K = 9 x 32 channels, 36 MFMAs per
  // wave and tile) take 16 fragments (16x16 ...) so that the per-workgroup prologue / epilogue is paid half as often
  // 2-D tiles of 8, 12 or 16 fragments (8x16, 4x32, ... pixels).  More fragments per wave = fewer LDS fragment reads per
  // MFMA ((TM + TN) / (TM * TN): the tap loop runs at the LDS read rate with four 8-fragment workgroups on a CU) and the
  // prologue / epilogue paid less often, but fewer workgroups per CU (LDS, registers) and a coarser last round:
  //  * 128-byte slabs: 8 fragments; 12 when the 8-fragment grid fits one round of 4 workgroups per CU anyway (measured on
  //    the 128-channel layer: 25.9 -> 23.4 us; the 64-channel layer, two rounds either way, 27.9 -> 28.9 us);
  //  * 64-byte slabs (32 channels: K = 9 x 32, 36 MFMAs per wave and 8-fragment tile): 16 where the grid stays full.
  const char* nfe = getenv("AST_PCONV_NF");                    // experiments: force the fragment count of 2-D tiles
  struct Cand { double eff; int th, twf, ph, pw, tiles_h, tiles_w; } cand[3] = {{0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0}};
  for (int k = 0; k < 3; ++k) {
    const int nf = 8 + 4 * k;
    for (int twf = 1; twf <= nf; twf *= 2) {
      if (nf % twf) continue;
      const int th = nf / twf, twp = twf * 16;
      const int ph = th + (dhmax - dhmin), pw = twp + (dwmax - dwmin);
      if (ph * pw > max_px) continue;
      const int tiles_h = (g.Hm + th - 1) / th, tiles_w = (g.Wm + twp - 1) / twp;
      const double eff = (double)g.Hm * g.Wm / ((double)tiles_h * tiles_w * nf * 16.0) - 1e-3 * (ph * pw) / 180.0;
      if (eff > cand[k].eff) cand[k] = Cand{eff, th, twf, ph, pw, tiles_h, tiles_w};
    }
  }
  const int nctp = (g.Cd + tn * 16 - 1) / (tn * 16);
  auto wgs = [&](const Cand& c) { return (long)g.N * c.tiles_h * c.tiles_w * nctp; };
  int pick = 0;
  if (nfe) pick = (atoi(nfe) - 8) / 4;
  else if (slb == 64 && cand[2].eff > 0 && (long)g.N * cand[2].tiles_h * cand[2].tiles_w >= 1024 && cand[2].eff + 0.03 > cand[0].eff) pick = 2;
  else if (slb == 128 && cand[1].eff > 0 && cand[0].eff > 0 && wgs(cand[0]) <= 1024 && wgs(cand[1]) >= 384 && cand[1].eff + 0.05 > cand[0].eff) pick = 1;
  if (pick >= 0 && pick < 3 && cand[pick].eff > 0) {
    const Cand& c = cand[pick];
    best = c.eff; pp.tm = 2 + pick; pp.TH = c.th; pp.TWF = c.twf; pp.PH = c.ph; pp.PW = c.pw; pp.tiles_h = c.tiles_h; pp.tiles_w = c.tiles_w;
  }
  // narrow images (the deep layers: 18x38, 9x19 pixels): TH full-width rows per workgroup, 8 or 12 fragments
  for (int nf = 8; nf <= 12; nf += 4)
    for (int th = 1; th * g.Wm <= nf * 16 && th <= g.Hm; ++th) {
      const int ph = th + (dhmax - dhmin), pw = g.Wm + (dwmax - dwmin);
      if (ph * pw > max_px) continue;
      const int tiles_h = (g.Hm + th - 1) / th;
      const double eff = (double)g.Hm * g.Wm / ((double)tiles_h * nf * 16.0) - 1e-3 * (ph * pw) / 180.0 - 0.02;   // prefer 2-D tiles on a tie
      if (eff > best) { best = eff; pp.rows = 1; pp.tm = nf / 4; pp.TH = th; pp.TWF = 1; pp.PH = ph; pp.PW = pw; pp.tiles_h = tiles_h; pp.tiles_w = 1; }
    }
  if (best < 0.6) return false;
  pp.dhmin = dhmin; pp.dwmin = dwmin;
  const long spatial = (long)g.N * pp.tiles_h * pp.tiles_w;
  if (tn == 4 && spatial * ((g.Cd + 63) / 64) < 200) tn = 2;      // few tiles: 32-channel tiles double the workgroups
  pp.nct = (g.Cd + tn * 16 - 1) / (tn * 16);
  pp.lds = pp.PH * pp.PW * slb + 2 * tn * 16 * slb;
  pp.tapq[0] = pp.tapq[1] = pp.tapq[2] = 0;
  for (int t = 0; t < g.ntaps; ++t) {
    const int dh = (g.tap[t] & 255) - 64, dw = ((g.tap[t] >> 8) & 255) - 64, wt = g.tap[t] >> 16;
    const int toff = (dh - dhmin) * pp.PW + (dw - dwmin);      // patch pixel offset of the tap
    if (toff >= 4096 || wt >= 16) return false;
    pp.tapq[t >> 2] |= (unsigned long long)(toff | (wt << 12)) << ((t & 3) * 16);
  }
  pp.m_nct = pc_magic(pp.nct); pp.m_per_img = pc_magic(pp.tiles_h * pp.tiles_w); pp.m_tiles_w = pc_magic(pp.tiles_w);
  pp.m_twf = pc_magic(pp.TWF);
  pp.m_pw20 = (unsigned)(((1u << 20) + pp.PW - 1) / pp.PW); pp.m_twf20 = (unsigned)(((1u << 20) + g.Wm - 1) / g.Wm);
  // exactness of the multiply-shift divisions (see pc_div / pc_div20) and 24-bit operands of the address products
  if ((double)g.N * pp.tiles_h * pp.tiles_w * pp.nct * std::max(pp.nct, pp.tiles_h * pp.tiles_w) >= 4294967296.0) return false;
  if (pp.PW >= 512 || g.Wm >= 512 || g.Ws >= (1 << 12) || g.Hs >= (1 << 12) || g.Cs * ES >= (1 << 13)) return false;
  const char* mt = getenv("AST_PCONV_MIN_TILES");
  const long min_tiles = mt ? atol(mt) : 192;
  if (spatial * pp.nct < min_tiles) return false;                 // under-filled grids keep the K-split plans
  return pp.lds <= 64 * 1024;
}

template <typename T, int SLB, int TM, int TN>
int launch_pconv(const void* src, const void* wgt, const float* bias, void* dst, const ast_gather_t& g, const PconvPlan& pp, int flags,
                 float* ws, const void* bn_x, const float* bn_scale, const float* bn_shift, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    AST_HIP(hipFuncSetAttribute((const void*)pconv_kernel<T, SLB, TM, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr_set = true;
  }
  const int tiles = g.N * pp.tiles_h * pp.tiles_w * pp.nct;
  const unsigned src_bytes = (unsigned)((size_t)g.N * g.Hs * g.Ws * g.Cs * sizeof(T));
  const unsigned wgt_bytes = (unsigned)((size_t)g.Cd * g.wtaps * g.Cs * sizeof(T));
  // (a persistent variant -- <= 3 workgroups per CU walking their XCD's tiles with the next patch prefetched into
  // registers -- measured SLOWER, 41 -> 54 us on the 64->64-channel layer: the prefetch registers cost a wave per SIMD,
  // and what the kernel lacks is overlap between workgroups, not bandwidth)
  const int grid = (tiles + 7) / 8 * 8;
  hipLaunchKernelGGL((pconv_kernel<T, SLB, TM, TN>), dim3(grid), dim3(256), pp.lds, s, (const T*)src, (const T*)wgt, bias, (T*)dst,
                     g, pp, flags, ws, src_bytes, wgt_bytes, (const T*)bn_x, bn_scale, bn_shift);
  AST_CHECK_LAUNCH();
  return 0;
}

struct IgemmPlan { int bm, bn, kch, nsplit, kt_per_split, depth, kgroups; const void* bn_x; const float* bn_scale; const float* bn_shift; };

IgemmPlan plan_igemm(const ast_gather_t& g, int M, int dtype) {
  const int E = dtype == AST_BF16 ? 8 : 4;
  const int nchunks = g.ntaps * (g.Cs / E);
  IgemmPlan p;
  // tile plan from tools/igemm_sweep.py on MI355X (profiles/r01): 64-row tiles win on every layer of the
  // B=8 step (more workgroups per CU matter more than operand reuse at these sizes)
  if (g.Cd > 64) { p.bm = 64; p.bn = (M >= 30000) ? 128 : 64; }
  else if (g.Cd > 32) { p.bm = 64; p.bn = 64; }
  else if (g.Cd > 16) { p.bn = 32; p.bm = M >= 4096 ? 128 : 64; }
  else { p.bn = 16; p.bm = M >= 4096 ? 128 : 64; }
  p.kch = (p.bn >= 64 && nchunks >= 16) ? 8 : 4;
  const int KT = (nchunks + p.kch - 1) / p.kch;
  const long blocks = (long)((M + p.bm - 1) / p.bm) * ((g.Cd + p.bn - 1) / p.bn);
  p.nsplit = 1;
  p.kgroups = 1;
  // under-filled grid with a long K loop: split K inside the workgroup (sweep: b5 35 -> 24 us, b4 47 -> 41 us; b3, with
  // 684 tiles, loses)
  static const long kg_blocks = getenv("AST_IGEMM_KG_BLOCKS") ? atol(getenv("AST_IGEMM_KG_BLOCKS")) : 400;
  // K tiles of 64 elements per group and barrier (kch 8): 21.0 -> 19.0 us on average over the 25 such launches of a replayed
  // step (tools/knob_ab.sh; back-to-back isolated launches had preferred 32)
  static const int kg_kch = getenv("AST_IGEMM_KG_KCH") ? atoi(getenv("AST_IGEMM_KG_KCH")) : 8;
  if (p.bm == 64 && p.bn == 64 && blocks < kg_blocks && KT >= 16) { p.kgroups = 4; p.kch = kg_kch == 4 ? 4 : 8; }
  else if (blocks < 200 && KT >= 16 && (long)M * g.Cd <= (1L << 20)) p.nsplit = blocks < 120 ? 4 : 2;
  p.depth = 2;                                          // depth 4 measured no better (the loop is not latency-bound)
  if (const char* f = getenv("AST_IGEMM_FORCE")) {      // tuning aid: "bm,bn,kch,nsplit[,kgroups]"
    int a, b, c, d, e = 0;
    const int nf = sscanf(f, "%d,%d,%d,%d,%d", &a, &b, &c, &d, &e);
    if (nf >= 4) { p.bm = a; p.bn = b; p.kch = c; p.nsplit = std::max(1, d); p.depth = 2; p.kgroups = 1; }
    if (nf >= 5) p.kgroups = (e == 4 && a == 64 && b == 64) ? 4 : 1;
  }
  {
    const int KT2 = (nchunks + p.kch - 1) / p.kch;
    p.nsplit = std::max(1, std::min(p.nsplit, std::max(1, KT2)));
    p.kt_per_split = (KT2 + p.nsplit - 1) / p.nsplit;
    p.nsplit = KT2 > 0 ? (KT2 + p.kt_per_split - 1) / p.kt_per_split : 1;
  }
  if (p.kt_per_split < 1) p.kt_per_split = 1;
  return p;
}

// igemm_direct_kernel applies to narrow layers whose weights fit in registers (see the kernel)
bool direct_ok(const ast_gather_t& g, const IgemmPlan& p, int dtype) {
  static const bool enabled = !(getenv("AST_IGEMM_DIRECT") && atoi(getenv("AST_IGEMM_DIRECT")) == 0);
  const int E = dtype == AST_BF16 ? 8 : 4;
  const int nchunks = g.ntaps * (g.Cs / E);
  // measured per layer (tools/layer_profile.py, B=8 step): wins 15-30 % for <= 16 output channels and K <= 12 chunks
  // (2 M pixels x 8 ch x 72: 79 -> 59 us; 2.4 M x 16 x 72: 100 -> 75 us); loses for 32 channels or long K, where the
  // weight registers (TN x NKS fragments) push occupancy to 2 waves and every chunk still crosses L1 once (whole step:
  // 7.04 -> 6.91 ms with this rule, 7.10 with everything up to 32 channels x 36 chunks)
  static const int max_cd = getenv("AST_IGEMM_DIRECT_CD") ? atoi(getenv("AST_IGEMM_DIRECT_CD")) : 16;
  static const int max_chunks = getenv("AST_IGEMM_DIRECT_CHUNKS") ? atoi(getenv("AST_IGEMM_DIRECT_CHUNKS")) : 12;
  if ((long)g.N * g.Hd * g.Wd >= (1L << 31)) return false;     // the kernel keeps destination pixel indices in 32 bits
  return enabled && g.Cd <= std::min(max_cd, 32) && nchunks <= std::min(max_chunks, 12) && p.nsplit == 1 && p.kgroups == 1 &&
         !getenv("AST_IGEMM_FORCE");
}
int direct_jt(int M) {                                       // pixel tiles of 16 per wave: keep >= ~2048 workgroups when M allows
  static const int env = getenv("AST_IGEMM_DIRECT_JT") ? atoi(getenv("AST_IGEMM_DIRECT_JT")) : 0;
  if (env > 0) return std::min(env, 64);
  return std::max(1, std::min(8, M / (64 * 2048)));
}

template <typename T, int TN, int NKS>
int launch_direct(const void* src, const void* wgt, const float* bias, void* dst, const ast_gather_t& g, int M, int flags, float* ws,
                  const IgemmPlan& p, hipStream_t s) {
  const int E = 16 / sizeof(T), cpc = g.Cs / E;
  int shift = -1;
  if ((cpc & (cpc - 1)) == 0) { shift = 0; while ((1 << shift) < cpc) ++shift; }
  const int jt = direct_jt(M);
  const int tiles = ((M + 64 * jt - 1) / (64 * jt)) * ((g.Cd + TN * 16 - 1) / (TN * 16));
  const unsigned src_bytes = (unsigned)((size_t)g.N * g.Hs * g.Ws * g.Cs * sizeof(T));
  const unsigned wgt_bytes = (unsigned)((size_t)g.Cd * g.wtaps * g.Cs * sizeof(T));
  hipLaunchKernelGGL((igemm_direct_kernel<T, TN, NKS>), dim3((tiles + 7) / 8 * 8), dim3(256), 0, s, (const T*)src, (const T*)wgt, bias, (T*)dst, g,
                     M, flags, ws, shift, src_bytes, wgt_bytes, 1.0f / (float)(g.Hm * g.Wm), 1.0f / (float)g.Wm, (const T*)p.bn_x, p.bn_scale,
                     p.bn_shift, jt);
  AST_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int dispatch_direct(const void* src, const void* wgt, const float* bias, void* dst, const ast_gather_t& g, int M, int flags, float* ws,
                    const IgemmPlan& p, hipStream_t s) {
  const int nks = (g.ntaps * (g.Cs / (16 / (int)sizeof(T))) + 3) / 4;
#define AST_DK(TN_, K_) return launch_direct<T, TN_, K_>(src, wgt, bias, dst, g, M, flags, ws, p, s)
  if (g.Cd <= 16) { if (nks <= 1) AST_DK(1, 1); if (nks <= 2) AST_DK(1, 2); AST_DK(1, 3); }
  if (nks <= 1) AST_DK(2, 1); if (nks <= 2) AST_DK(2, 2); AST_DK(2, 3);
#undef AST_DK
}

template <typename T, int BM, int BN, int WM, int WN, int KCH, int D, int KG, bool UT>
int launch_igemm_ut(const void* src, const void* wgt, const float* bias, void* dst, const ast_gather_t& g, int M, int flags,
                 float* ws, const IgemmPlan& p, hipStream_t s) {
  constexpr int LDS = KG * 2 * (KCH / 4) * (BM + BN) * 64 + 64;
  static_assert(KG == 1 || (KG - 1) * (BM / 16) * (BN / 16) / 4 * 256 * 16 <= KG * 2 * (KCH / 4) * (BM + BN) * 64, "reduce buffer fits");
  static bool attr_set = false;
  if (!attr_set) {
    AST_HIP(hipFuncSetAttribute((const void*)igemm_kernel<T, BM, BN, WM, WN, KCH, D, KG, UT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  const int E = 16 / sizeof(T);
  const int cpc = g.Cs / E;
  int shift = -1;
  if ((cpc & (cpc - 1)) == 0) { shift = 0; while ((1 << shift) < cpc) ++shift; }
  const int mtiles = (flags & 64) ? g.N * ((g.Hm * g.Wm + BM - 1) / BM) : (M + BM - 1) / BM;     // bit 6: image-aligned tiles
  const int tiles = mtiles * ((g.Cd + BN - 1) / BN);
  dim3 grid((tiles + 7) / 8 * 8, 1, p.nsplit);
  if (p.nsplit > 1 && !(flags & 4)) AST_HIP(hipMemsetAsync(ws, 0, sizeof(float) * (size_t)M * g.Cd, s));
  const unsigned src_bytes = (unsigned)((size_t)g.N * g.Hs * g.Ws * g.Cs * sizeof(T));
  const unsigned wgt_bytes = (unsigned)((size_t)g.Cd * g.wtaps * g.Cs * sizeof(T));
  hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WM, WN, KCH, D, KG, UT>), grid, dim3(256 * KG), LDS, s, (const T*)src, (const T*)wgt, bias, (T*)dst, g, M,
                     flags, ws, p.kt_per_split, shift, src_bytes, wgt_bytes, 1.0f / (float)(g.Hm * g.Wm), 1.0f / (float)g.Wm,
                     (const T*)p.bn_x, p.bn_scale, p.bn_shift);
  if (p.nsplit > 1) {
    const size_t total = (size_t)M * (g.Cd >> 2);
    hipLaunchKernelGGL((splitk_finish_kernel<T>), dim3((unsigned)std::min<size_t>((total + 255) / 256, 2048)), dim3(256), 0, s, ws, bias,
                       (T*)dst, g, M, flags);
  }
  AST_CHECK_LAUNCH();
  return 0;
}

template <typename T, int BM, int BN, int WM, int WN, int KCH, int D, int KG>
int launch_igemm(const void* src, const void* wgt, const float* bias, void* dst, const ast_gather_t& g, int M, int flags,
                 float* ws, const IgemmPlan& p, hipStream_t s) {
  const int cpc = g.Cs / (16 / (int)sizeof(T));
  if (cpc % KCH == 0) return launch_igemm_ut<T, BM, BN, WM, WN, KCH, D, KG, true>(src, wgt, bias, dst, g, M, flags, ws, p, s);
  return launch_igemm_ut<T, BM, BN, WM, WN, KCH, D, KG, false>(src, wgt, bias, dst, g, M, flags, ws, p, s);
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
  if ((long)g->N * g->Hs * g->Ws * g->Cs * 4 >= (1L << 31) || (long)g->N * g->Hm * g->Wm >= (1L << 31) ||
      (long)g->Cd * g->wtaps * g->Cs * 4 >= (1L << 31)) AST_FAIL("%s: tensor exceeds the 2 GiB buffer-addressing range", who);
  return 0;
}

}  // namespace

extern "C" long ast_igemm_ws_floats(const ast_gather_t* gp, int dtype) {
  if (!gp || check_gather(gp, "ast_igemm_ws_floats")) return -1;
  const int M = gp->N * gp->Hm * gp->Wm;
  {
    PconvPlan pp; int slb = 0, tn = 0;
    if (plan_pconv(*gp, dtype, pp, slb, tn)) return 0;      // the patch kernel never splits K
  }
  const IgemmPlan p = plan_igemm(*gp, M, dtype);
  return p.nsplit > 1 ? (long)M * gp->Cd : 0;
}

extern "C" int ast_igemm_plan(const ast_gather_t* gp, int dtype, int* out5) {
  if (!gp || !out5 || check_gather(gp, "ast_igemm_plan")) return -1;
  const IgemmPlan p = plan_igemm(*gp, gp->N * gp->Hm * gp->Wm, dtype);
  out5[0] = p.bm; out5[1] = p.bn; out5[2] = p.kch; out5[3] = p.nsplit; out5[4] = p.kgroups;
  {
    PconvPlan pp; int slb = 0, tn = 0;
    if (plan_pconv(*gp, dtype, pp, slb, tn)) {  // patch kernel: reported as BM = -(tile rows), kch = -(slab bytes)
      out5[0] = -pp.TH; out5[1] = tn * 16; out5[2] = -slb; out5[3] = pp.rows ? -(pp.tm * 4) : 1; out5[4] = 1;
      return 0;
    }
  }
  if (direct_ok(*gp, p, dtype)) {                            // LDS-free narrow-layer kernel: kch = 0 marks it
    out5[0] = 64 * direct_jt(gp->N * gp->Hm * gp->Wm); out5[1] = gp->Cd <= 16 ? 16 : 32; out5[2] = 0;
  }
  return 0;
}

extern "C" int ast_igemm_bn(const void* src, const void* wgt, const float* bias, void* dst, const ast_gather_t* gp,
                            int dtype, int flags, float* ws, long ws_floats, const void* bn_x, const float* bn_scale,
                            const float* bn_shift, void* stream) {
  if (int rc = check_gather(gp, "ast_igemm")) return rc;
  if (!src || !wgt || !dst) AST_FAIL("ast_igemm: null pointer");
  const ast_gather_t g = *gp;
  const int M = g.N * g.Hm * g.Wm;
  hipStream_t s = (hipStream_t)stream;
  IgemmPlan p = plan_igemm(g, M, dtype);
  p.bn_x = bn_x; p.bn_scale = bn_scale; p.bn_shift = bn_shift;
  {
    PconvPlan pp; int slb = 0, tn = 0;
    if (plan_pconv(g, dtype, pp, slb, tn)) {
      if ((flags & 16) && ((flags & 11) || !ws || ws_floats < 64L * g.Cd * 3 || !bn_x || !bn_scale || !bn_shift))
        AST_FAIL("ast_igemm: fused BatchNorm-backward sums need plain stores, a zeroed [64][Cd][3] table and the layer's x / scale / shift");
      if ((flags & 8) && ((flags & 3) || !ws || ws_floats < ((flags & 64) ? (long)g.N : 64L) * g.Cd * 2)) AST_FAIL("ast_igemm: fused channel statistics need plain stores and a zeroed [64][Cd][2] (per image: [N][Cd][2]) table");
#define AST_PC(S_, M_, N_) return launch_pconv<T, S_, M_, N_>(src, wgt, bias, dst, g, pp, flags, ws, bn_x, bn_scale, bn_shift, s)
      AST_DISPATCH_T(dtype, {
        if (slb == 128 && pp.tm == 2) { if (tn == 4) AST_PC(128, 2, 4); AST_PC(128, 2, 2); }
        if (slb == 128) { if (tn == 4) AST_PC(128, 3, 4); AST_PC(128, 3, 2); }
        if (pp.tm == 2) { if (tn == 4) AST_PC(64, 2, 4); AST_PC(64, 2, 2); }
        if (pp.tm == 4) { if (tn == 4) AST_PC(64, 4, 4); AST_PC(64, 4, 2); }
        if (tn == 4) AST_PC(64, 3, 4); AST_PC(64, 3, 2);
      });
#undef AST_PC
    }
  }
  // argument checks of the gathered and direct kernels (the patch kernel's are above): a null or short workspace here is a
  // write through a bad device pointer -- a GPU fault, not an error code
  if (p.nsplit > 1) {
    if (flags & (8 | 16)) AST_FAIL("ast_igemm: fused statistics (flags 8 / 16) are not available for a split-K plan (ast_igemm_ws_floats > 0)");
    if (!ws || ws_floats < (long)M * g.Cd) AST_FAIL("ast_igemm: this plan splits K and needs a workspace of %ld floats (ast_igemm_ws_floats), got %ld", (long)M * g.Cd, ws ? ws_floats : 0L);
  }
  if ((flags & 16) && ((flags & 11) || !ws || ws_floats < 64L * g.Cd * 3 || !bn_x || !bn_scale || !bn_shift))
    AST_FAIL("ast_igemm: fused BatchNorm-backward sums need plain stores, a zeroed [64][Cd][3] table and the layer's x / scale / shift");
  if ((flags & 8) && !(flags & 64) && ((flags & 3) || !ws || ws_floats < 64L * g.Cd * 2))
    AST_FAIL("ast_igemm: fused channel statistics need plain stores and a zeroed [64][Cd][2] table");
  if (flags & 64) {
    if (!(flags & 8) || (flags & 3) || p.nsplit > 1 || direct_ok(g, p, dtype)) AST_FAIL("ast_igemm: per-image statistics (flag 64) need flag 8, plain stores and the gathered kernel (ast_igemm_plan: kch > 0, no split)");
    if (!ws || ws_floats < (long)g.N * g.Cd * 2) AST_FAIL("ast_igemm: per-image statistics need a zeroed [N][Cd][2] table");
  }
  if (direct_ok(g, p, dtype)) { AST_DISPATCH_T(dtype, { return dispatch_direct<T>(src, wgt, bias, dst, g, M, flags, ws, p, s); }); }
#define AST_IG(BM_, BN_, WM_, WN_, K_) return launch_igemm<T, BM_, BN_, WM_, WN_, K_, 2, 1>(src, wgt, bias, dst, g, M, flags, ws, p, s)
#define AST_IG4(BM_, BN_, WM_, WN_, K_) return launch_igemm<T, BM_, BN_, WM_, WN_, K_, 2, 4>(src, wgt, bias, dst, g, M, flags, ws, p, s)
  AST_DISPATCH_T(dtype, {
    if (p.bm == 128 && p.bn == 128) { if (p.kch == 8) AST_IG(128, 128, 2, 2, 8); else AST_IG(128, 128, 2, 2, 4); }
    if (p.bm == 128 && p.bn == 64) { if (p.kch == 8) AST_IG(128, 64, 2, 2, 8); else AST_IG(128, 64, 2, 2, 4); }
    if (p.bm == 64 && p.bn == 64 && p.kgroups == 4) { if (p.kch == 8) AST_IG4(64, 64, 2, 2, 8); else AST_IG4(64, 64, 2, 2, 4); }
    if (p.bm == 64 && p.bn == 64) { if (p.kch == 8) AST_IG(64, 64, 2, 2, 8); else AST_IG(64, 64, 2, 2, 4); }
    if (p.bm == 64 && p.bn == 128) { if (p.kch == 8) AST_IG(64, 128, 1, 4, 8); else AST_IG(64, 128, 1, 4, 4); }
    if (p.bm == 128 && p.bn == 32) AST_IG(128, 32, 4, 1, 4);
    if (p.bm == 128 && p.bn == 16) AST_IG(128, 16, 4, 1, 4);
    if (p.bm == 256 && p.bn == 32) AST_IG(256, 32, 4, 1, 4);
    if (p.bm == 64 && p.bn == 32) AST_IG(64, 32, 4, 1, 4);
    if (p.bm == 256 && p.bn == 16) AST_IG(256, 16, 4, 1, 4);
    AST_IG(64, 16, 4, 1, 4);
  });
#undef AST_IG
#undef AST_IG4
  return 0;
}

extern "C" int ast_igemm(const void* src, const void* wgt, const float* bias, void* dst, const ast_gather_t* gp,
                         int dtype, int flags, float* ws, long ws_floats, void* stream) {
  return ast_igemm_bn(src, wgt, bias, dst, gp, dtype, flags & ~48, ws, ws_floats, nullptr, nullptr, nullptr, stream);
}

extern "C" int ast_wgrad(const void* dy, const void* src, float* dw, const ast_gather_t* gp, int dtype, void* stream);
extern "C" int ast_wgrad_rep(const void* dy, const void* src, float* dw, const ast_gather_t* gp, int dtype, int nrep, void* stream) {
  if (nrep < 1 || nrep > 64 || !gp) AST_FAIL("ast_wgrad_rep: 1..64 replicas");
  g_wg_nrep = nrep;
  g_wg_rep_stride = (long)gp->Cd * gp->wtaps * gp->Cs;
  const int rc = ast_wgrad(dy, src, dw, gp, dtype, stream);
  g_wg_nrep = 1; g_wg_rep_stride = 0;
  return rc;
}

extern "C" int ast_wgrad(const void* dy, const void* src, float* dw, const ast_gather_t* gp, int dtype, void* stream) {
  if (int rc = check_gather(gp, "ast_wgrad")) return rc;
  if (!dy || !src || !dw) AST_FAIL("ast_wgrad: null pointer");
  const ast_gather_t g = *gp;
  if (g.ntaps == 0) return 0;
  const int P = g.N * g.Hm * g.Wm;
  if ((long)P * g.Cd * 4 >= (1L << 31)) AST_FAIL("ast_wgrad: dy exceeds the 2 GiB buffer-addressing range");
  hipStream_t s = (hipStream_t)stream;
  const int nct_all = (g.ntaps * g.Cs + 15) / 16;          // column tiles of the whole (tap, channel) space
  const int bmw = g.Cd > 32 ? 64 : (g.Cd > 16 ? 32 : 16);
  // all columns in one workgroup when the accumulators fit (<= 20 tiles x BMW/16 row tiles <= 20 per wave)
  int nct;
  if (bmw == 64) nct = nct_all <= 8 ? (nct_all <= 4 ? 4 : 8) : 12;
  else nct = nct_all <= 4 ? 4 : (nct_all <= 8 ? 8 : (nct_all <= 12 ? 12 : 20));
  WHaloPlan whp;
  const bool halo = plan_wgrad_halo(g, dtype, nct, bmw, whp);
#define AST_WG(B_, N_) do { if (halo) return launch_wgrad_halo<T, B_, N_>(dy, src, dw, g, whp, s); \
                            return launch_wgrad<T, B_, N_>(dy, src, dw, g, P, s); } while (0)
  AST_DISPATCH_T(dtype, {
    if (bmw == 64) { if (nct == 4) AST_WG(64, 4); if (nct == 8) AST_WG(64, 8); AST_WG(64, 12); }
    if (bmw == 32) { if (nct == 4) AST_WG(32, 4); if (nct == 8) AST_WG(32, 8); if (nct == 12) AST_WG(32, 12); AST_WG(32, 20); }
    if (nct == 4) AST_WG(16, 4); if (nct == 8) AST_WG(16, 8); if (nct == 12) AST_WG(16, 12); AST_WG(16, 20);
  });
#undef AST_WG
  return 0;
}
