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
#include "gemm_common.h"

namespace {


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
// flags bits 8-11: log2 of the number of statistics slots (0 = 64): wide layers take fewer slots so that the consumer, which
// reduces the table itself (ast_bn_apply_fwd / _bwd), reads at most C x slots = 4096 entries
__host__ __device__ __forceinline__ int stat_slot_mask(const int flags) { const int l = (flags >> 8) & 15; return l ? (1 << l) - 1 : 63; }

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
  if (stats || bstats) epi_flush<T, TN>(ec, g, st1, st2, per_image ? ~img : (tix & stat_slot_mask(flags)), bn0 + wn * WTN, fr, fq, red);
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
  if (stats || bstats) epi_flush<T, TN>(ec, g, st1, st2, tix & stat_slot_mask(flags), bn0, fr, fq, red);
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
struct PconvPlan { int TH, TWF, PH, PW, dhmin, dwmin, tiles_h, tiles_w, nct, lds, rows, tm, wall; unsigned m_nct, m_per_img, m_tiles_w, m_pw20, m_twf, m_twf20; unsigned long long tapq[3]; };
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

// WALL ("weights of all taps"): the per-tap weight stage above costs one workgroup barrier and one exposed weight-load latency
// per tap for TM*TN*KS MFMAs per wave -- 12 of them (192 cycles) on the 512-channel layer, where a tap step measures ~800 cycles.
// With WALL the weights of ALL taps of a channel slab sit in LDS beside the patch (ntaps x BN x SLB bytes), the next slab's patch
// and weights are fetched into registers while this slab's ntaps x TM x TN x KS MFMAs run, and a slab costs two barriers
// (hand-over of the registers to LDS) instead of ntaps.
template <typename T, int SLB, int TM, int TN, bool WALL>
__global__ __launch_bounds__(256, (WALL ? 2 : (TM * TN <= 8 ? 4 : 2))) void pconv_kernel(const T* __restrict__ src, const T* __restrict__ wgt, const float* __restrict__ bias,
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
  if constexpr (WALL) {
    u32x4 pr[PC_MAXPL], wra[AST_MAX_TAPS][NWL];
    auto fetch = [&](int s) __attribute__((always_inline)) {
      const unsigned sb = (unsigned)(s * SLB);
#pragma unroll
      for (int i = 0; i < PC_MAXPL; ++i) pr[i] = __builtin_amdgcn_raw_buffer_load_b128(srcR, goff[i] + sb, 0, 0);
#pragma unroll
      for (int t = 0; t < AST_MAX_TAPS; ++t)
        if (t < g.ntaps) {
          const unsigned w0 = (tap_entry(t) >> 12) * wslice + sb;
#pragma unroll
          for (int k = 0; k < NWL; ++k) wra[t][k] = __builtin_amdgcn_raw_buffer_load_b128(wgtR, woff[k] + w0, 0, 0);
        }
    };
    fetch(0);
    for (int s = 0; s < nslab; ++s) {
      // (the previous slab ended with a barrier: every read of the patch and of the weights is done)
#pragma unroll
      for (int i = 0; i < PC_MAXPL; ++i)
        if (p0 + i * PSTEP < npix) *reinterpret_cast<u32x4*>(pl + l0 + i * (PSTEP * SLB)) = pr[i];
#pragma unroll
      for (int t = 0; t < AST_MAX_TAPS; ++t)
        if (t < g.ntaps) {
#pragma unroll
          for (int k = 0; k < NWL; ++k)
            if (wl[k] >= 0) *reinterpret_cast<u32x4*>(wbuf + t * (BN * SLB) + wl[k]) = wra[t][k];
        }
      __syncthreads();
      if (s == 0) PC_STAMP(2);
      if (s + 1 < nslab) fetch(s + 1);              // in flight under this slab's MFMAs
      __builtin_amdgcn_sched_barrier(0);            // (keep the loads here: the scheduler would sink them to their first use)
      for (int t = 0; t < g.ntaps; ++t) {
        const int toff = (int)(tap_entry(t) & 0xfffu);
        const unsigned char* wcur = wbuf + t * (BN * SLB);
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
      }
      __syncthreads();
    }
  } else {
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
  if (stats || bstats) epi_flush<TN>(ws, bstats, g.Cd, st1, st2, (flags & 64) ? ~n : (tix & stat_slot_mask(flags)), bn0, fr, fq,
                                       ((flags & 64) || stat_slot_mask(flags) < 63) ? reinterpret_cast<float*>(pl) : nullptr);   // per-image tables only: with 64 slots the per-wave atomics are as fast (and one barrier pair cheaper)      // bit 6: per-image slots (tiles never straddle images)
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
  // tiny images (5x10: the last block): one 4-fragment tile per image, only where nothing larger fits
  if (best < 0.6 && slb == 128)
    for (int th = 1; th * g.Wm <= 64 && th <= g.Hm; ++th) {
      const int ph = th + (dhmax - dhmin), pw = g.Wm + (dwmax - dwmin);
      if (ph * pw > max_px) continue;
      const int tiles_h = (g.Hm + th - 1) / th;
      const double eff = (double)g.Hm * g.Wm / ((double)tiles_h * 64.0) - 1e-3 * (ph * pw) / 180.0 - 0.02;
      if (eff > best) { best = eff; pp.rows = 1; pp.tm = 1; pp.TH = th; pp.TWF = 1; pp.PH = ph; pp.PW = pw; pp.tiles_h = tiles_h; pp.tiles_w = 1; }
    }
  if (best < 0.6) return false;
  pp.dhmin = dhmin; pp.dwmin = dwmin;
  const long spatial = (long)g.N * pp.tiles_h * pp.tiles_w;
  if (tn == 4 && spatial * ((g.Cd + 63) / 64) < 200) tn = 2;      // few tiles: 32-channel tiles double the workgroups
  // WALL (see the kernel): all taps' weights of a slab resident, the next slab prefetched under this slab's MFMAs.  It pays where
  // the per-tap steps are short AND there are slabs to overlap: >= 4 channel slabs, with 32-channel tiles so that two workgroups
  // still share a CU (<= 80 KB).  Isolated, 512 channels: 39.7 -> 26.2 us (492 TF/s); forced onto the 64 / 128 / 256-channel layers
  // with their 64-channel tiles (one workgroup per CU, one or two slabs) it loses: 28.0 -> 51.9, 23.2 -> 32.3, 30.7 -> 44.1 us
  // (profiles/r03/pconv_wall_layers.txt).  AST_PCONV_WALL = 0: off, 1: wherever it fits, 2: wherever it fits with 32-channel tiles.
  {
    const char* we = getenv("AST_PCONV_WALL");
    const int mode = we ? atoi(we) : -1, nslab = rowb / slb;
    const int patch = pp.PH * pp.PW * slb;
    pp.wall = 0;
    if (dtype == AST_BF16 && g.ntaps >= 3 && mode != 0) {
      if (mode == 1) pp.wall = patch + g.ntaps * tn * 16 * slb <= 150 * 1024;
      else if ((mode == 2 || nslab >= 4) && patch + g.ntaps * 32 * slb <= 80 * 1024) { pp.wall = 1; tn = 2; }
    }
    pp.lds = patch + (pp.wall ? g.ntaps : 2) * tn * 16 * slb;
  }
  pp.nct = (g.Cd + tn * 16 - 1) / (tn * 16);
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
  return pp.lds <= (pp.wall ? 150 : 64) * 1024;
}

template <typename T, int SLB, int TM, int TN, bool WALL>
int launch_pconv(const void* src, const void* wgt, const float* bias, void* dst, const ast_gather_t& g, const PconvPlan& pp, int flags,
                 float* ws, const void* bn_x, const float* bn_scale, const float* bn_shift, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    AST_HIP(hipFuncSetAttribute((const void*)pconv_kernel<T, SLB, TM, TN, WALL>, hipFuncAttributeMaxDynamicSharedMemorySize, (WALL ? 150 : 64) * 1024));
    attr_set = true;
  }
  const int tiles = g.N * pp.tiles_h * pp.tiles_w * pp.nct;
  const unsigned src_bytes = (unsigned)((size_t)g.N * g.Hs * g.Ws * g.Cs * sizeof(T));
  const unsigned wgt_bytes = (unsigned)((size_t)g.Cd * g.wtaps * g.Cs * sizeof(T));
  // (a persistent variant -- <= 3 workgroups per CU walking their XCD's tiles with the next patch prefetched into
  // registers -- measured SLOWER, 41 -> 54 us on the 64->64-channel layer: the prefetch registers cost a wave per SIMD,
  // and what the kernel lacks is overlap between workgroups, not bandwidth)
  const int grid = (tiles + 7) / 8 * 8;
  hipLaunchKernelGGL((pconv_kernel<T, SLB, TM, TN, WALL>), dim3(grid), dim3(256), pp.lds, s, (const T*)src, (const T*)wgt, bias, (T*)dst,
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
      if ((flags & 16) && ((flags & 11) || !ws || ws_floats < (stat_slot_mask(flags) + 1L) * g.Cd * 3 || !bn_x || !bn_scale || !bn_shift))
        AST_FAIL("ast_igemm: fused BatchNorm-backward sums need plain stores, a zeroed [64][Cd][3] table and the layer's x / scale / shift");
      if ((flags & 8) && ((flags & 3) || !ws || ws_floats < ((flags & 64) ? (long)g.N : stat_slot_mask(flags) + 1L) * g.Cd * 2)) AST_FAIL("ast_igemm: fused channel statistics need plain stores and a zeroed [64][Cd][2] (per image: [N][Cd][2]) table");
#define AST_PC(S_, M_, N_) do { if (pp.wall) return launch_pconv<T, S_, M_, N_, true>(src, wgt, bias, dst, g, pp, flags, ws, bn_x, bn_scale, bn_shift, s); \
                                return launch_pconv<T, S_, M_, N_, false>(src, wgt, bias, dst, g, pp, flags, ws, bn_x, bn_scale, bn_shift, s); } while (0)
      AST_DISPATCH_T(dtype, {
        if (slb == 128 && pp.tm == 1) { if (tn == 4) AST_PC(128, 1, 4); AST_PC(128, 1, 2); }
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
  if ((flags & 16) && ((flags & 11) || !ws || ws_floats < (stat_slot_mask(flags) + 1L) * g.Cd * 3 || !bn_x || !bn_scale || !bn_shift))
    AST_FAIL("ast_igemm: fused BatchNorm-backward sums need plain stores, a zeroed [64][Cd][3] table and the layer's x / scale / shift");
  if ((flags & 8) && !(flags & 64) && ((flags & 3) || !ws || ws_floats < (stat_slot_mask(flags) + 1L) * g.Cd * 2))
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
