// Weight gradients of the gathered GEMMs (Conv2d / ConvTranspose2d / Linear): dw[cd][wtap][c] += sum_pix dy[pix][cd] * src[gather(pix,tap)][c].
// Split from igemm.hip (same data layout and geometry struct: see its header comment).
#include "gemm_common.h"

namespace {

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
// The flush of a workgroup's dW tile.  D (MFMA accumulator) layout: lane (li, gq) holds rows 16 i + 4 gq + r, column 16 ct + li, so a
// wave-instruction of atomics straight from the registers touched 4 rows x 64 B.  The memory-side atomic units take their full
// rate only for 256 contiguous bytes (or 2 x 128 B) per wave-instruction (MI355X_MICROARCH.md "Global float atomics"; the
// 4 x 64 B shape measured 0.87 TB/s of added bytes here against ~1.3 TB/s): the tile goes through LDS (free at this point) and
// every wave adds whole 64-float row segments -- 64 consecutive (tap, channel) columns of one output channel = 256 contiguous
// bytes when the source has >= 64 channels, two 128-byte runs for 32.  Ends with every thread of the (remaining) workgroup.
// mode 0: atomics from the registers (default), 1: atomics in 256-byte rows through LDS, 2: PLAIN STORES in 256-byte rows through LDS
// (slab mode: the pixel slice owns its copy of dW -- ast_wgrad_slab -- so nothing else adds to these addresses in this launch).
template <int BMW, int NCT, int RT, int CTW, typename WtOf>
__device__ __forceinline__ void flush_tile_rows(const f32x4 (&acc)[RT][CTW], float* tile, float* dw, const ast_gather_t& g, const int cd0,
                                                const int col0, const int ncols, const int wave, const int lane, WtOf wt_of, const int mode) {
  constexpr int BNW = NCT * 16;
  const int li = lane & 15, gq = lane >> 4;
  if (mode == 0) {                                      // A/B (AST_WGRAD_FLUSH_LDS=0): atomics straight from the accumulator registers
#pragma unroll
    for (int j = 0; j < CTW; ++j) {
      const int ct = wave + 4 * j;
      const int col = col0 + ct * 16 + li;
      if (ct >= NCT || col >= ncols) continue;
      const int t = col / g.Cs, c = col - t * g.Cs;
      const int wtc = wt_of(t);
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cd = cd0 + i * 16 + gq * 4 + r;
          if (cd < g.Cd) unsafeAtomicAdd(dw + ((size_t)cd * g.wtaps + wtc) * g.Cs + c, acc[i][j][r]);
        }
    }
    return;
  }
  __syncthreads();                                   // the staging / reduction area is free
#pragma unroll
  for (int j = 0; j < CTW; ++j) {
    const int ct = wave + 4 * j;
    if (ct < NCT) {
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(i * 16 + gq * 4 + r) * BNW + ct * 16 + li] = acc[i][j][r];
    }
  }
  __syncthreads();
#pragma unroll
  for (int cc = 0; cc < BNW / 64; ++cc) {
    const int col = col0 + cc * 64 + lane;
    if (col >= ncols) continue;
    const int t = col / g.Cs, c = col - t * g.Cs;
    float* dcol = dw + (size_t)wt_of(t) * g.Cs + c;
    if (mode == 2) {
      for (int row = wave; row < BMW; row += 4) {
        const int cd = cd0 + row;
        if (cd < g.Cd) dcol[(size_t)cd * g.wtaps * g.Cs] = tile[row * BNW + cc * 64 + lane];
      }
    } else {
      for (int row = wave; row < BMW; row += 4) {
        const int cd = cd0 + row;
        if (cd < g.Cd) unsafeAtomicAdd(dcol + (size_t)cd * g.wtaps * g.Cs, tile[row * BNW + cc * 64 + lane]);
      }
    }
  }
}

// Default: atomics straight from the accumulator registers.  The through-LDS form (AST_WGRAD_FLUSH_LDS=1: 256 contiguous bytes per
// atomic wave-instruction) measured 1-1.5 us SLOWER per layer with 8 gradient replicas (64-channel layer 40.1 -> 41.5 us, 128: 41.0 ->
// 42.2, 256: 41.8 -> 43.3; profiles/r03/wg_flush.txt): at ~11 adders per address the 4 x 64 B shape is not what bounds the flush.
inline bool wg_flush_direct() { const char* e = getenv("AST_WGRAD_FLUSH_LDS"); return !(e && atoi(e) != 0); }
static thread_local int g_wg_nrep = 1;
static thread_local int g_wg_slab = 0;            // ast_wgrad_slab: g_wg_nrep slabs, one per pixel slice, plain stores
static thread_local int g_wg_slices = 0;          // out: pixel slices of the last launch
inline int wg_nrep_arg() { return g_wg_nrep | ((g_wg_slab ? 2 : (wg_flush_direct() ? 0 : 1)) << 16); }
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
  constexpr int TILE = BMW * BNW * 4;                                          // the [row][column] f32 image of the flush
  constexpr int TAP_OFF = std::max(std::max(PG * GROUP_BYTES, RED), TILE);
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
  dw += (size_t)(bz % (nrep & 0xffff)) * rep_stride;          // this pixel slice's gradient replica / slab (nrep bits 16-17: flush mode)
  flush_tile_rows<BMW, NCT, RT, CTW>(acc, reinterpret_cast<float*>(wl_all), dw, g, cd0, col0, ncols, wave, lane,
                                     [&](int t) { return taptab[t] >> 16; }, (nrep >> 16) & 3);
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
  constexpr int LDS = std::max(std::max(PG * GROUP, RED), BMW * NCT * 16 * 4) + 64;
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
  if (g_wg_slab) nsplit = std::min(nsplit, g_wg_nrep);         // slab mode: one copy of dW per pixel slice
  int pps = (P + nsplit - 1) / nsplit;
  pps = (pps + BKP - 1) / BKP * BKP;
  nsplit = (P + pps - 1) / pps;
  const unsigned dy_bytes = (unsigned)((size_t)P * g.Cd * sizeof(T));
  const unsigned src_bytes = (unsigned)((size_t)g.N * g.Hs * g.Ws * g.Cs * sizeof(T));
  const int total = gx * gy * nsplit;
  g_wg_slices = nsplit;
  hipLaunchKernelGGL((wgrad_kernel<T, BMW, NCT, PG>), dim3((total + 7) / 8 * 8), dim3(256 * PG), LDS, s, (const T*)dy, (const T*)src, dw, g, P, pps,
                     dy_bytes, src_bytes, 1.0f / (float)(g.Hm * g.Wm), 1.0f / (float)g.Wm, gx, gy, nsplit, wg_nrep_arg(), g_wg_rep_stride);
  AST_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------
// wgrad_rows_kernel (bf16; 3x3 stride-1 "same" convolutions with Cs and Cd multiples of 64: conv2 of every ResBlock).
// What bounds the gathered kernel above is the number of cache LINES its loads touch, not bytes or latency: a K trip of 64 pixels
// x 192 columns fetches 64 x (128 B of dy + 3 taps x 128 B of source) as 16-byte chunks laid out 4 lanes to a 64-byte block --
// 512 half-line touches per trip, ~4 cycles each in the CU's texture path = the 2 000 cycles a trip measures (0.86-1.1 us with
// one or two workgroups on the CU, from L2 or from HBM alike, with register staging or with an LDS-DMA ring: profiles/r03/
// wg_ring_layers.txt).  The three kw taps of a pixel are the SAME three source lines its neighbours read, so this kernel stages
// source LINES, not gathered columns:
//   workgroup = 64 output channels x (one kernel row kh, its three kw taps, 64 source channels) x a slice of the pixels;
//   K trip = 64 consecutive pixels p0.. of the flattened (n, h, w) grid: 64 dy lines (128 B = 64 channels each) and the 66 source
//   lines p0 + (kh-1) W - 1 ... + 64 -- in a stride-1 "same" convolution tap (kh, kw) of pixel p is line p + (kh-1) W + (kw-1) of
//   the flattened source, across row and image ends too, where the convolution wants zeros instead: a 16-bit AND mask per
//   (pixel, kw) -- h + kh - 1 and w + kw - 1 inside the image -- computed by one wave per trip, kept beside the tile and applied
//   to the B fragments (8 consecutive pixels of one channel per lane = one ds_read_b128 of masks);
//   130 full-line touches per trip instead of 512 halves, by LDS-DMA (8 lanes x 16 B = one line, 8 lines per wave-instruction)
//   into a ring of four stages, one s_barrier per trip, counted s_waitcnt vmcnt (tiles t+1 .. t+3 in flight under tile t's MFMAs).
// LDS images are [line][128 B]; the MFMA operands are read transposed (ds_read_b64_tr_b16) at a 128-byte pitch, so the 16-byte
// chunk c of line r is kept at position c ^ row_swz(r) -- applied on the SOURCE side of the DMA, whose LDS side is lane-linear --
// which spreads any 8 rows {a .. a+3, a+8 .. a+11} that a half-wave reads over all 64 banks.
// All reads and writes of the ring are inline assembly with hand-placed waits: the compiler puts s_waitcnt vmcnt(0) in front of
// every LDS access it can see while an LDS-DMA is in flight.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int row_swz(int r) { return (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1; }

template <int NST>
__global__ __launch_bounds__(256, 2) void wgrad_rows_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ src, float* __restrict__ dw,
                                                            const ast_gather_t g, const int P, const int pps, const unsigned dy_bytes,
                                                            const unsigned src_bytes, const float rcp_hw, const float rcp_w, const int gx,
                                                            const int gy, const int gz, const int nrep, const long rep_stride) {
  constexpr int RT = 4, CTW = 3;
  constexpr int DYB = 64 * 128, XB = 72 * 128, MB = 3 * 64 * 2;      // dy lines, source lines (66 used), masks [kw][pixel] u16
  static_assert(NST >= 2 && NST <= 4, "ring stages");
  constexpr int STAGE = DYB + XB + MB;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char wq_all[];
  constexpr int RING = NST * STAGE > 64 * 192 * 4 ? NST * STAGE : 64 * 192 * 4;      // the flush tile overlays the ring
  int* taptab = reinterpret_cast<int*>(wq_all + RING);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = gridDim.x >> 3;                 // XCD-aware order: the workgroups that share a dy slice sit on one XCD
  const int tix = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (tix >= gx * gy * gz) return;
  const int bx = tix / (gz * gy), bz = (tix / gy) % gz, by = tix % gy;
  const int kh = by % 3, cs0 = (by / 3) * 64, cd0 = bx * 64;
  const int W = g.Wm, H = g.Hm;
  WG_STAMP(0); WG_STAMP(7);
  const int p_begin = bz * pps, p_end = min(P, p_begin + pps);
  const __amdgpu_buffer_rsrc_t dyR = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srcR = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
#pragma unroll
  for (int t = 0; t < AST_MAX_TAPS; ++t)
    if (tid == t) taptab[t] = g.tap[t];
  __syncthreads();                                  // (before any DMA is in flight)
  const int wt0 = taptab[kh * 3] >> 16, wt1 = taptab[kh * 3 + 1] >> 16, wt2 = taptab[kh * 3 + 2] >> 16;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)wq_all;

  // loader role: wave w fills lines 16w .. 16w+15 of both images (two wave-instructions of 8 lines each); wave 0 also the two halo
  // lines 64, 65 of the source.  Lane = (line lr of the instruction, chunk position cp); it FETCHES chunk cp ^ row_swz(line).
  const int lr = lane >> 3, cp = lane & 7;
  int dyp[2], xgl[3];                               // pixel of the lane's dy line / flattened source line, at trip 0
  unsigned dyo[2], xo[3];                           // their byte offsets
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int line = a < 2 ? 16 * wave + 8 * a + lr : 64 + lr;
    const int c = cp ^ row_swz(line);
    if (a < 2) {
      dyp[a] = p_begin + line;
      dyo[a] = (unsigned)((dyp[a] * g.Cd + cd0) * 2 + c * 16);
    }
    xgl[a] = p_begin + (kh - 1) * W - 1 + line;
    xo[a] = (unsigned)((xgl[a] * g.Cs + cs0) * 2 + c * 16);
  }
  auto issue_tile = [&](int kt) __attribute__((always_inline)) {
    unsigned char* st = wq_all + (kt % NST) * STAGE;
    const int adv = kt * 64;
#pragma unroll
    for (int a = 0; a < 2; ++a)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dyR, (lds_void_t*)(st + (2 * wave + a) * 1024), 16,
                                               dyp[a] + adv < p_end ? dyo[a] + (unsigned)(adv * g.Cd * 2) : OOB, 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 2; ++a)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srcR, (lds_void_t*)(st + DYB + (2 * wave + a) * 1024), 16,
                                               (unsigned)(xgl[a] + adv) < (unsigned)P ? xo[a] + (unsigned)(adv * g.Cs * 2) : OOB, 0, 0, 0);
    if (wave == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srcR, (lds_void_t*)(st + DYB + 8 * 1024), 16,
                                               (lr < 2 && (unsigned)(xgl[2] + adv) < (unsigned)P) ? xo[2] + (unsigned)(adv * g.Cs * 2) : OOB, 0, 0, 0);
    if (wave == 1) {                                // the tile's masks: lane = pixel
      const int p = p_begin + adv + lane;
      const int n = fdiv(p, H * W, rcp_hw), rem = p - n * H * W;
      const int h = fdiv(rem, W, rcp_w), w = rem - h * W;
      const bool hv = kh == 0 ? h >= 1 : (kh == 2 ? h <= H - 2 : true);
      const unsigned m0 = hv && w >= 1 ? 0xffffu : 0u, m1 = hv ? 0xffffu : 0u, m2 = hv && w <= W - 2 ? 0xffffu : 0u;
      const unsigned ma = lds0 + (kt % NST) * STAGE + DYB + XB + lane * 2;
      asm volatile("ds_write_b16 %0, %1\n\tds_write_b16 %0, %2 offset:128\n\tds_write_b16 %0, %3 offset:256" :: "v"(ma), "v"(m0), "v"(m1), "v"(m2) : "memory");
    }
  };

  f32x4 acc[RT][CTW];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // consumer role: wave w owns the 16 source channels 16w .. 16w+15 of each of the three kw taps (column tile j = kw) and all four
  // 16-row tiles of the output channels.  Lane (li, gq): rows 8 gq + (li >> 2) (+ 4) of a 32-pixel k-step, 8 bytes (li & 3) of the tile.
  const int li = lane & 15, gq = lane >> 4, q4 = li >> 2;
  const int sub = (li & 1) * 8, ch = (li & 3) >> 1;
  unsigned aoff[RT], boff[CTW][2];                  // byte offsets inside a stage (k-step 0; k-step 1 = + 32 lines = + 4096)
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int r = 8 * gq + q4;                      // row_swz(r) == row_swz(r + 4) == row_swz(r + 32)
    aoff[i] = r * 128 + (((2 * i + ch) ^ row_swz(r)) << 4) + sub;
  }
#pragma unroll
  for (int j = 0; j < CTW; ++j)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int r = 8 * gq + q4 + j + 4 * hh;       // source line of pixel row (8 gq + q4 + 4 hh), tap kw = j
      boff[j][hh] = DYB + r * 128 + (((2 * wave + ch) ^ row_swz(r)) << 4) + sub;
    }
  const unsigned moff = DYB + XB + 16 * gq;         // masks of the lane's 8 pixels: [kw][32 ks + 8 gq ..] u16

  bf16x4 alo[2][RT], ahi[2][RT], blo[2][CTW], bhi[2][CTW];
  u32x4v mk[2][CTW];
  auto read_kstep = [&](unsigned st, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < RT; ++i)
      asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:512" : "=&v"(alo[ks][i]), "=&v"(ahi[ks][i]) : "v"(st + ks * 4096 + aoff[i]));
#pragma unroll
    for (int j = 0; j < CTW; ++j) {
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(blo[ks][j]) : "v"(st + ks * 4096 + boff[j][0]));
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bhi[ks][j]) : "v"(st + ks * 4096 + boff[j][1]));
      asm volatile("ds_read_b128 %0, %1" : "=v"(mk[ks][j]) : "v"(st + moff + ks * 64 + j * 128));
    }
  };
  auto wait_kstep = [&](int ks) __attribute__((always_inline)) {               // every LDS read issued so far has returned; ties the
#pragma unroll                                                                // fragments to the wait so that no use is scheduled above it
    for (int i = 0; i < RT; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(alo[ks][i]), "+v"(ahi[ks][i]));
#pragma unroll
    for (int j = 0; j < CTW; ++j) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(blo[ks][j]), "+v"(bhi[ks][j]), "+v"(mk[ks][j]));
  };
  auto mma_kstep = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < CTW; ++j) {
      const u32x2v lo = __builtin_bit_cast(u32x2v, blo[ks][j]) & u32x2v{mk[ks][j][0], mk[ks][j][1]};
      const u32x2v hi = __builtin_bit_cast(u32x2v, bhi[ks][j]) & u32x2v{mk[ks][j][2], mk[ks][j][3]};
      const bf16x8 bf = __builtin_bit_cast(bf16x8, u32x4v{lo[0], lo[1], hi[0], hi[1]});
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const bf16x8 af = bf16x8{alo[ks][i][0], alo[ks][i][1], alo[ks][i][2], alo[ks][i][3], ahi[ks][i][0], ahi[ks][i][1], ahi[ks][i][2], ahi[ks][i][3]};
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[i][j], 0, 0, 0);
      }
    }
  };

  const int nk = (p_end - p_begin + 63) / 64;
  for (int t = 0; t < NST - 1; ++t)
    if (t < nk) issue_tile(t);
  WG_STAMP(1);
#ifdef AST_STAMPS
  long long ph[5] = {0, 0, 0, 0, 0};              // cycles of thread 0 in: wait for the tile's DMAs, barrier, reads of k-step 0 + DMA issue, reads 1 + MFMAs 0, MFMAs 1
  long long tph = __builtin_readcyclecounter();
#endif
  for (int kt = 0; kt < nk; ++kt) {
    // tiles kt+1 .. min(kt+2, nk-1) may stay in flight; this wave's DMAs of tile kt must have landed (vmcnt counts them in issue
    // order: 5 per tile on wave 0, 4 on the others) and its mask writes too
    const int younger = min(NST - 2, nk - 1 - kt);
    if (wave == 0) {
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    WG_PH(0);
    __builtin_amdgcn_s_barrier();                   // every wave's pieces of tile kt are in LDS; stage (kt+3) % 4 (read in trip kt-1) is free
    asm volatile("" ::: "memory");
    WG_PH(1);
    const unsigned st = lds0 + (kt % NST) * STAGE;
    read_kstep(st, 0);
    if (kt + NST - 1 < nk) issue_tile(kt + NST - 1);
    wait_kstep(0);
    WG_PH(2);
    read_kstep(st, 1);                              // in flight under the MFMAs of k-step 0
    mma_kstep(0);
    wait_kstep(1);
    WG_PH(3);
    mma_kstep(1);
    WG_PH(4);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef AST_STAMPS
  if (threadIdx.x == 0 && tix < 4096) { ast_wg_stamps[tix * 8 + 5] = (unsigned long long)((ph[0] << 32) | (ph[1] & 0xffffffffll)); ast_wg_phase[tix * 4 + 0] = ph[2]; ast_wg_phase[tix * 4 + 1] = ph[3]; ast_wg_phase[tix * 4 + 2] = ph[4]; ast_wg_phase[tix * 4 + 3] = nk; }
#endif
  WG_STAMP(2); WG_STAMP(3);

  // flush: column tile j of wave w = tap (kh, kw = j), source channels cs0 + 16 w + li; rows = output channels cd0 + 16 i + 4 gq + r
  dw += (size_t)(bz % (nrep & 0xffff)) * rep_stride;
  const int mode = (nrep >> 16) & 3;
  const int wts[3] = {wt0, wt1, wt2};
  if (mode == 0) {
#pragma unroll
    for (int j = 0; j < CTW; ++j)
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          unsafeAtomicAdd(dw + ((size_t)(cd0 + i * 16 + gq * 4 + r) * g.wtaps + wts[j]) * g.Cs + cs0 + wave * 16 + li, acc[i][j][r]);
#ifdef AST_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    WG_STAMP(4); WG_STAMP(6);
    return;
  }
  float* tile = reinterpret_cast<float*>(wq_all);   // [64 output channels][3 kw x 64 source channels]
  __syncthreads();                                  // every wave has finished reading the ring
#pragma unroll
  for (int j = 0; j < CTW; ++j)
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[(i * 16 + gq * 4 + r) * 192 + j * 64 + wave * 16 + li] = acc[i][j][r];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < CTW; ++j) {
    float* dcol = dw + (size_t)wts[j] * g.Cs + cs0 + lane;         // 64 lanes = 256 contiguous bytes of one output channel's tap row
    for (int row = wave; row < 64; row += 4) {
      float* d = dcol + (size_t)(cd0 + row) * g.wtaps * g.Cs;
      const float v = tile[row * 192 + j * 64 + lane];
      if (mode == 2) *d = v; else unsafeAtomicAdd(d, v);
    }
  }
#ifdef AST_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  WG_STAMP(4); WG_STAMP(6);
}

// 3x3, stride 1, "same" size, taps in (kh, kw) order at offsets kh - 1, kw - 1, 64-channel multiples on both sides
bool wgrad_rows_eligible(const ast_gather_t& g, int dtype) {
  if (dtype != AST_BF16 || g.ntaps != 9 || g.sh != 1 || g.sw != 1 || g.Hm != g.Hs || g.Wm != g.Ws || (g.Cs & 63) || (g.Cd & 63)) return false;
  for (int t = 0; t < 9; ++t) {
    const int dh = (g.tap[t] & 255) - 64 + g.oh, dw = ((g.tap[t] >> 8) & 255) - 64 + g.ow;
    if (dh != t / 3 - 1 || dw != t % 3 - 1) return false;
  }
  return true;
}

template <int NST>
int launch_wgrad_rows(const void* dy, const void* src, float* dw, const ast_gather_t& g, int P, hipStream_t s) {
  constexpr int STAGE = 64 * 128 + 72 * 128 + 384;
  constexpr int LDS = (NST * STAGE > 64 * 192 * 4 ? NST * STAGE : 64 * 192 * 4) + 64;
  static bool attr_set = false;
  if (!attr_set) {
    AST_HIP(hipFuncSetAttribute((const void*)wgrad_rows_kernel<NST>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  const int gx = g.Cd / 64, gy = 3 * (g.Cs / 64);
  const int tiles = gx * gy;
  const char* wte = getenv("AST_WGRAD_WG_TARGET");
  const int wg_target = wte ? atoi(wte) : 384;          // 1.5 workgroups per CU (71 KB of LDS each); 256 and 512 measured slower (profiles/r03/wg_rows_layers.txt)
  int nsplit = std::max(1, std::min((P + 4 * 64 - 1) / (4 * 64), (wg_target + tiles - 1) / tiles));
  if (g_wg_slab) nsplit = std::min(nsplit, g_wg_nrep);
  int pps = (P + nsplit - 1) / nsplit;
  pps = (pps + 63) / 64 * 64;
  nsplit = (P + pps - 1) / pps;
  const unsigned dy_bytes = (unsigned)((size_t)P * g.Cd * 2);
  const unsigned src_bytes = (unsigned)((size_t)P * g.Cs * 2);
  const int total = gx * gy * nsplit;
  g_wg_slices = nsplit;
  hipLaunchKernelGGL(wgrad_rows_kernel<NST>, dim3((total + 7) / 8 * 8), dim3(256), LDS, s, (const bf16_t*)dy, (const bf16_t*)src, dw, g, P, pps,
                     dy_bytes, src_bytes, 1.0f / (float)(g.Hm * g.Wm), 1.0f / (float)g.Wm, gx, gy, nsplit, wg_nrep_arg(), g_wg_rep_stride);
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
  constexpr int TILE = BMW * NCT * 16 * 4;                                      // the [row][column] f32 image of the flush
  int* taptab = reinterpret_cast<int*>(wl_all + max(max(PG * group_bytes, RED), TILE));   // behind every use of the staging area

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
  dw += (size_t)(blockIdx.z % (nrep & 0xffff)) * rep_stride;      // this slice's gradient replica / slab (see g_wg_nrep)
  flush_tile_rows<BMW, NCT, RT, CTW>(acc, reinterpret_cast<float*>(wl_all), dw, g, cd0, col0, ncols, wave, lane,
                                     [&](int t) { return taptab[16 + t]; }, (nrep >> 16) & 3);
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
  const int lds = std::max(std::max(PG * hp.lds, PG > 1 ? RED : 0), BMW * NCT * 16 * 4) + 160;
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
  int gz = std::max(1, std::min((hp.ntiles + PG - 1) / PG, wg_target / (gx * gy)));
  if (g_wg_slab) gz = std::min(gz, g_wg_nrep);
  g_wg_slices = gz;
  const unsigned dy_bytes = (unsigned)((size_t)g.N * g.Hm * g.Wm * g.Cd * sizeof(T));
  const unsigned src_bytes = (unsigned)((size_t)g.N * g.Hs * g.Ws * g.Cs * sizeof(T));
  hipLaunchKernelGGL((wgrad_halo_kernel<T, BMW, NCT, PG>), dim3(gx, gy, gz), dim3(256 * PG), lds, s, (const T*)dy, (const T*)src, dw, g, hp,
                     dy_bytes, src_bytes, wg_nrep_arg(), g_wg_rep_stride);
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
// Tap-tile weight gradient for >= 64 output channels ("wgrad_tap"): wave-autonomous, no barrier in the K loop.
//
// Why (round-2 stamps of wgrad_kernel on the 64-channel layer, profiles/r02/z_wgrad_stamps_kloop.txt): a 64-pixel trip of the
// cooperative kernel was 3 300 cycles for 384 cycles of MFMA -- LDS store 690, barrier 650, issuing the next loads 808, reads +
// MFMA 533, barrier 621 -- i.e. the phases of ONE workgroup per CU ran back to back, two barriers per trip between two pixel
// groups; and its 64 x 192 tiles left 85 pixel slices whose 12 MB of f32 atomics (4 x 64 B per wave-instruction) took a third
// of the launch.  Here:
//  * the output tile is 64 channels x 64 columns of the (tap, channel) space (one tap for >= 64 source channels), so a layer has
//    3x more tiles and 3x fewer pixel slices: a third of the flush bytes (flush bytes = slices x |dW|);
//  * every WAVE owns its K steps end to end: it fetches 32 pixels (bf16; 16 in f32) of its dy rows and of its tap's source rows
//    with eight 16-byte buffer loads per lane (whole 128-byte rows: 8 pixels = 1 KB contiguous per wave-instruction for dy),
//    writes them to its PRIVATE 8 KB of LDS, and reads both operands back transposed (ds_read_b64_tr_b16; f32: ds_read_b32)
//    into 4 + 4 fragments for 16 MFMAs on its own 64 x 64 accumulator.  LDS operations of one wave execute in order, so no
//    barrier and no second buffer are needed; eight waves per CU are in different phases and hide each other's latencies;
//  * the LDS image is conflict-free for the transposed reads without padding: the 32-byte segment s of pixel row r is stored
//    at s ^ (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) (a 32-lane group of ds_read_b64_tr_b16 touches rows {0-3, 8-11} + 4k of
//    one segment: 8 rows x 32 B must cover the 64 banks once); f32: 64-byte segment s at s ^ (r & 1);
//  * at the end the waves' partial tiles are summed through LDS and every wave flushes 64/NW whole rows: one atomic
//    wave-instruction = 64 consecutive floats of one dW row = 256 contiguous bytes (the shape the memory-side atomic units
//    take at full rate; MI355X_MICROARCH.md "Global float atomics"); a single-slice launch adds with plain read-modify-write.
// ---------------------------------------------------------------------------
template <typename T> struct WtCfg;
template <> struct WtCfg<bf16_t> { static constexpr int KPX = 32; };
template <> struct WtCfg<float> { static constexpr int KPX = 16; };

template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void wgrad_tap_kernel(const T* __restrict__ dy, const T* __restrict__ src, float* __restrict__ dw,
                                                            const ast_gather_t g, const int P, const int pps, const unsigned dy_bytes,
                                                            const unsigned src_bytes, const float rcp_hw, const float rcp_w, const int gx,
                                                            const int gy, const int gz, const int nrep, const long rep_stride, const int rmw) {
  constexpr int ES = sizeof(T), E = 16 / ES;
  constexpr int KPX = WtCfg<T>::KPX;               // pixels per K step of one wave
  constexpr int RB = 64 * ES;                      // bytes of a 64-channel row piece: 128 / 256
  constexpr int CPR = RB / 16;                     // 16-byte chunks per row piece: 8 / 16
  constexpr int RPP = 64 / CPR;                    // pixel rows per load pass of the wave: 8 / 4
  constexpr int NLD = KPX / RPP;                   // load passes per operand and K step: 4
  constexpr int WAVE_LDS = 2 * KPX * RB;           // one wave's staging: dy rows, then source rows (8 KB)
  constexpr int RED_LDS = 64 * 64 * 4;             // one wave's partial tile in the final reduction (16 KB)
  constexpr int RPW = 64 / NW;                     // dW rows flushed per wave
  constexpr unsigned OOB = 0x80000000u;
  static_assert(64 % NW == 0 && NLD == 4 && WAVE_LDS <= RED_LDS, "tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char wt_lds[];
  int* taptab = reinterpret_cast<int*>(wt_lds + NW * RED_LDS);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware order (as wgrad_kernel): XCD x takes the x-th contiguous eighth of the (slice, row tile, column tile) sequence, so an
  // L2 holds one band of pixels and serves it to all the tiles (taps) that read it
  const int chunk = gridDim.x >> 3;
  const int tix = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  const int tiles = gx * gy;
  if (tix >= tiles * gz) return;
  const int bz = tix / tiles, tile = tix - bz * tiles;
  const int bx = tile / gy, by = tile - bx * gy;
  const int cd0 = bx * 64, col0 = by * 64;
  const int ncols = g.ntaps * g.Cs;
  const int HWm = g.Hm * g.Wm;
  const int p_begin = bz * pps, p_end = min(P, p_begin + pps);
  if (p_begin >= p_end) return;                    // (uniform per workgroup)
  const __amdgpu_buffer_rsrc_t dyR = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srcR = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
#pragma unroll
  for (int t = 0; t < AST_MAX_TAPS; ++t)
    if ((int)threadIdx.x == t) taptab[t] = g.tap[t];      // static index: a dynamic one would spill the by-value struct to scratch
  __syncthreads();

  // ---- loader role of the lane: chunk c of pixel rows r0, r0 + RPP, ... of the K step
  const int c = lane % CPR, r0 = lane / CPR;
  const int cdc = cd0 + c * E;
  const bool yok = cdc < g.Cd;
  const int colc = col0 + c * E;
  const bool xok = colc < ncols;
  int xdh = 0, xdw = 0, xdelta = 0;
  {
    const int t = xok ? colc / g.Cs : 0, ch = xok ? colc - t * g.Cs : 0;
    int wt;
    decode_tap(taptab[t], xdh, xdw, wt);
    xdelta = ((xdh * g.Ws + xdw) * g.Cs + ch) * ES;
  }
  int lofs[NLD];                                   // LDS byte offset of (row r0 + i*RPP, chunk c) inside an operand image
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int r = r0 + i * RPP;
    if constexpr (ES == 2) lofs[i] = r * RB + ((((c >> 1) ^ (((r >> 1) & 1) | (((r >> 3) & 1) << 1))) << 5) | ((c & 1) << 4));
    else lofs[i] = r * RB + ((((c >> 2) ^ (r & 1)) << 6) | ((c & 3) << 4));
  }
  unsigned char* Yw = wt_lds + wave * WAVE_LDS;
  unsigned char* Xw = Yw + KPX * RB;

  u32x4 yreg[NLD], xreg[NLD];
  auto load_step = [&](int ks) __attribute__((always_inline)) {
    const int pb = p_begin + ks * KPX + r0;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int p = pb + i * RPP;
      const bool pv = p < p_end;
      const int pp = pv ? p : 0;
      yreg[i] = __builtin_amdgcn_raw_buffer_load_b128(dyR, (pv && yok) ? (unsigned)((pp * g.Cd + cdc) * ES) : OOB, 0, 0);
      const int n = fdiv(pp, HWm, rcp_hw), rem = pp - n * HWm;
      const int hm = fdiv(rem, g.Wm, rcp_w), wq = rem - hm * g.Wm;
      const int hs0 = hm * g.sh + g.oh, ws0 = wq * g.sw + g.ow;
      const bool ok = pv && xok && (unsigned)(hs0 + xdh) < (unsigned)g.Hs && (unsigned)(ws0 + xdw) < (unsigned)g.Ws;
      xreg[i] = __builtin_amdgcn_raw_buffer_load_b128(srcR, ok ? (unsigned)((((n * g.Hs + hs0) * g.Ws + ws0) * g.Cs) * ES + xdelta) : OOB, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, gq = lane >> 4;
  const int nks = (p_end - p_begin + KPX - 1) / KPX;
  int ks = wave;
  if (ks < nks) load_step(ks);
  for (; ks < nks; ks += NW) {                     // trip counts differ between the waves: nothing inside is workgroup-wide
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      *reinterpret_cast<u32x4*>(Yw + lofs[i]) = yreg[i];
      *reinterpret_cast<u32x4*>(Xw + lofs[i]) = xreg[i];
    }
    asm volatile("" ::: "memory");
    if (ks + NW < nks) load_step(ks + NW);         // in flight while this step is consumed
    if constexpr (ES == 2) {
      typedef __attribute__((address_space(3))) bf16x4 lds_b4;
      // lane 4q + p of a 16-lane group supplies pixel row q, channels 4p..4p+3 of the group's 4 x 16 block; after the
      // hardware transpose lane l holds 4 consecutive pixels of channel l
      const int r_lo = 8 * gq + (li >> 2), r_hi = r_lo + 4;
      const int g_lo = ((r_lo >> 1) & 1) | (((r_lo >> 3) & 1) << 1), g_hi = ((r_hi >> 1) & 1) | (((r_hi >> 3) & 1) << 1);
      const int b_lo = r_lo * RB + (li & 3) * 8, b_hi = r_hi * RB + (li & 3) * 8;
      bf16x8 af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Yw + b_lo + ((i ^ g_lo) << 5)));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Yw + b_hi + ((i ^ g_hi) << 5)));
        af[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Xw + b_lo + ((j ^ g_lo) << 5)));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(Xw + b_hi + ((j ^ g_hi) << 5)));
        const bf16x8 bf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s4 = 0; s4 < KPX / 4; ++s4) {
        const int r = 4 * s4 + gq;
        const int base = r * RB + li * 4;
        float af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const float*>(Yw + base + ((i ^ (r & 1)) << 6));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float bf = *reinterpret_cast<const float*>(Xw + base + ((j ^ (r & 1)) << 6));
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf, acc[i][j], 0, 0, 0);
        }
      }
    }
    asm volatile("" ::: "memory");
  }

  // ---- sum the NW partial tiles through LDS.  D[row = cd 16 i + 4 gq + r][col = 16 j + li]
  __syncthreads();                                 // every wave is done with its staging (the reduction area overlays it)
  float* red = reinterpret_cast<float*>(wt_lds) + wave * (64 * 64);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(16 * i + 4 * gq + r) * 64 + 16 * j + li] = acc[i][j][r];
  __syncthreads();
  // wave w owns rows RPW*w ..; lane = column: one flush instruction = 64 consecutive floats of one dW row (256 contiguous bytes)
  const int col = col0 + lane;
  const bool cok = col < ncols;
  const int t = cok ? col / g.Cs : 0, ch = cok ? col - t * g.Cs : 0;
  const int wtc = taptab[t] >> 16;
  float* dwr = dw + (size_t)(bz % nrep) * rep_stride;      // this pixel slice's gradient replica (see g_wg_nrep)
  const float* rbase = reinterpret_cast<const float*>(wt_lds);
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = RPW * wave + rr, cd = cd0 + row;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += rbase[w * (64 * 64) + row * 64 + lane];
    if (cok && cd < g.Cd) {
      float* dst = dwr + ((size_t)cd * g.wtaps + wtc) * g.Cs + ch;
      if (rmw) *dst += v;                          // single pixel slice: this workgroup is the tile's only adder in the launch
      else unsafeAtomicAdd(dst, v);
    }
  }
}

template <typename T>
int launch_wgrad_tap(const void* dy, const void* src, float* dw, const ast_gather_t& g, int P, hipStream_t s) {
  constexpr int NW = 8, KPX = WtCfg<T>::KPX;
  constexpr int LDS = NW * 64 * 64 * 4 + 64;
  static bool attr_set = false;
  if (!attr_set) {
    AST_HIP(hipFuncSetAttribute((const void*)wgrad_tap_kernel<T, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  const int gx = (g.Cd + 63) / 64, gy = (g.ntaps * g.Cs + 63) / 64, tiles = gx * gy;
  // pixel slices: about one workgroup (8 waves) per CU.  More slices = more flush bytes (slices x |dW| of f32 atomics at the chip's
  // ~1.3 TB/s), fewer = idle CUs; a slice keeps at least two K steps per wave.
  const char* we = getenv("AST_WGRAD_TAP_WGS");
  const int wg_target = we ? atoi(we) : 256;
  int gz = std::max(1, std::min((P + 2 * KPX * NW - 1) / (2 * KPX * NW), (wg_target + tiles / 2) / tiles));
  int pps = (P + gz - 1) / gz;
  pps = (pps + KPX - 1) / KPX * KPX;
  gz = (P + pps - 1) / pps;
  const unsigned dy_bytes = (unsigned)((size_t)P * g.Cd * sizeof(T));
  const unsigned src_bytes = (unsigned)((size_t)g.N * g.Hs * g.Ws * g.Cs * sizeof(T));
  const int total = tiles * gz;
  g_wg_slices = gz;
  hipLaunchKernelGGL((wgrad_tap_kernel<T, NW>), dim3((total + 7) / 8 * 8), dim3(64 * NW), LDS, s, (const T*)dy, (const T*)src, dw, g, P, pps,
                     dy_bytes, src_bytes, 1.0f / (float)(g.Hm * g.Wm), 1.0f / (float)g.Wm, gx, gy, gz, g_wg_nrep, g_wg_rep_stride, gz == 1 ? 1 : 0);
  AST_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int ast_wgrad(const void* dy, const void* src, float* dw, const ast_gather_t* gp, int dtype, void* stream);
extern "C" int ast_wgrad_rep(const void* dy, const void* src, float* dw, const ast_gather_t* gp, int dtype, int nrep, void* stream) {
  if (nrep < 1 || nrep > 64 || !gp) AST_FAIL("ast_wgrad_rep: 1..64 replicas");
  g_wg_nrep = nrep;
  g_wg_rep_stride = (long)gp->Cd * gp->wtaps * gp->Cs;
  const int rc = ast_wgrad(dy, src, dw, gp, dtype, stream);
  g_wg_nrep = 1; g_wg_rep_stride = 0;
  return rc;
}

// ---- slab mode: every pixel slice STORES its partial dW into its own copy; ast_slab_sum adds the copies up ----------------------
// The atomic flush of the pixel-rich layers is rate-bound, not contention-bound: slices x |dW| bytes of f32 atomics at the chip's
// ~1.3 TB/s (measured here: 0.19 us per slice of the 147 KB gradient = 0.77 TB/s; tools/halo_probe.sh: time = trips x 1.6 us +
// slices x 0.19 us), a third to a half of the launch.  Plain 256-byte-row stores move the same bytes at the HBM rate.
namespace {
struct SlabRec { float* base; unsigned n, slabs; };                    // n floats per copy, copies n floats apart
struct SlabArgs { SlabRec rec[AST_MAX_SLAB_RECS]; unsigned first_block[AST_MAX_SLAB_RECS + 1]; int nrec; };
// copy 0 <- sum of the `slabs` copies; a workgroup owns 1024 consecutive floats of one record, eight copies in flight per thread
__global__ __launch_bounds__(256) void slab_sum_kernel(const SlabArgs a) {
  int r = 0;
#pragma unroll 1
  while (r + 1 < a.nrec && blockIdx.x >= a.first_block[r + 1]) ++r;
  // (scalar selects, not a dynamic index into the by-value struct: that would put it in scratch memory)
  float* base = nullptr; unsigned n = 0, slabs = 0, fb = 0;
#pragma unroll
  for (int i = 0; i < AST_MAX_SLAB_RECS; ++i)
    if (i == r) { base = a.rec[i].base; n = a.rec[i].n; slabs = a.rec[i].slabs; fb = a.first_block[i]; }
  const unsigned i4 = ((blockIdx.x - fb) * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  f32x4 acc = *reinterpret_cast<const f32x4*>(base + i4);
  unsigned s = 1;
  for (; s + 8 <= slabs; s += 8) {
    f32x4 t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const f32x4*>(base + (size_t)(s + k) * n + i4);
    acc += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  }
  for (; s < slabs; ++s) acc += *reinterpret_cast<const f32x4*>(base + (size_t)s * n + i4);
  *reinterpret_cast<f32x4*>(base + i4) = acc;
}
}  // namespace

extern "C" int ast_wgrad(const void* dy, const void* src, float* dw, const ast_gather_t* gp, int dtype, void* stream);
extern "C" int ast_wgrad_slab(const void* dy, const void* src, float* slabs, const ast_gather_t* gp, int dtype, int nslabs, int* slices_out,
                              void* stream) {
  if (nslabs < 1 || nslabs > 4096 || !gp || !slices_out) AST_FAIL("ast_wgrad_slab: 1..4096 slabs and a slices output");
  g_wg_nrep = nslabs; g_wg_slab = 1; g_wg_slices = 0;
  g_wg_rep_stride = (long)gp->Cd * gp->wtaps * gp->Cs;
  const int rc = ast_wgrad(dy, src, slabs, gp, dtype, stream);
  *slices_out = g_wg_slices;
  g_wg_nrep = 1; g_wg_slab = 0; g_wg_rep_stride = 0;
  return rc;
}

extern "C" int ast_slab_sum(float* const* bases, const int64_t* floats_per_copy, const int* slabs, int nrec, void* stream) {
  if (!bases || !floats_per_copy || !slabs || nrec < 1 || nrec > AST_MAX_SLAB_RECS) AST_FAIL("ast_slab_sum: 1..%d records", AST_MAX_SLAB_RECS);
  SlabArgs a;
  unsigned blocks = 0;
  a.nrec = nrec;
  for (int i = 0; i < AST_MAX_SLAB_RECS; ++i) {
    a.rec[i] = SlabRec{nullptr, 0, 0};
    a.first_block[i] = blocks;
    if (i < nrec) {
      if (!bases[i] || floats_per_copy[i] < 4 || (floats_per_copy[i] & 3) || floats_per_copy[i] > (1L << 30) || slabs[i] < 1 || (((uintptr_t)bases[i]) & 15))
        AST_FAIL("ast_slab_sum: record %d: 16-byte aligned base, a multiple of 4 floats per copy, >= 1 copies", i);
      a.rec[i] = SlabRec{bases[i], (unsigned)floats_per_copy[i], (unsigned)slabs[i]};
      blocks += (unsigned)((floats_per_copy[i] + 1023) / 1024);
    }
  }
  a.first_block[AST_MAX_SLAB_RECS] = blocks;
  hipLaunchKernelGGL(slab_sum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  AST_CHECK_LAUNCH();
  return 0;
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
  // >= 64 output channels: the wave-autonomous tap-tile kernel (AST_WGRAD_TAP=0: the cooperative kernels below, for A/B)
  // Opt-in (AST_WGRAD_TAP=1): measured SLOWER than the cooperative kernels below -- isolated 44.4 vs 39.4 us on the 64-channel
  // layer, 71.8 vs 41.3 (256 channels), 67.2 vs 39.6 (512); only the stride-2 layers gain (24.9 vs 30.9) -- and the whole step
  // 7.29 vs 6.2 ms: its nine tap tiles re-read every dy / source row nine times through L1 (398 MB per launch at ~9 TB/s: the
  // fabric, not the MFMAs), and a workgroup takes a whole CU (128 KB of LDS) away from the other streams' kernels.  DESIGN 9.3.
  const char* te = getenv("AST_WGRAD_TAP");                  // read per call (host side only): tests toggle it at run time
  const bool tap_on = te && atoi(te) != 0;
  if (tap_on && !g_wg_slab && g.Cd >= 64) { AST_DISPATCH_T(dtype, { return launch_wgrad_tap<T>(dy, src, dw, g, P, s); }); }
  const char* re = getenv("AST_WGRAD_ROWS");                 // read per call (host side only): tests toggle it at run time
  if (!(re && atoi(re) == 0) && wgrad_rows_eligible(g, dtype)) {
    if ((long)P * g.Cs * 2 >= (1L << 31)) AST_FAIL("ast_wgrad: source exceeds the 2 GiB buffer-addressing range");
    const char* se = getenv("AST_WGRAD_ROWS_STAGES");       // ring depth: 4 stages = 71 KB of LDS (two workgroups per CU), 3 = 53 KB (three)
    const int nst = se ? atoi(se) : 4;
    if (nst == 2) return launch_wgrad_rows<2>(dy, src, dw, g, P, s);
    if (nst == 3) return launch_wgrad_rows<3>(dy, src, dw, g, P, s);
    return launch_wgrad_rows<4>(dy, src, dw, g, P, s);
  }
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

