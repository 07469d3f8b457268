// Constant-Q front end (utilityFunctions.py:39-60, `librosa.cqt(y, sr=22050, n_bins=84, hop_length=256)`) and the
// polyphase resampler of `load_audio` (utilityFunctions.py:105-122, `torchaudio.functional.resample`).
//
// librosa's recursive CQT is, per octave o (top octave first): a rectangular-window STFT of the o-times-halved signal
// at hop 256 >> o, multiplied by the sparsified FFT of 12 wavelets.  Both steps are linear in the frame, so the host
// folds them into 12 complex time-domain kernels W[k][n] (ast_amd/cqt.py) and one octave is a strided correlation:
//     C[k][t] = sum_n y_o[t * hop_o - nfft/2 + n] * W[k][n]           (zero outside the signal: pad_mode="constant")
// 345 frames x 12 filters x 256 taps per octave per clip: HBM/latency-bound byte work, no MFMA.  Between octaves the
// signal is halved by a linear-phase FIR (ast_resample_poly with orig=2, new=1), which is also the kernel behind
// load_audio's 44.1/48 kHz -> 22.05 kHz conversion.
#include "ast_common.h"

#define AST_CQT_MAX_OCTAVES 12
#define AST_CQT_FR 8                     // frames per workgroup of the octave kernel

namespace {

// one workgroup per (8 frames, clip, octave); their sample span is staged in LDS once and read by the 4 waves, 3 filters
// each.  NPL > 0: nfft == 64 * NPL and nf <= 12 -- every lane's kernel taps are fetched into registers once, before the
// span barrier, and reused for the 8 frames (one frame per workgroup spent ~4 us of latency per 256-tap dot product).
struct CqtOctaves {                     // one entry per octave of the launch (blockIdx.z)
  const float* y[AST_CQT_MAX_OCTAVES];  // the octave's (halved) signals, B rows of n[o] floats
  int n[AST_CQT_MAX_OCTAVES], hop[AST_CQT_MAX_OCTAVES], lo[AST_CQT_MAX_OCTAVES], nf[AST_CQT_MAX_OCTAVES], row0[AST_CQT_MAX_OCTAVES];
};

template <int NPL>
__global__ __launch_bounds__(256) void cqt_octave_kernel(const CqtOctaves oc, const float* __restrict__ w_re_all,
                                                         const float* __restrict__ w_im_all, const float* __restrict__ scale_all,
                                                         const int nfft, float* __restrict__ out, const int T, const int ld, const int bin_off) {
  const int o = blockIdx.z;
  const float* __restrict__ y = oc.y[o];
  const int n = oc.n[o], hop = oc.hop[o], nf = oc.nf[o], bin0 = bin_off + oc.lo[o];
  const long y_stride = n;
  const float* __restrict__ w_re = w_re_all + (size_t)oc.row0[o] * nfft;
  const float* __restrict__ w_im = w_im_all + (size_t)oc.row0[o] * nfft;
  const float* __restrict__ scale = scale_all + oc.lo[o];
  extern __shared__ float span[];                                // the samples of FR consecutive frames
  const int t0 = blockIdx.x * AST_CQT_FR, b = blockIdx.y;
  const float* yb = y + (size_t)b * y_stride;
  const int start = t0 * hop - nfft / 2;
  const int nfr = min(AST_CQT_FR, T - t0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NR = NPL > 0 ? NPL : 1;
  float wr[3][NR], wi[3][NR];
  if (NPL > 0) {
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      const int k = min(wave + 4 * f, nf - 1);
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        wr[f][i] = w_re[(size_t)k * nfft + lane + 64 * i];
        wi[f][i] = w_im[(size_t)k * nfft + lane + 64 * i];
      }
    }
  }
  const int nspan = (nfr - 1) * hop + nfft;
  for (int i = threadIdx.x; i < nspan; i += 256) {
    const int s = start + i;
    span[i] = (s >= 0 && s < n) ? yb[s] : 0.0f;
  }
  __syncthreads();
  if constexpr (NPL > 0) {
    // 8 frames x 3 filters x (re, im) = 48 per-lane partial sums.  Three transposed reductions of 16 values each
    // (row16_transpose_sum: lane l of a 16-lane row ends with the row's sum of value l), two cross-row exchanges, and
    // lanes 0..15 store one result each -- instead of 48 separate wave sums (which made the kernel VALU-issue bound).
    static_assert(AST_CQT_FR * 6 == 48, "three groups of sixteen");
    float part[AST_CQT_FR * 6];
#pragma unroll
    for (int fr = 0; fr < AST_CQT_FR; ++fr) {
      const float* frame = span + fr * hop;                            // frames past nfr read stale LDS: never stored
      float fv[NR];
#pragma unroll
      for (int i = 0; i < NR; ++i) fv[i] = frame[lane + 64 * i];
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        float re = 0.0f, im = 0.0f;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          re = __builtin_fmaf(fv[i], wr[f][i], re);
          im = __builtin_fmaf(fv[i], wi[f][i], im);
        }
        part[fr * 6 + f * 2] = re;
        part[fr * 6 + f * 2 + 1] = im;
      }
    }
    const int l = lane & 15;
#pragma unroll
    for (int grp = 0; grp < 3; ++grp) {
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = part[grp * 16 + q];
      float val = row16_transpose_sum(v, l);
      val += __shfl_xor(val, 16);
      val += __shfl_xor(val, 32);
      const int q = grp * 16 + l, fr = q / 6, f = (q - fr * 6) >> 1, c = q & 1;
      const int k = wave + 4 * f, t = t0 + fr;
      if (lane < 16 && fr < nfr && k < nf)
        out[((size_t)b * 2 * T + (size_t)c * T + t) * ld + bin0 + k] = val * scale[k];   // (B, 2, T, ld): real plane, then imaginary
    }
  } else {
    for (int fr = 0; fr < nfr; ++fr) {
      const float* frame = span + fr * hop;
      const int t = t0 + fr;
      for (int k = wave; k < nf; k += 4) {
        float re = 0.0f, im = 0.0f;
        const float* pr = w_re + (size_t)k * nfft;
        const float* pi = w_im + (size_t)k * nfft;
        for (int i = lane; i < nfft; i += 64) {
          const float v = frame[i];
          re = __builtin_fmaf(v, pr[i], re);
          im = __builtin_fmaf(v, pi[i], im);
        }
        re = wave_sum_dpp(re);
        im = wave_sum_dpp(im);
        if (lane == 0) {
          float* o = out + ((size_t)b * 2 * T + t) * ld + bin0 + k;
          o[0] = re * scale[k];
          o[(size_t)T * ld] = im * scale[k];
        }
      }
    }
  }
}

// y[b][i*nnew + p] = gain * sum_k kern[p][k] * x[b][i*orig + k - width]   (x = 0 outside [0, n)); any ratio.
// One thread per output; its taps are a serial chain of load pairs, so eight independent accumulators keep eight
// pairs in flight.  (Splitting one output's taps over 8 lanes was 3x slower: it turns the broadcast tap loads and the
// stride-`orig` signal loads into scattered ones and the kernel becomes L1-throughput bound.)
__global__ __launch_bounds__(256) void resample_poly_kernel(const float* __restrict__ x, const int n, const float* __restrict__ kern,
                                                            const int orig, const int nnew, const int klen, const int width,
                                                            float* __restrict__ y, const int m, const float gain) {
  const int b = blockIdx.y;
  const float* xb = x + (size_t)b * n;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < m; j += gridDim.x * 256) {
    const int i = j / nnew, p = j - i * nnew;
    const float* kp = kern + (size_t)p * klen;
    const int s0 = i * orig - width;
    const int k0 = s0 < 0 ? -s0 : 0, k1 = min(klen, n - s0);
    const float* xs = xb + s0;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
      float kv[8], xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { kv[u] = kp[k + u]; xv[u] = xs[k + u]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = __builtin_fmaf(kv[u], xv[u], a[u]);
    }
    for (; k < k1; ++k) a[0] = __builtin_fmaf(kp[k], xs[k], a[0]);
    y[(size_t)b * m + j] = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) * gain;
  }
}

// The 2:1 case (the six halvings of a CQT, 44.1 -> 22.05 kHz in load_audio): y[j] = gain * sum_k h[k] x[2j + k - width].
// A workgroup stages its input span in LDS split into even and odd samples, so that for a fixed tap the 64 lanes of
// a wave (64 consecutive outputs) read 64 consecutive words -- conflict-free -- and the tap itself is wave-uniform
// (an LDS broadcast).  R outputs per thread (j, j+256, ...) share each tap.  8 x 44100 outputs x 359 taps: 151 -> ~10 us.
template <int R>
__global__ __launch_bounds__(256) void decimate2_kernel(const float* __restrict__ x, const int n, const float* __restrict__ h, const int klen,
                                                        const int width, float* __restrict__ y, const int m, const float gain) {
  extern __shared__ float sm[];
  const int b = blockIdx.y, t = threadIdx.x;
  const float* xb = x + (size_t)b * n;
  const int j0 = blockIdx.x * (256 * R);
  const int ne = 256 * R + (klen + 1) / 2;                    // even (and odd) samples the tile needs
  float* xe = sm;
  float* xo = sm + ne;
  float* hs = sm + 2 * ne;                                     // taps: LDS broadcast reads, one latency class with the samples
  for (int i = t; i < klen; i += 256) hs[i] = h[i];
  const int base = 2 * j0 - width;
  for (int i = t; i < 2 * ne; i += 256) {
    const int s = base + i;
    const float v = (s >= 0 && s < n) ? xb[s] : 0.0f;
    ((i & 1) ? xo : xe)[i >> 1] = v;
  }
  __syncthreads();
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.0f;
  const int pairs = klen >> 1;
#pragma unroll 8
  for (int k2 = 0; k2 < pairs; ++k2) {
    const float h0 = hs[2 * k2], h1 = hs[2 * k2 + 1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      acc[r] = __builtin_fmaf(h0, xe[t + 256 * r + k2], acc[r]);
      acc[r] = __builtin_fmaf(h1, xo[t + 256 * r + k2], acc[r]);
    }
  }
  if (klen & 1) {
    const float h0 = hs[klen - 1];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = __builtin_fmaf(h0, xe[t + 256 * r + pairs], acc[r]);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int j = j0 + t + 256 * r;
    if (j < m) y[(size_t)b * m + j] = acc[r] * gain;
  }
}

// z-score + overlap windows of the CQT planes into the bins behind the STFT's (dataloader.py:9-18,
// utilityFunctions.py:240-263): x[b][s][c][w][bin0 + k] = (cqt[b][c][s*step + w][k] - mean[c][k]) / (std[c][k] + 1e-8),
// 0 past the last frame (the zero-padded tail window)
__global__ __launch_bounds__(256) void cqt_sections_kernel(const float* __restrict__ cqt, const int T, const int nb,
                                                           const float* __restrict__ mean, const float* __restrict__ std_,
                                                           float* __restrict__ x, const int S, const int win, const int step,
                                                           const int F_total, const int bin0, const size_t total) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % nb);
    size_t r = i / nb;
    const int w = (int)(r % win); r /= win;
    const int c = (int)(r & 1); r >>= 1;
    const int s = (int)(r % S);
    const size_t b = r / S;
    const int t = s * step + w;
    float v = 0.0f;
    if (t < T) v = (cqt[((b * 2 + c) * T + t) * nb + k] - mean[c * nb + k]) / (std_[c * nb + k] + 1e-8f);
    x[((((b * S + s) * 2 + c) * win + w) * (size_t)F_total) + bin0 + k] = v;
  }
}

}  // namespace

extern "C" int ast_cqt_sections(const float* cqt, int Bc, int T, int nb, const float* mean, const float* std_, float* x, int S, int win,
                                int step, int F_total, int bin0, void* stream) {
  if (!cqt || !mean || !std_ || !x) AST_FAIL("ast_cqt_sections: null pointer");
  if (Bc < 1 || T < 1 || nb < 1 || S < 1 || win < 1 || step < 1 || bin0 < 0 || bin0 + nb > F_total)
    AST_FAIL("ast_cqt_sections: bad shape (Bc=%d T=%d nb=%d S=%d win=%d step=%d F=%d bin0=%d)", Bc, T, nb, S, win, step, F_total, bin0);
  const size_t total = (size_t)Bc * S * 2 * win * nb;
  hipLaunchKernelGGL(cqt_sections_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, cqt, T,
                     nb, mean, std_, x, S, win, step, F_total, bin0, total);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_cqt_octaves(const float* const* ys, const int* ns, const int* hops, const int* los, const int* nfs, const int* row0s,
                               int n_oct, int B, const float* w_re, const float* w_im, const float* scale, int nfft, float* out, int T,
                               int ld, int bin_off, void* stream) {
  if (!ys || !ns || !hops || !los || !nfs || !row0s || !w_re || !w_im || !scale || !out) AST_FAIL("ast_cqt_octaves: null pointer");
  if (n_oct < 1 || n_oct > AST_CQT_MAX_OCTAVES || B < 1 || B > 65535 || nfft < 2 || nfft > 8192 || (nfft & 1) || T < 1 || bin_off < 0)
    AST_FAIL("ast_cqt_octaves: bad shape (n_oct=%d B=%d nfft=%d T=%d)", n_oct, B, nfft, T);
  CqtOctaves oc;
  bool fast = nfft == 256;
  for (int o = 0; o < n_oct; ++o) {
    if (!ys[o] || ns[o] < 1 || hops[o] < 1 || los[o] < 0 || nfs[o] < 1 || row0s[o] < 0 || bin_off + los[o] + nfs[o] > ld)
      AST_FAIL("ast_cqt_octaves: bad octave %d (n=%d hop=%d lo=%d nf=%d row0=%d ld=%d)", o, ns[o], hops[o], los[o], nfs[o], row0s[o], ld);
    oc.y[o] = ys[o]; oc.n[o] = ns[o]; oc.hop[o] = hops[o]; oc.lo[o] = los[o]; oc.nf[o] = nfs[o]; oc.row0[o] = row0s[o];
    fast = fast && nfs[o] <= 12;
  }
  int hop_max = 1;
  for (int o = 0; o < n_oct; ++o) hop_max = std::max(hop_max, hops[o]);
  const size_t lds = ((size_t)(AST_CQT_FR - 1) * hop_max + nfft) * sizeof(float);
  if (lds > 64 * 1024) AST_FAIL("ast_cqt_octaves: hop %d x %d frames + nfft %d exceeds the LDS span", hop_max, AST_CQT_FR, nfft);
  const dim3 grid((T + AST_CQT_FR - 1) / AST_CQT_FR, B, n_oct);
  if (fast)
    hipLaunchKernelGGL(cqt_octave_kernel<4>, grid, dim3(256), lds, (hipStream_t)stream, oc, w_re, w_im, scale, nfft, out, T, ld, bin_off);
  else
    hipLaunchKernelGGL(cqt_octave_kernel<0>, grid, dim3(256), lds, (hipStream_t)stream, oc, w_re, w_im, scale, nfft, out, T, ld, bin_off);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_resample_poly(const float* x, int B, int n, const float* kern, int orig, int nnew, int klen, int width, float* y, int m,
                                 float gain, void* stream) {
  if (!x || !kern || !y) AST_FAIL("ast_resample_poly: null pointer");
  if (B < 1 || B > 65535 || n < 1 || orig < 1 || nnew < 1 || klen < 1 || width < 0 || m < 1)
    AST_FAIL("ast_resample_poly: bad shape (B=%d n=%d orig=%d new=%d klen=%d width=%d m=%d)", B, n, orig, nnew, klen, width, m);
  // every output must come from a polyphase row that exists: j/nnew*orig stays an int
  if ((long)((m - 1) / nnew) * orig + klen >= (1L << 31)) AST_FAIL("ast_resample_poly: signal too long");
  if (orig == 2 && nnew == 1 && klen <= 4096) {
    if ((long)m * B >= 256 * 1024) {
      const size_t lds = (2 * (size_t)(256 * 4 + (klen + 1) / 2) + klen) * sizeof(float);
      hipLaunchKernelGGL(decimate2_kernel<4>, dim3((m + 1023) / 1024, B), dim3(256), lds, (hipStream_t)stream, x, n, kern, klen, width, y, m, gain);
    } else {
      const size_t lds = (2 * (size_t)(256 + (klen + 1) / 2) + klen) * sizeof(float);
      hipLaunchKernelGGL(decimate2_kernel<1>, dim3((m + 255) / 256, B), dim3(256), lds, (hipStream_t)stream, x, n, kern, klen, width, y, m, gain);
    }
    AST_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(resample_poly_kernel, dim3(std::min((m + 255) / 256, 4096), B), dim3(256), 0, (hipStream_t)stream, x, n, kern, orig, nnew,
                     klen, width, y, m, gain);
  AST_CHECK_LAUNCH();
  return 0;
}
