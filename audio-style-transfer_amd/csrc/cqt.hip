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

namespace {

// one workgroup per (frame, clip); the frame is staged in LDS once and read by the 4 waves, 3 filters each
__global__ __launch_bounds__(256) void cqt_octave_kernel(const float* __restrict__ y, const int n, const long y_stride,
                                                         const float* __restrict__ w_re, const float* __restrict__ w_im,
                                                         const float* __restrict__ scale, const int nf, const int nfft, const int hop,
                                                         float* __restrict__ out, const int T, const int ld, const int bin0) {
  extern __shared__ float frame[];
  const int t = blockIdx.x, b = blockIdx.y;
  const float* yb = y + (size_t)b * y_stride;
  const int start = t * hop - nfft / 2;
  for (int i = threadIdx.x; i < nfft; i += 256) {
    const int s = start + i;
    frame[i] = (s >= 0 && s < n) ? yb[s] : 0.0f;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = wave; k < nf; k += 4) {
    float re = 0.0f, im = 0.0f;
    const float* wr = w_re + (size_t)k * nfft;
    const float* wi = w_im + (size_t)k * nfft;
    for (int i = lane; i < nfft; i += 64) {
      const float v = frame[i];
      re = __builtin_fmaf(v, wr[i], re);
      im = __builtin_fmaf(v, wi[i], im);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      re += __shfl_xor(re, off);
      im += __shfl_xor(im, off);
    }
    if (lane == 0) {
      float* o = out + ((size_t)b * 2 * T + t) * ld + bin0 + k;      // (B, 2, T, ld): real plane, then imaginary plane
      o[0] = re * scale[k];
      o[(size_t)T * ld] = im * scale[k];
    }
  }
}

// y[b][i*nnew + p] = gain * sum_k kern[p][k] * x[b][i*orig + k - width]   (x = 0 outside [0, n))
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
    float acc = 0.0f;
    for (int k = k0; k < k1; ++k) acc = __builtin_fmaf(kp[k], xb[s0 + k], acc);
    y[(size_t)b * m + j] = acc * gain;
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

extern "C" int ast_cqt_octave(const float* y, int B, int n, long y_stride, const float* w_re, const float* w_im, const float* scale,
                              int nf, int nfft, int hop, float* out, int T, int ld, int bin0, void* stream) {
  if (!y || !w_re || !w_im || !scale || !out) AST_FAIL("ast_cqt_octave: null pointer");
  if (B < 1 || n < 1 || y_stride < n || nf < 1 || nfft < 2 || nfft > 8192 || (nfft & 1) || hop < 1 || T < 1 || bin0 < 0 || bin0 + nf > ld)
    AST_FAIL("ast_cqt_octave: bad shape (B=%d n=%d nf=%d nfft=%d hop=%d T=%d ld=%d bin0=%d)", B, n, nf, nfft, hop, T, ld, bin0);
  if (B > 65535) AST_FAIL("ast_cqt_octave: at most 65535 clips per launch");
  hipLaunchKernelGGL(cqt_octave_kernel, dim3(T, B), dim3(256), nfft * sizeof(float), (hipStream_t)stream, y, n, y_stride, w_re, w_im, scale, nf,
                     nfft, hop, out, T, ld, bin0);
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
  hipLaunchKernelGGL(resample_poly_kernel, dim3(std::min((m + 255) / 256, 4096), B), dim3(256), 0, (hipStream_t)stream, x, n, kern, orig, nnew,
                     klen, width, y, m, gain);
  AST_CHECK_LAUNCH();
  return 0;
}
