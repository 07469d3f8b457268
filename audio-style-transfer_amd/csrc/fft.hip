// STFT front-end (utilityFunctions.py:12-37): 1024-point Hann-windowed frames,
// hop 256, reflect padding, as an LDS-staged radix-4 Stockham FFT -- one
// 256-thread workgroup per frame, one radix-4 butterfly per thread per stage
// (5 stages).  The epilogue z-scores each bin (dataloader.py:9-13) and writes the
// frame straight into its row of the (B,S,2,287,F) section tensor
// (utilityFunctions.py:240-263), so the spectrogram never exists in HBM in
// (freq,time) order and no separate normalise / window / collate pass runs.
#include "ast_common.h"
#include "../../include/ast_hip.h"

namespace {
constexpr int NFFT = 1024, HOP = 256, NBIN = 513;

__global__ __launch_bounds__(256) void stft_sections_kernel(const float* __restrict__ wave, int nsamp, int T, const float* __restrict__ mean,
                                                             const float* __restrict__ stdv, float* __restrict__ x, int S, int win, int step,
                                                             int Ftot) {
  __shared__ float2 buf[2][NFFT];
  const int row = blockIdx.x, s = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int t = s * step + row;
  float* xr = x + ((((size_t)b * S + s) * 2 + 0) * win + row) * Ftot;
  float* xi = x + ((((size_t)b * S + s) * 2 + 1) * win + row) * Ftot;
  if (t >= T) {            // zero padded tail rows of the last section
    for (int f = tid; f < NBIN; f += 256) { xr[f] = 0.f; xi[f] = 0.f; }
    return;
  }
  const float* w = wave + (size_t)b * nsamp;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = tid + 256 * k;
    int idx = t * HOP + j - NFFT / 2;
    if (idx < 0) idx = -idx;
    if (idx >= nsamp) idx = 2 * (nsamp - 1) - idx;
    const float hann = 0.5f - 0.5f * cospif(2.f * j / NFFT);
    buf[0][j] = make_float2(w[idx] * hann, 0.f);
  }
  __syncthreads();
  int cur = 0;
#pragma unroll
  for (int p = 1; p < NFFT; p *= 4) {
    const int k = tid & (p - 1);
    const int j = ((tid - k) << 2) + k;
    const float alpha = -(float)k / (2.f * p);          // in units of pi
    float s1, c1, s2, c2, s3, c3;
    sincospif(alpha, &s1, &c1); sincospif(2.f * alpha, &s2, &c2); sincospif(3.f * alpha, &s3, &c3);
    const float2 a0 = buf[cur][tid], a1 = buf[cur][tid + 256], a2 = buf[cur][tid + 512], a3 = buf[cur][tid + 768];
    const float2 u0 = a0;
    const float2 u1 = make_float2(a1.x * c1 - a1.y * s1, a1.x * s1 + a1.y * c1);
    const float2 u2 = make_float2(a2.x * c2 - a2.y * s2, a2.x * s2 + a2.y * c2);
    const float2 u3 = make_float2(a3.x * c3 - a3.y * s3, a3.x * s3 + a3.y * c3);
    const float2 v0 = make_float2(u0.x + u2.x, u0.y + u2.y), v1 = make_float2(u0.x - u2.x, u0.y - u2.y);
    const float2 v2 = make_float2(u1.x + u3.x, u1.y + u3.y);
    const float2 d = make_float2(u1.x - u3.x, u1.y - u3.y);
    const float2 v3 = make_float2(d.y, -d.x);            // -i * (u1 - u3)
    buf[cur ^ 1][j] = make_float2(v0.x + v2.x, v0.y + v2.y);
    buf[cur ^ 1][j + p] = make_float2(v1.x + v3.x, v1.y + v3.y);
    buf[cur ^ 1][j + 2 * p] = make_float2(v0.x - v2.x, v0.y - v2.y);
    buf[cur ^ 1][j + 3 * p] = make_float2(v1.x - v3.x, v1.y - v3.y);
    cur ^= 1;
    __syncthreads();
  }
  for (int f = tid; f < NBIN; f += 256) {
    const float2 z = buf[cur][f];
    xr[f] = (z.x - mean[f]) / (stdv[f] + 1e-8f);
    xi[f] = (z.y - mean[NBIN + f]) / (stdv[NBIN + f] + 1e-8f);
  }
}
// ---- inverse STFT (utilityFunctions.py:62-82, torch.istft defaults) -------------------
// frame kernel: Hermitian extension of the one-sided bins, inverse 1024-point FFT (same Stockham
// network on the conjugated spectrum), periodic Hann window -> frames[t][1024].
__global__ __launch_bounds__(256) void istft_frames_kernel(const float* __restrict__ spec, int T, float* __restrict__ frames) {
  __shared__ float2 buf[2][NFFT];
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* re = spec + ((size_t)b * 2 + 0) * T * NBIN + (size_t)t * NBIN;
  const float* im = spec + ((size_t)b * 2 + 1) * T * NBIN + (size_t)t * NBIN;
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) {
    const int k = tid + 256 * k4;
    float xr, xi;
    if (k <= NFFT / 2) { xr = re[k]; xi = (k == 0 || k == NFFT / 2) ? 0.f : im[k]; }     // c2r ignores imag of DC / Nyquist
    else { xr = re[NFFT - k]; xi = -im[NFFT - k]; }
    buf[0][k] = make_float2(xr, -xi);                  // conj(X): x = conj(FFT(conj(X))) / N
  }
  __syncthreads();
  int cur = 0;
#pragma unroll
  for (int p = 1; p < NFFT; p *= 4) {
    const int k = tid & (p - 1);
    const int j = ((tid - k) << 2) + k;
    const float alpha = -(float)k / (2.f * p);
    float s1, c1, s2, c2, s3, c3;
    sincospif(alpha, &s1, &c1); sincospif(2.f * alpha, &s2, &c2); sincospif(3.f * alpha, &s3, &c3);
    const float2 a0 = buf[cur][tid], a1 = buf[cur][tid + 256], a2 = buf[cur][tid + 512], a3 = buf[cur][tid + 768];
    const float2 u0 = a0;
    const float2 u1 = make_float2(a1.x * c1 - a1.y * s1, a1.x * s1 + a1.y * c1);
    const float2 u2 = make_float2(a2.x * c2 - a2.y * s2, a2.x * s2 + a2.y * c2);
    const float2 u3 = make_float2(a3.x * c3 - a3.y * s3, a3.x * s3 + a3.y * c3);
    const float2 v0 = make_float2(u0.x + u2.x, u0.y + u2.y), v1 = make_float2(u0.x - u2.x, u0.y - u2.y);
    const float2 v2 = make_float2(u1.x + u3.x, u1.y + u3.y);
    const float2 d = make_float2(u1.x - u3.x, u1.y - u3.y);
    const float2 v3 = make_float2(d.y, -d.x);
    buf[cur ^ 1][j] = make_float2(v0.x + v2.x, v0.y + v2.y);
    buf[cur ^ 1][j + p] = make_float2(v1.x + v3.x, v1.y + v3.y);
    buf[cur ^ 1][j + 2 * p] = make_float2(v0.x - v2.x, v0.y - v2.y);
    buf[cur ^ 1][j + 3 * p] = make_float2(v1.x - v3.x, v1.y - v3.y);
    cur ^= 1;
    __syncthreads();
  }
  float* fr = frames + ((size_t)b * T + t) * NFFT;
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) {
    const int j = tid + 256 * k4;
    const float hann = 0.5f - 0.5f * cospif(2.f * j / NFFT);
    fr[j] = buf[cur][j].x * (1.f / NFFT) * hann;       // real part of conj(...) = real part
  }
}
// overlap-add of the 4 frames that cover each sample, divided by the window^2 envelope, centre trimmed
__global__ void istft_ola_kernel(const float* __restrict__ frames, int T, int nout, float* __restrict__ wave) {
  const int b = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nout; i += gridDim.x * blockDim.x) {
    const int j = i + NFFT / 2;                        // position in the padded signal
    float acc = 0.f, env = 0.f;
    const int t_hi = min(T - 1, j / HOP), t_lo = max(0, (j - NFFT + HOP) / HOP);
    for (int t = t_lo; t <= t_hi; ++t) {
      const int o = j - t * HOP;
      if (o < 0 || o >= NFFT) continue;
      const float hann = 0.5f - 0.5f * cospif(2.f * o / NFFT);
      acc += frames[((size_t)b * T + t) * NFFT + o];
      env += hann * hann;
    }
    wave[(size_t)b * nout + i] = env > 1e-11f ? acc / env : acc;
  }
}
}  // namespace

namespace {
// sections2spectrogram (utilityFunctions.py:265-283): count-normalised overlap-average of S windows of `wind` frames at
// step `hop` back to one (2, n_time, F) spectrogram per clip, truncated to `out_T` frames.  One thread per 4 bins.
__global__ __launch_bounds__(256) void overlap_avg_kernel(const float* __restrict__ sec, float* __restrict__ out, int S, int wind,
                                                          int hop, int F_in, int F_out, int out_T, size_t total4) {
  const int f4n = (F_out + 3) >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
    const int f = (int)(i % f4n) * 4;
    size_t r = i / f4n;
    const int t = (int)(r % out_T); r /= out_T;
    const int c = (int)(r % 2);
    const size_t b = r / 2;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int cnt = 0;
    const int i_hi = min(S - 1, t / hop), i_lo = max(0, (t - wind + hop) / hop);
    for (int k = i_lo; k <= i_hi; ++k) {
      const int tt = t - k * hop;
      if (tt < 0 || tt >= wind) continue;
      const float* p = sec + ((((b * S + k) * 2 + c) * wind + tt) * (size_t)F_in) + f;
#pragma unroll
      for (int q = 0; q < 4; ++q) if (f + q < F_out) acc[q] += p[q];
      ++cnt;
    }
    const float inv = 1.f / (float)max(cnt, 1);
    float* o = out + ((b * 2 + c) * (size_t)out_T + t) * F_out + f;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (f + q < F_out) o[q] = acc[q] * inv;
  }
}
}  // namespace

namespace {
// dataloader.py:9-13 normalize: (x - mean[c][f]) / (std[c][f] + eps) over a (C, T, F) spectrogram
__global__ __launch_bounds__(256) void zscore_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                     const float* __restrict__ std_, float* __restrict__ out, int T, int F, float eps,
                                                     size_t total) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int f = (int)(i % F);
    const int c = (int)(i / ((size_t)T * F));
    out[i] = (x[i] - mean[(size_t)c * F + f]) / (std_[(size_t)c * F + f] + eps);
  }
}
}  // namespace

namespace {
// Preprocessing_Dataset/compute_unified_stats.py:34-44 for one clip: per (channel, bin) mean over the T frames and the
// unbiased variance (torch.std(dim=1)**2), ADDED into running sums (the caller divides by the clip count at the end).
// One thread per (channel, bin); consecutive threads read consecutive bins of a frame (coalesced).
__global__ __launch_bounds__(256) void bin_stats_kernel(const float* __restrict__ x, float* __restrict__ mean_acc,
                                                        float* __restrict__ var_acc, int T, int F) {
  const int f = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  if (f >= F) return;
  const float* p = x + (size_t)c * T * F + f;
  double s = 0.0, q = 0.0;
  for (int t = 0; t < T; ++t) { const double v = p[(size_t)t * F]; s += v; q += v * v; }
  const double m = s / T;
  const double var = T > 1 ? fmax((q - s * m) / (T - 1), 0.0) : 0.0;
  mean_acc[(size_t)c * F + f] += (float)m;
  var_acc[(size_t)c * F + f] += (float)var;
}
}  // namespace

extern "C" int ast_bin_stats_acc(const float* x, float* mean_acc, float* var_acc, int C, int T, int F, void* stream) {
  if (!x || !mean_acc || !var_acc || C < 1 || T < 1 || F < 1) AST_FAIL("ast_bin_stats_acc: bad args");
  hipLaunchKernelGGL(bin_stats_kernel, dim3((F + 255) / 256, C), dim3(256), 0, (hipStream_t)stream, x, mean_acc, var_acc, T, F);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_zscore(const float* x, const float* mean, const float* std_, float* out, int C, int T, int F, float eps, void* stream) {
  if (!x || !mean || !std_ || !out || C < 1 || T < 1 || F < 1) AST_FAIL("ast_zscore: bad args");
  const size_t total = (size_t)C * T * F;
  hipLaunchKernelGGL(zscore_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, x, mean,
                     std_, out, T, F, eps, total);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_sections_overlap_avg(const float* sections, float* out, int Bc, int S, int wind, int hop, int F_in, int F_out,
                                        int out_T, void* stream) {
  if (!sections || !out || Bc < 1 || S < 1 || wind < 1 || hop < 1 || hop > wind || F_out < 1 || F_out > F_in || out_T < 1 ||
      out_T > hop * (S - 1) + wind)
    AST_FAIL("ast_sections_overlap_avg: bad args");
  const size_t total4 = (size_t)Bc * 2 * out_T * ((F_out + 3) / 4);
  const unsigned grid = (unsigned)std::min<size_t>((total4 + 255) / 256, 4096);
  hipLaunchKernelGGL(overlap_avg_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, sections, out, S, wind, hop, F_in, F_out, out_T,
                     total4);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_istft(const float* spec, int Bc, int T, float* frames_ws, float* wave, void* stream) {
  if (!spec || !frames_ws || !wave || Bc <= 0 || T <= 1) AST_FAIL("ast_istft: bad args");
  hipStream_t s = (hipStream_t)stream;
  const int nout = HOP * (T - 1);
  hipLaunchKernelGGL(istft_frames_kernel, dim3(T, Bc), dim3(256), 0, s, spec, T, frames_ws);
  hipLaunchKernelGGL(istft_ola_kernel, dim3((nout + 255) / 256, Bc), dim3(256), 0, s, frames_ws, T, nout, wave);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_stft_sections(const float* wave, int Bc, int nsamp, const float* mean, const float* std_, float* x, int S, int win,
                                 int step, int F_total, void* stream) {
  if (!wave || !mean || !std_ || !x || Bc <= 0 || nsamp <= NFFT / 2 || S <= 0 || win <= 0 || step <= 0 || F_total < NBIN)
    AST_FAIL("ast_stft_sections: bad args");
  const int T = 1 + nsamp / HOP;
  hipLaunchKernelGGL(stft_sections_kernel, dim3(win, S, Bc), dim3(256), 0, (hipStream_t)stream, wave, nsamp, T, mean, std_, x, S, win, step,
                     F_total);
  AST_CHECK_LAUNCH();
  return 0;
}
