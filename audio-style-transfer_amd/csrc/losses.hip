// Loss kernels: the fused complex-spectrogram reconstruction loss (one streaming
// pass, value + gradient) and the batch-coupled embedding losses of losses.py as
// single-workgroup wavefront-reduction kernels (B <= 64 rows of 256 floats).
#include "ast_common.h"
#include "../../include/ast_hip.h"

namespace {

constexpr float PI_F = 3.14159265358979323846f;

// ---- compute_comprehensive_loss (new_decoder.py:348-420) ---------------------------
// thread = one (b, t, f) complex bin, loops over the S sections so the section
// differences stay in registers; t+-1 neighbours come from L1/L2.
__global__ __launch_bounds__(1024) void recon_loss_kernel(const float* __restrict__ out, const float* __restrict__ tgt, int64_t tld,
                                                          int B, int S, int T, int Fq, float c_mse, float c_mag, float c_ph,
                                                          float c_tmp, float c_spc, float* __restrict__ sums, float* __restrict__ grad, int nslots) {
  __shared__ float red[17];
  const unsigned total = (unsigned)B * T * Fq;                  // < 2^31 (host check): 32-bit index math, no 64-bit divisions
  const size_t plane_o = (size_t)T * Fq, plane_t = (size_t)T * tld;
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int f = (int)(i % (unsigned)Fq);
    const unsigned r = i / (unsigned)Fq;
    const int t = (int)(r % (unsigned)T);
    const int b = (int)(r / (unsigned)T);
    float eprev[2] = {0.f, 0.f};     // e[s-1] at (t,f)
    for (int s = 0; s < S; ++s) {
      const size_t sec = (size_t)b * S + s;
      float o[2], g[2], e[2], eup[2], edn[2], enext[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float* po = out + (sec * 2 + c) * plane_o + (size_t)t * Fq + f;
        const float* pt = tgt + (sec * 2 + c) * plane_t + (size_t)t * tld + f;
        o[c] = po[0]; g[c] = pt[0]; e[c] = o[c] - g[c];
        eup[c] = t > 0 ? po[-(long)Fq] - pt[-tld] : 0.f;
        edn[c] = t + 1 < T ? po[Fq] - pt[tld] : 0.f;
        enext[c] = s + 1 < S ? po[2 * plane_o] - pt[2 * plane_t] : 0.f;
      }
      const float mo2 = o[0] * o[0] + o[1] * o[1], mt2 = g[0] * g[0] + g[1] * g[1];
      const float mo = sqrtf(mo2 + 1e-8f), mt = sqrtf(mt2 + 1e-8f);
      float dphi = atan2f(o[1], o[0]) - atan2f(g[1], g[0]);
      dphi = dphi + PI_F;
      dphi = dphi - 2.f * PI_F * floorf(dphi / (2.f * PI_F)) - PI_F;      // torch.remainder(., 2pi) - pi
      acc[0] += e[0] * e[0] + e[1] * e[1];
      acc[1] += (mo - mt) * (mo - mt);
      acc[2] += dphi * dphi;
      float gt[2] = {0.f, 0.f}, gs[2] = {0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (s + 1 < S) { const float d = enext[c] - e[c]; acc[3] += d * d; gt[c] -= d; }
        if (s > 0) gt[c] += e[c] - eprev[c];
        if (t + 1 < T) { const float d = edn[c] - e[c]; acc[4] += d * d; gs[c] -= d; }
        if (t > 0) gs[c] += e[c] - eup[c];
      }
      if (grad) {
        const float inv2 = 1.f / mo2;   // atan2 backward: (-im, re) / (re^2 + im^2)
        const float gph[2] = {-o[1] * inv2, o[0] * inv2};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          grad[(sec * 2 + c) * plane_o + (size_t)t * Fq + f] =
              2.f * (c_mse * e[c] + c_mag * (mo - mt) * o[c] / mo + c_ph * dphi * gph[c] + c_tmp * gt[c] + c_spc * gs[c]);
        }
      }
      eprev[0] = e[0]; eprev[1] = e[1];
    }
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const float v = block_sum(acc[k], red);
    if (threadIdx.x == 0) unsafeAtomicAdd(sums + (blockIdx.x % nslots) * 5 + k, v);      // nslots rows of 5: few adders per address
  }
}

// sums[k] = sum over the slots (fixed order), total = sum_k c[k] * sums[k], parts[k] = inv[k] * sums[k]: out = [5 sums][total][5 parts]
struct ReconFin { float c[5], inv[5]; };
__global__ __launch_bounds__(64) void recon_finish_kernel(const float* __restrict__ ws, int nslots, ReconFin a, float* __restrict__ out) {
  __shared__ float sk[5];
  const int k = threadIdx.x;
  if (k < 5) {
    float v = 0.f;
    for (int i = 0; i < nslots; ++i) v += ws[i * 5 + k];
    sk[k] = v;
    out[k] = v;
    out[6 + k] = v * a.inv[k];
  }
  __syncthreads();
  if (k == 0) out[5] = (((a.c[0] * sk[0] + a.c[1] * sk[1]) + a.c[2] * sk[2]) + a.c[3] * sk[3]) + a.c[4] * sk[4];
}

// ---- InfoNCE (losses.py:9-36), one workgroup of 1024 -------------------------------
__global__ __launch_bounds__(1024) void infonce_kernel(const float* __restrict__ x, const int* __restrict__ labels, int B, int D,
                                                        float temperature, float* __restrict__ loss, float* __restrict__ dx,
                                                        float* __restrict__ ws /* B*B sim + B*B G + B inv */) {
  __shared__ float red[17];
  float* sim = ws; float* G = ws + (size_t)B * B; float* inv = G + (size_t)B * B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  for (int i = wave; i < B; i += nw) {
    float q = 0.f;
    for (int d = lane; d < D; d += 64) q += x[(size_t)i * D + d] * x[(size_t)i * D + d];
    q = wave_sum(q);
    if (lane == 0) inv[i] = 1.f / fmaxf(sqrtf(q), 1e-12f);
  }
  __syncthreads();
  for (int p = wave; p < B * B; p += nw) {
    const int i = p / B, j = p % B;
    float q = 0.f;
    for (int d = lane; d < D; d += 64) q += x[(size_t)i * D + d] * x[(size_t)j * D + d];
    q = wave_sum(q);
    if (lane == 0) sim[p] = i == j ? -1e9f : q * inv[i] * inv[j];
  }
  __syncthreads();
  float part = 0.f;
  for (int i = tid; i < B; i += blockDim.x) {
    float mx = -INFINITY;
    for (int j = 0; j < B; ++j) mx = fmaxf(mx, sim[i * B + j] / temperature);
    float den = 0.f;
    for (int j = 0; j < B; ++j) den += __expf(sim[i * B + j] / temperature - mx);
    const float lse = mx + __logf(den);
    int cnt = 0; float sp = 0.f;
    for (int j = 0; j < B; ++j) if (j != i && labels[j] == labels[i]) { ++cnt; sp += sim[i * B + j] / temperature - lse; }
    const float cd = (float)max(cnt, 1);
    part += -sp / cd / B;
    for (int j = 0; j < B; ++j) {
      const float pj = j == i ? 0.f : __expf(sim[i * B + j] / temperature - lse);
      const float pos = (j != i && labels[j] == labels[i]) ? 1.f : 0.f;
      G[i * B + j] = ((cnt > 0 ? pj : 0.f) - pos / cd) / (B * temperature);   // d loss / d sim_ij
    }
  }
  part = block_sum(part, red);
  if (tid == 0) loss[0] = part;
  if (!dx) return;
  __syncthreads();
  // de_i = sum_j (G_ij + G_ji) e_j ; dx_i = (de_i - e_i <e_i,de_i>) * inv_i
  for (int i = wave; i < B; i += nw) {
    float de[8];
    float dot = 0.f;
    const int nd = (D + 63) / 64;
    for (int k = 0; k < nd && k < 8; ++k) {
      const int d = lane + 64 * k;
      float a = 0.f;
      if (d < D)
        for (int j = 0; j < B; ++j) a += (G[i * B + j] + G[j * B + i]) * x[(size_t)j * D + d] * inv[j];
      de[k] = a;
      if (d < D) dot += a * x[(size_t)i * D + d] * inv[i];
    }
    dot = wave_sum(dot);
    for (int k = 0; k < nd && k < 8; ++k) {
      const int d = lane + 64 * k;
      if (d < D) dx[(size_t)i * D + d] = (de[k] - x[(size_t)i * D + d] * inv[i] * dot) * inv[i];
    }
  }
}

// ---- margin loss (losses.py:45-57) -------------------------------------------------
__global__ __launch_bounds__(64) void margin_kernel(const float* __restrict__ c, int C, int D, float margin, float* __restrict__ loss,
                                                     float* __restrict__ dc) {
  const int lane = threadIdx.x;
  const int npairs = C * (C - 1) / 2;
  if (dc) for (int i = lane; i < C * D; i += 64) dc[i] = 0.f;
  float total = 0.f;
  for (int i = 0; i < C; ++i)
    for (int j = i + 1; j < C; ++j) {
      float q = 0.f;
      for (int d = lane; d < D; d += 64) { const float v = c[i * D + d] - c[j * D + d]; q += v * v; }
      const float dist = sqrtf(wave_sum(q));
      const float h = fmaxf(margin - dist, 0.f);
      total += h * h / npairs;
      if (dc && h > 0.f && dist > 0.f) {
        const float k = -2.f * h / (dist * npairs);
        for (int d = lane; d < D; d += 64) {
          const float v = c[i * D + d] - c[j * D + d];
          dc[i * D + d] += k * v; dc[j * D + d] -= k * v;
        }
      }
    }
  if (lane == 0) loss[0] = total;
}

// ---- HSIC with the reference's median heuristic (losses.py:138-191) -----------------
// ws: d2[(2B)^2] | K[B^2] | L[B^2] | rmK[B] | rmL[B] | misc[8] ; ihist in LDS.
__global__ __launch_bounds__(1024) void hsic_kernel(const float* __restrict__ s, const float* __restrict__ c, int B, int D,
                                                     float* __restrict__ loss, float* __restrict__ ds, float* __restrict__ dc,
                                                     float* __restrict__ ws) {
  __shared__ float red[17];
  __shared__ unsigned hist[256];
  __shared__ unsigned sel_prefix, sel_rank, sel_index;
  const int M = 2 * B, n = M * M;
  float* d2 = ws; float* K = d2 + n; float* L = K + B * B; float* rmK = L + B * B; float* rmL = rmK + B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  auto row = [&](int i) { return i < B ? s + (size_t)i * D : c + (size_t)(i - B) * D; };
  for (int p = wave; p < n; p += nw) {
    const int i = p / M, j = p % M;
    const float* a = row(i); const float* b = row(j);
    float q = 0.f;
    for (int d = lane; d < D; d += 64) { const float v = a[d] - b[d]; q += v * v; }
    q = wave_sum(q);
    if (lane == 0) d2[p] = q;
  }
  __syncthreads();
  // radix select of rank n/2-1 (lower median of the full matrix, diagonal included) on the
  // bit pattern of dist = sqrt(d2) (monotone for non-negative floats)
  if (tid == 0) { sel_prefix = 0; sel_rank = (unsigned)(n / 2 - 1); sel_index = 0xffffffffu; }
  __syncthreads();
  for (int pass = 3; pass >= 0; --pass) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned shift = pass * 8;
    const unsigned himask = pass == 3 ? 0u : (0xffffffffu << (shift + 8));
    const unsigned pref = sel_prefix;
    for (int p = tid; p < n; p += blockDim.x) {
      const unsigned bits = __float_as_uint(sqrtf(d2[p]));
      if ((bits & himask) == pref) atomicAdd(&hist[(bits >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned r = sel_rank, bsel = 0;
      for (unsigned bkt = 0; bkt < 256; ++bkt) { if (r < hist[bkt]) { bsel = bkt; break; } r -= hist[bkt]; }
      sel_rank = r; sel_prefix = pref | (bsel << shift);
    }
    __syncthreads();
  }
  const unsigned sbits = sel_prefix;
  for (int p = tid; p < n; p += blockDim.x)
    if (__float_as_uint(sqrtf(d2[p])) == sbits) atomicMin(&sel_index, (unsigned)p);
  __syncthreads();
  const float sigma = __uint_as_float(sbits);
  const float i2s2 = 1.f / (2.f * sigma * sigma);
  for (int p = tid; p < B * B; p += blockDim.x) {
    const int i = p / B, j = p % B;
    K[p] = __expf(-d2[i * M + j] * i2s2);
    L[p] = __expf(-d2[(B + i) * M + (B + j)] * i2s2);
  }
  __syncthreads();
  for (int i = tid; i < B; i += blockDim.x) {
    float a = 0.f, b = 0.f;
    for (int j = 0; j < B; ++j) { a += K[i * B + j]; b += L[i * B + j]; }
    rmK[i] = a / B; rmL[i] = b / B;
  }
  __syncthreads();
  float tk = 0.f, tl = 0.f;
  for (int i = tid; i < B; i += blockDim.x) { tk += rmK[i]; tl += rmL[i]; }
  const float mK = block_sum(tk, red) / B;
  const float mL = block_sum(tl, red) / B;
  const float nrm = 1.f / ((float)(B - 1) * (B - 1));
  // hsic = sum Kc_ij L_ij * nrm ; gsig = sum (Lc K nS + Kc L nC) * nrm / sigma^3
  float hp = 0.f, gsp = 0.f;
  for (int p = tid; p < B * B; p += blockDim.x) {
    const int i = p / B, j = p % B;
    const float Kc = K[p] - rmK[i] - rmK[j] + mK, Lc = L[p] - rmL[i] - rmL[j] + mL;
    hp += Kc * L[p];
    gsp += Lc * K[p] * d2[i * M + j] + Kc * L[p] * d2[(B + i) * M + (B + j)];
  }
  const float hsic = block_sum(hp, red) * nrm;
  const float gsig = block_sum(gsp, red) * nrm / (sigma * sigma * sigma);
  if (tid == 0) loss[0] = hsic;
  if (!ds || !dc) return;
  const unsigned pq = sel_index;
  const int pi = (int)(pq / M), qi = (int)(pq % M);
  const float gd = (pi != qi && sigma > 0.f) ? gsig / sigma : 0.f;
  // dS_i = -(2/sigma^2) sum_j Lc_ij K_ij (S_i - S_j) * nrm  (+ sigma path on rows p, q)
  for (int r = wave; r < M; r += nw) {
    const bool is_s = r < B;
    const int i = is_s ? r : r - B;
    const float* X = is_s ? s : c;
    const float* Aij = is_s ? K : L;
    const float* rmO = is_s ? rmL : rmK;       // centred matrix of the OTHER kernel
    const float* Oij = is_s ? L : K;
    const float mO = is_s ? mL : mK;
    float* dX = is_s ? ds : dc;
    for (int d = lane; d < D; d += 64) {
      float a = 0.f;
      for (int j = 0; j < B; ++j) {
        const float Oc = Oij[i * B + j] - rmO[i] - rmO[j] + mO;
        a += Oc * Aij[i * B + j] * (X[(size_t)i * D + d] - X[(size_t)j * D + d]);
      }
      a *= -4.f * i2s2 * nrm;
      if (r == pi) a += gd * (row(pi)[d] - row(qi)[d]);
      if (r == qi) a -= gd * (row(pi)[d] - row(qi)[d]);
      dX[(size_t)i * D + d] = a;
    }
  }
}

// ---- cross-covariance penalty (losses.py:146-150): ||Sc^T Cc / (B-1)||_F^2 = sum_ij <Sc_i,Sc_j><Cc_i,Cc_j> / (B-1)^2
// ws: mean_s[D] | mean_c[D] | GS[B*B] | GC[B*B]
__global__ __launch_bounds__(1024) void crosscov_kernel(const float* __restrict__ s, const float* __restrict__ c, int B, int D,
                                                         float* __restrict__ loss, float* __restrict__ ds, float* __restrict__ dc,
                                                         float* __restrict__ ws) {
  __shared__ float red[17];
  float* ms = ws; float* mc = ws + D; float* GS = mc + D; float* GC = GS + (size_t)B * B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  for (int d = tid; d < D; d += blockDim.x) {
    float a = 0.f, b = 0.f;
    for (int i = 0; i < B; ++i) { a += s[(size_t)i * D + d]; b += c[(size_t)i * D + d]; }
    ms[d] = a / B; mc[d] = b / B;
  }
  __syncthreads();
  for (int p = wave; p < B * B; p += nw) {
    const int i = p / B, j = p % B;
    float a = 0.f, b = 0.f;
    for (int d = lane; d < D; d += 64) {
      a += (s[(size_t)i * D + d] - ms[d]) * (s[(size_t)j * D + d] - ms[d]);
      b += (c[(size_t)i * D + d] - mc[d]) * (c[(size_t)j * D + d] - mc[d]);
    }
    a = wave_sum(a); b = wave_sum(b);
    if (lane == 0) { GS[p] = a; GC[p] = b; }
  }
  __syncthreads();
  const float nrm = 1.f / ((float)(B - 1) * (B - 1));
  float part = 0.f;
  for (int p = tid; p < B * B; p += blockDim.x) part += GS[p] * GC[p];
  part = block_sum(part, red) * nrm;
  if (tid == 0) loss[0] = part;
  if (!ds || !dc) return;
  // d/dS_i = 2 nrm sum_j GC_ij Sc_j (centring adds nothing: columns of GC sum to 0), same for C with GS
  for (int idx = tid; idx < B * D; idx += blockDim.x) {
    const int i = idx / D, d = idx % D;
    float a = 0.f, b = 0.f;
    for (int j = 0; j < B; ++j) {
      a += GC[i * B + j] * (s[(size_t)j * D + d] - ms[d]);
      b += GS[i * B + j] * (c[(size_t)j * D + d] - mc[d]);
    }
    ds[idx] = 2.f * nrm * a; dc[idx] = 2.f * nrm * b;
  }
}

__global__ void cross_entropy_kernel(const float* __restrict__ logits, const int* __restrict__ target, int R, int C, float* __restrict__ loss,
                                     float* __restrict__ dl) {
  __shared__ float red[17];
  float part = 0.f;
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    const float* z = logits + (size_t)r * C;
    float mx = -INFINITY;
    for (int k = 0; k < C; ++k) mx = fmaxf(mx, z[k]);
    float den = 0.f;
    for (int k = 0; k < C; ++k) den += __expf(z[k] - mx);
    const float lse = mx + __logf(den);
    part += (lse - z[target[r]]) / R;
    if (dl) for (int k = 0; k < C; ++k) dl[(size_t)r * C + k] = (__expf(z[k] - lse) - (k == target[r] ? 1.f : 0.f)) / R;
  }
  part = block_sum(part, red);
  if (threadIdx.x == 0) loss[0] = part;
}

__global__ void softmax_entropy_kernel(const float* __restrict__ logits, int R, int C, float* __restrict__ loss, float* __restrict__ dl) {
  __shared__ float red[17];
  float part = 0.f;
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    const float* z = logits + (size_t)r * C;
    float mx = -INFINITY;
    for (int k = 0; k < C; ++k) mx = fmaxf(mx, z[k]);
    float den = 0.f;
    for (int k = 0; k < C; ++k) den += __expf(z[k] - mx);
    float H = 0.f, gp = 0.f;
    for (int k = 0; k < C; ++k) {
      const float p = __expf(z[k] - mx) / den;
      H -= p * __logf(p + 1e-8f);
      gp += -(__logf(p + 1e-8f) + p / (p + 1e-8f)) * p;
    }
    part += H / R;
    if (dl) for (int k = 0; k < C; ++k) {
      const float p = __expf(z[k] - mx) / den;
      const float g = -(__logf(p + 1e-8f) + p / (p + 1e-8f));
      dl[(size_t)r * C + k] = p * (g - gp) / R;
    }
  }
  part = block_sum(part, red);
  if (threadIdx.x == 0) loss[0] = part;
}

__global__ void scale_kernel(const float* __restrict__ x, const float* __restrict__ dscale, float hscale, float* __restrict__ y, size_t n, int accumulate) {
  const float sc = hscale * (dscale ? dscale[0] : 1.f);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = accumulate ? y[i] + sc * x[i] : sc * x[i];
}

}  // namespace

extern "C" int ast_recon_loss(const float* out, const float* tgt, int64_t tgt_ld, int B, int S, int T, int Fq, float c_mse, float c_mag,
                              float c_phase, float c_temporal, float c_spectral, float* sums, float* grad, void* stream) {
  if (!out || !tgt || !sums || B <= 0 || S <= 0 || T <= 0 || Fq <= 0 || tgt_ld < Fq) AST_FAIL("ast_recon_loss: bad args");
  hipStream_t s = (hipStream_t)stream;
  AST_HIP(hipMemsetAsync(sums, 0, 5 * sizeof(float), s));
  const size_t total = (size_t)B * T * Fq;
  if (total >= (1ull << 31)) AST_FAIL("ast_recon_loss: more than 2^31 bins per section");
  // every workgroup ends in 5 same-address f32 atomics, which serialise at ~10 ns each (1024 workgroups of 256: ~50 of the
  // kernel's 95 us; 4608: +250 us): the same 256 K threads as 256 workgroups of 1024
  static const int max_blocks = getenv("AST_RECON_BLOCKS") ? atoi(getenv("AST_RECON_BLOCKS")) : 256;
  const int grid = (int)std::min<size_t>((total + 1023) / 1024, (size_t)std::max(1, max_blocks));
  hipLaunchKernelGGL(recon_loss_kernel, dim3(grid), dim3(1024), 0, s, out, tgt, tgt_ld, B, S, T, Fq, c_mse, c_mag, c_phase, c_temporal,
                     c_spectral, sums, grad, 1);
  AST_CHECK_LAUNCH();
  return 0;
}

// One thread per (b, t, f) bin (256-thread workgroups: every load of the pass in flight at once instead of 4-5 dependent trips per
// thread), the workgroups' five partial sums added into AST_RECON_SLOTS rows (72 adders per address instead of 4 608), and a
// finishing launch that reduces the rows in a fixed order and forms the weighted total and the five reported means -- which took
// three element-wise launches behind the 256-workgroup pass (47.9 us on the step's chain).
extern "C" int ast_recon_loss_total(const float* out, const float* tgt, int64_t tgt_ld, int B, int S, int T, int Fq, const float* coef5,
                                    const float* inv5, float* ws, float* res11, float* grad, void* stream) {
  if (!out || !tgt || !coef5 || !inv5 || !ws || !res11 || B <= 0 || S <= 0 || T <= 0 || Fq <= 0 || tgt_ld < Fq) AST_FAIL("ast_recon_loss_total: bad args");
  hipStream_t s = (hipStream_t)stream;
  AST_HIP(hipMemsetAsync(ws, 0, AST_RECON_SLOTS * 5 * sizeof(float), s));
  const size_t total = (size_t)B * T * Fq;
  if (total >= (1ull << 31)) AST_FAIL("ast_recon_loss_total: more than 2^31 bins per section");
  const int grid = (int)std::min<size_t>((total + 255) / 256, 65536);
  hipLaunchKernelGGL(recon_loss_kernel, dim3(grid), dim3(256), 0, s, out, tgt, tgt_ld, B, S, T, Fq, coef5[0], coef5[1], coef5[2], coef5[3],
                     coef5[4], ws, grad, AST_RECON_SLOTS);
  ReconFin a;
  for (int k = 0; k < 5; ++k) { a.c[k] = coef5[k]; a.inv[k] = inv5[k]; }
  hipLaunchKernelGGL(recon_finish_kernel, dim3(1), dim3(64), 0, s, ws, AST_RECON_SLOTS, a, res11);
  AST_CHECK_LAUNCH();
  return 0;
}

extern "C" int ast_infonce(const float* emb, const int32_t* labels, int B, int D, float temperature, float* loss, float* demb, float* ws,
                           void* stream) {
  if (!emb || !labels || !loss || !ws || B < 1 || B > 256 || D < 1 || D > 512) AST_FAIL("ast_infonce: bad args (B<=256, D<=512)");
  hipLaunchKernelGGL(infonce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, emb, labels, B, D, temperature, loss, demb, ws);
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_margin(const float* cls, int C, int D, float margin, float* loss, float* dcls, void* stream) {
  if (!cls || !loss || C < 2) AST_FAIL("ast_margin: bad args");
  hipLaunchKernelGGL(margin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cls, C, D, margin, loss, dcls);
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_hsic(const float* s, const float* c, int B, int D, float* loss, float* ds, float* dc, float* ws, void* stream) {
  if (!s || !c || !loss || !ws || B < 2 || B > 128) AST_FAIL("ast_hsic: bad args (2<=B<=128)");
  hipLaunchKernelGGL(hsic_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, s, c, B, D, loss, ds, dc, ws);
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_crosscov(const float* s, const float* c, int B, int D, float* loss, float* ds, float* dc, float* ws, void* stream) {
  if (!s || !c || !loss || !ws || B < 2 || B > 256) AST_FAIL("ast_crosscov: bad args (2<=B<=256)");
  hipLaunchKernelGGL(crosscov_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, s, c, B, D, loss, ds, dc, ws);
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_cross_entropy(const float* logits, const int32_t* target, int R, int C, float* loss, float* dlogits, void* stream) {
  if (!logits || !target || !loss || R < 1 || C < 1) AST_FAIL("ast_cross_entropy: bad args");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, R, C, loss, dlogits);
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_softmax_entropy(const float* logits, int R, int C, float* loss, float* dlogits, void* stream) {
  if (!logits || !loss || R < 1 || C < 1) AST_FAIL("ast_softmax_entropy: bad args");
  hipLaunchKernelGGL(softmax_entropy_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, R, C, loss, dlogits);
  AST_CHECK_LAUNCH();
  return 0;
}
extern "C" int ast_scale(const float* x, const float* dscale, float hscale, float* y, int64_t n, int accumulate, void* stream) {
  if (!x || !y || n < 0) AST_FAIL("ast_scale: bad args");
  if (n == 0) return 0;
  const int grid = (int)std::min<size_t>(((size_t)n + 255) / 256, 4096);
  hipLaunchKernelGGL(scale_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, dscale, hscale, y, (size_t)n, accumulate);
  AST_CHECK_LAUNCH();
  return 0;
}
