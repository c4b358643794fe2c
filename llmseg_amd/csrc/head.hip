// Mask-selection head kernels (HBM-bound): fused upsample+mask-pooling, cosine scoring, the two live losses
// (softmax-KL align + weighted-MSE IoP regression), the north_star-named dice/BCE losses, and the shifted
// cross-entropy over the LM logits.  See include/llmseg_hip.h for the reference lines each one replaces.
#include <algorithm>
#include "common.h"
#include "llmseg_hip.h"

namespace {

// ---- pooled[k][:] = (segs[k] . U) . feat / (sum segs[k] + 1e-8) -------------------------------------------------------
// Stage 1 (this kernel, one workgroup per proposal): pull the S x S mask back through the ADJOINT of the bilinear
// interpolation, separably and without atomics: rows first (tmp[py][sx] = sum_px m[py][px] Ux[px][sx], half of the rows at a
// time in LDS), then columns (acc[sy][sx] += sum_py Uy[py][sy] tmp[py][sx]).  `segs` (K*S*S bf16, the only large operand) is
// read from HBM once.  Because every bilinear weight row sums to 1, sum_s acc[s] == sum_p m[p], so the normaliser comes for
// free.  Output: wn[k][s] = acc[s] / (sum + 1e-8) as bf16 (+ optional raw fp32 acc / sum for the backward pass).
// Stage 2 is a plain GEMM pooled = wn . feat on the matrix cores (feat is channels-last [g*g][C] = "W stored [K][N]").
// The [C][S][S] upsampled tensor of the reference is never formed.
__device__ __forceinline__ float bilin_w(int p, int cell, float scale, int g) {
  // weight of destination pixel p on source cell `cell` (F.interpolate bilinear, align_corners=False, clamped at the borders)
  const float s = fmaxf(0.f, ((float)p + 0.5f) * scale - 0.5f);
  const int c0 = (int)s, c1 = min(c0 + 1, g - 1);
  const float l1 = s - (float)c0;
  return (c0 == cell ? 1.f - l1 : 0.f) + (c1 == cell ? l1 : 0.f);
}

__global__ __launch_bounds__(256) void mask_pullback_kernel(const bf16_t* __restrict__ segs, bf16_t* __restrict__ wn, float* __restrict__ pb_out,
                                                           float* __restrict__ wsum_out, int g, int S) {
  extern __shared__ float lds[];                 // tmp[HALF][g] | acc[g][g] | red[16]
  const int HALF = (S + 1) / 2;
  float* tmp = lds;
  float* acc = lds + HALF * g;
  float* red = acc + g * g;
  const int k = blockIdx.x, tid = threadIdx.x;
  const float scale = (float)g / (float)S;
  const bf16_t* m = segs + (long)k * S * S;
  for (int i = tid; i < g * g; i += blockDim.x) acc[i] = 0.f;
  const int rows_par = blockDim.x / g;          // rows handled per sweep (blockDim.x is a multiple of g; host checks)
  const int sx = tid % g, ro = tid / g;
  // taps of this thread's source column sx (a superset range; bilin_w() zeroes the rest)
  const int plo = max(0, (int)floorf(((float)sx - 0.5f) / scale - 0.5f) - 1);
  const int phi = min(S - 1, (int)ceilf(((float)sx + 1.5f) / scale - 0.5f) + 1);
  for (int h0 = 0; h0 < S; h0 += HALF) {
    const int h1 = min(S, h0 + HALF);
    __syncthreads();                             // acc zero-fill / previous half's column pass done
    if (phi - plo < 16) {                        // common case (S/g <= 6): tap weights of this column live in registers
      float wx[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) wx[j] = (plo + j <= phi) ? bilin_w(plo + j, sx, scale, g) : 0.f;
      for (int py = h0 + ro; py < h1; py += rows_par) {
        const bf16_t* row = m + (long)py * S;
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += bf2f(row[min(plo + j, S - 1)]) * wx[j];
        tmp[(py - h0) * g + sx] = t;
      }
    } else {
      for (int py = h0 + ro; py < h1; py += rows_par) {
        float t = 0.f;
        for (int px = plo; px <= phi; ++px) t += bf2f(m[(long)py * S + px]) * bilin_w(px, sx, scale, g);
        tmp[(py - h0) * g + sx] = t;
      }
    }
    __syncthreads();
    for (int sy = ro; sy < g; sy += rows_par) {
      const int qlo = max(h0, (int)floorf(((float)sy - 0.5f) / scale - 0.5f) - 1);
      const int qhi = min(h1 - 1, (int)ceilf(((float)sy + 1.5f) / scale - 0.5f) + 1);
      float a = 0.f;
      for (int py = qlo; py <= qhi; ++py) a += tmp[(py - h0) * g + sx] * bilin_w(py, sy, scale, g);
      acc[sy * g + sx] += a;
    }
  }
  __syncthreads();
  float part = 0.f;
  for (int i = tid; i < g * g; i += blockDim.x) part += acc[i];
  const float wsum = block_sum(part, red);
  const float inv = 1.f / (wsum + 1e-8f);
  for (int i = tid; i < g * g; i += blockDim.x) {
    wn[(long)k * g * g + i] = f2bf(acc[i] * inv);
    if (pb_out) pb_out[(long)k * g * g + i] = acc[i];
  }
  if (wsum_out && tid == 0) wsum_out[k] = wsum;
}

// The same pull-back for the shape the path really has (S = 256 proposals pixels, g = 64 cells, 4 pixels per cell) with COALESCED loads:
// the mask is walked in chunks of 8 rows; thread t loads ONE 16-byte piece (row t / 32, pixels 8c .. 8c + 7, c = t % 32) -- a row is 32
// lanes x 16 B = one 512-byte burst -- and computes in registers the horizontal partial sums of the four source columns its 8 pixels
// touch (2c-1, 2c, 2c+1, 2c+2: a source column gathers from pixels 4 sx - 2 .. 4 sx + 5); the two outer partials go to the neighbour
// lanes by shuffle.  The 8 x 64 row results pass through LDS once for the vertical taps (same stencil), accumulated in acc[64][64].
// The previous kernel issued 16 scalar 2-byte loads per thread per row (0.47 TB/s); this one streams the masks (read once) with the
// next chunk's load in flight behind the current chunk's arithmetic.
__global__ __launch_bounds__(256) void mask_pullback_s256_kernel(const bf16_t* __restrict__ segs, bf16_t* __restrict__ wn, float* __restrict__ pb_out,
                                                                float* __restrict__ wsum_out) {
  constexpr int S = 256, G = 64;
  __shared__ float tmp[2][8][G];                 // horizontal sums of the current 8-row chunk (double-buffered: one barrier per chunk)
  __shared__ float acc[G * G];
  __shared__ float red[16];
  const int k = blockIdx.x, tid = threadIdx.x;
  const int r = tid >> 5, c = tid & 31;          // this thread's row inside a chunk / 8-pixel piece inside the row
  const float scale = 0.25f;
  const bf16_t* m = segs + (long)k * S * S + r * S + c * 8;
  // horizontal tap weights of this thread's 8 pixels onto its four source columns (exact bilinear adjoint incl. the clamped borders)
  float wa[8], wb[8], wc[2], wd[2];
#pragma unroll
  for (int j = 0; j < 8; ++j) { wa[j] = bilin_w(8 * c + j, 2 * c, scale, G); wb[j] = bilin_w(8 * c + j, 2 * c + 1, scale, G); }
#pragma unroll
  for (int j = 0; j < 2; ++j) { wc[j] = c > 0 ? bilin_w(8 * c + j, 2 * c - 1, scale, G) : 0.f; wd[j] = c < 31 ? bilin_w(8 * c + 6 + j, 2 * c + 2, scale, G) : 0.f; }
  // vertical pass: thread t owns (target slot q = t / 64 -> sy = 2C - 1 + q, column sx = t % 64) of every chunk C
  const int q = tid >> 6, sx = tid & 63;
  for (int i = tid; i < G * G; i += 256) acc[i] = 0.f;
  uint4 cur = *reinterpret_cast<const uint4*>(m);
  for (int C = 0; C < S / 8; ++C) {
    uint4 nxt = cur;
    if (C + 1 < S / 8) nxt = *reinterpret_cast<const uint4*>(m + (long)(C + 1) * 8 * S);
    float e[8];
    unpack8(cur, e);
    float ha = 0.f, hb = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ha = fmaf(e[j], wa[j], ha); hb = fmaf(e[j], wb[j], hb); }
    const float hc = e[0] * wc[0] + e[1] * wc[1];          // -> source column 2c - 1 (owned by lane c - 1 as its hb)
    const float hd = e[6] * wd[0] + e[7] * wd[1];          // -> source column 2c + 2 (owned by lane c + 1 as its ha)
    const float from_left = __shfl_up(hd, 1, 32), from_right = __shfl_down(hc, 1, 32);
    if (c > 0) ha += from_left;
    if (c < 31) hb += from_right;
    float* tb = &tmp[C & 1][r][0];
    *reinterpret_cast<float2*>(tb + 2 * c) = make_float2(ha, hb);
    __syncthreads();
    const int sy = 2 * C - 1 + q;
    if (sy >= 0 && sy < G) {
      float a = 0.f;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) a = fmaf(tmp[C & 1][rr][sx], bilin_w(8 * C + rr, sy, scale, G), a);
      acc[sy * G + sx] += a;                                // (sy, sx) is touched by exactly one thread per chunk
    }
    cur = nxt;
  }
  __syncthreads();
  float part = 0.f;
  for (int i = tid; i < G * G; i += 256) part += acc[i];
  const float wsum = block_sum(part, red);
  const float inv = 1.f / (wsum + 1e-8f);
  for (int i = tid; i < G * G / 8; i += 256) {
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = acc[i * 8 + j] * inv;
    *reinterpret_cast<uint4*>(wn + (long)k * G * G + i * 8) = pack8(o);
    if (pb_out) {
      *reinterpret_cast<float4*>(pb_out + (long)k * G * G + i * 8) = make_float4(acc[i * 8], acc[i * 8 + 1], acc[i * 8 + 2], acc[i * 8 + 3]);
      *reinterpret_cast<float4*>(pb_out + (long)k * G * G + i * 8 + 4) = make_float4(acc[i * 8 + 4], acc[i * 8 + 5], acc[i * 8 + 6], acc[i * 8 + 7]);
    }
  }
  if (wsum_out && tid == 0) wsum_out[k] = wsum;
}

// ---- cosine scores: one wave per proposal ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cosine_kernel(const bf16_t* __restrict__ t, const bf16_t* __restrict__ e, float* __restrict__ sim, int K, int D) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  float dot = 0.f, ne = 0.f, ntt = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float a = bf2f(t[d]), b = bf2f(e[(long)k * D + d]);
    dot += a * b; ne += b * b; ntt += a * a;
  }
  dot = wave_sum(dot); ne = wave_sum(ne); ntt = wave_sum(ntt);
  if (lane == 0) sim[k] = dot / (sqrtf(ne) * sqrtf(ntt));
}

// ---- align (KL) + IoP regression losses: one workgroup per (image, round) item; items are contiguous in every operand -----
__global__ __launch_bounds__(256) void align_reg_kernel(const bf16_t* __restrict__ e, const bf16_t* __restrict__ t, const float* __restrict__ gt_iou,
                                                       const bf16_t* __restrict__ pred, const float* __restrict__ gt_iop, float* __restrict__ out,
                                                       float* __restrict__ d_e, float* __restrict__ d_t, float* __restrict__ d_pred, int K, int D,
                                                       float tau) {
  extern __shared__ float sm[];                  // cos[K], enorm[K], gsoft[K], red[16]
  float* cs = sm; float* en = sm + K; float* gs = sm + 2 * K; float* red = sm + 3 * K;
  {
    const long it = blockIdx.x;
    e += it * K * D; t += it * D; gt_iou += it * K; pred += it * K; gt_iop += it * K; out += it * 2;
    if (d_e) d_e += it * K * D;
    if (d_t) d_t += it * D;
    if (d_pred) d_pred += it * K;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float tn2 = 0.f;
  for (int d = threadIdx.x; d < D; d += blockDim.x) { const float a = bf2f(t[d]); tn2 += a * a; }
  tn2 = block_sum(tn2, red);
  const float tn = sqrtf(tn2);
  for (int k = wv; k < K; k += nw) {
    float dot = 0.f, ne = 0.f;
    for (int d = lane; d < D; d += 64) { const float a = bf2f(t[d]), b = bf2f(e[(long)k * D + d]); dot += a * b; ne += b * b; }
    dot = wave_sum(dot); ne = wave_sum(ne);
    if (lane == 0) { en[k] = sqrtf(ne); cs[k] = dot / (sqrtf(ne) * tn); }
  }
  __syncthreads();
  // softmax over K of cos/tau and gt/tau
  float mx_s = -1e30f, mx_g = -1e30f;
  for (int k = threadIdx.x; k < K; k += blockDim.x) { mx_s = fmaxf(mx_s, cs[k] / tau); mx_g = fmaxf(mx_g, gt_iou[k] / tau); }
  mx_s = block_max(mx_s, red); mx_g = block_max(mx_g, red);
  float se = 0.f, ge = 0.f;
  for (int k = threadIdx.x; k < K; k += blockDim.x) { se += __expf(cs[k] / tau - mx_s); ge += __expf(gt_iou[k] / tau - mx_g); }
  se = block_sum(se, red); ge = block_sum(ge, red);
  const float lse_s = mx_s + __logf(se), lse_g = mx_g + __logf(ge);
  float kl = 0.f, rg = 0.f;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const float lg = gt_iou[k] / tau - lse_g, ls = cs[k] / tau - lse_s;
    const float pg = __expf(lg);
    gs[k] = pg;
    kl += pg > 0.f ? pg * (lg - ls) : 0.f;          // F.kl_div(log q, p): p * (log p - log q), 0 where p == 0
    const float pr = bf2f(pred[k]), g = gt_iop[k];
    const float w = __expf(g - 1.f);
    rg += (pr - g) * (pr - g) * w;
    if (d_pred && blockIdx.y == 0) d_pred[k] = 2.f * (pr - g) * w * 50.f / (float)K;
  }
  kl = block_sum(kl, red); rg = block_sum(rg, red);
  if (threadIdx.x == 0 && blockIdx.y == 0) { out[0] = kl; out[1] = rg / (float)K * 50.f; }
  if (d_e || d_t) {
    // dKL/dcos_k = (softmax_s[k] - softmax_g[k]) / tau;  cos_k = <e_k, t> / (|e_k||t|)
    // round 5: the per-key factors are formed ONCE (gs[k] <- dKL/dcos_k, en[k] <- 1 / |e_k|): the column loop below used to evaluate an exp and
    // three divisions per (k, d) inside a serial chain of K iterations per thread (164 us for two items of K = 256 -- as long as two Llama
    // GEMMs); same arithmetic per element otherwise (gk, then the two products), so the values are unchanged up to the reciprocal's rounding
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
      gs[k] = (__expf(cs[k] / tau - lse_s) - gs[k]) / tau;
      en[k] = 1.f / en[k];
    }
    __syncthreads();
    // gridDim.y workgroups share an item's columns (each has formed the item's statistics itself: K x D bf16 out of L2); inside a workgroup
    // the 4 waves take every 4th key of a 64-column block, the four partial d_t sums meet in LDS in wave order (fixed: no atomics)
    const float itn = 1.f / tn;
    const int cols = (D + (int)gridDim.y - 1) / (int)gridDim.y, d_lo = (int)blockIdx.y * cols, d_hi = min(D, d_lo + cols);
    for (int d0 = d_lo; d0 < d_hi; d0 += 64) {
      const int d = d0 + lane;
      const bool on = d < d_hi;
      float acc_t = 0.f;
      const float td = on ? bf2f(t[d]) : 0.f;
      if (on) {
#pragma unroll 4
        for (int k = wv; k < K; k += nw) {
          const float gk = gs[k], ie = en[k], c = cs[k];
          const float ed = bf2f(e[(long)k * D + d]);
          if (d_e) d_e[(long)k * D + d] = gk * (td * ie * itn - c * ed * ie * ie);
          acc_t += gk * (ed * ie * itn - c * td * itn * itn);
        }
      }
      __syncthreads();                               // (red is reused per 64-column block)
      if (d_t) {
        float* part = sm + 3 * K + 16;               // [nw][64] behind red[16]
        part[wv * 64 + lane] = acc_t;
        __syncthreads();
        if (wv == 0 && on) {
          float sum = 0.f;
          for (int w2 = 0; w2 < nw; ++w2) sum += part[w2 * 64 + lane];
          d_t[d] = sum;
        }
      }
    }
  }
}

// ---- dice + BCE-with-logits, one workgroup per mask ----------------------------------------------------------------------
// part[m][2] = this mask's (dice, bce) terms; fold_column_kernel adds the masks in a fixed order (no atomics)
__global__ __launch_bounds__(256) void dice_bce_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ part, long HW,
                                                      float num_masks) {
  __shared__ float red[16];
  const long m = blockIdx.x;
  float sxy = 0.f, sx = 0.f, sy = 0.f, bce = 0.f;
  for (long i = threadIdx.x; i < HW; i += blockDim.x) {
    const float l = x[m * HW + i], t = y[m * HW + i];
    const float s = 1.f / (1.f + __expf(-l));
    sxy += s * t; sx += s; sy += t;
    bce += fmaxf(l, 0.f) - l * t + log1pf(__expf(-fabsf(l)));
  }
  sxy = block_sum(sxy, red); sx = block_sum(sx, red); sy = block_sum(sy, red); bce = block_sum(bce, red);
  if (threadIdx.x == 0) {
    const float sc = 1000.f, eps = 1e-6f;
    const float dice = 1.f - (2.f * sxy / sc + eps) / (sx / sc + sy / sc + eps);
    part[2 * m] = dice / (num_masks + 1e-8f);
    part[2 * m + 1] = bce / (float)HW / (num_masks + 1e-8f);
  }
}

// Backward of the pair above in one launch: dx = g[0] * d(dice)/dx + g[1] * d(bce)/dx, one workgroup per mask (sums first, then the
// per-pixel gradient).  dice_m = 1 - N / D with N = 2 sum(s t) / sc + eps, D = (sum s + sum t) / sc + eps, s = sigmoid(x):
// d dice_m / dx_i = -(2 t_i D - N) / (sc D^2) * s_i (1 - s_i);  d bce / dx_i = (s_i - t_i) / HW; both / (num_masks + 1e-8).
__global__ __launch_bounds__(256) void dice_bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g,
                                                          float* __restrict__ dx, long HW, float num_masks) {
  __shared__ float red[16];
  const long m = blockIdx.x;
  float sxy = 0.f, sx = 0.f, sy = 0.f;
  for (long i = threadIdx.x; i < HW; i += blockDim.x) {
    const float l = x[m * HW + i], t = y[m * HW + i];
    const float s = 1.f / (1.f + __expf(-l));
    sxy += s * t; sx += s; sy += t;
  }
  sxy = block_sum(sxy, red); sx = block_sum(sx, red); sy = block_sum(sy, red);
  const float sc = 1000.f, eps = 1e-6f;
  const float Nn = 2.f * sxy / sc + eps, D = sx / sc + sy / sc + eps;
  const float inv = 1.f / (num_masks + 1e-8f), gd = g[0] * inv / (sc * D * D), gb = g[1] * inv / (float)HW;
  for (long i = threadIdx.x; i < HW; i += blockDim.x) {
    const float l = x[m * HW + i], t = y[m * HW + i];
    const float s = 1.f / (1.f + __expf(-l));
    dx[m * HW + i] = -gd * (2.f * t * D - Nn) * s * (1.f - s) + gb * (s - t);
  }
}

// ---- shifted CE: one workgroup per (n, t) with a valid label -------------------------------------------------------------
// part[row][2] = (nll, 1) of a scored position, (0, 0) of an ignored one; fold_column_kernel adds the rows in a fixed order (no atomics)
__global__ __launch_bounds__(256) void ce_kernel(const bf16_t* __restrict__ logits, const int64_t* __restrict__ labels, float* __restrict__ part, int T,
                                                long V, long ldl) {
  __shared__ float red[16];
  const int n = blockIdx.x / (T - 1), t = blockIdx.x % (T - 1);
  const long lab = labels[(long)n * T + t + 1];
  if (lab < 0 || lab >= V) {                              // ignore_index (-100)
    if (threadIdx.x == 0) { part[2 * (long)blockIdx.x] = 0.f; part[2 * (long)blockIdx.x + 1] = 0.f; }
    return;
  }
  const bf16_t* row = logits + ((long)n * T + t) * ldl;
  float mx = -1e30f, s = 0.f;
  if (ce_row_fast(row, V, ldl)) {                         // the row is read once (registers), 8-byte loads
    CeRow r;
    r.load(row, V);
    r.stats(red, mx, s);
  } else {
    for (long i = threadIdx.x; i < V; i += blockDim.x) mx = fmaxf(mx, bf2f(row[i]));
    mx = block_max(mx, red);
    for (long i = threadIdx.x; i < V; i += blockDim.x) s += __expf(bf2f(row[i]) - mx);
    s = block_sum(s, red);
  }
  if (threadIdx.x == 0) {
    part[2 * (long)blockIdx.x] = mx + __logf(s) - bf2f(row[lab]);
    part[2 * (long)blockIdx.x + 1] = 1.f;
  }
}

}  // namespace

// SAM mask post-processing (sam.py:137-172): low-res logits 256 x 256 -> bilinear (align_corners = False) to img x img, crop to the
// input size, bilinear to the original size, all in fp32 -- fused: every output pixel evaluates its four stage-1 samples on the fly
// (16 reads of a 256 KiB mask that lives in L2).  torch's upsample_bilinear2d arithmetic: src = max(0, scale (o + 0.5) - 0.5),
// i0 = floor(src), i1 = i0 + (i0 < in - 1), weights (1 - frac, frac), value = w_y0 (w_x0 v00 + w_x1 v01) + w_y1 (w_x0 v10 + w_x1 v11).
// `nested` = the layout the two stride-2 transposed convolutions of the mask decoder leave when run as GEMMs on token-major rows:
// pixel (Y, X) sits at ((y 64 + x) 4 + dy1 2 + dx1) 4 + dy2 2 + dx2 with Y = 4 y + 2 dy1 + dy2 (no pixel-shuffle pass is needed).
__device__ __forceinline__ float sam_low(const float* __restrict__ m, int Y, int X, int nested) {
  if (!nested) return m[Y * 256 + X];
  const int tok = (Y >> 2) * 64 + (X >> 2);
  return m[(tok * 4 + ((Y >> 1) & 1) * 2 + ((X >> 1) & 1)) * 4 + (Y & 1) * 2 + (X & 1)];
}
__device__ __forceinline__ void bil_coord(int o, float scale, int in, int& i0, int& i1, float& w0, float& w1) {
  const float src = fmaxf(0.f, __fmul_rn(scale, (float)o + 0.5f) - 0.5f);
  i0 = (int)src;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  w1 = src - (float)i0;
  w0 = 1.f - w1;
}
__global__ __launch_bounds__(256) void sam_postprocess_kernel(const float* __restrict__ low, float* __restrict__ out, int img, int in_h, int in_w, int oh, int ow,
                                                             int nested) {
  const int b = blockIdx.z, oy = blockIdx.y;
  const int ox = blockIdx.x * blockDim.x + threadIdx.x;
  if (ox >= ow) return;
  const float* m = low + (long)b * 65536;
  const float s1 = 256.f / (float)img, s2y = (float)in_h / (float)oh, s2x = (float)in_w / (float)ow;
  int y0, y1, x0, x1;
  float wy0, wy1, wx0, wx1;
  bil_coord(oy, s2y, in_h, y0, y1, wy0, wy1);
  bil_coord(ox, s2x, in_w, x0, x1, wx0, wx1);
  auto stage1 = [&](int Y, int X) {                        // value of the img x img intermediate at (Y, X)
    int a0, a1, c0, c1;
    float u0, u1, v0, v1;
    bil_coord(Y, s1, 256, a0, a1, u0, u1);
    bil_coord(X, s1, 256, c0, c1, v0, v1);
    return __fadd_rn(__fmul_rn(u0, __fadd_rn(__fmul_rn(v0, sam_low(m, a0, c0, nested)), __fmul_rn(v1, sam_low(m, a0, c1, nested)))),
                     __fmul_rn(u1, __fadd_rn(__fmul_rn(v0, sam_low(m, a1, c0, nested)), __fmul_rn(v1, sam_low(m, a1, c1, nested)))));
  };
  const float r = __fadd_rn(__fmul_rn(wy0, __fadd_rn(__fmul_rn(wx0, stage1(y0, x0)), __fmul_rn(wx1, stage1(y0, x1)))),
                            __fmul_rn(wy1, __fadd_rn(__fmul_rn(wx0, stage1(y1, x0)), __fmul_rn(wx1, stage1(y1, x1)))));
  out[((long)b * oh + oy) * ow + ox] = r;
}

extern "C" int llmseg_sam_postprocess(const float* low, float* out, int32_t n_masks, int32_t img_size, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                                      int32_t nested, void* stream) {
  LL_CHECK(low && out && n_masks > 0 && img_size > 0 && in_h > 0 && in_w > 0 && in_h <= img_size && in_w <= img_size && out_h > 0 && out_w > 0,
           "sam_postprocess: bad arguments");
  LL_LAUNCH_KERNEL(sam_postprocess_kernel, dim3((unsigned)((out_w + 255) / 256), (unsigned)out_h, (unsigned)n_masks), dim3(256), 0, (hipStream_t)stream, low, out,
                     img_size, in_h, in_w, out_h, out_w, nested);
  LL_LAUNCH_CHECK("sam_postprocess");
  return LLMSEG_OK;
}

// ---- SAM "everything" mode (automatic_mask_generator.py:264-324), the per-candidate work at the ORIGINAL resolution without ever
// materialising the logits there: one pass that evaluates the fused post-processing per pixel and reduces it to
//   st[c] = { |m > thr + off|, |m > thr - off|, |m > thr|, min x, min y, max x, max y of m > thr }      (int32 x 7)
// (stability score = st0 / st1, amg.py:156-176; box = amg.py:303-346), skipping candidates whose predicted IoU already fails; and a
// second pass that writes the binary masks of the survivors only.
// The fused value is separable in its index arithmetic: every low-resolution address is rowpart(Y) + colpart(X), every weight belongs to a
// row or to a column.  A workgroup owns 256 columns x 64 rows of one candidate: a thread computes its column's coordinates once, the 64
// row records are built by the first 64 threads into LDS and read as broadcasts, so that a pixel costs 16 loads, 16 adds and the 28
// fp32 operations of the two bilinear stages (same operations in the same order as sam_postprocess_kernel: bit-identical values).
struct PostRow { int f[2][2]; float u[2][2]; float w[2]; };      // [y0 / y1][a0 / a1]: row offsets + stage-1 weights; stage-2 weights
struct PostCol { int g[2][2]; float v[2][2]; float w[2]; };      // [x0 / x1][c0 / c1]
__device__ __forceinline__ int low_rowpart(int Y, int nested) { return nested ? (Y >> 2) * 1024 + ((Y >> 1) & 1) * 8 + (Y & 1) * 2 : Y * 256; }
__device__ __forceinline__ int low_colpart(int X, int nested) { return nested ? (X >> 2) * 16 + ((X >> 1) & 1) * 4 + (X & 1) : X; }
__device__ __forceinline__ PostRow post_row(int oy, float s1, float s2y, int in_h, int nested) {
  PostRow r;
  int y[2];
  bil_coord(oy, s2y, in_h, y[0], y[1], r.w[0], r.w[1]);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int a0, a1;
    bil_coord(y[i], s1, 256, a0, a1, r.u[i][0], r.u[i][1]);
    r.f[i][0] = low_rowpart(a0, nested); r.f[i][1] = low_rowpart(a1, nested);
  }
  return r;
}
__device__ __forceinline__ PostCol post_col(int ox, float s1, float s2x, int in_w, int nested) {
  PostCol c;
  int x[2];
  bil_coord(ox, s2x, in_w, x[0], x[1], c.w[0], c.w[1]);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int c0, c1;
    bil_coord(x[i], s1, 256, c0, c1, c.v[i][0], c.v[i][1]);
    c.g[i][0] = low_colpart(c0, nested); c.g[i][1] = low_colpart(c1, nested);
  }
  return c;
}
// The low-resolution samples of a column change only when a row offset changes (every ~4th output row at 256 -> 1024): they are kept in
// registers and reloaded per changed row offset (a wave-uniform decision).  16 gathers per pixel, each touching 16 cache lines per wave,
// made the first form of these kernels bound by the texture addresser (262 CU clocks per 64 pixels).
struct PostCache { int f[2][2]; float L[2][2][2][2]; };          // [y0 / y1][a0 / a1] row offset held, samples [..][..][x0 / x1][c0 / c1]
__device__ __forceinline__ void post_cache_reset(PostCache& k) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int a = 0; a < 2; ++a) k.f[i][a] = -1;
}
__device__ __forceinline__ float post_value(const float* __restrict__ m, const PostRow& r, const PostCol& c, PostCache& k) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int a = 0; a < 2; ++a)
      if (r.f[i][a] != k.f[i][a]) {
        k.f[i][a] = r.f[i][a];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) k.L[i][a][j][cc] = m[r.f[i][a] + c.g[j][cc]];
      }
  float s[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      s[i][j] = __fadd_rn(__fmul_rn(r.u[i][0], __fadd_rn(__fmul_rn(c.v[j][0], k.L[i][0][j][0]), __fmul_rn(c.v[j][1], k.L[i][0][j][1]))),
                          __fmul_rn(r.u[i][1], __fadd_rn(__fmul_rn(c.v[j][0], k.L[i][1][j][0]), __fmul_rn(c.v[j][1], k.L[i][1][j][1]))));
  return __fadd_rn(__fmul_rn(r.w[0], __fadd_rn(__fmul_rn(c.w[0], s[0][0]), __fmul_rn(c.w[1], s[0][1]))),
                   __fmul_rn(r.w[1], __fadd_rn(__fmul_rn(c.w[0], s[1][0]), __fmul_rn(c.w[1], s[1][1]))));
}
constexpr int POST_ROWS = 64;

__global__ __launch_bounds__(256) void sam_mask_stats_kernel(const float* __restrict__ low, const float* __restrict__ iou, float iou_thr, int32_t* __restrict__ st,
                                                            int img, int in_h, int in_w, int oh, int ow, int nested, float thr, float off) {
  const int c = blockIdx.z;
  if (iou && !(iou[c] > iou_thr)) return;                  // predicted-IoU filter first (automatic_mask_generator.py:290-292)
  __shared__ PostRow rows[POST_ROWS];
  const float* m = low + (long)c * 65536;
  const float s1 = 256.f / (float)img, s2y = (float)in_h / (float)oh, s2x = (float)in_w / (float)ow;
  const int oy0 = blockIdx.y * POST_ROWS, nrow = min(POST_ROWS, oh - oy0);
  const int ox = blockIdx.x * 256 + threadIdx.x;
  if ((int)threadIdx.x < nrow) rows[threadIdx.x] = post_row(oy0 + threadIdx.x, s1, s2y, in_h, nested);
  const PostCol pc = post_col(min(ox, ow - 1), s1, s2x, in_w, nested);
  __syncthreads();
  int hi = 0, lo = 0, ar = 0, mny = 1 << 30, mxy = -1;
  if (ox < ow) {
    PostCache pk;
    post_cache_reset(pk);
    for (int r = 0; r < nrow; ++r) {
      const float v = post_value(m, rows[r], pc, pk);
      hi += v > thr + off;
      lo += v > thr - off;
      if (v > thr) { ++ar; mny = min(mny, oy0 + r); mxy = max(mxy, oy0 + r); }
    }
  }
  int mnx = ar ? ox : 1 << 30, mxx = ar ? ox : -1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    hi += __shfl_xor(hi, o, 64); lo += __shfl_xor(lo, o, 64); ar += __shfl_xor(ar, o, 64);
    mnx = min(mnx, __shfl_xor(mnx, o, 64)); mny = min(mny, __shfl_xor(mny, o, 64));
    mxx = max(mxx, __shfl_xor(mxx, o, 64)); mxy = max(mxy, __shfl_xor(mxy, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    int32_t* s = st + (long)c * 7;
    if (hi) atomicAdd(s + 0, hi);
    if (lo) atomicAdd(s + 1, lo);
    if (ar) { atomicAdd(s + 2, ar); atomicMin(s + 3, mnx); atomicMin(s + 4, mny); atomicMax(s + 5, mxx); atomicMax(s + 6, mxy); }
  }
}

// binary masks (uint8 0 / 1) of the selected candidates at the original resolution
__global__ __launch_bounds__(256) void sam_binarize_kernel(const float* __restrict__ low, const int32_t* __restrict__ sel, uint8_t* __restrict__ out, int img, int in_h,
                                                          int in_w, int oh, int ow, int nested, float thr) {
  __shared__ PostRow rows[POST_ROWS];
  const int k = blockIdx.z;
  const float* m = low + (long)sel[k] * 65536;
  const float s1 = 256.f / (float)img, s2y = (float)in_h / (float)oh, s2x = (float)in_w / (float)ow;
  const int oy0 = blockIdx.y * POST_ROWS, nrow = min(POST_ROWS, oh - oy0);
  const int ox = blockIdx.x * 256 + threadIdx.x;
  if ((int)threadIdx.x < nrow) rows[threadIdx.x] = post_row(oy0 + threadIdx.x, s1, s2y, in_h, nested);
  const PostCol pc = post_col(min(ox, ow - 1), s1, s2x, in_w, nested);
  __syncthreads();
  if (ox >= ow) return;
  uint8_t* o = out + ((long)k * oh + oy0) * ow + ox;
  PostCache pk;
  post_cache_reset(pk);
  for (int r = 0; r < nrow; ++r) o[(long)r * ow] = post_value(m, rows[r], pc, pk) > thr ? 1 : 0;
}

// Greedy box NMS (torchvision.ops.nms semantics, one category): `order` = candidate indices by decreasing score; keep[i] = 1 if box
// order[i] survives: a kept box suppresses every later box with IoU > thr, areas (x2 - x1)(y2 - y1).  One workgroup; n <= 8192.
__global__ __launch_bounds__(1024) void nms_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ order, int n, float thr, uint8_t* __restrict__ keep) {
  extern __shared__ uint8_t dead[];
  for (int i = threadIdx.x; i < n; i += blockDim.x) dead[i] = 0;
  __syncthreads();
  for (int i = 0; i < n; ++i) {
    if (!dead[i]) {                                        // uniform: read after the barrier of the previous iteration
      const float* a = boxes + 4 * order[i];
      const float ax1 = a[0], ay1 = a[1], ax2 = a[2], ay2 = a[3], aa = (ax2 - ax1) * (ay2 - ay1);
      for (int j = i + 1 + threadIdx.x; j < n; j += blockDim.x) {
        if (dead[j]) continue;
        const float* b = boxes + 4 * order[j];
        const float iw = fmaxf(fminf(ax2, b[2]) - fmaxf(ax1, b[0]), 0.f), ih = fmaxf(fminf(ay2, b[3]) - fmaxf(ay1, b[1]), 0.f);
        const float inter = iw * ih;
        if (inter / (aa + (b[2] - b[0]) * (b[3] - b[1]) - inter) > thr) dead[j] = 1;
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) keep[i] = dead[i] ? 0 : 1;
}

extern "C" int llmseg_sam_mask_stats(const float* low, const float* iou, float iou_thresh, int32_t* stats, int32_t n_masks, int32_t img_size, int32_t in_h,
                                     int32_t in_w, int32_t out_h, int32_t out_w, int32_t nested, float mask_threshold, float offset, void* stream) {
  LL_CHECK(low && stats && n_masks > 0 && img_size > 0 && in_h > 0 && in_w > 0 && in_h <= img_size && in_w <= img_size && out_h > 0 && out_w > 0,
           "sam_mask_stats: bad arguments");
  LL_CHECK((out_h + POST_ROWS - 1) / POST_ROWS < 65536 && n_masks < 65536, "sam_mask_stats: grid limit");
  LL_LAUNCH_KERNEL(sam_mask_stats_kernel, dim3((unsigned)((out_w + 255) / 256), (unsigned)((out_h + POST_ROWS - 1) / POST_ROWS), (unsigned)n_masks), dim3(256), 0, (hipStream_t)stream, low, iou, iou_thresh, stats, img_size, in_h, in_w, out_h,
                     out_w, nested, mask_threshold, offset);
  LL_LAUNCH_CHECK("sam_mask_stats");
  return LLMSEG_OK;
}

extern "C" int llmseg_sam_binarize(const float* low, const int32_t* sel, uint8_t* out, int32_t n_sel, int32_t img_size, int32_t in_h, int32_t in_w, int32_t out_h,
                                   int32_t out_w, int32_t nested, float mask_threshold, void* stream) {
  LL_CHECK(low && sel && out && n_sel > 0 && img_size > 0 && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0, "sam_binarize: bad arguments");
  LL_CHECK(n_sel < 65536, "sam_binarize: grid limit");
  LL_LAUNCH_KERNEL(sam_binarize_kernel, dim3((unsigned)((out_w + 255) / 256), (unsigned)((out_h + POST_ROWS - 1) / POST_ROWS), (unsigned)n_sel), dim3(256), 0, (hipStream_t)stream, low, sel, out,
                     img_size, in_h, in_w, out_h, out_w, nested, mask_threshold);
  LL_LAUNCH_CHECK("sam_binarize");
  return LLMSEG_OK;
}

extern "C" int llmseg_nms(const float* boxes, const int32_t* order, int32_t n, float iou_threshold, uint8_t* keep, void* stream) {
  LL_CHECK(boxes && order && keep && n > 0 && n <= 8192, "nms: 1 <= n <= 8192 boxes");
  LL_LAUNCH_KERNEL(nms_kernel, dim3(1), dim3(1024), (size_t)n, (hipStream_t)stream, boxes, order, n, iou_threshold, keep);
  LL_LAUNCH_CHECK("nms");
  return LLMSEG_OK;
}

extern "C" int llmseg_mask_pullback(const void* segs, void* ws, float* pulled_back, float* wsum, int32_t K, int32_t g, int32_t S, void* stream) {
  LL_CHECK(segs && ws && K > 0 && g > 0 && S >= g, "mask_pullback: bad arguments");
  LL_CHECK(256 % g == 0 && ((g * g) & 7) == 0, "mask_pullback: feature grid %d must divide 256", g);
  if (S == 256 && g == 64 && ((((uintptr_t)segs) | ((uintptr_t)ws)) & 15) == 0 && (!pulled_back || (((uintptr_t)pulled_back) & 15) == 0)) {
    LL_LAUNCH_KERNEL(mask_pullback_s256_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)segs, (bf16_t*)ws, pulled_back, wsum);
    LL_LAUNCH_CHECK("mask_pullback");
    return LLMSEG_OK;
  }
  const size_t lds = ((size_t)((S + 1) / 2) * g + (size_t)g * g + 16) * sizeof(float);
  LL_CHECK(lds <= 64 * 1024, "mask_pullback: S=%d g=%d need %zu bytes of LDS", S, g, lds);
  LL_LAUNCH_KERNEL(mask_pullback_kernel, dim3(K), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)segs, (bf16_t*)ws, pulled_back, wsum, g, S);
  LL_LAUNCH_CHECK("mask_pullback");
  return LLMSEG_OK;
}

extern "C" int llmseg_upsample_maskpool(const void* feat, const void* segs, void* pooled, void* ws, float* pulled_back, float* wsum, int32_t K,
                                        int32_t C, int32_t g, int32_t S, void* stream) {
  LL_CHECK(feat && pooled && C > 0 && (C & 7) == 0, "upsample_maskpool: bad arguments");
  const int rc = llmseg_mask_pullback(segs, ws, pulled_back, wsum, K, g, S, stream);
  if (rc != LLMSEG_OK) return rc;
  // pooled[K][C] = wn[K][g*g] . feat[g*g][C]
  llmseg_gemm_args ga = {};
  ga.struct_size = sizeof(ga);
  ga.A = ws; ga.W = feat; ga.C = pooled;
  ga.M = K; ga.N = C; ga.K = (int64_t)g * g;
  ga.lda = (int64_t)g * g; ga.ldw = C; ga.ldc = C;
  ga.batch = 1; ga.alpha = 1.f; ga.act = LLMSEG_ACT_NONE; ga.trans_w = 1;
  return llmseg_gemm_bf16(&ga, stream);
}

extern "C" int llmseg_cosine_scores(const void* t, const void* e, float* sim, int32_t K, int32_t D, void* stream) {
  LL_CHECK(t && e && sim && K > 0 && D > 0, "cosine_scores: bad arguments");
  LL_LAUNCH_KERNEL(cosine_kernel, dim3((K + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)t, (const bf16_t*)e, sim, K, D);
  LL_LAUNCH_CHECK("cosine_scores");
  return LLMSEG_OK;
}

extern "C" int llmseg_align_reg_loss(const void* e, const void* t, const float* gt_iou, const void* pred_iou, const float* gt_iop, float* out,
                                     float* d_e, float* d_t, float* d_pred, int32_t K, int32_t D, float tau, int32_t items, void* stream) {
  LL_CHECK(e && t && gt_iou && pred_iou && gt_iop && out && K > 0 && D > 0 && tau > 0.f && items > 0, "align_reg_loss: bad arguments");
  const size_t lds = ((size_t)3 * K + 16 + 4 * 64) * sizeof(float);
  LL_CHECK(lds <= 64 * 1024, "align_reg_loss: K=%d too large", K);
  const unsigned ny = (d_e || d_t) ? (unsigned)std::max(1, std::min(8, D / 64)) : 1u;      // column chunks of an item's gradient pass
  LL_LAUNCH_KERNEL(align_reg_kernel, dim3(items, ny), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)e, (const bf16_t*)t, gt_iou,
                     (const bf16_t*)pred_iou, gt_iop, out, d_e, d_t, d_pred, K, D, tau);
  LL_LAUNCH_CHECK("align_reg_loss");
  return LLMSEG_OK;
}

extern "C" int llmseg_dice_bce(const float* logits, const float* targets, float* out, int32_t M, int64_t HW, float num_masks, void* workspace,
                               int64_t workspace_bytes, void* stream) {
  LL_CHECK(logits && targets && out && M > 0 && HW > 0, "dice_bce: bad arguments");
  LL_CHECK(workspace && workspace_bytes >= (int64_t)M * 8, "dice_bce: workspace of >= 8 M bytes (per-mask terms) is required");
  LL_LAUNCH_KERNEL(dice_bce_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, targets, (float*)workspace, (long)HW, num_masks);
  LL_LAUNCH_KERNEL(fold_column_kernel, dim3(2), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (long)M, out, 1.f);
  LL_LAUNCH_CHECK("dice_bce");
  return LLMSEG_OK;
}

extern "C" int llmseg_dice_bce_bwd(const float* logits, const float* targets, const float* g, float* dlogits, int32_t M, int64_t HW, float num_masks,
                                   void* stream) {
  LL_CHECK(logits && targets && g && dlogits && M > 0 && HW > 0, "dice_bce_bwd: bad arguments");
  LL_LAUNCH_KERNEL(dice_bce_bwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, targets, g, dlogits, (long)HW, num_masks);
  LL_LAUNCH_CHECK("dice_bce_bwd");
  return LLMSEG_OK;
}

extern "C" int llmseg_ce_loss(const void* logits, const int64_t* labels, float* acc, int32_t N, int32_t T, int64_t V, int64_t ldl, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  LL_CHECK(logits && labels && acc && N > 0 && T > 1 && V > 0 && ldl >= V, "ce_loss: bad arguments");
  const long rows = (long)N * (T - 1);
  LL_CHECK(workspace && workspace_bytes >= rows * 8, "ce_loss: workspace of >= 8 N (T - 1) bytes (per-position terms) is required");
  LL_LAUNCH_KERNEL(ce_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, labels, (float*)workspace, T, (long)V,
                     (long)ldl);
  LL_LAUNCH_KERNEL(fold_column_kernel, dim3(2), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, rows, acc, 1.f);
  LL_LAUNCH_CHECK("ce_loss");
  return LLMSEG_OK;
}

// ---- gIoU / cIoU metric: 2-class intersection / union histograms (reference utils/utils.py:119-132) -------------------------
// pred/target uint8 [n]; target == ignore_index is excluded.  out int64[6] += {I0, I1, U0, U1, T0, T1}.
namespace {
__global__ __launch_bounds__(256) void inter_union_kernel(const uint8_t* __restrict__ pred, const uint8_t* __restrict__ tgt, long n, int ignore,
                                                         unsigned long long* __restrict__ out) {
  __shared__ unsigned long long sh[6];
  if (threadIdx.x < 6) sh[threadIdx.x] = 0;
  __syncthreads();
  unsigned int c[6] = {0, 0, 0, 0, 0, 0};    // I0 I1 P0 P1 T0 T1
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int t = tgt[i];
    if (t == ignore) continue;
    const int p = pred[i];
    if (p < 2) { c[2 + p]++; if (p == t) c[p]++; }
    if (t < 2) c[4 + t]++;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    unsigned int v = c[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sh[k], (unsigned long long)v);
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    atomicAdd(&out[threadIdx.x], sh[threadIdx.x]);                                                     // intersection
    atomicAdd(&out[2 + threadIdx.x], sh[2 + threadIdx.x] + sh[4 + threadIdx.x] - sh[threadIdx.x]);     // union = P + T - I
    atomicAdd(&out[4 + threadIdx.x], sh[4 + threadIdx.x]);                                             // target area
  }
}
}  // namespace

extern "C" int llmseg_intersection_union(const uint8_t* pred, const uint8_t* target, int64_t n, int32_t ignore_index, int64_t* out, void* stream) {
  LL_CHECK(pred && target && out && n > 0, "intersection_union: bad arguments");
  long g = (n + 256 * 16 - 1) / (256 * 16);
  LL_LAUNCH_KERNEL(inter_union_kernel, dim3((unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g))), dim3(256), 0, (hipStream_t)stream, pred, target, (long)n,
                     ignore_index, (unsigned long long*)out);
  LL_LAUNCH_CHECK("intersection_union");
  return LLMSEG_OK;
}

// ---- validate_threshold inner loop (reference training.py:712-766) fused: union of the selected proposals at original
// resolution -> nearest resize of prediction and ground truth to out_h x out_w -> 2-class I/U with ignore 255 (the arg-max variant,
// training.py:605-687, selects ONE proposal and scores at the ground truth's own resolution: out = Hg x Wg).
// segs uint8 [H][W][K] (the reader's (H, W, K) layout), select uint8 [K] (pred_iou > threshold), gt uint8 [Hg][Wg].
namespace {
__global__ __launch_bounds__(256) void union_resize_iou_kernel(const uint8_t* __restrict__ segs, const uint8_t* __restrict__ select, const uint8_t* __restrict__ gt,
                                                              int H, int W, int K, int Hg, int Wg, int outh, int outw, int ignore,
                                                              unsigned long long* __restrict__ res) {
  __shared__ unsigned long long sh[6];
  extern __shared__ uint8_t sel[];
  if (threadIdx.x < 6) sh[threadIdx.x] = 0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) sel[k] = select[k];
  __syncthreads();
  unsigned int c[6] = {0, 0, 0, 0, 0, 0};
  const long n = (long)outh * outw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / outw), x = (int)(i % outw);
    // F.interpolate(mode="nearest"): src = min(floor(dst * in / out), in - 1)
    const int sy = min((int)floorf((float)y * ((float)H / (float)outh)), H - 1), sx = min((int)floorf((float)x * ((float)W / (float)outw)), W - 1);
    const int gy = min((int)floorf((float)y * ((float)Hg / (float)outh)), Hg - 1), gx = min((int)floorf((float)x * ((float)Wg / (float)outw)), Wg - 1);
    const int t = gt[(long)gy * Wg + gx];
    if (t == ignore) continue;
    const uint8_t* px = segs + ((long)sy * W + sx) * K;
    int p = 0;
    for (int k = 0; k < K; ++k) p |= (sel[k] & (px[k] != 0));
    c[2 + p]++;
    if (p == t) c[p]++;
    if (t < 2) c[4 + t]++;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    unsigned int v = c[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sh[k], (unsigned long long)v);
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    atomicAdd(&res[threadIdx.x], sh[threadIdx.x]);
    atomicAdd(&res[2 + threadIdx.x], sh[2 + threadIdx.x] + sh[4 + threadIdx.x] - sh[threadIdx.x]);
    atomicAdd(&res[4 + threadIdx.x], sh[4 + threadIdx.x]);
  }
}
}  // namespace

extern "C" int llmseg_union_resize_iou(const uint8_t* segs, const uint8_t* select, const uint8_t* gt, int32_t H, int32_t W, int32_t K, int32_t Hg,
                                       int32_t Wg, int32_t out_h, int32_t out_w, int32_t ignore_index, int64_t* out, void* stream) {
  LL_CHECK(segs && select && gt && out && H > 0 && W > 0 && K > 0 && Hg > 0 && Wg > 0 && out_h > 0 && out_w > 0, "union_resize_iou: bad arguments");
  LL_LAUNCH_KERNEL(union_resize_iou_kernel, dim3(1024), dim3(256), (size_t)K, (hipStream_t)stream, segs, select, gt, H, W, K, Hg, Wg, out_h, out_w,
                     ignore_index, (unsigned long long*)out);
  LL_LAUNCH_CHECK("union_resize_iou");
  return LLMSEG_OK;
}
