// Backward-pass and optimizer kernels for the trainable part of the LLM-Seg hot path (LoRA'd Llama stack, embed/lm_head,
// text_hidden_fcs, mask-selection head).  All HBM-bound streaming kernels; GEMM-shaped gradients go through
// llmseg_gemm_bf16 with trans_a / trans_w.  See include/llmseg_hip.h for the reference ops each one differentiates.
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "llmseg_hip.h"

namespace {

inline unsigned grid_for(long total) {
  long g = (total + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// out[n] += sum_m x[m][n]   (bias gradients); block = 64 columns x 4 row-lanes, grid.y splits the rows.  part != NULL: the row slice's
// sum goes to part[slice][N] (folded in slice order by fold_slices_kernel); part == NULL (one slice): the column's only writer adds it.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, float* __restrict__ part, long M, long N, long ld) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long col = (long)blockIdx.x * 64 + c;
  const long rows_per = (M + gridDim.y - 1) / gridDim.y;
  const long r0 = (long)blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float s = 0.f;
  if (col < N)
    for (long r = r0 + rl; r < r1; r += 4) s += bf2f(x[r * ld + col]);
  red[rl][c] = s;
  __syncthreads();
  if (rl == 0 && col < N) {
    const float v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    if (part) part[(long)blockIdx.y * N + col] = v;
    else out[col] += v;
  }
}

// LayerNorm / RMSNorm backward, one wave per row.  CPL > 0: x and dy (<= 64*CPL 16-byte chunks per row) are read once and kept
// in registers; CPL == 0: multi-pass fallback.  Weight / bias gradients (fp32, dw may be NULL), no atomics (fixed summation order):
// ACC = true (narrow rows, CPL <= 2) walks rows grid-stride with per-lane register partial sums, folds the workgroup's 4 waves in LDS
// and writes ONE partial per column per workgroup to part[workgroup][2][cols] (fold_slices_kernel adds them in workgroup order; a
// single workgroup adds to dw / db itself); ACC = false (wide rows) stores the row's (mean, rstd) to `stats` and leaves the column
// sums to norm_dwdb_kernel.
template <int CPL, bool ACC>
__global__ __launch_bounds__(256, 2) void norm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                         bf16_t* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, long rows, int cols,
                                                         float eps, int rms, const bf16_t* __restrict__ dres, float* __restrict__ part,
                                                         float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int nch = cols >> 3;
  constexpr int NR = CPL > 0 ? CPL : 1;
  float aw[ACC ? NR : 1][8], ab[ACC ? NR : 1][8];
  if constexpr (ACC) {
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) { aw[i][e] = 0.f; ab[i][e] = 0.f; }
  }
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
    const bf16_t* xr = x + row * cols;
    const bf16_t* gr = dy + row * cols;
    uint4 xc[NR], gc[NR];
    if constexpr (CPL > 0) {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int c = lane + 64 * i;
        const bool on = c < nch;
        xc[i] = on ? *reinterpret_cast<const uint4*>(xr + c * 8) : make_uint4(0, 0, 0, 0);
        gc[i] = on ? *reinterpret_cast<const uint4*>(gr + c * 8) : make_uint4(0, 0, 0, 0);
      }
    }
    // the body sees chunk index c, cache slot i and the cached / freshly loaded 16-byte pieces xv (x) and gv (dy)
#define LL_FOR_CHUNKS(...)                                                                       \
  if constexpr (CPL > 0) {                                                                       \
    _Pragma("unroll") for (int i = 0; i < NR; ++i) {                                             \
      const int c = lane + 64 * i;                                                               \
      if (c < nch) { const uint4 xv = xc[i], gv = gc[i]; __VA_ARGS__ }                           \
    }                                                                                            \
  } else {                                                                                       \
    for (int c = lane; c < nch; c += 64) {                                                       \
      constexpr int i = 0;                                                                       \
      const uint4 xv = *reinterpret_cast<const uint4*>(xr + c * 8), gv = *reinterpret_cast<const uint4*>(gr + c * 8); \
      __VA_ARGS__                                                                                \
    }                                                                                            \
  }
    float f[8], g[8], ww[8];
    float s = 0.f;
    if (!rms) {
      LL_FOR_CHUNKS({ (void)gv; (void)i; unpack8(xv, f); _Pragma("unroll") for (int e = 0; e < 8; ++e) s += f[e]; })
      s = wave_sum(s);
    }
    const float mean = rms ? 0.f : s / (float)cols;
    float v = 0.f;
    LL_FOR_CHUNKS({ (void)gv; (void)i; unpack8(xv, f); _Pragma("unroll") for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; v += d * d; } })
    v = wave_sum(v);
    const float rstd = rsqrtf(v / (float)cols + eps);
    if constexpr (!ACC) { if (stats && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; } }
    // c1 = mean(g), c2 = mean(g * xhat) with g = dy * w
    float c1 = 0.f, c2 = 0.f;
    LL_FOR_CHUNKS({
      (void)i;
      unpack8(xv, f); unpack8(gv, g);
      unpack8(*reinterpret_cast<const uint4*>(w + c * 8), ww);
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {
        const float xh = (f[e] - mean) * rstd, gg = g[e] * ww[e];
        c1 += gg; c2 += gg * xh;
      }
    })
    c1 = wave_sum(c1) / (float)cols; c2 = wave_sum(c2) / (float)cols;
    if (rms) c1 = 0.f;
    LL_FOR_CHUNKS({
      float o[8];
      unpack8(xv, f); unpack8(gv, g);
      unpack8(*reinterpret_cast<const uint4*>(w + c * 8), ww);
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {
        const float xh = (f[e] - mean) * rstd;
        o[e] = rstd * (g[e] * ww[e] - c1 - xh * c2);
        if constexpr (ACC) { aw[i][e] += g[e] * xh; ab[i][e] += g[e]; }
      }
      if (dres) {                                            // + the gradient that reaches x through the residual connection
        float rr[8];
        unpack8(*reinterpret_cast<const uint4*>(dres + row * cols + c * 8), rr);
        _Pragma("unroll") for (int e = 0; e < 8; ++e) o[e] += rr[e];
      }
      *reinterpret_cast<uint4*>(dx + row * cols + c * 8) = pack8(o);
    })
#undef LL_FOR_CHUNKS
  }
  if constexpr (ACC) {
    // fold the workgroup's 4 waves in LDS, then ONE partial per column per workgroup
    __shared__ float red[2][4][NR * 64 * 8];
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[0][wv][(i * 64 + lane) * 8 + e] = aw[i][e];
        red[1][wv][(i * 64 + lane) * 8 + e] = ab[i][e];
      }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NR * 64 * 8; idx += blockDim.x) {
      const int i = idx / (64 * 8), ln = (idx / 8) % 64, e = idx % 8;
      const int c = ln + 64 * i;
      if (c < nch) {
        const float sw = (red[0][0][idx] + red[0][1][idx]) + (red[0][2][idx] + red[0][3][idx]);
        const float sb = (red[1][0][idx] + red[1][1][idx]) + (red[1][2][idx] + red[1][3][idx]);
        if (part) {
          part[((long)blockIdx.x * 2 + 0) * cols + c * 8 + e] = sw;
          part[((long)blockIdx.x * 2 + 1) * cols + c * 8 + e] = sb;
        } else {
          if (dw) dw[c * 8 + e] += sw;
          if (db) db[c * 8 + e] += sb;
        }
      }
    }
  }
}

// Norm backward for a FROZEN weight (no dw / db) with a WORKGROUP per row: short, wide activations (Llama at two images per step: 638 rows x
// 4096 columns) leave most of the chip idle with one wave per row (14 us for 21 MB); 4 waves per row keep 4 x the loads in flight.
// the row arithmetic of norm_bwd_wg_kernel, shared with the fused split-K tail (reduce_lora_normbwd_kernel) so that both produce the same bits:
// xc / gc / wc = this thread's 8-element chunks (c = thread + 256 i) of x, dy and the norm weight; writes dx = norm backward + dres
template <int CPT>
__device__ __forceinline__ void normbwd_row(const uint4 (&xc)[CPT], const uint4 (&gc)[CPT], const uint4 (&wc)[CPT], float* red, bf16_t* __restrict__ dx, long row,
                                            int cols, float eps, int rms, const bf16_t* __restrict__ dres) {
  const int nch = cols >> 3;
  float f[8], g[8], ww[8];
  float s = 0.f;
  if (!rms) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) { unpack8(xc[i], f); _Pragma("unroll") for (int e = 0; e < 8; ++e) s += f[e]; }
    s = block_sum(s, red);
  }
  const float mean = rms ? 0.f : s / (float)cols;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i)
    if (threadIdx.x + 256 * i < nch) { unpack8(xc[i], f); _Pragma("unroll") for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; v += d * d; } }
  v = block_sum(v, red);
  const float rstd = rsqrtf(v / (float)cols + eps);
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i)
    if (threadIdx.x + 256 * i < nch) {
      unpack8(xc[i], f); unpack8(gc[i], g); unpack8(wc[i], ww);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float xh = (f[e] - mean) * rstd, gg = g[e] * ww[e]; c1 += gg; c2 += gg * xh; }
    }
  c1 = rms ? 0.f : block_sum(c1, red) / (float)cols;
  c2 = block_sum(c2, red) / (float)cols;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < nch) {
      float o[8];
      unpack8(xc[i], f); unpack8(gc[i], g); unpack8(wc[i], ww);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float xh = (f[e] - mean) * rstd; o[e] = rstd * (g[e] * ww[e] - c1 - xh * c2); }
      if (dres) {
        float rr[8];
        unpack8(*reinterpret_cast<const uint4*>(dres + row * cols + c * 8), rr);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += rr[e];
      }
      *reinterpret_cast<uint4*>(dx + row * cols + c * 8) = pack8(o);
    }
  }
}

template <int CPT>
__global__ __launch_bounds__(256) void norm_bwd_wg_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                         bf16_t* __restrict__ dx, long rows, int cols, float eps, int rms, const bf16_t* __restrict__ dres) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const int nch = cols >> 3;
  const bf16_t* xr = x + row * cols;
  const bf16_t* gr = dy + row * cols;
  uint4 xc[CPT], gc[CPT], wc[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = threadIdx.x + 256 * i;
    const bool on = c < nch;
    xc[i] = on ? *reinterpret_cast<const uint4*>(xr + c * 8) : make_uint4(0, 0, 0, 0);
    gc[i] = on ? *reinterpret_cast<const uint4*>(gr + c * 8) : make_uint4(0, 0, 0, 0);
    wc[i] = on ? *reinterpret_cast<const uint4*>(w + c * 8) : make_uint4(0, 0, 0, 0);
  }
  normbwd_row<CPT>(xc, gc, wc, red, dx, row, cols, eps, rms, dres);
}

// Column sums of a wide norm's backward from the stored row statistics: dw[c] = sum_r dy[r][c] (x[r][c] - mean_r) rstd_r, db[c] = sum_r dy[r][c].
// Block = 64 columns x 4 row-lanes, grid.y = row slices -> part[slice][2][cols] (slices folded in order), or straight into dw / db for one slice.
__global__ __launch_bounds__(256) void norm_dwdb_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ stats,
                                                       float* __restrict__ dw, float* __restrict__ db, float* __restrict__ part, long rows, long cols) {
  __shared__ float red[2][4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long col = (long)blockIdx.x * 64 + c;
  const long rows_per = (rows + gridDim.y - 1) / gridDim.y;
  const long r0 = (long)blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float sw = 0.f, sb = 0.f;
  if (col < cols)
    for (long r = r0 + rl; r < r1; r += 4) {
      const float g = bf2f(dy[r * cols + col]);
      sw += g * (bf2f(x[r * cols + col]) - stats[2 * r]) * stats[2 * r + 1];
      sb += g;
    }
  red[0][rl][c] = sw; red[1][rl][c] = sb;
  __syncthreads();
  if (rl == 0 && col < cols) {
    const float vw = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
    const float vb = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    if (part) { part[((long)blockIdx.y * 2 + 0) * cols + col] = vw; part[((long)blockIdx.y * 2 + 1) * cols + col] = vb; }
    else { if (dw) dw[col] += vw; if (db) db[col] += vb; }
  }
}

// d_gu[r][c] = d_out * up * silu'(gate), d_gu[r][I+c] = d_out * silu(gate)
// dout rows have stride ld_dout; dout may ALIAS the up half of dgu's rows (the two-launch route of the GEMM's fused swiglu-backward epilogue): every
// thread reads its own 8 elements of dout before it overwrites them, so no restrict on those two
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ gu, const bf16_t* dout, bf16_t* dgu,
                                                        long rows, long I, long ld_dout) {
  const long ich = I >> 3, total = rows * ich;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / ich, c = i % ich;
    float g[8], u[8], d[8], og[8], ou[8];
    unpack8(*reinterpret_cast<const uint4*>(gu + row * 2 * I + c * 8), g);
    unpack8(*reinterpret_cast<const uint4*>(gu + row * 2 * I + I + c * 8), u);
    unpack8(*reinterpret_cast<const uint4*>(dout + row * ld_dout + c * 8), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) swiglu_bwd1(d[e], g[e], u[e], og[e], ou[e]);
    *reinterpret_cast<uint4*>(dgu + row * 2 * I + c * 8) = pack8(og);
    *reinterpret_cast<uint4*>(dgu + row * 2 * I + I + c * 8) = pack8(ou);
  }
}

// dpre = dy * f'(y) from the OUTPUT y: relu -> (y > 0), sigmoid -> y (1 - y)
__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y, bf16_t* __restrict__ out, long n, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float g = bf2f(dy[i]), yy = bf2f(y[i]);
    out[i] = f2bf(act == LLMSEG_ACT_RELU ? (yy > 0.f ? g : 0.f) : g * yy * (1.f - yy));
  }
}

// P[b][q][k] = softmax_k(scale * S[b][q][k] + mask), one wave per (b, q) row; columns >= T are written as zero
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, bf16_t* __restrict__ P, long BH, int Tq, int T, int ld,
                                                          float scale, int causal, const uint8_t* __restrict__ key_mask, int heads) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= BH * Tq) return;
  const long bh = row / Tq;
  const int q = (int)(row % Tq);
  const float* s = S + row * ld;
  bf16_t* p = P + row * ld;
  const uint8_t* km = key_mask ? key_mask + (bh / heads) * T : nullptr;
  float mx = -1e30f;
  for (int k = lane; k < T; k += 64) {
    const bool ok = (!causal || k <= q) && (!km || km[k]);
    if (ok) mx = fmaxf(mx, s[k] * scale);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int k = lane; k < T; k += 64) {
    const bool ok = (!causal || k <= q) && (!km || km[k]);
    if (ok) sum += __expf(s[k] * scale - mx);
  }
  sum = wave_sum(sum);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  for (int k = lane; k < ld; k += 64) {
    const bool ok = k < T && (!causal || k <= q) && (!km || km[k]);
    p[k] = f2bf(ok ? __expf(s[k] * scale - mx) * inv : 0.f);
  }
}

// dS = scale * P * (dP - sum_k P dP), one wave per row; pad columns written as zero
__global__ __launch_bounds__(256) void attn_ds_kernel(const bf16_t* __restrict__ P, const float* __restrict__ dP, bf16_t* __restrict__ dS, long rows, int T,
                                                     int ld, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* p = P + row * ld;
  const float* d = dP + row * ld;
  float dl = 0.f;
  for (int k = lane; k < T; k += 64) dl += bf2f(p[k]) * d[k];
  dl = wave_sum(dl);
  for (int k = lane; k < ld; k += 64) dS[row * ld + k] = f2bf(k < T ? scale * bf2f(p[k]) * (d[k] - dl) : 0.f);
}

// dlogits[n][t][:] = coef * (softmax(logits[n][t]) - onehot(labels[n][t+1])), zero rows where the label is ignored / t == T-1
__global__ __launch_bounds__(256) void ce_bwd_kernel(const bf16_t* __restrict__ logits, const int64_t* __restrict__ labels, const float* __restrict__ coef,
                                                    bf16_t* __restrict__ dl, int T, long V, long ldl) {
  __shared__ float red[16];
  const int n = blockIdx.x / T, t = blockIdx.x % T;
  const bf16_t* row = logits + ((long)n * T + t) * ldl;
  bf16_t* out = dl + ((long)n * T + t) * ldl;
  const long lab = t + 1 < T ? labels[(long)n * T + t + 1] : -100;
  if (lab < 0 || lab >= V) {
    for (long i = threadIdx.x; i < V; i += blockDim.x) out[i] = 0;
    return;
  }
  if (ce_row_fast(row, V, ldl) && (((uintptr_t)out) & 7) == 0) {      // one read of the row (registers), one 8-byte write per quad
    CeRow r;
    r.load(row, V);
    float mx, s;
    r.stats(red, mx, s);
    const float c = coef[0], inv = 1.f / s;
    const long nq = V >> 2;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const long k = threadIdx.x + 256L * i;
      if (k < nq) {
        const long e0 = 4 * k;
        const float g0 = c * (__expf(__uint_as_float(r.q[i].x << 16) - mx) * inv - (e0 == lab ? 1.f : 0.f));
        const float g1 = c * (__expf(__uint_as_float(r.q[i].x & 0xffff0000u) - mx) * inv - (e0 + 1 == lab ? 1.f : 0.f));
        const float g2 = c * (__expf(__uint_as_float(r.q[i].y << 16) - mx) * inv - (e0 + 2 == lab ? 1.f : 0.f));
        const float g3 = c * (__expf(__uint_as_float(r.q[i].y & 0xffff0000u) - mx) * inv - (e0 + 3 == lab ? 1.f : 0.f));
        *reinterpret_cast<uint2*>(out + e0) = make_uint2(pack2bf(g0, g1), pack2bf(g2, g3));
      }
    }
    return;
  }
  float mx = -1e30f;
  for (long i = threadIdx.x; i < V; i += blockDim.x) mx = fmaxf(mx, bf2f(row[i]));
  mx = block_max(mx, red);
  float s = 0.f;
  for (long i = threadIdx.x; i < V; i += blockDim.x) s += __expf(bf2f(row[i]) - mx);
  s = block_sum(s, red);
  const float c = coef[0], inv = 1.f / s;
  for (long i = threadIdx.x; i < V; i += blockDim.x) out[i] = f2bf(c * (__expf(bf2f(row[i]) - mx) * inv - (i == lab ? 1.f : 0.f)));
}

// dst[idx[i]][:] += src[i][:]  (fp32 accumulation; embedding / gather gradients).  idx < 0 rows are skipped.
// No atomics: workgroup i handles source row i and is the OWNER of destination row d = idx[i] iff no earlier source row has the same
// destination (a scan of idx[0 .. i), 256 entries per step); the owner walks idx[i ..) in chunks of 1024, collects the source rows that
// hit d IN ASCENDING ORDER (one ballot per wave, prefix over the 4 waves) and adds them column by column -- the sum of a destination
// row is formed by one workgroup in source-row order, so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const bf16_t* __restrict__ src, const int64_t* __restrict__ idx, float* __restrict__ dst,
                                                              long n, long cols) {
  __shared__ int s_list[1024];
  __shared__ int s_cnt[4];
  const long i = blockIdx.x;
  const long d = idx[i];
  if (d < 0) return;
  for (long j0 = 0; j0 < i; j0 += 256) {
    const long j = j0 + threadIdx.x;
    if (__syncthreads_or(j < i && idx[j] == d)) return;          // an earlier source row owns this destination
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long base = i; base < n; base += 1024) {
    // matches of this chunk, ascending: wave w scans rows base + 256 w + lane + 64 q (q = 0..3) -> per-thread 4 candidates out of order;
    // simpler and ordered: four sub-steps of 256 consecutive rows each
    int total = 0;
    for (int q = 0; q < 4; ++q) {
      const long j = base + 256 * q + threadIdx.x;
      const bool hit = j < n && idx[j] == d;
      const unsigned long long bal = __ballot(hit);
      if (lane == 0) s_cnt[wave] = __popcll(bal);
      __syncthreads();
      int off = total;
      for (int w2 = 0; w2 < wave; ++w2) off += s_cnt[w2];
      if (hit) s_list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (int)(j - base);
      total += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
      __syncthreads();
    }
    if (total == 0) continue;
    for (long c = threadIdx.x; c < cols; c += 256) {
      float sacc = 0.f;
      for (int t = 0; t < total; ++t) sacc += bf2f(src[(base + s_list[t]) * cols + c]);
      dst[d * cols + c] += sacc;
    }
    __syncthreads();
  }
}

// sum of squares of a bf16 or fp32 buffer: one partial per workgroup -> part[blockIdx.x] (fold_column_kernel adds them in a fixed order)
__global__ __launch_bounds__(256) void sumsq_kernel(const void* __restrict__ x, long n, int is_f32, float* __restrict__ part) {
  __shared__ float red[16];
  float s = 0.f;
  if (is_f32 && (((uintptr_t)x) & 15) == 0) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
      const float4 v = x4[i];
      s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0)
      for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) { const float v = reinterpret_cast<const float*>(x)[i]; s += v * v; }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
      const float v = is_f32 ? reinterpret_cast<const float*>(x)[i] : bf2f(reinterpret_cast<const bf16_t*>(x)[i]);
      s += v * v;
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// Fused AdamW on fp32 master weights with bf16 model copy (DeepSpeed bf16 + AdamW(beta=(0.9,0.95), wd) semantics restated):
// g = grad * gscale[0];  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  master -= lr * (mhat / (sqrt(vhat) + eps) + wd * master)
__global__ __launch_bounds__(256) void adamw_kernel(bf16_t* __restrict__ p, float* __restrict__ master, const void* __restrict__ grad, int grad_f32,
                                                   float* __restrict__ m, float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2, const float* __restrict__ gscale) {
  const float gs = gscale ? gscale[0] : 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float g = gs * (grad_f32 ? reinterpret_cast<const float*>(grad)[i] : bf2f(reinterpret_cast<const bf16_t*>(grad)[i]));
    const float mi = b1 * m[i] + (1.f - b1) * g;
    const float vi = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mi; v[i] = vi;
    float w = master[i];
    w -= lr * ((mi / bc1) / (sqrtf(vi / bc2) + eps) + wd * w);
    master[i] = w;
    p[i] = f2bf(w);
  }
}

}  // namespace

#define AL16(p) ((((uintptr_t)(p)) & 15) == 0)

extern "C" int llmseg_colsum(const void* x, float* out, int64_t M, int64_t N, int64_t ld, void* workspace, int64_t workspace_bytes, void* stream) {
  LL_CHECK(x && out && M > 0 && N > 0 && ld >= N, "colsum: bad arguments");
  // row slices: >= 32 rows each (8 per row-lane) and enough of them that the launch reaches ~256 workgroups -- the head's bias gradients are 512 rows
  // x 256..2048 columns, and with 256-row slices every thread walked 64 rows one dependent load after the other (10-22 us per call)
  const long colwg = (N + 63) / 64;
  long gy = min((long)64, max((long)1, min((long)((M + 31) / 32), (long)((256 + colwg - 1) / colwg))));
  gy = min(gy, workspace ? (long)(workspace_bytes / (N * 4)) : 0L);          // row slices the scratch can hold; <= 1: one slice, no scratch
  float* part = gy > 1 ? (float*)workspace : nullptr;
  if (gy < 1) gy = 1;
  LL_LAUNCH_KERNEL(colsum_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)gy), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, out, part, (long)M,
                     (long)N, (long)ld);
  if (part)
    LL_LAUNCH_KERNEL(fold_slices_kernel, dim3(fold_grid(N)), dim3(256), 0, (hipStream_t)stream, (const float*)part, (int)gy, (long)N, (long)N, (long)N,
                       FoldOut{{out, nullptr, nullptr, nullptr}});
  LL_LAUNCH_CHECK("colsum");
  return LLMSEG_OK;
}

extern "C" int llmseg_norm_bwd(const void* dy, const void* x, const void* w, void* dx, float* dw, float* db, int64_t rows, int64_t cols, float eps,
                               int rms, void* workspace, int64_t workspace_bytes, void* stream) {
  return llmseg_norm_bwd_add(dy, x, w, nullptr, dx, dw, db, rows, cols, eps, rms, workspace, workspace_bytes, stream);
}

extern "C" int llmseg_norm_bwd_add(const void* dy, const void* x, const void* w, const void* dres, void* dx, float* dw, float* db, int64_t rows, int64_t cols,
                                   float eps, int rms, void* workspace, int64_t workspace_bytes, void* stream) {
  LL_CHECK(dy && x && w && dx && rows > 0 && cols > 0 && (cols & 7) == 0 && AL16(dy) && AL16(x) && AL16(w) && AL16(dx) && AL16(dres), "norm_bwd: bad arguments");
  const int cpl = (int)(((cols >> 3) + 63) / 64);
  const bool wgrad = dw || db;
  if (!wgrad && rows >= 64 && cols >= 2048 && cols <= 8192) {        // wide rows, frozen weight: a workgroup per row (also for tall activations: the
    // wave-per-row kernel holds x and dy of a 4096-wide row in 64 VGPRs per lane under a 128-VGPR bound and its counters showed 1.48 x the algorithmic bytes)
    const int cpt = (int)(((cols >> 3) + 255) / 256);
#define LL_NORMBW(C)                                                                                                                                    \
  LL_LAUNCH_KERNEL(norm_bwd_wg_kernel<C>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, \
                   (bf16_t*)dx, (long)rows, (int)cols, eps, rms, (const bf16_t*)dres)
    if (cpt <= 1) LL_NORMBW(1); else if (cpt <= 2) LL_NORMBW(2); else LL_NORMBW(4);
#undef LL_NORMBW
    LL_LAUNCH_CHECK("norm_bwd");
    return LLMSEG_OK;
  }
  const bool acc = wgrad && cpl <= 2;
  const long wgs = (rows + 3) / 4;
  const long ws_floats = workspace ? workspace_bytes / 4 : 0;
  long G = acc ? std::min<long>(std::max<long>(1, (rows + 31) / 32), 256) : wgs;                          // ACC: every wave walks >= 8 rows
  float *part = nullptr, *stats = nullptr;
  if (acc) {
    G = std::min<long>(G, ws_floats / (2 * cols));                                                         // partial slots the scratch can hold
    if (G > 1) part = (float*)workspace; else G = 1;
  } else if (wgrad) {
    LL_CHECK(ws_floats >= 2 * rows + 2 * cols, "norm_bwd: weight gradients of a wide norm need workspace >= (2 rows + 2 cols) floats (row statistics)");
    stats = (float*)workspace;
  }
  const dim3 grid((unsigned)G);
#define LL_NORMB(C, A)                                                                                                                       \
  LL_LAUNCH_KERNEL((norm_bwd_kernel<C, A>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, \
                     (bf16_t*)dx, dw, db, (long)rows, (int)cols, eps, rms, (const bf16_t*)dres, part, stats)
  if (acc) { if (cpl <= 1) LL_NORMB(1, true); else LL_NORMB(2, true); }
  else if (cpl <= 1) LL_NORMB(1, false); else if (cpl <= 2) LL_NORMB(2, false); else if (cpl <= 4) LL_NORMB(4, false);
  else if (cpl <= 8) LL_NORMB(8, false); else LL_NORMB(0, false);
#undef LL_NORMB
  if (acc && part)
    LL_LAUNCH_KERNEL(fold_slices_kernel, dim3(fold_grid(2 * cols)), dim3(256), 0, (hipStream_t)stream, (const float*)part, (int)G, (long)(2 * cols),
                       (long)(2 * cols), (long)cols, FoldOut{{dw, db, nullptr, nullptr}});
  if (stats) {                                                     // wide rows: column sums from the stored (mean, rstd), row slices folded in order
    float* p2 = stats + ((2 * rows + 63) / 64) * 64;
    long gy = std::min<long>(64, (rows + 255) / 256);
    gy = std::min<long>(gy, (ws_floats - (p2 - stats)) / (2 * cols));
    if (gy <= 1) { gy = 1; p2 = nullptr; }
    LL_LAUNCH_KERNEL(norm_dwdb_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)gy), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x,
                       (const float*)stats, dw, db, p2, (long)rows, (long)cols);
    if (p2)
      LL_LAUNCH_KERNEL(fold_slices_kernel, dim3(fold_grid(2 * cols)), dim3(256), 0, (hipStream_t)stream, (const float*)p2, (int)gy, (long)(2 * cols),
                         (long)(2 * cols), (long)cols, FoldOut{{dw, db, nullptr, nullptr}});
  }
  LL_LAUNCH_CHECK("norm_bwd");
  return LLMSEG_OK;
}

extern "C" int llmseg_swiglu_bwd(const void* gu, const void* dout, void* dgu, int64_t rows, int64_t I, void* stream) {
  LL_CHECK(gu && dout && dgu && rows > 0 && I > 0 && (I & 7) == 0 && AL16(gu) && AL16(dout) && AL16(dgu), "swiglu_bwd: bad arguments");
  LL_LAUNCH_KERNEL(swiglu_bwd_kernel, dim3(grid_for(rows * (I >> 3))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gu, (const bf16_t*)dout,
                     (bf16_t*)dgu, (long)rows, (long)I, (long)I);
  LL_LAUNCH_CHECK("swiglu_bwd");
  return LLMSEG_OK;
}

// library-internal (gemm.hip, the two-launch route of llmseg_gemm_args.fx = LLMSEG_FX_SWIGLU_BWD): d(out) rows of stride ld_dout, possibly inside dgu's own rows
extern "C" __attribute__((visibility("hidden"))) int llmseg_swiglu_bwd_ld(const void* gu, const void* dout, void* dgu, int64_t rows, int64_t I, int64_t ld_dout, void* stream) {
  LL_CHECK(gu && dout && dgu && rows > 0 && I > 0 && (I & 7) == 0 && (ld_dout & 7) == 0 && AL16(gu) && AL16(dout) && AL16(dgu), "swiglu_bwd: bad arguments");
  LL_LAUNCH_KERNEL(swiglu_bwd_kernel, dim3(grid_for(rows * (I >> 3))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gu, (const bf16_t*)dout,
                     (bf16_t*)dgu, (long)rows, (long)I, (long)ld_dout);
  LL_LAUNCH_CHECK("swiglu_bwd");
  return LLMSEG_OK;
}

extern "C" int llmseg_act_bwd(const void* dy, const void* y, void* out, int64_t n, int act, void* stream) {
  LL_CHECK(dy && y && out && n > 0 && (act == LLMSEG_ACT_RELU || act == LLMSEG_ACT_SIGMOID), "act_bwd: only relu/sigmoid are differentiated");
  LL_LAUNCH_KERNEL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)out, (long)n, act);
  LL_LAUNCH_CHECK("act_bwd");
  return LLMSEG_OK;
}

extern "C" int llmseg_softmax_rows(const float* S, void* P, int64_t BH, int32_t Tq, int32_t Tk, int32_t ld, float scale, int32_t causal,
                                   const uint8_t* key_mask, int32_t heads, void* stream) {
  LL_CHECK(S && P && BH > 0 && Tq > 0 && Tk > 0 && ld >= Tk && heads > 0 && (!causal || Tq == Tk), "softmax_rows: bad arguments");
  LL_LAUNCH_KERNEL(softmax_rows_kernel, dim3((unsigned)((BH * Tq + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, (bf16_t*)P, (long)BH, Tq, Tk, ld,
                     scale, causal, key_mask, heads);
  LL_LAUNCH_CHECK("softmax_rows");
  return LLMSEG_OK;
}

extern "C" int llmseg_attn_ds(const void* P, const float* dP, void* dS, int64_t rows, int32_t T, int32_t ld, float scale, void* stream) {
  LL_CHECK(P && dP && dS && rows > 0 && T > 0 && ld >= T, "attn_ds: bad arguments");
  LL_LAUNCH_KERNEL(attn_ds_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)P, dP, (bf16_t*)dS, (long)rows, T,
                     ld, scale);
  LL_LAUNCH_CHECK("attn_ds");
  return LLMSEG_OK;
}

extern "C" int llmseg_ce_bwd(const void* logits, const int64_t* labels, const float* coef, void* dlogits, int32_t N, int32_t T, int64_t V, int64_t ldl,
                             void* stream) {
  LL_CHECK(logits && labels && coef && dlogits && N > 0 && T > 1 && V > 0 && ldl >= V, "ce_bwd: bad arguments");
  LL_LAUNCH_KERNEL(ce_bwd_kernel, dim3((unsigned)(N * T)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, labels, coef, (bf16_t*)dlogits, T,
                     (long)V, (long)ldl);
  LL_LAUNCH_CHECK("ce_bwd");
  return LLMSEG_OK;
}

extern "C" int llmseg_scatter_add_rows(const void* src, const int64_t* idx, float* dst, int64_t n, int64_t cols, void* stream) {
  LL_CHECK(src && idx && dst && n > 0 && cols > 0, "scatter_add_rows: bad arguments");
  LL_CHECK(n < (1L << 31), "scatter_add_rows: too many source rows");
  LL_LAUNCH_KERNEL(scatter_add_rows_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, idx, dst, (long)n, (long)cols);
  LL_LAUNCH_CHECK("scatter_add_rows");
  return LLMSEG_OK;
}

extern "C" int llmseg_sumsq(const void* x, int64_t n, int is_f32, float* out, void* workspace, int64_t workspace_bytes, void* stream) {
  LL_CHECK(x && out && n > 0, "sumsq: bad arguments");
  LL_CHECK(workspace && workspace_bytes >= 4, "sumsq: workspace (>= 4 bytes, 8 KiB for every partial) is required");
  long g = std::min<long>(2048, (n + 2047) / 2048);                       // >= 8 elements per thread
  g = std::max<long>(1, std::min<long>(g, workspace_bytes / 4));
  LL_LAUNCH_KERNEL(sumsq_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, (long)n, is_f32, (float*)workspace);
  LL_LAUNCH_KERNEL(fold_column_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (long)g, out, 1.f);
  LL_LAUNCH_CHECK("sumsq");
  return LLMSEG_OK;
}

extern "C" int llmseg_adamw(void* p, float* master, const void* grad, int grad_f32, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int64_t step, const float* grad_scale, void* stream) {
  LL_CHECK(p && master && grad && m && v && n > 0 && step > 0, "adamw: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  LL_LAUNCH_KERNEL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)p, master, grad, grad_f32, m, v, (long)n, lr, beta1, beta2,
                     eps, weight_decay, bc1, bc2, grad_scale);
  LL_LAUNCH_CHECK("adamw");
  return LLMSEG_OK;
}

// ---- rank-8 LoRA kernels (peft Linear, r = 8): the skinny products a 128x128-tile GEMM wastes >90 % of its tile on ------------
namespace {
constexpr int LR = 8;

struct DropP { const unsigned long* rng; uint32_t stream, thr; float scale; uint32_t seg; };     // thr == 0: no dropout; seg: rows per segment (0 = none)

// Philox counter index of the 8 elements starting at (row m, column k) of a dense [M][width] activation, and what to add to the stream's
// offset: with segments (llmseg_dropout.seg_rows) segment s = m / seg draws the mask of a separate pass at offset + s over its own rows.
__device__ __forceinline__ unsigned long drop_index(const DropP& dp, unsigned long m, unsigned long width, unsigned long k, uint32_t& off_add) {
  off_add = 0;
  if (dp.seg) {
    off_add = (uint32_t)(m / dp.seg);
    m -= (unsigned long)off_add * dp.seg;
  }
  return (m * width + k) >> 3;
}

// Two rank-8 down-projections in ONE launch (q and v branches of a LoRA'd q|k|v projection):
//   y[m][0..7] = alpha * drop0(x0)[m][:] . W0^T,   y[m][8..15] = alpha * drop1(x1)[m][:] . W1^T,   y[m][16..16+zero_cols) = 0
// rows of y at pitch ldy (the [M][64] extension operand of the qkv GEMM is [x Aq^T | x Av^T | 0]); x0 == x1 in the forward pass (one
// load feeds both branches, each with its own dropout stream), two column blocks of dY in the backward pass.  w_kr = 0: W stored
// [8][K]; 1: W stored [K][8].  drop(x) = x * mask / (1 - p), mask element index m K + k.  nb = 1 computes the first branch only.
// RW rows per wave: 4 for tall activations (each W chunk is loaded once for four rows), 1 for short ones (M = 2 x 319: four times the
// waves in flight -- the kernel is latency-bound there).
template <int RW>
__global__ __launch_bounds__(256) void lora_down_kernel(const bf16_t* __restrict__ x0, const bf16_t* __restrict__ x1, long ldx, const bf16_t* __restrict__ w0,
                                                       const bf16_t* __restrict__ w1, bf16_t* __restrict__ y, long ldy, long M, int K, int w_kr, float alpha,
                                                       int zero_cols, int nb, DropP dp) {
  const int lane = threadIdx.x & 63;
  const long m0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (m0 >= M) return;
  float acc[2][RW][LR];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int r = 0; r < LR; ++r) acc[b][i][r] = 0.f;
  const bool same = x0 == x1;
#pragma unroll 2
  for (int k = lane * 8; k < K; k += 64 * 8) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (b >= nb) break;
      const bf16_t* xb = b == 0 ? x0 : x1;
      const bf16_t* wb = b == 0 ? w0 : w1;
      float xv[RW][8], wv[8];
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const long m = min(m0 + i, M - 1);
        unpack8(*reinterpret_cast<const uint4*>(xb + m * ldx + k), xv[i]);
        if (dp.thr) {
          uint32_t oa;
          const unsigned long di = drop_index(dp, (unsigned long)m, (unsigned long)K, (unsigned long)k, oa);
          dropout8(xv[i], di, dp.stream + b, dp.rng, dp.thr, dp.scale, oa);
        }
      }
      (void)same;
      if (w_kr) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          unpack8(*reinterpret_cast<const uint4*>(wb + (long)(k + j) * LR), wv);
#pragma unroll
          for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int r = 0; r < LR; ++r) acc[b][i][r] += xv[i][j] * wv[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < LR; ++r) {
          unpack8(*reinterpret_cast<const uint4*>(wb + (long)r * K + k), wv);
#pragma unroll
          for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[b][i][r] += xv[i][j] * wv[j];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < RW; ++i) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < LR; ++r) acc[b][i][r] = wave_sum(acc[b][i][r]) * alpha;
    if (m0 + i < M) {
      bf16_t* yr = y + (m0 + i) * ldy;
      if (lane == 0) *reinterpret_cast<uint4*>(yr) = pack8(acc[0][i]);
      else if (lane == 1 && nb > 1) *reinterpret_cast<uint4*>(yr + 8) = pack8(acc[1][i]);
      else if (lane >= nb && (lane - nb) * 8 < zero_cols) *reinterpret_cast<uint4*>(yr + lane * 8) = make_uint4(0, 0, 0, 0);
    }
  }
}

// The same product on the matrix cores (W stored [8][K]).  What bounds this product at M = 638 is the operand fetch rate of a CU
// (~16 B/clk): the per-row kernel above spreads over 160 CUs but re-reads both 8 x K weight blocks for every row (82 MB through L2:
// 20 us), a 16-row MFMA tile per workgroup reads every byte once but keeps only 40 CUs busy (20 us again).  So the 16-row tile is
// ALSO split along K over gridDim.y workgroups (320 workgroups at M = 638, K = 4096): a wave's v_mfma_f32_16x16x32_bf16 computes
// D[rank][m] += W[rank][k..k+32] . X[m][k..k+32]^T for the 16 rows (rows 8..15 of the W operand are zero), the 4 waves of a workgroup
// are folded through LDS, and a K-slice's fp32 partial goes to the caller's scratch [slice][M][16]; lora_down_finish_kernel adds the
// slices in order (deterministic), scales and writes the bf16 rows.  With gridDim.y == 1 (tall activations: enough row tiles to fill
// the chip) the first kernel writes the bf16 rows itself.  Dropout: the Philox mask zeroes elements of the bf16 X fragment (exact), the
// 1 / (1 - p) scale is folded into alpha -- no extra rounding.  Branches with different dropout streams need their own MFMA (X differs).
__global__ __launch_bounds__(256) void lora_down_mfma_kernel(const bf16_t* __restrict__ x0, const bf16_t* __restrict__ x1, long ldx, const bf16_t* __restrict__ w0,
                                                            const bf16_t* __restrict__ w1, bf16_t* __restrict__ y, long ldy, long M, int K, float alpha,
                                                            int zero_cols, int nb, DropP dp, float* __restrict__ part) {
  __shared__ float red[4][2][8][16];                   // [wave][branch][rank][row]
  __shared__ float fin[16][16];                        // [row][branch * 8 + rank]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kc = lane >> 4;
  const long m0 = (long)blockIdx.x * 16;
  const long m = min(m0 + r, M - 1);
  const int kq = K / (4 * (int)gridDim.y);             // K-range of one wave (a multiple of 32)
  const int kbeg = ((int)blockIdx.y * 4 + wave) * kq;
  const bool same = x0 == x1;
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  auto masked = [&](uint4 v, unsigned long idx, uint32_t stream, uint32_t off_add) {
    const unsigned long seed = dp.rng[0], off = dp.rng[1] + off_add;
    const Philox8 ph = philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), stream, (uint32_t)off, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t* u = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t lo = (ph.w[j] & 0xffffu) >= dp.thr ? 0x0000ffffu : 0u, hi = (ph.w[j] >> 16) >= dp.thr ? 0xffff0000u : 0u;
      u[j] &= lo | hi;
    }
    return v;
  };
#pragma unroll 4
  for (int k = kbeg + kc * 8; k < kbeg + kq; k += 32) {
    uint4 xa = *reinterpret_cast<const uint4*>(x0 + m * ldx + k);
    uint4 xb = (nb > 1 && !same) ? *reinterpret_cast<const uint4*>(x1 + m * ldx + k) : xa;
    const uint4 wa = r < LR ? *reinterpret_cast<const uint4*>(w0 + (long)r * K + k) : make_uint4(0, 0, 0, 0);
    const uint4 wb = (nb > 1 && r < LR) ? *reinterpret_cast<const uint4*>(w1 + (long)r * K + k) : make_uint4(0, 0, 0, 0);
    if (dp.thr) {
      uint32_t oa;
      const unsigned long idx = drop_index(dp, (unsigned long)m, (unsigned long)K, (unsigned long)k, oa);
      xa = masked(xa, idx, dp.stream, oa);
      if (nb > 1) xb = masked(xb, idx, dp.stream + 1, oa);
    }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa), __builtin_bit_cast(bf16x8_t, xa), acc0, 0, 0, 0);
    if (nb > 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wb), __builtin_bit_cast(bf16x8_t, xb), acc1, 0, 0, 0);
  }
  if (kc < 2) {                                        // acc[e] = D[rank 4 kc + e][row r]
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[wave][0][4 * kc + e][r] = acc0[e]; red[wave][1][4 * kc + e][r] = acc1[e]; }
  }
  __syncthreads();
  {
    const int b = threadIdx.x >> 7, rank = (threadIdx.x >> 4) & 7, mm = threadIdx.x & 15;
    const float sum = (red[0][b][rank][mm] + red[1][b][rank][mm]) + (red[2][b][rank][mm] + red[3][b][rank][mm]);
    if (part) {                                        // K-slice partial: [slice][M][16]
      if (m0 + mm < M) part[((long)blockIdx.y * M + m0 + mm) * 16 + b * 8 + rank] = sum;
    } else fin[mm][b * 8 + rank] = sum * alpha * (dp.thr ? dp.scale : 1.f);
  }
  if (part) return;
  __syncthreads();
  const int nch = nb + zero_cols / 8;                  // 16-byte chunks per output row
  for (int t = threadIdx.x; t < 16 * nch; t += blockDim.x) {
    const int mm = t / nch, ch = t - mm * nch;
    if (m0 + mm >= M) continue;
    uint4 o = make_uint4(0, 0, 0, 0);
    if (ch < nb) o = pack8(&fin[mm][ch * 8]);
    *reinterpret_cast<uint4*>(y + (m0 + mm) * ldy + ch * 8) = o;
  }
}

__device__ __forceinline__ void lora_down_finish_body(const float* __restrict__ part, int S, bf16_t* __restrict__ y, long ldy, long M, float scale,
                                                      int zero_cols, int nb, long bid, long nblocks) {
  const int nch = nb + zero_cols / 8;
  const long total = M * nch;
  for (long t = bid * blockDim.x + threadIdx.x; t < total; t += nblocks * blockDim.x) {
    const long mm = t / nch;
    const int ch = (int)(t - mm * nch);
    uint4 o = make_uint4(0, 0, 0, 0);
    if (ch < nb) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
      for (int s2 = 0; s2 < S; ++s2) {
        const float* pp = part + ((long)s2 * M + mm) * 16 + ch * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += pp[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= scale;
      o = pack8(v);
    }
    *reinterpret_cast<uint4*>(y + mm * ldy + ch * 8) = o;
  }
}
__global__ __launch_bounds__(256) void lora_down_finish_kernel(const float* __restrict__ part, int S, bf16_t* __restrict__ y, long ldy, long M, float scale,
                                                              int zero_cols, int nb) {
  lora_down_finish_body(part, S, y, ldy, M, scale, zero_cols, nb, blockIdx.x, gridDim.x);
}

// out(n,r) += alpha * sum_m drop(a)[m][n] * b[m][r]  (the caller zero-fills `out` or accumulates into a gradient arena; no atomics:
// with several row slices a workgroup's 512 sums go to part[slice][branch][N * 8] and fold_slices_kernel adds the slices in order).  out_rn = 0: out [N][8]; 1: out [8][N].  b has row pitch ldb.  blockIdx.z selects one of up to two independent
// products that share the shapes (the q and v branches of one layer: dAq / dAv read the same activation with their own dropout streams,
// dBq / dBv two column blocks of dY).
// Workgroup = 64 columns x one slice of rows, split again over its 4 waves: lane = (row-lane 0..7, column chunk 0..7), a wave reads 8
// rows x 128 contiguous bytes per step (16 B per lane, three steps in flight); the 8 row-lanes are folded with shuffles, the 4 waves in
// LDS, so ONE partial per output cell per workgroup.
// Up to FOUR products per launch (blockIdx.z), each with its own operands, pitches, output layout, scale and dropout stream (drop[z] = 0: no
// mask, else stream id + 1): the whole weight-gradient work of a LoRA'd q|k|v projection -- dBq, dBv, dAq, dAv -- is one launch + one fold.
struct OuterP { const bf16_t* a[4]; const bf16_t* b[4]; float* out[4]; long lda[4], ldb[4]; int out_rn[4]; float alpha[4]; uint32_t drop[4]; };
__global__ __launch_bounds__(256) void lora_outer_kernel(OuterP q, long M, long N, DropP dp, float* __restrict__ part) {
  __shared__ float red[4][8][8 * LR];                  // [wave][column chunk][8 columns x 8 ranks]
  const int z = blockIdx.z;
  const bf16_t* __restrict__ a = q.a[z];
  const bf16_t* __restrict__ b = q.b[z];
  float* __restrict__ out = q.out[z];
  const long lda = q.lda[z], ldb = q.ldb[z];
  const int out_rn = q.out_rn[z];
  const float alpha = q.alpha[z];
  if (q.drop[z] == 0) dp.thr = 0;
  const uint32_t dstream = q.drop[z] - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cl = lane & 7, rl = lane >> 3;
  const long n = ((long)blockIdx.x * 8 + cl) * 8;      // the workgroup's 4 waves cover the SAME 64 columns, each its own quarter of the row slice
  const long per = (M + gridDim.y * 4 - 1) / (gridDim.y * 4);
  const long m0 = ((long)blockIdx.y * 4 + wave) * per, m1 = min(M, m0 + per);
  const bool on = n < N;
  float acc[8][LR];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < LR; ++r) acc[j][r] = 0.f;
  if (on) {
    long m = m0 + rl;
    for (; m + 16 < m1; m += 24) {
      uint4 av[3], bv[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        av[u] = *reinterpret_cast<const uint4*>(a + (m + 8 * u) * lda + n);
        bv[u] = *reinterpret_cast<const uint4*>(b + (m + 8 * u) * ldb);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        float x[8], y[8];
        unpack8(av[u], x); unpack8(bv[u], y);
        if (dp.thr) {
          uint32_t oa;
          const unsigned long di = drop_index(dp, (unsigned long)(m + 8 * u), (unsigned long)N, (unsigned long)n, oa);
          dropout8(x, di, dstream, dp.rng, dp.thr, dp.scale, oa);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int r = 0; r < LR; ++r) acc[j][r] = fmaf(x[j], y[r], acc[j][r]);
      }
    }
    for (; m < m1; m += 8) {
      float x[8], y[8];
      unpack8(*reinterpret_cast<const uint4*>(a + m * lda + n), x);
      unpack8(*reinterpret_cast<const uint4*>(b + m * ldb), y);
      if (dp.thr) {
        uint32_t oa;
        const unsigned long di = drop_index(dp, (unsigned long)m, (unsigned long)N, (unsigned long)n, oa);
        dropout8(x, di, dstream, dp.rng, dp.thr, dp.scale, oa);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < LR; ++r) acc[j][r] = fmaf(x[j], y[r], acc[j][r]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < LR; ++r) {
      float v = acc[j][r];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      acc[j][r] = v * alpha;
    }
  if (rl == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < LR; ++r) red[wave][cl][j * LR + r] = acc[j][r];
  }
  __syncthreads();
  // 512 outputs (64 columns x 8 ranks) per workgroup: two per thread (row slices of other workgroups hold the other terms of the same cell)
  float* __restrict__ dstp = part ? part + ((long)blockIdx.y * gridDim.z + z) * N * LR : out;
  for (int idx = threadIdx.x; idx < 8 * 8 * LR; idx += blockDim.x) {
    const int c = idx / (8 * LR), jr = idx % (8 * LR), j = jr / LR, r = jr % LR;
    const long nn = ((long)blockIdx.x * 8 + c) * 8 + j;
    if (nn < N) {
      const float v = (red[0][c][jr] + red[1][c][jr]) + (red[2][c][jr] + red[3][c][jr]);
      const long cell = out_rn ? (long)r * N + nn : nn * LR + r;
      if (part) dstp[cell] = v;
      else dstp[cell] += v;
    }
  }
}

// y[m][n..n+7] += alpha * sum over nb branches of mask_b(m, n..) * sum_r xa[m][8 b + r] * W_b(n,r).  w_rn = 0: W stored [N][8]; 1: W
// stored [8][N].  xa row pitch ldxa (branch b reads columns 8 b .. 8 b + 7).  With dropout the product is masked like the forward input
// was (dX of the LoRA branches: ((dq Bq) Aq) * mask_q / (1 - p) + ((dv Bv) Av) * mask_v / (1 - p)) -- one pass over y for both.
__global__ __launch_bounds__(256) void lora_apply_kernel(bf16_t* __restrict__ y, long ldy, const bf16_t* __restrict__ xa, long ldxa,
                                                        const bf16_t* __restrict__ w0, const bf16_t* __restrict__ w1, long M, long N, int w_rn,
                                                        float alpha, int nb, DropP dp) {
  const long nch = N >> 3, total = M * nch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / nch, c = i % nch;
    float yv[8], wv[8];
    unpack8(*reinterpret_cast<const uint4*>(y + m * ldy + c * 8), yv);
    for (int b = 0; b < nb; ++b) {
      const bf16_t* w = b == 0 ? w0 : w1;
      float xv[8], dv[8];
      unpack8(*reinterpret_cast<const uint4*>(xa + m * ldxa + 8 * b), xv);
      if (w_rn) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dv[j] = 0.f;
#pragma unroll
        for (int r = 0; r < LR; ++r) {
          unpack8(*reinterpret_cast<const uint4*>(w + (long)r * N + c * 8), wv);
#pragma unroll
          for (int j = 0; j < 8; ++j) dv[j] += xv[r] * wv[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          unpack8(*reinterpret_cast<const uint4*>(w + (c * 8 + j) * LR), wv);
          float s = 0.f;
#pragma unroll
          for (int r = 0; r < LR; ++r) s += xv[r] * wv[r];
          dv[j] = s;
        }
      }
      if (dp.thr) {
        uint32_t oa;
        const unsigned long di = drop_index(dp, (unsigned long)m, (unsigned long)N, (unsigned long)(c * 8), oa);     // == i without segments
        dropout8(dv, di, dp.stream + b, dp.rng, dp.thr, dp.scale, oa);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) yv[j] += alpha * dv[j];
    }
    *reinterpret_cast<uint4*>(y + m * ldy + c * 8) = pack8(yv);
  }
}

// Split-K tail of a dX product that feeds a pre-norm's backward (llmseg_gemm_args.nb_x, round 6): a workgroup per row sums the row's S fp32 slab rows and
// rounds to bf16 (splitk_reduce_kernel's arithmetic: alpha = 1, no bias / activation / residual), optionally adds the LoRA branches' dX to it with
// lora_apply_kernel's arithmetic and rounds again (LORA: y += alpha * mask_b * (xa[:, 8b..] . W_b), W stored [8][N]), and -- holding that row as dy -- writes
// the norm backward + dres with norm_bwd_wg_kernel's arithmetic and thread -> column mapping: three launches and two round trips of the [M][N] gradient
// (reduce 6.9 + lora_apply 10.9 + norm_bwd 7.6 us per Llama layer at 2 images) in one pass.  Same bits as the three launches.
// lp != NULL (LORA only): the LoRA operand t = [s dq Bq | s dv Bv] arrives as the K-slice PARTIALS of llmseg_lora_down (fp32 [lS][rows][16]): threads 0..15 add
// the row's slices in order, scale and round to bf16 (lora_down_finish_kernel's arithmetic), hand the 16 values to the workgroup through LDS and store them (+ zero
// columns) into xa for the weight-gradient kernel -- the finish launch of the backward's rank-8 down projection rides here.
template <int CPT, bool LORA>
__global__ __launch_bounds__(256) void reduce_lora_normbwd_kernel(const float* __restrict__ slab, int S, long slab_sz, const bf16_t* __restrict__ x,
                                                                 const bf16_t* __restrict__ w, bf16_t* __restrict__ dx, long rows, int cols, float eps, int rms,
                                                                 const bf16_t* __restrict__ dres, bf16_t* __restrict__ xa, long ldxa,
                                                                 const bf16_t* __restrict__ w0, const bf16_t* __restrict__ w1, float alpha, int nb, DropP dp,
                                                                 const float* __restrict__ lp, int lS, float lscale, int lzero) {
  __shared__ float red[16];
  __shared__ float tl[16];
  const long row = blockIdx.x;
  const int nch = cols >> 3;
  const long N = cols;
  if (LORA && lp != nullptr) {
    if (threadIdx.x < 16) {
      float v = 0.f;
      if ((int)threadIdx.x < 8 * nb) {
        for (int s2 = 0; s2 < lS; ++s2) v += lp[((long)s2 * rows + row) * 16 + threadIdx.x];
        v *= lscale;
      }
      const bf16_t vb = f2bf(v);
      tl[threadIdx.x] = bf2f(vb);
      xa[row * ldxa + threadIdx.x] = vb;
    } else if ((int)threadIdx.x < 16 + lzero) xa[row * ldxa + threadIdx.x] = 0;
    __syncthreads();
  }
  uint4 xc[CPT], gc[CPT], wc[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = threadIdx.x + 256 * i;
    xc[i] = gc[i] = wc[i] = make_uint4(0, 0, 0, 0);
    if (c < nch) {
      xc[i] = *reinterpret_cast<const uint4*>(x + row * cols + c * 8);
      wc[i] = *reinterpret_cast<const uint4*>(w + c * 8);
      const float* sp = slab + row * N + c * 8;
      float4 a0 = *reinterpret_cast<const float4*>(sp), a1 = *reinterpret_cast<const float4*>(sp + 4);
      for (int s2 = 1; s2 < S; ++s2) {
        const float4 b0 = *reinterpret_cast<const float4*>(sp + s2 * slab_sz), b1 = *reinterpret_cast<const float4*>(sp + s2 * slab_sz + 4);
        a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
        a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
      }
      const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      gc[i] = pack8(v);                                   // the bf16 product row the reduce launch would have stored
      if (LORA) {
        float yv[8], wv[8];
        unpack8(gc[i], yv);
        for (int b = 0; b < nb; ++b) {
          const bf16_t* wl = b == 0 ? w0 : w1;
          float xv[8], dv[8];
          if (lp != nullptr) {
#pragma unroll
            for (int r = 0; r < 8; ++r) xv[r] = tl[8 * b + r];
          } else unpack8(*reinterpret_cast<const uint4*>(xa + row * ldxa + 8 * b), xv);
#pragma unroll
          for (int j = 0; j < 8; ++j) dv[j] = 0.f;
#pragma unroll
          for (int r = 0; r < LR; ++r) {
            unpack8(*reinterpret_cast<const uint4*>(wl + (long)r * N + c * 8), wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) dv[j] += xv[r] * wv[j];
          }
          if (dp.thr) {
            uint32_t oa;
            const unsigned long di = drop_index(dp, (unsigned long)row, (unsigned long)N, (unsigned long)(c * 8), oa);
            dropout8(dv, di, dp.stream + b, dp.rng, dp.thr, dp.scale, oa);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) yv[j] += alpha * dv[j];
        }
        gc[i] = pack8(yv);                                // ... and the bf16 row lora_apply would have left
      }
    }
  }
  normbwd_row<CPT>(xc, gc, wc, red, dx, row, cols, eps, rms, dres);
}

// The two [*, 64] extension operands of a LoRA'd q|k|v projection (llmseg_gemm_args.A2 / W2), rebuilt from the CURRENT LoRA
// matrices on every call (no cache to go stale when the optimizer updates them in place):
//   w2b [3H][64]: rows of the q block = [s Bq | 0], k block = 0, v block = [0 | s Bv | 0]     (forward: qkv += [xAq | xAv | 0] . w2b^T)
//   w2a [H][64]:  row h = [Aq[:, h] | Av[:, h] | 0]                                             (backward: dx += [tq | tv | 0] . w2a^T)
__device__ __forceinline__ void lora_pack_body(const bf16_t* __restrict__ aq, const bf16_t* __restrict__ bq, const bf16_t* __restrict__ av,
                                               const bf16_t* __restrict__ bv, bf16_t* __restrict__ w2b, bf16_t* __restrict__ w2a, bf16_t* __restrict__ bt,
                                               long H, float s, long bid, long nblocks) {
  const long total = 4 * H * 8 + (bt ? 2 * H : 0);                   // (3H + H) rows x 8 chunks of 8 columns (+ the B^T copies: 16 rows x H / 8 chunks)
  for (long i = bid * blockDim.x + threadIdx.x; i < total; i += nblocks * blockDim.x) {
    if (i >= 4 * H * 8) {                                            // bt [16][H]: rows 0..7 = Bq^T, 8..15 = Bv^T (the backward's rank-8 down projection wants K-contiguous rows)
      const long j = i - 4 * H * 8;
      const int row16 = (int)(j / (H >> 3));
      const long h0 = (j - (long)row16 * (H >> 3)) * 8;
      const bf16_t* src = row16 < 8 ? bq : bv;
      uint32_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (uint32_t)src[(h0 + 2 * e) * LR + (row16 & 7)] | ((uint32_t)src[(h0 + 2 * e + 1) * LR + (row16 & 7)] << 16);
      *reinterpret_cast<uint4*>(bt + (long)row16 * H + h0) = make_uint4(o[0], o[1], o[2], o[3]);
      continue;
    }
    const long row = i >> 3;
    const int ch = (int)(i & 7);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (row < 3 * H) {
      if (w2b == nullptr) continue;
      if (row < H && ch == 0) { unpack8(*reinterpret_cast<const uint4*>(bq + row * LR), v); }
      else if (row >= 2 * H && ch == 1) { unpack8(*reinterpret_cast<const uint4*>(bv + (row - 2 * H) * LR), v); }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= s;
      *reinterpret_cast<uint4*>(w2b + row * 64 + ch * 8) = pack8(v);
    } else {
      if (w2a == nullptr) continue;
      const long h = row - 3 * H;
      if (ch < 2) {
        const bf16_t* src = ch == 0 ? aq : av;
#pragma unroll
        for (int r = 0; r < LR; ++r) v[r] = bf2f(src[(long)r * H + h]);
      }
      *reinterpret_cast<uint4*>(w2a + h * 64 + ch * 8) = pack8(v);
    }
  }
}
__global__ __launch_bounds__(256) void lora_pack_kernel(const bf16_t* __restrict__ aq, const bf16_t* __restrict__ bq, const bf16_t* __restrict__ av,
                                                       const bf16_t* __restrict__ bv, bf16_t* __restrict__ w2b, bf16_t* __restrict__ w2a, bf16_t* __restrict__ bt,
                                                       long H, float s) {
  lora_pack_body(aq, bq, av, bv, w2b, w2a, bt, H, s, blockIdx.x, gridDim.x);
}
// lora_down's K-slice finish and lora_pack in ONE launch (round 6: the two are independent -- activations vs. weights -- and each is a ~4 us grid-stride pass):
// workgroups [0, gf) add the slices, [gf, gridDim.x) build the extension operands.  Same arithmetic as the two kernels.
__global__ __launch_bounds__(256) void lora_finish_pack_kernel(const float* __restrict__ part, int S, bf16_t* __restrict__ y, long ldy, long M, float scale, int zero_cols,
                                                              int nb, int gf, const bf16_t* __restrict__ aq, const bf16_t* __restrict__ bq,
                                                              const bf16_t* __restrict__ av, const bf16_t* __restrict__ bv, bf16_t* __restrict__ w2b,
                                                              bf16_t* __restrict__ w2a, bf16_t* __restrict__ bt, long H, float s) {
  if ((int)blockIdx.x < gf) lora_down_finish_body(part, S, y, ldy, M, scale, zero_cols, nb, blockIdx.x, gf);
  else lora_pack_body(aq, bq, av, bv, w2b, w2a, bt, H, s, (long)blockIdx.x - gf, (long)gridDim.x - gf);
}
// out[c][r] = in[r][c] for r < rows (zero for rows <= r < rows_pad): 64 x 64 tiles through LDS, 16-byte global accesses both ways.
__global__ __launch_bounds__(256) void transpose_pad_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, long rows, long cols, long ld_in,
                                                           long ld_out, long rows_pad) {
  __shared__ bf16_t tile[64][66];
  const long r0 = (long)blockIdx.y * 64, c0 = (long)blockIdx.x * 64;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = threadIdx.x + i * 256;
    const int r = idx >> 3, ch = idx & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r0 + r < rows && c0 + ch * 8 < cols) v = *reinterpret_cast<const uint4*>(in + (r0 + r) * ld_in + c0 + ch * 8);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&tile[r][ch * 8]);
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = threadIdx.x + i * 256;
    const int c = idx >> 3, rh = idx & 7;
    if (c0 + c >= cols || r0 + rh * 8 >= rows_pad) continue;
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = (uint32_t)tile[rh * 8 + 2 * j][c] | ((uint32_t)tile[rh * 8 + 2 * j + 1][c] << 16);
    *reinterpret_cast<uint4*>(out + (c0 + c) * ld_out + r0 + rh * 8) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

}  // namespace

static DropP make_drop(const llmseg_dropout* d) {
  DropP dp{nullptr, 0u, 0u, 1.f, 0u};
  if (d && d->rng_state && d->drop_thr > 0) {
    dp.rng = (const unsigned long*)d->rng_state; dp.stream = d->stream; dp.thr = d->drop_thr; dp.seg = d->seg_rows;
    dp.scale = 65536.f / (65536.f - (float)d->drop_thr);
  }
  return dp;
}

// library-internal (gemm.hip: the K-sliced route of llmseg_gemm_args.nb_x).  slab: fp32 [S][M][N]; la_t == NULL: no LoRA term.
extern "C" __attribute__((visibility("hidden"))) int llmseg_reduce_lora_normbwd(const float* slab, int S, int64_t M, int64_t N, const void* x, const void* w, void* dx, float eps,
                                                                                int rms, const void* dres, void* la_t, int64_t la_ldt, const void* la_w0,
                                                                                const void* la_w1, float la_alpha, const llmseg_dropout* la_drop, const float* la_part,
                                                                                int la_S, float la_scale, int la_zero, void* stream) {
  LL_CHECK(slab && S >= 1 && M > 0 && N >= 2048 && N <= 8192 && (N & 7) == 0 && x && w && dx && AL16(x) && AL16(w) && AL16(dx) && AL16(dres) && AL16(slab),
           "reduce_lora_normbwd: bad arguments");
  LL_CHECK(!la_t || (la_w0 && AL16(la_t) && AL16(la_w0) && AL16(la_w1) && (la_ldt & 7) == 0), "reduce_lora_normbwd: bad LoRA operands");
  LL_CHECK(!la_part || (la_t && la_S >= 1 && la_zero >= 0 && la_zero <= 240 && la_ldt >= 16 + la_zero), "reduce_lora_normbwd: bad LoRA partials");
  const DropP dp = make_drop(la_drop);
  const int cpt = (int)(((N >> 3) + 255) / 256), nb = la_w1 ? 2 : 1;
#define LL_RLN(C, L)                                                                                                                                     \
  LL_LAUNCH_KERNEL((reduce_lora_normbwd_kernel<C, L>), dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, slab, S, (long)(M * N), (const bf16_t*)x,    \
                   (const bf16_t*)w, (bf16_t*)dx, (long)M, (int)N, eps, rms, (const bf16_t*)dres, (bf16_t*)la_t, (long)la_ldt, (const bf16_t*)la_w0, \
                   (const bf16_t*)la_w1, la_alpha, nb, dp, la_part, la_S, la_scale, la_zero)
  if (la_t) { if (cpt <= 1) LL_RLN(1, true); else if (cpt <= 2) LL_RLN(2, true); else LL_RLN(4, true); }
  else { if (cpt <= 1) LL_RLN(1, false); else if (cpt <= 2) LL_RLN(2, false); else LL_RLN(4, false); }
#undef LL_RLN
  LL_LAUNCH_CHECK("reduce_lora_normbwd");
  return LLMSEG_OK;
}
#define LL_DROP_OK(d) (!(d) || (d)->drop_thr < 65536u)

extern "C" int llmseg_lora_down(const void* x0, const void* x1, int64_t ldx, const void* w0, const void* w1, void* y, int64_t ldy, int64_t M, int64_t K,
                                int32_t w_kr, float alpha, int32_t zero_cols, const llmseg_dropout* drop, void* stream) {
  return llmseg_lora_down_ws(x0, x1, ldx, w0, w1, y, ldy, M, K, w_kr, alpha, zero_cols, drop, nullptr, 0, stream);
}

// a lora_pack job waiting to ride in the finish launch of the lora_down call being issued on this thread (llmseg_lora_down_pack)
// llmseg_lora_down_parts: the K-sliced route leaves its fp32 partials for the consumer (no finish launch) and reports the slice count and the finish scale
struct PartsReq { bool active, done; int S; float scale; };
static thread_local PartsReq g_parts_req = {false, false, 0, 0.f};
struct PackReq { const bf16_t *aq, *bq, *av, *bv; bf16_t *w2b, *w2a, *bt; long H; float s; bool active, done; };
static thread_local PackReq g_pack_req = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, false, false};

extern "C" int llmseg_lora_down_ws(const void* x0, const void* x1, int64_t ldx, const void* w0, const void* w1, void* y, int64_t ldy, int64_t M, int64_t K,
                                   int32_t w_kr, float alpha, int32_t zero_cols, const llmseg_dropout* drop, void* scratch, int64_t scratch_bytes, void* stream) {
  const int nb = (x1 && w1) ? 2 : 1;
  LL_CHECK(x0 && w0 && y && (!x1) == (!w1) && M > 0 && K > 0 && (K & 7) == 0 && (ldx & 7) == 0 && (ldy & 7) == 0 && ldy >= 8 * nb + zero_cols &&
               zero_cols >= 0 && (zero_cols & 7) == 0 && zero_cols <= 480 && AL16(x0) && AL16(x1) && AL16(w0) && AL16(w1) && AL16(y) && LL_DROP_OK(drop),
           "lora_down: bad arguments");
  const DropP dp = make_drop(drop);
  const bf16_t *X0 = (const bf16_t*)x0, *X1 = (const bf16_t*)(x1 ? x1 : x0), *W0 = (const bf16_t*)w0, *W1 = (const bf16_t*)(w1 ? w1 : w0);
  static const int force = getenv("LLMSEG_LORA_DOWN") ? atoi(getenv("LLMSEG_LORA_DOWN")) : 0;      // tuning / bisecting: 1 = per-row kernel, 2 = MFMA without K slices
  if (!w_kr && (K & 127) == 0 && force != 1) {
    if (force == 2) scratch = nullptr;
    // MFMA form: 16-row tiles; K slices so that the launch has >= ~256 workgroups (each slice a multiple of 4 waves x 32 columns)
    const long tiles = (M + 15) / 16;
    int S = 1;
    while (tiles * S < 256 && (K % (4 * 32 * S * 2)) == 0 && scratch && (int64_t)S * 2 * M * 16 * 4 <= scratch_bytes && S < 32) S *= 2;
    float* part = S > 1 ? (float*)scratch : nullptr;
    LL_LAUNCH_KERNEL(lora_down_mfma_kernel, dim3((unsigned)tiles, (unsigned)S), dim3(256), 0, (hipStream_t)stream, X0, X1, (long)ldx, W0, W1, (bf16_t*)y, (long)ldy,
                       (long)M, (int)K, alpha, zero_cols, nb, dp, part);
    if (S > 1 && g_parts_req.active) {
      g_parts_req.done = true; g_parts_req.S = S; g_parts_req.scale = alpha * (dp.thr ? dp.scale : 1.f);
    } else if (S > 1 && g_pack_req.active) {
      const PackReq& q = g_pack_req;
      const unsigned gf = grid_for(M * (nb + zero_cols / 8)), gp = grid_for(4 * q.H * 8 + 2 * q.H);
      LL_LAUNCH_KERNEL(lora_finish_pack_kernel, dim3(gf + gp), dim3(256), 0, (hipStream_t)stream, (const float*)part, S, (bf16_t*)y, (long)ldy, (long)M,
                         alpha * (dp.thr ? dp.scale : 1.f), zero_cols, nb, (int)gf, q.aq, q.bq, q.av, q.bv, q.w2b, q.w2a, q.bt, q.H, q.s);
      g_pack_req.done = true;
    } else if (S > 1)
      LL_LAUNCH_KERNEL(lora_down_finish_kernel, dim3(grid_for(M * (nb + zero_cols / 8))), dim3(256), 0, (hipStream_t)stream, (const float*)part, S, (bf16_t*)y,
                         (long)ldy, (long)M, alpha * (dp.thr ? dp.scale : 1.f), zero_cols, nb);
  } else if (M <= 2048)
    LL_LAUNCH_KERNEL(lora_down_kernel<1>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, X0, X1, (long)ldx, W0, W1, (bf16_t*)y, (long)ldy, (long)M,
                       (int)K, w_kr, alpha, zero_cols, nb, dp);
  else
    LL_LAUNCH_KERNEL(lora_down_kernel<4>, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, (hipStream_t)stream, X0, X1, (long)ldx, W0, W1, (bf16_t*)y, (long)ldy, (long)M,
                       (int)K, w_kr, alpha, zero_cols, nb, dp);
  LL_LAUNCH_CHECK("lora_down");
  return LLMSEG_OK;
}

static int lora_outer_launch(const OuterP& q, int nz, int64_t M, int64_t N, const DropP& dp, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  // 64 columns per workgroup (its 4 waves split the rows): enough row slices to put >= 2 workgroups on every CU, each wave >= 24 rows
  const long colwg = (N + 63) / 64;
  long gy = max(max((long)1, 256 / (colwg * nz)), min((long)32, M / 512));
  gy = min(gy, workspace ? (long)(workspace_bytes / (nz * N * LR * 4)) : 0L);          // row slices the scratch can hold; <= 1: one slice, no scratch
  float* part = gy > 1 ? (float*)workspace : nullptr;
  if (gy < 1) gy = 1;
  LL_LAUNCH_KERNEL(lora_outer_kernel, dim3((unsigned)colwg, (unsigned)gy, (unsigned)nz), dim3(256), 0, stream, q, (long)M, (long)N, dp, part);
  if (part)
    LL_LAUNCH_KERNEL(fold_slices_kernel, dim3(fold_grid(nz * N * LR)), dim3(256), 0, stream, (const float*)part, (int)gy, (long)(nz * N * LR),
                     (long)(nz * N * LR), (long)(N * LR), FoldOut{{q.out[0], q.out[1], q.out[2], q.out[3]}});
  LL_LAUNCH_CHECK("lora_outer");
  return LLMSEG_OK;
}

extern "C" int llmseg_lora_outer(const void* a0, const void* a1, int64_t lda, const void* b0, const void* b1, int64_t ldb, float* out0, float* out1, int64_t M,
                                 int64_t N, int32_t out_rn, float alpha, const llmseg_dropout* drop, void* workspace, int64_t workspace_bytes, void* stream) {
  const int nz = (a1 && b1 && out1) ? 2 : 1;
  LL_CHECK(a0 && b0 && out0 && ((!a1) == (!b1)) && ((!a1) == (!out1)) && M > 0 && N > 0 && (N & 7) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 && AL16(a0) &&
               AL16(a1) && AL16(b0) && AL16(b1) && LL_DROP_OK(drop), "lora_outer: bad arguments (N, lda, ldb multiples of 8)");
  OuterP q{};
  const DropP dp = make_drop(drop);
  for (int z = 0; z < nz; ++z) {
    q.a[z] = (const bf16_t*)(z ? a1 : a0); q.b[z] = (const bf16_t*)(z ? b1 : b0); q.out[z] = z ? out1 : out0;
    q.lda[z] = lda; q.ldb[z] = ldb; q.out_rn[z] = out_rn; q.alpha[z] = alpha; q.drop[z] = dp.thr ? dp.stream + z + 1 : 0;
  }
  return lora_outer_launch(q, nz, M, N, dp, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int llmseg_lora_wgrads(const void* dq, const void* dv, int64_t ldd, const void* x, int64_t ldx, const void* xa, int64_t ldxa, const void* t,
                                  int64_t ldt, float* gbq, float* gbv, float* gaq, float* gav, int64_t M, int64_t H, float s, const llmseg_dropout* drop,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  LL_CHECK(dq && dv && x && xa && t && gbq && gbv && gaq && gav && M > 0 && H > 0 && (H & 7) == 0 && (ldd & 7) == 0 && (ldx & 7) == 0 && (ldxa & 7) == 0 &&
               (ldt & 7) == 0 && ldxa >= 16 && ldt >= 16 && AL16(dq) && AL16(dv) && AL16(x) && AL16(xa) && AL16(t) && LL_DROP_OK(drop),
           "lora_wgrads: bad arguments");
  const DropP dp = make_drop(drop);
  OuterP q{};
  // dBq [H][8] += s dq^T (drop_q(x) Aq^T),  dBv likewise;  dAq [8][H] += (s dq Bq)^T drop_q(x),  dAv likewise (streams: q = drop->stream, v = + 1)
  q.a[0] = (const bf16_t*)dq; q.b[0] = (const bf16_t*)xa;     q.out[0] = gbq; q.lda[0] = ldd; q.ldb[0] = ldxa; q.out_rn[0] = 0; q.alpha[0] = s;   q.drop[0] = 0;
  q.a[1] = (const bf16_t*)dv; q.b[1] = (const bf16_t*)xa + 8; q.out[1] = gbv; q.lda[1] = ldd; q.ldb[1] = ldxa; q.out_rn[1] = 0; q.alpha[1] = s;   q.drop[1] = 0;
  q.a[2] = (const bf16_t*)x;  q.b[2] = (const bf16_t*)t;      q.out[2] = gaq; q.lda[2] = ldx; q.ldb[2] = ldt;  q.out_rn[2] = 1; q.alpha[2] = 1.f; q.drop[2] = dp.thr ? dp.stream + 1 : 0;
  q.a[3] = (const bf16_t*)x;  q.b[3] = (const bf16_t*)t + 8;  q.out[3] = gav; q.lda[3] = ldx; q.ldb[3] = ldt;  q.out_rn[3] = 1; q.alpha[3] = 1.f; q.drop[3] = dp.thr ? dp.stream + 2 : 0;
  return lora_outer_launch(q, 4, M, H, dp, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int llmseg_lora_apply(void* y, int64_t ldy, const void* xa, int64_t ldxa, const void* w0, const void* w1, int64_t M, int64_t N, int32_t w_rn,
                                 float alpha, const llmseg_dropout* drop, void* stream) {
  LL_CHECK(y && xa && w0 && M > 0 && N > 0 && (N & 7) == 0 && (ldy & 7) == 0 && (ldxa & 7) == 0 && ldxa >= (w1 ? 16 : 8) && AL16(y) && AL16(xa) && AL16(w0) &&
               AL16(w1) && LL_DROP_OK(drop), "lora_apply: bad arguments");
  LL_CHECK(!(drop && drop->drop_thr) || ldy == N, "lora_apply: the dropout mask indexes y as a dense [M][N] matrix");
  LL_LAUNCH_KERNEL(lora_apply_kernel, dim3(grid_for(M * (N >> 3))), dim3(256), 0, (hipStream_t)stream, (bf16_t*)y, (long)ldy, (const bf16_t*)xa,
                     (long)ldxa, (const bf16_t*)w0, (const bf16_t*)w1, (long)M, (long)N, w_rn, alpha, w1 ? 2 : 1, make_drop(drop));
  LL_LAUNCH_CHECK("lora_apply");
  return LLMSEG_OK;
}

extern "C" int llmseg_lora_pack(const void* aq, const void* bq, const void* av, const void* bv, void* w2b, void* w2a, void* bt, int64_t H, float s, void* stream) {
  LL_CHECK(aq && bq && av && bv && (w2b || w2a || bt) && H > 0 && (H & 7) == 0 && AL16(aq) && AL16(bq) && AL16(av) && AL16(bv) && AL16(w2b) && AL16(w2a) && AL16(bt),
           "lora_pack: bad arguments");
  LL_LAUNCH_KERNEL(lora_pack_kernel, dim3(grid_for(4 * H * 8 + 2 * H)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)aq, (const bf16_t*)bq, (const bf16_t*)av,
                     (const bf16_t*)bv, (bf16_t*)w2b, (bf16_t*)w2a, (bf16_t*)bt, (long)H, s);
  LL_LAUNCH_CHECK("lora_pack");
  return LLMSEG_OK;
}

// llmseg_lora_down_ws WITHOUT its finish launch where it runs as K slices: *S_out = slice count (the fp32 partials [S][M][16] are then in `scratch` and y is NOT
// written: llmseg_gemm_args.nb_lora_part finishes them), *S_out = 0 when the call completed y itself (no slices on this shape).  *scale_out = the finish scale.
extern "C" int llmseg_lora_down_parts(const void* x0, const void* x1, int64_t ldx, const void* w0, const void* w1, void* y, int64_t ldy, int64_t M, int64_t K,
                                      int32_t w_kr, float alpha, int32_t zero_cols, const llmseg_dropout* drop, void* scratch, int64_t scratch_bytes, int32_t* S_out,
                                      float* scale_out, void* stream) {
  LL_CHECK(S_out && scale_out, "lora_down_parts: S_out / scale_out");
  static const bool off = getenv("LLMSEG_NO_LORA_PARTS") != nullptr;      // A/B switch: always finish here
  g_parts_req = PartsReq{!off, false, 0, 0.f};
  const int rc = llmseg_lora_down_ws(x0, x1, ldx, w0, w1, y, ldy, M, K, w_kr, alpha, zero_cols, drop, scratch, scratch_bytes, stream);
  *S_out = g_parts_req.done ? g_parts_req.S : 0;
  *scale_out = g_parts_req.scale;
  g_parts_req.active = false;
  return rc;
}
// library-internal (gemm.hip: the two-launch route of llmseg_gemm_args.nb_lora_part)
extern "C" __attribute__((visibility("hidden"))) int llmseg_lora_down_finish(const float* part, int S, void* y, int64_t ldy, int64_t M, float scale, int zero_cols, int nb, void* stream) {
  LL_LAUNCH_KERNEL(lora_down_finish_kernel, dim3(grid_for(M * (nb + zero_cols / 8))), dim3(256), 0, (hipStream_t)stream, part, S, (bf16_t*)y, (long)ldy, (long)M, scale,
                     zero_cols, nb);
  LL_LAUNCH_CHECK("lora_down_finish");
  return LLMSEG_OK;
}

// llmseg_lora_down_ws followed by llmseg_lora_pack as one entry point: where the down projection runs as K slices, the pack rides in its finish launch (same bits)
extern "C" int llmseg_lora_down_pack(const void* x0, const void* x1, int64_t ldx, const void* w0, const void* w1, void* y, int64_t ldy, int64_t M, int64_t K,
                                     int32_t w_kr, float alpha, int32_t zero_cols, const llmseg_dropout* drop, void* scratch, int64_t scratch_bytes, const void* aq,
                                     const void* bq, const void* av, const void* bv, void* w2b, void* w2a, void* bt, int64_t H, float s, void* stream) {
  LL_CHECK(aq && bq && av && bv && (w2b || w2a || bt) && H > 0 && (H & 7) == 0 && AL16(aq) && AL16(bq) && AL16(av) && AL16(bv) && AL16(w2b) && AL16(w2a) && AL16(bt),
           "lora_down_pack: bad pack arguments");
  static const bool off = getenv("LLMSEG_NO_FINISH_PACK") != nullptr;      // A/B switch: always the two launches
  g_pack_req = PackReq{(const bf16_t*)aq, (const bf16_t*)bq, (const bf16_t*)av, (const bf16_t*)bv, (bf16_t*)w2b, (bf16_t*)w2a, (bf16_t*)bt, (long)H, s, !off, false};
  const int rc = llmseg_lora_down_ws(x0, x1, ldx, w0, w1, y, ldy, M, K, w_kr, alpha, zero_cols, drop, scratch, scratch_bytes, stream);
  const bool done = g_pack_req.done;
  g_pack_req.active = false;
  if (rc != LLMSEG_OK || done) return rc;
  return llmseg_lora_pack(aq, bq, av, bv, w2b, w2a, bt, H, s, stream);
}

extern "C" int llmseg_transpose_pad(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out, int64_t rows_pad, void* stream) {
  LL_CHECK(in && out && rows > 0 && cols > 0 && rows_pad >= rows && ld_out >= rows_pad && ld_in >= cols, "transpose_pad: bad arguments");
  LL_CHECK((cols & 7) == 0 && (ld_in & 7) == 0 && (ld_out & 7) == 0 && (rows_pad & 7) == 0 && AL16(in) && AL16(out),
           "transpose_pad: cols, rows_pad and leading dimensions must be multiples of 8, pointers 16-byte aligned");
  LL_LAUNCH_KERNEL(transpose_pad_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)((rows_pad + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)in, (bf16_t*)out, (long)rows, (long)cols, (long)ld_in, (long)ld_out, (long)rows_pad);
  LL_LAUNCH_CHECK("transpose_pad");
  return LLMSEG_OK;
}
