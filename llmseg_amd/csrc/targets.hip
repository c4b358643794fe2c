// Proposal decode + training-target computation on the device: what the reference does per sample on CPU
// workers -- pycocotools RLE decode (utils/sam_mask_reader.py:69-113), zero-pad to square + antialiased bilinear resize to 256 x 256
// (utils/reason_seg_dataset.py:166-173), IoU / IoP of every proposal against the ground truth (utils/utils.py:174-272) -- as three
// HBM-bound kernels on uint8 masks.  Integer results (pixels, intersections, areas) are exact; the IoU / IoP divisions are IEEE double
// divisions of those integers, i.e. bit-identical to numpy's.  All index tables (nearest-neighbour source rows / columns, resampling
// taps and weights) are computed on the host in float64 with the library formulas and passed in, so no device floating-point
// contraction can move a sample point.
#include <algorithm>
#include "common.h"
#include "llmseg_hip.h"

namespace {

// COCO RLE -> dense masks.  ends: inclusive prefix sums of the run lengths of all K masks, concatenated (uint32); offs[k] .. offs[k+1]
// is mask k's slice.  Runs alternate 0 / 1 starting with 0 and walk the image in COLUMN-major order (pixel index p = x H + y).
// out: [K][H][W] (hwk = 0) or [H][W][K] (hwk = 1, the reader's layout).
__global__ __launch_bounds__(256) void rle_decode_kernel(const uint32_t* __restrict__ ends, const int64_t* __restrict__ offs, uint8_t* __restrict__ out,
                                                        int K, int H, int W, int hwk) {
  const long n = (long)K * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int k, y, x;
    if (hwk) { k = (int)(i % K); const long r = i / K; x = (int)(r % W); y = (int)(r / W); }
    else { x = (int)(i % W); const long r = i / W; y = (int)(r % H); k = (int)(r / H); }
    const uint32_t p = (uint32_t)x * (uint32_t)H + (uint32_t)y;
    long lo = offs[k], hi = offs[k + 1];                 // first run whose end exceeds p
    const long base = lo;
    while (lo < hi) {
      const long mid = (lo + hi) >> 1;
      if (ends[mid] > p) hi = mid; else lo = mid + 1;
    }
    out[i] = (uint8_t)((lo - base) & 1);                 // beyond the last run (malformed input): parity of the run count, as pycocotools leaves zeros
  }
}

// Per proposal k: S_k = |seg_k|, I_k = |seg_k & gt'|, where gt' is the ground truth resampled to the proposals' H x W grid by the
// nearest-neighbour maps gy[H], gx[W] (skimage.transform.resize(order=0) == scipy.ndimage.zoom(grid_mode=True), tables from the host).
// segs [K][H][W] uint8 (non-zero = inside).  cnt int64 [K][2] += {I_k, S_k}; gcnt int64[1] += |gt'| (added by the k = 0 blocks only).
__global__ __launch_bounds__(256) void mask_targets_kernel(const uint8_t* __restrict__ segs, const uint8_t* __restrict__ gt, const int32_t* __restrict__ gy,
                                                          const int32_t* __restrict__ gx, int H, int W, int Wg, unsigned long long* __restrict__ cnt,
                                                          unsigned long long* __restrict__ gcnt) {
  __shared__ unsigned long long sh[3];
  if (threadIdx.x < 3) sh[threadIdx.x] = 0;
  __syncthreads();
  const int k = blockIdx.y;
  const uint8_t* s = segs + (long)k * H * W;
  unsigned int ci = 0, cs = 0, cg = 0;
  const long n = (long)H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / W), x = (int)(i % W);
    const unsigned int g = gt[(long)gy[y] * Wg + gx[x]] != 0, v = s[i] != 0;
    ci += v & g; cs += v; cg += g;
  }
  unsigned int v3[3] = {ci, cs, cg};
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    unsigned int v = v3[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sh[j], (unsigned long long)v);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (sh[0]) atomicAdd(&cnt[2 * k], sh[0]);
    if (sh[1]) atomicAdd(&cnt[2 * k + 1], sh[1]);
    if (k == 0 && sh[2]) atomicAdd(gcnt, sh[2]);
  }
}

// iou_k = I / (S + G - I), iop_k = I / S as IEEE doubles of the exact integer counts (numpy: np.sum(bool) / np.sum(bool); 0 / 0 = nan)
__global__ void targets_finalize_kernel(const unsigned long long* __restrict__ cnt, const unsigned long long* __restrict__ gcnt, double* __restrict__ iou,
                                        double* __restrict__ iop, int K) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const double I = (double)cnt[2 * k], S = (double)cnt[2 * k + 1], G = (double)gcnt[0];
  iou[k] = I / (S + G - I);
  iop[k] = I / S;
}

// Antialiased separable resampling (torch F.interpolate(mode="bilinear", antialias=True, align_corners=False) of the zero-padded square
// mask): out[k][oy][ox] = sum_j wy[oy][j] sum_i wx[ox][i] in[k][y0[oy] + j][x0[ox] + i], samples beyond H / W are the zero padding.
// Tap tables (first index, count, weights in float64) come from the host.  Accumulation in double, one rounding to bf16.
__global__ __launch_bounds__(256) void resize_aa_kernel(const uint8_t* __restrict__ segs, bf16_t* __restrict__ out, int H, int W, int OS,
                                                       const int32_t* __restrict__ y0, const int32_t* __restrict__ ny, const double* __restrict__ wy,
                                                       const int32_t* __restrict__ x0, const int32_t* __restrict__ nx, const double* __restrict__ wx, int taps) {
  const int k = blockIdx.z, oy = blockIdx.y;
  const int ox = blockIdx.x * blockDim.x + threadIdx.x;
  if (ox >= OS) return;
  const uint8_t* s = segs + (long)k * H * W;
  const int ys = y0[oy], yn = ny[oy], xs = x0[ox], xn = nx[ox];
  double acc = 0.0;
  for (int j = 0; j < yn; ++j) {
    const int y = ys + j;
    if (y >= H) break;                                   // zero padding below the image
    double row = 0.0;
    for (int i = 0; i < xn; ++i) {
      const int x = xs + i;
      if (x < W && s[(long)y * W + x]) row += wx[(long)ox * taps + i];
    }
    acc += wy[(long)oy * taps + j] * row;
  }
  out[((long)k * OS + oy) * OS + ox] = f2bf((float)acc);
}

// ---- one pass over the proposals (round 4) -----------------------------------------------------------------------------------------
// gtp[y][x] = (gt[gy[y]][gx[x]] != 0): the ground truth on the proposals' grid, built once per (image, ground truth) instead of being
// gathered again for every proposal
__global__ __launch_bounds__(256) void gt_resample_kernel(const uint8_t* __restrict__ gt, const int32_t* __restrict__ gy, const int32_t* __restrict__ gx,
                                                         uint8_t* __restrict__ out, int H, int W, int Wg) {
  const long n = (long)H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / W), x = (int)(i % W);
    out[i] = gt[(long)gy[y] * Wg + gx[x]] != 0;
  }
}

// Every byte of a proposal is read ONCE (16-byte loads, through the `order` index -- no gathered copy of the selected proposals) and feeds
// (a) the integer counts |seg|, |seg & gt'_g| for up to 4 ground truths and (b) the antialiased resize, done separably with EXACTLY the
// arithmetic of resize_aa_kernel: rowsum[y][ox] = sum_i wx[ox][i] in[y][x0 + i] (double, i ascending), then out[oy][ox] = bf16(sum_j wy[oy][j]
// rowsum[y0 + j][ox]) (double, j ascending) -- the per-row sums are shared by the ~2 output rows that use them instead of being recomputed
// per output pixel (adding 0.0 where the old kernel skipped a tap is exact).  Grid = (proposal, slice of output rows): a workgroup walks its
// output rows in order, stages the input rows they need FOUR at a time in LDS, keeps their row sums in a ring of RING rows x OS doubles, and
// counts the input rows [first row of its slice, first row of the next slice) -- a disjoint cover of the image; counts are integer atomics.
// Limits: OS <= 256 (one thread per output column), taps + 4 <= RING, W <= 2048.
constexpr int TG_ROWS = 4, TG_MAXW = 2048, TG_MAXG = 4, TG_WREG = 12;
template <int RING>
__global__ __launch_bounds__(256) void proposal_targets_kernel(const uint8_t* __restrict__ masks, const int64_t* __restrict__ order, const uint8_t* __restrict__ gtp,
                                                              int n_gt, bf16_t* __restrict__ out, int H, int W, int OS, const int32_t* __restrict__ y0,
                                                              const int32_t* __restrict__ ny, const double* __restrict__ wy, const int32_t* __restrict__ x0,
                                                              const int32_t* __restrict__ nx, const double* __restrict__ wx, int taps,
                                                              unsigned long long* __restrict__ cnt, unsigned long long* __restrict__ gcnt, int K) {
  __shared__ double ring[RING][256];
  __shared__ __attribute__((aligned(16))) uint8_t rows[TG_ROWS][TG_MAXW + 16];    // + 16: the word reads of the last column's window
  __shared__ unsigned int s_cnt[1 + 2 * TG_MAXG];
  const int k = blockIdx.x, t = threadIdx.x;
  const int oy_a = (int)((long)blockIdx.y * OS / gridDim.y), oy_b = (int)((long)(blockIdx.y + 1) * OS / gridDim.y);
  const int cnt_a = blockIdx.y == 0 ? 0 : y0[oy_a], cnt_b = blockIdx.y + 1 == gridDim.y ? H : min(H, y0[oy_b]);      // input rows this slice counts
  const uint8_t* m = masks + (order ? order[k] : (long)k) * H * W;
  if (t < 1 + 2 * TG_MAXG) s_cnt[t] = 0;
  const int ox = t;
  const bool col_on = ox < OS;
  const int xs = col_on ? x0[ox] : 0, xn = col_on ? nx[ox] : 0;
  unsigned int c_s = 0, c_i[TG_MAXG] = {0, 0, 0, 0}, c_g[TG_MAXG] = {0, 0, 0, 0};
  const int wch = (W + 15) >> 4;                                   // 16-byte chunks per row (the last may be partial)
  const bool vec_ok = (W & 15) == 0 && ((((uintptr_t)m) & 15) == 0) && (gtp == nullptr || (((uintptr_t)gtp) & 15) == 0);
  auto nzmask = [](uint32_t w) {            // per byte: 1 if any bit is set (each fold is masked so that no bit crosses a byte boundary)
    w = (w | (w >> 4)) & 0x0f0f0f0fu; w = (w | (w >> 2)) & 0x03030303u; return (w | (w >> 1)) & 0x01010101u;
  };
  auto stage = [&](int ya, int nrows) {                            // rows ya .. ya + nrows - 1 -> LDS, counting the ones this slice owns
    for (int idx = t; idx < nrows * wch; idx += 256) {
      const int r = idx / wch, c = idx - r * wch;
      const int y = ya + r;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (vec_ok) v = *reinterpret_cast<const uint4*>(m + (long)y * W + c * 16);
      else {
        uint8_t* vb = reinterpret_cast<uint8_t*>(&v);
        for (int e = 0; e < 16; ++e) { const int x = c * 16 + e; vb[e] = x < W ? m[(long)y * W + x] : 0; }
      }
      *reinterpret_cast<uint4*>(&rows[r][c * 16]) = v;
      if (y < cnt_a || y >= cnt_b) continue;
      const uint32_t n0 = nzmask(v.x), n1 = nzmask(v.y), n2 = nzmask(v.z), n3 = nzmask(v.w);      // non-zero bytes -> 0x01 each (masks hold 0 / 1 or 0 / 255)
      c_s += __popc(n0) + __popc(n1) + __popc(n2) + __popc(n3);
      for (int g = 0; g < n_gt; ++g) {
        uint4 q = make_uint4(0, 0, 0, 0);
        const uint8_t* gp = gtp + ((long)g * H + y) * W;
        if (vec_ok) q = *reinterpret_cast<const uint4*>(gp + c * 16);
        else {
          uint8_t* qb = reinterpret_cast<uint8_t*>(&q);
          for (int e = 0; e < 16; ++e) { const int x = c * 16 + e; qb[e] = x < W ? gp[x] : 0; }
        }
        const uint32_t g0 = nzmask(q.x), g1 = nzmask(q.y), g2 = nzmask(q.z), g3 = nzmask(q.w);
        c_i[g] += __popc(n0 & g0) + __popc(n1 & g1) + __popc(n2 & g2) + __popc(n3 & g3);
        if (k == 0) c_g[g] += __popc(g0) + __popc(g1) + __popc(g2) + __popc(g3);
      }
    }
  };
  double wreg[TG_WREG];                                            // this column's horizontal weights (register-resident when there are <= 12 taps)
  const bool in_regs = taps <= TG_WREG;
#pragma unroll
  for (int i = 0; i < TG_WREG; ++i) {
    const bool on = col_on && in_regs && i < xn && xs + i < W;      // a tap beyond the row (zero padding) or beyond the window weighs nothing
    wreg[i] = on ? wx[(long)ox * taps + i] : 0.0;
  }
  const int xw = min(xs, TG_MAXW - 1) & ~3;                         // the window's 12 bytes = 4 aligned LDS words + one byte-align per 4 taps
  const unsigned sh = (unsigned)(min(xs, TG_MAXW - 1) & 3);
  auto rowsums = [&](int ya, int nrows) {                          // rowsum rows ya .. into the ring: branch-free, the four rows' chains interleave
    if (!col_on) return;
    if (in_regs) {
      double acc[TG_ROWS];
#pragma unroll
      for (int r = 0; r < TG_ROWS; ++r) acc[r] = 0.0;
#pragma unroll
      for (int r = 0; r < TG_ROWS; ++r) {                            // rows beyond nrows hold stale bytes: computed, never stored
        const uint32_t* wp = reinterpret_cast<const uint32_t*>(&rows[r][xw]);
        const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
        const uint32_t b[3] = {__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh), __builtin_amdgcn_alignbyte(w3, w2, sh)};
#pragma unroll
        for (int i = 0; i < TG_WREG; ++i) acc[r] += ((b[i >> 2] >> (8 * (i & 3))) & 0xffu) ? wreg[i] : 0.0;
      }
#pragma unroll
      for (int r = 0; r < TG_ROWS; ++r)
        if (r < nrows) ring[(ya + r) & (RING - 1)][ox] = acc[r];
    } else {
      for (int r = 0; r < nrows; ++r) {
        double acc = 0.0;
        for (int i = 0; i < xn; ++i) {
          const int x = xs + i;
          if (x < W && rows[r][x]) acc += wx[(long)ox * taps + i];
        }
        ring[(ya + r) & (RING - 1)][ox] = acc;
      }
    }
  };
  int next_y = blockIdx.y == 0 ? 0 : y0[oy_a];
  for (int oy = oy_a; oy < oy_b; ++oy) {
    const int ya = y0[oy], need = min(H, ya + ny[oy]);
    while (next_y < need) {
      const int nrows = min(TG_ROWS, H - next_y);                  // staging ahead of `need` is fine: the ring holds taps + 4 rows
      __syncthreads();                                             // the previous batch's readers are done with `rows`
      stage(next_y, nrows);
      __syncthreads();
      rowsums(next_y, nrows);
      next_y += nrows;
    }
    __syncthreads();                                               // ring rows of this output row are complete
    if (col_on) {
      double acc = 0.0;
      const int yn = ny[oy];
      for (int j = 0; j < yn; ++j) {
        const int y = ya + j;
        if (y >= H) break;                                         // zero padding below the image
        acc += wy[(long)oy * taps + j] * ring[y & (RING - 1)][ox];
      }
      out[((long)k * OS + oy) * OS + ox] = f2bf((float)acc);
    }
  }
  // rows this slice owns but none of its windows reached: counts only
  while (next_y < cnt_b) {
    const int nrows = min(TG_ROWS, cnt_b - next_y);
    __syncthreads();
    stage(next_y, nrows);
    next_y += nrows;
  }
  auto fold = [&](unsigned int v, int slot) {                      // wave shuffles, then LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((t & 63) == 0 && v) atomicAdd(&s_cnt[slot], v);
  };
  __syncthreads();
  fold(c_s, 0);
  for (int g = 0; g < n_gt; ++g) { fold(c_i[g], 1 + 2 * g); if (k == 0) fold(c_g[g], 2 + 2 * g); }
  __syncthreads();
  if (t < n_gt) {                                                  // integer atomics onto zero-filled counters: exact in any order
    if (s_cnt[1 + 2 * t]) atomicAdd(&cnt[((long)t * K + k) * 2], (unsigned long long)s_cnt[1 + 2 * t]);
    if (s_cnt[0]) atomicAdd(&cnt[((long)t * K + k) * 2 + 1], (unsigned long long)s_cnt[0]);
    if (k == 0 && s_cnt[2 + 2 * t]) atomicAdd(&gcnt[t], (unsigned long long)s_cnt[2 + 2 * t]);
  }
}

}  // namespace

extern "C" int llmseg_gt_resample(const uint8_t* gt, const int32_t* gy, const int32_t* gx, uint8_t* out, int32_t H, int32_t W, int32_t Hg, int32_t Wg, void* stream) {
  LL_CHECK(gt && gy && gx && out && H > 0 && W > 0 && Hg > 0 && Wg > 0, "gt_resample: bad arguments");
  const long n = (long)H * W;
  LL_LAUNCH_KERNEL(gt_resample_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, gt, gy, gx, out, H, W, Wg);
  LL_LAUNCH_CHECK("gt_resample");
  return LLMSEG_OK;
}

extern "C" int llmseg_proposal_targets(const uint8_t* masks, const int64_t* order, const uint8_t* gtp, int32_t n_gt, void* out, int32_t K, int32_t H, int32_t W,
                                       int32_t out_size, const int32_t* y0, const int32_t* ny, const double* wy, const int32_t* x0, const int32_t* nx,
                                       const double* wx, int32_t taps, int64_t* counts, int64_t* gt_area, double* iou, double* iop, void* stream) {
  LL_CHECK(masks && out && y0 && ny && wy && x0 && nx && wx && K > 0 && H > 0 && W > 0, "proposal_targets: bad arguments");
  LL_CHECK(out_size > 0 && out_size <= 256 && taps > 0 && taps + TG_ROWS <= 32 && W <= TG_MAXW,
           "proposal_targets: out_size <= 256, taps <= 28, W <= 2048 (use llmseg_mask_targets + llmseg_resize_aa beyond)");
  LL_CHECK(n_gt >= 0 && n_gt <= TG_MAXG && (n_gt == 0 || (gtp && counts && gt_area && iou && iop)), "proposal_targets: 0..4 ground truths with their outputs");
  if (n_gt) {
    LL_CHECK(hipMemsetAsync(counts, 0, (size_t)n_gt * K * 2 * sizeof(int64_t), (hipStream_t)stream) == hipSuccess, "proposal_targets: memset failed");
    LL_CHECK(hipMemsetAsync(gt_area, 0, (size_t)n_gt * sizeof(int64_t), (hipStream_t)stream) == hipSuccess, "proposal_targets: memset failed");
  }
  // slices of output rows per proposal: enough workgroups for ~4 per CU (the kernel is latency-bound per workgroup), each >= 16 output rows
  int ns = (int)std::min<long>(std::max<long>(1, 1024 / K), std::max(1, out_size / 16));
  const dim3 grid((unsigned)K, (unsigned)ns);
#define LL_PT(R)                                                                                                                                              \
  LL_LAUNCH_KERNEL(proposal_targets_kernel<R>, grid, dim3(256), 0, (hipStream_t)stream, masks, order, gtp, n_gt, (bf16_t*)out, H, W, out_size, y0, ny, wy, x0, nx, \
                   wx, taps, (unsigned long long*)counts, (unsigned long long*)gt_area, K)
  if (taps + TG_ROWS <= 16) LL_PT(16); else LL_PT(32);
#undef LL_PT
  for (int g = 0; g < n_gt; ++g)
    LL_LAUNCH_KERNEL(targets_finalize_kernel, dim3((unsigned)((K + 63) / 64)), dim3(64), 0, (hipStream_t)stream, (const unsigned long long*)counts + (long)g * K * 2,
                     (const unsigned long long*)gt_area + g, iou + (long)g * K, iop + (long)g * K, K);
  LL_LAUNCH_CHECK("proposal_targets");
  return LLMSEG_OK;
}

extern "C" int llmseg_rle_decode(const uint32_t* run_ends, const int64_t* offsets, uint8_t* out, int32_t K, int32_t H, int32_t W, int32_t hwk, void* stream) {
  LL_CHECK(run_ends && offsets && out && K > 0 && H > 0 && W > 0 && (long)H * W < (1L << 32), "rle_decode: bad arguments");
  const long n = (long)K * H * W;
  LL_LAUNCH_KERNEL(rle_decode_kernel, dim3((unsigned)((n + 255) / 256 > 65535 * 16 ? 65535 * 16 : (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, run_ends,
                     offsets, out, K, H, W, hwk);
  LL_LAUNCH_CHECK("rle_decode");
  return LLMSEG_OK;
}

extern "C" int llmseg_mask_targets(const uint8_t* segs, const uint8_t* gt, const int32_t* gy, const int32_t* gx, int32_t K, int32_t H, int32_t W, int32_t Hg,
                                   int32_t Wg, int64_t* counts, int64_t* gt_area, double* iou, double* iop, void* stream) {
  LL_CHECK(segs && gt && gy && gx && counts && gt_area && iou && iop && K > 0 && H > 0 && W > 0 && Hg > 0 && Wg > 0, "mask_targets: bad arguments");
  const long n = (long)H * W;
  const unsigned bx = (unsigned)((n + 256 * 16 - 1) / (256 * 16));
  LL_LAUNCH_KERNEL(mask_targets_kernel, dim3(bx < 1 ? 1 : (bx > 256 ? 256 : bx), (unsigned)K), dim3(256), 0, (hipStream_t)stream, segs, gt, gy, gx, H, W, Wg,
                     (unsigned long long*)counts, (unsigned long long*)gt_area);
  LL_LAUNCH_KERNEL(targets_finalize_kernel, dim3((unsigned)((K + 63) / 64)), dim3(64), 0, (hipStream_t)stream, (const unsigned long long*)counts,
                     (const unsigned long long*)gt_area, iou, iop, K);
  LL_LAUNCH_CHECK("mask_targets");
  return LLMSEG_OK;
}

extern "C" int llmseg_resize_aa(const uint8_t* segs, void* out, int32_t K, int32_t H, int32_t W, int32_t out_size, const int32_t* y0, const int32_t* ny,
                                const double* wy, const int32_t* x0, const int32_t* nx, const double* wx, int32_t taps, void* stream) {
  LL_CHECK(segs && out && y0 && ny && wy && x0 && nx && wx && K > 0 && H > 0 && W > 0 && out_size > 0 && taps > 0, "resize_aa: bad arguments");
  LL_LAUNCH_KERNEL(resize_aa_kernel, dim3((unsigned)((out_size + 255) / 256), (unsigned)out_size, (unsigned)K), dim3(256), 0, (hipStream_t)stream, segs,
                     (bf16_t*)out, H, W, out_size, y0, ny, wy, x0, nx, wx, taps);
  LL_LAUNCH_CHECK("resize_aa");
  return LLMSEG_OK;
}
