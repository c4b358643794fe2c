// Proposal decode + training-target computation on the device: what the reference does per sample on CPU
// workers -- pycocotools RLE decode (utils/sam_mask_reader.py:69-113), zero-pad to square + antialiased bilinear resize to 256 x 256
// (utils/reason_seg_dataset.py:166-173), IoU / IoP of every proposal against the ground truth (utils/utils.py:174-272) -- as three
// HBM-bound kernels on uint8 masks.  Integer results (pixels, intersections, areas) are exact; the IoU / IoP divisions are IEEE double
// divisions of those integers, i.e. bit-identical to numpy's.  All index tables (nearest-neighbour source rows / columns, resampling
// taps and weights) are computed on the host in float64 with the library formulas and passed in, so no device floating-point
// contraction can move a sample point.
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

}  // namespace

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
