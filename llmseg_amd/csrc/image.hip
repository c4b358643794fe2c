// Image-side pieces of SAM "everything" mode beyond the default single crop (SURVEY.md 8f N1 remainder), all HBM / latency-bound byte work:
//
//   llmseg_image_resize_u8     `SamPredictor.set_image` -> `ResizeLongestSide.apply_image` (model/segment_anything/predictor.py:34-60,
//                              utils/transforms.py:27-35): Pillow's 8-bit BILINEAR `Image.resize`, bit for bit.  The crop of a crop layer
//                              (automatic_mask_generator.py:254-257) is an origin pointer + a row stride: it is never copied.
//   llmseg_sam_preprocess      `Sam.preprocess` (modeling/sam.py:174-186): (x - mean) / std, zero-padded to the square frame, bf16 CHW.
//   llmseg_mask_small_regions  `remove_small_regions` (utils/amg.py:267-291) in "holes" then "islands" mode as
//                              `postprocess_small_regions` calls it (automatic_mask_generator.py:326-372): 8-connected components by
//                              union-find on the pixel grid instead of cv2.connectedComponentsWithStats.
//   llmseg_mask_boxes          `batched_mask_to_box` (utils/amg.py:303-346) + area, on uint8 masks.
//
// Pillow's resampling (src/libImaging/Resample.c): per output index the taps are triangle weights around center = (xx + 0.5) * scale over
// [center - support, center + support), support = max(scale, 1), normalised in double and rounded to 22-bit fixed point; a pass is
// out = clip8((2^21 + sum in * k) >> 22); horizontal pass first (uint8 intermediate), then vertical.  The tap table is computed on the device
// in double with the SAME operation order and no fused multiply-adds (a contraction changes the rounding of w * 2^22 + 0.5).
#include "common.h"
#include "llmseg_hip.h"

namespace {

constexpr int PIL_BITS = 32 - 8 - 2;
constexpr int MAX_TAPS = 64;

// table row xx: [0] = first input index, [1] = tap count, [2 ..] = taps (ksize of them, zero-filled)
__global__ __launch_bounds__(256) void pil_taps_kernel(int32_t* __restrict__ tab, int in_size, int out_size, int ksize) {
#pragma clang fp contract(off)
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= out_size) return;
  const double scale = __ddiv_rn((double)in_size, (double)out_size);
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = filterscale;                       // bilinear: 1.0 * filterscale
  const double ss = __ddiv_rn(1.0, filterscale);
  const double center = __dmul_rn(__dadd_rn((double)xx, 0.5), scale);
  int lo = (int)__dadd_rn(__dsub_rn(center, support), 0.5);
  if (lo < 0) lo = 0;
  int hi = (int)__dadd_rn(__dadd_rn(center, support), 0.5);
  if (hi > in_size) hi = in_size;
  const int n = hi - lo;
  double w[MAX_TAPS];
  double ww = 0.0;
  for (int x = 0; x < n; ++x) {
    double a = __dmul_rn(__dadd_rn(__dsub_rn((double)(x + lo), center), 0.5), ss);
    if (a < 0.0) a = -a;
    const double v = a < 1.0 ? __dsub_rn(1.0, a) : 0.0;
    w[x] = v;
    ww = __dadd_rn(ww, v);
  }
  int32_t* row = tab + (long)xx * (ksize + 2);
  row[0] = lo; row[1] = n;
  for (int x = 0; x < ksize; ++x) {
    int k = 0;
    if (x < n) {
      const double v = ww != 0.0 ? __ddiv_rn(w[x], ww) : w[x];
      const double f = __dmul_rn(v, (double)(1 << PIL_BITS));
      k = v < 0.0 ? (int)__dadd_rn(-0.5, f) : (int)__dadd_rn(0.5, f);
    }
    row[2 + x] = k;
  }
}

// one resampling pass over interleaved 8-bit pixels: out[o][i][c] = clip8(sum_t in[lo_o + t][i][c] * k_o[t]); `o` is the resampled axis.
// HORIZ: o = column (in/out rows are the same), else o = row.  One thread per output byte; consecutive threads = consecutive bytes of an output row.
template <bool HORIZ>
__global__ __launch_bounds__(256) void pil_pass_kernel(const uint8_t* __restrict__ in, long in_row_stride, uint8_t* __restrict__ out, long out_row_stride,
                                                       const int32_t* __restrict__ tab, int ksize, int out_h, int out_w, int ch) {
  const int xb = blockIdx.x * blockDim.x + threadIdx.x;      // byte within the output row
  const int y = blockIdx.y;
  if (xb >= out_w * ch) return;
  const int x = xb / ch, c = xb - x * ch;
  const int32_t* row = tab + (long)(HORIZ ? x : y) * (ksize + 2);
  const int lo = row[0], n = row[1];
  int acc = 1 << (PIL_BITS - 1);
  if (HORIZ) {
    const uint8_t* src = in + (long)y * in_row_stride + (long)lo * ch + c;
    for (int t = 0; t < n; ++t) acc += (int)src[(long)t * ch] * row[2 + t];
  } else {
    const uint8_t* src = in + (long)lo * in_row_stride + xb;
    for (int t = 0; t < n; ++t) acc += (int)src[(long)t * in_row_stride] * row[2 + t];
  }
  acc >>= PIL_BITS;
  out[(long)y * out_row_stride + xb] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
}

struct Norm3 { float mean[3], inv_std_is_div[3]; };

// out bf16 [3][S][S]: (x - mean) / std inside [h][w], zero elsewhere (F.pad after the normalisation)
__global__ __launch_bounds__(256) void sam_preprocess_kernel(const uint8_t* __restrict__ in, bf16_t* __restrict__ out, int h, int w, int S, Norm3 nm) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= S) return;
  const bool inside = y < h && x < w;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = 0.f;
    if (inside) v = ((float)in[((long)y * w + x) * 3 + c] - nm.mean[c]) / nm.inv_std_is_div[c];
    out[((long)c * S + y) * S + x] = f2bf(v);
  }
}

// ---- 8-connected components on [K][H][W] uint8 masks: union-find with the smallest linear index as the root ----
__device__ __forceinline__ int cc_find(const int* L, int a) {
  int p = __atomic_load_n(L + a, __ATOMIC_RELAXED);
  while (p != a) { a = p; p = __atomic_load_n(L + a, __ATOMIC_RELAXED); }
  return a;
}
__device__ __forceinline__ void cc_union(int* L, int a, int b) {
  while (true) {
    a = cc_find(L, a); b = cc_find(L, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }             // a > b: hang a under b
    const int old = atomicMin(L + a, b);
    if (old == a) return;
    a = old;                                                 // someone re-rooted a meanwhile: merge its new parent with b
  }
}

// working foreground: HOLES ? mask == 0 : mask != 0
template <bool HOLES>
__global__ __launch_bounds__(256) void cc_init_kernel(const uint8_t* __restrict__ m, int* __restrict__ L, int* __restrict__ sz, long n_img) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_img) return;
  const long g = (long)blockIdx.y * n_img + i;
  const bool fg = HOLES ? m[g] == 0 : m[g] != 0;
  L[g] = fg ? (int)i : -1;
  sz[g] = 0;
}
__global__ __launch_bounds__(256) void cc_merge_kernel(int* Lall, int H, int W) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  int* L = Lall + (long)blockIdx.z * H * W;
  const int p = y * W + x;
  if (L[p] < 0) return;                                       // the sign of an entry never changes
  if (x > 0 && L[p - 1] >= 0) cc_union(L, p, p - 1);
  if (y > 0) {
    const int q = p - W;
    if (L[q] >= 0) cc_union(L, p, q);
    if (x > 0 && L[q - 1] >= 0) cc_union(L, p, q - 1);
    if (x + 1 < W && L[q + 1] >= 0) cc_union(L, p, q + 1);
  }
}
__global__ __launch_bounds__(256) void cc_count_kernel(int* Lall, int* szall, long n_img) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_img) return;
  int* L = Lall + (long)blockIdx.y * n_img;
  if (L[i] < 0) return;
  const int r = cc_find(L, (int)i);
  __atomic_store_n(L + i, r, __ATOMIC_RELAXED);               // path compression: every stored value stays an ancestor
  atomicAdd(szall + (long)blockIdx.y * n_img + r, 1);
}
// per mask: st[0] = components below the threshold, st[1] = components at / above it, best = (size << 32) | ~root of the largest (first in raster order on ties)
__global__ __launch_bounds__(256) void cc_stats_kernel(const int* __restrict__ Lall, const int* __restrict__ szall, long n_img, int min_area, int* __restrict__ st,
                                                       unsigned long long* __restrict__ best) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_img) return;
  const long g = (long)blockIdx.y * n_img + i;
  if (Lall[g] != (int)i) return;                              // roots only
  const int s = szall[g];
  atomicAdd(st + 2 * blockIdx.y + (s < min_area ? 0 : 1), 1);
  atomicMax(best + blockIdx.y, ((unsigned long long)(unsigned)s << 32) | (0xffffffffu - (unsigned)i));
}
template <bool HOLES>
__global__ __launch_bounds__(256) void cc_apply_kernel(uint8_t* __restrict__ m, const int* __restrict__ Lall, const int* __restrict__ szall, long n_img, int min_area,
                                                       const int* __restrict__ st, const unsigned long long* __restrict__ best, uint8_t* __restrict__ changed) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (st[2 * k] == 0) return;                                 // no small component: the mask is returned as it is, changed = False
  if (i == 0) changed[k] = 1;
  if (i >= n_img) return;
  const long g = (long)k * n_img + i;
  const int r = Lall[g];
  if (r < 0) return;                                          // not working foreground: holes leave the mask's own pixels, islands leave background
  const bool small = szall[(long)k * n_img + r] < min_area;
  if (HOLES) {
    if (small) m[g] = 1;                                      // mask = isin(regions, [0] + small)
  } else {
    bool keep = !small;
    if (st[2 * k + 1] == 0) keep = (unsigned)r == 0xffffffffu - (unsigned)(best[k] & 0xffffffffu);   // every region is small: keep the largest
    if (!keep) m[g] = 0;
  }
}

// box[k] = {min x, min y, max x, max y} (zeros when empty), area[k]
__global__ __launch_bounds__(256) void mask_boxes_kernel(const uint8_t* __restrict__ m, int H, int W, int* __restrict__ acc) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
  const bool on = x < W && m[((long)k * H + y) * W + x] != 0;
  const unsigned long long b = __ballot(on);
  if (b == 0) return;
  const int lane = threadIdx.x & 63;
  if (lane == 0) {
    const int x0 = x + __builtin_ctzll(b), x1 = x + 63 - __builtin_clzll(b);
    int* a = acc + 5 * k;
    atomicMin(a + 0, x0); atomicMin(a + 1, y); atomicMax(a + 2, x1); atomicMax(a + 3, y); atomicAdd(a + 4, __builtin_popcountll(b));
  }
}
__global__ void mask_boxes_init_kernel(int* acc, int K) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  acc[5 * k] = 0x7fffffff; acc[5 * k + 1] = 0x7fffffff; acc[5 * k + 2] = -1; acc[5 * k + 3] = -1; acc[5 * k + 4] = 0;
}
__global__ void mask_boxes_finish_kernel(const int* acc, int K, int32_t* boxes, int32_t* areas) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const bool empty = acc[5 * k + 4] == 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) boxes[4 * k + j] = empty ? 0 : acc[5 * k + j];
  if (areas) areas[k] = acc[5 * k + 4];
}

inline long align256(long v) { return (v + 255) & ~255L; }
inline int pil_ksize(int in_size, int out_size) {
  const double scale = (double)in_size / (double)out_size;
  const double support = scale < 1.0 ? 1.0 : scale;
  return (int)ceil(support) * 2 + 1;
}

}  // namespace

extern "C" int64_t llmseg_image_resize_workspace(int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w, int32_t channels) {
  if (in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || channels <= 0) return -1;
  const long tw = align256((long)out_w * (pil_ksize(in_w, out_w) + 2) * 4), th = align256((long)out_h * (pil_ksize(in_h, out_h) + 2) * 4);
  return tw + th + align256((long)in_h * out_w * channels);
}

extern "C" int llmseg_image_resize_u8(const uint8_t* in, int64_t in_row_stride, uint8_t* out, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                                      int32_t channels, void* workspace, int64_t workspace_bytes, void* stream) {
  LL_CHECK(in && out && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0 && channels > 0 && channels <= 4 && in_row_stride >= (int64_t)in_w * channels,
           "image_resize: bad arguments");
  const int kw = pil_ksize(in_w, out_w), kh = pil_ksize(in_h, out_h);
  LL_CHECK(kw <= MAX_TAPS && kh <= MAX_TAPS, "image_resize: down-scaling by more than %dx is not supported", (MAX_TAPS - 1) / 2);
  LL_CHECK(out_h < 65536 && in_h < 65536, "image_resize: grid limit");
  LL_CHECK(workspace && workspace_bytes >= llmseg_image_resize_workspace(in_h, in_w, out_h, out_w, channels), "image_resize: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  int32_t* tabw = (int32_t*)ws;
  int32_t* tabh = (int32_t*)(ws + align256((long)out_w * (kw + 2) * 4));
  uint8_t* mid = (uint8_t*)((char*)tabh + align256((long)out_h * (kh + 2) * 4));
  const bool do_h = out_w != in_w, do_v = out_h != in_h;
  const uint8_t* src = in;
  long src_stride = in_row_stride;
  if (!do_h && !do_v) {                                      // Pillow returns a copy
    LL_CHECK(hipMemcpy2DAsync(out, (size_t)out_w * channels, in, (size_t)in_row_stride, (size_t)in_w * channels, (size_t)in_h, hipMemcpyDeviceToDevice, s) == hipSuccess,
             "image_resize: copy failed");
    return LLMSEG_OK;
  }
  if (do_h) {
    uint8_t* dst = do_v ? mid : out;
    LL_LAUNCH_KERNEL(pil_taps_kernel, dim3((unsigned)((out_w + 255) / 256)), dim3(256), 0, s, tabw, in_w, out_w, kw);
    LL_LAUNCH_KERNEL(pil_pass_kernel<true>, dim3((unsigned)((out_w * channels + 255) / 256), (unsigned)in_h), dim3(256), 0, s, src, src_stride, dst, (long)out_w * channels,
                       tabw, kw, in_h, out_w, channels);
    src = dst; src_stride = (long)out_w * channels;
  }
  if (do_v) {
    LL_LAUNCH_KERNEL(pil_taps_kernel, dim3((unsigned)((out_h + 255) / 256)), dim3(256), 0, s, tabh, in_h, out_h, kh);
    LL_LAUNCH_KERNEL(pil_pass_kernel<false>, dim3((unsigned)((out_w * channels + 255) / 256), (unsigned)out_h), dim3(256), 0, s, src, src_stride, out, (long)out_w * channels,
                       tabh, kh, out_h, out_w, channels);
  }
  LL_LAUNCH_CHECK("image_resize");
  return LLMSEG_OK;
}

extern "C" int llmseg_sam_preprocess(const uint8_t* in, void* out, int32_t h, int32_t w, int32_t img_size, const float* mean, const float* std_, void* stream) {
  LL_CHECK(in && out && mean && std_ && h > 0 && w > 0 && h <= img_size && w <= img_size && img_size < 65536, "sam_preprocess: bad arguments");
  Norm3 nm;
  for (int c = 0; c < 3; ++c) { nm.mean[c] = mean[c]; nm.inv_std_is_div[c] = std_[c]; }
  LL_LAUNCH_KERNEL(sam_preprocess_kernel, dim3((unsigned)((img_size + 255) / 256), (unsigned)img_size), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, h, w, img_size, nm);
  LL_LAUNCH_CHECK("sam_preprocess");
  return LLMSEG_OK;
}

extern "C" int64_t llmseg_mask_small_regions_workspace(int32_t K, int32_t H, int32_t W) {
  if (K <= 0 || H <= 0 || W <= 0) return -1;
  return 2 * align256((long)K * H * W * 4) + align256((long)K * 8) + align256((long)K * 8);
}

extern "C" int llmseg_mask_small_regions(uint8_t* masks, int32_t K, int32_t H, int32_t W, int32_t min_area, uint8_t* changed, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  LL_CHECK(masks && changed && K > 0 && K < 65536 && H > 0 && W > 0 && H < 65536 && (long)H * W < 0x7fffffffL && min_area >= 0, "mask_small_regions: bad arguments");
  LL_CHECK(workspace && workspace_bytes >= llmseg_mask_small_regions_workspace(K, H, W), "mask_small_regions: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const long n = (long)H * W;
  char* ws = (char*)workspace;
  int* L = (int*)ws;
  int* sz = (int*)(ws + align256((long)K * n * 4));
  int* st = (int*)((char*)sz + align256((long)K * n * 4));
  unsigned long long* best = (unsigned long long*)((char*)st + align256((long)K * 8));
  const dim3 flat((unsigned)((n + 255) / 256), (unsigned)K), tile((unsigned)((W + 255) / 256), (unsigned)H, (unsigned)K);
  LL_CHECK(hipMemsetAsync(changed, 0, (size_t)K, s) == hipSuccess, "mask_small_regions: memset failed");
  for (int mode = 0; mode < 2; ++mode) {                     // "holes", then "islands" on its result (automatic_mask_generator.py:347-350)
    LL_CHECK(hipMemsetAsync(st, 0, (size_t)(align256((long)K * 8) + align256((long)K * 8)), s) == hipSuccess, "mask_small_regions: memset failed");
    if (mode == 0) LL_LAUNCH_KERNEL(cc_init_kernel<true>, flat, dim3(256), 0, s, masks, L, sz, n);
    else LL_LAUNCH_KERNEL(cc_init_kernel<false>, flat, dim3(256), 0, s, masks, L, sz, n);
    LL_LAUNCH_KERNEL(cc_merge_kernel, tile, dim3(256), 0, s, L, H, W);
    LL_LAUNCH_KERNEL(cc_count_kernel, flat, dim3(256), 0, s, L, sz, n);
    LL_LAUNCH_KERNEL(cc_stats_kernel, flat, dim3(256), 0, s, L, sz, n, min_area, st, best);
    if (mode == 0) LL_LAUNCH_KERNEL(cc_apply_kernel<true>, flat, dim3(256), 0, s, masks, L, sz, n, min_area, st, best, changed);
    else LL_LAUNCH_KERNEL(cc_apply_kernel<false>, flat, dim3(256), 0, s, masks, L, sz, n, min_area, st, best, changed);
  }
  LL_LAUNCH_CHECK("mask_small_regions");
  return LLMSEG_OK;
}

extern "C" int llmseg_mask_boxes(const uint8_t* masks, int32_t K, int32_t H, int32_t W, int32_t* boxes, int32_t* areas, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
  LL_CHECK(masks && boxes && K > 0 && K < 65536 && H > 0 && H < 65536 && W > 0, "mask_boxes: bad arguments");
  LL_CHECK(workspace && workspace_bytes >= (int64_t)K * 20, "mask_boxes: workspace of 20 bytes per mask");
  hipStream_t s = (hipStream_t)stream;
  int* acc = (int*)workspace;
  LL_LAUNCH_KERNEL(mask_boxes_init_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, s, acc, K);
  LL_LAUNCH_KERNEL(mask_boxes_kernel, dim3((unsigned)((W + 255) / 256), (unsigned)H, (unsigned)K), dim3(256), 0, s, masks, H, W, acc);
  LL_LAUNCH_KERNEL(mask_boxes_finish_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, s, acc, K, boxes, areas);
  LL_LAUNCH_CHECK("mask_boxes");
  return LLMSEG_OK;
}
