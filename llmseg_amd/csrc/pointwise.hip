// HBM-bound glue kernels for the LLM-Seg hot path on gfx950: row norms, RoPE, SwiGLU, positional adds, im2col,
// LLaVA embedding splice, row gather.  All are streaming kernels: 16-byte (8 x bf16) accesses per lane, wave64
// shuffles for row statistics, fp32 math, one rounding to bf16 on store (see include/llmseg_hip.h for the
// reference ops each one replaces).
#include <cstdlib>
#include "common.h"
#include "llmseg_hip.h"

namespace {

// ---- LayerNorm / RMSNorm: one wave per row.  CPL > 0: the row (<= 64*CPL 16-byte chunks) is read ONCE and kept in registers
// for the three reductions / the normalise pass; CPL == 0: generic multi-pass fallback for very wide rows. ------------------
template <int CPL>
__global__ __launch_bounds__(256) void norm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                  const bf16_t* __restrict__ bias, bf16_t* __restrict__ y, long rows, int cols,
                                                  long ldx, long ldy, float eps, int rms, const int32_t* __restrict__ row_map) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * ldx;
  const int nch = cols >> 3;
  constexpr int NR = CPL > 0 ? CPL : 1;
  uint4 xc[NR];
  float f[8];
  float s = 0.f, v = 0.f;
  if (CPL > 0) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int c = lane + 64 * i;
      xc[i] = c < nch ? *reinterpret_cast<const uint4*>(xr + c * 8) : make_uint4(0, 0, 0, 0);
    }
  }
  if (!rms) {
    if (CPL > 0) {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        unpack8(xc[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[e];
      }
    } else {
      for (int c = lane; c < nch; c += 64) {
        unpack8(*reinterpret_cast<const uint4*>(xr + c * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[e];
      }
    }
    s = wave_sum(s);
  }
  const float mean = rms ? 0.f : s / (float)cols;
  if (CPL > 0) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      if (lane + 64 * i < nch) {
        unpack8(xc[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; v += d * d; }
      }
    }
  } else {
    for (int c = lane; c < nch; c += 64) {
      unpack8(*reinterpret_cast<const uint4*>(xr + c * 8), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; v += d * d; }
    }
  }
  v = wave_sum(v);
  const float rstd = rsqrtf(v / (float)cols + eps);
  long orow = row;
  if (row_map) { orow = row_map[row]; if (orow < 0) return; }
  bf16_t* yr = y + orow * ldy;
  auto emit = [&](int c, const uint4& xv) {
    float g[8], o[8], ff[8];
    unpack8(xv, ff);
    unpack8(*reinterpret_cast<const uint4*>(w + c * 8), g);
    if (rms) {
      // HF LlamaRMSNorm rounds the normalised value to the activation dtype before the weight multiply
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = g[e] * bf2f(f2bf(ff[e] * rstd));
    } else {
      float bb[8];
      if (bias) unpack8(*reinterpret_cast<const uint4*>(bias + c * 8), bb);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (ff[e] - mean) * rstd * g[e] + (bias ? bb[e] : 0.f);
    }
    *reinterpret_cast<uint4*>(yr + c * 8) = pack8(o);
  };
  if (CPL > 0) {
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (lane + 64 * i < nch) emit(lane + 64 * i, xc[i]);
  } else {
    for (int c = lane; c < nch; c += 64) emit(c, *reinterpret_cast<const uint4*>(xr + c * 8));
  }
}

// The same normalisation with a WORKGROUP per row (4 waves, CPT 16-byte chunks per thread, block-wide reductions through LDS): for short,
// wide activations -- Llama at two images per step is 638 rows x 4096 columns -- one wave per row puts 160 workgroups on 256 CUs and the
// kernel is bound by the latency of one wave's loads (9.8 us for 10 MB); a workgroup per row keeps 4 x the loads in flight.
template <int CPT>
__global__ __launch_bounds__(256) void norm_wg_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias,
                                                     bf16_t* __restrict__ y, long rows, int cols, long ldx, long ldy, float eps, int rms,
                                                     const int32_t* __restrict__ row_map) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const bf16_t* xr = x + row * ldx;
  const int nch = cols >> 3;
  uint4 xc[CPT];
  float f[8];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = threadIdx.x + 256 * i;
    xc[i] = c < nch ? *reinterpret_cast<const uint4*>(xr + c * 8) : make_uint4(0, 0, 0, 0);
  }
  float s = 0.f, v = 0.f;
  if (!rms) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      unpack8(xc[i], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += f[e];
    }
    s = block_sum(s, red);
  }
  const float mean = rms ? 0.f : s / (float)cols;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    if (threadIdx.x + 256 * i < nch) {
      unpack8(xc[i], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; v += d * d; }
    }
  }
  v = block_sum(v, red);
  const float rstd = rsqrtf(v / (float)cols + eps);
  long orow = row;
  if (row_map) { orow = row_map[row]; if (orow < 0) return; }
  bf16_t* yr = y + orow * ldy;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < nch) {
      float g[8], o[8];
      unpack8(xc[i], f);
      unpack8(*reinterpret_cast<const uint4*>(w + c * 8), g);
      if (rms) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = g[e] * bf2f(f2bf(f[e] * rstd));      // HF LlamaRMSNorm: round before the weight multiply
      } else {
        float bb[8];
        if (bias) unpack8(*reinterpret_cast<const uint4*>(bias + c * 8), bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f[e] - mean) * rstd * g[e] + (bias ? bb[e] : 0.f);
      }
      *reinterpret_cast<uint4*>(yr + c * 8) = pack8(o);
    }
  }
}

// ---- RoPE (rotate-half), in place ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ x, const float* __restrict__ cs, const float* __restrict__ sn,
                                                  long rows, long T, int heads, int hd, long ld) {
  const int hc = hd >> 4;                       // 8-wide chunks in half a head
  const long total = rows * heads * hc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % hc);
    const long rh = i / hc;
    const int h = (int)(rh % heads);
    const long row = rh / heads;
    const long pos = row % T;
    bf16_t* p1 = x + row * ld + (long)h * hd + c * 8;
    bf16_t* p2 = p1 + (hd >> 1);
    float a[8], b[8], o1[8], o2[8];
    unpack8(*reinterpret_cast<const uint4*>(p1), a);
    unpack8(*reinterpret_cast<const uint4*>(p2), b);
    const float* cp = cs + pos * (hd >> 1) + c * 8;
    const float* sp = sn + pos * (hd >> 1) + c * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      o1[e] = rope_lo(a[e], b[e], cp[e], sp[e]);
      o2[e] = rope_hi(a[e], b[e], cp[e], sp[e]);
    }
    *reinterpret_cast<uint4*>(p1) = pack8(o1);
    *reinterpret_cast<uint4*>(p2) = pack8(o2);
  }
}

// decode step: RoPE of the new token's q (in place) and k (into the cache) at the device-side position, v copied into the cache
__global__ __launch_bounds__(256) void rope_kv_append_kernel(bf16_t* __restrict__ qkv, long ld, const float* __restrict__ cs, const float* __restrict__ sn,
                                                            bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, long cstride, const int32_t* __restrict__ pos_dev,
                                                            long N, int heads, int hd) {
  const int hc = hd >> 4;
  const long D = (long)heads * hd, per = (long)heads * hc, total = N * 3 * per;
  const long pos = *pos_dev;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % hc);
    const int h = (int)((i / hc) % heads);
    const int sec = (int)((i / per) % 3);                 // 0 = q, 1 = k, 2 = v
    const long n = i / (3 * per);
    bf16_t* p1 = qkv + n * ld + sec * D + (long)h * hd + c * 8;
    bf16_t* p2 = p1 + (hd >> 1);
    const uint4 u1 = *reinterpret_cast<const uint4*>(p1), u2 = *reinterpret_cast<const uint4*>(p2);
    bf16_t* d1 = (sec == 1 ? kc : vc) + n * cstride + pos * D + (long)h * hd + c * 8;
    if (sec == 2) {
      *reinterpret_cast<uint4*>(d1) = u1;
      *reinterpret_cast<uint4*>(d1 + (hd >> 1)) = u2;
      continue;
    }
    float a[8], b[8], o1[8], o2[8];
    unpack8(u1, a);
    unpack8(u2, b);
    const float* cp = cs + pos * (hd >> 1) + c * 8;
    const float* sp = sn + pos * (hd >> 1) + c * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      o1[e] = rope_lo(a[e], b[e], cp[e], sp[e]);
      o2[e] = rope_hi(a[e], b[e], cp[e], sp[e]);
    }
    if (sec == 0) { *reinterpret_cast<uint4*>(p1) = pack8(o1); *reinterpret_cast<uint4*>(p2) = pack8(o2); }
    else { *reinterpret_cast<uint4*>(d1) = pack8(o1); *reinterpret_cast<uint4*>(d1 + (hd >> 1)) = pack8(o2); }
  }
}

// elementwise activation, in place or not (rows of 8-element chunks)
__global__ __launch_bounds__(256) void act_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n8, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float v[8];
    unpack8(reinterpret_cast<const uint4*>(x)[i], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], act);
    reinterpret_cast<uint4*>(y)[i] = pack8(v);
  }
}

__global__ __launch_bounds__(256) void swiglu_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ out, long rows, long I,
                                                    long ldgu, long ldo) {
  const long ich = I >> 3, total = rows * ich;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / ich, c = i % ich;
    float g[8], u[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(gu + row * ldgu + c * 8), g);
    unpack8(*reinterpret_cast<const uint4*>(gu + row * ldgu + I + c * 8), u);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = swiglu_fwd1(g[e], u[e]);
    *reinterpret_cast<uint4*>(out + row * ldo + c * 8) = pack8(o);
  }
}

__global__ __launch_bounds__(256) void add_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ add, bf16_t* __restrict__ y,
                                                      long rows, long cols, long add_rows) {
  const long nch = cols >> 3, total = rows * nch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / nch, c = i % nch;
    float a[8], b[8];
    unpack8(*reinterpret_cast<const uint4*>(x + row * cols + c * 8), a);
    unpack8(*reinterpret_cast<const uint4*>(add + (row % add_rows) * cols + c * 8), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += b[e];
    *reinterpret_cast<uint4*>(y + row * cols + c * 8) = pack8(a);
  }
}

// one thread per (patch row, channel, in-patch row): p contiguous pixels; tail threads zero-fill the K padding
__global__ __launch_bounds__(256) void patchify_kernel(const bf16_t* __restrict__ img, bf16_t* __restrict__ cols, int B, int H, int W, int p,
                                                      long ldo, long rows_per_img, long row_off) {
  const int gh = H / p, gw = W / p;
  const int runs = 3 * p + 1;                   // +1: padding run
  const long total = (long)B * gh * gw * runs;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int run = (int)(i % runs);
    const long patch = i / runs;
    const int px = (int)(patch % gw);
    const int py = (int)((patch / gw) % gh);
    const int b = (int)(patch / ((long)gw * gh));
    bf16_t* dst = cols + ((long)b * rows_per_img + row_off + (long)py * gw + px) * ldo;
    if (run == 3 * p) {
      for (long k = 3L * p * p; k < ldo; ++k) dst[k] = 0;
    } else {
      const int c = run / p, iy = run % p;
      const bf16_t* src = img + (((long)b * 3 + c) * H + (long)py * p + iy) * W + (long)px * p;
      dst += (long)c * p * p + (long)iy * p;
      for (int j = 0; j < p; ++j) dst[j] = src[j];
    }
  }
}

__global__ __launch_bounds__(256) void im2col3x3_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ cols, int B, int H, int W, int C) {
  const int cch = C >> 3;
  const long total = (long)B * H * W * 9 * cch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cch);
    long r = i / cch;
    const int tap = (int)(r % 9); r /= 9;
    const int xx = (int)(r % W); r /= W;
    const int yy = (int)(r % H);
    const int b = (int)(r / H);
    const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = *reinterpret_cast<const uint4*>(x + (((long)b * H + sy) * W + sx) * C + c * 8);
    *reinterpret_cast<uint4*>(cols + (((long)b * H + yy) * W + xx) * 9L * C + (long)tap * C + c * 8) = v;
  }
}

// one block per output position (n, t)
__global__ __launch_bounds__(256) void embed_splice_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ embed,
                                                          const bf16_t* __restrict__ feats, bf16_t* __restrict__ out, int N, int L, int P, int Hd,
                                                          long vocab, long fstride) {
  __shared__ int s_pos;
  const int Tn = L - 1 + P;
  const int n = blockIdx.x / Tn, t = blockIdx.x % Tn;
  if (threadIdx.x == 0) s_pos = L;
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += blockDim.x)
    if (ids[(long)n * L + i] == -200) atomicMin(&s_pos, i);
  __syncthreads();
  const int ip = s_pos;
  const bf16_t* src;
  if (t >= ip && t < ip + P) {
    src = feats + (long)n * fstride + (long)(t - ip) * Hd;
  } else {
    long id = ids[(long)n * L + (t < ip ? t : t - P + 1)];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    src = embed + id * Hd;
  }
  bf16_t* dst = out + ((long)n * Tn + t) * Hd;
  for (int c = threadIdx.x; c < (Hd >> 3); c += blockDim.x)
    *reinterpret_cast<uint4*>(dst + c * 8) = *reinterpret_cast<const uint4*>(src + c * 8);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ x, const int64_t* __restrict__ idx, bf16_t* __restrict__ out,
                                                         long n, long cols, long ldx) {
  const long nch = cols >> 3, total = n * nch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / nch, c = i % nch;
    *reinterpret_cast<uint4*>(out + r * cols + c * 8) = *reinterpret_cast<const uint4*>(x + idx[r] * ldx + c * 8);
  }
}

inline unsigned grid_for(long total) {
  long g = (total + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

#define AL16(p) ((((uintptr_t)(p)) & 15) == 0)

extern "C" int llmseg_norm(const void* x, const void* w, const void* b, void* y, int64_t rows, int64_t cols, int64_t ldx, int64_t ldy,
                           float eps, int rms, const int32_t* row_map, void* stream) {
  LL_CHECK(x && w && y && rows > 0 && cols > 0, "norm: bad arguments");
  LL_CHECK((cols & 7) == 0 && (ldx & 7) == 0 && (ldy & 7) == 0, "norm: cols/ld must be multiples of 8");
  LL_CHECK(AL16(x) && AL16(w) && AL16(y) && (b == nullptr || AL16(b)), "norm: pointers must be 16-byte aligned");
  const int cpl = (int)(((cols >> 3) + 63) / 64);
  static const long wg_max_rows = getenv("LLMSEG_NORM_WG_MAX") ? atol(getenv("LLMSEG_NORM_WG_MAX")) : 2048;      // A/B switch (tools/hbm_kernels.py)
  if (rows >= 64 && rows < wg_max_rows && cols >= 2048 && cols <= 8192) {    // short and wide: a workgroup per row (rows < 64: the decode path keeps its kernel)
    const int cpt = (int)(((cols >> 3) + 255) / 256);
#define LL_NORMW(C)                                                                                                                             \
  LL_LAUNCH_KERNEL(norm_wg_kernel<C>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, \
                   (bf16_t*)y, (long)rows, (int)cols, (long)ldx, (long)ldy, eps, rms, row_map)
    if (cpt <= 1) LL_NORMW(1); else if (cpt <= 2) LL_NORMW(2); else LL_NORMW(4);
#undef LL_NORMW
    LL_LAUNCH_CHECK("norm");
    return LLMSEG_OK;
  }
  const dim3 grid((unsigned)((rows + 3) / 4));
#define LL_NORM(C)                                                                                                                   \
  LL_LAUNCH_KERNEL(norm_kernel<C>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, \
                     (bf16_t*)y, (long)rows, (int)cols, (long)ldx, (long)ldy, eps, rms, row_map)
  if (cpl <= 1) LL_NORM(1); else if (cpl <= 2) LL_NORM(2); else if (cpl <= 3) LL_NORM(3); else if (cpl <= 4) LL_NORM(4);
  else if (cpl <= 8) LL_NORM(8); else if (cpl <= 16) LL_NORM(16); else LL_NORM(0);
#undef LL_NORM
  LL_LAUNCH_CHECK("norm");
  return LLMSEG_OK;
}

extern "C" int llmseg_rope(void* x, const float* cos, const float* sin, int64_t rows, int64_t T, int32_t heads, int32_t head_dim,
                           int64_t ld, void* stream) {
  LL_CHECK(x && cos && sin && rows > 0 && T > 0 && heads > 0, "rope: bad arguments");
  LL_CHECK((head_dim & 15) == 0 && (ld & 7) == 0 && AL16(x), "rope: head_dim %% 16 and 16-byte alignment required");
  const long total = rows * heads * (head_dim >> 4);
  LL_LAUNCH_KERNEL(rope_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, cos, sin, (long)rows, (long)T, heads,
                     head_dim, (long)ld);
  LL_LAUNCH_CHECK("rope");
  return LLMSEG_OK;
}

extern "C" int llmseg_rope_kv_append(void* qkv, int64_t ld, const float* cos, const float* sin, void* kcache, void* vcache, int64_t cache_stride_n,
                                     const int32_t* pos_dev, int64_t N, int32_t heads, int32_t head_dim, void* stream) {
  LL_CHECK(qkv && cos && sin && kcache && vcache && pos_dev && N > 0 && heads > 0, "rope_kv_append: bad arguments");
  LL_CHECK((head_dim & 15) == 0 && (ld & 7) == 0 && (cache_stride_n & 7) == 0 && AL16(qkv) && AL16(kcache) && AL16(vcache),
           "rope_kv_append: head_dim %% 16 and 16-byte alignment required");
  const long total = N * 3 * heads * (head_dim >> 4);
  LL_LAUNCH_KERNEL(rope_kv_append_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)qkv, (long)ld, cos, sin, (bf16_t*)kcache,
                     (bf16_t*)vcache, (long)cache_stride_n, pos_dev, (long)N, heads, head_dim);
  LL_LAUNCH_CHECK("rope_kv_append");
  return LLMSEG_OK;
}

extern "C" int llmseg_act(const void* x, void* y, int64_t n, int32_t act, void* stream) {
  LL_CHECK(x && y && n > 0 && (n & 7) == 0 && AL16(x) && AL16(y), "act: n %% 8 and 16-byte alignment required");
  LL_LAUNCH_KERNEL(act_kernel, dim3(grid_for(n >> 3)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, (long)(n >> 3), act);
  LL_LAUNCH_CHECK("act");
  return LLMSEG_OK;
}

extern "C" int llmseg_swiglu(const void* gu, void* out, int64_t rows, int64_t I, int64_t ldgu, int64_t ldo, void* stream) {
  LL_CHECK(gu && out && rows > 0 && I > 0 && (I & 7) == 0 && (ldgu & 7) == 0 && (ldo & 7) == 0 && AL16(gu) && AL16(out), "swiglu: bad arguments");
  LL_LAUNCH_KERNEL(swiglu_kernel, dim3(grid_for(rows * (I >> 3))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gu, (bf16_t*)out,
                     (long)rows, (long)I, (long)ldgu, (long)ldo);
  LL_LAUNCH_CHECK("swiglu");
  return LLMSEG_OK;
}

extern "C" int llmseg_add_rows(const void* x, const void* add, void* y, int64_t rows, int64_t cols, int64_t add_rows, void* stream) {
  LL_CHECK(x && add && y && rows > 0 && (cols & 7) == 0 && add_rows > 0 && AL16(x) && AL16(add) && AL16(y), "add_rows: bad arguments");
  LL_LAUNCH_KERNEL(add_rows_kernel, dim3(grid_for(rows * (cols >> 3))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)add,
                     (bf16_t*)y, (long)rows, (long)cols, (long)add_rows);
  LL_LAUNCH_CHECK("add_rows");
  return LLMSEG_OK;
}

extern "C" int llmseg_patchify(const void* img, void* cols, int32_t B, int32_t H, int32_t W, int32_t p, int64_t ldo, int64_t out_rows_per_img,
                               int64_t out_row_offset, void* stream) {
  LL_CHECK(img && cols && B > 0 && p > 0 && H % p == 0 && W % p == 0 && ldo >= 3L * p * p, "patchify: bad arguments");
  const long total = (long)B * (H / p) * (W / p) * (3 * p + 1);
  LL_LAUNCH_KERNEL(patchify_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)img, (bf16_t*)cols, B, H, W, p,
                     (long)ldo, (long)out_rows_per_img, (long)out_row_offset);
  LL_LAUNCH_CHECK("patchify");
  return LLMSEG_OK;
}

extern "C" int llmseg_im2col3x3(const void* x, void* cols, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  LL_CHECK(x && cols && B > 0 && H > 0 && W > 0 && (C & 7) == 0 && AL16(x) && AL16(cols), "im2col3x3: bad arguments");
  LL_LAUNCH_KERNEL(im2col3x3_kernel, dim3(grid_for((long)B * H * W * 9 * (C >> 3))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)cols, B, H, W, C);
  LL_LAUNCH_CHECK("im2col3x3");
  return LLMSEG_OK;
}

extern "C" int llmseg_embed_splice(const int64_t* ids, const void* embed, const void* img_feats, void* out, int32_t N, int32_t L, int32_t P,
                                   int32_t H, int64_t vocab, int64_t feats_stride_n, void* stream) {
  LL_CHECK(ids && embed && img_feats && out && N > 0 && L > 0 && P > 0 && (H & 7) == 0 && (feats_stride_n & 7) == 0 && AL16(embed) && AL16(img_feats) &&
           AL16(out), "embed_splice: bad arguments");
  LL_LAUNCH_KERNEL(embed_splice_kernel, dim3((unsigned)(N * (L - 1 + P))), dim3(256), 0, (hipStream_t)stream, ids, (const bf16_t*)embed,
                     (const bf16_t*)img_feats, (bf16_t*)out, N, L, P, H, (long)vocab, (long)feats_stride_n);
  LL_LAUNCH_CHECK("embed_splice");
  return LLMSEG_OK;
}

extern "C" int llmseg_gather_rows(const void* x, const int64_t* idx, void* out, int64_t n, int64_t cols, int64_t ldx, void* stream) {
  LL_CHECK(x && idx && out && n > 0 && (cols & 7) == 0 && (ldx & 7) == 0 && AL16(x) && AL16(out), "gather_rows: bad arguments");
  LL_LAUNCH_KERNEL(gather_rows_kernel, dim3(grid_for(n * (cols >> 3))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, idx, (bf16_t*)out,
                     (long)n, (long)cols, (long)ldx);
  LL_LAUNCH_CHECK("gather_rows");
  return LLMSEG_OK;
}
