// Shared device helpers for the gfx950 kernels (wave64, bf16 storage as uint16).
#pragma once
#include "llmseg_hip.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
// two fp32 -> packed bf16x2 with ONE v_cvt_pk_bf16_f32 (hardware RNE); lo goes to bits 0..15
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// fp32 -> bf16 -> fp32 (hardware RNE): what a value is after a bf16 store + load
__device__ __forceinline__ float round_bf(float x) { return __uint_as_float(pack2bf(x, 0.f) << 16); }
// rotate-half RoPE of the pair (a, b) = (x[d], x[d + head_dim/2]) with cos c, sin s: ONE explicit fma form, shared by llmseg_rope, the
// decode-step kernels and the fused epilogues (attention backward store, q|k|v GEMM), so that the fused and unfused routes agree bit for bit
__device__ __forceinline__ float rope_lo(float a, float b, float c, float s) { return fmaf(a, c, -(b * s)); }
__device__ __forceinline__ float rope_hi(float a, float b, float c, float s) { return fmaf(b, c, a * s); }

// HF LlamaMLP act(gate) * up and its gradient, per element: ONE form shared by llmseg_swiglu / llmseg_swiglu_bwd and the GEMM's fused epilogues
__device__ __forceinline__ float swiglu_fwd1(float g, float u) { return g / (1.f + __expf(-g)) * u; }
// d(gate) = d * up * silu'(gate), d(up) = d * silu(gate)
__device__ __forceinline__ void swiglu_bwd1(float d, float g, float u, float& og, float& ou) {
  const float sg = 1.f / (1.f + __expf(-g));
  ou = d * g * sg;
  og = d * u * sg * (1.f + g * (1.f - sg));
}

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
}

// epilogue / pointwise activations (LLMSEG_ACT_* of llmseg_hip.h)
__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case LLMSEG_ACT_RELU: return fmaxf(v, 0.f);
    case LLMSEG_ACT_GELU: {   // exact (erf) GELU; erf via Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7 (far below bf16 resolution)
      const float z = fabsf(v) * 0.70710678118654752f;
      const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
      const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
      const float erfa = 1.f - poly * __expf(-z * z);
      return 0.5f * v * (1.f + copysignf(erfa, v));
    }
    case LLMSEG_ACT_QUICKGELU: return v * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v));
    case LLMSEG_ACT_SILU: return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
    case LLMSEG_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.f + __expf(-v));
    default: return v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` is >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

// One logits row of a cross-entropy kernel, read ONCE: the row (V <= 32768 bf16, V % 4 == 0, 8-byte aligned) lives in registers as 32 x
// 8-byte quads per thread of a 256-thread workgroup (the round-2 kernels walked the row twice / three times with 2-byte loads: counters
// showed 2.0 x the algorithmic bytes at 1.0-1.5 TB/s).  -> row max and sum of exp(x - max), block-wide.
struct CeRow {
  uint2 q[32];
  __device__ __forceinline__ void load(const bf16_t* __restrict__ row, long V) {
    const long nq = V >> 2;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const long k = threadIdx.x + 256L * i;
      q[i] = k < nq ? *reinterpret_cast<const uint2*>(row + 4 * k) : make_uint2(0xff80ff80u, 0xff80ff80u);     // -inf bf16 pairs: exp() = 0
    }
  }
  __device__ __forceinline__ void stats(float* red, float& mx, float& s) const {
    float m = -1e30f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      m = fmaxf(m, fmaxf(__uint_as_float(q[i].x << 16), __uint_as_float(q[i].x & 0xffff0000u)));
      m = fmaxf(m, fmaxf(__uint_as_float(q[i].y << 16), __uint_as_float(q[i].y & 0xffff0000u)));
    }
    mx = block_max(m, red);
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i)
      t += __expf(__uint_as_float(q[i].x << 16) - mx) + __expf(__uint_as_float(q[i].x & 0xffff0000u) - mx) +
           __expf(__uint_as_float(q[i].y << 16) - mx) + __expf(__uint_as_float(q[i].y & 0xffff0000u) - mx);
    s = block_sum(t, red);
  }
};
__device__ __forceinline__ bool ce_row_fast(const bf16_t* row, long V, long ldl) {
  return V <= 32768 && (V & 3) == 0 && (ldl & 3) == 0 && (((uintptr_t)row) & 7) == 0 && blockDim.x == 256;
}

// ---- counter-based RNG for LoRA dropout (peft Linear: dropout(p = 0.05) on the LoRA branch input, reference training.py:91,218-226) ----
// Philox4x32-10 (Salmon et al. 2011): counter = (idx_lo, idx_hi, stream, offset), key = (seed_lo, seed_hi).  One call yields the
// keep bits of EIGHT consecutive elements: element e = 8 * idx + j uses the 16-bit field j of the 128-bit output (field j = bits
// 16 (j & 1) .. +15 of word j >> 1) and is KEPT when field >= drop_thr (drop_thr = round(p * 65536)); kept values are scaled by
// 65536 / (65536 - drop_thr).  Nothing is stored: forward and backward regenerate the mask from (seed, offset, stream, idx).
// rng_state points at device memory {seed, offset} (uint64 each) so that a replayed hipGraph sees a fresh offset every micro-step.
struct Philox8 { uint32_t w[4]; };
__device__ __forceinline__ Philox8 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox8{{c0, c1, c2, c3}};
}
// multiplies f[0..7] (elements 8 idx .. 8 idx + 7) by the dropout mask * scale
// off_add: added to the stream's offset (the segment index of llmseg_dropout.seg_rows; 0 without segments)
__device__ __forceinline__ void dropout8(float* f, unsigned long idx, uint32_t stream, const unsigned long* rng, uint32_t thr, float scale, uint32_t off_add = 0) {
  const unsigned long seed = rng[0], off = rng[1] + off_add;
  const Philox8 r = philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), stream, (uint32_t)off, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t field = (r.w[j >> 1] >> (16 * (j & 1))) & 0xffffu;
    f[j] = field >= thr ? f[j] * scale : 0.f;
  }
}


// ---- fixed-order reductions (round 4) ------------------------------------------------------------------------------------------------
// No kernel of the library adds floating-point numbers with atomics: a sum whose terms come from several workgroups is written as
// per-workgroup PARTIALS into caller-owned scratch (`workspace`) and folded by one of the two kernels below in a fixed order, so a
// training step is bit-reproducible (the reference's resume contract, training.py:404-421,460-477: same weights + optimizer state ->
// the same continuation; with fp32 atomicAdd two runs from the same state differed in isolated elements).
// (1) wide outputs, few partials: out[i] += sum_{p < P} part[p * pstride + i], p ascending, one thread per output element; the n outputs are
//     up to four equal segments of n0 elements, each with its own destination (NULL: skipped).
struct FoldOut { float* p[4]; };
static __global__ __launch_bounds__(256) void fold_slices_kernel(const float* __restrict__ part, int P, long pstride, long n, long n0, FoldOut outs) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long seg = i / n0;                       // outputs [seg n0, (seg + 1) n0) belong to outs.p[seg] (<= 4 equal segments; NULL = skipped)
    float* o = outs.p[seg];
    if (!o) continue;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += part[(long)p * pstride + i];
    o[i - seg * n0] += s;
  }
}
// (2) few outputs (n = gridDim.x), many partials: part[p * n + i]; thread t adds p = t, t + 256, ... in order, then a fixed shuffle /
//     LDS tree over the 256 threads; out[i] += scale * sum.
static __global__ __launch_bounds__(256) void fold_column_kernel(const float* __restrict__ part, long P, float* __restrict__ out, float scale) {
  __shared__ float red[16];
  const long n = gridDim.x, i = blockIdx.x;
  float s = 0.f;
  for (long p = threadIdx.x; p < P; p += 256) s += part[p * n + i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[i] += scale * s;
}
static inline unsigned fold_grid(long n) { const long g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g)); }

// every kernel launch of the library goes through this macro: llmseg_launch_count() lets a benchmark report launches per micro-step
void llmseg_count_launch();
#define LL_LAUNCH_KERNEL(...)              \
  do {                                     \
    llmseg_count_launch();                 \
    hipLaunchKernelGGL(__VA_ARGS__);       \
  } while (0)

void llmseg_set_error(const char* fmt, ...);
#define LL_CHECK(cond, ...)                 \
  do {                                      \
    if (!(cond)) {                          \
      llmseg_set_error(__VA_ARGS__);        \
      return LLMSEG_EINVAL;                 \
    }                                       \
  } while (0)
#define LL_LAUNCH_CHECK(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      llmseg_set_error("%s launch failed: %s", name, hipGetErrorString(e__)); \
      return LLMSEG_ELAUNCH;                                               \
    }                                                                      \
  } while (0)
