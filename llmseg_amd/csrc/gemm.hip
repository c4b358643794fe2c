// bf16 MFMA GEMM for gfx950:  C[m][n] = epi(alpha * sum_k A[m][k] * W[n][k])   ("TN": both operands K-contiguous)
//
// Replaces every nn.Linear / 1x1-conv / patch-embed on the LLM-Seg hot path (see include/llmseg_hip.h).
//
// Three kernels, one epilogue (the MFMA "A" operand is the WEIGHT fragment and the "B" operand the ACTIVATION fragment, so a
// lane ends up holding 4 consecutive output columns of one output row: bias / activation / LayerScale / residual fuse into the
// epilogue, which bounces through LDS so that stores and residual loads are whole 128-byte lines):
//   Q  gemm_bf16_tn_pp_kernel    256 x 256 tile, 8 waves in two groups one barrier apart ("ping-pong"), half-tile LDS-DMA ring
//                                (buffer_load ... lds) with counted vmcnt, v_mfma_f32_16x16x32_bf16, one workgroup per CU.  Used when the tile count fills whole rounds of the CUs.
//   G  gemm_bf16_tn_glds_kernel  128 x 128 tile, 4 waves, one LDS buffer filled by LDS-DMA, 4 workgroups per CU hide each other's
//                                latency.  Used for everything else with K % 64 == 0.
//   R  gemm_bf16_tn_kernel       128 x 128 tile, global -> VGPR -> LDS staging (double-buffered); any K % 8 == 0, zero-filled K
//                                tail, and the transposed-operand layouts of the backward pass.
// LDS-DMA (global_load_lds_dwordx4) writes LDS linearly (wave-uniform base + lane*16), so the XOR swizzle that makes the
// ds_read_b128 fragment loads conflict-free on gfx950's 16-lane service groups is applied to the per-lane SOURCE address
// (lane -> LDS slot (row, cpos) -> global chunk cpos ^ ((row>>1)&7) of that row; the 8 lanes of a row still read one 128-byte
// line).  Workgroup ids are remapped so that each of the 8 XCDs (private L2s) walks a contiguous, M-grouped range of tiles.
// Kernels tried and dropped (numbers in DESIGN.md): 256 x 128 LDS-DMA tiles, a lock-step 256 x 256 two-stage kernel, a 4-stage
// BK = 32 ring, a persistent ping-pong with the next tile's DMA issued before the epilogue, K-half ping-pong phases, a 4-wave
// 128 x 128-per-wave kernel with in-wave fragment prefetch (LDS-DMA issue from the MFMA wave costs ~170 cycles per instruction).
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "llmseg_hip.h"

namespace {

constexpr int BN = 128, BK = 64, NT = 256;
constexpr int GROUP_M = 8;

struct GemmP {
  const bf16_t* A; const bf16_t* W; void* C;
  const bf16_t* bias; const bf16_t* gamma; const bf16_t* res;
  int M, N, K;
  long lda, ldw, ldc, ldr, sA, sW, sC, sA2, sW2, sC2;
  int batch1;
  float alpha;
  int act;
  int tiles_m, tiles_n;
  int c_vec, r_vec, b_vec;   // host-verified alignment for vector C stores / residual loads / bias+gamma loads
  const bf16_t* A2; const bf16_t* W2; long lda2, ldw2;   // optional extension K-tile (64 wide): C += alpha * A2 . W2^T (ping-pong kernel)
  int group_m;               // M-tiles per group of the ping-pong kernel's tile walk
  int skew;                  // per-XCD rotation of the tile walk (de-phases the 8 XCDs' HBM/MALL channel access)
  int kt_total;              // split-K (ping-pong kernel): total K-tiles of the product; 0 = blockIdx.y is a batch index, K = whole contraction
  int accum;                 // fp32 output only: C += result (gradient accumulation into an fp32 arena)
  int k_split_total;         // register-staging kernel as K-slices: total K (elements); blockIdx.y % batch1 = slice, p.K = elements per slice; 0 = off
  const bf16_t* a_norm_w; float a_norm_eps; int a_swiglu;     // skinny route: transform of the A rows while they are loaded (decode-step fusions)
  // fused Llama-layer epilogues of the 128 x 256 two-phase kernel (llmseg_gemm_args.fx, round 6): see epilogue_fx
  int fx, fx_T, fx_cols, fx_I; const float* fx_cos; const float* fx_sin; bf16_t* fx_out; const bf16_t* fx_in; long fx_ld;
  int ablate;                // loader-wave experiment only (LLMSEG_LW_ABLATE; results are garbage): bit 0 = no MFMAs, bit 1 = no fragment reads, bit 2 = no DMA after the prologue, bit 3 = no per-K-tile barrier
};

// exact-erf GELU on a pair (packed fp32 VALU: v_pk_fma / v_pk_mul).  Same Abramowitz-Stegun 7.1.26 erf as apply_act, rearranged:
// 0.5 v (1 + erf(v/sqrt2)) = max(v, 0) - 0.5|v| * t*poly(t) * exp(-v^2/2),  t = 1 / (1 + p|v|/sqrt2);  exp via exp2.
__device__ __forceinline__ f32x2_t gelu2(f32x2_t v) {
  const f32x2_t a = __builtin_elementwise_abs(v);
  const f32x2_t den = __builtin_elementwise_fma(a, (f32x2_t)(0.3275911f * 0.70710678118654752f), (f32x2_t)(1.f));
  f32x2_t t;
  t.x = __builtin_amdgcn_rcpf(den.x); t.y = __builtin_amdgcn_rcpf(den.y);
  f32x2_t p = __builtin_elementwise_fma(t, (f32x2_t)(1.061405429f), (f32x2_t)(-1.453152027f));
  p = __builtin_elementwise_fma(p, t, (f32x2_t)(1.421413741f));
  p = __builtin_elementwise_fma(p, t, (f32x2_t)(-0.284496736f));
  p = __builtin_elementwise_fma(p, t, (f32x2_t)(0.254829592f));
  const f32x2_t zs = a * 0.84932180028801907f;          // |v| * sqrt(log2(e) / 2): exp(-v^2/2) = exp2(-zs^2)
  const f32x2_t nz2 = -(zs * zs);
  f32x2_t e;
  e.x = __builtin_amdgcn_exp2f(nz2.x); e.y = __builtin_amdgcn_exp2f(nz2.y);
  const f32x2_t hp = (a * 0.5f) * (p * t);
  return __builtin_elementwise_fma(-hp, e, __builtin_elementwise_max(v, (f32x2_t)(0.f)));
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * (BK * 2) + (((chunk ^ (row >> 1)) & 7) << 4); }

struct TileXY { int m0, n0; };

// workgroup -> tile: XCD-contiguous (bid % 8 is the XCD), then GROUP_M-grouped, M fastest
template <int BM>
__device__ __forceinline__ TileXY tile_of(const GemmP& p, int bid) {
  const int nwg = p.tiles_m * p.tiles_n;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int len = q + (xcd < r ? 1 : 0);                                   // tiles in this XCD's contiguous chunk
    const int idx = ((bid >> 3) + xcd * p.skew) % len;                       // rotate the walk inside the chunk
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective for any nwg
  }
  const int per_group = GROUP_M * p.tiles_n;
  const int grp = bid / per_group;
  const int first_m = grp * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  TileXY t;
  t.m0 = (first_m + (bid % per_group) % gsz) * BM;
  t.n0 = ((bid % per_group) / gsz) * BN;
  return t;
}

// one BK=64 slab: 4 k-steps x (MI activation + 2 weight fragments, 2*MI MFMAs) for this wave's (32*MI) x 64 sub-tile
template <int MI>
__device__ __forceinline__ void mma_slab(const char* abase, const char* wbase, int wm, int wn, int frow, int fhalf, f32x16_t (&acc)[2][MI]) {
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) {
    bf16x8_t af[MI], wf[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(wbase + lds_off(wn * 64 + j * 32 + frow, ks * 2 + fhalf));
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(abase + lds_off(wm * 32 * MI + i * 32 + frow, ks * 2 + fhalf));
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
  }
}

__device__ __forceinline__ void ld4bf(const bf16_t* p, bool vec, int nv, float* o) {
  if (vec && nv == 4) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
    o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = e < nv ? bf2f(p[e]) : 0.f;
  }
}

// Accumulator views: how a wave's (32*MI) x 64 fp32 sub-tile sits in registers, and how pass i (32 output rows) of it is written
// into the per-wave 32 x 64 slab (row = output row m, 16-byte chunk = 4 consecutive n, chunk XOR (row & 15)).
template <int MI>
struct Acc32 {                       // v_mfma_f32_32x32x16: acc[j][i][4g+e] = C[n = 32j + 8g + 4*(lane>>5) + e][m = 32i + (lane&31)]
  const f32x16_t (&a)[2][MI];
  __device__ __forceinline__ void write(float* slab, int lane, int i) const {
    const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int chunk = (j * 32 + 8 * g + 4 * fhalf) >> 2;
        *reinterpret_cast<float4*>(slab + frow * 64 + ((chunk ^ (frow & 15)) << 2)) =
            make_float4(a[j][i][4 * g], a[j][i][4 * g + 1], a[j][i][4 * g + 2], a[j][i][4 * g + 3]);
      }
  }
};
template <int MI>
struct Acc16 {                       // v_mfma_f32_16x16x32: a[nb][mb][e] = C[n = 16nb + 4*(lane>>4) + e][m = 16mb + (lane&15)], nb < 4, mb < 2 MI
  const f32x4_t (&a)[4][2 * MI];
  __device__ __forceinline__ void write(float* slab, int lane, int i) const {
    const int r15 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int row = h * 16 + r15, chunk = nb * 4 + q;
        *reinterpret_cast<float4*>(slab + row * 64 + ((chunk ^ (row & 15)) << 2)) =
            make_float4(a[nb][2 * i + h][0], a[nb][2 * i + h][1], a[nb][2 * i + h][2], a[nb][2 * i + h][3]);
      }
  }
};

// Coalesced epilogue: the accumulator fragment (lane = one output row, 4 x 4 consecutive columns) is bounced through a
// per-wave 32 x 64 fp32 LDS slab (XOR-swizzled 16-byte chunks: conflict-free both ways) so that afterwards 16 adjacent lanes
// hold one output row's 64 consecutive columns: residual loads and C stores become full 128-byte lines instead of 8-byte
// pieces scattered over 32 rows (measured on 32768x1280x1280 + bias + residual: 480 -> see profiles).  LDS ops of one wave
// execute in order, so no barrier is needed; the slab aliases the (finished) operand tiles.
template <bool OUT_F32, int MI, class ACC>
__device__ __forceinline__ void epilogue_lds_edge(const GemmP& p, const ACC& acc, char* smem, int wave, int m0, int n0, int wm, int wn,
                                               int lane, long bz) {
  float* slab = reinterpret_cast<float*>(smem) + wave * (32 * 64);
  const int frow = lane & 31, fhalf = lane >> 5;
  const int c = lane & 15, rsub = lane >> 4;
  const int n = n0 + wn * 64 + c * 4;
  const int nv = min(4, p.N - n);
  const bool vec_ok = p.c_vec != 0, res_vec = p.r_vec != 0, b_vec = p.b_vec != 0;
  float bs[4] = {0.f, 0.f, 0.f, 0.f}, gm[4] = {1.f, 1.f, 1.f, 1.f};
  if (nv > 0) {
    if (p.bias) ld4bf(p.bias + n, b_vec, nv, bs);
    if (p.gamma) ld4bf(p.gamma + n, b_vec, nv, gm);
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    uint2 rpre[8];
    if (p.res && res_vec && nv == 4) {          // issue the 8 residual loads of this pass first: they fly during the LDS bounce
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int m = min(m0 + wm * 32 * MI + i * 32 + it * 4 + rsub, p.M - 1);
        rpre[it] = *reinterpret_cast<const uint2*>(p.res + bz + (long)m * p.ldr + n);
      }
    }
    acc.write(slab, lane, i);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 4 + rsub;
      const float4 a4 = *reinterpret_cast<const float4*>(slab + row * 64 + ((c ^ (row & 15)) << 2));
      const int m = m0 + wm * 32 * MI + i * 32 + row;
      if (m >= p.M || nv <= 0) continue;
      float v[4] = {a4.x * p.alpha + bs[0], a4.y * p.alpha + bs[1], a4.z * p.alpha + bs[2], a4.w * p.alpha + bs[3]};
      if (p.act != LLMSEG_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
      }
      if (p.gamma) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gm[e];
      }
      if (p.res) {
        if (res_vec && nv == 4) {
          v[0] += __uint_as_float(rpre[it].x << 16); v[1] += __uint_as_float(rpre[it].x & 0xffff0000u);
          v[2] += __uint_as_float(rpre[it].y << 16); v[3] += __uint_as_float(rpre[it].y & 0xffff0000u);
        } else {
          float rr[4];
          ld4bf(p.res + bz + (long)m * p.ldr + n, false, nv, rr);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += rr[e];
        }
      }
      if (OUT_F32) {
        float* cp = reinterpret_cast<float*>(p.C) + bz + (long)m * p.ldc + n;
        if (p.accum) {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (e < nv) v[e] += cp[e];
        }
        if (vec_ok && nv == 4) {
          *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (e < nv) cp[e] = v[e];
        }
      } else {
        bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + bz + (long)m * p.ldc + n;
        if (vec_ok && nv == 4) {
          *reinterpret_cast<uint2*>(cp) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (e < nv) cp[e] = f2bf(v[e]);
        }
      }
    }
  }
}

// Interior tiles (the wave's whole (32*MI) x 64 block inside C, every row start 4-element aligned): the same LDS bounce as
// straight-line code - no per-lane bounds or alignment branches, the activation resolved once per tile (ACT is a template
// argument of the body), LDS offsets hoisted out of the pass loop, running 64-bit row pointers instead of a 64-bit multiply per
// row.  The branchy edge version above costs ~10-18 us per 256 x 256 tile (instruction-bound with 2 waves/SIMD).
template <bool OUT_F32, int MI, int ACT, class ACC>
__device__ __forceinline__ void epilogue_lds_body(const GemmP& p, const ACC& acc, float* slab, int mw, int nw, int lane, long bz) {
  const int frow = lane & 31, fhalf = lane >> 5;
  const int c = lane & 15, rsub = lane >> 4;
  const int n = nw + c * 4;
  float bs[4] = {0.f, 0.f, 0.f, 0.f}, gm[4] = {1.f, 1.f, 1.f, 1.f};
  if (p.bias) ld4bf(p.bias + n, true, 4, bs);
  if (p.gamma) ld4bf(p.gamma + n, true, 4, gm);
  int roff[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) roff[it] = (it * 4 + rsub) * 64 + ((c ^ ((it * 4 + rsub) & 15)) << 2);
  const long row0 = (long)(mw + rsub);
  char* cp = reinterpret_cast<char*>(p.C) + (bz + row0 * p.ldc + n) * (OUT_F32 ? 4 : 2);
  const long cstep = 4 * p.ldc * (OUT_F32 ? 4 : 2);
  const bf16_t* rp = p.res ? p.res + bz + row0 * p.ldr + n : nullptr;
  const long rstep = 4 * p.ldr;
  const float alpha = p.alpha;
  const bool has_gamma = p.gamma != nullptr;
  const bool accum = OUT_F32 && p.accum != 0;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    uint2 rpre[8];
    if (rp) {
#pragma unroll
      for (int it = 0; it < 8; ++it) { rpre[it] = *reinterpret_cast<const uint2*>(rp); rp += rstep; }
    }
    acc.write(slab, lane, i);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const float4 a4 = *reinterpret_cast<const float4*>(slab + roff[it & 3] + (it >> 2) * (16 * 64));
      float v[4] = {fmaf(a4.x, alpha, bs[0]), fmaf(a4.y, alpha, bs[1]), fmaf(a4.z, alpha, bs[2]), fmaf(a4.w, alpha, bs[3])};
      if (ACT == LLMSEG_ACT_GELU) {
        const f32x2_t g0 = gelu2(f32x2_t{v[0], v[1]}), g1 = gelu2(f32x2_t{v[2], v[3]});
        v[0] = g0.x; v[1] = g0.y; v[2] = g1.x; v[3] = g1.y;
      } else if (ACT != LLMSEG_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], ACT);
      }
      if (has_gamma) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gm[e];
      }
      if (rp) {
        v[0] += __uint_as_float(rpre[it].x << 16); v[1] += __uint_as_float(rpre[it].x & 0xffff0000u);
        v[2] += __uint_as_float(rpre[it].y << 16); v[3] += __uint_as_float(rpre[it].y & 0xffff0000u);
      }
      if (OUT_F32) {
        if (accum) {
          const float4 o4 = *reinterpret_cast<const float4*>(cp);
          v[0] += o4.x; v[1] += o4.y; v[2] += o4.z; v[3] += o4.w;
        }
        *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
      } else *reinterpret_cast<uint2*>(cp) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
      cp += cstep;
    }
  }
}

template <bool OUT_F32, int MI, class ACC>
__device__ __forceinline__ void epilogue_lds(const GemmP& p, const ACC& acc, char* smem, int wave, int m0, int n0, int wm, int wn,
                                             int lane, long bz) {
  const int mw = m0 + wm * 32 * MI, nw = n0 + wn * 64;
  const bool interior = mw + 32 * MI <= p.M && nw + 64 <= p.N && p.c_vec && p.b_vec && (!p.res || p.r_vec);
  if (!interior) {
    epilogue_lds_edge<OUT_F32, MI, ACC>(p, acc, smem, wave, m0, n0, wm, wn, lane, bz);
    return;
  }
  float* slab = reinterpret_cast<float*>(smem) + wave * (32 * 64);
  switch (p.act) {
    case LLMSEG_ACT_NONE: epilogue_lds_body<OUT_F32, MI, LLMSEG_ACT_NONE, ACC>(p, acc, slab, mw, nw, lane, bz); break;
    case LLMSEG_ACT_GELU: epilogue_lds_body<OUT_F32, MI, LLMSEG_ACT_GELU, ACC>(p, acc, slab, mw, nw, lane, bz); break;
    case LLMSEG_ACT_QUICKGELU: epilogue_lds_body<OUT_F32, MI, LLMSEG_ACT_QUICKGELU, ACC>(p, acc, slab, mw, nw, lane, bz); break;
    case LLMSEG_ACT_SILU: epilogue_lds_body<OUT_F32, MI, LLMSEG_ACT_SILU, ACC>(p, acc, slab, mw, nw, lane, bz); break;
    case LLMSEG_ACT_RELU: epilogue_lds_body<OUT_F32, MI, LLMSEG_ACT_RELU, ACC>(p, acc, slab, mw, nw, lane, bz); break;
    default: epilogue_lds_body<OUT_F32, MI, LLMSEG_ACT_SIGMOID, ACC>(p, acc, slab, mw, nw, lane, bz); break;
  }
}

// ---- fused Llama-layer epilogues of the 128 x 256 two-phase kernel (llmseg_gemm_args.fx; round 6) -------------------------------------------------
// Each replaces a pointwise launch that used to re-read what the GEMM has just written, and reproduces that launch's bits: the product is rounded to
// bf16 first (what the unfused route stores) and then goes through the pointwise kernel's own arithmetic (common.h: rope_lo / rope_hi, swiglu_*).
//   FX_ROPE        q|k|v projection: every head (128 columns) of the columns < fx_cols is rotated (rotate-half, position = row % fx_T) before the store
//   FX_SWIGLU      gate|up projection: C = gate|up as before AND fx_out = silu(gate) * up
//   FX_SWIGLU_BWD  dX of down_proj: the product is d(silu(gate) * up); C = d(gate|up) computed from it and the saved gate|up (fx_in)
// ROPE and SWIGLU need two columns a fixed distance apart (64 inside a head; gate column c and up column c) in ONE lane: the kernel then gives wave wn
// the W-tile rows {b .. b + 31} and {b + PS .. b + PS + 31} (ROPE: b = 128 (wn / 2) + 32 (wn % 2), PS = 64; SWIGLU: b = 32 wn, PS = 128 and the tile's 256 W
// rows are 128 gate rows + the 128 up rows of the same columns) instead of 64 consecutive ones, so that accumulator blocks nb and nb + 2 hold the two
// elements of a pair, and after the LDS bounce lane (cc = lane % 8, r8 = lane / 8) reads slab chunks cc and cc + 8 of a row: 4 + 4 paired columns.
enum { FX_NONE = 0, FX_ROPE = 1, FX_SWIGLU = 2, FX_SWIGLU_BWD = 3 };

__device__ __forceinline__ void st4bf(bf16_t* p, const float* v) { *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3])); }
__device__ __forceinline__ void ld4bf_v(const bf16_t* p, float* o) {
  const uint2 r = *reinterpret_cast<const uint2*>(p);
  o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u); o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
}

template <int MI, int FX, class ACC>
__device__ __forceinline__ void epilogue_fx(const GemmP& p, const ACC& acc, char* smem, int wave, int m0, int n0, int wm, int wn, int lane) {
  float* slab = reinterpret_cast<float*>(smem) + wave * (32 * 64);
  bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
  const int mw = m0 + wm * 32 * MI;
  if constexpr (FX == FX_SWIGLU_BWD) {
    const int c = lane & 15, rsub = lane >> 4;
    const int n = n0 + wn * 64 + c * 4;                     // column of d(out): 4 consecutive per lane
    if (n >= p.N) return;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      acc.write(slab, lane, i);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + rsub, m = mw + i * 32 + row;
        const float4 a4 = *reinterpret_cast<const float4*>(slab + row * 64 + ((c ^ (row & 15)) << 2));
        if (m >= p.M) continue;
        const float d[4] = {round_bf(a4.x), round_bf(a4.y), round_bf(a4.z), round_bf(a4.w)};
        float g[4], u[4], og[4], ou[4];
        ld4bf_v(p.fx_in + (long)m * p.fx_ld + n, g);
        ld4bf_v(p.fx_in + (long)m * p.fx_ld + p.fx_I + n, u);
#pragma unroll
        for (int e = 0; e < 4; ++e) swiglu_bwd1(d[e], g[e], u[e], og[e], ou[e]);
        st4bf(C + (long)m * p.ldc + n, og);
        st4bf(C + (long)m * p.ldc + p.fx_I + n, ou);
      }
    }
  } else {
    const int cc = lane & 7, r8 = lane >> 3;
    const int lcol = (FX == FX_ROPE ? (wn >> 1) * 128 + (wn & 1) * 32 : wn * 32) + cc * 4;      // tile-local column of the first element of the pair
    const bool rot = FX == FX_ROPE && n0 < p.fx_cols;
    const float inv_T = 1.f / (float)p.fx_T;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      acc.write(slab, lane, i);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + r8, m = mw + i * 32 + row;
        const float4 x1 = *reinterpret_cast<const float4*>(slab + row * 64 + ((cc ^ (row & 15)) << 2));
        const float4 x2 = *reinterpret_cast<const float4*>(slab + row * 64 + (((cc + 8) ^ (row & 15)) << 2));
        if (m >= p.M) continue;
        const float v1[4] = {round_bf(x1.x), round_bf(x1.y), round_bf(x1.z), round_bf(x1.w)};
        const float v2[4] = {round_bf(x2.x), round_bf(x2.y), round_bf(x2.z), round_bf(x2.w)};
        if constexpr (FX == FX_ROPE) {
          float o1[4] = {v1[0], v1[1], v1[2], v1[3]}, o2[4] = {v2[0], v2[1], v2[2], v2[3]};
          if (rot) {
            int pos = m - __float2int_rz((float)m * inv_T) * p.fx_T;      // m % fx_T without an integer division (m < 2^22)
            pos = pos < 0 ? pos + p.fx_T : (pos >= p.fx_T ? pos - p.fx_T : pos);
            const int d = (wn & 1) * 32 + cc * 4;
            const float4 c4 = *reinterpret_cast<const float4*>(p.fx_cos + (long)pos * 64 + d), s4 = *reinterpret_cast<const float4*>(p.fx_sin + (long)pos * 64 + d);
            const float cv[4] = {c4.x, c4.y, c4.z, c4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { o1[e] = rope_lo(v1[e], v2[e], cv[e], sv[e]); o2[e] = rope_hi(v1[e], v2[e], cv[e], sv[e]); }
          }
          bf16_t* cp = C + (long)m * p.ldc + n0 + lcol;
          st4bf(cp, o1);
          st4bf(cp + 64, o2);
        } else {
          float h[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = swiglu_fwd1(v1[e], v2[e]);
          bf16_t* cp = C + (long)m * p.ldc + n0 + lcol;        // n0 = 128 x the tile's column index: gate columns n0 .., up columns fx_I + n0 ..
          st4bf(cp, v1);
          st4bf(cp + p.fx_I, v2);
          st4bf(p.fx_out + (long)m * p.fx_ld + n0 + lcol, h);
        }
      }
    }
  }
}

template <int MI>
__device__ __forceinline__ void zero_acc(f32x16_t (&acc)[2][MI]) {
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
}

// ---- variant R: global -> VGPR -> LDS staging, 128 x 128 tile, 2 LDS buffers -------------------------------------------
// Handles every layout the backward pass needs without materialising transposes:
//   TA = false: A stored [M][K] (K contiguous)        TA = true: A stored [K][M] (M contiguous; "A^T given")
//   TW = false: W stored [N][K] (K contiguous)        TW = true: W stored [K][N] (N contiguous)
// A K-contiguous operand needs K % 8 == 0; the K tail is zero-filled per 16-byte chunk.  A transposed-stored operand is
// loaded as 4(k) x 8(rows) blocks, transposed in registers (v_perm) and written as 8-byte LDS rows; any K is fine there.
__device__ __forceinline__ uint32_t perm_lo16(uint32_t a, uint32_t b) { return (a & 0xffffu) | (b << 16); }
__device__ __forceinline__ uint32_t perm_hi16(uint32_t a, uint32_t b) { return (a >> 16) | (b & 0xffff0000u); }

template <bool TR>
struct OperandStage {
  const bf16_t* ptr[4];
  int off[4];       // LDS byte offsets (normal: 4 x 16-byte chunks; transposed: base of 8 x 8-byte rows)
  uint4 r[4];
  int kq;           // normal: first k of this thread's chunk inside the tile; transposed: first stored row inside the tile
  long ld;

  __device__ __forceinline__ void init(const bf16_t* g, long ld_, int x0, int R, int tid) {
    ld = ld_;
    if (!TR) {
      const int kc = tid & 7, r0 = tid >> 3;
      kq = kc * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ptr[i] = g + (long)min(x0 + r0 + 32 * i, R - 1) * ld + kc * 8;     // rows past R: clamp (results discarded)
        off[i] = lds_off(r0 + 32 * i, kc);
      }
    } else {
      const int kb = tid & 15, mb = tid >> 4;                               // 4 stored rows kb*4.., 8 columns mb*8..
      kq = kb * 4;
      const int col = min(x0 + mb * 8, ((R - 1) >> 3) << 3);                 // stay inside the row allocation (ld % 8 == 0)
#pragma unroll
      for (int j = 0; j < 4; ++j) ptr[j] = g + (long)(kb * 4 + j) * ld + col;
      off[0] = mb * 8; off[1] = kb;                                          // row base, k-quad
    }
  }
  __device__ __forceinline__ void load(int t, int K) {
    if (!TR) {
      if (t * BK + kq < K) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const uint4*>(ptr[i] + (long)t * BK);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = make_uint4(0, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        r[j] = (t * BK + kq + j < K) ? *reinterpret_cast<const uint4*>(ptr[j] + (long)t * BK * ld) : make_uint4(0, 0, 0, 0);
    }
  }
  __device__ __forceinline__ void store(char* base) const {
    if (!TR) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(base + off[i]) = r[i];
    } else {
      const int row0 = off[0], kb = off[1];
      const int ch = kb >> 1, sub = (kb & 1) * 8;
      // r[j] = 8 consecutive rows' values for stored row j; out row e gets (r0[e], r1[e], r2[e], r3[e]) = 8 bytes
      const uint2 o0 = make_uint2(perm_lo16(r[0].x, r[1].x), perm_lo16(r[2].x, r[3].x));
      const uint2 o1 = make_uint2(perm_hi16(r[0].x, r[1].x), perm_hi16(r[2].x, r[3].x));
      const uint2 o2 = make_uint2(perm_lo16(r[0].y, r[1].y), perm_lo16(r[2].y, r[3].y));
      const uint2 o3 = make_uint2(perm_hi16(r[0].y, r[1].y), perm_hi16(r[2].y, r[3].y));
      const uint2 o4 = make_uint2(perm_lo16(r[0].z, r[1].z), perm_lo16(r[2].z, r[3].z));
      const uint2 o5 = make_uint2(perm_hi16(r[0].z, r[1].z), perm_hi16(r[2].z, r[3].z));
      const uint2 o6 = make_uint2(perm_lo16(r[0].w, r[1].w), perm_lo16(r[2].w, r[3].w));
      const uint2 o7 = make_uint2(perm_hi16(r[0].w, r[1].w), perm_hi16(r[2].w, r[3].w));
      *reinterpret_cast<uint2*>(base + lds_off(row0 + 0, ch) + sub) = o0;
      *reinterpret_cast<uint2*>(base + lds_off(row0 + 1, ch) + sub) = o1;
      *reinterpret_cast<uint2*>(base + lds_off(row0 + 2, ch) + sub) = o2;
      *reinterpret_cast<uint2*>(base + lds_off(row0 + 3, ch) + sub) = o3;
      *reinterpret_cast<uint2*>(base + lds_off(row0 + 4, ch) + sub) = o4;
      *reinterpret_cast<uint2*>(base + lds_off(row0 + 5, ch) + sub) = o5;
      *reinterpret_cast<uint2*>(base + lds_off(row0 + 6, ch) + sub) = o6;
      *reinterpret_cast<uint2*>(base + lds_off(row0 + 7, ch) + sub) = o7;
    }
  }
};

template <bool OUT_F32, bool TA, bool TW>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_tn_kernel(GemmP p) {
  constexpr int MI = 2, BM = 128, TILE = 128 * BK * 2, BUF = 2 * TILE;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const TileXY tc = tile_of<BM>(p, blockIdx.x);
  const int m0 = tc.m0, n0 = tc.n0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long b1 = blockIdx.y % p.batch1, b2 = blockIdx.y / p.batch1;
  const long bz = b1 * p.sC + b2 * p.sC2;          // element offset of this batch entry in C / residual

  OperandStage<TA> sa;
  OperandStage<TW> sw;
  sa.init(p.A + b1 * p.sA + b2 * p.sA2, p.lda, m0, p.M, tid);
  sw.init(p.W + b1 * p.sW + b2 * p.sW2, p.ldw, n0, p.N, tid);
  const int Kloc = p.k_split_total > 0 ? min(p.K, p.k_split_total - (int)b1 * p.K) : p.K;      // K-slice b1 of a split launch (the last may be shorter)
  const int nt = (Kloc + BK - 1) / BK;

  f32x16_t acc[2][MI];
  zero_acc<MI>(acc);
  const int frow = lane & 31, fhalf = lane >> 5;

  sa.load(0, Kloc); sw.load(0, Kloc);
  sa.store(smem); sw.store(smem + TILE);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) { sa.load(t + 1, Kloc); sw.load(t + 1, Kloc); }
    const char* abase = smem + (t & 1) * BUF;
    mma_slab<MI>(abase, abase + TILE, wm, wn, frow, fhalf, acc);
    if (t + 1 < nt) { char* nb = smem + ((t + 1) & 1) * BUF; sa.store(nb); sw.store(nb + TILE); }
    __syncthreads();
  }
  epilogue_lds<OUT_F32, MI>(p, Acc32<MI>{acc}, smem, wave, m0, n0, wm, wn, lane, bz);
}

// ---- variant G: direct global -> LDS DMA, K % 64 == 0 ---------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <bool OUT_F32, int MI, int NBUF>
__global__ __launch_bounds__(NT, NBUF == 2 ? (MI == 4 ? 1 : 2) : (MI == 4 ? 2 : 4)) void gemm_bf16_tn_glds_kernel(GemmP p) {
  constexpr int BM = 64 * MI;
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, BUF = A_BYTES + W_BYTES;
  constexpr int A_DMA = BM / 32, W_DMA = BN / 32;              // 1-KiB DMA instructions per wave per tile
  __shared__ __attribute__((aligned(16))) char smem[NBUF * BUF];
  const TileXY tc = tile_of<BM>(p, blockIdx.x);
  const int m0 = tc.m0, n0 = tc.n0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const long b1 = blockIdx.y % p.batch1, b2 = blockIdx.y / p.batch1;
  const long bz = b1 * p.sC + b2 * p.sC2;          // element offset of this batch entry in C / residual
  const bf16_t* __restrict__ Ag = p.A + b1 * p.sA + b2 * p.sA2;
  const bf16_t* __restrict__ Wg = p.W + b1 * p.sW + b2 * p.sW2;

  // DMA instruction i of this wave fills LDS rows (i*4 + wave)*8 .. +7 of the operand tile (1 KiB)
  const bf16_t* a_src[A_DMA];
  const bf16_t* w_src[W_DMA];
#pragma unroll
  for (int i = 0; i < A_DMA; ++i) {
    const int row = (i * 4 + wave) * 8 + (lane >> 3);
    a_src[i] = Ag + (long)min(m0 + row, p.M - 1) * p.lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int i = 0; i < W_DMA; ++i) {
    const int row = (i * 4 + wave) * 8 + (lane >> 3);
    w_src[i] = Wg + (long)min(n0 + row, p.N - 1) * p.ldw + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
  }
  const int nt = p.K / BK;

  auto issue = [&](int t, int buf) {
    char* base = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < W_DMA; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[i] + (long)t * BK), (lds_ptr_t)(base + A_BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < A_DMA; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[i] + (long)t * BK), (lds_ptr_t)(base + (i * 4 + wave) * 1024), 16, 0, 0);
  };

  f32x16_t acc[2][MI];
  zero_acc<MI>(acc);
  const int frow = lane & 31, fhalf = lane >> 5;

  if (NBUF == 2) {
    issue(0, 0);
    __syncthreads();                       // hipcc drains the DMA (vmcnt(0)) ahead of the barrier
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
      const char* abase = smem + (t & 1) * BUF;
      mma_slab<MI>(abase, abase + A_BYTES, wm, wn, frow, fhalf, acc);
      __syncthreads();
    }
  } else {
    for (int t = 0; t < nt; ++t) {
      issue(t, 0);
      __syncthreads();
      mma_slab<MI>(smem, smem + A_BYTES, wm, wn, frow, fhalf, acc);
      __syncthreads();
    }
  }
  epilogue_lds<OUT_F32, MI>(p, Acc32<MI>{acc}, smem, wave, m0, n0, wm, wn, lane, bz);
}


// ---- variant G4 (round 6): 128 x 128 tile, 4 waves, FOUR-stage LDS-DMA ring with counted waits ---------------------------------------------
// For the products that put at most one workgroup on a CU (CLIP at two sequences: M = 514; the mask-selection head: M = C K = 512, N = 256 ... 2048):
// there the 128 x 128 kernel above has nothing to hide its fetch latency with (its one or two buffers wait for every K-tile: 16.6-17.3 us on
// 514 x 3072 x 1024 whichever kernel ran it, profiles/r04g_small_gemm_variants.txt).  Here THREE K-tiles (96 KiB) are in flight while the fourth
// stage is computed: iteration t = { s_waitcnt vmcnt(stages still allowed in flight) | s_barrier | issue K-tile t + 3 into the buffer iteration
// t - 1 read (every wave has passed the barrier, so every wave has finished reading it) | 16 MFMAs on stage t }.  Same lane -> (row, chunk) DMA
// mapping, swizzle and MFMA order as the kernel above: identical bits.  K % 64 == 0; 128 KiB of LDS, one workgroup per CU.
#define G4_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
template <bool OUT_F32>
__global__ __launch_bounds__(NT, 1) void gemm_bf16_tn_g4_kernel(GemmP p) {
  constexpr int MI = 2, BM = 128, NS = 4;
  constexpr int A_BYTES = BM * BK * 2, BUF = A_BYTES + BN * BK * 2;          // 16 KiB + 16 KiB per stage
  __shared__ __attribute__((aligned(16))) char smem[NS * BUF];
  const TileXY tc = tile_of<BM>(p, blockIdx.x);
  const int m0 = tc.m0, n0 = tc.n0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const long b1 = blockIdx.y % p.batch1, b2 = blockIdx.y / p.batch1;
  const long bz = b1 * p.sC + b2 * p.sC2;
  const bf16_t* __restrict__ Ag = p.A + b1 * p.sA + b2 * p.sA2;
  const bf16_t* __restrict__ Wg = p.W + b1 * p.sW + b2 * p.sW2;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ag + (long)m0 * p.lda), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Wg + (long)n0 * p.ldw), 0, 0x7fffffff, 0x00020000);
  // DMA instruction i of this wave fills LDS rows (i*4 + wave)*8 .. +7 of an operand tile (1 KiB); per-lane byte offsets relative to the tile's first row
  int a_off[4], w_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wave) * 8 + (lane >> 3);
    const int sw = ((lane & 7) ^ ((row >> 1) & 7)) << 3;
    a_off[i] = (int)(((long)min(row, p.M - 1 - m0) * p.lda + sw) * 2);
    w_off[i] = (int)(((long)min(row, p.N - 1 - n0) * p.ldw + sw) * 2);
  }
  const int nt = p.K / BK;
  auto issue = [&](int t, int st) {
    char* base = smem + st * BUF;
    const int koff = t * (BK * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(base + A_BYTES + (i * 4 + wave) * 1024), 16, w_off[i], koff, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(base + (i * 4 + wave) * 1024), 16, a_off[i], koff, 0, 0);
  };
  f32x16_t acc[2][MI];
  zero_acc<MI>(acc);
  const int frow = lane & 31, fhalf = lane >> 5;
  issue(0, 0);
  if (nt > 1) issue(1, 1);
  if (nt > 2) issue(2, 2);
  for (int t = 0; t < nt; ++t) {
    const int ahead = min(nt - 1 - t, 2);            // K-tiles issued beyond t: 8 DMA instructions of this wave each
    if (ahead == 2) G4_VM(16); else if (ahead == 1) G4_VM(8); else G4_VM(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (t + 3 < nt) issue(t + 3, (t + 3) & 3);
    const char* abase = smem + (t & 3) * BUF;
    mma_slab<MI>(abase, abase + A_BYTES, wm, wn, frow, fhalf, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();                                   // the epilogue's slabs alias stage 0
  epilogue_lds<OUT_F32, MI>(p, Acc32<MI>{acc}, smem, wave, m0, n0, wm, wn, lane, bz);
}

// ---- variant Q-LW (round 6 experiment): the 128 x 256 tile with DEDICATED LOADER WAVES --------------------------------------------------------------
// Hypothesis: at M = 638 the 128 x 256 tile is bound neither by the fabric (operands mostly L2 hits: a W tile is shared by the five row tiles of
// an XCD) nor by the vector-memory path (48 KiB per K-tile = 0.75 of its 64 B/clk at full MFMA rate) but by the ISSUE cost of the LDS-DMA instructions inside
// the waves that also issue the MFMAs: a `buffer_load ... lds` piece stalls its wave for 60-185 cycles (MI355X_MICROARCH.md), six of them per wave and K-tile
// against 32 MFMAs (~512 cycles).  Here 8 consumer waves (2 per SIMD; the wave tiling, fragment reads, MFMA order and epilogue of the ping-pong kernels: same
// bits) never touch global memory; 4 loader waves (one per SIMD) issue all 48 pieces of a K-tile, two K-tiles ahead, into a three-stage LDS ring.
// ONE workgroup barrier per K-tile:  loader: s_waitcnt vmcnt(tile t landed) | barrier | issue tile t + 2 (into the stage tile t - 1 was read from: every
// consumer has passed this barrier, so it has finished tile t - 1);  consumer: barrier | 16 fragment reads + 32 MFMAs on tile t.
constexpr int NTL = 768;
template <bool OUT_F32>
__global__ __launch_bounds__(NTL, 1) void gemm_bf16_tn_lw_kernel(GemmP p) {
  constexpr int MI = 2, BMB = 128, BNB = 256, NS = 3;
  constexpr int A_BYTES = BMB * BK * 2, W_BYTES = BNB * BK * 2, BUF = A_BYTES + W_BYTES;      // 16 + 32 KiB per stage, 144 KiB in all
  __shared__ __attribute__((aligned(16))) char smem[NS * BUF];
  int bid = blockIdx.x;
  const int nwg = p.tiles_m * p.tiles_n;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int len = q + (xcd < r ? 1 : 0);
    const int idx = ((bid >> 3) + xcd * p.skew) % len;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int per_group = p.group_m * p.tiles_n;
  const int first_m = (bid / per_group) * p.group_m;
  const int gsz = min(p.tiles_m - first_m, p.group_m);
  const int m0 = (first_m + (bid % per_group) % gsz) * BMB;
  const int n0 = ((bid % per_group) / gsz) * BNB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long b1 = blockIdx.y % p.batch1, b2 = blockIdx.y / p.batch1;
  const long bz = b1 * p.sC + b2 * p.sC2;
  int nt = p.K / BK;
  if (p.kt_total > 0) nt = min(nt, p.kt_total - (int)b1 * nt);                                 // split-K: the last slice may be shorter

  if (wave >= 8) {
    // ------------------------------------------------------------------ loader wave l: A pieces l, l + 4, .. (4 of 16), W pieces l, l + 4, .. (8 of 32)
    const int l = wave - 8;
    const bf16_t* __restrict__ Ag = p.A + b1 * p.sA + b2 * p.sA2;
    const bf16_t* __restrict__ Wg = p.W + b1 * p.sW + b2 * p.sW2;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ag + (long)m0 * p.lda), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Wg + (long)n0 * p.ldw), 0, 0x7fffffff, 0x00020000);
    int a_off[4], w_off[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (l + 4 * j) * 8 + (lane >> 3);
      a_off[j] = (int)(((long)min(row, p.M - 1 - m0) * p.lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = (l + 4 * j) * 8 + (lane >> 3);
      w_off[j] = (int)(((long)min(row, p.N - 1 - n0) * p.ldw + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2);
    }
    auto issue = [&](int t, int st) {
      char* base = smem + st * BUF + l * 1024;
      const int koff = t * (BK * 2);
#pragma unroll
      for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(base + A_BYTES + j * 4096), 16, w_off[j], koff, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(base + j * 4096), 16, a_off[j], koff, 0, 0);
    };
    issue(0, 0);
    if (nt > 1) issue(1, 1);
    int st2 = 2;                                           // stage of tile t + 2
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt) G4_VM(12); else G4_VM(0);            // tile t has landed (tile t + 1 may still be in flight)
      __builtin_amdgcn_sched_barrier(0);
      if (!(p.ablate & 8)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < nt && !(p.ablate & 4)) issue(t + 2, st2);
      st2 = st2 == NS - 1 ? 0 : st2 + 1;
    }
    __builtin_amdgcn_s_barrier();                          // (the consumers' barrier ahead of the epilogue)
    return;
  }

  // ---------------------------------------------------------------------- consumer waves: 2 x 4, each 64 x 64 of the tile
  const int wm = wave >> 2, wn = wave & 3;
  f32x4_t acc[4][2 * MI];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int mb = 0; mb < 2 * MI; ++mb) acc[nb][mb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
  int st = 0;
  for (int t = 0; t < nt; ++t) {
    if (!(p.ablate & 8)) __builtin_amdgcn_s_barrier();      // (bit 3: no per-K-tile barrier -- with bits 1 + 2 the bare MFMA stream of this wave arrangement)
    __builtin_amdgcn_sched_barrier(0);
    const char* base = smem + ((p.ablate & 2) ? 0 : st) * BUF;
    bf16x8_t wf[4][2], af[4][2];
    if (!(p.ablate & 2) || t == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) wf[nb][ks] = *reinterpret_cast<const bf16x8_t*>(base + A_BYTES + lds_off(wn * 64 + nb * 16 + frow, ks * 4 + fq));
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) af[mb][ks] = *reinterpret_cast<const bf16x8_t*>(base + lds_off(wm * 64 + mb * 16 + frow, ks * 4 + fq));
      }
    }
    if (!(p.ablate & 1)) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb][ks], af[mb][ks], acc[nb][mb], 0, 0, 0);
    } else {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[nb][0][0] += __builtin_bit_cast(f32x4_t, wf[nb][0])[0] + __builtin_bit_cast(f32x4_t, af[nb][1])[1] + __builtin_bit_cast(f32x4_t, wf[nb][1])[2] + __builtin_bit_cast(f32x4_t, af[nb][0])[3];
    }
    __builtin_amdgcn_sched_barrier(0);
    st = st == NS - 1 ? 0 : st + 1;
  }
  __builtin_amdgcn_s_barrier();                            // every consumer has finished reading LDS: the epilogue's slabs alias the stages
  epilogue_lds<OUT_F32, MI>(p, Acc16<MI>{acc}, smem, wave, m0, n0, wm, wn, lane, bz);
}

// ---- variant Q: (64 MI) x 256 tile, 8 waves in two groups that run ONE BARRIER APART ("ping-pong") -------------------------
// MI = 4: 256 x 256 (waves 2 x 4, each 128 x 64); MI = 2: 128 x 256 (each wave 64 x 64) for short matrices (Llama at 2 images per
// micro-step has M = 638 rows: five 128-row tiles waste 0.3 % of the rows, three 256-row tiles 17 %).
// Each K-tile (BK = 64) is four phases; a phase is {LDS fragment reads + one half-tile of DMA issue} | barrier | {4 MI MFMAs} |
// barrier.  Waves wm = 1 execute one extra barrier up front, so on every SIMD the wm = 0 wave's MFMA section overlaps the
// wm = 1 wave's read/DMA section and vice versa: the matrix pipe sees back-to-back MFMAs while the partner hides LDS latency.
//   operands per buffer (2 buffers): A[64 MI][64], W[256][64] bf16, 128-byte rows, chunk XOR (row>>1)&7
//   half-tiles (one DMA issue by all 8 waves: MI/2 instructions for A, 2 for W):  A0/A1 = the first / second half of BOTH wave
//   rows' blocks, W0/W1 = the first / second 32 rows of every 64-row wave column block.
//   phase:        p0                p1               p2               p3
//   reads         W0, A0            W1               A1               -
//   MFMAs (4 MI x v_mfma_f32_16x16x32_bf16 over 2 MI accumulators each)  W0 x A0 | W1 x A0 | W1 x A1 | W0 x A1
//   DMA issue     W1(t+1)           A1(t+1)          A0(t+2)          W0(t+2)          (issue sequence S[g+6] at phase g)
//   s_waitcnt     vmcnt(MI + 4)     vmcnt(MI + 4)    -                vmcnt(MI + 4)    (retires what phase g+1 reads)
// Ordering rules (MI355X_MICROARCH.md "LDS-DMA"): a half-tile is read one phase AFTER the phase whose pre-barrier vmcnt retired
// it (both groups' waits precede a barrier the reader has passed); a slot is re-issued >= 2 phases after its last read (the
// lagging group's reads retire one barrier later).  Four half-tiles stay in flight per workgroup.  K % 64 == 0, >= 2 K-tiles.
// LDS-DMA as `buffer_load_dwordx4 ... offen lds`: a raw buffer resource based at the tile's first row, tile-relative 32-bit per-lane
// byte offsets, the K advance in the scalar offset (no 64-bit VALU address arithmetic per instruction, 8 fewer VGPRs than pointers)
// Split-K (p.kt_total > 0): blockIdx.y is the K-slice; slice s covers K-tiles [s q, min((s+1) q, kt_total)), q = p.K / 64, and
// writes its fp32 partial tile to slab s of the workspace (plain OUT_F32 stores); splitk_reduce_kernel sums the slabs and applies
// the epilogue.  Short matrices with a long contraction (Llama o / down and the dX GEMMs at M = 638: 48 tiles of 256 x 256)
// otherwise leave 80 % of the CUs idle.
#define PP_KOFF(tt) ((int)(((tt) - (EXT ? 1 : 0)) * (BK * 2)))
#define PP_DMA_X(ptr, dst) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ptr), (lds_ptr_t)(dst), 16, 0, 0)
// first LDS row of this wave's 8-row DMA group: A half h, instruction i (i < MI / 2); W half h, instruction i (i < 2)
#define PP_AROW0(h, i) (MI == 4 ? ((i) * 128 + (h) * 64 + wave * 8) : ((wave >> 2) * 64 + (h) * 32 + (wave & 3) * 8))
#define PP_WROW0(h, i) (((i) * 2 + (wave >> 2)) * 64 + (h) * 32 + (wave & 3) * 8)
// NB: the LDS address handed to the DMA builtins must not be a value-dependent expression of a template parameter (hipcc 7.2 then
// silently drops the kernel's host stub): the per-wave LDS offsets live in the runtime tables a_lds / w_lds.
// cache-policy bits of the DMA loads (experiment switch, side builds only: -DPP_AUX_A=n -DPP_AUX_W=n; bit 0 sc0, bit 1 nt, bit 4 sc1)
#ifndef PP_AUX_A
#define PP_AUX_A 0
#endif
#ifndef PP_AUX_W
#define PP_AUX_W 0
#endif
#define PP_BL(rs, dst, voff, tt, aux) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst), 16, voff, PP_KOFF(tt), 0, aux)
#define PP_ISSUE_A(h, tt, base)                                                  \
  do {                                                                           \
    PP_BL(rsrc_a, (base) + a_lds[h][0], a_off[h][0], tt, PP_AUX_A);                        \
    if constexpr (MI == 4) PP_BL(rsrc_a, (base) + a_lds[h][1], a_off[h][1], tt, PP_AUX_A); \
  } while (0)
#define PP_ISSUE_W(h, tt, base)                                                  \
  do {                                                                           \
    PP_BL(rsrc_w, (base) + w_lds[h][0], w_off[h][0], tt, PP_AUX_W);                        \
    PP_BL(rsrc_w, (base) + w_lds[h][1], w_off[h][1], tt, PP_AUX_W);                        \
  } while (0)
// the optional extension tile (A2, W2) is K-tile 0, fetched in the prologue with the same lane -> (row, chunk) mapping
#define PP_ISSUE_AX(h, base)                                                     \
  do {                                                                           \
    PP_DMA_X(ext_a(h, 0), (base) + a_lds[h][0]);                                 \
    if constexpr (MI == 4) PP_DMA_X(ext_a(h, 1), (base) + a_lds[h][1]);          \
  } while (0)
#define PP_ISSUE_WX(h, base)                                                     \
  do {                                                                           \
    PP_DMA_X(ext_w(h, 0), (base) + w_lds[h][0]);                                 \
    PP_DMA_X(ext_w(h, 1), (base) + w_lds[h][1]);                                 \
  } while (0)
#define PP_VMI(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// counted waits: FULL = four half-tiles may stay in flight (2 A + 2 W = MI + 4 instructions), then the drain ladder
#define PP_VM_FULL do { if constexpr (MI == 4) PP_VMI(8); else PP_VMI(6); } while (0)
#define PP_VM_WA do { if constexpr (MI == 4) PP_VMI(4); else PP_VMI(3); } while (0)     /* W + A half-tile in flight */
#define PP_VM_A do { if constexpr (MI == 4) PP_VMI(2); else PP_VMI(1); } while (0)      /* one A half-tile in flight */
#define PP_VM_0 PP_VMI(0)
#define PP_NOP ((void)0)
#define PP_PHASE(READS, ISSUE, WAIT, MMA)               \
  do {                                                  \
    READS; ISSUE; WAIT;                                 \
    __builtin_amdgcn_sched_barrier(0);                  \
    __builtin_amdgcn_s_barrier();                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                  \
    __builtin_amdgcn_s_setprio(1);                      \
    MMA;                                                \
    __builtin_amdgcn_s_setprio(0);                      \
    __builtin_amdgcn_sched_barrier(0);                  \
    __builtin_amdgcn_s_barrier();                       \
    __builtin_amdgcn_sched_barrier(0);                  \
  } while (0)

constexpr int NTB = 512;

// 16x16x32 fragments: lane -> (row lane&15, 16-byte k chunk lane>>4) of a 16-row block, two k-steps of 32 per K-tile
#define P16_READ_W(dst, j, base)                                                                                              \
  _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                         \
      dst[nb][ks] = *reinterpret_cast<const bf16x8_t*>((base) + A_BYTES + lds_off(((j) ? wrow1 : wrow0) + nb * 16 + frow, ks * 4 + fq))
#define P16_READ_A(mb0, base)                                                                                                 \
  _Pragma("unroll") for (int mb = 0; mb < MI; ++mb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                        \
      af[mb][ks] = *reinterpret_cast<const bf16x8_t*>((base) + lds_off(wm * (32 * MI) + ((mb0) + mb) * 16 + frow, ks * 4 + fq))
#define P16_MMA(wfx, nb0, mb0)                                                                                                \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)                         \
      _Pragma("unroll") for (int mb = 0; mb < MI; ++mb)                                                                       \
          acc[(nb0) + nb][(mb0) + mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfx[nb][ks], af[mb][ks], acc[(nb0) + nb][(mb0) + mb], 0, 0, 0)

template <bool OUT_F32, bool EXT, int MI, int FX = FX_NONE>
__global__ __launch_bounds__(NTB, 2) void gemm_bf16_tn_pp_kernel(GemmP p) {
  constexpr int BMB = 64 * MI, BNB = 256;
  constexpr int BN_STEP = FX == FX_SWIGLU ? 128 : BNB;        // FX_SWIGLU: a tile is 128 gate columns + the 128 up columns of the same index (see epilogue_fx)
  constexpr int A_BYTES = BMB * BK * 2, W_BYTES = BNB * BK * 2, BUF = A_BYTES + W_BYTES;     // 64 KiB (MI 4) / 48 KiB (MI 2) per buffer
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  int bid = blockIdx.x;
  const int nwg = p.tiles_m * p.tiles_n;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int len = q + (xcd < r ? 1 : 0);
    const int idx = ((bid >> 3) + xcd * p.skew) % len;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int per_group = p.group_m * p.tiles_n;
  const int first_m = (bid / per_group) * p.group_m;
  const int gsz = min(p.tiles_m - first_m, p.group_m);
  const int m0 = (first_m + (bid % per_group) % gsz) * BMB;
  const int n0 = ((bid % per_group) / gsz) * BN_STEP;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const long b1 = blockIdx.y % p.batch1, b2 = blockIdx.y / p.batch1;
  const long bz = b1 * p.sC + b2 * p.sC2;
  const bf16_t* __restrict__ Ag = p.A + b1 * p.sA + b2 * p.sA2;
  const bf16_t* __restrict__ Wg = p.W + b1 * p.sW + b2 * p.sW2;

  // per-lane DMA sources: [half][instruction]; the lane's LDS slot is (row0 + lane/8, chunk lane%8), it fetches global chunk
  // (lane%8) ^ swizzle(row) of that row.  Byte offsets relative to the tile's first row (rows past the matrix edge are clamped).
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ag + (long)m0 * p.lda), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Wg + (long)n0 * p.ldw), 0, 0x7fffffff, 0x00020000);
  int a_off[2][2], w_off[2][2];      // per-lane global byte offsets
  int a_lds[2][2], w_lds[2][2];      // LDS byte offset of the wave's 8-row DMA group inside a buffer (wave-uniform)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra0 = i < MI / 2 ? PP_AROW0(h, i) : 0, ra = ra0 + (lane >> 3);
      a_lds[h][i] = ra0 * 128;
      a_off[h][i] = (int)(((long)min(ra, p.M - 1 - m0) * p.lda + (((lane & 7) ^ ((ra >> 1) & 7)) << 3)) * 2);
      // fused epilogues: a W half-tile is the set of rows the waves read as their FIRST / SECOND 32-column block (what phases p0 / p1 read), so with
      // the paired wave -> row mapping half h = rows with (row % 128) / 64 == h (RoPE) or row / 128 == h (SwiGLU); the DMA ring's timing is unchanged
      const int rw0 = FX == FX_ROPE ? i * 128 + h * 64 + wave * 8 : FX == FX_SWIGLU ? h * 128 + i * 64 + wave * 8 : PP_WROW0(h, i), rw = rw0 + (lane >> 3);
      w_lds[h][i] = A_BYTES + rw0 * 128;
      const int gw = (FX == FX_SWIGLU && rw >= 128) ? rw - 128 + p.fx_I : rw;
      w_off[h][i] = (int)(((long)min(gw, p.N - 1 - n0) * p.ldw + (((lane & 7) ^ ((rw >> 1) & 7)) << 3)) * 2);
    }
  }
  int nt_main = p.K / BK;
  if (p.kt_total > 0) nt_main = min(nt_main, p.kt_total - (int)b1 * nt_main);     // split-K: the last slice may be shorter
  const int nt = nt_main + (EXT ? 1 : 0);
  auto ext_a = [&](int h, int i) {
    const int ra = PP_AROW0(h, i) + (lane >> 3);
    return p.A2 + (long)min(m0 + ra, p.M - 1) * p.lda2 + (((lane & 7) ^ ((ra >> 1) & 7)) << 3);
  };
  auto ext_w = [&](int h, int i) {
    const int rw = (FX == FX_ROPE ? i * 128 + h * 64 + wave * 8 : PP_WROW0(h, i)) + (lane >> 3);
    return p.W2 + (long)min(n0 + rw, p.N - 1) * p.ldw2 + (((lane & 7) ^ ((rw >> 1) & 7)) << 3);
  };

  // v_mfma_f32_16x16x32_bf16: 4 MI MFMAs over 2 MI independent accumulators per phase (the 32x32x16 form alternates two, and a
  // dependent MFMA issues only every ~42 cycles): acc[nb][mb] = C block (n = 16 nb.., m = 16 mb..), 4 x 2MI blocks of f32x4
  f32x4_t acc[4][2 * MI];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int mb = 0; mb < 2 * MI; ++mb) acc[nb][mb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[MI][2], wf0[2][2], wf1[2][2];
  const int frow = lane & 15, fq = lane >> 4;
  // first W-tile row of this wave's two 32-column halves: 64 consecutive columns, or (fused epilogues) two blocks a pair distance apart
  const int wrow0 = FX == FX_ROPE ? (wn >> 1) * 128 + (wn & 1) * 32 : FX == FX_SWIGLU ? wn * 32 : wn * 64;
  const int wrow1 = wrow0 + (FX == FX_ROPE ? 64 : FX == FX_SWIGLU ? 128 : 32);

  // prologue: S[0..5] = A0(0) W0(0) W1(0) A1(0) A0(1) W0(1)
  if (EXT) { PP_ISSUE_AX(0, smem); PP_ISSUE_WX(0, smem); PP_ISSUE_WX(1, smem); PP_ISSUE_AX(1, smem); }
  else { PP_ISSUE_A(0, 0, smem); PP_ISSUE_W(0, 0, smem); PP_ISSUE_W(1, 0, smem); PP_ISSUE_A(1, 0, smem); }
  PP_ISSUE_A(0, 1, smem + BUF); PP_ISSUE_W(0, 1, smem + BUF);
  PP_VM_FULL;
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();            // the stagger
  __builtin_amdgcn_sched_barrier(0);

  int t = 0;
  for (; t < nt - 2; ++t) {
    char* cur = smem + (t & 1) * BUF;
    char* oth = smem + ((t + 1) & 1) * BUF;
    PP_PHASE(P16_READ_W(wf0, 0, cur); P16_READ_A(0, cur), PP_ISSUE_W(1, t + 1, oth), PP_VM_FULL, P16_MMA(wf0, 0, 0));
    PP_PHASE(P16_READ_W(wf1, 1, cur), PP_ISSUE_A(1, t + 1, oth), PP_VM_FULL, P16_MMA(wf1, 2, 0));
    PP_PHASE(P16_READ_A(MI, cur), PP_ISSUE_A(0, t + 2, cur), PP_NOP, P16_MMA(wf1, 2, MI));
    PP_PHASE(PP_NOP, PP_ISSUE_W(0, t + 2, cur), PP_VM_FULL, P16_MMA(wf0, 0, MI));
  }
  {   // K-tile nt-2: the last two half-tiles are issued, then the queue drains
    char* cur = smem + (t & 1) * BUF;
    char* oth = smem + ((t + 1) & 1) * BUF;
    PP_PHASE(P16_READ_W(wf0, 0, cur); P16_READ_A(0, cur), PP_ISSUE_W(1, t + 1, oth), PP_VM_FULL, P16_MMA(wf0, 0, 0));
    PP_PHASE(P16_READ_W(wf1, 1, cur), PP_ISSUE_A(1, t + 1, oth), PP_VM_FULL, P16_MMA(wf1, 2, 0));
    PP_PHASE(P16_READ_A(MI, cur), PP_NOP, PP_NOP, P16_MMA(wf1, 2, MI));
    PP_PHASE(PP_NOP, PP_NOP, PP_VM_WA, P16_MMA(wf0, 0, MI));
    cur = oth;   // K-tile nt-1
    PP_PHASE(P16_READ_W(wf0, 0, cur); P16_READ_A(0, cur), PP_NOP, PP_VM_A, P16_MMA(wf0, 0, 0));
    PP_PHASE(P16_READ_W(wf1, 1, cur), PP_NOP, PP_VM_0, P16_MMA(wf1, 2, 0));
    PP_PHASE(P16_READ_A(MI, cur), PP_NOP, PP_NOP, P16_MMA(wf1, 2, MI));
    PP_PHASE(PP_NOP, PP_NOP, PP_NOP, P16_MMA(wf0, 0, MI));
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();            // re-join: every wave's reads and DMA are retired past this point
  if constexpr (FX != FX_NONE) epilogue_fx<MI, FX>(p, Acc16<MI>{acc}, smem, wave, m0, n0, wm, wn, lane);
  else epilogue_lds<OUT_F32, MI>(p, Acc16<MI>{acc}, smem, wave, m0, n0, wm, wn, lane, bz);
}

// ---- variant Q2: 128 x 256 tile, TWO phases per K-tile, THREE LDS buffers -------------------------------------------------------
// The 128 x 256 form of the ping-pong kernel above spends four phases of 8 MFMAs (128 matrix-pipe cycles per wave) per K-tile; a phase's
// fixed cost (two barriers, the LDS-DMA issue, the exposed part of the fragment-read latency: ~130-190 cycles measured per phase on both
// tile heights) is then as long as its MFMA burst -- 1.0 us per K-tile against 0.49 us of matrix time.  Here a K-tile is TWO phases of 16
// MFMAs: phase A = {W0, W1} x A0 (12 fragment reads), phase B = {W1, W0} x A1 (4 reads); half as many barriers per K-tile, the same DMA
// instruction count.  A half-tile can then no longer be re-issued two phases after its last read inside a two-buffer ring, so the ring is
// three whole K-tiles (3 x 48 KiB = 144 KiB of the CU's 160 KiB): tile t + 2 is issued into the buffer tile t - 1 was read from, half of it
// in each phase (A: A0 + W0, B: W1 + A1 -- each region >= 2 phases after its last read, the lagging group's reads included), and one
// counted wait per K-tile (phase B: vmcnt(6) = tile t + 2 may stay in flight, tile t + 1 is complete) retires what phase A of the next
// tile reads, one barrier later.
template <bool OUT_F32, bool EXT, int FX = FX_NONE>
__global__ __launch_bounds__(NTB, 2) void gemm_bf16_tn_pp2_kernel(GemmP p) {
  constexpr int MI = 2;
  constexpr int BMB = 128, BNB = 256;
  constexpr int BN_STEP = FX == FX_SWIGLU ? 128 : BNB;        // FX_SWIGLU: a tile is 128 gate columns + the 128 up columns of the same index
  constexpr int A_BYTES = BMB * BK * 2, W_BYTES = BNB * BK * 2, BUF = A_BYTES + W_BYTES;     // 48 KiB per buffer
  __shared__ __attribute__((aligned(16))) char smem[3 * BUF];
  int bid = blockIdx.x;
  const int nwg = p.tiles_m * p.tiles_n;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int len = q + (xcd < r ? 1 : 0);
    const int idx = ((bid >> 3) + xcd * p.skew) % len;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int per_group = p.group_m * p.tiles_n;
  const int first_m = (bid / per_group) * p.group_m;
  const int gsz = min(p.tiles_m - first_m, p.group_m);
  const int m0 = (first_m + (bid % per_group) % gsz) * BMB;
  const int n0 = ((bid % per_group) / gsz) * BN_STEP;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const long b1 = blockIdx.y % p.batch1, b2 = blockIdx.y / p.batch1;
  const long bz = b1 * p.sC + b2 * p.sC2;
  const bf16_t* __restrict__ Ag = p.A + b1 * p.sA + b2 * p.sA2;
  const bf16_t* __restrict__ Wg = p.W + b1 * p.sW + b2 * p.sW2;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ag + (long)m0 * p.lda), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Wg + (long)n0 * p.ldw), 0, 0x7fffffff, 0x00020000);
  int a_off[2][2], w_off[2][2];
  int a_lds[2][2], w_lds[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra0 = i < MI / 2 ? PP_AROW0(h, i) : 0, ra = ra0 + (lane >> 3);
      a_lds[h][i] = ra0 * 128;
      a_off[h][i] = (int)(((long)min(ra, p.M - 1 - m0) * p.lda + (((lane & 7) ^ ((ra >> 1) & 7)) << 3)) * 2);
      const int rw0 = PP_WROW0(h, i), rw = rw0 + (lane >> 3);
      w_lds[h][i] = A_BYTES + rw0 * 128;
      // FX_SWIGLU: tile rows 128 .. 255 are the up_proj rows fx_I + n0 .. of the gate|up weight (n0 = 128 x the tile's column index)
      const int gw = (FX == FX_SWIGLU && rw >= 128) ? rw - 128 + p.fx_I : rw;
      w_off[h][i] = (int)(((long)min(gw, p.N - 1 - n0) * p.ldw + (((lane & 7) ^ ((rw >> 1) & 7)) << 3)) * 2);
    }
  }
  int nt_main = p.K / BK;
  if (p.kt_total > 0) nt_main = min(nt_main, p.kt_total - (int)b1 * nt_main);
  const int nt = nt_main + (EXT ? 1 : 0);
  auto ext_a = [&](int h, int i) {
    const int ra = PP_AROW0(h, i) + (lane >> 3);
    return p.A2 + (long)min(m0 + ra, p.M - 1) * p.lda2 + (((lane & 7) ^ ((ra >> 1) & 7)) << 3);
  };
  auto ext_w = [&](int h, int i) {
    const int rw = PP_WROW0(h, i) + (lane >> 3);
    return p.W2 + (long)min(n0 + rw, p.N - 1) * p.ldw2 + (((lane & 7) ^ ((rw >> 1) & 7)) << 3);
  };

  f32x4_t acc[4][2 * MI];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int mb = 0; mb < 2 * MI; ++mb) acc[nb][mb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[MI][2], wf0[2][2], wf1[2][2];
  const int frow = lane & 15, fq = lane >> 4;
  // first W-tile row of this wave's two 32-column halves: 64 consecutive columns, or (fused epilogues) two blocks a pair distance apart
  const int wrow0 = FX == FX_ROPE ? (wn >> 1) * 128 + (wn & 1) * 32 : FX == FX_SWIGLU ? wn * 32 : wn * 64;
  const int wrow1 = wrow0 + (FX == FX_ROPE ? 64 : FX == FX_SWIGLU ? 128 : 32);

  char* cur = smem;               // K-tile t
  char* nxt = smem + BUF;         // K-tile t + 1
  char* nn = smem + 2 * BUF;      // K-tile t + 2 (= the buffer K-tile t - 1 was read from)
  // prologue: K-tiles 0 and 1 whole (6 + 6 DMA instructions per wave); K-tile 0 must have landed before the first barrier
  if (EXT) { PP_ISSUE_AX(0, cur); PP_ISSUE_WX(0, cur); PP_ISSUE_WX(1, cur); PP_ISSUE_AX(1, cur); }
  else { PP_ISSUE_A(0, 0, cur); PP_ISSUE_W(0, 0, cur); PP_ISSUE_W(1, 0, cur); PP_ISSUE_A(1, 0, cur); }
  PP_ISSUE_A(0, 1, nxt); PP_ISSUE_W(0, 1, nxt); PP_ISSUE_W(1, 1, nxt); PP_ISSUE_A(1, 1, nxt);
  PP_VMI(6);
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();            // the stagger
  __builtin_amdgcn_sched_barrier(0);

#define PP2_READ_A P16_READ_W(wf0, 0, cur); P16_READ_W(wf1, 1, cur); P16_READ_A(0, cur)
#define PP2_MMA_A P16_MMA(wf0, 0, 0); P16_MMA(wf1, 2, 0)
#define PP2_MMA_B P16_MMA(wf1, 2, MI); P16_MMA(wf0, 0, MI)
  int t = 0;
  for (; t < nt - 2; ++t) {
    PP_PHASE(PP2_READ_A, PP_ISSUE_A(0, t + 2, nn); PP_ISSUE_W(0, t + 2, nn), PP_NOP, PP2_MMA_A);
    PP_PHASE(P16_READ_A(MI, cur), PP_ISSUE_W(1, t + 2, nn); PP_ISSUE_A(1, t + 2, nn), PP_VMI(6), PP2_MMA_B);
    char* const tmp = cur; cur = nxt; nxt = nn; nn = tmp;
  }
  // K-tile nt - 2: nothing left to issue, the queue drains (K-tile nt - 1 complete before its phase A)
  PP_PHASE(PP2_READ_A, PP_NOP, PP_NOP, PP2_MMA_A);
  PP_PHASE(P16_READ_A(MI, cur), PP_NOP, PP_VM_0, PP2_MMA_B);
  cur = nxt;
  PP_PHASE(PP2_READ_A, PP_NOP, PP_NOP, PP2_MMA_A);
  PP_PHASE(P16_READ_A(MI, cur), PP_NOP, PP_NOP, PP2_MMA_B);
#undef PP2_READ_A
#undef PP2_MMA_A
#undef PP2_MMA_B
  if (wm == 0) __builtin_amdgcn_s_barrier();            // re-join: every wave's reads and DMA are retired past this point
  if constexpr (FX != FX_NONE) epilogue_fx<MI, FX>(p, Acc16<MI>{acc}, smem, wave, m0, n0, wm, wn, lane);
  else epilogue_lds<OUT_F32, MI>(p, Acc16<MI>{acc}, smem, wave, m0, n0, wm, wn, lane, bz);
}

// ---- variant V: skinny GEMM, M <= 8 rows (the decode step of generation: one token per sequence) ---------------------------------
// C[m][n] = epi(alpha * sum_k A[m][k] W[n][k]) is a WEIGHT STREAM: every W row is read once (13.2 GB per Llama-7B token), the few A
// rows come from L1 / L2.  HBM-bound, so no MFMA and no LDS: one wave owns 4 consecutive W rows, its 64 lanes walk K in 16-byte
// chunks (1 KiB per row per instruction, fully coalesced), two K-steps in flight per lane (8 W loads + 2 MT A loads), fp32 FMA
// accumulators acc[4][MT], a 6-step butterfly at the end, lane (r, m) applies the epilogue and stores.  Workgroup = 4 waves = 16 W rows:
// N = 4096 gives 256 workgroups (one per CU, 40 KiB of loads in flight each).  K % 8 == 0, K-contiguous A and W, batch == 1.
// AT: transform applied to the A rows on load, so that a decode step needs no separate launch for it (each wave walks ALL of K, so it
// sees whole rows):  1 = A := RMSNorm(A) * a_norm_w with llmseg_norm's arithmetic (fp32 sum of squares, bf16 rounding before AND
// after the weight multiply, as the stored bf16 output of the norm kernel has);  2 = A := silu(g) * u of rows [g | u] of width 2K
// (llmseg_swiglu's arithmetic, rounded to bf16).  Both reproduce the two-launch result bit for bit up to the order of the fp32 dot
// product, which is this kernel's own either way.
template <int MT, int AT, bool SK = false>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmP p, int out_f32) {
  constexpr int KS = MT >= 4 ? 1 : 2;                // K-steps (512 columns) per trip
  constexpr int STEP = KS * 512;
  constexpr int ADV = SK ? 4 * STEP : STEP;          // SK (few W rows: N <= 8192): the 4 waves of a workgroup share 4 rows and interleave the trips
  constexpr int XB = AT == 2 ? MT : 1;               // second operand of the A transform: the norm gain (one row) or the up half (per row)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = SK ? blockIdx.x * 4 : (blockIdx.x * 4 + wave) * 4;
  if (n0 >= p.N) return;
  const bf16_t* wr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) wr[r] = p.W + (long)min(n0 + r, p.N - 1) * p.ldw;
  float acc[4][MT];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[r][m] = 0.f;
  const int K = p.K;
  auto loadw = [&](uint4 (&w)[KS][4], int k) {
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) w[t][r] = *reinterpret_cast<const uint4*>(wr[r] + min(k + t * 512, K - 8));   // branch-free: past the end a
  };                                                                                                            // valid chunk is re-read, its A chunk is zero
  // raw A chunks of a trip (the transform runs at FMA time): xa = the 8 A values, xb = the norm gain / the up half
  auto loadx = [&](uint4 (&xa)[KS][MT], uint4 (&xb)[KS][XB], int k) {
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      const int kk = k + t * 512, kc = min(kk, K - 8);
      const bool ok = kk < K;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const bf16_t* ar = p.A + (long)min(m, p.M - 1) * p.lda;
        const uint4 v = *reinterpret_cast<const uint4*>(ar + kc);
        xa[t][m] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
        if constexpr (AT == 2) xb[t][m] = *reinterpret_cast<const uint4*>(ar + K + kc);
      }
      if constexpr (AT == 1) xb[t][0] = *reinterpret_cast<const uint4*>(p.a_norm_w + kc);
    }
  };
  float rstd[MT];
  auto xform = [&](const uint4& a, const uint4& b, int m) -> uint4 {
    if constexpr (AT == 0) return a;
    float f[8], g[8], o[8];
    unpack8(a, f);
    unpack8(b, g);
    if constexpr (AT == 1) {                           // gain * bf16(x * rstd), both roundings by v_cvt_pk_bf16_f32 (RNE, as f2bf)
      const float r = rstd[m];
      const uint4 xn = make_uint4(pack2bf(f[0] * r, f[1] * r), pack2bf(f[2] * r, f[3] * r), pack2bf(f[4] * r, f[5] * r), pack2bf(f[6] * r, f[7] * r));
      unpack8(xn, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = g[e] * f[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = swiglu_fwd1(f[e], g[e]);
    }
    return pack8(o);
  };
  auto fma8 = [&](const uint4& w, const uint4& x, float& a) {
    a = fmaf(__uint_as_float(w.x << 16), __uint_as_float(x.x << 16), a); a = fmaf(__uint_as_float(w.x & 0xffff0000u), __uint_as_float(x.x & 0xffff0000u), a);
    a = fmaf(__uint_as_float(w.y << 16), __uint_as_float(x.y << 16), a); a = fmaf(__uint_as_float(w.y & 0xffff0000u), __uint_as_float(x.y & 0xffff0000u), a);
    a = fmaf(__uint_as_float(w.z << 16), __uint_as_float(x.z << 16), a); a = fmaf(__uint_as_float(w.z & 0xffff0000u), __uint_as_float(x.z & 0xffff0000u), a);
    a = fmaf(__uint_as_float(w.w << 16), __uint_as_float(x.w << 16), a); a = fmaf(__uint_as_float(w.w & 0xffff0000u), __uint_as_float(x.w & 0xffff0000u), a);
  };
  auto compute = [&](const uint4 (&w)[KS][4], const uint4 (&xa)[KS][MT], const uint4 (&xb)[KS][XB]) {
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const uint4 x = xform(xa[t][m], xb[t][AT == 2 ? m : 0], m);
#pragma unroll
        for (int r = 0; r < 4; ++r) fma8(w[t][r], x, acc[r][m]);
      }
  };
  // two trips in flight: the loads of trip i + 1 are issued before the FMAs of trip i; the first W rows are requested before the
  // RMSNorm prologue, so that a cold A (written by the previous kernel on another XCD) does not delay the weight stream
  uint4 wA[KS][4], wB[KS][4], xA[KS][MT], xB[KS][MT <= 2 ? MT : 1], gA[KS][XB], gB[KS][MT <= 2 ? XB : 1];
  int k = lane * 8 + (SK ? wave * STEP : 0);
  loadw(wA, k);
  if constexpr (AT == 1) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const bf16_t* ar = p.A + (long)min(m, p.M - 1) * p.lda;
      float ss = 0.f, f[8];
      for (int kk = lane * 8 + (SK ? wave * 512 : 0); kk < K; kk += SK ? 2048 : 512) {
        unpack8(*reinterpret_cast<const uint4*>(ar + kk), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
      }
      rstd[m] = wave_sum(ss);
    }
    if constexpr (SK) {                              // the workgroup's waves share the rows: each sums a quarter of the columns
      __shared__ float ssq[4][MT];
      if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) ssq[wave][m] = rstd[m];
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < MT; ++m) rstd[m] = (ssq[0][m] + ssq[1][m]) + (ssq[2][m] + ssq[3][m]);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) rstd[m] = rsqrtf(rstd[m] / (float)K + p.a_norm_eps);
  }
  const int trips_all = (K + STEP - 1) / STEP;
  const int trips = SK ? (trips_all - wave + 3) / 4 : trips_all;          // wave-uniform
  if constexpr (MT <= 2) {
    loadx(xA, gA, k);
    for (int t = 0; t < trips; t += 2) {
      loadw(wB, k + ADV);
      loadx(xB, gB, k + ADV);
      compute(wA, xA, gA);
      k += ADV;
      if (t + 1 >= trips) break;
      loadw(wA, k + ADV);
      loadx(xA, gA, k + ADV);
      compute(wB, xB, gB);
      k += ADV;
    }
  } else {                                           // 4-8 rows: A chunks (cache hits) single-buffered, requested ahead of the next W rows
    for (int t = 0; t < trips; t += 2) {             // (loads return in order: a later request would wait for the prefetch)
      loadx(xA, gA, k);
      loadw(wB, k + ADV);
      compute(wA, xA, gA);
      k += ADV;
      if (t + 1 >= trips) break;
      loadx(xA, gA, k);
      loadw(wA, k + ADV);
      compute(wB, xA, gA);
      k += ADV;
    }
  }
  float mine = 0.f;                                  // lane r * MT + m keeps C[m][n0 + r]
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float t = wave_sum(acc[r][m]);
      if (lane == r * MT + m) mine = t;
    }
  if constexpr (SK) {
    __shared__ float red[4][4 * MT];
    if (lane < 4 * MT) red[wave][lane] = mine;
    __syncthreads();
    if (wave != 0) return;
    if (lane < 4 * MT) mine = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  }
  if (lane < 4 * MT) {
    const int r = lane / MT, m = lane - r * MT, n = n0 + r;
    if (n < p.N && m < p.M) {
      float v = mine * p.alpha;
      if (p.bias) v += bf2f(p.bias[n]);
      v = apply_act(v, p.act);
      if (p.gamma) v *= bf2f(p.gamma[n]);
      if (p.res) v += bf2f(p.res[(long)m * p.ldr + n]);
      if (out_f32) {
        float* cp = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
        *cp = p.accum ? *cp + v : v;
      } else reinterpret_cast<bf16_t*>(p.C)[(long)m * p.ldc + n] = f2bf(v);
    }
  }
}

// Split-K tail: C = epi(alpha * sum_s slab[s]) with the GEMM's epilogue (bias, activation, LayerScale, residual, bf16 or fp32 out,
// optional += into an fp32 C).  slab: fp32 [S][M][N] (dense), one thread per 4 consecutive columns (N % 4 == 0).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmP p, const float* __restrict__ slab, int S, int out_f32) {
  const long n4 = p.N >> 2, total = (long)p.M * n4, slab_sz = (long)p.M * p.N;
  slab += (long)blockIdx.y * S * slab_sz;                      // strided-batched product: entry y has its own S slabs and its own C (no residual)
  const long cz = (long)blockIdx.y * p.sC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / n4;
    const int n = (int)(i - m * n4) * 4;
    const float* sp = slab + m * p.N + n;
    float4 a = *reinterpret_cast<const float4*>(sp);
    for (int s = 1; s < S; ++s) {
      const float4 b = *reinterpret_cast<const float4*>(sp + s * slab_sz);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x * p.alpha, a.y * p.alpha, a.z * p.alpha, a.w * p.alpha};
    if (p.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bf2f(p.bias[n + e]);
    }
    if (p.act != LLMSEG_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
    }
    if (p.gamma) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= bf2f(p.gamma[n + e]);
    }
    if (p.res) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bf2f(p.res[m * p.ldr + n + e]);
    }
    if (out_f32) {
      float* cp = reinterpret_cast<float*>(p.C) + cz + m * p.ldc + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) cp[e] = p.accum ? cp[e] + v[e] : v[e];
    } else {
      bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + cz + m * p.ldc + n;
      if (p.c_vec) *reinterpret_cast<uint2*>(cp) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) cp[e] = f2bf(v[e]);
      }
    }
  }
}

// Split-K tail of a residual-stream product WITH the stream's next pre-norm (llmseg_gemm_args.norm_out, round 5): a workgroup per output row sums the row's S
// slab rows, adds the residual, stores the bf16 row x -- splitk_reduce_kernel's arithmetic for alpha = 1, no bias / activation / LayerScale -- and, holding that
// row in registers, writes RMSNorm(x) * w with norm_wg_kernel<CPT>'s arithmetic and the SAME thread -> column mapping (chunk c = thread + 256 i: identical partial
// sums, identical block reduction), so both outputs carry the bits the two-launch route produces.  One launch and one pass over the row instead of two.
template <int CPT>
__global__ __launch_bounds__(256) void splitk_reduce_rmsnorm_kernel(GemmP p, const float* __restrict__ slab, int S, const bf16_t* __restrict__ nw, float eps,
                                                                   bf16_t* __restrict__ nout, long ldn) {
  __shared__ float red[16];
  const long m = blockIdx.x;
  const int nch = p.N >> 3;
  const long slab_sz = (long)p.M * p.N;
  uint4 xc[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = threadIdx.x + 256 * i;
    xc[i] = make_uint4(0, 0, 0, 0);
    if (c < nch) {
      const int n = c * 8;
      const float* sp = slab + m * p.N + n;
      float4 a0 = *reinterpret_cast<const float4*>(sp), a1 = *reinterpret_cast<const float4*>(sp + 4);
      for (int s2 = 1; s2 < S; ++s2) {
        const float4 b0 = *reinterpret_cast<const float4*>(sp + s2 * slab_sz), b1 = *reinterpret_cast<const float4*>(sp + s2 * slab_sz + 4);
        a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
        a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
      }
      float v[8] = {a0.x * p.alpha, a0.y * p.alpha, a0.z * p.alpha, a0.w * p.alpha, a1.x * p.alpha, a1.y * p.alpha, a1.z * p.alpha, a1.w * p.alpha};
      if (p.res) {
        float r[8];
        unpack8(*reinterpret_cast<const uint4*>(p.res + m * p.ldr + n), r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
      }
      xc[i] = pack8(v);
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + n) = xc[i];
    }
  }
  float f[8], v2 = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    if (threadIdx.x + 256 * i < nch) {
      unpack8(xc[i], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = f[e] - 0.f; v2 += d * d; }
    }
  }
  v2 = block_sum(v2, red);
  const float rstd = rsqrtf(v2 / (float)p.N + eps);
  bf16_t* yr = nout + m * ldn;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < nch) {
      float g[8], o[8];
      unpack8(xc[i], f);
      unpack8(*reinterpret_cast<const uint4*>(nw + c * 8), g);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = g[e] * bf2f(f2bf(f[e] * rstd));      // HF LlamaRMSNorm: round before the weight multiply
      *reinterpret_cast<uint4*>(yr + c * 8) = pack8(o);
    }
  }
}

// Split-K tail of dX(o_proj) with the attention backward's row statistic (llmseg_gemm_args.dl_o, round 6): a workgroup per row sums the row's S slab rows and
// stores the bf16 row dO (splitk_reduce_kernel's arithmetic, no bias / activation / residual) and, holding it, writes delta[b][h][q] = sum_d dO[q][h][d] O[q][h][d]
// for every head of width 128 with attn_delta_kernel's arithmetic (16 lanes per head, one 8-element chunk each, fma in element order, xor-8/4/2/1 butterfly):
// the bits `llmseg_attn_bwd`'s own delta launch would produce -- which it then skips (`delta_ready`).  Row m = b T + q.
template <int CPT>
__global__ __launch_bounds__(256) void splitk_reduce_delta_kernel(GemmP p, const float* __restrict__ slab, int S, const bf16_t* __restrict__ O, long ldo,
                                                                 float* __restrict__ delta, int heads, int T) {
  const long m = blockIdx.x;
  const int nch = p.N >> 3;
  const long slab_sz = (long)p.M * p.N;
  const long b = m / T, q = m - b * T;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < nch) {                                              // (nch is a multiple of 16: whole 16-lane groups are in or out together)
      const int n = c * 8;
      const float* sp = slab + m * p.N + n;
      float4 a0 = *reinterpret_cast<const float4*>(sp), a1 = *reinterpret_cast<const float4*>(sp + 4);
      for (int s2 = 1; s2 < S; ++s2) {
        const float4 b0 = *reinterpret_cast<const float4*>(sp + s2 * slab_sz), b1 = *reinterpret_cast<const float4*>(sp + s2 * slab_sz + 4);
        a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
        a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
      }
      const float v[8] = {a0.x * p.alpha, a0.y * p.alpha, a0.z * p.alpha, a0.w * p.alpha, a1.x * p.alpha, a1.y * p.alpha, a1.z * p.alpha, a1.w * p.alpha};
      const uint4 g = pack8(v);
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + n) = g;
      float of[8], df[8];
      unpack8(*reinterpret_cast<const uint4*>(O + m * ldo + n), of);
      unpack8(g, df);
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) part = fmaf(of[e], df[e], part);
#pragma unroll
      for (int mm = 8; mm >= 1; mm >>= 1) part += __shfl_xor(part, mm, 64);
      if ((c & 15) == 0) delta[(b * heads + (c >> 4)) * T + q] = part;
    }
  }
}

// a pending second output of the call being dispatched on this thread (llmseg_gemm_args.norm_out): the K-sliced ping-pong route consumes it in its reduce launch
struct NormReq { const bf16_t* w; bf16_t* out; long ldn; float eps; bool active, done; };
static thread_local NormReq g_norm_req = {nullptr, nullptr, 0, 0.f, false, false};
// a pending norm-backward tail of the call being dispatched on this thread (llmseg_gemm_args.nb_x): the K-sliced ping-pong route consumes it in its reduce launch
struct NbReq { const llmseg_gemm_args* a; void* out; bool active, done; };
static thread_local NbReq g_nb_req = {nullptr, nullptr, false, false};
// a pending delta tail (llmseg_gemm_args.dl_o) of the call being dispatched on this thread
struct DlReq { const llmseg_gemm_args* a; bool active, done; };
static thread_local DlReq g_dl_req = {nullptr, false, false};

template <bool OUT_F32, int MI, int NBUF>
void launch_glds(const GemmP& p, dim3 grid, hipStream_t s) {
  LL_LAUNCH_KERNEL((gemm_bf16_tn_glds_kernel<OUT_F32, MI, NBUF>), grid, dim3(NT), 0, s, p);
}

}  // namespace

// profiling hooks (capi.cpp)
void llmseg_prof_begin(hipStream_t s);
void llmseg_prof_end(hipStream_t s, double flops);
void llmseg_prof_tag(long a, long b, long c, long d);

// tuning knob (tools/gemm_bench.py): bits 0-3 kernel (0 = register staging 128x128; 2 = LDS-DMA 128x128; 3 = four-stage LDS-DMA 128x128; 8 / 9 = LDS-DMA ping-pong
// 256x256 / 128x256; 5 (default) = cost model), bits 4-7 = XCD skew + 1, bits 8-12 = forced split-K slice count for 8 / 9.
static int g_gemm_variant = 5, g_gemm_skew = 13, g_gemm_split = 0, g_gemm_pp2 = getenv("LLMSEG_GEMM_PP2") ? atoi(getenv("LLMSEG_GEMM_PP2")) : 1;
static const int g_gemm_rsplit = getenv("LLMSEG_GEMM_NO_RSPLIT") ? 0 : 1;      // K-slices for the register-staging kernel (A/B switch)
static int num_cus() {
  static int n = [] { int dev = 0, v = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v > 0 ? v : 256; }();
  return n;
}
extern "C" int llmseg_gemm_set_variant(int v) {
  g_gemm_variant = v & 15;
  if ((v >> 4) & 15) g_gemm_skew = ((v >> 4) & 15) - 1;
  g_gemm_split = (v >> 8) & 31;
  g_gemm_pp2 = (v >> 13) & 3 ? ((v >> 13) & 3) - 1 : g_gemm_pp2;     // bits 13-14: 128 x 256 kernel form + 1 (1 = four phases / two buffers, 2 = two phases / three buffers, 3 = loader waves)
  return LLMSEG_OK;
}

namespace {
// Cost model of the K % 64 == 0 kernels (microseconds; constants fitted to tools/gemm_bench.py on MI355X, profiles/r02*_gemm*.txt).
// One ping-pong workgroup owns a CU: a K-tile of the 256 x 256 kernel takes ~1.7 us (1.25 PF/s over 256 CUs), of the 128 x 256
// kernel ~1.0 us; prologue + epilogue ~7 / 4.5 us.  The 128 x 128 kernel shares a CU between up to 4 workgroups (2.4 us per K-tile
// each when all four are resident, latency-bound 1.3 us when alone).
struct GemmPlan { int variant, split; double us; };
inline double pp_cost(long M, long N, int nt, int mi, int S, long ncu, bool f32out) {
  const long tiles = ((M + 64 * mi - 1) / (64 * mi)) * ((N + 255) / 256);
  const long rounds = (tiles * S + ncu - 1) / ncu;
  const int q = (nt + S - 1) / S;
  // (round 5: re-fitting these constants to the two-phase kernel from two isolated shapes -- 0.8 + 6.4 / 1.64 + 4.5 -- moved the K-slice plans of
  // the Llama N = 4096 shapes and cost 4 % at 2 images, 6 % at 24: measured and reverted, profiles/r05g_gemm_dispatch.md)
  const double it = mi == 4 ? 1.7 : 1.0, fix = (mi == 4 ? 7.0 : 4.5) + ((S > 1 || f32out) ? 1.0 : 0.0);
  // Round 5: without K-slices a partially filled last round is priced at 0.5 + 0.5 x its fill instead of a whole round -- a CU that shares the
  // fabric with fewer neighbours fetches its operands faster (measured, tools/gemm_bench.py at the 2-image shapes: 8192 x 1280 x 1280 on 160
  // workgroups of 256 x 256 takes 35 us, not 41; the whole-round price made the 128 x 256 tile win SAM proj / lin1 / q|k|v-windows, where the
  // 256 x 256 tile measures +9 / +6 / +5 %).  K-sliced plans keep the whole-round price their slice counts were tuned with.
  double eff_rounds = (double)rounds;
  if (S == 1) {
    const long full = tiles / ncu, rem = tiles - full * ncu;
    eff_rounds = (double)full + (rem ? 0.5 + 0.5 * (double)rem / (double)ncu : 0.0);
  }
  double us = eff_rounds * (q * it + fix);
  if (S > 1) us += 2.5 + ((double)(S + 1) * M * N * 4.0) / 4.0e6;     // reduce launch: slabs read once (mostly from the Infinity Cache)
  return us;
}
inline double glds_cost(long M, long N, int nt, long ncu) {
  const long tiles = ((M + 127) / 128) * ((N + 127) / 128);
  const long full = tiles / (4 * ncu), rem = tiles - full * 4 * ncu;
  double us = (double)full * (nt * 2.4 + 4.0);
  if (rem > 0) { const double w = (double)((rem + ncu - 1) / ncu); us += nt * std::max(1.3, 0.6 * w) + 4.0; }
  return us;
}
inline bool split_ok(int nt, int S) {       // every slice needs >= 2 K-tiles (the kernel's pipeline depth)
  if (S <= 1) return S == 1;
  const int q = (nt + S - 1) / S;
  return q >= 2 && nt - (S - 1) * q >= 2;
}
}  // namespace

static int gemm_dispatch(const llmseg_gemm_args* a, void* stream, int force_variant);

extern "C" int llmseg_norm(const void* x, const void* w, const void* b, void* y, int64_t rows, int64_t cols, int64_t ldx, int64_t ldy, float eps, int rms,
                           const int32_t* row_map, void* stream);

extern "C" int llmseg_rope(void* x, const float* cos, const float* sin, int64_t rows, int64_t T, int32_t heads, int32_t head_dim, int64_t ld, void* stream);
extern "C" int llmseg_swiglu(const void* gu, void* out, int64_t rows, int64_t I, int64_t ldgu, int64_t ldo, void* stream);
extern "C" __attribute__((visibility("hidden"))) int llmseg_swiglu_bwd_ld(const void* gu, const void* dout, void* dgu, int64_t rows, int64_t I, int64_t ld_dout, void* stream);

static thread_local bool g_fx_done = false;        // set by gemm_dispatch when the fused-epilogue kernel ran for the call being dispatched on this thread
static const bool g_fx_off = getenv("LLMSEG_GEMM_NO_FX") != nullptr;      // A/B switch: always the GEMM + pointwise launch

// llmseg_gemm_args.fx: the fused kernel where the call takes the 128 x 256 two-phase kernel in one K-slice (the Llama layer at 2 images per micro-step),
// the GEMM followed by the pointwise launch it replaces everywhere else.  Same bits either way.
static int gemm_fx(const llmseg_gemm_args* a, void* stream) {
  LL_CHECK(a->fx >= 1 && a->fx <= 3, "gemm: unknown fx %d", a->fx);
  LL_CHECK(!a->out_f32 && a->batch <= 1 && a->batch2 <= 1 && a->alpha == 1.f && !a->bias && !a->gamma && !a->residual && a->act == LLMSEG_ACT_NONE && !a->norm_out &&
               !a->trans_a && !a->trans_w && !a->accumulate, "gemm: fx needs a plain bf16 product (batch 1, alpha 1, no bias / activation / gamma / residual / norm_out)");
  if (a->fx == LLMSEG_FX_ROPE) {
    LL_CHECK(a->fx_cos && a->fx_sin && a->fx_T > 0 && a->fx_cols > 0 && (a->fx_cols % 128) == 0 && a->fx_cols <= a->N && (a->ldc & 3) == 0 &&
                 ((((uintptr_t)a->fx_cos) | ((uintptr_t)a->fx_sin)) & 15) == 0 && (((uintptr_t)a->C) & 7) == 0,
             "gemm: fx rope needs fp32 [fx_T][64] tables (16-byte aligned), fx_cols a multiple of 128 (heads of width 128) and <= N");
  } else if (a->fx == LLMSEG_FX_SWIGLU) {
    LL_CHECK(a->fx_out && (a->N & 15) == 0 && (a->fx_ld & 7) == 0 && a->fx_ld >= a->N / 2 && (a->ldc & 7) == 0 && !a->A2 && ((((uintptr_t)a->fx_out) | ((uintptr_t)a->C)) & 15) == 0,
             "gemm: fx swiglu needs N = 2 I with I %% 8 == 0, fx_out bf16 [M][fx_ld >= I], 16-byte aligned rows, no extension operands");
  } else {
    LL_CHECK(a->fx_in && (a->N & 7) == 0 && (a->fx_ld & 7) == 0 && a->fx_ld >= 2 * a->N && a->ldc >= 2 * a->N && (a->ldc & 7) == 0 && !a->A2 &&
                 ((((uintptr_t)a->fx_in) | ((uintptr_t)a->C)) & 15) == 0,
             "gemm: fx swiglu_bwd needs N = I, fx_in = gate|up bf16 [M][fx_ld >= 2 I], C = d(gate|up) [M][ldc >= 2 I], 16-byte aligned rows");
  }
  g_fx_done = false;
  llmseg_gemm_args g = *a;
  if (a->fx == LLMSEG_FX_SWIGLU_BWD) g.C = (bf16_t*)a->C + a->N;      // unfused route: d(out) lands in the up half of C's rows, llmseg_swiglu_bwd then works in place
  const int rc = gemm_dispatch(&g, stream, -1);
  if (rc != LLMSEG_OK || g_fx_done) return rc;
  if (a->fx == LLMSEG_FX_ROPE) return llmseg_rope(a->C, a->fx_cos, a->fx_sin, a->M, a->fx_T, (int32_t)(a->fx_cols / 128), 128, a->ldc, stream);
  if (a->fx == LLMSEG_FX_SWIGLU) return llmseg_swiglu(a->C, a->fx_out, a->M, a->N / 2, a->ldc, a->fx_ld, stream);
  LL_CHECK(a->fx_ld == 2 * a->N && a->ldc == 2 * a->N, "gemm: fx swiglu_bwd on this shape (two-launch route) needs dense gate|up and d(gate|up) rows");
  return llmseg_swiglu_bwd_ld(a->fx_in, g.C, a->C, a->M, a->N, a->ldc, stream);
}

extern "C" __attribute__((visibility("hidden"))) int llmseg_reduce_lora_normbwd(const float* slab, int S, int64_t M, int64_t N, const void* x, const void* w, void* dx, float eps,
                                                                                int rms, const void* dres, void* la_t, int64_t la_ldt, const void* la_w0,
                                                                                const void* la_w1, float la_alpha, const llmseg_dropout* la_drop, const float* la_part,
                                                                                int la_S, float la_scale, int la_zero, void* stream);
extern "C" __attribute__((visibility("hidden"))) int llmseg_lora_down_finish(const float* part, int S, void* y, int64_t ldy, int64_t M, float scale, int zero_cols, int nb, void* stream);
extern "C" int llmseg_norm_bwd_add(const void* dy, const void* x, const void* w, const void* dres, void* dx, float* dw, float* db, int64_t rows, int64_t cols,
                                   float eps, int rms, void* workspace, int64_t workspace_bytes, void* stream);
extern "C" int llmseg_lora_apply(void* y, int64_t ldy, const void* xa, int64_t ldxa, const void* w0, const void* w1, int64_t M, int64_t N, int32_t w_rn,
                                 float alpha, const llmseg_dropout* drop, void* stream);
static const bool g_nb_off = getenv("LLMSEG_GEMM_NO_NB") != nullptr;      // A/B switch: always the product + lora_apply + norm_bwd launches

// llmseg_gemm_args.nb_x: C = norm_bwd(dy = the bf16 product [+ LoRA term], nb_x, nb_w) + nb_dres.  K-sliced products fold all of it into their reduce launch;
// every other route writes the product to the tail of the caller's workspace and runs llmseg_lora_apply / llmseg_norm_bwd_add behind it.  Same bits.
static int gemm_nb(const llmseg_gemm_args* a, void* stream) {
  LL_CHECK(a->nb_w && !a->out_f32 && a->batch <= 1 && a->batch2 <= 1 && a->alpha == 1.f && !a->bias && !a->gamma && !a->residual && a->act == LLMSEG_ACT_NONE &&
               !a->norm_out && !a->fx && !a->accumulate && (a->N & 7) == 0 && a->ldc == a->N,
           "gemm: nb_x needs a plain bf16 product (batch 1, alpha 1, no bias / activation / gamma / residual / norm_out / fx) with dense C rows (ldc == N)");
  LL_CHECK(((((uintptr_t)a->nb_x) | ((uintptr_t)a->nb_w) | ((uintptr_t)a->nb_dres) | ((uintptr_t)a->C)) & 15) == 0, "gemm: nb_x / nb_w / nb_dres / C must be 16-byte aligned");
  LL_CHECK(!a->nb_lora_t || (a->nb_lora_w0 && (a->nb_lora_ldt & 7) == 0), "gemm: nb_lora_t needs nb_lora_w0 ([8][N]) and a row pitch that is a multiple of 8");
  LL_CHECK(!a->nb_lora_part || (a->nb_lora_t && a->nb_lora_S >= 1 && a->nb_lora_zero >= 0 && (a->nb_lora_zero & 7) == 0 && a->nb_lora_ldt >= 16 + a->nb_lora_zero),
           "gemm: nb_lora_part needs nb_lora_t (the bf16 [M][>= 16 + zero] buffer the finished operand is written to), the slice count and the zero-column count");
  const int64_t tmp_bytes = ((a->M * a->N * 2 + 255) / 256) * 256;
  LL_CHECK(a->workspace && a->workspace_bytes >= tmp_bytes && (((uintptr_t)a->workspace) & 255) == 0, "gemm: nb_x needs a workspace of >= M N 2 bytes (256-byte aligned)");
  llmseg_gemm_args g = *a;
  g.workspace_bytes = (a->workspace_bytes - tmp_bytes) & ~(int64_t)255;
  void* tmp = (char*)a->workspace + g.workspace_bytes;
  g.C = tmp;                                                  // the two-launch routes leave the bf16 product here; the fused tail writes a->C itself
  g.nb_x = nullptr;
  g_nb_req = NbReq{a, a->C, !g_nb_off, false};
  const int rc = gemm_dispatch(&g, stream, -1);
  const bool done = g_nb_req.done;
  g_nb_req.active = false;
  if (rc != LLMSEG_OK || done) return rc;
  if (a->nb_lora_t && a->nb_lora_part) {          // the LoRA operand is still K-slice partials: finish them first (what the fused tail does in its own launch)
    const int rc0 = llmseg_lora_down_finish(a->nb_lora_part, a->nb_lora_S, (void*)a->nb_lora_t, a->nb_lora_ldt, a->M, a->nb_lora_scale, a->nb_lora_zero, a->nb_lora_w1 ? 2 : 1, stream);
    if (rc0 != LLMSEG_OK) return rc0;
  }
  if (a->nb_lora_t) {
    const int rc2 = llmseg_lora_apply(tmp, a->N, a->nb_lora_t, a->nb_lora_ldt, a->nb_lora_w0, a->nb_lora_w1, a->M, a->N, 1, a->nb_lora_alpha,
                                      (const llmseg_dropout*)a->nb_lora_drop, stream);
    if (rc2 != LLMSEG_OK) return rc2;
  }
  return llmseg_norm_bwd_add(tmp, a->nb_x, a->nb_w, a->nb_dres, a->C, nullptr, nullptr, a->M, a->N, a->nb_eps, a->nb_rms, nullptr, 0, stream);
}

extern "C" __attribute__((visibility("hidden"))) int llmseg_attn_delta128(const void* O, int64_t ldo, const void* dO, int64_t lddo, float* delta, int64_t batch, int32_t heads,
                                                                          int64_t T, void* stream);

// llmseg_gemm_args.dl_o: the product is dO of an attention (dX of o_proj) and the call also returns delta = rowsum_d(dO * O) per head: inside the reduce launch of
// a K-sliced product, by the attention backward's own delta kernel behind the product otherwise.  Same bits.
static int gemm_dl(const llmseg_gemm_args* a, void* stream) {
  LL_CHECK(a->dl_out && a->dl_heads > 0 && a->dl_T > 0 && !a->out_f32 && a->batch <= 1 && a->batch2 <= 1 && !a->bias && !a->gamma && !a->residual && a->act == LLMSEG_ACT_NONE &&
               !a->norm_out && !a->fx && !a->nb_x && !a->accumulate && a->N == (int64_t)a->dl_heads * 128 && (a->M % a->dl_T) == 0 && (a->ldc & 7) == 0 && (a->dl_ldo & 7) == 0 &&
               ((((uintptr_t)a->dl_o) | ((uintptr_t)a->C)) & 15) == 0,
           "gemm: dl_o needs a plain bf16 product of width heads x 128 over batch x T rows, 16-byte aligned rows");
  static const bool off = getenv("LLMSEG_GEMM_NO_DL") != nullptr;      // A/B switch
  g_dl_req = DlReq{a, !off, false};
  llmseg_gemm_args g = *a;
  g.dl_o = nullptr;
  const int rc = gemm_dispatch(&g, stream, -1);
  const bool done = g_dl_req.done;
  g_dl_req.active = false;
  if (rc != LLMSEG_OK || done) return rc;
  return llmseg_attn_delta128(a->dl_o, a->dl_ldo, a->C, a->ldc, a->dl_out, a->M / a->dl_T, a->dl_heads, a->dl_T, stream);
}

extern "C" int llmseg_gemm_bf16(const llmseg_gemm_args* a, void* stream) {
  if (a && a->struct_size == sizeof(*a) && a->fx) return gemm_fx(a, stream);
  if (a && a->struct_size == sizeof(*a) && a->nb_x) return gemm_nb(a, stream);
  if (a && a->struct_size == sizeof(*a) && a->dl_o) return gemm_dl(a, stream);
  if (!(a && a->struct_size == sizeof(*a) && a->norm_out)) return gemm_dispatch(a, stream, -1);
  // second output RMSNorm(C) * norm_w: the K-sliced route folds it into its reduce launch, every other route gets llmseg_norm behind the product
  LL_CHECK(a->norm_w && !a->out_f32 && a->batch <= 1 && a->batch2 <= 1 && (a->N & 7) == 0 && (a->ldn & 7) == 0 && a->ldn >= a->N && (a->ldc & 7) == 0 &&
               ((((uintptr_t)a->norm_w) | ((uintptr_t)a->norm_out) | ((uintptr_t)a->C)) & 15) == 0 && !a->accumulate,
           "gemm: norm_out needs norm_w, bf16 output, batch 1, N, ldc and ldn multiples of 8, 16-byte aligned pointers");
  g_norm_req = NormReq{(const bf16_t*)a->norm_w, (bf16_t*)a->norm_out, (long)a->ldn, a->norm_eps, true, false};
  const int rc = gemm_dispatch(a, stream, -1);
  const bool done = g_norm_req.done;
  g_norm_req.active = false;
  if (rc != LLMSEG_OK || done) return rc;
  return llmseg_norm(a->C, a->norm_w, nullptr, a->norm_out, a->M, a->N, a->ldc, a->ldn, a->norm_eps, 1, nullptr, stream);
}

// force_variant >= 0: an internal caller fixes the kernel (8 / 9, one K-slice)
static int gemm_dispatch(const llmseg_gemm_args* a, void* stream, int force_variant) {
  LL_CHECK(a && a->struct_size == sizeof(*a), "%s: ABI mismatch: caller's struct_size %u != %zu (bind against include/llmseg_hip.h version %d)",
           "gemm", a ? a->struct_size : 0u, sizeof(*a), LLMSEG_ABI_VERSION);
  LL_CHECK(a && a->A && a->W && a->C, "gemm: null pointer");
  LL_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "gemm: bad shape M=%ld N=%ld K=%ld", (long)a->M, (long)a->N, (long)a->K);
  const bool ta = a->trans_a != 0, tw = a->trans_w != 0;
  LL_CHECK((a->lda & 7) == 0 && (a->ldw & 7) == 0, "gemm: lda/ldw must be multiples of 8 (lda=%ld ldw=%ld)", (long)a->lda, (long)a->ldw);
  // a K-contiguous operand is read in 16-byte chunks: K % 8 == 0, or its rows are padded (ld >= roundup8(K)) with ZEROS
  const long k8 = (a->K + 7) & ~7L;
  LL_CHECK((ta || (a->K & 7) == 0 || a->lda >= k8) && (tw || (a->K & 7) == 0 || a->ldw >= k8),
           "gemm: K=%ld is not a multiple of 8 and the K-contiguous operand is not padded", (long)a->K);
  LL_CHECK((!ta || a->lda >= a->M) && (!tw || a->ldw >= a->N), "gemm: transposed operand needs ld >= rows");
  LL_CHECK((((uintptr_t)a->A) & 15) == 0 && (((uintptr_t)a->W) & 15) == 0, "gemm: A/W must be 16-byte aligned");
  LL_CHECK(((a->strideA | a->strideW | a->strideA2 | a->strideW2) & 7) == 0, "gemm: batch strides of A/W must be multiples of 8");
  LL_CHECK(!a->accumulate || a->out_f32, "gemm: accumulate needs fp32 output");
  const int esz = a->out_f32 ? 4 : 2;
  GemmP p;
  p.A = (const bf16_t*)a->A; p.W = (const bf16_t*)a->W; p.C = a->C;
  p.bias = (const bf16_t*)a->bias; p.gamma = (const bf16_t*)a->gamma; p.res = (const bf16_t*)a->residual;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
  p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc; p.ldr = a->residual ? a->ldr : 0;
  const long batch1 = a->batch > 0 ? a->batch : 1, batch2 = a->batch2 > 0 ? a->batch2 : 1;
  const long batch = batch1 * batch2;
  p.batch1 = (int)batch1;
  p.sA = a->strideA; p.sW = a->strideW; p.sC = a->strideC;
  p.sA2 = a->strideA2; p.sW2 = a->strideW2; p.sC2 = a->strideC2;
  p.alpha = a->alpha; p.act = a->act;
  p.kt_total = 0; p.k_split_total = 0; p.accum = a->accumulate ? 1 : 0;
  p.fx = 0; p.fx_T = a->fx_T; p.fx_cols = (int)a->fx_cols; p.fx_I = a->fx == LLMSEG_FX_SWIGLU ? (int)(a->N / 2) : (int)a->N;
  p.fx_cos = a->fx_cos; p.fx_sin = a->fx_sin; p.fx_out = (bf16_t*)a->fx_out; p.fx_in = (const bf16_t*)a->fx_in; p.fx_ld = a->fx_ld;
  { static const int abl = getenv("LLMSEG_LW_ABLATE") ? atoi(getenv("LLMSEG_LW_ABLATE")) : 0; p.ablate = abl; }
  // vector stores/loads need 4-element alignment of every row start; otherwise the kernel goes element-wise
  p.c_vec = ((((uintptr_t)a->C) % (4 * esz)) == 0 && (a->ldc & 3) == 0 && ((a->strideC | a->strideC2) & 3) == 0) ? 1 : 0;
  p.r_vec = (p.res && (((uintptr_t)p.res) & 7) == 0 && (p.ldr & 3) == 0 && ((a->strideC | a->strideC2) & 3) == 0) ? 1 : 0;
  p.b_vec = ((p.bias == nullptr || (((uintptr_t)p.bias) & 7) == 0) && (p.gamma == nullptr || (((uintptr_t)p.gamma) & 7) == 0)) ? 1 : 0;

  p.skew = g_gemm_skew;
  p.A2 = (const bf16_t*)a->A2; p.W2 = (const bf16_t*)a->W2; p.lda2 = a->lda2; p.ldw2 = a->ldw2;
  static const int group_m_env = getenv("LLMSEG_GEMM_GROUP_M") ? atoi(getenv("LLMSEG_GEMM_GROUP_M")) : 0;   // tuning override
  const int nt = p.K / BK;
  const long ncu = num_cus();
  int variant = (p.K % BK == 0 && !ta && !tw) ? (force_variant >= 0 ? force_variant : g_gemm_variant) : 0;
  if (variant != 0 && variant != 2 && variant != 3 && variant != 8 && variant != 9) variant = 5;
  if ((variant == 8 || variant == 9) && nt < (a->A2 ? 1 : 2)) variant = 2;
  // split-K needs a dense-enough problem for the slab layout [S][M][N], 4-column alignment and room in the caller's workspace
  const bool can_split = batch == 1 && (p.N & 3) == 0 && (p.ldc & 3) == 0 && a->workspace != nullptr &&
                         (((uintptr_t)a->workspace) & 15) == 0 && (!p.res || (p.ldr & 3) == 0);
  auto ws_fits = [&](int S) { return (double)(S + (a->A2 ? 1 : 0)) * p.M * p.N * 4.0 <= (double)a->workspace_bytes; };   // + the extension product's slab
  int split = 1;
  if (variant == 5) {
    // auto: minimum of the cost model over {128 x 128 DMA kernel, ping-pong 256 x 256 / 128 x 256 with 1..16 K-slices}
    GemmPlan best{2, 1, glds_cost(p.M, p.N, nt, ncu) * (double)batch};
    if (nt >= (a->A2 ? 1 : 2)) {
      for (int mi = 4; mi >= 2; mi -= 2) {
        for (int S = 1; S <= 16; ++S) {
          if (S > 1 && (!can_split || !split_ok(nt, S) || !ws_fits(S))) continue;
          if (S > 1 && ((p.M + 64 * mi - 1) / (64 * mi)) * ((p.N + 255) / 256) * S > ncu) break;     // slices only to fill ONE round of the CUs
          const double us = pp_cost(p.M, p.N, nt, mi, S, ncu, a->out_f32 != 0) * (double)batch;
          if (us < best.us * 0.97 || (us < best.us && S == 1)) best = GemmPlan{mi == 4 ? 8 : 9, S, us};
        }
      }
    }
    // round 6: the four-stage 128 x 128 kernel for products that fit ONE round of the CUs (at most one workgroup per CU: nothing else hides the
    // fetch latency there): ~0.55 us per K-tile behind a three-tile-deep DMA queue + ~4 us of first-tile latency and epilogue (tools/gemm_bench.py, GEMM_SET=clip)
    static const int g4_env = getenv("LLMSEG_GEMM_G4") ? atoi(getenv("LLMSEG_GEMM_G4")) : 0;      // A/B switch: 1 = let the cost model pick it
    if (g4_env && !a->A2 && nt >= 2) {
      const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
      if (t128 <= ncu) {
        const double us = nt * 0.55 + 4.0;
        if (us < best.us) best = GemmPlan{3, 1, us};
      }
    }
    variant = best.variant; split = best.split;
  } else if (variant == 8 || variant == 9) {
    split = (g_gemm_split > 1 && force_variant < 0) ? g_gemm_split : 1;
    if (split > 1) LL_CHECK(can_split && split_ok(nt, split) && ws_fits(split), "gemm: forced split-K %d not possible for this call", split);
  }
  if (p.A2) {
    // C = epi(alpha * (A.W^T + A2.W2^T)), A2 [M][64], W2 [N][64]: fused as one more K-tile of the ping-pong kernel; any other
    // kernel runs the product as a second launch that accumulates onto C (linear epilogues only)
    LL_CHECK(p.W2 && batch == 1 && !ta && !tw && (p.lda2 & 7) == 0 && (p.ldw2 & 7) == 0 && p.lda2 >= 64 && p.ldw2 >= 64 &&
                 (((uintptr_t)p.A2 | (uintptr_t)p.W2) & 15) == 0, "gemm: bad extension operands (A2 [M][64], W2 [N][64], 16-byte aligned rows)");
    LL_CHECK(!a->out_f32, "gemm: extension operands need bf16 output");
    if ((variant == 8 || variant == 9) && split > 1) {
      // split-K: the extension product is one more fp32 slab (a K = 64 launch of its own), summed by the reduce kernel
      llmseg_gemm_args g2 = *a;
      g2.A = a->A2; g2.W = a->W2; g2.lda = a->lda2; g2.ldw = a->ldw2; g2.K = 64; g2.A2 = g2.W2 = nullptr;
      g2.bias = g2.gamma = g2.residual = nullptr; g2.alpha = 1.f; g2.act = LLMSEG_ACT_NONE; g2.out_f32 = 1; g2.accumulate = 0;
      g2.C = (float*)a->workspace + (long)split * p.M * p.N; g2.ldc = p.N; g2.workspace = nullptr; g2.workspace_bytes = 0;
      g2.norm_w = nullptr; g2.norm_out = nullptr;              // (the caller's second output belongs to the whole product, not to this slab)
      const bool pend = g_norm_req.active;
      g_norm_req.active = false;
      const int rc = llmseg_gemm_bf16(&g2, stream);
      g_norm_req.active = pend;
      if (rc != LLMSEG_OK) return rc;
    } else if (variant != 8 && variant != 9) {
      LL_CHECK(a->act == LLMSEG_ACT_NONE && !a->gamma, "gemm: extension operands on this shape need a linear epilogue");
      llmseg_gemm_args g1 = *a, g2 = *a;
      g1.A2 = g1.W2 = nullptr;
      g1.norm_w = g2.norm_w = nullptr; g1.norm_out = g2.norm_out = nullptr;      // second output: after BOTH launches (the entry point's llmseg_norm)
      g_norm_req.active = false;
      int rc = llmseg_gemm_bf16(&g1, stream);
      if (rc != LLMSEG_OK) return rc;
      g2.A = a->A2; g2.W = a->W2; g2.lda = a->lda2; g2.ldw = a->ldw2; g2.K = 64; g2.bias = nullptr; g2.residual = a->C; g2.ldr = a->ldc;
      g2.A2 = g2.W2 = nullptr;
      return llmseg_gemm_bf16(&g2, stream);
    }
  }
  p.a_norm_w = (const bf16_t*)a->a_norm_w; p.a_norm_eps = a->a_norm_eps; p.a_swiglu = a->a_swiglu;
  const bool a_xform = p.a_norm_w || p.a_swiglu;
  LL_CHECK(!a_xform || (p.M <= 8 && !ta && !tw && batch == 1 && !p.A2 && (p.K & 7) == 0 && !(p.a_norm_w && p.a_swiglu) &&
                        (!p.a_norm_w || (((uintptr_t)p.a_norm_w) & 15) == 0) && (!p.a_swiglu || p.lda >= 2 * p.K)),
           "gemm: A-row transforms (a_norm_w / a_swiglu) are decode-step fusions of the M <= 8 route");
  if (p.M <= 8 && !ta && !tw && batch == 1 && !p.A2 && (p.K & 7) == 0 && (g_gemm_variant == 5 || a_xform)) {
    // skinny GEMM (decode steps, single-row head GEMMs): a weight stream, HBM-bound
    hipStream_t s = (hipStream_t)stream;
    llmseg_prof_begin(s);
    llmseg_prof_tag(p.M, p.N, p.K, 3000 + (p.res ? 20 : 0) + p.act * 2 + (a->out_f32 ? 1 : 0));
    static const int sk_env = getenv("LLMSEG_SKINNY_SK") ? atoi(getenv("LLMSEG_SKINNY_SK")) : 1;
    const bool sk = sk_env == 2 ? p.K >= 2048 : (sk_env && p.N <= 8192 && p.K >= 2048);      // fewer than 2 waves per SIMD otherwise: the workgroup's waves split K instead
    const dim3 grid((unsigned)(sk ? (p.N + 3) / 4 : (p.N + 15) / 16));
    const int of = a->out_f32 ? 1 : 0;
#define LL_SKINNY_(AT, SKV)                                                                                   \
    do {                                                                                                      \
      if (p.M == 1) LL_LAUNCH_KERNEL((gemm_skinny_kernel<1, AT, SKV>), grid, dim3(256), 0, s, p, of);      \
      else if (p.M == 2) LL_LAUNCH_KERNEL((gemm_skinny_kernel<2, AT, SKV>), grid, dim3(256), 0, s, p, of); \
      else if (p.M <= 4) LL_LAUNCH_KERNEL((gemm_skinny_kernel<4, AT, SKV>), grid, dim3(256), 0, s, p, of); \
      else LL_LAUNCH_KERNEL((gemm_skinny_kernel<8, AT, SKV>), grid, dim3(256), 0, s, p, of);               \
    } while (0)
#define LL_SKINNY(AT) do { if (sk) LL_SKINNY_(AT, true); else LL_SKINNY_(AT, false); } while (0)
    if (p.a_norm_w) LL_SKINNY(1); else if (p.a_swiglu) LL_SKINNY(2); else LL_SKINNY(0);
#undef LL_SKINNY_
#undef LL_SKINNY
    llmseg_prof_end(s, 2.0 * (double)a->M * (double)a->N * (double)a->K);
    LL_LAUNCH_CHECK("gemm_skinny");
    return LLMSEG_OK;
  }
  const bool pp = variant == 8 || variant == 9;
  const int bm = variant == 8 ? 256 : 128, bn = pp ? 256 : BN;
  p.tiles_m = (p.M + bm - 1) / bm; p.tiles_n = (p.N + bn - 1) / bn;
  // ping-pong tile walk (tools/gemm_bench.py sweeps): short matrices (Llama, <= 32 row tiles) with few column tiles (N = 4096: o, down,
  // the dX products) keep all of M in one group so a W column tile is fetched once per XCD; with many column tiles (qkv, gate|up, lm_head at
  // 16-24 images: 20-30 row tiles x 48-126 column tiles) an XCD's 32 concurrent tiles would be ONE column tile deep and re-stream all of A
  // (63 MB at 24 images) per column tile -- 8 row tiles per group make the concurrent set 8 x 4 (+5..8 %: qkv 1017 -> 1100, gate|up
  // 1215 -> 1300, lm_head 1234 -> 1310 TF/s at 16 images); tall ones (SAM, 384+ row tiles) walk 4 row tiles per group (+3..6 % at K = 5120)
  static const bool old_walk = getenv("LLMSEG_GEMM_OLD_WALK") != nullptr;     // A/B switch
  p.group_m = group_m_env > 0 ? group_m_env : (p.tiles_m <= 32 ? ((p.tiles_n > 16 && p.tiles_m > 8 && !old_walk) ? 8 : p.tiles_m) : 4);
  hipStream_t s = (hipStream_t)stream;
  llmseg_prof_begin(s);
  llmseg_prof_tag(p.M, p.N, p.K, (variant == 9 && g_gemm_pp2 ? 7 : variant) * 1000 + (ta ? 200 : 0) + (tw ? 100 : 0) + (p.res ? 20 : 0) + p.act * 2 + (a->out_f32 ? 1 : 0) + 40 * (batch > 1) +
                  10000 * (split > 1 ? split : 0));
  const bool f = a->out_f32 != 0;
  // Register-staging kernel (transposed operands / K % 64 != 0) on a grid that leaves most CUs idle with a long serial K loop (a lone
  // workgroup takes 1.2-3 us per K-tile, all of it exposed latency: 512 x 256 x 2048 with W stored [K][N] was 75 us on 8 workgroups):
  // K-slices as the (inner) batch index, fp32 slabs in the caller's workspace, epilogue in the reduce launch.
  if (variant == 0 && batch2 == 1 && (p.N & 3) == 0 && (p.ldc & 3) == 0 && a->workspace != nullptr && (((uintptr_t)a->workspace) & 15) == 0 &&
      (!p.res || ((p.ldr & 3) == 0 && batch == 1)) && ((p.sC & 3) == 0 || batch == 1) && g_gemm_rsplit) {
    const long tiles = (long)p.tiles_m * p.tiles_n * batch;
    const int ntr = (p.K + BK - 1) / BK;
    int S = (tiles * 4 <= ncu && ntr >= 8) ? (int)std::min<long>({(long)ncu / std::max<long>(tiles, 1), (long)ntr / 2, 32L}) : 1;
    while (S > 1 && (double)S * batch * p.M * p.N * 4.0 > (double)a->workspace_bytes) --S;
    if (S > 1) {
      const int q = (ntr + S - 1) / S;
      S = (ntr + q - 1) / q;
    }
    if (S > 1) {
      const int q = (ntr + S - 1) / S;
      static const bool log_it = getenv("LLMSEG_RSPLIT_LOG") != nullptr;
      if (log_it) fprintf(stderr, "rsplit M=%d N=%d K=%d ta=%d tw=%d S=%d q=%d f32=%d acc=%d res=%d bias=%d act=%d lda=%ld ldw=%ld ldc=%ld batch=%ld alpha=%g\n", p.M, p.N, p.K,
                          (int)ta, (int)tw, S, q, (int)f, p.accum, p.res != nullptr, p.bias != nullptr, p.act, p.lda, p.ldw, p.ldc, batch, p.alpha);
      GemmP ps = p;
      ps.K = q * BK; ps.k_split_total = p.K; ps.batch1 = S;
      ps.sA = ta ? (long)q * BK * p.lda : (long)q * BK; ps.sW = tw ? (long)q * BK * p.ldw : (long)q * BK;
      ps.sA2 = p.sA; ps.sW2 = p.sW;                                   // the call's own batch index moves to the outer slot
      ps.C = a->workspace; ps.ldc = p.N; ps.sC = (long)p.M * p.N; ps.sC2 = (long)S * p.M * p.N;
      ps.bias = ps.gamma = ps.res = nullptr; ps.ldr = 0; ps.alpha = 1.f; ps.act = LLMSEG_ACT_NONE; ps.accum = 0;
      ps.c_vec = 1; ps.r_vec = 0; ps.b_vec = 1;
      const dim3 grid(p.tiles_m * p.tiles_n, (unsigned)(S * batch));
      const int key = (ta ? 2 : 0) | (tw ? 1 : 0);
      switch (key) {
        case 0: LL_LAUNCH_KERNEL((gemm_bf16_tn_kernel<true, false, false>), grid, dim3(NT), 0, s, ps); break;
        case 1: LL_LAUNCH_KERNEL((gemm_bf16_tn_kernel<true, false, true>), grid, dim3(NT), 0, s, ps); break;
        case 2: LL_LAUNCH_KERNEL((gemm_bf16_tn_kernel<true, true, false>), grid, dim3(NT), 0, s, ps); break;
        default: LL_LAUNCH_KERNEL((gemm_bf16_tn_kernel<true, true, true>), grid, dim3(NT), 0, s, ps); break;
      }
      const long total4 = (long)p.M * (p.N >> 2);
      const unsigned rg = (unsigned)std::min<long>((total4 + 255) / 256, 4096);
      LL_LAUNCH_KERNEL(splitk_reduce_kernel, dim3(rg, (unsigned)batch), dim3(256), 0, s, p, (const float*)a->workspace, S, f ? 1 : 0);
      llmseg_prof_end(s, 2.0 * (double)a->M * (double)a->N * (double)a->K * (double)batch);
      LL_LAUNCH_CHECK("gemm_bf16_tn (K-sliced)");
      return LLMSEG_OK;
    }
  }
  if (pp && split > 1) {
    // K-slices as the batch dimension of the ping-pong kernel: slice s reads A / W columns [s q 64, ...), writes fp32 slab s
    GemmP ps = p;
    const int q = (nt + split - 1) / split;
    ps.K = q * BK; ps.kt_total = nt; ps.batch1 = split;
    ps.sA = ps.sW = (long)q * BK; ps.sA2 = ps.sW2 = ps.sC2 = 0;
    ps.C = a->workspace; ps.ldc = p.N; ps.sC = (long)p.M * p.N;
    ps.bias = ps.gamma = ps.res = nullptr; ps.ldr = 0; ps.alpha = 1.f; ps.act = LLMSEG_ACT_NONE; ps.accum = 0;
    ps.c_vec = 1; ps.r_vec = 0; ps.b_vec = 1; ps.A2 = ps.W2 = nullptr;
    dim3 grid(p.tiles_m * p.tiles_n, (unsigned)split);
    if (variant == 8) LL_LAUNCH_KERNEL((gemm_bf16_tn_pp_kernel<true, false, 4>), grid, dim3(NTB), 0, s, ps);
    else if (g_gemm_pp2 == 2) LL_LAUNCH_KERNEL((gemm_bf16_tn_lw_kernel<true>), grid, dim3(NTL), 0, s, ps);      // loader-wave form (experiment)
    else if (g_gemm_pp2) LL_LAUNCH_KERNEL((gemm_bf16_tn_pp2_kernel<true, false>), grid, dim3(NTB), 0, s, ps);
    else LL_LAUNCH_KERNEL((gemm_bf16_tn_pp_kernel<true, false, 2>), grid, dim3(NTB), 0, s, ps);
    const long total4 = (long)p.M * (p.N >> 2);
    const unsigned rg = (unsigned)std::min<long>((total4 + 255) / 256, 4096);
    // the row kernel only where llmseg_norm would run its workgroup-per-row kernel on this shape (same arithmetic, same bits) and the epilogue is the plain residual add
    static const long wg_max_rows = getenv("LLMSEG_NORM_WG_MAX") ? atol(getenv("LLMSEG_NORM_WG_MAX")) : 2048;
    static const bool no_fuse = getenv("LLMSEG_GEMM_NO_NORM_FUSE") != nullptr;      // A/B switch: always the two-launch route
    const bool fuse_norm = g_norm_req.active && !no_fuse && !f && p.alpha == 1.f && !p.bias && !p.gamma && p.act == LLMSEG_ACT_NONE && p.M >= 64 && p.M < wg_max_rows &&
                           p.N >= 2048 && p.N <= 8192 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (!p.res || (p.ldr & 7) == 0) && (g_norm_req.ldn & 7) == 0 &&
                           ((((uintptr_t)p.C) | ((uintptr_t)p.res) | ((uintptr_t)g_norm_req.w) | ((uintptr_t)g_norm_req.out)) & 15) == 0;
    const bool fuse_nb = g_nb_req.active && !f && p.alpha == 1.f && !p.bias && !p.gamma && !p.res && p.act == LLMSEG_ACT_NONE && p.M >= 64 && p.N >= 2048 && p.N <= 8192 &&
                         (p.N & 7) == 0;       // == where llmseg_norm_bwd_add runs its workgroup-per-row kernel (same arithmetic, same bits)
    const bool fuse_dl = g_dl_req.active && !f && !p.bias && !p.gamma && !p.res && p.act == LLMSEG_ACT_NONE && p.N <= 8192 && (p.N & 127) == 0 && (p.ldc & 7) == 0;
    if (fuse_dl) {
      const llmseg_gemm_args* q = g_dl_req.a;
      const int S2 = split + (p.A2 ? 1 : 0);
      const int cpt = ((p.N >> 3) + 255) / 256;
#define LL_RDL(C) LL_LAUNCH_KERNEL(splitk_reduce_delta_kernel<C>, dim3((unsigned)p.M), dim3(256), 0, s, p, (const float*)a->workspace, S2, (const bf16_t*)q->dl_o, \
                                   (long)q->dl_ldo, q->dl_out, (int)q->dl_heads, (int)q->dl_T)
      if (cpt <= 1) LL_RDL(1); else if (cpt <= 2) LL_RDL(2); else LL_RDL(4);
#undef LL_RDL
      g_dl_req.done = true;
    } else if (fuse_nb) {
      const llmseg_gemm_args* q = g_nb_req.a;
      const int rc = llmseg_reduce_lora_normbwd((const float*)a->workspace, split + (p.A2 ? 1 : 0), p.M, p.N, q->nb_x, q->nb_w, g_nb_req.out, q->nb_eps, q->nb_rms, q->nb_dres,
                                                (void*)q->nb_lora_t, q->nb_lora_ldt, q->nb_lora_w0, q->nb_lora_w1, q->nb_lora_alpha, (const llmseg_dropout*)q->nb_lora_drop,
                                                q->nb_lora_part, q->nb_lora_S, q->nb_lora_scale, q->nb_lora_zero, stream);
      if (rc != LLMSEG_OK) return rc;
      g_nb_req.done = true;
    } else if (fuse_norm) {
      const int S2 = split + (p.A2 ? 1 : 0);
      const int cpt = ((p.N >> 3) + 255) / 256;
      if (cpt <= 1) LL_LAUNCH_KERNEL(splitk_reduce_rmsnorm_kernel<1>, dim3((unsigned)p.M), dim3(256), 0, s, p, (const float*)a->workspace, S2, g_norm_req.w, g_norm_req.eps, g_norm_req.out, g_norm_req.ldn);
      else if (cpt <= 2) LL_LAUNCH_KERNEL(splitk_reduce_rmsnorm_kernel<2>, dim3((unsigned)p.M), dim3(256), 0, s, p, (const float*)a->workspace, S2, g_norm_req.w, g_norm_req.eps, g_norm_req.out, g_norm_req.ldn);
      else LL_LAUNCH_KERNEL(splitk_reduce_rmsnorm_kernel<4>, dim3((unsigned)p.M), dim3(256), 0, s, p, (const float*)a->workspace, S2, g_norm_req.w, g_norm_req.eps, g_norm_req.out, g_norm_req.ldn);
      g_norm_req.done = true;
    } else
    LL_LAUNCH_KERNEL(splitk_reduce_kernel, dim3(rg), dim3(256), 0, s, p, (const float*)a->workspace, split + (p.A2 ? 1 : 0), f ? 1 : 0);
  } else {
    dim3 grid(p.tiles_m * p.tiles_n, (unsigned)batch);
    switch (variant) {
      case 2: f ? launch_glds<true, 2, 1>(p, grid, s) : launch_glds<false, 2, 1>(p, grid, s); break;
      case 3:
        if (f) LL_LAUNCH_KERNEL((gemm_bf16_tn_g4_kernel<true>), grid, dim3(NT), 0, s, p);
        else LL_LAUNCH_KERNEL((gemm_bf16_tn_g4_kernel<false>), grid, dim3(NT), 0, s, p);
        break;
      case 8: {
        // fused Llama-layer epilogues at large M (the fused accumulation window, 24-image micro-batches): the 256 x 256 tile's FX forms
        const bool fx_ok8 = a->fx && !g_fx_off && !f && batch == 1 &&
                            (a->fx == LLMSEG_FX_ROPE ? (p.N % 256) == 0 && p.A2 != nullptr
                             : a->fx == LLMSEG_FX_SWIGLU ? (p.fx_I % 128) == 0 : (p.N % 64) == 0);
        if (fx_ok8) {
          GemmP q = p;
          q.fx = a->fx;
          if (a->fx == LLMSEG_FX_SWIGLU_BWD) q.C = (bf16_t*)a->C - a->N;
          if (a->fx == LLMSEG_FX_ROPE) LL_LAUNCH_KERNEL((gemm_bf16_tn_pp_kernel<false, true, 4, FX_ROPE>), grid, dim3(NTB), 0, s, q);
          else if (a->fx == LLMSEG_FX_SWIGLU) LL_LAUNCH_KERNEL((gemm_bf16_tn_pp_kernel<false, false, 4, FX_SWIGLU>), grid, dim3(NTB), 0, s, q);
          else LL_LAUNCH_KERNEL((gemm_bf16_tn_pp_kernel<false, false, 4, FX_SWIGLU_BWD>), grid, dim3(NTB), 0, s, q);
          g_fx_done = true;
          break;
        }
        if (p.A2) LL_LAUNCH_KERNEL((gemm_bf16_tn_pp_kernel<false, true, 4>), grid, dim3(NTB), 0, s, p);      // bf16 out only (checked above)
        else if (f) LL_LAUNCH_KERNEL((gemm_bf16_tn_pp_kernel<true, false, 4>), grid, dim3(NTB), 0, s, p);
        else LL_LAUNCH_KERNEL((gemm_bf16_tn_pp_kernel<false, false, 4>), grid, dim3(NTB), 0, s, p);
        break;
      }
      case 9:
        if (g_gemm_pp2 == 2 && !p.A2) {                     // loader-wave form (experiment; the LoRA extension tile keeps the two-phase kernel)
          if (f) LL_LAUNCH_KERNEL((gemm_bf16_tn_lw_kernel<true>), grid, dim3(NTL), 0, s, p);
          else LL_LAUNCH_KERNEL((gemm_bf16_tn_lw_kernel<false>), grid, dim3(NTL), 0, s, p);
          break;
        }
        if (g_gemm_pp2) {
          // fused Llama-layer epilogues (llmseg_gemm_args.fx): this kernel, one K-slice; shapes whose tiles hold whole pairs (gemm_fx runs the pointwise launch otherwise)
          const bool fx_ok = a->fx && !g_fx_off && !f && batch == 1 &&
                             (a->fx == LLMSEG_FX_ROPE ? (p.N % 256) == 0 && p.A2 != nullptr
                              : a->fx == LLMSEG_FX_SWIGLU ? (p.fx_I % 128) == 0 : (p.N % 64) == 0);
          if (fx_ok) {
            GemmP q = p;
            q.fx = a->fx;
            if (a->fx == LLMSEG_FX_SWIGLU_BWD) q.C = (bf16_t*)a->C - a->N;      // gemm_fx pointed C at the up half for the two-launch route: back to the row start
            if (a->fx == LLMSEG_FX_ROPE) LL_LAUNCH_KERNEL((gemm_bf16_tn_pp2_kernel<false, true, FX_ROPE>), grid, dim3(NTB), 0, s, q);
            else if (a->fx == LLMSEG_FX_SWIGLU) LL_LAUNCH_KERNEL((gemm_bf16_tn_pp2_kernel<false, false, FX_SWIGLU>), grid, dim3(NTB), 0, s, q);
            else LL_LAUNCH_KERNEL((gemm_bf16_tn_pp2_kernel<false, false, FX_SWIGLU_BWD>), grid, dim3(NTB), 0, s, q);
            g_fx_done = true;
            break;
          }
          if (p.A2) LL_LAUNCH_KERNEL((gemm_bf16_tn_pp2_kernel<false, true>), grid, dim3(NTB), 0, s, p);
          else if (f) LL_LAUNCH_KERNEL((gemm_bf16_tn_pp2_kernel<true, false>), grid, dim3(NTB), 0, s, p);
          else LL_LAUNCH_KERNEL((gemm_bf16_tn_pp2_kernel<false, false>), grid, dim3(NTB), 0, s, p);
          break;
        }
        if (p.A2) LL_LAUNCH_KERNEL((gemm_bf16_tn_pp_kernel<false, true, 2>), grid, dim3(NTB), 0, s, p);
        else if (f) LL_LAUNCH_KERNEL((gemm_bf16_tn_pp_kernel<true, false, 2>), grid, dim3(NTB), 0, s, p);
        else LL_LAUNCH_KERNEL((gemm_bf16_tn_pp_kernel<false, false, 2>), grid, dim3(NTB), 0, s, p);
        break;
      default: {
        const int key = (f ? 4 : 0) | (ta ? 2 : 0) | (tw ? 1 : 0);
        switch (key) {
          case 0: LL_LAUNCH_KERNEL((gemm_bf16_tn_kernel<false, false, false>), grid, dim3(NT), 0, s, p); break;
          case 1: LL_LAUNCH_KERNEL((gemm_bf16_tn_kernel<false, false, true>), grid, dim3(NT), 0, s, p); break;
          case 2: LL_LAUNCH_KERNEL((gemm_bf16_tn_kernel<false, true, false>), grid, dim3(NT), 0, s, p); break;
          case 3: LL_LAUNCH_KERNEL((gemm_bf16_tn_kernel<false, true, true>), grid, dim3(NT), 0, s, p); break;
          case 4: LL_LAUNCH_KERNEL((gemm_bf16_tn_kernel<true, false, false>), grid, dim3(NT), 0, s, p); break;
          case 5: LL_LAUNCH_KERNEL((gemm_bf16_tn_kernel<true, false, true>), grid, dim3(NT), 0, s, p); break;
          case 6: LL_LAUNCH_KERNEL((gemm_bf16_tn_kernel<true, true, false>), grid, dim3(NT), 0, s, p); break;
          default: LL_LAUNCH_KERNEL((gemm_bf16_tn_kernel<true, true, true>), grid, dim3(NT), 0, s, p); break;
        }
      }
    }
  }
  llmseg_prof_end(s, 2.0 * (double)a->M * (double)a->N * (double)a->K * (double)batch);
  LL_LAUNCH_CHECK("gemm_bf16_tn");
  return LLMSEG_OK;
}
