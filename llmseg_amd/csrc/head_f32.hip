// fp32-activation kernels for the INFERENCE side of the mask-selection head: mask pooling, `text_hidden_fcs`, the two
// LISA_TwoWayAttentionBlocks, the final attention, the IoU / embedding heads and the cosine scores evaluated with fp32 activations on the
// bf16 trunk outputs (reference model/LISA.py:340-408, model/transformer.py:215-341).  The trunk (Llama, SAM, CLIP) keeps the reference's
// dtype; the head is ~1 GFLOP per image, so nothing here needs the matrix cores: plain fp32 FMA chains in ascending k -- the arithmetic the
// fp32 CPU oracle performs, up to the order of a dot product's partial sums.  Weights stay the model's bf16 tensors (what the reference holds).
// Measured on the builder's A/B (profiles/r04b_spread_fulldepth.md): the bf16 head contributes ~40 % of the error of pred_similarity / pred_iou.
#include "common.h"
#include "llmseg_hip.h"

namespace {

__device__ __forceinline__ float act_f32(float v, int act) {
  switch (act) {
    case LLMSEG_ACT_RELU: return fmaxf(v, 0.f);
    case LLMSEG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case LLMSEG_ACT_SILU: return v / (1.f + expf(-v));
    case LLMSEG_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    case LLMSEG_ACT_QUICKGELU: return v / (1.f + expf(-1.702f * v));
    default: return v;
  }
}

// y[m][n] = act(alpha * sum_k x[m][k] * W(n, k) + bias[n]) + res[m][n];  x, res, y fp32; W, bias bf16.
// W_KN = false: W stored [N][K] (an nn.Linear weight); true: W stored [K][N] (a channels-last feature map as the right operand of the mask pooling).
// 64 x 64 output tile per 256-thread workgroup, 4 x 4 outputs per thread, K in steps of 16 through LDS; every output is one FMA chain in ascending k.
constexpr int LT = 64, LK = 16;
template <bool W_KN>
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x, long ldx, const bf16_t* __restrict__ W, long ldw, const bf16_t* __restrict__ bias,
                                                         const float* __restrict__ res, long ldr, float* __restrict__ y, long ldy, int M, int N, int K, int act,
                                                         float alpha) {
  __shared__ float xs[LK][LT + 4];
  __shared__ float ws[LK][LT + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * LT, n0 = blockIdx.x * LT;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += LK) {
    {   // x tile: 64 rows x 16 k, thread -> (row tid / 4, k quad tid % 4)
      const int r = tid >> 2, kq = (tid & 3) * 4;
      const int m = m0 + r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + kq + e;
        xs[kq + e][r] = (m < M && k < K) ? x[(long)m * ldx + k] : 0.f;
      }
    }
    if (!W_KN) {   // W [N][K]: thread -> (n tid / 4, k quad tid % 4)
      const int r = tid >> 2, kq = (tid & 3) * 4;
      const int n = n0 + r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + kq + e;
        ws[kq + e][r] = (n < N && k < K) ? bf2f(W[(long)n * ldw + k]) : 0.f;
      }
    } else {       // W [K][N]: thread -> (k tid / 16, n quad tid % 16)
      const int kk = tid >> 4, nq = (tid & 15) * 4;
      const int k = k0 + kk;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + nq + e;
        ws[kk][nq + e] = (n < N && k < K) ? bf2f(W[(long)k * ldw + n]) : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < LK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = xs[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = ws[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= N) continue;
      float v = acc[i][j] * alpha + (bias ? bf2f(bias[n]) : 0.f);
      v = act_f32(v, act);
      if (res) v += res[(long)m * ldr + n];
      y[(long)m * ldy + n] = v;
    }
  }
}

// LayerNorm over the last dimension, fp32 in / out, bf16 weight + bias; one wave per row; two passes (mean, then centred variance).
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, const bf16_t* __restrict__ w, const bf16_t* __restrict__ b, float* __restrict__ y,
                                                            long rows, int D, float eps) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xr[i];
  const float mean = wave_sum(s) / (float)D;
  float v = 0.f;
  for (int i = lane; i < D; i += 64) { const float d = xr[i] - mean; v = fmaf(d, d, v); }
  const float rstd = 1.f / sqrtf(wave_sum(v) / (float)D + eps);
  float* yr = y + row * D;
  for (int i = lane; i < D; i += 64) yr[i] = (xr[i] - mean) * rstd * bf2f(w[i]) + (b ? bf2f(b[i]) : 0.f);
}

// softmax(q k^T * scale) v for small problems, everything fp32.  One thread per query (128 queries per workgroup), keys / values of the (batch,
// head) pair staged through LDS 64 rows at a time; online softmax with the running maximum (the reference subtracts the row maximum too).
template <int HD>
__global__ __launch_bounds__(128) void attn_f32_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, float* __restrict__ o,
                                                       long qsb, long qsh, long qsr, long ksb, long ksh, long ksr, long vsb, long vsh, long vsr, long osb, long osh, long osr,
                                                       int Nq, int Nk, float scale) {
  __shared__ float ks[64][HD + 1];
  __shared__ float vs[64][HD + 1];
  const int b = blockIdx.z, h = blockIdx.y;
  const int qi = blockIdx.x * 128 + threadIdx.x;
  const bool live = qi < Nq;
  float qr[HD], acc[HD];
  const float* qp = q + b * qsb + h * qsh + (long)(live ? qi : 0) * qsr;
#pragma unroll
  for (int d = 0; d < HD; ++d) { qr[d] = qp[d] * scale; acc[d] = 0.f; }
  float mx = -INFINITY, l = 0.f;
  for (int j0 = 0; j0 < Nk; j0 += 64) {
    const int nj = min(64, Nk - j0);
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * HD; e += 128) {
      const int r = e / HD, d = e % HD;
      const bool ok = r < nj;
      ks[r][d] = ok ? k[b * ksb + h * ksh + (long)(j0 + r) * ksr + d] : 0.f;
      vs[r][d] = ok ? v[b * vsb + h * vsh + (long)(j0 + r) * vsr + d] : 0.f;
    }
    __syncthreads();
    for (int r = 0; r < nj; ++r) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) s = fmaf(qr[d], ks[r][d], s);
      const float mn = fmaxf(mx, s);
      const float corr = expf(mx - mn), p = expf(s - mn);
      l = l * corr + p;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[d] = fmaf(acc[d], corr, p * vs[r][d]);
      mx = mn;
    }
  }
  if (live) {
    float* op = o + b * osb + h * osh + (long)qi * osr;
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < HD; ++d) op[d] = acc[d] * inv;
  }
}

// sim[k] = <t / |t|, e_k / |e_k|> (LISA.py:398-403), fp32; one wave per row of e.
__global__ __launch_bounds__(256) void cosine_f32_kernel(const float* __restrict__ t, const float* __restrict__ e, float* __restrict__ sim, int K, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= K) return;
  const float* er = e + (long)row * D;
  float tt = 0.f, ee = 0.f, te = 0.f;
  for (int i = lane; i < D; i += 64) { const float a = t[i], b = er[i]; tt = fmaf(a, a, tt); ee = fmaf(b, b, ee); te = fmaf(a, b, te); }
  tt = wave_sum(tt); ee = wave_sum(ee); te = wave_sum(te);
  if (lane == 0) sim[row] = te / (sqrtf(tt) * sqrtf(ee));
}

}  // namespace

extern "C" int llmseg_linear_f32(const float* x, int64_t ldx, const void* W, int64_t ldw, int32_t w_kn, const void* bias, const float* residual, int64_t ldr,
                                 float* y, int64_t ldy, int32_t M, int32_t N, int32_t K, int32_t act, float alpha, void* stream) {
  LL_CHECK(x && W && y && M > 0 && N > 0 && K > 0, "linear_f32: bad arguments");
  LL_CHECK(ldx >= K && ldy >= N && (w_kn ? ldw >= N : ldw >= K) && (!residual || ldr >= N), "linear_f32: a leading dimension is shorter than its row");
  LL_CHECK(act >= LLMSEG_ACT_NONE && act <= LLMSEG_ACT_SIGMOID, "linear_f32: unknown activation %d", act);
  const dim3 grid((N + LT - 1) / LT, (M + LT - 1) / LT);
  if (w_kn)
    LL_LAUNCH_KERNEL((linear_f32_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (const bf16_t*)W, (long)ldw, (const bf16_t*)bias, residual,
                     (long)ldr, y, (long)ldy, M, N, K, act, alpha);
  else
    LL_LAUNCH_KERNEL((linear_f32_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (const bf16_t*)W, (long)ldw, (const bf16_t*)bias, residual,
                     (long)ldr, y, (long)ldy, M, N, K, act, alpha);
  LL_LAUNCH_CHECK("linear_f32");
  return LLMSEG_OK;
}

extern "C" int llmseg_layernorm_f32(const float* x, const void* w, const void* b, float* y, int64_t rows, int32_t D, float eps, void* stream) {
  LL_CHECK(x && w && y && rows > 0 && D > 0, "layernorm_f32: bad arguments");
  LL_LAUNCH_KERNEL(layernorm_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)w, (const bf16_t*)b, y, (long)rows, D, eps);
  LL_LAUNCH_CHECK("layernorm_f32");
  return LLMSEG_OK;
}

extern "C" int llmseg_attn_f32(const float* q, const float* k, const float* v, float* o, const int64_t* strides, int32_t batch, int32_t heads, int32_t Nq, int32_t Nk,
                               int32_t head_dim, float scale, void* stream) {
  LL_CHECK(q && k && v && o && strides && batch > 0 && heads > 0 && Nq > 0 && Nk > 0, "attn_f32: bad arguments");
  LL_CHECK(head_dim == 32 || head_dim == 64, "attn_f32: head_dim %d (32 and 64 are built: the mask-selection head is 8 x 32)", head_dim);
  const dim3 grid((Nq + 127) / 128, heads, batch);
  const int64_t* s = strides;       // {q, k, v, o} x {batch, head, row} in elements
#define ATTN_F32_ARGS q, k, v, o, (long)s[0], (long)s[1], (long)s[2], (long)s[3], (long)s[4], (long)s[5], (long)s[6], (long)s[7], (long)s[8], (long)s[9], (long)s[10], (long)s[11], Nq, Nk, scale
  if (head_dim == 32) LL_LAUNCH_KERNEL((attn_f32_kernel<32>), grid, dim3(128), 0, (hipStream_t)stream, ATTN_F32_ARGS);
  else LL_LAUNCH_KERNEL((attn_f32_kernel<64>), grid, dim3(128), 0, (hipStream_t)stream, ATTN_F32_ARGS);
#undef ATTN_F32_ARGS
  LL_LAUNCH_CHECK("attn_f32");
  return LLMSEG_OK;
}

extern "C" int llmseg_cosine_f32(const float* t, const float* e, float* sim, int32_t K, int32_t D, void* stream) {
  LL_CHECK(t && e && sim && K > 0 && D > 0, "cosine_f32: bad arguments");
  LL_LAUNCH_KERNEL(cosine_f32_kernel, dim3((K + 3) / 4), dim3(256), 0, (hipStream_t)stream, t, e, sim, K, D);
  LL_LAUNCH_CHECK("cosine_f32");
  return LLMSEG_OK;
}
