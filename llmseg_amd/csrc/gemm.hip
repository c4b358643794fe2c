// bf16 MFMA GEMM for gfx950:  C[m][n] = epi(alpha * sum_k A[m][k] * W[n][k])   ("TN": both operands K-contiguous)
//
// Replaces every nn.Linear / 1x1-conv / patch-embed on the LLM-Seg hot path (see include/llmseg_hip.h).
//
// Design (CDNA4): 256-thread workgroup = 4 wave64, 128x128 output tile, BK = 64.  Each wave owns a 64x64
// sub-tile as 2x2 v_mfma_f32_32x32x16_bf16 accumulators (64 fp32 regs/lane).  The MFMA "A" operand is the
// WEIGHT fragment and the "B" operand the ACTIVATION fragment, so a lane ends up holding 4 consecutive
// output columns (n) of one output row (m): bias/LayerScale/residual/activation fuse into the epilogue and
// the bf16 store is 8 bytes per lane.
// Tiles are staged global -> VGPR -> LDS (16-byte chunks, XOR-swizzled on (row>>1)&7 so that the
// ds_read_b128 fragment loads are bank-conflict free for the 16-lane service groups of gfx950) and the loop
// is software pipelined: the global loads of tile t+1 are issued before the MFMAs of tile t and written to the
// other LDS buffer afterwards -> one barrier per K-tile.  Workgroup ids are remapped so that the 8 XCDs
// (private L2s) each walk a contiguous, M-grouped range of tiles.
#include "common.h"
#include "llmseg_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NT = 256;
constexpr int TILE_BYTES = BM * BK * 2;       // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;     // A + W
constexpr int GROUP_M = 8;

struct GemmP {
  const bf16_t* A; const bf16_t* W; void* C;
  const bf16_t* bias; const bf16_t* gamma; const bf16_t* res;
  int M, N, K;
  long lda, ldw, ldc, ldr, sA, sW, sC;
  float alpha;
  int act;
  int tiles_m, tiles_n;
  int c_vec, r_vec;   // host-verified alignment for vector C stores / residual loads
};

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case LLMSEG_ACT_RELU: return fmaxf(v, 0.f);
    case LLMSEG_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    case LLMSEG_ACT_QUICKGELU: return v / (1.f + __expf(-1.702f * v));
    case LLMSEG_ACT_SILU: return v / (1.f + __expf(-v));
    case LLMSEG_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * (BK * 2) + (((chunk ^ (row >> 1)) & 7) << 4); }

template <bool OUT_F32>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_tn_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];

  // ---- workgroup -> tile: XCD-contiguous (bid % 8 is the XCD), then GROUP_M-grouped, M fastest ----------------
  const int nwg = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective for any nwg
  }
  const int per_group = GROUP_M * p.tiles_n;
  const int grp = bid / per_group;
  const int first_m = grp * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int tile_m = first_m + (bid % per_group) % gsz;
  const int tile_n = (bid % per_group) / gsz;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long bz = blockIdx.y;
  const bf16_t* __restrict__ Ag = p.A + bz * p.sA;
  const bf16_t* __restrict__ Wg = p.W + bz * p.sW;

  // ---- staging map: thread owns chunk kc of rows (tid>>3)+32*i ------------------------------------------------
  const int kc = tid & 7, r0 = tid >> 3;
  const bf16_t* a_ptr[4];
  const bf16_t* w_ptr[4];
  int st_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + 32 * i;
    a_ptr[i] = Ag + (long)min(m0 + r, p.M - 1) * p.lda + kc * 8;   // rows past M/N: clamp (results discarded)
    w_ptr[i] = Wg + (long)min(n0 + r, p.N - 1) * p.ldw + kc * 8;
    st_off[i] = lds_off(r, kc);
  }
  const int nt = (p.K + BK - 1) / BK;
  uint4 ra[4], rw[4];

  auto load_tile = [&](int t) {
    const int k = t * BK + kc * 8;
    if (k < p.K) {      // K % 8 == 0: a 16-byte chunk is entirely inside or entirely outside
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const uint4*>(a_ptr[i] + (long)t * BK);
        rw[i] = *reinterpret_cast<const uint4*>(w_ptr[i] + (long)t * BK);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) { ra[i] = make_uint4(0, 0, 0, 0); rw[i] = make_uint4(0, 0, 0, 0); }
    }
  };
  auto store_tile = [&](int buf) {
    char* base = smem + buf * BUF_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(base + st_off[i]) = ra[i];
      *reinterpret_cast<uint4*>(base + TILE_BYTES + st_off[i]) = rw[i];
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) load_tile(t + 1);
    const char* abase = smem + (t & 1) * BUF_BYTES;
    const char* wbase = abase + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8_t af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ar = wm * 64 + i * 32 + frow;
        af[i] = *reinterpret_cast<const bf16x8_t*>(abase + lds_off(ar, ks * 2 + fhalf));
        const int wr = wn * 64 + i * 32 + frow;
        wf[i] = *reinterpret_cast<const bf16x8_t*>(wbase + lds_off(wr, ks * 2 + fhalf));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
    }
    if (t + 1 < nt) store_tile((t + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m][n..n+3] for 4 column groups per accumulator -----------------------------------
  const bool vec_ok = p.c_vec != 0;
  const bool res_vec = p.r_vec != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + frow;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * fhalf;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[j][i][4 * g + e] * p.alpha;
        const int nv = min(4, p.N - n);
        if (p.bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (e < nv) v[e] += bf2f(p.bias[n + e]);
        }
        if (p.act != LLMSEG_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
        }
        if (p.gamma) {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (e < nv) v[e] *= bf2f(p.gamma[n + e]);
        }
        if (p.res) {
          const bf16_t* rp = p.res + bz * p.sC + (long)m * p.ldr + n;
          if (res_vec && nv == 4) {
            const uint2 rv = *reinterpret_cast<const uint2*>(rp);
            v[0] += __uint_as_float(rv.x << 16); v[1] += __uint_as_float(rv.x & 0xffff0000u);
            v[2] += __uint_as_float(rv.y << 16); v[3] += __uint_as_float(rv.y & 0xffff0000u);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < nv) v[e] += bf2f(rp[e]);
          }
        }
        if (OUT_F32) {
          float* cp = reinterpret_cast<float*>(p.C) + bz * p.sC + (long)m * p.ldc + n;
          if (vec_ok && nv == 4) {
            *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < nv) cp[e] = v[e];
          }
        } else {
          bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + bz * p.sC + (long)m * p.ldc + n;
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
}

}  // namespace

// profiling hooks (capi.cpp)
void llmseg_prof_begin(hipStream_t s);
void llmseg_prof_end(hipStream_t s, double flops);

extern "C" int llmseg_gemm_bf16(const llmseg_gemm_args* a, void* stream) {
  LL_CHECK(a && a->A && a->W && a->C, "gemm: null pointer");
  LL_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "gemm: bad shape M=%ld N=%ld K=%ld", (long)a->M, (long)a->N, (long)a->K);
  LL_CHECK((a->K & 7) == 0 && (a->lda & 7) == 0 && (a->ldw & 7) == 0, "gemm: K/lda/ldw must be multiples of 8 (K=%ld lda=%ld ldw=%ld)",
           (long)a->K, (long)a->lda, (long)a->ldw);
  LL_CHECK((((uintptr_t)a->A) & 15) == 0 && (((uintptr_t)a->W) & 15) == 0, "gemm: A/W must be 16-byte aligned");
  LL_CHECK((a->strideA & 7) == 0 && (a->strideW & 7) == 0, "gemm: batch strides of A/W must be multiples of 8");
  const int esz = a->out_f32 ? 4 : 2;
  GemmP p;
  p.A = (const bf16_t*)a->A; p.W = (const bf16_t*)a->W; p.C = a->C;
  p.bias = (const bf16_t*)a->bias; p.gamma = (const bf16_t*)a->gamma; p.res = (const bf16_t*)a->residual;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
  p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc; p.ldr = a->residual ? a->ldr : 0;
  const long batch = a->batch > 0 ? a->batch : 1;
  p.sA = a->strideA; p.sW = a->strideW; p.sC = a->strideC;
  p.alpha = a->alpha; p.act = a->act;
  p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
  // vector stores/loads need 4-element alignment of every row start; otherwise the kernel goes element-wise
  p.c_vec = ((((uintptr_t)a->C) % (4 * esz)) == 0 && (a->ldc & 3) == 0 && (a->strideC & 3) == 0) ? 1 : 0;
  p.r_vec = (p.res && (((uintptr_t)p.res) & 7) == 0 && (p.ldr & 3) == 0 && (a->strideC & 3) == 0) ? 1 : 0;
  dim3 grid(p.tiles_m * p.tiles_n, (unsigned)batch);
  hipStream_t s = (hipStream_t)stream;
  llmseg_prof_begin(s);
  if (a->out_f32) hipLaunchKernelGGL(gemm_bf16_tn_kernel<true>, grid, dim3(NT), 0, s, p);
  else hipLaunchKernelGGL(gemm_bf16_tn_kernel<false>, grid, dim3(NT), 0, s, p);
  llmseg_prof_end(s, 2.0 * (double)a->M * (double)a->N * (double)a->K * (double)batch);
  LL_LAUNCH_CHECK("gemm_bf16_tn");
  return LLMSEG_OK;
}
