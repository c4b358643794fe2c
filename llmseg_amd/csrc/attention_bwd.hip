// Fused attention backward for gfx950 (flash-style: scores are recomputed tile by tile, nothing of size Nq x Nk touches HBM).
//
// Used for the trainable attentions of the LLM-Seg path: Llama causal + key-padding self attention (hd 128, LoRA on q/v ->
// gradients flow to every layer's qkv; reference call site llava_llama.py:93-102 through HF LlamaAttention) and the
// mask-selection head's self attention (model/transformer.py:319-341).
//
//   P  = exp2(S * scale*log2e - LSE2)           S = Q K^T (raw), LSE2 = the forward kernel's row log2-sum-exp
//   dV = P^T dO      dP = dO V^T      D = rowsum(dO * O)      dS = scale * P * (dP - D)      dQ = dS K      dK = dS^T Q
//
// ONE kernel template, three modes, each producing one gradient with the forward kernel's structure (4 wave64, every wave owns
// 32 "owner" rows whose fragments live in registers as MFMA-B operands; the other side is walked in tiles of 64 rows staged
// through LDS; accumulators hold the TRANSPOSED output so a lane owns one owner row):
//   mode  owner   staged tile (natural | natural | transposed)   MFMAs per tile
//   dQ    query   K | V | K^T                                    S^T = K.Q^T, dP^T = V.dO^T, dQ^T += K^T.dS^T
//   dK    key     Q | dO | Q^T                                   S = Q.K^T,   dP = dO.V^T,   dK^T += Q^T.dS
//   dV    key     Q | - | dO^T                                   S = Q.K^T,                  dV^T += dO^T.P
// In the score accumulator a lane owns one owner column and its registers run over the staged rows, so P / dS feed the output
// MFMA as the B operand in exactly the k-slot order they already have (no lane movement), as in the forward kernel; the staged
// operand of that MFMA is transposed while it is staged (4x8 register transposes, 8-byte LDS writes).  Row statistics are
// per-lane scalars in mode dQ and come from a 64-entry LDS table in the key-owner modes.  dQ also writes D for the dK pass.
// 8 tile-GEMMs instead of the minimal 5 buy three simple passes at 2 waves/SIMD without a dQ atomic-add pass; the three passes share ONE
// launch (attn_bwd_all_kernel), D = rowsum(dO * O) comes from a small launch ahead of it.
#include <cstdlib>
#include "common.h"
#include "llmseg_hip.h"

namespace {

constexpr int BO = 128, BT = 64, NT = 256;
constexpr float LOG2E = 1.4426950408889634f;
enum { MODE_DQ = 0, MODE_DK = 1, MODE_DV = 2 };

struct BwdP {
  const bf16_t* Q; const bf16_t* K; const bf16_t* V; const bf16_t* O; const bf16_t* dO;
  bf16_t* dQ; bf16_t* dK; bf16_t* dV;
  long qs[3], ks[3], vs[3], os[3], dos[3], dqs[3], dks[3], dvs[3];     // (batch, head, row) strides in elements
  const float* lse; float* delta;
  int batch, heads, Nq, Nk;
  float scale, scale_log2;
  int causal;
  const uint8_t* key_mask;
};

__device__ __forceinline__ uint32_t bperm_lo(uint32_t a, uint32_t b) { return (a & 0xffffu) | (b << 16); }
__device__ __forceinline__ uint32_t bperm_hi(uint32_t a, uint32_t b) { return (a >> 16) | (b & 0xffff0000u); }

template <int HD>
constexpr int bwd_smem_bytes(bool has2) { return BT * ((HD + 8) * 2) * (has2 ? 2 : 1) + HD * ((BT + 4) * 2) + BT * 8; }

// One owner block (128 owner rows) of one mode.  `smem`: bwd_smem_bytes<HD>(MODE != MODE_DV) bytes of LDS, 16-byte aligned.
template <int HD, int MODE>
__device__ __forceinline__ void attn_bwd_body(const BwdP& p, const int own_block, char* smem) {
  constexpr int KS = HD / 16, DT = HD / 32, CH = HD / 8;
  constexpr int PK = (HD + 8) * 2;            // natural tile row pitch (bytes): CH + 1 chunks -> conflict-free ds_read_b128
  constexpr int PV = (BT + 4) * 2;            // transposed tile row pitch (bytes)
  constexpr bool OWNER_Q = MODE == MODE_DQ;
  constexpr bool HAS2 = MODE != MODE_DV;      // second natural tile + second owner fragment set (the dP product)
  static_assert(HD % 32 == 0 && 16 * CH <= NT, "head_dim in {32, 64, 128}");
  char* Y1 = smem;
  char* Y2 = smem + BT * PK;
  char* Yt = smem + BT * PK * (HAS2 ? 2 : 1);
  float* st_l = reinterpret_cast<float*>(Yt + HD * PV);
  float* st_d = st_l + BT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int own0 = own_block * BO;
  const int Nown = OWNER_Q ? p.Nq : p.Nk, Nst = OWNER_Q ? p.Nk : p.Nq;
  const int wo0 = own0 + wave * 32;
  const int ow = wo0 + ql;
  const int owc = min(ow, Nown - 1);

  const bf16_t* Qg = p.Q + (long)b * p.qs[0] + (long)h * p.qs[1];
  const bf16_t* Kg = p.K + (long)b * p.ks[0] + (long)h * p.ks[1];
  const bf16_t* Vg = p.V + (long)b * p.vs[0] + (long)h * p.vs[1];
  const bf16_t* dOg = p.dO + (long)b * p.dos[0] + (long)h * p.dos[1];
  const long stat0 = ((long)b * p.heads + h) * p.Nq;

  // ---- owner fragments (MFMA B operands): col = owner row, k-slots = 8 consecutive d ----------------------------------
  const bf16_t* x1 = OWNER_Q ? Qg + (long)owc * p.qs[2] : Kg + (long)owc * p.ks[2];
  const bf16_t* x2 = OWNER_Q ? dOg + (long)owc * p.dos[2] : Vg + (long)owc * p.vs[2];
  bf16x8_t xf1[KS], xf2[HAS2 ? KS : 1];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    xf1[ks] = *reinterpret_cast<const bf16x8_t*>(x1 + ks * 16 + half * 8);
    if (HAS2) xf2[ks] = *reinterpret_cast<const bf16x8_t*>(x2 + ks * 16 + half * 8);
  }
  bool own_ok = ow < Nown;
  float lse_o = 0.f, d_o = 0.f;
  if (OWNER_Q) {
    lse_o = p.lse[stat0 + owc];
    const bf16_t* orow = p.O + (long)b * p.os[0] + (long)h * p.os[1] + (long)owc * p.os[2];
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float of[8], df[8];
      unpack8(*reinterpret_cast<const uint4*>(orow + ks * 16 + half * 8), of);
      unpack8(__builtin_bit_cast(uint4, xf2[HAS2 ? ks : 0]), df);
#pragma unroll
      for (int e = 0; e < 8; ++e) part = fmaf(of[e], df[e], part);
    }
    d_o = part + __shfl_xor(part, 32, 64);              // = delta[ow] (attn_delta_kernel writes the table the dK blocks read)
  } else if (p.key_mask) {
    own_ok = own_ok && (p.key_mask[(long)b * p.Nk + owc] != 0);
  }

  // ---- staged-tile range (causal: keys <= query) -------------------------------------------------------------------------
  int t_begin = 0, t_end = (Nst + BT - 1) / BT;
  if (p.causal) {
    if (OWNER_Q) t_end = min(t_end, (min(own0 + BO, p.Nq) + BT - 1) / BT);
    else t_begin = own0 / BT;
  }

  const bf16_t* sa = OWNER_Q ? Kg : Qg;                   // staged tensor A: natural -> Y1 (and transposed -> Yt unless mode dV)
  const long sa_r = OWNER_Q ? p.ks[2] : p.qs[2];
  const bf16_t* sb = OWNER_Q ? Vg : dOg;                  // staged tensor B: natural -> Y2 (modes dQ, dK) / transposed -> Yt (mode dV)
  const long sb_r = OWNER_Q ? p.vs[2] : p.dos[2];
  const int kq = tid & 15, cc = tid >> 4;
  const bool st_on = tid < 16 * CH;

  f32x16_t acc[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;

  for (int t = t_begin; t < t_end; ++t) {
    const int t0 = t * BT;
    __syncthreads();                                      // the previous tile's fragment reads are done
    if (st_on) {
      uint4 ra[4], rb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long row = min(t0 + 4 * kq + j, Nst - 1);
        ra[j] = *reinterpret_cast<const uint4*>(sa + row * sa_r + cc * 8);
        rb[j] = *reinterpret_cast<const uint4*>(sb + row * sb_r + cc * 8);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<uint4*>(Y1 + (4 * kq + j) * PK + cc * 16) = ra[j];
        if (HAS2) *reinterpret_cast<uint4*>(Y2 + (4 * kq + j) * PK + cc * 16) = rb[j];
      }
      // 4 rows x 8 d -> 8 d-rows of 4 consecutive staged rows (8 bytes each)
      const uint4 r0 = HAS2 ? ra[0] : rb[0], r1 = HAS2 ? ra[1] : rb[1], r2 = HAS2 ? ra[2] : rb[2], r3 = HAS2 ? ra[3] : rb[3];
      char* dst = Yt + (8 * cc) * PV + 8 * kq;
      *reinterpret_cast<uint2*>(dst + 0 * PV) = make_uint2(bperm_lo(r0.x, r1.x), bperm_lo(r2.x, r3.x));
      *reinterpret_cast<uint2*>(dst + 1 * PV) = make_uint2(bperm_hi(r0.x, r1.x), bperm_hi(r2.x, r3.x));
      *reinterpret_cast<uint2*>(dst + 2 * PV) = make_uint2(bperm_lo(r0.y, r1.y), bperm_lo(r2.y, r3.y));
      *reinterpret_cast<uint2*>(dst + 3 * PV) = make_uint2(bperm_hi(r0.y, r1.y), bperm_hi(r2.y, r3.y));
      *reinterpret_cast<uint2*>(dst + 4 * PV) = make_uint2(bperm_lo(r0.z, r1.z), bperm_lo(r2.z, r3.z));
      *reinterpret_cast<uint2*>(dst + 5 * PV) = make_uint2(bperm_hi(r0.z, r1.z), bperm_hi(r2.z, r3.z));
      *reinterpret_cast<uint2*>(dst + 6 * PV) = make_uint2(bperm_lo(r0.w, r1.w), bperm_lo(r2.w, r3.w));
      *reinterpret_cast<uint2*>(dst + 7 * PV) = make_uint2(bperm_hi(r0.w, r1.w), bperm_hi(r2.w, r3.w));
    }
    if (!OWNER_Q && tid < BT) {
      const long qi = stat0 + min(t0 + tid, p.Nq - 1);
      st_l[tid] = p.lse[qi];
      if (MODE == MODE_DK) st_d[tid] = p.delta[qi];
    }
    __syncthreads();

    // wave-level skips: nothing owned, or (causal) this tile lies entirely on the masked side of this wave's rows
    if (wo0 >= Nown) continue;
    if (p.causal && (OWNER_Q ? t0 > wo0 + 31 : t0 + BT - 1 < wo0)) continue;

    f32x16_t s[2], dp[2];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[jb][e] = 0.f; dp[jb][e] = 0.f; }
    // k-step outer: consecutive MFMAs cycle through the (up to four) accumulators
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(Y1 + (jb * 32 + ql) * PK + (2 * ks + half) * 16);
        s[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xf1[ks], s[jb], 0, 0, 0);
        if (HAS2) {
          const bf16x8_t a2 = *reinterpret_cast<const bf16x8_t*>(Y2 + (jb * 32 + ql) * PK + (2 * ks + half) * 16);
          dp[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xf2[ks], dp[jb], 0, 0, 0);
        }
      }
    // key-padding mask of the staged keys (mode dQ): one byte load per lane + a ballot per tile, pre-shifted by the lane half's 4
    uint32_t km_lo = 0xffffffffu, km_hi = 0xffffffffu;
    if (OWNER_Q && p.key_mask) {
      const unsigned long long kb = __ballot(p.key_mask[(long)b * p.Nk + min(t0 + lane, p.Nk - 1)] != 0) >> (4 * half);
      km_lo = (uint32_t)kb; km_hi = (uint32_t)(kb >> 32);
    }
    // P (mode dV) or dS (modes dQ, dK), in place
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int loc = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int sr = t0 + loc;
        bool ok = own_ok && sr < Nst;
        if (p.causal) ok = ok && (OWNER_Q ? sr <= ow : ow <= sr);
        if (OWNER_Q) ok = ok && (((jb ? km_hi : km_lo) >> ((r & 3) + 8 * (r >> 2))) & 1u);
        const float lse = OWNER_Q ? lse_o : st_l[loc];
        const float pv = ok ? __builtin_amdgcn_exp2f(fmaf(s[jb][r], p.scale_log2, -lse)) : 0.f;
        if (MODE == MODE_DV) {
          s[jb][r] = pv;
        } else {
          const float dd = OWNER_Q ? d_o : st_d[loc];
          s[jb][r] = pv * (dp[jb][r] - dd) * p.scale;
        }
      }
    // out^T += Yt . (P | dS): k-steps of 16 staged rows; the B fragment of step ss = accumulator regs 8*(ss&1)..+7 of block ss>>1
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
      const int jb = ss >> 1, rb = 8 * (ss & 1);
      const uint4 pu = make_uint4(pack2bf(s[jb][rb + 0], s[jb][rb + 1]), pack2bf(s[jb][rb + 2], s[jb][rb + 3]),
                                  pack2bf(s[jb][rb + 4], s[jb][rb + 5]), pack2bf(s[jb][rb + 6], s[jb][rb + 7]));
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pu);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const char* vrow = Yt + (d * 32 + ql) * PV + (16 * ss + 4 * half) * 2;
        const uint2 va = *reinterpret_cast<const uint2*>(vrow);
        const uint2 vb = *reinterpret_cast<const uint2*>(vrow + 16);
        const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, make_uint4(va.x, va.y, vb.x, vb.y));
        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, acc[d], 0, 0, 0);
      }
    }
  }

  // ---- store: lane holds out[owner][d .. d+3] groups ----------------------------------------------------------------------
  if (ow < Nown) {
    bf16_t* orow = MODE == MODE_DQ   ? p.dQ + (long)b * p.dqs[0] + (long)h * p.dqs[1] + (long)ow * p.dqs[2]
                   : MODE == MODE_DK ? p.dK + (long)b * p.dks[0] + (long)h * p.dks[1] + (long)ow * p.dks[2]
                                     : p.dV + (long)b * p.dvs[0] + (long)h * p.dvs[1] + (long)ow * p.dvs[2];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = d * 32 + 8 * g + 4 * half;
        *reinterpret_cast<uint2*>(orow + dd) = make_uint2(pack2bf(acc[d][4 * g], acc[d][4 * g + 1]), pack2bf(acc[d][4 * g + 2], acc[d][4 * g + 3]));
      }
  }
}

template <int HD, int MODE>
__global__ __launch_bounds__(NT, 2) void attn_bwd_kernel(BwdP p) {
  __shared__ __attribute__((aligned(16))) char smem[bwd_smem_bytes<HD>(MODE != MODE_DV)];
  attn_bwd_body<HD, MODE>(p, blockIdx.x, smem);
}

// All three gradients in ONE launch: blockIdx.x = [dQ owner blocks | dK owner blocks | dV owner blocks].  At the Llama shape (T = 319,
// 2 sequences x 32 heads) each pass is 192 workgroups on 256 CUs with a serial chain of <= 5 tiles -- latency-bound, 43 + 32 + 29 us as
// three launches; side by side the 576 workgroups share the CUs (two resident per CU) and the launch costs what its longest pass costs.
// The dK blocks read D = rowsum(dO * O) from `delta`, which therefore comes from its own small launch (attn_delta_kernel) ahead of this
// one instead of from the dQ pass.  Causal: the heavy blocks (late query blocks, early key blocks) get the low workgroup ids.
template <int HD>
__global__ __launch_bounds__(NT, 2) void attn_bwd_all_kernel(BwdP p, int nbq, int nbk) {
  __shared__ __attribute__((aligned(16))) char smem[bwd_smem_bytes<HD>(true)];
  const int x = blockIdx.x;
  if (x < nbq) attn_bwd_body<HD, MODE_DQ>(p, p.causal ? nbq - 1 - x : x, smem);
  else if (x < nbq + nbk) attn_bwd_body<HD, MODE_DK>(p, x - nbq, smem);
  else attn_bwd_body<HD, MODE_DV>(p, x - nbq - nbk, smem);
}

// delta[b][h][q] = sum_d dO[b][h][q][d] * O[b][h][q][d]: 16 lanes per row (one 16-byte load each per 128 d), 16 rows per workgroup
template <int HD>
__global__ __launch_bounds__(256) void attn_delta_kernel(BwdP p) {
  const int tid = threadIdx.x, sub = tid & 15;
  const long row = (long)blockIdx.x * 16 + (tid >> 4), rows = (long)p.batch * p.heads * p.Nq;
  if (row >= rows) return;
  const int q = (int)(row % p.Nq), h = (int)((row / p.Nq) % p.heads), b = (int)(row / ((long)p.Nq * p.heads));
  const bf16_t* o = p.O + (long)b * p.os[0] + (long)h * p.os[1] + (long)q * p.os[2];
  const bf16_t* g = p.dO + (long)b * p.dos[0] + (long)h * p.dos[1] + (long)q * p.dos[2];
  float part = 0.f;
#pragma unroll
  for (int c = sub; c < HD / 8; c += 16) {
    float of[8], df[8];
    unpack8(*reinterpret_cast<const uint4*>(o + c * 8), of);
    unpack8(*reinterpret_cast<const uint4*>(g + c * 8), df);
#pragma unroll
    for (int e = 0; e < 8; ++e) part = fmaf(of[e], df[e], part);
  }
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
  if (sub == 0) p.delta[row] = part;
}

static const bool g_bwd_split = getenv("LLMSEG_ATTN_BWD_SPLIT") != nullptr;       // A/B: the three passes as three launches

template <int HD>
void launch_bwd(const BwdP& p, hipStream_t s) {
  const int nbq = (p.Nq + BO - 1) / BO, nbk = (p.Nk + BO - 1) / BO;
  const long rows = (long)p.batch * p.heads * p.Nq;
  LL_LAUNCH_KERNEL((attn_delta_kernel<HD>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, p);
  if (g_bwd_split) {
    const dim3 gq(nbq, p.heads, p.batch), gk(nbk, p.heads, p.batch);
    LL_LAUNCH_KERNEL((attn_bwd_kernel<HD, MODE_DQ>), gq, dim3(NT), 0, s, p);
    LL_LAUNCH_KERNEL((attn_bwd_kernel<HD, MODE_DK>), gk, dim3(NT), 0, s, p);
    LL_LAUNCH_KERNEL((attn_bwd_kernel<HD, MODE_DV>), gk, dim3(NT), 0, s, p);
  } else {
    LL_LAUNCH_KERNEL((attn_bwd_all_kernel<HD>), dim3(nbq + 2 * nbk, p.heads, p.batch), dim3(NT), 0, s, p, nbq, nbk);
  }
}

}  // namespace

extern "C" int llmseg_attn_bwd(const llmseg_attn_bwd_args* a, void* stream) {
  LL_CHECK(a && a->struct_size == sizeof(*a), "%s: ABI mismatch: caller's struct_size %u != %zu (bind against include/llmseg_hip.h version %d)",
           "attn_bwd", a ? a->struct_size : 0u, sizeof(*a), LLMSEG_ABI_VERSION);
  LL_CHECK(a && a->Q && a->K && a->V && a->O && a->dO && a->dQ && a->dK && a->dV && a->lse && a->delta, "attn_bwd: null pointer");
  LL_CHECK(a->batch > 0 && a->heads > 0 && a->Nq > 0 && a->Nk > 0, "attn_bwd: bad sizes");
  LL_CHECK(a->head_dim == 32 || a->head_dim == 64 || a->head_dim == 128, "attn_bwd: head_dim %d unsupported (32, 64, 128)", a->head_dim);
  LL_CHECK(!a->causal || a->Nq == a->Nk, "attn_bwd: causal needs Nq == Nk");
  const int64_t all8 = a->q_stride_b | a->q_stride_h | a->q_stride_row | a->k_stride_b | a->k_stride_h | a->k_stride_row | a->v_stride_b |
                       a->v_stride_h | a->v_stride_row | a->o_stride_b | a->o_stride_h | a->o_stride_row | a->do_stride_b | a->do_stride_h |
                       a->do_stride_row;
  LL_CHECK((all8 & 7) == 0, "attn_bwd: Q/K/V/O/dO strides must be multiples of 8 elements");
  const int64_t all4 = a->dq_stride_b | a->dq_stride_h | a->dq_stride_row | a->dk_stride_b | a->dk_stride_h | a->dk_stride_row | a->dv_stride_b |
                       a->dv_stride_h | a->dv_stride_row;
  LL_CHECK((all4 & 3) == 0, "attn_bwd: dQ/dK/dV strides must be multiples of 4 elements");
  LL_CHECK((((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V | (uintptr_t)a->O | (uintptr_t)a->dO) & 15) == 0 &&
               (((uintptr_t)a->dQ | (uintptr_t)a->dK | (uintptr_t)a->dV) & 7) == 0, "attn_bwd: misaligned pointer");
  BwdP p;
  p.Q = (const bf16_t*)a->Q; p.K = (const bf16_t*)a->K; p.V = (const bf16_t*)a->V; p.O = (const bf16_t*)a->O; p.dO = (const bf16_t*)a->dO;
  p.dQ = (bf16_t*)a->dQ; p.dK = (bf16_t*)a->dK; p.dV = (bf16_t*)a->dV;
  p.qs[0] = a->q_stride_b; p.qs[1] = a->q_stride_h; p.qs[2] = a->q_stride_row;
  p.ks[0] = a->k_stride_b; p.ks[1] = a->k_stride_h; p.ks[2] = a->k_stride_row;
  p.vs[0] = a->v_stride_b; p.vs[1] = a->v_stride_h; p.vs[2] = a->v_stride_row;
  p.os[0] = a->o_stride_b; p.os[1] = a->o_stride_h; p.os[2] = a->o_stride_row;
  p.dos[0] = a->do_stride_b; p.dos[1] = a->do_stride_h; p.dos[2] = a->do_stride_row;
  p.dqs[0] = a->dq_stride_b; p.dqs[1] = a->dq_stride_h; p.dqs[2] = a->dq_stride_row;
  p.dks[0] = a->dk_stride_b; p.dks[1] = a->dk_stride_h; p.dks[2] = a->dk_stride_row;
  p.dvs[0] = a->dv_stride_b; p.dvs[1] = a->dv_stride_h; p.dvs[2] = a->dv_stride_row;
  p.lse = a->lse; p.delta = a->delta;
  p.batch = a->batch; p.heads = a->heads; p.Nq = a->Nq; p.Nk = a->Nk;
  p.scale = a->scale; p.scale_log2 = a->scale * LOG2E;
  p.causal = a->causal; p.key_mask = a->key_mask;
  hipStream_t s = (hipStream_t)stream;
  switch (a->head_dim) {
    case 32: launch_bwd<32>(p, s); break;
    case 64: launch_bwd<64>(p, s); break;
    default: launch_bwd<128>(p, s); break;
  }
  LL_LAUNCH_CHECK("attn_bwd");
  return LLMSEG_OK;
}
