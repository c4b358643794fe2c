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
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "llmseg_hip.h"

namespace {

constexpr int BO = 128, BT = 64, NT = 256;
#ifndef BWD_ABLATE
#define BWD_ABLATE 0           // side builds (tools): 1 = no tile compute, 2 = no dq/dk/dv stores, 3 = tiles staged once, 4 = no P / dS element pass, 5 = no owner-fragment loads
#endif
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
  const float* rope_cos; const float* rope_sin;      // optional: rotation of the dQ / dK rows at the store (fp32 [Nq][HD / 2])
};

__device__ __forceinline__ uint32_t bperm_lo(uint32_t a, uint32_t b) { return (a & 0xffffu) | (b << 16); }
__device__ __forceinline__ uint32_t bperm_hi(uint32_t a, uint32_t b) { return (a >> 16) | (b & 0xffff0000u); }

template <int HD>
constexpr int bwd_smem_bytes(bool has2) { return BT * ((HD + 8) * 2) * (has2 ? 2 : 1) + HD * ((BT + 4) * 2) + BT * 8; }

// One owner block (128 owner rows) of one mode.  `smem`: bwd_smem_bytes<HD>(MODE != MODE_DV) bytes of LDS, 16-byte aligned.
template <int HD, int MODE>
__device__ __forceinline__ void attn_bwd_body(const BwdP& p, const int own_block, const int b, const int h, char* smem) {
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
    xf1[ks] = *reinterpret_cast<const bf16x8_t*>((BWD_ABLATE == 5 ? (OWNER_Q ? Qg : Kg) : x1) + ks * 16 + half * 8);
    if (HAS2) xf2[ks] = *reinterpret_cast<const bf16x8_t*>((BWD_ABLATE == 5 ? (OWNER_Q ? dOg : Vg) : x2) + ks * 16 + half * 8);
  }
  bool own_ok = ow < Nown;
  float lse_o = 0.f, d_o = 0.f, d_os = 0.f;
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
    d_os = d_o * p.scale;
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
  const bool st_on = (16 * CH >= NT) || tid < 16 * CH;    // head_dim 128: every thread stages (no branch around the prefetch)

  f32x16_t acc[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;

  // Staging registers of the NEXT tile (round 6): its rows are fetched while the current tile is computed and written to LDS at the top of
  // the next iteration -- the chain of <= 5 tiles per workgroup used to pay one exposed global round trip per tile (51 -> see profiles).
  uint4 ra0 = make_uint4(0, 0, 0, 0), ra1 = ra0, ra2 = ra0, ra3 = ra0, rb0 = ra0, rb1 = ra0, rb2 = ra0, rb3 = ra0;
  float r_lse = 0.f, r_dl = 0.f;
  unsigned km_nxt = 1u, km_cur = 1u;
#define BW_LOAD(T)                                                                                              \
  {                                                                                                             \
    const int t0_ = (T) * BT;                                                                                   \
    if (OWNER_Q && p.key_mask) km_nxt = p.key_mask[(long)b * p.Nk + min(t0_ + lane, p.Nk - 1)];               \
    if (!OWNER_Q && tid < BT) {                                                                                 \
      const long qi_ = stat0 + min(t0_ + tid, p.Nq - 1);                                                        \
      r_lse = p.lse[qi_];                                                                                       \
      if (MODE == MODE_DK) r_dl = p.delta[qi_];                                                                 \
    }                                                                                                           \
    if (st_on) {                                                                                                \
      const long w0_ = min(t0_ + 4 * kq + 0, Nst - 1), w1_ = min(t0_ + 4 * kq + 1, Nst - 1);                    \
      const long w2_ = min(t0_ + 4 * kq + 2, Nst - 1), w3_ = min(t0_ + 4 * kq + 3, Nst - 1);                    \
      ra0 = *reinterpret_cast<const uint4*>(sa + w0_ * sa_r + cc * 8); rb0 = *reinterpret_cast<const uint4*>(sb + w0_ * sb_r + cc * 8); \
      ra1 = *reinterpret_cast<const uint4*>(sa + w1_ * sa_r + cc * 8); rb1 = *reinterpret_cast<const uint4*>(sb + w1_ * sb_r + cc * 8); \
      ra2 = *reinterpret_cast<const uint4*>(sa + w2_ * sa_r + cc * 8); rb2 = *reinterpret_cast<const uint4*>(sb + w2_ * sb_r + cc * 8); \
      ra3 = *reinterpret_cast<const uint4*>(sa + w3_ * sa_r + cc * 8); rb3 = *reinterpret_cast<const uint4*>(sb + w3_ * sb_r + cc * 8); \
    }                                                                                                           \
  }
#define BW_PIN4(r) "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w)
  // hipcc otherwise hoists the register transposes of BW_STORE (and with them the wait for the whole prefetch) in front of the tile's MFMAs
#define BW_PIN                                                                                                  \
  {                                                                                                             \
    asm volatile("" : BW_PIN4(ra0), BW_PIN4(ra1), BW_PIN4(ra2), BW_PIN4(ra3));                                  \
    asm volatile("" : BW_PIN4(rb0), BW_PIN4(rb1), BW_PIN4(rb2), BW_PIN4(rb3), "+v"(r_lse), "+v"(r_dl), "+v"(km_nxt)); \
  }
#define BW_STORE                                                                                                \
  {                                                                                                             \
    if (st_on) {                                                                                                \
      *reinterpret_cast<uint4*>(Y1 + (4 * kq + 0) * PK + cc * 16) = ra0;                                        \
      *reinterpret_cast<uint4*>(Y1 + (4 * kq + 1) * PK + cc * 16) = ra1;                                        \
      *reinterpret_cast<uint4*>(Y1 + (4 * kq + 2) * PK + cc * 16) = ra2;                                        \
      *reinterpret_cast<uint4*>(Y1 + (4 * kq + 3) * PK + cc * 16) = ra3;                                        \
      if (HAS2) {                                                                                               \
        *reinterpret_cast<uint4*>(Y2 + (4 * kq + 0) * PK + cc * 16) = rb0;                                      \
        *reinterpret_cast<uint4*>(Y2 + (4 * kq + 1) * PK + cc * 16) = rb1;                                      \
        *reinterpret_cast<uint4*>(Y2 + (4 * kq + 2) * PK + cc * 16) = rb2;                                      \
        *reinterpret_cast<uint4*>(Y2 + (4 * kq + 3) * PK + cc * 16) = rb3;                                      \
      }                                                                                                         \
      /* 4 rows x 8 d -> 8 d-rows of 4 consecutive staged rows (8 bytes each) */                                \
      const uint4 r0 = HAS2 ? ra0 : rb0, r1 = HAS2 ? ra1 : rb1, r2 = HAS2 ? ra2 : rb2, r3 = HAS2 ? ra3 : rb3;   \
      char* dst = Yt + (8 * cc) * PV + 8 * kq;                                                                  \
      *reinterpret_cast<uint2*>(dst + 0 * PV) = make_uint2(bperm_lo(r0.x, r1.x), bperm_lo(r2.x, r3.x));         \
      *reinterpret_cast<uint2*>(dst + 1 * PV) = make_uint2(bperm_hi(r0.x, r1.x), bperm_hi(r2.x, r3.x));         \
      *reinterpret_cast<uint2*>(dst + 2 * PV) = make_uint2(bperm_lo(r0.y, r1.y), bperm_lo(r2.y, r3.y));         \
      *reinterpret_cast<uint2*>(dst + 3 * PV) = make_uint2(bperm_hi(r0.y, r1.y), bperm_hi(r2.y, r3.y));         \
      *reinterpret_cast<uint2*>(dst + 4 * PV) = make_uint2(bperm_lo(r0.z, r1.z), bperm_lo(r2.z, r3.z));         \
      *reinterpret_cast<uint2*>(dst + 5 * PV) = make_uint2(bperm_hi(r0.z, r1.z), bperm_hi(r2.z, r3.z));         \
      *reinterpret_cast<uint2*>(dst + 6 * PV) = make_uint2(bperm_lo(r0.w, r1.w), bperm_lo(r2.w, r3.w));         \
      *reinterpret_cast<uint2*>(dst + 7 * PV) = make_uint2(bperm_hi(r0.w, r1.w), bperm_hi(r2.w, r3.w));         \
    }                                                                                                           \
    if (!OWNER_Q && tid < BT) {                                                                                 \
      st_l[tid] = r_lse;                                                                                        \
      if (MODE == MODE_DK) st_d[tid] = r_dl;                                                                    \
    }                                                                                                           \
    km_cur = km_nxt;                                                                                            \
  }

  if (t_begin < t_end) BW_LOAD(t_begin)
  // the owner fragments (and the first tile) must have LANDED before the loop: a load still pending at the loop header becomes a vmcnt
  // wait inside every iteration, which would drain that iteration's prefetch early
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    asm volatile("" : "+v"(xf1[ks]));
    if (HAS2) asm volatile("" : "+v"(xf2[ks]));
  }
  BW_PIN

  for (int t = t_begin; t < t_end; ++t) {
    const int t0 = t * BT;
    __syncthreads();                                      // the previous tile's fragment reads are done
    BW_STORE
    __syncthreads();
    if (BWD_ABLATE != 3) BW_LOAD(min(t + 1, t_end - 1))      // unconditional (the last iteration re-fetches its own tile): a branch around the loads costs register copies that wait for them

    // wave-level skips: nothing owned, or (causal) this tile lies entirely on the masked side of this wave's rows
    const bool skip = wo0 >= Nown || (p.causal && (OWNER_Q ? t0 > wo0 + 31 : t0 + BT - 1 < wo0));
    if (!skip && BWD_ABLATE != 1) {
      // key-padding mask of the staged keys (mode dQ): one byte per lane (fetched with the tile) + a ballot, pre-shifted by the lane half's 4
      uint32_t km_lo = 0xffffffffu, km_hi = 0xffffffffu;
      if (OWNER_Q && p.key_mask) {
        const unsigned long long kb = __ballot(km_cur != 0) >> (4 * half);
        km_lo = (uint32_t)kb; km_hi = (uint32_t)(kb >> 32);
      }
      // one 32-row block of the staged tile at a time (scores, then its share of the output product): 32 score registers live instead of 64,
      // which is what makes room for the prefetch registers at 2 waves per SIMD
#pragma unroll 1
      for (int jb = 0; jb < 2; ++jb) {
        f32x16_t s, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(Y1 + (jb * 32 + ql) * PK + (2 * ks + half) * 16);
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xf1[ks], s, 0, 0, 0);
          if (HAS2) {
            const bf16x8_t a2 = *reinterpret_cast<const bf16x8_t*>(Y2 + (jb * 32 + ql) * PK + (2 * ks + half) * 16);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xf2[ks], dp, 0, 0, 0);
          }
        }
        // P (mode dV) or dS (modes dQ, dK), in place.  Branch-free: the row statistics of the 16 staged rows this lane's registers run over are four
        // 16-byte LDS reads per table, issued together (the per-element `ok ? ... : 0` with an LDS read inside compiled to 32 branches with a
        // dependent ds_read_b32 each: 18 of the launch's 51 us), the exponential is computed for every element and the mask is a select.
        if (BWD_ABLATE != 4) {
          const uint32_t kmw = jb ? km_hi : km_lo;
          const int sr0 = t0 + jb * 32 + 4 * half;
          // wave-uniform: no element of this 32-row block of the tile is masked for any of the wave's owner rows (interior of the causal triangle,
          // all rows real, no padded key): the mask arithmetic (4-5 of ~9 VALU operations per element) is skipped
          const int blk0 = t0 + jb * 32;
          const bool open_blk = __all(own_ok) && blk0 + 32 <= Nst && (!p.causal || (OWNER_Q ? blk0 + 31 <= wo0 : wo0 + 31 <= blk0)) &&
                                (!OWNER_Q || __all((kmw & 0x0f0f0f0fu) == 0x0f0f0f0fu));      // the 16 mask bits this lane's registers use
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float lg[4] = {lse_o, lse_o, lse_o, lse_o}, dg[4] = {d_os, d_os, d_os, d_os};
            if (!OWNER_Q) {
              const float4 l4 = *reinterpret_cast<const float4*>(st_l + jb * 32 + 8 * g + 4 * half);
              lg[0] = l4.x; lg[1] = l4.y; lg[2] = l4.z; lg[3] = l4.w;
              if (MODE == MODE_DK) {
                const float4 d4 = *reinterpret_cast<const float4*>(st_d + jb * 32 + 8 * g + 4 * half);
                dg[0] = d4.x * p.scale; dg[1] = d4.y * p.scale; dg[2] = d4.z * p.scale; dg[3] = d4.w * p.scale;
              }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * g + e;
              float pv = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2, -lg[e]));
              if (!open_blk) {
                const int sr = sr0 + e + 8 * g;
                bool ok = own_ok && sr < Nst;
                if (p.causal) ok = ok && (OWNER_Q ? sr <= ow : ow <= sr);
                if (OWNER_Q) ok = ok && ((kmw >> (e + 8 * g)) & 1u);
                pv = ok ? pv : 0.f;
              }
              if (MODE == MODE_DV) s[r] = pv;
              else s[r] = pv * fmaf(dp[r], p.scale, -dg[e]);        // dS = scale * P * (dP - D)
            }
          }
        }
        // out^T += Yt . (P | dS): k-steps of 16 staged rows; the B fragment of step ss = registers 8*(ss&1)..+7 of block ss>>1
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int ss = 2 * jb + s2, rb = 8 * s2;
          const uint4 pu = make_uint4(pack2bf(s[rb + 0], s[rb + 1]), pack2bf(s[rb + 2], s[rb + 3]),
                                      pack2bf(s[rb + 4], s[rb + 5]), pack2bf(s[rb + 6], s[rb + 7]));
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
    }
    BW_PIN
  }
#undef BW_LOAD
#undef BW_STORE
#undef BW_PIN
#undef BW_PIN4

  // ---- store: lane holds out[owner][d .. d+3] groups ----------------------------------------------------------------------
  // Optional rotation of the dQ / dK rows (round 6): the inverse RoPE of the gradient (`rope_cos`, `rope_sin` = the NEGATED sine table) at
  // position = owner row, with llmseg_rope's arithmetic on the bf16-rounded values -- bit for bit what a separate llmseg_rope launch over
  // the stored rows gives.  The two elements of a rotate-half pair (d, d + HD/2) sit in accumulators d and d + DT/2 of the same lane.
  if (ow < Nown && (BWD_ABLATE != 2 || acc[0][0] == 12345.f)) {
    bf16_t* orow = MODE == MODE_DQ   ? p.dQ + (long)b * p.dqs[0] + (long)h * p.dqs[1] + (long)ow * p.dqs[2]
                   : MODE == MODE_DK ? p.dK + (long)b * p.dks[0] + (long)h * p.dks[1] + (long)ow * p.dks[2]
                                     : p.dV + (long)b * p.dvs[0] + (long)h * p.dvs[1] + (long)ow * p.dvs[2];
    if (MODE != MODE_DV && p.rope_cos != nullptr) {
      const float* cs = p.rope_cos + (long)ow * (HD / 2);
      const float* sn = p.rope_sin + (long)ow * (HD / 2);
#pragma unroll
      for (int d = 0; d < DT / 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = d * 32 + 8 * g + 4 * half;
          const float4 c4 = *reinterpret_cast<const float4*>(cs + dd), s4 = *reinterpret_cast<const float4*>(sn + dd);
          const float cv[4] = {c4.x, c4.y, c4.z, c4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w};
          float o1[4], o2[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = round_bf(acc[d][4 * g + e]), bb = round_bf(acc[d + DT / 2][4 * g + e]);
            o1[e] = rope_lo(a, bb, cv[e], sv[e]);
            o2[e] = rope_hi(a, bb, cv[e], sv[e]);
          }
          *reinterpret_cast<uint2*>(orow + dd) = make_uint2(pack2bf(o1[0], o1[1]), pack2bf(o1[2], o1[3]));
          *reinterpret_cast<uint2*>(orow + dd + HD / 2) = make_uint2(pack2bf(o2[0], o2[1]), pack2bf(o2[2], o2[3]));
        }
    } else {
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = d * 32 + 8 * g + 4 * half;
          *reinterpret_cast<uint2*>(orow + dd) = make_uint2(pack2bf(acc[d][4 * g], acc[d][4 * g + 1]), pack2bf(acc[d][4 * g + 2], acc[d][4 * g + 3]));
        }
    }
  }
}

template <int HD, int MODE>
__global__ __launch_bounds__(NT, 2) void attn_bwd_kernel(BwdP p) {
  __shared__ __attribute__((aligned(16))) char smem[bwd_smem_bytes<HD>(MODE != MODE_DV)];
  attn_bwd_body<HD, MODE>(p, blockIdx.x, blockIdx.z, blockIdx.y, smem);
}

// All three gradients in ONE launch: blockIdx.x = [dQ owner blocks | dK owner blocks | dV owner blocks].  At the Llama shape (T = 319,
// 2 sequences x 32 heads) each pass is 192 workgroups on 256 CUs with a serial chain of <= 5 tiles -- latency-bound, 43 + 32 + 29 us as
// three launches; side by side the 576 workgroups share the CUs (two resident per CU) and the launch costs what its longest pass costs.
// The dK blocks read D = rowsum(dO * O) from `delta`, which therefore comes from its own small launch (attn_delta_kernel) ahead of this
// one instead of from the dQ pass.  Causal: the heavy blocks (late query blocks, early key blocks) get the low workgroup ids.
template <int HD>
__global__ __launch_bounds__(NT, 2) void attn_bwd_all_kernel(BwdP p, int nbq, int nbk) {
  __shared__ __attribute__((aligned(16))) char smem[bwd_smem_bytes<HD>(true)];
  // 1-D grid, (batch, head) fastest: workgroup id -> (class, b, h), the classes ordered by their tile-chain length so that the workgroups the
  // dispatcher starts LAST (the grid is 576 workgroups for 512 resident slots at the Llama shape) are the shortest ones.  Causal: a dQ block of
  // late queries and a dK / dV block of early keys are the long chains -> class c walks [dq(n-1-c) | dk(c) | dv(c)].
  const int bh = p.batch * p.heads;
  const int cls = blockIdx.x / bh, r = blockIdx.x - cls * bh;
  const int b = r / p.heads, h = r - b * p.heads;
  const int lvl = cls / 3, mode = cls - lvl * 3;           // level 0 = the longest chains
  if (mode == 0) { if (lvl < nbq) attn_bwd_body<HD, MODE_DQ>(p, p.causal ? nbq - 1 - lvl : lvl, b, h, smem); }
  else if (mode == 1) { if (lvl < nbk) attn_bwd_body<HD, MODE_DK>(p, lvl, b, h, smem); }
  else { if (lvl < nbk) attn_bwd_body<HD, MODE_DV>(p, lvl, b, h, smem); }
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
void launch_bwd(const BwdP& p, hipStream_t s, bool delta_ready) {
  const int nbq = (p.Nq + BO - 1) / BO, nbk = (p.Nk + BO - 1) / BO;
  const long rows = (long)p.batch * p.heads * p.Nq;
  if (!delta_ready) LL_LAUNCH_KERNEL((attn_delta_kernel<HD>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, p);
  if (g_bwd_split) {
    const dim3 gq(nbq, p.heads, p.batch), gk(nbk, p.heads, p.batch);
    LL_LAUNCH_KERNEL((attn_bwd_kernel<HD, MODE_DQ>), gq, dim3(NT), 0, s, p);
    LL_LAUNCH_KERNEL((attn_bwd_kernel<HD, MODE_DK>), gk, dim3(NT), 0, s, p);
    LL_LAUNCH_KERNEL((attn_bwd_kernel<HD, MODE_DV>), gk, dim3(NT), 0, s, p);
  } else {
    LL_LAUNCH_KERNEL((attn_bwd_all_kernel<HD>), dim3((unsigned)(3 * std::max(nbq, nbk) * p.heads * p.batch)), dim3(NT), 0, s, p, nbq, nbk);
  }
}

}  // namespace

// library-internal (gemm.hip: the two-launch route of llmseg_gemm_args.dl_o): delta[b][h][q] of packed [batch * T][heads * 128] O / dO rows
extern "C" __attribute__((visibility("hidden"))) int llmseg_attn_delta128(const void* O, int64_t ldo, const void* dO, int64_t lddo, float* delta, int64_t batch, int32_t heads,
                                                                          int64_t T, void* stream) {
  BwdP p{};
  p.O = (const bf16_t*)O; p.dO = (const bf16_t*)dO; p.delta = delta;
  p.os[0] = T * ldo; p.os[1] = 128; p.os[2] = ldo;
  p.dos[0] = T * lddo; p.dos[1] = 128; p.dos[2] = lddo;
  p.batch = (int)batch; p.heads = heads; p.Nq = (int)T; p.Nk = (int)T;
  const long rows = (long)batch * heads * T;
  LL_LAUNCH_KERNEL((attn_delta_kernel<128>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, p);
  LL_LAUNCH_CHECK("attn_delta");
  return LLMSEG_OK;
}

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
  p.rope_cos = a->rope_cos; p.rope_sin = a->rope_sin;
  LL_CHECK((a->rope_cos == nullptr) == (a->rope_sin == nullptr), "attn_bwd: rope_cos and rope_sin come together");
  LL_CHECK(!a->rope_cos || (a->head_dim >= 64 && a->Nq == a->Nk && (((uintptr_t)a->rope_cos | (uintptr_t)a->rope_sin) & 15) == 0),
           "attn_bwd: the fused rotation needs head_dim 64 or 128, Nq == Nk (self attention: position = row) and 16-byte aligned tables");
  hipStream_t s = (hipStream_t)stream;
  switch (a->head_dim) {
    case 32: launch_bwd<32>(p, s, a->delta_ready != 0); break;
    case 64: launch_bwd<64>(p, s, a->delta_ready != 0); break;
    default: launch_bwd<128>(p, s, a->delta_ready != 0); break;
  }
  LL_LAUNCH_CHECK("attn_bwd");
  return LLMSEG_OK;
}
