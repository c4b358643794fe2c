// Fused attention forward for gfx950 (flash-style: online softmax, scores never leave the CU).
//
// One kernel template covers the four attention shapes on the LLM-Seg hot path (include/llmseg_hip.h):
//   Llama causal + key-padding (hd 128), CLIP/DINOv2 ViT global (hd 64), SAM ViT-H windowed/global with
//   decomposed relative position (hd 80), mask-selection head (hd 32).
//
// CDNA4 mapping: workgroup = 4 wave64 = 128 query rows (8 waves = 256 rows for long non-causal sequences); each wave owns 32
// queries and walks K/V in tiles of 64 keys staged through LDS (two tile buffers on the plain / global-grid paths: the next tile is
// written while the current one is read, one barrier per tile).  Scores are computed TRANSPOSED, S^T = K . Q^T, with v_mfma_f32_32x32x16_bf16
// (A = K fragment from LDS, B = Q fragment held in registers for the whole kernel), so that every lane owns
// one query column: the online-softmax max/sum are in-lane reductions plus ONE cross-half exchange, and
// the O rescale is a per-lane scalar.  The exponentiated scores are packed to bf16 in registers and fed back
// as the B operand of O^T = V^T . P^T; the k-slot order of that MFMA is chosen to be exactly the order the
// S^T accumulator registers already have, so P never moves between lanes.  V is transposed while it is staged
// (global rows -> 4x8 register transpose -> 8-byte LDS writes), K is stored row-major with a one-chunk pad
// (conflict-free ds_read_b128 on gfx950's 16-lane service groups).
#include "common.h"
#include "llmseg_hip.h"
#include <algorithm>
#include <cstdlib>
#define AL16(p) ((((uintptr_t)(p)) & 15) == 0)

namespace {

#ifndef ATTN_ABLATE
#define ATTN_ABLATE 0          // side builds (tools/attn_ablate.sh): 1 = exp -> identity, 2 = K/V staged once, 3 = 2 + no barrier, 4 = no PV MFMAs, 5 = no QK MFMAs
#endif
#ifndef WIN_ABLATE
#define WIN_ABLATE 0           // side builds of attn_win14_kernel: 1 = exp -> identity, 2 = K/V staged once, 3 = no rel-pos, 4 = no P.V MFMAs, 5 = no Q.K^T MFMAs
#endif
#if ATTN_ABLATE == 1
#define ATTN_EXP(x) (x)
#else
#define ATTN_EXP(x) __builtin_amdgcn_exp2f(x)
#endif
constexpr int BKV = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG = -1.0e30f;

struct AttnP {
  const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O;
  long qsb, qsh, qsr, ksb, ksh, ksr, vsb, vsh, vsr, osb, osh, osr;
  int batch, heads, Nq, Nk;
  float scale_log2;   // scale * log2(e): scores are exponentiated with exp2(raw * scale_log2 - max * scale_log2)
  float inv_scale;    // 1 / scale: relative-position bias enters the MFMA accumulator in the raw (unscaled) domain
  int causal;
  const uint8_t* key_mask;
  const float* rel_h; const float* rel_w; int rel_ld, gh, gw;
  const int32_t* o_row_map;
  float* lse;                                   // optional [batch][heads][Nq]: row log2-sum-exp for llmseg_attn_bwd
  const bf16_t* rtab_h; const bf16_t* rtab_w;   // REL == 4: bf16 [32][head_dim] relative-position tables (rows >= 2*14-1 are zero)
  const int32_t* nk_dev;                        // optional: the key count is read from device memory (decode steps replayed from a hipGraph)
  int xcd_nqb;                                  // > 0: 1-D grid, XCD-grouped (see attn_fwd_kernel); = query blocks per (batch, head)
  // window gather (attn_win14_dma_kernel<true>): Q / K / V / O rows are UNPARTITIONED token rows of images on a win_grid x win_grid grid; window (wy, wx) of image i
  // (batch index i * win_nw^2 + wy * win_nw + wx) reads token (14 wy + iy, 14 wx + ix) or, outside the grid, the pad rows (the q|k|v projection of a zero row = its bias)
  int win_grid, win_nw;
  const bf16_t* pad_q; const bf16_t* pad_k; const bf16_t* pad_v;
};

typedef short short4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t perm_lo(uint32_t a, uint32_t b) { return (a & 0xffffu) | (b << 16); }
__device__ __forceinline__ uint32_t perm_hi(uint32_t a, uint32_t b) { return (a >> 16) | (b & 0xffff0000u); }

// REL: 0 = no bias, 1 = generic grid (slow, general), 2 = grid_w == BKV (a K tile is exactly one key row: kh uniform,
// kw = offset), 3 = SAM's 14x14 window (196 keys = 4 unrolled tiles, per-query bias rows live in 28 registers)
// NW = waves per workgroup (32 queries each): 4 by default; 8 for long sequences (every workgroup re-reads ALL keys, so 256
// queries per workgroup halve the L2 -> LDS traffic and the staging work per query; the K/V tile and LDS footprint are the same).
template <int HD, int REL, int NW = 4>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(AttnP p) {
  if (p.nk_dev) p.Nk = min(p.Nk, *p.nk_dev);   // Nk (host) = capacity, *nk_dev = keys present now
  constexpr int NT = NW * 64, BQ = NW * 32;
  constexpr int KS = HD / 16;                 // k-steps of QK^T
  constexpr int DT = (HD + 31) / 32;          // 32-row blocks of O^T
  constexpr int CH = HD / 8;                  // 16-byte chunks per row
  constexpr int PK = (HD + 8) * 2;            // K_lds row pitch (bytes): CH+1 chunks -> odd
  constexpr int PV = (BKV + 4) * 2;           // Vt row pitch (bytes) = 136 = 8 * 17
  constexpr bool WIN = (REL == 3 || REL == 4);            // SAM 14x14 window; 4 = q.R^T computed here (no rel_h/rel_w arrays)
  constexpr int G_SLAB = 32 * 33;                          // per-wave fp32 [32 queries][33] slab per table (REL == 4)
  // K/V tile buffers: two for the plain / global-grid paths (the next tile is written while the current one is read: ONE barrier per
  // tile), one for the window paths (whose per-wave rel-pos slabs already take 34 KiB) and the generic-grid path (single lut)
  constexpr bool DB = (REL == 0 || REL == 2);
  constexpr int TILE_BYTES = BKV * PK + DT * 32 * PV;
  __shared__ __attribute__((aligned(16))) char smem[(DB ? 2 : 1) * TILE_BYTES + BKV * 4 + (REL == 4 ? NW * 2 * G_SLAB * 4 : 0)];
  char* Ks = smem;                // tile being READ
  char* Vt = smem + BKV * PK;
  char* Ks_w = smem;              // tile being WRITTEN (== Ks unless DB)
  char* Vt_w = smem + BKV * PK;
  uint32_t* lut = reinterpret_cast<uint32_t*>(smem + (DB ? 2 : 1) * TILE_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, half = lane >> 5;
  // Workgroup -> (batch, head, query block).  Long sequences (xcd_nqb > 0, 1-D grid): every query block of a (batch, head) re-reads ALL
  // of that head's K / V (1.3 MB at 4096 x 80), so the blocks of one head must share an L2: workgroup id i runs on XCD i % 8, which
  // therefore takes the heads {i % 8 + 8 j} and walks their query blocks consecutively (the 32 CUs of an XCD hold two heads' blocks at a
  // time: 2.6 MB of K / V in its 4 MiB L2).  The 3-D grid put the 16 blocks of a head on 8 different XCDs: every block fetched K / V
  // through the fabric again (5.7 x the algorithmic bytes, profiles/traffic.json round 2).
  int b = blockIdx.z, h = blockIdx.y, qb = blockIdx.x;
  if (p.xcd_nqb > 0) {
    const int id = blockIdx.x, j = id >> 3;
    const int bh = (j / p.xcd_nqb) * 8 + (id & 7);
    if (bh >= p.batch * p.heads) return;
    qb = j % p.xcd_nqb; b = bh / p.heads; h = bh - b * p.heads;
  }
  const int q0 = qb * BQ;
  const int q = q0 + wave * 32 + ql;
  const int qc = min(q, p.Nq - 1);

  const bf16_t* Qg = p.Q + (long)b * p.qsb + (long)h * p.qsh;
  const bf16_t* Kg = p.K + (long)b * p.ksb + (long)h * p.ksh;
  const bf16_t* Vg = p.V + (long)b * p.vsb + (long)h * p.vsh;

  // ---- Q fragments (B operand): col = query, k-slots = 8 consecutive d ------------------------------------------
  bf16x8_t qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8_t*>(Qg + (long)qc * p.qsr + ks * 16 + half * 8);

  // ---- relative-position setup -----------------------------------------------------------------------------------
  int qh = 0, qw = 0;
  const float* relh_row = nullptr;
  const float* relw_row = nullptr;
  float bw_cache[REL == 2 ? 32 : 1];
  if (REL != 0) {
    qh = qc / p.gw; qw = qc - qh * p.gw;
    const long rrow = (((long)h * p.batch + b) * p.Nq + qc) * p.rel_ld;   // [heads][batch*Nq][rel_ld]
    if (REL != 4) {
      relh_row = p.rel_h + rrow;
      relw_row = p.rel_w + rrow;
    }
    if (REL == 2) {
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kw = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          bw_cache[jb * 16 + r] = relw_row[qw - kw + p.gw - 1] * p.inv_scale;
        }
    }
  }

  // REL == 3: per-query bias rows in registers, with the lane-half (which selects key or key+4) folded into the LOAD
  // address so that the tile loop only ever uses compile-time indices: bh_s[kh] = Bh[kh], bh_x[kh] = Bh[kh + half],
  // bw_y[kw] = Bw[(kw + 4*half) % 14].
  float bh_s[WIN ? 14 : 1], bh_x[WIN ? 13 : 1], bw_y[WIN ? 14 : 1];
  if constexpr (REL == 4) {
    // Fused decomposed rel-pos (image_encoder.py:354-392): G^T = R . Q^T on the matrix cores (R = the 27-row table, zero
    // padded to 32 rows; Q fragments are already in registers), bounced through a per-wave LDS slab because each lane needs the
    // entries at its own (qh - kh + 13) / (qw - kw + 13): replaces two skinny fp32 GEMM launches + 41 scattered loads per lane.
    float* gs = reinterpret_cast<float*>(smem + TILE_BYTES + BKV * 4) + wave * (2 * G_SLAB);
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      const bf16_t* tab = tb == 0 ? p.rtab_h : p.rtab_w;
      f32x16_t g;
#pragma unroll
      for (int e = 0; e < 16; ++e) g[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t rf = *reinterpret_cast<const bf16x8_t*>(tab + ql * HD + ks * 16 + half * 8);
        g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf, qf[ks], g, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) gs[tb * G_SLAB + ql * 33 + (e & 3) + 8 * (e >> 2) + 4 * half] = g[e];
    }
    const float* gh = gs + ql * 33;
    const float* gw = gs + G_SLAB + ql * 33;
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      bh_s[j] = gh[qh - j + 13] * p.inv_scale;
      bw_y[j] = gw[qw - ((j + 4 * half) % 14) + 13] * p.inv_scale;
    }
#pragma unroll
    for (int j = 0; j < 13; ++j) bh_x[j] = gh[qh - (j + half) + 13] * p.inv_scale;
  }
  if constexpr (REL == 3) {
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      bh_s[j] = relh_row[qh - j + 13] * p.inv_scale;
      const int kw = (j + 4 * half) % 14;
      bw_y[j] = relw_row[qw - kw + 13] * p.inv_scale;
    }
#pragma unroll
    for (int j = 0; j < 13; ++j) bh_x[j] = relh_row[qh - (j + half) + 13] * p.inv_scale;
  }

  int kv_len = p.Nk;
  if (p.causal) kv_len = min(p.Nk, q0 + BQ);
  const int ntiles = (kv_len + BKV - 1) / BKV;

  // ---- staging registers (explicit scalars: arrays captured by the staging code end up in scratch) ---------------
  constexpr int KIT = (BKV * CH + NT - 1) / NT;        // K items (16B chunks) per thread, <= 4
  static_assert(KIT <= 4 && 16 * CH <= NT, "staging map assumes head_dim <= 128");
  uint4 kreg0 = make_uint4(0, 0, 0, 0), kreg1 = kreg0, kreg2 = kreg0, kreg3 = kreg0;
  uint4 vreg0 = kreg0, vreg1 = kreg0, vreg2 = kreg0, vreg3 = kreg0;
  const int v_kq = tid & 15, v_c = tid >> 4;
  const bool v_on = (16 * CH >= NT) || tid < 16 * CH;     // head_dim 128 with 4 waves: every thread stages V
  const int v_cl = (16 * CH >= NT) ? v_c : v_c % CH;      // the LOADS are unconditional (threads past 16 CH re-read a valid chunk, only the LDS stores are predicated):
                                                          // a branch around the prefetch costs register copies at its join that wait for the loads in front of the MFMAs

#define LL_K_LOAD(IT, REG)                                                                          \
  if constexpr (KIT > IT) {                                                                         \
    const int idx = tid + IT * NT;                                                                  \
    if ((IT + 1) * NT <= BKV * CH || idx < BKV * CH) {                                              \
      const int key = idx / CH, c = idx - key * CH;                                                 \
      REG = *reinterpret_cast<const uint4*>(Kg + (long)min(k0s + key, p.Nk - 1) * p.ksr + c * 8);   \
    }                                                                                               \
  }
#define LL_K_STORE(IT, REG)                                                                         \
  if constexpr (KIT > IT) {                                                                         \
    const int idx = tid + IT * NT;                                                                  \
    if ((IT + 1) * NT <= BKV * CH || idx < BKV * CH) {                                              \
      const int key = idx / CH, c = idx - key * CH;                                                 \
      *reinterpret_cast<uint4*>(Ks_w + key * PK + c * 16) = REG;                                    \
    }                                                                                               \
  }
#define LL_STAGE_LOAD(T)                                                                            \
  {                                                                                                 \
    const int k0s = (T) * BKV;                                                                      \
    LL_K_LOAD(0, kreg0) LL_K_LOAD(1, kreg1) LL_K_LOAD(2, kreg2) LL_K_LOAD(3, kreg3)                 \
    {                                                                                               \
      const bf16_t* vp = Vg + v_cl * 8;                                                             \
      vreg0 = *reinterpret_cast<const uint4*>(vp + (long)min(k0s + 4 * v_kq + 0, p.Nk - 1) * p.vsr); \
      vreg1 = *reinterpret_cast<const uint4*>(vp + (long)min(k0s + 4 * v_kq + 1, p.Nk - 1) * p.vsr); \
      vreg2 = *reinterpret_cast<const uint4*>(vp + (long)min(k0s + 4 * v_kq + 2, p.Nk - 1) * p.vsr); \
      vreg3 = *reinterpret_cast<const uint4*>(vp + (long)min(k0s + 4 * v_kq + 3, p.Nk - 1) * p.vsr); \
    }                                                                                               \
  }
  // V: 4 keys x 8 d per thread, 4x8 register transpose, 8-byte writes Vt[8c + d][4kq .. 4kq+3]
#define LL_STAGE_STORE(T)                                                                           \
  {                                                                                                 \
    LL_K_STORE(0, kreg0) LL_K_STORE(1, kreg1) LL_K_STORE(2, kreg2) LL_K_STORE(3, kreg3)             \
    if (v_on) {                                                                                     \
      char* dst = Vt_w + (8 * v_c) * PV + 8 * v_kq;                                                 \
      *reinterpret_cast<uint2*>(dst + 0 * PV) = make_uint2(perm_lo(vreg0.x, vreg1.x), perm_lo(vreg2.x, vreg3.x)); \
      *reinterpret_cast<uint2*>(dst + 1 * PV) = make_uint2(perm_hi(vreg0.x, vreg1.x), perm_hi(vreg2.x, vreg3.x)); \
      *reinterpret_cast<uint2*>(dst + 2 * PV) = make_uint2(perm_lo(vreg0.y, vreg1.y), perm_lo(vreg2.y, vreg3.y)); \
      *reinterpret_cast<uint2*>(dst + 3 * PV) = make_uint2(perm_hi(vreg0.y, vreg1.y), perm_hi(vreg2.y, vreg3.y)); \
      *reinterpret_cast<uint2*>(dst + 4 * PV) = make_uint2(perm_lo(vreg0.z, vreg1.z), perm_lo(vreg2.z, vreg3.z)); \
      *reinterpret_cast<uint2*>(dst + 5 * PV) = make_uint2(perm_hi(vreg0.z, vreg1.z), perm_hi(vreg2.z, vreg3.z)); \
      *reinterpret_cast<uint2*>(dst + 6 * PV) = make_uint2(perm_lo(vreg0.w, vreg1.w), perm_lo(vreg2.w, vreg3.w)); \
      *reinterpret_cast<uint2*>(dst + 7 * PV) = make_uint2(perm_hi(vreg0.w, vreg1.w), perm_hi(vreg2.w, vreg3.w)); \
    }                                                                                               \
    if (REL == 1 && tid < BKV) {                                                                    \
      const int key = min((T) * BKV + tid, p.Nk - 1);                                               \
      const int kh = key / p.gw;                                                                    \
      lut[tid] = ((uint32_t)kh << 16) | (uint32_t)(key - kh * p.gw);                                \
    }                                                                                               \
  }

  f32x16_t o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = NEG, l_run = 0.f;

  // key-padding byte of this lane's key in tile T.  The double-buffered loop fetches it one tile AHEAD, in front of that tile's staging loads: a
  // load inside the tile body would sit behind them in the in-order vmcnt queue, and its wait would expose the whole prefetch (22 -> see profiles).
#define LL_PIN_V asm volatile("" : "+v"(vreg0.x), "+v"(vreg0.y), "+v"(vreg0.z), "+v"(vreg0.w), "+v"(vreg1.x), "+v"(vreg1.y), "+v"(vreg1.z), "+v"(vreg1.w), \
                                  "+v"(vreg2.x), "+v"(vreg2.y), "+v"(vreg2.z), "+v"(vreg2.w), "+v"(vreg3.x), "+v"(vreg3.y), "+v"(vreg3.z), "+v"(vreg3.w));
#define LL_KM_LOAD(T) (p.key_mask ? (unsigned)p.key_mask[(long)b * p.Nk + min((T) * BKV + lane, p.Nk - 1)] : 1u)
  unsigned km_cur = 1u, km_nxt = 1u;
  if constexpr (DB) km_cur = LL_KM_LOAD(0);
  LL_STAGE_LOAD(0)
  LL_STAGE_STORE(0)
  // the Q fragments must have LANDED before the tile loop: hipcc sinks their loads behind tile 0's staging loads, and a fragment still pending at
  // the loop header becomes a vmcnt wait inside EVERY iteration -- which then drains that iteration's prefetch in front of the softmax
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
  asm volatile("" : "+v"(km_cur));
  __syncthreads();

  // One K/V tile.  TT = tile index expression, TC = the same as a LITERAL when the tile loop is unrolled (REL == 3), else 0.
  // Scores are moved to the log2 domain; masking is compiled out for interior tiles (uniform branch).
#define LL_BIAS_INIT(TC)                                                                                            \
  _Pragma("unroll") for (int jb = 0; jb < 2; ++jb)                                                                  \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                  \
    float v = 0.f;                                                                                                  \
    if constexpr (REL == 2) v = bh_tile + bw_cache[jb * 16 + r];                                                    \
    if constexpr (REL == 1) {                                                                                       \
      const uint32_t kk = lut[jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];                                         \
      const int kh = (int)(kk >> 16), kw = (int)(kk & 0xffffu);                                                     \
      v = (relh_row[qh - kh + p.gh - 1] + relw_row[qw - kw + p.gw - 1]) * p.inv_scale;                              \
    }                                                                                                               \
    if constexpr (WIN) {        /* 14x14 window, everything below folds to constants after unrolling */             \
      const int key0 = (TC) * 64 + jb * 32 + (r & 3) + 8 * (r >> 2);   /* this register's key for half 0; half 1: +4 */ \
      const int kh0 = key0 / 14 > 13 ? 13 : key0 / 14, kw0 = key0 % 14;                                             \
      const bool cross = kw0 >= 10 && kh0 < 13;                        /* key0 + 4 falls into the next key row */  \
      v = (cross ? bh_x[kh0 > 12 ? 12 : kh0] : bh_s[kh0]) + bw_y[kw0];                                              \
    }                                                                                                               \
    s[jb][r] = v;                                                                                                   \
  }

#define LL_SCORE_LOOP(MASKED)                                                                                       \
  _Pragma("unroll") for (int jb = 0; jb < 2; ++jb)                                                                  \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                  \
    float v = s[jb][r];                                                                                             \
    if (MASKED) {                                                                                                   \
      const int key = k0 + jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;                                             \
      bool ok = key < p.Nk;                                                                                         \
      if (p.causal) ok = ok && (key <= q);                                                                          \
      /* key-padding mask: bit (key - k0) of the tile's ballot, pre-shifted by this lane half's 4 (a constant bit test) */ \
      ok = ok && (((jb ? km_hi : km_lo) >> ((r & 3) + 8 * (r >> 2))) & 1u);                                         \
      v = ok ? v : NEG;                                                                                             \
      s[jb][r] = v;                                                                                                 \
    }                                                                                                               \
    mx = fmaxf(mx, v);                                                                                              \
  }

#define LL_TILE_BODY(TT, TC)                                                                                        \
  {                                                                                                                 \
    const int k0 = (TT) * BKV;                                                                                      \
    constexpr bool LAST_WIN = WIN && ((TC) == 3);             /* keys 192..195 only */                              \
    float bh_tile = 0.f;                                                                                            \
    if constexpr (REL == 2) {                                                                                       \
      bh_tile = bh_next;                                                                                            \
      if ((TT) + 1 < ntiles) bh_next = relh_row[qh - ((TT) + 1) + p.gh - 1] * p.inv_scale; /* prefetch next row */ \
    }                                                                                                               \
    f32x16_t s[2];                                                                                                  \
    LL_BIAS_INIT(TC)          /* raw-domain bias (or 0) is the C input of the QK^T MFMAs */                         \
    /* k-step outer, key block inner: consecutive MFMAs alternate the two accumulators (a dependent MFMA issues late) */ \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                             \
      _Pragma("unroll") for (int jb = 0; jb < 2; ++jb) {                                                            \
        if (!(LAST_WIN && jb == 1)) {                                                                               \
          const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (jb * 32 + ql) * PK + (2 * ks + half) * 16);  \
          if (ATTN_ABLATE != 5) s[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[jb], 0, 0, 0);        \
          else s[jb][ks] += (float)kf[0];                                                                               \
        }                                                                                                           \
      }                                                                                                             \
    }                                                                                                               \
    float mx = NEG;                                                                                                 \
    uint32_t km_lo = 0xffffffffu, km_hi = 0xffffffffu;                                                              \
    unsigned long long kb_all = ~0ull;                                                                              \
    if (p.key_mask) {       /* ONE byte load per lane per tile (km_cur: fetched a tile ahead, LL_KM_LOAD) + a ballot instead of 32 broadcast byte loads per lane */ \
      kb_all = __ballot(km_cur != 0);                                                                               \
      const unsigned long long kb = kb_all >> (4 * half);                                                           \
      km_lo = (uint32_t)kb; km_hi = (uint32_t)(kb >> 32);                                                           \
    }                                                                                                               \
    /* wave-uniform: the tile needs no masking when it ends inside Nk, no key of it is padded, and (causal) its last key is not after */ \
    /* the wave's FIRST query -- the interior of the causal triangle skips the per-element mask arithmetic */                          \
    const bool need_mask = (k0 + BKV > p.Nk) || kb_all != ~0ull || (p.causal && k0 + BKV - 1 > q0 + wave * 32);     \
    if (need_mask) { LL_SCORE_LOOP(true) } else { LL_SCORE_LOOP(false) }                                            \
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                                                                         \
    const float m_new = fmaxf(m_run, mx);                                                                           \
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);                                     \
    const float nmc = -m_new * p.scale_log2;                                                                        \
    const bool moved = m_new != m_run;                                                                              \
    m_run = m_new;                                                                                                  \
    float lsum = 0.f;                                                                                               \
    _Pragma("unroll") for (int jb = 0; jb < 2; ++jb)                                                                \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                \
      const float pv = ATTN_EXP(fmaf(s[jb][r], p.scale_log2, nmc));                                                 \
      s[jb][r] = pv;                                                                                                \
      lsum += pv;                                                                                                   \
    }                                                                                                               \
    l_run = l_run * alpha + lsum;                                                                                   \
    if (__any(moved)) {        /* wave-uniform: after the first tiles the running max rarely moves */               \
      _Pragma("unroll") for (int d = 0; d < DT; ++d)                                                                \
      _Pragma("unroll") for (int e = 0; e < 16; ++e) o[d][e] *= alpha;                                              \
    }                                                                                                               \
    /* O^T += V^T . P^T : k-steps of 16 keys; P fragment for step ss = accumulator regs 8*(ss&1)..+7 of block ss>>1 */ \
    _Pragma("unroll") for (int ss = 0; ss < (LAST_WIN ? 1 : 4); ++ss) {                                             \
      const int jb = ss >> 1, rb = 8 * (ss & 1);                                                                    \
      const uint4 pu = make_uint4(pack2bf(s[jb][rb + 0], s[jb][rb + 1]), pack2bf(s[jb][rb + 2], s[jb][rb + 3]),     \
                                  pack2bf(s[jb][rb + 4], s[jb][rb + 5]), pack2bf(s[jb][rb + 6], s[jb][rb + 7]));    \
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pu);                                                         \
      _Pragma("unroll") for (int d = 0; d < DT; ++d) {                                                              \
        const char* vrow = Vt + (d * 32 + ql) * PV + (16 * ss + 4 * half) * 2;                                      \
        const uint2 va = *reinterpret_cast<const uint2*>(vrow);                                                     \
        const uint2 vb = *reinterpret_cast<const uint2*>(vrow + 16);                                                \
        const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, make_uint4(va.x, va.y, vb.x, vb.y));                       \
        if (ATTN_ABLATE != 4) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);                \
        else o[d][ss] += (float)vf[0] + (float)pf[0];                                                               \
      }                                                                                                             \
    }                                                                                                               \
  }

  float bh_next = 0.f;
  if constexpr (REL == 2) bh_next = relh_row[qh + p.gh - 1] * p.inv_scale;
  if constexpr (WIN) {               // Nk == 196: exactly 4 tiles, fully unrolled so that every bias index is a constant
    LL_STAGE_LOAD(1) LL_TILE_BODY(0, 0) __syncthreads(); LL_STAGE_STORE(1) __syncthreads();
    LL_STAGE_LOAD(2) LL_TILE_BODY(1, 1) __syncthreads(); LL_STAGE_STORE(2) __syncthreads();
    LL_STAGE_LOAD(3) LL_TILE_BODY(2, 2) __syncthreads(); LL_STAGE_STORE(3) __syncthreads();
    LL_TILE_BODY(3, 3)
  } else if constexpr (DB) {
    for (int t = 0; t < ntiles; ++t) {
#if ATTN_ABLATE == 2 || ATTN_ABLATE == 3
      LL_TILE_BODY(t, 0)
#if ATTN_ABLATE == 2
      __syncthreads();
#endif
#else
      Ks = smem + (t & 1) * TILE_BYTES; Vt = Ks + BKV * PK;
      Ks_w = smem + ((t + 1) & 1) * TILE_BYTES; Vt_w = Ks_w + BKV * PK;
      if (t + 1 < ntiles) { km_nxt = LL_KM_LOAD(t + 1); LL_STAGE_LOAD(t + 1) }
      if (!(p.causal && t * BKV > q0 + wave * 32 + 31)) LL_TILE_BODY(t, 0)      // causal: a tile wholly after this wave's last query contributes exact zeros
      LL_PIN_V        // hipcc otherwise hoists the V transposes (and with them the wait for ALL of the prefetch) in front of the tile's MFMAs
      if (t + 1 < ntiles) LL_STAGE_STORE(t + 1)      // the other buffer: its last readers passed the barrier of iteration t-1
      km_cur = km_nxt;
      __syncthreads();
#endif
    }
  } else {
    for (int t = 0; t < ntiles; ++t) {
      km_cur = LL_KM_LOAD(t);
      if (t + 1 < ntiles) LL_STAGE_LOAD(t + 1)
      LL_TILE_BODY(t, 0)
      __syncthreads();
      if (t + 1 < ntiles) LL_STAGE_STORE(t + 1)
      __syncthreads();
    }
  }

  // ---- normalise and store: lane holds O[q][d..d+3] groups ------------------------------------------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (p.lse && half == 0 && q < p.Nq) p.lse[((long)b * p.heads + h) * p.Nq + q] = fmaf(m_run, p.scale_log2, __builtin_amdgcn_logf(l_tot));
  if (q < p.Nq) {
    bf16_t* orow;
    bool skip = false;
    if (p.o_row_map) {
      const int row = p.o_row_map[(long)b * p.Nq + q];
      skip = row < 0;
      orow = p.O + (long)h * p.osh + (long)(skip ? 0 : row) * p.osr;
    } else {
      orow = p.O + (long)b * p.osb + (long)h * p.osh + (long)q * p.osr;
    }
    if (!skip) {
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = d * 32 + 8 * g + 4 * half;
          if (dd < HD)
            *reinterpret_cast<uint2*>(orow + dd) =
                make_uint2(pack2bf(o[d][4 * g] * inv, o[d][4 * g + 1] * inv), pack2bf(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv));
        }
    }
  }
}

// ---- SAM 14 x 14 window attention (head_dim 80, 196 tokens, decomposed rel-pos from the tables): ONE workgroup per (window, head) ----
// 8 waves x 32 queries = 256 query slots (196 used); all 196 keys (padded to 7 blocks of 32) and V^T stay resident in LDS, so there is
// no tile loop, no online-softmax rescale and one barrier: every lane holds the complete score row of its query (7 x 16 fp32 registers
// per half-lane), takes the exact max / sum, and feeds P to the P.V MFMAs in accumulator order.  K / V are fetched ONCE per workgroup
// (global -> registers, issued first so they fly behind the rel-pos MFMAs; V transposed 4 x 8 in registers on the way into LDS).
// The general kernel above walked 64-key tiles with two barriers each, gave half of its workgroups 68 of 128 query rows and a
// 4-of-64-key last tile: 177 TF/s on this shape.
__global__ __launch_bounds__(512, 2) void attn_win14_kernel(AttnP p) {
  constexpr int HD = 80, KS = HD / 16, DT = 3, NKB = 7, NKP = NKB * 32, NKEY = 196, NT = 512;
  constexpr int CH = HD / 8;                  // 16-byte chunks per K row
  constexpr int PK = (HD + 8) * 2;            // K row pitch (bytes): 11 chunks -> conflict-free ds_read_b128
  constexpr int PV = (NKP + 4) * 2;           // V^T row pitch (bytes) = 8 * 57
  constexpr int G_SLAB = 32 * 33;
  __shared__ __attribute__((aligned(16))) char smem[NKP * PK + DT * 32 * PV + 8 * 2 * G_SLAB * 4];
  char* Ks = smem;
  char* Vt = smem + NKP * PK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, half = lane >> 5;
  const int q = wave * 32 + ql;
  const int qc = min(q, NKEY - 1);
  const int qh = qc / 14, qw = qc - qh * 14;
  // persistent workgroups: item = (window, head), heads fastest.  Workgroup w sits on XCD w % 8; give each XCD runs of 32 consecutive
  // items (two windows x 16 heads: the heads of a token row share cache lines) per sweep of 256.
  const int nitems = p.batch * p.heads;
  const int wg = (int)blockIdx.x, nwg = (int)gridDim.x;
  const int vwg = (nwg % 8 == 0) ? (wg % 8) * (nwg / 8) + wg / 8 : wg;
  int item = vwg;
  if (item >= nitems) return;
  int b = item / p.heads, h = item - b * p.heads;
  const bf16_t* Qg = p.Q + (long)b * p.qsb + (long)h * p.qsh;
  const bf16_t* Kg = p.K + (long)b * p.ksb + (long)h * p.ksh;
  const bf16_t* Vg = p.V + (long)b * p.vsb + (long)h * p.vsh;

  // per-lane offsets are 32-bit element offsets from the (uniform) item base: one VGPR each, SGPR-base addressing
  const int ksr = (int)p.ksr, vsr = (int)p.vsr, qoff = qc * (int)p.qsr + half * 8;
  // ---- K / V global loads into registers (explicit scalars: arrays indexed by the staging loops end up in scratch) ----
  uint4 kr0, kr1, kr2, kr3, kr4, va0, va1, va2, va3, vb0, vb1, vb2, vb3;
  kr0 = kr1 = kr2 = kr3 = kr4 = va0 = va1 = va2 = va3 = vb0 = vb1 = vb2 = vb3 = make_uint4(0, 0, 0, 0);
#define WIN_K_LOAD(IT, REG)                                                                        \
  {                                                                                                \
    const int idx = tid + IT * NT;                                                                 \
    if (idx < NKP * CH) {                                                                          \
      const int key = idx / CH, c = idx - key * CH;                                                \
      REG = *reinterpret_cast<const uint4*>(Kg + (min(key, NKEY - 1) * ksr + c * 8));              \
    }                                                                                              \
  }
#define WIN_K_STORE(IT, REG)                                                                       \
  {                                                                                                \
    const int idx = tid + IT * NT;                                                                 \
    if (idx < NKP * CH) {                                                                          \
      const int key = idx / CH, c = idx - key * CH;                                                \
      *reinterpret_cast<uint4*>(Ks + key * PK + c * 16) = REG;                                     \
    }                                                                                              \
  }
  // V: item = (key quad kq of 56, column chunk c of 10): 4 keys x 8 d
#define WIN_V_LOAD(IT, R0, R1, R2, R3)                                                             \
  {                                                                                                \
    const int idx = tid + IT * NT;                                                                 \
    if (idx < (NKP / 4) * CH) {                                                                    \
      const int kq = idx / CH, c = idx - kq * CH;                                                  \
      R0 = *reinterpret_cast<const uint4*>(Vg + (min(4 * kq + 0, NKEY - 1) * vsr + c * 8));        \
      R1 = *reinterpret_cast<const uint4*>(Vg + (min(4 * kq + 1, NKEY - 1) * vsr + c * 8));        \
      R2 = *reinterpret_cast<const uint4*>(Vg + (min(4 * kq + 2, NKEY - 1) * vsr + c * 8));        \
      R3 = *reinterpret_cast<const uint4*>(Vg + (min(4 * kq + 3, NKEY - 1) * vsr + c * 8));        \
    }                                                                                              \
  }
#define WIN_V_STORE(IT, R0, R1, R2, R3)                                                            \
  {                                                                                                \
    const int idx = tid + IT * NT;                                                                 \
    if (idx < (NKP / 4) * CH) {                                                                    \
      const int kq = idx / CH, c = idx - kq * CH;                                                  \
      char* dst = Vt + (8 * c) * PV + 8 * kq;                                                      \
      *reinterpret_cast<uint2*>(dst + 0 * PV) = make_uint2(perm_lo(R0.x, R1.x), perm_lo(R2.x, R3.x)); \
      *reinterpret_cast<uint2*>(dst + 1 * PV) = make_uint2(perm_hi(R0.x, R1.x), perm_hi(R2.x, R3.x)); \
      *reinterpret_cast<uint2*>(dst + 2 * PV) = make_uint2(perm_lo(R0.y, R1.y), perm_lo(R2.y, R3.y)); \
      *reinterpret_cast<uint2*>(dst + 3 * PV) = make_uint2(perm_hi(R0.y, R1.y), perm_hi(R2.y, R3.y)); \
      *reinterpret_cast<uint2*>(dst + 4 * PV) = make_uint2(perm_lo(R0.z, R1.z), perm_lo(R2.z, R3.z)); \
      *reinterpret_cast<uint2*>(dst + 5 * PV) = make_uint2(perm_hi(R0.z, R1.z), perm_hi(R2.z, R3.z)); \
      *reinterpret_cast<uint2*>(dst + 6 * PV) = make_uint2(perm_lo(R0.w, R1.w), perm_lo(R2.w, R3.w)); \
      *reinterpret_cast<uint2*>(dst + 7 * PV) = make_uint2(perm_hi(R0.w, R1.w), perm_hi(R2.w, R3.w)); \
    }                                                                                              \
  }
  // Q fragments (B operand of S^T = K . Q^T): col = query, k-slots = 8 consecutive d
  bf16x8_t qn[KS];
#define WIN_LOAD_ALL()                                                                             \
  WIN_K_LOAD(0, kr0) WIN_K_LOAD(1, kr1) WIN_K_LOAD(2, kr2) WIN_K_LOAD(3, kr3) WIN_K_LOAD(4, kr4)   \
  WIN_V_LOAD(0, va0, va1, va2, va3) WIN_V_LOAD(1, vb0, vb1, vb2, vb3)                              \
  _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) qn[ks] = *reinterpret_cast<const bf16x8_t*>(Qg + (qoff + ks * 16));
  WIN_LOAD_ALL()

  while (true) {
  // ---- this item's K -> LDS row-major (padded pitch), V -> LDS transposed, Q -> fragments; then the NEXT item's loads are issued and
  //      fly behind this item's MFMAs (one workgroup per CU: without the prefetch HBM idles during compute and the CU during loads) ----
#if WIN_ABLATE == 2
  if (item == vwg) {
#endif
  WIN_K_STORE(0, kr0) WIN_K_STORE(1, kr1) WIN_K_STORE(2, kr2) WIN_K_STORE(3, kr3) WIN_K_STORE(4, kr4)
  WIN_V_STORE(0, va0, va1, va2, va3) WIN_V_STORE(1, vb0, vb1, vb2, vb3)
#if WIN_ABLATE == 2
  }
#endif
  bf16x8_t qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = qn[ks];
  const int cb = b, chd = h;
  __syncthreads();
  item += nwg;
  const bool more = item < nitems;
  if (more) {
    b = item / p.heads; h = item - b * p.heads;
    Qg = p.Q + (long)b * p.qsb + (long)h * p.qsh;
    Kg = p.K + (long)b * p.ksb + (long)h * p.ksh;
    Vg = p.V + (long)b * p.vsb + (long)h * p.vsh;
#if WIN_ABLATE != 2
    WIN_LOAD_ALL()
#endif
  }

  // ---- scores of the whole key range: s[kb][r] = S^T[key = 32 kb + (r&3) + 8 (r>>2) + 4 half][query], rel-pos bias added after (keeps the bias registers off the MFMA phase's peak) ----
  f32x16_t s[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {                               // 7 independent accumulators between dependent MFMAs
      const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (kb * 32 + ql) * PK + (2 * ks + half) * 16);
      if (WIN_ABLATE != 5) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
      else s[kb][ks] += (float)kf[0] + (float)qf[ks][0];
    }
  // ---- decomposed rel-pos (image_encoder.py:354-392): G^T = R . Q^T on the matrix cores, bounced through a per-wave slab so that each
  //      lane can pick the entries at its own (qh - kh + 13) / (qw - kw + 13); the lane-half (key or key + 4) is folded into the LOAD ----
  if (WIN_ABLATE != 3) {
  float bh_s[14], bh_x[13], bw_y[14];
  {
    float* gs = reinterpret_cast<float*>(smem + NKP * PK + DT * 32 * PV) + wave * (2 * G_SLAB);
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      const bf16_t* tab = tb == 0 ? p.rtab_h : p.rtab_w;
      f32x16_t g;
#pragma unroll
      for (int e = 0; e < 16; ++e) g[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t rf = *reinterpret_cast<const bf16x8_t*>(tab + ql * HD + ks * 16 + half * 8);
        g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf, qf[ks], g, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) gs[tb * G_SLAB + ql * 33 + (e & 3) + 8 * (e >> 2) + 4 * half] = g[e];
    }
    const float* gh = gs + ql * 33;
    const float* gw = gs + G_SLAB + ql * 33;
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      bh_s[j] = gh[qh - j + 13] * p.inv_scale;
      bw_y[j] = gw[qw - ((j + 4 * half) % 14) + 13] * p.inv_scale;
    }
#pragma unroll
    for (int j = 0; j < 13; ++j) bh_x[j] = gh[qh - (j + half) + 13] * p.inv_scale;
  }
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key0 = kb * 32 + (r & 3) + 8 * (r >> 2);            // this register's key for half 0; half 1: + 4
      const int kh0 = key0 / 14 > 13 ? 13 : key0 / 14, kw0 = key0 % 14;
      const bool cross = kw0 >= 10 && kh0 < 13;                    // key0 + 4 falls into the next key row
      s[kb][r] += (cross ? bh_x[kh0 > 12 ? 12 : kh0] : bh_s[kh0]) + bw_y[kw0];
    }
  }
  // keys >= 196 live in block 6 only: key = 192 + (r&3) + 8 (r>>2) + 4 half is valid for half 0, r < 4
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (half != 0 || r >= 4) s[NKB - 1][r] = NEG;
  float mx = NEG;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float nmc = -mx * p.scale_log2;
  float lsum = 0.f;
  uint32_t pk[NKB][8];                                               // P in bf16 pairs, accumulator order (halves the live registers of the P.V phase)
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
#if WIN_ABLATE == 1
      const float p0 = fmaf(s[kb][r], p.scale_log2, nmc), p1 = fmaf(s[kb][r + 1], p.scale_log2, nmc);
#else
      const float p0 = __builtin_amdgcn_exp2f(fmaf(s[kb][r], p.scale_log2, nmc));
      const float p1 = __builtin_amdgcn_exp2f(fmaf(s[kb][r + 1], p.scale_log2, nmc));
#endif
      lsum += p0 + p1;
      pk[kb][r >> 1] = pack2bf(p0, p1);
    }

  // ---- O^T = V^T . P^T: k-steps of 16 keys; the P fragment of step (kb, u) = accumulator regs 8u .. 8u+7 of block kb ----
  f32x16_t o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (kb == NKB - 1 && u == 1) continue;                         // keys 208..223: all padding
      const int rb = 8 * u, ss = 2 * kb + u;
      const uint4 pu = make_uint4(pk[kb][rb / 2 + 0], pk[kb][rb / 2 + 1], pk[kb][rb / 2 + 2], pk[kb][rb / 2 + 3]);
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pu);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const char* vrow = Vt + (d * 32 + ql) * PV + (16 * ss + 4 * half) * 2;
        const uint2 va = *reinterpret_cast<const uint2*>(vrow);
        const uint2 vb = *reinterpret_cast<const uint2*>(vrow + 16);
        const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, make_uint4(va.x, va.y, vb.x, vb.y));
        if (WIN_ABLATE != 4) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
        else o[d][ss & 15] += (float)vf[0] + (float)pf[0];
      }
    }

  // ---- normalise and store through the window un-partition map ----
  const float l_tot = lsum + __shfl_xor(lsum, 32, 64);
  const float inv = 1.f / l_tot;
  if (q < NKEY) {
    bf16_t* orow;
    bool skip = false;
    if (p.o_row_map) {
      const int row = p.o_row_map[(long)cb * p.Nq + q];
      skip = row < 0;
      orow = p.O + (long)chd * p.osh + (long)(skip ? 0 : row) * p.osr;
    } else {
      orow = p.O + (long)cb * p.osb + (long)chd * p.osh + (long)q * p.osr;
    }
    if (!skip) {
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = d * 32 + 8 * g + 4 * half;
          if (dd < HD)
            *reinterpret_cast<uint2*>(orow + dd) =
                make_uint2(pack2bf(o[d][4 * g] * inv, o[d][4 * g + 1] * inv), pack2bf(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv));
        }
    }
  }
  if (!more) break;
  __syncthreads();                              // every wave is done with this item's K / V^T before the next item overwrites them
  }
#undef WIN_LOAD_ALL
#undef WIN_K_LOAD
#undef WIN_K_STORE
#undef WIN_V_LOAD
#undef WIN_V_STORE
}

// ---- the same window attention with K / V arriving by LDS-DMA (round 3) -------------------------------------------------------------------
// What bounded attn_win14_kernel (side builds with one section removed, 600 windows x 16 heads, tools/attn_ablate.sh): shipped 487 us; K / V
// staged once 300 us (-38 %); no rel-pos 327 us (-33 %); exp -> identity, no P.V MFMAs, no Q.K^T MFMAs: -0..13 %.  I.e. the register
// staging (13 16-byte loads, 5 + 16 LDS stores and 32 permutes per thread and item) and the rel-pos bounce (64 scattered 4-byte LDS stores +
// 41 loads + 41 multiplies + 448 adds per lane and item), not the matrix pipe or the exponentials.  Here:
//   * K and V rows go global -> LDS by `buffer_load ... lds` (1 KiB per wave-instruction, 4-5 per wave and operand), row-major at their
//     natural 160-byte pitch, into the buffers of the NEXT item while this item computes (two K + two V buffers, one barrier per item);
//     K rows 8..15 of every 16 are stored rotated by one 16-byte chunk (the DMA's per-lane SOURCE address is free) so that the
//     `ds_read_b128` K fragments of a 16-lane service group touch 64 distinct banks; V^T fragments come from the row-major image through
//     two `ds_read_b64_tr_b16` each (profiles/r02_tr_b16_probe.md: no transposes anywhere);
//   * the rel-pos tables' MFMA fragments stay in registers for the whole kernel; G^T = R . Q^T is bounced through ONE per-wave slab (table h,
//     then table w) with 16-byte stores; the bias enters the score accumulators as their INITIAL value (one add per score instead of two
//     after the MFMAs).
// GATHER (round 6): the kernel applies window_partition (+ its zero padding) itself -- Q / K / V are the q|k|v rows of the UNPARTITIONED tokens, a window row outside the
// image reads the projection's bias (what the reference's padded zero token projects to: it pads AFTER norm1, image_encoder.py:178-183), the output goes to the token's own row.
// The q|k|v GEMM of a windowed block then runs on the 4096 real tokens of an image instead of the 4900 rows of its 25 padded windows (-16.4 % of that product, same bits).
template <bool GATHER>
__global__ __launch_bounds__(512, 2) void attn_win14_dma_kernel(AttnP p) {
  constexpr int HD = 80, KS = HD / 16, DT = 3, NKB = 7, NKEY = 196, ROWB = HD * 2;
  constexpr int KBYTES = NKEY * ROWB;                    // 31360: rows >= 196 of the last key block are never real keys (masked below)
  constexpr int VROWS = 208, VBYTES = VROWS * ROWB;      // 33280: 13 k-steps of 16 keys; rows 196..207 hold copies of row 195 (finite, times P = 0)
  constexpr int K_DMA = (KBYTES + 1023) / 1024, V_DMA = (VBYTES + 1023) / 1024;      // 31 / 33 one-KiB pieces
  constexpr int SPITCH = 28, SLAB = 32 * SPITCH;         // per-wave fp32 slab [32 queries][28 table rows]
  __shared__ __attribute__((aligned(16))) char smem[2 * KBYTES + 2 * VBYTES + 8 * SLAB * 4];       // 157,952 B
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, half = lane >> 5;
  const int q = wave * 32 + ql;
  const int qc = min(q, NKEY - 1);
  const int qh = qc / 14, qw = qc - qh * 14;
  const int nitems = p.batch * p.heads;
  const int wg = (int)blockIdx.x, nwg = (int)gridDim.x;
  const int vwg = (nwg % 8 == 0) ? (wg % 8) * (nwg / 8) + wg / 8 : wg;
  int item = vwg;
  if (item >= nitems) return;
  int b = item / p.heads, h = item - b * p.heads;

  // per-lane DMA source offsets (bytes from the item's K / V base): piece j = wave + 8 k fills LDS bytes [1024 j, 1024 j + 1024)
  int kvoff[4], vvoff[5];
  bool kval[4], vval[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int j = wave + 8 * k, L = j * 1024 + lane * 16;
    const int row = L / ROWB, slot = (L - row * ROWB) >> 4;
    const int rc = min(row, NKEY - 1);
    if (k < 4) {
      int c = slot - ((row >> 3) & 1);                   // LDS slot `slot` of a rotated row holds global chunk slot - 1 (mod 10)
      if (c < 0) c += HD / 8;
      kvoff[k] = GATHER ? ((rc / 14) | ((rc % 14) << 8) | (c * 16) << 16) : (rc * (int)p.ksr + c * 8) * 2;       // GATHER: (iy, ix, byte offset in the row), packed
      kval[k] = j < K_DMA && L < KBYTES;
    }
    vvoff[k] = GATHER ? ((rc / 14) | ((rc % 14) << 8) | (slot * 16) << 16) : (rc * (int)p.vsr + slot * 8) * 2;
    vval[k] = j < V_DMA && L < VBYTES;
  }
  // GATHER: source of window row (iy, ix) [packed in pk] of the window whose first token is (y0, x0); img = the operand's first row of that image and head (wave-uniform),
  // pad = the pad row at that head; rs2 = row pitch in bytes.  Branch-free: the offset inside an image fits 32 bits (g^2 rows x pitch), only the two selects depend on `in`.
  auto gsrc = [&](const char* img, const char* pad, int rs2, int pk, int y0, int x0) -> const char* {
    const int ty = y0 + (pk & 255), tx = x0 + ((pk >> 8) & 255);
    const bool in = (ty < p.win_grid) & (tx < p.win_grid);
    const unsigned off = (unsigned)((ty * p.win_grid + tx) * rs2);
    return (in ? img : pad) + ((in ? off : 0u) + (unsigned)(pk >> 16));
  };
  // fragment read offsets: K row ql of a key block, chunk 2 ks + half at its (rotated) slot; V^T through the transposing read
  int koff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) koff[ks] = ql * ROWB + (((2 * ks + half + ((ql >> 3) & 1)) % (HD / 8)) << 4);
  const int voff = (4 * half + ((lane & 15) >> 2)) * ROWB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  // rel-pos table fragments (A operands of G^T = R . Q^T), resident: rows >= 27 of the tables are zero
  bf16x8_t rth[KS], rtw[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    rth[ks] = *reinterpret_cast<const bf16x8_t*>(p.rtab_h + ql * HD + ks * 16 + half * 8);
    rtw[ks] = *reinterpret_cast<const bf16x8_t*>(p.rtab_w + ql * HD + ks * 16 + half * 8);
  }
  const int qoff = qc * (int)p.qsr + half * 8;
  bf16x8_t qn[KS];
  typedef __attribute__((address_space(3))) void* lds_p;
  typedef __attribute__((address_space(3))) short4v* lds_tr;

  // One 1-KiB LDS-DMA piece as an asm statement: hipcc must NOT see it -- a `buffer_load ... lds` builtin is tracked as a store to `smem`, and
  // because the next item's buffers cannot be proven distinct from this item's, every fragment read after it got an `s_waitcnt vmcnt`
  // that drained the DMA before the compute started (measured in the ISA: vmcnt(5) right behind the issue).  M0 is written in the same
  // statement that reads it (cdna_hip_programming.md 5.7); completion is counted by the explicit vmcnt(0) at the top of the item loop.
#define WIND_DMA16(LDS_BYTE_ADDR, GSRC)                                                                             \
  {                                                                                                                 \
    unsigned keep_;                                                                                                 \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"        \
                 : "=&s"(keep_) : "v"(GSRC), "s"(LDS_BYTE_ADDR) : "memory");                                        \
  }
#define WIND_ISSUE(KD, VD)                                                                                          \
  {                                                                                                                 \
    const unsigned kd_ = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_p)(KD)) + wave * 1024;           \
    const unsigned vd_ = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_p)(VD)) + wave * 1024;           \
    if constexpr (GATHER) {                                                                                         \
      const int nw2_ = p.win_nw * p.win_nw, im_ = b / nw2_, w_ = b - im_ * nw2_, wy_ = w_ / p.win_nw;               \
      const int y0_ = wy_ * 14, x0_ = (w_ - wy_ * p.win_nw) * 14;                                                   \
      const long r0_ = (long)im_ * p.win_grid * p.win_grid;                                                         \
      const char* ki_ = reinterpret_cast<const char*>(p.K + r0_ * p.ksr + (long)h * p.ksh);                         \
      const char* vi_ = reinterpret_cast<const char*>(p.V + r0_ * p.vsr + (long)h * p.vsh);                         \
      const char* qi_ = reinterpret_cast<const char*>(p.Q + r0_ * p.qsr + (long)h * p.qsh);                         \
      const char* kp_ = reinterpret_cast<const char*>(p.pad_k + (long)h * p.ksh);                                   \
      const char* vp_ = reinterpret_cast<const char*>(p.pad_v + (long)h * p.vsh);                                   \
      const char* qp_ = reinterpret_cast<const char*>(p.pad_q + (long)h * p.qsh);                                   \
      _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                                 \
        if (kval[k]) WIND_DMA16(kd_ + 8192 * k, gsrc(ki_, kp_, (int)p.ksr * 2, kvoff[k], y0_, x0_))                 \
      _Pragma("unroll") for (int k = 0; k < 5; ++k)                                                                 \
        if (vval[k]) WIND_DMA16(vd_ + 8192 * k, gsrc(vi_, vp_, (int)p.vsr * 2, vvoff[k], y0_, x0_))                 \
      const bf16_t* Qg = reinterpret_cast<const bf16_t*>(gsrc(qi_, qp_, (int)p.qsr * 2, qh | (qw << 8), y0_, x0_)) + half * 8; \
      _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) qn[ks] = *reinterpret_cast<const bf16x8_t*>(Qg + ks * 16);  \
    } else {                                                                                                        \
      const char* Kg = reinterpret_cast<const char*>(p.K + (long)b * p.ksb + (long)h * p.ksh);                      \
      const char* Vg = reinterpret_cast<const char*>(p.V + (long)b * p.vsb + (long)h * p.vsh);                      \
      const bf16_t* Qg = p.Q + (long)b * p.qsb + (long)h * p.qsh;                                                   \
      _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                                 \
        if (kval[k]) WIND_DMA16(kd_ + 8192 * k, Kg + kvoff[k])                                                      \
      _Pragma("unroll") for (int k = 0; k < 5; ++k)                                                                 \
        if (vval[k]) WIND_DMA16(vd_ + 8192 * k, Vg + vvoff[k])                                                      \
      _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) qn[ks] = *reinterpret_cast<const bf16x8_t*>(Qg + (qoff + ks * 16)); \
    }                                                                                                               \
  }

  char* Kc = smem;                                   // buffers of the item being computed
  char* Vc = smem + 2 * KBYTES;
  char* Kn = smem + KBYTES;                          // buffers the next item is DMA'd into
  char* Vn = smem + 2 * KBYTES + VBYTES;
  float* const gs = reinterpret_cast<float*>(smem + 2 * KBYTES + 2 * VBYTES) + wave * SLAB;
  WIND_ISSUE(Kc, Vc)

  while (true) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of the item (and its Q rows) have arrived
    __syncthreads();                                           // ... everyone's have, and everyone is done with the previous item's buffers
    bf16x8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = qn[ks];
    const int cb = b, chd = h;
    item += nwg;
    const bool more = item < nitems;
    if (more) {
      b = item / p.heads; h = item - b * p.heads;
      WIND_ISSUE(Kn, Vn)                                       // flies behind this item's compute
    }

    // ---- decomposed rel-pos: G^T = R . Q^T per table, bounced through the per-wave slab; lane-half folded into the LOAD index ----
    float bh_s[14], bh_x[13];
    f32x2_t bw_y2[7];                                             // key-column bias as register PAIRS (columns 2j, 2j + 1): one v_pk_add_f32 per two accumulator slots
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      f32x16_t g;
#pragma unroll
      for (int e = 0; e < 16; ++e) g[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tb == 0 ? rth[ks] : rtw[ks], qf[ks], g, 0, 0, 0);
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)                              // regs 4 gq .. + 3 = table rows 8 gq + 4 half + 0..3 of query ql
        if (!(gq == 3 && half == 1))
          *reinterpret_cast<float4*>(gs + ql * SPITCH + 8 * gq + 4 * half) = make_float4(g[4 * gq], g[4 * gq + 1], g[4 * gq + 2], g[4 * gq + 3]);
      if (tb == 0) {
        const float* gh = gs + ql * SPITCH + qh;
#pragma unroll
        for (int j = 0; j < 14; ++j) bh_s[j] = gh[13 - j] * p.inv_scale;
#pragma unroll
        for (int j = 0; j < 13; ++j) bh_x[j] = gh[13 - j - half] * p.inv_scale;
      } else {
        const float* gw = gs + ql * SPITCH + qw + 13;
#pragma unroll
        for (int j = 0; j < 7; ++j) bw_y2[j] = f32x2_t{gw[-((2 * j + 4 * half) % 14)], gw[-((2 * j + 1 + 4 * half) % 14)]} * p.inv_scale;
      }
    }

    // ---- scores: accumulators start at the bias, then 35 MFMAs (7 independent accumulators between dependent ones) ----
    f32x16_t s[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {                             // slots r, r + 1 = keys key0, key0 + 1: same key row (key0 and 14 are even)
        const int key0 = kb * 32 + (r & 3) + 8 * (r >> 2);            // this register's key for half 0; half 1: + 4
        const int kh0 = key0 / 14 > 13 ? 13 : key0 / 14, kw0 = key0 % 14;
        const bool cross = kw0 >= 10 && kh0 < 13;                    // key0 + 4 falls into the next key row
        const f32x2_t v = bw_y2[kw0 >> 1] + (cross ? bh_x[kh0 > 12 ? 12 : kh0] : bh_s[kh0]);
        s[kb][r] = v.x; s[kb][r + 1] = v.y;
      }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Kc + kb * (32 * ROWB) + koff[ks]);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
      }
    // keys >= 196 live in block 6 only: key = 192 + (r&3) + 8 (r>>2) + 4 half is valid for half 0, r < 4
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (half != 0 || r >= 4) s[NKB - 1][r] = NEG;
    float mx = NEG;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float nmc = -mx * p.scale_log2;
    f32x2_t lsum2 = {0.f, 0.f};
    const f32x2_t sc2 = {p.scale_log2, p.scale_log2}, nmc2 = {nmc, nmc};
    uint32_t pk[NKB][8];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {                             // packed fp32: one v_pk_fma / v_pk_add per two scores
        const f32x2_t t2 = __builtin_elementwise_fma(f32x2_t{s[kb][r], s[kb][r + 1]}, sc2, nmc2);
        const f32x2_t e2 = {__builtin_amdgcn_exp2f(t2.x), __builtin_amdgcn_exp2f(t2.y)};
        lsum2 += e2;
        pk[kb][r >> 1] = pack2bf(e2.x, e2.y);
      }
    const float lsum = lsum2.x + lsum2.y;

    // ---- O^T = V^T . P^T: k-steps of 16 keys; V^T fragments = two transposing reads of the row-major V image ----
    f32x16_t o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (kb == NKB - 1 && u == 1) continue;                         // keys 208..223: all padding
        const int rb = 8 * u, ss = 2 * kb + u;
        const uint4 pu = make_uint4(pk[kb][rb / 2 + 0], pk[kb][rb / 2 + 1], pk[kb][rb / 2 + 2], pk[kb][rb / 2 + 3]);
        const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pu);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const char* va = Vc + ss * (16 * ROWB) + voff + d * 64;     // keys 16 ss + 4 half + 0..3 | + 8: the P fragment's k-slot order
          const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)va);
          const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(va + 8 * ROWB));
          const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
        }
      }

    // ---- normalise and store through the window un-partition map ----
    const float l_tot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = 1.f / l_tot;
    if (q < NKEY) {
      bf16_t* orow;
      bool skip = false;
      if constexpr (GATHER) {                        // window_unpartition + crop: the query's own token row, padding queries store nothing
        const int nw2 = p.win_nw * p.win_nw, im = cb / nw2, w = cb - im * nw2, wy = w / p.win_nw;
        const int ty = wy * 14 + qh, tx = (w - wy * p.win_nw) * 14 + qw;
        skip = ty >= p.win_grid || tx >= p.win_grid;
        orow = p.O + (long)chd * p.osh + (skip ? 0 : ((long)im * p.win_grid + ty) * p.win_grid + tx) * p.osr;
      } else if (p.o_row_map) {
        const int row = p.o_row_map[(long)cb * p.Nq + q];
        skip = row < 0;
        orow = p.O + (long)chd * p.osh + (long)(skip ? 0 : row) * p.osr;
      } else {
        orow = p.O + (long)cb * p.osb + (long)chd * p.osh + (long)q * p.osr;
      }
      if (!skip) {
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int dd = d * 32 + 8 * g + 4 * half;
            if (dd < HD)
              *reinterpret_cast<uint2*>(orow + dd) =
                  make_uint2(pack2bf(o[d][4 * g] * inv, o[d][4 * g + 1] * inv), pack2bf(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv));
          }
      }
    }
    if (!more) break;
    { char* x = Kc; Kc = Kn; Kn = x; x = Vc; Vc = Vn; Vn = x; }
  }
#undef WIND_ISSUE
#undef WIND_DMA16
}

static int g_attn_win_new = 2;      // SAM 14 x 14 windows: 0 = the general tiled kernel, 1 = resident-window kernel (register staging), 2 = LDS-DMA form
// ---- SAM global attention (4096 tokens on a 64 x 64 grid, head_dim 80, decomposed rel-pos) with K / V tiles arriving by LDS-DMA (round 3) ----
// attn_fwd_kernel<80, 2, 8> spends 23 % of its time staging K / V through registers and 14 % at its per-tile barrier (side builds:
// profiles/r03_attn_experiments.md).  Same arithmetic (8 waves x 32 queries, transposed scores, online softmax), but: a step is TWO key rows of
// the grid (128 keys: half as many barriers and softmax updates per key); its K and V go global -> LDS by `global_load_lds_dwordx4` into the
// OTHER of two 128-key buffers while this step computes (vmcnt(0) + one barrier per step); K rows 8..15 of every 16 are rotated by one
// 16-byte chunk (conflict-free `ds_read_b128`), V^T fragments come from the row-major image through two `ds_read_b64_tr_b16` per MFMA
// operand; the per-query rel-pos row of the key-row axis lives in LDS (a `ds_read_b32` per key row instead of a global load whose
// compiler-counted wait would drain the DMA).
__global__ __launch_bounds__(512, 2) void attn_glob80_dma_kernel(AttnP p) {
  constexpr int HD = 80, KS = HD / 16, DT = 3, ROWB = HD * 2;
  constexpr int KT = 2, KEYS = KT * BKV, TB = KEYS * ROWB;                              // 128 keys, 20480 B per K or V buffer
  constexpr int NPIECE = TB / 1024;                                                     // 20 one-KiB pieces per step and operand
  constexpr int BQ = 256, BHP = 65;                                                     // bias table pitch (floats)
  __shared__ __attribute__((aligned(16))) char smem[4 * TB + BQ * BHP * 4];             // 81,920 + 66,560 B
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, half = lane >> 5;
  int b = blockIdx.z, h = blockIdx.y, qb = blockIdx.x;
  if (p.xcd_nqb > 0) {
    const int id = blockIdx.x, j = id >> 3;
    const int bh = (j / p.xcd_nqb) * 8 + (id & 7);
    if (bh >= p.batch * p.heads) return;
    qb = j % p.xcd_nqb; b = bh / p.heads; h = bh - b * p.heads;
  }
  const int q = qb * BQ + wave * 32 + ql;                   // Nq % 256 == 0: every slot is a real query
  const int nsteps = p.Nk / KEYS;
  const bf16_t* Qg = p.Q + (long)b * p.qsb + (long)h * p.qsh;
  const char* Kg = reinterpret_cast<const char*>(p.K + (long)b * p.ksb + (long)h * p.ksh);
  const char* Vg = reinterpret_cast<const char*>(p.V + (long)b * p.vsb + (long)h * p.vsh);

  bf16x8_t qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(Qg + (long)q * p.qsr + ks * 16 + half * 8);
  const int qh = q / p.gw, qw = q - qh * p.gw;
  const long rrow = (((long)h * p.batch + b) * p.Nq + q) * p.rel_ld;
  // key-column bias of this query, as register PAIRS in accumulator order (pair k of block jb = accumulator slots 2k, 2k + 1): the
  // accumulators start a step as `pair + key-row bias` through one v_pk_add_f32 per pair, with no moves to re-order
  f32x2_t bw2[16];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = 2 * k;
      const long at = rrow + qw - (jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) + p.gw - 1;
      bw2[jb * 8 + k] = f32x2_t{p.rel_w[at] * p.inv_scale, p.rel_w[at - 1] * p.inv_scale};
    }
  // key-row bias of this query for every key row: bh[kh] = rel_h[qh - kh + gh - 1]; half 0 writes the even rows, half 1 the odd ones
  float* const bhs = reinterpret_cast<float*>(smem + 4 * TB) + (wave * 32 + ql) * BHP;
  for (int t = half; t < p.gh; t += 2) bhs[t] = p.rel_h[rrow + qh - t + p.gh - 1] * p.inv_scale;

  // DMA: piece j of a step covers LDS bytes [1024 j, +1024) of the buffer; this wave issues pieces wave, wave + 8 and (waves 0..3) wave + 16
  int kvo[3], vvo[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int L = (wave + 8 * k) * 1024 + lane * 16;
    const int row = L / ROWB, slot = (L - row * ROWB) >> 4;
    int c = slot - ((row >> 3) & 1);
    if (c < 0) c += HD / 8;
    kvo[k] = (row * (int)p.ksr + c * 8) * 2;
    // V: the five 32-byte blocks of an EVEN row sit one block to the right (cyclically): the four rows x two blocks a 32-lane group of
    // ds_read_b64_tr_b16 touches then fall on 64 distinct banks (pitch 160 B: rows 0 and 3 of a group shared 8 banks, 2x the LDS cycles)
    const int vblk = ((slot >> 1) + ((row & 1) ? 0 : 4)) % 5;
    vvo[k] = (row * (int)p.vsr + (vblk * 2 + (slot & 1)) * 8) * 2;
  }
  const bool three = wave < NPIECE - 16;                    // waves 0..3 carry a third piece per operand
  typedef __attribute__((address_space(3))) void* lds_p;
  typedef __attribute__((address_space(3))) short4v* lds_tr;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_p)smem);
#define GLOB_DMA16(LDS_BYTE_ADDR, GSRC)                                                                             \
  {                                                                                                                 \
    unsigned keep_;                                                                                                 \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"        \
                 : "=&s"(keep_) : "v"(GSRC), "s"(LDS_BYTE_ADDR) : "memory");                                        \
  }
#define GLOB_ISSUE(T)                                                                                               \
  {                                                                                                                 \
    const unsigned sl_ = lds0 + (unsigned)(((T) & 1) * TB) + wave * 1024;                                           \
    const long ko_ = (long)(T) * KEYS * p.ksr * 2, vo_ = (long)(T) * KEYS * p.vsr * 2;                              \
    GLOB_DMA16(sl_, Kg + ko_ + kvo[0]) GLOB_DMA16(sl_ + 8192, Kg + ko_ + kvo[1])                                    \
    GLOB_DMA16(sl_ + 2 * TB, Vg + vo_ + vvo[0]) GLOB_DMA16(sl_ + 2 * TB + 8192, Vg + vo_ + vvo[1])                  \
    if (three) { GLOB_DMA16(sl_ + 16384, Kg + ko_ + kvo[2]) GLOB_DMA16(sl_ + 2 * TB + 16384, Vg + vo_ + vvo[2]) }   \
  }
  int koff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) koff[ks] = ql * ROWB + (((2 * ks + half + ((ql >> 3) & 1)) % (HD / 8)) << 4);
  int voff[DT];                                             // per 32-dim block of V: this lane's (row, rotated 32-byte block, 8-byte piece)
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    const int vrow = 4 * half + ((lane & 15) >> 2);
    const int blk = min(2 * d + ((lane >> 4) & 1), HD / 16 - 1);      // dims 80..95 do not exist: those lanes re-read block 4 (a broadcast)
    voff[d] = vrow * ROWB + ((blk + ((vrow & 1) ? 0 : 1)) % 5) * 32 + (lane & 3) * 8;
  }

  f32x16_t o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = NEG, l_run = 0.f;

  GLOB_ISSUE(0)
  for (int t = 0; t < nsteps; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of step t (the only DMA in flight)
    __syncthreads();                                        // everyone's pieces; everyone is done with step t - 1 (= the buffer step t + 1 goes into)
    if (t + 1 < nsteps) GLOB_ISSUE(t + 1)
    const char* Ks = smem + (t & 1) * TB;
    const char* Vs = Ks + 2 * TB;
    f32x16_t s[2 * KT];
#pragma unroll
    for (int kr = 0; kr < KT; ++kr) {
      const float bh_row = bhs[KT * t + kr];
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const f32x2_t v = bw2[jb * 8 + k] + bh_row;
          s[2 * kr + jb][2 * k] = v.x; s[2 * kr + jb][2 * k + 1] = v.y;
        }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int jb = 0; jb < 2 * KT; ++jb) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + jb * (32 * ROWB) + koff[ks]);
        s[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[jb], 0, 0, 0);
      }
    float mx = NEG;
#pragma unroll
    for (int jb = 0; jb < 2 * KT; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[jb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
    const float nmc = -m_new * p.scale_log2;
    const bool moved = m_new != m_run;
    m_run = m_new;
    f32x2_t lsum2 = {0.f, 0.f};
    const f32x2_t sc2 = {p.scale_log2, p.scale_log2}, nmc2 = {nmc, nmc};
#pragma unroll
    for (int jb = 0; jb < 2 * KT; ++jb)
#pragma unroll
      for (int k = 0; k < 8; ++k) {                      // packed fp32: one v_pk_fma / v_pk_add per two scores
        const f32x2_t t2 = __builtin_elementwise_fma(f32x2_t{s[jb][2 * k], s[jb][2 * k + 1]}, sc2, nmc2);
        const f32x2_t e2 = {__builtin_amdgcn_exp2f(t2.x), __builtin_amdgcn_exp2f(t2.y)};
        s[jb][2 * k] = e2.x; s[jb][2 * k + 1] = e2.y;
        lsum2 += e2;
      }
    l_run = l_run * alpha + (lsum2.x + lsum2.y);
    if (__any(moved)) {
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
    }
#pragma unroll
    for (int ss = 0; ss < 4 * KT; ++ss) {
      const int jb = ss >> 1, rb = 8 * (ss & 1);
      const uint4 pu = make_uint4(pack2bf(s[jb][rb + 0], s[jb][rb + 1]), pack2bf(s[jb][rb + 2], s[jb][rb + 3]),
                                  pack2bf(s[jb][rb + 4], s[jb][rb + 5]), pack2bf(s[jb][rb + 6], s[jb][rb + 7]));
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pu);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const char* va = Vs + ss * (16 * ROWB) + voff[d];
        const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)va);
        const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(va + 8 * ROWB));
        const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
      }
    }
  }
#undef GLOB_ISSUE
#undef GLOB_DMA16
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  bf16_t* orow;
  bool skip = false;
  if (p.o_row_map) {
    const int row = p.o_row_map[(long)b * p.Nq + q];
    skip = row < 0;
    orow = p.O + (long)h * p.osh + (long)(skip ? 0 : row) * p.osr;
  } else {
    orow = p.O + (long)b * p.osb + (long)h * p.osh + (long)q * p.osr;
  }
  if (!skip) {
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = d * 32 + 8 * g + 4 * half;
        if (dd < HD)
          *reinterpret_cast<uint2*>(orow + dd) =
              make_uint2(pack2bf(o[d][4 * g] * inv, o[d][4 * g + 1] * inv), pack2bf(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv));
      }
  }
}

static const int g_attn_xcd = getenv("LLMSEG_ATTN_NO_XCD") ? 0 : 1;
static const int g_attn_glob_dma = getenv("LLMSEG_ATTN_NO_GLOB_DMA") ? 0 : 1;      // LDS-DMA form of SAM global attention (A/B switch)
static int g_attn_win_wgs = 256;    // persistent workgroups of the window kernel: one per CU      // tuning knob (tools): 0 = the general tiled kernel on the window shape

template <int HD>
int launch_hd(const AttnP& p, hipStream_t s) {
  const bool wide = p.Nq >= 1024 && !p.causal;                       // long non-causal sequences: 256 queries per workgroup
  const int bq = wide ? 256 : 128;
  dim3 grid((p.Nq + bq - 1) / bq, p.heads, p.batch);
  AttnP px = p;                                                        // wide kernels: 1-D XCD-grouped grid (g_attn_xcd = 0: the plain 3-D grid, A/B)
  px.xcd_nqb = g_attn_xcd ? (int)grid.x : 0;
  const dim3 xgrid = g_attn_xcd ? dim3((unsigned)(8 * ((p.batch * p.heads + 7) / 8) * (int)grid.x)) : grid;
  constexpr int NT4 = 256, NT8 = 512;
  if (p.rtab_h != nullptr && HD == 80 && g_attn_win_new == 2 && p.lse == nullptr && p.ksr == p.vsr) {
    // persistent workgroups walk (window, head) items: with 800 items (2 images) 256 workgroups need 4 rounds with the last one 12 % full; the same 4 rounds on 200
    // workgroups finish at the same time and leave 56 CUs to the other stream (CLIP -> Llama run beside the frozen SAM encoder) for the whole launch
    static const bool balance = getenv("LLMSEG_WIN_NO_BALANCE") == nullptr;      // A/B switch
    const int items = p.batch * p.heads;
    int wgs = std::min(items, g_attn_win_wgs);
    if (balance && wgs > 0) {
      const int rounds = (items + wgs - 1) / wgs;
      wgs = std::min(wgs, ((items + rounds - 1) / rounds + 7) & ~7);       // a multiple of 8 keeps the kernel's XCD grouping of the heads of a window
    }
    if (p.win_grid > 0) LL_LAUNCH_KERNEL(attn_win14_dma_kernel<true>, dim3((unsigned)wgs), dim3(NT8), 0, s, p);
    else LL_LAUNCH_KERNEL(attn_win14_dma_kernel<false>, dim3((unsigned)wgs), dim3(NT8), 0, s, p);
  }
  else if (p.rtab_h != nullptr && HD == 80 && g_attn_win_new && p.lse == nullptr) LL_LAUNCH_KERNEL(attn_win14_kernel, dim3((unsigned)std::min(p.batch * p.heads, g_attn_win_wgs)), dim3(NT8), 0, s, p);
  else if (p.rtab_h != nullptr) LL_LAUNCH_KERNEL((attn_fwd_kernel<HD == 80 ? 80 : HD, HD == 80 ? 4 : 0>), dim3((p.Nq + 127) / 128, p.heads, p.batch), dim3(NT4), 0, s, p);
  else if (p.rel_h == nullptr) {
    if (wide) LL_LAUNCH_KERNEL((attn_fwd_kernel<HD, 0, 8>), xgrid, dim3(NT8), 0, s, px);
    else LL_LAUNCH_KERNEL((attn_fwd_kernel<HD, 0>), grid, dim3(NT4), 0, s, p);
  } else if (HD == 80 && p.gh == 14 && p.gw == 14 && p.Nk == 196 && !p.causal && !p.key_mask)
    LL_LAUNCH_KERNEL((attn_fwd_kernel<HD == 80 ? 80 : HD, HD == 80 ? 3 : 1>), dim3((p.Nq + 127) / 128, p.heads, p.batch), dim3(NT4), 0, s, p);
  else if (p.gw == BKV && (p.Nk % BKV) == 0) {
    if (HD == 80 && wide && g_attn_glob_dma && (p.Nq % 256) == 0 && p.Nk == p.gh * BKV && (p.gh & 1) == 0 && !p.causal && !p.key_mask && !p.lse && !p.nk_dev)
      LL_LAUNCH_KERNEL(attn_glob80_dma_kernel, xgrid, dim3(NT8), 0, s, px);
    else if (wide) LL_LAUNCH_KERNEL((attn_fwd_kernel<HD, 2, 8>), xgrid, dim3(NT8), 0, s, px);
    else LL_LAUNCH_KERNEL((attn_fwd_kernel<HD, 2>), grid, dim3(NT4), 0, s, p);
  } else LL_LAUNCH_KERNEL((attn_fwd_kernel<HD, 1>), dim3((p.Nq + 127) / 128, p.heads, p.batch), dim3(NT4), 0, s, p);
  return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// Decode step (one new token per sequence): RoPE of q and k at the device-side position, k / v appended to the cache, and
// softmax(q K^T * scale) V over the pos + 1 cached keys, in one launch (+ a merge when the keys of a head are split over workgroups).
// A CU takes only ~16 B/clk of missing lines, so the ~170 KB of K and V rows of one head (330 keys) are spread over `splits` workgroups
// until the launch covers the chip.  16 lanes own one key (8 dims each, one 16-byte load per K and V row); every 16-lane group runs an
// online softmax over its keys with all of a trip's K and V rows requested together (one HBM round trip per 64 keys per workgroup).
struct DecP {
  const bf16_t* qkv; long ld;
  const float *cs, *sn;
  bf16_t *kc, *vc; long cstride;
  const int32_t* pos_dev;
  int heads, splits;
  float scale;
  bf16_t* out; long ldo;
  float* part;                                         // [N][heads][splits][130]: 128 unnormalised outputs, running max, running sum
};

__global__ __launch_bounds__(256) void decode_attn_kernel(DecP p) {
  constexpr int HD = 128, UN = 4;
  __shared__ float qs[HD];
  __shared__ __align__(16) bf16_t knew[HD], vnew[HD];
  __shared__ float red[4][HD], rm[4], rl[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = tid >> 4, c = tid & 15;
  const int h = blockIdx.x / p.splits, sp = blockIdx.x - h * p.splits, n = blockIdx.y;
  const long D = (long)p.heads * HD;
  const int pos = *p.pos_dev, nk = pos + 1;
  const int per = ((nk + p.splits - 1) / p.splits + 15) & ~15;
  const int j0 = sp * per, j1 = min(nk, j0 + per);
  const bool owner = pos >= j0 && pos < j1;              // the workgroup whose key range holds the new token appends it
  const bf16_t* row = p.qkv + (long)n * p.ld + (long)h * HD;
  bf16_t* kb = p.kc + (long)n * p.cstride + (long)h * HD;
  bf16_t* vb = p.vc + (long)n * p.cstride + (long)h * HD;
  if (tid < 128) {
    const int i = tid & 63;
    if (tid < 64 || owner) {
      const bf16_t* src = row + (tid < 64 ? 0 : D);
      const float a = bf2f(src[i]), b = bf2f(src[i + 64]);
      const float cv = p.cs[(long)pos * 64 + i], sv = p.sn[(long)pos * 64 + i];
      const bf16_t o1 = f2bf(rope_lo(a, b, cv, sv)), o2 = f2bf(rope_hi(a, b, cv, sv));
      if (tid < 64) { qs[i] = bf2f(o1); qs[i + 64] = bf2f(o2); }
      else {
        knew[i] = o1; knew[i + 64] = o2;
        kb[(long)pos * D + i] = o1; kb[(long)pos * D + i + 64] = o2;
      }
    }
  } else if (tid < 192 && owner) {
    const int i = tid - 128;
    const bf16_t v1 = row[2 * D + i], v2 = row[2 * D + i + 64];
    vnew[i] = v1; vnew[i + 64] = v2;
    vb[(long)pos * D + i] = v1; vb[(long)pos * D + i + 64] = v2;
  }
  __syncthreads();
  float q[8], acc[8], m = -1e30f, l = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { q[e] = qs[c * 8 + e]; acc[e] = 0.f; }
  for (int jb = j0; jb < j1; jb += 16 * UN) {
    uint4 kk[UN], vv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = jb + u * 16 + g;
      kk[u] = make_uint4(0, 0, 0, 0); vv[u] = kk[u];
      if (j < j1) {
        if (j == pos) { kk[u] = *reinterpret_cast<const uint4*>(knew + c * 8); vv[u] = *reinterpret_cast<const uint4*>(vnew + c * 8); }
        else { kk[u] = *reinterpret_cast<const uint4*>(kb + (long)j * D + c * 8); vv[u] = *reinterpret_cast<const uint4*>(vb + (long)j * D + c * 8); }
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = jb + u * 16 + g;
      float kf[8], vf[8], s = 0.f;
      unpack8(kk[u], kf);
      unpack8(vv[u], vf);
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(q[e], kf[e], s);
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
      const bool valid = j < j1;
      s = valid ? s * p.scale : -1e30f;
      const float mn = fmaxf(m, s), corr = __expf(m - mn), pj = valid ? __expf(s - mn) : 0.f;
      l = l * corr + pj;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = acc[e] * corr + pj * vf[e];
      m = mn;
    }
  }
  // the four key groups of a wave, then the four waves
  float mw = fmaxf(m, __shfl_xor(m, 16));
  mw = fmaxf(mw, __shfl_xor(mw, 32));
  const float f = __expf(m - mw);
  l *= f; l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
#pragma unroll
  for (int e = 0; e < 8; ++e) { float a = acc[e] * f; a += __shfl_xor(a, 16); a += __shfl_xor(a, 32); acc[e] = a; }
  if (lane < 16) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][c * 8 + e] = acc[e];
    if (lane == 0) { rm[wave] = mw; rl[wave] = l; }
  }
  __syncthreads();
  if (tid < HD) {
    const float M = fmaxf(fmaxf(rm[0], rm[1]), fmaxf(rm[2], rm[3]));
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const float e = __expf(rm[w] - M); L += e * rl[w]; o += e * red[w][tid]; }
    if (p.splits == 1) p.out[(long)n * p.ldo + (long)h * HD + tid] = f2bf(o / L);
    else {
      float* pp = p.part + (((long)n * p.heads + h) * p.splits + sp) * 130;
      pp[tid] = o;
      if (tid == 0) { pp[128] = M; pp[129] = L; }
    }
  }
}

__global__ __launch_bounds__(128) void decode_attn_merge_kernel(const float* __restrict__ part, int heads, int splits, bf16_t* __restrict__ out, long ldo) {
  const int h = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const float* pp = part + ((long)n * heads + h) * splits * 130;
  float mv[16], lv[16], ov[16], M = -1e30f;            // every partial requested before the first use: one memory round trip
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int sc = min(s, splits - 1);
    mv[s] = pp[sc * 130 + 128]; lv[s] = pp[sc * 130 + 129]; ov[s] = pp[sc * 130 + tid];
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) M = fmaxf(M, mv[s]);
  float L = 0.f, o = 0.f;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float e = s < splits ? __expf(mv[s] - M) : 0.f;
    L += e * lv[s]; o += e * ov[s];
  }
  out[(long)n * ldo + (long)h * 128 + tid] = f2bf(o / L);
}

extern "C" int llmseg_decode_attn(const void* qkv, int64_t ld, const float* cos, const float* sin, void* kcache, void* vcache, int64_t cache_stride_n,
                                  const int32_t* pos_dev, int64_t N, int32_t heads, int32_t head_dim, float scale, void* out, int64_t ldo,
                                  void* scratch, int64_t scratch_bytes, void* stream) {
  LL_CHECK(qkv && cos && sin && kcache && vcache && pos_dev && out && N > 0 && N < 65536 && heads > 0, "decode_attn: bad arguments");
  LL_CHECK(head_dim == 128, "decode_attn: head_dim 128 only (use llmseg_rope_kv_append + llmseg_attn_fwd otherwise)");
  LL_CHECK((ld & 7) == 0 && (cache_stride_n & 7) == 0 && AL16(qkv) && AL16(kcache) && AL16(vcache), "decode_attn: 16-byte alignment required");
  int splits = 1;
  if (scratch) {
    splits = (int)std::min<int64_t>(16, std::max<int64_t>(1, 256 / (N * heads)));
    while (splits > 1 && (int64_t)N * heads * splits * 130 * 4 > scratch_bytes) --splits;
  }
  DecP p{(const bf16_t*)qkv, (long)ld, cos, sin, (bf16_t*)kcache, (bf16_t*)vcache, (long)cache_stride_n, pos_dev, heads, splits, scale,
         (bf16_t*)out, (long)ldo, (float*)scratch};
  LL_LAUNCH_KERNEL(decode_attn_kernel, dim3(heads * splits, (unsigned)N), dim3(256), 0, (hipStream_t)stream, p);
  LL_LAUNCH_CHECK("decode_attn");
  if (splits > 1) {
    LL_LAUNCH_KERNEL(decode_attn_merge_kernel, dim3(heads, (unsigned)N), dim3(128), 0, (hipStream_t)stream, (const float*)scratch, heads, splits,
                       (bf16_t*)out, (long)ldo);
    LL_LAUNCH_CHECK("decode_attn_merge");
  }
  return LLMSEG_OK;
}

extern "C" int llmseg_attn_set_variant(int v) {
  g_attn_win_new = v & 3;
  g_attn_win_wgs = (v >> 4) > 0 ? (v >> 4) : 256;
  return LLMSEG_OK;
}

extern "C" int llmseg_attn_fwd(const llmseg_attn_args* a, void* stream) {
  LL_CHECK(a && a->struct_size == sizeof(*a), "%s: ABI mismatch: caller's struct_size %u != %zu (bind against include/llmseg_hip.h version %d)",
           "attn", a ? a->struct_size : 0u, sizeof(*a), LLMSEG_ABI_VERSION);
  LL_CHECK(a && a->Q && a->K && a->V && a->O, "attn: null pointer");
  LL_CHECK(a->batch > 0 && a->heads > 0 && a->Nq > 0 && a->Nk > 0, "attn: bad sizes");
  LL_CHECK(a->head_dim == 32 || a->head_dim == 64 || a->head_dim == 80 || a->head_dim == 128, "attn: head_dim %d unsupported", a->head_dim);
  LL_CHECK(((a->q_stride_row | a->k_stride_row | a->v_stride_row | a->q_stride_h | a->k_stride_h | a->v_stride_h |
             a->q_stride_b | a->k_stride_b | a->v_stride_b) & 7) == 0, "attn: Q/K/V strides must be multiples of 8 elements");
  LL_CHECK(((a->o_stride_row | a->o_stride_h | a->o_stride_b) & 3) == 0, "attn: O strides must be multiples of 4 elements");
  LL_CHECK((((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V) & 15) == 0 && (((uintptr_t)a->O) & 7) == 0, "attn: misaligned pointer");
  if (a->rel_tab_h || a->rel_tab_w) {
    LL_CHECK(a->rel_tab_h && a->rel_tab_w && !a->rel_h && !a->rel_w && a->head_dim == 80 && a->grid_h == 14 && a->grid_w == 14 && a->Nk == 196 &&
             a->Nq == 196 && !a->causal && !a->key_mask, "attn: fused rel-pos tables are supported for SAM's 14x14 windows (head_dim 80) only");
    LL_CHECK((((uintptr_t)a->rel_tab_h | (uintptr_t)a->rel_tab_w) & 15) == 0, "attn: rel tables must be 16-byte aligned");
  }
  if (a->rel_h || a->rel_w) {
    LL_CHECK(a->rel_h && a->rel_w && a->grid_h > 0 && a->grid_w > 0 && a->grid_h * a->grid_w == a->Nk && a->Nq == a->Nk &&
             a->rel_ld >= 2 * (a->grid_h > a->grid_w ? a->grid_h : a->grid_w) - 1, "attn: bad relative-position arguments");
    LL_CHECK(a->grid_h < 65536 && a->grid_w < 65536, "attn: grid too large");
  }
  AttnP p;
  p.Q = (const bf16_t*)a->Q; p.K = (const bf16_t*)a->K; p.V = (const bf16_t*)a->V; p.O = (bf16_t*)a->O;
  p.qsb = a->q_stride_b; p.qsh = a->q_stride_h; p.qsr = a->q_stride_row;
  p.ksb = a->k_stride_b; p.ksh = a->k_stride_h; p.ksr = a->k_stride_row;
  p.vsb = a->v_stride_b; p.vsh = a->v_stride_h; p.vsr = a->v_stride_row;
  p.osb = a->o_stride_b; p.osh = a->o_stride_h; p.osr = a->o_stride_row;
  p.batch = a->batch; p.heads = a->heads; p.Nq = a->Nq; p.Nk = a->Nk;
  p.scale_log2 = a->scale * LOG2E;
  p.inv_scale = 1.0f / a->scale;
  p.causal = a->causal; p.key_mask = a->key_mask;
  p.rel_h = a->rel_h; p.rel_w = a->rel_w; p.rel_ld = a->rel_ld; p.gh = a->grid_h; p.gw = a->grid_w;
  p.o_row_map = a->o_row_map;
  p.lse = a->lse;
  p.rtab_h = (const bf16_t*)a->rel_tab_h; p.rtab_w = (const bf16_t*)a->rel_tab_w;
  p.nk_dev = a->nk_dev;
  p.xcd_nqb = 0;
  p.win_grid = a->win_grid; p.win_nw = a->win_nw;
  p.pad_q = (const bf16_t*)a->pad_q; p.pad_k = (const bf16_t*)a->pad_k; p.pad_v = (const bf16_t*)a->pad_v;
  if (a->win_grid != 0) {
    LL_CHECK(a->rel_tab_h && g_attn_win_new == 2 && !a->lse && a->k_stride_row == a->v_stride_row, "attn: the window-gather form needs the LDS-DMA window kernel (rel_tab_*, no lse)");
    LL_CHECK(a->win_grid > 0 && a->win_nw == (a->win_grid + 13) / 14 && a->win_nw < 19 && a->batch % (a->win_nw * a->win_nw) == 0, "attn: bad window-gather grid");
    LL_CHECK(a->pad_q && a->pad_k && a->pad_v && (((uintptr_t)a->pad_q | (uintptr_t)a->pad_k | (uintptr_t)a->pad_v) & 15) == 0, "attn: window-gather pad rows missing / misaligned");
  }
  LL_CHECK(!a->nk_dev || (!a->rel_tab_h && !a->rel_h && !a->causal), "attn: nk_dev is for plain (decode-step) attention");
  hipStream_t s = (hipStream_t)stream;
  switch (a->head_dim) {
    case 32: launch_hd<32>(p, s); break;
    case 64: launch_hd<64>(p, s); break;
    case 80: launch_hd<80>(p, s); break;
    default: launch_hd<128>(p, s); break;
  }
  LL_LAUNCH_CHECK("attn_fwd");
  return LLMSEG_OK;
}
