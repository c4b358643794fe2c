/* libllmseg_hip.so -- C ABI of the MI355X (gfx950) kernels behind LLM-Seg's `model_forward` hot path.
 *
 * The reference (wangjunchi/LLMSeg) has no FFI boundary: its path is PyTorch eager ops
 * (SURVEY.md §8b).  Each entry point below replaces the eager op sequence cited next to it
 * (paths relative to the reference tree).  All pointers are DEVICE pointers owned by the caller
 * (PyTorch's caching allocator); kernels never allocate, never synchronise, and launch on the
 * `stream` argument (a hipStream_t passed as void*).  bf16 tensors are raw uint16 storage.
 * Every function returns 0 on success or a negative LLMSEG_E* code; llmseg_last_error() gives text.
 * Rows of every bf16 matrix must be 16-byte aligned (pointer and leading dimension % 8 == 0)
 * unless stated otherwise.
 */
#ifndef LLMSEG_HIP_H
#define LLMSEG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LLMSEG_OK 0
#define LLMSEG_EINVAL (-1)
#define LLMSEG_ELAUNCH (-2)

/* epilogue activation */
enum { LLMSEG_ACT_NONE = 0, LLMSEG_ACT_RELU = 1, LLMSEG_ACT_GELU = 2, LLMSEG_ACT_QUICKGELU = 3, LLMSEG_ACT_SILU = 4,
       LLMSEG_ACT_SIGMOID = 5 };

/* ABI guard.  Every argument struct starts with `struct_size`: the caller writes sizeof() of the struct AS ITS OWN BINDING DECLARES IT;
 * an entry point whose struct does not have exactly that size returns LLMSEG_EINVAL ("ABI mismatch") before reading any other field,
 * so a binding written against an older header (fields were appended in every round) fails loudly instead of having the library read
 * past the caller's struct.  llmseg_struct_size(which) returns the library's sizeof (0 = llmseg_gemm_args, 1 = llmseg_attn_args,
 * 2 = llmseg_attn_bwd_args, 3 = llmseg_dropout; -1 for an unknown index) so a binding can assert at load time;
 * llmseg_version() is bumped whenever a struct or a signature changes (9: llmseg_attn_args.win_grid / win_nw / pad_q / pad_k / pad_v; 7: the fp32-activation head entry points; 4: the reduction entry points take a workspace; 5 = this header:
 * llmseg_dropout.seg_rows; 6: llmseg_gemm_args.norm_w / norm_eps / norm_out / ldn). */
#define LLMSEG_ABI_VERSION 9

/* Determinism (round 4).  No kernel adds floating-point numbers with atomics: every sum whose terms come from several workgroups is
 * written as per-workgroup partials into CALLER-OWNED scratch (`workspace`, `workspace_bytes`; any device memory, 256-byte aligned, not
 * shared with a concurrently running stream) and folded in a fixed order by a second launch, so the same inputs give the same bits on
 * every run (the reference's resume contract, training.py:404-421,460-477).  LLMSEG_REDUCE_WS_BYTES is enough for every call the model
 * makes; entry points that can fall back to a single partial (llmseg_colsum, llmseg_norm_bwd* for rows <= 8 KiB, llmseg_lora_outer)
 * accept NULL / a smaller buffer and stay deterministic (slower), the others return LLMSEG_EINVAL when it is too small. */
#define LLMSEG_REDUCE_WS_BYTES (16 << 20)
int llmseg_version(void);
/* kernel launches issued by this library since it was loaded (all streams; a captured hipGraph counts at capture, not at replay):
 * a benchmark reads the difference across one eager micro-step to report launches per step */
int64_t llmseg_launch_count(void);
int64_t llmseg_struct_size(int which);
const char* llmseg_last_error(void);

/* ---- GEMM ------------------------------------------------------------------------------------
 * C[b][m][n] = epi( alpha * sum_k A[b][m][k] * W[b][n][k] ),  A,W bf16 (K contiguous), fp32 accumulate on MFMA.
 * epi(v) = residual[m][n] + gamma[n] * act(v + bias[n])   (each term optional, NULL = absent).
 * Replaces every nn.Linear / 1x1 conv / patch-embed conv on the path: HF LlamaAttention/LlamaMLP projections and
 * lm_head (model/llava/model/language_model/llava_llama.py:93-105), SAM qkv/proj/mlp
 * (model/segment_anything/modeling/image_encoder.py:238-258, common.py:25-26), neck convs (:92-108),
 * mm_projector (model/llava/model/llava_arch.py:93-96), text_hidden_fcs / lisa_* heads (model/LISA.py:54-121).
 * out_f32 != 0 writes fp32 C (ldc in elements of the output type).  batch/strides (in elements) give a strided-batched GEMM.
 */
typedef struct {
  uint32_t struct_size;  /* = sizeof(llmseg_gemm_args) of the caller's declaration (ABI guard, see above) */
  uint32_t reserved0;    /* 0 */
  const void* A; const void* W; void* C;
  const void* bias;      /* bf16 [N] or NULL */
  const void* gamma;     /* bf16 [N] or NULL (DINOv2 LayerScale) */
  const void* residual;  /* bf16 [M][ldr] or NULL */
  int64_t M, N, K;
  int64_t lda, ldw, ldc, ldr;
  int64_t batch, strideA, strideW, strideC;
  float alpha;
  int act;
  int out_f32;
  int trans_a;   /* 1: A is stored [K][M] (M contiguous, lda >= M) -- used by the backward pass (dW = dY^T X) */
  int trans_w;   /* 1: W is stored [K][N] (N contiguous, ldw >= N) -- used by the backward pass (dX = dY W) */
  int64_t batch2, strideA2, strideW2, strideC2;   /* optional outer batch dimension (0/1 = none): entry (b1, b2) is at b1*stride + b2*stride2 */
  /* optional 64-wide extension of the contraction: C = epi(alpha * (A.W^T + A2.W2^T)), A2 bf16 [M][64] (lda2), W2 bf16 [N][64]
   * (ldw2); NULL = none.  Fuses low-rank updates (the r = 8 LoRA products of q_proj / v_proj, zero-padded to 64 columns) into the
   * big GEMM as one more K-tile instead of a read-modify-write pass over C.  batch == 1, no trans_*. */
  const void* A2; const void* W2; int64_t lda2, ldw2;
  /* optional caller-owned scratch (fp32 partial tiles of the split-K path: short matrices with a long contraction, e.g. the
   * Llama o / down projections and every dX GEMM at M = 2 x 319 rows, run as S K-slices + one reduce launch); NULL / too small =
   * no split-K.  Must not be shared between streams that run concurrently. */
  void* workspace; int64_t workspace_bytes;
  /* out_f32 only: C += result instead of C = result (weight gradients accumulate over micro-steps in an fp32 arena, as the
   * reference's DeepSpeed engine accumulates them: training.py:79-82 gradient_accumulation_steps, :292-332 bf16 config) */
  int accumulate;
  /* decode-step fusions, M <= 8 only (one token per sequence: every kernel launch saved is ~8 us of a ~100 us layer):
   *   a_norm_w (bf16 [K]) : A := RMSNorm(A) * a_norm_w on load (HF LlamaRMSNorm arithmetic, eps = a_norm_eps);
   *   a_swiglu            : A rows are [gate | up] of width 2K (lda >= 2K), A := silu(gate) * up on load (HF LlamaMLP).
   * Same bits as llmseg_norm / llmseg_swiglu followed by the GEMM. */
  const void* a_norm_w; float a_norm_eps; int a_swiglu;
  /* optional second output (ABI 6): norm_out[m][:] = RMSNorm(C[m][:]) * norm_w, bf16 [M][ldn], with exactly the arithmetic of llmseg_norm(rms = 1) applied to
   * the bf16 C this call writes (HF LlamaRMSNorm: fp32 statistics, rounding before the weight multiply) -- the residual stream's next pre-norm
   * (llava_llama.py:93-102: o_proj / down_proj + residual, then the following LlamaRMSNorm).  bf16 output, batch 1, N % 8 == 0, ldn % 8 == 0.  When the product runs
   * as K-slices (Llama o / down at M = 2 x 319) the reduce launch computes it on the row it has just summed (one launch and one pass over the row instead of
   * two); otherwise the library runs llmseg_norm behind the GEMM.  Same bits either way.  NULL = off. */
  const void* norm_w; float norm_eps; int reserved1; void* norm_out; int64_t ldn;
  /* fused Llama-layer epilogues (ABI 8): the pointwise launch that used to follow the product runs inside the GEMM's store.  Plain bf16 product only
   * (batch 1, alpha 1, no bias / activation / gamma / residual / norm_out / trans_*).  Same bits as the product followed by that launch: the library runs
   * exactly that two-launch route on every shape its fused kernel (the 128 x 256 tile in one K-slice: the Llama layer at 2 x 319 rows) does not take.
   *   LLMSEG_FX_ROPE        C = q|k|v [M][N]: the heads (width 128) of columns < fx_cols are rotated (rotate-half; fx_cos / fx_sin fp32 [fx_T][64], position
   *                         = row % fx_T) as llmseg_rope(C, ...) would after the product (HF LlamaAttention: apply_rotary_pos_emb on q and k,
   *                         call site llava_llama.py:93-102);
   *   LLMSEG_FX_SWIGLU      N = 2 I, C = gate|up [M][2 I] as without fx AND fx_out[m][c] = silu(gate[m][c]) * up[m][c], bf16 [M][fx_ld] (HF LlamaMLP), as
   *                         llmseg_swiglu would after the product;
   *   LLMSEG_FX_SWIGLU_BWD  N = I: the product is d(silu(gate) * up) [M][I] (dX of down_proj); fx_in = the saved gate|up [M][fx_ld >= 2 I]; C = d(gate|up)
   *                         bf16 [M][ldc >= 2 I], as llmseg_swiglu_bwd would compute from the stored product. */
  int fx; int fx_T; int64_t fx_cols; const float* fx_cos; const float* fx_sin; void* fx_out; const void* fx_in; int64_t fx_ld;
  /* norm-backward tail (ABI 8): the product is the gradient dy [M][N] of a pre-norm's OUTPUT (dX of q|k|v or of gate|up: HF LlamaDecoderLayer,
   * hidden = residual + sublayer(norm(hidden))), and what the caller wants is the gradient of the norm's INPUT:
   *   C = norm_backward(dy, nb_x, nb_w; nb_eps, nb_rms) + nb_dres          (llmseg_norm_bwd_add: frozen norm weight, nb_dres = the residual branch's gradient or NULL)
   * with, optionally, the LoRA branches' dX added to dy first (llmseg_lora_apply with w_rn = 1: dy += nb_lora_alpha * mask_b * (nb_lora_t[:, 8 b ..] . nb_lora_w_b),
   * nb_lora_t bf16 [M][nb_lora_ldt >= 16], nb_lora_w0 / nb_lora_w1 bf16 [8][N] (w1 NULL = one branch), nb_lora_drop = llmseg_dropout* or NULL).  A K-sliced product
   * (the Llama dX products at 2 x 319 rows) does all of it in its reduce launch -- one pass over the gradient instead of three launches; otherwise the library
   * runs the product into the tail of `workspace` (>= M N 2 bytes, required) followed by llmseg_lora_apply / llmseg_norm_bwd_add.  Same bits either way.
   * Plain bf16 product, ldc == N, N % 8 == 0.  nb_x NULL = off. */
  const void* nb_x; const void* nb_w; const void* nb_dres; float nb_eps; int nb_rms;
  const void* nb_lora_t; int64_t nb_lora_ldt; const void* nb_lora_w0; const void* nb_lora_w1; float nb_lora_alpha; int reserved2; const void* nb_lora_drop;
  /* delta tail (ABI 8): the product is dO [batch * dl_T][dl_heads * 128] of an attention (dX of o_proj; rows = b * dl_T + q) whose output was dl_o (bf16, row pitch
   * dl_ldo); the call also writes dl_out[b][h][q] = sum_d dO * O (fp32 [batch][dl_heads][dl_T]) -- llmseg_attn_bwd's `delta`, which can then be passed with
   * delta_ready = 1.  In the reduce launch of a K-sliced product, by the attention backward's delta kernel behind the product otherwise; same bits.  NULL = off. */
  const void* dl_o; int64_t dl_ldo; float* dl_out; int32_t dl_heads; int32_t dl_T;
  /* with nb_lora_t (ABI 8): the LoRA operand still as the K-slice partials llmseg_lora_down_parts left (fp32 [nb_lora_S][M][16]); the tail adds them in order, scales by
   * nb_lora_scale, rounds to bf16 (llmseg_lora_down's finish arithmetic), uses the result AND stores it to nb_lora_t[:, 0:16] (+ nb_lora_zero zero columns) for
   * llmseg_lora_wgrads: the finish launch of the backward's rank-8 down projection rides in the reduce launch.  NULL = nb_lora_t is an input. */
  const float* nb_lora_part; int32_t nb_lora_S; float nb_lora_scale; int32_t nb_lora_zero; int32_t reserved3;
} llmseg_gemm_args;
enum { LLMSEG_FX_NONE = 0, LLMSEG_FX_ROPE = 1, LLMSEG_FX_SWIGLU = 2, LLMSEG_FX_SWIGLU_BWD = 3 };
int llmseg_gemm_bf16(const llmseg_gemm_args* args, void* stream);
/* tuning knob (results are identical up to fp32 summation order of split-K; only speed differs):
 * bits 0-3: GEMM kernel for K % 64 == 0 shapes: 0 = register staging 128x128, 2 = LDS-DMA 128x128, 8 = LDS-DMA 256x256 ping-pong,
 *           9 = LDS-DMA 128x256 ping-pong, 5 = auto [default: a cost model picks kernel and split count];
 * bits 4-7: XCD skew + 1 (0 = keep); bits 8-12: forced split-K slice count for variants 8 / 9 (0 = 1 slice). */
int llmseg_gemm_set_variant(int variant);

/* ---- fused attention forward -----------------------------------------------------------------
 * O[b][h][q][:] = softmax_k( scale * Q.K^T + bias + mask ) V, online softmax in fp32, bf16 MFMA.
 * Q/K/V/O are addressed as base + b*stride_b + h*stride_h + row*stride_row (elements); head_dim in {32,64,80,128}.
 *  - causal + key_mask: HF LlamaAttention eager path (transformers 4.29; call site llava_llama.py:93-102)
 *  - rel_h/rel_w: SAM decomposed relative position (image_encoder.py:244-251, 354-392):
 *      bias(q,k) = rel_h[h][b][q][qh-kh+grid_h-1] + rel_w[h][b][q][qw-kw+grid_w-1], q=(qh,qw), k=(kh,kw) on a grid_h x grid_w grid,
 *      rel_* are fp32 [heads][batch*Nq][rel_ld] (the q . R^T products, as the strided-batched llmseg_gemm_bf16 with out_f32 writes them)
 *  - plain: CLIP / DINOv2 ViT attention, mask-selection head attention (model/transformer.py:319-341)
 * o_row_map (int32 [batch][Nq] or NULL): output row for query (b,q) inside O (rows of stride o_stride_row, ignoring
 * o_stride_b); negative = skip.  Used to fold SAM's window_unpartition + crop (image_encoder.py:291-318) into the store.
 */
typedef struct {
  uint32_t struct_size;  /* = sizeof(llmseg_attn_args) (ABI guard) */
  uint32_t reserved0;
  const void* Q; const void* K; const void* V; void* O;
  int64_t q_stride_b, q_stride_h, q_stride_row;
  int64_t k_stride_b, k_stride_h, k_stride_row;
  int64_t v_stride_b, v_stride_h, v_stride_row;
  int64_t o_stride_b, o_stride_h, o_stride_row;
  int32_t batch, heads, Nq, Nk, head_dim;
  float scale;
  int32_t causal;
  const uint8_t* key_mask;   /* [batch][Nk] 1 = attend, or NULL */
  const float* rel_h; const float* rel_w; int32_t rel_ld, grid_h, grid_w;
  const int32_t* o_row_map;
  /* Fused variant for SAM's 14x14 windows (head_dim 80): instead of rel_h/rel_w pass the bf16 tables themselves,
   * [32][head_dim] with rows >= 27 zero; q . R^T is then computed inside the kernel (grid_h = grid_w = 14 still required). */
  const void* rel_tab_h; const void* rel_tab_w;
  /* optional fp32 [batch][heads][Nq]: log2-sum-exp of every score row (max*scale*log2e + log2 sum), what llmseg_attn_bwd needs */
  float* lse;
  /* optional device int32: the number of keys present NOW (<= Nk, which is then the capacity of K / V); read by the kernel, so that a
   * decode step captured in a hipGraph can be replayed while the KV cache grows (llava_llama.py:137-163 prepare_inputs_for_generation) */
  const int32_t* nk_dev;
  /* window gather (ABI 9; only with rel_tab_h / rel_tab_w): the kernel applies SAM's window_partition + zero padding (image_encoder.py:263-289) and window_unpartition +
   * crop (:291-318) itself.  Q / K / V / O rows are then the UNPARTITIONED token rows of images on a win_grid x win_grid grid (row = image * win_grid^2 + y * win_grid + x,
   * *_stride_row apart; *_stride_b and o_row_map are ignored), batch = images * win_nw^2 windows (win_nw = ceil(win_grid / 14)); row (iy, ix) of window (wy, wx) is token
   * (14 wy + iy, 14 wx + ix) or, outside the grid, the row pad_q / pad_k / pad_v (bf16, head h at + h * *_stride_h): the reference pads AFTER norm1 (:178-183), so a padded
   * token's q|k|v row is the projection's bias -- the q|k|v product then runs on the real tokens only (4096 instead of 4900 rows per 1024^2 image).  win_grid = 0: off. */
  int32_t win_grid; int32_t win_nw;
  const void* pad_q; const void* pad_k; const void* pad_v;
} llmseg_attn_args;
int llmseg_attn_fwd(const llmseg_attn_args* args, void* stream);
/* tuning knob (identical results up to fp32 summation order): bit 0 = 1 [default]: SAM 14x14 windows on the resident-window kernel
 * (one workgroup per window and head, no key tiling); 0: the general tiled kernel */
int llmseg_attn_set_variant(int variant);

/* ---- fused attention backward ----------------------------------------------------------------
 * Gradients of O = softmax(scale * Q.K^T + mask) V w.r.t. Q, K, V (plain / causal / key_mask forms; no relative position:
 * the SAM tower is frozen).  Scores are recomputed tile by tile from Q, K and the forward's `lse`; nothing of size Nq x Nk is
 * written.  HF LlamaAttention backward (the LoRA gradients of training.py:218-226 flow through every layer's q/v) and the
 * mask-selection head's self attention (model/transformer.py:319-341).  head_dim in {32, 64, 128}; causal needs Nq == Nk.
 * Every tensor is addressed as base + b*stride_b + h*stride_h + row*stride_row (elements); dQ/dK/dV may be strided views of
 * one packed buffer.  delta is an fp32 [batch][heads][Nq] workspace (rowsum(dO * O), written by the dQ pass). */
typedef struct {
  uint32_t struct_size;  /* = sizeof(llmseg_attn_bwd_args) (ABI guard) */
  uint32_t reserved0;
  const void* Q; const void* K; const void* V; const void* O; const void* dO;
  void* dQ; void* dK; void* dV;
  int64_t q_stride_b, q_stride_h, q_stride_row;
  int64_t k_stride_b, k_stride_h, k_stride_row;
  int64_t v_stride_b, v_stride_h, v_stride_row;
  int64_t o_stride_b, o_stride_h, o_stride_row;
  int64_t do_stride_b, do_stride_h, do_stride_row;
  int64_t dq_stride_b, dq_stride_h, dq_stride_row;
  int64_t dk_stride_b, dk_stride_h, dk_stride_row;
  int64_t dv_stride_b, dv_stride_h, dv_stride_row;
  int32_t batch, heads, Nq, Nk, head_dim;
  float scale;
  int32_t causal;
  const uint8_t* key_mask;   /* [batch][Nk] 1 = attend, or NULL */
  const float* lse;
  float* delta;
  /* optional (ABI 8): fp32 [Nq][head_dim / 2] tables of a rotate-half rotation applied to every dQ and dK row (position = row index,
   * Nq == Nk) before it is stored, with llmseg_rope's arithmetic on the bf16-rounded gradient -- bit for bit what llmseg_rope over the
   * stored dQ | dK gives.  The Llama layer passes (cos, -sin): the inverse rotation of HF apply_rotary_pos_emb's backward
   * (modeling_llama.py, transformers 4.29), which used to be a launch of its own per layer.  head_dim 64 or 128. */
  const float* rope_cos; const float* rope_sin;
  /* 1: `delta` already holds rowsum(dO * O) (llmseg_gemm_args.dl_o wrote it): the delta launch is skipped (ABI 8) */
  int32_t delta_ready; int32_t reserved1;
} llmseg_attn_bwd_args;
int llmseg_attn_bwd(const llmseg_attn_bwd_args* args, void* stream);

/* ---- row-wise normalisation ------------------------------------------------------------------
 * y[row_map ? row_map[r] : r][:] = norm(x[r][:]) * w (+ b); statistics in fp32.
 * rms != 0: HF LlamaRMSNorm (variance over x^2, no mean, no bias).  Otherwise nn.LayerNorm / SAM LayerNorm2d in
 * channels-last form (common.py:31-43).  row_map folds SAM's window_partition (image_encoder.py:263-288) into the store
 * (padding rows of y are left untouched -- the caller zero-fills them once).
 */
int llmseg_norm(const void* x, const void* w, const void* b, void* y, int64_t rows, int64_t cols, int64_t ldx, int64_t ldy,
                float eps, int rms, const int32_t* row_map, void* stream);

/* RoPE, rotate-half form, applied in place to `heads` heads of width head_dim starting at x (row stride ld):
 * positions = row % T (HF LlamaRotaryEmbedding, position_ids = arange(T)); cos/sin fp32 [T][head_dim/2]. */
int llmseg_rope(void* x, const float* cos, const float* sin, int64_t rows, int64_t T, int32_t heads, int32_t head_dim,
                int64_t ld, void* stream);
/* Decode step of generation (HF LlamaAttention with past_key_values; call site llava_llama.py:93-102,137-163): qkv bf16 [N][3 * heads *
 * head_dim] (row stride ld) holds one new token per sequence.  With pos = *pos_dev: q is rotated in place at position pos, k is
 * rotated and written to kcache[n][pos], v is copied to vcache[n][pos] (caches bf16 [N][capacity][heads * head_dim], sequence stride
 * cache_stride_n elements).  cos/sin fp32 [capacity][head_dim/2].  The position lives in device memory so that the step can be
 * replayed from a hipGraph. */
int llmseg_rope_kv_append(void* qkv, int64_t ld, const float* cos, const float* sin, void* kcache, void* vcache, int64_t cache_stride_n,
                          const int32_t* pos_dev, int64_t N, int32_t heads, int32_t head_dim, void* stream);

/* The same decode step with the attention in the launch (head_dim 128): with pos = *pos_dev, k is rotated and written to kcache[n][pos], v
 * copied to vcache[n][pos], and out[n][h*128 ..] = softmax(rot(q_h) K_h^T * scale) V_h over the pos + 1 cached keys (HF LlamaAttention
 * with past_key_values and a one-token query: no mask).  qkv is not modified.  scratch (optional, fp32, >= N * heads * splits * 130 * 4
 * bytes for splits = min(16, 256 / (N * heads))): the keys of a head are split over workgroups until the launch covers the chip and a
 * second launch merges the partial softmaxes; without scratch one workgroup walks all keys of a head. */
int llmseg_decode_attn(const void* qkv, int64_t ld, const float* cos, const float* sin, void* kcache, void* vcache, int64_t cache_stride_n,
                       const int32_t* pos_dev, int64_t N, int32_t heads, int32_t head_dim, float scale, void* out, int64_t ldo,
                       void* scratch, int64_t scratch_bytes, void* stream);

/* y[i] = act(x[i]), bf16, n % 8 == 0, in place allowed (the GELU between LayerNorm2d and the second transposed convolution of SAM's
 * mask decoder, mask_decoder.py:53-63: every other activation on the path rides in a GEMM epilogue) */
int llmseg_act(const void* x, void* y, int64_t n, int32_t act, void* stream);
/* Sam.postprocess_masks (model/segment_anything/modeling/sam.py:137-172), fused: low fp32 [n_masks][256*256] mask logits ->
 * F.interpolate(bilinear, align_corners=False) to img_size^2 -> crop [:in_h, :in_w] -> F.interpolate to (out_h, out_w); out fp32
 * [n_masks][out_h][out_w].  nested = 1: `low` is in the row order the GEMM-form transposed convolutions emit (see head.hip). */
int llmseg_sam_postprocess(const float* low, float* out, int32_t n_masks, int32_t img_size, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                           int32_t nested, void* stream);
/* SAM "everything" mode (model/segment_anything/automatic_mask_generator.py:264-324, utils/amg.py:78-88,156-176,303-346), per candidate
 * mask at the ORIGINAL resolution without materialising the logits there.  low fp32 [n][256*256] (nested as above):
 *   llmseg_sam_mask_stats: stats int32 [n][7] += { |m > thr + off|, |m > thr - off|, |m > thr|, min x, min y, max x, max y } of the
 *     post-processed mask m; the caller initialises rows to {0, 0, 0, INT_MAX, INT_MAX, -1, -1}.  iou (optional, fp32 [n]): candidates
 *     with !(iou > iou_thresh) are skipped (pred_iou_thresh filter).  stability score = stats[0] / stats[1]; box = stats[3..6].
 *   llmseg_sam_binarize: out uint8 [n_sel][out_h][out_w] = m[sel[k]] > thr for the surviving candidates.
 *   llmseg_nms: greedy box NMS with torchvision.ops.nms semantics (one category): boxes fp32 [*][4] XYXY, order int32 [n] = candidate
 *     indices by decreasing score, keep uint8 [n] (position in `order`). */
int llmseg_sam_mask_stats(const float* low, const float* iou, float iou_thresh, int32_t* stats, int32_t n_masks, int32_t img_size, int32_t in_h,
                          int32_t in_w, int32_t out_h, int32_t out_w, int32_t nested, float mask_threshold, float offset, void* stream);
int llmseg_sam_binarize(const float* low, const int32_t* sel, uint8_t* out, int32_t n_sel, int32_t img_size, int32_t in_h, int32_t in_w, int32_t out_h,
                        int32_t out_w, int32_t nested, float mask_threshold, void* stream);
int llmseg_nms(const float* boxes, const int32_t* order, int32_t n, float iou_threshold, uint8_t* keep, void* stream);
/* SAM everything mode beyond the default single crop (SURVEY.md 8f N1 remainder; csrc/image.hip), byte work on the device:
 *   llmseg_image_resize_u8: `ResizeLongestSide.apply_image` (model/segment_anything/utils/transforms.py:27-35, called by
 *     `SamPredictor.set_image`, predictor.py:34-60) = Pillow's 8-bit BILINEAR `Image.resize`, bit-identical.  in uint8 [in_h][in_w][channels]
 *     with `in_row_stride` bytes between rows -- a crop of a crop layer (automatic_mask_generator.py:254-257) is an origin pointer + the full
 *     image's row stride; out uint8 [out_h][out_w][channels] dense.  workspace >= llmseg_image_resize_workspace(...) bytes.
 *   llmseg_sam_preprocess: `Sam.preprocess` (modeling/sam.py:174-186): in uint8 [h][w][3] -> out bf16 [3][img_size][img_size],
 *     (x - mean[c]) / std[c] inside the image, zero in the padding; mean / std are HOST pointers to 3 floats.
 *   llmseg_mask_small_regions: `remove_small_regions(mask, min_area, "holes")` then `(..., "islands")` (utils/amg.py:267-291, as
 *     `postprocess_small_regions` applies them, automatic_mask_generator.py:347-350) on masks uint8 [K][H][W] IN PLACE (0 / non-zero in,
 *     0 / 1 written where a pixel changes); changed uint8 [K] = 1 when either pass altered the mask.  8-connected components
 *     (cv2.connectedComponentsWithStats(.., 8) in the reference) by union-find; when every island is below min_area the largest is kept
 *     (first in raster order among equals).  workspace >= llmseg_mask_small_regions_workspace(K, H, W) bytes.
 *   llmseg_mask_boxes: `batched_mask_to_box` (utils/amg.py:303-346): boxes int32 [K][4] XYXY (inclusive; zeros for an empty mask),
 *     areas int32 [K] (optional); workspace >= 20 bytes per mask. */
int64_t llmseg_image_resize_workspace(int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w, int32_t channels);
int llmseg_image_resize_u8(const uint8_t* in, int64_t in_row_stride, uint8_t* out, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                           int32_t channels, void* workspace, int64_t workspace_bytes, void* stream);
int llmseg_sam_preprocess(const uint8_t* in, void* out, int32_t h, int32_t w, int32_t img_size, const float* mean, const float* std_, void* stream);
int64_t llmseg_mask_small_regions_workspace(int32_t K, int32_t H, int32_t W);
int llmseg_mask_small_regions(uint8_t* masks, int32_t K, int32_t H, int32_t W, int32_t min_area, uint8_t* changed, void* workspace,
                              int64_t workspace_bytes, void* stream);
int llmseg_mask_boxes(const uint8_t* masks, int32_t K, int32_t H, int32_t W, int32_t* boxes, int32_t* areas, void* workspace, int64_t workspace_bytes,
                      void* stream);
/* out[r][c] = silu(gu[r][c]) * gu[r][I + c]   (HF LlamaMLP: down(silu(gate(x)) * up(x)); gu = x.[Wgate;Wup]^T) */
int llmseg_swiglu(const void* gu, void* out, int64_t rows, int64_t I, int64_t ldgu, int64_t ldo, void* stream);

/* y[r][:] = x[r][:] + add[(r % add_rows)][:]  (positional embeddings; x may alias y) */
int llmseg_add_rows(const void* x, const void* add, void* y, int64_t rows, int64_t cols, int64_t add_rows, void* stream);

/* im2col for stride==kernel patch embedding: img bf16 [B][3][H][W] -> cols bf16 [B*gh*gw][ldo], column = c*p*p + i*p + j,
 * zero-filled up to ldo.  (PatchEmbed image_encoder.py:395-426; CLIP/DINOv2 patch convs.)  out_row_offset/out_rows_per_img
 * let the caller leave a CLS row at the front of each image's block. */
int llmseg_patchify(const void* img, void* cols, int32_t B, int32_t H, int32_t W, int32_t p, int64_t ldo,
                    int64_t out_rows_per_img, int64_t out_row_offset, void* stream);

/* 3x3 / pad 1 im2col on a channels-last map: x bf16 [B][H][W][C] -> cols [B*H*W][9*C], column = (ky*3+kx)*C + c
 * (SAM neck conv, image_encoder.py:100-106; the weight is re-laid-out to match at load time). */
int llmseg_im2col3x3(const void* x, void* cols, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);

/* LLaVA splice (llava_arch.py:185-208): out[n][t][:] = embed[ids] for text positions, img_feats[n][t-img_pos] inside the
 * image span.  ids int64 [N][L] with exactly one IMAGE token (-200) per row; out [N][L-1+P][H]; sequence n's P feature
 * rows start at img_feats + n*feats_stride_n (elements), so a CLS row per image can be skipped in place. */
int llmseg_embed_splice(const int64_t* ids, const void* embed, const void* img_feats, void* out, int32_t N, int32_t L,
                        int32_t P, int32_t H, int64_t vocab, int64_t feats_stride_n, void* stream);

/* gather rows: out[i][:] = x[idx[i]][:] (bf16, cols % 8 == 0) -- [SEG] hidden-state gather (LISA.py:322-323) */
int llmseg_gather_rows(const void* x, const int64_t* idx, void* out, int64_t n, int64_t cols, int64_t ldx, void* stream);

/* Fused bilinear-upsample + mask pooling (LISA.py:350-361, 201-218):
 * pooled[k][c] = sum_p segs[k][p] * Up(feat)[c][p] / (sum_p segs[k][p] + 1e-8), Up = F.interpolate(size=S, bilinear,
 * align_corners=False) of the channels-last bf16 map feat [g*g][C].  Computed as (segs . U) . feat, i.e. the mask is
 * pulled back through the adjoint of the interpolation, so the [C][S][S] upsampled tensor is never materialised and
 * `segs` (K*S*S bf16, the only large operand) is read from HBM exactly once.  feat bf16 [g*g][C]; pooled bf16 [K][C].
 * ws: caller-provided bf16 [K][g*g] workspace (the normalised pulled-back masks; stage 2 is an MFMA GEMM over it).
 * Optional outputs for the backward pass (NULL = skip): pulled_back fp32 [K][g*g] = segs . U, wsum fp32 [K] = sum_p segs. */
/* Stage 1 of the above alone: ws[k][:] = (segs[k] . U) / (sum_p segs[k][p] + 1e-8) as bf16 [K][g*g] (+ the optional fp32 outputs),
 * so that a caller can pool several images with ONE strided-batched llmseg_gemm_bf16 (trans_w) over their feature maps. */
int llmseg_mask_pullback(const void* segs, void* ws, float* pulled_back, float* wsum, int32_t K, int32_t g, int32_t S, void* stream);
int llmseg_upsample_maskpool(const void* feat, const void* segs, void* pooled, void* ws, float* pulled_back, float* wsum, int32_t K,
                             int32_t C, int32_t g, int32_t S, void* stream);

/* Cosine scoring (LISA.py:398-403): sim[k] = <t,e_k> / (|t||e_k|); t bf16 [D], e bf16 [K][D]; sim fp32 [K]. */
int llmseg_cosine_scores(const void* t, const void* e, float* sim, int32_t K, int32_t D, void* stream);

/* ---- ABI 7: the mask-selection head with fp32 ACTIVATIONS (inference scores; weights stay the model's bf16 tensors) ------------------------
 * What these replace on the reference side: the nn.Linear / nn.LayerNorm / Attention modules of model/transformer.py:215-341 and the scoring of
 * model/LISA.py:340-408 when the reference is run in fp32 on the CPU (its own "CPU path" of north_star).  Plain fp32 FMA chains, ascending k. */
/* y[m][n] = act(alpha * sum_k x[m][k] * W(n,k) + bias[n]) + residual[m][n];  x fp32 [M][ldx], residual fp32 [M][ldr] or NULL, y fp32 [M][ldy];
 * W bf16: w_kn = 0 -> stored [N][ldw] (nn.Linear weight, F.linear); w_kn = 1 -> stored [K][ldw] (right operand given K-major: the channels-last
 * feature map of the mask pooling, LISA.py:201-218); bias bf16 [N] or NULL. */
int llmseg_linear_f32(const float* x, int64_t ldx, const void* W, int64_t ldw, int32_t w_kn, const void* bias, const float* residual, int64_t ldr,
                      float* y, int64_t ldy, int32_t M, int32_t N, int32_t K, int32_t act, float alpha, void* stream);
/* nn.LayerNorm over the last dimension (transformer.py:236-283 norm1..4): x, y fp32 [rows][D] contiguous; w, b bf16 [D] (b may be NULL). */
int llmseg_layernorm_f32(const float* x, const void* w, const void* b, float* y, int64_t rows, int32_t D, float eps, void* stream);
/* softmax(q k^T * scale) v (transformer.py:319-341), all fp32; strides[12] = {q, k, v, o} x {batch, head, row} in elements; head_dim 32 or 64. */
int llmseg_attn_f32(const float* q, const float* k, const float* v, float* o, const int64_t* strides, int32_t batch, int32_t heads, int32_t Nq, int32_t Nk,
                    int32_t head_dim, float scale, void* stream);
/* llmseg_cosine_scores on fp32 operands: t fp32 [D], e fp32 [K][D] -> sim fp32 [K]. */
int llmseg_cosine_f32(const float* t, const float* e, float* sim, int32_t K, int32_t D, void* stream);

/* softmax_align_loss + iou_regression_loss (model/loss.py:50-94) for `items` (image, round) pairs with the same K, one workgroup
 * each, every operand contiguous over items (e [items][K][D], t [items][D], gt_iou / pred_iou / gt_iop [items][K]); fp32 results:
 * out[i][0] = KL(softmax(gt_iou/tau) || softmax(cos(e_k,t)/tau)) summed; out[i][1] = mean((p-g)^2 exp(g-1)) * 50.
 * Optional gradients: d_e fp32 [items][K][D], d_t fp32 [items][D], d_pred fp32 [items][K] (of out[i][0] resp. out[i][1]; NULL = skip). */
int llmseg_align_reg_loss(const void* e, const void* t, const float* gt_iou, const void* pred_iou, const float* gt_iop,
                          float* out, float* d_e, float* d_t, float* d_pred, int32_t K, int32_t D, float tau, int32_t items, void* stream);

/* dice_loss + sigmoid_ce_loss (model/loss.py:4-47; named by the north_star, no caller in the reference):
 * logits bf16/fp32-as-float [M][HW] given as fp32, targets fp32; out[0] = dice (scale 1000, eps 1e-6), out[1] = bce,
 * both summed over masks / (num_masks + 1e-8).  out must be zeroed by the caller.  workspace >= 8 M bytes (per-mask terms, folded in order). */
int llmseg_dice_bce(const float* logits, const float* targets, float* out, int32_t M, int64_t HW, float num_masks, void* workspace,
                    int64_t workspace_bytes, void* stream);
/* gradient of g[0] * dice + g[1] * bce w.r.t. the logits (g: device fp32[2], the upstream gradients of the two losses) */
int llmseg_dice_bce_bwd(const float* logits, const float* targets, const float* g, float* dlogits, int32_t M, int64_t HW, float num_masks, void* stream);

/* Shifted cross-entropy over bf16 logits (llava_llama.py:108-118): rows = N*T positions, labels int64 [N][T] already in
 * spliced form; position (n,t) is scored against labels[n][t+1]; ignore_index -100.  acc fp32[2] += {sum nll, count}.
 * workspace >= 8 N (T - 1) bytes (per-position terms, folded in order). */
int llmseg_ce_loss(const void* logits, const int64_t* labels, float* acc, int32_t N, int32_t T, int64_t V, int64_t ldl,
                   void* workspace, int64_t workspace_bytes, void* stream);


/* gIoU / cIoU bookkeeping (reference utils/utils.py:119-132 `intersectionAndUnionGPU`, K = 2 classes): pred/target uint8 [n],
 * pixels with target == ignore_index are dropped; out int64[6] += {I0, I1, U0, U1, T0, T1} (exact integer counts). */
int llmseg_intersection_union(const uint8_t* pred, const uint8_t* target, int64_t n, int32_t ignore_index, int64_t* out, void* stream);

/* validate_threshold's per-image body (training.py:712-766) in one pass: union of the proposals with select[k] != 0 (segs uint8
 * [H][W][K], the reader's layout), nearest-resize of that union and of gt (uint8 [Hg][Wg], 255 = ignore) to out_h x out_w, 2-class I/U.
 * `validate` (arg-max of the similarity, training.py:605-687) is the same with a one-hot select and out = Hg x Wg.
 * out int64[6] += {I0, I1, U0, U1, T0, T1}. */
int llmseg_union_resize_iou(const uint8_t* segs, const uint8_t* select, const uint8_t* gt, int32_t H, int32_t W, int32_t K, int32_t Hg, int32_t Wg,
                            int32_t out_h, int32_t out_w, int32_t ignore_index, int64_t* out, void* stream);

/* ---- proposal decode + training targets on the device (the reference does this per sample on CPU workers) -----
 * rle_decode: COCO run-length proposals -> dense uint8 masks (pycocotools `mask_util.decode`, utils/sam_mask_reader.py:86-87).  run_ends =
 *   inclusive prefix sums of every mask's run lengths, concatenated; offsets int64 [K+1] delimits mask k's slice; runs alternate 0 / 1 from 0
 *   and walk the image column-major.  out uint8 [K][H][W] (hwk = 0) or [H][W][K] (hwk = 1, the reader's layout).
 * mask_targets: `compute_all_iou` / `compute_all_iop` (utils/utils.py:234-272) for all K proposals: ground truth gt [Hg][Wg] resampled to the
 *   proposals' H x W grid through the nearest-neighbour index maps gy [H], gx [W] (skimage.transform.resize(order=0) rule, built on the host
 *   in float64), counts int64 [K][2] += {|seg & gt|, |seg|}, gt_area int64 [1] += |gt'| (caller zero-fills both), then
 *   iou[k] = I / (S + G - I), iop[k] = I / S as IEEE doubles (0 / 0 = nan, as numpy gives the reference).  Integer parts are exact.
 * resize_aa: proposal maps (utils/reason_seg_dataset.py:166-173): masks uint8 [K][H][W], zero-padded bottom / right to the square of side
 *   max(H, W), resampled to out_size x out_size with torch's antialiased bilinear filter, written as bf16 [K][out][out].  Tap tables per
 *   output index (first source index, tap count, float64 weights [out][taps]) come from the host (aten `_compute_indices_weights_aa`). */
/* One pass over the proposals (round 4): gt_resample writes the ground truth on the proposals' grid once (gtp uint8 [H][W] = gt[gy[y]][gx[x]] != 0);
 * proposal_targets reads every selected proposal ONCE -- proposal k is masks[order[k]] (order int64 [K] or NULL = identity: no gathered copy) --
 * and produces the bf16 [K][out][out] antialiased maps (the arithmetic of llmseg_resize_aa, evaluated separably: bit-identical) AND, for n_gt <= 4
 * ground truths gtp [n_gt][H][W], counts int64 [n_gt][K][2] = {|seg & gt'|, |seg|}, gt_area int64 [n_gt], iou / iop double [n_gt][K] (all plainly
 * written).  Limits: out_size <= 256, taps <= 28, W <= 2048; beyond them use llmseg_mask_targets + llmseg_resize_aa. */
int llmseg_gt_resample(const uint8_t* gt, const int32_t* gy, const int32_t* gx, uint8_t* out, int32_t H, int32_t W, int32_t Hg, int32_t Wg, void* stream);
int llmseg_proposal_targets(const uint8_t* masks, const int64_t* order, const uint8_t* gtp, int32_t n_gt, void* out, int32_t K, int32_t H, int32_t W,
                            int32_t out_size, const int32_t* y0, const int32_t* ny, const double* wy, const int32_t* x0, const int32_t* nx,
                            const double* wx, int32_t taps, int64_t* counts, int64_t* gt_area, double* iou, double* iop, void* stream);
int llmseg_rle_decode(const uint32_t* run_ends, const int64_t* offsets, uint8_t* out, int32_t K, int32_t H, int32_t W, int32_t hwk, void* stream);
int llmseg_mask_targets(const uint8_t* segs, const uint8_t* gt, const int32_t* gy, const int32_t* gx, int32_t K, int32_t H, int32_t W, int32_t Hg,
                        int32_t Wg, int64_t* counts, int64_t* gt_area, double* iou, double* iop, void* stream);
int llmseg_resize_aa(const uint8_t* segs, void* out, int32_t K, int32_t H, int32_t W, int32_t out_size, const int32_t* y0, const int32_t* ny,
                     const double* wy, const int32_t* x0, const int32_t* nx, const double* wx, int32_t taps, void* stream);

/* ---- backward pass + optimizer (trainable part: LoRA'd Llama stack, embed/lm_head, text_hidden_fcs, mask-selection head) -----
 * GEMM-shaped gradients use llmseg_gemm_bf16 with trans_a / trans_w (dX = dY W, dW = dY^T X); the kernels below are the
 * streaming pieces.  Gradients of the loss kernels: llmseg_align_reg_loss (d_e, d_t, d_pred). */
/* out[n] += sum_m x[m][n]  (bias gradients, fp32 accumulation; caller zero-fills out).  workspace: up to 64 N floats (row slices). */
int llmseg_colsum(const void* x, float* out, int64_t M, int64_t N, int64_t ld, void* workspace, int64_t workspace_bytes, void* stream);
/* LayerNorm / RMSNorm backward (contiguous rows): dx bf16; dw/db fp32 accumulated (NULL = frozen weight).  workspace (only read when dw or
 * db is given): rows of <= 1024 columns use up to 256 x 2 cols floats of partials (NULL: one workgroup); wider rows NEED 2 rows + 2 cols
 * floats (row statistics) and use up to 64 x 2 cols more. */
int llmseg_norm_bwd(const void* dy, const void* x, const void* w, void* dx, float* dw, float* db, int64_t rows, int64_t cols,
                    float eps, int rms, void* workspace, int64_t workspace_bytes, void* stream);
/* same, dx = norm backward + dres (bf16 [rows][cols] or NULL): x of a pre-norm block also feeds the residual connection, and the
 * gradient arriving that way is added here instead of in a separate pass (HF LlamaDecoderLayer: hidden = residual + sublayer(norm(hidden))) */
int llmseg_norm_bwd_add(const void* dy, const void* x, const void* w, const void* dres, void* dx, float* dw, float* db, int64_t rows, int64_t cols,
                        float eps, int rms, void* workspace, int64_t workspace_bytes, void* stream);
/* SwiGLU backward: gu [rows][2I] (gate|up), dout [rows][I] -> dgu [rows][2I] */
int llmseg_swiglu_bwd(const void* gu, const void* dout, void* dgu, int64_t rows, int64_t I, void* stream);
/* out = dy * f'(y) computed from the OUTPUT y of a fused GEMM epilogue (act = RELU or SIGMOID) */
int llmseg_act_bwd(const void* dy, const void* y, void* out, int64_t n, int act, void* stream);
/* Materialised attention probabilities for the backward pass (T <= a few hundred: Llama T=319, head K<=512):
 * P[b][q][k] = softmax_k(scale * S[b][q][k] + causal/key mask), S fp32 [BH][Tq][ld], P bf16 same shape, columns >= Tk zero.
 * key_mask uint8 [BH/heads][Tk] or NULL; causal requires Tq == Tk. */
int llmseg_softmax_rows(const float* S, void* P, int64_t BH, int32_t Tq, int32_t Tk, int32_t ld, float scale, int32_t causal,
                        const uint8_t* key_mask, int32_t heads, void* stream);
/* dS[r][k] = scale * P[r][k] * (dP[r][k] - sum_k' P[r][k'] dP[r][k'])  (softmax backward; rows = BH*T) */
int llmseg_attn_ds(const void* P, const float* dP, void* dS, int64_t rows, int32_t T, int32_t ld, float scale, void* stream);
/* dlogits = coef[0] * (softmax(logits) - onehot(shifted label)); zero for ignored positions (llava_llama.py:108-118 backward) */
int llmseg_ce_bwd(const void* logits, const int64_t* labels, const float* coef, void* dlogits, int32_t N, int32_t T, int64_t V,
                  int64_t ldl, void* stream);
/* dst[idx[i]][:] += src[i][:] in fp32 (embedding / row-gather gradients; idx < 0 skipped); source rows that share a destination are added
 * in ascending source order by one workgroup (no atomics) */
int llmseg_scatter_add_rows(const void* src, const int64_t* idx, float* dst, int64_t n, int64_t cols, void* stream);
/* Rank-8 LoRA products (peft==0.4.0 Linear with r = 8, lora_dropout 0.05: training.py:91,218-226) -- skinny shapes a tiled GEMM
 * cannot fill.  Every kernel handles the q AND the v branch of one layer in one launch (second operand set NULL = one branch);
 * branch b uses dropout stream drop->stream + b:
 *   lora_down:  y[m][8b..8b+7] = alpha * drop_b(x_b)[M][K] . W_b^T  (W stored [8][K], or [K][8] when w_kr), rows of y at pitch ldy,
 *               the following zero_cols columns of each row zero-filled                            fwd drop(x).A^T, bwd dq.Bq | dv.Bv
 *   lora_outer: out_b(n,r) += alpha * sum_m drop_b(a_b)[m][n] b_b[m][r]  (fp32 [N][8], or [8][N] when out_rn; accumulates)   dB, dA
 *   lora_apply: y[M][N] += alpha * sum_b mask_b * (xa[M][8b..8b+7] . W_b^T)  (W stored [N][8], or [8][N] when w_rn)   bwd dx of the LoRA branches
 *   lora_pack:  the two [*][64] extension operands of llmseg_gemm_args (A2 / W2) for a LoRA'd q|k|v projection, from the current
 *               LoRA matrices: w2b [3H][64] = rows [s Bq | 0], 0, [0 | s Bv | 0];  w2a [H][64] = rows [Aq[:,h] | Av[:,h] | 0];
 *               bt [16][H] = Bq^T | Bv^T (K-contiguous rows: the backward's dq.Bq | dv.Bv then runs on the MFMA form of lora_down);
 *               every output is optional (NULL)
 * Dropout (NULL = none, as in eval mode) is counter-based, nothing is stored: Philox4x32-10 with key = rng_state[0] (seed), counter
 * = (element index / 8, stream, rng_state[1] (offset)); the 16-bit field j of the 128-bit output decides element 8 idx + j, kept
 * when field >= drop_thr (= round(p * 65536)) and scaled by 65536 / (65536 - drop_thr); element index = row * width + column of
 * the dense [M][width] activation.  rng_state is DEVICE memory (a captured hipGraph reads a fresh offset on every replay).
 * seg_rows > 0 (ABI 5): the M rows are consecutive SEGMENTS of seg_rows rows -- the micro-batches of a gradient-accumulation window run as
 * one pass -- and segment s = row / seg_rows draws the mask a separate pass over it would draw at offset + s: element index =
 * (row % seg_rows) * width + column, counter offset = rng_state[1] + s. */
typedef struct {
  const uint64_t* rng_state;   /* device: {seed, offset} */
  uint32_t stream;             /* which dropout module (layer * 2 + {q = 0, v = 1}) */
  uint32_t drop_thr;           /* round(p * 65536); 0 = no dropout */
  uint32_t seg_rows;           /* 0 = one segment (all rows at rng_state[1]) */
  uint32_t reserved0;          /* 0 */
} llmseg_dropout;
int llmseg_lora_down(const void* x0, const void* x1, int64_t ldx, const void* w0, const void* w1, void* y, int64_t ldy, int64_t M, int64_t K,
                     int32_t w_kr, float alpha, int32_t zero_cols, const llmseg_dropout* drop, void* stream);
/* lora_down with caller-owned scratch (fp32, >= 2 * 32 * M * 16 * 4 bytes lets every split be taken): short activations (M = 2 x 319) are
 * split along K over several workgroups per 16-row tile so that the whole chip fetches the operand, a second launch adds the slices */
int llmseg_lora_down_ws(const void* x0, const void* x1, int64_t ldx, const void* w0, const void* w1, void* y, int64_t ldy, int64_t M, int64_t K,
                        int32_t w_kr, float alpha, int32_t zero_cols, const llmseg_dropout* drop, void* scratch, int64_t scratch_bytes, void* stream);
/* workspace: up to 32 row slices x (1 or 2 branches) x 8 N floats of partials (NULL: one slice) */
int llmseg_lora_outer(const void* a0, const void* a1, int64_t lda, const void* b0, const void* b1, int64_t ldb, float* out0, float* out1, int64_t M,
                      int64_t N, int32_t out_rn, float alpha, const llmseg_dropout* drop, void* workspace, int64_t workspace_bytes, void* stream);
/* The four weight gradients of a LoRA'd q|k|v projection in ONE launch (+ one fold): with d = dqkv (dq, dv: column blocks at row pitch ldd),
 * xa = [drop_q(x) Aq^T | drop_v(x) Av^T] [M][>= 16] (the forward's extension operand), t = [s dq Bq | s dv Bv] [M][>= 16]:
 *   gbq [H][8] += s dq^T xa[:, 0:8],  gbv [H][8] += s dv^T xa[:, 8:16],  gaq [8][H] += t[:, 0:8]^T drop_q(x),  gav [8][H] += t[:, 8:16]^T drop_v(x)
 * (dropout streams drop->stream / + 1, as llmseg_lora_down uses them).  workspace: up to 32 row slices x 4 x 8 H floats. */
int llmseg_lora_wgrads(const void* dq, const void* dv, int64_t ldd, const void* x, int64_t ldx, const void* xa, int64_t ldxa, const void* t, int64_t ldt,
                       float* gbq, float* gbv, float* gaq, float* gav, int64_t M, int64_t H, float s, const llmseg_dropout* drop, void* workspace,
                       int64_t workspace_bytes, void* stream);
int llmseg_lora_apply(void* y, int64_t ldy, const void* xa, int64_t ldxa, const void* w0, const void* w1, int64_t M, int64_t N, int32_t w_rn,
                      float alpha, const llmseg_dropout* drop, void* stream);
int llmseg_lora_pack(const void* aq, const void* bq, const void* av, const void* bv, void* w2b, void* w2a, void* bt, int64_t H, float s, void* stream);
/* llmseg_lora_down_ws that leaves a K-sliced product UNFINISHED (ABI 8): *S_out = slice count with the fp32 partials [S][M][16] in `scratch` and y untouched (hand them
 * to llmseg_gemm_args.nb_lora_part), or 0 when the shape took no slices and y is complete; *scale_out = alpha * dropout scale (the finish factor). */
int llmseg_lora_down_parts(const void* x0, const void* x1, int64_t ldx, const void* w0, const void* w1, void* y, int64_t ldy, int64_t M, int64_t K,
                           int32_t w_kr, float alpha, int32_t zero_cols, const llmseg_dropout* drop, void* scratch, int64_t scratch_bytes, int32_t* S_out,
                           float* scale_out, void* stream);
/* llmseg_lora_down_ws followed by llmseg_lora_pack in one call (ABI 8): the two are independent (activations vs. weights), so where the down projection runs
 * as K slices the pack rides in its finish launch -- one launch fewer per LoRA'd projection; the same bits as the two calls. */
int llmseg_lora_down_pack(const void* x0, const void* x1, int64_t ldx, const void* w0, const void* w1, void* y, int64_t ldy, int64_t M, int64_t K,
                          int32_t w_kr, float alpha, int32_t zero_cols, const llmseg_dropout* drop, void* scratch, int64_t scratch_bytes, const void* aq,
                          const void* bq, const void* av, const void* bv, void* w2b, void* w2a, void* bt, int64_t H, float s, void* stream);
/* out[c][r] = in[r][c] (bf16; in [rows][cols] with leading dimension ld_in, out [cols][ld_out]); rows r in [rows, rows_pad) of the
 * source are taken as zero.  Lets the weight / input gradients of a wide trainable Linear (lm_head: dX = dY.W, dW = dY^T.X) run on
 * the K-contiguous LDS-DMA GEMM kernels with the contraction dimension padded to a multiple of 64. */
int llmseg_transpose_pad(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out, int64_t rows_pad, void* stream);
/* out[0] += sum x^2 (global gradient-norm clipping, training.py:301 "gradient_clipping": 1.0).  workspace: required, 8 KiB holds every
 * workgroup's partial (fewer bytes = fewer workgroups). */
int llmseg_sumsq(const void* x, int64_t n, int is_f32, float* out, void* workspace, int64_t workspace_bytes, void* stream);
/* Fused AdamW on fp32 master weights + bf16 model copy (DeepSpeed config training.py:292-332: betas (0.9, 0.95), wd 0).
 * grad is bf16 (grad_f32 = 0) or fp32; grad_scale (device fp32 scalar or NULL) carries 1/accum and the clip coefficient. */
int llmseg_adamw(void* p, float* master, const void* grad, int grad_f32, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int64_t step, const float* grad_scale, void* stream);

/* ---- per-kernel timing (bench roofline): HIP events recorded around every GEMM launch on its own stream ---------- */
int llmseg_prof_enable(int on);                 /* 1 = record events around GEMM launches */
/* syncs the events, resets; totals over every GEMM launch, and the same for the dominant kernel only
 * (gemm_bf16_tn_glds_kernel<false, 2, 1>: bf16-out 128x128 LDS-DMA variant) so that it can be compared with rocprofv3's per-kernel row */
/* name of the GEMM kernel class that took the most time in the last collected window (its totals are dom_*) */
const char* llmseg_prof_dominant_kernel(void);
/* algorithmic bytes (A + W + C read / written once, bf16) summed over that class's launches in the last collected window */
double llmseg_prof_dominant_bytes(void);
/* composition of that class in the last collected window (a timed record is one user-level llmseg_gemm_bf16 CALL): out4 = {calls, calls that ran
 * as K-slices, K-slices summed over those calls, kernel launches = calls + one splitk_reduce_kernel per K-sliced call} */
int llmseg_prof_dominant_info(int64_t* out4);
int llmseg_prof_collect(double* total_ms, double* total_flops, int64_t* launches, double* dom_ms, double* dom_flops, int64_t* dom_launches);

#ifdef __cplusplus
}
#endif
#endif
