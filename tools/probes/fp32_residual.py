"""A/B: would an fp32 RESIDUAL STREAM in the Llama stack and the SAM ViT-H encoder bring the full-depth mask scores within the
flat 1e-3 of north_star?  Emulation on the device, inference only: residual adds and the inputs of every norm in fp32 (torch ops on the fp32 stream),
GEMM operands / attention unchanged (bf16 in, fp32 accumulate), the GEMMs that feed the stream write fp32 (`out_f32`).  Compared with the shipped bf16
stream against the SAME fp32 oracle run, seed by seed.  What an implementation would cost: +2 bytes / element on the reads and writes of the stream.

    python tools/probes/fp32_residual.py [n_seeds]      (one fp32 oracle forward per seed on the host: ~50 s each on 128 threads)
"""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from llmseg_amd import lisa as hip_lisa, ops, params as hp, synthetic      # noqa: E402
from llmseg_amd.trainable import _Direct                                    # noqa: E402
from oracle import lisa as olisa, llama as ol, sam_encoder as osam, vit as ovit      # noqa: E402
from tests.fulldepth_checks import _LazyState, _e                           # noqa: E402

BF = torch.bfloat16


def rms32(x32, w, eps):
    """HF LlamaRMSNorm on an fp32 stream: statistics and normalisation in fp32, cast, scale."""
    y = x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + eps)
    return (w * y.to(BF)).contiguous()


def llama_fp32res(m, embeds, key_mask_u8, Fn=_Direct, kv_out=None, drop_seg_rows=0):
    c = m.config.llama
    N, T, H = embeds.shape
    x32 = embeds.reshape(N * T, H).float()
    rope = m._rope(T)
    s = c.lora_alpha / c.lora_r
    P = m.params
    for i in range(c.layers):
        p = f"model.layers.{i}."
        h = rms32(x32, P[p + "input_layernorm.weight"], c.eps)
        lp = p + "self_attn."
        qkv = Fn.lora_qkv(h, P[p + "qkv"], P[lp + "q_proj.lora_A.default.weight"], P[lp + "q_proj.lora_B.default.weight"],
                          P[lp + "v_proj.lora_A.default.weight"], P[lp + "v_proj.lora_B.default.weight"], s)
        a = Fn.rope_attn(qkv, rope, N, T, c.heads, c.head_dim, True, key_mask_u8)
        x32 = x32 + ops.gemm(a, P[p + "self_attn.o_proj.weight"], out_f32=True)
        h = rms32(x32, P[p + "post_attention_layernorm.weight"], c.eps)
        gu = ops.gemm(h, P[p + "gate_up"])
        x32 = x32 + ops.gemm(ops.swiglu(gu, c.inter), P[p + "mlp.down_proj.weight"], out_f32=True)
    return rms32(x32, P["model.norm.weight"], c.eps).view(N, T, H)


def sam_fp32res(m, images):
    s, P, d = m.config.sam, m.params, m.prepare()
    B = images.shape[0]
    g, D, nh = s.grid, s.dim, s.heads
    hd = D // nh
    sp = "model.visual_model.image_encoder."
    ln = lambda x32, n: F.layer_norm(x32, (x32.shape[-1],), P[n + ".weight"].float(), P[n + ".bias"].float(), s.eps).to(BF)
    cols = ops.patchify(images, s.patch, 3 * s.patch ** 2)
    x32 = ops.gemm(cols, d["sam.patch_w"], bias=P[sp + "patch_embed.proj.bias"], out_f32=True)
    x32 = x32 + d["sam.pos"].float().repeat(B, 1)
    part, unpart, n_win, per_img = m._window_maps(B, g, s.window)
    winbuf = torch.zeros((B * per_img, D), device=x32.device, dtype=BF)
    for i in range(s.depth):
        p = f"{sp}blocks.{i}."
        glob = i in s.global_idx
        sz = g if glob else s.window
        ld = d[f"sam.relh.{i}"].shape[0]
        h = ln(x32, p + "norm1")
        if glob:
            batch, n_tok = B, g * g
        else:
            winbuf.index_copy_(0, part.long(), h)                       # padding rows stay zero
            h = winbuf
            batch, n_tok = B * n_win, s.window * s.window
        qkv = ops.gemm(h, P[p + "attn.qkv.weight"], bias=P[p + "attn.qkv.bias"])
        rows = batch * n_tok
        a = torch.empty((B * g * g, D), device=x32.device, dtype=BF)
        if (not glob) and hd == 80 and sz == 14:
            ops.attention_packed(qkv, batch, n_tok, nh, hd, out=a, rel_tab_h=d[f"sam.relh.{i}"], rel_tab_w=d[f"sam.relw.{i}"], grid_hw=(sz, sz), o_row_map=unpart)
        else:
            rel = torch.empty((2, nh, rows, ld), device=x32.device, dtype=torch.float32)
            for j, tab in enumerate((d[f"sam.relh.{i}"], d[f"sam.relw.{i}"])):
                ops.gemm_batched(qkv, tab, rel[j], M=rows, N=2 * sz - 1, K=hd, lda=3 * D, ldw=hd, ldc=ld, batch=nh, sA=hd, sW=0, sC=rows * ld, out_f32=True)
            ops.attention_packed(qkv, batch, n_tok, nh, hd, out=a, rel_h=rel[0], rel_w=rel[1], rel_ld=ld, grid_hw=(sz, sz), o_row_map=None if glob else unpart)
        x32 = x32 + ops.gemm(a, P[p + "attn.proj.weight"], bias=P[p + "attn.proj.bias"], out_f32=True)
        h = ops.gemm(ln(x32, p + "norm2"), P[p + "mlp.lin1.weight"], bias=P[p + "mlp.lin1.bias"], act=ops.ACT_GELU)
        x32 = x32 + ops.gemm(h, P[p + "mlp.lin2.weight"], bias=P[p + "mlp.lin2.bias"], out_f32=True)
    y = ops.gemm(x32.to(BF), d["sam.neck0_w"])
    y = ops.norm(y, P[sp + "neck.1.weight"], P[sp + "neck.1.bias"], eps=s.eps)
    y = ops.gemm(ops.im2col3x3(y, B, g, g, s.out_chans), d["sam.neck2_w"])
    return ops.norm(y, P[sp + "neck.3.weight"], P[sp + "neck.3.bias"], eps=s.eps)


def one_seed(seed, K=256, L=64):
    dev = "cuda"
    hcfg = hp.LisaConfig(backbone="sam", build_unused_towers=False)
    hcfg.llama = hp.LlamaConfig(lora_r=8)
    m = hip_lisa.LISAForCausalLM(hcfg, device=dev).init_random(seed=5 + seed)
    m.eval()
    m.overlap_towers = False
    ocfg = olisa.LisaCfg(llama=ol.LlamaCfg(lora_r=8), clip=ovit.VitCfg(eps=1e-5, img=224), sam=osam.SamCfg(), backbone="sam")
    batch = synthetic.make_batch(1, img_size=1024, L=L, K=K, device=dev, seed=4321 + seed, soft=True)
    inf = dict(images=batch["images"], images_clip=batch["images_clip"], input_ids=batch["input_ids"], labels=None, attention_masks=batch["attention_masks"],
               offset=batch["offset"], sam_segs_list=batch["sam_segs_list"])
    with torch.no_grad():
        base = m.model_forward(**inf, inference=True, return_aux=True)
        m.__dict__["_llama"] = lambda e, km, Fn, kv_out=None, drop_seg_rows=0: llama_fp32res(m, e, km, Fn)
        both_l = m.model_forward(**inf, inference=True, return_aux=True)             # fp32 stream in Llama only
        m.__dict__["_sam_encoder_cl"] = lambda im: sam_fp32res(m, im)
        both = m.model_forward(**inf, inference=True, return_aux=True)               # + SAM
        m.__dict__.pop("_llama"); m.__dict__.pop("_sam_encoder_cl")
    torch.cuda.synchronize()
    host = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    cpu = lambda t: t.detach().cpu().float() if (torch.is_tensor(t) and t.is_floating_point()) else (t.cpu() if torch.is_tensor(t) else t)
    b = {k: ([cpu(t) for t in v] if isinstance(v, list) else cpu(v)) for k, v in inf.items()}
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = olisa.model_forward(_LazyState(host, torch.float32), ocfg, **b, inference=True, return_aux=True)
    t_ref = time.perf_counter() - t0
    B, C, g, _ = ref["feats"].shape
    rf = ref["feats"].permute(0, 2, 3, 1).reshape(B * g * g, C)
    row = {}
    for tag, got in (("bf16 stream (shipped)", base), ("fp32 stream: Llama", both_l), ("fp32 stream: Llama + SAM", both)):
        row[tag] = dict(feats=_e(got["feats"].view(B * g * g, C), rf), hidden=_e(got["hidden"], ref["hidden"]), logits=_e(got["logits"], ref["logits"]),
                        emb=_e(got["pred_embeddings"][0], ref["pred_embeddings"][0]), sim=_e(got["pred_similarity"][0], ref["pred_similarity"][0]),
                        iou=_e(got["pred_iou"][0], ref["pred_iou"][0]))
    del m
    torch.cuda.empty_cache()
    return row, t_ref


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    print("# fp32 residual stream A/B at full depth (BASELINE configs[1]: 32-layer Llama-7B + LoRA, CLIP-L, 32-block SAM ViT-H, 1 image, K = 256)\n")
    print("max |HIP - fp32 oracle| per output; `sim` = pred_similarity, `iou` = pred_iou (north_star's flat bound on both: 1e-3)\n")
    print("| seed | variant | SAM feats | Llama hidden | logits | [SEG] embedding | sim | iou | sim <= 1e-3 | iou <= 1e-3 |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    met = {}
    for s in range(n):
        row, t_ref = one_seed(s)
        for tag, r in row.items():
            print(f"| {s} | {tag} | {r['feats']:.2e} | {r['hidden']:.2e} | {r['logits']:.2e} | {r['emb']:.2e} | {r['sim']:.2e} | {r['iou']:.2e} | "
                  f"{'yes' if r['sim'] <= 1e-3 else 'no'} | {'yes' if r['iou'] <= 1e-3 else 'no'} |", flush=True)
            m_ = met.setdefault(tag, [0, 0])
            m_[0] += r["sim"] <= 1e-3
            m_[1] += r["iou"] <= 1e-3
        print(f"<!-- seed {s}: fp32 oracle forward {t_ref:.1f} s -->", flush=True)
    print()
    for tag, (a, b) in met.items():
        print(f"- {tag}: pred_similarity within 1e-3 on {a} / {n} seeds, pred_iou on {b} / {n}")
