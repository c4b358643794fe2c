"""GPU: KV-cache greedy generation (llmseg_amd/generate.py) against the oracle's cache-free greedy loop on the same bf16-rounded weights.
A greedy choice is only comparable where the fp32 margin (top-1 minus top-2 logit) exceeds what bf16 arithmetic can move: the first
step where the tokens differ must be such an undecidable step, and everything after it is skipped for that sequence."""
import torch

from oracle import cases, generate as ogen
from tests import model_checks as mc

DEV = "cuda"
MARGIN = 0.08          # logits of the tiny model are O(1); the bf16 path moves them by ~1e-2


def check_generate(max_new=6, lora_r=0):
    """lora_r = 8: the decode steps run on q|k|v weights with the LoRA deltas merged in (prefill: the unmerged training forward)."""
    cfg = cases.tiny_lisa_cfg(lora_r=lora_r)
    m, sd = mc.build_pair(cfg)
    batch = mc._round_batch(cases.tiny_lisa_batch())
    clip, ids0 = batch["images_clip"][:2], batch["input_ids"][:2]
    L = ids0.shape[1]
    res = []
    with torch.no_grad():
        margins = []
        seq_r, hid_r = ogen.greedy_generate(sd, cfg, clip, ids0, max_new_tokens=max_new, eos_token_id=None, margins=margins)
        seq_g, hid_g = m.generate(clip.to(DEV), ids0.to(DEV), max_new_tokens=max_new, eos_token_id=None)
        seq_g, hid_g = seq_g.cpu(), hid_g.float().cpu()
        assert seq_g.shape == seq_r.shape and hid_g.shape == hid_r.shape, (seq_g.shape, seq_r.shape, hid_g.shape, hid_r.shape)
        mg = torch.stack(margins, 1)                                   # [N, steps]
        same_upto = []
        for n in range(seq_r.shape[0]):
            k = 0
            while k < max_new and seq_g[n, L + k] == seq_r[n, L + k]:
                k += 1
            if k < max_new:
                res.append((f"generate: seq {n} first differing step {k} has margin {float(mg[n, k]):.3f} (must be undecidable)", float(mg[n, k]), MARGIN))
            same_upto.append(k)
        res.append(("generate: steps compared (need >= 3 of %d on every sequence)" % max_new, 3.0 - min(same_upto), 0.0))
        Tp = hid_r.shape[1] - (max_new - 1)
        scale = max(1.0, hid_r.abs().max().item())
        res.append(("generate: prefill hidden", (hid_g[:, :Tp] - hid_r[:, :Tp]).abs().max().item(), 3e-2 * scale))
        for n in range(seq_r.shape[0]):
            k = min(same_upto[n], max_new - 1)                         # decode step j is fed token j: valid while tokens 0..j agree
            if k > 0:
                res.append((f"generate: decode hidden seq {n} ({k} cached steps)", (hid_g[n, Tp:Tp + k] - hid_r[n, Tp:Tp + k]).abs().max().item(), 3e-2 * scale))
        # one sequence alone: the decode step with RMSNorm / SwiGLU on the skinny GEMM's A load (N = 1 only) against the same oracle run
        seq_1, hid_1 = m.generate(clip[:1].to(DEV), ids0[:1].to(DEV), max_new_tokens=max_new, eos_token_id=None)
        seq_1, hid_1 = seq_1.cpu(), hid_1.float().cpu()
        k = 0
        while k < max_new and seq_1[0, L + k] == seq_r[0, L + k]:
            k += 1
        if k < max_new:
            res.append((f"generate: single sequence first differing step {k} has margin {float(mg[0, k]):.3f} (must be undecidable)", float(mg[0, k]), MARGIN))
        res.append(("generate: single sequence steps compared (need >= 3)", 3.0 - k, 0.0))
        k = min(k, max_new - 1)
        if k > 0:
            res.append((f"generate: single-sequence decode hidden ({k} cached steps)", (hid_1[0, Tp:Tp + k] - hid_r[0, Tp:Tp + k]).abs().max().item(), 3e-2 * scale))
        # eos / pad rule: sequence 0 finishes at its third new token
        if min(same_upto) >= 3:
            eos = int(seq_r[0, L + 2])
            seq_re, _ = ogen.greedy_generate(sd, cfg, clip, ids0, max_new_tokens=max_new, eos_token_id=eos, pad_token_id=0)
            seq_ge, hid_ge = m.generate(clip.to(DEV), ids0.to(DEV), max_new_tokens=max_new, eos_token_id=eos, pad_token_id=0)
            ok = seq_ge.shape == seq_re.shape and bool((seq_ge.cpu()[0] == seq_re[0]).all())
            res.append(("generate: eos -> pad rule on the finished row", 0.0 if ok else 1.0, 0.0))
            one, _ = m.generate(clip[:1].to(DEV), ids0[:1].to(DEV), max_new_tokens=max_new, eos_token_id=eos, pad_token_id=0)
            res.append(("generate: early stop when every row finished", float(abs(one.shape[1] - (L + 3))), 0.0))
        # [SEG] rows through text_hidden_fcs
        fake = seq_r.clone()
        fake[0, L + 1] = cfg.seg_token_idx
        er = ogen.seg_embeddings(sd, cfg, fake, hid_r)
        eg = m.seg_embeddings(fake.to(DEV), hid_r.to(DEV, torch.bfloat16))
        for n in range(2):
            assert eg[n].shape == er[n].shape
            if er[n].numel():
                res.append((f"generate: [SEG] embeddings seq {n}", (eg[n].float().cpu() - er[n]).abs().max().item(), 3e-2 * max(1.0, er[n].abs().max().item())))
    return res
