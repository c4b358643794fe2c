"""Oracle: the generation half of `LISAForCausalLM.evaluate` (test infrastructure; reference `model/LISA.py:477-521`).

`evaluate` calls `self.generate(images=..., input_ids=..., max_new_tokens=..., num_beams=1, output_hidden_states=True,
return_dict_in_generate=True)`: with those arguments HF runs greedy search -- third party, `transformers==4.29.0`
(`requirements.txt:276`) `generation/utils.py::greedy_search`, restated here:
    every step: forward -> logits of the last position -> argmax; finished rows emit `pad_token_id`;
    a row finishes when it emits `eos_token_id`; stop when all rows finished or `max_new_tokens` tokens were added.
Each step's forward is the reference's own `LlavaLlamaForCausalLM.forward` (llava_llama.py:55-135, restated in oracle/lisa.py and
pinned by tests/golden/lisa_tiny.pt).  The reference then takes `outputs.hidden_states[-1]`, the hidden states of the LAST
forward; its mask arithmetic (LISA.py:497-506: `output_ids[:, 1:]` plus 255 leading zeros) has the length of the whole final
sequence minus one, i.e. it assumes that forward saw every token but the last one -- the shipped configs generate with
`use_cache = False` (training.py sets it before saving).  With a KV cache the same tensor is the concatenation of the prefill's
and every decode step's hidden states, which is what the product returns.
PARITY: pinned by oracle/make_goldens.py::gold_generate against a greedy loop over the imported reference forward AND against
`model.generate(..., num_beams=1, use_cache=False[, eos_token_id, pad_token_id])` of the imported reference on the installed
transformers (5.x; the pinned 4.29 is not installed): identical sequences with and without an early eos
(tests/golden/generate_tiny.pt, key `hf_generate_agrees`).
"""
import torch

from . import lisa as _lisa


def greedy_generate(sd, cfg, images_clip, input_ids, max_new_tokens=32, eos_token_id=2, pad_token_id=0, forward=None, margins=None):
    """-> (sequences [N, L + n_new] int64, hidden [N, T_total - 1, H]: final-norm hidden state of every token but the last).
    `forward(ids) -> (logits, hidden)` overrides the restated forward (make_goldens passes the imported reference's).
    `margins`: a list that receives, per step, the top-1 minus top-2 logit of every row (how decidable the greedy choice was)."""
    ids = input_ids.clone()
    N = ids.shape[0]
    unfinished = torch.ones(N, dtype=torch.long)
    hidden = None
    for _ in range(max_new_tokens):
        if forward is None:
            _, logits, hidden = _lisa.llava_forward(sd, cfg, images_clip, torch.ones_like(ids, dtype=torch.bool), ids)
        else:
            logits, hidden = forward(ids)
        nxt = logits[:, -1, :].float().argmax(-1)
        if margins is not None:
            t2 = logits[:, -1, :].float().topk(2, -1).values
            margins.append(t2[:, 0] - t2[:, 1])
        if eos_token_id is not None:
            nxt = nxt * unfinished + pad_token_id * (1 - unfinished)
        ids = torch.cat([ids, nxt[:, None]], 1)
        if eos_token_id is not None:
            unfinished = unfinished * (nxt != eos_token_id).long()
            if int(unfinished.max()) == 0:
                break
    return ids, hidden


def seg_embeddings(sd, cfg, output_ids, hidden):
    """LISA.py:497-521: text_hidden_fcs over every hidden state, rows where the NEXT token is [SEG] (255 = image-token expansion - 1),
    split per sequence."""
    import torch.nn.functional as F
    m = output_ids[:, 1:] == cfg.seg_token_idx
    m = torch.cat([torch.zeros(m.shape[0], cfg.n_img_tokens - 1, dtype=torch.bool), m], 1)
    assert m.shape[1] == hidden.shape[1], (m.shape, hidden.shape)
    h = F.linear(F.relu(F.linear(hidden, sd["model.text_hidden_fcs.0.0.weight"], sd["model.text_hidden_fcs.0.0.bias"])),
                 sd["model.text_hidden_fcs.0.2.weight"], sd["model.text_hidden_fcs.0.2.bias"])
    cnt = m.int().sum(-1)
    off = [0] + cnt.cumsum(0).tolist()
    pe = h[m]
    return [pe[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def evaluate(sd, cfg, images_clip, images, input_ids, resize_list, original_size_list, max_new_tokens=32, eos_token_id=2, pad_token_id=0):
    """`LISAForCausalLM.evaluate` (LISA.py:477-559) from the restated pieces: generation, [SEG] embeddings, SAM image embedding
    (oracle/lisa.py::visual_features), prompt encoder + mask decoder + post-processing (oracle/sam_decoder.py).
    -> (output_ids, [fp32 [n_seg, H, W]], aux)"""
    from . import sam_decoder as sdec
    ids, hidden = greedy_generate(sd, cfg, images_clip, input_ids, max_new_tokens, eos_token_id, pad_token_id)
    pe = seg_embeddings(sd, cfg, ids, hidden)
    feats = _lisa.visual_features(sd, cfg, images)                      # [B, 256, 64, 64]
    masks = []
    for i in range(len(pe)):
        if pe[i].shape[0] == 0:
            masks.append(torch.empty((0,) + tuple(original_size_list[i])))
            continue
        low, _ = sdec.decode_masks(sd, feats[i:i + 1], pe[i])
        masks.append(sdec.postprocess_masks(low, resize_list[i], original_size_list[i], cfg.sam.img)[:, 0])
    return ids, masks, dict(hidden=hidden, pred_embeddings=pe, feats=feats)
