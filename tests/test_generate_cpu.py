"""CPU: the generation oracle (oracle/generate.py) against the fixture made from the imported reference (tests/golden/generate_tiny.pt:
greedy loop over the reference forward, cross-checked against HF `generate` there)."""
import torch

from oracle import cases, generate as gen


def test_generate_oracle_matches_reference_fixture(golden):
    g = golden("generate_tiny.pt")
    assert g["hf_generate_agrees"]
    cfg = cases.tiny_lisa_cfg()
    sd = cases.tiny_lisa_state(cfg)
    batch = cases.tiny_lisa_batch()
    clip, ids0 = batch["images_clip"][:2], batch["input_ids"][:2]
    with torch.no_grad():
        seq, hid = gen.greedy_generate(sd, cfg, clip, ids0, max_new_tokens=6, eos_token_id=None)
        assert torch.equal(seq, g["sequences"])
        assert (hid - g["hidden"]).abs().max() < 1e-4
        seq_e, hid_e = gen.greedy_generate(sd, cfg, clip, ids0, max_new_tokens=6, eos_token_id=g["eos"], pad_token_id=0)
        assert torch.equal(seq_e, g["sequences_eos"])
        assert abs(float(hid_e.double().sum()) - float(g["hidden_eos_sum"])) < 1e-2
        # finished rows emit the pad id, the loop runs until the other row is done
        L = g["prompt_len"]
        assert seq_e[0, L + 2] == g["eos"] and (seq_e[0, L + 3:] == 0).all() and seq_e.shape[1] == L + 6
        # all rows finished -> early stop
        both = gen.greedy_generate(sd, cfg, clip[:1], ids0[:1], max_new_tokens=6, eos_token_id=g["eos"], pad_token_id=0)[0]
        assert both.shape[1] == L + 3
        # [SEG] gather: length bookkeeping of LISA.py:497-506
        fake = seq.clone()
        fake[0, L + 1] = cfg.seg_token_idx
        emb = gen.seg_embeddings(sd, cfg, fake, hid)
        for n in range(2):
            assert emb[n].shape == (int((fake[n, 1:] == cfg.seg_token_idx).sum()), cfg.out_dim)
        assert emb[0].shape[0] == emb[1].shape[0] + 1
