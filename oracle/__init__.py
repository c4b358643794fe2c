"""CPU oracle for the LLM-Seg `model_forward` hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `llmseg_amd/` may import this package.
Allowed importers: `tests/`, `__graft_entry__.smoke()`, and the `cpu_baseline`
leg of `bench.py` (where it is the thing timed as the CPU baseline, never the
product path).

What it is: a from-scratch, *functional* pure-PyTorch restatement (flat
state-dict in, tensors out) of the arithmetic the reference runs on the path
`LISAForCausalLM.model_forward` (reference `model/LISA.py:225-474`).  Every
function cites the reference file:line it follows.  State-dict key names are
the reference's, so a reference checkpoint maps 1:1.

How it is pinned: the reference ships NO tests or golden
vectors, so the pin is "outputs of the reference itself run here":
`oracle/ref_harness.py` imports `/root/reference` (read-only, build container
only) behind a set of import shims, `oracle/make_goldens.py` runs reference
modules and this restatement on the same seeded inputs/weights, asserts they
agree, and writes small fixtures to `tests/golden/`.  Pinned that way:
SAM ViT-H encoder, mask-selection head, losses, LLaVA splice + CE, and the
end-to-end `model_forward` (train + inference).  The Llama / CLIP blocks are
third-party (`transformers==4.29.0` in the reference; 5.15.0 installed here) and
are pinned against the installed eager implementation.

Parity UNPINNED (third-party code absent from /root/reference *and* from this
image): LoRA (`peft==0.4.0`), DINOv2 hub code (`facebookresearch/dinov2`,
no commit pin; HF `Dinov2Model` is the structural stand-in), DeepSpeed
AdamW/WarmupDecayLR.  Their published algorithms are restated and marked so.
"""
