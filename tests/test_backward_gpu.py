"""GPU: backward kernels, autograd Functions and whole-model gradients against torch autograd / the oracle / the fixture."""
import pytest

pytestmark = pytest.mark.gpu


def _assert(res):
    from tests._lines import record
    record(res)
    bad = [(n, e, t) for n, e, t in res if not (e <= t)]
    assert not bad, "; ".join(f"{n}: err {e:.3e} > tol {t:.3e}" for n, e, t in bad)


def test_gemm_layouts():
    from tests import backward_checks as bc
    _assert(bc.check_gemm_layouts())


def test_autograd_ops():
    from tests import backward_checks as bc
    _assert(bc.check_autograd_ops())


def test_lora_paths():
    from tests import backward_checks as bc
    _assert(bc.check_lora_paths())


def test_arena_ops():
    from tests import backward_checks as bc
    _assert(bc.check_arena_ops())


def test_adamw():
    from tests import backward_checks as bc
    _assert(bc.check_adamw())


def test_model_grads(golden):
    from tests import backward_checks as bc
    _assert(bc.check_model_grads(golden))


def test_model_grads_lora_arena():
    from tests import backward_checks as bc
    _assert(bc.check_model_grads_lora("sam"))


def test_full_depth_configs2_gradients():
    """BASELINE configs[2] at its own depth: 32-layer Llama-7B (LoRA r 8 + dropout) + CLIP-L + 32-block SAM ViT-H, B = 2, K = 256, one
    micro-step into the fp32 arena vs autograd through the oracle on the host.  Writes the per-tensor ratio table to
    gpurun_out/fulldepth_grads.md (copied to profiles/)."""
    import os
    free_gb = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2 ** 30
    if free_gb < 128:
        pytest.skip(f"host has {free_gb:.0f} GB free; the full-depth fp32 oracle with autograd wants ~90 GB")
    from tests import fulldepth_checks as fc
    table, logs = [], []
    res = fc.check_full_depth_gradients(table=table, log=lambda s: (print(s), logs.append(s)))
    for n, e, t in res:
        print(f"{n}: err {e:.3e} tol {t:.3e}")
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "fulldepth_grads.md"), "w") as fh:
            fh.write("# Full-depth fwd+bwd gradient parity (BASELINE configs[2]: 32-layer Llama-7B + LoRA r 8 + dropout 0.05, CLIP-L, SAM ViT-H, B = 2, K = 256)\n\n"
                     "HIP fp32 arena after ONE micro-step vs autograd through the fp32 oracle on the host; yardstick = the same oracle in bf16 on the CPU.\n\n"
                     + "\n".join(f"- {s}" for s in logs) + "\n\n"
                     "| tensor | rms(ref) | RMS err HIP | RMS err bf16-CPU | ratio | max err HIP | max err bf16-CPU | ratio | flipped-gate rows excluded | err / tol |\n|---|---|---|---|---|---|---|---|---|---|\n"
                     + "\n".join(table) + "\n\n" + "\n".join(f"- {n}: {e:.3e} (tol {t:.3e})" for n, e, t in res if "arena grad" not in n or "quantile" in n or "median" in n) + "\n")
    except OSError:
        pass
    _assert(res)


def test_trainer_eager_and_graph():
    from tests import backward_checks as bc
    _assert(bc.check_trainer_graph_vs_eager())


def test_trainer_configs4_k512_accum8():
    """BASELINE configs[4] as a workload: 512 candidate masks per image, gradient accumulation 8 (one optimizer step = 8 micro-steps),
    hipGraph micro-step, against the fp32 oracle loop."""
    from tests import backward_checks as bc
    res, _ = bc.check_trainer(use_graph=True, opt_steps=2, accum=8, K=512)
    _assert(res)


def test_fused_accumulation_window_equals_micro_steps():
    """One fused pass over the k micro-batches of an optimizer step == k micro-steps (arena, losses, dropout stream)."""
    from tests import backward_checks as bc
    _assert(bc.check_fused_accum(3))


def test_trainer_configs3_mix_9_3_1_batch1_window():
    """BASELINE configs[3]'s per-GPU workload: batch 1, source drawn 9:3:1 -> 1-3 conversations per image (three batch structures, three
    hipGraphs), one accumulation window of 10 micro-steps against the oracle; graph == eager bit for bit."""
    from tests import backward_checks as bc
    _assert(bc.check_mix_window())


def test_window_step_with_batched_frozen_towers():
    """The frozen towers once per accumulation window (`Trainer.window_step`): exact plumbing, window pass vs micro-steps, graph + prefetch."""
    from tests import backward_checks as bc
    _assert(bc.check_window_towers(3))


def test_backward_cut_at_the_llama_output_for_the_exchange_overlap():
    from tests import backward_checks as bc
    _assert(bc.check_overlap_exchange())


def test_graph_trainer_with_rotating_batches():
    from tests import backward_checks as bc
    _assert(bc.check_graph_rotating_batches())


def test_micro_step_is_bit_reproducible():
    from tests import backward_checks as bc
    _assert(bc.check_determinism())


def test_checkpoint_save_resume_and_reference_layout(tmp_path):
    from tests import backward_checks as bc
    _assert(bc.check_checkpoint_resume(str(tmp_path)))
