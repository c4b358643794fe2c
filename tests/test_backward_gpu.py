"""GPU: backward kernels, autograd Functions and whole-model gradients against torch autograd / the oracle / the fixture."""
import pytest

pytestmark = pytest.mark.gpu


def _assert(res):
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


def test_trainer_eager_and_graph():
    from tests import backward_checks as bc
    _assert(bc.check_trainer_graph_vs_eager())


def test_trainer_configs4_k512_accum8():
    """BASELINE configs[4] as a workload: 512 candidate masks per image, gradient accumulation 8 (one optimizer step = 8 micro-steps),
    hipGraph micro-step, against the fp32 oracle loop."""
    from tests import backward_checks as bc
    res, _ = bc.check_trainer(use_graph=True, opt_steps=2, accum=8, K=512)
    _assert(res)


def test_graph_trainer_with_rotating_batches():
    from tests import backward_checks as bc
    _assert(bc.check_graph_rotating_batches())


def test_micro_step_is_bit_reproducible():
    from tests import backward_checks as bc
    _assert(bc.check_determinism())


def test_checkpoint_save_resume_and_reference_layout(tmp_path):
    from tests import backward_checks as bc
    _assert(bc.check_checkpoint_resume(str(tmp_path)))
