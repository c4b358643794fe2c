"""GPU: every HIP kernel, called through the C ABI, against an fp32 PyTorch / oracle computation of the same op."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"


def _run(fn):
    res = fn()
    from tests._lines import record
    record(res)
    bad = [(n, e, t) for n, e, t in res if not (e <= t)]
    assert not bad, "kernel parity failures: " + "; ".join(f"{n}: err {e:.3e} > tol {t:.3e}" for n, e, t in bad)


def test_gemm():
    from tests import kernel_checks as kc
    _run(kc.check_gemm)


def test_gemm_hot_shapes():
    from tests import kernel_checks as kc
    _run(kc.check_gemm_hot_shapes)


def test_gemm_norm_out():
    from tests import kernel_checks as kc
    _run(kc.check_gemm_norm_out)


def test_gemm_fused_llama_epilogues():
    from tests import kernel_checks as kc
    _run(kc.check_gemm_fx)


def test_gemm_norm_backward_tail():
    from tests import kernel_checks as kc
    _run(kc.check_gemm_normbwd_tail)


def test_gemm_delta_tail():
    from tests import kernel_checks as kc
    _run(kc.check_gemm_delta_tail)


def test_attention():
    from tests import kernel_checks as kc
    _run(kc.check_attention)


def test_head_f32_kernels():
    from tests import kernel_checks as kc
    _run(kc.check_head_f32)


def test_decode_attn():
    from tests import kernel_checks as kc
    _run(kc.check_decode_attn)


def test_pointwise():
    from tests import kernel_checks as kc
    _run(kc.check_pointwise)


def test_head():
    from tests import kernel_checks as kc
    _run(kc.check_head)


def test_metric(golden):
    from tests import kernel_checks as kc
    _run(lambda: kc.check_metric(golden))


def test_validate_body():
    from tests import kernel_checks as kc
    _run(kc.check_validate_body)
