"""GPU: the generation path of evaluate()."""
import pytest

pytestmark = pytest.mark.gpu


def _assert(res):
    bad = [(n, e, t) for n, e, t in res if not (e <= t)]
    assert not bad, "; ".join(f"{n}: err {e:.3e} > tol {t:.3e}" for n, e, t in bad)


def test_kv_cache_greedy_generation_matches_oracle():
    from tests import generate_checks as gc
    _assert(gc.check_generate())


def test_kv_cache_generation_with_merged_lora_matches_oracle():
    from tests import generate_checks as gc
    _assert(gc.check_generate(lora_r=8))


def test_sam_mask_decoder_and_postprocess_match_oracle():
    from tests import sam_decoder_checks as sc
    _assert(sc.check_sam_decoder())


def test_evaluate_end_to_end_matches_oracle():
    from tests import sam_decoder_checks as sc
    _assert(sc.check_evaluate())


def test_everything_mode_proposals_match_oracle():
    from tests import amg_checks as ac
    _assert(ac.check_amg())


def test_everything_mode_crop_layers_match_oracle():
    from tests import amg_checks as ac
    _assert(ac.check_amg_crops())
