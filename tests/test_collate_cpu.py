"""CPU: `llmseg_amd.collate` against the fixture recorded from the imported reference `collate_fn_new`
(oracle/make_goldens.py::gold_collate -> tests/golden/collate.pt; the same stand-in tokenizer on both sides)."""
import pytest
import torch

from llmseg_amd import collate
from oracle import cases
from oracle.stub_tokenizer import StubTokenizer


def _prompt(msgs):
    t = collate.CONV_TEMPLATES["llava_v1"]
    return t.get_prompt([m for q, a in msgs for m in ((t.roles[0], q), (t.roles[1], a))])


def test_prompts_are_the_reference_template(golden):
    g = golden("collate.pt")
    assert cases.collate_conversations(_prompt) == g["conversations"]          # strings recorded from conv_llava_v1.get_prompt()
    assert collate.single_turn_prompt(*cases.COLLATE_QUESTIONS[4]) == g["conversations"][2][0]


@pytest.mark.parametrize("inference", [False, True])
def test_collate_equals_reference_fixture(golden, inference):
    g = golden("collate.pt")
    tok = StubTokenizer(model_max_length=g["model_max_length"])
    out = collate.collate_fn_new(cases.collate_samples(g["conversations"], inference), tokenizer=tok, conv_type="llava_v1", use_mm_start_end=True)
    exp = g["infer" if inference else "train"]
    for k in ("input_ids", "labels", "attention_masks", "offset"):
        assert out[k].dtype == exp[k].dtype and torch.equal(out[k], exp[k]), k
    ids, lab, am = out["input_ids"], out["labels"], out["attention_masks"]
    assert bool(((ids == collate.IMAGE_TOKEN_INDEX).sum(1) == 1).all())         # what make_plan asserts
    assert bool((ids[~am] == tok.pad_token_id).all()) and bool((lab[~am] == collate.IGNORE_INDEX).all())     # unk right-padding, never labelled
    assert bool((lab[:, 0] == collate.IGNORE_INDEX).all())
    keep = lab != collate.IGNORE_INDEX
    assert bool((lab[keep] == ids[keep]).all())                                 # labels are the ids of the answer tokens
    if inference:
        assert ids.shape[1] > tok.model_max_length - 255                       # validation batches are never cut
        assert int((ids == 32000).sum()) == 6
    else:
        assert ids.shape[1] == tok.model_max_length - 255                      # dataset.py:141-148
        assert int((ids[3] == 32000).sum()) == 0                               # the long conversation lost its [SEG] to the cut
    assert out["inference"] is inference and out["offset"].tolist() == [0, 2, 4, 5]
    assert [tuple(s.shape) for s in out["sam_segs_list"]] == [(6, 4, 4), (7, 4, 4), (8, 4, 4)] and out["masks_list"][0].dtype == torch.float32
    assert out["sam_iops_list"] == [None] * 3 if inference else all(t.dtype == torch.float64 for t in out["sam_iops_list"])


def test_plain_conversation_and_no_start_end(golden):
    g = golden("collate.pt")["plain"]
    tok = StubTokenizer()
    out = collate.collate_fn_new(cases.collate_samples(g["conversations"], False), tokenizer=tok, use_mm_start_end=False)
    assert torch.equal(out["input_ids"], g["input_ids"]) and torch.equal(out["labels"], g["labels"])
    assert collate.IMAGE_TOKEN_INDEX not in out["input_ids"]


def test_tokenizer_image_token_and_errors():
    tok = StubTokenizer()
    ids = collate.tokenizer_image_token("a <image> b <image>", tok)
    assert ids[0] == tok.bos_token_id and ids.count(collate.IMAGE_TOKEN_INDEX) == 2 and ids.count(tok.bos_token_id) == 1
    with pytest.raises(ValueError):
        collate.tokenizer_image_token("x", tok, return_tensors="np")
    two_answers = [[_prompt([("<image>\n q", "one ASSISTANT: two")])]]
    with pytest.raises(AssertionError):                                        # the reference asserts one answer per round too
        collate.collate_fn_new(cases.collate_samples(two_answers), tokenizer=tok)


def test_dict_to_cuda_dtypes(golden):
    g = golden("collate.pt")
    tok = StubTokenizer()
    d = collate.dict_to_cuda(collate.collate_fn_new(cases.collate_samples(g["conversations"], False), tokenizer=tok), torch.bfloat16, device="cpu")
    got = {k: str(v.dtype if torch.is_tensor(v) else v[0].dtype) for k, v in d.items()
           if torch.is_tensor(v) or (isinstance(v, list) and v and torch.is_tensor(v[0]))}
    assert got == g["dtypes"]
    kw = collate.model_kwargs(d)
    assert "conversation_list" not in kw and "input_ids" in kw and kw["inference"] is False
