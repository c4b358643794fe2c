"""A deterministic stand-in for the Llama tokenizer the reference loads (`training.py:121-135`: `AutoTokenizer.from_pretrained(...,
model_max_length=512, padding_side="right", use_fast=False)`, `pad_token = unk_token`, `[SEG]` / `<im_start>` / `<im_end>` added).
Test infrastructure: there is no tokenizer model offline, so the SAME class is handed to the imported reference `collate_fn_new`
(oracle/make_goldens.py::gold_collate) and to `llmseg_amd.collate` -- what the fixture pins is the collate arithmetic, not sentencepiece
(PARITY UNPINNED for the tokenization itself).

It keeps the two properties of the sentencepiece tokenizer that the collate's label arithmetic leans on (`utils/dataset.py:111-121`):
  * every call prepends `bos`;
  * a space belongs to the word that follows it, and a TRAILING space is a token of its own -- so the tokens of "... ASSISTANT: " are the
    prefix of the tokens of "... ASSISTANT: Sure" plus one (the reference's `instruction_len = len(...) - 2`).
Added tokens and `</s>` are single ids; everything else is hashed (CRC-32) into [3, 31999)."""
import re
import types
import zlib

_SPECIAL = {"</s>": 2, "<s>": 1, "[SEG]": 32000, "<im_start>": 32001, "<im_end>": 32002}
_UNIT = re.compile(r" ?[A-Za-z0-9]+| ?[^A-Za-z0-9\s]|\s")


class StubTokenizer:
    bos_token_id = 1
    eos_token_id = 2
    unk_token_id = 0
    pad_token_id = 0                      # training.py:129: pad_token = unk_token

    def __init__(self, model_max_length=512, specials=None):
        self.model_max_length = model_max_length
        self.specials = dict(_SPECIAL if specials is None else specials)

    def _ids(self, text):
        """Added tokens are cut out first (as the slow HF tokenizer does), the text between them is tokenised on its own."""
        out = []
        cut = re.compile("(" + "|".join(re.escape(k) for k in sorted(self.specials, key=len, reverse=True)) + ")")
        for chunk in cut.split(text):
            if chunk in self.specials:
                out.append(self.specials[chunk])
            else:
                out += [self._hash(u) for u in _UNIT.findall(chunk)]
        return out

    @staticmethod
    def _hash(u):
        return 3 + zlib.crc32(u.encode("utf-8")) % (31999 - 3)

    def __call__(self, text, add_special_tokens=True, **_):
        ids = ([self.bos_token_id] if add_special_tokens else []) + self._ids(text)
        return types.SimpleNamespace(input_ids=ids)
