"""The collate / prompt contract that produces `model_forward`'s kwargs: reference `collate_fn_new`
(`utils/dataset.py:33-170`), `tokenizer_image_token` (`model/llava/mm_utils.py:19-44`), `dict_to_cuda` (`utils/utils.py:157-171`) and the
single-turn prompt the datasets build through `conversation_lib` (`model/llava/conversation.py:31-62,355-365`, e.g.
`utils/reason_seg_dataset.py:245-256`, `utils/dataset.py:570-592`).  Same function names, argument meaning, return keys and error behaviour.

Host code by nature (string handling + a tokenizer the caller injects -- the reference's is the LLaVA sentencepiece tokenizer with
`pad_token = unk_token`, `training.py:121-135`); nothing here touches the device except `dict_to_cuda`.  What the device path needs from it
(`TrainableMixin.make_plan`): `input_ids` with exactly one -200 per sequence, right padding with the pad (= unk) id and `attention_masks`
False there, labels -100 over system + user turns, padding and everything past the last answer, `offset` = conversations per image
(prefix sums), sequences cut to `model_max_length - 255` when training (the 255 extra rows the image tokens add must still fit).
"""
from dataclasses import dataclass

import torch

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"


@dataclass(frozen=True)
class ConvTemplate:
    """The fields of a `conversation_lib.Conversation` that the datasets and the collate read (conversation.py:17-28)."""
    system: str
    roles: tuple
    sep: str
    sep2: str

    def get_prompt(self, messages):
        """`SeparatorStyle.TWO` (conversation.py:53-62): system + sep, then "ROLE: message" + sep / sep2 alternating; an empty message
        leaves "ROLE:" open (generation prompts)."""
        out = self.system + self.sep
        for i, (role, msg) in enumerate(messages):
            out += f"{role}: {msg}{(self.sep, self.sep2)[i % 2]}" if msg else f"{role}:"
        return out


CONV_TEMPLATES = {
    # conversation.py:355-365 (`conv_llava_v1`, what `--conv_type llava_v1` selects, training.py:178-180)
    "llava_v1": ConvTemplate(
        system="A chat between a curious human and an artificial intelligence assistant. "
               "The assistant gives helpful, detailed, and polite answers to the human's questions.",
        roles=("USER", "ASSISTANT"), sep=" ", sep2="</s>"),
}


def single_turn_prompt(question, answer, conv_type="llava_v1"):
    """One (question, answer) conversation as the datasets emit it (`conv.messages = []; append_message(roles[0], q);
    append_message(roles[1], a); conv.get_prompt()`)."""
    t = CONV_TEMPLATES[conv_type]
    return t.get_prompt([(t.roles[0], question), (t.roles[1], answer)])


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    """mm_utils.py:19-44: tokenise the text around every "<image>" and put `image_token_index` in between.  Each chunk is tokenised on its own
    (so each carries the tokenizer's `bos`); the result keeps ONE leading `bos` when the first chunk has one."""
    chunks = [tokenizer(c).input_ids for c in prompt.split(DEFAULT_IMAGE_TOKEN)]
    has_bos = bool(chunks) and len(chunks[0]) > 0 and chunks[0][0] == tokenizer.bos_token_id
    skip = 1 if has_bos else 0
    ids = [chunks[0][0]] if has_bos else []
    for i, c in enumerate(chunks):
        if i > 0:
            ids.append(image_token_index)            # (the reference inserts [index] * (skip + 1) and drops the first `skip` of it)
        ids.extend(c[skip:])
    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")


def _unlabelled_spans(conversation, tokenizer, round_sep, answer_sep):
    """Token spans [start, stop) of one conversation that carry no label, and the token count the walk ends on (dataset.py:103-126).
    Rounds are the pieces between `round_sep` ("</s>"); inside a round everything up to and including `answer_sep` (" ASSISTANT: ") is the
    instruction.  Lengths are re-tokenised per piece: a round costs len(tokens(round)) positions (its `bos` stands in for the "</s>" that the
    split removed), an instruction len(tokens(instruction)) - 2 (minus `bos`, minus the trailing-space token that fuses with the answer's
    first word in the full string)."""
    if DEFAULT_IMAGE_TOKEN in conversation:
        count = lambda s: len(tokenizer_image_token(s, tokenizer))
    else:
        count = lambda s: len(tokenizer(s).input_ids)
    spans, pos = [(0, 1)], 1                                    # bos
    for rnd in conversation.split(round_sep):
        if rnd == "":
            break
        pieces = rnd.split(answer_sep)
        assert len(pieces) == 2, (len(pieces), rnd)             # the reference asserts the same: one answer per round
        spans.append((pos, pos + count(pieces[0] + answer_sep) - 2))
        pos += count(rnd)
    return spans, pos


def collate_fn_new(batch, tokenizer=None, conv_type="llava_v1", use_mm_start_end=True, local_rank=-1):
    """dataset.py:33-170.  `batch`: list of sample dicts (keys as the datasets return them: image_path, images, images_clip, conversations,
    masks, label, resize, questions, sampled_classes, segs, ious, iops, inference, optional segs_origin / bbox) -> the kwargs of
    `model_forward` plus the pass-through lists."""
    get = lambda key, default=None: [d.get(key, default) for d in batch]
    conversation_list = [c for d in batch for c in d.get("conversations", [])]
    offset, n = [0], 0
    for d in batch:
        n += len(d.get("conversations", []))
        offset.append(n)
    if use_mm_start_end:
        tag = DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN
        conversation_list = [c.replace(DEFAULT_IMAGE_TOKEN, tag) for c in conversation_list]

    pad = tokenizer.pad_token_id
    seqs = [tokenizer_image_token(c, tokenizer, return_tensors="pt") for c in conversation_list]
    input_ids = torch.nn.utils.rnn.pad_sequence(seqs, batch_first=True, padding_value=pad)
    attention_masks = input_ids.ne(pad)                        # NB: a genuine unk token inside a prompt is masked too (pad = unk), as in the reference
    targets = input_ids.clone()

    conv = CONV_TEMPLATES.get(conv_type, CONV_TEMPLATES["llava_v1"])
    answer_sep = conv.sep + conv.roles[1] + ": " if conv_type == "llava_v1" else "[/INST] "
    for conversation, target in zip(conversation_list, targets):
        total_len = int(target.ne(pad).sum())
        spans, end = _unlabelled_spans(conversation, tokenizer, conv.sep2, answer_sep)
        for a, b in spans:
            target[a:b] = IGNORE_INDEX
        target[end:] = IGNORE_INDEX
        if end < tokenizer.model_max_length:
            assert end == total_len, (end, total_len)          # the per-piece token counts must add up to the whole (dataset.py:138-139)

    inferences = get("inference")
    if inferences[0] == False:                                  # noqa: E712 -- `None` must NOT truncate (the reference compares with ==)
        keep = tokenizer.model_max_length - 255
        if input_ids.shape[1] > keep:
            input_ids, targets, attention_masks = input_ids[:, :keep], targets[:, :keep], attention_masks[:, :keep]

    return {
        "image_paths": get("image_path"),
        "images": torch.stack(get("images"), dim=0),
        "images_clip": torch.stack(get("images_clip"), dim=0),
        "input_ids": input_ids,
        "labels": targets,
        "attention_masks": attention_masks,
        "masks_list": [d.get("masks").float() for d in batch],
        "label_list": get("label"),
        "resize_list": get("resize"),
        "offset": torch.LongTensor(offset),
        "questions_list": get("questions"),
        "sampled_classes_list": get("sampled_classes"),
        "inference": inferences[0],
        "conversation_list": conversation_list,
        "sam_segs_list": get("segs"),
        "sam_ious_list": get("ious"),
        "sam_iops_list": get("iops"),
        "origin_segs_list": get("segs_origin"),
        "bbox_list": get("bbox"),
    }


def dict_to_cuda(input_dict, torch_dtype=torch.bfloat16, device="cuda"):
    """utils.py:157-171: tensors and lists of tensors to the device; `images` / `images_clip` and the proposal maps in the training dtype,
    everything else (ids, masks, float64 IoU / IoP targets) keeps its dtype."""
    for k, v in input_dict.items():
        if isinstance(v, torch.Tensor):
            v = v.to(device, non_blocking=True)
            input_dict[k] = v.to(dtype=torch_dtype) if k in ("images", "images_clip") else v
        elif isinstance(v, list) and len(v) > 0 and isinstance(v[0], torch.Tensor):
            v = [t.to(device, non_blocking=True) for t in v]
            input_dict[k] = [t.to(dtype=torch_dtype) for t in v] if k == "sam_segs_list" else v
    return input_dict


MODEL_FORWARD_KEYS = ("images", "images_clip", "input_ids", "labels", "attention_masks", "offset", "masks_list", "label_list", "resize_list",
                      "sam_segs_list", "sam_ious_list", "sam_iops_list", "inference")


def model_kwargs(collated):
    """The entries of a collated dict that `model_forward` takes by name (the reference passes the whole dict and lets `**kwargs` swallow the
    rest, `training.py:546`; the string lists would defeat a hipGraph key)."""
    return {k: collated[k] for k in MODEL_FORWARD_KEYS if k in collated}


def reason_seg_sample(image, image_clip, sents, gt_masks, proposal_records, device, inference, is_sentence=True, answers=None, image_path="",
                      resize=None, top=50, ignore_label=255):
    """One sample dict in the datasets' format from already-decoded inputs: `ValDataSet_ReasonSeg.__getitem__` (utils/dataset.py:561-656,
    `inference=True`: one "[SEG]." conversation per sentence, no IoU targets) or `ReasonSegDataset.__getitem__` (utils/reason_seg_dataset.py:127-282,
    `inference=False`: answers from the caller, IoU / IoP targets per sampled mask).  The image pipeline (cv2 decode, ResizeLongestSide, CLIP
    processor) stays outside: `image` [3, S, S] / `image_clip` [3, 224, 224] arrive preprocessed.  The proposal work -- area sort, top-`top`, RLE
    decode, pad to square, antialiased resize to 256 x 256, IoU / IoP against every ground truth -- runs on `device` (llmseg_amd/targets.py, N2).
    gt_masks: uint8 [C, H, W] (one per sentence when training; validation passes the image's single mask)."""
    from . import targets
    gt_masks = torch.as_tensor(gt_masks)
    if inference:
        tmpl = "\n {} Please output segmentation mask." if is_sentence else "\n What is {} in this image? Please output segmentation mask."
        questions = [DEFAULT_IMAGE_TOKEN + tmpl.format(s.strip()) for s in sents]
        answers = ["[SEG]."] * len(sents)
    else:
        questions = list(sents)                                   # training: the caller has already drawn the question templates
        assert answers is not None and len(answers) == len(questions)
    t = targets.proposals_and_targets(proposal_records, [] if inference else list(gt_masks), device, top=top)
    d = {"image_path": image_path, "images": image, "images_clip": image_clip,
         "conversations": [single_turn_prompt(q, a) for q, a in zip(questions, answers)],
         "masks": gt_masks, "label": torch.ones(gt_masks.shape[1], gt_masks.shape[2]) * ignore_label, "resize": resize,
         "questions": None if inference else questions, "sampled_classes": None if inference else list(sents),
         "segs": t["sam_segs"], "ious": None if inference else t["sam_ious"], "inference": inference,
         "segs_origin": t["segs_origin"].permute(1, 2, 0).contiguous() if inference else None,            # the reader's [H, W, K] layout
         "bbox": t["bbox"] if inference else None}
    if not inference:
        d["iops"] = t["sam_iops"]
    return d
