"""Seeded test cases shared by oracle/make_goldens.py (reference side, build container)
and tests/ (oracle + HIP side).  Test infrastructure.

Each case is (config, state-dict seed, inputs) built only from `seeded.uniform`, so a
fixture stores just the expected outputs.
"""
import torch

from . import seeded
from .lisa import LisaCfg
from .llama import LlamaCfg
from .sam_encoder import SamCfg
from .vit import VitCfg

V = 32004
SEG, IM_START, IM_END, IMG = 32000, 32001, 32002, -200


def sam_small_cfg():
    # 2 heads x hd 80; one windowed (14, zero-padding 30->42 exercised) + one global block
    return SamCfg(img=480, patch=16, dim=160, depth=2, heads=2, window=14, global_idx=(1,), out_chans=256)


def sam_small_case(batch=1):
    cfg = sam_small_cfg()
    sd = seeded.fill_state_dict(seeded.sam_shapes(cfg, pfx=""), 11)
    img = seeded.uniform((batch, 3, cfg.img, cfg.img), 12, -2, 2)
    return cfg, sd, img


def tiny_lisa_cfg(backbone="dinov2", lora_r=0):
    return LisaCfg(
        llama=LlamaCfg(hidden=256, inter=512, layers=2, heads=2, vocab=V, lora_r=lora_r),
        clip=VitCfg(dim=64, layers=3, heads=2, mlp=128, patch=14, img=224, eps=1e-5),
        dino=VitCfg(dim=1024, layers=2, heads=16, mlp=4096, patch=14, img=518, eps=1e-6),
        sam=SamCfg(depth=0) if backbone == "dinov2" else SamCfg(dim=160, depth=2, heads=2, global_idx=(1,)),
        backbone=backbone, seg_token_idx=SEG)


def tiny_lisa_state(cfg, seed=3):
    return seeded.fill_state_dict(seeded.lisa_shapes(cfg), seed)


def prompt_ids(n_seq, L, seed, seg_pos=None):
    ids = seeded.uniform((n_seq, L), seed, 3, 31999).long()
    ids[:, 0], ids[:, 1], ids[:, 2], ids[:, 3] = 1, IM_START, IMG, IM_END
    ids[:, (L - 5) if seg_pos is None else seg_pos] = SEG
    return ids


def tiny_lisa_batch(img_size=896, K=16, L=24):
    """2 images, 3 conversations (offset [0,2,3]), one right-padded sequence."""
    B = 2
    ids = prompt_ids(3, L, 21)
    labels = ids.clone()
    labels[:, :10] = -100
    am = torch.ones(3, L, dtype=torch.bool)
    am[1, L - 2:] = False
    return dict(
        images=seeded.uniform((B, 3, img_size, img_size), 22, -2, 2),
        images_clip=seeded.uniform((B, 3, 224, 224), 23, -2, 2),
        input_ids=ids, labels=labels, attention_masks=am, offset=torch.tensor([0, 2, 3]),
        sam_segs_list=[(seeded.uniform((K, 256, 256), 24 + b) > 0.4).float() for b in range(B)],
        sam_ious_list=[seeded.uniform((c, K), 30 + b, 0, 1).double() for b, c in enumerate([2, 1])],
        sam_iops_list=[seeded.uniform((c, K), 40 + b, 0, 1).double() for b, c in enumerate([2, 1])])


def first_image_inference(batch):
    """The validation shape (LISA.py:268-290): one image, one conversation."""
    return dict(images=batch["images"][:1], images_clip=batch["images_clip"][:1], input_ids=batch["input_ids"][:1],
                labels=None, attention_masks=batch["attention_masks"][:1], offset=torch.tensor([0, 1]),
                sam_segs_list=batch["sam_segs_list"][:1])


def loss_inputs():
    return dict(P=seeded.uniform((256, 256), 1), t=seeded.uniform((1, 256), 2), iou=seeded.uniform((256, 1), 3, 0, 1),
                pr=seeded.uniform((256, 1), 4, 0, 1), x=seeded.uniform((3, 64, 64), 5, -3, 3),
                y=(seeded.uniform((3, 64, 64), 6) > 0).float())


def head_case(C=2, K=256, D=256):
    """K = 256: BASELINE configs[1]/[2]; K = 512: configs[4] (the weights are the same seeded head)."""
    sd = seeded.fill_state_dict(seeded.head_shapes(512, D), 51)
    pooled = seeded.uniform((K, D), 52 if K == 256 else 52 + K, -1, 1)
    text = seeded.uniform((C, D), 53, -1, 1)
    return sd, pooled, text


def iou_metric_cases():
    out = []
    for i, (p_thr, ign) in enumerate([(0.0, False), (0.3, True), (2.0, False)]):   # last: empty prediction
        pred = (seeded.uniform((64, 64), 60 + i) > p_thr).long()
        tgt = (seeded.uniform((64, 64), 70 + i) > 0.2).long()
        if ign:
            tgt[seeded.uniform((64, 64), 80 + i) > 0.8] = 255
        out.append((pred, tgt))
    out.append((torch.zeros(64, 64, dtype=torch.long), torch.zeros(64, 64, dtype=torch.long)))   # empty union
    return out


def sam_decoder_state(seed=11):
    """Seeded weights of SAM's prompt encoder (text path) + mask decoder under the reference's state-dict names."""
    from . import sam_decoder as sdec
    sd = seeded.fill_state_dict(sdec.decoder_shapes(), seed)
    k = sdec.PFX + "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"
    sd[k] = seeded.uniform((2, 128), 1234, -1.5, 1.5)                 # the reference draws it from randn (scale 1)
    return sd


def sam_decoder_case(b=2):
    """image embedding [1, 256, 64, 64] (what the SAM encoder's neck emits) and b [SEG] text embeddings [b, 256]."""
    return seeded.uniform((1, 256, 64, 64), 51, -1.0, 1.0), seeded.uniform((b, 256), 52, -1.0, 1.0)


def amg_thresholds():
    """Everything-mode thresholds for the SEEDED decoder (random weights give small logits and arbitrary IoU predictions: the
    reference's defaults 0.88 / 0.95 / offset 1.0 would keep nothing; on `amg_embedding_case` these keep 112 of 192 candidates and 8 records after NMS; the stability filter itself is exercised with 0.08 in the tests)."""
    return dict(pred_iou_thresh=0.1, stability_score_thresh=0.0, stability_score_offset=0.05, box_nms_thresh=0.7)


def amg_embedding_case():
    """A spatially STRUCTURED image embedding [1, 256, 64, 64] (a few Gaussian bumps with per-channel coefficients over a constant background vector):
    with i.i.d. noise every seeded mask covers the whole image and box NMS degenerates to one survivor."""
    ys, xs = torch.meshgrid(torch.arange(64.0), torch.arange(64.0), indexing="ij")
    cen = [(12, 14, 7.0), (20, 48, 9.0), (44, 20, 8.0), (50, 50, 6.0), (32, 32, 14.0), (8, 56, 5.0)]
    bumps = torch.stack([torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / (2 * sg * sg)) for cy, cx, sg in cen], 0)       # [6, 64, 64]
    coef = seeded.uniform((256, len(cen)), 61, -2.0, 2.0)
    emb = torch.einsum("ck,kyx->cyx", coef, bumps) + seeded.uniform((256, 1, 1), 62, -1.0, 1.0)       # + a constant background vector
    return emb[None]


def amg_image_case(h=300, w=400):
    """A seeded uint8 RGB image [h, w, 3] (numpy) for everything mode with crop layers: coloured Gaussian blobs over a grey gradient + mild
    per-pixel noise, so that crops of it are images with structure of their own."""
    import numpy as np
    ys, xs = torch.meshgrid(torch.arange(float(h)), torch.arange(float(w)), indexing="ij")
    img = torch.stack([90 + 40 * xs / w, 100 + 30 * ys / h, 110 - 30 * xs / w], -1)
    blobs = [(0.22, 0.20, 0.09, (150, -60, -40)), (0.30, 0.72, 0.12, (-70, 120, -30)), (0.70, 0.30, 0.11, (-50, -40, 140)), (0.78, 0.80, 0.07, (120, 110, -80)),
             (0.50, 0.50, 0.05, (-90, -90, -90)), (0.12, 0.90, 0.05, (100, -20, 100)), (0.90, 0.08, 0.06, (60, 130, 60))]
    for cy, cx, sg, col in blobs:
        g = torch.exp(-(((ys - cy * h) / (sg * h)) ** 2 + ((xs - cx * w) / (sg * h)) ** 2) / 2)
        img = img + g[..., None] * torch.tensor(col, dtype=torch.float32)
    img = img + seeded.uniform((h, w, 3), 71, -6.0, 6.0)
    return img.clamp(0, 255).round().to(torch.uint8).numpy()


def amg_standin_encoder():
    """A seeded stand-in for the SAM image encoder in the crop-layer tests (the encoder itself is row A6, tested elsewhere): the preprocessed
    frame [1, 3, 1024, 1024] -> [1, 256, 64, 64] = a fixed 256 x 9 map of 16 x 16-pooled colour features (r, g, b, their products and squares)
    + a constant vector.  The SAME function stands in for the encoder of the imported reference generator and of the restatement."""
    coef = seeded.uniform((256, 9), 72, -1.2, 1.2)
    bias = seeded.uniform((256, 1, 1), 73, -1.0, 1.0)

    def encode(x):
        p = torch.nn.functional.avg_pool2d(x.float(), 16)[0]                       # [3, 64, 64]
        r, g, b = p[0], p[1], p[2]
        f = torch.stack([r, g, b, r * g, g * b, b * r, r * r, g * g, b * b], 0)
        return (torch.einsum("ck,kyx->cyx", coef, f) + bias)[None]
    return encode


# ------------------------------------------------------------------------------------------------ A14: collate / prompt contract
COLLATE_QUESTIONS = [
    ("<image>\n What is the dog in this image? Please output segmentation mask.", "Sure, [SEG]."),
    ("<image>\n the left chair Please respond with segmentation mask.", "It is [SEG]."),
    ("<image>\n Can you segment the red car parked next to the tree in this image?", "Sure, the segmentation result is [SEG]."),
    ("<image>\n " + " ".join(f"word{i} and then something number {i}," for i in range(60)) + " which object is it?", "[SEG]."),   # > 512 - 255 tokens: truncated in training
    ("<image>\n the cup", "Sure, it is [SEG]."),
]


def collate_conversations(prompt):
    """Conversation strings of 3 images: image 0 carries two single-turn conversations, image 1 ONE conversation of two rounds (two [SEG],
    the second round without <image>) followed by a long single-turn one (hits the training truncation), image 2 a short one (right padding).
    `prompt(messages)`: [(question, answer), ...] -> the conversation string (the reference's `conv.get_prompt()` or this package's template)."""
    q = COLLATE_QUESTIONS
    return [[prompt([q[0]]), prompt([q[1]])],
            [prompt([q[2], ("And where is the other one?", "It is [SEG].")]), prompt([q[3]])],
            [prompt([q[4]])]]


def collate_samples(conversations, inference=False, K=6, img=16, clip=8, seg=4, ragged=True):
    """Sample dicts in the datasets' format (`utils/reason_seg_dataset.py:266-282`; validation: `utils/dataset.py:640-656`, which has no
    'iops' / 'questions') with tiny seeded tensors: what is under test is the collate, not the image pipeline."""
    out = []
    for b, convs in enumerate(conversations):
        C = len(convs)
        Kb = K + b if ragged else K
        d = {"image_path": f"img{b}.jpg", "images": seeded.uniform((3, img, img), 200 + b, -2, 2), "images_clip": seeded.uniform((3, clip, clip), 210 + b, -2, 2),
             "conversations": convs, "masks": (seeded.uniform((C, 5, 7), 220 + b) > 0).to(torch.uint8), "label": torch.ones(5, 7) * 255,
             "resize": (12, 16), "segs": seeded.uniform((Kb, seg, seg), 230 + b, 0, 1), "inference": inference,
             "segs_origin": None, "bbox": None}
        if inference:
            d.update(questions=None, sampled_classes=None, ious=None)
        else:
            d.update(questions=[f"q{b}{c}" for c in range(C)], sampled_classes=[f"c{b}{c}" for c in range(C)],
                     ious=seeded.uniform((C, Kb), 240 + b, 0, 1).double(), iops=seeded.uniform((C, Kb), 250 + b, 0, 1).double())
        out.append(d)
    return out


# ------------------------------------------------------------------------------------------------ N2: the reference's own target functions
def target_case(n_masks=57, h=45, w=70, gt_hw=(61, 97)):
    """Seeded binary proposals [n, h, w] with distinct-ish areas (some ties: the sort must be stable) + a ground truth at another resolution."""
    import numpy as np
    ys, xs = torch.meshgrid(torch.arange(float(h)), torch.arange(float(w)), indexing="ij")
    cy, cx = seeded.uniform((n_masks,), 301, 0, h), seeded.uniform((n_masks,), 302, 0, w)
    ry, rx = seeded.uniform((n_masks,), 303, 2, h / 2), seeded.uniform((n_masks,), 304, 2, w / 2)
    m = (((ys[None] - cy[:, None, None]) / ry[:, None, None]) ** 2 + ((xs[None] - cx[:, None, None]) / rx[:, None, None]) ** 2) <= 1.0
    m[5] = m[4]; m[20] = m[19]                                                     # area ties
    m[9] = False                                                                   # an empty proposal: IoU 0 / IoP 0/0 = nan
    gy, gx = torch.meshgrid(torch.arange(float(gt_hw[0])), torch.arange(float(gt_hw[1])), indexing="ij")
    gt = (((gy - 30) / 14) ** 2 + ((gx - 50) / 25) ** 2 <= 1.0)
    return m.to(torch.uint8).numpy(), gt.to(torch.uint8).numpy()
