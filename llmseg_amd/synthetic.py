"""Synthetic `model_forward` batches of the shapes BASELINE.json benchmarks, generated on the device.

Mirrors the kwargs `collate_fn_new` hands to the model (reference `utils/dataset.py:33-170`): one conversation per
image, 64-token prompt with `<im_start> <image> <im_end>` at positions 1..3 and `[SEG]` near the end.
"""
import torch

SEG, IM_START, IM_END, IMG = 32000, 32001, 32002, -200


def make_batch(B, img_size=1024, L=64, K=256, seg_size=256, device="cuda", seed=1234, vocab=32004, soft=False, convs=None):
    """convs: conversations per image (list of B ints; default one each) -- the datasets emit one conversation per sampled class / sentence of
    an image (`num_classes_per_sample` = 3, e.g. utils/sem_seg_dataset.py, utils/refer_seg_dataset.py), all sharing the image."""
    if convs is not None:
        return _make_multi(B, img_size, L, K, seg_size, device, seed, soft, list(convs))
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ids = torch.randint(3, 31999, (B, L), device=device, generator=g)
    ids[:, 0], ids[:, 1], ids[:, 2], ids[:, 3] = 1, IM_START, IMG, IM_END
    ids[:, L - 3] = SEG
    labels = ids.clone()
    labels[:, : L // 2] = -100
    segs = []
    for _ in range(B):
        r = torch.rand((K, seg_size, seg_size), device=device, generator=g)
        segs.append(r.to(torch.bfloat16) if soft else (r > 0.7).to(torch.bfloat16))
    return dict(
        images=torch.randn((B, 3, img_size, img_size), device=device, generator=g).to(torch.bfloat16),
        images_clip=torch.randn((B, 3, 224, 224), device=device, generator=g).to(torch.bfloat16),
        input_ids=ids, labels=labels, attention_masks=torch.ones((B, L), dtype=torch.bool, device=device),
        offset=torch.arange(B + 1, device=device),
        masks_list=[None] * B, label_list=[None] * B, resize_list=[None] * B,
        sam_segs_list=segs,
        sam_ious_list=[torch.rand((1, K), device=device, generator=g) for _ in range(B)],
        sam_iops_list=[torch.rand((1, K), device=device, generator=g) for _ in range(B)])


def _make_multi(B, img_size, L, K, seg_size, device, seed, soft, convs):
    assert len(convs) == B and all(c >= 1 for c in convs)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    N = sum(convs)
    ids = torch.randint(3, 31999, (N, L), device=device, generator=g)
    ids[:, 0], ids[:, 1], ids[:, 2], ids[:, 3] = 1, IM_START, IMG, IM_END
    ids[:, L - 3] = SEG
    labels = ids.clone()
    labels[:, : L // 2] = -100
    segs = []
    for _ in range(B):
        r = torch.rand((K, seg_size, seg_size), device=device, generator=g)
        segs.append(r.to(torch.bfloat16) if soft else (r > 0.7).to(torch.bfloat16))
    off = [0]
    for c in convs:
        off.append(off[-1] + c)
    return dict(
        images=torch.randn((B, 3, img_size, img_size), device=device, generator=g).to(torch.bfloat16),
        images_clip=torch.randn((B, 3, 224, 224), device=device, generator=g).to(torch.bfloat16),
        input_ids=ids, labels=labels, attention_masks=torch.ones((N, L), dtype=torch.bool, device=device),
        offset=torch.tensor(off, device=device),
        masks_list=[None] * B, label_list=[None] * B, resize_list=[None] * B,
        sam_segs_list=segs,
        sam_ious_list=[torch.rand((c, K), device=device, generator=g) for c in convs],
        sam_iops_list=[torch.rand((c, K), device=device, generator=g) for c in convs])


class HybridSampler:
    """Which source the next sample comes from: `HybridDataset.__getitem__` (reference utils/dataset.py:499-502) draws a dataset index with
    probability sample_rate / sum(sample_rate) and ignores `idx` (`--dataset sem_seg||refer_seg||reason_seg --sample_rates 9,3,1`, training.py:66-70;
    BASELINE configs[3]).  The datasets themselves are out of scope; what the sources differ in on the device path is the number of conversations an
    image carries: `num_classes_per_sample` = 3 classes of a sem_seg image, 1-3 sentences of a refer_seg image, 1-2 of a reason_seg image."""

    SOURCES = ("sem_seg", "refer_seg", "reason_seg")

    def __init__(self, rates=(9, 3, 1), seed=0):
        import numpy as np
        self.p = np.asarray(rates, dtype=np.float64) / float(sum(rates))
        self.rng = np.random.default_rng(seed)

    def draw(self):
        """-> (source name, conversations of the sampled image)."""
        i = int(self.rng.choice(len(self.p), p=self.p))
        src = self.SOURCES[i]
        c = 3 if src == "sem_seg" else int(self.rng.integers(1, 4)) if src == "refer_seg" else int(self.rng.integers(1, 3))
        return src, c
