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


class SyntheticSegDataset:
    """A dataset in the reference datasets' own item format (what `collate_fn_new` consumes; utils/reason_seg_dataset.py:127-282 for training items,
    utils/dataset.py:561-656 for validation items), made of seeded noise: item i is a pure function of (seed, i).  For the driver's plumbing
    (`python -m llmseg_amd.run --dataset_module synthetic`) and its tests -- images / proposals / ground truths carry no meaning.
    Every item goes through the real sample builder `collate.reason_seg_sample`: proposal records as COCO RLE + area (the reference's file format,
    prepare_datasets/*), decoded, area-sorted, resampled to 256 x 256 and scored against the ground truths on the device (llmseg_amd/targets.py)."""

    ANSWERS = ("It is [SEG].", "Sure, [SEG].", "Sure, it is [SEG].", "Sure, the segmentation result is [SEG].", "[SEG].")      # utils/utils.py:38-45
    WORDS = ("the", "thing", "that", "matters", "most", "left", "of", "object", "person", "holding", "red", "cup", "near", "window")

    def __init__(self, n, device, img_size=1024, inference=False, seed=0, proposals=12, hw=(96, 128), max_sentences=3):
        self.n, self.device, self.img, self.inference, self.seed, self.K, self.hw, self.max_s = int(n), device, img_size, inference, int(seed), proposals, hw, max_sentences

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        from . import collate, targets
        if not 0 <= i < self.n:
            raise IndexError(i)
        g = torch.Generator().manual_seed(self.seed * 7919 + int(i))
        H, W = self.hw
        thr = torch.rand((self.K, 1, 1), generator=g) * 0.7 + 0.2
        masks = (torch.rand((self.K, H, W), generator=g) > thr).to(torch.uint8)
        recs = [{"segmentation": r, "area": int(mk.sum()), "bbox": [0, 0, 1, k]} for k, (r, mk) in enumerate(zip(targets.rle_encode_masks(masks), masks))]
        n_s = 1 if self.inference else int(torch.randint(1, self.max_s + 1, (1,), generator=g))
        words = lambda: " ".join(self.WORDS[int(j)] for j in torch.randint(0, len(self.WORDS), (5,), generator=g))
        image = torch.randn((3, self.img, self.img), generator=g)
        image_clip = torch.randn((3, 224, 224), generator=g)
        if self.inference:
            gt = (torch.rand((1, H + 3, W + 5), generator=g) > 0.5).to(torch.uint8)
            return collate.reason_seg_sample(image, image_clip, [words() + " "], gt, recs, self.device, inference=True, image_path=f"synthetic_val_{i}.jpg",
                                             resize=(self.img, self.img), top=self.K)
        gt = (torch.rand((n_s, H, W), generator=g) > 0.5).to(torch.uint8)
        qs = [collate.DEFAULT_IMAGE_TOKEN + "\n " + words() + " Please output segmentation mask." for _ in range(n_s)]
        ans = [self.ANSWERS[int(torch.randint(0, len(self.ANSWERS), (1,), generator=g))] for _ in range(n_s)]
        return collate.reason_seg_sample(image, image_clip, qs, gt, recs, self.device, inference=False, answers=ans, image_path=f"synthetic_train_{i}.jpg",
                                         resize=(self.img, self.img), top=self.K)
