"""Synthetic `model_forward` batches of the shapes BASELINE.json benchmarks, generated on the device.

Mirrors the kwargs `collate_fn_new` hands to the model (reference `utils/dataset.py:33-170`): one conversation per
image, 64-token prompt with `<im_start> <image> <im_end>` at positions 1..3 and `[SEG]` near the end.
"""
import torch

SEG, IM_START, IM_END, IMG = 32000, 32001, 32002, -200


def make_batch(B, img_size=1024, L=64, K=256, seg_size=256, device="cuda", seed=1234, vocab=32004, soft=False):
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
