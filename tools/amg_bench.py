"""Everything-mode timing on one MI355X: 32 x 32 point grid, random decoder weights (thresholds chosen so that a few
hundred candidates survive the filters, as with real weights), 1024 x 1024 original image.
usage: python tools/amg_bench.py [points_per_batch=256]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llmseg_amd.lisa import LISAForCausalLM  # noqa: E402
from llmseg_amd.params import LisaConfig, LlamaConfig, SamConfig, VitConfig  # noqa: E402

ppb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
cfg = LisaConfig(backbone="sam", build_unused_towers=False, sam_decoder=True)
cfg.llama = LlamaConfig(layers=1)
cfg.clip = VitConfig(layers=1)
m = LISAForCausalLM(cfg, device=dev).init_random(seed=0)
m.prepare()
img = torch.randn(1, 3, 1024, 1024, device=dev).to(torch.bfloat16)
torch.cuda.synchronize(); t0 = time.perf_counter()
feats = m._sam_encoder_cl(img)
torch.cuda.synchronize(); t_enc = time.perf_counter() - t0
torch.cuda.synchronize(); t0 = time.perf_counter()
feats = m._sam_encoder_cl(img)
torch.cuda.synchronize(); t_enc = time.perf_counter() - t0
kw = dict(points_per_side=32, points_per_batch=ppb, pred_iou_thresh=-1e9, stability_score_thresh=0.0, stability_score_offset=0.02, box_nms_thresh=0.7)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m.generate_proposals(feats[:4096], (1024, 1024), (1024, 1024), **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"run {it}: everything mode {dt * 1e3:.1f} ms (+ SAM ViT-H encoder {t_enc * 1e3:.1f} ms) for 1024 points / 3072 candidates, {out['masks'].shape[0]} records, "
          f"points_per_batch {ppb}; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")

# beyond the default configuration: one crop layer (5 crops, 32 x 32 + 4 x 16 x 16 points) and the small-region clean-up, from a uint8 image
if len(sys.argv) > 2 and sys.argv[2] == "crops":
    u8 = (torch.rand(1024, 1024, 3, device=dev) * 255).to(torch.uint8)
    for tag, extra in (("set_image only (resize + preprocess + ViT-H)", None), ("crop_n_layers=1", dict(crop_n_layers=1, crop_n_points_downscale_factor=2)),
                       ("crop_n_layers=1 + min_mask_region_area=100", dict(crop_n_layers=1, crop_n_points_downscale_factor=2, min_mask_region_area=100))):
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if extra is None:
                m.set_image(u8, (100, 50, 700, 500)); k = 0
            else:
                k = m.generate_masks(u8, **kw, **extra)["masks"].shape[0]
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{tag}: {dt * 1e3:.1f} ms, {k} records; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    K = 256
    masks = (torch.rand(K, 1024, 1024, device=dev) > 0.45).to(torch.uint8)
    from llmseg_amd import ops  # noqa: E402
    for it in range(2):
        mm = masks.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        ops.mask_small_regions_(mm, 100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"mask_small_regions on {K} random 1024 x 1024 masks (worst case: ~50 % speckle): {dt * 1e3:.1f} ms")
