"""GPU box: time of the N2 target computation for one image (256 dense proposals at 1024 x 1024), alone on the device: the fused one-pass kernel
vs the round-3 route (gathered copy + mask_targets + resize_aa)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from llmseg_amd import targets as T  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
K, H, W = 300, 1024, 1024
masks = (torch.rand((K, H, W), device=dev, generator=g) > 0.7).to(torch.uint8)
areas = masks.flatten(1).sum(1)
gt = (torch.rand((H, W), device=dev, generator=g) > 0.5).to(torch.uint8)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


order = torch.argsort(areas, descending=True, stable=True)[:256]
print("argsort + slice: %.3f ms" % timeit(lambda: torch.argsort(areas, descending=True, stable=True)[:256]))
print("fused (proposal_targets_fused, order given): %.3f ms" % timeit(lambda: T.proposal_targets_fused(masks, order, [gt])))
print("proposals_and_targets_dense(want_origin=False): %.3f ms" % timeit(lambda: T.proposals_and_targets_dense(masks, areas, [gt], top=256, want_origin=False)))


def old():
    segs = masks[order].contiguous()
    T.mask_targets(segs, gt)
    T.resize_square_aa(segs, 256)


print("round-3 route (gather + mask_targets + resize_aa): %.3f ms" % timeit(old))
a = T.proposal_targets_fused(masks, order, [gt])
segs = masks[order].contiguous()
iou, iop, cnt = T.mask_targets(segs, gt)
print("maps identical:", torch.equal(a[0], T.resize_square_aa(segs, 256)), " iou identical:", torch.equal(a[1][0], iou), " counts identical:", torch.equal(a[3][0], cnt))
