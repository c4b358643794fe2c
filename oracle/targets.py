"""Oracle: proposal decode + IoU / IoP targets + proposal maps (test infrastructure), restating the reference's CPU data path:
`utils/sam_mask_reader.py:49-113` (sort by area, top 50, `mask_util.decode`, pad to square), `utils/utils.py:174-272`
(`compute_iou`, `compute_iop`, `compute_all_iou`, `compute_all_iop`), `utils/reason_seg_dataset.py:166-181` (antialiased resize to 256).

Third-party pieces: `pycocotools==2.0.7` RLE codec (absent here: the published maskApi.c algorithm restated, PARITY UNPINNED for the
string codec; pinned by round trip + hand-built run lists), `skimage==0.22.0` `resize(order=0, anti_aliasing=False)` (absent: it calls
`scipy.ndimage.zoom(order=0, grid_mode=True)`, which IS installed -- the restated index rule is pinned against it in
tests/test_targets_cpu.py), `torch.nn.functional.interpolate(antialias=True)` (installed: called directly, as the reference does)."""
import numpy as np
import torch
import torch.nn.functional as F


def rle_encode(mask):
    """uint8 [H, W] -> COCO RLE dict with the compressed counts string (maskApi.c rleEncode + rleToString)."""
    h, w = mask.shape
    flat = np.asarray(mask, dtype=np.uint8).T.reshape(-1)      # column-major walk
    cnts, prev, run = [], 0, 0
    for v in flat:
        if v != prev:
            cnts.append(run); run = 0; prev = v
        run += 1
    cnts.append(run)
    s = []
    for i, c in enumerate(cnts):
        x = int(c)
        if i > 2:
            x -= int(cnts[i - 2])
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            s.append(chr(ch + 48))
    return {"size": [h, w], "counts": "".join(s)}


def rle_decode(rle):
    """COCO RLE dict -> uint8 [H, W] (maskApi.c rleFrString + rleDecode)."""
    h, w = rle["size"]
    c = rle["counts"]
    if isinstance(c, (list, tuple)):
        cnts = [int(v) for v in c]
    else:
        if isinstance(c, bytes):
            c = c.decode("ascii")
        cnts, p = [], 0
        while p < len(c):
            x, k, more = 0, 0, True
            while more:
                ch = ord(c[p]) - 48
                x |= (ch & 0x1f) << (5 * k)
                more = bool(ch & 0x20)
                p += 1; k += 1
                if not more and (ch & 0x10):
                    x |= -1 << (5 * k)
            if len(cnts) > 2:
                x += cnts[-2]
            cnts.append(x)
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, v = 0, 0
    for n in cnts:
        flat[pos:pos + n] = v
        pos += n; v ^= 1
    return flat.reshape(w, h).T.copy()


def extract_sam_segs(masks, top=50):
    """sam_mask_reader.py:69-113 -> segs_origin uint8 [H, W, K], segs_square float64 [S, S, K]."""
    ms = sorted(masks, key=lambda m: m["area"], reverse=True)[:top]
    segs = np.stack([rle_decode(m["segmentation"]) for m in ms], -1)
    h, w, _ = segs.shape
    sq = np.pad(segs.astype(np.float64), ((0, max(h, w) - h), (0, max(h, w) - w), (0, 0)))
    return {"segs_origin": segs, "segs_square": sq, "bbox": [m["bbox"] for m in ms]}


def resize_nearest(gt, H, W):
    """skimage.transform.resize(gt, (H, W), anti_aliasing=False, preserve_range=True, order=0): float64 output, source index
    floor(((i + 0.5) * in / out - 0.5) + 0.5) per axis (scipy.ndimage.zoom, order 0, grid_mode=True)."""
    gt = np.asarray(gt)

    def idx(out_n, in_n):
        zoom = np.float64(in_n) / np.float64(out_n)
        c = (np.arange(out_n, dtype=np.float64) + 0.5) * zoom - 0.5
        return np.clip(np.floor(c + 0.5), 0, in_n - 1).astype(np.int64)
    return gt[idx(H, gt.shape[0])][:, idx(W, gt.shape[1])].astype(np.float64)


def compute_all_iou_iop(segs_hwk, gt):
    """utils.py:234-272 for all proposals -> (ious, iops) float64 [K] (0 / 0 = nan, as numpy gives the reference)."""
    H, W, K = segs_hwk.shape
    g = resize_nearest(gt, H, W)
    ious, iops = [], []
    with np.errstate(divide="ignore", invalid="ignore"):
        for i in range(K):
            s = segs_hwk[:, :, i]
            inter, union = np.logical_and(s, g), np.logical_or(s, g)
            ious.append(np.sum(inter) / np.sum(union))
            iops.append(np.sum(inter) / np.sum(s))
    return np.array(ious), np.array(iops)


def proposal_maps(segs_square, out=256, dtype=torch.bfloat16):
    """reason_seg_dataset.py:166-173: float64 [S, S, K] -> [K, 256, 256] in the training dtype."""
    t = torch.from_numpy(segs_square).permute(2, 0, 1).contiguous()
    return F.interpolate(t.unsqueeze(0), size=(out, out), mode="bilinear", align_corners=False, antialias=True).squeeze(0).to(dtype)
