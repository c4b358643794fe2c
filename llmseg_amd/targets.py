"""Proposal decode + IoU / IoP targets + the 256 x 256 proposal maps on the device: the per-sample work the
reference does on CPU data-loader workers, feeding `model_forward`'s `sam_segs_list` / `sam_ious_list` / `sam_iops_list`.

Mirrors, with the same names and argument meaning where they exist in the reference:
  * `SAM_Mask_Reader.extract_sam_segs` (utils/sam_mask_reader.py:69-113): proposals sorted by area, top 50, RLE -> dense, padded square;
  * `compute_all_iou` / `compute_all_iop` (utils/utils.py:234-272): ground truth resampled to the proposals' grid with
    `skimage.transform.resize(order=0)`, then |seg & gt| / |seg | gt| and |seg & gt| / |seg|;
  * the dataset's proposal maps (utils/reason_seg_dataset.py:166-173): float64 zero-padded square -> `F.interpolate(size=(256, 256),
    mode="bilinear", align_corners=False, antialias=True)` -> bf16.
Host side: parsing the COCO run-length strings and building the (tiny) index / tap tables in float64; every per-pixel operation runs in
libllmseg_hip.so (`llmseg_rle_decode`, `llmseg_mask_targets`, `llmseg_resize_aa`)."""
import ctypes as C

import numpy as np
import torch

from . import _lib

BF16 = torch.bfloat16


def rle_counts(rle):
    """COCO RLE dict {'size': [h, w], 'counts': str | bytes | list} -> run lengths (uint32).  The compressed string packs each count
    in 5-bit groups, low group first, bit 5 = continuation, bit 4 of the last group = sign; counts from the fourth on are deltas against
    the count two positions back (pycocotools maskApi rleFrString)."""
    cnts = rle["counts"]
    if isinstance(cnts, (list, tuple, np.ndarray)):
        return np.asarray(cnts, dtype=np.uint32)
    if isinstance(cnts, str):
        cnts = cnts.encode("ascii")
    b = np.frombuffer(cnts, dtype=np.uint8).astype(np.int64) - 48
    last = (b & 0x20) == 0                                   # final group of each count
    idx = np.cumsum(np.concatenate([[0], last[:-1]]))        # which count a group belongs to
    start = np.concatenate([[0], np.nonzero(last)[0][:-1] + 1])
    pos = np.arange(len(b)) - start[idx]                     # group position inside its count
    vals = np.zeros(int(idx[-1]) + 1, dtype=np.int64)
    np.add.at(vals, idx, (b & 0x1f) << (5 * pos))
    ends = np.nonzero(last)[0]
    neg = (b[ends] & 0x10) != 0
    vals[neg] |= -1 << (5 * (pos[ends][neg] + 1))
    for i in range(3, len(vals)):                            # delta coding (sequential by definition; masks have O(1e3) runs)
        vals[i] += vals[i - 2]
    return vals.astype(np.uint32)


def rle_encode_masks(masks):
    """uint8 / bool [K, H, W] (tensor or array) -> [{'size': [H, W], 'counts': str}]: pycocotools `mask.encode` output (maskApi rleEncode +
    rleToString: column-major runs starting with zeros; counts from the fourth on as deltas against the count two back; 5-bit groups, low
    first, bit 5 = continuation, bit 4 of the last group = sign).  Host-side: this is the reference's FILE format (prepare_datasets/*),
    the path itself consumes the dense masks."""
    m = masks.detach().cpu().numpy() if torch.is_tensor(masks) else np.asarray(masks)
    K, H, W = m.shape
    out = []
    for k in range(K):
        flat = np.ascontiguousarray(m[k].astype(bool).T).reshape(-1)
        ch = np.flatnonzero(flat[1:] != flat[:-1]) + 1
        idx = np.concatenate([[0], ch, [H * W]])
        cnts = np.diff(idx).astype(np.int64)
        if flat[0]:
            cnts = np.concatenate([[0], cnts])
        d = cnts.copy()
        d[3:] -= cnts[1:-2]
        chars = []
        for x in d.tolist():
            more = True
            while more:
                c = x & 0x1f
                x >>= 5
                more = (x != -1) if (c & 0x10) else (x != 0)
                if more:
                    c |= 0x20
                chars.append(c + 48)
        out.append({"size": [int(H), int(W)], "counts": bytes(chars).decode("ascii")})
    return out


def nearest_index(out_n, in_n):
    """Source index of every output index for skimage.transform.resize(order=0, anti_aliasing=False) == scipy.ndimage.zoom(order=0,
    grid_mode=True): coordinate (i + 0.5) * (in / out) - 0.5 in float64, nearest = floor(c + 0.5)."""
    zoom = np.float64(in_n) / np.float64(out_n)
    c = (np.arange(out_n, dtype=np.float64) + 0.5) * zoom - 0.5
    return np.clip(np.floor(c + 0.5), 0, in_n - 1).astype(np.int32)


def aa_taps(in_size, out_size):
    """Tap tables of torch's antialiased bilinear resampling (aten UpSampleKernel `_compute_indices_weights_aa`, triangle filter, float64):
    -> (first [out] int32, count [out] int32, weights [out, taps] float64)."""
    scale = np.float64(in_size) / np.float64(out_size)
    support = scale if scale >= 1.0 else np.float64(1.0)
    invscale = 1.0 / scale if scale >= 1.0 else np.float64(1.0)
    taps = int(np.ceil(support)) * 2 + 1
    first = np.zeros(out_size, np.int32)
    count = np.zeros(out_size, np.int32)
    w = np.zeros((out_size, taps), np.float64)
    for i in range(out_size):
        center = scale * (i + 0.5)
        xmin = max(int(center - support + 0.5), 0)
        xsize = min(int(center + support + 0.5), in_size) - xmin
        ws = np.array([max(0.0, 1.0 - abs((j + xmin - center + 0.5) * invscale)) for j in range(xsize)], dtype=np.float64)
        tot = ws.sum()
        first[i], count[i] = xmin, xsize
        w[i, :xsize] = ws / tot if tot != 0 else ws
    return first, count, w


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _LRU:
    """Cache of device tables capped at `cap` entries (least recently used evicted): a dataset of arbitrary image sizes must not grow it without
    bound.  `get(key, make)` is ONE locked lookup-or-build (loader threads share the cache).  Keys carry the device AND the stream the table was
    built on: an evicted table goes back to that stream's pool of the caching allocator, where only later work of the SAME stream can reuse the
    block -- a table shared across streams could be handed out again while a side stream's kernel still reads it (bench.py's loader runs the
    target kernels on a side stream)."""

    def __init__(self, cap):
        import collections
        import threading
        self.cap, self._d, self._lock = cap, collections.OrderedDict(), threading.Lock()

    def get(self, key, make):
        with self._lock:
            if key in self._d:
                self._d.move_to_end(key)
                return self._d[key]
        v = make()                                     # built outside the lock (a host -> device copy); a racing duplicate is harmless
        with self._lock:
            v = self._d.setdefault(key, v)
            self._d.move_to_end(key)
            while len(self._d) > self.cap:
                self._d.popitem(last=False)
            return v

    def __len__(self):
        return len(self._d)


_TABLES = _LRU(64)


def _where(dev):
    dev = torch.device(dev)
    return (str(dev), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)


def _dev_nearest(out_n, in_n, dev):
    """`nearest_index` as a device tensor, built once per (sizes, device, stream): a loader calls this every sample, and a pageable host->device
    copy stalls the host until the stream has drained (measured: 5 ms of idle GPU per micro-step with the tables rebuilt per call)."""
    return _TABLES.get(("nn", out_n, in_n) + _where(dev), lambda: torch.from_numpy(nearest_index(out_n, in_n)).to(dev))


def _dev_taps(in_size, out_size, dev):
    def make():
        first, count, w = aa_taps(in_size, out_size)
        return (torch.from_numpy(first).to(dev), torch.from_numpy(count).to(dev), torch.from_numpy(w).to(dev), w.shape[1])
    return _TABLES.get(("aa", in_size, out_size) + _where(dev), make)


def _p(t):
    return C.c_void_p(t.data_ptr())


def decode_rles(rles, device, hwk=False):
    """list of COCO RLE dicts (same size) -> uint8 [K, H, W] (or [H, W, K], `mask_util.decode`'s layout) on the device."""
    H, W = rles[0]["size"]
    runs = [rle_counts(r) for r in rles]
    assert all(tuple(r["size"]) == (H, W) for r in rles)
    ends = np.concatenate([np.cumsum(r, dtype=np.uint64).astype(np.uint32) for r in runs])
    offs = np.concatenate([[0], np.cumsum([len(r) for r in runs])]).astype(np.int64)
    K = len(rles)
    d_ends = torch.from_numpy(ends.view(np.int32)).to(device)
    d_offs = torch.from_numpy(offs).to(device)
    out = torch.empty((H, W, K) if hwk else (K, H, W), device=device, dtype=torch.uint8)
    _lib.check(_lib.load().llmseg_rle_decode(_p(d_ends), _p(d_offs), _p(out), K, H, W, 1 if hwk else 0, _stream()), "rle_decode")
    return out


def mask_targets(segs, gt):
    """segs uint8 [K, H, W] (device), gt uint8 [Hg, Wg] (device, non-zero = object) -> (iou, iop) float64 [K] -- `compute_all_iou` /
    `compute_all_iop` of the reference for all K proposals in one pass -- plus the exact integer counts [K, 2] = (|seg & gt|, |seg|)."""
    K, H, W = segs.shape
    Hg, Wg = gt.shape
    dev = segs.device
    gy, gx = _dev_nearest(H, Hg, dev), _dev_nearest(W, Wg, dev)
    cnt = torch.zeros((K, 2), device=dev, dtype=torch.int64)
    garea = torch.zeros((1,), device=dev, dtype=torch.int64)
    iou = torch.empty((K,), device=dev, dtype=torch.float64)
    iop = torch.empty((K,), device=dev, dtype=torch.float64)
    _lib.check(_lib.load().llmseg_mask_targets(_p(segs.contiguous()), _p(gt.contiguous()), _p(gy), _p(gx), K, H, W, Hg, Wg, _p(cnt), _p(garea), _p(iou), _p(iop),
                                               _stream()), "mask_targets")
    return iou, iop, cnt


def resize_square_aa(segs, out_size=256):
    """segs uint8 [K, H, W] -> bf16 [K, out, out]: zero-pad to the square of side max(H, W) (bottom / right, `preprocess_mask`) and
    resample with the antialiased bilinear filter."""
    K, H, W = segs.shape
    side = max(H, W)
    dev = segs.device
    d_first, d_count, d_w, n_taps = _dev_taps(side, out_size, dev)
    out = torch.empty((K, out_size, out_size), device=dev, dtype=BF16)
    _lib.check(_lib.load().llmseg_resize_aa(_p(segs.contiguous()), _p(out), K, H, W, out_size, _p(d_first), _p(d_count), _p(d_w), _p(d_first), _p(d_count), _p(d_w),
                                            n_taps, _stream()), "resize_aa")
    return out


def extract_sam_segs(masks, device, top=50):
    """`SAM_Mask_Reader.extract_sam_segs` on the device: `masks` = the image's proposal records ({'segmentation': RLE, 'area', 'bbox'}).
    -> {"segs_origin": uint8 [K, H, W], "bbox": [...]} (the zero-padded square is never materialised: `resize_square_aa` pads on the fly)."""
    ms = sorted(masks, key=lambda m: m["area"], reverse=True)[:top]
    return {"segs_origin": decode_rles([m["segmentation"] for m in ms], device), "bbox": [m["bbox"] for m in ms]}


def _stack_targets(rows, K, dev):
    """[C] list of float64 [K] -> [C, K]; no ground truth (validation samples carry none, utils/dataset.py:640-656) -> an empty [0, K], on every route."""
    return torch.stack(rows) if rows else torch.empty((0, K), device=dev, dtype=torch.float64)


def proposals_and_targets(masks, gt_masks, device, top=50, out_size=256):
    """Everything `model_forward` needs about one image's proposals: -> dict(sam_segs bf16 [K, 256, 256], sam_ious / sam_iops float64 [C, K]
    for the C sampled ground-truth masks (uint8 [Hg, Wg] tensors or arrays), segs_origin uint8 [K, H, W])."""
    d = extract_sam_segs(masks, device, top)
    segs = d["segs_origin"]
    ious, iops = [], []
    for g in gt_masks:
        g = torch.as_tensor(np.asarray(g) if not torch.is_tensor(g) else g).to(device=device, dtype=torch.uint8)
        iou, iop, _ = mask_targets(segs, g)
        ious.append(iou); iops.append(iop)
    K = segs.shape[0]
    return {"sam_segs": resize_square_aa(segs, out_size), "sam_ious": _stack_targets(ious, K, segs.device), "sam_iops": _stack_targets(iops, K, segs.device),
            "segs_origin": segs, "bbox": d["bbox"]}


FUSED_LIMITS = dict(out_size=256, taps=28, width=2048, n_gt=4)      # llmseg_proposal_targets' limits (csrc/targets.hip)


def proposal_targets_fused(masks, order, gt_masks, out_size=256):
    """One pass over the selected proposals masks[order] (uint8 [*, H, W] on the device; order int64 [K] or None): -> (maps bf16 [K, out, out],
    ious, iops float64 [C, K], counts int64 [C, K, 2]).  None when the shape is outside the fused kernel's limits."""
    Kall, H, W = masks.shape
    K = Kall if order is None else int(order.numel())
    dev = masks.device
    side = max(H, W)
    d_first, d_count, d_w, n_taps = _dev_taps(side, out_size, dev)
    if out_size > FUSED_LIMITS["out_size"] or n_taps > FUSED_LIMITS["taps"] or W > FUSED_LIMITS["width"] or K == 0:
        return None
    lib = _lib.load()
    gts = [torch.as_tensor(np.asarray(g) if not torch.is_tensor(g) else g).to(device=dev, dtype=torch.uint8).contiguous() for g in gt_masks]
    C_ = len(gts)
    out = torch.empty((K, out_size, out_size), device=dev, dtype=BF16)
    ious = torch.empty((C_, K), device=dev, dtype=torch.float64)
    iops = torch.empty((C_, K), device=dev, dtype=torch.float64)
    cnts = torch.empty((C_, K, 2), device=dev, dtype=torch.int64)
    masks = masks.contiguous()
    first = True
    for c0 in range(0, max(C_, 1), FUSED_LIMITS["n_gt"]):                      # the first <= 4 ground truths ride in the fused pass that also makes the maps
        grp = gts[c0:c0 + FUSED_LIMITS["n_gt"]]
        if not first:
            # further ground truths need counts only: the popcount kernel on the selected proposals (one gathered copy, shared by all of them)
            # instead of re-running the whole resampling pass into a scratch map per group of four
            segs_sel = masks if order is None else masks[order].contiguous()
            for c in range(c0, C_):
                ious[c], iops[c], cnts[c] = mask_targets(segs_sel, gts[c])
            break
        n = len(grp)
        gtp = torch.empty((max(n, 1), H, W), device=dev, dtype=torch.uint8)
        for i, g in enumerate(grp):
            Hg, Wg = g.shape
            _lib.check(lib.llmseg_gt_resample(_p(g), _p(_dev_nearest(H, Hg, dev)), _p(_dev_nearest(W, Wg, dev)), _p(gtp[i]), H, W, Hg, Wg, _stream()), "gt_resample")
        garea = torch.empty((max(n, 1),), device=dev, dtype=torch.int64)
        _lib.check(lib.llmseg_proposal_targets(_p(masks), None if order is None else _p(order.contiguous()), _p(gtp), n, _p(out), K, H, W, out_size,
                                               _p(d_first), _p(d_count), _p(d_w), _p(d_first), _p(d_count), _p(d_w), n_taps,
                                               _p(cnts[c0:c0 + n]) if n else None, _p(garea) if n else None, _p(ious[c0:c0 + n]) if n else None,
                                               _p(iops[c0:c0 + n]) if n else None, _stream()), "proposal_targets")
        first = False
    return out, ious, iops, cnts


def proposals_and_targets_dense(masks, areas, gt_masks, top=50, out_size=256, want_origin=True):
    """The same for proposals that are already dense masks on the device (uint8 [K, H, W] + their areas, e.g. the output of
    `LISAForCausalLM.generate_proposals`): largest `top` by area (sam_mask_reader.py:69-75), no RLE round trip.  The selected proposals are
    read once, through the `order` index (`llmseg_proposal_targets`); `want_origin=False` skips the gathered uint8 copy `segs_origin`
    (a loader that only feeds `model_forward` never reads it)."""
    order = torch.argsort(areas, descending=True, stable=True)[:top]
    fused = proposal_targets_fused(masks, order, gt_masks, out_size)
    if fused is not None:
        maps, ious, iops, _ = fused
        res = {"sam_segs": maps, "sam_ious": ious, "sam_iops": iops, "order": order}
        if want_origin:
            res["segs_origin"] = masks[order].contiguous()
        return res
    segs = masks[order].contiguous()
    ious, iops = [], []
    for g in gt_masks:
        g = torch.as_tensor(np.asarray(g) if not torch.is_tensor(g) else g).to(device=segs.device, dtype=torch.uint8)
        iou, iop, _ = mask_targets(segs, g)
        ious.append(iou); iops.append(iop)
    K = segs.shape[0]
    return {"sam_segs": resize_square_aa(segs, out_size), "sam_ious": _stack_targets(ious, K, segs.device), "sam_iops": _stack_targets(iops, K, segs.device),
            "segs_origin": segs, "order": order}
