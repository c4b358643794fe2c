"""Generate tests/golden/* by running the imported reference.  BUILD CONTAINER ONLY
(`python -m oracle.make_goldens`; needs /root/reference).  Test infrastructure.

For every case: run the reference module on seeded inputs + seeded weights, assert the
oracle restatement reproduces it (fp32, tight), then store the reference outputs.
Fixtures hold data only (expected outputs + the case parameters), never reference code.
"""
import os
import sys
from functools import partial

import torch
import torch.nn as nn

from . import cases, lisa, losses, mask_head, ref_harness as rh, sam_encoder, seeded

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
TOL = 2e-5


def _check(name, ref, mine, tol=TOL):
    d = (ref.float() - mine.float()).abs().max().item()
    s = ref.float().abs().max().item()
    print(f"  {name}: max|ref-oracle| = {d:.3e} (|ref|max {s:.3e})")
    assert d <= tol * max(1.0, s), name


def gold_losses():
    import model.loss as RL
    i = cases.loss_inputs()
    ref = dict(align=RL.softmax_align_loss(i["P"], i["t"], i["iou"]), reg=RL.iou_regression_loss(i["pr"], i["iou"]),
               dice=RL.dice_loss(i["x"], i["y"], 3), bce=RL.sigmoid_ce_loss(i["x"], i["y"], 3))
    mine = dict(align=losses.softmax_align(i["P"], i["t"], i["iou"]), reg=losses.iop_regression(i["pr"], i["iou"]),
                dice=losses.dice(i["x"], i["y"], 3), bce=losses.sigmoid_ce(i["x"], i["y"], 3))
    for k in ref:
        _check("loss." + k, ref[k], mine[k], 1e-6)
    # gradients of the two live losses w.r.t. their inputs (for the backward kernels)
    P = i["P"].clone().requires_grad_(True); t = i["t"].clone().requires_grad_(True)
    RL.softmax_align_loss(P, t, i["iou"]).backward()
    pr = i["pr"].clone().requires_grad_(True)
    RL.iou_regression_loss(pr, i["iou"]).backward()
    torch.save({**{k: v.detach() for k, v in ref.items()}, "dP": P.grad, "dt": t.grad, "dpr": pr.grad},
               os.path.join(OUT, "losses.pt"))


def gold_iou_metric():
    sys.modules.setdefault("cv2", sys.modules.get("cv2"))
    from utils.utils import intersectionAndUnionGPU
    exp = []
    for pred, tgt in cases.iou_metric_cases():
        ri, ru, rt = intersectionAndUnionGPU(pred.clone().float(), tgt.clone().float(), 2, ignore_index=255)
        mi, mu, mt = losses.intersection_and_union(pred, tgt, 2, 255)
        _check("iou.I", ri, mi, 0); _check("iou.U", ru, mu, 0)
        exp.append(torch.stack([ri, ru, rt]))
    torch.save({"IUT": torch.stack(exp)}, os.path.join(OUT, "iou_metric.pt"))


def gold_sam_small():
    from model.segment_anything.modeling.image_encoder import ImageEncoderViT
    cfg, sd, img = cases.sam_small_case(batch=1)
    ref = ImageEncoderViT(depth=cfg.depth, embed_dim=cfg.dim, img_size=cfg.img, mlp_ratio=4,
                          norm_layer=partial(nn.LayerNorm, eps=1e-6), num_heads=cfg.heads, patch_size=cfg.patch,
                          qkv_bias=True, use_rel_pos=True, global_attn_indexes=list(cfg.global_idx),
                          window_size=cfg.window, out_chans=cfg.out_chans).eval()
    ref.load_state_dict(sd, strict=True)
    with torch.no_grad():
        a = ref(img)
        b = sam_encoder.sam_image_encoder(sd, "", img, cfg)
        # per-block activations, useful when bisecting a kernel mismatch
        x = ref.patch_embed(img) + ref.pos_embed
        blk = []
        for bl in ref.blocks:
            x = bl(x)
            blk.append(x.clone())
    _check("sam_small.out", a, b)
    torch.save({"out": a, "block0": blk[0], "block1": blk[1]}, os.path.join(OUT, "sam_encoder_small.pt"))


def gold_head():
    from model.transformer import Attention, LISA_TwoWayAttentionBlock
    sd, pooled, text = cases.head_case()
    layers = nn.ModuleList([LISA_TwoWayAttentionBlock(256, 8, 2048, attention_downsample_rate=1) for _ in range(2)])
    fin, nrm = Attention(256, 8, downsample_rate=1), nn.LayerNorm(256)
    iou_h = nn.Sequential(nn.Linear(256, 128), nn.ReLU(), nn.Linear(128, 1), nn.Sigmoid())
    emb_h = nn.Sequential(nn.Linear(256, 2048), nn.ReLU(), nn.Linear(2048, 256))
    mods = {"lisa_attention_layers": layers, "lisa_final_attn": fin, "lisa_norm_final_attn": nrm,
            "lisa_iou_head": iou_h, "lisa_embedding_head": emb_h}
    for n, m in mods.items():
        m.load_state_dict({k[len("model." + n) + 1:]: v for k, v in sd.items() if k.startswith("model." + n + ".")},
                          strict=True)

    def run(pooled, text):
        with torch.no_grad():
            C = text.shape[0]
            s, t = pooled.unsqueeze(0).expand(C, -1, -1), text.unsqueeze(1)       # LISA.py:363-372
            for l in layers:
                s, t = l(queries=s, keys=t)
            s = nrm(s + fin(q=s, k=t, v=t))
            r_iou, r_emb = iou_h(s), emb_h(s)
            m_iou, m_emb = mask_head.mask_head(sd, "model.", pooled, text)
        _check(f"head.iou K={pooled.shape[0]}", r_iou, m_iou); _check(f"head.emb K={pooled.shape[0]}", r_emb, m_emb)
        return r_iou, r_emb
    r_iou, r_emb = run(pooled, text)
    # BASELINE configs[4]: 512 candidate masks -- the embeddings as a strided sample (file size), the scores in full
    _, pooled5, text5 = cases.head_case(K=512)
    r_iou5, r_emb5 = run(pooled5, text5)
    torch.save({"iou": r_iou, "emb": r_emb, "iou_k512": r_iou5, "emb_k512_cols8": r_emb5[:, :, ::8].clone(),
                "emb_k512_rownorm": r_emb5.norm(dim=-1)}, os.path.join(OUT, "mask_head.pt"))


def _tiny_ref():
    """The imported reference model at the tiny configuration, loaded with the seeded tiny state."""
    cfg = cases.tiny_lisa_cfg()
    rh.setup(clip_cfg_kwargs=dict(hidden_size=cfg.clip.dim, intermediate_size=cfg.clip.mlp,
                                  num_hidden_layers=cfg.clip.layers, num_attention_heads=cfg.clip.heads),
             dino_cfg_kwargs=dict(num_hidden_layers=cfg.dino.layers))
    m = rh.build_lisa(dict(hidden_size=cfg.llama.hidden, intermediate_size=cfg.llama.inter,
                           num_hidden_layers=cfg.llama.layers, num_attention_heads=cfg.llama.heads,
                           num_key_value_heads=cfg.llama.heads, vocab_size=cfg.llama.vocab,
                           max_position_embeddings=2048, rms_norm_eps=cfg.llama.eps), seg_token_idx=cfg.seg_token_idx)
    sd = cases.tiny_lisa_state(cfg)
    ref_sd = {}
    for k, v in sd.items():
        if k.startswith("model.visual_model_dinov2.") or k.startswith("model.visual_model."):
            continue
        ref_sd[k.replace("vision_tower.vision_tower.vision_model.", "vision_tower.vision_tower.hf.")] = v
    ref_sd.update(rh.hub_to_hf_dino_names(sd, "model.visual_model_dinov2.", "model.visual_model_dinov2.m.", cfg.dino.layers))
    res = m.load_state_dict(ref_sd, strict=False)
    assert not res.unexpected_keys and all(k.startswith("model.visual_model.") for k in res.missing_keys), res
    return cfg, sd, m


def gold_generate():
    """evaluate()'s generation (LISA.py:487-521): greedy loop over the IMPORTED reference forward vs the restatement."""
    from oracle import generate as gen
    cfg, sd, m = _tiny_ref()
    m.eval()
    batch = cases.tiny_lisa_batch()
    clip, ids0 = batch["images_clip"][:2], batch["input_ids"][:2]

    def ref_forward(ids):
        with torch.no_grad():
            o = super(type(m), m).forward(images=clip, attention_mask=torch.ones_like(ids, dtype=torch.bool), input_ids=ids,
                                          output_hidden_states=True)
        return o.logits, o.hidden_states
    out = {}
    with torch.no_grad():
        seq_r, hid_r = gen.greedy_generate(None, cfg, clip, ids0, max_new_tokens=6, eos_token_id=None, forward=ref_forward)
        seq_m, hid_m = gen.greedy_generate(sd, cfg, clip, ids0, max_new_tokens=6, eos_token_id=None)
        assert torch.equal(seq_r, seq_m), (seq_r, seq_m)
        _check("generate.hidden", hid_r, hid_m, 1e-4)
        eos = int(seq_r[0, ids0.shape[1] + 2])          # make sequence 0 finish at its third new token
        seq_re, hid_re = gen.greedy_generate(None, cfg, clip, ids0, max_new_tokens=6, eos_token_id=eos, pad_token_id=0, forward=ref_forward)
        seq_me, hid_me = gen.greedy_generate(sd, cfg, clip, ids0, max_new_tokens=6, eos_token_id=eos, pad_token_id=0)
        assert torch.equal(seq_re, seq_me), (seq_re, seq_me)
        _check("generate.hidden(eos)", hid_re, hid_me, 1e-4)
        # margins of the greedy choices (top-1 minus top-2 logit) so that a bf16 run knows which steps are decidable
        lg, _ = ref_forward(seq_r[:, :-1])
        L = ids0.shape[1]
        top2 = lg[:, -6:, :].float().topk(2, -1).values
        out.update(sequences=seq_r, hidden=hid_r, eos=eos, sequences_eos=seq_re, hidden_eos_sum=hid_re.double().sum(),
                   margins=(top2[..., 0] - top2[..., 1]), prompt_len=L)
        print("   generated", seq_r[:, L:].tolist(), "eos-run", seq_re[:, L:].tolist(), "min margin", float(out["margins"].min()))
        # HF's own driver on the installed transformers (5.x, not the pinned 4.29): informative only
        try:
            g = m.generate(images=clip, input_ids=ids0, max_new_tokens=6, num_beams=1, do_sample=False, use_cache=False)
            ge = m.generate(images=clip, input_ids=ids0, max_new_tokens=6, num_beams=1, do_sample=False, use_cache=False, eos_token_id=eos,
                            pad_token_id=0)
            seq = lambda x: x if torch.is_tensor(x) else x.sequences
            print("   HF generate agrees:", torch.equal(seq(g), seq_r), "with eos/pad:", torch.equal(seq(ge), seq_re))
            out["hf_generate_agrees"] = bool(torch.equal(seq(g), seq_r) and torch.equal(seq(ge), seq_re))
        except Exception as e:      # noqa: BLE001
            print("   HF generate on transformers", __import__("transformers").__version__, "not usable with the reference model:", type(e).__name__, str(e)[:120])
    torch.save(out, os.path.join(OUT, "generate_tiny.pt"))


def gold_sam_decoder():
    """evaluate()'s mask path (LISA.py:523-557): the imported reference PromptEncoder / MaskDecoder / postprocess vs the restatement."""
    from oracle import sam_decoder as sdec, seeded
    cfg = cases.tiny_lisa_cfg()
    rh.setup(clip_cfg_kwargs=dict(hidden_size=cfg.clip.dim, intermediate_size=cfg.clip.mlp,
                                  num_hidden_layers=cfg.clip.layers, num_attention_heads=cfg.clip.heads),
             dino_cfg_kwargs=dict(num_hidden_layers=cfg.dino.layers))
    from model.segment_anything.modeling import MaskDecoder, PromptEncoder, TwoWayTransformer
    from model.segment_anything.modeling.sam import Sam
    pe = PromptEncoder(embed_dim=256, image_embedding_size=(64, 64), input_image_size=(1024, 1024), mask_in_chans=16)
    md = MaskDecoder(num_multimask_outputs=3, transformer=TwoWayTransformer(depth=2, embedding_dim=256, mlp_dim=2048, num_heads=8),
                     transformer_dim=256, iou_head_depth=3, iou_head_hidden_dim=256)
    sd = cases.sam_decoder_state()
    r1 = pe.load_state_dict({k[len(sdec.PFX + "prompt_encoder."):]: v for k, v in sd.items() if ".prompt_encoder." in k}, strict=False)
    r2 = md.load_state_dict({k[len(sdec.PFX + "mask_decoder."):]: v for k, v in sd.items() if ".mask_decoder." in k}, strict=True)
    assert not r1.unexpected_keys, r1
    emb, text = cases.sam_decoder_case()
    with torch.no_grad():
        sparse, dense = pe(points=None, boxes=None, masks=None, text_embeds=text[:, None, :])
        low, iou = md(image_embeddings=emb, image_pe=pe.get_dense_pe(), sparse_prompt_embeddings=sparse, dense_prompt_embeddings=dense,
                      multimask_output=False)
        holder = type("S", (), {"image_encoder": type("E", (), {"img_size": 1024})()})()
        post = Sam.postprocess_masks(holder, low, input_size=(683, 1024), original_size=(427, 640))
        m_low, m_iou = sdec.decode_masks(sd, emb, text)
        m_post = sdec.postprocess_masks(m_low, (683, 1024), (427, 640))
    _check("sam_decoder.dense_pe", pe.get_dense_pe()[0], sdec.dense_pe(sd), 1e-5)
    _check("sam_decoder.low_res_masks", low, m_low, 1e-4)
    _check("sam_decoder.iou", iou, m_iou, 1e-4)
    _check("sam_decoder.postprocess", post, m_post, 1e-4)
    torch.save({"low_res_sub": low[:, 0, ::4, ::4].clone(), "low_res_sum": low.double().sum(), "iou": iou, "post_sub": post[:, 0, ::7, ::9].clone(),
                "post_sum": post.double().sum(), "input_size": (683, 1024), "original_size": (427, 640)}, os.path.join(OUT, "sam_decoder.pt"))


def gold_amg():
    """SAM everything mode (automatic_mask_generator.py:127-324) from the image embedding on: the imported reference generator (with
    the image embedding set on its predictor) vs the restatement.  torchvision's batched_nms is absent -> the restated nms is shimmed
    into the reference for the last step (that step is therefore NOT pinned)."""
    from oracle import amg as oamg, sam_decoder as sdec
    import numpy as np
    cfg = cases.tiny_lisa_cfg()
    rh.setup(clip_cfg_kwargs=dict(hidden_size=cfg.clip.dim, intermediate_size=cfg.clip.mlp,
                                  num_hidden_layers=cfg.clip.layers, num_attention_heads=cfg.clip.heads),
             dino_cfg_kwargs=dict(num_hidden_layers=cfg.dino.layers))
    import model.segment_anything.automatic_mask_generator as ramg
    from model.segment_anything.modeling import MaskDecoder, PromptEncoder, TwoWayTransformer
    from model.segment_anything.modeling.sam import Sam
    ramg.batched_nms = lambda boxes, scores, idxs, iou_threshold: oamg.nms(boxes, scores, iou_threshold)
    pe = PromptEncoder(embed_dim=256, image_embedding_size=(64, 64), input_image_size=(1024, 1024), mask_in_chans=16)
    md = MaskDecoder(num_multimask_outputs=3, transformer=TwoWayTransformer(depth=2, embedding_dim=256, mlp_dim=2048, num_heads=8),
                     transformer_dim=256, iou_head_depth=3, iou_head_hidden_dim=256)
    sd = cases.sam_decoder_state()
    pe.load_state_dict({k[len(sdec.PFX + "prompt_encoder."):]: v for k, v in sd.items() if ".prompt_encoder." in k}, strict=False)
    md.load_state_dict({k[len(sdec.PFX + "mask_decoder."):]: v for k, v in sd.items() if ".mask_decoder." in k}, strict=True)
    enc = torch.nn.Module()
    enc.img_size = 1024
    sam = Sam(image_encoder=enc, prompt_encoder=pe, mask_decoder=md)
    # the vendored predictor (predictor.py:233) still calls the prompt encoder without the `text_embeds` argument LISA added to it
    # (prompt_encoder.py:140-146): the generator cannot run in the reference tree as it stands (the authors used the pip package
    # offline); pass text_embeds=None, which is what the upstream signature means
    fwd = pe.forward
    pe.forward = lambda points, boxes, masks, text_embeds=None: fwd(points, boxes, masks, text_embeds)
    thr = cases.amg_thresholds()
    gen = ramg.SamAutomaticMaskGenerator(sam, points_per_side=8, points_per_batch=16, pred_iou_thresh=thr["pred_iou_thresh"],
                                         stability_score_thresh=thr["stability_score_thresh"], stability_score_offset=thr["stability_score_offset"],
                                         box_nms_thresh=thr["box_nms_thresh"])
    emb = cases.amg_embedding_case()
    orig, inp = (427, 640), (683, 1024)
    pr = gen.predictor
    pr.features, pr.original_size, pr.input_size, pr.is_image_set = emb, orig, inp, True
    grid = gen.point_grids[0] * np.array([[orig[1], orig[0]]])
    assert np.array_equal(gen.point_grids[0], oamg.build_point_grid(8))
    with torch.no_grad():
        gen.stability_score_thresh = 0.08                         # first batch: with an active stability filter
        rb = gen._process_batch(grid[:16], orig, [0, 0, orig[1], orig[0]], orig)
        gen.stability_score_thresh = thr["stability_score_thresh"]
        mb = oamg.process_batch(sd, emb, grid[:16], inp, orig, thr["pred_iou_thresh"], 0.08, thr["stability_score_offset"])
    assert len(rb["rles"]) == mb["masks"].shape[0] and len(rb["rles"]) > 4, (len(rb["rles"]), mb["masks"].shape)
    assert rb["rles"] == oamg.mask_to_rle(mb["masks"])
    assert torch.equal(rb["boxes"].long(), mb["boxes"].long())
    _check("amg.iou_preds", rb["iou_preds"], mb["iou_preds"], 1e-5)
    _check("amg.stability", rb["stability_score"], mb["stability_score"], 1e-6)
    assert torch.equal(rb["points"], mb["points"])
    # whole pipeline (NMS step shimmed, see the docstring)
    with torch.no_grad():
        data = ramg.MaskData()
        for i in range(0, len(grid), 16):
            data.cat(gen._process_batch(grid[i:i + 16], orig, [0, 0, orig[1], orig[0]], orig))
        keep = oamg.nms(data["boxes"].float(), data["iou_preds"], thr["box_nms_thresh"])
        data.filter(keep)
        mine = oamg.generate(sd, emb, inp, orig, points_per_side=8, points_per_batch=16, **thr)
    assert data["rles"] == mine["rles"] and len(mine["rles"]) >= 3
    print("   kept", mb["masks"].shape[0], "of 48 in the first batch;", len(mine["rles"]), "records after NMS of", sum(1 for _ in data["rles"]))
    torch.save({"n_first_batch": mb["masks"].shape[0], "boxes": mine["boxes"], "iou_preds": mine["iou_preds"], "stability_score": mine["stability_score"],
                "points": mine["points"], "rle_counts": [r["counts"] for r in mine["rles"]], "areas": mine["masks"].flatten(1).sum(1),
                "original_size": orig, "input_size": inp}, os.path.join(OUT, "amg.pt"))


def gold_amg_crops():
    """Everything mode beyond the default configuration: crop layers + small-region clean-up (automatic_mask_generator.py:199-262,326-372).
    The imported generator runs end to end on a seeded uint8 image -- `set_image` per crop (Pillow resize through the two thin torchvision
    wrappers the reference imports, spelled here with Pillow itself; `Sam.preprocess`), per-layer point grids, the crop-edge filter, un-cropping,
    cross-crop NMS, `postprocess_small_regions` -- with a seeded stand-in for the image encoder (the same function on both sides) and three
    shims for absent third-party code: torchvision's `batched_nms` / `box_area` (restated) and `cv2.connectedComponentsWithStats`
    (scipy.ndimage.label): those three steps are NOT pinned, everything around them is."""
    from oracle import amg as oamg, sam_decoder as sdec
    import numpy as np
    from PIL import Image
    from scipy import ndimage
    import sys
    import types
    import model.segment_anything.automatic_mask_generator as ramg
    import model.segment_anything.utils.transforms as rtr
    from model.segment_anything.modeling import MaskDecoder, PromptEncoder, TwoWayTransformer
    from model.segment_anything.modeling.sam import Sam
    ramg.batched_nms = lambda boxes, scores, idxs, iou_threshold: oamg.nms(boxes, scores, iou_threshold)
    ramg.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    rtr.to_pil_image = lambda a: Image.fromarray(a)                                    # torchvision: uint8 HWC ndarray -> mode RGB
    rtr.resize = lambda im, size: im.resize((size[1], size[0]), Image.BILINEAR)       # torchvision: PIL image -> Image.resize(size[::-1], BILINEAR)

    def cc_stats(working, connectivity):
        assert connectivity == 8
        regions, n = ndimage.label(working, structure=np.ones((3, 3), np.int32))
        stats = np.zeros((n + 1, 5), np.int64)
        stats[:, -1] = np.bincount(regions.ravel(), minlength=n + 1)
        return n + 1, regions, stats, None
    sys.modules["cv2"] = types.SimpleNamespace(connectedComponentsWithStats=cc_stats)
    try:
        pe = PromptEncoder(embed_dim=256, image_embedding_size=(64, 64), input_image_size=(1024, 1024), mask_in_chans=16)
        md = MaskDecoder(num_multimask_outputs=3, transformer=TwoWayTransformer(depth=2, embedding_dim=256, mlp_dim=2048, num_heads=8),
                         transformer_dim=256, iou_head_depth=3, iou_head_hidden_dim=256)
        sd = cases.sam_decoder_state()
        pe.load_state_dict({k[len(sdec.PFX + "prompt_encoder."):]: v for k, v in sd.items() if ".prompt_encoder." in k}, strict=False)
        md.load_state_dict({k[len(sdec.PFX + "mask_decoder."):]: v for k, v in sd.items() if ".mask_decoder." in k}, strict=True)
        encode = cases.amg_standin_encoder()

        class Enc(torch.nn.Module):
            img_size = 1024

            def forward(self, x):
                return encode(x)
        sam = Sam(image_encoder=Enc(), prompt_encoder=pe, mask_decoder=md)
        fwd = pe.forward                                              # see gold_amg: the vendored predictor omits LISA's `text_embeds` argument
        pe.forward = lambda points, boxes, masks, text_embeds=None: fwd(points, boxes, masks, text_embeds)
        thr = cases.amg_thresholds()
        img = cases.amg_image_case()
        gold = {}
        for tag, min_area in (("crops", 0), ("crops_clean", 12)):
            kw = dict(points_per_side=8, points_per_batch=16, crop_n_layers=1, crop_n_points_downscale_factor=2, min_mask_region_area=min_area)
            gen = ramg.SamAutomaticMaskGenerator(sam, output_mode="binary_mask", **kw, **thr)
            with torch.no_grad():
                recs = gen.generate(img)
                mine = oamg.generate_crops(sd, encode, img, **kw, **thr)
            assert len(recs) == mine["masks"].shape[0] and len(recs) >= 8, (len(recs), mine["masks"].shape)
            for k, r in enumerate(recs):
                assert np.array_equal(r["segmentation"], mine["masks"][k].numpy()), (tag, k)
                x0, y0, x1, y1 = mine["boxes"][k].tolist()
                assert r["bbox"] == [x0, y0, x1 - x0, y1 - y0], (tag, k, r["bbox"], mine["boxes"][k])
                assert r["area"] == int(mine["masks"][k].sum())
                cx0, cy0, cx1, cy1 = mine["crop_boxes"][k].tolist()
                assert r["crop_box"] == [cx0, cy0, cx1 - cx0, cy1 - cy0]
                assert abs(r["predicted_iou"] - float(mine["iou_preds"][k])) < 1e-5 and abs(r["stability_score"] - float(mine["stability_score"][k])) < 1e-6
                assert r["point_coords"] == [mine["points"][k].tolist()]
            ncrop = len({tuple(c) for c in mine["crop_boxes"].tolist()})
            print(f"   {tag}: {len(recs)} records from {ncrop} crops, areas {int(mine['masks'].flatten(1).sum(1).min())}..{int(mine['masks'].flatten(1).sum(1).max())}")
            gold[tag] = {"boxes": mine["boxes"].long(), "crop_boxes": mine["crop_boxes"].long(), "points": mine["points"], "iou_preds": mine["iou_preds"],
                         "stability_score": mine["stability_score"], "areas": mine["masks"].flatten(1).sum(1), "rle_counts": [r["counts"] for r in oamg.mask_to_rle(mine["masks"])]}
        # the Pillow resize restatement (oracle/pil_resize.py) against Pillow on the crops of this image and on odd sizes
        from oracle import pil_resize
        for (x0, y0, x1, y1) in oamg.generate_crop_boxes(img.shape[:2], 1, 512 / 1500)[0]:
            c = img[y0:y1, x0:x1]
            nh, nw = oamg.preprocess_shape(*c.shape[:2])
            assert np.array_equal(np.array(Image.fromarray(c).resize((nw, nh), Image.BILINEAR)), pil_resize.resize_bilinear_u8(c, nh, nw))
        small = pil_resize.resize_bilinear_u8(img, 77, 131)
        assert np.array_equal(np.array(Image.fromarray(img).resize((131, 77), Image.BILINEAR)), small)
        gold["resize_77x131_sum"] = int(small.astype(np.int64).sum())
        gold["resize_77x131_sample"] = torch.as_tensor(small[::9, ::11].copy())
        torch.save(gold, os.path.join(OUT, "amg_crops.pt"))
    finally:
        del sys.modules["cv2"]


def gold_lisa_tiny():
    cfg = cases.tiny_lisa_cfg()
    rh.setup(clip_cfg_kwargs=dict(hidden_size=cfg.clip.dim, intermediate_size=cfg.clip.mlp,
                                  num_hidden_layers=cfg.clip.layers, num_attention_heads=cfg.clip.heads),
             dino_cfg_kwargs=dict(num_hidden_layers=cfg.dino.layers))
    m = rh.build_lisa(dict(hidden_size=cfg.llama.hidden, intermediate_size=cfg.llama.inter,
                           num_hidden_layers=cfg.llama.layers, num_attention_heads=cfg.llama.heads,
                           num_key_value_heads=cfg.llama.heads, vocab_size=cfg.llama.vocab,
                           max_position_embeddings=2048, rms_norm_eps=cfg.llama.eps), seg_token_idx=cfg.seg_token_idx)
    sd = cases.tiny_lisa_state(cfg)
    ref_sd = {}
    for k, v in sd.items():
        if k.startswith("model.visual_model_dinov2.") or k.startswith("model.visual_model."):
            continue
        ref_sd[k.replace("vision_tower.vision_tower.vision_model.", "vision_tower.vision_tower.hf.")] = v
    ref_sd.update(rh.hub_to_hf_dino_names(sd, "model.visual_model_dinov2.", "model.visual_model_dinov2.m.", cfg.dino.layers))
    res = m.load_state_dict(ref_sd, strict=False)
    assert not res.unexpected_keys and all(k.startswith("model.visual_model.") for k in res.missing_keys), res

    batch = cases.tiny_lisa_batch()
    extra = dict(masks_list=[None, None], label_list=[None, None], resize_list=[None, None])
    m.train()
    for p in m.parameters():
        p.requires_grad_(True)
    out = m(**batch, **extra, inference=False)
    with torch.no_grad():
        mine = lisa.model_forward(sd, cfg, **batch, inference=False, return_aux=True)
    for k in ("loss", "ce_loss", "align_loss", "regression_loss"):
        _check("lisa.train." + k, out[k].detach(), torch.as_tensor(mine[k]), 1e-5)
    out["loss"].backward()
    gn = {"text_fc2_w": m.model.text_hidden_fcs[0][2].weight.grad,
          "lm_head_rows": m.lm_head.weight.grad[::1000].clone(),
          "iou_head0_w": m.model.lisa_iou_head[0].weight.grad,
          "q_proj_l1": m.model.layers[1].self_attn.q_proj.weight.grad,
          "final_attn_q_w": m.model.lisa_final_attn.q_proj.weight.grad}
    m.eval()
    inf = cases.first_image_inference(batch)
    with torch.no_grad():
        r = m(**inf, masks_list=[None], label_list=[None], resize_list=[None], sam_ious_list=None,
              sam_iops_list=None, inference=True)
        o = lisa.model_forward(sd, cfg, **inf, inference=True, return_aux=True)
        ro = super(type(m), m).forward(images=inf["images_clip"], attention_mask=inf["attention_masks"],
                                       input_ids=inf["input_ids"], output_hidden_states=True)
    _check("lisa.inf.sim", r["pred_similarity"][0], o["pred_similarity"][0], 1e-5)
    _check("lisa.inf.iou", r["pred_iou"][0], o["pred_iou"][0], 1e-5)
    _check("lisa.inf.logits", ro.logits, o["logits"], 1e-4)
    _check("lisa.inf.hidden", ro.hidden_states, o["hidden"], 1e-4)
    torch.save({"train": {k: out[k].detach() for k in ("loss", "ce_loss", "align_loss", "regression_loss")},
                "grads": gn,
                "pred_similarity": r["pred_similarity"][0], "pred_iou": r["pred_iou"][0],
                "logits_sample": ro.logits[0, ::7, ::997].clone(), "hidden": ro.hidden_states[0].clone(),
                "feats_sample": o["feats"][0, ::8, ::4, ::4].clone()}, os.path.join(OUT, "lisa_tiny.pt"))


def gold_collate():
    """A14: the imported `collate_fn_new` (utils/dataset.py:33-170) + `tokenizer_image_token` + `conv_llava_v1.get_prompt()` against
    `llmseg_amd.collate` on seeded sample dicts, with the SAME stand-in tokenizer on both sides (oracle/stub_tokenizer.py: the tokenization is
    unpinned, the collate arithmetic is what is pinned).  Training (truncated to 512 - 255) and inference variants."""
    from llmseg_amd import collate as mine
    from oracle.stub_tokenizer import StubTokenizer
    D, _, _ = rh.setup_utils()
    import model.llava.conversation as CL
    from model.llava.mm_utils import tokenizer_image_token as ref_tit
    CL.default_conversation = CL.conv_templates["llava_v1"]                 # training.py:178-180

    def ref_prompt(msgs):
        conv = CL.default_conversation.copy()
        conv.messages = []
        for q, a in msgs:
            conv.append_message(conv.roles[0], q)
            conv.append_message(conv.roles[1], a)
        return conv.get_prompt()
    t = mine.CONV_TEMPLATES["llava_v1"]
    my_prompt = lambda msgs: t.get_prompt([m for q, a in msgs for m in ((t.roles[0], q), (t.roles[1], a))])
    convs = cases.collate_conversations(ref_prompt)
    assert convs == cases.collate_conversations(my_prompt), "prompt template differs from conv_llava_v1"
    assert mine.single_turn_prompt(*cases.COLLATE_QUESTIONS[0]) == ref_prompt([cases.COLLATE_QUESTIONS[0]])
    tok = StubTokenizer(model_max_length=512)
    for c in [x for cs in convs for x in cs]:
        assert ref_tit(c, tok) == mine.tokenizer_image_token(c, tok), "tokenizer_image_token"
        assert torch.equal(ref_tit(c, tok, return_tensors="pt"), mine.tokenizer_image_token(c, tok, return_tensors="pt"))
    fix = {"conversations": convs, "model_max_length": 512}
    for inference in (False, True):
        ref = D.collate_fn_new(cases.collate_samples(convs, inference), tokenizer=tok, conv_type="llava_v1", use_mm_start_end=True, local_rank=0)
        got = mine.collate_fn_new(cases.collate_samples(convs, inference), tokenizer=tok, conv_type="llava_v1", use_mm_start_end=True, local_rank=0)
        assert list(ref.keys()) == list(got.keys()), (list(ref.keys()), list(got.keys()))
        for k, v in ref.items():
            g = got[k]
            if torch.is_tensor(v):
                assert v.dtype == g.dtype and torch.equal(v, g), k
            elif isinstance(v, list) and v and torch.is_tensor(v[0]):
                assert all(a.dtype == b.dtype and torch.equal(a, b) for a, b in zip(v, g)) and len(v) == len(g), k
            else:
                assert v == g, (k, v, g)
        L = ref["input_ids"].shape[1]
        n_seg = int((ref["input_ids"] == 32000).sum())
        print(f"  collate inference={inference}: ids {tuple(ref['input_ids'].shape)}, labelled {int((ref['labels'] != -100).sum())} tokens, "
              f"padded {int((~ref['attention_masks']).sum())}, [SEG] x {n_seg}, offset {ref['offset'].tolist()}")
        assert (L == 512 - 255) == (not inference), L          # the long conversation is cut in training only
        fix["infer" if inference else "train"] = {k: ref[k] for k in ("input_ids", "labels", "attention_masks", "offset")}
    # use_mm_start_end=False and a conversation without <image> (plain tokenizer path of the label walk)
    plain = [[ref_prompt([("What is shown here?", "A dog.")])]]
    ref = D.collate_fn_new(cases.collate_samples(plain, False), tokenizer=tok, use_mm_start_end=False)
    got = mine.collate_fn_new(cases.collate_samples(plain, False), tokenizer=tok, use_mm_start_end=False)
    assert torch.equal(ref["input_ids"], got["input_ids"]) and torch.equal(ref["labels"], got["labels"])
    fix["plain"] = {"conversations": plain, "input_ids": ref["input_ids"], "labels": ref["labels"]}
    # dict_to_cuda's dtype contract (utils/utils.py:157-171; .cuda() is the identity under the harness)
    import utils.utils as UU
    r = UU.dict_to_cuda(D.collate_fn_new(cases.collate_samples(convs, False), tokenizer=tok), torch.bfloat16)
    g = mine.dict_to_cuda(mine.collate_fn_new(cases.collate_samples(convs, False), tokenizer=tok), torch.bfloat16, device="cpu")
    for k in r:
        if torch.is_tensor(r[k]):
            assert r[k].dtype == g[k].dtype, k
        elif isinstance(r[k], list) and r[k] and torch.is_tensor(r[k][0]):
            assert [x.dtype for x in r[k]] == [x.dtype for x in g[k]], k
    fix["dtypes"] = {k: str(v.dtype if torch.is_tensor(v) else v[0].dtype) for k, v in r.items()
                     if torch.is_tensor(v) or (isinstance(v, list) and v and torch.is_tensor(v[0]))}
    torch.save(fix, os.path.join(OUT, "collate.pt"))


def gold_targets():
    """N2: the reference's OWN target functions -- `compute_iou` / `compute_iop` / `compute_all_iou` / `compute_all_iop` (utils/utils.py:174-272)
    and `SAM_Mask_Reader.extract_sam_segs` (utils/sam_mask_reader.py:69-113: area sort, top 50, decode, pad to square) -- against
    oracle/targets.py (skimage's nearest resize spelled with scipy, pycocotools' decode with the restated codec: see `ref_harness.setup_utils`)."""
    import json
    import tempfile
    import numpy as np
    from . import targets as ot
    _, UU, SR = rh.setup_utils()
    masks, gt = cases.target_case()
    rles = [ot.rle_encode(m) for m in masks]
    recs = [{"segmentation": r, "area": int(m.sum()), "bbox": [0, 0, 1, 1 + i]} for i, (r, m) in enumerate(zip(rles, masks))]
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "sam_masks.json")
        with open(f, "w") as fh:
            json.dump([{"image": "other.jpg", "masks": recs[:3]}, {"image": "img.jpg", "masks": recs}], fh)
        reader = SR.SAM_Mask_Reader(f)
        ref = reader.extract_sam_segs("img.jpg")
    mine = ot.extract_sam_segs(recs)
    assert ref["segs_origin"].shape == (masks.shape[1], masks.shape[2], 50) and ref["segs_square"].dtype == np.float64
    for k in ("segs_origin", "segs_square"):
        assert ref[k].dtype == mine[k].dtype and np.array_equal(ref[k], mine[k]), k
    assert ref["bbox"] == mine["bbox"]
    r_iou, r_iop = UU.compute_all_iou(ref["segs_origin"], gt), UU.compute_all_iop(ref["segs_origin"], gt)
    m_iou, m_iop = ot.compute_all_iou_iop(mine["segs_origin"], gt)
    assert np.array_equal(r_iou, m_iou, equal_nan=True) and np.array_equal(r_iop, m_iop, equal_nan=True)
    one = UU.compute_iou(ref["segs_origin"][:, :, 0], ot.resize_nearest(gt, *ref["segs_origin"].shape[:2]))
    assert one == m_iou[0]
    # fewer than 50 proposals, one of them empty (|seg| = 0: IoP = 0 / 0 = nan, IoU = 0 as numpy gives the reference)
    sub = ref["segs_origin"][:, :, :3].copy()
    sub[:, :, 1] = 0
    with np.errstate(divide="ignore", invalid="ignore"):
        s_iou, s_iop = UU.compute_all_iou(sub, gt), UU.compute_all_iop(sub, gt)
    q_iou, q_iop = ot.compute_all_iou_iop(sub, gt)
    assert np.array_equal(s_iou, q_iou, equal_nan=True) and np.array_equal(s_iop, q_iop, equal_nan=True) and np.isnan(s_iop[1]) and s_iou[1] == 0
    print(f"  targets: 50 of {len(recs)} proposals kept, IoU max {np.nanmax(r_iou):.4f}; empty proposal: IoU {s_iou[1]}, IoP {s_iop[1]}")
    torch.save({"order_bbox_h": torch.tensor([b[3] for b in ref["bbox"]]), "ious": torch.from_numpy(r_iou), "iops": torch.from_numpy(r_iop),
                "square_sum": torch.from_numpy(ref["segs_square"].sum((0, 1))), "empty_ious": torch.from_numpy(s_iou), "empty_iops": torch.from_numpy(s_iop)},
               os.path.join(OUT, "targets_ref.pt"))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    cfg = cases.tiny_lisa_cfg()
    rh.setup(clip_cfg_kwargs=dict(hidden_size=cfg.clip.dim, intermediate_size=cfg.clip.mlp,
                                  num_hidden_layers=cfg.clip.layers, num_attention_heads=cfg.clip.heads),
             dino_cfg_kwargs=dict(num_hidden_layers=cfg.dino.layers))
    every = (gold_losses, gold_iou_metric, gold_sam_small, gold_head, gold_lisa_tiny, gold_generate, gold_sam_decoder, gold_amg, gold_amg_crops,
             gold_collate, gold_targets)
    only = set(sys.argv[1:])                                 # `python -m oracle.make_goldens gold_collate gold_targets`: just these
    for f in every:
        if only and f.__name__ not in only:
            continue
        print(f.__name__)
        f()
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
