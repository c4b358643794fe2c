"""CPU: the oracle restatement reproduces the fixtures generated from the imported reference
(oracle/make_goldens.py).  This is the pin that lets the GPU tests trust the oracle."""
import torch

from oracle import cases, lisa, losses, mask_head, sam_encoder


def close(a, b, tol=2e-5):
    a, b = torch.as_tensor(a).float(), torch.as_tensor(b).float()
    return (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())


def test_losses(golden):
    g, i = golden("losses.pt"), cases.loss_inputs()
    assert close(losses.softmax_align(i["P"], i["t"], i["iou"]), g["align"], 1e-6)
    assert close(losses.iop_regression(i["pr"], i["iou"]), g["reg"], 1e-6)
    assert close(losses.dice(i["x"], i["y"], 3), g["dice"], 1e-6)
    assert close(losses.sigmoid_ce(i["x"], i["y"], 3), g["bce"], 1e-6)
    P = i["P"].clone().requires_grad_(True)
    t = i["t"].clone().requires_grad_(True)
    losses.softmax_align(P, t, i["iou"]).backward()
    assert close(P.grad, g["dP"], 1e-5) and close(t.grad, g["dt"], 1e-5)


def test_iou_metric(golden):
    g = golden("iou_metric.pt")["IUT"]
    for n, (pred, tgt) in enumerate(cases.iou_metric_cases()):
        i, u, t = losses.intersection_and_union(pred, tgt, 2, 255)
        assert torch.equal(torch.stack([i, u, t]), g[n])


def test_sam_encoder_small(golden):
    g = golden("sam_encoder_small.pt")
    cfg, sd, img = cases.sam_small_case(batch=1)
    with torch.no_grad():
        out = sam_encoder.sam_image_encoder(sd, "", img, cfg)
    assert close(out, g["out"])


def test_mask_head(golden):
    g = golden("mask_head.pt")
    sd, pooled, text = cases.head_case()
    with torch.no_grad():
        iou, emb = mask_head.mask_head(sd, "model.", pooled, text)
    assert close(iou, g["iou"]) and close(emb, g["emb"])
    sd, pooled, text = cases.head_case(K=512)                         # BASELINE configs[4]: 512 candidate masks
    with torch.no_grad():
        iou, emb = mask_head.mask_head(sd, "model.", pooled, text)
    assert close(iou, g["iou_k512"]) and close(emb[:, :, ::8], g["emb_k512_cols8"]) and close(emb.norm(dim=-1), g["emb_k512_rownorm"])


def test_lisa_tiny_train_and_inference(golden):
    g = golden("lisa_tiny.pt")
    cfg = cases.tiny_lisa_cfg()
    sd = cases.tiny_lisa_state(cfg)
    batch = cases.tiny_lisa_batch()
    with torch.no_grad():
        out = lisa.model_forward(sd, cfg, **batch, inference=False)
        for k in ("loss", "ce_loss", "align_loss", "regression_loss"):
            assert close(out[k], g["train"][k], 1e-5), k
        o = lisa.model_forward(sd, cfg, **cases.first_image_inference(batch), inference=True, return_aux=True)
    assert close(o["pred_similarity"][0], g["pred_similarity"], 1e-5)
    assert close(o["pred_iou"][0], g["pred_iou"], 1e-5)
    assert close(o["logits"][0, ::7, ::997], g["logits_sample"], 1e-4)
    assert close(o["hidden"][0], g["hidden"], 1e-4)


def test_lisa_tiny_grads(golden):
    """Autograd through the oracle reproduces the reference's parameter gradients."""
    g = golden("lisa_tiny.pt")["grads"]
    cfg = cases.tiny_lisa_cfg()
    sd = cases.tiny_lisa_state(cfg)
    names = {"text_fc2_w": "model.text_hidden_fcs.0.2.weight", "lm_head_rows": "lm_head.weight",
             "iou_head0_w": "model.lisa_iou_head.0.weight", "q_proj_l1": "model.layers.1.self_attn.q_proj.weight",
             "final_attn_q_w": "model.lisa_final_attn.q_proj.weight"}
    for n in names.values():
        sd[n].requires_grad_(True)
    lisa.model_forward(sd, cfg, **cases.tiny_lisa_batch(), inference=False)["loss"].backward()
    for k, n in names.items():
        got = sd[n].grad[::1000] if k == "lm_head_rows" else sd[n].grad
        assert close(got, g[k], 2e-4), k
    # softmax over a single key is constant -> exactly-zero grads (SURVEY.md §7 hard parts)
    assert sd["model.lisa_final_attn.q_proj.weight"].grad.abs().max().item() == 0.0


def test_validation_selection_rules_cpu():
    """The two remaining validation loops (reference training.py:872-1078) as oracle bodies: the arg-max proposal is always part of the
    iou+iop prediction; the top-5 rule yields an EMPTY prediction when no IoP passes; both reduce to the arg-max body when exactly the
    arg-max proposal passes."""
    import torch
    from oracle import metric
    g = torch.Generator().manual_seed(0)
    segs = (torch.rand(40, 50, 12, generator=g) > 0.7).to(torch.uint8)
    gt = (torch.rand(80, 100, generator=g) > 0.5).to(torch.uint8)
    sim, iop = torch.rand(12, generator=g), torch.rand(12, generator=g)
    k = int(torch.argmax(sim))
    only_k = torch.zeros(12)
    only_k[k] = 1.0
    a = metric.argmax_iou(segs, sim, gt)
    for body in (metric.iou_iop_iou, metric.top_iou_iou):
        b = body(segs, sim, only_k, gt, threshold=0.5)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    none = metric.top_iou_iou(segs, sim, iop, gt, threshold=2.0)
    assert none[0][1] == 0 and none[1][1] == int((gt == 1).sum())          # empty prediction: class-1 intersection 0, union = the target
    with_k = metric.iou_iop_iou(segs, sim, iop, gt, threshold=2.0)
    assert torch.equal(with_k[0], a[0])                                    # nothing passes: the arg-max proposal alone
