#!/usr/bin/env python
"""Benchmark of the LLM-Seg `model_forward` hot path on MI355X (contract: see the task's bench.py section).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of `LISAForCausalLM.model_forward` over one synthetic batch of `--batch` images per rank
(BASELINE.json configs[1]: synthetic 1024x1024 images + random-init LLaVA-7B / CLIP-L / SAM ViT-H, 64-token prompt,
256 candidate masks, bf16, forward only).  Inputs are resident in HBM before the timed region.  The path shards over
independent images: no data-path collective, weak scaling.  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak of MI355X (/opt/skills/guides/MI355X_MICROARCH.md)


def cpu_baseline(threads=None, reps=5):
    """Time the CPU oracle (== reference arithmetic, fp32) on a bounded sample of the same workload and scale to 1 image:
    1 windowed + 1 global SAM ViT-H block, patch-embed + neck, 1 Llama-7B layer at T=319, 2 CLIP-L layers, lm_head, and the
    full mask-pooling + selection head at K=256.  This is the ONLY place bench.py touches oracle/."""
    import torch.nn.functional as F
    from oracle import llama as ol, mask_head as oh, sam_encoder as osam, seeded, vit as ovit
    if threads:
        torch.set_num_threads(threads)
    cores = torch.get_num_threads()
    rn = lambda *s: torch.randn(*s) * 0.02

    def fill(shapes):
        return {k: (torch.ones(v) if ("norm" in k and k.endswith("weight")) else rn(*v)) for k, v in shapes.items()}

    def timeit(fn):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return min(ts)

    with torch.no_grad():
        scfg = osam.SamCfg(depth=2, global_idx=(1,))
        ssd = fill(seeded.sam_shapes(scfg, pfx=""))
        x = rn(1, 64, 64, 1280) * 50
        t_win = timeit(lambda: osam.sam_block(ssd, "blocks.0.", x, scfg, 14))
        t_glob = timeit(lambda: osam.sam_block(ssd, "blocks.1.", x, scfg, 0))
        s0 = osam.SamCfg(depth=0)
        s0sd = fill(seeded.sam_shapes(s0, pfx=""))
        img = torch.randn(1, 3, 1024, 1024)
        t_sam0 = timeit(lambda: osam.sam_image_encoder(s0sd, "", img, s0))
        lcfg = ol.LlamaCfg(layers=1, vocab=8)
        lsd = fill(seeded.llama_shapes(lcfg))
        h = rn(1, 319, 4096) * 50
        cos, sin = ol.rope_tables(319, 128, 1e4, "cpu")
        mask = ol.additive_mask(torch.ones(1, 319, dtype=torch.bool), 319, torch.float32, "cpu")
        t_llama = timeit(lambda: ol.decoder_layer(lsd, "model.layers.0.", h, mask, cos, sin, lcfg))
        ccfg = ovit.VitCfg(layers=2)
        csd = fill(seeded.clip_shapes(ccfg, pfx=""))
        ic = torch.randn(1, 3, 224, 224)
        t_clip2 = timeit(lambda: ovit.clip_vision_features(csd, "", ic, ccfg, select_layer=2))
        wl = rn(32004, 4096)
        t_lm = timeit(lambda: F.linear(h[0], wl))
        hsd = fill(seeded.head_shapes(256))
        feat = rn(1, 256, 64, 64) * 50
        segs = (torch.rand(256, 256, 256) > 0.7).float()
        txt = rn(1, 256) * 50

        def head():
            up = oh.upsample_feats(feat, 256)
            return oh.mask_head(hsd, "model.", oh.mask_pooling(up[0], segs), txt)
        t_head = timeit(head)
    t_img = t_sam0 + 28 * t_win + 4 * t_glob + 32 * t_llama + 23 / 2 * t_clip2 + t_lm + t_head
    return {"value": 1.0 / t_img, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": ("fp32 oracle, 1 image: measured 1 windowed + 1 global SAM-H block, patch-embed+neck, 1 Llama-7B layer "
                       "(T=319), 2 CLIP-L layers, lm_head, mask-pool+head (K=256); scaled to 28+4 / 32 / 23 layers "
                       f"(sample CPU time {t_sam0 + t_win + t_glob + t_llama + t_clip2 + t_lm + t_head:.1f}s x{reps + 1} runs)")}


def attention_flops(cfg, B, T):
    """Dense attention FLOPs per step (heads * 2*Nq*Nk*hd * 2), as SURVEY.md §8d counts them."""
    f = 0.0
    s = cfg.sam
    if cfg.backbone == "sam":
        hd = s.dim // s.heads
        n_glob = len([i for i in range(s.depth) if i in s.global_idx])
        nw = ((s.grid + s.window - 1) // s.window) ** 2
        f += n_glob * s.heads * 4.0 * (s.grid ** 2) ** 2 * hd
        f += (s.depth - n_glob) * s.heads * nw * 4.0 * (s.window ** 2) ** 2 * hd
    c = cfg.clip
    n_clip = c.layers + 1 + cfg.select_layer
    f += n_clip * c.heads * 4.0 * 257 ** 2 * (c.dim // c.heads)
    l = cfg.llama
    f += l.layers * l.heads * 4.0 * T * T * l.head_dim
    return f * B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=24,
                    help="images per rank per step (24: every Llama GEMM, M = 24 x 319 rows, fills whole rounds of 256 x 256 tiles)")
    ap.add_argument("--backbone", default="sam", choices=["sam", "dinov2"])
    ap.add_argument("--masks", type=int, default=256)
    ap.add_argument("--prompt-len", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--small", action="store_true", help="reduced depth (debug only; result marked invalid)")
    ap.add_argument("--mode", default="train", choices=["fwd", "train"],
                    help="train (default; BASELINE metric 'fwd+bwd'): fwd+bwd with LoRA r=8 + DDP, optimizer step every --accum steps, and a "
                         "forward-only pass (BASELINE configs[1]) reported under 'fwd_only'; fwd: forward only")
    ap.add_argument("--force-ddp", action="store_true", help="wrap in DDP even at world size 1 (exercises the reducer on one GPU)")
    ap.add_argument("--accum", type=int, default=10, help="gradient-accumulation micro-steps per optimizer step (reference: 10)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.force_ddp:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from llmseg_amd import ops, synthetic
    from llmseg_amd.lisa import LISAForCausalLM
    from llmseg_amd.params import LisaConfig, LlamaConfig, SamConfig, VitConfig

    cfg = LisaConfig(backbone=args.backbone, build_unused_towers=False)
    train = args.mode == "train"
    if train:
        cfg.llama = LlamaConfig(lora_r=8)
    if args.small:
        cfg.llama = LlamaConfig(layers=2, lora_r=8 if train else 0)
        cfg.sam = SamConfig(depth=2, global_idx=(1,))
        cfg.clip = VitConfig(layers=3)
    model = LISAForCausalLM(cfg, device=dev).init_random(seed=0)
    model.prepare()
    img = 1024 if args.backbone == "sam" else 896
    batch = synthetic.make_batch(args.batch, img_size=img, L=args.prompt_len, K=args.masks, device=dev, seed=1234 + rank)

    if train:
        from llmseg_amd.train import Trainer
        model.set_trainable()
        trainer = Trainer(model, lr=3e-4, grad_accum=args.accum, device_ids=[local], force_ddp=args.force_ddp)

        def step():
            return trainer.micro_step(batch)
    else:
        def step():
            with torch.no_grad():
                return model.model_forward(**batch, inference=False)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    ops.prof_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.prof_enable(False)
    prof = ops.prof_collect()
    gemm_ms, gemm_flops, gemm_launches = prof["all"]
    dom_ms, dom_flops, dom_launches = prof["dominant"]
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(out["loss"].detach())
    assert loss == loss, "NaN loss"

    fwd_only = None
    if train:                                   # BASELINE configs[1]: the same batch, forward only (no grad)
        def fstep():
            with torch.no_grad():
                return model.model_forward(**batch, inference=False)
        for _ in range(max(1, args.warmup)):
            fstep()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        f0 = time.perf_counter()
        for _ in range(args.steps):
            fstep()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        fdt = time.perf_counter() - f0
        if dist:
            t = torch.tensor([fdt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            fdt = float(t.item())
        fwd_only = {"value": args.batch * world * args.steps / fdt, "unit": "images/s", "ms_per_step": fdt / args.steps * 1e3,
                    "workload": "BASELINE.json configs[1]: forward-only model_forward, same batch"}

    if rank == 0:
        T = args.prompt_len - 1 + cfg.n_img_tokens
        ms = dt / args.steps * 1e3
        total_imgs = args.batch * world * args.steps
        ach = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        ach_all = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        model_flops = gemm_flops / args.steps + attention_flops(cfg, args.batch, T)
        res = {
            "metric": "images/sec (1024x1024, 64-tok prompt) model_forward " + ("fwd+bwd" if train else "fwd"), "value": total_imgs / dt,
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[1]: synthetic %dx%d + random-init LLaVA-7B(Llama-7B+CLIP-L/14)/%s, "
                                    "%s, %d candidate masks, %d-token prompt") % (
                img, img, "SAM-ViT-H" if args.backbone == "sam" else "DINOv2-L",
                ("fwd+bwd train step (CE + align + IoP losses, LoRA r=8 on q/v + trainable embed/lm_head/text_fcs/lisa_*, frozen towers, "
                 "torch DDP, AdamW step every %d micro-steps inside the timed region)" % args.accum) if train else
                "forward-only model_forward (training-mode forward incl. lm_head+CE+align+IoP losses, no backward)",
                args.masks, args.prompt_len),
                       "images_per_gpu_per_step": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "valid": not args.small},
            # dominant kernel = the GEMM kernel class with the largest total time (the 256x256 ping-pong LDS-DMA GEMM on this
            # workload); achieved = its algorithmic 2MNK per launch / its HIP-event duration
            "roofline": {"bound": "mfma", "kernel": prof["dominant_kernel"], "achieved": ach, "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": None, "launches_per_step": dom_launches / args.steps,
                         "avg_launch_us": dom_ms * 1e3 / max(1, dom_launches), "time_share_of_step": dom_ms / (dt * 1e3),
                         "all_gemm_kernels": {"achieved": ach_all, "launches_per_step": gemm_launches / args.steps,
                                              "avg_launch_us": gemm_ms * 1e3 / max(1, gemm_launches), "time_share_of_step": gemm_ms / (dt * 1e3)}},
            "model_tflop_per_image": model_flops / args.batch / 1e12,
            "model_mfma_frac": model_flops / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
            "loss": loss,
            "peak_hbm_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
        }
        if fwd_only is not None:
            res["fwd_only"] = fwd_only
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
    else:
        res = None
    if dist:
        dist.barrier()
        dist.destroy_process_group()      # RCCL prints its version banner on teardown: keep the JSON line the LAST line of stdout
    if res is not None:
        sys.stdout.flush()
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
