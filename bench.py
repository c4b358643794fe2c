#!/usr/bin/env python
"""Benchmark of the LLM-Seg `model_forward` hot path on MI355X (contract: see the task's bench.py section).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one fwd+bwd pass of `LISAForCausalLM.model_forward` over one synthetic micro-batch of `--batch` images per rank:
BASELINE.json configs[2] (synthetic 1024x1024 images + random-init LLaVA-7B / CLIP-L / SAM ViT-H, 64-token prompt, 256 candidate
masks, bf16, LoRA r=8 with dropout 0.05 on q/v, CE + align + IoP losses, batch_size = 2, AdamW step every 10 micro-steps inside the
timed region).  The same batch forward-only is BASELINE configs[1] (`fwd_only`).  Inputs are resident in HBM before the timed region.
The path shards over independent images: no data-path collective in the micro-step, one all-reduce of the fp32 gradient arena per
optimizer step, weak scaling.  One JSON line on rank 0.  `--extra-batch B` (default 24) adds the same measurement at B images per step
under `batch_<B>` (the throughput-optimal micro-batch on 288 GB of HBM; 0 = skip).
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


def cpu_baseline(threads=None, budget_s=45.0):
    """The CPU oracle (== reference arithmetic, checked against the imported reference in oracle/make_goldens.py) timed on the host
    cores.  This is the ONLY place bench.py touches oracle/."""
    from tools import cpu_baseline as cb
    return cb.run(threads=threads, budget_s=budget_s)


def pmc_traffic(kernel, B):
    """HBM bytes per user-level GEMM CALL of the dominant class from the fabric counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
    FETCH_SIZE x 2 on gfx950; tools/pmc_traffic.sh -> profiles/traffic.json, collected on an MI355X with this same bench command at the same batch).
    The roofline's class is "every call the dispatch sends to one tile kernel" (see `class_composition`): its traffic is the launch-weighted sum over that
    kernel's three forms -- `<false, ..>` bf16 output, `<true, ..>` fp32 K-slice slabs, `<false, true, ..>` with the LoRA extension tile -- plus the
    `splitk_reduce*` launches (attributed to the class that owns most K-sliced launches), divided by the number of calls (= launches of the three
    forms), i.e. the same population `algorithmic_bytes_per_launch` is averaged over (round 4 divided one form's bytes by the whole class's calls).
    The counters cannot be read from inside this process: (None, provenance) when no record for this batch is committed.
    -> (bytes per call | None, {"file", "sha1", "collected", "definition"})."""
    import hashlib
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        raw = open(path, "rb").read()
        doc = json.loads(raw)
    except (OSError, ValueError):
        return None, None
    rec = doc.get(f"batch_{B}", {})
    ks = rec.get("kernels", {})
    base = kernel.split("<")[0] + "<"
    forms = {k: v for k, v in ks.items() if k.startswith(base)}
    prov = {"file": "profiles/traffic.json", "sha1": hashlib.sha1(raw).hexdigest()[:16], "collected": rec.get("source"),
            "definition": "sum over the class's kernel forms (+ split-K reduce launches) of bytes x launches, / calls of the class"}
    if not forms:
        return None, prov
    calls = sum(v["launches"] for v in forms.values())
    total = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in forms.values())
    sliced = lambda b: sum(v["launches"] for k, v in ks.items() if k.startswith(b) and k[len(b):].startswith("true"))
    others = [k.split("<")[0] + "<" for k in ks if k.startswith("gemm_bf16_tn_pp") and not k.startswith(base)]
    reds = [v for k, v in ks.items() if k.startswith("splitk_reduce") or k.startswith("reduce_lora_normbwd")]     # the plain reduce, its reduce + RMSNorm form (`norm_out`) and the reduce + LoRA dX + norm-backward tail (`nb_x`)
    if reds and sliced(base) >= max([sliced(o) for o in set(others)] + [0]):
        total += sum(r["hbm_bytes_per_launch"] * r["launches"] for r in reds)
    prov["forms"] = {k: {"launches": v["launches"], "hbm_bytes_per_launch": v["hbm_bytes_per_launch"]} for k, v in forms.items()}
    return total / max(1, calls), prov


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


def _sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize()


def timed(step, steps, warmup, dist, dev, after_warmup=None):
    """W untimed + exactly K timed calls of step(), bracketed by barrier + synchronize; max over ranks.  -> (seconds, last output)
    after_warmup: untimed hook between the warm-up and the first barrier (first execution of code the warm-up steps do not reach)."""
    out = None
    for _ in range(warmup):
        out = step()
    if after_warmup is not None:
        after_warmup()
    _sync(dev)
    if dist:
        dist.barrier()
    _sync(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    _sync(dev)
    if dist:
        dist.barrier()
    _sync(dev)
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def measure(model, cfg, args, B, dev, dist, rank, world, local, use_graph, trainer_cls):
    """Train-mode (fwd+bwd, optimizer step every --accum micro-steps) and forward-only throughput at B images per rank per step, plus
    the per-kernel GEMM timing (HIP events on the launch stream, eager launches of the same step function)."""
    from llmseg_amd import ops, synthetic
    img = 1024 if args.backbone == "sam" else 896
    batch = synthetic.make_batch(B, img_size=img, L=args.prompt_len, K=args.masks, device=dev, seed=1234 + rank)
    plan = model.make_plan(**batch)
    res = {"images_per_gpu_per_step": B}
    T = args.prompt_len - 1 + cfg.n_img_tokens
    if args.mode == "train":
        trainer = trainer_cls(model, lr=3e-4, grad_accum=args.accum, device_ids=[local], use_graph=use_graph, ddp_wrapper=args.ddp_wrapper,
                              force_ddp=args.force_ddp, leaf_stream=args.leaf_stream, time_comm=dist is not None,
                              overlap_exchange=(True if args.overlap_exchange else False if args.no_overlap_exchange else None))
        # warm-up covers the eager warm-up calls of the graph path + the capture itself
        def first_optimizer_step():
            # the warm-up micro-steps never reach the optimizer (one step per --accum micro-steps): run it once untimed -- on a cold box its
            # first execution (kernel code pages, lazy AdamW state) cost 50 ms, 10 % of a 10-step timed region -- and restart the
            # accumulation window so that the timed region contains exactly steps / accum optimizer steps
            trainer.optimizer_step()
            trainer.micro = 0
            trainer.comm_times_ms()                                    # (drop the untimed step's exchange record)
        dt, out = timed(lambda: trainer.micro_step(batch, plan), args.steps, args.warmup + (3 if use_graph else 0), dist, dev, first_optimizer_step)
        loss = float(out["loss"].detach())
        assert loss == loss, "NaN loss"
        res.update(value=B * world * args.steps / dt, ms_per_step=dt / args.steps * 1e3, loss=loss,
                   graph=bool(use_graph and trainer.graph_error is None and any(e["graph"] is not None for e in trainer._graphs.values())))
        if trainer.graph_error:
            res["graph_error"] = trainer.graph_error[:200]
        if dist is not None and trainer.arena is not None:
            # what the data-parallel exchange costs: `exposed_ms` = compute-stream time per optimizer step from issuing the collectives to holding the
            # reduced arena + its norm (inside the timed region: nothing overlaps it); `allreduce_ms` = the same buffers exchanged alone
            # after the timed region (barrier + sync on both sides, max over ranks), dense and as shipped (embedding block as touched rows)
            exp_ms = trainer.comm_times_ms()
            flat = trainer.arena.flat
            def alone(fn, reps=3):
                fn(); _sync(dev); dist.barrier(); _sync(dev)
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                _sync(dev); dist.barrier(); _sync(dev)
                t = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item())
            scratch = torch.zeros_like(flat)
            dense_ms = alone(lambda: dist.all_reduce(scratch))
            del scratch
            eo, en = trainer.arena.block_of.get(trainer._embed_key, (0, 0))
            res["grad_exchange"] = {"optimizer_steps_timed": len(exp_ms), "exposed_ms_per_optimizer_step": (sum(exp_ms) / len(exp_ms)) if exp_ms else None,
                                    "allreduce_ms_dense_fp32_arena_alone": dense_ms, "arena_bytes": int(flat.numel() * 4), "embedding_block_bytes": int(en * 4),
                                    "sparse_embedding_rows": bool(trainer.sparse_embed), "world": world, "overlap_exchange": bool(trainer.overlap_exchange),
                                    "wire_dtype": str(trainer.wire_dtype).replace("torch.", "") if trainer.wire_dtype is not None else "float32",
                                    "micro_steps_per_optimizer_step": args.accum, "share_of_optimizer_step": ((sum(exp_ms) / len(exp_ms)) / (dt / args.steps * 1e3 * args.accum)) if exp_ms else None,
                                    "note": "one exchange per optimizer step; without --overlap-exchange everything is issued after the last micro-step's backward, with it the arena's tail "
                                            "(lm_head, text_hidden_fcs, lisa_*: ~half of the arena) leaves between the two halves of that backward and `exposed_ms` covers the rest; the embedding "
                                            "table's block travels as an all-gather of its non-zero rows, the rest as asynchronous 128 MB all-reduce pieces"}
        # per-kernel timing: events cannot be recorded inside a replayed hipGraph, so the same micro-step runs eagerly (identical launches)
        # for a few steps right after the timed region, with an event pair around every GEMM launch on its stream
        trainer.use_graph = False
        trainer.leaf_stream = False
        overlap, model.overlap_towers = model.overlap_towers, False      # one stream: a launch's duration must not include another stream's kernels
        from llmseg_amd import _lib
        n0 = _lib.load().llmseg_launch_count()
        trainer._eager_step(batch, plan)                               # fwd + bwd only: never the optimizer, whatever --accum is; the window counter is untouched
        torch.cuda.synchronize()
        if trainer.arena is not None:                                  # (that extra gradient must not reach a later optimizer step)
            trainer.arena.zero_()
        else:
            for p_ in trainer.params:
                p_.grad = None
        res["launches_per_micro_step"] = {"library_kernels": int(_lib.load().llmseg_launch_count() - n0),
                                          "note": "kernels libllmseg_hip.so launches for one fwd+bwd micro-step (the hipGraph replays the same nodes plus "
                                                  "PyTorch's glue kernels: zero-fills, casts, the loss sum)"}
        ops.prof_enable(True)
        n_prof = min(args.steps, 3)
        for _ in range(n_prof):
            trainer.micro_step(batch, plan)
        torch.cuda.synchronize()
        ops.prof_enable(False)
        prof = ops.prof_collect()
        model.overlap_towers = overlap
        trainer.close()
        del trainer
    else:
        def fstep():
            with torch.no_grad():
                return model.model_forward(**batch, inference=False, plan=plan)
        dt, out = timed(fstep, args.steps, args.warmup, dist, dev)
        res.update(value=B * world * args.steps / dt, ms_per_step=dt / args.steps * 1e3, loss=float(out["loss"]), graph=False)
        overlap, model.overlap_towers = model.overlap_towers, False
        ops.prof_enable(True)
        n_prof = min(args.steps, 3)
        for _ in range(n_prof):
            fstep()
        torch.cuda.synchronize()
        ops.prof_enable(False)
        prof = ops.prof_collect()
        model.overlap_towers = overlap
    gemm_ms, gemm_flops, gemm_launches = prof["all"]
    dom_ms, dom_flops, dom_launches = prof["dominant"]
    ach = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    ach_all = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    oth_ms, oth_flops = gemm_ms - dom_ms, gemm_flops - dom_flops
    ach_oth = oth_flops / (oth_ms * 1e-3) / 1e12 if oth_ms > 0 else 0.0
    model_flops = gemm_flops / n_prof + attention_flops(cfg, B, T)
    traffic, traffic_src = pmc_traffic(prof["dominant_kernel"], B)
    di = prof.get("dominant_info", {})
    per = lambda key: di.get(key, 0) / n_prof
    composition = ("the class = every llmseg_gemm_bf16 CALL the dispatch sends to this tile kernel with bf16 output (a timed record spans the whole call): "
                   "%.0f calls per step, of which %.0f ran as K-slices (%.1f slices on average: ONE launch of the tile kernel's fp32-slab form -- rocprofv3 row "
                   "`...<true, ...>` -- with the slices as its batch index, + one `splitk_reduce_kernel` launch that applies the epilogue -- `splitk_reduce_rmsnorm_kernel` where the call also asks for the next RMSNorm, `norm_out`: that row's time is inside the call) and %.0f as a single "
                   "launch of the bf16-out form (row `...<false, false, 0>`) or its LoRA extension-tile form (`...<false, true, ..>`); round 6: the timed calls also carry the "
                   "pointwise work that used to be launches of its own -- RoPE / SwiGLU / SwiGLU-backward in the store of the `<.., 1|2|3>` forms, and the LoRA dX + pre-norm backward "
                   "in the K-slice reduce launch `reduce_lora_normbwd_kernel` -- so the fraction prices those epilogues as GEMM time; %.0f kernel launches per step in all"
                   % (per("calls"), per("calls_as_k_slices"), di.get("k_slices", 0) / max(1, di.get("calls_as_k_slices", 0)),
                      per("calls") - per("calls_as_k_slices"), per("kernel_launches"))) if di else None
    res["roofline"] = {"bound": "mfma", "kernel": prof["dominant_kernel"], "class_composition": composition,
                       "hottest_symbol": "by rocprofv3's per-symbol tables (profiles/r06p_b2_kernel_stats.md, r06p_b2_1stream_kernel_stats.md) the hottest single SYMBOL of the replayed step at 2 images is "
                                         "gemm_bf16_tn_pp_kernel<false, false, 4> (the 256 x 256 tile: the SAM products at M = 8192; ~0.35 of peak; 22-28 % of the kernel time), level with "
                                         "the K-slice form of the class below (`...pp2_kernel<true, false, 0>`, 21-22 %): its calls are counted under `other_gemm_classes` below.  `kernel` above names the dominant CLASS by total time = the 128 x 256 tile's three "
                                         "symbols (bf16-out, fp32 K-slice slabs, LoRA extension tile) + their reduce launches (the Llama products at M = 638)",
                       "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                       "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                       "algorithmic_bytes_per_launch": prof["dominant_alg_bytes"] / max(1, dom_launches), "launches_per_step": dom_launches / n_prof,
                       "launches_are": "user-level GEMM calls of the class (see class_composition); kernel launches per step: %.0f" % per("kernel_launches") if di else None,
                       "avg_launch_us": dom_ms * 1e3 / max(1, dom_launches), "time_share_of_step": dom_ms / n_prof / res["ms_per_step"],
                       "timing": "HIP events around every GEMM launch, %d eager single-stream steps of the same micro-step after the timed region" % n_prof,
                       "all_gemm_kernels": {"achieved": ach_all, "frac": ach_all / PEAK_BF16_TFLOPS, "launches_per_step": gemm_launches / n_prof,
                                            "avg_launch_us": gemm_ms * 1e3 / max(1, gemm_launches),
                                            "time_share_of_step": gemm_ms / n_prof / res["ms_per_step"]},
                       # every GEMM call that is NOT in the dominant class (at 2 images: the SAM products on the 256 x 256 tile kernel, the small CLIP / head
                       # products): a dispatch change that moves calls between classes moves `frac` without the step getting slower or faster -- the pair
                       # (frac, other_gemm_classes.frac) with their time shares is the whole picture
                       "other_gemm_classes": {"achieved": ach_oth, "frac": ach_oth / PEAK_BF16_TFLOPS, "launches_per_step": (gemm_launches - dom_launches) / n_prof,
                                              "time_share_of_step": oth_ms / n_prof / res["ms_per_step"]}}
    res["model_tflop_per_image"] = model_flops / B / 1e12
    res["model_mfma_frac"] = model_flops / (res["ms_per_step"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS
    if args.mode == "train" and not args.no_fwd_only:       # BASELINE configs[1]: the same batch, forward only (no grad)
        def fstep():
            with torch.no_grad():
                return model.model_forward(**batch, inference=False, plan=plan)
        fdt, _ = timed(fstep, args.steps, max(1, args.warmup), dist, dev)
        res["fwd_only"] = {"value": B * world * args.steps / fdt, "unit": "images/s", "ms_per_step": fdt / args.steps * 1e3,
                           "workload": "BASELINE.json configs[1]: forward-only model_forward (incl. lm_head + CE + align + IoP losses), same batch"}
    return res


def accum_fused(model, cfg, args, B, dev, dist, rank, world, local, trainer_cls):
    """The SAME optimizer step as the headline -- `--accum` micro-batches of B images, per-micro-batch loss normalisation, AdamW on the averaged
    gradient -- with the accumulation window folded into ONE pass (`Trainer(fused_accum=k)`, `make_plan(micro_batches=k)`): every GEMM runs at
    k x the rows (M = 638 -> 6380 at 2 x 10), the weights are read once per optimizer step instead of k times.  Gradient equality with the k
    micro-steps: tests/backward_checks.py::check_fused_accum.  A timed step = one pass over k micro-batches + the optimizer step.  Reported beside the
    headline, never as it: BASELINE configs[2] is batch_size = 2 per step."""
    from llmseg_amd import synthetic
    from llmseg_amd.train import merge_micro_batches
    img = 1024 if args.backbone == "sam" else 896
    k = args.accum
    merged = merge_micro_batches([synthetic.make_batch(B, img_size=img, L=args.prompt_len, K=args.masks, device=dev, seed=1234 + rank + 101 * j) for j in range(k)])
    plan = model.make_plan(**merged, micro_batches=k)
    trainer = trainer_cls(model, lr=3e-4, grad_accum=1, device_ids=[local], use_graph=True, fused_accum=k)
    steps = max(2, args.steps // 2)
    dt, out = timed(lambda: trainer.micro_step(merged, plan), steps, 4, dist, dev)
    res = {"value": B * k * world * steps / dt, "unit": "images/s", "ms_per_optimizer_step": dt / steps * 1e3, "ms_per_micro_batch": dt / steps / k * 1e3,
           "micro_batches_per_pass": k, "images_per_pass_per_gpu": B * k, "timed_optimizer_steps": steps, "loss_sum_over_micro_batches": float(out["loss"].detach()),
           "graph": bool(trainer.graph_error is None and any(e["graph"] is not None for e in trainer._graphs.values())),
           "what": "one fused fwd+bwd pass over the %d micro-batches of an optimizer step (%d images, Llama GEMMs at M = %d) + AdamW, replayed from a hipGraph; "
                   "same gradient as %d micro-steps (per-micro-batch CE / image means, per-micro-batch dropout masks)" % (k, B * k, B * k * (args.prompt_len - 1 + cfg.n_img_tokens), k)}
    if trainer.graph_error:
        res["graph_error"] = trainer.graph_error[:200]
    trainer.close()
    return res


def window_towers(model, cfg, args, B, dev, dist, rank, world, local, trainer_cls, convs=None):
    """The headline's optimizer step -- `--accum` micro-batches of B images, fwd+bwd each, AdamW on the accumulated gradient -- with the two FROZEN
    towers (SAM ViT-H, CLIP-L + projector) run ONCE per accumulation window over all accum x B images (`Trainer.window_step`): their GEMMs see
    M = accum x B x 4096 rows, the micro-steps (hipGraph replays) run CLIP-less / SAM-less with the tower outputs as inputs, and the towers of
    window w + 1 are issued on the side stream beside the micro-steps of window w.  Same gradient as the headline's micro-steps (features are
    inputs: tests/backward_checks.py::check_window_towers).  Timed: whole windows (tower pass + accum micro-steps + optimizer step).
    convs: conversations per image drawn per micro-batch (BASELINE configs[3]: batch 1, 1-3 conversations) instead of one per image."""
    from llmseg_amd import synthetic
    img = 1024 if args.backbone == "sam" else 896
    k = args.accum
    if convs is None:
        batches = [synthetic.make_batch(B, img_size=img, L=args.prompt_len, K=args.masks, device=dev, seed=1234 + rank + 101 * j) for j in range(k)]
    else:
        batches = [synthetic.make_batch(B, img_size=img, L=args.prompt_len, K=args.masks, device=dev, seed=555 + rank + 101 * j, convs=[c] * B) for j, c in enumerate(convs)]
    plans = [model.make_plan(**b) for b in batches]
    trainer = trainer_cls(model, lr=3e-4, grad_accum=k, device_ids=[local], use_graph=True)
    pending = [trainer.encode_window(batches, prefetch=True)]

    chain_only = os.environ.get("LLMSEG_BENCH_CHAIN_ONLY") is not None      # profiling aid: no tower pass at all (one stale result reused) = the trainable chain alone

    def window():
        tw = pending.pop()
        pending.append(tw if chain_only else trainer.encode_window(batches, prefetch=True))     # the next window's towers: beside this window's micro-steps
        return trainer.window_step(batches, plans, towers=tw)[-1]
    windows = max(2, args.steps // k)
    dt, out = timed(window, windows, 3, dist, dev)
    n_img = B * k
    res = {"value": n_img * world * windows / dt, "unit": "images/s", "ms_per_micro_step": dt / windows / k * 1e3, "ms_per_window": dt / windows * 1e3,
           "micro_batches_per_window": k, "images_per_tower_pass_per_gpu": n_img, "timed_windows": windows, "loss": float(out["loss"].detach()),
           "graphs": sum(1 for e in trainer._graphs.values() if e["graph"] is not None),
           "graph": bool(trainer.graph_error is None and any(e["graph"] is not None for e in trainer._graphs.values())),
           "what": "the same optimizer step as the headline (%d micro-batches of %d image(s), fwd+bwd + AdamW), frozen SAM-H / CLIP-L towers run once per window on "
                   "all %d images (prefetched on a side stream one window ahead), micro-steps replayed from hipGraphs with the tower outputs as inputs" % (k, B, n_img)}
    if convs is not None:
        res["conversations_per_micro_batch"] = list(convs)
    if trainer.graph_error:
        res["graph_error"] = trainer.graph_error[:200]
    trainer.close()
    return res


def mix_9_3_1(model, cfg, args, dev, dist, rank, world, local, trainer_cls):
    """BASELINE configs[3]'s per-GPU workload on synthetic data: batch_size = 1 image per micro-step, the source of every sample drawn 9:3:1 from
    sem_seg / refer_seg / reason_seg as `HybridDataset` draws it (utils/dataset.py:499-502), i.e. 1-3 conversations on the image (N = 1..3 sequences
    through CLIP + Llama, one SAM forward).  One hipGraph per batch structure (3 of them), grad-accum 10, AdamW inside the timed region."""
    from llmseg_amd import synthetic
    img = 1024 if args.backbone == "sam" else 896
    sampler = synthetic.HybridSampler((9, 3, 1), seed=2024 + rank)
    n = 3 * args.accum                                                        # three optimizer steps
    draws = [sampler.draw() for _ in range(n)]
    pool = {c: synthetic.make_batch(1, img_size=img, L=args.prompt_len, K=args.masks, device=dev, seed=555 + c + rank, convs=[c]) for c in (1, 2, 3)}
    plans = {c: model.make_plan(**b) for c, b in pool.items()}
    trainer = trainer_cls(model, lr=3e-4, grad_accum=args.accum, device_ids=[local], use_graph=True)
    for c in (1, 2, 3):                                                       # eager warm-ups + capture of every structure, outside the timed region
        for _ in range(4):
            trainer.micro_step(pool[c], plans[c])
    it = [0]

    def step():
        c = draws[it[0] % n][1]
        it[0] += 1
        return trainer.micro_step(pool[c], plans[c])

    def restart():
        trainer.optimizer_step()
        trainer.micro = 0
    dt, out = timed(step, n, 0, dist, dev, restart)
    counts = {s: sum(1 for d in draws if d[0] == s) for s in synthetic.HybridSampler.SOURCES}
    res = {"value": world * n / dt, "unit": "images/s", "ms_per_step": dt / n * 1e3, "timed_micro_steps": n, "batch_size_per_gpu": 1,
           "sources_drawn": counts, "conversations_drawn": {str(c): sum(1 for d in draws if d[1] == c) for c in (1, 2, 3)},
           "graphs": sum(1 for e in trainer._graphs.values() if e["graph"] is not None), "loss": float(out["loss"].detach()),
           "what": "BASELINE.json configs[3] per-GPU workload, synthetic: 1 image per micro-step, source drawn 9:3:1 (sem_seg / refer_seg / reason_seg) -> 1-3 "
                   "conversations per image, grad-accum %d, one hipGraph per batch structure" % args.accum}
    trainer.close()
    return res


def loader_in_loop(model, cfg, args, B, dev, dist, rank, world, local, trainer_cls, resident_ms):
    """The same fwd+bwd micro-step fed the way a data loader feeds it (reference training.py:521-532, utils/dataset.py:33-170):
    every micro-step gets a DIFFERENT batch -- token ids / labels / masks arrive as host tensors (what `collate_fn_new` returns), images and the
    image's dense SAM proposals sit in device memory -- and, inside the timed region, per micro-step: N2 on the device
    (`proposals_and_targets_dense`: top-K by area, IoU / IoP against the ground truth, antialiased 256 x 256 proposal maps), a fresh
    `make_plan` (host index plumbing, no device->host sync: the ids are host tensors), the copy of the batch into the hipGraph's input
    buffers, the graph replay.  The targets of micro-step i + 1 are issued on a second stream beside micro-step i (a loader prefetches).
    `input_ms` = what all of that adds to the resident-batch step; `targets_gpu_ms` = the N2 kernels' own duration per micro-step."""
    import time
    from llmseg_amd import synthetic, targets
    img = 1024 if args.backbone == "sam" else 896
    R = 3
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    sets = []
    for r in range(R):
        b = synthetic.make_batch(B, img_size=img, L=args.prompt_len, K=args.masks, device=dev, seed=777 + 13 * r + rank)
        host = {k: b[k].cpu() for k in ("input_ids", "labels", "attention_masks", "offset")}
        # dense proposals of every image at the image's resolution (what SAM everything mode emits, N1) + one ground-truth mask
        props = [(torch.rand((args.masks, img, img), device=dev, generator=g) > 0.7).to(torch.uint8) for _ in range(B)]
        areas = [p.flatten(1).sum(1) for p in props]
        gts = [(torch.rand((img, img), device=dev, generator=g) > 0.6).to(torch.uint8) for _ in range(B)]
        sets.append((b, host, props, areas, gts))
    trainer = trainer_cls(model, lr=3e-4, grad_accum=args.accum, device_ids=[local], use_graph=True)
    t_plan = [0.0, 0]
    it = [0]
    n2_events = []

    side = torch.cuda.Stream(device=dev)

    def prepare(i):
        """The loader's device-side stage for micro-step i, issued on its own stream so that it runs beside the previous micro-step (a loader
        prefetches): N2 targets of every image.  -> (targets, event)"""
        b, host, props, areas, gts = sets[i % R]
        with torch.cuda.stream(side):                     # no dependence on the main stream: the proposals are resident, the outputs are fresh tensors
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            tg = [targets.proposals_and_targets_dense(props[j], areas[j], [gts[j]], top=args.masks, want_origin=False) for j in range(B)]
            e1.record()
        n2_events.append((e0, e1))
        return tg, e1

    pending = [prepare(0)]

    def step():
        b, host, props, areas, gts = sets[it[0] % R]
        tg, ready = pending[0]
        cur = torch.cuda.current_stream()
        cur.wait_event(ready)
        for t_ in tg:
            for k in ("sam_segs", "sam_ious", "sam_iops"):
                t_[k].record_stream(cur)
        it[0] += 1
        batch = dict(b)
        batch["sam_segs_list"] = [t["sam_segs"] for t in tg]
        batch["sam_ious_list"] = [t["sam_ious"] for t in tg]
        batch["sam_iops_list"] = [t["sam_iops"] for t in tg]
        t0 = time.perf_counter()
        plan = model.make_plan(host["input_ids"], host["labels"], host["attention_masks"], host["offset"], sam_segs_list=batch["sam_segs_list"])
        t_plan[0] += time.perf_counter() - t0
        t_plan[1] += 1
        out = trainer.micro_step(batch, plan)
        pending[0] = prepare(it[0])                       # the next micro-step's targets run beside this one's graph
        return out

    def first_optimizer_step():
        trainer.optimizer_step()
        trainer.micro = 0
        t_plan[0], t_plan[1] = 0.0, 0
    dt, out = timed(step, args.steps, args.warmup + 3, dist, dev, first_optimizer_step)
    ms = dt / args.steps * 1e3
    res = {"value": B * world * args.steps / dt, "unit": "images/s", "ms_per_step": ms, "input_ms": ms - resident_ms,
           "make_plan_host_ms": t_plan[0] / max(1, t_plan[1]) * 1e3, "distinct_batches": R,
           "targets_gpu_ms": sum(a.elapsed_time(b_) for a, b_ in n2_events[-args.steps:]) / max(1, min(args.steps, len(n2_events))),
           "graph": bool(trainer.graph_error is None and any(e["graph"] is not None for e in trainer._graphs.values())), "loss": float(out["loss"].detach()),
           "what": "per micro-step inside the timed region: proposals_and_targets_dense on the device for every image (%d dense proposals at %dx%d; "
                   "issued on a second stream one micro-step ahead), make_plan from host token tensors, copy into the captured graph's inputs, replay" % (args.masks, img, img)}
    trainer.close()
    return res


def neighbours(model, cfg, dev, prompt_len):
    """Side measurements of the rows next to the path, same model, outside the timed region: N3 = `evaluate()`'s generation
    (KV-cache decode, ms per token vs the 13.2 GB weight stream), N1 = SAM everything mode from the image embedding (32 x 32 point grid).
    Never fatal for the benchmark line."""
    import time
    out = {}
    try:
        model.eval()
        g = torch.Generator().manual_seed(7)
        ids = torch.randint(3, 31999, (1, prompt_len), generator=g)
        ids[:, 0], ids[:, 1], ids[:, 2], ids[:, 3] = 1, 32001, -200, 32002
        clip = torch.randn(1, 3, 224, 224, generator=g).to(dev, torch.bfloat16)
        c = cfg.llama
        wbytes = 2.0 * (c.layers * (4 * c.hidden * c.hidden + 3 * c.hidden * c.inter) + c.vocab * c.hidden)

        def run(n):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            model.generate(clip, ids, max_new_tokens=n, eos_token_id=None)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        run(3)                                                   # warm-up + graph capture
        t_a, t_b = min(run(2), run(2)), min(run(18), run(18))
        per = (t_b - t_a) / 16
        out["generation"] = {"ms_per_token": per * 1e3, "tokens_per_s": 1.0 / per, "prefill_plus_first_token_ms": t_a * 1e3 - per * 1e3, "sequences": 1,
                              "weight_stream_gb_per_token": wbytes / 1e9, "hbm_tb_per_s": wbytes / per / 1e12, "frac_of_8tbps": wbytes / per / 8e12,
                              "what": "LISAForCausalLM.generate (evaluate()'s greedy decode, KV cache, LoRA r=8): (18-token run - 2-token run) / 16"}
        img = torch.randn(1, 3, 1024, 1024, generator=g).to(dev, torch.bfloat16)
        feats = model._sam_encoder_cl(img)
        kw = dict(points_per_side=32, pred_iou_thresh=-1e9, stability_score_thresh=0.0, stability_score_offset=0.02)
        model.generate_proposals(feats, (1024, 1024), (1024, 1024), **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rec = model.generate_proposals(feats, (1024, 1024), (1024, 1024), **kw)
        torch.cuda.synchronize()
        out["everything_mode"] = {"ms_per_image": (time.perf_counter() - t0) * 1e3, "points": 1024, "candidates": 3072, "records": int(rec["masks"].shape[0]),
                                   "what": "generate_proposals from the image embedding, 1024x1024 original: 1024 point prompts through the multimask "
                                           "decoder, statistics pass over all 3072 candidates (filters opened), NMS, binarisation of the survivors; random "
                                           "decoder weights, so the survivor count is not representative (profiles/r02_amg.md: 52 ms with ~1000 records)"}
    except Exception as e:      # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"[:300]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2, help="images per rank per micro-step (BASELINE configs[2]: batch_size=2)")
    ap.add_argument("--extra-batch", type=int, default=24,
                    help="also measure this many images per micro-step (24: every Llama GEMM, M = 24 x 319 rows, fills whole rounds of "
                         "256 x 256 tiles) and report it under batch_<B>; 0 = skip")
    ap.add_argument("--backbone", default="sam", choices=["sam", "dinov2"])
    ap.add_argument("--masks", type=int, default=256)
    ap.add_argument("--prompt-len", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--small", action="store_true", help="reduced depth (debug only; result marked invalid)")
    ap.add_argument("--mode", default="train", choices=["fwd", "train"],
                    help="train (default; BASELINE metric 'fwd+bwd'): fwd+bwd with LoRA r=8, optimizer step every --accum steps, and a "
                         "forward-only pass (BASELINE configs[1]) reported under 'fwd_only'; fwd: forward only")
    ap.add_argument("--no-fwd-only", action="store_true", help="skip the forward-only measurement (profiling runs)")
    ap.add_argument("--no-overlap", action="store_true", help="issue the frozen backbone on the main stream instead of its own HIP stream")
    ap.add_argument("--leaf-stream", action="store_true", help="issue the arena's weight-gradient kernels on a side stream (A/B of autograd.Leaves; measured slower)")
    ap.add_argument("--overlap-exchange", action="store_true", help="cut every backward at the Llama output and issue the arena tail's all-reduce between the halves of a window's last micro-step (Trainer(overlap_exchange=True))")
    ap.add_argument("--multirank-defaults", action="store_true", help="(with --force-dist on one GPU) build every Trainer with the defaults of a > 1-rank run: embedding rows, bf16 wire, overlapped exchange")
    ap.add_argument("--no-overlap-exchange", action="store_true", help="exchange everything after the window's last backward (the default under > 1 rank is to overlap the arena's tail with it)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying the captured hipGraph")
    ap.add_argument("--ddp-wrapper", action="store_true", help="torch DDP wrapper + bf16 .grad instead of the fp32 gradient arena")
    ap.add_argument("--force-ddp", "--force-dist", dest="force_ddp", action="store_true",
                    help="initialise the RCCL process group even at world size 1 (the gradient all-reduce / the DDP wrapper then run on one GPU)")
    ap.add_argument("--lora-dropout", type=float, default=0.05, help="peft lora_dropout (reference training.py:91)")
    ap.add_argument("--no-neighbours", action="store_true", help="skip the generation / everything-mode side measurements (SURVEY.md 8f N3, N1)")
    ap.add_argument("--accum", type=int, default=10, help="gradient-accumulation micro-steps per optimizer step (reference: 10)")
    ap.add_argument("--no-k512", action="store_true", help="skip the BASELINE configs[4] side measurement (512 candidate masks, grad-accum 8) reported under batch_<B>_k512")
    ap.add_argument("--no-accum-fused", action="store_true", help="skip the fused-accumulation-window side measurement (the --accum micro-batches of an optimizer step as one pass)")
    ap.add_argument("--window-towers-only", action="store_true", help="profiling aid: measure only the window-towers side line")
    ap.add_argument("--no-window-towers", action="store_true", help="skip the side measurement with the frozen towers batched over the accumulation window (Trainer.window_step)")
    ap.add_argument("--no-mix", action="store_true", help="skip the BASELINE configs[3] side measurement (batch 1, sources drawn 9:3:1 -> 1-3 conversations per image)")
    ap.add_argument("--no-loader", action="store_true", help="skip the loader-in-the-loop side measurement (a different batch + device-side targets + a fresh plan every micro-step)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start N ranks ourselves (one process per GPU, the launch line the driver uses) and
        # let rank 0 of THAT job print the line -- never time one GPU and report it as N (reference launch: scripts/train_10epoch.sh:10-22)
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} requested but this node exposes {have} GPU(s); refusing to report a {args.gpus}-GPU figure")
        import subprocess
        # --standalone: the launcher picks a free rendezvous port itself (binding one here and releasing it can race with another process)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; the two must agree")
    if local >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} (LOCAL_RANK {local}) has no GPU: this node exposes {torch.cuda.device_count()}")
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
        assert dist.get_world_size() == world and dist.get_rank() == rank

    from llmseg_amd.lisa import LISAForCausalLM
    from llmseg_amd.params import LisaConfig, LlamaConfig, SamConfig, VitConfig
    from llmseg_amd.train import Trainer as _Trainer
    if args.multirank_defaults:
        # one GPU, world-1 RCCL group, but every Trainer of this run is built with what a Trainer under > 1 rank defaults to (embedding block as rows, bf16 on the
        # wire, the exchange overlapped with the backward): the code path the driver's N > 1 runs take, with identity collectives
        assert dist is not None, "--multirank-defaults needs --force-dist"

        def Trainer(*a, **kw):
            for k, v in (("sparse_embed", True), ("wire_dtype", torch.bfloat16), ("overlap_exchange", True)):
                if kw.get(k) is None:
                    kw[k] = v
            return _Trainer(*a, **kw)
    else:
        Trainer = _Trainer

    cfg = LisaConfig(backbone=args.backbone, build_unused_towers=False, sam_decoder=(args.backbone == "sam"))
    train = args.mode == "train"
    if train:
        cfg.llama = LlamaConfig(lora_r=8, lora_dropout=args.lora_dropout)
    if args.small:
        cfg.llama = LlamaConfig(layers=2, lora_r=8 if train else 0, lora_dropout=args.lora_dropout if train else 0.0)
        cfg.sam = SamConfig(depth=2, global_idx=(1,))
        cfg.clip = VitConfig(layers=3)
    model = LISAForCausalLM(cfg, device=dev).init_random(seed=0)
    model.prepare()
    model.overlap_towers = not args.no_overlap
    if os.environ.get("LLMSEG_CE_FULL"):
        model.ce_gather_first = False
    if train:
        model.set_trainable()
    use_graph = train and not args.no_graph and not args.ddp_wrapper

    if args.window_towers_only:                # profiling aid (tools/r06_final.sh): ONLY the window-towers side line, printed as its own small JSON line
        r = window_towers(model, cfg, args, args.batch, dev, dist, rank, world, local, Trainer)
        if rank == 0:
            print(json.dumps({"metric": "images/sec (1024x1024, 64-tok prompt) model_forward fwd+bwd, frozen towers once per accumulation window (side line)", "value": r["value"],
                              "unit": "images/s", "n_gpus": world, "side_line": True, "window_towers": r}), flush=True)
        if dist:
            dist.barrier(); dist.destroy_process_group()
        return
    main_res = measure(model, cfg, args, args.batch, dev, dist, rank, world, local, use_graph, Trainer)
    extra = None
    if args.extra_batch and args.extra_batch != args.batch:
        extra = measure(model, cfg, args, args.extra_batch, dev, dist, rank, world, local, use_graph, Trainer)
    k512 = None
    if train and not args.no_k512 and not args.small:
        # BASELINE configs[4]'s workload shape on this model: 512 candidate masks per image, gradient accumulation 8 (training.py:79-82)
        a4 = argparse.Namespace(**vars(args))
        a4.masks, a4.accum, a4.no_fwd_only = 512, 8, True
        a4.steps = max(a4.accum, (args.steps // a4.accum) * a4.accum)           # whole accumulation windows inside the timed region
        k512 = measure(model, cfg, a4, args.batch, dev, dist, rank, world, local, use_graph, Trainer)
        k512["workload"] = "BASELINE.json configs[4] shape: %d candidate masks per image, grad-accum %d, %d timed micro-steps" % (a4.masks, a4.accum, a4.steps)
    fused = None
    if train and use_graph and not args.no_accum_fused and args.accum > 1:
        fused = accum_fused(model, cfg, args, args.batch, dev, dist, rank, world, local, Trainer)
    wtow = None
    if train and use_graph and not args.no_window_towers and args.accum > 1:
        wtow = window_towers(model, cfg, args, args.batch, dev, dist, rank, world, local, Trainer)
    mix = None
    if train and use_graph and not args.no_mix and not args.small:
        mix = mix_9_3_1(model, cfg, args, dev, dist, rank, world, local, Trainer)
        if not args.no_window_towers:
            from llmseg_amd import synthetic
            sampler = synthetic.HybridSampler((9, 3, 1), seed=2024 + rank)
            mix["window_towers"] = window_towers(model, cfg, args, 1, dev, dist, rank, world, local, Trainer, convs=[sampler.draw()[1] for _ in range(args.accum)])
    loader = None
    if train and use_graph and not args.no_loader and not args.small:
        loader = loader_in_loop(model, cfg, args, args.batch, dev, dist, rank, world, local, Trainer, main_res["ms_per_step"])

    if rank == 0:
        img = 1024 if args.backbone == "sam" else 896
        what = ("fwd+bwd train step (CE + align + IoP losses, LoRA r=8 dropout %.2f on q/v + trainable embed/lm_head/text_fcs/lisa_*, frozen towers, "
                "fp32 gradient arena, AdamW step every %d micro-steps inside the timed region%s)" % (
                    args.lora_dropout, args.accum, ", micro-step replayed from a hipGraph" if main_res.get("graph") else "")) if train else \
            "forward-only model_forward (training-mode forward incl. lm_head+CE+align+IoP losses, no backward)"
        res = {
            "metric": "images/sec (1024x1024, 64-tok prompt) model_forward " + ("fwd+bwd" if train else "fwd"), "value": main_res["value"],
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[%d]: synthetic %dx%d + random-init LLaVA-7B(Llama-7B+CLIP-L/14)/%s, batch_size=%d per GPU, "
                                    "%s, %d candidate masks, %d-token prompt") % (
                2 if train else 1, img, img, "SAM-ViT-H" if args.backbone == "sam" else "DINOv2-L", args.batch, what, args.masks, args.prompt_len),
                       "images_per_gpu_per_step": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "process_group_world_size": (dist.get_world_size() if dist else None),     # what RCCL reports (None: no process group at 1 GPU)
                       "graph": main_res.get("graph", False), "valid": not args.small},
            # dominant kernel = the GEMM kernel class with the largest total time; achieved = its algorithmic 2MNK per launch / its
            # HIP-event duration
            "roofline": main_res["roofline"],
            "model_tflop_per_image": main_res["model_tflop_per_image"],
            "model_mfma_frac": main_res["model_mfma_frac"],
            "loss": main_res["loss"],
            "launches_per_micro_step": main_res.get("launches_per_micro_step"),
            "grad_exchange": main_res.get("grad_exchange"),
            "peak_hbm_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
        }
        if "graph_error" in main_res:
            res["graph_error"] = main_res["graph_error"]
        if "fwd_only" in main_res:
            res["fwd_only"] = main_res["fwd_only"]
        if extra is not None:
            res[f"batch_{args.extra_batch}"] = extra
        if k512 is not None:
            res[f"batch_{args.batch}_k512"] = k512
        if fused is not None:
            res["accum_fused"] = fused
        if wtow is not None:
            res["window_towers"] = wtow
        if mix is not None:
            res["mix_9_3_1_batch_1"] = mix
        if loader is not None:
            res["loader_in_loop"] = loader
        if world == 1 and not args.no_neighbours and not args.small and args.backbone == "sam":
            res["neighbours"] = neighbours(model, cfg, dev, args.prompt_len)
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
