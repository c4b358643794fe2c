"""GPU (one MI355X): the distributed plumbing of the training step on a world-size-1 RCCL process group -- the gradient all-reduce of
the fp32 arena, the torch-DDP-wrapper alternative, and `bench.py` launched exactly as the driver launches it for N > 1
(`python -m torch.distributed.run ... bench.py --gpus N`).  Multi-rank arithmetic is covered on the CPU (tests/test_train_cpu.py, gloo)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_window(mode, steps=4, accum=2):
    """mode: 'arena' (no process group) | 'arena_dist' (all-reduce on a world-1 group) | 'grad' (plain .grad) | 'ddp' (DDP wrapper)"""
    from llmseg_amd.train import Trainer
    from tests import backward_checks as bc, model_checks as mc
    cfg, m, sd, batch = bc._lora_case("sam", p_drop=0.05)
    m.set_dropout_seed(5, 0)
    kw = dict(lr=1e-3, grad_accum=accum, warmup=0, total_steps=50)
    if mode in ("grad", "ddp"):
        tr = Trainer(m, ddp_wrapper=True, force_ddp=(mode == "ddp"), device_ids=[0], **kw)
        assert (tr.ddp is not None) == (mode == "ddp") and tr.arena is None
    else:
        tr = Trainer(m, **kw)
        assert tr.arena is not None and tr.dist_on == (mode == "arena_dist")
    db = mc._dev(batch)
    losses = []
    for _ in range(steps):
        if tr.arena is None and cfg.llama.lora_dropout > 0:
            m.advance_dropout()
        losses.append(float(tr.micro_step(db)["loss"]))
    params = torch.cat([w.detach().flatten().cpu() for w in tr.opt.master])
    tr.close()
    return losses, params, tr.opt_steps


def test_world1_process_group_paths():
    import torch.distributed as dist
    l0, p0, n0 = _run_window("arena")
    g0, q0, _ = _run_window("grad")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        l1, p1, n1 = _run_window("arena_dist")
        g1, q1, _ = _run_window("ddp")
    finally:
        dist.destroy_process_group()
    assert n0 == n1 == 2
    # the all-reduce over one rank is the identity: same losses and the same parameters (fp32 atomics make the last bits vary run to run)
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 2e-3, (l0, l1)
    assert (p0 - p1).abs().max().item() < 2e-3 * 1e-3 + 1e-4, (p0 - p1).abs().max().item()
    assert max(abs(a - b) for a, b in zip(g0, g1)) < 2e-3, (g0, g1)
    assert (q0 - q1).abs().max().item() < 2e-4, (q0 - q1).abs().max().item()
    # arena (fp32 accumulation) vs bf16 .grad accumulation: the same training trajectory up to bf16 gradient rounding
    assert max(abs(a - b) for a, b in zip(l0, g0)) < 5e-2, (l0, g0)


def test_bench_under_torchrun_world1():
    """bench.py through `torch.distributed.run` (1 rank) with the process group forced on: the launch line, env parsing, barrier /
    all_reduce(MAX) timing, teardown and the JSON contract."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29573",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--small", "--extra-batch", "0", "--no-cpu-baseline", "--force-dist"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["config"]["graph"] is True, d
