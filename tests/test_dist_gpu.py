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
        tr = Trainer(m, overlap_exchange=(mode == "arena_dist_overlap"), time_comm=(mode == "arena_dist_overlap"), **kw)
        assert tr.arena is not None and tr.dist_on == mode.startswith("arena_dist")
    db = mc._dev(batch)
    losses = []
    for _ in range(steps):
        losses.append(float(tr.micro_step(db)["loss"]))       # the trainer advances the dropout offset in every gradient mode
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
        l2, p2, n2 = _run_window("arena_dist_overlap")      # the tail's all-reduce issued between the two halves of the window's last backward
        g1, q1, _ = _run_window("ddp")
    finally:
        dist.destroy_process_group()
    assert n0 == n1 == 2
    # the all-reduce over one rank is the identity and no kernel sums with atomics: the same losses and parameters, bit for bit
    assert l0 == l1, (l0, l1)
    assert torch.equal(p0, p1), (p0 - p1).abs().max().item()
    assert l0 == l2 and n2 == 2 and torch.equal(p0, p2), ((p0 - p2).abs().max().item(), l0, l2)
    assert g0 == g1, (g0, g1)
    assert torch.equal(q0, q1), (q0 - q1).abs().max().item()
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


# ---------------------------------------------------------------------------------------------------------------------------------
# Two ranks of the REAL HIP arena Trainer sharing cuda:0 over gloo (no 2-GPU box is available to the build): init broadcast, per-rank
# dropout streams, the chunked asynchronous all-reduce of the fp32 arena, replicas bit-identical and equal to one process that
# accumulates the union of both ranks' micro-batches.  Reference: training.py:292-332, 369-381 (DeepSpeed engine = broadcast + DP all-reduce).
SEED, LR, ACCUM, OPT_STEPS = 77, 2e-3, 2, 2


def _rank_batch(batch, r):
    """Rank r's micro-batch: rank 0 the seeded case, rank 1 a different one of the same structure."""
    if r == 0:
        return batch
    b = dict(batch)
    b["images"] = batch["images"].flip(0) * 0.5
    b["images_clip"] = batch["images_clip"].flip(-1)
    b["sam_segs_list"] = [t.flip(1) for t in batch["sam_segs_list"]]
    b["sam_ious_list"] = [1.0 - t for t in batch["sam_ious_list"]]
    return b


def _two_rank_worker(rank, world, port, ret, tmp, wire):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from llmseg_amd.train import Trainer, rank_dropout_seed
    from tests import backward_checks as bc, model_checks as mc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, m, sd, batch = bc._lora_case("sam", p_drop=0.05)
        if rank == 1:                                   # a replica that starts elsewhere: the constructor's broadcast must erase this
            with torch.no_grad():
                for p in m.trainable_parameters():
                    p.add_(0.03)
        m.set_dropout_seed(SEED, 0)
        tr = Trainer(m, lr=LR, grad_accum=ACCUM, warmup=0, total_steps=20, reduce_chunk_mb=8, **({"wire_dtype": None, "overlap_exchange": False} if wire == "fp32" else
                                                                                                  {"overlap_exchange": False} if wire == "no_overlap" else {}))
        assert tr.arena is not None and tr.dist_on and tr.arena.flat.numel() > 2 * tr.reduce_chunk, (tr.arena.flat.numel(), tr.reduce_chunk)
        assert tr.sparse_embed and tr.wire_dtype == (None if wire == "fp32" else torch.bfloat16) and tr.overlap_exchange == (wire == "default"), (tr.sparse_embed, tr.wire_dtype)
        if rank == 0:
            torch.save(torch.tensor(tr.arena.block_of[tr._embed_key]), os.path.join(tmp, "embed_block.pt"))
        assert int(m.dropout_state()[0]) == rank_dropout_seed(SEED, rank)
        tr.check_replicas()
        first = {}
        tr.grad_hook = lambda t, ss: first.setdefault("g", (t.arena.flat.detach().cpu().clone(), float(ss)))     # the reduced arena of step 0
        db = mc._dev(_rank_batch(batch, rank))
        losses = [float(tr.micro_step(db)["loss"].detach()) for _ in range(OPT_STEPS * ACCUM)]
        tr.check_replicas()                             # still identical after two optimizer steps (incl. the clip coefficient)
        if rank == 0:
            torch.save(first["g"][0], os.path.join(tmp, "arena0.pt"))
        ret[rank] = (losses, torch.cat([w.detach().flatten().cpu() for w in tr.opt.master]), tr.opt_steps, first["g"][1])
        tr.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("wire", ["fp32", "default"])      # ("no_overlap" -- bf16 wire, everything after the backward -- is a valid third value: dropped from the suite for time)
def test_two_ranks_share_one_gpu_gloo(tmp_path, wire):
    """wire = "default": what a multi-rank Trainer does unless told otherwise -- dense pieces in bf16 on the wire (the reference's DeepSpeed bf16 engine,
    training.py:314-329), the embedding block as fp32 rows, the arena's tail leaving between the two halves of the window's last backward (DeepSpeed's
    `overlap_comm`); "no_overlap": the same wire, everything exchanged after the backward; "fp32": `wire_dtype=None`, no overlap."""
    import torch.multiprocessing as mp
    from llmseg_amd.train import Trainer, rank_dropout_seed
    from tests import backward_checks as bc, model_checks as mc
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_two_rank_worker, args=(world, {"fp32": 29581, "no_overlap": 29583, "default": 29585}[wire], ret, str(tmp_path), wire), nprocs=world, join=True)
    (l0, p0, n0, ss0), (l1, p1, n1, ss1) = ret[0], ret[1]
    assert n0 == n1 == OPT_STEPS and ss0 == ss1
    assert torch.equal(p0, p1), (p0 - p1).abs().max().item()           # replicas bit-identical after the exchanges
    assert max(abs(a - b) for a, b in zip(l0, l1)) > 1e-4, "the two ranks saw the same data / masks"
    # one process over the union: accumulation window = ACCUM x world micro-batches, each under the dropout stream its rank used.  Compared
    # at the level of what is exchanged -- the accumulated fp32 gradient arena of optimizer step 0 (same weights on both sides) -- because
    # parameters after AdamW amplify rounding noise of near-zero gradients into +-lr steps.
    cfg, m, sd, batch = bc._lora_case("sam", p_drop=0.05)
    tr = Trainer(m, lr=LR, grad_accum=ACCUM * world, warmup=0, total_steps=20)
    first = {}
    tr.grad_hook = lambda t, ss: first.setdefault("g", (t.arena.flat.detach().cpu().clone(), float(ss)))
    dbs = [mc._dev(_rank_batch(batch, r)) for r in range(world)]
    losses = {0: [], 1: []}
    for s in range(ACCUM):
        for r in range(world):
            m.set_dropout_seed(rank_dropout_seed(SEED, r), s)          # micro_step advances to s + 1, as on rank r
            losses[r].append(float(tr.micro_step(dbs[r])["loss"].detach()))
    pu = torch.cat([w.detach().flatten().cpu() for w in tr.opt.master])
    tr.close()
    assert tr.opt_steps == 1
    for r, lr_ in ((0, l0), (1, l1)):
        assert max(abs(a - b) for a, b in zip(losses[r], lr_[:ACCUM])) < 2e-3, (losses[r], lr_)
    g_dist, g_union = torch.load(os.path.join(str(tmp_path), "arena0.pt")), first["g"][0]
    # identical up to the fp32 association of the two accumulation orders: (a + b) + (c + d) across ranks vs ((a + b) + c) + d in one process
    scale = g_union.abs().max().item()
    eo, en = [int(v) for v in torch.load(os.path.join(str(tmp_path), "embed_block.pt"))]
    d = (g_dist - g_union).abs()
    assert scale > 1e-3 and d[eo:eo + en].max().item() <= 2e-5 * scale + 1e-7, (d[eo:eo + en].max().item(), scale)      # the embedding rows: fp32 on the wire either way
    d[eo:eo + en] = 0
    if wire == "fp32":
        assert d.max().item() <= 2e-5 * scale + 1e-7, (d.max().item(), scale)
        assert abs(ss0 - first["g"][1]) <= 1e-4 * first["g"][1]
    else:                                                    # every dense element: its two per-rank sums rounded to bf16, added in bf16
        assert d.max().item() <= 2.0 ** -7 * scale and d.norm().item() <= 2.0 ** -7 * g_union.norm().item(), (d.max().item(), scale, d.norm().item(), g_union.norm().item())
        assert d.max().item() > 0 and abs(ss0 - first["g"][1]) <= 2e-2 * first["g"][1]
