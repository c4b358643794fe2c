"""CPU: the data-parallel trainer (DDP over gloo, world_size 2): gradient accumulation with no_sync, one all-reduce per optimizer
step, clipping, WarmupDecayLR.  The optimizer arithmetic is injected (CPU restatement of llmseg_adamw) because the product's
optimizer kernels are HIP-only; what is under test here is the distributed / accumulation logic of llmseg_amd.train."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from llmseg_amd.train import Trainer, warmup_decay_lr


class CpuAdamW:
    """Same update rule as llmseg_adamw (fp32 master, bias correction, decoupled weight decay, external grad scale)."""

    def __init__(self, params, betas=(0.9, 0.95), eps=1e-8, wd=0.0):
        self.params = list(params)
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.b, self.eps, self.wd, self.t = betas, eps, wd, 0

    def grad_sumsq(self):
        return sum((p.grad.float() ** 2).sum() for p in self.params if p.grad is not None).reshape(1)

    def step(self, lr, grad_scale):
        self.t += 1
        for p, m, v in zip(self.params, self.m, self.v):
            g = p.grad * grad_scale
            m.mul_(self.b[0]).add_(g, alpha=1 - self.b[0])
            v.mul_(self.b[1]).addcmul_(g, g, value=1 - self.b[1])
            mh, vh = m / (1 - self.b[0] ** self.t), v / (1 - self.b[1] ** self.t)
            p.data.add_(-(lr * (mh / (vh.sqrt() + self.eps) + self.wd * p.data)))


class ArenaLinearFn(torch.autograd.Function):
    """y = x W^T + b whose weight / bias gradients are ACCUMULATED into the tensors' `_g32` arena views (what the HIP Functions do kernel-side)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.g = (w._g32, b._g32)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        ctx.g[0].add_(dy.t() @ x)
        ctx.g[1].add_(dy.sum(0))
        return dy @ w, None, None


class ArenaToy(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = nn.Linear(6, 5)
        self.b = nn.Linear(5, 1)

    def forward(self, x, y):
        h = torch.tanh(ArenaLinearFn.apply(x, self.a.weight, self.a.bias))
        return {"loss": ((ArenaLinearFn.apply(h, self.b.weight, self.b.bias) - y) ** 2).mean()}


class CpuArenaAdamW(CpuAdamW):
    """reads the fp32 arena views instead of .grad"""

    def sumsq_flat(self, flat):
        return (flat.double() ** 2).sum().float().reshape(1)

    def step(self, lr, grad_scale):
        for p in self.params:
            p.grad = p._g32.clone()
        super().step(lr, grad_scale)
        for p in self.params:
            p.grad = None


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = nn.Linear(6, 5)
        self.b = nn.Linear(5, 1)

    def forward(self, x, y):
        return {"loss": ((self.b(torch.tanh(self.a(x))) - y) ** 2).mean()}


def _data(rank, step):
    g = torch.Generator().manual_seed(100 * step + rank)
    return torch.randn(4, 6, generator=g), torch.randn(4, 1, generator=g)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = Toy()
    tr = Trainer(m, lr=1e-2, clip=1.0, grad_accum=3, warmup=2, total_steps=10, optimizer=CpuAdamW([p for p in m.parameters()]))
    for s in range(6):
        x, y = _data(rank, s)
        tr.micro_step(dict(x=x, y=y))
    ret[rank] = (tr.opt_steps, torch.cat([p.detach().flatten() for p in m.parameters()]))
    dist.destroy_process_group()


def _arena_worker(rank, world, port, ret):
    """The arena path of the trainer (one all-reduce of the flat fp32 gradient buffer per optimizer step) + bench.py's timing plumbing
    (barrier, per-rank wall clock, all_reduce(MAX)) under gloo on the CPU."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = ArenaToy()
    if rank == 1:                                   # a replica that starts from different weights (different seed / partial load) ...
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.3)
    # ... is overwritten by rank 0's at construction (no DDP wrapper in arena mode to do it); 100-element all-reduce chunks -> 3 pieces
    # check_every=1: the trainer itself verifies the replicas after every optimizer step (round 4: the clip coefficient needs no collective any
    # more, so a periodic check is the safety net); counted through a wrapper
    tr = Trainer(m, lr=1e-2, clip=1.0, grad_accum=3, warmup=2, total_steps=10, optimizer=lambda ps: CpuArenaAdamW(ps), use_arena=True,
                 reduce_chunk_mb=400 / (1 << 20), check_every=1)
    assert tr.arena is not None and tr.ddp is None and tr.reduce_chunk == 100 and tr.arena.flat.numel() > 200
    checks, orig_check = [0], tr.check_replicas
    tr.check_replicas = lambda: (checks.__setitem__(0, checks[0] + 1), orig_check())[1]
    tr.check_replicas()
    import bench
    step_no = [0]

    def step():
        x, y = _data(rank, step_no[0])
        step_no[0] += 1
        return tr.micro_step(dict(x=x, y=y))
    dt, out = bench.timed(step, 5, 1, dist, torch.device("cpu"))
    assert all(p.grad is None for p in m.parameters())
    assert checks[0] == 1 + tr.opt_steps == 3, checks             # the explicit call above + one per optimizer step
    tr.check_replicas()
    final = torch.cat([p.detach().flatten() for p in m.parameters()]).clone()
    if rank == 1:                                   # a diverged replica is detected
        with torch.no_grad():
            m.a.bias.add_(1e-3)
    try:
        tr.check_replicas()
        diverged = False
    except AssertionError:
        diverged = True
    assert diverged
    ret[rank] = (tr.opt_steps, final, dt, float(out["loss"]))
    dist.destroy_process_group()


def test_arena_gloo_matches_single_process_and_bench_plumbing():
    world, port = 2, 29613
    ret = mp.Manager().dict()
    mp.spawn(_arena_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0][0] == 2 and ret[1][0] == 2                       # 1 warm-up + 5 timed micro-steps / accum 3
    assert torch.equal(ret[0][1], ret[1][1])                        # replicas stay bit-identical
    assert ret[0][2] == ret[1][2] and ret[0][2] > 0                 # bench.timed: MAX over ranks -> the same number on every rank
    m = Toy()                                                       # same init (seed 0), plain autograd
    opt = CpuAdamW(list(m.parameters()))
    for o in range(2):
        for p in m.parameters():
            p.grad = None
        for s in range(3 * o, 3 * o + 3):
            for r in range(world):
                x, y = _data(r, s)
                (m(x, y)["loss"] / world).backward()
        ss = opt.grad_sumsq()
        norm = ss.sqrt() / 3
        coef = torch.clamp(1.0 / (norm + 1e-6), max=1.0) / 3
        opt.step(warmup_decay_lr(o, 1e-2, 2, 10), coef)
    ref = torch.cat([p.detach().flatten() for p in m.parameters()])
    assert torch.allclose(ret[0][1], ref, atol=1e-6), (ret[0][1] - ref).abs().max()


def test_ddp_gloo_matches_single_process():
    world, port = 2, 29611
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0][0] == 2 and ret[1][0] == 2                       # 6 micro-steps / accum 3
    assert torch.allclose(ret[0][1], ret[1][1], atol=0, rtol=0)    # replicas stay bit-identical
    # single process over the union of both ranks' micro-batches (DDP averages over ranks)
    m = Toy()
    opt = CpuAdamW(list(m.parameters()))
    for o in range(2):
        for p in m.parameters():
            p.grad = None
        for s in range(3 * o, 3 * o + 3):
            for r in range(world):
                x, y = _data(r, s)
                (m(x, y)["loss"] / world).backward()
        ss = opt.grad_sumsq()
        norm = ss.sqrt() / 3
        coef = torch.clamp(1.0 / (norm + 1e-6), max=1.0) / 3
        opt.step(warmup_decay_lr(o, 1e-2, 2, 10), coef)
    ref = torch.cat([p.detach().flatten() for p in m.parameters()])
    assert torch.allclose(ret[0][1], ref, atol=1e-6), (ret[0][1] - ref).abs().max()


class ArenaEmbedFn(torch.autograd.Function):
    """rows = W[ids] whose gradient is scatter-ACCUMULATED into W's `_g32` arena view (what `llmseg_scatter_add_rows` does kernel-side)."""

    @staticmethod
    def forward(ctx, ids, w):
        ctx.ids, ctx.g = ids, w._g32
        return w[ids]

    @staticmethod
    def backward(ctx, dy):
        ctx.g.index_add_(0, ctx.ids.reshape(-1), dy.reshape(-1, dy.shape[-1]))
        return None, None


class _Inner(nn.Module):
    def __init__(self):
        super().__init__()
        self.embed_tokens = nn.Embedding(50, 6)


class ArenaEmbedToy(nn.Module):
    """`model.embed_tokens.weight` (a table whose gradient touches few rows) + a dense head: the two kinds of arena blocks the exchange treats apart."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.model = _Inner()
        self.head = nn.Linear(6, 1)

    def forward(self, input_ids, y):
        e = ArenaEmbedFn.apply(input_ids, self.model.embed_tokens.weight).mean(1)
        return {"loss": ((ArenaLinearFn.apply(torch.tanh(e), self.head.weight, self.head.bias) - y) ** 2).mean()}


def _embed_data(rank, step):
    g = torch.Generator().manual_seed(1000 * step + rank)
    return torch.randint(0, 50, (3, 4), generator=g), torch.randn(3, 1, generator=g)


def _embed_worker(rank, world, port, ret, sparse, wire=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = ArenaEmbedToy()
    tr = Trainer(m, lr=1e-2, clip=1.0, grad_accum=2, warmup=2, total_steps=10, optimizer=lambda ps: CpuArenaAdamW(ps), use_arena=True,
                 reduce_chunk_mb=64 / (1 << 20), check_every=1, sparse_embed=sparse, wire_dtype=wire)
    assert tr.sparse_embed == sparse and (not sparse or tr._embed_cols == 6)
    seen = {}
    tr.grad_hook = lambda t, ss: seen.update(flat=t.arena.flat.clone(), ss=float(ss))
    for s_ in range(4):
        ids, y = _embed_data(rank, s_)
        tr.micro_step(dict(input_ids=ids, y=y))
    ret[(sparse if wire is None else "bf16", rank)] = (tr.opt_steps, torch.cat([p.detach().flatten() for p in m.parameters()]).clone(), seen["flat"], seen["ss"])
    dist.destroy_process_group()


def test_sparse_embedding_row_exchange_equals_dense_all_reduce():
    """The embedding block travels as an all-gather of (row, values) lists (`Trainer._exchange_embed_rows`): the reduced arena equals the dense
    all-reduce's (fp32 summation order over two ranks: exact), replicas stay bit-identical, rows no rank touched stay exactly zero."""
    world = 2
    ret = mp.Manager().dict()
    for sparse, port in ((True, 29617), (False, 29619)):
        mp.spawn(_embed_worker, args=(world, port, ret, sparse), nprocs=world, join=True)
    for sparse in (True, False):
        assert ret[(sparse, 0)][0] == 2 and torch.equal(ret[(sparse, 0)][1], ret[(sparse, 1)][1]) and torch.equal(ret[(sparse, 0)][2], ret[(sparse, 1)][2])
    assert torch.allclose(ret[(True, 0)][2], ret[(False, 0)][2], atol=1e-7) and torch.allclose(ret[(True, 0)][1], ret[(False, 0)][1], atol=1e-7)
    assert abs(ret[(True, 0)][3] - ret[(False, 0)][3]) <= 1e-6 * abs(ret[(False, 0)][3])
    touched = set()
    for r in range(world):
        for s_ in (2, 3):                                            # the second accumulation window (what the hook saw last)
            touched |= set(_embed_data(r, s_)[0].reshape(-1).tolist())
    emb = ret[(True, 0)][2][: 50 * 6].view(50, 6)
    rows_nz = set(torch.nonzero(emb.abs().sum(1)).flatten().tolist())
    assert rows_nz <= touched and len(rows_nz) >= len(touched) - 2


def test_bf16_wire_option():
    """`Trainer(wire_dtype=torch.bfloat16)`: the dense pieces are exchanged in bf16 (the reference's DeepSpeed engine reduces bf16 gradients), the embedding
    rows and the arena stay fp32: replicas bit-identical, the reduced dense part within bf16 rounding of the fp32 exchange, the embedding block equal to it."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_embed_worker, args=(world, 29621, ret, True, torch.bfloat16), nprocs=world, join=True)
    mp.spawn(_embed_worker, args=(world, 29623, ret, True), nprocs=world, join=True)
    a, b = ret[("bf16", 0)], ret[(True, 0)]
    assert torch.equal(a[1], ret[("bf16", 1)][1]) and torch.equal(a[2], ret[("bf16", 1)][2])
    emb = slice(0, 50 * 6)
    assert torch.equal(a[2][emb], b[2][emb])                                                     # rows: fp32 on the wire either way
    dense_a, dense_b = a[2][50 * 6:], b[2][50 * 6:]
    assert not torch.equal(dense_a, dense_b) and torch.allclose(dense_a, dense_b, rtol=2 ** -7, atol=1e-6)


def test_warmup_decay_lr():
    assert warmup_decay_lr(0, 1.0, 100, 5000) == 0.0
    assert abs(warmup_decay_lr(50, 1.0, 100, 5000) - 0.5) < 1e-12
    assert warmup_decay_lr(100, 1.0, 100, 5000) == 1.0
    assert abs(warmup_decay_lr(2550, 1.0, 100, 5000) - 0.5) < 1e-12
    assert warmup_decay_lr(5000, 1.0, 100, 5000) == 0.0


def test_graph_inputs_and_rank_dropout_seed():
    """What a replayed hipGraph copies: only the tensors the captured kernels read; pass-through entries (ground-truth masks of another
    size, extra keys) neither break the copy nor enter the key, any shape change of a read tensor does (train.py:282)."""
    from llmseg_amd import train as T
    mk = lambda k, gt: dict(images=torch.zeros(2, 3, 8, 8), images_clip=torch.zeros(2, 3, 4, 4), input_ids=torch.zeros(2, 5, dtype=torch.int64),
                            labels=torch.zeros(2, 5, dtype=torch.int64), attention_masks=torch.ones(2, 5, dtype=torch.bool), offset=torch.arange(3),
                            sam_segs_list=[torch.zeros(k, 4, 4), torch.zeros(k, 4, 4)], sam_ious_list=[torch.zeros(1, k)] * 2,
                            sam_iops_list=[torch.zeros(1, k)] * 2, masks_list=[torch.zeros(1, gt, gt)] * 2, label_list=[torch.zeros(gt, gt)] * 2,
                            resize_list=[(gt, gt)] * 2, inference=False)
    a, b, c = mk(3, 7), mk(3, 9), mk(4, 7)
    b["extra_key"] = "x"
    assert T._input_sig(a) == T._input_sig(b) != T._input_sig(c)
    b["images"] += 1.0
    b["sam_segs_list"][1] += 2.0
    T._copy_batch(a, b)                                            # masks_list 7x7 vs 9x9: untouched, no error
    assert float(a["images"].min()) == 1.0 and float(a["sam_segs_list"][1].min()) == 2.0 and a["masks_list"][0].shape[-1] == 7
    seeds = {T.rank_dropout_seed(0x5EED, r) for r in range(8)}
    assert len(seeds) == 8 and T.rank_dropout_seed(0x5EED, 0) == 0x5EED and all(0 <= v < 2 ** 63 for v in seeds)


def test_fused_window_plan_and_merge(monkeypatch):
    """Host logic of the fused accumulation window (`merge_micro_batches` + `make_plan(micro_batches=k)`): offsets re-based, one CE segment per
    micro-batch with its own trailing row, item weights 1 / (rounds x images of the micro-batch), dropout segment = rows of one micro-batch, a plan
    signature that differs from the unfused one; misuse fails loudly.  (`make_plan` pins its uploads: patched out, there is no device here.)"""
    import types
    from llmseg_amd import trainable
    from llmseg_amd.train import merge_micro_batches
    from oracle import cases
    monkeypatch.setattr(torch.Tensor, "pin_memory", lambda self: self)
    base = cases.tiny_lisa_batch(img_size=32)
    batches = []
    for j in range(3):
        b = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in base.items()}
        b["labels"][:, :10 + j] = -100                                    # different label counts per micro-batch: different CE denominators
        batches.append(b)
    merged = merge_micro_batches(batches)
    assert merged["offset"].tolist() == [0, 2, 3, 5, 6, 8, 9] and merged["input_ids"].shape[0] == 9 and len(merged["sam_segs_list"]) == 6

    class Fake(trainable.TrainableMixin):
        pass
    f = Fake()
    f.config, f.device_, f.seg_token_idx = types.SimpleNamespace(n_img_tokens=256), torch.device("cpu"), cases.SEG
    p1 = [f.make_plan(**b) for b in batches]
    pk = f.make_plan(**merged, micro_batches=3)
    assert pk.micro == 3 and pk.loss_div == 1.0 and p1[0].loss_div == 2.0 and pk.sig != f.make_plan(**merged).sig
    T = 24 - 1 + 256
    assert pk.drop_seg_rows == 3 * T and p1[0].drop_seg_rows == 0
    sizes = [int(p.ce_rows.numel()) for p in p1]
    assert pk.ce_segs == [(0, sizes[0]), (sizes[0], sizes[0] + sizes[1]), (sizes[0] + sizes[1], sum(sizes))] and len(set(sizes)) == 3
    off = 0
    for j, p in enumerate(p1):                                          # segment j = micro-batch j's own rows (shifted by its sequences) and labels
        a, b_ = pk.ce_segs[j]
        assert torch.equal(pk.ce_rows[a:b_], p.ce_rows + j * 3 * T) and torch.equal(pk.ce_labels[0, a:b_], p.ce_labels[0])
        assert int(pk.ce_labels[0, a]) == -100                           # every segment starts with its own ignored trailing-row label
    assert torch.allclose(pk.tensors["loss_w16"], torch.cat([p.tensors["loss_w16"] / 2.0 for p in p1]))       # 1 / rounds, / 2 images per micro-batch
    import pytest
    with pytest.raises(AssertionError):
        f.make_plan(**merged, micro_batches=2)                           # 9 sequences / 6 images do not split into 2 equal micro-batches
    # two micro-batches of 2 images whose sequence counts differ (4 and 2): N = 6 splits evenly, the micro-batches do not
    ids4 = torch.cat([base["input_ids"], base["input_ids"][:1]])
    big = dict(input_ids=torch.cat([ids4, base["input_ids"][:2]]), labels=torch.cat([ids4, base["input_ids"][:2]]), attention_masks=torch.ones(6, 24, dtype=torch.bool),
               offset=torch.tensor([0, 2, 4, 5, 6]), sam_segs_list=base["sam_segs_list"] * 2)
    with pytest.raises(AssertionError, match="same number of sequences"):
        f.make_plan(**big, micro_batches=2)
