"""Data-parallel training step for the hot path: torch.distributed over RCCL/xGMI in place of the reference's DeepSpeed ZeRO-2 engine
(reference `training.py:292-332` config, `:480-602` loop).

What is kept from the reference's recipe: micro-batches with gradient accumulation, ONE all-reduce of the trainable-gradient set per
optimizer step (nothing is reduced on the other accumulation micro-steps -- DDP's `no_sync()`), AdamW(betas, wd 0) on fp32 master
weights with bf16 model copies, global-norm clipping at 1.0, WarmupDecayLR (linear 0 -> lr over 100 steps, then linear decay to 0 at
total_steps).  ZeRO sharding is dropped on purpose: optimizer state for the trainable set is ~3.5 GB on a 288 GB device.  DeepSpeed
itself is not installed here: PARITY UNPINNED for optimizer/schedule details (published formulas).

MI355X-first structure:
  * `GradArena`: ONE flat fp32 buffer holds the gradient of every trainable tensor.  The backward kernels accumulate into it directly
    (`C += dY^T X` GEMM epilogue, fixed-order partial sums of the skinny LoRA / norm / bias / embedding kernels), so accumulation over micro-steps is
    fp32 end to end, there is no per-parameter bf16 `.grad`, the global norm is ONE reduction kernel and the data-parallel exchange is ONE
    `all_reduce` on a contiguous 1.2 GB buffer (what DDP's reducer does with its buckets, minus the copy into them).
  * `use_graph`: the whole fwd+bwd micro-step (~3000 kernel launches) is captured once per batch structure in a hipGraph and replayed;
    inputs are copied into the graph's static buffers, the dropout offset lives in device memory, the optimizer step and the collective
    stay outside the graph.
  * `ddp_wrapper=True` keeps the plain `torch.nn.parallel.DistributedDataParallel` wrapper (bf16 `.grad`, reducer buckets) as an alternative.

The optimizer / clipping arithmetic runs in libllmseg_hip.so (`llmseg_adamw`, `llmseg_sumsq`); `optimizer` can be replaced
(tests inject a CPU restatement to exercise the distributed/accumulation logic under gloo).
"""
import contextlib

import torch
import torch.distributed as dist

from .trainable import rank_dropout_seed  # noqa: F401  (re-exported: the tests import it from here)


def warmup_decay_lr(step, lr, warmup=100, total=5000):
    """DeepSpeed WarmupDecayLR with warmup_type='linear', warmup_min_lr=0 (training.py:304-313). `step` counts optimizer steps
    already taken."""
    if step < warmup:
        return lr * step / max(1, warmup)
    return lr * max(0.0, (total - step) / max(1.0, total - warmup))


class GradArena:
    """Flat fp32 gradient storage of a `LISAForCausalLM`'s trainable tensors.  Every trainable Parameter gets `p._g32`, an fp32 view of
    its block; fused weight groups (q|k|v of the head attentions) occupy ONE block laid out like the fused tensor, whose backing
    tensor gets the fused view, so a fused forward GEMM has a fused weight-gradient GEMM."""

    ALIGN = 64          # elements: every block starts 256-byte aligned (vector accesses of the GEMM epilogue)

    def __init__(self, model):
        self.model = model
        if hasattr(model, "params") and hasattr(model.params, "flat"):           # LISAForCausalLM: ParamTree with fused weight groups
            from .params import fused_groups
            flat = model.params.flat
            named = list(model.params.named_parameters())
            groups = {g: [m for m in members if m in flat] for g, members in fused_groups(model.config) if g in flat}
        else:                                                                     # any nn.Module whose Functions honour `_g32`
            named = list(model.named_parameters())
            flat, groups = dict(named), {}
        member_of = {m: g for g, ms in groups.items() for m in ms}
        blocks, seen, total = [], set(), 0
        self.params = []
        for name, p in named:
            if not p.requires_grad:
                continue
            self.params.append(p)
            key = member_of.get(name, name)
            if key in seen:
                continue
            seen.add(key)
            n = flat[key].numel() if key in groups else p.numel()
            blocks.append((key, total, n))
            total += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.numel = sum(p.numel() for p in self.params)
        self._touched, views = [], {}
        for key, off, n in blocks:
            if key in groups:
                buf = flat[key]
                gv = self.flat[off:off + n].view(buf.shape)
                self._attach(buf, gv, make_leaf=True)
                r = 0
                for m in groups[key]:
                    rows = flat[m].shape[0]
                    self._attach(flat[m], gv[r:r + rows])
                    r += rows
                if key.endswith("cross_attn_image_to_token.qkv.weight") or key.endswith("cross_attn_image_to_token.qkv.bias"):
                    # the k|v half as one operand (rows D.. of the fused tensor): a leaf view that carries its arena block
                    D = buf.shape[0] // 3
                    kv = buf[D:].detach()        # an alias without an autograd view relation to `buf` (which is itself a leaf that is updated in place)
                    self._attach(kv, gv[D:], make_leaf=True)
                    views[key.replace("qkv.", "kv.")] = kv
            else:
                p = flat[key]
                self._attach(p, self.flat[off:off + n].view(p.shape))
                if key == "model.lisa_dino_conv.weight":
                    w2d = p.detach().view(p.shape[0], p.shape[1]).detach()
                    self._attach(w2d, p._g32.view(p.shape[0], p.shape[1]), make_leaf=True)
                    views["model.lisa_dino_conv.weight2d"] = w2d
        model.__dict__["_arena_views"] = views
        self.block_of = {key: (off, n) for key, off, n in blocks}       # name (or fused-group name) -> (offset, elements) inside `flat`


    def _attach(self, t, view, make_leaf=False):
        t._g32 = view
        if make_leaf and not t.requires_grad:
            t.requires_grad_(True)           # so that an autograd Function whose other inputs are frozen still runs its backward
            self._touched.append((t, True))
        else:
            self._touched.append((t, False))

    def detach(self):
        for t, was_made_leaf in self._touched:
            if hasattr(t, "_g32"):
                del t._g32
            if was_made_leaf:
                t.requires_grad_(False)
        self.model.__dict__.pop("_arena_views", None)
        self._touched = []

    def zero_(self):
        self.flat.zero_()


class HipAdamW:
    """fp32 master / m / v per trainable parameter; fused update + bf16 write-back in one kernel per parameter.  Gradients are read
    from the parameter's arena view (`p._g32`, fp32) when it has one, else from `p.grad`."""

    def __init__(self, params, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0):
        from . import ops
        self.ops = ops
        self.params = list(params)
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        self.master = [p.detach().float().clone() for p in self.params]
        self.m = [torch.zeros_like(t) for t in self.master]
        self.v = [torch.zeros_like(t) for t in self.master]
        self.t = 0

    def resync_master(self):
        """Re-read the fp32 master copies from the (bf16) parameters: call after loading weights into the model."""
        with torch.no_grad():
            for w, p in zip(self.master, self.params):
                w.copy_(p.detach().float())

    @staticmethod
    def _grad(p):
        g = getattr(p, "_g32", None)
        return g if g is not None else p.grad

    def grad_sumsq(self):
        acc = torch.zeros(1, device=self.params[0].device, dtype=torch.float32)
        for p in self.params:
            g = self._grad(p)
            if g is not None:
                self.ops.sumsq(g.contiguous(), acc)
        return acc

    def step(self, lr, grad_scale):
        """grad_scale: device fp32 scalar multiplying every gradient (1/accum x clip coefficient)."""
        self.t += 1
        for p, w, m, v in zip(self.params, self.master, self.m, self.v):
            g = self._grad(p)
            if g is None:
                continue
            g = g if g.is_contiguous() else g.contiguous()
            if p.is_contiguous():
                self.ops.adamw_(p.data, w, g, m, v, lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t, grad_scale)
            else:   # parameter is a strided view: update a contiguous copy and write it back
                tmp = p.data.contiguous()
                self.ops.adamw_(tmp, w, g, m, v, lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t, grad_scale)
                p.data.copy_(tmp)

    def state_dict(self):
        return {"t": self.t, "master": self.master, "m": self.m, "v": self.v}

    def load_state_dict(self, sd):
        self.t = int(sd["t"])
        with torch.no_grad():
            for dst, src in ((self.master, sd["master"]), (self.m, sd["m"]), (self.v, sd["v"])):
                for d, s in zip(dst, src):
                    d.copy_(s)


# What a captured fwd+bwd micro-step reads from the batch (`TrainableMixin._model_forward`): everything else of the collate dict
# (labels, attention masks, offset -> consumed on the host by `make_plan`; masks_list / label_list / resize_list -> pass-through)
# never reaches a kernel, so it is neither copied into the graph's buffers nor part of the graph key.
GRAPH_INPUTS = ("images", "images_clip", "tower_visual", "tower_clip", "input_ids", "sam_segs_list", "sam_ious_list", "sam_iops_list")


def _graph_inputs(batch):
    """-> {key: tensor | [tensors]} of the graph-read entries present in `batch`."""
    return {k: batch[k] for k in GRAPH_INPUTS if batch.get(k) is not None}


def _input_sig(batch):
    """Shapes + dtypes of every graph-read tensor: part of the graph key, so a batch whose inputs differ in any shape gets its own graph
    (or the eager path) instead of a failing `copy_`."""
    def one(t):
        return (tuple(t.shape), str(t.dtype))
    return tuple((k, tuple(one(t) for t in v) if isinstance(v, (list, tuple)) else one(v)) for k, v in _graph_inputs(batch).items())


def _copy_batch(dst, src):
    """Copy the graph-read inputs of `src` into the captured tensors `dst` (same key set and shapes: guaranteed by the graph key)."""
    for k, v in _graph_inputs(src).items():
        d = dst[k]
        if torch.is_tensor(v):
            if d.data_ptr() != v.data_ptr():
                d.copy_(v, non_blocking=True)
        else:
            for dd, vv in zip(d, v):
                if dd.data_ptr() != vv.data_ptr():
                    dd.copy_(vv, non_blocking=True)


def merge_micro_batches(batches):
    """The `grad_accum` micro-batches of one optimizer step (collated dicts of the SAME token length and sequences per micro-batch) as ONE
    batch for a fused pass (`Trainer(fused_accum=k)`, `make_plan(micro_batches=k)`): tensors concatenated along the batch axis, `offset`
    re-based, per-image lists extended.  Reference: the k micro-steps of `training.py:532-547` (`gradient_accumulation_steps`, :79-82)."""
    b0 = batches[0]
    out = {}
    for key, v in b0.items():
        vals = [b[key] for b in batches]
        if key == "offset":
            parts, base = [vals[0].reshape(-1)[:1]], 0
            for o in vals:
                o = o.reshape(-1)
                parts.append(o[1:] + base)
                base = base + int(o[-1])
            out[key] = torch.cat(parts)
        elif torch.is_tensor(v):
            out[key] = torch.cat(vals, 0)
        elif isinstance(v, (list, tuple)):
            out[key] = [x for val in vals for x in val]
        else:
            out[key] = v
    return out


class WindowTowers:
    """Outputs of the frozen towers for the images of one accumulation window (`Trainer.encode_window`): `parts[j]` = (tower_visual, tower_clip)
    of micro-batch j, views into two tensors kept alive here."""

    def __init__(self, parts, event, keep, stream):
        self.parts, self.event, self.keep, self.stream = parts, event, keep, stream

    def wait(self):
        """The current stream waits for the pass (no host synchronisation)."""
        if self.event is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(self.event)
            if self.stream is not None:
                for t in self.keep:
                    t.record_stream(cur)
            self.event = None


class Trainer:
    """One process per GPU.  `module(**batch)` must return a dict with a scalar "loss".

    Default (HIP model): gradient arena + optional hipGraph micro-step.  `ddp_wrapper=True`, or a module that is not a
    `LISAForCausalLM` (the gloo tests' toy modules), uses plain `.grad` tensors and torch DDP."""

    def __init__(self, module, lr=3e-4, betas=(0.9, 0.95), weight_decay=0.0, clip=1.0, grad_accum=10, warmup=100, total_steps=5000,
                 optimizer=None, device_ids=None, force_ddp=False, ddp_wrapper=False, use_graph=False, graph_warmup=2, use_arena=None,
                 reduce_chunk_mb=128, sync_init=True, check_every=100, leaf_stream=False, fused_accum=1, sparse_embed=None, time_comm=False, wire_dtype="auto",
                 max_graphs=None, overlap_exchange=None):
        """optimizer: None = HipAdamW; or a factory `params -> optimizer` / an optimizer object (CPU tests).  use_arena: None = automatic
        (the HIP model with the built-in optimizer), True = force the fp32 gradient arena (the module's autograd Functions must honour `_g32`)."""
        self.module = module
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.dist_on = dist.is_initialized()
        self.ddp = None
        self.arena = None
        is_hip_model = hasattr(module, "params") and hasattr(module, "make_plan")
        if use_arena is None:
            use_arena = is_hip_model and optimizer is None and not ddp_wrapper
        if use_arena:
            self.arena = GradArena(module)
            self.params = self.arena.params
        else:
            if self.world > 1 or (force_ddp and dist.is_initialized()):
                self.ddp = torch.nn.parallel.DistributedDataParallel(module, device_ids=device_ids, broadcast_buffers=False,
                                                                   gradient_as_bucket_view=False)
            self.params = [p for p in module.parameters() if p.requires_grad]
        if callable(optimizer) and not hasattr(optimizer, "step"):
            optimizer = optimizer(self.params)
        self.opt = optimizer if optimizer is not None else HipAdamW(self.params, betas, weight_decay=weight_decay)
        if hasattr(self.opt, "resync_master") and hasattr(module, "__dict__"):
            module.__dict__.setdefault("_weight_hooks", []).append(self.opt.resync_master)
        self.lr, self.clip, self.accum, self.warmup, self.total = lr, clip, grad_accum, warmup, total_steps
        # fused_accum = k > 1: every `micro_step` receives the concatenation of k micro-batches (`merge_micro_batches`) and a plan built with
        # `make_plan(..., micro_batches=k)`: ONE pass whose loss is the sum of the k micro-batch losses, so the arena holds what k micro-steps
        # would have accumulated (every GEMM at k x the rows: M = 638 -> 6380 at 2 images x 10); an optimizer step then follows every
        # `grad_accum` such passes (normally grad_accum = 1) and averages over grad_accum * k micro-batches.
        self.fused = int(fused_accum)
        assert self.fused >= 1
        # sparse_embed (default: on whenever a process group and an `embed_tokens` block exist): the embedding table's gradient block (0.52 GB of the
        # 1.16 GB arena at Llama-7B) has non-zero rows only for the tokens the ranks saw in this accumulation window (<= accum x N x L of 32004 rows:
        # 98 % zeros at the benchmark's batch), so it is exchanged as an all-gather of (row index, row) lists instead of a dense all-reduce --
        # `_exchange_embed_rows`.  The row set is read from the BLOCK ITSELF (rows with a non-zero entry; two row reductions over 0.52 GB, ~0.2 ms), not
        # from a record of the ids the micro-steps saw: whatever wrote into the block -- a direct `_eager_step`, another Function, a batch dict
        # without `input_ids` -- is exchanged.  The reference reduces it densely (DeepSpeed ZeRO-2 buckets, training.py:321-329).
        self._embed_key = "model.embed_tokens.weight"
        has_embed = self.arena is not None and self._embed_key in getattr(self.arena, "block_of", {})
        self.sparse_embed = (self.dist_on and self.world > 1 and has_embed) if sparse_embed is None else (bool(sparse_embed) and has_embed)     # (world 1: measured 1.4 ms of pure overhead)
        if self.sparse_embed:
            self._embed_cols = int(next(p for p in self.params if getattr(p, "_g32", None) is not None and p.dim() == 2 and
                                        p._g32.data_ptr() == self.arena.flat[self.arena.block_of[self._embed_key][0]:].data_ptr()).shape[1])
        # wire_dtype: what the DENSE pieces cross the fabric in.  "auto" (default) = bf16 for the HIP model under a process group of more than one rank --
        # the reference's DeepSpeed bf16 engine reduces bf16 gradients (training.py:314-329): half the bytes of an fp32 exchange -- and fp32 otherwise;
        # None / torch.float32 = fp32.  Accumulation over micro-steps, the embedding rows and everything after the exchange stay fp32 either way, and
        # every rank widens the same bf16 sums, so replicas stay bit-identical.
        if isinstance(wire_dtype, str):
            assert wire_dtype == "auto", wire_dtype
            wire_dtype = torch.bfloat16 if (self.dist_on and self.world > 1 and is_hip_model and self.arena is not None) else None
        self.wire_dtype = wire_dtype
        self.time_comm = bool(time_comm)                                 # bench / tests: event pair around the exchange of every optimizer step
        # overlap_exchange (the reference's DeepSpeed `overlap_comm`, training.py:321-329): every micro-step's backward runs in two halves cut at
        # the Llama output -- A: loss -> lm_head / CE, text_hidden_fcs, the mask-selection head (their gradients are the TAIL of the arena: blocks
        # `model.text_hidden_fcs*`, `model.lisa_*`, `lm_head` follow the decoder layers in parameter order); B: the decoder stack (LoRA blocks) and the
        # embedding rows.  On the LAST micro-step of a window the tail's all-reduce pieces are issued between A and B and travel while B (~19 ms at two
        # images) computes; after B only the head of the arena (LoRA: 16 MB, + the embedding rows) is left to exchange.  Same kernels in the same order
        # on the same data: the arena holds the same bits as without the cut.  With hipGraphs a micro-step is TWO graphs (A, B) replayed back to back.
        # Default (None): ON under a process group of more than one rank, off otherwise.  Measured at world 1 over RCCL (profiles/r06_world1_exchange.md): the
        # two-graph micro-step costs nothing (41.98 vs 42.04 ms per micro-step).
        if overlap_exchange is None:
            overlap_exchange = self.dist_on and self.world > 1
        self.overlap_exchange = bool(overlap_exchange) and self.arena is not None and is_hip_model
        self._tail_start = None
        self._pending = None
        if self.overlap_exchange:
            tail = lambda k: k.startswith("model.text_hidden_fcs") or k.startswith("model.lisa_") or k.startswith("lm_head")
            blocks = sorted(self.arena.block_of.items(), key=lambda kv: kv[1][0])
            first = next((i for i, (k, _) in enumerate(blocks) if tail(k)), None)
            ok = first is not None and all(tail(k) for k, _ in blocks[first:]) and not any(tail(k) for k, _ in blocks[:first])
            assert ok, "overlap_exchange: the arena's tail is not exactly {text_hidden_fcs, lisa_*, lm_head} (parameter order changed?)"
            self._tail_start = blocks[first][1][0]
        self.comm_ms = []
        self.micro = 0
        self.opt_steps = 0
        self.is_hip_model = is_hip_model
        self.use_graph = bool(use_graph) and self.arena is not None and is_hip_model
        self.graph_warmup = graph_warmup
        self._graphs = {}
        # max_graphs: bound on the captured graphs kept alive (each owns the activations of a whole micro-step: ~15 GB at two images).  A benchmark sees
        # one batch structure, BASELINE configs[3] three; a real loader produces a new (sequences, token length, labelled-token count, [SEG] positions)
        # combination for most batches -- those run eagerly (`graph_warmup` calls before a capture), and when the cache is full the least recently
        # replayed graph is dropped.  None = unbounded.
        self.max_graphs = max_graphs
        self._graph_clock = 0
        self.graph_error = None
        self.grad_hook = None
        self.reduce_chunk = max(1, int(reduce_chunk_mb * (1 << 20) // 4))       # fp32 elements per all-reduce chunk of the arena
        # weight-gradient kernels of the arena on a side stream (autograd.Leaves).  Measured (profiles/r04f): 43.81 ms with it vs 42.95 ms without at
        # two images per step, 347.5 vs 346.6 ms at 24 -- a replayed hipGraph already runs its kernels back to back, and the second branch only
        # takes CUs from the GEMM chain.  Off by default; kept as a switch.
        self.leaf_stream = bool(leaf_stream)
        self.check_every = int(check_every)                                      # > 0: `check_replicas()` after every N-th optimizer step (two 16-byte all-reduces)
        self.rank = dist.get_rank() if self.dist_on else 0
        if is_hip_model and self.dist_on:
            module.set_dropout_rank(self.rank)                                      # every rank draws its own LoRA-dropout masks (idempotent: derived from the base seed)
        if self.dist_on and self.ddp is None and sync_init:
            self.sync_params()                                                      # what DDP's constructor / DeepSpeed's engine do: rank 0's weights everywhere
            if hasattr(module, "__dict__"):
                module.__dict__.setdefault("_weight_hooks", []).append(self.sync_params)

    # ------------------------------------------------------------------------------------------------ replica consistency
    def sync_params(self, src=0):
        """Broadcast rank `src`'s trainable parameters (one flat buffer, one collective), then re-read the fp32 masters from them.
        Runs at construction and after every wholesale weight change (`load_state_dict` / checkpoint load -> `_weight_hooks`), so a
        per-rank difference at init (LoRA init under different seeds, a partial load) cannot persist: only gradients are exchanged later.
        COLLECTIVE: while a Trainer with a process group is attached, `model.load_state_dict` / `load_checkpoint` must be called on EVERY
        rank (as the reference's `model_engine.load_checkpoint` is, training.py:404-421); a rank-0-only load would wait here forever.
        Pass `sync_init=False` and call `trainer.sync_params()` yourself to control when the broadcast happens."""
        if not self.dist_on or not self.params:
            return
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in self.params])
            dist.broadcast(flat, src=src)
            o = 0
            for p in self.params:
                n = p.numel()
                p.detach().copy_(flat[o:o + n].view(p.shape))
                o += n
        if hasattr(self.opt, "resync_master"):
            self.opt.resync_master()

    def replica_checksum(self):
        """[sum, sum of squares] (float64) over the trainable parameters and -- when the optimizer keeps them -- its fp32 masters and
        moments.  `check_replicas()` compares it across ranks."""
        ts = [p.detach() for p in self.params]
        for k in ("master", "m", "v"):
            ts += list(getattr(self.opt, k, []))
        acc = torch.zeros(2, dtype=torch.float64, device=ts[0].device)
        for t in ts:
            d = t.double()
            acc[0] += d.sum()
            acc[1] += (d * d).sum()
        return acc

    def check_replicas(self):
        """Assert that every rank holds the same trainable state (all-reduce MIN / MAX of the checksum)."""
        c = self.replica_checksum()
        if self.dist_on:
            lo, hi = c.clone(), c.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert torch.equal(lo, hi), f"replicas diverged: checksum range {lo.tolist()} .. {hi.tolist()}"
        return c

    def close(self):
        """Detach from the model: arena views, weight hooks, captured graphs (a model can then be handed to another Trainer)."""
        if self.arena is not None:
            self.arena.detach()
        hooks = getattr(self.module, "__dict__", {}).get("_weight_hooks", [])
        for h in (getattr(self.opt, "resync_master", None), self.sync_params):
            if h is not None and h in hooks:
                hooks.remove(h)
        self._graphs = {}

    # ------------------------------------------------------------------------------------------------ micro-step
    def micro_step(self, batch, plan=None):
        """Forward + backward of one micro-batch; runs the optimizer on every `grad_accum`-th call.  Returns the loss dict.
        `plan` (HIP model only): the batch's `BatchPlan`; built here when absent (one device synchronisation)."""
        last = (self.micro + 1) % self.accum == 0
        drop_on = self.is_hip_model and getattr(self.module.config.llama, "lora_dropout", 0.0) > 0
        if drop_on:
            self.module.advance_dropout()                # a new mask every micro-step in EVERY gradient mode (arena, .grad, DDP wrapper)
        if self.fused > 1:
            assert self.arena is not None and self.is_hip_model and plan is not None and plan.micro == self.fused, \
                "fused_accum: pass the merged batch with its plan (make_plan(..., micro_batches=fused_accum))"
        if self.arena is not None and self.is_hip_model:
            if plan is None:
                plan = self.module.make_plan(**batch)
            out = self._graph_step(batch, plan, last) if self.use_graph else self._eager_step(batch, plan, last)
        elif self.arena is not None:
            out = self.module(**batch)
            out["loss"].backward()
        else:
            fwd = self.ddp if self.ddp is not None else self.module
            sync_ctx = contextlib.nullcontext() if (last or self.ddp is None) else self.ddp.no_sync()
            with sync_ctx:
                out = fwd(**batch)
                out["loss"].backward()
        if drop_on and self.fused > 1:
            self.module.advance_dropout(self.fused - 1)  # segment j of the fused pass used offset + j: the next pass starts after the window
        self.micro += 1
        if last:
            self.optimizer_step()
        return out

    # ------------------------------------------------------------------------------------------------ accumulation window with batched towers
    def encode_window(self, batches, prefetch=False):
        """The two FROZEN towers (SAM ViT-H / DINOv2, CLIP-L + mm_projector: no gradient, LISA.py:173-199, clip_encoder.py:41-60) for ALL images of
        one accumulation window in ONE pass: their GEMMs run at accum x B x 4096 rows instead of B x 4096 (SAM at two images per micro-step:
        M = 8192 -> 81 920, where the same kernels reach 1.1-1.35 instead of 0.74-1.04 PFLOP/s), the window attention fills the chip at batch 1.
        -> `WindowTowers`: per micro-batch views (tower_visual_j, tower_clip_j) into the two result tensors -- inputs of the micro-steps
        (`window_step`), which then run CLIP -> Llama -> head without either tower.
        prefetch=True: issued on the model's side stream (after everything already queued on the current stream), so the pass for window w + 1
        can run beside the micro-steps of window w; `WindowTowers.wait()` makes the current stream wait for it."""
        imgs = torch.cat([b["images"] for b in batches], 0)
        clips = torch.cat([b["images_clip"] for b in batches], 0)
        cur = torch.cuda.current_stream()
        side = self.module._tower_stream() if prefetch else None
        if side is not None:
            side.wait_stream(cur)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            vis, clip = self.module.encode_towers(imgs, clips)
            ev = torch.cuda.Event()
            ev.record()
        if side is not None:
            imgs.record_stream(side)
            clips.record_stream(side)
        n_img = imgs.shape[0]
        rv, rc = vis.shape[0] // n_img, clip.shape[0] // n_img
        parts, o = [], 0
        for b in batches:
            n = b["images"].shape[0]
            parts.append((vis[o * rv:(o + n) * rv], clip[o * rc:(o + n) * rc]))
            o += n
        return WindowTowers(parts, ev, (vis, clip), side)

    def window_step(self, batches, plans=None, towers=None):
        """One whole accumulation window (`grad_accum` micro-batches -> one optimizer step; reference loop training.py:532-547) with the frozen
        towers batched over the window: `encode_window` once, then the micro-steps with the tower outputs as inputs.  The trainable part sees
        exactly the tensors it would have computed itself from `images` / `images_clip`, so the arena holds the same gradient (bit for bit whenever
        the towers' kernels pick the same summation order at both row counts -- tests/backward_checks.py::check_window_towers).
        `towers`: a `WindowTowers` issued earlier (`encode_window(batches, prefetch=True)`).  -> list of the micro-steps' loss dicts."""
        assert len(batches) == self.accum and self.micro % self.accum == 0, "window_step: one call = one accumulation window"
        assert self.is_hip_model and self.arena is not None
        if towers is None:
            towers = self.encode_window(batches)
        towers.wait()
        outs = []
        for j, (b, (tv, tc)) in enumerate(zip(batches, towers.parts)):
            mb = {k: v for k, v in b.items() if k not in ("images", "images_clip")}
            mb.update(images=None, images_clip=None, tower_visual=tv, tower_clip=tc)
            out = self.micro_step(mb, None if plans is None else plans[j])
            # (a replayed hipGraph returns ITS output buffers: the next replay of the same structure overwrites them -- hand out copies of the scalars)
            outs.append({k: (v.detach().clone() if torch.is_tensor(v) and v.numel() == 1 else v) for k, v in out.items()})
        return outs

    def _backward(self, loss):
        """loss.backward() with the arena's weight-gradient kernels on the side stream (`autograd.Leaves`), joined before returning."""
        from .autograd import Leaves
        Leaves.on = self.leaf_stream
        try:
            loss.backward()
        finally:
            Leaves.join()
            Leaves.on = False

    def _eager_step(self, batch, plan, last=False):
        if not self.overlap_exchange:
            out = self.module.model_forward(**batch, plan=plan)
            self._backward(out["loss"])
            return out
        self.module.__dict__["_split_backward"] = True
        try:
            out = self.module.model_forward(**batch, plan=plan)
            root, leaf = self.module.__dict__.pop("_split_pair")
        finally:
            self.module.__dict__.pop("_split_backward", None)
        self._backward(out["loss"])                      # half A: everything downstream of the Llama output
        if last:
            self._issue_tail_exchange()
        root.backward(leaf.grad)                         # half B: the decoder stack, LoRA blocks, embedding rows
        return out

    def _issue_tail_exchange(self):
        """Between the two halves of the window's last backward: the arena's tail is final -- its all-reduce pieces leave now and travel beside half B."""
        if self.dist_on and self._pending is None:
            self._pending = self._issue_dense([(self._tail_start, self.arena.flat.numel())])

    def _issue_dense(self, spans):
        """Asynchronous all-reduce of arena spans in `reduce_chunk`-element pieces (bf16 copies when `wire_dtype` says so) -> [(piece, wire buffer, work)]."""
        flat = self.arena.flat
        pieces = [flat[o:min(o + self.reduce_chunk, b)] for a, b in spans for o in range(a, b, self.reduce_chunk)]
        wire = [pc if self.wire_dtype in (None, pc.dtype) else pc.to(self.wire_dtype) for pc in pieces]
        return [(pc, wb, dist.all_reduce(wb, async_op=True)) for pc, wb in zip(pieces, wire)]

    def _graph_step(self, batch, plan, last=False):
        ent = self._graphs.setdefault((plan.sig, _input_sig(batch)), {"calls": 0, "graph": None})
        if ent["graph"] is None and (ent["calls"] < self.graph_warmup or self.graph_error is not None):
            ent["calls"] += 1
            return self._eager_step(batch, plan, last)       # eager warm-up (lazy caches, workspace) -- also a real micro-step
        if ent["graph"] is None:
            if self.max_graphs is not None:
                live = [(e.get("used", 0), k) for k, e in self._graphs.items() if e.get("graph") is not None]
                while len(live) >= max(1, int(self.max_graphs)):
                    live.sort()
                    _, victim = live.pop(0)
                    del self._graphs[victim]                     # its graph, input buffers and memory pool go with it
            try:
                self._capture(ent, batch, plan)
            except Exception as e:                       # noqa: BLE001 -- fall back to eager launches, keep training
                self.graph_error = repr(e)
                import warnings
                warnings.warn(f"llmseg_amd.Trainer: hipGraph capture of the micro-step failed ({self.graph_error}); training continues with eager "
                              "launches at roughly half the speed", RuntimeWarning, stacklevel=3)
                torch.cuda.synchronize()
                ent["graph"] = None
                return self._eager_step(batch, plan, last)
        self._graph_clock += 1
        ent["used"] = self._graph_clock
        _copy_batch(ent["batch"], batch)
        if plan is not ent["last_plan"]:                 # plans are immutable once built: the same object again needs no second upload
            ent["plan"].copy_tensors_from(plan)
            ent["last_plan"] = plan
        ent["graph"].replay()
        if ent.get("graph_b") is not None:               # overlap_exchange: the micro-step is two graphs, the tail's exchange leaves between them
            if last:
                self._issue_tail_exchange()
            ent["graph_b"].replay()
        return ent["out"]

    def graph_buffers(self, batch, plan):
        """The captured graph's own input tensors for batches of this structure (None before the capture): passing THESE objects to
        `micro_step` skips the copy."""
        ent = self._graphs.get((plan.sig, _input_sig(batch)))
        return None if ent is None or ent.get("graph") is None else ent["batch"]

    def _capture(self, ent, batch, plan):
        """Capture fwd+bwd of one micro-step.  The graph reads its inputs from buffers IT OWNS (clones of this call's tensors): every
        later micro-step copies its batch into them (~80 MB at two images: 25 us of a 44 ms step), so a caller's tensors are never
        aliased, overwritten or read stale (round 3 captured the caller's own tensors and skipped the copy on a pointer match: a loader
        rotating several resident sets then replayed on the set that was present at capture).  A loader that wants zero copies writes
        straight into `graph_buffers(batch, plan)`."""
        sb = {k: ([x.clone() for x in v] if isinstance(v, (list, tuple)) else v.clone()) for k, v in _graph_inputs(batch).items()}
        for k in ("labels", "attention_masks", "offset"):          # positional arguments of model_forward that a planned forward never reads
            sb[k] = batch.get(k)
        for k in ("images", "images_clip"):                        # absent when the frozen towers' outputs are the inputs (`window_step`)
            sb.setdefault(k, None)
        torch.cuda.synchronize()
        gplan = plan.clone()
        # capture_error_mode="thread_local": with a process group the RCCL watchdog thread polls its work items with hipEventQuery; under the default "global" mode
        # that call, made by ANOTHER thread while this one captures, is an error ("operation not permitted when stream is capturing") that aborts the process --
        # seen in round 6 when a capture followed a collective within the watchdog's polling interval (bench.py under torchrun, --small).  Only this thread's own
        # unsafe calls need to be errors.
        mode = dict(capture_error_mode="thread_local")
        g = torch.cuda.CUDAGraph()
        gb = keep = None
        if not self.overlap_exchange:
            with torch.cuda.graph(g, **mode):
                out = self.module.model_forward(**sb, plan=gplan)
                self._backward(out["loss"])
        else:
            self.module.__dict__["_split_backward"] = True
            try:
                with torch.cuda.graph(g, **mode):        # graph A: forward + the backward of everything downstream of the Llama output
                    out = self.module.model_forward(**sb, plan=gplan)
                    root, leaf = self.module.__dict__.pop("_split_pair")
                    self._backward(out["loss"])
            finally:
                self.module.__dict__.pop("_split_backward", None)
                self.module.__dict__.pop("_split_pair", None)
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=g.pool(), **mode):    # graph B: the decoder stack's backward from the gradient graph A left in `leaf.grad`
                root.backward(leaf.grad)
            keep = (leaf, leaf.grad)                     # (graph B reads this buffer on every replay)
        ent.update(graph=g, graph_b=gb, keep=keep, batch=sb, plan=gplan, last_plan=plan, out={k: v.detach() for k, v in out.items() if torch.is_tensor(v)})

    # ------------------------------------------------------------------------------------------------ optimizer step
    def optimizer_step(self):
        lr = warmup_decay_lr(self.opt_steps, self.lr, self.warmup, self.total)
        scale = 1.0 / (self.accum * self.fused)
        if self.arena is not None:
            ss = self._reduce_and_sumsq()
            if self.dist_on:
                scale /= self.world
        else:
            ss = self.opt.grad_sumsq()                  # gradients hold the SUM over `accum` micro-steps (DDP already averaged over ranks)
        # Every rank applies THE SAME clip coefficient without a collective: the all-reduced arena is bit-identical on every rank and the
        # squared-norm reduction has a fixed summation order (no atomics since round 4; round 3 needed a 4-byte all-reduce(MAX) here because
        # the last bit of `ss` varied run to run and replicas drifted).  `check_every` re-verifies the replicas periodically.
        if self.grad_hook is not None:                   # observer (tests): the reduced gradient sum + its squared norm, before the update consumes them
            self.grad_hook(self, ss)
        norm = torch.sqrt(ss) * scale
        coef = torch.clamp(self.clip / (norm + 1e-6), max=1.0) * scale if self.clip and self.clip > 0 else torch.full_like(norm, scale)
        self.opt.step(lr, coef.reshape(1).float().contiguous())
        if self.arena is not None:
            self.arena.zero_()
        else:
            for p in self.params:
                p.grad = None
        self.opt_steps += 1
        if self.check_every > 0 and self.dist_on and self.opt_steps % self.check_every == 0:
            self.check_replicas()
        return float(lr)

    def _reduce_and_sumsq(self):
        """The data-parallel exchange + the global gradient norm.  Order of issue: (1) the embedding block's row set and the 8-byte all-gather of its
        size -- the one host synchronisation of an optimizer step, taken BEFORE anything large is queued, so the host does not sit behind the dense
        pieces; (2) the dense part of the flat fp32 arena as `reduce_chunk`-element all-reduce pieces, ALL issued asynchronously (RCCL runs them back
        to back on its own stream: per-link ring traffic is the same as one call); (3) the all-gather of the embedding rows, queued behind them;
        (4) stream-level waits, bf16 wire pieces widened back into the arena; (5) ONE squared-norm reduction over the finished arena in a fixed
        order (identical on every rank: the clip coefficient needs no collective).  Also issued at world size 1, where the collectives are the
        identity.  -> device fp32 [1] sum of squares."""
        flat = self.arena.flat
        sumsq = self.opt.sumsq_flat if hasattr(self.opt, "sumsq_flat") else None
        ss = torch.zeros(1, device=flat.device, dtype=torch.float32)
        if not self.dist_on:
            return sumsq(flat) if sumsq is not None else (self.opt.ops.sumsq(flat, ss), ss)[1]
        ev = None
        if self.time_comm and flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        rows_plan = self._embed_rows_plan() if self.sparse_embed else None
        # dense part: everything but the embedding block when that one travels as rows
        spans = [(0, flat.numel())]
        if self.sparse_embed:
            eo, en = self.arena.block_of[self._embed_key]
            spans = [(a, b) for a, b in ((0, eo), (eo + en, flat.numel())) if b > a]
        issued, self._pending = (self._pending or []), None
        if issued:                                       # overlap_exchange: the tail left between the two halves of the last backward
            spans = [(a, min(b, self._tail_start)) for a, b in spans if a < self._tail_start]
        issued = issued + self._issue_dense(spans)
        if rows_plan is not None:
            self._exchange_embed_rows(*rows_plan)        # queued behind the dense pieces on the collective stream
        for pc, wb, w in issued:
            w.wait()                                     # stream-level wait on a device backend: the host runs ahead
            if wb is not pc:
                pc.copy_(wb)                             # back to the fp32 arena (every rank widens the same bf16 sums: replicas stay identical)
        if sumsq is not None:
            ss = sumsq(flat)
        else:
            self.opt.ops.sumsq(flat, ss)
        if ev is not None:
            ev[1].record()
            self._comm_events = getattr(self, "_comm_events", []) + [ev]
        return ss

    def _embed_rows_plan(self):
        """-> (rows, n): this rank's non-zero rows of the embedding block (ascending, distinct: `nonzero` of a row mask) and the largest count over
        the ranks (one 8-byte all-gather + the ONLY host synchronisation of an optimizer step)."""
        eo, en = self.arena.block_of[self._embed_key]
        block = self.arena.flat[eo:eo + en].view(-1, self._embed_cols)
        rows = ((torch.amax(block, 1) != 0) | (torch.amin(block, 1) != 0)).nonzero().flatten()      # (a NaN row compares != 0: it travels and shows up in the norm)
        cnt = torch.tensor([rows.numel()], device=block.device, dtype=torch.int64)      # (`nonzero` already synchronised with the host)
        cnts = [torch.zeros_like(cnt) for _ in range(self.world)]
        dist.all_gather(cnts, cnt)
        return rows, max(1, int(torch.stack(cnts).max()))

    def _exchange_embed_rows(self, rows, n):
        """Data-parallel sum of the embedding table's gradient block as ROWS: index lists padded to the largest count `n`, all-gather of indices
        [world, n] and rows [world, n, H]; then the block is rebuilt as the sum over ranks IN RANK ORDER (`index_add_` with indices that are distinct
        within a rank -- `rows` comes from `nonzero` -- so no two additions race on an element; padding adds +0.0 to row 0): every rank holds the same bits."""
        eo, en = self.arena.block_of[self._embed_key]
        H = self._embed_cols
        block = self.arena.flat[eo:eo + en].view(-1, H)
        idx = torch.full((n,), -1, device=block.device, dtype=torch.int64)
        idx[: rows.numel()] = rows
        vals = torch.zeros((n, H), device=block.device, dtype=block.dtype)
        vals[: rows.numel()] = block[rows]
        all_idx = [torch.empty_like(idx) for _ in range(self.world)]
        all_val = [torch.empty_like(vals) for _ in range(self.world)]
        dist.all_gather(all_idx, idx)
        dist.all_gather(all_val, vals)
        block.zero_()
        for i_r, v_r in zip(all_idx, all_val):           # rank order: the same sum on every rank
            block.index_add_(0, i_r.clamp(min=0), v_r * (i_r >= 0).to(v_r.dtype)[:, None])

    def comm_times_ms(self):
        """`time_comm`: milliseconds the compute stream spent per optimizer step from issuing the gradient exchange to holding the reduced arena
        and its norm (nothing else runs on it meanwhile: this IS the exposed time of the exchange); clears the record."""
        evs, self._comm_events = getattr(self, "_comm_events", []), []
        if evs:
            torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in evs]

    # ------------------------------------------------------------------------------------------------ checkpoint (reference: training.py:404-421, 460-477)
    def state_dict(self):
        return {"opt": self.opt.state_dict() if hasattr(self.opt, "state_dict") else None, "micro": self.micro, "opt_steps": self.opt_steps}

    def load_state_dict(self, sd):
        if sd.get("opt") is not None and hasattr(self.opt, "load_state_dict"):
            self.opt.load_state_dict(sd["opt"])
        self.micro, self.opt_steps = int(sd["micro"]), int(sd["opt_steps"])
