"""Data-parallel training step for the hot path: plain torch DDP over RCCL/xGMI in place of the reference's DeepSpeed
ZeRO-2 engine (reference `training.py:292-332` config, `:480-602` loop).

What is kept from the reference's recipe: micro-batches with gradient accumulation (`no_sync()` on all but the last
micro-step, so ONE bucketed all-reduce of the ~0.58 GB trainable-gradient set per optimizer step), AdamW(betas, wd 0) on fp32
master weights with bf16 model copies, global-norm clipping at 1.0, WarmupDecayLR (linear 0 -> lr over 100 steps, then linear
decay to 0 at total_steps).  ZeRO sharding is dropped on purpose: optimizer state for the trainable set is ~3.5 GB on a
288 GB device.  DeepSpeed itself is not installed here: PARITY UNPINNED for optimizer/schedule details (published formulas).

The optimizer / clipping arithmetic runs in libllmseg_hip.so (`llmseg_adamw`, `llmseg_sumsq`); `opt_step` can be replaced
(tests inject a CPU restatement to exercise the distributed/accumulation logic under gloo).
"""
import contextlib
import math

import torch
import torch.distributed as dist


def warmup_decay_lr(step, lr, warmup=100, total=5000):
    """DeepSpeed WarmupDecayLR with warmup_type='linear', warmup_min_lr=0 (training.py:304-313). `step` counts optimizer steps
    already taken."""
    if step < warmup:
        return lr * step / max(1, warmup)
    return lr * max(0.0, (total - step) / max(1.0, total - warmup))


class HipAdamW:
    """fp32 master / m / v per trainable parameter; fused update + bf16 write-back in one kernel per parameter."""

    def __init__(self, params, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0):
        from . import ops
        self.ops = ops
        self.params = list(params)
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        self.master = [p.detach().float().clone() for p in self.params]
        self.m = [torch.zeros_like(t) for t in self.master]
        self.v = [torch.zeros_like(t) for t in self.master]
        self.t = 0

    def grad_sumsq(self):
        acc = torch.zeros(1, device=self.params[0].device, dtype=torch.float32)
        for p in self.params:
            if p.grad is not None:
                self.ops.sumsq(p.grad.contiguous(), acc)
        return acc

    def step(self, lr, grad_scale):
        """grad_scale: device fp32 scalar multiplying every gradient (1/accum x clip coefficient)."""
        self.t += 1
        for p, w, m, v in zip(self.params, self.master, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            if p.is_contiguous():
                self.ops.adamw_(p.data, w, g, m, v, lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t, grad_scale)
            else:   # parameter is a strided view: update a contiguous copy and write it back
                tmp = p.data.contiguous()
                self.ops.adamw_(tmp, w, g, m, v, lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t, grad_scale)
                p.data.copy_(tmp)


class Trainer:
    """One process per GPU.  `module(**batch)` must return a dict with a scalar "loss"."""

    def __init__(self, module, lr=3e-4, betas=(0.9, 0.95), weight_decay=0.0, clip=1.0, grad_accum=10, warmup=100, total_steps=5000,
                 optimizer=None, device_ids=None, force_ddp=False):
        self.module = module
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.ddp = None
        if self.world > 1 or (force_ddp and dist.is_initialized()):
            self.ddp = torch.nn.parallel.DistributedDataParallel(module, device_ids=device_ids, broadcast_buffers=False,
                                                               gradient_as_bucket_view=False)
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.opt = optimizer if optimizer is not None else HipAdamW(self.params, betas, weight_decay=weight_decay)
        self.lr, self.clip, self.accum, self.warmup, self.total = lr, clip, grad_accum, warmup, total_steps
        self.micro = 0
        self.opt_steps = 0

    def micro_step(self, batch):
        """Forward + backward of one micro-batch; runs the optimizer on every `grad_accum`-th call.  Returns the loss dict."""
        last = (self.micro + 1) % self.accum == 0
        fwd = self.ddp if self.ddp is not None else self.module
        sync_ctx = contextlib.nullcontext() if (last or self.ddp is None) else self.ddp.no_sync()
        with sync_ctx:
            out = fwd(**batch)
            out["loss"].backward()
        self.micro += 1
        if last:
            self.optimizer_step()
        return out

    def optimizer_step(self):
        lr = warmup_decay_lr(self.opt_steps, self.lr, self.warmup, self.total)
        # gradients hold the SUM over `accum` micro-steps (DDP already averaged over ranks): scale by 1/accum, then clip
        ss = self.opt.grad_sumsq()
        norm = torch.sqrt(ss) / self.accum
        coef = torch.clamp(self.clip / (norm + 1e-6), max=1.0) / self.accum if self.clip and self.clip > 0 else torch.full_like(norm, 1.0 / self.accum)
        self.opt.step(lr, coef.reshape(1).float().contiguous())
        for p in self.params:
            p.grad = None
        from . import autograd
        autograd.PARAM_EPOCH += 1            # parameters changed in place: invalidate operand caches derived from them
        self.opt_steps += 1
        return float(lr)
