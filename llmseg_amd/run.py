"""Training / evaluation driver around the hot path: the reference's `main` + `train` (`training.py:336-602`) with the DeepSpeed engine
replaced by `llmseg_amd.train.Trainer` (one process per GPU, `torch.distributed` over RCCL).

    python -m llmseg_amd.run --version <LLaVA dir> --vision_pretrained sam_vit_h_4b8939.pth --dataset_module pkg.mod:factory \
        --epochs 10 --steps_per_epoch 500 --grad_accumulation_steps 10 --lr 3e-4 [--eval_only | --no_eval] [--auto_resume]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m llmseg_amd.run ...

What is kept from the reference, flag for flag (`training.py:29-118`): the model / LoRA / loss-weight / optimizer flags, `--epochs` x
`--steps_per_epoch` optimizer steps of `--grad_accumulation_steps` micro-batches of `--batch_size` images, WarmupDecayLR over
epochs x steps_per_epoch, per-epoch `validate` (arg-max) + `validate_threshold` (`:449-452`), "save only when gIoU improves" into
`<log_base_dir>/<exp_name>/ckpt_model` with the `meta_log_giou..._ciou....pth` marker (`:460-477`), `--auto_resume` from that directory
with `start_epoch = global_step // steps_per_epoch` (`:404-421`), `--eval_only` = one `validate_threshold(threshold=0.5)` pass (`:425-434`),
the iterator restart when the loader runs dry (`:520-525`), loss meters reduced over ranks at `--print_freq` (`:556-562`).

What is NOT here (out of scope): the dataset classes and their cv2 / json pipelines, wandb / tensorboard.  Data enters through a
protocol instead: `--dataset_module pkg.mod:factory` names a callable `factory(args, tokenizer, device) -> (train_dataset, val_dataset)`; a
dataset is anything with `__len__` and `__getitem__(i) -> sample dict` in the datasets' own format (what `collate_fn_new` consumes;
`collate.reason_seg_sample` builds one from decoded inputs).  `--dataset_module synthetic` = `synthetic.SyntheticSegDataset` (plumbing / tests).
`main()` also takes ready objects (`model=`, `tokenizer=`, `train_dataset=`, `val_dataset=`): the tests drive it that way.

Differences from the reference, stated: (1) a resumed run continues the data stream where the interrupted run left it (the sampler is a seeded
permutation that can skip consumed micro-batches; the reference restarts its loader), so kill + `--auto_resume` reproduces the uninterrupted
run bit for bit -- tests/test_run_gpu.py; (2) `best_score` is restored from the `meta_log_*` marker on resume (the reference restarts it at 0,
so its first epoch after a resume always overwrites the best checkpoint); (3) loss meters accumulate on the device and are read at
`--print_freq` (the reference calls `.item()` every micro-step: one host synchronisation each)."""
import argparse
import glob
import importlib
import os
import re
import shutil
import sys
import time
import warnings
from functools import partial

import torch
import torch.distributed as dist


def parse_args(argv=None):
    """The reference's flags (training.py:29-118), same names and defaults where the path honours them.  Flags of subsystems that are out of
    scope (`--vis_save_path`, `--visualize`, the per-source data lists) are accepted so that the reference's launch scripts parse, and ignored;
    flags that would silently change the arithmetic here (`--precision` other than bf16, 8 / 4-bit loading, `--train_mask_decoder`) are rejected."""
    p = argparse.ArgumentParser(description="LLM-Seg training / evaluation on MI355X (llmseg_amd)")
    p.add_argument("--local_rank", default=int(os.environ.get("LOCAL_RANK", 0)), type=int)
    p.add_argument("--version", default="", help="HF LLaVA directory (config.json + weights + tokenizer)")
    p.add_argument("--precision", default="bf16", choices=["fp32", "bf16", "fp16"])
    p.add_argument("--image_size", default=1024, type=int, help="segmentation-backbone input side (1024: SAM ViT-H; the reference's DINOv2 default is 896)")
    p.add_argument("--model_max_length", default=512, type=int)
    p.add_argument("--lora_r", default=8, type=int)
    p.add_argument("--vision-tower", "--vision_tower", dest="vision_tower", default="openai/clip-vit-large-patch14", type=str)
    p.add_argument("--load_in_8bit", action="store_true", default=False)
    p.add_argument("--load_in_4bit", action="store_true", default=False)
    p.add_argument("--backbone", default="sam", choices=["sam", "dinov2"], help="segmentation backbone feeding the mask pooling (not a reference flag: the reference hard-wires DINOv2)")
    p.add_argument("--dataset", default="refer_seg||reason_seg", type=str)
    p.add_argument("--sample_rates", default="10, 1", type=str)
    p.add_argument("--sem_seg_data", default="ade20k||cocostuff||pascal_part||paco_lvis||mapillary", type=str)
    p.add_argument("--refer_seg_data", default="refclef||refcoco||refcoco+||refcocog", type=str)
    p.add_argument("--vqa_data", default="llava_instruct_150k", type=str)
    p.add_argument("--reason_seg_data", default="ReasonSeg|train", type=str)
    p.add_argument("--val_dataset", default="ReasonSeg|val", type=str)
    p.add_argument("--dataset_dir", default="./dataset", type=str)
    p.add_argument("--sam_masks_dir", default="./processed_data", type=str)
    p.add_argument("--dataset_module", default="", type=str, help="pkg.mod:factory -> (train_dataset, val_dataset); 'synthetic' = the built-in synthetic dataset")
    p.add_argument("--log_base_dir", default="./runs", type=str)
    p.add_argument("--exp_name", default="debug", type=str)
    p.add_argument("--epochs", default=10, type=int)
    p.add_argument("--steps_per_epoch", default=500, type=int)
    p.add_argument("--batch_size", default=1, type=int, help="batch size per device per step")
    p.add_argument("--grad_accumulation_steps", default=10, type=int)
    p.add_argument("--val_batch_size", default=1, type=int)
    p.add_argument("--workers", default=0, type=int, help="loader workers (0: samples are built in this process; the N2 target kernels run on the device)")
    p.add_argument("--lr", default=0.0003, type=float)
    p.add_argument("--ce_loss_weight", default=1.0, type=float)
    p.add_argument("--align_loss_weight", default=1.0, type=float)
    p.add_argument("--regression_loss_weight", default=1.0, type=float)
    p.add_argument("--lora_alpha", default=16, type=int)
    p.add_argument("--lora_dropout", default=0.05, type=float)
    p.add_argument("--lora_target_modules", default="q_proj,v_proj", type=str)
    p.add_argument("--explanatory", default=0.1, type=float)
    p.add_argument("--beta1", default=0.9, type=float)
    p.add_argument("--beta2", default=0.95, type=float)
    p.add_argument("--num_classes_per_sample", default=3, type=int)
    p.add_argument("--exclude_val", action="store_true", default=False)
    p.add_argument("--no_eval", action="store_true", default=False)
    p.add_argument("--eval_only", action="store_true", default=False)
    p.add_argument("--vision_pretrained", default="", type=str)
    p.add_argument("--out_dim", default=256, type=int)
    p.add_argument("--weight", default="", type=str)
    p.add_argument("--resume", default="", type=str)
    p.add_argument("--print_freq", default=1, type=int)
    p.add_argument("--start_epoch", default=0, type=int)
    p.add_argument("--gradient_checkpointing", action="store_true", default=True)
    p.add_argument("--train_mask_decoder", action="store_true", default=False)
    p.add_argument("--use_mm_start_end", action="store_true", default=True)
    p.add_argument("--auto_resume", action="store_true", default=True)
    p.add_argument("--no_auto_resume", dest="auto_resume", action="store_false", help="(the reference's --auto_resume is store_true with default True, i.e. cannot be switched off)")
    p.add_argument("--conv_type", default="llava_v1", type=str, choices=["llava_v1", "llava_llama_2"])
    p.add_argument("--visualize", action="store_true", default=False)
    p.add_argument("--vis_save_path", default="./vis_output", type=str)
    p.add_argument("--iou_selection_only", action="store_true", default=False)
    p.add_argument("--seed", default=0, type=int, help="data order / LoRA init / dropout base seed")
    p.add_argument("--no_graph", action="store_true", default=False, help="eager launches instead of hipGraph replays of the micro-step")
    p.add_argument("--max_graphs", default=8, type=int, help="captured micro-step graphs kept alive (one per batch structure; least recently used dropped)")
    p.add_argument("--window_towers", action="store_true", default=False, help="run the frozen towers once per accumulation window (Trainer.window_step)")
    args = p.parse_args(argv)
    if args.precision != "bf16":
        p.error("--precision: the HIP path computes in bf16 with fp32 accumulation (the reference's training precision, training.py:151-156)")
    if args.load_in_8bit or args.load_in_4bit:
        p.error("--load_in_8bit / --load_in_4bit: bitsandbytes quantisation is not part of this path")
    if args.train_mask_decoder:
        p.error("--train_mask_decoder: the mask decoder is not on model_forward's training path (LISA.py:340-474 never calls it)")
    if sorted(t.strip() for t in args.lora_target_modules.split(",")) != ["q_proj", "v_proj"]:
        p.error("--lora_target_modules: the fused q|k|v kernels carry LoRA on q_proj and v_proj (the reference's setting)")
    if args.val_batch_size != 1:
        p.error("--val_batch_size must be 1 (training.py:383)")
    return args


# ---------------------------------------------------------------------------------------------------------------- model / tokenizer
def init_tokenizer(args):
    """training.py:121-137: the LLaVA sentencepiece tokenizer, pad = unk, `[SEG]` + the two image tags added."""
    import transformers
    from .collate import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN
    tok = transformers.AutoTokenizer.from_pretrained(args.version, cache_dir=None, model_max_length=args.model_max_length, padding_side="right", use_fast=False)
    tok.pad_token = tok.unk_token
    tok.add_tokens("[SEG]")
    if args.use_mm_start_end:
        tok.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
    return tok


def seg_token_index(tokenizer):
    return tokenizer("[SEG]", add_special_tokens=False).input_ids[0]


def init_model(args, tokenizer, device):
    """training.py:139-243 (`init_LISA_model`) through `LISAForCausalLM.from_pretrained`."""
    from .lisa import LISAForCausalLM
    return LISAForCausalLM.from_pretrained(
        args.version, device=device, backbone=args.backbone, lora_r=args.lora_r, lora_alpha=args.lora_alpha, lora_dropout=args.lora_dropout,
        seed=args.seed, vocab_size=len(tokenizer), seg_token_idx=seg_token_index(tokenizer), out_dim=args.out_dim,
        ce_loss_weight=args.ce_loss_weight, align_loss_weight=args.align_loss_weight, regression_loss_weight=args.regression_loss_weight,
        vision_pretrained=args.vision_pretrained or None, vision_tower=args.vision_tower, use_mm_start_end=args.use_mm_start_end)


def load_datasets(args, tokenizer, device):
    spec = args.dataset_module
    if not spec:
        raise SystemExit("llmseg_amd.run: no data source.  The reference's dataset classes are out of scope here: pass --dataset_module pkg.mod:factory "
                         "(factory(args, tokenizer, device) -> (train_dataset, val_dataset), items = sample dicts as collate_fn_new takes them) or 'synthetic'.")
    if spec == "synthetic":
        from .synthetic import SyntheticSegDataset
        n = args.batch_size * args.grad_accumulation_steps * args.steps_per_epoch
        return (SyntheticSegDataset(n, device, img_size=args.image_size, inference=False, seed=args.seed),
                SyntheticSegDataset(max(2, min(8, n)), device, img_size=args.image_size, inference=True, seed=args.seed + 1))
    mod, _, fn = spec.partition(":")
    return getattr(importlib.import_module(mod), fn or "factory")(args, tokenizer, device)


# ---------------------------------------------------------------------------------------------------------------- meters / loaders
class AverageMeter:
    """utils/utils.py:55-97 with the sum kept on the device: `update` enqueues an add (no host synchronisation, and the value is read before the
    next hipGraph replay overwrites the loss buffer), `all_reduce` sums [sum, count] over the ranks, `avg` reads it back."""

    def __init__(self, name, fmt=":.4f", device="cpu"):
        self.name, self.fmt, self.device = name, fmt, device
        self.reset()

    def reset(self):
        self.total = torch.zeros(2, dtype=torch.float64, device=self.device)       # [sum, count]
        self.last = None

    def update(self, val, n=1):
        v = val.detach().to(self.total.dtype).reshape(()) if torch.is_tensor(val) else torch.tensor(float(val), dtype=self.total.dtype, device=self.device)
        self.total[0] += v * n
        self.total[1] += n
        self.last = v.clone()

    def all_reduce(self):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.total)

    @property
    def avg(self):
        s, c = self.total.tolist()
        return s / (c + 1e-5) if c else 0.0

    def __str__(self):
        val = float(self.last) if self.last is not None else 0.0
        return ("{name} {val" + self.fmt + "} ({avg" + self.fmt + "})").format(name=self.name, val=val, avg=self.avg)


class MicroBatchSampler:
    """The index stream of one rank: pass p over the dataset is a permutation seeded by (seed, p), dealt round-robin to the ranks, cut into
    micro-batches of `batch_size`; the stream continues into pass p + 1 when a pass runs dry (the reference re-creates its iterator,
    training.py:520-525).  `skip(n)` drops n micro-batches without touching the data -- a resumed run continues where the interrupted one stopped."""

    def __init__(self, n_items, batch_size, rank=0, world=1, seed=0, shuffle=True):
        assert n_items >= 1 and batch_size >= 1
        self.n, self.bs, self.rank, self.world, self.seed, self.shuffle = int(n_items), int(batch_size), rank, world, int(seed), shuffle
        self.passes = 0              # passes started (== restarts + 1)
        self._it = self._stream()

    def _pass(self, p):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed * 1000003 + p)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        pad = (-len(order)) % self.world                              # DistributedSampler semantics: wrap around so every rank gets the same count
        order += order[:pad]
        mine = order[self.rank::self.world]
        return [mine[i:i + self.bs] for i in range(0, len(mine) - self.bs + 1, self.bs)] or [(mine * self.bs)[: self.bs]]

    def _stream(self):
        while True:
            batches = self._pass(self.passes)
            self.passes += 1
            for b in batches:
                yield b

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._it)

    def skip(self, n):
        for _ in range(int(n)):
            next(self._it)


def train_batches(dataset, sampler, collate):
    """-> iterator of collated micro-batches (host side of `collate_fn_new`)."""
    for idx in sampler:
        yield collate([dataset[i] for i in idx])


def val_samples(dataset, collate, device, rank=0, world=1):
    """One collated image at a time, dealt to the ranks as `DistributedSampler(shuffle=False, drop_last=False)` deals them (training.py:385-387:
    the tail wraps around, so a few images may count twice at world > 1, as in the reference) -> the sample dicts of `validate.*`."""
    from .collate import dict_to_cuda
    from .validate import sample_from_collated
    n = len(dataset)
    order = list(range(n)) + list(range((-n) % world))
    for i in order[rank::world]:
        yield sample_from_collated(dict_to_cuda(collate([dataset[i]]), torch.bfloat16, device=device))


# ---------------------------------------------------------------------------------------------------------------- the loops
def train_epoch(trainer, model, batches, epoch, args, device, rank, log):
    """training.py:480-602: `steps_per_epoch` optimizer steps of `grad_accumulation_steps` micro-batches.  -> dict of the epoch's last meter averages."""
    from .collate import dict_to_cuda, model_kwargs
    names = (("Loss", "loss"), ("CeLoss", "ce_loss"), ("AlignLoss", "align_loss"), ("RegressionLoss", "regression_loss"))
    meters = {k: AverageMeter(n, device=device) for n, k in names}
    t_batch, t_data = AverageMeter("Time", ":6.3f"), AverageMeter("Data", ":6.3f")
    model.train()
    last, end = {}, time.time()
    width = len(str(args.steps_per_epoch))
    for global_step in range(args.steps_per_epoch):
        window, plans = [], []
        for _ in range(args.grad_accumulation_steps):
            col = next(batches)                                          # (the sampler never runs dry: it restarts its pass, training.py:520-525)
            t_data.update(time.time() - end)
            # the plan from the HOST copies of the token tensors (what the collate returns): no device -> host synchronisation per micro-step
            plans.append(model.make_plan(col["input_ids"], col["labels"], col["attention_masks"], col["offset"], sam_segs_list=col["sam_segs_list"]))
            window.append(model_kwargs(dict_to_cuda(col, torch.bfloat16, device=device)))
        outs = trainer.window_step(window, plans) if args.window_towers else None
        for j, kw in enumerate(window):
            out = outs[j] if outs is not None else trainer.micro_step(kw, plans[j])
            for k, mt in meters.items():
                mt.update(out[k], kw["images"].shape[0])
        t_batch.update(time.time() - end)
        end = time.time()
        if global_step % args.print_freq == 0:
            for mt in list(meters.values()) + [t_batch, t_data]:
                mt.all_reduce()
            last = {k: mt.avg for k, mt in meters.items()}
            if rank == 0:
                log("\t".join(["Epoch: [%d][%*d/%d]" % (epoch, width, global_step + 1, args.steps_per_epoch), str(t_batch)] + [str(mt) for mt in meters.values()]))
            for mt in list(meters.values()) + [t_batch, t_data]:
                mt.reset()
    return last


def _best_from_markers(log_dir):
    """(best gIoU, its cIoU) recorded by the `meta_log_giou{:.3f}_ciou{:.3f}.pth` markers of earlier epochs (training.py:466-473)."""
    best = (0.0, 0.0)
    for f in glob.glob(os.path.join(log_dir, "meta_log_giou*_ciou*.pth")):
        m = re.search(r"meta_log_giou([0-9.]+)_ciou([0-9.]+)\.pth$", f)
        if m:
            best = max(best, (float(m.group(1)), float(m.group(2))))
    return best


def main(argv=None, *, model=None, tokenizer=None, train_dataset=None, val_dataset=None, log=print, device=None):
    """-> summary dict (start_epoch, epochs run, per-epoch validation results, best score, checkpoints written)."""
    from . import checkpoint as ck
    from . import validate as V
    from .collate import collate_fn_new
    from .train import Trainer
    args = parse_args(argv)
    args.log_dir = os.path.join(args.log_base_dir, args.exp_name)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    own_group = False
    if world > 1 and not dist.is_initialized():
        torch.cuda.set_device(args.local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", args.local_rank))
        own_group = True
    if dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    device = torch.device("cuda", args.local_rank) if device is None else torch.device(device)
    if rank == 0:
        os.makedirs(args.log_dir, exist_ok=True)
    if tokenizer is None:
        tokenizer = init_tokenizer(args)
    if model is None:
        model = init_model(args, tokenizer, device)
    if args.weight:
        ck.load_reference_checkpoint(model, args.weight)
    model.set_trainable()
    if train_dataset is None and val_dataset is None:
        train_dataset, val_dataset = load_datasets(args, tokenizer, device)
    if train_dataset is not None and rank == 0:
        log(f"Training with {len(train_dataset)} examples.")
    if val_dataset is not None and rank == 0:
        log(f"Validation with {len(val_dataset)} examples.")
    collate = partial(collate_fn_new, tokenizer=tokenizer, conv_type=args.conv_type, use_mm_start_end=args.use_mm_start_end, local_rank=args.local_rank)
    if hasattr(model, "set_dropout_seed"):
        model.set_dropout_seed(0x5EED + args.seed, 0)
    trainer = Trainer(model, lr=args.lr, betas=(args.beta1, args.beta2), weight_decay=0.0, clip=1.0, grad_accum=args.grad_accumulation_steps, warmup=100,
                      total_steps=args.epochs * args.steps_per_epoch, use_graph=not args.no_graph, device_ids=[args.local_rank], max_graphs=args.max_graphs)
    summary = {"start_epoch": args.start_epoch, "epochs": [], "saved": [], "best_score": 0.0, "resumed_from": None}
    try:
        # resume (training.py:404-421)
        if args.auto_resume and len(args.resume) == 0:
            resume = os.path.join(args.log_dir, "ckpt_model")
            if os.path.exists(os.path.join(resume, "latest")):
                args.resume = resume
        if args.resume:
            info = ck.load_checkpoint(args.resume, model, trainer, steps_per_epoch=args.steps_per_epoch)
            args.start_epoch = info["start_epoch"]
            summary.update(start_epoch=args.start_epoch, resumed_from=args.resume, optimizer_restored=info["optimizer_restored"])
            if rank == 0:
                log("resume training from {}, start from epoch {}".format(args.resume, args.start_epoch))

        def run_validation(fn, **kw):
            assert val_dataset is not None, "validation asked for (no --no_eval) without a validation dataset"
            model.eval()
            return fn(model, val_samples(val_dataset, collate, device, rank, world), **kw)

        if args.eval_only:                                              # training.py:425-434
            r = run_validation(V.validate_threshold, threshold=0.5)
            if rank == 0:
                log("giou: {:.4f}, ciou: {:.4f}".format(r["giou"], r["ciou"]))
            summary["eval"] = r
            return summary

        best_score, cur_ciou = _best_from_markers(args.log_dir) if args.resume else (0.0, 0.0)
        sampler = MicroBatchSampler(len(train_dataset), args.batch_size, rank, world, seed=args.seed)
        sampler.skip(args.start_epoch * args.steps_per_epoch * args.grad_accumulation_steps)
        batches = train_batches(train_dataset, sampler, collate)
        is_best = False
        for epoch in range(args.start_epoch, args.epochs):
            meters = train_epoch(trainer, model, batches, epoch, args, device, rank, log)
            rec = {"epoch": epoch, "train": meters}
            if not args.no_eval:                                        # training.py:449-457
                r = run_validation(V.validate)
                if not args.iou_selection_only:
                    r = run_validation(V.validate_threshold)
                giou, ciou = r["giou"], r["ciou"]
                if rank == 0:
                    log("results from threshold: giou={}, ciou={}".format(giou, ciou))
                is_best = giou > best_score
                best_score = max(giou, best_score)
                cur_ciou = ciou if is_best else cur_ciou
                rec.update(giou=giou, ciou=ciou, is_best=is_best)
            if args.no_eval or is_best:                                 # training.py:460-477
                save_dir = os.path.join(args.log_dir, "ckpt_model")
                if rank == 0:
                    torch.save({"epoch": epoch}, os.path.join(args.log_dir, "meta_log_giou{:.3f}_ciou{:.3f}.pth".format(best_score, cur_ciou)))
                    if os.path.exists(save_dir):
                        shutil.rmtree(save_dir)
                if dist.is_initialized():
                    dist.barrier()
                ck.save_checkpoint(save_dir, model, trainer, global_step=trainer.opt_steps, rank=rank)
                if dist.is_initialized():
                    dist.barrier()
                summary["saved"].append(trainer.opt_steps)
                rec["saved"] = True
            summary["epochs"].append(rec)
        summary.update(best_score=best_score, cur_ciou=cur_ciou, opt_steps=trainer.opt_steps, sampler_passes=sampler.passes)
        if trainer.graph_error:
            warnings.warn(f"the micro-step ran eagerly: {trainer.graph_error}", RuntimeWarning)
        return summary
    finally:
        trainer.close()
        if own_group:
            dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
