"""Checkpoint interchange: read the reference's DeepSpeed checkpoints, save / resume this trainer in the same
directory layout.

The reference saves with `model_engine.save_checkpoint(log_dir/ckpt_model)` and resumes with `load_checkpoint` + the `latest` tag file
(reference `training.py:404-421,460-477`; layout documented in its `README.md:118-130`):

    <dir>/latest                                   text file holding the tag, e.g. "global_step5000"
    <dir>/<tag>/mp_rank_00_model_states.pt         {"module": state_dict of the engine's module, "global_steps": ..., ...}
    <dir>/<tag>/bf16_zero_pp_rank_<r>_mp_rank_00_optim_states.pt     ZeRO-2 optimizer partitions, one per data-parallel rank

The module is `PeftModel(LISAForCausalLM)`: its keys carry the prefix `base_model.model.`; LoRA keys are `...q_proj.lora_A.default.weight`.
Everything after the prefix is a key of this package's state dict (llmseg_amd/params.py keeps the reference's names), except buffers that
are recomputed here (`rotary_emb.inv_freq`, `pixel_mean` / `pixel_std`).  SAM's prompt encoder / mask decoder are loaded when the model
was built with `LisaConfig(sam_decoder=True)` (`evaluate()` reads them) and reported under `ignored` otherwise.

Written by this package: the same directory layout and the same `module` key names, so the reference's loader (and ours) can read the
weights; the optimizer state is ONE file in this package's own format (`llmseg_optim_states.pt`: fp32 master weights, Adam moments, step
counters) -- DeepSpeed's rank-partitioned flat buffers are not WRITTEN; they are READ (`load_zero2_optimizer_states`, round 5; PARITY UNPINNED:
deepspeed==0.10.0 is absent, the layout is restated from its published zero_to_fp32 logic).
"""
import os
import re
import warnings

import torch

PEFT_PREFIX = "base_model.model."
_DROP = (re.compile(r"\.rotary_emb\.inv_freq$"), re.compile(r"^model\.visual_model\.pixel_(mean|std)$"))      # buffers recomputed here
_LORA = re.compile(r"\.lora_[AB]\.")


def reference_key(name):
    """Key of a reference checkpoint (`module` dict of mp_rank_00_model_states.pt) -> key of this package's state dict, or None for a
    buffer this package recomputes.  Whether the model HAS the key (e.g. the SAM decoder tensors) is decided against the model's own
    parameter set in `load_reference_checkpoint`, not here."""
    for pfx in ("module.", PEFT_PREFIX):
        if name.startswith(pfx):
            name = name[len(pfx):]
    if any(p.search(name) for p in _DROP):
        return None
    return name


def resolve(path):
    """A checkpoint directory (with `latest`), a tag directory or the model-states file itself -> (model-states file, tag)."""
    if os.path.isdir(path):
        latest = os.path.join(path, "latest")
        if os.path.exists(latest):
            with open(latest) as fh:
                tag = fh.read().strip().splitlines()[0].strip()
            return os.path.join(path, tag, "mp_rank_00_model_states.pt"), tag
        return os.path.join(path, "mp_rank_00_model_states.pt"), os.path.basename(os.path.normpath(path))
    return path, os.path.basename(os.path.dirname(path))


def load_reference_checkpoint(model, path, strict=False):
    """Load the weights of a DeepSpeed checkpoint of the reference (or one written by `save_checkpoint`) into `model`.
    -> dict(missing=[...], ignored=[...], tag=..., global_steps=...).  Tensors of the model that the checkpoint lacks are listed in
    `missing`; anything but LoRA matrices among them raises a `RuntimeWarning` (they stay at their initial values), `strict=True` raises."""
    f, tag = resolve(path)
    try:
        blob = torch.load(f, map_location="cpu", weights_only=True)
    except Exception:                                   # DeepSpeed pickles config objects next to the tensors
        blob = torch.load(f, map_location="cpu", weights_only=False)
    sd_in = blob["module"] if "module" in blob else blob
    sd, ignored = {}, []
    for k, v in sd_in.items():
        nk = reference_key(k)
        if nk is None or not torch.is_tensor(v):
            ignored.append(k)
        else:
            sd[nk] = v
    own = {n for n, _ in model.params.named_parameters()}
    ignored += [k for k in sd if k not in own]
    sd = {k: v for k, v in sd.items() if k in own}
    missing, _ = model.load_state_dict(sd, strict=False)
    missing = list(missing)
    if strict and missing:
        raise KeyError(f"checkpoint lacks {len(missing)} tensors, e.g. {missing[:5]}")
    hard = [k for k in missing if not _LORA.search(k)]
    if hard:
        warnings.warn(f"checkpoint {f} lacks {len(hard)} non-LoRA tensors of the model (left at their initial values), e.g. {hard[:4]}",
                      RuntimeWarning, stacklevel=2)
    return {"missing": missing, "ignored": ignored, "tag": tag, "global_steps": blob.get("global_steps") if isinstance(blob, dict) else None}


def save_checkpoint(save_dir, model, trainer=None, global_step=0, trainable_only=False, rank=0):
    """Write <save_dir>/global_step<N>/{mp_rank_00_model_states.pt, llmseg_optim_states.pt} + <save_dir>/latest (rank 0 writes)."""
    tag = f"global_step{int(global_step)}"
    d = os.path.join(save_dir, tag)
    if rank != 0:
        return d
    os.makedirs(d, exist_ok=True)
    req = {n for n, p in model.params.named_parameters() if p.requires_grad}
    module = {PEFT_PREFIX + k: v.detach().to("cpu") for k, v in model.state_dict().items() if (not trainable_only or k in req)}
    torch.save({"module": module, "global_steps": int(global_step), "dp_world_size": getattr(trainer, "world", 1), "mp_world_size": 1,
                "writer": "llmseg_amd"}, os.path.join(d, "mp_rank_00_model_states.pt"))
    if trainer is not None:
        st = trainer.state_dict()
        st["param_names"] = [n for n, p in model.params.named_parameters() if p.requires_grad]
        if hasattr(model, "dropout_state"):
            st["dropout_state"] = torch.tensor([model.dropout_base_seed(), int(model.dropout_state()[1])], dtype=torch.int64)     # BASE seed + offset
        torch.save(st, os.path.join(d, "llmseg_optim_states.pt"))
    with open(os.path.join(save_dir, "latest"), "w") as fh:
        fh.write(tag)
    return d


def load_zero2_optimizer_states(tag_dir, model, trainer):
    """Adam state of a DeepSpeed ZeRO-2 bf16 checkpoint (`bf16_zero_pp_rank_<r>_mp_rank_00_optim_states.pt`, one file per data-parallel rank,
    next to `mp_rank_00_model_states.pt`) -> this trainer's fp32 masters / moments / step, so that a resume from an authors' TRAINING checkpoint
    (`training.py:404-421`) continues Adam instead of restarting it.  PARITY UNPINNED: deepspeed==0.10.0 is absent; the layout is restated from its
    published `zero_to_fp32.py` / `stage_1_and_2.py`:
      * the model-states file holds `param_shapes`: a list (one entry per optimizer param group) of {parameter name: shape} in the order the group was
        flattened (the engine's trainable parameters in `named_parameters()` order);
      * every rank file holds `optimizer_state_dict` with `single_partition_of_fp32_groups[g]` (that rank's slice of group g's flat fp32 master vector) and
        `base_optimizer_state["state"][g]` = {"step", "exp_avg", "exp_avg_sq"} (the same slice of the moments);
      * the flat vector of a group = its parameters concatenated, zero-padded to a multiple of 2 x world, cut into `world` equal partitions.
    Raises when the files do not have that shape; -> dict(restored=[names], skipped=[names of the file the model does not train], step=int)."""
    import glob
    # TRUST ASSUMPTION (here and in `load_reference_checkpoint`): DeepSpeed's rank files pickle more than tensors, so they are unpickled in full --
    # load only checkpoints you would hand to the reference's own `model_engine.load_checkpoint` (training.py:404-421).
    ms = torch.load(os.path.join(tag_dir, "mp_rank_00_model_states.pt"), map_location="cpu", weights_only=False)
    shapes = ms.get("param_shapes")
    if not shapes:
        raise ValueError("the model-states file carries no `param_shapes` (not a DeepSpeed ZeRO checkpoint)")
    shapes = shapes if isinstance(shapes, (list, tuple)) else [shapes]
    files = sorted(glob.glob(os.path.join(tag_dir, "*zero_pp_rank_*_mp_rank_00_optim_states.pt")),
                   key=lambda f: int(re.search(r"zero_pp_rank_(\d+)_", os.path.basename(f)).group(1)))
    if not files:
        raise FileNotFoundError(f"no *zero_pp_rank_<r>_mp_rank_00_optim_states.pt under {tag_dir}")
    parts = [torch.load(f, map_location="cpu", weights_only=False)["optimizer_state_dict"] for f in files]
    world = len(parts)
    if any(int(p.get("zero_stage", 2)) > 2 for p in parts):
        raise ValueError("ZeRO-3 checkpoints partition every parameter separately: not read here (the reference trains with stage 2, training.py:321)")
    names = [n for n, p in model.params.named_parameters() if p.requires_grad]
    slot = {n: i for i, n in enumerate(names)}
    restored, skipped, step = [], [], 0
    for g, group in enumerate(shapes):
        flat = {"master": torch.cat([p["single_partition_of_fp32_groups"][g].float().reshape(-1) for p in parts])}
        st = [p["base_optimizer_state"]["state"][g] for p in parts]
        flat["m"] = torch.cat([s_["exp_avg"].float().reshape(-1) for s_ in st])
        flat["v"] = torch.cat([s_["exp_avg_sq"].float().reshape(-1) for s_ in st])
        g_step = st[0].get("step")
        if g_step is None:                                   # some optimizers keep `step` in the param group, not in the per-parameter state
            pgs = parts[0]["base_optimizer_state"].get("param_groups", [])
            g_step = pgs[g].get("step") if g < len(pgs) and isinstance(pgs[g], dict) else None
        if g_step is None:
            g_step = ms.get("global_steps")
        step = max(step, int(g_step or 0))
        need = sum(int(torch.Size(shp).numel()) for shp in group.values())
        align = 2 * world
        if not (need <= flat["master"].numel() <= (need + align - 1) // align * align + align):
            raise ValueError(f"group {g}: {need} parameter elements do not fit {flat['master'].numel()} partitioned elements over {world} ranks")
        off = 0
        for name, shp in group.items():
            n = int(torch.Size(shp).numel())
            key = reference_key(name)
            if key in slot:
                i = slot[key]
                for dst, src in ((trainer.opt.master[i], flat["master"]), (trainer.opt.m[i], flat["m"]), (trainer.opt.v[i], flat["v"])):
                    if tuple(dst.shape) != tuple(torch.Size(shp)):
                        raise ValueError(f"{name}: optimizer state of shape {tuple(shp)} for a parameter of shape {tuple(dst.shape)}")
                    dst.copy_(src[off:off + n].view(dst.shape))
                restored.append(key)
            else:
                skipped.append(name)
            off += n
    if step == 0 and restored and any(float(m_.abs().max()) > 0 for m_ in trainer.opt.m):
        import warnings
        warnings.warn("ZeRO-2 optimizer partitions carry warm Adam moments but no step count (state, param_groups and global_steps all lack it): bias correction "
                      "and the LR warm-up would restart at 0 -- set trainer.opt.t / trainer.opt_steps yourself", RuntimeWarning, stacklevel=2)
    trainer.opt.t = step
    return {"restored": restored, "skipped": skipped, "step": step, "not_in_file": [n for n in names if n not in set(restored)]}


def load_checkpoint(load_dir, model, trainer=None, steps_per_epoch=500):
    """Resume (`--auto_resume`, training.py:404-421): weights, then -- when this package wrote the checkpoint -- the optimizer state.
    -> dict(tag, global_steps, start_epoch, optimizer_restored)."""
    info = load_reference_checkpoint(model, load_dir)
    f, tag = resolve(load_dir)
    opt_file = os.path.join(os.path.dirname(f), "llmseg_optim_states.pt")
    restored = False
    if trainer is not None:
        if os.path.exists(opt_file):
            st = torch.load(opt_file, map_location="cpu", weights_only=True)
            names = [n for n, p in model.params.named_parameters() if p.requires_grad]
            assert st.get("param_names") == names, "optimizer state belongs to a different trainable set"
            trainer.load_state_dict(st)
            if "dropout_state" in st and hasattr(model, "dropout_state"):
                seed, off = (int(v) for v in st["dropout_state"].tolist())      # the base seed; every rank's module derives its own key
                model.set_dropout_seed(seed, off)
            restored = True
        elif hasattr(trainer.opt, "master") and [x for x in os.listdir(os.path.dirname(f)) if "zero_pp_rank_" in x and x.endswith("_optim_states.pt")]:
            z = load_zero2_optimizer_states(os.path.dirname(f), model, trainer)      # a reference TRAINING checkpoint: its ZeRO-2 Adam partitions
            restored = not z["not_in_file"]
            trainer.opt_steps = z["step"]
        elif hasattr(trainer.opt, "resync_master"):
            trainer.opt.resync_master()                 # weights only (a released checkpoint): fresh Adam moments on the loaded weights
    m = re.search(r"(\d+)$", tag or "")
    steps = int(m.group(1)) if m else (info.get("global_steps") or 0)
    return {"tag": tag, "global_steps": steps, "start_epoch": steps // max(1, steps_per_epoch), "optimizer_restored": restored,
            "missing": info["missing"], "ignored": info["ignored"]}
