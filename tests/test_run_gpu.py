"""GPU: the training / evaluation driver (`python -m llmseg_amd.run`; reference `training.py:336-602`) end to end on a tiny model + the built-in
synthetic dataset: from dataset items -> `collate_fn_new` -> `dict_to_cuda` -> `make_plan` -> `Trainer.micro_step` (hipGraphs per batch
structure) -> `validate` / `validate_threshold` -> save-if-better -> `--auto_resume`.  The claim that matters: a run killed after an epoch and
relaunched with the same command reproduces the uninterrupted run BIT FOR BIT (weights, Adam state, dropout stream, data stream)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(seed_data=3):
    from llmseg_amd.synthetic import SyntheticSegDataset
    from oracle.stub_tokenizer import StubTokenizer            # the tests' stand-in for the LLaVA sentencepiece tokenizer (none offline)
    from tests import backward_checks as bc
    cfg, m, sd, _ = bc._lora_case("sam")
    tok = StubTokenizer()
    train = SyntheticSegDataset(64, "cuda", img_size=cfg.sam.img, inference=False, seed=seed_data)
    val = SyntheticSegDataset(3, "cuda", img_size=cfg.sam.img, inference=True, seed=seed_data + 1)
    return m, tok, train, val


def _argv(tmp, name, *extra):
    return ["--log_base_dir", str(tmp), "--exp_name", name, "--epochs", "3", "--steps_per_epoch", "2", "--grad_accumulation_steps", "2", "--batch_size", "1",
            "--lr", "2e-3"] + list(extra)


def _masters(m):
    return torch.cat([p.detach().float().flatten().cpu() for p in m.trainable_parameters()])


def test_kill_and_auto_resume_equals_the_uninterrupted_run(tmp_path):
    from llmseg_amd import run
    logs = []
    m, tok, train, val = _setup()
    w0 = _masters(m)
    sa = run.main(_argv(tmp_path, "a", "--no_eval"), model=m, tokenizer=tok, train_dataset=train, val_dataset=None, log=logs.append)
    wa = _masters(m)
    assert sa["saved"] == [2, 4, 6] and sa["opt_steps"] == 6 and not torch.equal(w0, wa)
    assert any("Epoch: [2][2/2]" in l and "CeLoss" in l and "AlignLoss" in l and "RegressionLoss" in l for l in logs), logs[-3:]
    # the same command, killed when epoch 1 starts to report (epoch 0's checkpoint is on disk) ...

    class Kill(Exception):
        pass

    def log_kill(msg):
        if "Epoch: [1]" in msg:
            raise Kill()
    m, tok, train, val = _setup()
    with pytest.raises(Kill):
        run.main(_argv(tmp_path, "b", "--no_eval"), model=m, tokenizer=tok, train_dataset=train, val_dataset=None, log=log_kill)
    assert open(os.path.join(tmp_path, "b", "ckpt_model", "latest")).read() == "global_step2"
    # ... and relaunched in a fresh process' worth of state: new model object (same pretrained init), new trainer, --auto_resume finds ckpt_model
    m, tok, train, val = _setup()
    sb = run.main(_argv(tmp_path, "b", "--no_eval"), model=m, tokenizer=tok, train_dataset=train, val_dataset=None, log=logs.append)
    wb = _masters(m)
    assert sb["start_epoch"] == 1 and sb["optimizer_restored"] and [e["epoch"] for e in sb["epochs"]] == [1, 2] and sb["opt_steps"] == 6
    assert torch.equal(wa, wb), f"resumed run differs from the uninterrupted one: {(wa != wb).sum().item()} elements, max {(wa - wb).abs().max().item():.3e}"
    # the last epoch's meters too (same data, same weights)
    assert sa["epochs"][-1]["train"] == sb["epochs"][-1]["train"], (sa["epochs"][-1]["train"], sb["epochs"][-1]["train"])


def test_validation_save_if_better_and_eval_only(tmp_path):
    from llmseg_amd import run, validate as V
    m, tok, train, val = _setup()
    s = run.main(_argv(tmp_path, "c", "--epochs", "2", "--window_towers"), model=m, tokenizer=tok, train_dataset=train, val_dataset=val, log=lambda x: None)
    assert len(s["epochs"]) == 2 and all(0.0 <= e["giou"] <= 1.0 and 0.0 <= e["ciou"] <= 1.0 for e in s["epochs"])
    best = max(e["giou"] for e in s["epochs"])
    assert s["best_score"] == best and bool(s["saved"]) == (best > 0.0)
    assert [e["is_best"] for e in s["epochs"]] == [s["epochs"][0]["giou"] > 0.0, s["epochs"][1]["giou"] > max(0.0, s["epochs"][0]["giou"])]
    # --eval_only on the weights the run left: one validate_threshold(0.5) pass == calling the loop directly
    e = run.main(_argv(tmp_path, "d", "--eval_only", "--no_auto_resume"), model=m, tokenizer=tok, train_dataset=None, val_dataset=val, log=lambda x: None)
    from functools import partial
    from llmseg_amd.collate import collate_fn_new
    m.eval()
    direct = V.validate_threshold(m, run.val_samples(val, partial(collate_fn_new, tokenizer=tok), torch.device("cuda", 0)), threshold=0.5)
    assert e["eval"]["images"] == 3 and e["eval"]["giou"] == direct["giou"] and e["eval"]["ciou"] == direct["ciou"]


def _two_rank_driver(rank, world, port, ret, tmp):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from llmseg_amd import run
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, tok, train, val = _setup()
        logs = []
        s = run.main(_argv(tmp, "ddp", "--epochs", "1"), model=m, tokenizer=tok, train_dataset=train, val_dataset=val, log=logs.append)
        ret[rank] = (s["opt_steps"], s["epochs"][0]["giou"], s["epochs"][0]["ciou"], s["epochs"][0]["train"], _masters(m), len(logs), s["saved"])
    finally:
        dist.destroy_process_group()


def test_driver_under_two_ranks_sharing_the_gpu(tmp_path):
    """The driver with a process group of two (gloo; both ranks on cuda:0 -- no 2-GPU box is available to the build): each rank draws its own slice of the data stream, the gradient
    exchange keeps the replicas bit-identical, the loss meters and the gIoU / cIoU sums are reduced over the ranks (same numbers on both), only rank 0 logs and writes the checkpoint."""
    import torch.multiprocessing as mp
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_two_rank_driver, args=(world, 29611, ret, str(tmp_path)), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    assert a[0] == b[0] == 2 and torch.equal(a[4], b[4]), (a[4] - b[4]).abs().max().item()      # replicas identical after two optimizer steps
    assert a[1] == b[1] and a[2] == b[2] and a[3] == b[3], (a[1:4], b[1:4])                     # reduced validation metrics and train meters agree
    assert a[5] > 0 and b[5] == 0                                                                # rank 0 logs
    if a[6]:
        assert sorted(os.listdir(os.path.join(tmp_path, "ddp", "ckpt_model"))) == ["global_step2", "latest"]
