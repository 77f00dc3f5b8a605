"""train.py (the reference's entry point on the HIP engine) and the per-step log of internevo_amd/trainlog.py."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(os.path.dirname(__file__), "golden")

CFG = """
model_type = "INTERNLM2_PUBLIC"
data = dict(seq_len=128, micro_num=2, micro_bsz=1, total_steps={steps}, train_folder=None, fixed_random_dataset_seqlen=True)
grad_scaler = dict(fp16=dict(initial_scale=2**16, min_scale=1, growth_interval=1000), growth_factor=2, backoff_factor=0.5, max_scale=2**24, hysteresis=2)
hybrid_zero_optimizer = dict(overlap_sync_grad=True, overlap_sync_param=False, reduce_bucket_size=512 * 1024 * 1024, clip_grad_norm=1.0)
loss = dict(label_smoothing=0)
adam = dict(lr=1e-3, adam_beta1=0.9, adam_beta2=0.95, adam_beta2_c=0, adam_eps=1e-8, weight_decay=0.01)
lr_scheduler = dict(total_steps={steps}, init_steps=0, warmup_ratio=0.01, eta_min=1e-5, last_epoch=-1)
model = dict(checkpoint=False, num_attention_heads=4, vocab_size=512, hidden_size=256, num_layers=2, no_bias=True, mlp_ratio=3.5,
             dtype="torch.bfloat16", layer_norm_epsilon=1e-5, num_kv_attention_heads=2, use_flash_attn=True)
parallel = dict(zero1=dict(size=-1), tensor=dict(size=1, mode="mtp"), pipeline=dict(size=1), weight=dict(size=1))
ckpt = dict(enable_save_ckpt={save}, save_ckpt_folder="local:{folder}", checkpoint_every=2, load_ckpt_info={load}, auto_resume=False)
"""


def _load_info(path):
    return f'dict(path="local:{path}", content=("all",), ckpt_type="internevo")'


def test_megatron_flops_and_tgs_windows_match_reference_golden():
    """get_megatron_flops against the value the REAL reference function returned (tests/golden/ops.json) and the TGS windows against a
    hand-rolled replay of pipeline.py:511-545."""
    import json

    from internevo_amd.trainlog import TgsStatistic, get_megatron_flops, line

    ops = json.load(open(os.path.join(G, "ops.json")))
    got = get_megatron_flops(1.0, checkpoint=False, seq_len=4096, hidden_size=4096, num_layers=32, vocab_size=92544, global_batch_size=4,
                             global_world_size=1, mlp_ratio=3.5)
    assert abs(got - ops["flops_7b_internlm2_4096"]) <= 1e-9 * got
    got = get_megatron_flops(2.0, checkpoint=True, seq_len=2048, hidden_size=4096, num_layers=32, vocab_size=103168, global_batch_size=16,
                             global_world_size=8, mlp_ratio=8 / 3)
    assert abs(got - ops["flops_ckpt"]) <= 1e-9 * got
    t = TgsStatistic()
    toks, times = [16384.0] * 12, [0.5 + 0.01 * i for i in range(12)]
    for i, (a, b) in enumerate(zip(toks, times)):
        w = t.update(a, b)
        assert w["tgs/last_tgs_1"] == round(a / b, 2)
        assert w["tgs/tgs_all"] == round(sum(toks[: i + 1]) / sum(times[: i + 1]), 2)
        assert w["tgs/tgs_avg"] == round(sum(round(x / y, 2) for x, y in zip(toks[: i + 1], times[: i + 1])) / (i + 1), 2)
    assert w["tgs/last_tgs_10"] == round(sum(toks[:10]) / sum(times[:10]), 2) and w["tgs/last_tgs_50"] == 0
    assert line({"a": 1, "b": {"x": 2.0}}) == "a=1 b={'x': 2.0} "


@pytest.mark.gpu
def test_train_entry_runs_saves_and_resumes(dev, tmp_path):
    """python train.py --config <an InternEvo config file> --launcher torch: 4 steps with a checkpoint every 2, then a second
    run resumed from the step-2 checkpoint must reproduce steps 2 and 3 exactly (same loss, grad norm, metric)."""
    sys.path.insert(0, ROOT)
    import train

    folder = str(tmp_path / "ckpts")
    cfg1 = tmp_path / "cfg1.py"
    cfg1.write_text(CFG.format(steps=4, save=True, folder=folder, load="None"))
    lines = []
    run1 = train.main(["--config", str(cfg1), "--launcher", "torch"], log=lines.append)
    assert len(run1) == 4 and sorted(os.listdir(folder)) == ["2", "4"]
    keys = list(run1[0])
    assert keys[:13] == ["tflops", "step", "loss", "tgs (tokens/gpu/second)", "tgs/last_tgs_1", "tgs/tgs_all", "tgs/tgs_avg", "tgs/tgs_SMA",
                         "tgs/last_tgs_10", "tgs/last_tgs_50", "lr", "loss_scale", "grad_norm"], "the reference's key order (pipeline.py:556-570)"
    assert keys[13:23] == ["micro_num", "num_consumed_tokens", "inf_nan_skip_batches", "num_samples_in_batch", "largest_length", "largest_batch",
                           "smallest_batch", "adam_beta2", "fwd_bwd_time", "acc"]
    assert run1[3]["num_consumed_tokens"] == 4 * 2 * 128 and run1[0]["loss"] > run1[3]["loss"]
    assert any(l.startswith("tflops=") for l in lines)
    cfg2 = tmp_path / "cfg2.py"
    cfg2.write_text(CFG.format(steps=4, save=False, folder=folder, load=_load_info(os.path.join(folder, "2"))))
    run2 = train.main(["--config", str(cfg2), "--launcher", "torch"], log=lines.append)
    assert [r["step"] for r in run2] == [2, 3]
    for a, b in zip(run1[2:], run2):
        assert (a["loss"], a["grad_norm"], a["acc"], a["perplexity"], a["lr"]) == (b["loss"], b["grad_norm"], b["acc"], b["perplexity"], b["lr"])
        assert a["num_consumed_tokens"] == b["num_consumed_tokens"], "context.pt carries the token count across the restart"
    # the folder is a complete InternEvo checkpoint: model, optimizer shard + plan, and the logging rank's run state
    from internevo_amd import checkpoint as C

    assert sorted(os.listdir(os.path.join(folder, "2"))) == ["2.step", "context.pt", "gpus-1_wp-0_tp-0_dp-0_pp-0_zo-0.pt", "model_tp0_pp0.pt",
                                                             "optimizer_tp0_pp0_zo0.pt", "sampler.pt", "schedulder.pt", "topo_tp0_pp0.json"]
    rs = C.load_run_state(os.path.join(folder, "2"))
    assert rs["context"] == dict(batch_count=1, num_consumed_samples_in_epoch=4, num_consumed_tokens=2 * 2 * 128, inf_nan_skip_batches=0,
                                 step_count=2, tensorboard_folder=None)
    assert rs["sampler"]["batch_count"] == 2 and rs["scheduler"]["after_scheduler_dict"]["last_epoch"] == 2
    # auto_resume (the reference's DEFAULT, checkpoint_manager.py:296-305): load_ckpt_info is overridden by the latest complete checkpoint under
    # save_ckpt_folder -- the folder with the largest {step}.step flag, here "4" -- and by nothing (a new run) when there is none
    import shutil

    shutil.rmtree(os.path.join(folder, "4"))       # ("2" is the latest complete checkpoint now)
    cfg3 = tmp_path / "cfg3.py"
    cfg3.write_text(CFG.format(steps=4, save=True, folder=folder, load=_load_info(str(tmp_path / "nowhere"))).replace(", auto_resume=False", ""))
    run3 = train.main(["--config", str(cfg3), "--launcher", "torch"], log=lines.append)
    assert [r["step"] for r in run3] == [2, 3] and any("Found latest ckpt" in ln and ln.rstrip(".").endswith("step: 2") for ln in lines)
    assert [(r["loss"], r["grad_norm"]) for r in run3] == [(r["loss"], r["grad_norm"]) for r in run2]
    # ... and with saving disabled the reference drops save_ckpt_folder (initialize/launch.py:219-225): auto_resume finds nothing there and starts a new run
    cfg3b = tmp_path / "cfg3b.py"
    cfg3b.write_text(CFG.format(steps=2, save=False, folder=folder, load=_load_info(str(tmp_path / "nowhere"))).replace(", auto_resume=False", ""))
    run3b = train.main(["--config", str(cfg3b), "--launcher", "torch"], log=lines.append)
    assert [r["step"] for r in run3b] == [0, 1], "enable_save_ckpt=False: old checkpoints under save_ckpt_folder are not resumed from"
    cfg4 = tmp_path / "cfg4.py"
    cfg4.write_text(CFG.format(steps=2, save=False, folder=str(tmp_path / "empty"), load=_load_info(str(tmp_path / "nowhere"))).replace(", auto_resume=False", ""))
    run4 = train.main(["--config", str(cfg4), "--launcher", "torch"], log=lines.append)
    assert [r["step"] for r in run4] == [0, 1], "auto_resume with nothing saved yet: a new run (the shipped configs' load_ckpt_info placeholders are never opened)"
    # auto_resume off: content = ("model",) takes the weights only -- step 0 of a fresh schedule on the trained weights; a missing folder is an error
    cfg5 = tmp_path / "cfg5.py"
    cfg5.write_text(CFG.format(steps=4, save=False, folder=folder, load=_load_info(os.path.join(folder, "2")).replace('("all",)', '("model",)')))
    run5 = train.main(["--config", str(cfg5), "--launcher", "torch"], log=lines.append)
    assert [r["step"] for r in run5] == [0, 1, 2, 3] and run5[0]["loss"] < run1[1]["loss"] and run5[0]["lr"] == run1[0]["lr"]
    cfg6 = tmp_path / "cfg6.py"
    cfg6.write_text(CFG.format(steps=2, save=False, folder=folder, load=_load_info(str(tmp_path / "nowhere"))))
    with pytest.raises(FileNotFoundError):
        train.main(["--config", str(cfg6), "--launcher", "torch"], log=lines.append)


@pytest.mark.gpu
def test_train_entry_skips_the_batches_data_skip_batches_names(dev, tmp_path):
    """data.skip_batches = "1" (train.py:187,208-212 of the reference): batch 1 is drawn from the loader and not trained on -- the run logs the reference's line,
    reports steps 0 and 2 only, and step 2 sees the very batch an unskipped run sees at step 2 (the data stream moved on)."""
    sys.path.insert(0, ROOT)
    import train

    base = CFG.format(steps=3, save=False, folder=str(tmp_path / "none"), load="None")
    plain, skip = tmp_path / "plain.py", tmp_path / "skip.py"
    plain.write_text(base)
    skip.write_text(base.replace("fixed_random_dataset_seqlen=True)", 'fixed_random_dataset_seqlen=True, skip_batches="1")'))
    lines = []
    run_plain = train.main(["--config", str(plain), "--launcher", "torch"], log=lines.append)
    run_skip = train.main(["--config", str(skip), "--launcher", "torch"], log=lines.append)
    assert [r["step"] for r in run_plain] == [0, 1, 2] and [r["step"] for r in run_skip] == [0, 2]
    assert sum(ln == "Skip batch count:`1`..." for ln in lines) == 1
    data_keys = ("num_samples_in_batch", "largest_length", "largest_batch", "smallest_batch")
    assert all(run_skip[1][k] == run_plain[2][k] for k in data_keys) and run_skip[0]["loss"] == run_plain[0]["loss"]
    assert run_skip[1]["loss"] != run_plain[2]["loss"]    # (one optimizer step fewer)


@pytest.mark.gpu
def test_train_entry_default_model_type_is_the_dense_internlm1_model_and_resumes(dev, tmp_path):
    """A config WITHOUT `model_type` (configs/7B_sft.py) is the reference's dense InternLM-1 model (launch.py:78-79): train.py runs it on the dense
    engine's InternLM-1 block, writes InternEvo checkpoints every 2 steps and a second run resumed from the step-2 folder reproduces steps 2 and 3 exactly."""
    sys.path.insert(0, ROOT)
    import train

    v1 = CFG.replace('model_type = "INTERNLM2_PUBLIC"\n', "").replace("no_bias=True, mlp_ratio=3.5", "mlp_ratio=8 / 3").replace("num_kv_attention_heads=2, ", "")
    assert "model_type" not in v1
    folder = str(tmp_path / "ckpts")
    cfg1 = tmp_path / "cfg1.py"
    cfg1.write_text(v1.format(steps=4, save=True, folder=folder, load="None"))
    lines = []
    run1 = train.main(["--config", str(cfg1), "--launcher", "torch"], log=lines.append)
    assert len(run1) == 4 and sorted(os.listdir(folder)) == ["2", "4"] and run1[0]["loss"] > run1[3]["loss"]
    assert sorted(os.listdir(os.path.join(folder, "2"))) == ["2.step", "context.pt", "gpus-1_wp-0_tp-0_dp-0_pp-0_zo-0.pt", "model_tp0_pp0.pt", "optimizer_tp0_pp0_zo0.pt",
                                                             "sampler.pt", "schedulder.pt", "topo_tp0_pp0.json"]
    sd = torch.load(os.path.join(folder, "2", "model_tp0_pp0.pt"), weights_only=False)
    assert list(sd)[:3] == ["model.embedding.weight", "model.blocks.0.mixer.Wqkv.weight", "model.blocks.0.mixer.Wqkv.bias"]
    cfg2 = tmp_path / "cfg2.py"
    cfg2.write_text(v1.format(steps=4, save=False, folder=folder, load=_load_info(os.path.join(folder, "2"))))
    run2 = train.main(["--config", str(cfg2), "--launcher", "torch"], log=lines.append)
    assert [r["step"] for r in run2] == [2, 3]
    for a, b in zip(run1[2:], run2):
        assert (a["loss"], a["grad_norm"], a["lr"], a["loss_scale"]) == (b["loss"], b["grad_norm"], b["lr"], b["loss_scale"])


@pytest.mark.gpu
def test_train_entry_on_a_tokenized_folder(dev, tmp_path):
    """data.train_folder = a folder of tokenized .bin / .meta files (tests/golden/folder_fixture.py): the loader's packs (pinned
    against the real pipeline in test_tokenized_folder_pipeline_matches_reference) drive the HIP engine; the metric splits by the
    folder's dataset types; the per-type token counts add up to the tokens with a label."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, G)
    import train
    from folder_fixture import write_folder

    root = str(tmp_path / "data")
    write_folder(root)
    cfg = tmp_path / "cfg.py"
    text = CFG.format(steps=3, save=False, folder=str(tmp_path / "ck"), load="None")
    text = text.replace("seq_len=128, micro_num=2, micro_bsz=1,", "seq_len=64, micro_num=3, micro_bsz=2, min_length=5,").replace(
        "train_folder=None", f"train_folder={root!r}")
    cfg.write_text(text)
    lines = []
    run = train.main(["--config", str(cfg), "--launcher", "torch"], log=lines.append)
    assert [r["step"] for r in run] == [0, 1, 2]
    for r in run:
        assert r["loss"] == r["loss"] and r["loss"] < 7.0 and r["grad_norm"]["0_default"] > 0
        assert {"acc/cn", "acc/en", "tokens/cn", "tokens/en", "loss/cn", "loss/en"} <= set(r)
        assert r["tokens/cn"] + r["tokens/en"] > 0 and r["tokens/cn"] + r["tokens/en"] <= 3 * 128
    assert run[2]["num_consumed_tokens"] == 3 * 3 * 128
    # the same batches as the loader alone yields
    from internevo_amd.data import FolderLoader

    it = iter(FolderLoader(root, 64, 2, 3, 5))
    _, labels = next(it)
    assert run[0]["tokens/cn"] + run[0]["tokens/en"] == int((labels > 0).sum()) or run[0]["tokens/cn"] + run[0]["tokens/en"] == int((labels != -100).sum())


@pytest.mark.gpu
def test_train_entry_validates_every_n_steps(dev, tmp_path):
    """data.valid_every = 2: after successful steps 2 and 4 the loop runs evaluate_on_val_dls over the default validation set and
    prints the reference's line; the numbers are those of a direct forward_only sweep."""
    sys.path.insert(0, ROOT)
    import train

    cfg = tmp_path / "cfg.py"
    text = CFG.format(steps=4, save=False, folder=str(tmp_path / "ck"), load="None")
    cfg.write_text(text.replace("train_folder=None,", "train_folder=None, valid_every=2, valid_micro_num=4, valid_folder=None,"))
    lines = []
    run = train.main(["--config", str(cfg), "--launcher", "torch"], log=lines.append)
    val = [l for l in lines if l.startswith("Validation on val: ")]
    assert len(val) == 2 and val[0].startswith("Validation on val: step=2 val/val_loss=") and " val/val_acc=" in val[0] and " val/val_plex=" in val[0]
    assert "val/val_loss" not in run[0] and "val/val_loss" in run[1] and "val/val_loss" in run[3]
    assert run[3]["val/val_loss"] < run[1]["val/val_loss"] < 6.3 and 0.0 <= run[3]["val/val_acc"] <= 1.0
    # training is unaffected by the interleaved evaluation: same losses as the run without it
    cfg2 = tmp_path / "cfg2.py"
    cfg2.write_text(text)
    run2 = train.main(["--config", str(cfg2), "--launcher", "torch"], log=lines.append)
    assert [r["loss"] for r in run] == [r["loss"] for r in run2] and [r["grad_norm"] for r in run] == [r["grad_norm"] for r in run2]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_line_contract(dev):
    """`python bench.py` prints ONE JSON line with the driver's keys, the roofline object of the dominant kernel and the CPU baseline
    (here on the tiny plumbing config so that it takes seconds; the driver runs the default = configs[1], InternLM2-7B)."""
    import json
    import subprocess

    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "tiny", "--steps", "2", "--warmup", "1"], capture_output=True,
                         text=True, timeout=500, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "tokens/s" and d["dtype"] == "bf16" and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - d["config"]["tokens_per_step"]) <= 1e-6 * d["config"]["tokens_per_step"]
    r = d["roofline"]
    assert set(r) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["launches"] > 0
    c = d["cpu_baseline"]
    assert set(c) >= {"value", "unit", "cores", "kind", "sample"} and c["kind"] == "port" and c["unit"] == "tokens/s" and c["value"] > 0 and c["cores"] >= 1


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_line_of_two_staged_ranks(dev):
    """bench.py's OWN N > 1 code -- the self-launch under torch.distributed.run on 127.0.0.1, the barrier + max-over-ranks timing, the one-hot world-size
    vector, `params_in_sync_across_ranks`, the exposed-wait reduction -- on a one-GPU box: two ranks on cuda:0 with host-staged collectives (the
    IE_BENCH_BACKEND=gloo + --staged-test hook).  What the first 8-GPU run of the driver executes on RCCL is this code path with the other Backend; the line
    must say that THIS run is not a measurement.  (Metric definition: train/pipeline.py:506-509,550-556.)"""
    import json
    import subprocess

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", IE_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "tiny", "--staged-test", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=500, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]   # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    c = d["comm"]
    assert c["rccl_world_size_per_rank"] == [2, 2], c
    assert c["params_in_sync_across_ranks"] is True
    assert "NOT a measurement" in c["backend"] and "gloo" in c["backend"]
    assert c["data_parallel_size"] == 2 and c["zero_shards_per_bucket"] == 2 and c["zero_replicas"] == 1
    # two ranks = twice the tokens of one rank per step (weak scaling: per-GPU work fixed), value = whole-job tokens / the slowest rank's time
    assert d["config"]["tokens_per_step"] > 0 and d["config"]["tokens_per_step"] % 2 == 0
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - d["config"]["tokens_per_step"]) <= 1e-6 * d["config"]["tokens_per_step"]
    assert abs(d["tgs"] * 2 - d["value"]) <= 1e-9 * d["value"]
    w = c["exposed_wait_ms_per_step"]
    assert w["waits_per_step_mean"] > 0 and set(w["max_over_ranks"]) >= {"reduce_scatter", "all_gather"}, w   # the staged collectives are waited for on the stream
    # the same launch WITHOUT the command-line flag must refuse (a stray environment variable must not turn a benchmark into a host-staged run)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "tiny", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert bad.returncode != 0
