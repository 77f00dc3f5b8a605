"""InternEvo checkpoint format: tests/golden/ckpt_ref/ was written by the REAL reference's
save_model_checkpoint / save_optimizer_checkpoint after 2 training steps of a tiny bf16 InternLM2 (make_golden.py --ckpt),
tests/golden/ckpt.json records the structure it saw and the 2 steps it trained afterwards."""
import json
import os

import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")
REF = os.path.join(G, "ckpt_ref")


def _cfg():
    from internevo_amd.config import tiny

    c = json.load(open(os.path.join(G, "ckpt.json")))["config"]
    return tiny(c["hidden"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], c["seq_len"], c["micro_num"], 1e-3, c["total_steps"])


def test_reference_checkpoint_loads_and_structure_matches():
    from internevo_amd import checkpoint as C

    gold = json.load(open(os.path.join(G, "ckpt.json")))
    cfg = _cfg()
    ck = C.load_checkpoint(REF, cfg.model)
    assert [["model." + n, str(ck["params"][n].dtype), list(ck["params"][n].shape)] for n in C.state_dict_order(cfg.model)] == gold["model_keys"]
    assert [n for n, _ in C.zero_flat_order([(n, tuple(ck["params"][n].shape)) for n in C.state_dict_order(cfg.model)])] != C.state_dict_order(cfg.model)
    assert ck["adam_step"] == 2 and ck["scaler"] == dict(scale=65536.0, growth_step=2, hysteresis_step=0)
    assert abs(ck["lr"] - gold["base_param_groups"][0]["lr"]) < 1e-15
    # master weights round to the bf16 parameters the model file holds
    for n, p in ck["params"].items():
        assert torch.equal(ck["master"][n].to(torch.bfloat16), p), n
        assert ck["exp_avg"][n].shape == p.shape and float(ck["exp_avg_sq"][n].min()) >= 0.0


def _same_tree(a, b, path=""):
    """dict / list trees with the same keys in the same order; floats to 1e-12 relative, everything else exactly."""
    assert type(a) is type(b) or (isinstance(a, (int, float)) and isinstance(b, (int, float)) and not isinstance(a, bool) and not isinstance(b, bool)), (path, a, b)
    if isinstance(a, dict):
        assert list(a) == list(b), (path, list(a), list(b))
        for k in a:
            _same_tree(a[k], b[k], f"{path}.{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same_tree(x, y, f"{path}[{i}]")
    elif isinstance(a, float) or isinstance(b, float):
        assert abs(a - b) <= 1e-12 * max(abs(a), abs(b)), (path, a, b)
    else:
        assert a == b, (path, a, b)


def test_scheduler_state_dict_matches_the_reference_class():
    """tests/golden/sched_state.json: state_dict() of the real FineTuneCosineAnnealingWarmupLR after n = 0..11 steps, with and
    without warm-up and init steps.  CosineWarmupLR.state_dict restates it in closed form; load_state_dict finds n back."""
    from internevo_amd.schedule import CosineWarmupLR

    gold = json.load(open(os.path.join(G, "sched_state.json")))
    for case in gold["cases"]:
        for st in case["states"]:
            if st["n"] > case["total_steps"]:
                continue  # past the end of training torch's recursive cosine leaves the closed form; no run gets there
            s = CosineWarmupLR(case["base_lr"], case["total_steps"], case["warmup_ratio"], case["eta_min"], case["init_steps"])
            s.set_successful_steps(st["n"])
            assert abs(s.lr() - st["lr"]) <= 1e-12 * max(st["lr"], 1e-30)
            _same_tree(s.state_dict(), st["state"], f"total={case['total_steps']} n={st['n']}")
            t = CosineWarmupLR(case["base_lr"], case["total_steps"], case["warmup_ratio"], case["eta_min"], case["init_steps"])
            t.load_state_dict(st["state"])
            assert t.k == st["n"]
    bad = CosineWarmupLR(1e-3, 60, 0.1, 1e-5)
    with pytest.raises(ValueError):
        bad.load_state_dict(gold["cases"][0]["states"][3]["state"])


def test_sampler_resumes_from_the_reference_sampler_file():
    """sampler.pt written by the real reference after 2 batches: loading it puts a fresh sampler where an uninterrupted one is."""
    from internevo_amd import checkpoint as C
    from internevo_amd.data import SyntheticLoader

    gold = json.load(open(os.path.join(G, "ckpt.json")))
    c = gold["config"]
    rs = C.load_run_state(REF)
    assert rs["context"]["step_count"] == 2 and rs["context"]["batch_count"] == 1 and rs["scheduler"]["after_scheduler_dict"]["last_epoch"] == 2
    straight = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"]))
    for _ in range(gold["saved_after_step"]):
        next(straight)
    resumed = SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"])
    resumed.sampler.load_state_dict(rs["sampler"])
    it = iter(resumed)
    for _ in range(3):
        (a, la), (b, lb) = next(straight), next(it)
        assert torch.equal(a["input_ids"], b["input_ids"]) and torch.equal(la, lb)
    assert C.load_run_state(REF2) == dict(scheduler=None, sampler=None, context=None)  # the dp-2 fixture holds model + optimizer only


def test_saved_files_equal_the_reference_files(tmp_path):
    """load(reference files) -> save -> the same file set, the same keys / dtypes / shapes / values, tensor for tensor."""
    from internevo_amd import checkpoint as C

    gold = json.load(open(os.path.join(G, "ckpt.json")))
    cfg = _cfg()
    ck = C.load_checkpoint(REF, cfg.model)
    out = str(tmp_path / "ck")
    C.save_checkpoint(out, cfg.model, ck["params"], ck["master"], ck["exp_avg"], ck["exp_avg_sq"], ck["adam_step"], ck["scaler"], ck["lr"],
                      dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3))
    # ... and the run state the logging rank adds (checkpoint_manager.py:608-618): scheduler, sampler, TrainState counters
    from internevo_amd.data import SyntheticLoader
    from internevo_amd.schedule import CosineWarmupLR

    c = gold["config"]
    loader = SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"])
    it = iter(loader)
    for _ in range(gold["saved_after_step"]):
        _, labels = next(it)
    sched = CosineWarmupLR(1e-3, c["total_steps"], 0.01, 1e-5)
    sched.set_successful_steps(gold["saved_after_step"])
    C.save_run_state(out, sched.state_dict(), loader.sampler.state_dict(), batch_count=gold["saved_after_step"] - 1,
                     num_consumed_samples_in_epoch=gold["saved_after_step"] * labels.shape[0],
                     num_consumed_tokens=gold["saved_after_step"] * labels.nelement(), inf_nan_skip_batches=0, step_count=gold["saved_after_step"])
    assert sorted(f for f in os.listdir(out) if not f.endswith(".step")) == gold["files"]   # (+ the {step}.step flag CheckpointManager writes last; the fixture was written by the writers directly)
    ld = lambda folder, fn: torch.load(os.path.join(folder, fn), weights_only=False)  # noqa: E731
    assert ld(out, "context.pt") == ld(REF, "context.pt") == gold["context_state"]
    _same_tree(ld(out, "schedulder.pt"), ld(REF, "schedulder.pt"))
    sa, sb = ld(REF, "sampler.pt"), ld(out, "sampler.pt")
    assert list(sa) == list(sb)
    for k in sa:
        if k == "rng_state":
            assert sa[k][0] == sb[k][0] and (sa[k][1] == sb[k][1]).all() and tuple(sa[k][2:]) == tuple(sb[k][2:])
        elif k == "indices":
            assert sa[k].dtype == sb[k].dtype and (sa[k] == sb[k]).all()
        else:
            assert sa[k] == sb[k], k
    a = torch.load(os.path.join(REF, "model_tp0_pp0.pt"), weights_only=False)
    b = torch.load(os.path.join(out, "model_tp0_pp0.pt"), weights_only=False)
    assert list(a) == list(b) and all(a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]) for k in a)
    ra, rb = C._load(os.path.join(REF, "optimizer_tp0_pp0_zo0.pt")), C._load(os.path.join(out, "optimizer_tp0_pp0_zo0.pt"))
    assert list(ra) == gold["optimizer_top_keys"] and list(ra) == list(rb)
    assert ra["zero_devide_optim_plan"] == rb["zero_devide_optim_plan"]
    assert ra["grad_scaler"] == rb["grad_scaler"]
    assert torch.equal(ra["flat_fp32_weights"][0], rb["flat_fp32_weights"][0])
    sa, sb = ra["base_optim_states"]["state"][0], rb["base_optim_states"]["state"][0]
    assert float(sa["step"]) == float(sb["step"]) and sa["step"].dtype == sb["step"].dtype
    assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
    for ga, gb in zip(ra["base_optim_states"]["param_groups"], rb["base_optim_states"]["param_groups"]):
        assert list(ga) == list(gb), (list(ga), list(gb))
        for k in ga:
            if k != "optimizer_mode":
                assert ga[k] == gb[k], k
        mode = lambda o: getattr(o, "args", (getattr(o, "value", None),))  # noqa: E731  placeholder (no reference) or the real enum
        assert mode(ga["optimizer_mode"]) == mode(gb["optimizer_mode"]) == ("zero1",)
    assert C._load(os.path.join(REF, "gpus-1_wp-0_tp-0_dp-0_pp-0_zo-0.pt")) == C._load(os.path.join(out, "gpus-1_wp-0_tp-0_dp-0_pp-0_zo-0.pt"))
    # the enum really is pickled by reference to the reference's module path (so the reference unpickles its own enum)
    import zipfile

    raw = zipfile.ZipFile(os.path.join(out, "optimizer_tp0_pp0_zo0.pt")).read("optimizer_tp0_pp0_zo0/data.pkl")
    assert b"internlm.core.context.process_group_initializer" in raw and b"ParallelMode" in raw


def test_oracle_resumes_from_the_reference_checkpoint():
    """Training resumed from the reference's checkpoint reproduces the 2 steps the reference itself trained after saving."""
    from internevo_amd import checkpoint as C
    from internevo_amd.data import SyntheticLoader
    from oracle.step import OracleTrainer

    gold = json.load(open(os.path.join(G, "ckpt.json")))
    cfg = _cfg()
    c = gold["config"]
    tr = OracleTrainer(cfg, torch.bfloat16)
    tr.load_state(C.load_checkpoint(REF, cfg.model))
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"]))
    for _ in range(gold["saved_after_step"]):
        next(loader)  # the sampler position is part of the run state: two batches were consumed before the save
    for w in gold["steps"][gold["saved_after_step"]:]:
        batch, labels = next(loader)
        g = tr.train_step(batch, labels)
        assert abs(g["loss"] - w["loss"]) <= 2e-3 * abs(w["loss"]), (g["loss"], w["loss"])
        assert abs(g["grad_norm"] - w["grad_norm"]["0_default"]) <= 1e-2 * w["grad_norm"]["0_default"]
        assert abs(g["lr"] - w["lr"]) <= 1e-9 * w["lr"] and g["loss_scale"] == w["loss_scale"]


def test_reference_resumes_from_our_checkpoint():
    """tests/golden/ckpt_load.json: the oracle trained 2 steps, save_checkpoint() + save_run_state() wrote the files, the REAL
    reference resumed from them with its own loaders in its own order (load_model_checkpoint, load_context,
    load_optimizer_checkpoint, load_scheduler, load_sampler; make_golden.py --ckpt-load) and trained 2 more steps.  The oracle,
    simply continuing, must land on the same losses, grad norms AND learning rates, on the same batches."""
    from internevo_amd.data import SyntheticLoader
    from oracle.step import OracleTrainer

    gold = json.load(open(os.path.join(G, "ckpt_load.json")))
    cfg = _cfg()
    c = gold["config"]
    assert gold["resumed_from_batch"] == 2
    assert gold["resumed_train_state"] == dict(batch_count=2, inf_nan_skip_batches=0, num_consumed_samples_in_epoch=4,
                                               num_consumed_tokens=2 * c["micro_num"] * c["seq_len"], step_count=2)
    tr = OracleTrainer(cfg, torch.bfloat16)
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"]))
    for w in gold["oracle_steps_before_save"]:
        g = tr.train_step(*next(loader))
        assert g["loss"] == w["loss"] and g["grad_norm"] == w["grad_norm"], "the oracle itself is deterministic"
    for w in gold["reference_steps_after_load"]:
        g = tr.train_step(*next(loader))
        assert abs(g["lr"] - w["lr"]) <= 1e-12 * w["lr"], "the reference's scheduler, restored from our schedulder.pt, is where ours is"
        assert abs(g["loss"] - w["loss"]) <= 2e-3 * abs(w["loss"]), (g["loss"], w["loss"])
        assert abs(g["grad_norm"] - w["grad_norm"]["0_default"]) <= 1e-2 * w["grad_norm"]["0_default"]
        assert g["loss_scale"] == w["loss_scale"] and w["ok"]


REF2 = os.path.join(G, "ckpt_ref_dp2")


def _cmp_optimizer_files(C, a_path, b_path):
    ra, rb = C._load(a_path), C._load(b_path)
    assert list(ra) == list(rb)
    assert ra["zero_devide_optim_plan"] == rb["zero_devide_optim_plan"] and ra["grad_scaler"] == rb["grad_scaler"]
    assert torch.equal(ra["flat_fp32_weights"][0], rb["flat_fp32_weights"][0])
    sa, sb = ra["base_optim_states"]["state"][0], rb["base_optim_states"]["state"][0]
    assert float(sa["step"]) == float(sb["step"]) and torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
    for ga, gb in zip(ra["base_optim_states"]["param_groups"], rb["base_optim_states"]["param_groups"]):
        assert list(ga) == list(gb) and all(ga[k] == gb[k] for k in ga if k != "optimizer_mode")


def test_two_rank_zero_partition_matches_the_reference():
    """tests/golden/ckpt_ref_dp2/: the real reference on 2 data-parallel ranks (gloo), ZeRO-1 world 2 (make_golden.py --ckpt-mp).
    The greedy whole-parameter partition restated in checkpoint.zero_partition must reproduce its plan, and load -> save
    (all ranks, or one rank at a time the way the engine's ranks write) its files tensor for tensor."""
    from internevo_amd import checkpoint as C

    gold = json.load(open(os.path.join(G, "ckpt_dp2.json")))
    cfg = _cfg()
    assert C.saved_zero_world(REF2) == 2 and C.saved_zero_world(REF) == 1
    ck = C.load_checkpoint(REF2, cfg.model)
    assert ck["zero_world"] == 2 and ck["adam_step"] == 2
    shapes = {n: tuple(ck["params"][n].shape) for n in C.state_dict_order(cfg.model)}
    order = C.zero_flat_order(list(shapes.items()))
    part = C.zero_partition(order, 2)
    assert [C._plan_ids(order, idx) for idx in part] == gold["zero_devide_optim_plan"][0]
    assert sorted(i for idx in part for i in idx) == list(range(len(order)))
    for n, p in ck["params"].items():  # the merged shards cover every parameter; masters round to the saved bf16 weights
        assert torch.equal(ck["master"][n].to(torch.bfloat16), p), n
    hyper = dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3)
    import tempfile

    with tempfile.TemporaryDirectory() as out:
        C.save_checkpoint(out, cfg.model, ck["params"], ck["master"], ck["exp_avg"], ck["exp_avg_sq"], ck["adam_step"], ck["scaler"], ck["lr"],
                          hyper, zero_world=2)
        assert sorted(f for f in os.listdir(out) if not f.endswith(".step")) == gold["files"]   # (+ the {step}.step flag CheckpointManager writes last; the fixture was written by the writers directly)
        for r in (0, 1):
            _cmp_optimizer_files(C, os.path.join(REF2, f"optimizer_tp0_pp0_zo{r}.pt"), os.path.join(out, f"optimizer_tp0_pp0_zo{r}.pt"))
            fn = f"gpus-2_wp-0_tp-0_dp-{r}_pp-0_zo-{r}.pt"
            assert C._load(os.path.join(REF2, fn)) == C._load(os.path.join(out, fn))
    with tempfile.TemporaryDirectory() as out:  # rank by rank, each rank holding only the parameters it owns
        names = C.zero_rank_names(shapes, 2)
        for r in (1, 0):
            own = lambda d: {n: d[n] for n in names[r]}  # noqa: E731
            C.save_checkpoint(out, cfg.model, ck["params"] if r == 0 else None, own(ck["master"]), own(ck["exp_avg"]), own(ck["exp_avg_sq"]),
                              ck["adam_step"], ck["scaler"], ck["lr"], hyper, zero_world=2, zero_ranks=[r], write_model=(r == 0), shapes=shapes)
        assert sorted(f for f in os.listdir(out) if not f.endswith(".step")) == gold["files"]   # (+ the {step}.step flag CheckpointManager writes last; the fixture was written by the writers directly)
        for r in (0, 1):
            _cmp_optimizer_files(C, os.path.join(REF2, f"optimizer_tp0_pp0_zo{r}.pt"), os.path.join(out, f"optimizer_tp0_pp0_zo{r}.pt"))
        a = torch.load(os.path.join(REF2, "model_tp0_pp0.pt"), weights_only=False)
        b = torch.load(os.path.join(out, "model_tp0_pp0.pt"), weights_only=False)
        assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    # `want` keeps only the asked-for optimizer tensors; a missing shard or a foreign plan is refused
    some = set(names[1][:2])
    part_ck = C.load_checkpoint(REF2, cfg.model, want=some)
    assert set(part_ck["master"]) == some == set(part_ck["exp_avg"])
    import shutil

    with tempfile.TemporaryDirectory() as out:
        shutil.copytree(REF2, out, dirs_exist_ok=True)
        os.remove(os.path.join(out, "optimizer_tp0_pp0_zo0.pt"))
        with pytest.raises(FileNotFoundError):
            C.load_checkpoint(out, cfg.model)


def test_zero_partition_properties():
    """Greedy whole-parameter partition on the 7B shapes: every parameter on exactly one rank, loads within one largest parameter
    of each other, world 1 = the flat order itself."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import internlm2_7b
    from internevo_amd.layout import FlatLayout

    mc = internlm2_7b().model
    shapes = {n: s.shape for n, s in FlatLayout(mc, 1).params.items()}
    order = C.zero_flat_order([(n, shapes[n]) for n in C.state_dict_order(mc)])
    numel = lambda shp: int(torch.Size(shp).numel())  # noqa: E731
    assert C.zero_partition(order, 1) == [list(range(len(order)))]
    for w in (2, 8, 64):
        part = C.zero_partition(order, w)
        assert sorted(i for idx in part for i in idx) == list(range(len(order)))
        loads = [sum(numel(order[i][1]) for i in idx) for idx in part]
        assert max(loads) - min(loads) <= numel(order[0][1])
        assert all(idx == sorted(idx) for idx in part)  # flat order inside a rank = sorted order


def test_oracle_resumes_from_the_two_rank_reference_checkpoint():
    """The 2 steps the 2-rank reference trained after saving: global batch = 2 ranks x micro_num micro-batches; the oracle runs
    them as one rank with twice the micro-batches (same averaged gradient up to summation order)."""
    from internevo_amd import checkpoint as C
    from internevo_amd.data import SyntheticLoader
    from oracle.step import OracleTrainer

    gold = json.load(open(os.path.join(G, "ckpt_dp2.json")))
    cfg = _cfg()
    c = gold["config"]
    cfg.train.micro_num = 2 * c["micro_num"]
    tr = OracleTrainer(cfg, torch.bfloat16)
    tr.load_state(C.load_checkpoint(REF2, cfg.model))
    loaders = [iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"], data_rank=r, data_world_size=2)) for r in (0, 1)]
    for _ in range(gold["saved_after_step"]):
        [next(l) for l in loaders]
    for w in gold["steps"][gold["saved_after_step"]:]:
        parts = [next(l) for l in loaders]
        batch = {k: (sum((p[0][k] for p in parts), []) if isinstance(parts[0][0][k], list) else torch.cat([p[0][k] for p in parts]))
                 for k in parts[0][0]}
        labels = torch.cat([p[1] for p in parts])
        g = tr.train_step(batch, labels)
        # the reference logs rank 0's LOCAL loss; the grad norm is global
        assert abs(g["grad_norm"] - w["grad_norm"]["0_default"]) <= 1e-2 * w["grad_norm"]["0_default"], (g["grad_norm"], w["grad_norm"])
        assert abs(g["lr"] - w["lr"]) <= 1e-9 * w["lr"] and g["loss_scale"] == w["loss_scale"]


REFTP = os.path.join(G, "ckpt_ref_tp2")


def test_tensor_parallel_checkpoint_shards_match_the_reference():
    """tests/golden/ckpt_ref_tp2/: the real reference on 2 Megatron tensor-parallel ranks (make_golden.py --ckpt-tp), its weights
    the tp_shard parts of the closed-form full tensors.  load_checkpoint merges the two ranks' files back into the FULL tensors
    (= the closed-form init moved by two optimizer steps: every parameter whole, replicated norms equal on both ranks); cutting the
    merged state again and saving reproduces the reference's files tensor for tensor."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import tiny

    gold = json.load(open(os.path.join(G, "ckpt_tp2.json")))
    c = gold["config"]
    cfg = tiny(c["hidden"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], c["seq_len"], c["micro_num"], 1e-3, c["total_steps"])
    assert C.saved_tp_world(REFTP) == 2 and C.saved_tp_world(REF) == 1 and C.saved_zero_world(REFTP, 1) == 1
    ck = C.load_checkpoint(REFTP, cfg.model)
    assert ck["tp_world"] == 2 and ck["zero_world"] == 1 and ck["adam_step"] == 2
    from oracle.model import param_shapes

    full = param_shapes(cfg.model)
    for n in C.state_dict_order(cfg.model):
        assert tuple(ck["params"][n].shape) == tuple(full[n]), n
        assert tuple(ck["master"][n].shape) == tuple(full[n]) and torch.equal(ck["master"][n].to(torch.bfloat16), ck["params"][n]), n
    # the two ranks' copies of a replicated parameter agree; the sharded ones really differ
    a, b = (C._load_tp_rank(REFTP, cfg.model, t, 2, None) for t in (0, 1))
    assert torch.equal(a["params"]["norm.weight"], b["params"]["norm.weight"]) and torch.equal(a["master"]["norm.weight"], b["master"]["norm.weight"])
    assert not torch.equal(a["params"]["layers.0.attention.wo.weight"], b["params"]["layers.0.attention.wo.weight"])
    assert [C.tp_split_dim(n) for n in ("tok_embeddings.weight", "output.weight", "layers.0.attention.wqkv.weight", "layers.0.attention.wo.weight",
                                        "layers.0.feed_forward.w1.weight", "layers.0.feed_forward.w2.weight", "layers.0.ffn_norm.weight")] == [1, 0, 0, 1, 0, 1, None]
    hyper = dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3)
    import tempfile

    with tempfile.TemporaryDirectory() as out:
        for t in (0, 1):
            cut = lambda d: {n: C.tp_shard(n, v, t, 2).contiguous() for n, v in d.items()}  # noqa: E731
            C.save_checkpoint(out, cfg.model, cut(ck["params"]), cut(ck["master"]), cut(ck["exp_avg"]), cut(ck["exp_avg_sq"]), ck["adam_step"],
                              ck["scaler"], ck["lr"], hyper, tp_world=2, tp_rank=t)
        assert sorted(f for f in os.listdir(out) if not f.endswith(".step")) == gold["files"]   # (+ the {step}.step flag CheckpointManager writes last; the fixture was written by the writers directly)
        for t in (0, 1):
            _cmp_optimizer_files(C, os.path.join(REFTP, f"optimizer_tp{t}_pp0_zo0.pt"), os.path.join(out, f"optimizer_tp{t}_pp0_zo0.pt"))
            x = torch.load(os.path.join(REFTP, f"model_tp{t}_pp0.pt"), weights_only=False)
            y = torch.load(os.path.join(out, f"model_tp{t}_pp0.pt"), weights_only=False)
            assert list(x) == list(y) and all(x[k].shape == y[k].shape and torch.equal(x[k], y[k]) for k in x)
            fn = f"gpus-2_wp-0_tp-{t}_dp-0_pp-0_zo-0.pt"
            assert C._load(os.path.join(REFTP, fn)) == C._load(os.path.join(out, fn))


def test_oracle_resumes_from_the_tensor_parallel_reference_checkpoint():
    """The merged full tensors ARE the model: the single-rank oracle resumed from the 2-tensor-rank reference checkpoint reproduces
    the 2 steps the tensor-parallel reference trained after saving."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import tiny
    from internevo_amd.data import SyntheticLoader
    from oracle.step import OracleTrainer

    gold = json.load(open(os.path.join(G, "ckpt_tp2.json")))
    c = gold["config"]
    cfg = tiny(c["hidden"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], c["seq_len"], c["micro_num"], 1e-3, c["total_steps"])
    tr = OracleTrainer(cfg, torch.bfloat16)
    tr.load_state(C.load_checkpoint(REFTP, cfg.model))
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"]))
    for _ in range(gold["saved_after_step"]):
        next(loader)
    for w in gold["steps"][gold["saved_after_step"]:]:
        g = tr.train_step(*next(loader))
        assert abs(g["loss"] - w["loss"]) <= 2e-3 * abs(w["loss"]), (g["loss"], w["loss"])
        assert abs(g["grad_norm"] - w["grad_norm"]["0_default"]) <= 1e-2 * w["grad_norm"]["0_default"]
        assert abs(g["lr"] - w["lr"]) <= 1e-9 * w["lr"] and g["loss_scale"] == w["loss_scale"]


@pytest.mark.parametrize("tp,zw", [(1, 3), (2, 2), (4, 1), (4, 3)])
def test_checkpoint_layouts_round_trip(tmp_path, tp, zw):
    """Any (tensor-parallel, ZeRO) layout written rank by rank -- every rank passing only what it owns -- merges back into the same
    full tensors; the flat vectors of a rank hold its parameters in the partition order with nothing lost or duplicated."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import tiny
    from oracle.model import param_shapes

    cfg = tiny(128, 2, 4, 4, 256, 32, 2, 1e-3, 4)
    order = C.state_dict_order(cfg.model)
    full_shapes = param_shapes(cfg.model)
    g = torch.Generator().manual_seed(tp * 10 + zw)
    full = {k: {n: torch.randn(full_shapes[n], generator=g) for n in order} for k in ("params", "master", "exp_avg", "exp_avg_sq")}
    full["params"] = {n: v.to(torch.bfloat16) for n, v in full["params"].items()}
    hyper = dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3)
    scaler = dict(scale=1024.0, growth_step=7, hysteresis_step=1)
    out = str(tmp_path / "ck")
    total = 0
    for t in range(tp):
        local = {k: {n: C.tp_shard(n, v, t, tp).contiguous() for n, v in d.items()} for k, d in full.items()}
        shapes = {n: tuple(local["params"][n].shape) for n in order}
        owners = C.zero_rank_names(shapes, zw)
        assert sorted(n for names in owners for n in names) == sorted(order)
        for r in range(zw):
            own = lambda d: {n: d[n] for n in owners[r]}  # noqa: E731
            C.save_checkpoint(out, cfg.model, local["params"] if r == 0 else None, own(local["master"]), own(local["exp_avg"]), own(local["exp_avg_sq"]),
                              5, scaler, 3e-4, hyper, zero_world=zw, zero_ranks=[r], write_model=(r == 0), shapes=shapes, tp_world=tp, tp_rank=t)
            st = C._load(os.path.join(out, f"optimizer_tp{t}_pp0_zo{r}.pt"))
            total += st["flat_fp32_weights"][0].numel()
            assert os.path.exists(os.path.join(out, f"gpus-{tp * zw}_wp-0_tp-{t}_dp-{r}_pp-0_zo-{r}.pt"))
    replicated = sum(int(torch.Size(full_shapes[n]).numel()) for n in order if C.tp_split_dim(n) is None)  # norms live on every tensor rank
    assert total == sum(int(torch.Size(s_).numel()) for s_ in full_shapes.values()) + (tp - 1) * replicated, "every local element in exactly one flat vector"
    ck = C.load_checkpoint(out, cfg.model)
    assert (ck["tp_world"], ck["zero_world"], ck["adam_step"], ck["lr"], ck["scaler"]) == (tp, zw, 5, 3e-4, scaler)
    for k in ("params", "master", "exp_avg", "exp_avg_sq"):
        assert list(ck[k]) == order or set(ck[k]) == set(order)
        for n in order:
            assert torch.equal(ck[k][n], full[k][n]), (k, n)
    some = {order[1], order[-1]}
    part = C.load_checkpoint(out, cfg.model, want=some)
    assert set(part["master"]) == some and all(torch.equal(part["exp_avg"][n], full["exp_avg"][n]) for n in some)


def test_more_zero_ranks_than_parameters(tmp_path):
    """ZeRO ranks that hold NO parameter (the greedy whole-parameter partition over more ranks than parameters, hybrid_zero_optim.py:254-284) write what the
    reference's do -- the group's own parameter ids, no optimizer state, no flat weights (the rule pinned on tests/golden/ckpt_ref_moe_dp4/) -- and the reader
    merges the folder without them."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import tiny
    from oracle.model import param_shapes

    cfg = tiny(128, 2, 4, 4, 256, 32, 2, 1e-3, 4)
    order = C.state_dict_order(cfg.model)
    zw = len(order) + 3
    shapes = param_shapes(cfg.model)
    g = torch.Generator().manual_seed(5)
    full = {k: {n: torch.randn(shapes[n], generator=g) for n in order} for k in ("params", "master", "exp_avg", "exp_avg_sq")}
    full["params"] = {n: v.to(torch.bfloat16) for n, v in full["params"].items()}
    hyper = dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3)
    scaler = dict(scale=1024.0, growth_step=7, hysteresis_step=1)
    owners = C.zero_rank_names({n: tuple(shapes[n]) for n in order}, zw)
    assert sum(1 for o in owners if not o) == 3
    out = str(tmp_path / "ck")
    for r in range(zw):
        own = lambda d: {n: d[n] for n in owners[r]}  # noqa: E731
        C.save_checkpoint(out, cfg.model, full["params"] if r == 0 else None, own(full["master"]), own(full["exp_avg"]), own(full["exp_avg_sq"]), 5, scaler, 3e-4, hyper,
                          zero_world=zw, zero_ranks=[r], write_model=(r == 0), shapes={n: tuple(shapes[n]) for n in order})
    empty = next(r for r in range(zw) if not owners[r])
    st = C._load(os.path.join(out, f"optimizer_tp0_pp0_zo{empty}.pt"))
    assert st["base_optim_states"]["state"] == {} and st["flat_fp32_weights"] == {} and st["zero_devide_optim_plan"][0][empty] == []
    assert [g_["params"] for g_ in st["base_optim_states"]["param_groups"]] == [list(range(len(order))), []]
    ck = C.load_checkpoint(out, cfg.model)
    assert (ck["zero_world"], ck["adam_step"], ck["lr"], ck["scaler"]) == (zw, 5, 3e-4, scaler)
    for k in ("params", "master", "exp_avg", "exp_avg_sq"):
        for n in order:
            assert torch.equal(ck[k][n], full[k][n]), (k, n)


@pytest.mark.gpu
def test_engine_resumes_from_reference_checkpoint_and_round_trips(dev, tmp_path):
    """The HIP engine loads the reference's checkpoint and reproduces the 2 steps the reference trained after saving; a checkpoint
    it writes itself restores a second engine to a bit-identical continuation, and its files equal the reference's in structure."""
    from internevo_amd import checkpoint as C
    from internevo_amd.data import SyntheticLoader
    from internevo_amd.engine import InternLM2Engine

    gold = json.load(open(os.path.join(G, "ckpt.json")))
    cfg = _cfg()
    c = gold["config"]
    eng = InternLM2Engine(cfg, dev)
    eng.load_checkpoint(REF)
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"]))
    for _ in range(gold["saved_after_step"]):
        next(loader)
    batches = [next(loader) for _ in range(2)]
    out = str(tmp_path / "ck")
    eng.save_checkpoint(out)  # before training on: must equal what was loaded
    a, b = C.load_checkpoint(REF, cfg.model), C.load_checkpoint(out, cfg.model)
    for key in ("params", "master", "exp_avg", "exp_avg_sq"):
        assert all(torch.equal(a[key][n], b[key][n]) for n in a[key]), key
    assert (a["adam_step"], a["scaler"], a["lr"]) == (b["adam_step"], b["scaler"], b["lr"])
    tr = []
    for (batch, labels), w in zip(batches, gold["steps"][gold["saved_after_step"]:]):
        loss = float(eng.forward_backward(batch, labels))
        eng.step()
        st = eng.read_state()
        print(f"resumed: HIP {loss:.5f} / {st.grad_norm:.4f} | reference {w['loss']:.5f} / {w['grad_norm']['0_default']:.4f}")
        assert abs(loss - w["loss"]) <= 3e-3 * abs(w["loss"])
        assert abs(st.grad_norm - w["grad_norm"]["0_default"]) <= 2e-2 * w["grad_norm"]["0_default"]
        assert st.loss_scale == w["loss_scale"] and st.skip == 0
        tr.append((loss, st.grad_norm))
    eng2 = InternLM2Engine(cfg, dev)
    eng2.load_checkpoint(out)
    tr2 = []
    for batch, labels in batches:
        loss = float(eng2.forward_backward(batch, labels))
        eng2.step()
        tr2.append((loss, eng2.read_state().grad_norm))
    assert tr == tr2 and torch.equal(eng.params, eng2.params), "a restored engine continues bit-identically"


def test_stale_shards_of_a_larger_layout_are_removed_before_a_save(tmp_path):
    """Re-saving into a folder that still holds the shards of an earlier, larger layout: the loaders infer the layout from the highest
    file index present, so those files have to go before the new save is written (engine.save_checkpoint calls this on one rank)."""
    from internevo_amd import checkpoint as C

    names = ["model_tp0_pp0.pt", "model_tp1_pp0.pt", "topo_tp0_pp0.json", "topo_tp1_pp0.json"]
    names += [f"optimizer_tp{t}_pp0_zo{z}.pt" for t in range(2) for z in range(4)]
    names += [f"gpus-8_wp-0_tp-{t}_dp-{z}_pp-0_zo-{z}.pt" for t in range(2) for z in range(4)]
    names += ["context.pt", "sampler.pt", "schedulder.pt"]
    for n in names:
        (tmp_path / n).write_bytes(b"x")
    C.remove_stale_shards(str(tmp_path), zero_world=2, tp_world=1)
    left = sorted(os.listdir(tmp_path))
    assert left == sorted(["model_tp0_pp0.pt", "topo_tp0_pp0.json", "optimizer_tp0_pp0_zo0.pt", "optimizer_tp0_pp0_zo1.pt",
                           "context.pt", "sampler.pt", "schedulder.pt"]), left   # the gpus-8 plans name another world size: all gone
    assert C.saved_zero_world(str(tmp_path)) == 2 and C.saved_tp_world(str(tmp_path)) == 1


def _v1_model_cfg(gold):
    from internevo_amd.config import ModelConfig

    c = gold["config"]
    return ModelConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], num_layers=c["layers"], num_attention_heads=c["heads"], num_kv_attention_heads=c["heads"],
                       mlp_ratio=8 / 3, model_type="INTERNLM", num_experts=1)


def test_dense_internlm1_reference_checkpoint_loads_saves_and_resumes(tmp_path):
    """tests/golden/ckpt_ref_v1/: the REAL reference's checkpoint of the dense InternLM-1 model (model_type INTERNLM, its default; make_golden.py
    --ckpt-v1) after two training steps; ckpt_v1.json = the structure it saw and the two steps it trained afterwards.  The loader reads it (module
    order, ZeRO partition plan checked inside), the writer reproduces the model and optimizer files tensor for tensor, and the oracle resumed from
    it retraces the reference's next two steps."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import PathConfig, TrainConfig
    from internevo_amd.data import SyntheticLoader
    from oracle.moe_model import OracleMoETrainer

    gold = json.load(open(os.path.join(G, "ckpt_v1.json")))
    ref = os.path.join(G, "ckpt_ref_v1")
    mc = _v1_model_cfg(gold)
    ck = C.load_checkpoint(ref, mc)
    assert [["model." + n, str(ck["params"][n].dtype), list(ck["params"][n].shape)] for n in C.state_dict_order(mc)] == gold["model_keys"]
    assert ck["adam_step"] == 2 and ck["scaler"] == dict(scale=65536.0, growth_step=2, hysteresis_step=0)
    for n, p in ck["params"].items():
        assert torch.equal(ck["master"][n].to(torch.bfloat16), p), n
    out = str(tmp_path / "ck")
    C.save_checkpoint(out, mc, ck["params"], ck["master"], ck["exp_avg"], ck["exp_avg_sq"], ck["adam_step"], ck["scaler"], ck["lr"],
                      dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3))
    ld = lambda folder, fn: torch.load(os.path.join(folder, fn), weights_only=False)  # noqa: E731
    a, b = ld(ref, "model_tp0_pp0.pt"), ld(out, "model_tp0_pp0.pt")
    assert list(a) == list(b) and all(a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]) for k in a)
    oa, ob = C._load(os.path.join(ref, "optimizer_tp0_pp0_zo0.pt")), C._load(os.path.join(out, "optimizer_tp0_pp0_zo0.pt"))
    assert torch.equal(oa["flat_fp32_weights"][0].detach(), ob["flat_fp32_weights"][0]) and oa["zero_devide_optim_plan"] == ob["zero_devide_optim_plan"]
    for k in ("exp_avg", "exp_avg_sq"):
        assert torch.equal(oa["base_optim_states"]["state"][0][k], ob["base_optim_states"]["state"][0][k])
    # the oracle resumes where the reference went on
    c = gold["config"]
    tc = TrainConfig(seq_len=c["seq_len"], micro_bsz=1, micro_num=c["micro_num"], total_steps=c["total_steps"], lr=1e-3, fixed_random_dataset_seqlen=True)
    tr = OracleMoETrainer(PathConfig(mc, tc), torch.bfloat16)
    tr.load_state(ck)
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"]))
    for _ in range(gold["saved_after_step"]):
        next(loader)
    for w in gold["steps"][gold["saved_after_step"]:]:
        r = tr.train_step(*next(loader))
        total = sum(v * v for v in r["grad_norm"].values()) ** 0.5
        assert abs(r["loss"] - w["loss"]) <= 2e-3 * w["loss"] and abs(total - w["grad_norm"]["0_default"]) <= 1e-2 * total, (r, w)
        assert abs(r["lr"] - w["lr"]) <= 1e-12


def test_dense_internlm1_tensor_parallel_checkpoint_of_the_reference_merges_is_reproduced_and_resumes(tmp_path):
    """tests/golden/ckpt_ref_v1tp2/: the REAL reference's dense InternLM-1 model on two Megatron tensor-parallel ranks (make_golden.py --ckpt-v1tp).  Two things
    differ from the InternLM2 layout (ckpt_ref_tp2/): a rank's packed Wqkv rows (weight and bias) are "(three h/tp d)" of ITS heads, so the ranks' parts
    interleave along the head axis of the single-rank "(three h d)" instead of concatenating; and out_proj's bias (a row-parallel linear) exists on tensor rank 0
    only -- rank 1's files hold 19 parameters, rank 0's 21.  The loader merges both ranks' files into the full tensors, the writer reproduces each rank's files
    tensor for tensor from its cut, and the single-rank oracle resumed from the merge retraces the reference's next two steps."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import PathConfig, TrainConfig
    from internevo_amd.data import SyntheticLoader
    from oracle.moe_model import OracleMoETrainer, param_shapes

    g0, g1 = (json.load(open(os.path.join(G, f"ckpt_v1tp2_rank{r}.json"))) for r in (0, 1))
    ref = os.path.join(G, "ckpt_ref_v1tp2")
    mc = _v1_model_cfg(g0)
    hd = mc.head_dim
    assert ["model." + n for n in C.state_dict_order(mc, 0)] == [k[0] for k in g0["model_keys"]] and ["model." + n for n in C.state_dict_order(mc, 1)] == [k[0] for k in g1["model_keys"]]
    assert len(g0["model_keys"]) == len(g1["model_keys"]) + mc.num_layers   # (out_proj.bias)
    ck = C.load_checkpoint(ref, mc)
    assert ck["tp_world"] == 2 and ck["adam_step"] == 2 and list(ck["params"]) == C.state_dict_order(mc)
    full = param_shapes(mc)
    for n in ck["params"]:
        assert tuple(ck["params"][n].shape) == tuple(full[n]) == tuple(ck["master"][n].shape) == tuple(ck["exp_avg_sq"][n].shape), n
        assert torch.equal(ck["master"][n].to(torch.bfloat16), ck["params"][n]), n
    # the head-wise cut and its inverse
    w = ck["params"]["blocks.0.mixer.Wqkv.weight"]
    parts = [C.tp_shard("blocks.0.mixer.Wqkv.weight", w, t, 2, hd) for t in (0, 1)]
    assert torch.equal(C.tp_unshard("blocks.0.mixer.Wqkv.weight", parts, hd), w) and not torch.equal(torch.cat(parts), w)
    assert torch.equal(parts[1], torch.load(os.path.join(ref, "model_tp1_pp0.pt"), weights_only=False)["model.blocks.0.mixer.Wqkv.weight"])
    hyper = dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3)
    out = str(tmp_path / "ck")
    for t in (0, 1):
        cut = lambda d: {n: C.tp_shard(n, d[n], t, 2, hd).contiguous() for n in C.state_dict_order(mc, t)}  # noqa: E731
        C.save_checkpoint(out, mc, cut(ck["params"]), cut(ck["master"]), cut(ck["exp_avg"]), cut(ck["exp_avg_sq"]), ck["adam_step"], ck["scaler"], ck["lr"], hyper,
                          tp_world=2, tp_rank=t)
    assert sorted(f for f in os.listdir(out) if not f.endswith(".step")) == g0["files"]
    for t in (0, 1):
        _cmp_optimizer_files(C, os.path.join(ref, f"optimizer_tp{t}_pp0_zo0.pt"), os.path.join(out, f"optimizer_tp{t}_pp0_zo0.pt"))
        x, y = (torch.load(os.path.join(f, f"model_tp{t}_pp0.pt"), weights_only=False) for f in (ref, out))
        assert list(x) == list(y) and all(x[k].shape == y[k].shape and torch.equal(x[k], y[k]) for k in x)
        fn = f"gpus-2_wp-0_tp-{t}_dp-0_pp-0_zo-0.pt"
        assert C._load(os.path.join(ref, fn)) == C._load(os.path.join(out, fn))
    c = g0["config"]
    tc = TrainConfig(seq_len=c["seq_len"], micro_bsz=1, micro_num=c["micro_num"], total_steps=c["total_steps"], lr=1e-3, fixed_random_dataset_seqlen=True)
    tr = OracleMoETrainer(PathConfig(mc, tc), torch.bfloat16)
    tr.load_state(ck)
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, g0["num_samples"]))
    for _ in range(g0["saved_after_step"]):
        next(loader)
    for w_ in g0["steps"][g0["saved_after_step"]:]:
        r = tr.train_step(*next(loader))
        total = sum(v * v for v in r["grad_norm"].values()) ** 0.5
        assert abs(r["loss"] - w_["loss"]) <= 2e-3 * w_["loss"] and abs(total - w_["grad_norm"]["0_default"]) <= 1e-2 * total, (r, w_)


def test_pipeline_stage_checkpoint_files_of_the_reference_load_save_and_resume(tmp_path):
    """tests/golden/ckpt_ref_pp2/: the REAL reference with parallel.pipeline = dict(size=2) on two processes (make_golden.py --ckpt-pp) wrote one model /
    optimizer / plan / topo file per STAGE after two steps; a stage numbers its layers from 0 (model_tp0_pp1.pt holds layers.0, layers.1 = the model's
    layers 2, 3, then norm and output).  The loader merges the stages into the whole model under its own names (any layout can resume from the folder),
    the writer reproduces each stage's files tensor for tensor from the stage's slice, and the single-rank oracle resumed from the merge retraces the
    reference's next two steps (ckpt_pp2_rank1.json: the last stage reports the loss)."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import tiny
    from internevo_amd.data import SyntheticLoader
    from oracle.step import OracleTrainer

    ref = os.path.join(G, "ckpt_ref_pp2")
    g0, g1 = (json.load(open(os.path.join(G, f"ckpt_pp2_rank{r}.json"))) for r in (0, 1))
    c = g1["config"]
    cfg = tiny(c["hidden"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], c["seq_len"], c["micro_num"], 1e-3, c["total_steps"])
    mc = cfg.model
    assert C.saved_pp_world(ref) == 2
    assert [["model." + n, "torch.bfloat16", None] for n in C.stage_order(mc, 2, True, False)] == [[k[0], k[1], None] for k in g0["model_keys"]]
    assert ["model." + n for n in C.stage_order(mc, 2, False, True)] == [k[0] for k in g1["model_keys"]]
    ck = C.load_checkpoint(ref, mc)
    assert list(ck["params"]) == C.state_dict_order(mc) and ck["pp_world"] == 2 and ck["adam_step"] == 2
    assert set(ck["master"]) == set(ck["params"]) == set(ck["exp_avg"])
    # `want` (what one engine rank keeps of the optimizer state: names of the whole model, any stage) is applied while the flat vectors are cut
    some = {"layers.3.attention.wo.weight", "layers.0.ffn_norm.weight", "norm.weight"}
    part = C.load_checkpoint(ref, mc, want=some)
    assert set(part["master"]) == set(part["exp_avg_sq"]) == some and list(part["params"]) == C.state_dict_order(mc)
    assert all(torch.equal(part[k][n], ck[k][n]) for k in ("master", "exp_avg", "exp_avg_sq") for n in some)
    # every stage's files again from its slice of the merged state
    out = str(tmp_path / "ck")
    for p_, (lo, n) in enumerate([(0, 2), (2, 2)]):
        order = C.stage_order(mc, n, p_ == 0, p_ == 1)
        cut = lambda d: {k: d[C.stage_to_global(k, lo)] for k in order}  # noqa: E731
        C.save_checkpoint(out, mc, cut(ck["params"]), cut(ck["master"]), cut(ck["exp_avg"]), cut(ck["exp_avg_sq"]), ck["adam_step"], ck["scaler"], ck["lr"],
                          dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3), pp_world=2, pp_rank=p_, order=order)
    assert sorted(os.listdir(out)) == g1["files"]
    ld = lambda folder, fn: torch.load(os.path.join(folder, fn), weights_only=False)  # noqa: E731
    for p_ in (0, 1):
        a, b = ld(ref, f"model_tp0_pp{p_}.pt"), ld(out, f"model_tp0_pp{p_}.pt")
        assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
        oa, ob = C._load(os.path.join(ref, f"optimizer_tp0_pp{p_}_zo0.pt")), C._load(os.path.join(out, f"optimizer_tp0_pp{p_}_zo0.pt"))
        assert torch.equal(oa["flat_fp32_weights"][0].detach(), ob["flat_fp32_weights"][0]) and oa["zero_devide_optim_plan"] == ob["zero_devide_optim_plan"]
        assert torch.equal(oa["base_optim_states"]["state"][0]["exp_avg_sq"], ob["base_optim_states"]["state"][0]["exp_avg_sq"])
        assert C._load(os.path.join(ref, f"gpus-2_wp-0_tp-0_dp-0_pp-{p_}_zo-0.pt")) == C._load(os.path.join(out, f"gpus-2_wp-0_tp-0_dp-0_pp-{p_}_zo-0.pt"))
    # the single-rank oracle resumes where the two-stage reference went on
    tr = OracleTrainer(cfg, torch.bfloat16)
    tr.load_state(ck)
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, g1["num_samples"]))
    for _ in range(g1["saved_after_step"]):
        next(loader)
    for w in g1["steps"][g1["saved_after_step"]:]:
        r = tr.train_step(*next(loader))
        assert abs(r["loss"] - w["loss"]) <= 2e-3 * w["loss"] and abs(r["grad_norm"] - w["grad_norm"]["0_default"]) <= 1e-2 * r["grad_norm"], (r, w)


def test_pipeline_x_tensor_parallel_checkpoint_of_the_reference_merges_is_reproduced_and_resumes(tmp_path):
    """tests/golden/ckpt_ref_pp2tp2/: the REAL reference with parallel.pipeline size 2 AND parallel.tensor size 2 (mtp) on FOUR processes (make_golden.py
    --ckpt-pptp) wrote one model / optimizer / plan / topo file per (tensor rank, stage) after two steps.  The loader merges the tensor ranks of a stage into full
    tensors and the stages into the whole model (any layout resumes from the folder), the writer reproduces all sixteen files tensor for tensor from the (stage,
    tensor rank) cut of the merged state, and the single-rank oracle resumed from the merge retraces the reference's next two steps."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import tiny
    from internevo_amd.data import SyntheticLoader
    from oracle.model import param_shapes
    from oracle.step import OracleTrainer

    ref = os.path.join(G, "ckpt_ref_pp2tp2")
    gold = [json.load(open(os.path.join(G, f"ckpt_pp2tp2_rank{r}.json"))) for r in range(4)]
    assert [(g["ranks"]["TENSOR"][0], g["ranks"]["PIPELINE"][0]) for g in gold] == [(0, 0), (1, 0), (0, 1), (1, 1)]   # tensor ranks innermost, stages = blocks of ranks
    last = gold[2]   # (a last-stage rank reports the loss)
    c = last["config"]
    cfg = tiny(c["hidden"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], c["seq_len"], c["micro_num"], 1e-3, c["total_steps"])
    mc = cfg.model
    assert C.saved_pp_world(ref) == 2 and C.saved_tp_world(ref) == 2 and C.saved_zero_world(ref, 1, 1) == 1
    ck = C.load_checkpoint(ref, mc)
    assert list(ck["params"]) == C.state_dict_order(mc) and ck["pp_world"] == 2 and ck["tp_world"] == 2 and ck["adam_step"] == 2
    full = param_shapes(mc)
    for n in C.state_dict_order(mc):
        assert tuple(ck["params"][n].shape) == tuple(full[n]) == tuple(ck["exp_avg"][n].shape), n
        assert torch.equal(ck["master"][n].to(torch.bfloat16), ck["params"][n]), n
    some = {"layers.3.attention.wo.weight", "layers.0.ffn_norm.weight", "output.weight"}
    part = C.load_checkpoint(ref, mc, want=some)
    assert set(part["master"]) == some and all(torch.equal(part[k][n], ck[k][n]) for k in ("master", "exp_avg", "exp_avg_sq") for n in some)
    only = C.load_checkpoint(ref, mc, model_only=True)
    assert only["master"] is None and all(torch.equal(only["params"][n], ck["params"][n]) for n in ck["params"])
    # every (stage, tensor rank)'s files again from its cut of the merged state
    out = str(tmp_path / "ck")
    hyper = dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3)
    for p_, (lo, n) in enumerate([(0, 2), (2, 2)]):
        order = C.stage_order(mc, n, p_ == 0, p_ == 1)
        for t in (0, 1):
            cut = lambda d: {k: C.tp_shard(k, d[C.stage_to_global(k, lo)], t, 2).contiguous() for k in order}  # noqa: E731
            C.save_checkpoint(out, mc, cut(ck["params"]), cut(ck["master"]), cut(ck["exp_avg"]), cut(ck["exp_avg_sq"]), ck["adam_step"], ck["scaler"], ck["lr"],
                              hyper, tp_world=2, tp_rank=t, pp_world=2, pp_rank=p_, order=order)
    assert sorted(os.listdir(out)) == last["files"]
    ld = lambda folder, fn: torch.load(os.path.join(folder, fn), weights_only=False)  # noqa: E731
    for p_ in (0, 1):
        for t in (0, 1):
            a, b = ld(ref, f"model_tp{t}_pp{p_}.pt"), ld(out, f"model_tp{t}_pp{p_}.pt")
            assert list(a) == list(b) == [k[0] for k in gold[2 * p_ + t]["model_keys"]] and all(torch.equal(a[k], b[k]) for k in a)
            _cmp_optimizer_files(C, os.path.join(ref, f"optimizer_tp{t}_pp{p_}_zo0.pt"), os.path.join(out, f"optimizer_tp{t}_pp{p_}_zo0.pt"))
            fn = f"gpus-4_wp-0_tp-{t}_dp-0_pp-{p_}_zo-0.pt"
            assert C._load(os.path.join(ref, fn)) == C._load(os.path.join(out, fn))
    # the single-rank oracle resumes where the four-process reference went on
    tr = OracleTrainer(cfg, torch.bfloat16)
    tr.load_state(ck)
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, last["num_samples"]))
    for _ in range(last["saved_after_step"]):
        next(loader)
    for w in last["steps"][last["saved_after_step"]:]:
        r = tr.train_step(*next(loader))
        assert abs(r["loss"] - w["loss"]) <= 2e-3 * w["loss"] and abs(r["grad_norm"] - w["grad_norm"]["0_default"]) <= 1e-2 * r["grad_norm"], (r, w)


def test_interleaved_pipeline_checkpoint_of_the_reference_merges_is_reproduced_and_resumes(tmp_path):
    """tests/golden/ckpt_ref_pp2i/: the REAL reference under the INTERLEAVED pipeline schedule (parallel.pipeline size 2, model.num_chunks = 2; make_golden.py
    --ckpt-ppi).  A stage's model is a ModuleList of chunks: its state dict carries "<chunk>.model.<name>", every chunk numbers its layers from 0, and stage p holds
    the layers partition_chunks gives it (stage 0: model layers 0 and 2, stage 1: layers 1 and 3, with norm + head in ITS last chunk); one optimizer group per stage
    over all its chunks.  The loader merges the stages into the whole model in the model's own order, the writer reproduces every stage's files tensor for tensor, and
    the single-rank oracle resumed from the merge retraces the reference's next two steps."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import tiny
    from internevo_amd.data import SyntheticLoader
    from oracle.step import OracleTrainer

    ref = os.path.join(G, "ckpt_ref_pp2i")
    g0, g1 = (json.load(open(os.path.join(G, f"ckpt_pp2i_rank{r}.json"))) for r in (0, 1))
    c = g1["config"]
    cfg = tiny(c["hidden"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], c["seq_len"], c["micro_num"], 1e-3, c["total_steps"])
    mc = cfg.model
    namings = [C.stage_naming(mc, 2, p_, 2) for p_ in (0, 1)]
    assert [k for _, k, _ in namings[0]] == [k[0] for k in g0["model_keys"]] and [k for _, k, _ in namings[1]] == [k[0] for k in g1["model_keys"]]
    assert [n for n, _, _ in namings[1]] == g1["param_group_order"]["0"]   # (the names the stage's optimizer group is flattened in)
    assert [g for _, _, g in namings[0]][:2] == ["tok_embeddings.weight", "layers.0.attention.wqkv.weight"] and [g for _, _, g in namings[0]][8] == "layers.2.attention.wqkv.weight"
    ck = C.load_checkpoint(ref, mc)
    assert list(ck["params"]) == C.state_dict_order(mc) and ck["pp_world"] == 2 and ck["chunks"] == 2 and ck["adam_step"] == 2
    assert all(torch.equal(ck["master"][n].to(torch.bfloat16), ck["params"][n]) for n in ck["params"])
    some = {"layers.2.attention.wo.weight", "layers.1.ffn_norm.weight", "norm.weight"}
    part = C.load_checkpoint(ref, mc, want=some)
    assert set(part["master"]) == some and all(torch.equal(part[k][n], ck[k][n]) for k in ("master", "exp_avg", "exp_avg_sq") for n in some)
    out = str(tmp_path / "ck")
    for p_ in (0, 1):
        order = [n for n, _, _ in namings[p_]]
        cut = lambda d: {n: d[g] for n, _, g in namings[p_]}  # noqa: E731
        C.save_checkpoint(out, mc, cut(ck["params"]), cut(ck["master"]), cut(ck["exp_avg"]), cut(ck["exp_avg_sq"]), ck["adam_step"], ck["scaler"], ck["lr"],
                          dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3), pp_world=2, pp_rank=p_, order=order, chunked=True)
    assert sorted(os.listdir(out)) == g1["files"]
    ld = lambda folder, fn: torch.load(os.path.join(folder, fn), weights_only=False)  # noqa: E731
    for p_ in (0, 1):
        a, b = ld(ref, f"model_tp0_pp{p_}.pt"), ld(out, f"model_tp0_pp{p_}.pt")
        assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
        _cmp_optimizer_files(C, os.path.join(ref, f"optimizer_tp0_pp{p_}_zo0.pt"), os.path.join(out, f"optimizer_tp0_pp{p_}_zo0.pt"))
        assert C._load(os.path.join(ref, f"gpus-2_wp-0_tp-0_dp-0_pp-{p_}_zo-0.pt")) == C._load(os.path.join(out, f"gpus-2_wp-0_tp-0_dp-0_pp-{p_}_zo-0.pt"))
    tr = OracleTrainer(cfg, torch.bfloat16)
    tr.load_state(ck)
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, g1["num_samples"]))
    for _ in range(g1["saved_after_step"]):
        next(loader)
    for w in g1["steps"][g1["saved_after_step"]:]:
        r = tr.train_step(*next(loader))
        assert abs(r["loss"] - w["loss"]) <= 2e-3 * w["loss"] and abs(r["grad_norm"] - w["grad_norm"]["0_default"]) <= 1e-2 * r["grad_norm"], (r, w)


def test_dense_internlm1_pipeline_checkpoint_of_the_reference_merges_is_reproduced_and_resumes(tmp_path):
    """tests/golden/ckpt_ref_pp2v1/: the REAL reference's dense InternLM-1 model on two pipeline stages (make_golden.py --ckpt-ppv1): a stage's files carry
    `blocks.<k>.` numbered from 0 (embedding on the first stage, norm + head on the last), biases with their block.  Merge, tensor-for-tensor rewrite of both stages'
    files, and the single-rank oracle resumed from the merge retraces the reference's next two steps."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import PathConfig, TrainConfig
    from internevo_amd.data import SyntheticLoader
    from oracle.moe_model import OracleMoETrainer

    ref = os.path.join(G, "ckpt_ref_pp2v1")
    g0, g1 = (json.load(open(os.path.join(G, f"ckpt_pp2v1_rank{r}.json"))) for r in (0, 1))
    mc = _v1_model_cfg(g1)
    namings = [C.stage_naming(mc, 2, p_) for p_ in (0, 1)]
    assert [k for _, k, _ in namings[0]] == [k[0] for k in g0["model_keys"]] and [k for _, k, _ in namings[1]] == [k[0] for k in g1["model_keys"]]
    assert namings[1][0][0] == "blocks.0.mixer.Wqkv.weight" and namings[1][0][2] == "blocks.2.mixer.Wqkv.weight"
    ck = C.load_checkpoint(ref, mc)
    assert list(ck["params"]) == C.state_dict_order(mc) and ck["pp_world"] == 2 and ck["adam_step"] == 2
    assert all(torch.equal(ck["master"][n].to(torch.bfloat16), ck["params"][n]) for n in ck["params"])
    out = str(tmp_path / "ck")
    for p_ in (0, 1):
        cut = lambda d: {n: d[g] for n, _, g in namings[p_]}  # noqa: E731
        C.save_checkpoint(out, mc, cut(ck["params"]), cut(ck["master"]), cut(ck["exp_avg"]), cut(ck["exp_avg_sq"]), ck["adam_step"], ck["scaler"], ck["lr"],
                          dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3), pp_world=2, pp_rank=p_, order=[n for n, _, _ in namings[p_]])
    assert sorted(os.listdir(out)) == g1["files"]
    for p_ in (0, 1):
        a, b = (torch.load(os.path.join(f, f"model_tp0_pp{p_}.pt"), weights_only=False) for f in (ref, out))
        assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
        _cmp_optimizer_files(C, os.path.join(ref, f"optimizer_tp0_pp{p_}_zo0.pt"), os.path.join(out, f"optimizer_tp0_pp{p_}_zo0.pt"))
    c = g1["config"]
    tc = TrainConfig(seq_len=c["seq_len"], micro_bsz=1, micro_num=c["micro_num"], total_steps=c["total_steps"], lr=1e-3, fixed_random_dataset_seqlen=True)
    tr = OracleMoETrainer(PathConfig(mc, tc), torch.bfloat16)
    tr.load_state(ck)
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, g1["num_samples"]))
    for _ in range(g1["saved_after_step"]):
        next(loader)
    for w in g1["steps"][g1["saved_after_step"]:]:
        r = tr.train_step(*next(loader))
        total = sum(v * v for v in r["grad_norm"].values()) ** 0.5
        assert abs(r["loss"] - w["loss"]) <= 2e-3 * w["loss"] and abs(total - w["grad_norm"]["0_default"]) <= 1e-2 * total, (r, w)


def test_llama2_tensor_parallel_checkpoint_of_the_reference_merges_is_reproduced_and_resumes(tmp_path):
    """tests/golden/ckpt_ref_llama_tp2/: the REAL reference's LLAMA2 model (separate wq / wk / wv) on two tensor ranks -- BASELINE configs[2]'s family and layout
    (make_golden.py --ckpt-llama-tp).  The loader merges the ranks' files (every projection by rows / columns as checkpoint.tp_split_dim says), the writer reproduces
    them tensor for tensor, and the single-rank oracle (LLAMA2: adapt_hf False) resumed from the merge retraces the reference's next two steps."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import tiny
    from internevo_amd.data import SyntheticLoader
    from oracle.model import param_shapes
    from oracle.step import OracleTrainer

    gold = json.load(open(os.path.join(G, "ckpt_llama_tp2.json")))
    ref = os.path.join(G, "ckpt_ref_llama_tp2")
    c = gold["config"]
    cfg = tiny(c["hidden"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], c["seq_len"], c["micro_num"], 1e-3, c["total_steps"], model_type="LLAMA2")
    mc = cfg.model
    assert ["model." + n for n in C.state_dict_order(mc)] == [k[0] for k in gold["model_keys"]] and any(n.endswith("attention.wq.weight") for n in C.state_dict_order(mc))
    ck = C.load_checkpoint(ref, mc)
    full = param_shapes(mc)
    assert ck["tp_world"] == 2 and ck["adam_step"] == 2 and all(tuple(ck["params"][n].shape) == tuple(full[n]) for n in C.state_dict_order(mc))
    assert all(torch.equal(ck["master"][n].to(torch.bfloat16), ck["params"][n]) for n in ck["params"])
    hyper = dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3)
    out = str(tmp_path / "ck")
    for t in (0, 1):
        cut = lambda d: {n: C.tp_shard(n, v, t, 2).contiguous() for n, v in d.items()}  # noqa: E731
        C.save_checkpoint(out, mc, cut(ck["params"]), cut(ck["master"]), cut(ck["exp_avg"]), cut(ck["exp_avg_sq"]), ck["adam_step"], ck["scaler"], ck["lr"], hyper,
                          tp_world=2, tp_rank=t)
    assert sorted(f for f in os.listdir(out) if not f.endswith(".step")) == gold["files"]
    for t in (0, 1):
        _cmp_optimizer_files(C, os.path.join(ref, f"optimizer_tp{t}_pp0_zo0.pt"), os.path.join(out, f"optimizer_tp{t}_pp0_zo0.pt"))
        x, y = (torch.load(os.path.join(f, f"model_tp{t}_pp0.pt"), weights_only=False) for f in (ref, out))
        assert list(x) == list(y) and all(x[k].shape == y[k].shape and torch.equal(x[k], y[k]) for k in x)
    tr = OracleTrainer(cfg, torch.bfloat16)
    tr.load_state(ck)
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"]))
    for _ in range(gold["saved_after_step"]):
        next(loader)
    for w in gold["steps"][gold["saved_after_step"]:]:
        r = tr.train_step(*next(loader))
        assert abs(r["loss"] - w["loss"]) <= 2e-3 * w["loss"] and abs(r["grad_norm"] - w["grad_norm"]["0_default"]) <= 1e-2 * r["grad_norm"], (r, w)


def test_hybrid_zero_reference_checkpoint_loads_and_resumes(tmp_path):
    """tests/golden/ckpt_ref_dp4_zo2/: the REAL reference on four data-parallel ranks with parallel.zero1.size = 2 (hybrid ZeRO; make_golden.py --ckpt-hz): the optimizer
    state is sharded over groups of TWO ranks -- two optimizer files, written by the first zero group -- while every data rank writes its own plan file under the JOB's
    world size (`gpus-4_..._dp-{d}_..._zo-{d % 2}.pt`, hybrid_zero_optim.py:133-140, components.py:396-407).  The loader takes the layout from the optimizer files (the
    plan also sits inside them), merges the two shards, and the oracle resumed from the merge retraces the four-rank run's next two gradient norms.
    The writer reproduces the whole file set the way the engine's ranks call it: the two ranks of the first zero group write model / optimizer / plan files, the other
    two their plan files only (`plans_only`), all named after the job's four ranks."""
    from internevo_amd import checkpoint as C
    from internevo_amd.data import SyntheticLoader
    from oracle.step import OracleTrainer

    ref = os.path.join(G, "ckpt_ref_dp4_zo2")
    gold = [json.load(open(os.path.join(G, f"ckpt_dp4_zo2_rank{r}.json"))) for r in range(4)]
    assert [g["ranks"]["ZERO1"] for g in gold] == [[0, 2], [1, 2], [0, 2], [1, 2]] and [g["ranks"]["DATA"][0] for g in gold] == [0, 1, 2, 3]
    assert [g["rank_unique_id"] for g in gold] == [f"gpus-4_wp-0_tp-0_dp-{d}_pp-0_zo-{d % 2}.pt" for d in range(4)]
    assert [f for f in gold[0]["files"] if f.startswith("optimizer")] == ["optimizer_tp0_pp0_zo0.pt", "optimizer_tp0_pp0_zo1.pt"]
    cfg = _cfg()
    c = gold[0]["config"]
    assert C.saved_zero_world(ref) == 2
    ck = C.load_checkpoint(ref, cfg.model)
    assert ck["zero_world"] == 2 and ck["adam_step"] == 2 and all(torch.equal(ck["master"][n].to(torch.bfloat16), ck["params"][n]) for n in ck["params"])
    shapes = {n: tuple(ck["params"][n].shape) for n in C.state_dict_order(cfg.model)}
    flat = C.zero_flat_order(list(shapes.items()))
    plan = [C._plan_ids(flat, idx) for idx in C.zero_partition(flat, 2)]
    assert all(C._load(os.path.join(ref, g["rank_unique_id"]))[0] == plan for g in gold)   # every rank's plan file = the partition over the ZERO group
    out = str(tmp_path / "ck")
    hyper = dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3)
    for d in range(4):   # data-parallel rank d = zero rank d % 2 of zero group d // 2
        if d < 2:
            C.save_checkpoint(out, cfg.model, ck["params"] if d == 0 else None, ck["master"], ck["exp_avg"], ck["exp_avg_sq"], ck["adam_step"], ck["scaler"], ck["lr"], hyper,
                              zero_world=2, zero_ranks=[d], write_model=(d == 0), shapes=shapes, job_world=4, dp_ranks=[d])
        else:
            C.save_checkpoint(out, cfg.model, None, None, None, None, ck["adam_step"], ck["scaler"], ck["lr"], hyper, zero_world=2, shapes=shapes, job_world=4, dp_ranks=[d],
                              plans_only=True)
    assert sorted(os.listdir(out)) == gold[0]["files"]
    for g in gold:
        assert C._load(os.path.join(ref, g["rank_unique_id"])) == C._load(os.path.join(out, g["rank_unique_id"]))
    for z in (0, 1):
        _cmp_optimizer_files(C, os.path.join(ref, f"optimizer_tp0_pp0_zo{z}.pt"), os.path.join(out, f"optimizer_tp0_pp0_zo{z}.pt"))
    assert C.remove_stale_shards(out, 2, 1, job_world=4) == [] and len(C.remove_stale_shards(out, 2, 1)) == 4   # (the plan files name the job's world size)
    cfg.train.micro_num = 4 * c["micro_num"]
    tr = OracleTrainer(cfg, torch.bfloat16)
    tr.load_state(ck)
    loaders = [iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold[0]["num_samples"], data_rank=r, data_world_size=4)) for r in range(4)]
    for _ in range(gold[0]["saved_after_step"]):
        [next(l) for l in loaders]
    for w in gold[0]["steps"][gold[0]["saved_after_step"]:]:
        parts = [next(l) for l in loaders]
        batch = {k: (sum((p[0][k] for p in parts), []) if isinstance(parts[0][0][k], list) else torch.cat([p[0][k] for p in parts])) for k in parts[0][0]}
        g = tr.train_step(batch, torch.cat([p[1] for p in parts]))
        assert abs(g["grad_norm"] - w["grad_norm"]["0_default"]) <= 1e-2 * w["grad_norm"]["0_default"], (g["grad_norm"], w["grad_norm"])
        assert abs(g["lr"] - w["lr"]) <= 1e-9 * w["lr"] and g["loss_scale"] == w["loss_scale"]


def test_moe_reference_checkpoint_loads_saves_and_resumes(tmp_path):
    """tests/golden/ckpt_ref_moe/: the REAL reference's INTERNLM_MoE checkpoint (4 experts, top-2; make_golden.py --ckpt-moe) after two steps: the model
    file without the experts, one `model_moe_layer{l}_expert{e}_tp0.pt` per expert, and an optimizer file with THREE groups (default / fp32 = the gates /
    moe_ep_size_1 = the experts, optimizer_mode EXPERT_DATA).  checkpoint.load_moe_checkpoint reads it (module order, dtypes, the per-group plans
    checked), save_moe_checkpoint reproduces every file tensor for tensor and key for key, and the oracle resumed from it -- gating noise continued at
    the eighth call -- retraces the reference's next two steps."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import ModelConfig, PathConfig, TrainConfig
    from internevo_amd.data import SyntheticLoader
    from oracle.moe_model import OracleMoETrainer

    gold = json.load(open(os.path.join(G, "ckpt_moe.json")))
    ref, c = os.path.join(G, "ckpt_ref_moe"), gold["config"]
    mc = ModelConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], num_layers=c["layers"], num_attention_heads=c["heads"], num_kv_attention_heads=c["heads"],
                     mlp_ratio=4 / 3, model_type="INTERNLM_MoE", num_experts=c["num_experts"], moe_capacity_factor=c["capacity_factor"], moe_loss_coeff=0.1)
    ck = C.load_moe_checkpoint(ref, mc)
    assert [["model." + n, str(ck["params"][n].dtype), list(ck["params"][n].shape)] for n in C.state_dict_order(mc)] == gold["model_keys"]
    assert ck["adam_step"] == 2 and ck["scaler"] == dict(scale=65536.0, growth_step=2, hysteresis_step=0)
    assert [len(names) for _, names in C.moe_groups(mc)] == [len(v) for v in gold["param_group_order"].values()] == [15, 2, 24]
    out = str(tmp_path / "ck")
    C.save_moe_checkpoint(out, mc, ck["params"], ck["master"], ck["exp_avg"], ck["exp_avg_sq"], ck["adam_step"], ck["scaler"], ck["lr"],
                          dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3))
    assert sorted(os.listdir(out)) == [f for f in gold["files"] if f not in ("context.pt", "sampler.pt", "schedulder.pt")]
    ld = lambda folder, fn: torch.load(os.path.join(folder, fn), weights_only=False)  # noqa: E731
    for fn in os.listdir(out):
        if fn.startswith("model_"):
            a, b = ld(ref, fn), ld(out, fn)
            assert list(a) == list(b) and all(a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]) for k in a), fn
    oa, ob = C._load(os.path.join(ref, "optimizer_tp0_pp0_zo0.pt")), C._load(os.path.join(out, "optimizer_tp0_pp0_zo0.pt"))
    assert oa["zero_devide_optim_plan"] == ob["zero_devide_optim_plan"] == C._load(os.path.join(out, "gpus-1_wp-0_tp-0_dp-0_pp-0_zo-0.pt"))
    for g in (0, 1, 2):
        assert torch.equal(oa["flat_fp32_weights"][g].detach(), ob["flat_fp32_weights"][g])
        for k in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.equal(oa["base_optim_states"]["state"][g][k], ob["base_optim_states"]["state"][g][k]), (g, k)
        ga, gb = oa["base_optim_states"]["param_groups"][g], ob["base_optim_states"]["param_groups"][g]
        assert list(ga) == list(gb)
        assert all((str(ga[k]) == str(gb[k])) if k == "optimizer_mode" else (ga[k] == gb[k]) for k in ga), (ga, gb)
        assert ("expert_data" if g == 2 else "zero1") in str(gb["optimizer_mode"])
    # the oracle resumes where the reference went on
    tc = TrainConfig(seq_len=c["seq_len"], micro_bsz=1, micro_num=c["micro_num"], total_steps=c["total_steps"], lr=1e-3, fixed_random_dataset_seqlen=True)
    tr = OracleMoETrainer(PathConfig(mc, tc), torch.bfloat16)
    tr.load_state(ck)
    tr.calls = gold["saved_after_step"] * c["micro_num"] * c["layers"]   # gating calls so far (the harness seeds the k-th call with 5000 + k)
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"]))
    for _ in range(gold["saved_after_step"]):
        next(loader)
    for w in gold["steps"][gold["saved_after_step"]:]:
        r = tr.train_step(*next(loader))
        assert abs(r["loss"] - w["loss"]) <= 2e-3 * w["loss"] and abs(r["moe_loss"] - w["moe_loss"]) <= 3e-2 * w["moe_loss"], (r, w)
        for k, v in w["grad_norm"].items():
            assert abs(r["grad_norm"][k] - v) <= 3e-2 * v, (k, r["grad_norm"], w["grad_norm"])


def test_isp_layout_model_files_of_the_reference_load_and_are_reproduced(tmp_path):
    """tests/golden/ckpt_ref_isp2v1/ = the MODEL files a real two-process ISP run of the reference wrote (tensor = dict(size=2, mode="isp"), weight =
    dict(size=2), the dense InternLM-1 model -- configs/7B_isp_sft.py's layout in small; make_golden.py --ckpt-isp): `model_tp{t}_wp{w}_pp0.pt` with the
    embedding's columns / the head's rows of tensor rank t and the ISPLinear rows (weights AND biases) of weight rank w.  The reader merges them into the
    full model (the folder a user names in load_ckpt_info content=("model",) under that config); cutting the merged model again reproduces both files
    tensor for tensor; the merged weights are the ISP run's after two steps: the oracle (oracle.isp, which retraces that run) holds the same weights and
    sees the reference's step-2 loss on them."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import ModelConfig, PathConfig, TrainConfig
    from internevo_amd.data import SyntheticLoader
    from oracle.isp import OracleISPTrainer
    from oracle.moe_model import OracleMoETrainer, param_shapes

    gold = json.load(open(os.path.join(G, "ckpt_isp2v1.json")))
    c = gold["config"]
    mc = ModelConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], num_layers=c["layers"], num_attention_heads=c["heads"], num_kv_attention_heads=c["heads"],
                     mlp_ratio=8 / 3, model_type="INTERNLM", num_experts=1)
    ref = os.path.join(G, "ckpt_ref_isp2v1")
    assert sorted(C.saved_isp_layout(ref)) == [(0, 0), (1, 1)] and C.saved_tp_world(ref) == 0
    assert C.load_checkpoint(ref, mc)["master"] is None   # (this fixture holds model files only; optimizer shards of the layout: the next test)
    ck = C.load_checkpoint(ref, mc, model_only=True)
    assert {n: tuple(t.shape) for n, t in ck["params"].items()} == {n: tuple(s_) for n, s_ in param_shapes(mc).items()} and ck["master"] is None
    # what rank 0 of the reference held: the shapes recorded from its own state dict
    for key, dt, shape in gold["model_keys"]:
        n = key[len("model."):]
        assert list(C.isp_shard(n, ck["params"][n], 0, 2, 0, 2).shape) == shape, n
    for t, w in ((0, 0), (1, 1)):
        C.save_isp_model_shard(str(tmp_path), mc, ck["params"], t, 2, w, 2)
        ours = torch.load(os.path.join(tmp_path, f"model_tp{t}_wp{w}_pp0.pt"), weights_only=False)
        theirs = torch.load(os.path.join(ref, f"model_tp{t}_wp{w}_pp0.pt"), weights_only=False)
        assert list(ours) == list(theirs)
        for k in ours:
            assert ours[k].dtype == theirs[k].dtype and torch.equal(ours[k], theirs[k]), (t, w, k)
    assert sorted(os.listdir(tmp_path)) == gold["files"]
    # the weights are the run's: the oracle retraces the two steps before the save, and the loss of step 2 follows from the loaded weights alone
    tc = TrainConfig(seq_len=c["seq_len"], micro_bsz=1, micro_num=c["micro_num"], total_steps=c["total_steps"], lr=1e-3, fixed_random_dataset_seqlen=True)
    tr = OracleISPTrainer(OracleMoETrainer(PathConfig(mc, tc), torch.bfloat16), c["sp"])
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"]))
    for k in range(2):
        batch, labels = next(loader)
        r = tr.train_step(batch, labels)
        assert abs(r["loss"] - gold["steps"][k]["loss"]) <= 1e-3 * r["loss"]
    worst = max(float((tr.base.params[n].detach().float() - ck["params"][n].float()).abs().max()) for n in ck["params"])
    print("max |weight difference| oracle vs the reference's saved ISP shards after two steps:", worst)
    assert worst <= 1.6e-2   # (bf16 weights of magnitude <= 1.2: one or two ulps where an Adam sign decision differs)
    with torch.no_grad():
        for n in ck["params"]:
            tr.base.params[n].copy_(ck["params"][n])
    batch, labels = next(loader)
    M = batch["input_ids"].shape[0]
    from oracle.isp import isp_positions

    idx, cu = isp_positions(c["seq_len"], c["sp"], "INTERNLM")
    loss = sum(float(tr._loss(batch["input_ids"][i], labels[i], idx, cu).detach()) for i in range(M)) / M
    assert abs(loss - gold["steps"][2]["loss"]) <= 1e-3 * loss, (loss, gold["steps"][2]["loss"])


def _deep_equal(a, b, path=""):
    if torch.is_tensor(a):
        assert torch.is_tensor(b) and a.dtype == b.dtype and torch.equal(a, b), path
    elif isinstance(a, dict):
        assert list(a) == list(b), (path, list(a), list(b))
        for k in a:
            _deep_equal(a[k], b[k], f"{path}/{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _deep_equal(x, y, f"{path}[{i}]")
    else:
        assert a == b or repr(a) == repr(b), (path, a, b)


def test_isp_layout_optimizer_shards_of_the_reference_merge_are_reproduced_and_resume(tmp_path):
    """tests/golden/ckpt_ref_isp4v1/ = what a real FOUR-process ISP run of the reference wrote after two steps (tensor 2 (isp) x weight 2 -> two weight-data /
    data replicas; the dense InternLM-1 model; make_golden.py --ckpt-isp4): per rank `optimizer_tp{t}_wp{w}_pp0_dp{d}.pt` with THREE groups -- "default" (the
    rank's local ISPLinear row shards + norms, partitioned over the weight-data group), "embed_head" (its embedding columns / head rows, partitioned over the
    DATA group), "fp32" (empty) -- the reference's greedy whole-parameter partition of the LOCAL shapes, plus plan and model files.
      * the reader merges all shards into FULL fp32 state: the master weights round to the merged bf16 model bit for bit;
      * writing every rank's files from the merged state reproduces the reference's twelve files tensor for tensor (param_groups, plans, scaler included);
      * the rank coordinates the writer assumes are the reference's (ckpt_isp4v1_rank*.json records gpc's);
      * the oracle (oracle.isp over the union of the two data ranks' micro-batches) resumes from the merged state onto the reference's next two steps."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import ModelConfig, PathConfig, TrainConfig
    from internevo_amd.data import SyntheticLoader
    from oracle.isp import OracleISPTrainer
    from oracle.moe_model import OracleMoETrainer

    gold = [json.load(open(os.path.join(G, f"ckpt_isp4v1_rank{r}.json"))) for r in range(4)]
    c = gold[0]["config"]
    mc = ModelConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], num_layers=c["layers"], num_attention_heads=c["heads"], num_kv_attention_heads=c["heads"],
                     mlp_ratio=8 / 3, model_type="INTERNLM", num_experts=1)
    ref = os.path.join(G, "ckpt_ref_isp4v1")
    for r, g_ in enumerate(gold):
        co = C.isp_coords(r, 4, c["sp"], c["wp"])
        rk = g_["ranks"]
        assert (co["t"], co["w"], co["d"], co["z"]) == (rk["TENSOR"][0], rk["WEIGHT"][0], rk["DATA"][0], rk["ZERO1"][0]) and rk["WEIGHT_DATA"][0] == co["z"]
        assert (co["data_world"], co["zero_world"]) == (rk["DATA"][1], rk["ZERO1"][1])
        assert g_["rank_unique_id"] == f"gpus-4_wp-{co['w']}_tp-{co['t']}_dp-{co['d']}_pp-0_zo-{co['z']}.pt"
    ck = C.load_checkpoint(ref, mc)
    assert ck["adam_step"] == 2 and ck["isp"] == dict(world=4, sp=2, wp=2) and ck["scaler"]["scale"] == gold[0]["grad_scaler"]["_scale"]
    for n in ck["params"]:
        assert torch.equal(ck["master"][n].to(torch.bfloat16), ck["params"][n]), n
    # `want`: a rank asks for the optimizer tensors it holds -- only those are allocated and copied (coverage is still checked for every name)
    some = {"blocks.1.mixer.Wqkv.weight", "norm.weight", "head.weight"}
    part = C.load_checkpoint(ref, mc, want=some)
    assert set(part["master"]) == set(part["exp_avg"]) == set(part["exp_avg_sq"]) == some and set(part["params"]) == set(ck["params"])
    assert all(torch.equal(part[k][n], ck[k][n]) for k in ("master", "exp_avg", "exp_avg_sq") for n in some) and part["adam_step"] == 2
    hyper = dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3)
    for r in range(4):
        co = C.isp_coords(r, 4, 2, 2)
        # (a rank only needs the tensors its partitions name; the others may be shape-only)
        mine = set(C.isp_rank_names(mc, {n: tuple(t.shape) for n, t in ck["params"].items()}, r, 4, 2, 2))
        part = lambda d: {n: (t if n in mine else torch.empty(t.shape, device="meta")) for n, t in d.items()}  # noqa: E731
        C.save_isp_optimizer_shard(str(tmp_path), mc, r, 4, 2, 2, part(ck["master"]), part(ck["exp_avg"]), part(ck["exp_avg_sq"]), ck["adam_step"], ck["scaler"], ck["lr"], hyper)
        if r // 2 == 0:
            C.save_isp_model_shard(str(tmp_path), mc, ck["params"], co["t"], 2, co["w"], 2)
    assert sorted(os.listdir(tmp_path)) == sorted(os.listdir(ref)) == gold[0]["files"]
    for fn in gold[0]["files"]:
        if not fn.endswith(".json"):
            _deep_equal(C._load(os.path.join(tmp_path, fn)), C._load(os.path.join(ref, fn)), fn)
    # resume: two data ranks x micro_num micro-batches = one process on their union (mean loss = mean of the ranks' losses, same averaged gradients)
    tc = TrainConfig(seq_len=c["seq_len"], micro_bsz=1, micro_num=2 * c["micro_num"], total_steps=c["total_steps"], lr=1e-3, fixed_random_dataset_seqlen=True)
    base = OracleMoETrainer(PathConfig(mc, tc), torch.bfloat16)
    base.load_state(ck)
    tr = OracleISPTrainer(base, c["sp"])
    loaders = [iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold[0]["num_samples"], data_rank=d, data_world_size=2)) for d in range(2)]
    for _ in range(gold[0]["saved_after_step"]):
        for ld in loaders:
            next(ld)
    for k in range(2, 4):
        bl = [next(ld) for ld in loaders]
        batch = dict(input_ids=torch.cat([b["input_ids"] for b, _ in bl]))
        r = tr.train_step(batch, torch.cat([y for _, y in bl]))
        want_loss = (gold[0]["steps"][k]["loss"] + gold[2]["steps"][k]["loss"]) / 2     # ranks 0, 1 = data rank 0; ranks 2, 3 = data rank 1
        w = gold[0]["steps"][k]
        print(f"resumed step {k}: oracle loss {r['loss']:.5f} norms {r['grad_norm']} | reference {want_loss:.5f} {w['grad_norm']}")
        assert abs(r["loss"] - want_loss) <= 1e-3 * want_loss and abs(r["lr"] - w["lr"]) <= 1e-12
        for g_ in ("0_default", "1_embed_head"):
            assert abs(r["grad_norm"][g_] - w["grad_norm"][g_]) <= 1e-2 * w["grad_norm"][g_], (k, g_)


def test_moe_two_rank_reference_checkpoint_merges_is_reproduced_and_resumes(tmp_path):
    """tests/golden/ckpt_ref_moe_dp2/ = the files of a real TWO-rank run of the reference's INTERNLM_MoE model (its automatic expert parallelism: ep = 2, two
    of the four experts per rank; make_golden.py --ckpt-moe-mp): the model file without the experts (rank 0), one file per expert under its GLOBAL number
    (written by the rank that holds it), per rank an optimizer shard with three groups -- the dense parameters' and the gates' partition of the
    data-parallel group, and in group `moe_ep_size_2` ALL of the rank's own experts (the expert-data group has one rank) -- and its plan.
    The reader merges both ranks into the full state (master weights round to the model bit for bit); writing both ranks' files from it reproduces the
    reference's sixteen files tensor for tensor; the expert-parallel oracle resumes onto the reference's next two steps."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import ModelConfig, PathConfig, TrainConfig
    from internevo_amd.data import SyntheticLoader
    from oracle.moe_model import OracleMoEDataParallel

    gold = [json.load(open(os.path.join(G, f"ckpt_moe_dp2_rank{r}.json"))) for r in (0, 1)]
    c = gold[0]["config"]
    mc = ModelConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], num_layers=c["layers"], num_attention_heads=c["heads"], num_kv_attention_heads=c["heads"],
                     mlp_ratio=4 / 3, model_type="INTERNLM_MoE", num_experts=c["num_experts"], moe_capacity_factor=c["capacity_factor"], moe_loss_coeff=0.1)
    ref = os.path.join(G, "ckpt_ref_moe_dp2")
    assert gold[1]["ranks"] == {"DATA": [1, 2], "ZERO1": [1, 2], "EXPERT": [1, 2], "EXPERT_DATA": [0, 1]}
    ck = C.load_moe_checkpoint(ref, mc)
    assert ck["adam_step"] == 2 and ck["zero_world"] == 2 and len(ck["master"]) == len(C.state_dict_order(mc))
    for n in ck["params"]:
        assert torch.equal(ck["master"][n].to(ck["params"][n].dtype), ck["params"][n]), n
    hyper = dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3)
    for r in (0, 1):   # (a rank passes what it holds: everything dense, its own two experts)
        mine = {n for _, names in C.moe_groups(mc, 2, r) for n in names}
        part = lambda d: {n: t for n, t in d.items() if n in mine}  # noqa: E731
        C.save_moe_checkpoint(str(tmp_path), mc, part(ck["params"]), part(ck["master"]), part(ck["exp_avg"]), part(ck["exp_avg_sq"]), ck["adam_step"], ck["scaler"],
                              ck["lr"], hyper, world=2, rank=r)
    assert sorted(os.listdir(tmp_path)) == sorted(os.listdir(ref)) == gold[0]["files"]
    for fn in gold[0]["files"]:
        if not fn.endswith(".json"):
            ld = C._load if fn.startswith(("optimizer", "gpus")) else (lambda p_: torch.load(p_, weights_only=False))
            _deep_equal(ld(os.path.join(tmp_path, fn)), ld(os.path.join(ref, fn)), fn)
    # resume: the expert-parallel oracle (both ranks in one process) from the merged state onto the reference's steps 2 and 3
    tc = TrainConfig(seq_len=c["seq_len"], micro_bsz=1, micro_num=c["micro_num"], total_steps=c["total_steps"], lr=1e-3, fixed_random_dataset_seqlen=True)
    ora = OracleMoEDataParallel(PathConfig(mc, tc), 2)
    ora.load_state(ck)
    ora.calls_of = [2 * c["layers"] * c["micro_num"]] * 2     # gating calls each rank's two saved steps consumed (rank r's call k draws seed 5000 + 1000 r + k)
    loaders = [iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold[0]["num_samples"], data_rank=r, data_world_size=2)) for r in (0, 1)]
    for _ in range(2):
        for ld_ in loaders:
            next(ld_)
    for k in (2, 3):
        bl = [next(ld_) for ld_ in loaders]
        res = ora.train_step([b for b, _ in bl], [y for _, y in bl])
        for r in (0, 1):
            w = gold[r]["steps"][k]
            print(f"resumed step {k} rank {r}: oracle loss {res[r]['loss']:.5f} {res[r]['grad_norm']} | reference {w['loss']:.5f} {w['grad_norm']}")
            assert abs(res[r]["loss"] - w["loss"]) <= 1e-4 * w["loss"], (k, r)        # (measured 4e-6)
            for (g_, v), gw in zip(res[r]["grad_norm"].items(), w["grad_norm"].values()):
                assert abs(v - gw) <= 2e-3 * gw, (k, r, g_, v, gw)                     # (measured 2.5e-4)


def test_isp_layout_on_six_ranks_with_a_data_rank_that_holds_neither_embedding_nor_head(tmp_path):
    """tests/golden/ckpt_ref_isp6v1/ = a real SIX-process ISP run of the reference (tensor 2 (isp) x weight 2, THREE data replicas; make_golden.py --ckpt-isp6): the
    "embed_head" group's two parameters go to data ranks 0 and 1, data rank 2 holds none (hybrid_zero_optim.py:254-284) -- its files list the group's own
    parameters (ids 1, 2), carry no state and no flat weights for it.  The reader merges the six ranks; writing them again reproduces the reference's files tensor for tensor."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import ModelConfig

    gold = [json.load(open(os.path.join(G, f"ckpt_isp6v1_rank{r}.json"))) for r in range(6)]
    c = gold[0]["config"]
    mc = ModelConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], num_layers=c["layers"], num_attention_heads=c["heads"], num_kv_attention_heads=c["heads"],
                     mlp_ratio=8 / 3, model_type="INTERNLM", num_experts=1)
    ref = os.path.join(G, "ckpt_ref_isp6v1")
    for r, g_ in enumerate(gold):
        co = C.isp_coords(r, 6, c["sp"], c["wp"])
        rk = g_["ranks"]
        assert (co["t"], co["w"], co["d"], co["z"]) == (rk["TENSOR"][0], rk["WEIGHT"][0], rk["DATA"][0], rk["ZERO1"][0]) and rk["WEIGHT_DATA"][0] == co["z"]
        assert (co["data_world"], co["zero_world"]) == (rk["DATA"][1], rk["ZERO1"][1]) == (3, 3)
        assert g_["rank_unique_id"] == f"gpus-6_wp-{co['w']}_tp-{co['t']}_dp-{co['d']}_pp-0_zo-{co['z']}.pt"
    st2 = C._load(os.path.join(ref, "optimizer_tp1_wp1_pp0_dp2.pt"))
    assert [g_["params"] for g_ in st2["base_optim_states"]["param_groups"]] == [[0], [1, 2], []] and sorted(st2["base_optim_states"]["state"]) == [0]
    assert sorted(st2["flat_fp32_weights"]) == [0] and st2["zero_devide_optim_plan"][1][2] == []
    ck = C.load_checkpoint(ref, mc)
    assert ck["adam_step"] == 2 and ck["isp"] == dict(world=6, sp=2, wp=2) and ck["scaler"]["scale"] == gold[0]["grad_scaler"]["_scale"]
    for n in ck["params"]:
        assert torch.equal(ck["master"][n].to(torch.bfloat16), ck["params"][n]), n
    hyper = dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3)
    for r in range(6):
        co = C.isp_coords(r, 6, 2, 2)
        mine = set(C.isp_rank_names(mc, {n: tuple(t.shape) for n, t in ck["params"].items()}, r, 6, 2, 2))
        part = lambda d: {n: (t if n in mine else torch.empty(t.shape, device="meta")) for n, t in d.items()}  # noqa: E731
        C.save_isp_optimizer_shard(str(tmp_path), mc, r, 6, 2, 2, part(ck["master"]), part(ck["exp_avg"]), part(ck["exp_avg_sq"]), ck["adam_step"], ck["scaler"], ck["lr"], hyper)
        if r // 2 == 0:
            C.save_isp_model_shard(str(tmp_path), mc, ck["params"], co["t"], 2, co["w"], 2)
    assert sorted(os.listdir(tmp_path)) == sorted(os.listdir(ref)) == gold[0]["files"]
    for fn in gold[0]["files"]:
        if not fn.endswith(".json"):
            _deep_equal(C._load(os.path.join(tmp_path, fn)), C._load(os.path.join(ref, fn)), fn)


def test_moe_four_rank_reference_checkpoint_with_ranks_that_hold_no_gate(tmp_path):
    """tests/golden/ckpt_ref_moe_dp4/ = a real FOUR-rank run of the reference's INTERNLM_MoE model (make_golden.py --ckpt-moe-mp4): ep = 4, one expert per rank, and
    the two gate parameters of the fp32 group go to ZeRO ranks 0 and 1 -- ranks 2 and 3 hold NO parameter of that group (hybrid_zero_optim.py:254-284): their files
    carry no flat weights and no optimizer state for it, the group lists the gates themselves (ids 1, 2) and the expert group's id moves to 3.  The reader merges the four
    ranks; writing them again reproduces the reference's files tensor for tensor."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import ModelConfig

    gold = [json.load(open(os.path.join(G, f"ckpt_moe_dp4_rank{r}.json"))) for r in range(4)]
    c = gold[0]["config"]
    mc = ModelConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], num_layers=c["layers"], num_attention_heads=c["heads"], num_kv_attention_heads=c["heads"],
                     mlp_ratio=4 / 3, model_type="INTERNLM_MoE", num_experts=c["num_experts"], moe_capacity_factor=c["capacity_factor"], moe_loss_coeff=0.1)
    ref = os.path.join(G, "ckpt_ref_moe_dp4")
    st3 = C._load(os.path.join(ref, "optimizer_tp0_pp0_zo3.pt"))
    assert [g["params"] for g in st3["base_optim_states"]["param_groups"]] == [[0], [1, 2], [3]] and sorted(st3["base_optim_states"]["state"]) == [0, 3]
    assert sorted(st3["flat_fp32_weights"]) == [0, 2] and st3["zero_devide_optim_plan"][1][2:] == [[], []]
    ck = C.load_moe_checkpoint(ref, mc)
    assert ck["adam_step"] == 2 and ck["zero_world"] == 4 and len(ck["master"]) == len(C.state_dict_order(mc))
    for n in ck["params"]:
        assert torch.equal(ck["master"][n].to(ck["params"][n].dtype), ck["params"][n]), n
    hyper = dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3)
    for r in range(4):
        mine = {n for _, names in C.moe_groups(mc, 4, r) for n in names}
        part = lambda d: {n: t for n, t in d.items() if n in mine}  # noqa: E731
        C.save_moe_checkpoint(str(tmp_path), mc, part(ck["params"]), part(ck["master"]), part(ck["exp_avg"]), part(ck["exp_avg_sq"]), ck["adam_step"], ck["scaler"],
                              ck["lr"], hyper, world=4, rank=r)
    assert sorted(os.listdir(tmp_path)) == sorted(os.listdir(ref)) == gold[0]["files"]
    for fn in gold[0]["files"]:
        if not fn.endswith(".json"):
            ld = C._load if fn.startswith(("optimizer", "gpus")) else (lambda p_: torch.load(p_, weights_only=False))
            _deep_equal(ld(os.path.join(tmp_path, fn)), ld(os.path.join(ref, fn)), fn)


def test_moe_tensor_parallel_reference_checkpoint_merges_and_is_reproduced(tmp_path):
    """tests/golden/ckpt_ref_moe_tp2dp2/ = a real FOUR-rank run of the reference's INTERNLM_MoE model under data parallel 2 x Megatron tensor parallel 2
    (make_golden.py --ckpt-moe-tpdp): every expert a FeedForward over the tensor group (gshard_layer.py:421-433), expert parallelism 2 inside the data-parallel groups.
    Per tensor rank t: `model_tp{t}_pp0.pt` (its local parts; out_proj's bias on tensor rank 0 only), one `model_moe_layer{l}_expert{e}_tp{t}.pt` per expert, and per data rank
    d `optimizer_tp{t}_pp0_zo{d}.pt` + plan with the three groups partitioned from the LOCAL shapes.  The reader merges the four ranks into FULL tensors (the master weights
    round to the model bit for bit); cutting them again and writing every rank's files reproduces the reference's twenty-eight files tensor for tensor."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import ModelConfig
    from oracle.moe_model import param_shapes as moe_shapes

    gold = [json.load(open(os.path.join(G, f"ckpt_moe_tp2dp2_rank{r}.json"))) for r in range(4)]
    c = gold[0]["config"]
    mc = ModelConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], num_layers=c["layers"], num_attention_heads=c["heads"], num_kv_attention_heads=c["heads"],
                     mlp_ratio=4 / 3, model_type="INTERNLM_MoE", num_experts=c["num_experts"], moe_capacity_factor=c["capacity_factor"], moe_loss_coeff=0.1)
    ref = os.path.join(G, "ckpt_ref_moe_tp2dp2")
    for r, g_ in enumerate(gold):   # ranks: tensor groups are consecutive ranks, the expert groups live inside the data-parallel groups {t, t + 2}
        rk = g_["ranks"]
        assert (rk["TENSOR"], rk["DATA"], rk["EXPERT"], rk["EXPERT_DATA"]) == ([r % 2, 2], [r // 2, 2], [r // 2, 2], [0, 1])
        assert g_["rank_unique_id"] == f"gpus-4_wp-0_tp-{r % 2}_dp-{r // 2}_pp-0_zo-{r // 2}.pt"
    ck = C.load_moe_checkpoint(ref, mc)
    assert (ck["adam_step"], ck["zero_world"], ck["tp_world"]) == (2, 2, 2)
    full_shapes = moe_shapes(mc)
    assert {n: tuple(t.shape) for n, t in ck["params"].items()} == {n: tuple(s_) for n, s_ in full_shapes.items()}
    for n in ck["params"]:
        assert torch.equal(ck["master"][n].to(ck["params"][n].dtype), ck["params"][n]), n
    hyper = dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3)
    for t in range(2):
        for d in range(2):   # (a rank passes what it holds: its parts of everything dense and of its own two experts)
            mine = {n for _, names in C.moe_groups(mc, 2, d, t) for n in names}
            part = lambda dd: {n: C.tp_shard(n, x, t, 2, mc.head_dim).contiguous() for n, x in dd.items() if n in mine}  # noqa: E731
            C.save_moe_checkpoint(str(tmp_path), mc, part(ck["params"]), part(ck["master"]), part(ck["exp_avg"]), part(ck["exp_avg_sq"]), ck["adam_step"], ck["scaler"],
                                  ck["lr"], hyper, world=2, rank=d, tp_world=2, tp_rank=t)
    assert sorted(os.listdir(tmp_path)) == sorted(os.listdir(ref)) == gold[0]["files"] and len(gold[0]["files"]) == 28
    for fn in gold[0]["files"]:
        if not fn.endswith(".json"):
            ld = C._load if fn.startswith(("optimizer", "gpus")) else (lambda p_: torch.load(p_, weights_only=False))
            _deep_equal(ld(os.path.join(tmp_path, fn)), ld(os.path.join(ref, fn)), fn)


def test_stale_files_of_another_layout_are_removed_before_a_save(tmp_path):
    """A folder that held an ISP-layout save (or the expert files of a MoE save) and is written again in the plain layout must not keep the old files: load_checkpoint
    takes the ISP branch as soon as ANY model_tp*_wp* file is present and would load the stale weights (and the other way round)."""
    from internevo_amd import checkpoint as C

    def touch(*names):
        for n in names:
            open(os.path.join(tmp_path, n), "w").close()

    isp = ["model_tp0_wp0_pp0.pt", "model_tp1_wp1_pp0.pt", "optimizer_tp0_wp0_pp0_dp0.pt", "optimizer_tp1_wp1_pp0_dp0.pt", "gpus-4_wp-1_tp-1_dp-0_pp-0_zo-0.pt",
           "gpus-4_wp-0_tp-0_dp-0_pp-0_zo-0.pt"]
    plain = ["model_tp0_pp0.pt", "topo_tp0_pp0.json", "optimizer_tp0_pp0_zo0.pt", "optimizer_tp0_pp0_zo1.pt", "gpus-2_wp-0_tp-0_dp-0_pp-0_zo-0.pt", "gpus-2_wp-0_tp-0_dp-1_pp-0_zo-1.pt"]
    moe = ["model_moe_layer0_expert0_tp0.pt", "model_moe_layer1_expert3_tp0.pt"]
    keep = ["context.pt", "notes.txt"]
    touch(*isp, *plain, *moe, *keep)
    gone = C.remove_stale_shards(str(tmp_path), 2, 1)                                 # a plain 2-rank save
    assert sorted(gone) == sorted(isp + moe) and sorted(os.listdir(tmp_path)) == sorted(plain + keep)
    touch(*isp, *moe)
    gone = C.remove_stale_shards(str(tmp_path), 1, 2, 1, job_world=4, layout="isp", wp_world=2)   # an ISP save, sp 2 x wp 2 on 4 ranks
    assert sorted(gone) == sorted(plain + moe) and sorted(os.listdir(tmp_path)) == sorted(isp + keep)
    touch(*plain, *moe, "model_moe_layer2_expert0_tp0.pt", "model_moe_layer0_expert4_tp0.pt")
    gone = C.remove_stale_shards(str(tmp_path), 2, 1, layout="moe", num_experts=4, num_layers=2)  # a MoE save on 2 ranks: 2 layers, 4 experts
    assert sorted(gone) == sorted(isp + ["model_moe_layer2_expert0_tp0.pt", "model_moe_layer0_expert4_tp0.pt"])
    assert sorted(os.listdir(tmp_path)) == sorted(plain + moe + keep)
