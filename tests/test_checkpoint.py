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


def test_saved_files_equal_the_reference_files(tmp_path):
    """load(reference files) -> save -> the same file set, the same keys / dtypes / shapes / values, tensor for tensor."""
    from internevo_amd import checkpoint as C

    gold = json.load(open(os.path.join(G, "ckpt.json")))
    cfg = _cfg()
    ck = C.load_checkpoint(REF, cfg.model)
    out = str(tmp_path / "ck")
    C.save_checkpoint(out, cfg.model, ck["params"], ck["master"], ck["exp_avg"], ck["exp_avg_sq"], ck["adam_step"], ck["scaler"], ck["lr"],
                      dict(weight_decay=0.01, betas=(0.9, 0.95), eps=1e-8, initial_lr=1e-3))
    assert sorted(os.listdir(out)) == gold["files"]
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
    """tests/golden/ckpt_load.json: the oracle trained 2 steps, save_checkpoint() wrote the files, the REAL reference loaded them
    with its own load_model_checkpoint / load_optimizer_checkpoint and trained 2 more steps (make_golden.py --ckpt-load).  The
    oracle, simply continuing, must land on the same losses and grad norms (given the learning rates the resumed reference used:
    its scheduler files are not part of this slice, the harness re-stepped a fresh scheduler)."""
    from internevo_amd.data import SyntheticLoader
    from oracle.step import OracleTrainer

    gold = json.load(open(os.path.join(G, "ckpt_load.json")))
    cfg = _cfg()
    c = gold["config"]
    tr = OracleTrainer(cfg, torch.bfloat16)
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"]))
    for w in gold["oracle_steps_before_save"]:
        g = tr.train_step(*next(loader))
        assert g["loss"] == w["loss"] and g["grad_norm"] == w["grad_norm"], "the oracle itself is deterministic"
    for w in gold["reference_steps_after_load"]:
        tr._lr = lambda lr=w["lr"]: lr
        g = tr.train_step(*next(loader))
        assert abs(g["loss"] - w["loss"]) <= 2e-3 * abs(w["loss"]), (g["loss"], w["loss"])
        assert abs(g["grad_norm"] - w["grad_norm"]["0_default"]) <= 1e-2 * w["grad_norm"]["0_default"]
        assert g["loss_scale"] == w["loss_scale"] and w["ok"]


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
