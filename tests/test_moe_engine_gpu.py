"""INTERNLM_MoE training step on the HIP engine (internevo_amd/moe_engine.py) against the UNMODIFIED reference's CPU run of the same model
(tests/golden/train_moe_bf16.json: 2 layers, 4 experts, top-2, Gumbel noise injected per gating call) and against the pinned oracle
(oracle/moe_model.py) step by step: loss, moe loss, the three optimizer-group norms, and the trained weights."""
import json
import os

import pytest
from conftest import xport
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cfg(gold):
    from internevo_amd.config import ModelConfig, PathConfig, TrainConfig

    c = gold["config"]
    mc = ModelConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], num_layers=c["layers"], num_attention_heads=c["heads"], num_kv_attention_heads=c["heads"],
                     mlp_ratio=4 / 3, model_type="INTERNLM_MoE", num_experts=c["num_experts"], moe_capacity_factor=c["capacity_factor"], moe_loss_coeff=0.1)
    tc = TrainConfig(seq_len=c["seq_len"], micro_bsz=1, micro_num=c["micro_num"], total_steps=c["total_steps"], lr=1e-3, fixed_random_dataset_seqlen=True)
    return PathConfig(mc, tc)


def test_moe_engine_matches_reference_trajectory(dev):
    """Six training steps of the reference's INTERNLM_MoE run.  Routing is a discontinuous function of bf16 activations: a token whose two best
    (noisy) gate logits are within rounding noise goes to another expert on another machine (the CPU oracle itself moves between two
    hosts), and with 128 tokens a handful of such flips shifts every gradient by several per cent.  So the parity statement has two parts:
      1. the DISCRETE decisions: the HIP engine's expert choices equal the free-running oracle's except for near-ties -- every flipped token's
         competing logits are within 0.05 -- on at most 10 % of the choices (checked at step 0, where both hold identical weights);
      2. everything DOWNSTREAM of the decisions: with the oracle teacher-forced onto the engine's choices (oracle.moe.top2gating), loss,
         moe loss and the three optimizer-group norms agree step by step at the dense model's tolerances, and so do the trained weights.
    The reference's own numbers (tests/golden/train_moe_bf16.json; the free-running oracle retraces them on CPU,
    test_moe_model_oracle_retraces_the_reference_internlm_moe_training_run) are asserted for the first two steps, before the flips compound."""
    from internevo_amd.data import SyntheticLoader
    from internevo_amd.moe_engine import MoEEngine
    from oracle import moe as MO
    from oracle.model import moe_formula_init
    from oracle.moe_model import OracleMoETrainer

    gold = json.load(open(os.path.join(G, "train_moe_bf16.json")))
    cfg = _cfg(gold)
    eng = MoEEngine(cfg, dev, init_fn=moe_formula_init, noise_fn=lambda call, S, E: MO.gumbel_noise((S, E), 5000 + call).to(dev))
    ora = OracleMoETrainer(cfg, torch.bfloat16)
    free = OracleMoETrainer(cfg, torch.bfloat16)
    loader = iter(SyntheticLoader(cfg.train.seq_len, 1, cfg.train.micro_num, True, gold["num_samples"]))
    worst_loss = worst_norm = 0.0
    for k, w in enumerate(gold["steps"]):
        batch, labels = next(loader)
        eng.keep_routes = []
        loss, moe_loss = eng.forward_backward(batch, labels)
        eng.step()
        st = eng.read_state()
        forced = [[r[0].cpu().long() for r in micro] for micro in eng.keep_routes]
        ref = ora.train_step(batch, labels, forced)
        print(f"step {k}: HIP loss {float(loss):.5f} moe {float(moe_loss):.5f} norms {st.group_norms} | forced oracle {ref['loss']:.5f} {ref['moe_loss']:.5f} "
              f"{ref['grad_norm']} | reference {w['loss']:.5f} {w['moe_loss']:.5f}")
        assert st.skip == 0 and st.loss_scale == w["loss_scale"]
        if k == 0:   # part 1: the decisions, against the free-running oracle on identical weights
            free.backward(batch, labels)
            flips = total = 0
            for i, micro in enumerate(eng.keep_routes):
                for l, (he, hl) in enumerate(micro):
                    oe, noisy, lg = free.routes[i][l]["expert"], free.routes[i][l]["noisy"], free.routes[i][l]["logits"]
                    he = he.cpu().long()
                    for c in range(2):
                        for s_ in torch.nonzero(he[c] != oe[c]).squeeze(1).tolist():
                            score = lg if c == 0 else noisy   # first choice: argmax of the gates (= of the logits); second: of the noisy logits
                            gap = abs(float(score[s_, he[c][s_]] - score[s_, oe[c][s_]]))
                            assert gap <= 5e-2 or he[0][s_] != oe[0][s_], f"micro {i} layer {l} token {s_} choice {c}: flipped with a logit gap of {gap:.4f}"
                            flips += 1
                    total += 2 * he.shape[1]
            print(f"[parity moe] routing: {flips} of {total} choices differ from the CPU oracle, all near-ties")
            assert flips <= 0.10 * total
        if k < 2:    # the reference's own trajectory, before routing flips compound through the updates
            assert abs(float(loss) - w["loss"]) <= 1e-3 * w["loss"], (k, float(loss), w["loss"])   # (measured 3.9e-4 and 1.4e-4)
        # part 2: downstream of the decisions
        # steps 0-2 (measured: loss within 3e-4, norms within 2.5e-2) at the dense tolerances; from step 3 on the two weight sets have
        # drifted apart by bf16 update rounding and the engine's choices are no longer the oracle's own arg-maxes (a 2-layer, 128-token model at
        # lr 1e-3 is chaotic in its routing): only a sanity band
        loss_tol, norm_tol = (1e-3, 3e-2) if k < 3 else (3e-2, 5e-1)
        if k < 3:
            worst_loss = max(worst_loss, abs(float(loss) - ref["loss"]) / ref["loss"])
        assert abs(float(loss) - ref["loss"]) <= loss_tol * ref["loss"], (k, float(loss), ref["loss"])
        assert abs(float(moe_loss) - ref["moe_loss"]) <= (3e-2 if k < 3 else 3e-1) * ref["moe_loss"], (k, float(moe_loss), ref["moe_loss"])   # (bf16 sums: 0.6 % per ulp)
        for gname, v in ref["grad_norm"].items():
            if k < 3:
                worst_norm = max(worst_norm, abs(st.group_norms[gname] - v) / v)
            assert abs(st.group_norms[gname] - v) <= norm_tol * v, (k, gname, st.group_norms[gname], v)
    print(f"[parity moe] forced routing, steps 0-2: max relative loss deviation {worst_loss:.2e} (bound 1e-3), group norms {worst_norm:.2e} (bound 3e-2)")
    worst = 0.0
    for n, p in eng.named_parameters():
        worst = max(worst, float((p.float().cpu() - ora.params[n].detach().float()).abs().max()))
    print("max |param diff| vs the forced oracle after training:", worst)
    assert worst <= 1e-1


def _v1_cfg(gold):
    from internevo_amd.config import ModelConfig, PathConfig, TrainConfig

    c = gold["config"]
    mc = ModelConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], num_layers=c["layers"], num_attention_heads=c["heads"], num_kv_attention_heads=c["heads"],
                     mlp_ratio=8 / 3, model_type="INTERNLM", num_experts=1)
    tc = TrainConfig(seq_len=c["seq_len"], micro_bsz=1, micro_num=c["micro_num"], total_steps=c["total_steps"], lr=1e-3, fixed_random_dataset_seqlen=True)
    return PathConfig(mc, tc)


def test_dense_internlm1_engine_matches_reference_trajectory(dev):
    """model_type INTERNLM (modeling_internlm.py, configs/7B_sft.py -- the model of the reference's published numbers): the InternLM-1 block
    with a plain SwiGLU FeedForward = MoEEngine's dense branch.  Six steps against the UNMODIFIED reference's bf16 CPU run
    (tests/golden/train_v1_bf16.json, make_golden.py --run v1_bf16) and against the pinned oracle (oracle.moe_model with num_experts = 1, which
    retraces that run on CPU): loss within 1e-3 relative (north_star), total gradient norm within 2e-2, loss scale equal, no skipped step;
    the trained weights against the oracle's."""
    from internevo_amd.data import SyntheticLoader
    from internevo_amd.moe_engine import MoEEngine
    from oracle.model import moe_formula_init
    from oracle.moe_model import OracleMoETrainer

    gold = json.load(open(os.path.join(G, "train_v1_bf16.json")))
    cfg = _v1_cfg(gold)
    eng = MoEEngine(cfg, dev, init_fn=moe_formula_init)
    assert eng.dense and eng.ep == 1
    ora = OracleMoETrainer(cfg, torch.bfloat16)
    assert sorted(n for n, _ in eng.named_parameters()) == sorted(ora.params), "parameter names of the dense model"
    loader = iter(SyntheticLoader(cfg.train.seq_len, 1, cfg.train.micro_num, True, gold["num_samples"]))
    worst_loss = worst_norm = 0.0
    for k, w in enumerate(gold["steps"]):
        batch, labels = next(loader)
        loss, moe_loss = eng.forward_backward(batch, labels)
        eng.step()
        st = eng.read_state()
        ref = ora.train_step(batch, labels)
        total = sum(v * v for v in st.group_norms.values()) ** 0.5
        ref_total = sum(v * v for v in ref["grad_norm"].values()) ** 0.5
        want = w["grad_norm"]["0_default"]
        print(f"step {k}: HIP loss {float(loss):.5f} norm {total:.4f} | oracle {ref['loss']:.5f} {ref_total:.4f} | reference {w['loss']:.5f} {want:.4f}")
        assert st.skip == 0 and st.loss_scale == w["loss_scale"] and float(moe_loss) == 0.0
        worst_loss = max(worst_loss, abs(float(loss) - w["loss"]) / w["loss"], abs(float(loss) - ref["loss"]) / ref["loss"])
        worst_norm = max(worst_norm, abs(total - want) / want, abs(total - ref_total) / ref_total)
        assert abs(float(loss) - w["loss"]) <= 1e-3 * w["loss"], (k, float(loss), w["loss"])
        assert abs(float(loss) - ref["loss"]) <= 1e-3 * ref["loss"], (k, float(loss), ref["loss"])
        assert abs(total - want) <= 2e-2 * want and abs(total - ref_total) <= 2e-2 * ref_total, (k, total, want, ref_total)
    print(f"[parity dense v1] max relative loss deviation {worst_loss:.2e} (bound 1e-3), total gradient norm {worst_norm:.2e} (bound 2e-2)")
    worst = 0.0
    for n, p in eng.named_parameters():
        worst = max(worst, float((p.float().cpu() - ora.params[n].detach().float()).abs().max()))
    print("max |param diff| vs the oracle after training:", worst)
    assert worst <= 2e-2


def test_dense_internlm1_engine_resumes_from_the_reference_checkpoint_and_its_own(dev, tmp_path):
    """tests/golden/ckpt_ref_v1/ (the REAL reference's model + optimizer files of the dense InternLM-1 model after two steps, make_golden.py --ckpt-v1):
    the HIP engine loads them and its next two steps are the reference's (ckpt_v1.json: loss <= 1e-3, gradient norm <= 2e-2, same lr and loss
    scale); then its own save_checkpoint -> a fresh engine -> bit-identical buffers and a bit-identical next step; the saved files load back
    through the reference-format reader."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import ModelConfig, PathConfig, TrainConfig
    from internevo_amd.data import SyntheticLoader
    from internevo_amd.moe_engine import MoEEngine

    gold = json.load(open(os.path.join(G, "ckpt_v1.json")))
    c = gold["config"]
    mc = ModelConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], num_layers=c["layers"], num_attention_heads=c["heads"], num_kv_attention_heads=c["heads"],
                     mlp_ratio=8 / 3, model_type="INTERNLM", num_experts=1)
    tc = TrainConfig(seq_len=c["seq_len"], micro_bsz=1, micro_num=c["micro_num"], total_steps=c["total_steps"], lr=1e-3, fixed_random_dataset_seqlen=True)
    cfg = PathConfig(mc, tc)
    eng = MoEEngine(cfg, dev, seed=5)
    eng.load_checkpoint(os.path.join(G, "ckpt_ref_v1"))
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"]))
    for _ in range(gold["saved_after_step"]):
        next(loader)
    for k, w in enumerate(gold["steps"][gold["saved_after_step"]:]):
        batch, labels = next(loader)
        lr = eng.lr_sched.lr()
        loss, _ = eng.forward_backward(batch, labels)
        eng.step()
        st = eng.read_state()
        total = sum(v * v for v in st.group_norms.values()) ** 0.5
        print(f"resumed step {k}: HIP loss {float(loss):.5f} norm {total:.4f} lr {lr:.3e} | reference {w['loss']:.5f} {w['grad_norm']['0_default']:.4f} {w['lr']:.3e}")
        assert abs(float(loss) - w["loss"]) <= 1e-3 * w["loss"] and abs(total - w["grad_norm"]["0_default"]) <= 2e-2 * total
        assert abs(lr - w["lr"]) <= 1e-12 and st.loss_scale == w["loss_scale"] and st.skip == 0
    folder = str(tmp_path / "ck_v1")
    eng.save_checkpoint(folder)
    assert sorted(os.listdir(folder)) == ["gpus-1_wp-0_tp-0_dp-0_pp-0_zo-0.pt", "model_tp0_pp0.pt", "optimizer_tp0_pp0_zo0.pt", "topo_tp0_pp0.json"]
    fresh = MoEEngine(cfg, dev, seed=9)
    fresh.load_checkpoint(folder)
    torch.cuda.synchronize()
    for name in ("params", "master", "exp_avg", "exp_avg_sq"):
        assert torch.equal(getattr(eng, name), getattr(fresh, name)), name
    batch, labels = next(loader)
    nxt = []
    for e in (eng, fresh):
        loss, _ = e.forward_backward(batch, labels)
        e.step()
        nxt.append((float(loss), e.read_state().group_norms))
    assert nxt[0] == nxt[1] and torch.equal(eng.params, fresh.params)
    ck = C.load_checkpoint(folder, mc)
    assert ck["adam_step"] == 4 and set(ck["params"]) == {n for n, _ in eng.named_parameters()}


def test_moe_engine_resumes_from_the_reference_checkpoint_and_its_own(dev, tmp_path):
    """tests/golden/ckpt_ref_moe/ (the REAL reference's INTERNLM_MoE checkpoint after two steps: model file without the experts, one file per expert, three
    optimizer groups; make_golden.py --ckpt-moe): the HIP engine loads it and -- gating noise continued at the eighth call -- its next step is the
    reference's (ckpt_moe.json: loss <= 1e-3, moe loss 3e-2, the three group norms 3e-2); the step after within the sanity band of near-tie routing flips;
    then its own save_checkpoint -> a fresh engine -> bit-identical buffers (gates and their moments included) and a bit-identical next step."""
    from internevo_amd import checkpoint as C
    from internevo_amd.config import ModelConfig, PathConfig, TrainConfig
    from internevo_amd.data import SyntheticLoader
    from internevo_amd.moe_engine import MoEEngine
    from oracle import moe as MO

    gold = json.load(open(os.path.join(G, "ckpt_moe.json")))
    c = gold["config"]
    mc = ModelConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], num_layers=c["layers"], num_attention_heads=c["heads"], num_kv_attention_heads=c["heads"],
                     mlp_ratio=4 / 3, model_type="INTERNLM_MoE", num_experts=c["num_experts"], moe_capacity_factor=c["capacity_factor"], moe_loss_coeff=0.1)
    tc = TrainConfig(seq_len=c["seq_len"], micro_bsz=1, micro_num=c["micro_num"], total_steps=c["total_steps"], lr=1e-3, fixed_random_dataset_seqlen=True)
    cfg = PathConfig(mc, tc)
    noise = lambda call, S, E: MO.gumbel_noise((S, E), 5000 + call).to(dev)  # noqa: E731
    eng = MoEEngine(cfg, dev, seed=5, noise_fn=noise)
    eng.load_checkpoint(os.path.join(G, "ckpt_ref_moe"))
    eng.calls = gold["saved_after_step"] * c["micro_num"] * c["layers"]
    loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"]))
    for _ in range(gold["saved_after_step"]):
        next(loader)
    for k, w in enumerate(gold["steps"][gold["saved_after_step"]:]):
        batch, labels = next(loader)
        loss, moe_loss = eng.forward_backward(batch, labels)
        eng.step()
        st = eng.read_state()
        print(f"resumed step {k}: HIP loss {float(loss):.5f} moe {float(moe_loss):.5f} norms {st.group_norms} | reference {w['loss']:.5f} {w['moe_loss']:.5f} {w['grad_norm']}")
        lt, mt, nt = (1e-3, 3e-2, 3e-2) if k == 0 else (3e-2, 3e-1, 5e-1)
        assert abs(float(loss) - w["loss"]) <= lt * w["loss"] and abs(float(moe_loss) - w["moe_loss"]) <= mt * w["moe_loss"]
        for (gname, v), (_, want) in zip(st.group_norms.items(), w["grad_norm"].items()):
            assert abs(v - want) <= nt * want, (k, gname, v, want)
        assert st.loss_scale == w["loss_scale"] and st.skip == 0
    folder = str(tmp_path / "ck_moe")
    eng.save_checkpoint(folder)
    assert sorted(os.listdir(folder)) == [f for f in gold["files"] if f not in ("context.pt", "sampler.pt", "schedulder.pt")]
    fresh = MoEEngine(cfg, dev, seed=9, noise_fn=noise)
    fresh.load_checkpoint(folder)
    fresh.calls = eng.calls
    torch.cuda.synchronize()
    for name in ("params", "master", "exp_avg", "exp_avg_sq", "wg", "wg_m", "wg_v"):
        assert torch.equal(getattr(eng, name), getattr(fresh, name)), name
    batch, labels = next(loader)
    nxt = []
    for e in (eng, fresh):
        loss, moe_loss = e.forward_backward(batch, labels)
        e.step()
        nxt.append((float(loss), float(moe_loss), e.read_state().group_norms))
    assert nxt[0] == nxt[1] and torch.equal(eng.params, fresh.params) and torch.equal(eng.wg, fresh.wg)
    assert C.load_moe_checkpoint(folder, mc)["adam_step"] == 4


def test_moe_engine_runs_with_device_generated_noise_and_default_init(dev):
    """The production path: Gumbel noise from the device generator, the family's default initialisation; the loss must fall."""
    from internevo_amd.data import SyntheticLoader
    from internevo_amd.moe_engine import MoEEngine

    gold = json.load(open(os.path.join(G, "train_moe_bf16.json")))
    cfg = _cfg(gold)
    cfg.train.total_steps = 12
    eng = MoEEngine(cfg, dev, seed=3)
    loader = iter(SyntheticLoader(cfg.train.seq_len, 1, cfg.train.micro_num, True, 4000))
    losses = []
    for _ in range(12):
        batch, labels = next(loader)
        loss, _ = eng.forward_backward(batch, labels)
        eng.step()
        losses.append(float(loss))
    st = eng.read_state()
    assert st.skipped_total == 0 and all(v == v for v in losses)
    assert losses[-1] < losses[0] - 0.3, losses


def _ep_engine_worker(rank, world, port, q, steps):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(xport(port)), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from internevo_amd.data import SyntheticLoader
        from internevo_amd.moe_engine import MoEEngine
        from oracle import moe as MO
        from oracle.model import moe_formula_init

        gold = json.load(open(os.path.join(G, f"train_moe2_bf16_rank{rank}.json")))
        cfg = _cfg(gold)
        eng = MoEEngine(cfg, dev, None, world, rank, init_fn=moe_formula_init,
                        noise_fn=lambda call, S, E: MO.gumbel_noise((S, E), 5000 + 1000 * rank + call).to(dev))
        assert eng.ep == 2 and eng.p["blocks.0.mlp.w13"].shape[0] == 2 and eng.groups[2] == "2_moe_ep_size_2"
        loader = iter(SyntheticLoader(cfg.train.seq_len, 1, cfg.train.micro_num, True, gold["num_samples"], data_rank=rank, data_world_size=world))
        out = []
        for _ in range(steps):
            batch, labels = next(loader)
            eng.keep_routes = []
            loss, moe_loss = eng.forward_backward(batch, labels)
            eng.step()
            st = eng.read_state()
            routes = [[r[0].cpu().long().numpy() for r in micro] for micro in eng.keep_routes]
            out.append((float(loss), float(moe_loss), dict(st.group_norms), st.skip, st.loss_scale, routes))
        q.put((rank, out, {n: p.float().cpu().numpy() for n, p in eng.named_parameters()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_moe_engine_expert_parallel_on_two_ranks_matches_the_reference_rules(dev):
    """Two data-parallel ranks of MoEEngine = the reference's automatic expert parallelism (ep = 2: two of the four experts per rank, the dispatch
    buffers exchanged by all_to_all), against oracle.moe_model.OracleMoEDataParallel -- itself pinned on the unmodified reference's 2-rank run
    (tests/golden/train_moe2_bf16_rank*.json) -- with the oracle teacher-forced onto the engine's routing, as in the single-rank test:
    per-rank loss and moe loss, the three GLOBAL group norms (expert gradients summed over the ranks' tokens without 1 / ep, the moe norm as
    (sum of squares) / dp), identical on both ranks; the reference's own numbers at step 0; after the steps the two ranks hold the
    same dense parameters and different experts."""
    import torch.multiprocessing as mp

    from internevo_amd.data import SyntheticLoader
    from oracle.moe_model import OracleMoEDataParallel

    steps = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ep_engine_worker, args=(r, 2, 29895, q, steps)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, out, params = q.get(timeout=400)
        res[r] = (out, params)
    for p in procs:
        p.join(60)
    gold = [json.load(open(os.path.join(G, f"train_moe2_bf16_rank{r}.json"))) for r in (0, 1)]
    cfg = _cfg(gold[0])
    ora = OracleMoEDataParallel(cfg, 2)
    loaders = [iter(SyntheticLoader(cfg.train.seq_len, 1, cfg.train.micro_num, True, gold[0]["num_samples"], data_rank=r, data_world_size=2)) for r in (0, 1)]
    for k in range(steps):
        bl = [next(ld) for ld in loaders]
        forced = [[[torch.from_numpy(x) for x in micro] for micro in res[r][0][k][5]] for r in (0, 1)]
        ref = ora.train_step([b for b, _ in bl], [y for _, y in bl], forced)
        for r in (0, 1):
            loss, moe_loss, norms, skip, scale, _ = res[r][0][k]
            w = gold[r]["steps"][k]
            print(f"step {k} rank {r}: HIP loss {loss:.5f} moe {moe_loss:.5f} norms {norms} | forced oracle {ref[r]['loss']:.5f} {ref[r]['moe_loss']:.5f} "
                  f"{ref[r]['grad_norm']} | reference {w['loss']:.5f} {w['moe_loss']:.5f} {w['grad_norm']}")
            assert skip == 0 and scale == w["loss_scale"]
            assert abs(loss - ref[r]["loss"]) <= 1e-3 * ref[r]["loss"], (k, r, loss, ref[r]["loss"])
            assert abs(moe_loss - ref[r]["moe_loss"]) <= 3e-2 * ref[r]["moe_loss"]
            for (g, v), (g2, v2) in zip(norms.items(), ref[r]["grad_norm"].items()):
                assert abs(v - v2) <= 3e-2 * v2, (k, r, g, v, v2)
            if k == 0:   # the reference's own numbers on identical weights (its routing differs from the engine's in a few near-ties: from step 1 on
                #          the free-running trajectories drift apart, as in the single-rank test)
                assert abs(loss - w["loss"]) <= 2e-3 * w["loss"], (k, r, loss, w["loss"])
                for (g, v), gw in zip(norms.items(), w["grad_norm"].values()):
                    assert abs(v - gw) <= 3e-2 * gw, (k, r, g, v, gw)
        assert res[0][0][k][2] == res[1][0][k][2], "both ranks report the same global group norms"
    p0, p1 = res[0][1], res[1][1]
    dense = [n for n in p0 if ".experts." not in n]
    assert dense and all((p0[n] == p1[n]).all() for n in dense), "the dense parameters stay replicated"
    ex0, ex1 = {n for n in p0 if ".experts." in n}, {n for n in p1 if ".experts." in n}
    assert ex0 and not (ex0 & ex1), "the ranks hold different experts"
    assert all(".wrapped_experts.0." in n or ".wrapped_experts.1." in n for n in ex0), "rank 0 holds experts 0 and 1"
    assert all(".wrapped_experts.2." in n or ".wrapped_experts.3." in n for n in ex1), "rank 1 holds experts 2 and 3"


def _moe_ckpt_worker(rank, world, port, q, out_folder):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(xport(port)), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from internevo_amd.data import SyntheticLoader
        from internevo_amd.moe_engine import MoEEngine
        from oracle import moe as MO

        gold = json.load(open(os.path.join(G, f"ckpt_moe_dp2_rank{rank}.json")))
        cfg = _cfg(gold)
        calls0 = gold["saved_after_step"] * cfg.model.num_layers * cfg.train.micro_num    # gating calls the saved steps consumed on this rank
        eng = MoEEngine(cfg, dev, None, world, rank, seed=21 + rank,
                        noise_fn=lambda call, S, E: MO.gumbel_noise((S, E), 5000 + 1000 * rank + calls0 + call).to(dev))
        eng.load_checkpoint(os.path.join(G, "ckpt_ref_moe_dp2"))
        eng.save_checkpoint(out_folder)           # straight back out: must be the reference's files
        loader = iter(SyntheticLoader(cfg.train.seq_len, 1, cfg.train.micro_num, True, gold["num_samples"], data_rank=rank, data_world_size=world))
        for _ in range(gold["saved_after_step"]):
            next(loader)
        batch, labels = next(loader)
        lr = eng.lr_sched.lr()
        loss, moe_loss = eng.forward_backward(batch, labels)
        eng.step()
        st = eng.read_state()
        q.put((rank, float(loss), float(moe_loss), dict(st.group_norms), lr, st.loss_scale, st.skip))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_moe_engine_two_rank_checkpoint_of_the_reference_loads_and_is_written_back(dev, tmp_path):
    """tests/golden/ckpt_ref_moe_dp2/ (a real two-rank run of the reference's INTERNLM_MoE model with its automatic expert parallelism, after two steps) into
    MoEEngine on two ranks: every rank takes the dense parameters, the gates and ITS two experts out of the merged state; save_checkpoint straight after the
    load writes the reference's sixteen files back tensor for tensor (model file without the experts, one file per expert under its global number from the
    rank that holds it, per rank the three-group optimizer shard + plan); the next step runs on the reference's batch: its loss within 5e-3 of the
    reference's step 2 (the routing is free here: a near-tie of a gate logit may fall the other way), learning rate and loss scale equal."""
    import torch.multiprocessing as mp
    from test_checkpoint import _deep_equal

    from internevo_amd import checkpoint as C

    out = str(tmp_path / "moe_dp2_out")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_moe_ckpt_worker, args=(r, 2, 29737, q, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=400) for _ in range(2)), key=lambda x: x[0])
    for p in procs:
        p.join(60)
    ref = os.path.join(G, "ckpt_ref_moe_dp2")
    assert sorted(os.listdir(out)) == sorted(os.listdir(ref))
    for fn in sorted(os.listdir(ref)):
        if not fn.endswith(".json"):
            ld = C._load if fn.startswith(("optimizer", "gpus")) else (lambda p_: torch.load(p_, weights_only=False))
            _deep_equal(ld(os.path.join(out, fn)), ld(os.path.join(ref, fn)), fn)
    for rank, loss, moe_loss, norms, lr, scale, skip in res:
        w = json.load(open(os.path.join(G, f"ckpt_moe_dp2_rank{rank}.json")))["steps"][2]
        print(f"rank {rank} resumed step: HIP loss {loss:.5f} moe {moe_loss:.4f} norms {norms} | reference {w['loss']:.5f} {w['moe_loss']:.4f} {w['grad_norm']}")
        assert skip == 0 and scale == w["loss_scale"] and abs(lr - w["lr"]) <= 1e-12
        assert abs(loss - w["loss"]) <= 5e-3 * w["loss"]
        for (g_, v), gw in zip(norms.items(), w["grad_norm"].values()):
            assert abs(v - gw) <= 5e-2 * gw, (rank, g_, v, gw)


# ---------------------------------------------------------------------------------------------- MoE x Megatron tensor parallelism (round 5)
def _tp_engine_worker(rank, world, port, q, steps):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(xport(port)), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from internevo_amd.data import SyntheticLoader
        from internevo_amd.moe_engine import MoEEngine
        from oracle import moe as MO
        from oracle.model import moe_formula_init

        gold = json.load(open(os.path.join(G, f"train_moe_tp2_bf16_rank{rank}.json")))
        cfg = _cfg(gold)
        # the ranks of a tensor group read the same micro-batches and draw the same gate noise (data rank 0 of 1)
        eng = MoEEngine(cfg, dev, None, world, rank, init_fn=moe_formula_init, tp_size=2,
                        noise_fn=lambda call, S, E: MO.gumbel_noise((S, E), 5000 + call).to(dev))
        F = eng.F
        assert eng.tp == 2 and eng.ep == 1 and eng.dp_world == 1 and eng.groups[2] == "2_moe_ep_size_1"
        assert eng.p["blocks.0.mlp.w13"].shape == (4, 2 * F, 256) and eng.p["blocks.0.mlp.w2"].shape == (4, 256, F) and 2 * F == 512
        assert eng.p["blocks.0.mixer.Wqkv.weight"].shape == (3 * 128, 256) and eng.p["head.weight"].shape == (256, 256)
        loader = iter(SyntheticLoader(cfg.train.seq_len, 1, cfg.train.micro_num, True, gold["num_samples"]))
        out = []
        for _ in range(steps):
            batch, labels = next(loader)
            eng.keep_routes = []
            loss, moe_loss = eng.forward_backward(batch, labels)
            eng.step()
            st = eng.read_state()
            routes = [[r[0].cpu().long().numpy() for r in micro] for micro in eng.keep_routes]
            out.append((float(loss), float(moe_loss), dict(st.group_norms), st.skip, st.loss_scale, routes))
        q.put((rank, out, {n: p.float().cpu().numpy() for n, p in eng.named_parameters()}))
    except Exception:   # (a worker that dies leaves its peer in a collective: report instead of letting the test wait for its timeout)
        import traceback

        q.put((rank, traceback.format_exc(), None))
        q.close()
        q.join_thread()   # the report is on the pipe before this rank goes
        os._exit(1)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_moe_engine_tensor_parallel_on_two_ranks_matches_the_reference_rules(dev):
    """INTERNLM_MoE on two Megatron tensor ranks (parallel.tensor = dict(size=2, mode="mtp"); gshard_layer.py:421-433: every expert is a FeedForward over
    the TENSOR group): heads, expert FFN units and vocabulary rows cut in two, gate / routing / dispatch / combine replicated on the same tokens with the
    same noise, the experts' outputs and the column-parallel input gradients all-reduced, the vocabulary-parallel loss, group norms summed over the tensor
    group with the replicated parameters counted once.  Checked: both ranks report the same loss, moe loss, routing and GLOBAL group norms; against the
    ONE-rank oracle teacher-forced onto the engine's routing (the sharded computation is the same mathematics) at the single-rank test's tolerances; against
    the unmodified reference's own 2-process tensor-parallel run (tests/golden/train_moe_tp2_bf16_rank*.json) at step 0, on identical weights; after the
    steps the ranks hold identical replicated parameters and complementary shards whose concatenation is the forced oracle's weights."""
    import numpy as np
    import torch.multiprocessing as mp

    from internevo_amd.data import SyntheticLoader
    from oracle.moe_model import OracleMoETrainer

    steps = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tp_engine_worker, args=(r, 2, 29897, q, steps)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, out, params = q.get(timeout=240)
        assert params is not None, f"rank {r} failed:\n{out}"
        res[r] = (out, params)
    for p in procs:
        p.join(60)
    gold = [json.load(open(os.path.join(G, f"train_moe_tp2_bf16_rank{r}.json"))) for r in (0, 1)]
    assert gold[0]["steps"] == gold[1]["steps"], "the reference's two tensor ranks report the same numbers"
    cfg = _cfg(gold[0])
    ora = OracleMoETrainer(cfg, torch.bfloat16)
    loader = iter(SyntheticLoader(cfg.train.seq_len, 1, cfg.train.micro_num, True, gold[0]["num_samples"]))
    worst_loss = worst_norm = 0.0
    for k in range(steps):
        batch, labels = next(loader)
        a, b = res[0][0][k], res[1][0][k]
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and a[3] == b[3] == 0 and a[4] == b[4], (k, a[:5], b[:5])
        assert all((x == y).all() for ma, mb in zip(a[5], b[5]) for x, y in zip(ma, mb)), "both tensor ranks route every token alike"
        forced = [[torch.from_numpy(x) for x in micro] for micro in a[5]]
        ref = ora.train_step(batch, labels, forced)
        loss, moe_loss, norms, _, scale, _ = a
        w = gold[0]["steps"][k]
        print(f"step {k}: HIP tp2 loss {loss:.5f} moe {moe_loss:.5f} norms {norms} | forced one-rank oracle {ref['loss']:.5f} {ref['moe_loss']:.5f} {ref['grad_norm']} | "
              f"reference tp2 {w['loss']:.5f} {w['moe_loss']:.5f} {w['grad_norm']}")
        assert scale == w["loss_scale"]
        worst_loss = max(worst_loss, abs(loss - ref["loss"]) / ref["loss"])
        assert abs(loss - ref["loss"]) <= 1e-3 * ref["loss"], (k, loss, ref["loss"])
        assert abs(moe_loss - ref["moe_loss"]) <= 3e-2 * ref["moe_loss"], (k, moe_loss, ref["moe_loss"])
        for gname, v in ref["grad_norm"].items():
            worst_norm = max(worst_norm, abs(norms[gname] - v) / v)
            assert abs(norms[gname] - v) <= 3e-2 * v, (k, gname, norms[gname], v)
        if k == 0:   # the reference's own tensor-parallel run, on identical weights (from step 1 on routing near-ties drift the trajectories apart)
            assert abs(loss - w["loss"]) <= 2e-3 * w["loss"], (loss, w["loss"])
            assert abs(moe_loss - w["moe_loss"]) <= 3e-2 * w["moe_loss"]
            for gname, v in w["grad_norm"].items():
                assert abs(norms[gname] - v) <= 3e-2 * v, (gname, norms[gname], v)
    print(f"[parity moe tp2] forced routing, steps 0-2: max relative loss deviation {worst_loss:.2e} (bound 1e-3), group norms {worst_norm:.2e} (bound 3e-2)")
    p0, p1 = res[0][1], res[1][1]
    assert set(p0) == set(p1) == set(ora.params)
    worst = 0.0
    H, d = cfg.model.num_attention_heads, cfg.model.head_dim
    for n in p0:
        full = ora.params[n].detach().float().numpy()
        if n.endswith(("norm1.weight", "norm2.weight", "norm.weight", "gate.wg.weight", "out_proj.bias")) or n == "embedding.weight":
            assert (p0[n] == p1[n]).all(), f"{n}: replicated parameters stay identical"
            got = p0[n]
        elif "mixer.Wqkv" in n:   # every rank: "(three h d)" rows of ITS heads
            parts = [x.reshape((3, H // 2, d) + x.shape[1:]) for x in (p0[n], p1[n])]
            got = np.concatenate(parts, axis=1).reshape(full.shape)
        elif n.endswith(("out_proj.weight", "w2.weight")):
            got = np.concatenate([p0[n], p1[n]], axis=1)
        else:   # w1 / w3 / head: rows
            got = np.concatenate([p0[n], p1[n]], axis=0)
        assert got.shape == full.shape, (n, got.shape, full.shape)
        worst = max(worst, float(np.abs(got - full).max()))
    print("max |param diff| of the re-assembled shards vs the forced one-rank oracle after training:", worst)
    assert worst <= 1e-2   # (measured 4.5e-3 on weights of std 0.02 after the training steps: profiles/r05_moe_tensor_parallel_parity.log)


def _tp_dp_engine_worker(rank, world, port, q, steps):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(xport(port)), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from internevo_amd.data import SyntheticLoader
        from internevo_amd.moe_engine import MoEEngine
        from oracle import moe as MO
        from oracle.model import moe_formula_init

        gold = json.load(open(os.path.join(G, f"train_moe_tp2dp2_bf16_rank{rank}.json")))
        cfg = _cfg(gold)
        dp_rank = rank // 2
        eng = MoEEngine(cfg, dev, None, world, rank, init_fn=moe_formula_init, tp_size=2,
                        noise_fn=lambda call, S, E: MO.gumbel_noise((S, E), 5000 + 1000 * dp_rank + call).to(dev))
        assert eng.tp == 2 and eng.dp_world == 2 and eng.ep == 2 and eng.ep_rank == dp_rank and eng.groups[2] == "2_moe_ep_size_2"
        assert eng.p["blocks.0.mlp.w13"].shape == (2, 512, 256) and eng.p["blocks.0.mlp.w2"].shape == (2, 256, 256)
        loader = iter(SyntheticLoader(cfg.train.seq_len, 1, cfg.train.micro_num, True, gold["num_samples"], data_rank=dp_rank, data_world_size=2))
        out = []
        for _ in range(steps):
            batch, labels = next(loader)
            eng.keep_routes = []
            loss, moe_loss = eng.forward_backward(batch, labels)
            eng.step()
            st = eng.read_state()
            routes = [[r[0].cpu().long().numpy() for r in micro] for micro in eng.keep_routes]
            out.append((float(loss), float(moe_loss), dict(st.group_norms), st.skip, st.loss_scale, routes))
        q.put((rank, out, {n: p.float().cpu().numpy() for n, p in eng.named_parameters()}))
    except Exception:
        import traceback

        q.put((rank, traceback.format_exc(), None))
        q.close()
        q.join_thread()   # the report is on the pipe before this rank goes
        os._exit(1)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(1200)
def test_moe_engine_tensor_parallel_2_x_expert_parallel_2_on_four_ranks(dev):
    """Two tensor groups side by side (data parallel 2 x tensor 2, four staged ranks): the expert groups live INSIDE the data-parallel groups [0, 2] and
    [1, 3] (process_group_initializer.py:493-524), so every rank holds its tensor part of two of the four experts; the dispatch buffers cross the data-parallel
    ranks by all_to_all, the experts' partial outputs are summed over the tensor group.  Against the two-data-rank oracle (OracleMoEDataParallel, pinned on the
    reference's ep-2 run) teacher-forced onto the engine's routing, and against the unmodified reference's own 4-process run at step 0
    (tests/golden/train_moe_tp2dp2_bf16_rank*.json)."""
    import torch.multiprocessing as mp

    from internevo_amd.data import SyntheticLoader
    from oracle.moe_model import OracleMoEDataParallel

    steps = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tp_dp_engine_worker, args=(r, 4, 29899, q, steps)) for r in range(4)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(4):
        r, out, params = q.get(timeout=900)   # (four ranks import torch and build engines on one box, last in the suite's order: tests/conftest.py)
        assert params is not None, f"rank {r} failed:\n{out}"
        res[r] = (out, params)
    for p in procs:
        p.join(60)
    gold = [json.load(open(os.path.join(G, f"train_moe_tp2dp2_bf16_rank{r}.json"))) for r in range(4)]
    cfg = _cfg(gold[0])
    ora = OracleMoEDataParallel(cfg, 2)
    loaders = [iter(SyntheticLoader(cfg.train.seq_len, 1, cfg.train.micro_num, True, gold[0]["num_samples"], data_rank=r, data_world_size=2)) for r in (0, 1)]
    for k in range(steps):
        bl = [next(ld) for ld in loaders]
        for a, b in ((0, 1), (2, 3)):   # the ranks of a tensor group agree on everything
            x, y = res[a][0][k], res[b][0][k]
            assert x[:5] == y[:5], (k, a, b, x[:5], y[:5])
            assert all((u == v).all() for mu, mv in zip(x[5], y[5]) for u, v in zip(mu, mv))
        forced = [[[torch.from_numpy(x) for x in micro] for micro in res[r][0][k][5]] for r in (0, 2)]
        ref = ora.train_step([b for b, _ in bl], [y for _, y in bl], forced)
        for dp, r in enumerate((0, 2)):
            loss, moe_loss, norms, skip, scale, _ = res[r][0][k]
            w = gold[r]["steps"][k]
            print(f"step {k} data rank {dp}: HIP loss {loss:.5f} moe {moe_loss:.5f} norms {norms} | forced oracle {ref[dp]['loss']:.5f} {ref[dp]['moe_loss']:.5f} "
                  f"{ref[dp]['grad_norm']} | reference {w['loss']:.5f} {w['moe_loss']:.5f} {w['grad_norm']}")
            assert skip == 0 and scale == w["loss_scale"]
            assert abs(loss - ref[dp]["loss"]) <= 1e-3 * ref[dp]["loss"], (k, dp, loss, ref[dp]["loss"])
            assert abs(moe_loss - ref[dp]["moe_loss"]) <= 3e-2 * ref[dp]["moe_loss"]
            for (g, v), (g2, v2) in zip(norms.items(), ref[dp]["grad_norm"].items()):
                assert abs(v - v2) <= 3e-2 * v2, (k, dp, g, v, v2)
            if k == 0:
                assert abs(loss - w["loss"]) <= 2e-3 * w["loss"], (dp, loss, w["loss"])
                for (g, v), gw in zip(norms.items(), w["grad_norm"].values()):
                    assert abs(v - gw) <= 3e-2 * gw, (dp, g, v, gw)
        assert res[0][0][k][2] == res[2][0][k][2], "every rank reports the same global group norms"
    names = [set(res[r][1]) for r in range(4)]
    assert names[0] == names[1] and names[2] == names[3]
    ex0, ex2 = {n for n in names[0] if ".experts." in n}, {n for n in names[2] if ".experts." in n}
    assert ex0 and not (ex0 & ex2) and all(".wrapped_experts.0." in n or ".wrapped_experts.1." in n for n in ex0)
    assert all(".wrapped_experts.2." in n or ".wrapped_experts.3." in n for n in ex2)
    dense = [n for n in names[0] if ".experts." not in n]
    assert all((res[0][1][n] == res[2][1][n]).all() and (res[1][1][n] == res[3][1][n]).all() for n in dense), "a shard stays replicated over its data-parallel group"
    n = "blocks.0.mlp.moe_layer.experts.wrapped_experts.0.w1.weight"
    assert res[0][1][n].shape == (256, 256) and not (res[0][1][n] == res[1][1][n]).all(), "the tensor ranks hold different rows of an expert"
