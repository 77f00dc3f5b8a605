"""INTERNLM_MoE training step on the HIP engine (internevo_amd/moe_engine.py) against the UNMODIFIED reference's CPU run of the same model
(tests/golden/train_moe_bf16.json: 2 layers, 4 experts, top-2, Gumbel noise injected per gating call) and against the pinned oracle
(oracle/moe_model.py) step by step: loss, moe loss, the three optimizer-group norms, and the trained weights."""
import json
import os

import pytest
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
            assert abs(float(loss) - w["loss"]) <= 2e-3 * w["loss"], (k, float(loss), w["loss"])
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
