"""Fused metric pass: ie_ce_fwd_metric + ie_metric_accumulate and the AccPerplex host mirror vs
the REAL reference's AccPerplex / LossWithTypeId accumulators (tests/golden/metrics.*) and vs the oracle on random cases."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


def _feed(metric, dev, logits, labels, type_ids, dtype):
    from internevo_amd import kernels as K

    T = logits.shape[0]
    lg = logits.to(dev, dtype)
    lab = labels.to(dev)
    argmax = torch.empty(T, dtype=torch.int32, device=dev)
    nll = torch.empty(T, dtype=torch.float32, device=dev)
    K.ce_fwd(lg, lab, argmax_rows=argmax, nll_rows=nll)
    metric.update_fused(nll, argmax, lab)
    return argmax, nll


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_fused_metric_matches_reference_accumulators(dev, dtype):
    from internevo_amd.metrics import AccPerplex

    gold = json.load(open(os.path.join(G, "metrics.json")))
    z = np.load(os.path.join(G, "metrics.npz"))
    logits, labels, type_ids = (torch.from_numpy(z[k]) for k in ("logits", "labels", "type_ids"))  # logits are bf16-representable
    m = AccPerplex(dev, None, gold["dataset_types"])
    m.set_current_type_ids(type_ids)
    for i, want in enumerate(gold["trace"]):
        _feed(m, dev, logits[i], labels[i], type_ids[i], dtype)
        # integer-valued accumulators: exact
        assert float(m.right) == want["right"] and float(m.total) == want["total"] and float(m.token_num) == want["token_num"]
        assert m.ds_right.tolist() == want["ds_right"] and m.ds_tokens.tolist() == want["ds_tokens"]
        assert m.ds_token_num.tolist() == want["ds_token_num"]
        # fp32 sums: same terms, different summation order / lse-vs-ratio form
        assert abs(float(m.total_log_probs) - want["total_log_probs"]) <= 2e-6 * want["total_log_probs"]
        assert abs(float(m.loss) - want["loss"]) <= 2e-6 * want["loss"]
        np.testing.assert_allclose(m.ds_loss.cpu().numpy(), np.array(want["ds_loss"], dtype=np.float32), rtol=2e-6)
    res = m.get_metric(reset=True)
    assert list(res) == list(gold["get_metric"]), "same keys in the same order as the reference's dict"
    for k, v in gold["get_metric"].items():
        assert abs(res[k] - v) <= 1.01e-4, (k, res[k], v)
    assert float(m.total) == 0.0 and m.ds_tokens.tolist() == [0, 0, 0]


def test_fused_metric_argmax_ties_ragged_vocab_and_no_types(dev):
    """First-index argmax on ties (torch.argmax contract), a vocab that is not a multiple of the vector width, a strided
    logits view, all-ignored rows, no dataset types."""
    from internevo_amd.metrics import AccPerplex
    from oracle.ops import AccPerplexOracle

    gen = torch.Generator().manual_seed(5)
    T, V = 300, 1003
    buf = (torch.randn(T, V + 5, generator=gen) * 2).to(torch.bfloat16)
    logits = buf[:, :V]
    for r in range(0, T, 3):  # plant duplicated maxima
        j = torch.randint(0, V, (3,), generator=gen)
        logits[r, j] = 9.0
    labels = torch.randint(0, V, (T,), generator=gen)
    am = logits.float().argmax(-1)
    labels[::2] = am[::2]
    labels[5:40] = -100
    m = AccPerplex(dev)
    ora = AccPerplexOracle(None)
    for _ in range(2):
        argmax, nll = _feed(m, dev, logits, labels, None, torch.bfloat16)
        ora.update(logits.float(), labels)
    assert torch.equal(argmax.cpu().long(), am)
    assert float(m.right) == float(ora.right) and float(m.total) == float(ora.total)
    assert abs(float(m.total_log_probs) - float(ora.total_log_probs)) <= 3e-6 * float(ora.total_log_probs)
    a, b = m.get_metric(), ora.get_metric()
    assert list(a) == list(b)
    for k in a:
        assert abs(a[k] - b[k]) <= 1.01e-4 * max(1.0, abs(b[k])), (k, a[k], b[k])


def test_engine_metric_hook_matches_reference_training_metric(dev):
    """The engine drives the metric from inside the CE sweep of every micro-batch (SchedulerMetricHook.post_helper_func).
    Against the metric dicts the UNMODIFIED reference training loop produced step by step (tests/golden/train_pin_bf16.json):
    token counts exact, loss / perplexity to bf16-trajectory tolerance, accuracy to a few near-tie argmax flips; and
    attaching the metric must not change the training step by a bit."""
    import math

    from internevo_amd.config import tiny
    from internevo_amd.data import SyntheticLoader
    from internevo_amd.engine import InternLM2Engine
    from internevo_amd.metrics import AccPerplex
    from oracle.model import formula_init

    gold = json.load(open(os.path.join(G, "train_pin_bf16.json")))
    c = gold["config"]
    runs = []
    for with_metric in (False, True):
        cfg = tiny(c["hidden"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], c["seq_len"], c["micro_num"], 1e-3, c["total_steps"])
        eng = InternLM2Engine(cfg, dev, init_fn=formula_init)
        metric = AccPerplex(dev, None, ["en", "cn", "code"])
        if with_metric:
            eng.attach_metric(metric)
        loader = iter(SyntheticLoader(c["seq_len"], 1, c["micro_num"], True, gold["num_samples"]))
        tr = []
        for k in range(len(gold["steps"])):
            batch, labels = next(loader)
            loss = float(eng.forward_backward(batch, labels))
            eng.step()
            tr.append(loss)
            if with_metric:
                res, want = metric.get_metric(reset=True), gold["steps"][k]["metric"]
                print(k, res)
                assert list(res) == list(want)
                for key, w in want.items():
                    g = res[key]
                    if isinstance(w, float) and math.isnan(w):
                        assert math.isnan(g), key
                    elif key.startswith("tokens/"):
                        assert g == w, key
                    elif key.startswith("acc"):
                        assert abs(g - w) <= 0.012, (k, key, g, w)
                    else:
                        assert abs(g - w) <= 3e-3 * abs(w) + 1.01e-4, (k, key, g, w)
        eng.drain()  # the last step's AdamW runs on the optimizer stream
        runs.append((tr, eng.params.clone()))
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1])
