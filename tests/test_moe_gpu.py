"""GShard MoE layer on the HIP kernels (csrc/moe.hip, internevo_amd/moe.py) against
  * tests/golden/moe.npz        -- the REAL top2gating + dispatch / combine einsums on five routing regimes (balanced, capacity drops,
                                    min_capacity binding, one hot expert, exact ties),
  * tests/golden/moe_layer.npz  -- the REAL GShardMOELayer run forward and backward on CPU (output, l_aux, every gradient),
  * oracle.moe.moe_layer        -- the pinned index-form restatement, at a size the fixtures do not reach.
"""
import json
import os

import numpy as np
import pytest
from conftest import xport
import torch

from oracle import moe as MO  # tests may use the oracle; the product never does

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BF16 = torch.bfloat16


def _bf(a):
    return torch.from_numpy(a.copy()).view(BF16)


def _exact_logit_operands(logits):
    """x (bf16) and wg (fp32) with float(x) @ wg^T == logits EXACTLY: every fp32 logit is split into three bf16 pieces (8 + 8 + 8
    mantissa bits, by truncation: disjoint bit fields of one sign) that sit in their own columns; any summation order is exact in fp32."""
    S, E = logits.shape
    M = (3 * E + 7) // 8 * 8
    x = torch.zeros(S, M, dtype=BF16)
    wg = torch.zeros(E, M)
    rest = logits.clone()
    for j in range(3):
        piece = (rest.view(torch.int32) & -65536).view(torch.float32).to(BF16)   # TRUNCATED to bf16: the pieces are disjoint bit fields
        x[:, j * E : (j + 1) * E] = piece
        rest = rest - piece.float()
    assert float(rest.abs().max()) == 0.0, "three bf16 pieces hold every bit of an fp32 value"
    for e in range(E):
        for j in range(3):
            wg[e, j * E + e] = 1.0
    return x, wg


@pytest.mark.parametrize("case", ["balanced", "drops", "min_capacity", "hot_expert", "ties"])
def test_routing_matches_the_reference_gating_on_five_regimes(dev, case):
    from internevo_amd import kernels as K
    from internevo_amd.moe import MoELayer, capacity

    z = np.load(os.path.join(G, "moe.npz"))
    meta = {m["name"]: m for m in json.load(open(os.path.join(G, "moe.json")))}[case]
    S, E, cf, mincap = meta["S"], meta["E"], meta["capacity_factor"], meta["min_capacity"]
    logits, noise = torch.from_numpy(z[f"{case}.logits"]), torch.from_numpy(z[f"{case}.noise"])
    ref = MO.top2gating(logits, cf, mincap, noise)
    assert capacity(S, E, cf, mincap) == ref["capacity"] == meta["capacity"]
    x, wg = _exact_logit_operands(logits)
    lay = MoELayer(x.shape[1], 256, E, S, dev, cf, mincap)
    L, st = K._L(), K._stream
    xd, wgd, nd = x.to(dev), wg.to(dev), noise.to(dev)
    K.check(L.ie_moe_gate_fwd(K._p(xd), xd.stride(0), K._p(wgd), K._p(nd), S, x.shape[1], E, K._p(lay.logits), K._p(lay.gates), K._p(lay.expert), st()), "gate")
    K.check(L.ie_moe_route(K._p(lay.gates), K._p(lay.expert), S, E, lay.C, K._p(lay.row), K._p(lay.weight), K._p(lay.token_of), K._p(lay.l_aux),
                           K._p(lay.exp_counts), st()), "route")
    assert torch.equal(lay.logits.cpu(), logits), "the logits operand construction is exact"
    assert torch.equal(lay.expert.cpu().long(), ref["expert"]), "first / second choices"
    want_row = torch.where(ref["slot"] >= 0, ref["expert"] * ref["capacity"] + ref["slot"], torch.full_like(ref["slot"], -1))
    assert torch.equal(lay.row.cpu().long(), want_row), "slots in the capacity buffers (token order, second choices behind first ones, drops)"
    assert torch.equal(lay.exp_counts.cpu().long(), ref["exp_counts"])
    assert torch.allclose(lay.weight.cpu(), ref["weight"], rtol=2e-6, atol=1e-7)
    assert abs(float(lay.l_aux) - float(ref["l_aux"].to(BF16))) <= 1e-2 * float(ref["l_aux"]), (float(lay.l_aux), float(ref["l_aux"]))
    assert int((lay.row >= 0).sum()) == 2 * S - meta["dropped"]
    # dispatch / combine on the fixture's tokens and "expert outputs" (fp32 there; bf16 here)
    M = 16
    xt = torch.from_numpy(z[f"{case}.x"]).to(BF16)
    ein = torch.empty(E * lay.C, M, dtype=BF16, device=dev)
    xtd = xt.to(dev)
    K.check(L.ie_moe_dispatch(K._p(xtd), M, K._p(lay.token_of), E * lay.C, M, K._p(ein), st()), "dispatch")
    want = MO.dispatch(xt.float(), ref, E).reshape(E * lay.C, M)
    assert torch.equal(ein.float().cpu(), want), "dispatch = the boolean-mask einsum (a copy of the token row, zeros elsewhere)"
    eo = torch.from_numpy(z[f"{case}.expert_out"]).to(BF16).reshape(E * lay.C, M)
    out = torch.empty(S, M, dtype=BF16, device=dev)
    eod = eo.to(dev)
    K.check(L.ie_moe_combine_fwd(K._p(eod), K._p(lay.row), K._p(lay.weight), S, M, K._p(out), M, st()), "combine")
    want = MO.combine(eo.reshape(E, lay.C, M), ref)
    assert torch.allclose(out.float().cpu(), want.float(), rtol=1e-2, atol=1e-2)


def _run_layer(dev, x, dy, wg, w1, w3, w2, noise, cf, mincap, aux_coeff):
    from internevo_amd.moe import MoELayer

    S, M = x.shape
    E, F = wg.shape[0], w1.shape[1]
    lay = MoELayer(M, F, E, S, dev, cf, mincap)
    w13 = torch.cat([w1, w3], dim=1).contiguous().to(dev)          # [E, 2F, M]
    w2d, wgd, xd, dyd = w2.contiguous().to(dev), wg.to(dev), x.to(dev), dy.to(dev)
    out = torch.empty(S, M, dtype=BF16, device=dev)
    l_aux = lay.forward(xd, wgd, w13, w2d, out, noise=noise.to(dev))
    dx = torch.empty(S, M, dtype=BF16, device=dev)
    d_wg = torch.empty_like(wgd)
    d_w13, d_w2 = torch.empty_like(w13), torch.empty_like(w2d)
    lay.backward(dyd, wgd, w13, w2d, dx, d_wg, d_w13, d_w2, accumulate=False, loss_scale_dev=None, aux_factor=aux_coeff)
    return lay, out.cpu(), float(l_aux), dx.cpu(), d_wg.cpu(), d_w13[:, :F].cpu(), d_w13[:, F:].cpu(), d_w2.cpu()


def _close(a, b, what, rtol=2e-2):
    a, b = a.float(), b.float()
    tol = rtol * float(b.abs().max())
    err = float((a - b).abs().max())
    assert err <= tol, f"{what}: max |diff| {err:.3e} vs tolerance {tol:.3e}"


@pytest.mark.parametrize("case", ["balanced", "drops", "hot"])
def test_layer_matches_the_real_gshard_layer_forward_and_backward(dev, case):
    z = np.load(os.path.join(G, "moe_layer.npz"))
    meta = {m["name"]: m for m in json.load(open(os.path.join(G, "moe_layer.json")))}[case]
    S, E, M = meta["S"], meta["E"], meta["M"]
    x, dy = _bf(z[f"{case}.x"]).reshape(S, M), _bf(z[f"{case}.dy"]).reshape(S, M)
    wg = torch.from_numpy(z[f"{case}.wg"])
    w1, w3, w2 = (torch.stack([_bf(z[f"{case}.e{e}.{n}"]) for e in range(E)]) for n in ("w1", "w3", "w2"))
    lay, out, l_aux, dx, d_wg, d_w1, d_w3, d_w2 = _run_layer(dev, x, dy, wg, w1, w3, w2, torch.from_numpy(z[f"{case}.noise"]), meta["capacity_factor"],
                                                            meta["min_capacity"], meta["aux_coeff"])
    assert [int(c) for c in lay.exp_counts.cpu()] == meta["exp_counts"]
    assert l_aux == meta["l_aux"], "auxiliary loss (bf16-rounded like the reference's gate output)"
    _close(out, _bf(z[f"{case}.out"]).reshape(S, M), f"{case} output")
    _close(dx, _bf(z[f"{case}.dx"]).reshape(S, M), f"{case} d input")
    _close(d_wg, torch.from_numpy(z[f"{case}.d_wg"]), f"{case} d gate weight")
    for e in range(E):
        _close(d_w1[e], _bf(z[f"{case}.e{e}.d_w1"]), f"{case} d expert {e} w1")
        _close(d_w3[e], _bf(z[f"{case}.e{e}.d_w3"]), f"{case} d expert {e} w3")
        _close(d_w2[e], _bf(z[f"{case}.e{e}.d_w2"]), f"{case} d expert {e} w2")


def test_layer_matches_the_oracle_at_a_larger_size_with_drops(dev):
    """4096 tokens, 8 experts, capacity factor 0.5 (a third of the choices dropped), hidden 512: forward, auxiliary loss and every gradient
    against oracle.moe.moe_layer (itself pinned on the real layer)."""
    g = torch.Generator().manual_seed(5)
    S, E, M, F = 4096, 8, 512, 1024
    x = torch.randn(S, M, generator=g).to(BF16)
    dy = (torch.randn(S, M, generator=g) * 0.1).to(BF16)
    wg = torch.randn(E, M, generator=g) * 0.05
    w1, w3 = (torch.randn(E, F, M, generator=g) * 0.03).to(BF16), (torch.randn(E, F, M, generator=g) * 0.03).to(BF16)
    w2 = (torch.randn(E, M, F, generator=g) * 0.03).to(BF16)
    noise = MO.gumbel_noise((S, E), 9)
    lay, out, l_aux, dx, d_wg, d_w1, d_w3, d_w2 = _run_layer(dev, x, dy, wg, w1, w3, w2, noise, 0.5, 4, 0.01)
    xo, wgo = x.clone().requires_grad_(True), wg.clone().requires_grad_(True)
    w1o, w3o, w2o = w1.clone().requires_grad_(True), w3.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    ro, lo, r = MO.moe_layer(xo, wgo, w1o, w3o, w2o, noise, 0.5, 4)
    ((ro.float() * dy.float()).sum() + 0.01 * lo.float()).backward()
    dropped = int((r["slot"] < 0).sum())
    assert dropped > S // 4, "the case is meant to drop a good share of the choices"
    same = (lay.expert.cpu().long() == r["expert"]).float().mean()
    assert float(same) >= 0.999, f"routing agrees on {float(same):.4%} of the choices (fp32 summation order may flip a near-tie)"
    assert abs(l_aux - float(lo)) <= 1e-2 * float(lo)
    _close(out, ro.detach(), "output")
    _close(dx, xo.grad, "d input")
    _close(d_wg, wgo.grad, "d gate weight", 3e-2)
    _close(d_w1, w1o.grad, "d w1")
    _close(d_w3, w3o.grad, "d w3")
    _close(d_w2, w2o.grad, "d w2")


def test_gumbel_noise_is_reproducible_and_gumbel_distributed(dev):
    from internevo_amd import kernels as K

    n = 1 << 20
    a, b = torch.empty(n, device=dev), torch.empty(n, device=dev)
    L = K._L()
    K.check(L.ie_moe_gumbel_noise(K._p(a), n, 7, 0, K._stream()), "noise")
    K.check(L.ie_moe_gumbel_noise(K._p(b), n, 7, 0, K._stream()), "noise")
    assert torch.equal(a, b), "same (seed, offset) -> same noise"
    K.check(L.ie_moe_gumbel_noise(K._p(b), n, 7, n, K._stream()), "noise")
    assert not torch.equal(a, b)
    # Gumbel(0, 1): mean = Euler-Mascheroni, variance = pi^2 / 6
    assert abs(float(a.mean()) - 0.5772) < 5e-3 and abs(float(a.var()) - 1.6449) < 2e-2


def _ep_worker(rank, world, port, q, E=4):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(xport(port)), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from internevo_amd.moe import MoELayer

        g = torch.Generator().manual_seed(17)
        S, M, F = 512, 256, 512
        wg = torch.randn(E, M, generator=g) * 0.05
        w1, w3 = (torch.randn(E, F, M, generator=g) * 0.03).to(BF16), (torch.randn(E, F, M, generator=g) * 0.03).to(BF16)
        w2 = (torch.randn(E, M, F, generator=g) * 0.03).to(BF16)
        gr = torch.Generator().manual_seed(100 + rank)          # every rank has its OWN tokens
        x = torch.randn(S, M, generator=gr).to(BF16)
        dy = (torch.randn(S, M, generator=gr) * 0.1).to(BF16)
        noise = MO.gumbel_noise((S, E), 30 + rank)
        one = _run_layer(dev, x, dy, wg, w1, w3, w2, noise, 1.0, 4, 0.01)[1:]   # all experts local: what this rank's tokens must get
        El = E // world
        mine = slice(rank * El, (rank + 1) * El)
        w13 = torch.cat([w1, w3], dim=1)[mine].contiguous().to(dev)
        w2d, wgd, xd, dyd = w2[mine].contiguous().to(dev), wg.to(dev), x.to(dev), dy.to(dev)

        def run(chunks, overlap):
            lay = MoELayer(M, F, E, S, dev, 1.0, 4, ep_group=dist.group.WORLD, ep_size=world, ep_rank=rank, a2a_chunks=chunks, a2a_overlap=overlap)
            assert lay.nch == (chunks or 2) and lay.overlap == overlap
            out = torch.full((S, M), 7.0, dtype=BF16, device=dev)
            l_aux = lay.forward(xd, wgd, w13, w2d, out, noise=noise.to(dev))
            dx = torch.full((S, M), 7.0, dtype=BF16, device=dev)
            d_wg, d_w13, d_w2 = torch.empty_like(wgd), torch.empty_like(w13), torch.empty_like(w2d)
            lay.backward(dyd, wgd, w13, w2d, dx, d_wg, d_w13, d_w2, accumulate=False, loss_scale_dev=None, aux_factor=0.01)
            torch.cuda.synchronize()
            return lay, [out, l_aux.clone(), dx, d_wg, d_w13, d_w2]

        lay, got = run(None, True)            # the default: two pieces, every exchange under the other piece's expert products
        _, blocking = run(2, False)           # the same pieces, every exchange waited for at once
        lay1, whole = run(1, True)            # one piece: the round-5 layout
        same = [bool(torch.equal(a, b)) for a, b in zip(got, blocking)]
        rowwise = [bool(torch.equal(a, b)) for a, b in zip(got[:4], whole[:4])]   # outputs, l_aux, input and gate gradients do not depend on the piece count
        wdiff = [float((a.float() - b.float()).abs().max() / b.float().abs().max()) for a, b in zip(got[4:], whole[4:])]   # the weight gradients: bf16 sum order
        # the chunk-major rows (ie_moe_chunk_rows) against the formula: slot c of expert e at k E Cn + e Cn + (c mod Cn)
        C, Cn = lay.C, lay.C // 2
        r1 = lay1.row.cpu().long()
        e_, c_ = r1 // C, r1 % C
        want_row = torch.where(r1 >= 0, (c_ // Cn) * E * Cn + e_ * Cn + c_ % Cn, r1)
        t1 = lay1.token_of.cpu()
        want_tok = torch.empty_like(t1)
        rr = torch.arange(E * C)
        want_tok[(rr % C // Cn) * E * Cn + (rr // C) * Cn + rr % C % Cn] = t1
        perm_ok = bool(torch.equal(lay.row.cpu().long(), want_row)) and bool(torch.equal(lay.token_of.cpu(), want_tok))
        out, l_aux, dx, d_wg, d_w13, d_w2 = got
        q.put((rank, [t.float().numpy() if torch.is_tensor(t) else t for t in one],
               [out.float().cpu().numpy(), float(l_aux), dx.float().cpu().numpy(), d_wg.cpu().numpy(), d_w13[:, :F].float().cpu().numpy(),
                d_w13[:, F:].float().cpu().numpy(), d_w2.float().cpu().numpy()], dict(same=same, rowwise=rowwise, wdiff=wdiff, perm_ok=perm_ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("E", [4, 2], ids=["two_experts_per_rank", "one_expert_per_rank"])
def test_expert_parallel_layer_equals_the_all_local_layer(dev, E):
    """Expert parallelism (parallel.expert; gshard_layer.py:453-474: all_to_all of the [E, C, M] dispatch buffer) on two ranks, two of the four
    experts each (or one of two: all source ranks' rows of a piece are then ONE product's operand), every rank routing its OWN tokens: a rank's output,
    input gradient and gate gradient equal the layer that holds all experts locally (same kernels, same routing), and the gradient of an expert's weights is
    the SUM of what the two ranks' tokens contribute to it.  Round 6 -- the exchange in two pieces along the capacity, piece k + 1 on its way under piece k's
    expert products (the reference's exchange is blocking, gshard_layer.py:465-498): the overlapped form equals the same pieces exchanged one after the
    other BIT FOR BIT in every output; against ONE piece the outputs, the auxiliary loss, the input and the gate gradient are bit-identical (row-wise
    work) and the weight gradients differ in their bf16 summation order only; the chunk-major row numbering equals its formula."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ep_worker, args=(r, 2, 29891 + E, q, E)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, one, ep, chk = q.get(timeout=240)
        res[r] = (one, ep)
        assert all(chk["same"]), f"rank {r}: overlapped exchange differs from the blocking one: {chk['same']}"
        assert all(chk["rowwise"]), f"rank {r}: outputs / input gradients depend on the piece count: {chk['rowwise']}"
        assert max(chk["wdiff"]) <= 1e-2, chk["wdiff"]
        assert chk["perm_ok"], "ie_moe_chunk_rows: rows / inverse map differ from the chunk-major formula"
    for p in procs:
        p.join(60)
    El = E // 2
    for r in range(2):
        one, ep = res[r]
        for k, what in ((0, "output"), (2, "d input"), (3, "d gate weight")):
            _close(torch.from_numpy(ep[k]), torch.from_numpy(one[k]), f"rank {r} {what}", 1e-2)
        assert ep[1] == one[1], "auxiliary loss of the rank's own tokens"
        for k, what in ((4, "d w1"), (5, "d w3"), (6, "d w2")):
            for j in range(El):
                e = r * El + j
                want = torch.from_numpy(res[0][0][k][e]) + torch.from_numpy(res[1][0][k][e])
                _close(torch.from_numpy(ep[k][j]), want, f"rank {r} expert {e} {what}", 2e-2)
