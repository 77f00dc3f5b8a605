"""Ring attention (internevo_amd/seqpar.py: RingAttention; BASELINE.json north_star "ring attention send/recv over xGMI").  The reference has no such
mode; what is pinned is the result it must equal -- causal attention over the whole packed sequence, DistributedAttention's result
(multi_head_attention.py:56-135) -- and the block geometry it is built from.

CPU: ring_plan against the brute-force visibility matrix of packed causal attention.
GPU (`-m gpu`): the sequence group as processes on the box's GPU(s) over a comm.Backend (staged gloo on a one-GPU box, RCCL with a GPU per rank):
RingAttention forward + backward against the one-rank kernels on ragged packs incl. sequences that span several ranks, with fewer kv heads than
ranks (the head exchange's limit); and the engine in ring mode against one rank stepping through the same micro-batches."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from test_dp_gpu import backend  # noqa: E402,F401  (the fixture: staged gloo on one GPU, RCCL when the box has a GPU per rank)


def _visible(cu):
    T = cu[-1]
    seq = np.zeros(T, dtype=np.int64)
    for s, (a, b) in enumerate(zip(cu, cu[1:])):
        seq[a:b] = s
    t = np.arange(T)
    return (seq[:, None] == seq[None, :]) & (t[None, :] <= t[:, None])   # [query, key]


@pytest.mark.parametrize("cu,sp", [
    ([0, 64], 2), ([0, 64], 4), ([0, 10, 64], 4), ([0, 31, 33, 64], 4), ([0, 16, 32, 48, 64], 4), ([0, 5, 20, 21, 50, 64], 8),
    ([0, 1, 2, 3, 64], 8), ([0, 63, 64], 8), ([0, 17, 17, 40, 64], 2), ([0, 96], 3),
])
def test_ring_plan_covers_exactly_the_visible_pairs(cu, sp):
    """Own block (causal inside the local boundaries) + one unmasked rectangle per listed earlier block = the visibility matrix of packed causal
    attention, for every rank; blocks the plan skips hold nothing the rank's queries see."""
    from internevo_amd.seqpar import ring_plan

    vis = _visible(cu)
    T = cu[-1]
    Tl = T // sp
    for j in range(sp):
        pl = ring_plan(cu, sp, j)
        assert pl["Tl"] == Tl and len(pl["steps"]) == sp - 1 and pl["cu_local"][0] == 0 and pl["cu_local"][-1] == Tl
        got = np.zeros((Tl, T), dtype=bool)
        loc = _visible(pl["cu_local"])
        got[:, j * Tl : (j + 1) * Tl] = loc
        for s, (r, koff, Lk) in enumerate(pl["steps"], start=1):
            assert r == (j - s) % sp
            if Lk:
                assert r < j and koff + Lk == Tl and pl["Lq"] > 0
                got[: pl["Lq"], r * Tl + koff : (r + 1) * Tl] = True
        assert np.array_equal(got, vis[j * Tl : (j + 1) * Tl]), f"rank {j} of {sp}, boundaries {cu}"
        assert pl["max_local"] == max(b - a for a, b in zip(pl["cu_local"], pl["cu_local"][1:]))
    with pytest.raises(ValueError):
        ring_plan([0, 10], 4, 0)


# ------------------------------------------------------------------------------------------------------------------------------ GPU
def _problem(lens, hq, hkv, d, seed):
    g = torch.Generator().manual_seed(seed)
    T = sum(lens)
    bf = lambda t: t.to(torch.bfloat16)  # noqa: E731
    return (bf(torch.randn(T, hq, d, generator=g)), bf(torch.randn(T, 2, hkv, d, generator=g)), bf(torch.randn(T, hq, d, generator=g) * 0.5),
            [0] + list(np.cumsum(lens)))


def _ring_worker(rank, world, port, qu, lens, hq, hkv, d):
    import torch.distributed as dist
    from test_dp_gpu import _init_dist

    dev = _init_dist(rank, world, port)
    try:
        from internevo_amd.seqpar import RingAttention, SeqParallel

        q, kv, do, cu = _problem(lens, hq, hkv, d, 77)
        Tl = q.shape[0] // world
        sl = slice(rank * Tl, (rank + 1) * Tl)
        sq = SeqParallel(world, rank, world)
        ring = RingAttention(sq, hq, hkv, d, Tl, dev)
        ql, kvl, dol = q[sl].to(dev).contiguous(), kv[sl].to(dev).contiguous(), do[sl].to(dev).contiguous()
        out = torch.empty(Tl, hq, d, dtype=torch.bfloat16, device=dev)
        lse = torch.empty(hq, Tl, dtype=torch.float32, device=dev)
        pl = ring.plan(cu)
        ring.forward(ql, kvl, pl, out, lse)
        dq = torch.full((Tl, hq, d), float("nan"), dtype=torch.bfloat16, device=dev)
        dkv = torch.full((Tl, 2, hkv, d), float("nan"), dtype=torch.bfloat16, device=dev)
        ring.backward(dol, ql, kvl, out, lse, pl, dq, dkv)
        torch.cuda.synchronize()
        qu.put((rank, out.float().cpu().numpy(), lse.cpu().numpy(), dq.float().cpu().numpy(), dkv.float().cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("sp,lens,hq,hkv,d", [
    pytest.param(2, [300, 212], 4, 1, 128, marks=pytest.mark.ranks(2)),                       # one kv head for two ranks; the first sequence ends in rank 1's block
    pytest.param(4, [1024], 4, 2, 128, marks=pytest.mark.ranks(4)),                           # ONE sequence over all four ranks: three rectangles on the last rank
    pytest.param(4, [100, 500, 40, 384], 8, 2, 64, marks=pytest.mark.ranks(4)),               # ragged: a sequence inside one block, two spanning, head dim 64
    pytest.param(8, [700, 61, 3, 1284], 2, 1, 128, marks=pytest.mark.ranks(8)),               # eight ranks, one kv head; a 3-token sequence; blocks that see nothing
], ids=["sp2", "sp4_one_sequence", "sp4_ragged_d64", "sp8_one_kv_head"])
def test_ring_attention_equals_one_rank_attention(dev, backend, sp, lens, hq, hkv, d):  # noqa: F811
    """Forward (out, lse) and backward (dq, dk, dv) of the ring on sp ranks against the one-rank kernels on the whole pack, at the flash tests' bounds
    (the merged rows carry one fp32 merge per block; the gradients' sums are kept in fp32 until the blocks are home)."""
    from internevo_amd import kernels as K
    from test_dp_gpu import _collect

    ctx = mp.get_context("spawn")
    qu = ctx.Queue()
    procs = [ctx.Process(target=_ring_worker, args=(r, sp, 29870 + sp, qu, lens, hq, hkv, d)) for r in range(sp)]
    for p in procs:
        p.start()
    res = sorted(_collect(qu, procs, sp), key=lambda x: x[0])
    for p in procs:
        p.join(60)
    q, kv, do, cu = _problem(lens, hq, hkv, d, 77)
    cud = torch.tensor(cu, dtype=torch.int32, device=dev)
    qd, kvd = q.to(dev), kv.to(dev)
    out, lse = K.flash_attn_fwd(qd, kvd[:, 0], kvd[:, 1], cud, max(lens), None, True)
    dq, dk, dv = K.flash_attn_bwd(do.to(dev), qd, kvd[:, 0], kvd[:, 1], out, lse, cud, max(lens), None, True)
    got = [np.concatenate([r[i] for r in res], axis=(1 if i == 2 else 0)) for i in range(1, 5)]

    def rel(a, b):
        a, b = torch.from_numpy(a).double(), b.double().cpu()
        return float((a - b).norm() / b.norm())

    errs = {"out": rel(got[0], out.float()), "lse": rel(got[1], lse), "dq": rel(got[2], dq.float()), "dk": rel(got[3][:, 0], dk.float()), "dv": rel(got[3][:, 1], dv.float())}
    print(f"ring sp{sp} vs one rank (relative l2):", {k: f"{v:.2e}" for k, v in errs.items()})
    assert all(np.isfinite(g).all() for g in got)
    assert errs["out"] <= 5e-3 and errs["lse"] <= 1e-5 and errs["dq"] <= 5e-3 and errs["dk"] <= 5e-3 and errs["dv"] <= 5e-3, errs
    # ... and DIRECTLY against the oracle (oracle.ops.attention_varlen = the reference's CrossAttention per packed sequence, multi_head_attention.py:195-237,
    # in fp32 on the same bf16 inputs, gradients by autograd), at the flash tests' own bound: not only transitively through the one-rank kernels
    from oracle import ops as O

    qf, kvf = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    ref = O.attention_varlen(qf, kvf, cu, True, None)
    ref.backward(do.float())
    oerr = {"out": rel(got[0], ref.detach()), "dq": rel(got[2], qf.grad), "dk": rel(got[3][:, 0], kvf.grad[:, 0]), "dv": rel(got[3][:, 1], kvf.grad[:, 1])}
    print(f"ring sp{sp} vs the dense fp32 oracle (relative l2):", {k: f"{v:.2e}" for k, v in oerr.items()})
    assert all(v <= 5e-3 for v in oerr.values()), oerr


def _engine_worker(rank, world, port, qu, sp, heads, kv_heads, micro_num):
    import torch.distributed as dist
    from test_dp_gpu import _init_dist

    dev = _init_dist(rank, world, port)
    try:
        from internevo_amd.data import SyntheticLoader
        from internevo_amd.engine import InternLM2Engine
        from oracle.model import formula_init
        from test_multirank_gpu import _sp_cfg

        eng = InternLM2Engine(_sp_cfg(heads, kv_heads, micro_num), dev, None, world, rank, init_fn=formula_init, sp_size=sp)
        assert eng.ring_mode and eng.a_kv[0].shape == (256 // sp, 2, kv_heads, 64)
        loader = iter(SyntheticLoader(256, 1, micro_num, False, 4000, data_rank=eng.seqpar.data_rank, data_world_size=eng.seqpar.data_world))
        out = []
        for _ in range(3):
            batch, labels = next(loader)
            loss = eng.forward_backward(batch, labels)
            eng.step()
            out.append((float(loss), float(eng.read_state().grad_norm)))
        eng.drain()
        qu.put((rank, out, eng.params.float().cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.ranks(4)
def test_engine_with_ring_attention_equals_single_rank_step(dev, backend):  # noqa: F811
    """parallel.tensor = dict(size=4, mode="isp") on a model with TWO kv heads: the head exchange cannot split them over four ranks (Ulysses' limit),
    `attention="auto"` picks the ring.  Three training steps on four ranks must reproduce one rank stepping through the same ragged micro-batches
    (loss 1e-3, global grad norm 2e-2), with ISP's gradient averaging rule and ZeRO-1 over the four ranks."""
    from test_dp_gpu import _collect
    from test_multirank_gpu import _run_one_rank, _sp_cfg

    sp, heads, kv_heads, M = 4, 8, 2, 2
    ctx = mp.get_context("spawn")
    qu = ctx.Queue()
    procs = [ctx.Process(target=_engine_worker, args=(r, sp, 29890, qu, sp, heads, kv_heads, M)) for r in range(sp)]
    for p in procs:
        p.start()
    res = sorted(_collect(qu, procs, sp), key=lambda x: x[0])
    for p in procs:
        p.join(60)
    eng, ref = _run_one_rank(dev, _sp_cfg(heads, kv_heads, M), M, False, emulate_isp_grad_rule=sp)
    p0 = torch.from_numpy(res[0][2])
    for r in res[1:]:
        assert torch.equal(torch.from_numpy(r[2]), p0), f"rank {r[0]} disagrees with rank 0 on the parameters after the all-gather"
    for k in range(3):
        loss, gn = res[0][1][k]
        print(f"step {k}: ring sp{sp} loss {loss:.5f} gn {gn:.4f} | 1 rank loss {ref[k][0]:.5f} gn {ref[k][1]:.4f}")
        for r in res:
            assert r[1][k][0] == loss and abs(r[1][k][1] - gn) <= 1e-6 * gn
        assert abs(loss - ref[k][0]) <= 1e-3 * abs(ref[k][0]) and abs(gn - ref[k][1]) <= 2e-2 * ref[k][1]
