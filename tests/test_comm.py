"""comm.py on CPU (gloo): the staged test backend keeps the contract the engine's wait() placement is written against, and the zero /
replica groups of hybrid ZeRO are right for EVERY data-parallel group of a job (pipeline stages, tensor ranks)."""
import os
import sys

import pytest
from conftest import xport
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spawn(fn, world, port, *args, timeout=200):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=fn, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=timeout) for _ in range(world)]
    finally:
        for p in procs:
            p.join(30)
    return sorted(res, key=lambda x: x[0])


def _staged_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(xport(port)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from internevo_amd.comm import StagedGlooBackend, backend_for

        be = backend_for(None)
        ok = isinstance(be, StagedGlooBackend)
        # all-reduce: the destination keeps its old value until wait()
        t = torch.full((8,), float(rank + 1), dtype=torch.bfloat16)
        w = be.all_reduce(t, None)
        ok &= bool((t == rank + 1).all())
        w.wait()
        ok &= bool((t == 3.0).all())
        # in-place reduce-scatter(AVG) into the owner's slice of the bucket, then in-place all-gather from it (zero.py's two calls)
        full = torch.arange(8, dtype=torch.float32).to(torch.bfloat16) * (rank + 1)
        shard = full[rank * 4 : (rank + 1) * 4]
        before = full.clone()
        w = be.reduce_scatter(shard, full, None, avg=True)
        ok &= bool(torch.equal(full, before))
        w.wait()
        want = (torch.arange(8, dtype=torch.float32) * 1.5)[rank * 4 : (rank + 1) * 4]
        ok &= bool(torch.equal(shard.float(), want)) and bool(torch.equal(full[(1 - rank) * 4 : (2 - rank) * 4], before[(1 - rank) * 4 : (2 - rank) * 4]))
        w = be.all_gather(full, shard, None)
        w.wait()
        ok &= bool(torch.equal(full.float(), torch.arange(8, dtype=torch.float32) * 1.5))
        # all-to-all: chunk r of send -> rank r
        send = torch.tensor([10.0 * rank, 10.0 * rank + 1])
        recv = torch.full((2,), -1.0)
        w = be.all_to_all(recv, send, None)
        ok &= bool((recv == -1).all())
        w.wait()
        ok &= bool(torch.equal(recv, torch.tensor([float(rank), 10.0 + rank])))
        # paired send + receive (pipeline.py): the receive buffer is written in wait()
        peer = 1 - rank
        out, inp = torch.full((3,), float(rank)), torch.full((3,), -1.0)
        w = be.exchange([(out, peer)], [(inp, peer)])
        ok &= bool((inp == -1).all())
        w.wait()
        ok &= bool((inp == peer).all())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_staged_backend_lands_results_only_in_wait():
    res = _spawn(_staged_worker, 2, 29741)
    assert all(ok for _, ok in res), res


def _hybrid_worker(rank, world, port, q, pp, tp, zero):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(xport(port)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from internevo_amd.config import tiny
        from internevo_amd.layout import FlatLayout
        from internevo_amd.pipeline import PipeParallel
        from internevo_amd.tensorpar import TensorParallel
        from internevo_amd.zero import ZeroComm, job_dp_groups

        if pp > 1:
            par = PipeParallel(pp, rank, world)
            key = par.stage
        else:
            par = TensorParallel(tp, rank, world)
            key = par.tp_rank
        dp_world, dp_rank = par.dp_world, par.dp_rank
        groups = job_dp_groups(world, tp=tp, pp=pp)
        members = dist.get_process_group_ranks(par.dp_group)
        ok = members in groups and members[dp_rank] == rank
        L = FlatLayout(tiny(hidden=64, layers=1, heads=1, kv_heads=1, vocab=40).model, zero)
        comm = ZeroComm(L, par.dp_group, dp_world, dp_rank, zero_size=zero, dp_groups=groups)
        ok &= (comm.world, comm.rank, comm.replica, comm.n_replica) == (zero, dp_rank % zero, dp_rank // zero, dp_world // zero)
        ok &= dist.get_process_group_ranks(comm.group) == members[(dp_rank // zero) * zero : (dp_rank // zero + 1) * zero]
        ok &= dist.get_process_group_ranks(comm.replica_group) == members[dp_rank % zero :: zero]
        # gradients: every data-parallel group averages ITS OWN ranks' gradients (seeded by the global rank)
        grads = torch.randn(L.total, generator=torch.Generator().manual_seed(100 + rank)).to(torch.bfloat16)
        mean = sum(torch.randn(L.total, generator=torch.Generator().manual_seed(100 + r)).to(torch.bfloat16).float() for r in members) / dp_world
        for b in reversed(range(len(L.buckets))):
            comm.reduce_bucket_async(grads, b)
        comm.wait_all()
        for b in L.buckets:
            s, n = b.shard(comm.rank, comm.world)
            ok &= bool(torch.allclose(grads[s : s + n].float(), mean[s : s + n], rtol=1e-2, atol=1e-2))
        # parameters: all-gather inside the zero group only
        params = torch.zeros(L.total, dtype=torch.bfloat16)
        for b in L.buckets:
            s, n = b.shard(comm.rank, comm.world)
            params[s : s + n] = 100 * key + comm.rank + 1
            comm.gather_bucket_async(params, b.index)
        comm.wait_all_gathers()
        for b in L.buckets:
            for r in range(comm.world):
                s, n = b.shard(r, comm.world)
                ok &= bool((params[s : s + n] == 100 * key + r + 1).all())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(400)
@pytest.mark.parametrize("pp,tp", [(2, 1), (1, 2)], ids=["pipeline_stages", "tensor_ranks"])
def test_hybrid_zero_groups_for_every_data_parallel_group_of_the_job(pp, tp):
    """8 ranks = 2 model-parallel positions x 4 data-parallel ranks, parallel.zero1.size = 2: the second pipeline stage (consecutive ranks
    4..7) and the second tensor rank (ranks 1, 3, 5, 7) get their OWN zero and replica groups (round-2 advice: the stride heuristic
    left stage 1 with the whole data-parallel group and shards sized for two)."""
    res = _spawn(_hybrid_worker, 8, 29751 + pp, pp, tp, 2, timeout=300)
    assert all(ok for _, ok in res), res


def _rows_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(xport(port)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from internevo_amd.tensorpar import TensorParallel

        tp = TensorParallel(2, rank, world)
        T, C = 8, 4
        rl = tp.rows(T)
        ok = rl == slice(rank * 4, rank * 4 + 4)
        # a row-parallel product's partial sums -> summed into this rank's rows, in place; the other rows keep the partial (dead) values until wait()
        part = (torch.arange(T * C, dtype=torch.float32).view(T, C) * (rank + 1)).to(torch.bfloat16)
        before = part.clone()
        w = tp.reduce_scatter_rows_async(part)
        ok &= bool(torch.equal(part, before))
        w.wait()
        ok &= bool(torch.equal(part[rl].float(), (torch.arange(T * C, dtype=torch.float32).view(T, C) * 3)[rl]))
        # every rank's rows -> the whole tensor, in place
        full = torch.zeros(T, C, dtype=torch.bfloat16)
        full[rl] = rank + 1
        tp.all_gather_rows_async(full).wait()
        ok &= bool((full[:4] == 1).all() and (full[4:] == 2).all())
        # a norm weight's gradient: AVERAGED over the tensor group
        gsum = torch.full((C,), float(rank + 1))
        tp.all_reduce_avg(gsum)
        ok &= bool((gsum == 1.5).all())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_sequence_sharded_activation_exchanges_of_the_tensor_group():
    """tensorpar.rows / reduce_scatter_rows_async / all_gather_rows_async / all_reduce_avg (the msp / fsp modes) on two gloo ranks."""
    res = _spawn(_rows_worker, 2, 29761)
    assert all(ok for _, ok in res), res


def _pp_tp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(xport(port)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from internevo_amd.pipeline import PipeParallel
        from internevo_amd.tensorpar import TensorParallel
        from internevo_amd.zero import job_dp_groups

        pp, tp = 2, 2
        pipe = PipeParallel(pp, rank, world)                       # stages = blocks of world / pp consecutive ranks
        tpar = TensorParallel(tp, pipe.dp_rank, pipe.dp_world, stages=pp, stage=pipe.stage)
        per = world // pp
        base, local = pipe.stage * per, rank % per
        ok = dist.get_process_group_ranks(tpar.group) == [base + (local // tp) * tp + i for i in range(tp)]        # consecutive ranks inside the stage
        ok &= dist.get_process_group_ranks(tpar.dp_group) == [base + r for r in range(local % tp, per, tp)]         # same shard, same stage
        ok &= dist.get_process_group_ranks(tpar.dp_group) in job_dp_groups(world, tp=tp, pp=pp)
        ok &= dist.get_process_group_ranks(pipe.group) == [local, per + local]                                       # same position in every stage
        ok &= (tpar.tp_rank, tpar.dp_rank, tpar.dp_world) == (local % tp, local // tp, per // tp)
        t = torch.full((2,), float(rank))
        tpar.all_reduce_sum(t)
        ok &= bool((t == sum(dist.get_process_group_ranks(tpar.group))).all())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(400)
def test_tensor_groups_inside_pipeline_stages():
    """pipeline 2 x tensor 2 x data 2 on 8 gloo ranks: tensor groups and same-shard data-parallel groups live INSIDE a stage, pipelines connect equal
    positions of the stages (parallel_context.py: tensor innermost, then data, then pipeline), and job_dp_groups lists exactly those data-parallel groups."""
    res = _spawn(_pp_tp_worker, 8, 29771, timeout=300)
    assert all(ok for _, ok in res), res
