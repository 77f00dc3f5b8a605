"""CPU tests of the flat layout and of the ZeRO-1 exchange (gloo, world_size 2, real processes)."""
import os
import sys

import pytest
from conftest import xport
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flat_layout_invariants():
    from internevo_amd.config import internlm2_7b, tiny
    from internevo_amd.layout import ALIGN, FlatLayout

    for cfg, worlds in ((tiny().model, (1, 2, 8)), (internlm2_7b().model, (1, 8))):
        for w in worlds:
            L = FlatLayout(cfg, w)
            assert len(L.buckets) == cfg.num_layers + 2
            end = 0
            for b in L.buckets:
                assert b.start == end and b.start % ALIGN == 0 and b.size % (w * ALIGN) == 0 and b.used <= b.size
                end = b.start + b.size
            assert end == L.total
            # parameters tile their bucket without overlap
            for b in L.buckets:
                o = b.start
                for n in b.params:
                    s = L.params[n]
                    assert s.offset == o and s.offset % ALIGN == 0
                    o += (s.numel + ALIGN - 1) // ALIGN * ALIGN
            assert sum(s.numel for s in L.params.values()) == cfg.num_params()
            # w1 / w3 adjacency (one [2F, h] GEMM operand)
            a, c = L.params["layers.0.feed_forward.w1.weight"], L.params["layers.0.feed_forward.w3.weight"]
            assert c.offset == a.offset + a.numel
            assert L.local_numel() * w == L.total
    assert internlm2_7b().model.num_params() == 7_737_708_544  # 7.74 B


def test_reference_config_files_map(tmp_path):
    from internevo_amd.config import from_reference_dict, load_reference_config

    p = tmp_path / "cfg.py"
    p.write_text(
        "SEQ=2048\nmodel_type='INTERNLM2_PUBLIC'\n"
        "data=dict(seq_len=SEQ, micro_num=4, micro_bsz=1, total_steps=20)\n"
        "grad_scaler=dict(fp16=dict(initial_scale=2**16, min_scale=1, growth_interval=1000), growth_factor=2, backoff_factor=0.5, max_scale=2**24, hysteresis=2)\n"
        "hybrid_zero_optimizer=dict(overlap_sync_grad=True, overlap_sync_param=False, reduce_bucket_size=512*1024*1024, clip_grad_norm=1.0)\n"
        "loss=dict(label_smoothing=0)\nadam=dict(lr=1e-4, adam_beta1=0.9, adam_beta2=0.95, adam_beta2_c=0, adam_eps=1e-8, weight_decay=0.01)\n"
        "lr_scheduler=dict(total_steps=20, init_steps=0, warmup_ratio=0.01, eta_min=1e-5, last_epoch=-1)\n"
        "model=dict(checkpoint=False, num_attention_heads=32, vocab_size=92544, hidden_size=4096, num_layers=32, no_bias=True, mlp_ratio=3.5,\n"
        "           dtype='torch.bfloat16', layer_norm_epsilon=1e-5, num_kv_attention_heads=8, use_flash_attn=True)\n"
        "parallel=dict(zero1=dict(size=8), tensor=dict(size=1, mode='mtp'), pipeline=dict(size=1), weight=dict(size=1))\n"
    )
    cfg = load_reference_config(str(p), seq_len=4096)
    assert cfg.model.ffn_dim == 14336 and cfg.model.qkv_dim == 6144 and cfg.train.seq_len == 4096 and cfg.train.clip_grad_norm == 1.0
    # configs/7B_isp_sft.py: tensor=dict(size=2, mode="isp"), weight=dict(size=4, overlap=True, memory_pool=True)
    import copy
    import runpy

    g = {k: v for k, v in runpy.run_path(str(p)).items() if not k.startswith("__")}
    isp = copy.deepcopy(g)
    isp["parallel"] = dict(zero1=dict(size=-1), tensor=dict(size=2, mode="isp"), pipeline=dict(size=1, interleaved_overlap=True),
                           weight=dict(size=4, overlap=True, memory_pool=True))
    assert from_reference_dict(isp).train.sp_size == 2
    mtp = copy.deepcopy(g)
    mtp["parallel"] = dict(zero1=dict(size=8), tensor=dict(size=2, mode="mtp"), pipeline=dict(size=1))
    assert from_reference_dict(mtp).train.tp_size == 2 and from_reference_dict(mtp).train.sp_size == 1
    msp = copy.deepcopy(g)
    msp["parallel"] = dict(zero1=dict(size=8), tensor=dict(size=2, mode="msp"), pipeline=dict(size=1))
    pp = copy.deepcopy(g)
    pp["parallel"] = dict(zero1=dict(size=-1), tensor=dict(size=1, mode="mtp"), pipeline=dict(size=4))
    assert from_reference_dict(pp).train.pp_size == 4                       # non-interleaved 1F1B (pipeline.py)
    pp["parallel"]["tensor"] = dict(size=2, mode="mtp")                     # ... together with Megatron tensor parallelism of mode mtp
    assert (from_reference_dict(pp).train.pp_size, from_reference_dict(pp).train.tp_size) == (4, 2)
    pp["parallel"]["tensor"] = dict(size=2, mode="msp")                     # ... with the sequence-sharded tensor modes and with sequence parallelism (round 4)
    t = from_reference_dict(pp).train
    assert (t.pp_size, t.tp_size, t.tp_mode, t.sp_size) == (4, 2, "msp", 1)
    pp["parallel"]["tensor"] = dict(size=2, mode="isp")
    t = from_reference_dict(pp).train
    assert (t.pp_size, t.tp_size, t.sp_size) == (4, 1, 2)
    pp["parallel"]["weight"] = dict(size=2)                                 # ... but not with weight parallelism
    with pytest.raises(NotImplementedError):
        from_reference_dict(pp)
    pp["parallel"].pop("weight")
    pp["parallel"]["tensor"] = 1
    pp["model"]["num_chunks"] = 2                                           # interleaved 1F1B: two model chunks per stage
    pp["data"]["micro_num"] = 8
    assert from_reference_dict(pp).train.num_chunks == 2 and from_reference_dict(pp).train.pp_size == 4
    pp["data"]["micro_num"] = 6                                             # (the reference's own assertions: micro_num % pp, layers % chunks)
    with pytest.raises(ValueError):
        from_reference_dict(pp)
    pp["data"]["micro_num"], pp["model"]["num_chunks"] = 8, 3
    with pytest.raises(ValueError):
        from_reference_dict(pp)
    pp["parallel"]["pipeline"] = dict(size=1)                               # without pipeline stages there is nothing to interleave
    assert from_reference_dict(pp).train.num_chunks == 1
    # Megatron sequence parallelism (msp / fsp): the same parameter shards as mtp, activations between the linears sharded along the sequence (engine.py seq_shard)
    assert from_reference_dict(msp).train.tp_size == 2 and from_reference_dict(msp).train.sp_size == 1
    msp["parallel"]["tensor"]["mode"] = "fsp"
    assert from_reference_dict(msp).train.tp_size == 2
    msp["parallel"]["tensor"]["mode"] = "ring"
    with pytest.raises(NotImplementedError):
        from_reference_dict(msp)
    # settings that would change the arithmetic are refused, not silently ignored
    for path, value in ((("use_fp32_norm",), True), (("model", "norm_type"), "layernorm"), (("model", "apply_post_layer_norm"), True),
                        (("model", "attn_drop_rate"), 0.1),
                        (("model", "multiple_of"), 128), (("data", "rampup_batch_size"), "2 6 5"),
                        (("parallel", "zero1"), dict(size=8, fsdp=True)), (("model", "num_experts"), 4), (("model", "no_bias"), False),
                        (("data", "use_packed_dataset"), False)):
        c = copy.deepcopy(g)
        node = c
        for k in path[:-1]:
            node = node[k]
        node[path[-1]] = value
        with pytest.raises(NotImplementedError):
            from_reference_dict(c)
    # data.skip_batches is honoured (train.py draws the batch and moves on; data.BatchSkipper)
    sk = copy.deepcopy(g)
    sk["data"]["skip_batches"] = "1-3,5"
    assert from_reference_dict(sk).train.skip_batches == "1-3,5" and from_reference_dict(g).train.skip_batches == ""
    # ScaleColumnParallelLinearWithNormHead's options map (InternLM2 family, no pipeline stages)
    nh = copy.deepcopy(g)
    nh["model"].update(embed_grad_scale=0.1, norm_head=True)
    assert from_reference_dict(nh).model.embed_grad_scale == 0.1 and from_reference_dict(nh).model.norm_head is True
    nh["parallel"] = dict(zero1=dict(size=-1), tensor=dict(size=1, mode="mtp"), pipeline=dict(size=2))
    with pytest.raises(NotImplementedError):
        from_reference_dict(nh)
    from internevo_amd.config import ModelConfig
    from internevo_amd.layout import FlatLayout

    half = ModelConfig().tp_shard(2)
    assert (half.num_attention_heads, half.num_kv_attention_heads, half.head_dim, half.ffn_dim, half.qkv_dim) == (16, 4, 128, 7168, 3072)
    L = FlatLayout(half, 4)
    assert L.params["layers.0.attention.wo.weight"].shape == (4096, 2048) and L.params["layers.0.feed_forward.w2.weight"].shape == (4096, 7168)
    assert L.params["layers.0.attention.wqkv.weight"].shape == (3072, 4096) and L.params["tok_embeddings.weight"].shape == (92544, 4096)


def _worker(rank, world, port, q, zero=None):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(xport(port)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from internevo_amd.config import tiny
        from internevo_amd.layout import FlatLayout
        from internevo_amd.zero import ZeroComm

        mc = tiny(hidden=64, layers=2, heads=1, kv_heads=1, vocab=40).model
        z = zero or world          # shards per bucket: the whole data-parallel group, or parallel.zero1.size (hybrid ZeRO)
        L = FlatLayout(mc, z)
        comm = ZeroComm(L, None, world, rank, zero_size=zero)
        assert (comm.world, comm.rank, comm.replica) == (z, rank % z, rank // z)
        dp_rank, dp_world, rank, world = rank, world, comm.rank, comm.world   # below: shard index / shard count
        gen = torch.Generator().manual_seed(100 + dp_rank)
        grads = torch.randn(L.total, generator=gen).to(torch.bfloat16)
        all_grads = [torch.randn(L.total, generator=torch.Generator().manual_seed(100 + r)).to(torch.bfloat16) for r in range(dp_world)]
        mean = sum(g.float() for g in all_grads) / dp_world   # the average over the WHOLE data-parallel group, one or two hops
        for b in reversed(range(len(L.buckets))):
            comm.reduce_bucket_async(grads, b)
        comm.wait_all()
        ok = True
        sq_local = torch.zeros(1)
        for b in L.buckets:
            s, n = b.shard(rank, world)
            ok &= torch.allclose(grads[s : s + n].float(), mean[s : s + n], rtol=1e-2, atol=1e-2)
            sq_local += grads[s : s + n].float().pow(2).sum()
        comm.all_reduce_sum(sq_local)
        # every rank holds the same global squared norm of the averaged gradient
        ref_sq = mean.to(torch.bfloat16).float().pow(2).sum()
        ok &= abs(float(sq_local) - float(ref_sq)) <= 2e-2 * float(ref_sq)
        # parameter all-gather: each rank writes its shard id, afterwards everyone sees every shard
        params = torch.zeros(L.total, dtype=torch.bfloat16)
        for b in L.buckets:
            s, n = b.shard(rank, world)
            params[s : s + n] = rank + 1
            comm.gather_bucket_async(params, b.index)
        comm.wait_all_gathers()   # (the staged test backend lands a result only in wait(), like the stream order on RCCL)
        for b in L.buckets:
            for r in range(world):
                s, n = b.shard(r, world)
                ok &= bool((params[s : s + n] == r + 1).all())
        q.put((dp_rank, ok, float(sq_local)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_hybrid_zero_exchange_gloo_world4_zero2():
    """parallel.zero1.size = 2 on four data-parallel ranks: two shards per bucket, reduce-scatter inside the zero group + all-reduce
    across the replicas = the average over all four ranks; the all-gather stays inside the zero group."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port = 4, 29817
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, 2)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=200) for _ in range(world)]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok, _ in res), res
    assert all(abs(r[2] - res[0][2]) < 1e-6 * abs(res[0][2]) for r in res)


@pytest.mark.timeout(180)
def test_zero_exchange_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port = 2, 29811
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok, _ in res), res
    assert abs(res[0][2] - res[1][2]) < 1e-6 * abs(res[0][2])


def test_checkpoint_fraction_maps_like_the_reference():
    """launch.py:295-303 (True -> 1, False -> 0, else a fraction in [0,1]) and modeling_internlm2.py:857-861,910
    (layer lid is checkpointed iff lid < num_layers * fraction)."""
    from internevo_amd.config import ModelConfig

    for L, frac, want in [(32, 0.0, 0), (32, 1.0, 32), (32, 0.5, 16), (5, 0.5, 3), (4, 0.3, 2), (2, True, 2), (2, False, 0)]:
        assert ModelConfig(num_layers=L, checkpoint=float(frac)).checkpoint_layers == want


def _sp_exchange_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(xport(port)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from internevo_amd.seqpar import SeqParallel

        sp = SeqParallel(2, rank, world)  # world 4 = 2 data-parallel groups of 2 sequence ranks
        Tl, H, d = 6, 4, 8
        gen = torch.Generator().manual_seed(7 + sp.data_rank)     # both ranks of a sequence group derive the same full tensors
        q_full_heads = torch.randn(2 * Tl, H, d, generator=gen)   # [T, H, d]: all tokens, all heads (ground truth)
        kv_full_heads = torch.randn(2 * Tl, 2, H, d, generator=gen)
        lo = sp.sp_rank * Tl
        # what _SeqAllToAll computes (multi_head_attention.py:27-53): scatter heads, gather sequence
        want_q = q_full_heads[:, sp.sp_rank * (H // 2) : (sp.sp_rank + 1) * (H // 2)]
        want_kv = kv_full_heads[:, :, sp.sp_rank * (H // 2) : (sp.sp_rank + 1) * (H // 2)]
        # CPU stand-in for ie_seq_head_permute: [A][B][S][C] -> [S][A][B][C]
        def pack(x, B):
            A = x.shape[0]
            return x.reshape(A, B, 2, -1).permute(2, 0, 1, 3).contiguous()
        got_q = sp.all_to_all(pack(q_full_heads[lo : lo + Tl], 1), torch.empty(2 * Tl, H // 2, d))
        got_kv = sp.all_to_all(pack(kv_full_heads[lo : lo + Tl], 2), torch.empty(2 * Tl, 2, H // 2, d))
        ok = torch.equal(got_q, want_q) and torch.equal(got_kv, want_kv)
        # the inverse exchange brings "all tokens, my heads" back to "my tokens, all heads"
        back = sp.all_to_all(got_q.contiguous(), torch.empty(2, Tl, 1, (H // 2) * d))
        ctx_local = back.permute(1, 2, 0, 3).reshape(Tl, H, d)   # inverse permute [S][A][B][C] -> [A][B][S][C]
        ok = ok and torch.equal(ctx_local, q_full_heads[lo : lo + Tl])
        t = torch.tensor([float(rank + 1)])
        sp.all_reduce_sum(t)  # sums over the sequence group only: ranks (0,1) -> 3, ranks (2,3) -> 7
        ok = ok and float(t) == (3.0 if rank < 2 else 7.0) and (sp.data_rank, sp.data_world) == (rank // 2, 2)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_ulysses_exchange_gloo_world4():
    """SeqParallel's groups (consecutive ranks share a sequence) and its flat all_to_all_single exchange reproduce
    _SeqAllToAll's scatter-heads / gather-sequence result and its inverse, 2 sequence groups x 2 ranks over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sp_exchange_worker, args=(r, 4, 29861, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(4))
    for p in procs:
        p.join(30)
    assert res == [(0, True), (1, True), (2, True), (3, True)], res


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="the reference tree is only mounted in the build container")
def test_shipped_reference_configs_map():
    """The reference's own config files for the covered model families load through load_reference_config unchanged."""
    from internevo_amd.config import load_reference_config

    a = load_reference_config("/root/reference/configs/7B_internlm2.py", seq_len=4096)
    assert (a.model.model_type, a.model.vocab_size, a.model.num_kv_attention_heads, a.train.micro_num, a.model.adapt_hf) == ("INTERNLM2_PUBLIC", 92544, 8, 4, True)
    b = load_reference_config("/root/reference/configs/7B_llama2.py")
    assert (b.model.model_type, b.model.vocab_size, b.model.num_kv_attention_heads, b.model.adapt_hf, b.train.sp_size) == ("LLAMA2", 32000, 8, False, 1)
    # a config without `model_type` is the dense InternLM-1 model (initialize/launch.py:78-79): configs/7B_sft.py loads as that family,
    # configs/7B_isp_sft.py (the same model under tensor mode "isp", size 2) is refused instead of being trained as InternLM2
    c = load_reference_config("/root/reference/configs/7B_sft.py")
    assert (c.model.model_type, c.model.num_experts, c.model.num_kv_attention_heads, c.model.vocab_size) == ("INTERNLM", 1, 32, 103168)
    from internevo_amd.moe_engine import ffn_dim as dense_ffn

    assert dense_ffn(c.model) == 11008   # int(4096 * 8/3) rounded up to a multiple of 256 (modules/mlp.py:52)
    # BASELINE configs[3] as shipped: the same model under tensor = dict(size=2, mode="isp"), weight = dict(size=4) (configs/7B_isp_sft.py:175-180)
    i = load_reference_config("/root/reference/configs/7B_isp_sft.py")
    assert (i.model.model_type, i.model.attn_bias, i.model.num_kv_attention_heads, i.model.ffn_dim, i.train.sp_size, i.train.wp_size, i.train.tp_size,
            i.train.seq_len) == ("INTERNLM", True, 32, 11008, 2, 4, 1, 2048)
    # BASELINE configs[4]: model_type INTERNLM_MoE, 4 experts, top-2 (moe_engine.MoEEngine)
    d = load_reference_config("/root/reference/configs/7B_MoE4_sft.py")
    assert (d.model.model_type, d.model.num_experts, d.model.num_kv_attention_heads, d.model.moe_capacity_factor, d.model.moe_min_capacity,
            d.model.moe_loss_coeff) == ("INTERNLM_MoE", 4, 32, 1.0, 4, 0.1)
    from internevo_amd.moe_engine import ffn_dim

    assert ffn_dim(d.model) == 5632   # int(4096 * 4/3) rounded up to a multiple of 256 (modules/mlp.py:52)


def _tp_group_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(xport(port)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from internevo_amd.tensorpar import TensorParallel

        tp = TensorParallel(2, rank, world)  # ranks (0,1), (2,3) are tensor groups; (0,2), (1,3) hold the same shard
        t = torch.tensor([float(rank + 1)])
        tp.all_reduce_sum(t)
        ok = float(t) == (3.0 if rank < 2 else 7.0)
        d = torch.tensor([float(rank + 1)])
        dist.all_reduce(d, group=tp.dp_group)
        ok = ok and float(d) == (4.0 if rank % 2 == 0 else 6.0) and (tp.tp_rank, tp.dp_rank, tp.dp_world) == (rank % 2, rank // 2, 2)
        full = {"wqkv": torch.arange(24.0).reshape(6, 4), "wo": torch.arange(16.0).reshape(4, 4), "w2": torch.arange(32.0).reshape(4, 8),
                "norm": torch.arange(4.0), "embed": torch.arange(12.0).reshape(3, 4), "head": torch.arange(40.0).reshape(10, 4)}
        for kind, w in full.items():
            mine = tp.shard(kind, w)
            parts = [TensorParallel(1, 0, 1).shard(kind, w)] if kind in ("norm", "embed") else None
            other = type("O", (), {"tp": 2, "tp_rank": 1 - tp.tp_rank, "vocab_parallel": True})()
            theirs = TensorParallel.shard(other, kind, w)
            both = [mine, theirs] if tp.tp_rank == 0 else [theirs, mine]
            ok = ok and torch.equal(TensorParallel.unshard(kind, parts or both), w)
            if kind == "head":   # vocabulary rows: 5 of the 10 per rank (and the whole head when the mode is off)
                ok = ok and mine.shape == (5, 4) and TensorParallel(1, 0, 1).shard(kind, w).shape == (10, 4)
                whole = type("O", (), {"tp": 2, "tp_rank": tp.tp_rank, "vocab_parallel": False})()
                ok = ok and TensorParallel.shard(whole, kind, w).shape == (10, 4) and torch.equal(TensorParallel.unshard(kind, [w, w], False), w)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_tensor_parallel_groups_and_shards_gloo_world4():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tp_group_worker, args=(r, 4, 29867, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(4))
    for p in procs:
        p.join(30)
    assert res == [(0, True), (1, True), (2, True), (3, True)], res


def test_pipeline_partition_and_schedule():
    """partition_uniform (solver/pipeline_utils.py:9-34, num_chunks = 1): L // pp layers per stage, the LAST L % pp stages one more;
    the 1F1B order of a stage (pipeline_scheduler.py:430-560): pp - stage - 1 warm-up forwards, one-forward-one-backward, cool-down;
    the layout of a stage holds its layers under their global numbers, the embedding on the first and norm + head on the last stage."""
    from internevo_amd.config import tiny
    from internevo_amd.layout import FlatLayout
    from internevo_amd.pipeline import partition_uniform, schedule_1f1b

    assert partition_uniform(32, 4) == [(0, 8), (8, 16), (16, 24), (24, 32)]
    assert partition_uniform(10, 4) == [(0, 2), (2, 4), (4, 7), (7, 10)]
    with pytest.raises(ValueError):
        partition_uniform(2, 4)
    assert schedule_1f1b(0, 2, 3) == [("F", 0), ("F", 1), ("B", 0), ("F", 2), ("B", 1), ("B", 2)]
    assert schedule_1f1b(1, 2, 3) == [("F", 0), ("B", 0), ("F", 1), ("B", 1), ("F", 2), ("B", 2)]
    assert schedule_1f1b(0, 4, 2) == [("F", 0), ("F", 1), ("B", 0), ("B", 1)]
    for s in range(4):   # every micro-batch forwarded before it is backwarded, never more than pp - stage in flight
        inflight = worst = 0
        for kind, _ in schedule_1f1b(s, 4, 8):
            inflight += 1 if kind == "F" else -1
            worst = max(worst, inflight)
        assert inflight == 0 and worst == 4 - s
    import dataclasses

    mc = tiny(64, 5, 4, 2, 128, 32, 2).model
    full = FlatLayout(mc, 1)
    names = []
    for st, (lo, hi) in enumerate(partition_uniform(5, 2)):
        L = FlatLayout(dataclasses.replace(mc, num_layers=hi - lo), 2, lo, st == 0, st == 1)
        assert len(L.buckets) == (hi - lo) + 2 and (L.buckets[0].size == 0) == (st != 0) and (L.buckets[-1].size == 0) == (st != 1)
        assert all(L.params[n].shape == full.params[n].shape for n in L.params)
        names += list(L.params)
    assert names == list(full.params)


def test_interleaved_pipeline_partition_order_and_plan():
    """Interleaved 1F1B (model.num_chunks > 1): the layer ranges of partition_uniform (pipeline_utils.py:9-34), the micro-step order of
    InterleavedPipelineScheduler (pipeline_scheduler.py:925-945, 1327-1373) and the tick plan that moves the messages."""
    import os
    import sys

    from internevo_amd.pipeline import interleaved_order, interleaved_plan, partition_chunks

    assert partition_chunks(8, 2, 2) == [[(0, 2), (4, 6)], [(2, 4), (6, 8)]]
    assert partition_chunks(12, 4, 1) == [[(0, 3)], [(3, 6)], [(6, 9)], [(9, 12)]]
    assert partition_chunks(10, 2, 2) == [[(0, 2), (5, 7)], [(2, 5), (7, 10)]]       # 5 layers per chunk over 2 stages: the last stage takes the extra one
    with pytest.raises(ValueError):
        partition_chunks(9, 2, 2)
    ref_root = "/root/reference"
    if os.path.isdir(ref_root):   # the reference's own function (pure python), when the reference is mounted
        import importlib.util
        import types

        stub = types.ModuleType("internlm.utils.logger")
        stub.get_logger = lambda *_a, **_k: None
        saved = {k: sys.modules.get(k) for k in ("internlm", "internlm.utils", "internlm.utils.logger")}
        try:
            sys.modules.update({"internlm": types.ModuleType("internlm"), "internlm.utils": types.ModuleType("internlm.utils"), "internlm.utils.logger": stub})
            spec = importlib.util.spec_from_file_location("_ref_pipeline_utils", os.path.join(ref_root, "internlm/solver/pipeline_utils.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            for n, pp, c in [(8, 2, 2), (32, 4, 2), (30, 4, 2), (24, 4, 3), (12, 4, 1), (10, 2, 2)]:
                assert [list(map(tuple, x)) for x in mod.partition_uniform(n, pp, c)] == partition_chunks(n, pp, c), (n, pp, c)
        finally:
            for k, v in saved.items():
                if v is None:
                    sys.modules.pop(k, None)
                else:
                    sys.modules[k] = v

    # the reference's order, spelled out for stage 0 of 2 stages x 2 chunks x 4 micro-batches: 2 + 2 warm-up forwards, pairs, cool-down
    assert interleaved_order(0, 2, 2, 4) == [("F", 0, 0), ("F", 1, 0), ("F", 0, 1), ("F", 1, 1), ("F", 2, 0), ("B", 0, 1), ("F", 3, 0), ("B", 1, 1),
                                             ("F", 2, 1), ("B", 0, 0), ("F", 3, 1), ("B", 1, 0), ("B", 2, 1), ("B", 3, 1), ("B", 2, 0), ("B", 3, 0)]
    assert [k for k, _, _ in interleaved_order(1, 2, 2, 2)] == ["F"] * 4 + ["B"] * 4          # micro_num == pp: everything is warm-up
    with pytest.raises(ValueError):
        interleaved_order(0, 4, 2, 6)
    for pp, chunks, micro in [(2, 2, 2), (2, 2, 4), (4, 2, 8), (4, 3, 4), (2, 4, 6), (3, 2, 9)]:
        plan = interleaved_plan(pp, chunks, micro)
        ticks = len(plan[0])
        assert all(len(p) == ticks for p in plan)                                  # every stage makes the same number of exchanges
        last_v = pp * chunks - 1
        for s in range(pp):
            ops = [t["op"] for t in plan[s] if t["op"]]
            assert ops == interleaved_order(s, pp, chunks, micro)                  # the reference's order, only delayed
            arrived = set()
            for t in plan[s]:
                if t["op"]:
                    kind, m, c = t["op"]
                    v = c * pp + s
                    if (kind == "F" and v > 0) or (kind == "B" and v < last_v):
                        assert (kind, m, c) in arrived, (pp, chunks, micro, s, t["op"])   # an input is received in an EARLIER tick's exchange
                    if kind == "B":
                        assert ("F", m, c) in {o for o in ops[: ops.index(t["op"])]}      # the backward of a chunk follows its forward
                arrived |= {(k, m, c) for k, m, c, _ in t["recvs"]}
            for c in range(chunks):    # gradients accumulate over the micro-batches in ascending order, as without a pipeline
                assert [m for k, m, cc in ops if k == "B" and cc == c] == list(range(micro))
        for t in range(ticks):         # a send and its receive are in the same tick's exchange, between ring neighbours
            sends = sorted((k, m, c, s, to) for s in range(pp) for k, m, c, to in plan[s][t]["sends"])
            recvs = sorted((k, m, c, frm, s) for s in range(pp) for k, m, c, frm in plan[s][t]["recvs"])
            assert sends == recvs
            assert all(abs(a - b) in (1, pp - 1) for _, _, _, a, b in sends)
        assert ticks < 2 * micro * chunks + 2 * pp * chunks                        # bubble bounded by the fill / drain of the ring
    for pp in range(2, 9):             # the replay always completes (a stage whose input has not arrived idles, it never blocks the others)
        for chunks in (2, 3, 4):
            for k in (1, 2, 3):
                assert len(interleaved_plan(pp, chunks, pp * k)[0]) >= 2 * pp * k * chunks


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="the reference tree is not mounted")
def test_pure_python_style_config_loads_without_the_reference_installed(tmp_path):
    """configs/demo.py opens with `from internlm.utils.utils import read_base` / `with read_base(): from configs._base_... import *`
    (demo.py:2-6).  load_reference_config must resolve it in a process where `internlm` is NOT importable (round-2 review), and map the
    ISP config's parallel section (tensor 2 / isp, weight 4)."""
    import subprocess

    isp = "/root/reference/configs/7B_isp_sft.py"   # (no model_type: the dense InternLM-1 model, launch.py:78-79)
    code = ("import sys, importlib.util; sys.path.insert(0, %r); assert importlib.util.find_spec('internlm') is None; "
            "from internevo_amd.config import load_reference_config as L; c = L('/root/reference/configs/demo.py'); "
            "i = L(%r); "
            "print(c.model.num_layers, c.model.hidden_size, c.model.vocab_size, i.model.model_type, i.train.sp_size, i.train.wp_size, 'internlm' in sys.modules)") % (ROOT, isp)
    out = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), capture_output=True, text=True, check=True).stdout.split()
    assert out == ["32", "4096", "92544", "INTERNLM", "2", "4", "False"], out


def test_batch_skipper_equals_the_reference_class():
    """data.BatchSkipper against the real utils/common.py BatchSkipper (tests/golden/skipper.json, make_golden.py --skipper): the parsed spans and which of the
    batch counts 0..29 are skipped, for an empty string, a single count, intervals and open tails; descending intervals are refused as the reference asserts."""
    import json
    import os

    from internevo_amd.data import BatchSkipper

    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for case in json.load(open(os.path.join(G, "skipper.json"))):
        sk = BatchSkipper(case["skip_batches"])
        assert list(sk.spans) == case["spans"] and [n for n in range(30) if sk(n)] == case["skipped"], case["skip_batches"]
    with pytest.raises(AssertionError):
        BatchSkipper("5-7,2")


def test_beta2_scheduler_equals_the_reference_class():
    """schedule.Beta2Scheduler against the real solver/schedulers/beta2_scheduler.py class driving a torch AdamW (tests/golden/beta2.json, make_golden.py --beta2): the
    beta2 the optimizer step k uses, for c = 0 (every shipped config), the class default c = 0.8 over 60 steps (it leaves 0.95 at step 43) and a small init value."""
    import json
    import os

    from internevo_amd.schedule import Beta2Scheduler

    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for case in json.load(open(os.path.join(G, "beta2.json"))):
        sch = Beta2Scheduler(case["init_beta2"], case["c"])
        got = []
        for _ in case["beta2_at_step"]:
            got.append(sch.beta2())
            sch.step()
        assert got == case["beta2_at_step"], (case["init_beta2"], case["c"])
    assert any(b > 0.95 for b in json.load(open(os.path.join(G, "beta2.json")))[1]["beta2_at_step"])

