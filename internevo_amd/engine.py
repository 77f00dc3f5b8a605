"""The MI355X training-step engine: InternLM2 forward / backward / hybrid-ZeRO AdamW on hand-written
HIP kernels, one process per GPU.

It is the host-side mirror of the reference's hot path:
  PackedFlashLlama1D.forward            internlm/model/modeling_internlm2.py:966-1009   -> _forward_micro
  PackedFlashLlamaLayer1D._forward      :684-740                                        -> per-layer block
  MHA._packed_forward                   :404-478                                        -> wqkv / rotary / flash / wo
  FeedForward.forward                   modules/mlp.py:82-86                            -> w13 GEMM / SwiGLU / w2
  FlashGPTLMLoss.forward                losses/ce_loss.py:42-58                         -> fused CE on bf16 logits
  NonPipelineScheduler.forward_backward_step  core/scheduler/no_pipeline_scheduler.py:163-239 -> forward_backward
  HybridZeroOptimizer.backward/step/_step     solver/optimizer/hybrid_zero_optim.py:592-807   -> step

MI355X-first differences that do not change results beyond rounding order:
  * no autograd: the backward is an explicit reverse sweep over pre-allocated activation buffers
    (288 GB HBM holds params + grads + fp32 state + all activations of a micro-batch, so by default nothing
    is recomputed except the SwiGLU product; `model.checkpoint` layers are replayed from their input);
  * parameters and gradients live in single flat bf16 buffers (layout.py); weight-gradient GEMMs
    accumulate straight into the flat gradient buffer (bf16 `+=`, as autograd's AccumulateGrad does);
  * the loss scale, overflow check, clip factor and Adam step counter live on the device
    (IeStepState): a training step has no host synchronisation at all;
  * ZeRO-1 uses reduce-scatter(AVG) of gradient buckets + all-gather of updated bf16 shards
    (half the bytes of the reference's all-reduce + broadcast), launched per layer bucket as soon as
    the bucket's last weight-gradient GEMM is queued on the last micro-batch, overlapping backward.
"""
import dataclasses
import os
import math

import torch

from . import kernels as K
from ._lib import IeScalerConfig
from .config import PathConfig
from .layout import FlatLayout
from .schedule import Beta2Scheduler, CosineWarmupLR
from .seqpar import SeqParallel
from .pipeline import PipeParallel, interleaved_plan, partition_chunks, partition_uniform
from .tensorpar import TensorParallel
from .zero import ZeroComm, job_dp_groups

BF16 = torch.bfloat16
_LN2, _LOG2E = 0.6931471805599453, 1.4426950408889634


_HIP_RT = None


def _mem_budget(device):
    """Bytes the engine may still plan with when it decides its two memory-dependent switches (batched weight gradients, the kept SwiGLU product).  Default:
    what the device reports free now.  IE_MEM_BUDGET=own (tests/conftest.py: several test processes share one device): the device's total memory minus what
    THIS process has reserved -- the decision then depends on the engine's own sizes only, and a configuration that does not fit beside its neighbours
    fails with an out-of-memory error instead of silently running the other mode."""
    if os.environ.get("IE_MEM_BUDGET") == "own":
        return torch.cuda.get_device_properties(device).total_memory - torch.cuda.memory_reserved(device)
    return torch.cuda.mem_get_info(device)[0]


_SUMMED = object()   # _layer_forward's return value when the layer has added its residual in w2's epilogue (the sum is where the caller said)


def _optimizer_stream(device):
    """The optimizer's HIP stream: an ordinary stream.
    Round 5 tried a stream whose kernels may only use n of the 256 CUs (hipExtStreamCreateWithCUMask, n / 8 CUs of every XCD): 32 / 64 / 96 CUs -> 722-726 ms per
    step against 671-674 ms unmasked (profiles/r05_adamw_cu_mask_ab.log) -- the same loss whatever n: not what fewer CUs for the update should do, the masked queue
    itself cost the step.  Round 6 restricts the update by its LAUNCH SHAPE instead (ie_tune_adamw_cus: n workgroups, each alone on a CU; InternLM2Engine.adamw_cus).
    IE_ADAMW_STREAM_PRIORITY=low | high (A/B switch, round 6): the stream at the end of hipDeviceGetStreamPriorityRange -- six A B pairs on two boxes: level (mean +0.1 ms
    of 655), as is IE_SERIAL_ADAMW=1 against the side stream at round 6's kernel speeds (profiles/r06_step_adamw_stream_ab.log).  Default: an ordinary stream."""
    global _HIP_RT
    prio = os.environ.get("IE_ADAMW_STREAM_PRIORITY")   # (A/B switch, round 6: "low" / "high" = the ends of hipDeviceGetStreamPriorityRange)
    if prio and device.type == "cuda":
        import ctypes

        if _HIP_RT is None:
            _HIP_RT = ctypes.CDLL("libamdhip64.so")
        least, greatest, st = ctypes.c_int(), ctypes.c_int(), ctypes.c_void_p()
        with torch.cuda.device(device):
            rc = _HIP_RT.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest))
            rc = rc or _HIP_RT.hipStreamCreateWithPriority(ctypes.byref(st), ctypes.c_uint32(1), ctypes.c_int(least.value if prio == "low" else greatest.value))   # 1 = non-blocking
        if rc != 0 or not st.value:
            raise RuntimeError(f"hipStreamCreateWithPriority({prio}) failed with {rc}")
        print(f"[internevo_amd] optimizer stream priority {prio} ({least.value if prio == 'low' else greatest.value} of [{greatest.value}, {least.value}])", flush=True)
        return torch.cuda.ExternalStream(st.value, device=device)
    return torch.cuda.Stream(device=device)


class InternLM2Engine:
    def __init__(self, cfg: PathConfig, device, process_group=None, world_size=1, rank=0, init="normal", seed=1024, init_fn=None,
                 force_collectives=False, sp_size=None, emulate_isp_grad_rule=1, tp_size=None, batch_wgrad=None, merge_micro=None,
                 zero_size=None, vocab_parallel=None, pp_size=None, num_chunks=None, weight_parallel=None, scale_on_q=None, tp_mode=None, rs_under_w13=None,
                 sp_attention=None):
        """zero_size (default: the config's parallel.zero1.size): hybrid ZeRO -- the fp32 state is sharded over groups of zero_size
        consecutive data-parallel ranks and replicated across the groups (zero.py); -1 / None-and-unset = the whole data-parallel group.
        sp_size > 1: Ulysses / ISP sequence parallelism over groups of sp_size consecutive ranks (seqpar.py).
        sp_attention (default: the config's parallel.tensor.attention, "auto"): "ulysses" = the reference's head exchange around the attention,
        "ring" = seqpar.RingAttention (K / V blocks travel around the group; any kv head count), "auto" = ulysses where the kv heads divide, else ring.
        emulate_isp_grad_rule = n on a run WITHOUT sequence parallelism applies the gradient averaging rule of an sp = n ISP
        run (test hook: an sp = n run must then match it step for step).
        tp_mode (default: the config's parallel.tensor mode): "msp" / "fsp" = tensor parallelism with the activations BETWEEN the linears sharded
        along the sequence (model/utils.py:228-463, ops/linear.py:260-354): the residual stream, the norms and the residual adds live on this
        rank's T / tp token rows; a norm's output is all-gathered in front of the column-parallel product, a row-parallel product's partial sums are
        reduce-scattered, the backward mirrors it, and the norm weights' gradients (each rank's cover its own rows) are AVERAGED over the tensor
        group as the reference does (hybrid_zero_optim.py:315-353, ReduceOp.AVG) -- pinned on tests/golden/train_msp2_*.json.
        vocab_parallel (tensor parallelism only; default on): every tensor rank holds 1/tp of the head's vocabulary rows and the loss is
        computed vocabulary-parallel (tensorpar.py); False = the whole head on every rank.
        pp_size (default: the config's parallel.pipeline.size): pipeline parallelism, non-interleaved 1F1B (pipeline.py): this rank
        holds one contiguous range of layers (+ the embedding on the first stage, + norm / head / loss on the last).
        weight_parallel (ISP's weight parallelism, parallel.weight = dict(size=wp); core/communication/isp.py:31-526, ops/linear.py:357-378):
        True = every rank keeps only its 1/wp part of each LAYER's bf16 weights and gradients (wp = the zero group: cfg.train.wp_size, or the
        whole data-parallel group); a layer's weights are all-gathered into a two-slot pool right before the layer runs, forward and
        backward, the next layer's gather running on a side stream under this layer's kernels, and its weight gradients are
        reduce-scattered (AVG) out of a pool slot every micro-batch and accumulated in the resident shard.  None = automatic: on only when
        the config asks for wp > 1 AND the resident layout (weights whole on every GPU) does not fit this GPU's memory; False = resident.
        rs_under_w13 (default off; IE_RS_UNDER_W13=1): a layer bucket's gradient reduce-scatter is launched in front of the NEXT layer's w1 | w3 backward
        products (28 rounds of tiles each at the 7B shapes) instead of right behind the layer's last weight gradient, where it first meets the 4- and
        6-round wo / wqkv products of the layer below: a collective that holds a few CUs costs every product it overlaps one more round of tiles
         -- an A/B switch for the first multi-GPU runs, same results.
        scale_on_q (default off): the rotary kernel stores q already multiplied by softmax_scale * log2 e (ie_qkv_rotary_fwd_scaled) and
        attention runs at softmax_scale = ln 2, which lets the forward take the folded-softmax kernel (+7 % on the 4 x 4096 attention call,
        profiles/r03_flash_fwd_folded.md).  Same mathematics, but q is rounded to bf16 AFTER the scale instead of before it as the reference
        does: the attention inputs are no longer bit-identical to the reference's, and the tiny-model trajectories drift from the reference
        runs 3-4x faster (3e-3 instead of 7e-4 in the loss after six steps, measured) -- outside the 1e-3 parity bound, hence opt-in."""
        self.cfg = cfg
        self.mc, self.tc = cfg.model, cfg.train
        self.dev = device
        self.world, self.rank = world_size, rank
        self.job_world, self.job_rank = world_size, rank   # (self.world / self.rank end up counting the ZeRO shards)
        mc, tc = self.mc, self.tc
        if mc.head_dim not in (64, 128):
            raise NotImplementedError("head dim must be 64 or 128")
        K._L()  # fail loudly now if libinternevo_hip.so is missing
        tp_size = int(tc.tp_size if tp_size is None else tp_size)  # default: the config's parallel.tensor size (mode "mtp")
        sp_size = int(tc.sp_size if sp_size is None else sp_size)  # default: the config's parallel.tensor size (mode "isp")
        if tp_size > 1 and sp_size > 1:
            raise NotImplementedError("tensor parallelism and sequence parallelism are alternatives (parallel.tensor has ONE mode)")
        pp_size = int(getattr(tc, "pp_size", 1) if pp_size is None else pp_size)
        tp_ss = tp_size > 1 and (getattr(tc, "tp_mode", "mtp") if tp_mode is None else tp_mode) in ("msp", "fsp")
        if pp_size > 1 and mc.checkpoint_layers:
            raise NotImplementedError("pipeline parallelism without activation checkpointing (a stage keeps the activation sets of its in-flight micro-batches)")
        self.pp = pp_size
        # every data-parallel group of the job (hybrid ZeRO creates its sub-groups collectively over all of them); a caller-supplied
        # process group is taken as the job's only one
        # (under tensor / pipeline parallelism the engine derives the per-shard data-parallel groups itself, whatever group the caller handed in: every
        # rank of the job must then create the SAME list of hybrid-ZeRO sub-groups -- dist.new_group is collective over the default group)
        dp_groups = job_dp_groups(world_size, tp=tp_size, pp=pp_size) if (process_group is None or tp_size > 1 or pp_size > 1) else None
        self.pipe = PipeParallel(pp_size, rank, world_size)
        # this stage's layers in the reference's numbering: one range, or one range per model chunk (interleaved schedule); `chunks` = the
        # same ranges in local layer indices, `gid[l]` = the global number of local layer l
        nch = int(getattr(tc, "num_chunks", 1) if num_chunks is None else num_chunks) if pp_size > 1 else 1
        ranges = partition_chunks(mc.num_layers, pp_size, nch)[self.pipe.stage] if nch > 1 else [partition_uniform(mc.num_layers, pp_size)[self.pipe.stage]]
        self.gid = [l for lo, hi in ranges for l in range(lo, hi)]
        self.chunks, n0 = [], 0
        for lo, hi in ranges:
            self.chunks.append((n0, n0 + hi - lo))
            n0 += hi - lo
        self.nch = nch
        if pp_size > 1:   # data parallelism and ZeRO-1 run inside a stage
            process_group, world_size, rank = self.pipe.dp_group, self.pipe.dp_world, self.pipe.dp_rank
            self.world, self.rank = world_size, rank
            merge_micro, batch_wgrad = False, False   # the 1F1B schedule works on single micro-batches
        self.tpar = TensorParallel(tp_size, rank, world_size, vocab_parallel=True if vocab_parallel is None else vocab_parallel,
                                   embed_split=getattr(mc, "embed_split_hidden", False), stages=pp_size, stage=self.pipe.stage)
        self.embed_split = self.tpar.embed_split
        self.tp = tp_size
        self.rs_under_w13 = bool(int(os.environ.get("IE_RS_UNDER_W13", "0")) if rs_under_w13 is None else rs_under_w13)
        # the weight-gradient products' tail k-split (kernels.enable_wgrad_ksplit; IE_WGRAD_KSPLIT=0: A/B switch): a process-wide setting of the library
        if device.type == "cuda":
            K.enable_wgrad_ksplit(device, os.environ.get("IE_WGRAD_KSPLIT", "1") != "0")
        # DIAGNOSTIC (bench.py --hold-cus n; one rank only): where a data-parallel run would launch a bucket's reduce-scatter / all-gather, launch instead n idle
        # workgroups that each occupy a CU for the time the collective would take at `link` GB/s per xGMI link (a bucket's 1 / 8 per link), on a side stream as
        # RCCL's kernels run -- prices what the CUs a collective holds cost the products beside it.  No effect on any result.
        self.hold = None
        if os.environ.get("IE_HOLD_CUS") and world_size == 1:
            n_, link_ = (os.environ["IE_HOLD_CUS"].split(",") + ["100"])[:2]
            self.hold = (int(n_), float(link_))
            self._hold_rs, self._hold_ag = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
        self._rs_deferred = None
        self.bias = bool(getattr(mc, "attn_bias", False))   # the InternLM-1 block: Wqkv / out_proj with bias (model_type INTERNLM)
        self.ss = tp_ss   # sequence-sharded activations
        self.vp = self.tpar.vocab_parallel   # vocabulary-parallel head + loss
        self.lmc = mc.tp_shard(tp_size, self.vp)   # what this rank holds / computes: 1/tp of the heads, of the FFN width and of the head's rows
        if tp_size > 1 and not self.embed_split and self.lmc.embed_dim != mc.hidden_size:
            self.lmc = dataclasses.replace(self.lmc, embed_dim_override=None)
        if pp_size > 1:
            self.lmc = dataclasses.replace(self.lmc, num_layers=len(self.gid))
        if tp_size > 1:
            # data parallelism and ZeRO-1 run over the ranks that hold the same shard
            process_group, world_size, rank = self.tpar.dp_group, self.tpar.dp_world, self.tpar.dp_rank
            self.world, self.rank = world_size, rank
        # data-parallel group (dp_world ranks) -> ZeRO shards: `self.world` / `self.rank` count the SHARDS of the optimizer state
        # (= the data-parallel group unless parallel.zero1.size asks for hybrid ZeRO)
        self.dp_world, self.dp_rank = world_size, rank
        zs = tc.zero1_size if zero_size is None else zero_size
        zs = world_size if (zs is None or zs <= 0 or zs >= world_size) else int(zs)
        if world_size % zs:
            raise ValueError(f"parallel.zero1.size = {zs} must divide the data-parallel size {world_size}")
        wp_cfg = int(getattr(tc, "wp_size", 1) or 1)
        if weight_parallel is None:   # resident weights are the faster design: shard only what does not fit (static sizes: every rank decides alike)
            full = FlatLayout(self.lmc, 1, self.gid, self.pipe.first, self.pipe.last).total
            total = torch.cuda.get_device_properties(device).total_memory if device.type == "cuda" else 0
            weight_parallel = wp_cfg > 1 and total > 0 and 4 * full + 12 * full / max(zs, 1) > 0.7 * total
        self.wp_mode = bool(weight_parallel)
        if self.wp_mode:
            if tp_size > 1 or pp_size > 1 or (mc.embed_grad_scale != 1.0 or mc.norm_head):
                raise NotImplementedError("weight parallelism combines with data / sequence parallelism (the ISP configuration) only")
            if wp_cfg > 1:
                if zs != world_size and zs != wp_cfg:
                    raise NotImplementedError(f"parallel.weight.size = {wp_cfg} with parallel.zero1.size = {zs}: here the weight group IS the zero "
                                              "group (fp32 state, bf16 weights and gradients are cut the same way); leave zero1.size at -1")
                zs = wp_cfg
                if world_size % zs:
                    raise ValueError(f"parallel.weight.size = {zs} must divide the data-parallel size {world_size}")
        self.world, self.rank = zs, rank % zs
        self.layout = FlatLayout(self.lmc, zs, self.gid, self.pipe.first, self.pipe.last)
        L = self.layout
        self.comm = ZeroComm(L, process_group, world_size, rank, force_collectives, zero_size=zs, dp_groups=dp_groups)
        self.sp = sp_size
        # (rank / world_size are stage-local under pipeline parallelism: every stage of a pipeline reads the same batches, and inside a stage the ranks of a
        # sequence group do)
        self.seqpar = SeqParallel(sp_size, self.pipe.dp_rank if pp_size > 1 else rank, self.pipe.dp_world if pp_size > 1 else world_size, stages=pp_size, stage=self.pipe.stage)
        if tp_size > 1:  # ... and so does every rank of a tensor group (inside a stage, under pipeline parallelism)
            self.seqpar.data_rank, self.seqpar.data_world = self.tpar.dp_rank, self.tpar.dp_world
        self.isp_rule = sp_size if sp_size > 1 else int(emulate_isp_grad_rule)
        mode = str(getattr(tc, "sp_attention", "auto") if sp_attention is None else sp_attention)
        if mode not in ("auto", "ulysses", "ring"):
            raise ValueError(f"sp_attention = {mode!r}: 'auto', 'ulysses' or 'ring'")
        heads_divide = mc.num_kv_attention_heads % sp_size == 0 and mc.num_attention_heads % sp_size == 0
        self.ring_mode = sp_size > 1 and (mode == "ring" or (mode == "auto" and not heads_divide))
        self.overlap_gathered_rows = bool(int(os.environ.get("IE_OVERLAP_GATHERED_ROWS", "0")))   # msp / fsp: see _gathered_rows
        self.attn_bwd_spill = bool(int(os.environ.get("IE_ATTN_BWD_SPILL", "0")))
        if sp_size > 1 and tc.packed_length % sp_size:
            raise ValueError("sequence parallel size must divide the packed length")
        if sp_size > 1 and not self.ring_mode and not heads_divide:
            raise ValueError("sequence parallel size must divide the head counts for the head exchange (parallel.tensor.attention = 'ring' has no such limit)")

        # ---- flat parameter / gradient buffers + ZeRO-1 fp32 state of this rank's shards
        # Physical placement of the buckets inside `params` / `grads`.  Resident: the layout's own offsets.  Weight parallel: a layer bucket
        # occupies only this rank's shard there; its whole image exists in one of two pool slots while the layer runs (layer l -> slot l % 2).
        nb = len(L.buckets)
        self._wp_buckets = set(range(1, nb - 1)) if self.wp_mode else set()
        self._pbase, off = [], 0
        for b in L.buckets:
            self._pbase.append(off)
            off += b.size // zs if b.index in self._wp_buckets else b.size
        assert self.wp_mode or off == L.total
        self.params = torch.zeros(off, dtype=BF16, device=device)
        self.grads = torch.zeros(off, dtype=BF16, device=device)
        nloc = L.local_numel()
        self.master = torch.zeros(nloc, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(nloc, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(nloc, dtype=torch.float32, device=device)
        if self.wp_mode:
            slot = max(L.buckets[i].size for i in self._wp_buckets)
            self.pool_p = [torch.zeros(slot, dtype=BF16, device=device) for _ in range(2)]   # (padding stays zero: nothing writes it)
            self.pool_g = [torch.zeros(slot, dtype=BF16, device=device) for _ in range(2)]
            self.t_gshard = [torch.empty(slot // zs, dtype=BF16, device=device) for _ in range(2)]   # reduce-scatter target of a later micro-batch
            self.wp_stream = torch.cuda.Stream(device=device) if device.type == "cuda" else None
            self._wp_inflight, self._wp_rs, self._slot_layer = {}, [None, None], [None, None]

        def backing(spec, flat, pool):   # (storage tensor, element offset) a parameter's view lives in
            b = L.buckets[spec.bucket]
            if b.index in self._wp_buckets:
                return pool[(b.index - 1) % 2], spec.offset - b.start
            return flat, spec.offset - b.start + self._pbase[b.index]

        self.p, self.g, self.p13, self.g13 = {}, {}, {}, {}
        for n, sp_ in L.params.items():
            for dst, dst13, flat, pool in ((self.p, self.p13, self.params, getattr(self, "pool_p", None)), (self.g, self.g13, self.grads, getattr(self, "pool_g", None))):
                t, o = backing(sp_, flat, pool)
                dst[n] = t[o : o + sp_.numel].view(sp_.shape)
                if sp_.kind == "w1":     # w1 and w3 are adjacent: one [2F, h] GEMM operand
                    dst13[sp_.layer] = t[o : o + 2 * sp_.numel].view(2 * sp_.shape[0], sp_.shape[1])
        self._init_params(init, seed, init_fn)
        self.sync_master_from_params()

        # ---- device-resident step state
        self.state = K.step_state_new(device, tc.initial_scale)
        self.scaler_cfg = IeScalerConfig(tc.growth_factor, tc.backoff_factor, tc.min_scale, tc.max_scale, tc.growth_interval, tc.hysteresis,
                                         tc.clip_grad_norm, 1)
        self.lr_sched = CosineWarmupLR(tc.lr, tc.total_steps, tc.warmup_ratio, tc.eta_min, tc.init_steps)
        self.beta2_sched = Beta2Scheduler(tc.adam_beta2, tc.adam_beta2_c)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=device)
        # ISP (tensor mode "isp"): embedding and head are the optimizer group "1_embed_head" (train/utils.py:42-43), the rest "0_default"; every group has
        # its own norm and is unscaled / clipped by its OWN factor (hybrid_zero_optim.py:760-779,863-876; pinned on tests/golden/train_isp2*_bf16_rank*.json)
        self.isp_groups = self.isp_rule > 1
        if self.isp_groups:
            self.sumsq_g = torch.zeros(2, dtype=torch.float32, device=device)
            self.group_inv = torch.zeros(2, dtype=torch.float32, device=device)
            self.group_norm = torch.zeros(2, dtype=torch.float32, device=device)
        self.sumsq_ws = torch.empty(K._L().ie_sumsq_max_partials() * (len(L.buckets) + 1), dtype=torch.float32, device=device)

        # ---- scale_on_q: the rotary kernel stores bf16(q * scale * log2 e) (one rounding, like the unscaled q), attention is called with
        # softmax_scale = ln 2 (scale * log2 e inside the kernels becomes 1: scores come off the MFMA accumulators in log2 units, which is
        # what the folded-softmax forward kernel wants), and the rotary backward applies the chain rule's q_scale to dq in fp32
        self.scale_on_q = bool(int(os.environ.get("IE_SCALE_ON_Q", "0")) if scale_on_q is None else scale_on_q)
        self.attn_scale = _LN2 if self.scale_on_q else None          # (None: 1 / sqrt(head_dim) inside the attention kernels, as the reference)
        self.q_scale = (mc.head_dim ** -0.5) * _LOG2E if self.scale_on_q else 1.0
        self.dq_scale = self.q_scale          # chain rule: the attention backward returns dL/d(q~) of q~ = q_scale * q
        # ---- rotary tables (embedding.py:301-327: fp32 -> bf16), sized on demand
        self._rot_len = 0
        self._ensure_rotary(tc.seq_len)

        # ---- activation / workspace buffers for T tokens per micro-batch
        # merge_micro: run the micro_num micro-batches of a step as ONE pass over micro_num * packed_length tokens (see forward_backward)
        can_merge = tc.micro_num > 1 and sp_size == 1 and mc.checkpoint_layers == 0
        if merge_micro and not can_merge:
            raise ValueError("merge_micro needs micro_num > 1, no sequence parallelism and no activation checkpointing")
        if self.wp_mode:
            if batch_wgrad:
                raise ValueError("batch_wgrad keeps gradients in place across micro-batches; weight parallelism reduce-scatters them per micro-batch")
            batch_wgrad = False
        if merge_micro is None and batch_wgrad:
            merge_micro = False  # an explicit request for the staged per-micro-batch path
        if merge_micro is None:
            # automatic: on when the activations of micro_num micro-batches fit next to the weights and the optimizer state.  Decided
            # from static sizes and the device's TOTAL memory, so every rank of a job takes the same decision (the tensor-parallel
            # all-reduces of the two modes differ in number and size)
            lm = self.lmc
            per_token = 2 * (lm.num_layers * (6 * lm.hidden_size + 2 * lm.num_kv_attention_heads * lm.head_dim + 2 * lm.ffn_dim)
                             + lm.head_vocab + 12 * lm.hidden_size + 6 * lm.ffn_dim + 2 * lm.qkv_dim) + 64
            fixed = 4 * L.total + 12 * L.local_numel()
            total = torch.cuda.get_device_properties(device).total_memory if device.type == "cuda" else 0
            merge_micro = can_merge and fixed + per_token * tc.packed_length * tc.micro_num + (24 << 30) < 0.92 * total
        self.mm = tc.micro_num if merge_micro else 1       # micro-batches per pass
        self.n_pass = tc.micro_num // self.mm               # passes (gradient-accumulation steps) per optimizer step
        self.Tg = tc.packed_length * self.mm    # tokens of a pass
        self.T = self.Tg // sp_size             # tokens this rank owns (all of them without sequence parallelism)
        self.rl = self.tpar.rows(self.T) if self.ss else slice(None)   # the rows of the residual stream this rank works on (msp / fsp: T / tp of them)
        self._alloc(self.T)
        self.t_loss_seg = torch.empty(self.mm, 2, dtype=torch.float32, device=device)  # per micro-batch [mean loss, valid tokens]
        self.batch_wgrad = self._alloc_wgrad_stage(batch_wgrad)
        # The SwiGLU product of every layer kept from the forward (autograd saves it too: it is w2's input) instead of being written a second
        # time by the SwiGLU backward: 2 F bytes per token and layer less HBM traffic in backward (12 F -> 10 F), 2 F T bytes per layer more
        # memory (15 GB for the 7B merged pass).  On when that fits with room to spare; the values are the same bits either way.
        self.a_act = None
        if not self.batch_wgrad and device.type == "cuda" and os.environ.get("IE_KEEP_ACT", "1") != "0":   # (IE_KEEP_ACT=0: A/B switch)
            lm = self.lmc
            nslot = len(self.a_w13)
            # under pipeline parallelism a stage keeps one activation SET per in-flight micro-batch (up to pp, or the interleaved plan's slot count)
            sets = 1 if pp_size == 1 else (min(pp_size - self.pipe.stage, tc.micro_num) if self.nch == 1 else tc.micro_num)
            need = 2 * nslot * self.T * lm.ffn_dim * sets
            if need + (24 << 30) < _mem_budget(device):
                self.a_act = [torch.empty(self.T, lm.ffn_dim, dtype=BF16, device=device) for _ in range(nslot)]
        self.loss_acc = torch.zeros(1, dtype=torch.float32, device=device)  # sum over micro-batches of loss/micro_num
        # The optimizer runs on its own HIP stream: AdamW is HBM-bound (28 B per parameter), the next step's first forward
        # GEMMs are MFMA-bound, so bucket b+1's update overlaps the forward of layer b; per-bucket events order the two.
        self.opt_stream = _optimizer_stream(device)
        # AdamW beside the next step's forward (step()): the buckets behind the first adamw_full_buckets run on adamw_cus CUs (ie_tune_adamw_cus; 0 = whole chip)
        self.adamw_cus = int(os.environ.get("IE_ADAMW_CUS", "128") or 0)
        # the block's residual adds in the epilogues of wo / w2 (kernels.linear_fwd_add; IE_RES_IN_EPILOGUE=0: A/B switch): only where product and add are neighbours.
        # Measured (profiles/r06_step_residual_in_epilogue_abab.log): first with the addend read where it is used: the norm behind it 88 -> 46 us, the product
        # + 80 us, the step 0.15 % slower; with the addend requested one epilogue turn ahead: + 14 us per product, 653.6 / 653.8 -> 653.0 / 653.1 ms per step.
        self.res_in_epilogue = os.environ.get("IE_RES_IN_EPILOGUE", "1") != "0" and self.tp == 1 and not self.bias and not self.ss
        self.attn_bwd_rotary_fuse = os.environ.get("IE_ATTN_BWD_ROTARY_FUSE", "1") != "0"   # (A/B switch: kernels.flash_attn_bwd_qkv_rotary in _layer_backward)
        self.adamw_full_buckets = int(os.environ.get("IE_ADAMW_FULL_BUCKETS", "2") or 0)
        self._bucket_ready = [None] * len(self.layout.buckets)
        self._opt_done = None
        self.metric = None  # optional internevo_amd.metrics.AccPerplex (attach_metric)
        self.step_count = 0

    # ------------------------------------------------------------------------------------------ setup
    def _init_params(self, init, seed, init_fn):
        """modeling_internlm2.py:646-672 + :890-893,:955-961: normal(0.02); wo / w2 scaled by 1/sqrt(2*(layer+1)); norms = 1."""
        mc = self.mc
        if init_fn is not None:
            self.load_named_parameters({n: init_fn(n, shape) for n, shape in self.reference_param_shapes().items()}, sync_master=False)
            return
        # The FULL (un-sharded) tensors are drawn from one generator in the order of the single-rank layout and every tensor rank
        # keeps its cut (tensorpar.shard): all data-parallel ranks start equal, the shards of a tensor group are DIFFERENT pieces
        # of the same full model (the reference seeds TENSOR mode with seed + tp_rank for the same purpose,
        # parallel_context.py:639-641), and a tp = n run starts from exactly the weights of the tp = 1 run with the same seed.
        gen = torch.Generator(device=self.dev).manual_seed(seed)
        v1 = self._is_v1()
        for n, full in FlatLayout(mc, 1).params.items():   # (a pipeline stage draws the whole sequence too and keeps its layers)
            if full.kind == "norm" or full.kind in ("bqkv", "bo"):   # unit norm gains, zero biases
                if n in self.p:
                    self._store_param(n, self.tpar.shard(full.kind, (torch.ones if full.kind == "norm" else torch.zeros)(full.shape, dtype=BF16, device=self.dev)))
                continue
            std = mc.init_std
            if mc.use_scaled_init and full.kind in ("wo", "w2"):
                std = mc.init_std / math.sqrt(2.0 * (full.layer + 1))
            if v1:   # modeling_internlm.py:161-195,:331-333,:380-382: normal(0.006) Wqkv / w1 / w3, normal(0.0015) (0.006 / sqrt(2 (l + 1)) with use_scaled_init)
                     # out_proj / w2, normal(0.0052) embedding / head
                std = 0.0052 if full.kind in ("embed", "head") else 0.006
                if full.kind in ("wo", "w2"):
                    std = 0.006 / math.sqrt(2.0 * (full.layer + 1)) if mc.use_scaled_init else 0.0015
            w = torch.empty(full.shape, dtype=torch.float32, device=self.dev).normal_(0.0, std, generator=gen)
            if n in self.p:
                self._store_param(n, self.tpar.shard(full.kind, w))

    # ---- where a bucket lives (resident: the layout's offsets; weight parallel: layer buckets hold this rank's shard only) ----------
    def _shard(self, flat, b):
        """this rank's 1/world part of bucket b inside `flat` (params or grads)."""
        n = b.size // self.world
        s = self._pbase[b.index] + (0 if b.index in self._wp_buckets else self.rank * n)
        return flat[s : s + n]

    def _full(self, flat, b):
        """bucket b whole (resident buckets only)."""
        assert b.index not in self._wp_buckets
        return flat[self._pbase[b.index] : self._pbase[b.index] + b.size]

    def _store_param(self, name, value):
        """value: this rank's (tensor-parallel cut of the) WHOLE parameter -> its storage; under weight parallelism only the part of a layer
        parameter that falls into this rank's shard of the bucket is kept."""
        spec = self.layout.params[name]
        b = self.layout.buckets[spec.bucket]
        if b.index not in self._wp_buckets:
            self.p[name].copy_(value)
            return
        n = b.size // self.world
        s0 = b.start + self.rank * n
        a, z = max(s0, spec.offset), min(s0 + n, spec.offset + spec.numel)
        if a < z:
            self._shard(self.params, b)[a - s0 : z - s0].copy_(value.reshape(-1)[a - spec.offset : z - spec.offset])

    def sync_master_from_params(self):
        """fp32 master copy of this rank's shards (hybrid_zero_optim.py:214-233)."""
        L = self.layout
        for b, lo in zip(L.buckets, L.local_offsets()):
            n = b.size // self.world
            self.master[lo : lo + n].copy_(self._shard(self.params, b))
        if self.wp_mode:
            self._slot_layer = [None, None]   # whatever the pool holds is stale now

    def _ensure_rotary(self, seqlen):
        if seqlen <= self._rot_len:
            return
        d = self.mc.head_dim
        inv_freq = 1.0 / (self.mc.rope_base ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
        freqs = torch.outer(torch.arange(seqlen, dtype=torch.float32), inv_freq)
        self.cos = torch.cos(freqs).to(BF16).to(self.dev)
        self.sin = torch.sin(freqs).to(BF16).to(self.dev)
        self._rot_len = seqlen

    def _alloc(self, T):
        mc, dev = self.lmc, self.dev   # per-rank sizes (1/tp of the heads and of the FFN width under tensor parallelism)
        h, F, V, L = mc.hidden_size, mc.ffn_dim, mc.vocab_size, mc.num_layers
        hq, hkv, d = mc.num_attention_heads, mc.num_kv_attention_heads, mc.head_dim

        def e(*shape, dtype=BF16):
            return torch.empty(shape, dtype=dtype, device=dev)

        # saved per layer: the layer input always; everything else per activation SLOT -- a layer under activation
        # checkpointing (model.checkpoint, modeling_internlm2.py:857-861,910: lid < num_layers * fraction) shares slot 0
        # and is recomputed from its input in backward (solver/activation_checkpoint.py:40-172), the others own a slot
        nck = mc.checkpoint_layers
        self.slot = [0 if l < nck else l - nck + (1 if nck else 0) for l in range(L)]
        S = (L - nck) + (1 if nck else 0)
        self.a_x = [e(T, h) for _ in range(L)]        # layer input (residual stream)
        self.a_n1 = [e(T, h) for _ in range(S)]
        self.a_rstd1 = [e(T, dtype=torch.float32) for _ in range(S)]
        # attention works on ALL Tg tokens of the micro-batch with this rank's 1/sp of the heads (same element counts)
        sp, Tg = self.sp, self.Tg
        hql, hkvl = hq // sp, hkv // sp
        if self.ring_mode:      # ring attention keeps the local tokens with ALL heads (same element counts where the heads divide)
            Tg, hql, hkvl = T, hq, hkv
        self.a_q = [e(Tg, hql, d) for _ in range(S)]
        self.a_kv = [e(Tg, 2, hkvl, d) for _ in range(S)]
        self.a_ctx = [e(Tg, hql, d) for _ in range(S)]
        self.a_lse = [e(hql, Tg, dtype=torch.float32) for _ in range(S)]
        self.a_ctxl = [e(T, hq, d) for _ in range(S)] if (sp > 1 and not self.ring_mode) else self.a_ctx   # context of the local tokens, all heads (wo input)
        self.a_r2 = [e(T, h) for _ in range(S)]
        self.a_n2 = [e(T, h) for _ in range(S)]
        self.a_rstd2 = [e(T, dtype=torch.float32) for _ in range(S)]
        self.a_w13 = [e(T, 2 * F) for _ in range(S)]
        self.a_xf, self.a_nf, self.a_rstdf = e(T, h), e(T, h), e(T, dtype=torch.float32)
        # transient
        self.t_qkv = e(T, mc.qkv_dim)
        if self.bias:
            self.t_bias = e(max(mc.qkv_dim, h))   # a micro-batch's bias gradient on its way into the accumulated one
        self.t_h0, self.t_h1, self.t_h2 = e(T, h), e(T, h), e(T, h)
        self.t_h3 = e(T, h) if nck else None          # wo output of a recomputed layer (t_h0..2 carry gradients then)
        self.t_act = e(T, F)
        self.t_dact = e(T, F)
        self.t_dw13 = e(T, 2 * F)
        self.t_dq = e(Tg, hql, d)
        self.t_dkv = e(Tg, 2, hkvl, d)
        self.ring = None
        if self.ring_mode:
            from .seqpar import RingAttention

            self.ring = RingAttention(self.seqpar, hq, hkv, d, T, self.dev, None)   # (the scale is set below, with attn_scale)
            self.t_loss_red = e(2, dtype=torch.float32)
        elif sp > 1:  # local-token / all-head staging of the exchanges
            self.t_ql, self.t_kvl = e(T, hq, d), e(T, 2, hkv, d)
            self.t_xq, self.t_xkv = e(T, hq, d), e(T, 2, hkv, d)      # send / receive buffers
            self.t_dctx_full = e(Tg, hql, d)
            self.t_loss_red = e(2, dtype=torch.float32)
        self.t_logits = e(T, mc.head_vocab)
        # ScaleColumnParallelLinearWithNormHead (ops/linear.py:79-153): with embed_grad_scale != 1 / norm_head the head multiplies by a function of
        # its weight (kernels.head_weight_fwd), rebuilt in every forward pass like the reference does; the weight gradient goes through t_head_dw
        self.head_fn = (mc.embed_grad_scale != 1.0 or mc.norm_head) and self.pipe.last
        if self.head_fn:
            self.t_head_w = e(mc.head_vocab, mc.hidden_size)
            self.t_head_dw = e(mc.head_vocab, mc.hidden_size)
            self.t_head_inv = e(mc.head_vocab, dtype=torch.float32)
        if self.vp:
            self.t_lab_local = e(T, dtype=torch.int64)   # labels in this rank's vocabulary range (-1: valid, owned by another rank)
        self.t_loss_rows = e(T, dtype=torch.float32)
        self.t_lse = e(T, dtype=torch.float32)
        self.t_loss = e(2, dtype=torch.float32)       # [mean loss of the micro-batch, valid-token count]
        self.t_delta = e(K._L().ie_flash_attn_bwd_workspace(Tg, hql, hkvl, d), dtype=torch.float32)
        self.t_norm_ws = e(K._L().ie_rmsnorm_bwd_partials(T) * h, dtype=torch.float32)
        self.t_emb_ws = e(V + 1 + T, dtype=torch.int32)
        if getattr(self, "embed_split", False):
            self.t_emb_loc = e(T, mc.embed_dim)
        self.scale_view = self.state[:4].view(torch.float32)  # IeStepState.loss_scale, read by the CE backward on device

    def _alloc_wgrad_stage(self, want):
        """Batched weight gradients.  The reference (autograd) runs one weight-gradient GEMM per linear per micro-batch and adds
        it into the bf16 .grad; with micro_num micro-batches that is micro_num GEMMs of contraction length T each.  On 288 GB there
        is room to keep every linear's (dY, X) pair of ALL micro-batches of a step (7B, seq 4096, micro_num 4: 76 GB) and run ONE
        GEMM per linear with contraction length micro_num * T in the last micro-batch's backward: the tile prologue / epilogue and
        the bf16 read-modify-write of dW are paid once (+6 ... +14 % on the weight-gradient GEMMs, tools/wgrad_batch_probe.py),
        the sum over micro-batches is carried in the fp32 accumulators (rounded to bf16 once instead of micro_num times), and
        the gradient reduce-scatter of a bucket still starts right after its last weight gradient.  No copies: the producing
        kernels write straight into the micro-batch's rows of the [micro_num * T, cols] staging tensors.
        want: None = on when micro_num > 1, no activation checkpointing and the memory is there; True / False = forced."""
        tc, mc = self.tc, self.lmc
        M, T, L = self.n_pass, self.T, mc.num_layers
        h, F, V = mc.hidden_size, mc.ffn_dim, mc.vocab_size
        ctx_cols = mc.num_attention_heads * mc.head_dim
        V = mc.head_vocab
        cols = L * (3 * h + ctx_cols + 3 * F + mc.qkv_dim) + h + V  # n1, n2, d_out, d_r2 | ctx | act, dw13 | dqkv ; nf, logits
        need = 2 * M * T * cols
        possible = M > 1 and mc.checkpoint_layers == 0
        if want is None:
            free = _mem_budget(self.dev) if self.dev.type == "cuda" else 0
            want = possible and need + (16 << 30) < free
        elif want and not possible:
            raise ValueError("batch_wgrad needs micro_num > 1 and no activation checkpointing")
        if not want:
            return False

        def big(c):
            return torch.empty(M * T, c, dtype=BF16, device=self.dev)

        self.st_n1, self.st_n2 = [big(h) for _ in range(L)], [big(h) for _ in range(L)]
        self.st_ctx = [big(ctx_cols) for _ in range(L)]
        self.st_act, self.st_dw13 = [big(F) for _ in range(L)], [big(2 * F) for _ in range(L)]
        self.st_dqkv = [big(mc.qkv_dim) for _ in range(L)]
        self.st_dout, self.st_dr2 = [big(h) for _ in range(L)], [big(h) for _ in range(L)]
        self.st_nf, self.st_logits = big(h), big(V)
        self.batch_wgrad = True
        self._bind_micro(0)  # also releases the single-micro-batch buffers these replace
        return True

    def _bind_micro(self, i):
        """Point the forward's saved linear inputs at micro-batch i's rows of the weight-gradient staging tensors."""
        if not getattr(self, "batch_wgrad", False):
            return
        T, mc = self.T, self.lmc
        r = slice(i * T, (i + 1) * T)
        self._mrows = r
        self.a_n1 = [t[r] for t in self.st_n1]
        self.a_n2 = [t[r] for t in self.st_n2]
        ctx = [t[r].view(T, mc.num_attention_heads, mc.head_dim) for t in self.st_ctx]
        self.a_ctxl = ctx
        if self.sp == 1 or self.ring_mode:
            self.a_ctx = ctx  # without the head exchange the attention output IS the wo input
        self.a_nf, self.t_logits = self.st_nf[r], self.st_logits[r]

    # ------------------------------------------------------------------------------------------ forward / backward
    def _w13(self, l):
        return self.p13[self.gid[l]], self.g13[self.gid[l]]

    # ---- weight parallelism: a layer's weights exist whole only in its pool slot, around the layer's own kernels ----------------------
    def _wp_gather(self, l):
        """Start the all-gather of layer l's weight shards into slot l % 2 on the side stream: behind everything queued so far on the compute
        stream (the slot's previous tenant, layer l +- 2, is done by then) and behind the optimizer's update of the shard."""
        slot, b = l % 2, self.layout.buckets[1 + l]
        if self._slot_layer[slot] == l or l in self._wp_inflight:
            return
        self._slot_layer[slot] = None
        full, shard = self.pool_p[slot][: b.size], self._shard(self.params, b)
        if not self.comm.active:      # one rank: the shard is the bucket (the pool logic runs, the collectives are identities)
            self._wait_bucket(b.index)   # the optimizer stream may still be updating the shard (the multi-rank branch waits for the same event)
            full.copy_(shard)
            self._wp_inflight[l] = None
            return
        if self.wp_stream is None:   # CPU tensors (tests)
            self._wait_bucket(b.index)
            self._wp_inflight[l] = self.comm.all_gather_async(full, shard)
            return
        cur = torch.cuda.current_stream(self.dev)
        ev = torch.cuda.Event()
        ev.record(cur)
        with torch.cuda.stream(self.wp_stream):
            self.wp_stream.wait_event(ev)
            ready = self._bucket_ready[b.index]
            if ready is not None:
                self.wp_stream.wait_event(ready)
            self._wp_inflight[l] = self.comm.all_gather_async(full, shard)

    def _layer_ready(self, l, nxt):
        """Before the first kernel that reads layer l's weights.  Resident: order the stream behind the optimizer's work on the bucket.
        Weight parallel: wait for the gather of layer l, then start the gather of `nxt` (the layer this pass visits next) into the other
        slot, and make sure that slot's gradient image has left (its reduce-scatter of two layers ago)."""
        if not self.wp_mode:
            self._wait_bucket(1 + l)
            return
        self._wp_gather(l)
        work = self._wp_inflight.pop(l, None)
        if work is not None:
            work.wait()                 # the compute stream continues behind the collective
        self._slot_layer[l % 2] = l
        self._wp_finish_rs(l % 2)
        if nxt is not None:
            self._wp_gather(nxt)

    def _wp_reduce(self, l, first_micro):
        """Layer l's weight gradients of this micro-batch (whole, in pool slot l % 2) -> averaged over all ranks into this rank's gradient
        shard: written by the first micro-batch of a step, accumulated by the others (the reference's ISP hooks reduce-scatter every
        micro-batch too: isp.py:143-526)."""
        slot, b = l % 2, self.layout.buckets[1 + l]
        full, shard = self.pool_g[slot][: b.size], self._shard(self.grads, b)
        if not self.comm.active:
            shard.copy_(full[: shard.numel()]) if first_micro else shard.add_(full[: shard.numel()])
            return
        if first_micro:
            self._wp_rs[slot] = (self.comm.reduce_scatter_async(full, shard), None)
        else:
            tmp = self.t_gshard[slot][: shard.numel()]
            self._wp_rs[slot] = (self.comm.reduce_scatter_async(full, tmp), lambda: shard.add_(tmp))

    def _wp_finish_rs(self, slot):
        ent = self._wp_rs[slot]
        if ent is not None:
            work, fin = ent
            work.wait()
            if fin is not None:
                fin()
            self._wp_rs[slot] = None

    def _reduce_bucket(self, bi):
        """Start the averaging of a RESIDENT bucket's gradients over the data-parallel group (embedding, head; every bucket without weight
        parallelism): reduce-scatter into the owner's slice, in place."""
        b = self.layout.buckets[bi]
        if self.comm.active and b.size:
            self.comm.pending.append(self.comm.reduce_scatter_async(self._full(self.grads, b), self._shard(self.grads, b)))
        elif self.hold is not None and b.size:
            self._hold(self._hold_rs, b)

    def _hold(self, side, b):
        """(diagnostic) n CUs held on `side`, behind what the current stream holds now, for the time bucket b's collective would run on 8 ranks."""
        n, link = self.hold
        usec = max(int(2.0 * b.size / 8 / (link * 1e3)), 1)   # bytes per link / (GB/s) -> microseconds
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            side.wait_event(ev)
            K.hold_cus(n, usec)

    def _gathered_rows(self, x, product):
        """msp / fsp: `x` [T, C] holds this rank's token rows; product(rows) is a row-wise product that reads x[rows] (a column-parallel linear's forward,
        a row-parallel linear's input gradient).  The all-gather of the other ranks' rows runs UNDER the product of this rank's own rows, the rest
        follows behind the wait -- the reference's fsp overlap (FusedDenseFunc, model/utils.py:228-346: async all-gather of x under the weight cast,
        re-gather in backward under the input gradient).  Without sequence-sharded activations: product over all rows.  Same values either way (the
        products are row-wise); the gain needs xGMI to be measured."""
        if not self.ss:
            product(slice(None))
            return
        work = self.tpar.all_gather_rows_async(x)
        if not self.overlap_gathered_rows:   # default: ONE product over all rows behind the all-gather -- the row slices of the overlapped form can fall off
            work.wait()                      # the 256x256 / fused-epilogue schedules (pick_variant depends on the row count) and nothing on xGMI has shown
            product(slice(None))             # the overlap to pay yet; `overlap_gathered_rows=True` / IE_OVERLAP_GATHERED_ROWS=1 is the A/B switch
            return
        rl, T = self.rl, x.shape[0]
        product(rl)
        work.wait()
        if rl.start > 0:
            product(slice(0, rl.start))
        if rl.stop < T:
            product(slice(rl.stop, T))

    def _dev_cu(self, cu_h):
        """A micro-batch's sequence boundaries on the device; the host copy rides along (ring attention plans its blocks from it without a sync)."""
        cu = cu_h.to(self.dev, non_blocking=True)
        cu.host = cu_h
        return cu

    def _layer_forward(self, l, prev_ffn_out, cu, pos, max_seqlen, recompute, sum_dst=None):
        """One PackedFlashLlamaLayer1D (modeling_internlm2.py:684-740) into activation slot slot[l].
        recompute=False: the forward proper; the layer input a_x[l] = prev_ffn_out + previous layer's r2 is produced
        here (fused with the attention norm) and the w2 output is returned.
        recompute=True: the backward-time replay of a checkpointed layer from its saved input a_x[l]; same kernels on the
        same values (bit-identical activations), minus the w2 GEMM whose output backward does not need."""
        mc = self.lmc
        F, eps = mc.ffn_dim, mc.layer_norm_epsilon
        hkv, qpk, d = mc.num_kv_attention_heads, mc.q_per_kv, mc.head_dim
        p, s, rl = self.p, self.slot[l], self.rl
        pre = f"layers.{self.gid[l]}."   # (l counts this stage's layers; the names carry the reference's global layer numbers)
        if prev_ffn_out is None or prev_ffn_out is _SUMMED or recompute:   # (the first layer of the model / of a pipeline stage / of a model chunk, or the
            # layer below has added its residual in w2's epilogue: the input is in a_x[l])
            K.rmsnorm_fwd(self.a_x[l][rl], p[pre + "attention_norm.weight"], eps, self.a_n1[s][rl], self.a_rstd1[s][rl])
        else:
            K.add_rmsnorm_fwd(prev_ffn_out[rl], self.a_r2[self.slot[l - 1]][rl], p[pre + "attention_norm.weight"], eps, self.a_x[l][rl], self.a_n1[s][rl],
                              self.a_rstd1[s][rl])
        # (msp / fsp: all-gather in front of the column-parallel wqkv, under the product of the local rows; the gathered rows are kept: its weight gradient reads them)
        local = self.sp == 1 or self.ring_mode   # (ring attention: the local tokens keep all their heads)
        q_dst, kv_dst = (self.a_q[s], self.a_kv[s]) if local else (self.t_ql, self.t_kvl)
        if not self.bias and not self.ss:
            # wqkv product + GQA split + rotary in ONE launch where the persistent GEMM frame takes the product (d = 128): the [T, N] product never reaches memory
            K.linear_qkv_rotary_fwd(self.a_n1[s], p[pre + "attention.wqkv.weight"], self.cos, self.sin, pos, hkv, qpk, d, not mc.adapt_hf, q_dst, kv_dst, self.t_qkv,
                                    self.q_scale)
        else:
            self._gathered_rows(self.a_n1[s], lambda r: K.linear_fwd(self.a_n1[s][r], p[pre + "attention.wqkv.weight"], self.t_qkv[r]))
            if self.bias:   # the InternLM-1 block (multi_head_attention.py:371-396): Wqkv carries a bias
                K.bias_add(self.t_qkv, p[pre + "attention.wqkv.bias"])
            K.qkv_rotary_fwd(self.t_qkv, self.cos, self.sin, pos, hkv, qpk, d, not mc.adapt_hf, q_dst, kv_dst, self.q_scale)
        if not local:  # DistributedAttention: my tokens / all heads -> all tokens / my heads (multi_head_attention.py:117-126)
            xq = self.seqpar.scatter_heads_gather_seq_async(self.t_ql, 1, self.t_xq, self.a_q[s])     # q's exchange runs under kv's packing copy
            xkv = self.seqpar.scatter_heads_gather_seq_async(self.t_kvl, 2, self.t_xkv, self.a_kv[s])  # and the two exchanges beside each other
            xq.wait()
            xkv.wait()
        if self.ring_mode:   # K / V blocks travel around the sequence group instead of the heads (seqpar.RingAttention)
            self.ring.scale = self.attn_scale
            self.ring.forward(self.a_q[s], self.a_kv[s], self.ring.plan(cu.host), self.a_ctx[s], self.a_lse[s])
        else:
            K.flash_attn_fwd(self.a_q[s], self.a_kv[s][:, 0], self.a_kv[s][:, 1], cu, max_seqlen, self.attn_scale, True, self.a_ctx[s], self.a_lse[s])
        if self.sp > 1 and not self.ring_mode:    # ... and back: all tokens / my heads -> my tokens / all heads (:127)
            self.seqpar.scatter_seq_gather_heads(self.a_ctx[s], 1, self.t_xq, self.a_ctxl[s])
        attn_out = self.t_h3 if recompute else self.t_h0
        # round 6: the residual add in the product's epilogue where nothing stands between product and add (no tensor-parallel sum, no bias) and the persistent
        # GEMM frame takes the shape -- r2 = bf16(bf16(ctx wo^T) + x), the two-step form's roundings -- so the norm reads one matrix instead of two
        if self.res_in_epilogue and K.linear_fwd_add(self.a_ctxl[s].view(self.T, -1), p[pre + "attention.wo.weight"], self.a_x[l], self.a_r2[s]):
            K.rmsnorm_fwd(self.a_r2[s], p[pre + "ffn_norm.weight"], eps, self.a_n2[s], self.a_rstd2[s])
        else:
            K.linear_fwd(self.a_ctxl[s].view(self.T, -1), p[pre + "attention.wo.weight"], attn_out)
            # row-parallel wo: partial sums over the tensor group (no-op without tensor parallelism); msp / fsp: summed into this rank's rows only
            (self.tpar.reduce_scatter_rows_async(attn_out) if self.ss else self.tpar.all_reduce_sum_async(attn_out)).wait()
            if self.bias:   # out_proj's bias, once, on the summed output (the reference's row-parallel linear holds it on tensor rank 0 only, ops/linear.py:318-324)
                K.bias_add(attn_out[rl], p[pre + "attention.wo.bias"])
            K.add_rmsnorm_fwd(attn_out[rl], self.a_x[l][rl], p[pre + "ffn_norm.weight"], eps, self.a_r2[s][rl], self.a_n2[s][rl], self.a_rstd2[s][rl])
        w13, _ = self._w13(l)
        if recompute:
            self._gathered_rows(self.a_n2[s], lambda r: K.linear_fwd(self.a_n2[s][r], w13, self.a_w13[s][r]))
            return None
        act = self.t_act if self.a_act is None else self.a_act[s]
        # w1 | w3 product with the gate in its epilogue (one launch at the 7B shapes)
        self._gathered_rows(self.a_n2[s], lambda r: K.linear_swiglu_fwd(self.a_n2[s][r], w13, self.a_w13[s][r], act[r]))
        if sum_dst is not None and self.res_in_epilogue and K.linear_fwd_add(act, p[pre + "feed_forward.w2.weight"], self.a_r2[s], sum_dst):
            return _SUMMED   # (the next layer's input / the final norm's: ffn_out + r2 is in sum_dst already)
        K.linear_fwd(act, p[pre + "feed_forward.w2.weight"], self.t_h1)
        (self.tpar.reduce_scatter_rows_async(self.t_h1) if self.ss else self.tpar.all_reduce_sum_async(self.t_h1)).wait()   # row-parallel w2
        return self.t_h1

    def _chunk(self, chunk):
        """(first local layer, end, is the model's first part, is its last part) of a forward / backward pass: the whole stage, or one of
        its model chunks under the interleaved pipeline schedule."""
        if chunk is None:
            return 0, self.lmc.num_layers, self.pipe.first, self.pipe.last
        la, lb = self.chunks[chunk]
        return la, lb, self.pipe.first and chunk == 0, self.pipe.last and chunk == self.nch - 1

    def _forward_micro(self, ids, labels, cu, pos, max_seqlen, nseg=None, chunk=None):
        mc = self.lmc   # (this stage's layer count under pipeline parallelism)
        L, eps = mc.num_layers, mc.layer_norm_epsilon
        la, lb, is_first, is_last = self._chunk(chunk)
        p = self.p
        self._wait_bucket(0)              # bucket b's AdamW / all-gather of the previous step() may still be running on the optimizer stream
        if is_first and self.embed_split:
            # Embedding1D under tensor parallelism (modules/embedding.py:52-60): this rank's h / tp columns of the looked-up rows, all-gathered
            # along the hidden dimension (gather_forward_split_backward)
            K.embedding_fwd(p["tok_embeddings.weight"], ids, self.t_emb_loc)
            allc = self.tpar.all_gather(self.t_emb_loc)                                   # [tp, T, h / tp]
            self.a_x[0].view(self.T, self.tp, -1).copy_(allc.permute(1, 0, 2))
        elif is_first:
            K.embedding_fwd(p["tok_embeddings.weight"], ids, self.a_x[0])
            if mc.embed_grad_scale != 1.0:   # modeling_internlm2.py:970-973 (the value; the backward scales the gradient)
                K.grad_scale_mix(self.a_x[0], mc.embed_grad_scale)
        # (a later pipeline stage / model chunk received its input -- the previous one's output -- straight into a_x[la])
        ffn_out = None
        for l in range(la, lb):
            self._layer_ready(l, l + 1 if l + 1 < lb else None)
            # (sum_dst: where ffn_out + r2 belongs -- the next layer's input, the final norm's input -- for the layer to add it in w2's epilogue if it can)
            ffn_out = self._layer_forward(l, ffn_out, cu, pos, max_seqlen, False, self.a_x[l + 1] if l + 1 < lb else (self.a_xf if is_last else None))
        if not is_last:   # the output = the residual stream after the last layer; norm, head and loss live behind the model's last layer
            rl = self.rl    # (msp / fsp: this rank's token rows of it -- what travels to the same tensor rank of the next stage)
            torch.add(ffn_out[rl], self.a_r2[self.slot[lb - 1]][rl], out=self.t_send[rl])
            return
        self._wait_bucket(L + 1)
        rl = self.rl
        if ffn_out is _SUMMED:
            K.rmsnorm_fwd(self.a_xf, p["norm.weight"], eps, self.a_nf, self.a_rstdf)
        else:
            K.add_rmsnorm_fwd(ffn_out[rl], self.a_r2[self.slot[L - 1]][rl], p["norm.weight"], eps, self.a_xf[rl], self.a_nf[rl], self.a_rstdf[rl])
        if self.head_fn:
            K.head_weight_fwd(p["output.weight"], mc.embed_grad_scale, mc.norm_head, self.t_head_w, self.t_head_inv)
        # [T, V], or this tensor rank's [T, V / tp] columns (msp / fsp: the head is column-parallel: all-gather along the sequence in front of it, ops/linear.py:146-153)
        head_w = self.t_head_w if self.head_fn else p["output.weight"]
        self._gathered_rows(self.a_nf, lambda r: K.linear_fwd(self.a_nf[r], head_w, self.t_logits[r]))
        if self.mm > 1:
            # merged pass: the loss (and the metric) stay per micro-batch -- each has its own valid-token count (loss = mean over
            # micro-batches of the mean token loss, no_pipeline_scheduler.py:146)
            P = self.T // self.mm
            for i in range(self.mm if nseg is None else nseg):
                self._cross_entropy(slice(i * P, (i + 1) * P), labels, self.t_loss_seg[i])
        else:
            self._cross_entropy(slice(0, self.T), labels, self.t_loss)
        if self.sp > 1:
            # the loss is the mean over ALL tokens of the micro-batch (the reference gathers the sequence in front of the head,
            # ops/linear.py:146-153 gather_dim=1): combine the local (sum, count) over the sequence group; the backward
            # then divides by the global count
            self.t_loss_red[0] = self.t_loss[0] * self.t_loss[1]
            self.t_loss_red[1] = self.t_loss[1]
            torch.nan_to_num_(self.t_loss_red[0:1], nan=0.0)  # a rank whose tokens are all ignored: 0/0 * 0
            self.seqpar.all_reduce_sum(self.t_loss_red)
            self.t_loss[1] = self.t_loss_red[1]
            self.t_loss[0] = self.t_loss_red[0] / self.t_loss_red[1]

    def _cross_entropy(self, r, labels, out):
        """Loss of the token rows `r` of the current pass: out[0] = mean over the valid tokens, out[1] = their count; with a metric
        attached (SchedulerMetricHook.post_helper_func -> AccPerplex.update) the arg-max / NLL pass is fused into the same sweep over
        the logits.  Vocabulary-parallel head: the same kernels run on this rank's [rows, V / tp] columns with the labels mapped into
        its range (-1 = valid but owned by another rank: no target term here), then ONE all-gather of (local log-sum-exp, local
        target logit) per token over the tensor group gives the global log-sum-exp (kept in t_lse for the backward) and the loss."""
        logits, lab, rows, lse = self.t_logits[r], labels[r], self.t_loss_rows[r], self.t_lse[r]
        metric = self.metric is not None
        if not self.vp:
            if metric:
                K.ce_fwd(logits, lab, -100, self.tc.label_smoothing, rows, lse, out, self.t_argmax[r], self.t_nll[r])
                self.metric.update_fused(self.t_nll[r], self.t_argmax[r], lab)
            else:
                K.ce_fwd(logits, lab, -100, self.tc.label_smoothing, rows, lse, out)
            return
        Vl = self.lmc.head_vocab
        v0 = self.tpar.tp_rank * Vl
        ll = self.t_lab_local[r]
        here = (lab >= v0) & (lab < v0 + Vl)
        ll.copy_(torch.where(lab == -100, lab, torch.where(here, lab - v0, torch.full_like(lab, -1))))
        if metric:
            K.ce_fwd(logits, ll, -100, 0.0, rows, lse, out, self.t_argmax[r], self.t_nll[r])
        else:
            K.ce_fwd(logits, ll, -100, 0.0, rows, lse, out)
        # rows = local lse - logit[label] where the label is here, else 0  ->  the target logit this rank contributes
        eps = self.tc.label_smoothing
        mine = [lse, torch.where(here, lse - rows, torch.zeros_like(rows))]
        if eps > 0:   # the smoothing term needs the mean logit over the WHOLE vocabulary (ce_loss.py:15-36 / flash-attn's smoothed parallel loss): one more statistic
            mine.append(logits.sum(dim=1, dtype=torch.float32))   # (fp32 accumulation inside the reduction: no [T, V / tp] fp32 copy of the logits)
        stats = self.tpar.all_gather(torch.stack(mine))   # [tp, 2 or 3, rows]
        lse.copy_(torch.logsumexp(stats[:, 0], dim=0))
        nll = torch.where(lab != -100, lse - stats[:, 1].sum(dim=0), torch.zeros_like(rows))
        if eps > 0:
            smooth = lse - stats[:, 2].sum(dim=0) / float(Vl * self.tp)
            rows.copy_(torch.where(lab != -100, (1.0 - eps) * nll + eps * smooth, torch.zeros_like(rows)))
        else:
            rows.copy_(nll)
        K.ce_mean(rows, lab, -100, out)
        if metric:
            # first index of the row maximum over the whole vocabulary: the largest local maximum, the lowest rank on ties
            am = self.t_argmax[r]
            top = logits.gather(1, am.long().unsqueeze(1)).squeeze(1).float()
            cand = self.tpar.all_gather(torch.stack([top, (am + v0).float()]))                                     # [tp, 2, rows]
            win = cand[:, 0].max(dim=0).indices
            am.copy_(cand[:, 1].gather(0, win.unsqueeze(0)).squeeze(0).to(torch.int32))
            self.t_nll[r].copy_(nll)   # (the metric's loss is the plain negative log-likelihood, smoothing or not)
            self.metric.update_fused(self.t_nll[r], am, lab)

    def _vp_smoothing_target_fix(self, dlogits, lab_local, count):
        """Vocabulary-parallel head with label smoothing: ce_bwd ran with eps / tp, which makes the uniform term right (eps / V over the whole
        vocabulary) and leaves the target's coefficient at -(1 - eps / tp) where -(1 - eps) is wanted: add eps (1 - 1 / tp) g to the target's
        element on the rank that owns it, g = loss scale / (valid tokens * micro_num) as in the kernel."""
        eps = self.tc.label_smoothing
        if not self.vp or eps <= 0:
            return
        own = lab_local >= 0                                  # (-100: ignored, -1: valid but another rank's column)
        g = self.scale_view * (eps * (1.0 - 1.0 / self.tp) / self.tc.micro_num) / count     # one device scalar: no host synchronisation
        src = torch.where(own, g, torch.zeros_like(g)).to(dlogits.dtype).unsqueeze(1)
        dlogits.scatter_add_(1, lab_local.clamp(min=0).unsqueeze(1), src)

    def _backward_micro(self, ids, labels, cu, pos, max_seqlen, last_micro, first_micro=False, chunk=None):
        mc, tc = self.lmc, self.tc
        L, F = mc.num_layers, mc.ffn_dim
        la, lb, is_first, is_last = self._chunk(chunk)
        hkv, qpk, d = mc.num_kv_attention_heads, mc.q_per_kv, mc.head_dim
        p, g = self.p, self.g
        T = self.T
        ws = self.t_norm_ws
        # msp / fsp: the residual stream's gradients live on this rank's rows `rl`; a column-parallel product's input gradient is reduce-scattered
        # into them, and the gradient in front of a row-parallel product's backward is all-gathered (the mirror of the forward)
        ss, rl = self.ss, self.rl
        tp_sum = self.tpar.reduce_scatter_rows_async if ss else self.tpar.all_reduce_sum_async
        # the first micro-batch of a step WRITES the gradients, the others accumulate: no zero_grad pass over 15.5 GB and no read of
        # the old value in the first weight-gradient epilogues (bucket padding is zero from allocation and never written)
        acc = not first_micro
        acc_l = acc and not self.wp_mode   # layer gradients under weight parallelism: written whole into the pool slot, accumulated as shards
        if first_micro:
            self._wait_optimizer()  # the previous step's AdamW reads the gradients this backward is about to overwrite
        # d(loss_scale * loss / micro_num) / dlogits, in place over the logits (inplace_backward=True, ce_loss.py:31)
        bw = self.batch_wgrad
        r = self._mrows if bw else None

        def wgrad(dy, x, gw, dy_all, x_all, a=None):
            if not bw:
                K.linear_wgrad(dy, x, gw, acc if a is None else a)
            elif last_micro:  # every micro-batch's rows are in place: one GEMM over micro_num * T tokens
                K.linear_wgrad(dy_all, x_all, gw, False)

        def bgrad(dy, gb, dy_all, a):
            """gradient of a linear's bias = column sums of its output gradient (linear_bias_wgrad, model/utils.py:590-631), fp32 sums rounded once"""
            if bw:
                if last_micro:
                    K.colsum(dy_all, gb)
            elif a:
                tmp = self.t_bias[: gb.numel()]
                K.colsum(dy, tmp)
                K.add_bf16(gb, tmp, gb)
            else:
                K.colsum(dy, gb)

        if is_last:
            # (vocabulary-parallel head: t_lse holds the GLOBAL log-sum-exp, the labels are the ones mapped into this rank's range by the
            # forward: a label owned by another rank is valid without a one-hot term here)
            lab_b = self.t_lab_local if self.vp else labels
            # label smoothing on 1/tp of the vocabulary: the kernel's uniform term eps' / V_local is the full-vocabulary eps / V for eps' = eps / tp
            # (the target's coefficient is put right below)
            eps_k = tc.label_smoothing / self.tp if self.vp else tc.label_smoothing
            if self.mm > 1:
                P = T // self.mm
                for i in range(self.mm):
                    rs = slice(i * P, (i + 1) * P)
                    K.ce_bwd(self.t_logits[rs], lab_b[rs], self.t_lse[rs], self.scale_view, self.t_loss_seg[i, 1:2], 1.0 / tc.micro_num, -100, eps_k)
                    self._vp_smoothing_target_fix(self.t_logits[rs], lab_b[rs], self.t_loss_seg[i, 1:2])
            else:
                K.ce_bwd(self.t_logits, lab_b, self.t_lse, self.scale_view, self.t_loss[1:2], 1.0 / tc.micro_num, -100, eps_k)
                self._vp_smoothing_target_fix(self.t_logits, lab_b, self.t_loss[1:2])
            dlog = self.t_logits

            K.linear_dgrad(dlog, self.t_head_w if self.head_fn else p["output.weight"], self.t_h0)
            ar = tp_sum(self.t_h0) if self.vp else None   # column-parallel head: its input gradient is a partial sum
            if not self.head_fn:
                wgrad(dlog, self.a_nf, g["output.weight"], self.st_logits if bw else None, self.st_nf if bw else None)
            else:   # the gradient w.r.t. the weight the GEMM used, then through normalize / the gradient scale into the parameter's gradient
                if not bw:
                    K.linear_wgrad(dlog, self.a_nf, self.t_head_dw, False)
                    K.head_weight_bwd(self.t_head_dw, self.t_head_w, self.t_head_inv, mc.embed_grad_scale, mc.norm_head, g["output.weight"], acc)
                elif last_micro:
                    K.linear_wgrad(self.st_logits, self.st_nf, self.t_head_dw, False)
                    K.head_weight_bwd(self.t_head_dw, self.t_head_w, self.t_head_inv, mc.embed_grad_scale, mc.norm_head, g["output.weight"], False)
            if ar is not None:
                ar.wait()
        d_out = self.st_dout[L - 1][r] if bw else self.t_h1
        if is_last:
            K.rmsnorm_bwd(self.t_h0[rl], self.a_xf[rl], p["norm.weight"], self.a_rstdf[rl], None, g["norm.weight"], acc, ws, d_out[rl])
            # (msp / fsp: d(residual stream) is all-gathered in front of the last layer's row-parallel w2, at the top of the layer loop)
            if last_micro:
                if ss:
                    self.tpar.all_reduce_avg(g["norm.weight"])
                self._reduce_bucket(len(self.layout.buckets) - 1)
        # (an earlier pipeline stage / model chunk received the gradient of its output from the next one into t_h1 = d_out)
        spare = [self.t_h0, self.t_h2]
        for l in range(lb - 1, la - 1, -1):
            pre = f"layers.{self.gid[l]}."
            w13, gw13 = self._w13(l)
            sl = self.slot[l]
            if self.wp_mode:
                self._layer_ready(l, l - 1 if l - 1 >= la else None)
            if l < mc.checkpoint_layers:
                self._layer_forward(l, None, cu, pos, max_seqlen, True)
            # feed-forward
            t_act, t_dw13, t_qkv = (self.st_act[l][r], self.st_dw13[l][r], self.st_dqkv[l][r]) if bw else (self.t_act, self.t_dw13, self.t_qkv)
            # msp / fsp: d_out arrives on this rank's rows (from the norm backward above / of the layer above, or a later pipeline stage -- whole then, the
            # gather an identity); its all-gather in front of the row-parallel w2's backward runs under the local rows' input gradient
            if self.a_act is not None and l >= mc.checkpoint_layers:   # the product is still there from the forward
                t_act = self.a_act[sl]
                # d(act) = d_out @ w2 never reaches memory: the gate's backward sits in the product's epilogue (one launch at the 7B shapes)
                self._gathered_rows(d_out, lambda r: K.linear_dgrad_swiglu_bwd(d_out[r], p[pre + "feed_forward.w2.weight"], self.a_w13[sl][r], t_dw13[r], self.t_dact[r]))
            else:
                def ffn_bwd(r, t_act=t_act):
                    K.linear_dgrad(d_out[r], p[pre + "feed_forward.w2.weight"], self.t_dact[r])
                    K.swiglu_bwd(self.t_dact[r], self.a_w13[sl][r, :F], self.a_w13[sl][r, F:], t_dw13[r, :F], t_dw13[r, F:], t_act[r])

                self._gathered_rows(d_out, ffn_bwd)
            wgrad(d_out, t_act, g[pre + "feed_forward.w2.weight"], self.st_dout[l] if bw else None, self.st_act[l] if bw else None, acc_l)
            d_n2 = spare[0]
            if self._rs_deferred is not None:   # rs_under_w13: the bucket of the layer above leaves now, under this layer's two longest products
                self._reduce_bucket(self._rs_deferred)
                self._rs_deferred = None
            K.linear_dgrad(t_dw13, w13, d_n2)
            ar = tp_sum(d_n2)   # input gradient of the column-parallel w1 | w3: summed over the tensor group ...
            wgrad(t_dw13, self.a_n2[sl], gw13, self.st_dw13[l] if bw else None, self.st_n2[l] if bw else None, acc_l)   # ... under this GEMM
            ar.wait()
            d_r2 = self.st_dr2[l][r] if bw else spare[1]
            K.rmsnorm_bwd(d_n2[rl], self.a_r2[sl][rl], p[pre + "ffn_norm.weight"], self.a_rstd2[sl][rl], d_out[rl], g[pre + "ffn_norm.weight"], acc_l, ws, d_r2[rl])
            # attention (msp / fsp: d_r2's all-gather in front of the row-parallel wo's backward, under the local rows' input gradient)
            d_ctx = d_n2.view(-1)[: T * mc.num_attention_heads * d].view(T, mc.num_attention_heads * d)  # reuse ([T, h], or 1/tp of it)
            self._gathered_rows(d_r2, lambda r: K.linear_dgrad(d_r2[r], p[pre + "attention.wo.weight"], d_ctx[r]))
            if self.bias:
                bgrad(d_r2, g[pre + "attention.wo.bias"], self.st_dr2[l] if bw else None, acc_l)
            # _SeqAllToAll.backward: the mirrored exchanges (multi_head_attention.py:47-53); d_ctx travels under wo's weight gradient
            head_x = self.sp > 1 and not self.ring_mode
            xc = self.seqpar.scatter_heads_gather_seq_async(d_ctx.view(T, -1, d), 1, self.t_xq, self.t_dctx_full) if head_x else None
            wgrad(d_r2, self.a_ctxl[sl].view(T, -1), g[pre + "attention.wo.weight"], self.st_dr2[l] if bw else None, self.st_ctx[l] if bw else None, acc_l)
            d_ctx_full = xc.wait() if head_x else d_ctx.view(T, -1, d)
            fused_rot = False
            if self.ring_mode:
                self.ring.backward(d_ctx_full, self.a_q[sl], self.a_kv[sl], self.a_ctx[sl], self.a_lse[sl], self.ring.plan(cu.host), self.t_dq, self.t_dkv, self.t_delta)
            else:
                if self.attn_bwd_spill:   # (A/B switch IE_ATTN_BWD_SPILL=1: the five-product backward, kernels.flash_attn_bwd_spill; a no-op once the buffer fits)
                    K.flash_attn_bwd_spill(True, cu.numel() - 1, max_seqlen, self.a_q[sl].shape[1], True, self.dev)
                # round 6: the rotary embedding's backward and the GQA rearrange's in the attention kernels' stores, straight into the wqkv output gradient
                # (bit-identical to the two calls below; where the library does not fuse the shape it says so and touches nothing)
                if (self.attn_bwd_rotary_fuse and not head_x and mc.adapt_hf and self.dq_scale == 1.0 and not self.attn_bwd_spill
                        and K.flash_attn_bwd_qkv_rotary(d_ctx_full, self.a_q[sl], self.a_kv[sl][:, 0], self.a_kv[sl][:, 1], self.a_ctx[sl], self.a_lse[sl], cu,
                                                        max_seqlen, self.cos, self.sin, pos, t_qkv, self.attn_scale, self.t_delta)):
                    fused_rot = True
                else:
                    K.flash_attn_bwd(d_ctx_full, self.a_q[sl], self.a_kv[sl][:, 0], self.a_kv[sl][:, 1], self.a_ctx[sl], self.a_lse[sl], cu,
                                     max_seqlen, self.attn_scale, True, self.t_dq, self.t_dkv[:, 0], self.t_dkv[:, 1], self.t_delta)
            if fused_rot:
                pass
            elif not head_x:
                dq_l, dkv_l = self.t_dq, self.t_dkv
            else:
                xq = self.seqpar.scatter_seq_gather_heads_async(self.t_dq, 1, self.t_xq, self.t_ql)
                xkv = self.seqpar.scatter_seq_gather_heads_async(self.t_dkv, 2, self.t_xkv, self.t_kvl)   # both in flight; dq unpacks under dkv's
                dq_l, dkv_l = xq.wait(), xkv.wait()
            if not fused_rot:
                K.qkv_rotary_bwd(dq_l, dkv_l, self.cos, self.sin, pos, hkv, qpk, d, not mc.adapt_hf, t_qkv, self.dq_scale)
            if self.bias:
                bgrad(t_qkv, g[pre + "attention.wqkv.bias"], self.st_dqkv[l] if bw else None, acc_l)
            d_n1 = d_n2  # the full [T, h] buffer again (d_ctx was a view of its first 1/tp)
            K.linear_dgrad(t_qkv, p[pre + "attention.wqkv.weight"], d_n1)
            ar = tp_sum(d_n1)   # input gradient of the column-parallel wqkv, overlapped with its weight gradient
            wgrad(t_qkv, self.a_n1[sl], g[pre + "attention.wqkv.weight"], self.st_dqkv[l] if bw else None, self.st_n1[l] if bw else None, acc_l)
            ar.wait()
            if bw:    # the layer below reads its output gradient from its own staging rows (it is the dY of that layer's w2)
                d_x = self.st_dout[l - 1][r] if l > 0 else self.t_h1
            else:
                d_x = d_out  # the old d_out buffer is free now
            K.rmsnorm_bwd(d_n1[rl], self.a_x[l][rl], p[pre + "attention_norm.weight"], self.a_rstd1[sl][rl], d_r2[rl], g[pre + "attention_norm.weight"], acc_l, ws, d_x[rl])
            if ss:   # (the layer below starts with its row-parallel w2's backward, which gathers d_x's rows; layer 0: after the loop)
                if last_micro:
                    self.tpar.all_reduce_avg(g[pre + "attention_norm.weight"])
                    self.tpar.all_reduce_avg(g[pre + "ffn_norm.weight"])
            # rotate buffers: next d_out = d_x; spare = the two others
            if not bw:
                spare = [d_n2, d_r2]
            d_out = d_x
            if self.wp_mode:
                self._wp_reduce(l, first_micro)
            elif last_micro:
                if self.rs_under_w13 and l > la:
                    self._rs_deferred = 1 + l
                else:
                    self._reduce_bucket(1 + l)
        if not is_first:
            return d_out   # gradient of this stage's (chunk's) input: travels to the previous stage (msp / fsp: this rank's token rows of it are valid)
        if ss:   # the embedding's backward takes the whole sequence (embedding.py:57-58)
            self.tpar.all_gather_rows_async(d_out).wait()
        if mc.embed_grad_scale != 1.0:
            K.scale_bf16(d_out, mc.embed_grad_scale)   # d(s x + (1 - s) x.detach()) / dx = s
        if self.embed_split:   # ... split backward: this rank's columns of the gradient
            self.t_emb_loc.copy_(d_out.view(T, self.tp, -1)[:, self.tpar.tp_rank])
            d_out = self.t_emb_loc
        K.embedding_bwd(d_out, ids, g["tok_embeddings.weight"], acc, self.t_emb_ws)
        if last_micro:
            self._reduce_bucket(0)
        return None

    def attach_metric(self, metric):
        """get_scheduler_hooks(metric, ...) of the reference (train/pipeline.py): the metric sees every micro-batch's logits."""
        self.metric = metric
        if metric is not None and not hasattr(self, "t_argmax"):
            self.t_argmax = torch.empty(self.T, dtype=torch.int32, device=self.dev)
            self.t_nll = torch.empty(self.T, dtype=torch.float32, device=self.dev)

    def zero_grad(self):
        self.grads.zero_()

    def forward_backward(self, batch, labels):
        """One NonPipelineScheduler.forward_backward_step: micro_num micro-batches with gradient accumulation.
        batch: dict with input_ids [micro_num, T] int64, cu_seqlens (list of int32 [n+1]), indexes [micro_num, T] int64
        (host tensors).  Returns the device scalar sum_i loss_i / micro_num."""
        tc = self.tc
        M = batch["input_ids"].shape[0]
        assert M == tc.micro_num and batch["input_ids"].shape[1] * self.mm == self.Tg
        if self.mm > 1:
            return self._forward_backward_merged(batch, labels)
        if self.pp > 1:
            return self._forward_backward_interleaved(batch, labels) if self.nch > 1 else self._forward_backward_pipeline(batch, labels)
        lo, hi = self.seqpar.sp_rank * self.T, (self.seqpar.sp_rank + 1) * self.T  # this rank's tokens of every micro-batch
        self.loss_acc.zero_()
        ids_d = batch["input_ids"].to(self.dev, non_blocking=True)
        lab_d = labels.to(self.dev, non_blocking=True)
        pos_d = batch["indexes"].to(self.dev, non_blocking=True)
        if self.metric is not None and self.metric.ntypes:
            self.metric.set_current_type_ids(batch["type_ids"])  # train.py:239-240
        for i in range(M):
            cu_h = batch["cu_seqlens"][i]
            max_seqlen = int((cu_h[1:] - cu_h[:-1]).max())  # host-side: no `.item()` sync (modeling_internlm2.py:989 syncs here)
            self._ensure_rotary(int(batch["indexes"][i].max()) + 1)
            cu = self._dev_cu(cu_h)
            ids_i, lab_i, pos_i = ids_d[i, lo:hi], lab_d[i, lo:hi], pos_d[i, lo:hi]
            self._bind_micro(i)
            if self.metric is not None and self.metric.ntypes and self.sp > 1:
                self.metric.type_ids_local = (lo, hi)
            self._forward_micro(ids_i, lab_i, cu, pos_i, max_seqlen)
            self.loss_acc.add_(self.t_loss[0:1], alpha=1.0 / M)
            self._backward_micro(ids_i, lab_i, cu, pos_i, max_seqlen, i == M - 1, i == 0)
        return self.loss_acc

    # ---- pipeline parallelism: the 1F1B schedule of one stage (pipeline.py; pipeline_scheduler.py:430-560) ---------------------
    _ACT_SETS = ("a_x", "a_n1", "a_rstd1", "a_q", "a_kv", "a_ctx", "a_lse", "a_r2", "a_n2", "a_rstd2", "a_w13")

    def _act_set_names(self):
        return self._ACT_SETS + (("a_act",) if self.a_act is not None else ()) + (("a_ctxl",) if (self.sp > 1 and not self.ring_mode) else ())

    def _alloc_inflight_sets(self):
        """A stage keeps the saved activations of up to pp - stage micro-batches (forwarded, not yet backwarded): whole extra sets
        of the per-layer activation lists, swapped in by _bind_inflight."""
        n = min(self.pp - self.pipe.stage, self.tc.micro_num)
        first = {name: getattr(self, name) for name in self._act_set_names()}
        self._sets = [first] + [{name: [torch.empty_like(t) for t in lst] for name, lst in first.items()} for _ in range(n - 1)]
        h = self.lmc.hidden_size
        self.t_send = torch.empty(self.T, h, dtype=BF16, device=self.dev)     # this stage's output on its way to the next stage

    def _bind_inflight(self, i):
        for name, lst in self._sets[i % len(self._sets)].items():
            setattr(self, name, lst)
        if self.sp == 1 or self.ring_mode:
            self.a_ctxl = self.a_ctx   # (without the head exchange the attention output is the wo input)

    def _p2p(self, t):
        """What travels between stages of a [T, hidden] residual-stream tensor: all of it, or (msp / fsp) this rank's token rows."""
        return t[self.rl] if self.ss else t

    def _forward_backward_pipeline(self, batch, labels):
        """One PipelineScheduler.forward_backward_step of this stage: warm-up forwards, one-forward-one-backward, cool-down backwards.
        Between stages travel the [T, hidden] residual stream (forward, received straight into the first layer's input buffer) and its
        gradient (backward, received straight into the buffer the layer backward reads); in the steady state a stage's send and the
        matching receive are ONE paired exchange.  Returns the loss of the step (from the last stage) on every stage."""
        tc, P = self.tc, self.pipe
        M = tc.micro_num
        if not hasattr(self, "_sets"):
            self._alloc_inflight_sets()
        self.loss_acc.zero_()
        ids_d = batch["input_ids"].to(self.dev, non_blocking=True)
        lab_d = labels.to(self.dev, non_blocking=True)
        pos_d = batch["indexes"].to(self.dev, non_blocking=True)
        if self.metric is not None and self.metric.ntypes:
            self.metric.set_current_type_ids(batch["type_ids"])

        lo, hi = self.seqpar.sp_rank * self.T, (self.seqpar.sp_rank + 1) * self.T   # this rank's tokens of every micro-batch (all of them without sp)
        pr = self._p2p

        def args(i):
            cu_h = batch["cu_seqlens"][i]
            self._ensure_rotary(int(batch["indexes"][i].max()) + 1)
            return ids_d[i, lo:hi], lab_d[i, lo:hi], self._dev_cu(cu_h), pos_d[i, lo:hi], int((cu_h[1:] - cu_h[:-1]).max())

        def forward(i):
            self._bind_inflight(i)
            self._forward_micro(*args(i))
            if P.last:
                self.loss_acc.add_(self.t_loss[0:1], alpha=1.0 / M)

        def backward(i):
            self._bind_inflight(i)
            return self._backward_micro(*args(i), i == M - 1, i == 0)

        def x_in(i):      # where micro-batch i's input of this stage is received
            return self._sets[i % len(self._sets)]["a_x"][0]

        warm = min(self.pp - P.stage - 1, M)
        rem = M - warm
        for i in range(warm):
            P.exchange(recvs=[] if P.first else [(pr(x_in(i)), P.prev)])
            forward(i)
            P.exchange(sends=[(pr(self.t_send), P.next)])          # (warm > 0 only on stages before the last)
        if rem > 0:
            P.exchange(recvs=[] if P.first else [(pr(x_in(warm)), P.prev)])
        for i in range(rem):
            forward(warm + i)
            if not P.last:                                      # send_forward_recv_backward
                P.exchange(sends=[(pr(self.t_send), P.next)], recvs=[(pr(self.t_h1), P.next)])
            g_in = backward(i)
            if i == rem - 1:
                P.exchange(sends=[] if P.first else [(pr(g_in), P.prev)])
            else:                                               # send_backward_recv_forward
                P.exchange(sends=[] if P.first else [(pr(g_in), P.prev)], recvs=[] if P.first else [(pr(x_in(warm + i + 1)), P.prev)])
        for i in range(rem, M):
            P.exchange(recvs=[(pr(self.t_h1), P.next)])
            g_in = backward(i)
            P.exchange(sends=[] if P.first else [(pr(g_in), P.prev)])
        return P.broadcast_from_last(self.loss_acc)

    def _forward_backward_interleaved(self, batch, labels):
        """One InterleavedPipelineScheduler.forward_backward_step of this stage (pipeline.py: interleaved_plan): the stage's num_chunks model
        chunks work through micro_num * num_chunks forward and backward micro-steps in the reference's order, on the common clock of the
        plan -- per tick at most one micro-step, then ONE paired exchange with whatever this stage sends and receives behind that tick.  A
        forward input is received straight into its micro-batch's activation set (the first layer of the chunk), an output gradient into
        that micro-batch's own buffer until its backward micro-step runs.  Returns the loss of the step on every stage."""
        tc, P = self.tc, self.pipe
        M, C = tc.micro_num, self.nch
        if not hasattr(self, "_plan"):
            self._plan = interleaved_plan(self.pp, C, M)[P.stage]
            # activation sets: a micro-batch owns one from its first arrival / forward to its last backward micro-step
            start, end = {}, {}
            for t, tick in enumerate(self._plan):
                seen = [m for k, m, c, _ in tick["recvs"] if k == "F"] + ([tick["op"][1]] if tick["op"] else [])
                for m in seen:
                    start.setdefault(m, t)
                if tick["op"] and tick["op"][0] == "B":
                    end[tick["op"][1]] = t
            free, busy, self._slot_of = [], [], {}
            for m in sorted(start, key=lambda m: (start[m], m)):
                for other in [o for o in busy if end[o] < start[m]]:
                    busy.remove(other)
                    free.append(self._slot_of[other])
                self._slot_of[m] = free.pop(0) if free else len(set(self._slot_of.values()))
                busy.append(m)
            n = len(set(self._slot_of.values()))
            # (with micro_num == pp every forward runs before the first backward, on the model's last chunk too: the final norm's saved
            # values, the logits and the loss rows are per micro-batch as well, not only the layers' activations)
            names = self._act_set_names() + (("a_xf", "a_nf", "a_rstdf", "t_logits", "t_lse", "t_loss_rows", "t_loss") if P.last else ())
            first = {name: getattr(self, name) for name in names}
            clone = lambda v: [torch.empty_like(t) for t in v] if isinstance(v, list) else torch.empty_like(v)
            self._sets = [first] + [{name: clone(v) for name, v in first.items()} for _ in range(n - 1)]
            h = self.lmc.hidden_size
            self.t_send = torch.empty(self.T, h, dtype=BF16, device=self.dev)
            self._gbuf = {}
        self.loss_acc.zero_()
        ids_d = batch["input_ids"].to(self.dev, non_blocking=True)
        lab_d = labels.to(self.dev, non_blocking=True)
        pos_d = batch["indexes"].to(self.dev, non_blocking=True)
        if self.metric is not None and self.metric.ntypes:
            self.metric.set_current_type_ids(batch["type_ids"])

        lo, hi = self.seqpar.sp_rank * self.T, (self.seqpar.sp_rank + 1) * self.T
        pr = self._p2p

        def args(i):
            cu_h = batch["cu_seqlens"][i]
            self._ensure_rotary(int(batch["indexes"][i].max()) + 1)
            return ids_d[i, lo:hi], lab_d[i, lo:hi], self._dev_cu(cu_h), pos_d[i, lo:hi], int((cu_h[1:] - cu_h[:-1]).max())

        def gbuf(m, c):
            key = (self._slot_of[m], c)
            if key not in self._gbuf:
                self._gbuf[key] = torch.empty(self.T, self.lmc.hidden_size, dtype=BF16, device=self.dev)
            return self._gbuf[key]

        last_v = self.pp * C - 1
        for tick in self._plan:
            g_in = None
            if tick["op"] is not None:
                kind, m, c = tick["op"]
                self._bind_inflight(self._slot_of[m])
                v = c * self.pp + P.stage
                if kind == "F":
                    self._forward_micro(*args(m), chunk=c)
                    if v == last_v:
                        self.loss_acc.add_(self.t_loss[0:1], alpha=1.0 / M)
                else:
                    if v != last_v:
                        self.t_h1.copy_(gbuf(m, c))
                    g_in = self._backward_micro(*args(m), m == M - 1, m == 0, chunk=c)
            sends = [(pr(self.t_send if k == "F" else g_in), P.stage_rank[to]) for k, m, c, to in tick["sends"]]
            recvs = [(pr(self._sets[self._slot_of[m]]["a_x"][self.chunks[c][0]] if k == "F" else gbuf(m, c)), P.stage_rank[frm]) for k, m, c, frm in tick["recvs"]]
            P.exchange(sends=sends, recvs=recvs)
        return P.broadcast_from_last(self.loss_acc)

    def forward_only(self, input_ids, labels, metric=None):
        """One evaluation batch = NonPipelineScheduler.forward_backward_step(forward_only=True) as evaluate_on_val_dls drives it
        (eval/evaluation.py:45-147, data_process_func = None): input_ids / labels [B, seq_len] host tensors, every row one
        (zero-padded) sequence, B a multiple of micro_bsz; micro-batches of micro_bsz rows run the training forward (same kernels,
        nothing saved for a backward that matters, gradients untouched).  `metric`: an AccPerplex that sees these logits instead
        of the training metric.  Returns the device scalar mean over micro-batches of the mean token loss."""
        tc = self.tc
        if self.pp > 1 and self.nch > 1:
            raise NotImplementedError("forward-only (evaluation) passes under the INTERLEAVED pipeline schedule: run validation with model.num_chunks = 1")
        B, S = input_ids.shape
        if S != tc.seq_len or B % tc.micro_bsz:
            raise ValueError(f"evaluation batch {tuple(input_ids.shape)}: rows must be seq_len = {tc.seq_len} long, their number a multiple of micro_bsz = {tc.micro_bsz}")
        M = B // tc.micro_bsz                      # micro-batches of the evaluation batch
        rows = tc.micro_bsz * self.mm              # rows of one pass (a merge_micro engine takes several micro-batches at once)
        npass = -(-B // rows)
        if npass * rows != B:                      # pad with empty rows: no label, so neither the loss nor the metric sees them
            pad = npass * rows - B
            input_ids = torch.cat([input_ids, input_ids.new_zeros(pad, S)])
            labels = torch.cat([labels, labels.new_full((pad, S), -100)])
        lo, hi = self.seqpar.sp_rank * self.T, (self.seqpar.sp_rank + 1) * self.T
        ids_d = input_ids.reshape(npass, -1).to(self.dev, non_blocking=True)
        lab_d = labels.reshape(npass, -1).to(self.dev, non_blocking=True)
        pos_d = torch.arange(S, dtype=torch.int64).repeat(rows).to(self.dev, non_blocking=True)
        cu = self._dev_cu(torch.arange(rows + 1, dtype=torch.int32) * S)
        self._ensure_rotary(S)
        out = torch.zeros(1, dtype=torch.float32, device=self.dev)
        train_metric = self.metric
        self.attach_metric(metric)  # allocates the argmax / nll rows on first use
        if self.pp > 1:
            # PipelineScheduler._forward_only_step (pipeline_scheduler.py:340-428): every micro-batch walks the stages once -- receive the residual
            # stream from the previous stage, run this stage's layers, send it on; the last stage holds logits, loss and metric, and its loss is
            # handed to every stage at the end (as the training step does)
            P = self.pipe
            if not hasattr(self, "_sets"):
                self._alloc_inflight_sets()
            try:
                for i in range(npass):
                    self._bind_inflight(0)
                    P.exchange(recvs=[] if P.first else [(self._p2p(self.a_x[0]), P.prev)])
                    self._forward_micro(ids_d[i, lo:hi], lab_d[i, lo:hi], cu, pos_d[lo:hi], S)
                    if P.last:
                        out.add_(self.t_loss[0:1], alpha=1.0 / M)
                    else:
                        P.exchange(sends=[(self._p2p(self.t_send), P.next)])
            finally:
                self.metric = train_metric
            return P.broadcast_from_last(out)
        self._bind_micro(0)
        try:
            for i in range(npass):
                if self.mm > 1:
                    nseg = min(self.mm, M - i * self.mm)   # real micro-batches of this pass
                    self._forward_micro(ids_d[i, lo:hi], lab_d[i, lo:hi], cu, pos_d[lo:hi], S, nseg)
                    out.add_(self.t_loss_seg[:nseg, 0].sum(dim=0, keepdim=True), alpha=1.0 / M)
                else:
                    self._forward_micro(ids_d[i, lo:hi], lab_d[i, lo:hi], cu, pos_d[lo:hi], S)
                    out.add_(self.t_loss[0:1], alpha=1.0 / M)
        finally:
            self.metric = train_metric
        return out

    def _forward_backward_merged(self, batch, labels):
        """merge_micro: the micro_num micro-batches of a step as ONE varlen pass over micro_num * packed_length tokens.  The
        micro-batches are independent until their gradients are summed, so stacking their tokens changes nothing but the shapes:
        every GEMM sees micro_num x the rows (whole rounds of 256x256 tiles on 256 CUs where 4096 tokens left 1.5 or 3.5, weights
        streamed once per step instead of micro_num times), the weight gradient contracts over all tokens at once, attention walks
        the concatenated cu_seqlens, the cross-entropy is normalised per micro-batch.  Gradients are summed in fp32 and rounded to
        bf16 once (as with batch_wgrad)."""
        tc, M, P = self.tc, self.mm, self.tc.packed_length
        ids_d = batch["input_ids"].reshape(-1).to(self.dev, non_blocking=True)
        lab_d = labels.reshape(-1).to(self.dev, non_blocking=True)
        pos_d = batch["indexes"].reshape(-1).to(self.dev, non_blocking=True)
        cus = batch["cu_seqlens"]
        cu_h = torch.cat([cus[0].to(torch.int32)] + [cus[i][1:].to(torch.int32) + i * P for i in range(1, M)])
        max_seqlen = int((cu_h[1:] - cu_h[:-1]).max())
        self._ensure_rotary(int(batch["indexes"].max()) + 1)
        if self.metric is not None and self.metric.ntypes:
            self.metric.set_current_type_ids(batch["type_ids"])
        cu = self._dev_cu(cu_h)
        self._forward_micro(ids_d, lab_d, cu, pos_d, max_seqlen)
        torch.sum(self.t_loss_seg[:, 0:1], dim=0, out=self.loss_acc)
        self.loss_acc.mul_(1.0 / M)
        self._backward_micro(ids_d, lab_d, cu, pos_d, max_seqlen, True, True)
        return self.loss_acc

    # ------------------------------------------------------------------------------------------ optimizer
    def step(self):
        """HybridZeroOptimizer.step + Engine.step (engine.py:105-126): norm, scaler, clip, AdamW, param sync,
        schedulers -- all stream-ordered, no host sync.  Returns nothing; read results with `read_state()`."""
        tc, L = self.tc, self.layout
        self.comm.wait_all()
        # squared grad norm over this rank's (already averaged) shards, then summed over ranks (compute_norm, utils.py:265-378)
        if self.wp_mode:   # the last layers' reduce-scatters out of the pool
            for slot in (0, 1):
                self._wp_finish_rs(slot)
        shards = [self._shard(self.grads, b) for b in L.buckets]
        if self.isp_rule > 1:
            self._apply_isp_grad_rule(shards)
        if self.isp_groups:
            self._step_isp_groups(shards)
            return
        K.sumsq([x for x in shards if x.numel()], self.sumsq, False, self.sumsq_ws)   # (empty: a bucket this pipeline stage does not own)
        if self.tp > 1:
            # compute_norm (solver/optimizer/utils.py:265-378) counts a parameter that is replicated over the tensor group (norm
            # weights; here also embedding and head) on ONE rank only: every rank of the group subtracts (1 - 1/tp) of its
            # (identical) contribution, then the squared norm is summed over the data-parallel AND the tensor group
            rep = self._replicated_grad_slices(shards)
            if rep:
                rs = K.sumsq(rep)
                # inf - inf would turn an overflow (inf: the scaler must back off) into NaN (which it treats differently):
                # an overflowing replicated gradient leaves the total at inf
                self.sumsq.sub_(torch.where(torch.isfinite(rs), rs * (1.0 - 1.0 / self.tp), torch.zeros_like(rs)))
        self.comm.all_reduce_sum(self.sumsq)
        self.tpar.all_reduce_sum(self.sumsq)
        self.pipe.all_reduce_sum(self.sumsq)   # the norm (and the overflow decision) covers all stages of the pipeline (compute_norm: MODEL group)
        K.step_control(self.state, self.sumsq, self.scaler_cfg)
        lr = self.lr_sched.lr()
        beta2 = self.beta2_sched.beta2()
        main = torch.cuda.current_stream(self.dev)
        ev = torch.cuda.Event()
        ev.record(main)
        opt_stream = main if os.environ.get("IE_SERIAL_ADAMW") == "1" else self.opt_stream   # (A/B switch: AdamW in line with the step)
        with torch.cuda.stream(opt_stream):
            opt_stream.wait_event(ev)  # gradients, norm and step control are final
            launched = 0
            for b, lo, gsh in zip(L.buckets, L.local_offsets(), shards):
                n = b.size // self.world
                if n == 0:   # a bucket this pipeline stage does not own
                    continue
                # the first buckets over the whole chip (the next forward has nothing to run until they are done), the others on adamw_cus CUs beside it
                if self.adamw_cus and opt_stream is not main:
                    K.tune_adamw_cus(self.adamw_cus if launched >= self.adamw_full_buckets else 0)
                launched += 1
                K.adamw_step(gsh, self.master[lo : lo + n], self.exp_avg[lo : lo + n], self.exp_avg_sq[lo : lo + n], self._shard(self.params, b),
                             self.state, lr, tc.adam_beta1, beta2, tc.adam_eps, tc.weight_decay)
                if b.index not in self._wp_buckets and self.comm.active:   # (a weight-parallel layer is gathered when it runs, into its pool slot)
                    self.comm.gathers[b.index] = self.comm.all_gather_async(self._full(self.params, b), self._shard(self.params, b))
                elif self.hold is not None:
                    self._hold(self._hold_ag, b)
                done = torch.cuda.Event()
                done.record(opt_stream)
                self._bucket_ready[b.index] = done
            self._opt_done = done
            if self.adamw_cus:
                K.tune_adamw_cus(0)
        if self.wp_mode:
            self._slot_layer = [None, None]   # the pool holds the weights of before this update
        # the all-gathers are NOT waited for here: the next forward waits per bucket (comm.wait_gather), so the parameter
        # exchange overlaps the next step's first layers; read_state()/named_parameters() drain them explicitly.
        # Engine.step steps the schedulers only after a successful update; success lives on the device, so the
        # host-side schedule advances optimistically and is corrected lazily when a skip is observed (read_state()).
        self.lr_sched.step()
        self.beta2_sched.step()
        self.step_count += 1

    def _group_pieces(self):
        """ISP's two optimizer groups inside this rank's bucket shards: [(bucket, offset inside the shard, length, group)] -- group 1 ("1_embed_head") =
        the embedding bucket and the head's part of the last bucket, group 0 ("0_default") = everything else (bucket padding is zero in every buffer and
        belongs to whichever piece it falls into)."""
        if getattr(self, "_pieces", None) is None:
            L, out = self.layout, []
            head = L.params.get("output.weight")   # (None on a pipeline stage other than the last: its last bucket is empty)
            for b in L.buckets:
                s0, n = b.shard(self.rank, self.world)
                if n == 0:
                    continue
                if b.index == 0:
                    out.append((b, 0, n, 1))
                elif b.index != len(L.buckets) - 1:
                    out.append((b, 0, n, 0))
                else:
                    cut = min(max(head.offset - s0, 0), n)   # elements of this shard in front of the head
                    if cut:
                        out.append((b, 0, cut, 0))
                    if n - cut:
                        out.append((b, cut, n - cut, 1))
            self._pieces = out
        return self._pieces

    def _step_isp_groups(self, shards):
        """step() of an ISP run: two group norms, one overflow decision, every group clipped by its own norm; AdamW per piece with its group's factor."""
        tc, L = self.tc, self.layout
        pieces = self._group_pieces()
        for grp in (0, 1):
            mine = [shards[b.index][o : o + n] for b, o, n, g_ in pieces if g_ == grp]
            if mine:
                K.sumsq(mine, self.sumsq_g[grp : grp + 1], False, self.sumsq_ws)
            else:   # (a middle pipeline stage holds neither embedding nor head)
                self.sumsq_g[grp : grp + 1].zero_()
        self.comm.all_reduce_sum(self.sumsq_g)
        self.pipe.all_reduce_sum(self.sumsq_g)   # every group's norm (and the overflow decision) covers all stages of the pipeline
        K.step_control_groups(self.state, self.sumsq_g, self.scaler_cfg, self.group_inv, self.group_norm)
        lr, beta2 = self.lr_sched.lr(), self.beta2_sched.beta2()
        main = torch.cuda.current_stream(self.dev)
        ev = torch.cuda.Event()
        ev.record(main)
        opt_stream = main if os.environ.get("IE_SERIAL_ADAMW") == "1" else self.opt_stream
        offs = L.local_offsets()
        with torch.cuda.stream(opt_stream):
            opt_stream.wait_event(ev)
            done = None
            for i, (b, o, n, grp) in enumerate(pieces):
                lo = offs[b.index] + o
                if self.adamw_cus and opt_stream is not main:   # (as in step(): the first buckets over the whole chip, the others beside the forward)
                    K.tune_adamw_cus(self.adamw_cus if b.index >= self.adamw_full_buckets else 0)
                K.adamw_step_group(shards[b.index][o : o + n], self.master[lo : lo + n], self.exp_avg[lo : lo + n], self.exp_avg_sq[lo : lo + n],
                                   self._shard(self.params, b)[o : o + n], self.state, self.group_inv[grp : grp + 1], lr, tc.adam_beta1, beta2, tc.adam_eps, tc.weight_decay)
                if i + 1 < len(pieces) and pieces[i + 1][0] is b:
                    continue   # (the bucket's second piece follows)
                if b.index not in self._wp_buckets and self.comm.active:
                    self.comm.gathers[b.index] = self.comm.all_gather_async(self._full(self.params, b), self._shard(self.params, b))
                done = torch.cuda.Event()
                done.record(opt_stream)
                self._bucket_ready[b.index] = done
            self._opt_done = done
            if self.adamw_cus:
                K.tune_adamw_cus(0)
        if self.wp_mode:
            self._slot_layer = [None, None]
        self.lr_sched.step()
        self.beta2_sched.step()
        self.step_count += 1

    def _replicated_grad_slices(self, shards):
        """Views of this rank's ZeRO gradient shards that belong to parameters held whole by every rank of the tensor group."""
        L = self.layout
        out = []
        for spec in L.params.values():
            if spec.kind not in ("embed", "norm", "head", "bo") or (spec.kind == "head" and self.vp) or (spec.kind == "embed" and self.embed_split):
                continue
            b = L.buckets[spec.bucket]
            s0, n0 = b.shard(self.rank, self.world)
            a, z = max(s0, spec.offset), min(s0 + n0, spec.offset + spec.numel)
            if a < z:
                out.append(self._shard(self.grads, b)[a - s0 : z - s0])
        return out

    def _apply_isp_grad_rule(self, shards):
        """The gradient averaging of the reference's ISP mode, as its code reads (sp = size of the sequence group):
          * ISPLinear weights: reduce-scatter AVG over the weight group (model/utils.py:556) then all-reduce AVG over
            WEIGHT_DATA (hybrid_zero_optim.py:98,169) = the mean over ALL ranks of per-rank gradients that each cover only
            1/sp of a micro-batch's tokens -> 1/sp of the data-parallel mean gradient;
          * norm weights (IS_REPLICA_ZERO_PARALLEL): AVG over the weight group (hybrid_zero_optim.py:318-324,
            solver/optimizer/utils.py:119), then the same bucket all-reduce -> also 1/sp;
          * embedding and head ("embed_head" group, train/utils.py:42-43, reduced over DATA): their gradients come from the
            gathered sequence (ops/linear.py:146-153, modules/embedding.py:52-60), so they are the full mean gradient.
        Here every rank's weight gradient covers its local tokens and the ZeRO reduce-scatter averages over all ranks, i.e.
        everything arrives as 1/sp of the mean gradient: embedding and head are multiplied back by sp (exact in bf16).
        With emulate_isp_grad_rule (no sequence parallelism) the gradients arrive as the full mean: divide all, then same."""
        n_ = float(self.isp_rule)
        L = self.layout
        if self.sp == 1:
            for sh in shards:
                K.scale_bf16(sh, 1.0 / n_)
        for name in ("tok_embeddings.weight", "output.weight"):
            spec = L.params.get(name)
            if spec is None:   # (another pipeline stage holds it)
                continue
            for b, sh in zip(L.buckets, shards):
                s0, n0 = b.shard(self.rank, self.world)
                a, z = max(s0, spec.offset), min(s0 + n0, spec.offset + spec.numel)
                if a < z:
                    K.scale_bf16(sh[a - s0 : z - s0], n_)

    def _wait_bucket(self, b):
        """Order the current stream behind the optimizer-stream work on bucket b (AdamW, and its all-gather when world > 1)."""
        ev = self._bucket_ready[b]
        if ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(ev)
            self._bucket_ready[b] = None
        self.comm.wait_gather(b)

    def _wait_optimizer(self):
        if self._opt_done is not None:
            torch.cuda.current_stream(self.dev).wait_event(self._opt_done)
            self._opt_done = None

    def drain(self):
        """Order the current stream behind everything the last step() left running (call before touching eng.params directly)."""
        if getattr(self, "wp_mode", False) and hasattr(self, "_wp_rs"):
            for l in list(self._wp_inflight):
                w = self._wp_inflight.pop(l)
                if w is not None:
                    w.wait()
            for slot in (0, 1):
                self._wp_finish_rs(slot)
            self._slot_layer = [None, None]
        for b in range(len(getattr(self, "_bucket_ready", ()))):  # (also called while the constructor is still loading the weights)
            self._wait_bucket(b)
        if getattr(self, "_opt_done", None) is not None:
            self._wait_optimizer()
        self.comm.wait_all_gathers()

    def read_state(self):
        """Host copy of the step state (synchronises).  Also rewinds the host schedulers for skipped steps."""
        self.drain()
        st = K.step_state_read(self.state)
        self.lr_sched.set_successful_steps(st.adam_step)
        self.beta2_sched.set_successful_steps(st.adam_step)
        if self.isp_groups:   # the reference reports one norm per optimizer group (hybrid_zero_optim.py:801-803); st.grad_norm is the norm over both
            gn = [float(x) for x in self.group_norm.cpu()]
            st.group_norms = {"0_default": gn[0], "1_embed_head": gn[1]}
        return st

    # ------------------------------------------------------------------------------------------ utilities
    # LLAMA2 (modeling_llama.py:126-148) keeps wq / wk / wv as separate parameters; the kernels work on one projection in
    # InternLM2's layout [kv group][q_per_kv q heads, k, v][head_dim] (same product, rows permuted), so the engine stores that and
    # converts at the naming boundary (named_parameters, load_named_parameters, checkpoints).
    def _is_llama(self):
        return getattr(self.mc, "model_type", "INTERNLM2_PUBLIC") == "LLAMA2"

    # The dense InternLM-1 model (model_type INTERNLM: modeling_internlm.py, the model configs/7B_sft.py and configs/7B_isp_sft.py build) is the same
    # block with other parameter names, a Wqkv packed "(three h d)" (multi_head_attention.py:428-431) and biases on Wqkv / out_proj.  The engine keeps
    # its own names and InternLM2's row order -- with one q head per kv head [kv group][q, k, v][d] is [h][three][d] -- and converts at the naming boundary.
    _V1_LAYER = {"attention_norm.weight": "norm1.weight", "attention.wqkv.weight": "mixer.Wqkv.weight", "attention.wqkv.bias": "mixer.Wqkv.bias",
                 "attention.wo.weight": "mixer.out_proj.weight", "attention.wo.bias": "mixer.out_proj.bias", "ffn_norm.weight": "norm2.weight",
                 "feed_forward.w1.weight": "mlp.w1.weight", "feed_forward.w2.weight": "mlp.w2.weight", "feed_forward.w3.weight": "mlp.w3.weight"}
    _V1_TOP = {"tok_embeddings.weight": "embedding.weight", "norm.weight": "norm.weight", "output.weight": "head.weight"}

    def _is_v1(self):
        return getattr(self.mc, "model_type", "INTERNLM2_PUBLIC") == "INTERNLM"

    def _v1_name(self, n, to_reference):
        top = self._V1_TOP if to_reference else {v: k for k, v in self._V1_TOP.items()}
        if n in top:
            return top[n]
        src, dst = ("layers.", "blocks.") if to_reference else ("blocks.", "layers.")
        lay = self._V1_LAYER if to_reference else {v: k for k, v in self._V1_LAYER.items()}
        assert n.startswith(src), n
        l, rest = n[len(src):].split(".", 1)
        return f"{dst}{l}.{lay[rest]}"

    def _v1_qkv(self, t, to_reference):
        """Wqkv weight [3 H d, h] / bias [3 H d] of H heads (all of them, or a tensor rank's): engine rows [H][three][d] <-> reference rows "(three h d)"."""
        d = self.mc.head_dim
        H = t.shape[0] // (3 * d)
        v = t.reshape(H, 3, d, -1) if to_reference else t.reshape(3, H, d, -1)
        return v.permute(1, 0, 2, 3).reshape(t.shape)

    def _split_wqkv(self, t):
        mc = self.mc
        qpk = mc.q_per_kv
        v = t.reshape(-1, qpk + 2, mc.head_dim, t.shape[-1])  # [kv groups (all, or this rank's under tensor parallelism)][q.., k, v][d][h]
        return (v[:, :qpk].reshape(-1, t.shape[-1]), v[:, qpk].reshape(-1, t.shape[-1]), v[:, qpk + 1].reshape(-1, t.shape[-1]))

    def _fuse_wqkv(self, wq, wk, wv):
        mc = self.mc
        qpk, d = mc.q_per_kv, mc.head_dim
        hkv = wk.shape[0] // d
        return torch.cat([wq.reshape(hkv, qpk, d, -1), wk.reshape(hkv, 1, d, -1), wv.reshape(hkv, 1, d, -1)], dim=1).reshape(hkv * (qpk + 2) * d, -1)

    def _to_reference_names(self, named):
        """engine names -> the reference's parameter names (a copy for the LLAMA2 projections / the InternLM-1 Wqkv, the same tensors otherwise)."""
        if self._is_v1():
            return {self._v1_name(n, True): (self._v1_qkv(t, True) if ".attention.wqkv." in n else t) for n, t in named.items()}
        if not self._is_llama():
            return dict(named)
        out = {}
        for n, t in named.items():
            if n.endswith("attention.wqkv.weight"):
                pre = n[: -len("wqkv.weight")]
                out[pre + "wq.weight"], out[pre + "wk.weight"], out[pre + "wv.weight"] = self._split_wqkv(t)
            else:
                out[n] = t
        return out

    def _from_reference_names(self, named):
        if self._is_v1():
            return {self._v1_name(n, False): (self._v1_qkv(t, False) if ".mixer.Wqkv." in n else t) for n, t in named.items()}
        if not self._is_llama():
            return dict(named)
        out = {n: t for n, t in named.items() if not n.endswith(("attention.wq.weight", "attention.wk.weight", "attention.wv.weight"))}
        for l in range(self.mc.num_layers):
            pre = f"layers.{l}.attention."
            if pre + "wq.weight" in named:
                out[pre + "wqkv.weight"] = self._fuse_wqkv(named[pre + "wq.weight"], named[pre + "wk.weight"], named[pre + "wv.weight"])
        return out

    def reference_param_shapes(self):
        """Shapes of the FULL (un-sharded) parameters under the reference's names."""
        shapes = {n: s.shape for n, s in FlatLayout(self.mc, 1).params.items()}
        if self._is_v1():
            return {self._v1_name(n, True): shp for n, shp in shapes.items()}
        if self._is_llama():
            mc, out = self.mc, {}
            for n, shp in shapes.items():
                if n.endswith("attention.wqkv.weight"):
                    pre = n[: -len("wqkv.weight")]
                    out[pre + "wq.weight"] = (mc.num_attention_heads * mc.head_dim, shp[1])
                    out[pre + "wk.weight"] = (mc.num_kv_attention_heads * mc.head_dim, shp[1])
                    out[pre + "wv.weight"] = (mc.num_kv_attention_heads * mc.head_dim, shp[1])
                else:
                    out[n] = shp
            return out
        return shapes

    def named_parameters(self):
        """(reference name, bf16 tensor) pairs -- views of the flat buffer, except LLAMA2's wq / wk / wv (copies)."""
        self.drain()
        if self.wp_mode:
            return self._to_reference_names(self._gathered_params()).items()
        return self._to_reference_names(self.p).items()

    def _gathered_params(self):
        """Weight parallelism: every parameter whole (engine names) -- the resident ones as views, the layer parameters as COPIES out of a
        collective all-gather per layer bucket (tests, reporting; never on the step's path)."""
        L, out = self.layout, {}
        for b in L.buckets:
            if b.index in self._wp_buckets:
                full = torch.empty(b.size, dtype=BF16, device=self.dev)
                if self.comm.active:
                    self.comm.all_gather_async(full, self._shard(self.params, b)).wait()
                else:
                    full.copy_(self._shard(self.params, b))
            for n in b.params:
                spec = L.params[n]
                out[n] = full[spec.offset - b.start : spec.offset - b.start + spec.numel].view(spec.shape) if b.index in self._wp_buckets else self.p[n]
        return out

    # ------------------------------------------------------------------------------------------ checkpoints
    def _named_shard_views(self, flat_local):
        """name -> view of this rank's fp32 state for a parameter (world 1: the whole parameter)."""
        L = self.layout
        out = {}
        for b, lo in zip(L.buckets, L.local_offsets()):
            for n in b.params:
                s = L.params[n]
                out[n] = flat_local[lo + (s.offset - b.start) : lo + (s.offset - b.start) + s.numel].view(s.shape)
        return out

    def _engine_name(self, ref_name):
        """the reference's parameter name -> the engine parameter that holds it (LLAMA2: wq / wk / wv live in the fused wqkv)."""
        if self._is_llama() and ref_name.endswith(("attention.wq.weight", "attention.wk.weight", "attention.wv.weight")):
            return ref_name[: -len("wq.weight")] + "wqkv.weight"
        if self._is_v1():
            return self._v1_name(ref_name, False)
        return ref_name

    def _shard_pieces(self):
        """[(name, start inside the parameter, length, offset inside this rank's fp32 state)]: what this rank's contiguous bucket
        slices hold of every parameter (world 1: every parameter whole)."""
        L = self.layout
        out = []
        for b, lo in zip(L.buckets, L.local_offsets()):
            ss = b.size // self.world
            s0 = b.start + self.rank * ss
            for n in b.params:
                spec = L.params[n]
                a, e = max(spec.offset, s0), min(spec.offset + spec.numel, s0 + ss)
                if a < e:
                    out.append((n, a - spec.offset, e - a, lo + (a - s0)))
        return out

    def _isp_layout(self):
        """Sequence parallelism (tensor mode "isp") and / or weight parallelism: checkpoints take the reference's ISP layout."""
        return self.sp != 1 or self.wp_mode

    def _checkpoint_guard(self):
        if self._isp_layout():
            if self.pp != 1 or self.tp != 1:
                raise NotImplementedError("checkpoints of the ISP layout (tensor mode 'isp' / parallel.weight) without pipeline parallelism")
            zs_cfg = self.tc.zero1_size
            wp = max(int(getattr(self.tc, "wp_size", 1) or 1), 1)
            if self.job_world % wp or not (zs_cfg is None or zs_cfg <= 0 or zs_cfg >= self.job_world // wp):
                raise NotImplementedError("checkpoints of the ISP layout with parallel.zero1.size below the weight-data size")
            return

    def _local_reference_named(self, named):
        """engine-named tensors of this rank -> the reference's names AND the reference's tensor-parallel cut: the layer weights are
        already this tensor rank's part (same rule, tensorpar.py); embedding and head, which this engine keeps whole on every rank,
        are cut the way the reference's modules hold them (hidden columns / vocabulary rows, checkpoint.tp_shard)."""
        from . import checkpoint as C

        out = self._to_reference_names(named)
        if self.tp > 1:
            emb, head = ("embedding.weight", "head.weight") if self._is_v1() else ("tok_embeddings.weight", "output.weight")
            whole = ([] if self.embed_split else [emb]) + ([] if self.vp else [head])
            for n in whole:   # (what this engine keeps whole is cut the reference's way for its files; what it holds cut already is written as is)
                if n in out:
                    out[n] = C.tp_shard(n, out[n], self.tpar.tp_rank, self.tp)
            if self._is_v1() and self.tpar.tp_rank != 0:   # a row-parallel linear's bias lives on tensor rank 0 only (ops/linear.py:317-324); this engine keeps it replicated
                out = {n: t for n, t in out.items() if not n.endswith("mixer.out_proj.bias")}
        return out

    def save_model_isp(self, folder):
        """The MODEL files of the reference's ISP layout (checkpoint/components.py:221-226; checkpoint.save_isp_model_shard): `model_tp{t}_wp{w}_pp0.pt` with
        the embedding's hidden columns / the head's vocabulary rows of tensor (= sequence) rank t and the ISPLinear rows of weight rank w, written by the
        ranks the reference writes from (weight-data rank 0 or data rank 0); a reference job of the same tensor x weight sizes -- or this engine, in any
        layout -- loads them with load_ckpt_info content = ("model",).  Collective.  (save_checkpoint under sequence / weight parallelism writes these AND
        the optimizer shards of the layout.)"""
        from . import checkpoint as C

        if self.tp != 1 or self.pp != 1:
            raise NotImplementedError("save_model_isp: the ISP layout (tensor mode 'isp' + parallel.weight), without tensor mode mtp / msp / fsp or pipeline parallelism")
        sp, wp = self.sp, max(int(getattr(self.tc, "wp_size", 1) or 1), 1)
        named = {n: t.detach().to("cpu", copy=True) for n, t in self.named_parameters()}   # (collective under weight parallelism: every rank gathers)
        r = self.job_rank
        if r // wp == 0 or r // sp == 0:
            C.save_isp_model_shard(folder, self.mc, named, r % sp, sp, r % wp, wp)
        self.comm.barrier()

    def _save_checkpoint_isp(self, folder):
        """save_checkpoint under sequence / weight parallelism: the reference's ISP layout (checkpoint.py: save_isp_model_shard / save_isp_optimizer_shard) --
        every rank writes `optimizer_tp{t}_wp{w}_pp0_dp{d}.pt` (three groups: the partitions it would hold in a reference job of this size, of ITS local
        shards) + its plan file, the ranks with weight-data rank 0 or data rank 0 the model files.  The engine's fp32 state lives in contiguous bucket slices
        of the zero group, so every bucket is gathered once (collective) and every rank keeps the rows its partitions name.  Collective."""
        from . import checkpoint as C

        st = self.read_state()
        tc, L = self.tc, self.layout
        W, r = self.job_world, self.job_rank
        sp, wp = self.sp, max(int(getattr(tc, "wp_size", 1) or 1), 1)
        if r == 0:   # files of another layout (or of a larger ISP job) in the folder would be loaded instead of / merged into this save
            gone = C.remove_stale_shards(folder, 1, sp, 1, job_world=W, layout="isp", wp_world=wp, dp_world=W // sp)   # (dp shards / plans d >= W / sp of a smaller-sp save go too)
            if gone:
                print(f"[internevo_amd] save_checkpoint({folder}): removed {len(gone)} files of an earlier layout: {', '.join(gone)}", flush=True)
        if W > 1:                # (the job is the default group under ISP: behind this barrier every rank has been here, nobody writes before the cleanup)
            import torch.distributed as dist

            dist.barrier()
        full_shapes = self.reference_param_shapes()
        mine = {self._engine_name(n) for n in C.isp_rank_names(self.mc, full_shapes, r, W, sp, wp)}
        state = {}
        for key, flat in (("master", self.master), ("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
            named = {}
            for bi, (b, lo) in enumerate(zip(L.buckets, L.local_offsets())):   # every rank walks every bucket: the gather is collective
                full = self.comm.gather_full_bucket(flat, lo, bi)
                for n in b.params:
                    spec = L.params[n]
                    if n in mine:
                        named[n] = full[spec.offset - b.start : spec.offset - b.start + spec.numel].view(spec.shape).to("cpu", copy=True)
                    else:   # (only its shape is needed: the partition is computed from the shapes of ALL parameters)
                        named[n] = torch.empty(spec.shape, dtype=torch.float32, device="meta")
                del full
            state[key] = self._to_reference_names(named)   # (the naming boundary's row permutations are views: fine on the shape-only meta tensors too)
        hyper = dict(weight_decay=tc.weight_decay, betas=(tc.adam_beta1, tc.adam_beta2), eps=tc.adam_eps, initial_lr=tc.lr)
        scaler = dict(scale=st.loss_scale, growth_step=st.growth_step, hysteresis_step=st.hysteresis_step)
        C.save_isp_optimizer_shard(folder, self.mc, r, W, sp, wp, state["master"], state["exp_avg"], state["exp_avg_sq"], st.adam_step, scaler, self.lr_sched.lr(), hyper)
        self.save_model_isp(folder)

    def save_checkpoint(self, folder):
        """InternEvo's checkpoint files (checkpoint.py): per tensor rank (and pipeline stage) the model weights (written by its data-parallel
        rank 0) and one hybrid-ZeRO optimizer shard + partition plan per data-parallel rank, in the reference's whole-parameter partition
        (hybrid_zero_optim.py:254-284).  A pipeline stage writes `..._pp{stage}...` files whose layers are numbered from 0, as the reference's
        stage does (tests/golden/ckpt_ref_pp2/).  Collective."""
        from . import checkpoint as C

        self._checkpoint_guard()
        if self._isp_layout():
            return self._save_checkpoint_isp(folder)
        st = self.read_state()  # drains the optimizer stream
        tc, L, W, r = self.tc, self.layout, self.world, self.rank
        tp, t = self.tp, self.tpar.tp_rank
        pp, ps = self.pp, self.pipe.stage
        job = self.dp_world * tp * pp   # the ranks of the job: under hybrid ZeRO (W < dp_world) EVERY data-parallel rank writes a plan file named after it (ckpt_ref_dp4_zo2/)
        if self.dp_rank == 0 and t == 0 and ps == 0:   # shards of an earlier, larger layout in the same folder would be merged into this save by any loader
            gone = C.remove_stale_shards(folder, W, tp, pp, job_world=job)
            if gone:
                print(f"[internevo_amd] save_checkpoint({folder}): removed {len(gone)} shard files of an earlier, larger layout: {', '.join(gone)}", flush=True)

        def everyone():   # the data-parallel ranks of this stage, then the stages of this pipeline: behind it every rank of the job has been here
            self.comm.barrier()
            self.tpar.barrier()
            self.pipe.barrier()

        everyone()
        hyper = dict(weight_decay=tc.weight_decay, betas=(tc.adam_beta1, tc.adam_beta2), eps=tc.adam_eps, initial_lr=tc.lr)
        scaler = dict(scale=st.loss_scale, growth_step=st.growth_step, hysteresis_step=st.hysteresis_step)
        # this stage's parameters under the reference's names, module order; inside the files a stage numbers its layers from 0
        # (under the interleaved schedule a stage's files hold its model chunks, "<chunk>.<name>", every chunk numbering its layers from 0: checkpoint.stage_naming)
        naming = C.stage_naming(self.mc, pp, ps, self.nch if self.nch > 1 else 0, t)
        assert all(self._engine_name(g) in self.p for _, _, g in naming), "this stage's parameters are not the ones its files are named for"
        glob = [g for _, _, g in naming]
        to_local, to_global = {g: n for n, _, g in naming}, {n: g for n, _, g in naming}
        loc = to_local.__getitem__
        stage = dict(pp_world=pp, pp_rank=ps, order=[loc(n) for n in glob], chunked=self.nch > 1) if pp > 1 else {}
        cpu = lambda d: {loc(n): x.detach().to("cpu") for n, x in self._local_reference_named(d).items()}  # noqa: E731
        shapes = {}
        for n, shp in self.reference_param_shapes().items():  # FULL shapes -> this tensor rank's local shapes
            d = C.tp_split_dim(n)
            shapes[n] = tuple(x // tp if (tp > 1 and i == d) else x for i, x in enumerate(shp))
        shapes = {loc(n): shapes[n] for n in glob}
        plans = dict(job_world=job, dp_ranks=[self.dp_rank])   # this data-parallel rank's plan file, named after the job's world size and its ranks
        if W == 1:
            if self.dp_rank == 0:
                C.save_checkpoint(folder, self.mc, cpu(self.p), cpu(self._named_shard_views(self.master)), cpu(self._named_shard_views(self.exp_avg)),
                                  cpu(self._named_shard_views(self.exp_avg_sq)), st.adam_step, scaler, self.lr_sched.lr(), hyper, tp_world=tp, tp_rank=t, **stage, **plans)
            else:   # (zero1.size = 1 under data parallelism: rank 0 holds and writes the state, every rank its plan file)
                C.save_checkpoint(folder, self.mc, None, None, None, None, st.adam_step, scaler, self.lr_sched.lr(), hyper, shapes=shapes, tp_world=tp, tp_rank=t,
                                  plans_only=True, **stage, **plans)
            everyone()
            return
        mine = C.zero_rank_names(shapes, W)[r]              # the parameters the reference's ZeRO rank r owns (whole; stage-local names)
        need = {self._engine_name(to_global[n]) for n in mine}
        state = {}
        for key, flat in (("master", self.master), ("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
            named = {}
            for bi, (b, lo) in enumerate(zip(L.buckets, L.local_offsets())):  # every rank walks every bucket: the gather is collective
                full = self.comm.gather_full_bucket(flat, lo, bi)
                for n in b.params:
                    if n in need:
                        spec = L.params[n]
                        named[n] = full[spec.offset - b.start : spec.offset - b.start + spec.numel].view(spec.shape).to("cpu", copy=True)
                del full
            ref_named = self._local_reference_named(named)
            state[key] = {n: ref_named[to_global[n]] for n in mine}
        if self.comm.replica == 0:   # hybrid ZeRO: every zero group holds the same shards, the first one writes them
            C.save_checkpoint(folder, self.mc, cpu(self.p) if r == 0 else None, state["master"], state["exp_avg"], state["exp_avg_sq"], st.adam_step,
                              scaler, self.lr_sched.lr(), hyper, zero_world=W, zero_ranks=[r], write_model=(r == 0), shapes=shapes, tp_world=tp, tp_rank=t, **stage, **plans)
        else:                        # ... the others their plan files (hybrid_zero_optim.py:133-140: one per data-parallel rank)
            C.save_checkpoint(folder, self.mc, None, None, None, None, st.adam_step, scaler, self.lr_sched.lr(), hyper, zero_world=W, shapes=shapes, tp_world=tp, tp_rank=t,
                              plans_only=True, **stage, **plans)
        everyone()  # the folder is complete when any rank returns

    def load_checkpoint(self, folder, model_only=False):
        """Resume from InternEvo checkpoint files (written by the reference or by save_checkpoint, by ANY ZeRO-1 world, ANY
        tensor-parallel size and ANY pipeline size: the shards are merged into full tensors under the model's own names and re-cut into this
        engine's stage, tensor-parallel parts and bucket slices).  model_only: load_ckpt_info content = ("model",) -- the weights only; the fp32 master
        copy is rebuilt from them (reload_zero_fp32_buff, checkpoint_manager.py:553-557), optimizer moments, step and loss scale stay as they are."""
        from . import checkpoint as C

        if model_only:   # nothing of the optimizer's layout is touched: also under sequence / weight parallelism, where full checkpoints are refused
            if self._is_v1() and self.tp != 1:
                self._checkpoint_guard()
            ck = C.load_checkpoint(folder, self.mc, want=set(), model_only=True)
            self.load_named_parameters(ck["params"])   # every rank keeps its stage's layers / its tensor-parallel cut / its weight-parallel shard
            return
        self._checkpoint_guard()
        pieces = self._shard_pieces()
        want = set()
        for n in {p[0] for p in pieces}:
            want.update(self._to_reference_names({n: self.p[n]}).keys())
        ck = C.load_checkpoint(folder, self.mc, want=want)
        self.drain()
        kind = lambda n: self.layout.params[n].kind  # noqa: E731
        for n, full in self._from_reference_names(ck["params"]).items():
            if n in self.p:   # (a pipeline stage keeps its own layers; under weight parallelism a rank keeps its shard of a layer)
                self._store_param(n, self.tpar.shard(kind(n), full).to(self.dev, BF16))
        if ck["master"] is None:
            self.sync_master_from_params()
            return
        for flat, key in ((self.master, "master"), (self.exp_avg, "exp_avg"), (self.exp_avg_sq, "exp_avg_sq")):
            src = {n: self.tpar.shard(kind(n), full).reshape(-1) for n, full in self._from_reference_names(ck[key]).items() if n in self.p}
            for n, a, k, lo in pieces:
                flat[lo : lo + k].copy_(src[n][a : a + k].to(self.dev))
        st = K.step_state_read(self.state)
        st.loss_scale, st.growth_step, st.hysteresis_step = ck["scaler"]["scale"], ck["scaler"]["growth_step"], ck["scaler"]["hysteresis_step"]
        st.adam_step, st.skip, st.found_inf, st.found_nan = ck["adam_step"], 0, 0, 0
        self.state.copy_(torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).to(self.dev))
        self.lr_sched.set_successful_steps(ck["adam_step"])
        self.beta2_sched.set_successful_steps(ck["adam_step"])
        self.step_count = ck["adam_step"]

    def load_named_parameters(self, named, sync_master=True):
        """named: the reference's FULL parameter tensors by name; under tensor parallelism every rank keeps its shard."""
        self.drain()
        for n, t in self._from_reference_names(named).items():
            if n in self.p:   # (a pipeline stage keeps its own layers)
                self._store_param(n, self.tpar.shard(self.layout.params[n].kind, t).to(self.dev, BF16))
        if sync_master:
            self.sync_master_from_params()
