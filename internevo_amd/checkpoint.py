"""InternEvo's on-disk checkpoint format, single model-parallel shard (tp = pp = 1), any ZeRO-1 world.

Written / read so that a run can be handed over between the reference and this engine in both directions:

    <folder>/model_tp0_pp0.pt                      torch.save(OrderedDict), keys "model.<param>" in module order, model dtype
    <folder>/topo_tp0_pp0.json                     torch.save({}) (yes, a torch zip under a .json name)
    <folder>/optimizer_tp0_pp0_zo0.pt              HybridZeroOptimizer.state_dict():
        grad_scaler        {_scale (python float), _growth_step, _hysteresis_step}
        base_optim_states  torch.optim.AdamW.state_dict() over ONE flat fp32 parameter per group:
                           state {0: {step (0-d fp32), exp_avg [P], exp_avg_sq [P]}}, param_groups [default, fp32]
        flat_fp32_weights  {0: fp32 [P]}  -- the master weights
        zero_devide_optim_plan  as below
    <folder>/gpus-1_wp-0_tp-0_dp-0_pp-0_zo-0.pt   zero_devide_optim_plan: [[["<i>_<shape>" ...]], [[]]]

    <folder>/schedulder.pt (sic), sampler.pt, context.pt   the run state written by rank 0 (checkpoint_manager.py:608-618):
        the LR scheduler's state_dict, the batch sampler's (generator state, position, this epoch's order), TrainState's counters

(internlm/checkpoint/components.py:199-283,377-410; solver/optimizer/hybrid_zero_optim.py:133-140,254-284,882-936.)
The flat vectors hold the group's parameters in the ZeRO partition order: the module-order parameter list stably sorted by
numel, largest first, each parameter handed WHOLE to the rank that holds the fewest elements so far (`_partition_param_list`);
ZeRO world W gives W optimizer files `optimizer_tp0_pp0_zo{r}.pt` and W plan files `gpus-{W}_wp-0_tp-0_dp-{r}_pp-0_zo-{r}.pt`,
every one carrying the plan of ALL ranks.  (This engine shards each bucket into contiguous slices instead; engine.py re-cuts
the state on the way in and out, so a checkpoint can also be loaded into a different data-parallel size.)  `param_groups[*]["optimizer_mode"]` is the reference's ParallelMode
enum, pickled by reference: it is written through a stand-in module of the same dotted name when `internlm` is not importable
and read through a tolerant unpickler, so neither side needs the other installed.

Everything here works on named host tensors; `InternLM2Engine.save_checkpoint / load_checkpoint` (engine.py) map them onto
the flat device buffers.  Pinned by tests/golden/ckpt_ref/ (written by the real reference, tests/golden/make_golden.py --ckpt).
"""
import collections
import enum
import os
import pickle
import sys
import types

import torch

_PM_MODULE = "internlm.core.context.process_group_initializer"


def state_dict_order(model_cfg, tp_rank=0):
    """Parameter names in the reference's module order (PackedFlashLlama1D.state_dict()) on tensor-parallel rank tp_rank."""
    if getattr(model_cfg, "model_type", "INTERNLM2_PUBLIC") == "INTERNLM":   # PackedFlashInternLm1D (modeling_internlm.py; pinned by tests/golden/ckpt_v1.json)
        names = ["embedding.weight"]
        for l in range(model_cfg.num_layers):
            p = f"blocks.{l}."
            # (a row-parallel linear's bias exists on tensor rank 0 only, ops/linear.py:317-324; pinned by tests/golden/ckpt_v1tp2_rank1.json)
            names += [p + "mixer.Wqkv.weight", p + "mixer.Wqkv.bias", p + "mixer.out_proj.weight"] + ([p + "mixer.out_proj.bias"] if tp_rank == 0 else [])
            names += [p + "norm1.weight", p + "norm2.weight", p + "mlp.w1.weight", p + "mlp.w2.weight", p + "mlp.w3.weight"]
        return names + ["norm.weight", "head.weight"]
    if getattr(model_cfg, "model_type", "INTERNLM2_PUBLIC") == "INTERNLM_MoE":   # PackedFlashInternLm1D of modeling_moe.py (pinned by tests/golden/ckpt_moe.json)
        names = ["embedding.weight"]
        for l in range(model_cfg.num_layers):
            p = f"blocks.{l}."
            names += [p + "mixer.Wqkv.weight", p + "mixer.Wqkv.bias", p + "mixer.out_proj.weight"] + ([p + "mixer.out_proj.bias"] if tp_rank == 0 else [])
            names += [p + "norm1.weight", p + "norm2.weight", p + "mlp.moe_layer.gate.wg.weight"]
            for e in range(model_cfg.num_experts):
                names += [p + f"mlp.moe_layer.experts.wrapped_experts.{e}.w{k}.weight" for k in (1, 2, 3)]
        return names + ["norm.weight", "head.weight"]
    names = ["tok_embeddings.weight"]
    llama = getattr(model_cfg, "model_type", "INTERNLM2_PUBLIC") == "LLAMA2"  # modeling_llama.py: wq, wk, wv instead of wqkv
    for l in range(model_cfg.num_layers):
        p = f"layers.{l}."
        names += ([p + "attention.wq.weight", p + "attention.wk.weight", p + "attention.wv.weight"] if llama else [p + "attention.wqkv.weight"])
        names += [p + "attention.wo.weight", p + "attention_norm.weight", p + "ffn_norm.weight",
                  p + "feed_forward.w1.weight", p + "feed_forward.w2.weight", p + "feed_forward.w3.weight"]
    return names + ["norm.weight", "output.weight"]


def stage_order(model_cfg, n_layers, first, last, tp_rank=0):
    """Names in ONE pipeline stage's state dict (`model_tp{t}_pp{p}.pt`): the stage numbers its layers from 0 (modeling_internlm2.py:897-925), the
    embedding lives on the first stage, norm + head on the last (pinned by tests/golden/ckpt_pp2_rank*.json)."""
    import dataclasses

    full = state_dict_order(dataclasses.replace(model_cfg, num_layers=n_layers), tp_rank)
    head, tail = full[:1], full[-2:]
    return (head if first else []) + full[1:-2] + (tail if last else [])


def stage_to_global(name, layer_lo):
    """A stage-local parameter name -> the model's name (layer number + the stage's first layer)."""
    import re

    return re.sub(r"^(layers|blocks)\.(\d+)\.", lambda m: f"{m.group(1)}.{int(m.group(2)) + layer_lo}.", name)   # (InternLM2 / InternLM-1 names)


def global_to_stage(name, layer_lo):
    """The model's parameter name -> the name inside the stage whose first layer is layer_lo."""
    return stage_to_global(name, -layer_lo)


def stage_chunks(sd):
    """Number of model chunks in a stage's state dict: 0 = one model (keys "model.<name>"), c > 0 = the interleaved pipeline schedule's ModuleList of c chunks
    (keys "<chunk>.model.<name>", every chunk numbering its layers from 0; modeling_internlm2.py:1012-1050, pinned by tests/golden/ckpt_pp2i_rank*.json)."""
    import re

    idx = [int(m.group(1)) for m in (re.match(r"(\d+)\.model\.", k) for k in sd) if m]
    return max(idx) + 1 if idx else 0


def stage_naming(model_cfg, pp_world, pp_rank, chunks=0, tp_rank=0, layer_counts=None):
    """[(name inside the stage's files, key of its model state dict, the model's own name)] of one pipeline stage, in the stage's module order (= the order the
    stage's ZeRO partition is computed in).  chunks = 0: one model per stage, its layers numbered from 0 (partition_uniform with one chunk).  chunks >= 1
    (interleaved schedule): chunk c of stage p holds the layers partition_chunks(...)[p][c]; inside the files its names carry the prefix "<c>.".
    layer_counts: the layers found in the files (per chunk), checked against the partition."""
    from .pipeline import partition_chunks, partition_uniform

    L = model_cfg.num_layers
    if pp_world == 1 and not chunks:
        return [(n, "model." + n, n) for n in state_dict_order(model_cfg, tp_rank)]
    parts = partition_chunks(L, pp_world, chunks)[pp_rank] if chunks else [partition_uniform(L, pp_world)[pp_rank]]
    if layer_counts is not None and [hi - lo for lo, hi in parts] != list(layer_counts):
        raise ValueError(f"pipeline stage {pp_rank} of the checkpoint holds {list(layer_counts)} layers per chunk, this model's partition gives {[hi - lo for lo, hi in parts]}")
    out = []
    for c, (lo, hi) in enumerate(parts):
        for n in stage_order(model_cfg, hi - lo, lo == 0, hi == L, tp_rank):
            out.append((f"{c}.{n}", f"{c}.model.{n}", stage_to_global(n, lo)) if chunks else (n, "model." + n, stage_to_global(n, lo)))
    return out


def zero_flat_order(named_shapes):
    """ZeRO rank-0 order of a parameter group on one rank: stable sort by numel, descending (hybrid_zero_optim.py:254-284)."""
    def numel(shape):
        n = 1
        for d in shape:
            n *= d
        return n

    return sorted(named_shapes, key=lambda kv: numel(kv[1]), reverse=True)


def tp_split_dim(name):
    """The dimension along which the reference's Megatron tensor parallelism cuts a parameter (None = replicated): embedding over
    the hidden dim (embed_split_hidden), head and the column-parallel wqkv (wq / wk / wv) / w1 / w3 over output rows, the
    row-parallel wo / w2 over input columns; norms whole.  The layer rules are internevo_amd/tensorpar.py:shard's, pinned on a real
    2-rank mtp run (tests/golden/train_tp2_*.json) and on its checkpoint files (tests/golden/ckpt_ref_tp2/)."""
    if name in ("tok_embeddings.weight", "embedding.weight"):   # (InternLM2 / InternLM-1 names)
        return 1
    if name in ("output.weight", "head.weight") or name.endswith(("attention.wqkv.weight", "attention.wq.weight", "attention.wk.weight", "attention.wv.weight",
                                                                  "feed_forward.w1.weight", "feed_forward.w3.weight", "mlp.w1.weight", "mlp.w3.weight",
                                                                  "mixer.Wqkv.weight", "mixer.Wqkv.bias")):
        return 0
    if name.endswith(("attention.wo.weight", "feed_forward.w2.weight", "mixer.out_proj.weight", "mlp.w2.weight")):
        return 1
    if ".experts.wrapped_experts." in name:   # every expert of the MoE model is a FeedForward over the tensor group (gshard_layer.py:421-433; tests/golden/ckpt_ref_moe_tp2dp2/)
        return 1 if name.endswith(".w2.weight") else 0
    return None


def _by_heads(name):
    """The InternLM-1 block's packed Wqkv (weight and bias): rows "(three h d)" -- a tensor rank holds "(three h/tp d)" of ITS heads (multi_head_attention.py:
    ColumnParallelLinear + rearrange with the local head count), so the ranks' parts interleave along the head axis instead of concatenating."""
    return name.endswith(("mixer.Wqkv.weight", "mixer.Wqkv.bias"))


def tp_shard(name, full, tp_rank, tp_world, head_dim=None):
    d = tp_split_dim(name)
    if tp_world == 1 or d is None:
        return full
    if _by_heads(name):
        v = full.reshape(3, -1, head_dim, *full.shape[1:])
        n = v.shape[1] // tp_world
        return v.narrow(1, tp_rank * n, n).reshape(-1, *full.shape[1:])
    n = full.shape[d] // tp_world
    return full.narrow(d, tp_rank * n, n)


def tp_unshard(name, parts, head_dim=None):
    d = tp_split_dim(name)
    if len(parts) == 1 or d is None:
        return parts[0]   # (replicated -- or, out_proj's bias of the InternLM-1 block, held by tensor rank 0 alone)
    if _by_heads(name):
        return torch.cat([p.reshape(3, -1, head_dim, *p.shape[1:]) for p in parts], dim=1).reshape(-1, *parts[0].shape[1:])
    return torch.cat(list(parts), dim=d)


def zero_partition(ordered, zero_world):
    """ordered: zero_flat_order(...) output.  -> per rank the list of indices into `ordered` (greedy: next-largest parameter to
    the rank with the fewest elements, first such rank on ties; hybrid_zero_optim.py:254-284)."""
    per_rank = [[] for _ in range(zero_world)]
    numel = [0] * zero_world
    for i, (_, shape) in enumerate(ordered):
        k = 1
        for d in shape:
            k *= d
        r = numel.index(min(numel))
        per_rank[r].append(i)
        numel[r] += k
    return per_rank


def _plan_ids(ordered, indices):
    return [f"{i}_" + "_".join(str(d) for d in ordered[i][1]) for i in indices]


def zero_rank_names(shapes, zero_world):
    """Reference parameter names owned by each ZeRO rank, in flat order.  shapes: dict name -> shape in module order."""
    ordered = zero_flat_order([(n, tuple(shp)) for n, shp in shapes.items()])
    return [[ordered[i][0] for i in idx] for idx in zero_partition(ordered, zero_world)]


class _RefEnumModule:
    """Context: make `ParallelMode.ZERO1` picklable by reference under the reference's module path without the reference."""

    def __enter__(self):
        self.created = []
        try:
            import importlib

            self.pm = importlib.import_module(_PM_MODULE).ParallelMode
            return self.pm.ZERO1
        except Exception:  # noqa: BLE001 - the reference is simply not installed
            pass
        parts = _PM_MODULE.split(".")
        for i in range(1, len(parts) + 1):
            name = ".".join(parts[:i])
            if name not in sys.modules:
                sys.modules[name] = types.ModuleType(name)
                self.created.append(name)
        mod = sys.modules[_PM_MODULE]
        pm = enum.Enum("ParallelMode", {"ZERO1": "zero1", "EXPERT_DATA": "expert_data", "DATA": "data"}, module=_PM_MODULE)
        mod.ParallelMode = pm
        self.pm = pm
        return pm.ZERO1

    def __exit__(self, *exc):
        for name in reversed(self.created):
            sys.modules.pop(name, None)


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("internlm"):
            try:
                return super().find_class(module, name)
            except Exception:  # noqa: BLE001 - reference not installed: a placeholder that remembers its constructor args
                return type(name, (), {"__module__": module, "__init__": lambda self, *a, **k: setattr(self, "args", a),
                                       "__repr__": lambda self: f"{name}{getattr(self, 'args', ())}"})
        return super().find_class(module, name)


_pickle_shim = types.ModuleType("internevo_amd_refpickle")
_pickle_shim.Unpickler = _TolerantUnpickler
_pickle_shim.load = lambda f, **kw: _TolerantUnpickler(f, **kw).load()
_pickle_shim.__name__ = "pickle"


def _load(path):
    return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_pickle_shim)


def save_checkpoint(folder, model_cfg, params, master, exp_avg, exp_avg_sq, adam_step, scaler, lr, hyper, param_dtype=torch.bfloat16,
                    zero_world=1, zero_ranks=None, write_model=True, shapes=None, tp_world=1, tp_rank=0, pp_world=1, pp_rank=0, order=None, chunked=False,
                    job_world=None, dp_ranks=None, plans_only=False):
    """params / master / exp_avg / exp_avg_sq: dict name -> host tensor (master and moments fp32).  scaler: dict(scale, growth_step,
    hysteresis_step).  hyper: dict(weight_decay, betas, eps, initial_lr).  lr: the learning rate in effect (param_groups' "lr").
    zero_world > 1: one optimizer + plan file per rank in `zero_ranks` (default: all); the state dicts then only need the
    parameters those ranks own (zero_rank_names).  write_model=False skips the model / topology files (the reference writes them
    from data-parallel rank 0 only) and `params` may then be None if `shapes` (name -> shape, module order) is given.
    tp_world > 1: the tensors are tensor-parallel rank `tp_rank`'s LOCAL parts (tp_shard); the files carry that rank in their names
    and the ZeRO partition is computed from the local shapes, as every tensor rank of the reference does for itself.
    pp_world > 1: ONE pipeline stage's files (`..._pp{pp_rank}...`); `order` = the stage's local names (stage_order / stage_naming) and every dict is keyed by
    them; chunked: the stage holds the interleaved schedule's model chunks (names "<chunk>.<name>").
    Hybrid ZeRO (parallel.zero1.size = zero_world below the data-parallel size; pinned on tests/golden/ckpt_ref_dp4_zo2/): the optimizer files are those of ONE zero
    group, but every data-parallel rank d writes a plan file, named after the JOB's world size and its own ranks: `gpus-{job_world}_..._dp-{d}_..._zo-{d % zero_world}.pt`
    (hybrid_zero_optim.py:133-140).  job_world: the number of ranks in the job (default zero_world * tp_world * pp_world: no hybrid ZeRO); dp_ranks: the data-parallel
    ranks whose plan files this call writes (default: the ranks of `zero_ranks`, as data ranks of the first zero group); plans_only: write nothing but those plan files
    (the ranks outside the first zero group: the state dicts are not needed then)."""
    os.makedirs(folder, exist_ok=True)
    order = state_dict_order(model_cfg, tp_rank) if order is None else list(order)
    if write_model and not plans_only:
        import re

        # (a stage of the interleaved schedule: `order` carries "<chunk>.<name>", the ModuleList's state dict "<chunk>.model.<name>")
        key = (lambda n: re.sub(r"^(\d+)\.", r"\1.model.", n)) if chunked else (lambda n: "model." + n)  # noqa: E731
        sd = collections.OrderedDict((key(n), params[n].detach().to("cpu", param_dtype).contiguous()) for n in order)
        torch.save(sd, os.path.join(folder, f"model_tp{tp_rank}_pp{pp_rank}.pt"))
        torch.save({}, os.path.join(folder, f"topo_tp{tp_rank}_pp{pp_rank}.json"))
    if shapes is None:
        shapes = {n: tuple(params[n].shape) for n in order}
    flat_order = zero_flat_order([(n, tuple(shapes[n])) for n in order])
    partition = zero_partition(flat_order, zero_world)
    plan = [[_plan_ids(flat_order, idx) for idx in partition], [[] for _ in range(zero_world)]]
    gpus = zero_world * tp_world * pp_world if job_world is None else int(job_world)
    plan_file = lambda d: os.path.join(folder, f"gpus-{gpus}_wp-0_tp-{tp_rank}_dp-{d}_pp-{pp_rank}_zo-{d % zero_world}.pt")  # noqa: E731
    if plans_only:
        for d in dp_ranks:
            torch.save(plan, plan_file(d))
        return

    with _RefEnumModule() as zero1:
        tail = dict(lr=lr, betas=tuple(hyper["betas"]), eps=hyper["eps"], amsgrad=False, maximize=False, foreach=None, capturable=False,
                    differentiable=False, fused=True, decoupled_weight_decay=True)
        for r in (range(zero_world) if zero_ranks is None else zero_ranks):
            names = [flat_order[i][0] for i in partition[r]]

            def flat(named):
                return torch.cat([named[n].detach().to("cpu", torch.float32).reshape(-1) for n in names])

            # key order as the reference's groups carry it (train/utils.py:create_param_groups + torch.optim.AdamW defaults)
            # (a rank with an EMPTY partition -- more ZeRO ranks than parameters -- keeps the group's own parameters in the base optimizer, without state and without
            #  flat weights: hybrid_zero_optim.py:210-222,880-895; the rule is pinned on the MoE model's gate group, tests/golden/ckpt_ref_moe_dp4/)
            g_default = dict(name="default", weight_decay=hyper["weight_decay"], optimizer_mode=zero1, **tail, dtype=param_dtype,
                             initial_lr=hyper["initial_lr"], params=[0] if names else list(range(len(flat_order))))
            g_fp32 = dict(name="fp32", optimizer_mode=zero1, weight_decay=hyper["weight_decay"], **tail, dtype=None,
                          initial_lr=hyper["initial_lr"], params=[])
            states = {
                "grad_scaler": {"_scale": float(scaler["scale"]), "_growth_step": int(scaler["growth_step"]),
                                "_hysteresis_step": int(scaler["hysteresis_step"])},
                "base_optim_states": {
                    "state": {0: {"step": torch.tensor(float(adam_step), dtype=torch.float32), "exp_avg": flat(exp_avg), "exp_avg_sq": flat(exp_avg_sq)}} if names else {},
                    "param_groups": [g_default, g_fp32],
                },
                "flat_fp32_weights": {0: flat(master)} if names else {},
                "zero_devide_optim_plan": plan,  # the reference writes the state file BEFORE popping the plan (components.py:398-407)
            }
            torch.save(states, os.path.join(folder, f"optimizer_tp{tp_rank}_pp{pp_rank}_zo{r}.pt"))
            if dp_ranks is None:
                torch.save(plan, plan_file(r))
        for d in (dp_ranks or ()):
            torch.save(plan, plan_file(d))


def remove_stale_shards(folder, zero_world, tp_world, pp_world=1, job_world=None, layout="plain", wp_world=1, dp_world=None, num_experts=0, num_layers=0):
    """Before a save into an existing folder: drop the files of an earlier save that the loaders would merge into this one.  The loaders infer the
    saved layout from the files present (saved_zero_world / saved_tp_world as the reference's components.py:294-306 does; saved_isp_layout: ANY
    `model_tp{t}_wp{w}_pp0.pt` makes the folder an ISP-layout folder), so left-over `optimizer_tp0_pp0_zo3.pt` files next to a fresh 2-rank save would
    be merged as if they belonged to it, and left-over ISP files next to a plain save would be loaded INSTEAD of it.  Removed:
      * files of this save's own layout with an out-of-range index (a LARGER earlier layout);
      * every file of the OTHER layouts: layout = "plain" (model_tp{t}_pp{p}.pt, optimizer_tp{t}_pp{p}_zo{z}.pt, topo_*.json, plans with wp-0),
        "isp" (model_tp{t}_wp{w}_pp0.pt, optimizer_tp{t}_wp{w}_pp0_dp{d}.pt, plans gpus-{W}_wp-{w}_...; tp_world = the sequence-parallel size, dp_world the
        data-parallel ranks per (t, w)), "moe" (= plain + model_moe_layer{l}_expert{e}_tp0.pt: experts >= num_experts / layers >= num_layers go).
    Called by ONE rank, before anyone writes.  Returns the names of the files it removed (the engine logs them: this DELETES files in the user's
    checkpoint folder -- only names this module itself writes)."""
    removed = []
    if not os.path.isdir(folder):
        return removed
    import re

    isp = layout == "isp"
    job = (zero_world * tp_world * pp_world) if job_world is None else job_world
    for fn in sorted(os.listdir(folder)):
        stale = False
        m = re.fullmatch(r"optimizer_tp(\d+)_pp(\d+)_zo(\d+)\.pt", fn)
        if m:
            stale = isp or int(m.group(1)) >= tp_world or int(m.group(2)) >= pp_world or int(m.group(3)) >= zero_world
        m = re.fullmatch(r"(?:model_tp(\d+)_pp(\d+)\.pt|topo_tp(\d+)_pp(\d+)\.json)", fn)
        if m:
            stale = isp or int(m.group(1) or m.group(3)) >= tp_world or int(m.group(2) or m.group(4)) >= pp_world
        m = re.fullmatch(r"model_tp(\d+)_wp(\d+)_pp(\d+)\.pt", fn)
        if m:
            stale = not isp or int(m.group(1)) >= tp_world or int(m.group(2)) >= wp_world or int(m.group(3)) >= pp_world
        m = re.fullmatch(r"optimizer_tp(\d+)_wp(\d+)_pp(\d+)_dp(\d+)\.pt", fn)
        if m:
            stale = (not isp or int(m.group(1)) >= tp_world or int(m.group(2)) >= wp_world or int(m.group(3)) >= pp_world
                     or (dp_world is not None and int(m.group(4)) >= dp_world))
        m = re.fullmatch(r"gpus-(\d+)_wp-(\d+)_tp-(\d+)_dp-(\d+)_pp-(\d+)_zo-(\d+)\.pt", fn)
        if m:
            g, w, t, d, p_, z = (int(x) for x in m.groups())
            if isp:
                stale = g != job or w >= wp_world or t >= tp_world or p_ >= pp_world or (dp_world is not None and d >= dp_world)
            else:
                stale = w > 0 or g != job or t >= tp_world or p_ >= pp_world or z >= zero_world
        m = re.fullmatch(r"model_moe_layer(\d+)_expert(\d+)_tp(\d+)\.pt", fn)
        if m:
            stale = layout != "moe" or int(m.group(1)) >= num_layers or int(m.group(2)) >= num_experts or int(m.group(3)) >= tp_world
        if stale:
            os.remove(os.path.join(folder, fn))
            removed.append(fn)
    return removed


def saved_zero_world(folder, tp_rank=0, pp_rank=0):
    """Number of ZeRO-1 optimizer shards of a tensor rank in the folder (components.py:294-306 counts them the same way); 0 = weights only."""
    n, pre = 0, f"optimizer_tp{tp_rank}_pp{pp_rank}_zo"
    for fn in os.listdir(folder):
        if fn.startswith(pre) and fn.endswith(".pt"):
            n = max(n, int(fn[len(pre):-3]) + 1)
    return n


def saved_tp_world(folder):
    import re

    n = 0
    for fn in os.listdir(folder):
        m = re.fullmatch(r"model_tp(\d+)_pp0\.pt", fn)
        if m:
            n = max(n, int(m.group(1)) + 1)
    return n


# ---- model files of the ISP layout (tensor mode "isp" + weight parallelism; configs/7B_isp_sft.py) ---------------------------------------------------
# save_model_checkpoint under ISP (checkpoint/components.py:221-226) writes `model_tp{t}_wp{w}_pp{p}.pt` = the state dict of a rank with tensor rank t and
# weight rank w: the embedding cut over hidden COLUMNS and the head over vocabulary ROWS of the TENSOR group (modules/embedding.py:40-50, ops/linear.py:79-153),
# every ISPLinear weight and bias cut over output ROWS of the WEIGHT group (ops/linear.py:357-378), norm weights whole.  Only the (t, w) pairs that occur on
# ranks with weight-data rank 0 or data rank 0 are written (tensor 2 x weight 4: tp0_wp0, tp1_wp1, tp0_wp2, tp1_wp3).  Pinned on a real two-process
# checkpoint (tests/golden/ckpt_ref_isp2v1/, make_golden.py --ckpt-isp).
def isp_split(name):
    """('tp' | 'wp' | None, dim) -- over which group and along which dimension the ISP layout cuts a parameter (both block families' names)."""
    if name in ("tok_embeddings.weight", "embedding.weight"):
        return "tp", 1
    if name in ("output.weight", "head.weight"):
        return "tp", 0
    if name.endswith(("norm.weight", "norm1.weight", "norm2.weight")):
        return None, None
    return "wp", 0


def isp_shard(name, full, tp_rank, tp_world, wp_rank, wp_world):
    grp, d = isp_split(name)
    if grp is None:
        return full
    r, w = (tp_rank, tp_world) if grp == "tp" else (wp_rank, wp_world)
    n = full.shape[d] // w
    return full.narrow(d, r * n, n)


def saved_isp_layout(folder):
    """-> {(t, w): file name} of the `model_tp{t}_wp{w}_pp0.pt` files of a folder (empty: not an ISP-layout folder)."""
    import re

    out = {}
    for fn in os.listdir(folder):
        m = re.fullmatch(r"model_tp(\d+)_wp(\d+)_pp(\d+)\.pt", fn)
        if m:
            if int(m.group(3)) != 0:
                raise NotImplementedError("ISP-layout checkpoints with pipeline parallelism")
            out[(int(m.group(1)), int(m.group(2)))] = fn
    return out


def load_isp_model(folder, model_cfg):
    """The FULL parameters (reference names) out of an ISP-layout folder: tensor-group cuts taken from one file per tensor rank, weight-group cuts from one
    file per weight rank (every rank of either group occurs in at least one file)."""
    files = saved_isp_layout(folder)
    tp_world, wp_world = max(t for t, _ in files) + 1, max(w for _, w in files) + 1
    by_tp = {t: fn for (t, _), fn in sorted(files.items(), reverse=True)}
    by_wp = {w: fn for (_, w), fn in sorted(files.items(), reverse=True)}
    if sorted(by_tp) != list(range(tp_world)) or sorted(by_wp) != list(range(wp_world)):
        raise FileNotFoundError(f"{folder}: the ISP-layout model files {sorted(files.values())} do not cover tensor ranks 0..{tp_world - 1} and weight ranks 0..{wp_world - 1}")
    cache = {}

    def sd(fn):
        if fn not in cache:
            cache[fn] = torch.load(os.path.join(folder, fn), map_location="cpu", weights_only=False)
        return cache[fn]

    def get(fn, n):
        d = sd(fn)
        return d["model." + n if "model." + n in d else n].detach()

    out = {}
    for n in state_dict_order(model_cfg):
        grp, dim = isp_split(n)
        if grp is None:
            out[n] = get(next(iter(files.values())), n)
        elif grp == "tp":
            out[n] = torch.cat([get(by_tp[t], n) for t in range(tp_world)], dim=dim)
        else:
            out[n] = torch.cat([get(by_wp[w], n) for w in range(wp_world)], dim=dim)
    return out, tp_world, wp_world


def isp_coords(rank, world, sp, wp):
    """The group ranks of global rank `rank` in an ISP job of `world` ranks, tensor (= sequence) size sp, weight size wp, zero1.size -1
    (process_group_initializer.py: tensor and weight groups are blocks of consecutive ranks; pinned by tests/golden/ckpt_isp4v1_rank*.json):
    dict(t, w, d, z, data_world, zero_world) -- tensor rank, weight rank, data rank (= rank // sp), ZERO1 rank (= weight-data rank = rank // wp)."""
    if world % sp or world % wp:
        raise ValueError(f"an ISP job of {world} ranks with tensor size {sp} and weight size {wp}")
    return dict(t=rank % sp, w=rank % wp, d=rank // sp, z=rank // wp, data_world=world // sp, zero_world=world // wp)


def isp_groups(model_cfg):
    """[(group name, parameter names in module order)] of an ISP run's optimizer groups (train/utils.py:11-80): "default" = the layers' ISPLinear weights and
    biases + every norm weight, ZeRO over the weight-data group; "embed_head" = embedding + head, ZeRO over the DATA group; "fp32" (empty in a bf16 run)."""
    order = state_dict_order(model_cfg)
    eh = [n for n in order if isp_split(n)[0] == "tp"]
    return [("default", [n for n in order if n not in eh]), ("embed_head", eh), ("fp32", [])]


def _isp_group_layout(model_cfg, full_shapes, c, sp, wp):
    """Per group: (flat order of (name, LOCAL shape) on a rank with coordinates c, partition over the group's zero world)."""
    out = []
    for g, (_, names) in enumerate(isp_groups(model_cfg)):
        local = [(n, tuple(isp_shard(n, torch.empty(full_shapes[n], device="meta"), c["t"], sp, c["w"], wp).shape)) for n in names]
        fo = zero_flat_order(local)
        out.append((fo, zero_partition(fo, c["data_world"] if g == 1 else c["zero_world"])))
    return out


def save_isp_optimizer_shard(folder, model_cfg, rank, world, sp, wp, master, exp_avg, exp_avg_sq, adam_step, scaler, lr, hyper, param_dtype=torch.bfloat16):
    """What global rank `rank` of an ISP job writes (checkpoint/components.py:377-410): `optimizer_tp{t}_wp{w}_pp0_dp{d}.pt` -- three groups, the flat fp32
    master weights / AdamW moments of the rank's partitions of ITS local shards (group "default": partition z of the weight-data group; "embed_head":
    partition d of the data group), the reference's greedy whole-parameter partition of the LOCAL shapes -- and the plan file.  master / exp_avg /
    exp_avg_sq: name -> FULL fp32 host tensor, at least for the parameters this rank's partitions hold (isp_rank_names)."""
    os.makedirs(folder, exist_ok=True)
    c = isp_coords(rank, world, sp, wp)
    full_shapes = {n: tuple(t.shape) for n, t in master.items()} if len(master) == len(state_dict_order(model_cfg)) else None
    if full_shapes is None:
        raise ValueError("save_isp_optimizer_shard takes the shapes of ALL parameters from `master` (entries a rank does not own may be meta / empty tensors)")
    layout = _isp_group_layout(model_cfg, full_shapes, c, sp, wp)
    plan = [[_plan_ids(fo, idx) for idx in part] for fo, part in layout]

    def flat(named, g):
        fo, part = layout[g]
        mine = part[c["d"] if g == 1 else c["z"]]
        return torch.cat([isp_shard(fo[i][0], named[fo[i][0]].detach().to("cpu", torch.float32), c["t"], sp, c["w"], wp).reshape(-1) for i in mine])

    with _RefEnumModule() as zero1:
        pm = type(zero1)
        tail = dict(lr=lr, betas=tuple(hyper["betas"]), eps=hyper["eps"], amsgrad=False, maximize=False, foreach=None, capturable=False,
                    differentiable=False, fused=True, decoupled_weight_decay=True)
        wd, ilr = hyper["weight_decay"], hyper["initial_lr"]
        # (a rank whose partition of a group is EMPTY -- embedding + head over three or more data ranks -- lists the group's own parameters, without state and without
        #  flat weights, and the ids behind them move up: see save_moe_checkpoint; pinned on data rank 2 of tests/golden/ckpt_ref_isp6v1/)
        held = [bool(layout[g][1][c["d"] if g == 1 else c["z"]]) for g in (0, 1)]
        ids, nxt = [], 0
        for g in (0, 1):
            k = 1 if held[g] else len(layout[g][0])
            ids.append(list(range(nxt, nxt + k)))
            nxt += k
        pgs = [dict(name="default", weight_decay=wd, optimizer_mode=zero1, **tail, dtype=param_dtype, initial_lr=ilr, params=ids[0]),
               dict(name="embed_head", optimizer_mode=pm.DATA, weight_decay=wd, **tail, dtype=param_dtype, initial_lr=ilr, params=ids[1]),
               dict(name="fp32", optimizer_mode=zero1, weight_decay=wd, **tail, dtype=None, initial_lr=ilr, params=[])]
        states = {
            "grad_scaler": {"_scale": float(scaler["scale"]), "_growth_step": int(scaler["growth_step"]), "_hysteresis_step": int(scaler["hysteresis_step"])},
            "base_optim_states": {"state": {ids[g][0]: {"step": torch.tensor(float(adam_step), dtype=torch.float32), "exp_avg": flat(exp_avg, g), "exp_avg_sq": flat(exp_avg_sq, g)}
                                            for g in (0, 1) if held[g]},
                                  "param_groups": pgs},
            "flat_fp32_weights": {g: flat(master, g) for g in (0, 1) if held[g]},
            "zero_devide_optim_plan": plan,
        }
        torch.save(states, os.path.join(folder, f"optimizer_tp{c['t']}_wp{c['w']}_pp0_dp{c['d']}.pt"))
        torch.save(plan, os.path.join(folder, f"gpus-{world}_wp-{c['w']}_tp-{c['t']}_dp-{c['d']}_pp-0_zo-{c['z']}.pt"))


def isp_rank_names(model_cfg, full_shapes, rank, world, sp, wp):
    """The parameters (reference names) of which global rank `rank` of an ISP job holds optimizer state -- whole LOCAL shards, both groups."""
    c = isp_coords(rank, world, sp, wp)
    layout = _isp_group_layout(model_cfg, full_shapes, c, sp, wp)
    return [layout[g][0][i][0] for g in (0, 1) for i in layout[g][1][c["d"] if g == 1 else c["z"]]]


def load_isp_optimizer(folder, model_cfg, params, want=None):
    """The FULL fp32 master weights and AdamW moments out of the optimizer shards of an ISP-layout folder (`optimizer_tp{t}_wp{w}_pp0_dp{d}.pt` of every
    rank + the plan files, whose names carry the rank's zero1 position): every local shard is put back at its place in the full tensor.  params: the merged
    model (load_isp_model) for the shapes.  want (a set of reference names, or None = all): only these are allocated and copied -- a rank of a large job asks
    for what it holds, not 3 x fp32 of the whole model -- while coverage (every element of every parameter present in some shard) is still checked for ALL
    names, with one element counter per name.  -> dict(master, exp_avg, exp_avg_sq, adam_step, scaler, lr)."""
    import re

    files = {}
    for fn in os.listdir(folder):
        m = re.fullmatch(r"gpus-(\d+)_wp-(\d+)_tp-(\d+)_dp-(\d+)_pp-0_zo-(\d+)\.pt", fn)
        if m:
            world, w, t, d, z = (int(x) for x in m.groups())
            files[(t, w, d)] = (world, z)
    if not files:
        raise FileNotFoundError(f"{folder}: no partition-plan files (gpus-*_wp-*_tp-*_dp-*_pp-0_zo-*.pt) next to the ISP optimizer shards")
    world = next(iter(files.values()))[0]
    sp, wp = max(t for t, _, _ in files) + 1, max(w for _, w, _ in files) + 1
    full_shapes = {n: tuple(t.shape) for n, t in params.items()}
    keep = set(full_shapes) if want is None else set(want) & set(full_shapes)
    out = {k: {n: torch.zeros(full_shapes[n], dtype=torch.float32) for n in full_shapes if n in keep} for k in ("master", "exp_avg", "exp_avg_sq")}
    # coverage without a mask per element: the region of a parameter a (t, w) rank holds is a view (isp_shard) -- identified by its offset / shape / strides on a
    # storage-less tensor; the regions of different ranks are identical (replicated parameters: norms) or disjoint (cut ones)
    seen = {n: {} for n in full_shapes}
    probe = {n: torch.empty(full_shapes[n], device="meta") for n in full_shapes}
    meta = step = None
    for (t, w, d), (wld, z) in sorted(files.items()):
        st = _load(os.path.join(folder, f"optimizer_tp{t}_wp{w}_pp0_dp{d}.pt"))
        c = dict(t=t, w=w, d=d, z=z, data_world=wld // sp, zero_world=wld // wp)
        layout = _isp_group_layout(model_cfg, full_shapes, c, sp, wp)
        for g in (0, 1):
            fo, part = layout[g]
            mine = part[d if g == 1 else z]
            if list(st["zero_devide_optim_plan"][g][d if g == 1 else z]) != _plan_ids(fo, mine):
                raise ValueError(f"optimizer_tp{t}_wp{w}_pp0_dp{d}.pt: the partition plan of group {g} does not match this model / layout")
            if not mine:   # this rank holds no parameter of the group: no state, no flat weights (see save_isp_optimizer_shard)
                continue
            sid = st["base_optim_states"]["param_groups"][g]["params"][0]
            for key, vec in (("master", st["flat_fp32_weights"][g]), ("exp_avg", st["base_optim_states"]["state"][sid]["exp_avg"]),
                             ("exp_avg_sq", st["base_optim_states"]["state"][sid]["exp_avg_sq"])):
                o = 0
                for i in mine:
                    n, shape = fo[i]
                    k = 1
                    for dd in shape:
                        k *= dd
                    if n in keep:
                        isp_shard(n, out[key][n], t, sp, w, wp).copy_(vec.detach()[o : o + k].reshape(shape))
                    if key == "master":
                        reg = isp_shard(n, probe[n], t, sp, w, wp)
                        seen[n][(reg.storage_offset(), tuple(reg.shape), tuple(reg.stride()))] = k
                    o += k
                if o != vec.numel():
                    raise ValueError(f"optimizer_tp{t}_wp{w}_pp0_dp{d}.pt group {g}: {vec.numel()} elements, the partition holds {o}")
        gs, base = st["grad_scaler"], st["base_optim_states"]
        some = next(iter(base["state"].values()), None)   # (a rank without any parameter carries no step: the scaler and lr only)
        if some is not None:
            if step not in (None, int(float(some["step"]))):
                raise ValueError("the ISP optimizer shards disagree on the step")
            step = int(float(some["step"]))
        here = (float(base["param_groups"][0]["lr"]), float(gs["_scale"]), int(gs["_growth_step"]), int(gs["_hysteresis_step"]))
        if meta is not None and here != meta:
            raise ValueError("the ISP optimizer shards disagree on lr / loss scale")
        meta = here
    missing = [n for n, regs in seen.items() if sum(regs.values()) != int(torch.Size(full_shapes[n]).numel())]
    if missing:
        raise FileNotFoundError(f"{folder}: the optimizer shards present do not cover {missing[:4]} ... (files of some ranks are missing)")
    return dict(out, adam_step=step, lr=meta[0], scaler=dict(scale=meta[1], growth_step=meta[2], hysteresis_step=meta[3]), zero_world=world // wp,
                isp=dict(world=world, sp=sp, wp=wp))


def save_isp_model_shard(folder, model_cfg, full_params, tp_rank, tp_world, wp_rank, wp_world, param_dtype=torch.bfloat16):
    """One rank's `model_tp{t}_wp{w}_pp0.pt` (+ the topology file) from the FULL parameters (reference names): what save_model_checkpoint writes on a
    rank with tensor rank t and weight rank w."""
    os.makedirs(folder, exist_ok=True)
    sd = collections.OrderedDict()
    for n in state_dict_order(model_cfg):
        sd["model." + n] = isp_shard(n, full_params[n].detach().to("cpu"), tp_rank, tp_world, wp_rank, wp_world).to(param_dtype).contiguous().clone()
    torch.save(sd, os.path.join(folder, f"model_tp{tp_rank}_wp{wp_rank}_pp0.pt"))
    torch.save({}, os.path.join(folder, f"topo_tp{tp_rank}_wp{wp_rank}_pp0.json"))


def saved_pp_world(folder):
    n = 0
    import re

    for fn in os.listdir(folder):
        m = re.fullmatch(r"model_tp0_pp(\d+)\.pt", fn)
        if m:
            n = max(n, int(m.group(1)) + 1)
    return n


def _stage_layers(sd):
    """Number of layers in a stage's state dict (its layers are numbered from 0)."""
    import re

    idx = [int(m.group(1)) for m in (re.match(r"(?:model\.)?(?:layers|blocks)\.(\d+)\.", k) for k in sd) if m]
    return max(idx) + 1 if idx else 0


def _load_tp_rank(folder, model_cfg, t, tp_world, want, pp_rank=0, pp_world=1, sd=None, model_only=False, naming=None):
    """One tensor rank's files (of one pipeline stage) -> its LOCAL named tensors (all its ZeRO shards merged).  want: the names (as this stage's files
    carry them) whose optimizer tensors are kept; sd: the stage's model state dict if the caller has read it already; naming: stage_naming(...) of the stage."""
    if sd is None:
        sd = torch.load(os.path.join(folder, f"model_tp{t}_pp{pp_rank}.pt"), map_location="cpu", weights_only=False)
    if naming is None:
        order = state_dict_order(model_cfg, t) if pp_world == 1 else stage_order(model_cfg, _stage_layers(sd), pp_rank == 0, pp_rank == pp_world - 1, t)
        naming = [(n, "model." + n, n) for n in order]
    order = [n for n, _, _ in naming]
    params = {}
    for n, key, _ in naming:
        key = key if key in sd else n
        if key not in sd:
            raise KeyError(f"checkpoint has no parameter {n!r} (keys: {list(sd)[:4]} ...)")
        params[n] = sd[key].detach()
    out = dict(params=params, master=None, exp_avg=None, exp_avg_sq=None, adam_step=None, scaler=None, lr=None, zero_world=0)
    zero_world = 0 if model_only else saved_zero_world(folder, t, pp_rank)
    if zero_world == 0:
        return out
    flat_order = zero_flat_order([(n, tuple(params[n].shape)) for n in order])
    partition = zero_partition(flat_order, zero_world)
    plan_ids = [_plan_ids(flat_order, idx) for idx in partition]
    merged = dict(master={}, exp_avg={}, exp_avg_sq={})
    head = None
    for r in range(zero_world):
        opt_path = os.path.join(folder, f"optimizer_tp{t}_pp{pp_rank}_zo{r}.pt")
        if not os.path.exists(opt_path):
            raise FileNotFoundError(f"{opt_path}: the folder holds shards up to zo{zero_world - 1} but not this one")
        st = _load(opt_path)
        plan_path = os.path.join(folder, f"gpus-{zero_world * tp_world * pp_world}_wp-0_tp-{t}_dp-{r}_pp-{pp_rank}_zo-{r}.pt")
        plan = _load(plan_path) if os.path.exists(plan_path) else st.get("zero_devide_optim_plan")
        if plan is not None and list(plan[0]) != plan_ids:
            raise ValueError(f"zero_devide_optim_plan of the checkpoint does not match this model's partition over {zero_world} ranks")

        def unflat(vec, into):
            o = 0
            for i in partition[r]:
                n, shape = flat_order[i]
                k = 1
                for d in shape:
                    k *= d
                if want is None or n in want:
                    into[n] = vec.detach()[o : o + k].reshape(shape).to(torch.float32)  # the reference saves nn.Parameters (requires_grad)
                o += k
            if o != vec.numel():
                raise ValueError(f"flat optimizer vector of tp{t} zo{r} holds {vec.numel()} elements, this model's partition {o}")

        base = st["base_optim_states"]
        if not partition[r]:   # a ZeRO rank without parameters: its file carries the scaler and the groups only
            if base["state"] or st["flat_fp32_weights"]:
                raise ValueError(f"optimizer shard tp{t} zo{r} holds state, this model's partition over {zero_world} ranks gives that rank no parameter")
            continue
        s0 = base["state"][0]
        unflat(st["flat_fp32_weights"][0], merged["master"])
        unflat(s0["exp_avg"], merged["exp_avg"])
        unflat(s0["exp_avg_sq"], merged["exp_avg_sq"])
        gs = st["grad_scaler"]
        this = dict(adam_step=int(float(s0["step"])), lr=float(base["param_groups"][0]["lr"]),
                    scaler=dict(scale=float(gs["_scale"]), growth_step=int(gs["_growth_step"]), hysteresis_step=int(gs["_hysteresis_step"])))
        if head is None:
            head = this
        elif this != head:
            raise ValueError(f"optimizer shard tp{t} zo{r} disagrees with zo0 on the step / scaler / lr: {this} vs {head}")
        del st
    out.update(merged, zero_world=zero_world, **head)
    return out


def load_checkpoint(folder, model_cfg, want=None, model_only=False):
    """-> dict(params, master, exp_avg, exp_avg_sq (name -> host tensor, FULL tensors), adam_step, scaler, lr, zero_world, tp_world).
    Optimizer entries are None when the folder holds model weights only, or with model_only (load_ckpt_info content = ("model",): the optimizer files are not read).  Every shard in the folder is read and merged -- whatever
    ZeRO world and tensor-parallel size wrote them; `want` (a set of names) limits the optimizer tensors kept in memory."""
    if saved_isp_layout(folder):   # the ISP layout: model_tp{t}_wp{w}_pp0.pt + optimizer_tp{t}_wp{w}_pp0_dp{d}.pt, merged into FULL tensors
        params, tp_world, _ = load_isp_model(folder, model_cfg)
        out = dict(params=params, master=None, exp_avg=None, exp_avg_sq=None, adam_step=None, scaler=None, lr=None, zero_world=0, tp_world=tp_world)
        if not model_only and any(fn.startswith("optimizer_tp") for fn in os.listdir(folder)):
            out.update(load_isp_optimizer(folder, model_cfg, params, want=want))
        return out
    tp_world = saved_tp_world(folder)
    if tp_world == 0:
        raise FileNotFoundError(f"{folder}: no model_tp*_pp0.pt")
    pp_world = saved_pp_world(folder)
    if pp_world > 1:
        # one set of files per pipeline stage (and per tensor rank of the stage), each numbering its layers from 0: the tensor ranks of a stage are merged
        # into full tensors, the stages into the whole model under its own names (so that ANY layout -- another pipeline or tensor size, or none -- can
        # resume from the folder)
        out, seen = None, 0
        hd = model_cfg.head_dim
        same = ("adam_step", "lr", "scaler", "zero_world")
        chunks = None
        for p_ in range(pp_world):
            sds = [torch.load(os.path.join(folder, f"model_tp{t}_pp{p_}.pt"), map_location="cpu", weights_only=False) for t in range(tp_world)]
            ch = stage_chunks(sds[0])
            if chunks is not None and ch != chunks:
                raise ValueError(f"pipeline stage {p_} holds {ch} model chunks, stage 0 holds {chunks}")
            chunks = ch
            if ch:   # (the interleaved schedule: the layers of every chunk, from the "<c>.model.layers.<k>." keys)
                counts = [_stage_layers({k[len(f"{c}."):]: 0 for k in sds[0] if k.startswith(f"{c}.model.")}) for c in range(ch)]
            else:
                counts = [_stage_layers(sds[0])]
            namings = [stage_naming(model_cfg, pp_world, p_, ch, t, counts) for t in range(tp_world)]
            glob = {n: g for n, _, g in namings[0]}   # (tensor rank 0 holds every name of the stage)
            # (the optimizer tensors a caller does not want -- another stage's, another ZeRO rank's -- are dropped while the flat vectors are cut, not after)
            local_want = None if want is None else {n for n, g in glob.items() if g in want}
            ranks = [_load_tp_rank(folder, model_cfg, t, tp_world, local_want, p_, pp_world, sd=sds[t], model_only=model_only, naming=namings[t]) for t in range(tp_world)]
            st = ranks[0]
            for t, r in enumerate(ranks[1:], 1):
                if {k: r[k] for k in same} != {k: st[k] for k in same}:
                    raise ValueError(f"tensor rank {t} of pipeline stage {p_} disagrees with rank 0 on the step / scaler / lr / ZeRO world")
            whole = lambda key: None if st[key] is None else {glob[n]: (tp_unshard(glob[n], [r[key][n] for r in ranks if n in r[key]], hd) if tp_world > 1 else st[key][n]) for n in st[key]}  # noqa: E731
            named = {k: whole(k) for k in ("params", "master", "exp_avg", "exp_avg_sq")}
            if out is None:
                out = dict(st, **named)
            else:
                if {k: st[k] for k in same} != {k: out[k] for k in same}:
                    raise ValueError(f"pipeline stage {p_} disagrees with stage 0 on the step / scaler / lr / ZeRO world")
                for k, d in named.items():
                    if d is not None:
                        out[k].update(d)
            seen += sum(counts)
        if seen != model_cfg.num_layers:
            raise ValueError(f"the {pp_world} pipeline stages of the checkpoint hold {seen} layers, the model {model_cfg.num_layers}")
        order = state_dict_order(model_cfg)   # (the model's own order: with interleaved chunks the stages do not hold consecutive layers)
        for k in ("params", "master", "exp_avg", "exp_avg_sq"):
            if out[k] is not None:
                out[k] = {n: out[k][n] for n in order if n in out[k]}
        return dict(out, tp_world=tp_world, pp_world=pp_world, chunks=chunks or 1)
    ranks = [_load_tp_rank(folder, model_cfg, t, tp_world, want, model_only=model_only) for t in range(tp_world)]
    out = dict(ranks[0], tp_world=tp_world)
    if tp_world == 1:
        return out
    for t, r in enumerate(ranks[1:], 1):
        same = ("adam_step", "lr", "scaler", "zero_world")
        if {k: r[k] for k in same} != {k: ranks[0][k] for k in same}:
            raise ValueError(f"tensor rank {t} disagrees with rank 0 on the step / scaler / lr / ZeRO world")
    for key in ("params", "master", "exp_avg", "exp_avg_sq"):
        if ranks[0][key] is not None:
            out[key] = {n: tp_unshard(n, [r[key][n] for r in ranks if n in r[key]], model_cfg.head_dim) for n in ranks[0][key]}
    return out


# ---- the MoE model (model_type INTERNLM_MoE): three optimizer groups, one model file per expert ---------------------------------------------------------
# What the reference writes for it (checkpoint/components.py:31-93 try_load / try_save_moe_checkpoint, train/utils.py:25-80 create_param_groups; pinned
# by tests/golden/ckpt_ref_moe/, make_golden.py --ckpt-moe):
#   model_tp0_pp0.pt                               the state dict WITHOUT the experts (the gates, fp32 modules, are in it)
#   model_moe_layer{l}_expert{e}_tp0.pt            one expert's w1 / w2 / w3 under its GLOBAL expert number
#   optimizer_tp0_pp0_zo0.pt                       base_optim_states.state {0, 1, 2}, param_groups [default, fp32 (the gates), moe_ep_size_{ep} (the experts,
#                                                  optimizer_mode EXPERT_DATA)], flat_fp32_weights {0, 1, 2}, the plan with one list per group
# Covered: any number of data-parallel ranks (the reference's automatic expert parallelism, ep = min(dp, experts); pinned on tests/golden/ckpt_ref_moe_dp2/ too).
def moe_groups(model_cfg, ep_world=1, ep_rank=0, tp_rank=0):
    """[(group name, parameter names in module order)] of the three optimizer groups ON ONE RANK of an expert-parallel group of ep_world ranks: the dense
    parameters and the gates whole, the experts this rank holds (global numbers ep_rank * E / ep ... -- the reference's automatic expert parallelism,
    parallel_context.py:538-541)."""
    order = state_dict_order(model_cfg, tp_rank)
    gates = [n for n in order if n.endswith("gate.wg.weight")]
    El = max(model_cfg.num_experts, 1) // ep_world
    mine = range(ep_rank * El, (ep_rank + 1) * El)
    experts = [n for n in order if ".experts." in n and int(n.split(".")[6]) in mine]
    dense = [n for n in order if ".experts." not in n and n not in set(gates)]
    return [("default", dense), ("fp32", gates), (f"moe_ep_size_{ep_world}", experts)]


def _expert_file(name, tp_rank=0):
    parts = name.split(".")   # blocks.{l}.mlp.moe_layer.experts.wrapped_experts.{e}.w1.weight
    return f"model_moe_layer{parts[1]}_expert{parts[6]}_tp{tp_rank}.pt"


def _moe_layout(model_cfg, shapes, world, rank, tp_rank=0):
    """Per optimizer group on data-parallel rank `rank` of `world`: (flat order of (name, shape), partition over the group's zero world, this rank's position in
    it).  default / fp32: ZeRO over the data-parallel group; the expert group: over the EXPERT_DATA group (the ranks holding the same experts: rank // ep)."""
    ep = min(world, max(model_cfg.num_experts, 1))
    if world % ep or max(model_cfg.num_experts, 1) % ep:
        raise NotImplementedError(f"{world} data-parallel ranks with {model_cfg.num_experts} experts")
    out = []
    for g, (_, names) in enumerate(moe_groups(model_cfg, ep, rank % ep, tp_rank)):
        fo = zero_flat_order([(n, tuple(shapes[n])) for n in names])
        zw, zr = (world // ep, rank // ep) if g == 2 else (world, rank)
        out.append((fo, zero_partition(fo, zw), zr))
    return out, ep


def save_moe_checkpoint(folder, model_cfg, params, master, exp_avg, exp_avg_sq, adam_step, scaler, lr, hyper, param_dtype=torch.bfloat16, world=1, rank=0,
                        tp_world=1, tp_rank=0):
    """params: name -> host tensor (gates fp32, the rest in the model dtype); master / exp_avg / exp_avg_sq: name -> fp32 host tensor; the rest as
    save_checkpoint.  world > 1 (round 4): what data-parallel rank `rank` of `world` writes in a reference job with its automatic expert parallelism
    (ep = min(world, experts); pinned by tests/golden/ckpt_ref_moe_dp2/): its optimizer shard -- the dense parameters' and the gates' partition `rank` of the
    data-parallel group, and its partition of ITS experts over the expert-data group --, its plan file, one model file per expert it holds (under the
    expert's GLOBAL number; expert-data rank 0 only), and from rank 0 the model file without the experts.  The dicts need this rank's experts only.
    tp_world > 1 (Megatron tensor parallelism; pinned by tests/golden/ckpt_ref_moe_tp2dp2/): `world` / `rank` are the DATA-parallel size and rank, the tensors tensor
    rank tp_rank's LOCAL parts as the reference's modules hold them (tp_shard: embedding columns, head rows, the heads' Wqkv rows, out_proj / w2 columns, w1 / w3
    rows -- the experts' too; out_proj's bias on tensor rank 0 only); every file carries the tensor rank in its name, the partitions are computed from the local shapes."""
    os.makedirs(folder, exist_ok=True)
    shapes = {n: tuple(params[n].shape) for n in state_dict_order(model_cfg, tp_rank) if n in params}
    layout, ep = _moe_layout(model_cfg, shapes, world, rank, tp_rank)
    groups = moe_groups(model_cfg, ep, rank % ep, tp_rank)
    if rank == 0:   # (the reference: tensor rank t's file from data rank t % dp, components.py:256-262 -- the same content on every data rank)
        sd = collections.OrderedDict()
        for n in state_dict_order(model_cfg, tp_rank):
            if ".experts." not in n:
                sd["model." + n] = params[n].detach().to("cpu", torch.float32 if n.endswith("gate.wg.weight") else param_dtype).contiguous()
        torch.save(sd, os.path.join(folder, f"model_tp{tp_rank}_pp0.pt"))
        torch.save({}, os.path.join(folder, f"topo_tp{tp_rank}_pp0.json"))
    if rank // ep == 0:   # expert-data rank 0 (the reference: (tp_rank, expert_dp_rank) = (t, t % edp), components.py:270-277 -- the same content again)
        per_expert = collections.OrderedDict()
        for n in groups[2][1]:
            per_expert.setdefault(_expert_file(n, tp_rank), collections.OrderedDict())["model." + n] = params[n].detach().to("cpu", param_dtype).contiguous()
        for fn, esd in per_expert.items():
            torch.save(esd, os.path.join(folder, fn))
    plan = [[_plan_ids(fo, idx) for idx in part] for fo, part, _ in layout]

    def flat(named, g):
        fo, part, zr = layout[g]
        return torch.cat([named[fo[i][0]].detach().to("cpu", torch.float32).reshape(-1) for i in part[zr]])

    with _RefEnumModule() as zero1:
        pm = type(zero1)
        tail = dict(lr=lr, betas=tuple(hyper["betas"]), eps=hyper["eps"], amsgrad=False, maximize=False, foreach=None, capturable=False,
                    differentiable=False, fused=True, decoupled_weight_decay=True)
        wd, ilr = hyper["weight_decay"], hyper["initial_lr"]
        # A rank whose partition of a group is EMPTY (more ranks than parameters: the gates of a shallow model on many ranks; hybrid_zero_optim.py:254-284
        # no_params_ranks) keeps the group's own parameters in the base optimizer (:210-222 replaces them by the flat fp32 shard only where there is one): the
        # group lists ALL its parameters' ids, none of them has optimizer state, the group has no flat fp32 weights, and the ids behind it move up
        # (pinned on ranks 2 and 3 of tests/golden/ckpt_ref_moe_dp4/).
        held = [bool(part[zr]) for _, part, zr in layout]
        ids, nxt = [], 0
        for g in range(3):
            k = 1 if held[g] else len(groups[g][1])
            ids.append(list(range(nxt, nxt + k)))
            nxt += k
        # key order as the reference's groups carry it
        pgs = [dict(name="default", weight_decay=wd, optimizer_mode=zero1, **tail, dtype=param_dtype, initial_lr=ilr, params=ids[0]),
               dict(name="fp32", optimizer_mode=zero1, weight_decay=wd, **tail, dtype=torch.float32, initial_lr=ilr, params=ids[1]),
               dict(name=groups[2][0], moe=True, optimizer_mode=pm.EXPERT_DATA, weight_decay=wd, **tail, dtype=param_dtype, initial_lr=ilr, params=ids[2])]
        states = {
            "grad_scaler": {"_scale": float(scaler["scale"]), "_growth_step": int(scaler["growth_step"]), "_hysteresis_step": int(scaler["hysteresis_step"])},
            "base_optim_states": {
                "state": {ids[g][0]: {"step": torch.tensor(float(adam_step), dtype=torch.float32), "exp_avg": flat(exp_avg, g), "exp_avg_sq": flat(exp_avg_sq, g)}
                          for g in range(3) if held[g]},
                "param_groups": pgs,
            },
            "flat_fp32_weights": {g: flat(master, g) for g in range(3) if held[g]},
            "zero_devide_optim_plan": plan,
        }
        torch.save(states, os.path.join(folder, f"optimizer_tp{tp_rank}_pp0_zo{rank}.pt"))
        torch.save(plan, os.path.join(folder, f"gpus-{world * tp_world}_wp-0_tp-{tp_rank}_dp-{rank}_pp-0_zo-{rank}.pt"))


def _load_moe_tp_rank(folder, model_cfg, t):
    """One tensor rank's files of an INTERNLM_MoE checkpoint -> dict(params, master, exp_avg, exp_avg_sq (name -> LOCAL host tensor, all experts under their global
    numbers; optimizer entries None when the folder holds weights only), adam_step, scaler, lr, zero_world): every data-parallel rank's optimizer shard read and merged."""
    order = state_dict_order(model_cfg, t)
    sd = dict(torch.load(os.path.join(folder, f"model_tp{t}_pp0.pt"), map_location="cpu", weights_only=False))
    for n in order:
        if ".experts." in n and "model." + n not in sd:
            sd.update(torch.load(os.path.join(folder, _expert_file(n, t)), map_location="cpu", weights_only=False))
    params = {n: sd["model." + n].detach() for n in order}
    out = dict(params=params, master=None, exp_avg=None, exp_avg_sq=None, adam_step=None, scaler=None, lr=None, zero_world=0)
    world = saved_zero_world(folder, t)
    if world == 0:
        return out
    shapes = {n: tuple(params[n].shape) for n in order}
    merged = dict(master={}, exp_avg={}, exp_avg_sq={})
    meta = None
    for r in range(world):
        fn = f"optimizer_tp{t}_pp0_zo{r}.pt"
        st = _load(os.path.join(folder, fn))
        layout, ep = _moe_layout(model_cfg, shapes, world, r, t)
        if st["base_optim_states"]["param_groups"][2]["name"] != f"moe_ep_size_{ep}":
            raise ValueError(f"{fn}: expert group {st['base_optim_states']['param_groups'][2]['name']!r}, this model on {world} data-parallel ranks has moe_ep_size_{ep}")
        for g, (fo, part, zr) in enumerate(layout):
            if list(st["zero_devide_optim_plan"][g][zr]) != _plan_ids(fo, part[zr]):
                raise ValueError(f"{fn}: zero_devide_optim_plan of group {g} does not match this model")
            if not part[zr]:   # this rank holds no parameter of the group: no state, no flat weights, the group lists its own parameters (see save_moe_checkpoint)
                continue
            sid = st["base_optim_states"]["param_groups"][g]["params"][0]
            for key, vec in (("master", st["flat_fp32_weights"][g]), ("exp_avg", st["base_optim_states"]["state"][sid]["exp_avg"]),
                             ("exp_avg_sq", st["base_optim_states"]["state"][sid]["exp_avg_sq"])):
                o = 0
                for i in part[zr]:
                    n, shape = fo[i]
                    k = 1
                    for d in shape:
                        k *= d
                    merged[key][n] = vec.detach()[o : o + k].reshape(shape).to(torch.float32)
                    o += k
                if o != vec.numel():
                    raise ValueError(f"{fn} group {g}: the flat vector holds {vec.numel()} elements, this rank's partition {o}")
        gs, base = st["grad_scaler"], st["base_optim_states"]
        # (a data rank whose partition of the default group is empty writes no state entry 0 -- the writer keys the state by the first id of each non-empty
        # group: the step comes from ANY entry present, as load_isp_optimizer reads it; a shard without any state entry names no step)
        any_state = next(iter(base["state"].values()), None)
        here = (int(float(any_state["step"])) if any_state is not None else None, float(base["param_groups"][0]["lr"]), float(gs["_scale"]), int(gs["_growth_step"]),
                int(gs["_hysteresis_step"]))
        if meta is not None and (here[1:] != meta[1:] or (here[0] is not None and meta[0] is not None and here[0] != meta[0])):
            raise ValueError("the optimizer shards disagree on step / lr / loss scale")
        meta = here if (meta is None or meta[0] is None) else meta
    missing = [n for n in order if n not in merged["master"]]
    if missing:
        raise FileNotFoundError(f"{folder}: the optimizer shards present do not cover {missing[:3]} ...")
    out.update(merged, adam_step=meta[0], lr=meta[1], scaler=dict(scale=meta[2], growth_step=meta[3], hysteresis_step=meta[4]), zero_world=world)
    return out


def load_moe_checkpoint(folder, model_cfg):
    """-> dict(params, master, exp_avg, exp_avg_sq (name -> host tensor, FULL tensors, ALL experts under their global numbers), adam_step, scaler, lr, zero_world,
    tp_world) of an INTERNLM_MoE checkpoint (the reference's or save_moe_checkpoint's) written by ANY number of data-parallel ranks and ANY tensor-parallel size:
    every rank's optimizer shard is read and merged, the tensor ranks' parts put together (tp_unshard); optimizer entries None when the folder holds weights only."""
    tp_world = saved_tp_world(folder)
    if tp_world == 0:
        raise FileNotFoundError(f"{folder}: no model_tp*_pp0.pt")
    ranks = [_load_moe_tp_rank(folder, model_cfg, t) for t in range(tp_world)]
    out = dict(ranks[0], tp_world=tp_world)
    if tp_world == 1:
        return out
    same = ("adam_step", "lr", "scaler", "zero_world")
    for t, r in enumerate(ranks[1:], 1):
        if {k: r[k] for k in same} != {k: out[k] for k in same}:
            raise ValueError(f"tensor rank {t} disagrees with rank 0 on the step / scaler / lr / ZeRO world")
    hd = model_cfg.head_dim
    for key in ("params", "master", "exp_avg", "exp_avg_sq"):
        if out[key] is not None:
            out[key] = {n: tp_unshard(n, [r[key][n] for r in ranks if n in r[key]], hd).contiguous() for n in ranks[0][key]}
    return out


def latest_checkpoint(save_folder):
    """CheckpointManager.query_latest_snapshot_step_local (checkpoint_manager.py:497-512): the folder under `save_folder` that holds the `{step}.step`
    flag file with the largest step (the flag is written last, by the logging rank: a folder without it is incomplete) -> (folder or None, step)."""
    best, where = 0, None
    if save_folder and os.path.isdir(save_folder):
        for root, _, files in os.walk(save_folder, followlinks=True):
            for fn in files:
                if fn.endswith(".step") and fn[: -len(".step")].isdigit() and int(fn[: -len(".step")]) > best:
                    best, where = int(fn[: -len(".step")]), root
    return where, best


def save_run_state(folder, scheduler_state, sampler_state, batch_count, num_consumed_samples_in_epoch, num_consumed_tokens,
                   inf_nan_skip_batches, step_count, tensorboard_folder=None):
    """schedulder.pt / sampler.pt / context.pt as CheckpointManager.save_checkpoint writes them from the logging rank
    (checkpoint_manager.py:608-618).  scheduler_state: schedule.CosineWarmupLR.state_dict(); sampler_state:
    data.StaticBatchSampler.state_dict(); the rest are TrainState's counters (core/trainer.py:123-135): batch_count = index of the
    last batch run, step_count = successful optimizer steps."""
    os.makedirs(folder, exist_ok=True)
    torch.save(scheduler_state, os.path.join(folder, "schedulder.pt"))
    if sampler_state is not None:
        torch.save(sampler_state, os.path.join(folder, "sampler.pt"))
    torch.save({"batch_count": int(batch_count), "num_consumed_samples_in_epoch": int(num_consumed_samples_in_epoch),
                "num_consumed_tokens": int(num_consumed_tokens), "inf_nan_skip_batches": int(inf_nan_skip_batches),
                "step_count": int(step_count), "tensorboard_folder": tensorboard_folder}, os.path.join(folder, "context.pt"))
    # the integrity flag the reference writes last (checkpoint_manager.py:633-637); `auto_resume` looks for the largest one
    torch.save({"step": int(step_count)}, os.path.join(folder, f"{int(step_count)}.step"))


def load_run_state(folder):
    """-> dict(scheduler, sampler, context), each None when its file is absent.  A resumed run starts at batch
    context["batch_count"] + 1 (TrainState.load_state_dict, core/trainer.py:114-117)."""
    out = {}
    for key, fn in (("scheduler", "schedulder.pt"), ("sampler", "sampler.pt"), ("context", "context.pt")):
        path = os.path.join(folder, fn)
        out[key] = torch.load(path, map_location="cpu", weights_only=False) if os.path.exists(path) else None
    return out
