"""Configuration of the hot path, read from the reference's own `configs/*.py` schema.

`load_reference_config(path)` executes an InternEvo config file exactly like
internlm/core/context/parallel_context.py:77-127 (Config.from_file: the .py is run and its module
globals become the config) and maps the keys the training step needs onto `PathConfig`.
Anything the path does not implement (pipeline parallel, MoE ...) is rejected loudly
instead of being ignored.
"""
import dataclasses
import runpy
from typing import Optional


@dataclasses.dataclass
class ModelConfig:
    # configs/7B_internlm2.py:5-11,122-141
    vocab_size: int = 92544
    hidden_size: int = 4096
    num_layers: int = 32
    num_attention_heads: int = 32
    num_kv_attention_heads: int = 8
    mlp_ratio: float = 3.5
    layer_norm_epsilon: float = 1e-5
    rope_base: int = 10000
    model_type: str = "INTERNLM2_PUBLIC"  # or "LLAMA2" (modeling_llama.py): separate wq / wk / wv instead of the GQA-interleaved wqkv
    adapt_hf: bool = True           # builder default (modeling_internlm2.py:1071); False = even/odd de-interleave before rotary (:425-427)
    multiple_of: int = 256          # modules/mlp.py:52
    checkpoint: float = 0.0         # fraction of layers under activation checkpointing (launch.py:295-303; True -> 1, False -> 0)
    dtype: str = "torch.bfloat16"
    # explicit per-rank sizes of a tensor-parallel shard (engine-internal: ModelConfig.tp_shard); None = derived from the above
    head_dim_override: Optional[int] = None
    ffn_dim_override: Optional[int] = None
    head_vocab_override: Optional[int] = None   # tensor parallelism: vocabulary rows of the head on this rank
    embed_dim_override: Optional[int] = None    # tensor parallelism with embed_split_hidden: hidden columns of the embedding on this rank
    embed_split_hidden: bool = False            # model.embed_split_hidden (modules/embedding.py:24-60): under tensor parallelism every rank holds
                                                # h / tp columns of the embedding and the looked-up rows are all-gathered along the hidden dim
    # init (modeling_internlm2.py:646-672; scaled init for wo / w2)
    init_std: float = 0.02
    use_scaled_init: bool = True
    # model_type "INTERNLM_MoE" (modeling_moe.py, configs/7B_MoE4_sft.py): InternLM-1 block (MHA with biases, no GQA) + GShard top-2 MoE
    num_experts: int = 1
    moe_capacity_factor: float = 1.0     # moe = dict(capacity_factor, min_capacity, ...)
    moe_expert_fp8: bool = False         # moe = dict(..., expert_fp8=True): THIS repo's opt-in (the reference has no fp8 linear): the experts' forward products on e4m3
    moe_min_capacity: int = 4
    moe_loss_coeff: float = 1.0          # loss.moe_loss_coeff (launch.py:433-434 default)
    # ScaleColumnParallelLinearWithNormHead (ops/linear.py:79-153) and the embedding's gradient scale (modeling_internlm2.py:970-973)
    embed_grad_scale: float = 1.0        # s: x -> s x + (1 - s) x.detach() on the embedding output AND on the head weight (weight_scale = s)
    norm_head: bool = False              # the head multiplies by its weight with every row scaled to unit length (F.normalize)

    @property
    def attn_bias(self):
        """The InternLM-1 block (modeling_internlm.py / modeling_moe.py; multi_head_attention.py:371-408): Wqkv and out_proj carry a bias."""
        return self.model_type in ("INTERNLM", "INTERNLM_MoE")

    @property
    def head_dim(self):
        return self.head_dim_override or self.hidden_size // self.num_attention_heads

    @property
    def q_per_kv(self):
        return self.num_attention_heads // self.num_kv_attention_heads

    @property
    def qkv_dim(self):
        return (self.num_attention_heads + 2 * self.num_kv_attention_heads) * self.head_dim

    @property
    def ffn_dim(self):
        if self.ffn_dim_override:
            return self.ffn_dim_override
        f = int(self.hidden_size * self.mlp_ratio)
        return self.multiple_of * ((f + self.multiple_of - 1) // self.multiple_of)

    @property
    def checkpoint_layers(self):
        """modeling_internlm2.py:857-861,910: layer lid is checkpointed iff lid < num_layers * checkpoint_fraction."""
        lim = self.num_layers * float(self.checkpoint)
        return sum(1 for lid in range(self.num_layers) if lid < lim)

    @property
    def head_vocab(self):
        """Rows of the output head THIS rank holds: the whole vocabulary, or 1/tp of it under the vocabulary-parallel head."""
        return self.head_vocab_override or self.vocab_size

    @property
    def embed_dim(self):
        """Hidden columns of the embedding THIS rank holds."""
        return self.embed_dim_override or self.hidden_size

    def tp_shard(self, tp, vocab_parallel=True):
        """The model one rank of a tensor-parallel group of size tp holds (Megatron "mtp" split, model/ops/linear.py:205-337):
        1/tp of the attention heads (whole kv groups), 1/tp of the FFN width and -- vocab_parallel, the reference's
        `parallel_output=True` head (ops/linear.py:124-153) -- 1/tp of the head's vocabulary rows; hidden size, norms and the
        embedding are whole."""
        if tp == 1:
            return self
        if self.num_kv_attention_heads % tp or self.ffn_dim % tp:
            raise ValueError(f"tensor parallel size {tp} must divide the kv head count and the FFN width")
        if vocab_parallel and self.vocab_size % tp:
            raise ValueError(f"tensor parallel size {tp} must divide the vocabulary size {self.vocab_size}")
        return dataclasses.replace(self, num_attention_heads=self.num_attention_heads // tp, num_kv_attention_heads=self.num_kv_attention_heads // tp,
                                   head_dim_override=self.head_dim, ffn_dim_override=self.ffn_dim // tp,
                                   head_vocab_override=self.vocab_size // tp if vocab_parallel else None,
                                   embed_dim_override=self.hidden_size // tp if self.embed_split_hidden else None)

    def num_params(self):
        h, f, v = self.hidden_size, self.ffn_dim, self.vocab_size
        per_layer = self.qkv_dim * h + h * h + 3 * f * h + 2 * h + (self.qkv_dim + h if self.attn_bias else 0)
        return per_layer * self.num_layers + 2 * v * h + h


@dataclasses.dataclass
class TrainConfig:
    seq_len: int = 4096
    micro_bsz: int = 1
    micro_num: int = 4
    total_steps: int = 20
    fixed_random_dataset_seqlen: bool = False
    skip_batches: str = ""   # data.skip_batches, e.g. "1-3,5": batches drawn from the loader but not trained on (train.py:187,208-212; data.BatchSkipper)
    # adam (configs/7B_internlm2.py:92-99)
    lr: float = 1e-4
    adam_beta1: float = 0.9
    adam_beta2: float = 0.95
    adam_beta2_c: float = 0.0
    adam_eps: float = 1e-8
    weight_decay: float = 0.01
    # lr scheduler (:101-107)
    warmup_ratio: float = 0.01
    eta_min: float = 1e-5
    init_steps: int = 0
    # grad scaler (:73-83) / hybrid zero (:84-89)
    initial_scale: float = 2.0**16
    min_scale: float = 1.0
    growth_interval: int = 1000
    growth_factor: float = 2.0
    backoff_factor: float = 0.5
    max_scale: float = 2.0**24
    hysteresis: int = 2
    clip_grad_norm: float = 1.0
    label_smoothing: float = 0.0
    zero1_size: int = -1
    wp_size: int = 1                # parallel.weight = dict(size=wp): ISP weight parallelism -- the engine shards the layer weights over groups of wp ranks
                                    # only when they do not fit resident (engine.py, weight_parallel)
    sp_size: int = 1                # parallel.tensor = dict(size=sp, mode="isp"): Ulysses / ISP sequence parallelism (seqpar.py)
    sp_attention: str = "auto"      # parallel.tensor = dict(..., attention="ulysses" | "ring" | "auto") -- an extension of this repo: how the attention of an isp
                                    # run sees the whole sequence: the reference's head exchange (DistributedAttention), or K / V blocks travelling around the
                                    # sequence group (seqpar.RingAttention: no limit on the kv head count); auto = ulysses where the kv heads divide, else ring
    tp_size: int = 1                # parallel.tensor = dict(size=tp, mode="mtp"): Megatron tensor parallelism of the layers (tensorpar.py)
    tp_mode: str = "mtp"            # "msp" / "fsp": the same shards with the activations between the linears sharded along the sequence
    pp_size: int = 1                # parallel.pipeline = dict(size=pp): 1F1B pipeline parallelism (pipeline.py)
    num_chunks: int = 1             # model.num_chunks: model chunks per pipeline stage (> 1: the interleaved 1F1B schedule)

    @property
    def packed_length(self):
        return self.seq_len * self.micro_bsz


@dataclasses.dataclass
class PathConfig:
    model: ModelConfig
    train: TrainConfig


_UNSUPPORTED = "internevo_amd implements the data-parallel InternLM2 hot path only"


def _parse_dtype(s):
    s = str(s)
    return s if s.startswith("torch.") else "torch." + s


def from_reference_dict(cfg: dict, seq_len: Optional[int] = None) -> PathConfig:
    m, d = cfg["model"], cfg["data"]
    par = cfg.get("parallel", {})
    pipe = par.get("pipeline", {})
    pipe = pipe if isinstance(pipe, dict) else dict(size=pipe)
    pp_size = int(pipe.get("size", 1))
    tensor = par.get("tensor", {})
    tensor = tensor if isinstance(tensor, dict) else dict(size=tensor, mode="mtp")  # launch.py normalises an int the same way
    sp_size = tp_size = 1
    tp_mode = "mtp"
    sp_attention = str(tensor.get("attention", "auto"))
    if sp_attention not in ("auto", "ulysses", "ring"):
        raise ValueError(f"parallel.tensor.attention = {sp_attention!r}: 'auto', 'ulysses' or 'ring'")
    if tensor.get("size", 1) != 1:
        mode = tensor.get("mode", "mtp")
        if mode == "isp":
            sp_size = int(tensor["size"])
        elif mode in ("mtp", "msp", "fsp"):
            # msp / fsp (Megatron sequence parallelism, without / with overlap: model/ops/linear.py:338-441, utils.py:160-226) hold the
            # SAME parameter shards as mtp; the activations between the linears are sharded along the sequence (reduce-scatter after a
            # row-parallel linear + all-gather in front of the next column-parallel one, the same bytes as mtp's all-reduce), the norms and
            # residual adds run on T / tp rows, and the norm weights' gradients are AVERAGED over the tensor group (hybrid_zero_optim.py:315-353) --
            # engine.py `seq_shard`.  The two modes differ in the reference only in how the exchanges overlap the products.
            tp_size = int(tensor["size"])
            tp_mode = mode
        else:
            raise NotImplementedError(f"{_UNSUPPORTED}: tensor parallel mode {mode!r} (supported: 'mtp', 'msp', 'fsp', 'isp')")
    # parallel.weight (size, overlap, memory_pool): the weight-parallel size is honoured when the resident layout does not fit the GPU (or when
    # the engine is told to, weight_parallel=True); overlap / memory_pool describe what the engine always does then (prefetch on a side stream
    # into a two-slot pool)
    weight = par.get("weight", {})
    wp_size = int(weight.get("size", 1) if isinstance(weight, dict) else weight)
    if wp_size > 1 and tensor.get("mode", "mtp") != "isp" and tensor.get("size", 1) != 1:
        raise NotImplementedError(f"{_UNSUPPORTED}: parallel.weight.size > 1 outside tensor mode 'isp'")
    if pp_size > 1 and wp_size > 1:
        raise NotImplementedError(f"{_UNSUPPORTED}: pipeline parallelism together with parallel.weight.size > 1")
    model_type = cfg.get("model_type", "INTERNLM")   # the reference's default (initialize/launch.py:78-79)
    if model_type not in ("INTERNLM2_PUBLIC", "LLAMA2", "INTERNLM_MoE", "INTERNLM"):
        raise NotImplementedError(f"{_UNSUPPORTED}: model_type {model_type}")
    moe_kw = {}
    if model_type == "INTERNLM" or (model_type == "INTERNLM_MoE" and m.get("num_experts", 1) <= 1):
        # the dense InternLM-1 model (modeling_internlm.py; configs/7B_sft.py, configs/7B_isp_sft.py; INTERNLM_MoE with one expert builds the same block):
        # engine.InternLM2Engine's InternLM-1 block variant (packed Wqkv "(three h d)" + biases) under every parallel mode of that engine
        if not m.get("use_swiglu", True) or m.get("residual_in_fp32", False) or m.get("num_kv_attention_heads", m["num_attention_heads"]) != m["num_attention_heads"]:
            raise NotImplementedError(f"{_UNSUPPORTED}: InternLM-1 with use_swiglu=False / residual_in_fp32 / grouped-query attention")
        moe_kw = dict(num_experts=1)
        model_type = "INTERNLM"
    elif model_type == "INTERNLM_MoE":
        # configs/7B_MoE4_sft.py: GShard top-2 MoE in every block (internevo_amd/moe_engine.py)
        moe = cfg.get("moe", {}) or {}
        if m.get("moe_type", "GShard") != "GShard":
            raise NotImplementedError(f"{_UNSUPPORTED}: model.moe_type {m.get('moe_type')!r} (GShard only)")
        if m.get("moe_use_residual", False):
            raise NotImplementedError(f"{_UNSUPPORTED}: model.moe_use_residual")
        if moe.get("top_k", 1) != 2:
            raise NotImplementedError(f"{_UNSUPPORTED}: moe.top_k = {moe.get('top_k', 1)} (top-2 gating only)")
        if not moe.get("drop_tokens", True) or moe.get("noisy_gate_policy", None) not in (None, "None"):
            raise NotImplementedError(f"{_UNSUPPORTED}: moe.drop_tokens=False / moe.noisy_gate_policy")
        if sp_size > 1 or pp_size > 1 or (tp_size > 1 and tp_mode != "mtp"):
            raise NotImplementedError(f"{_UNSUPPORTED}: INTERNLM_MoE with sequence / pipeline parallelism or the sequence-sharded tensor modes msp / fsp "
                                      "(data, expert and Megatron 'mtp' tensor parallelism only)")
        if not m.get("use_swiglu", True) or m.get("residual_in_fp32", False):
            raise NotImplementedError(f"{_UNSUPPORTED}: INTERNLM_MoE with use_swiglu=False / residual_in_fp32")
        moe_kw = dict(num_experts=int(m["num_experts"]), moe_capacity_factor=float(moe.get("capacity_factor", 1.0)),
                      moe_min_capacity=int(moe.get("min_capacity", 4)), moe_loss_coeff=float(cfg.get("loss", {}).get("moe_loss_coeff", 1.0)),
                      moe_expert_fp8=bool(moe.get("expert_fp8", False)))
    elif m.get("num_experts", 1) > 1:
        raise NotImplementedError(f"{_UNSUPPORTED}: num_experts > 1 outside model_type INTERNLM_MoE")
    ck = m.get("checkpoint", False)
    ck = 1.0 if ck is True else 0.0 if ck is False else float(ck)
    if not 0.0 <= ck <= 1.0:
        raise ValueError(f'model.checkpoint: "{ck}" should >=0 and <=1')  # launch.py:300-303
    if not m.get("no_bias", True) and model_type not in ("INTERNLM_MoE", "INTERNLM"):   # (the InternLM-1 block always has attention biases)
        raise NotImplementedError(f"{_UNSUPPORTED}: linear bias")
    # settings that change the arithmetic of the step: refuse them instead of training something else than the config describes
    if cfg.get("use_fp32_norm", False):
        raise NotImplementedError(f"{_UNSUPPORTED}: use_fp32_norm (the norms run on bf16 activations with fp32 statistics)")
    if m.get("norm_type", "rmsnorm") != "rmsnorm":
        raise NotImplementedError(f"{_UNSUPPORTED}: model.norm_type {m.get('norm_type')!r} (rmsnorm only)")
    if m.get("apply_post_layer_norm", False):
        raise NotImplementedError(f"{_UNSUPPORTED}: model.apply_post_layer_norm")
    if (m.get("embed_grad_scale", 1) != 1 or m.get("norm_head", False)) and (model_type != "INTERNLM2_PUBLIC" or pp_size > 1):   # (norm_head: InternLM2's head only)
        raise NotImplementedError(f"{_UNSUPPORTED}: model.embed_grad_scale != 1 / model.norm_head outside the InternLM2 family or with pipeline parallelism")
    num_chunks = int(m.get("num_chunks", 1))
    if num_chunks > 1:
        # InterleavedPipelineScheduler (pipeline_scheduler.py:736-757) and partition_uniform (pipeline_utils.py:9-12) assert the same
        if pp_size == 1:
            num_chunks = 1   # (without pipeline parallelism the chunks of a rank are one model: nothing to interleave)
        elif d["micro_num"] % pp_size:
            raise ValueError(f"num_microbatches: {d['micro_num']} must be an integer multiple of pipeline parallel world size")
        elif m["num_layers"] % num_chunks:
            raise ValueError("Layer length should be divided by the number of chunks, otherwise parameter method is recomended")
    for key in ("drop_rate", "attn_drop_rate", "dropout"):
        if m.get(key, 0):
            raise NotImplementedError(f"{_UNSUPPORTED}: model.{key} > 0 (the path trains without dropout)")
    if m.get("multiple_of", 256) != 256:
        raise NotImplementedError(f"{_UNSUPPORTED}: model.multiple_of != 256")
    if not d.get("use_packed_dataset", True):
        raise NotImplementedError(f"{_UNSUPPORTED}: data.use_packed_dataset=False (the loaders build PackedDatasetWithCut batches only)")
    if d.get("rampup_batch_size", ""):
        raise NotImplementedError(f"{_UNSUPPORTED}: data.rampup_batch_size")
    zero1 = par.get("zero1", {})
    if isinstance(zero1, dict) and zero1.get("fsdp", False):
        raise NotImplementedError(f"{_UNSUPPORTED}: parallel.zero1.fsdp")
    if _parse_dtype(m.get("dtype", "torch.bfloat16")) != "torch.bfloat16":
        raise NotImplementedError(f"{_UNSUPPORTED}: the HIP path computes in bf16")
    model = ModelConfig(
        vocab_size=m["vocab_size"], hidden_size=m["hidden_size"], num_layers=m["num_layers"],
        num_attention_heads=m["num_attention_heads"], num_kv_attention_heads=m.get("num_kv_attention_heads", m["num_attention_heads"]),
        mlp_ratio=m.get("mlp_ratio", 4), layer_norm_epsilon=m.get("layer_norm_epsilon", 1e-5), rope_base=m.get("rope_base", 10000),
        # builder defaults differ: modeling_internlm2.py:1071 adapt_hf=True, modeling_llama.py:1039 adapt_hf=False
        adapt_hf=True if model_type in ("INTERNLM", "INTERNLM_MoE") else m.get("adapt_hf", model_type != "LLAMA2"), checkpoint=ck, model_type=model_type,
        embed_grad_scale=float(m.get("embed_grad_scale", 1)), norm_head=bool(m.get("norm_head", False)),
        embed_split_hidden=bool(m.get("embed_split_hidden", False)) and model_type != "INTERNLM_MoE", **moe_kw,
    )
    adam, ls, gs = cfg["adam"], cfg["lr_scheduler"], cfg["grad_scaler"]
    hz = cfg["hybrid_zero_optimizer"]
    train = TrainConfig(
        seq_len=seq_len or d["seq_len"], micro_bsz=d["micro_bsz"], micro_num=d["micro_num"], total_steps=d["total_steps"],
        fixed_random_dataset_seqlen=d.get("fixed_random_dataset_seqlen", False), skip_batches=str(d.get("skip_batches", "") or ""),
        lr=adam["lr"], adam_beta1=adam["adam_beta1"], adam_beta2=adam["adam_beta2"], adam_beta2_c=adam.get("adam_beta2_c", 0),
        adam_eps=adam["adam_eps"], weight_decay=adam["weight_decay"],
        warmup_ratio=ls.get("warmup_ratio", 0.0), eta_min=ls.get("eta_min", 0.0), init_steps=ls.get("init_steps", 0),
        initial_scale=gs["fp16"]["initial_scale"], min_scale=gs["fp16"].get("min_scale", 1), growth_interval=gs["fp16"]["growth_interval"],
        growth_factor=gs["growth_factor"], backoff_factor=gs["backoff_factor"], max_scale=gs.get("max_scale", 2**24), hysteresis=gs["hysteresis"],
        clip_grad_norm=hz["clip_grad_norm"], label_smoothing=cfg.get("loss", {}).get("label_smoothing", 0) or 0.0,
        zero1_size=par.get("zero1", {}).get("size", -1) if isinstance(par.get("zero1", {}), dict) else par.get("zero1", -1),
        sp_size=sp_size, tp_size=tp_size, tp_mode=tp_mode, pp_size=pp_size, num_chunks=num_chunks, wp_size=wp_size, sp_attention=sp_attention,
    )
    return PathConfig(model, train)


def run_reference_config(path: str) -> dict:
    """The variables an InternEvo `configs/*.py` defines (Config.from_file, parallel_context.py:77-127).  Pure-Python-style configs (configs/demo.py:2-6:
    `with read_base(): from configs._base_... import *`) resolve without the reference installed: `read_base` is an empty context manager
    (internlm/utils/utils.py:5-18) -- supplied here when `internlm` is not importable -- and the `configs.` package is found next to the
    file (the folder above the file's own folder goes onto sys.path for the duration of the run)."""
    import contextlib
    import importlib.util
    import os
    import sys
    import types

    path = os.path.abspath(path)
    root = os.path.dirname(os.path.dirname(path))
    stubs = []
    if importlib.util.find_spec("internlm") is None:
        for name in ("internlm", "internlm.utils", "internlm.utils.utils"):
            mod = types.ModuleType(name)
            mod.__path__ = []
            sys.modules[name] = mod
            stubs.append(name)
        sys.modules["internlm.utils.utils"].read_base = contextlib.contextmanager(lambda: (yield))
    sys.path.insert(0, root)
    before = set(sys.modules)
    try:
        g = runpy.run_path(path)
    finally:
        sys.path.remove(root)
        for name in stubs + [m for m in set(sys.modules) - before if m == "configs" or m.startswith("configs.")]:
            sys.modules.pop(name, None)
    return {k: v for k, v in g.items() if not k.startswith("__") and not isinstance(v, types.ModuleType) and not callable(v)}


def load_reference_config(path: str, seq_len: Optional[int] = None) -> PathConfig:
    """Run an InternEvo `configs/*.py` (run_reference_config) and map it onto this engine's PathConfig."""
    return from_reference_dict(run_reference_config(path), seq_len)


def internlm2_7b(seq_len=4096) -> PathConfig:
    """configs/7B_internlm2.py with BASELINE.json's seq-4096 override."""
    return PathConfig(ModelConfig(), TrainConfig(seq_len=seq_len))


def llama2_7b(seq_len=4096) -> PathConfig:
    """configs/7B_llama2.py (BASELINE.json configs[2]'s model; the shipped file has tensor size 1): LLAMA2, vocab 32000, GQA 32/8."""
    return PathConfig(ModelConfig(vocab_size=32000, model_type="LLAMA2", adapt_hf=False), TrainConfig(seq_len=seq_len))


def tiny(hidden=512, layers=2, heads=8, kv_heads=2, vocab=1024, seq_len=256, micro_num=2, lr=1e-3, total_steps=5, model_type="INTERNLM2_PUBLIC",
         embed_grad_scale=1.0, norm_head=False) -> PathConfig:
    """BASELINE.json configs[0]: the CPU-runnable plumbing case (SURVEY.md section 8d "tiny config")."""
    return PathConfig(
        ModelConfig(vocab_size=vocab, hidden_size=hidden, num_layers=layers, num_attention_heads=heads, num_kv_attention_heads=kv_heads,
                    model_type=model_type, adapt_hf=model_type != "LLAMA2", embed_grad_scale=embed_grad_scale, norm_head=norm_head),
        TrainConfig(seq_len=seq_len, micro_bsz=1, micro_num=micro_num, total_steps=total_steps, lr=lr, fixed_random_dataset_seqlen=True),
    )
