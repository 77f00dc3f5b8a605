"""Flat parameter / gradient layout of the InternLM2 path and its ZeRO-1 buckets (pure host logic).

The reference keeps 227 separate parameter tensors, flattens gradients into buckets on the fly
(`_flatten_dense_tensors`, hybrid_zero_optim.py:503-523), unflattens after the all-reduce and flattens
again into per-rank partitions for the optimizer (:738-799).  Here parameters and gradients live in
ONE contiguous bf16 buffer each from the start (no flatten / unflatten passes),
laid out in forward order and cut into buckets = {embedding, layer 0, ..., layer L-1, final norm + head}.
Every bucket is padded to a multiple of `world * ALIGN` elements so that
    reduce_scatter_tensor(bucket)  ->  this rank's contiguous 1/world shard of the bucket
    all_gather_into_tensor(bucket) <-  the updated bf16 shard
need no copies; the fp32 master / Adam state of a rank is the concatenation of its bucket shards.
Parameter names are the reference's (`PackedFlashLlama1D.named_parameters()`), so a state dict
round-trips with InternEvo checkpoints.
"""
import dataclasses
from typing import Dict, List, Tuple

ALIGN = 8  # elements (16 bytes of bf16)


@dataclasses.dataclass
class ParamSpec:
    name: str
    shape: Tuple[int, ...]
    offset: int      # element offset in the flat buffer
    bucket: int
    kind: str        # "embed" | "norm" | "wqkv" | "wo" | "w1" | "w3" | "w2" | "head" | "bqkv" | "bo" (the InternLM-1 block's attention biases)
    layer: int = -1

    @property
    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n


@dataclasses.dataclass
class Bucket:
    index: int
    start: int       # element offset (multiple of ALIGN)
    size: int        # padded size, multiple of world * ALIGN
    used: int        # elements actually holding parameters
    params: List[str]

    def shard(self, rank, world):
        n = self.size // world
        return self.start + rank * n, n


def _round_up(x, m):
    return (x + m - 1) // m * m


class FlatLayout:
    def __init__(self, model_cfg, world_size=1, layer_lo=0, first=True, last=True):
        """layer_lo / first / last: the layout of ONE pipeline stage -- model_cfg.num_layers layers numbered layer_lo ... in the
        parameter names (the reference's global layer numbers; a LIST of numbers for a stage that holds several model chunks), the
        embedding bucket only on the first stage and the norm + head bucket only on the last one.  The bucket list keeps its shape on every stage (index 0 = embedding, 1 + i = local layer i, last =
        norm + head): a bucket a stage does not own is EMPTY (size 0)."""
        self.cfg = model_cfg
        self.world = world_size
        self.layer_lo, self.first, self.last = layer_lo, first, last
        c = model_cfg
        h, f, v = c.hidden_size, c.ffn_dim, c.vocab_size
        self.params: Dict[str, ParamSpec] = {}
        self.buckets: List[Bucket] = []
        off = 0

        def open_bucket():
            return {"start": off, "names": []}

        def add(cur, name, shape, kind, layer=-1):
            nonlocal off
            spec = ParamSpec(name, tuple(shape), off, len(self.buckets), kind, layer)
            self.params[name] = spec
            cur["names"].append(name)
            off += _round_up(spec.numel, ALIGN)

        def close_bucket(cur):
            nonlocal off
            used = off - cur["start"]
            size = _round_up(used, self.world * ALIGN)
            self.buckets.append(Bucket(len(self.buckets), cur["start"], size, used, cur["names"]))
            off = cur["start"] + size

        cur = open_bucket()
        if first:
            add(cur, "tok_embeddings.weight", (v, getattr(c, "embed_dim", h)), "embed")   # all hidden columns, or this tensor rank's h / tp (embed_split_hidden)
        close_bucket(cur)
        layer_ids = list(layer_lo) if isinstance(layer_lo, (list, tuple)) else list(range(layer_lo, layer_lo + c.num_layers))
        assert len(layer_ids) == c.num_layers
        self.layer_ids = layer_ids
        for l in layer_ids:
            cur = open_bucket()
            p = f"layers.{l}."
            add(cur, p + "attention_norm.weight", (h,), "norm", l)
            add(cur, p + "attention.wqkv.weight", (c.qkv_dim, h), "wqkv", l)
            if getattr(c, "attn_bias", False):   # the InternLM-1 block (multi_head_attention.py:371-408): Wqkv and out_proj carry a bias
                add(cur, p + "attention.wqkv.bias", (c.qkv_dim,), "bqkv", l)
            add(cur, p + "attention.wo.weight", (h, c.num_attention_heads * c.head_dim), "wo", l)  # = (h, h) unless tensor-parallel
            if getattr(c, "attn_bias", False):
                add(cur, p + "attention.wo.bias", (h,), "bo", l)
            add(cur, p + "ffn_norm.weight", (h,), "norm", l)
            add(cur, p + "feed_forward.w1.weight", (f, h), "w1", l)   # w1 and w3 adjacent: one [2F, h] GEMM operand
            add(cur, p + "feed_forward.w3.weight", (f, h), "w3", l)
            add(cur, p + "feed_forward.w2.weight", (h, f), "w2", l)
            close_bucket(cur)
        cur = open_bucket()
        if last:
            add(cur, "norm.weight", (h,), "norm")
            add(cur, "output.weight", (c.head_vocab, h), "head")   # all vocabulary rows, or this tensor rank's 1/tp of them
        close_bucket(cur)
        self.total = off

    # ---- ZeRO-1 shard bookkeeping -------------------------------------------------------------
    def shard_sizes(self):
        return [b.size // self.world for b in self.buckets]

    def local_numel(self):
        return sum(self.shard_sizes())

    def local_offsets(self):
        """Offset of each bucket's shard inside a rank's concatenated fp32 master buffer."""
        offs, o = [], 0
        for n in self.shard_sizes():
            offs.append(o)
            o += n
        return offs

    def names(self):
        return list(self.params.keys())

    def reference_param_order(self):
        """Order of `model.named_parameters()` in the reference (modeling_internlm2.py: tok_embeddings, layers
        [attention.wqkv, attention.wo, feed_forward.w1, w2(?), ...], norm, output) is NOT relied upon; lookups go by name."""
        return self.names()
