"""Ulysses sequence parallelism of the attention and the data layout of the ISP mode (a19).

Reference behaviour being matched (`tensor=dict(size=sp, mode="isp")`, configs/7B_isp_sft.py):
  * every micro-batch is split contiguously along the packed sequence over the `tensor` group: rank j of the group owns
    tokens [j*T/sp, (j+1)*T/sp) (Embedding1D splits dim 1, modules/embedding.py:52-60; `indexes` is split the same way,
    modeling_internlm2.py:985-987); all token-wise work (norms, linears, SwiGLU, rotary) runs on the local tokens;
  * attention needs the whole sequence: `DistributedAttention` (modules/multi_head_attention.py:56-135) exchanges
    "my tokens, all heads" for "all tokens, my heads" with an all-to-all before the local attention (q: scatter heads /
    gather sequence, packed kv likewise) and the inverse exchange on the context; backward is the mirrored exchange;
  * ISPLinear (ops/linear.py:357-378, core/communication/isp.py) keeps 1/wp of every weight per rank, all-gathers it before
    each use (forward and backward) and reduce-scatters the weight gradient with AVG over the weight group.

MI355X redesign with the same results: 288 GB of HBM hold the whole bf16 model on every GPU, so the weights stay resident
(they are already exchanged once per step by the ZeRO-1 all-gather) and the per-layer weight all-gather / reduce-scatter
traffic of ISP (3x the parameter bytes per micro-batch) disappears; what remains of a19 is its gradient averaging RULE, which
the engine reproduces (engine.py `_apply_isp_grad_rule`).  The exchange itself is ONE `all_to_all_single` per tensor over
xGMI with a single packing copy on the send side (ie_seq_head_permute); the receive buffer already is the gathered tensor.
"""
import torch.distributed as dist

from . import kernels as K
from .comm import backend_for


class SeqParallel:
    def __init__(self, sp_size, rank, world_size, stages=1, stage=0):
        """rank / world_size: this rank inside ONE pipeline stage and the stage's size (the whole job without pipeline parallelism); stages / stage: the
        pipeline size and this rank's stage (stage s = the global ranks s * world_size ...): every rank of the job creates every group of every stage."""
        if world_size % sp_size != 0:
            raise ValueError(f"world size {world_size} is not a multiple of the sequence-parallel size {sp_size}")
        self.sp = sp_size
        self.sp_rank = rank % sp_size
        self.data_rank = rank // sp_size          # ranks of one sequence group read the same batch (data/build_dataloader.py:54-63)
        self.data_world = world_size // sp_size
        self.group = None
        self.ranks = [rank]                       # the global ranks of this rank's sequence group, in group order
        self.backend = None
        self.be = None
        if sp_size > 1:
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed must be initialised for sequence parallelism")
            # consecutive ranks share a sequence (parallel_context.py: the tensor group is the innermost dimension);
            # every rank has to take part in the creation of every group
            me = stage * world_size + rank
            for s_ in range(stages):
                for g in range(world_size // sp_size):
                    ranks = [s_ * world_size + r for r in range(g * sp_size, (g + 1) * sp_size)]
                    grp = dist.new_group(ranks)
                    if me in ranks:
                        self.group = grp
                        self.ranks = ranks
            self.be = backend_for(self.group)
            self.backend = self.be.name

    # ---- the exchange: send[r] (contiguous chunk r) -> rank r; recv[s] <- rank s -------------------------------------
    def all_to_all_async(self, send, recv):
        """Start the exchange; returns a handle whose .wait() orders the current stream behind it.  On RCCL it runs on c10d's
        stream (after the work already queued on the current stream), so kernels launched before .wait() overlap it -- the engine
        puts the neighbouring weight-gradient GEMM / the other tensor's packing copy there.  `recv` is valid only after .wait()."""
        # flat views: chunk r of the send buffer is its r-th 1/sp, whatever shape the caller gives the tensors
        return self.be.all_to_all(recv, send, self.group)

    def all_to_all(self, send, recv):
        self.all_to_all_async(send, recv).wait()
        return recv

    def all_reduce_sum(self, t):
        self.be.all_reduce(t, self.group).wait()
        return t

    # ---- heads <-> sequence ------------------------------------------------------------------------------------------
    def scatter_heads_gather_seq_async(self, x_local, B, pack_buf, out_full):
        """x_local [Tl, (B,) heads, d] -> out_full [T, (B,) heads/sp, d] (q, kv, d_ctx: _SeqAllToAll scatter_idx = head dim).
        Returns a handle; out_full is valid after .wait()."""
        Tl = x_local.shape[0]
        C = x_local.numel() // (Tl * B * self.sp)
        K.seq_head_permute(x_local, pack_buf, Tl, B, self.sp, C, inverse=False)
        return _Exchange(self.all_to_all_async(pack_buf, out_full), out_full)

    def scatter_heads_gather_seq(self, x_local, B, pack_buf, out_full):
        return self.scatter_heads_gather_seq_async(x_local, B, pack_buf, out_full).wait()

    def scatter_seq_gather_heads_async(self, x_full, B, recv_buf, out_local):
        """x_full [T, (B,) heads/sp, d] -> out_local [Tl, (B,) heads, d] (context, dq, dkv: the inverse exchange).
        Returns a handle; .wait() orders the stream behind the exchange and unpacks into out_local."""
        Tl = out_local.shape[0]
        C = out_local.numel() // (Tl * B * self.sp)
        sp = self.sp
        return _Exchange(self.all_to_all_async(x_full, recv_buf), out_local,
                         lambda: K.seq_head_permute(recv_buf, out_local, Tl, B, sp, C, inverse=True))

    def scatter_seq_gather_heads(self, x_full, B, recv_buf, out_local):
        return self.scatter_seq_gather_heads_async(x_full, B, recv_buf, out_local).wait()


class _Exchange:
    """An exchange in flight: wait() orders the current stream behind it, runs the unpacking copy if there is one, returns the result."""

    def __init__(self, work, out, after=None):
        self.work, self.out, self.after = work, out, after

    def wait(self):
        self.work.wait()
        if self.after is not None:
            self.after()
            self.after = None
        return self.out


# ---- ring attention -------------------------------------------------------------------------------------------------------------------------
def ring_plan(cu, sp, rank):
    """The geometry of ring attention for sequence rank `rank` of `sp`: `cu` = the boundaries of the packed sequences of one micro-batch (cu[0] = 0,
    cu[-1] = T, T % sp == 0); rank j owns tokens [j T/sp, (j+1) T/sp).  Causal attention inside packed sequences needs, besides the rank's own
    block (its local boundaries: `cu_local`), exactly ONE rectangle per earlier block: only the sequence that is running when the rank's block starts
    reaches into earlier blocks -- its `Lq` leading rows of the block against the rows [koff, T/sp) of block r that belong to it, all of which lie in
    front of all of those queries (no mask).  `steps[s - 1]` = (r, koff, Lk) for the block held in ring step s (the block of rank (rank - s) mod sp),
    Lk = 0 when that block holds nothing this rank's queries see."""
    cu = [int(c) for c in cu]
    T = cu[-1]
    if cu[0] != 0 or T % sp != 0 or any(b < a for a, b in zip(cu, cu[1:])):
        raise ValueError("ring_plan: boundaries must start at 0, ascend, and end at a multiple of the sequence-parallel size")
    Tl = T // sp
    a, b = rank * Tl, (rank + 1) * Tl
    cu_local = [0] + sorted({c - a for c in cu if a < c < b}) + [Tl]
    start0 = max(c for c in cu if c <= a)         # start of the sequence that token `a` belongs to
    Lq = cu_local[1] if start0 < a else 0
    steps = []
    for s in range(1, sp):
        r = (rank - s) % sp
        if r < rank and Lq > 0 and start0 < (r + 1) * Tl:
            koff = max(start0 - r * Tl, 0)
            steps.append((r, koff, Tl - koff))
        else:
            steps.append((r, 0, 0))
    return {"Tl": Tl, "cu_local": cu_local, "max_local": max(y - x for x, y in zip(cu_local, cu_local[1:])), "Lq": Lq, "steps": steps}


class RingAttention:
    """Sequence-parallel attention whose K / V blocks travel around the ranks of the sequence group (BASELINE.json north_star: "ring attention send/recv
    over xGMI overlapped with backward").  The reference has no such mode: the result to match is DistributedAttention's
    (multi_head_attention.py:56-135), i.e. causal attention over the whole packed sequence -- which this reproduces WITHOUT the head exchange, so the
    kv head count need not be a multiple of the group size (Ulysses' limit) and every exchange has a block's worth of products to hide under:

      forward   step 0: causal varlen attention of the own block (the product kernel);  step s = 1 .. sp-1: the block of rank j - s is here (sent on
                by rank j - 1 while that rank computed); if it holds earlier tokens of the sequence running at the block's start, the rectangle
                [those queries] x [those keys] (ie_flash_attn_fwd_x) is folded into the running fp32 result by log-sum-exp (ie_attn_merge);
      backward  the same ring once more, every block followed by its fp32 dK / dV sums: each rank adds its share (ie_flash_attn_bwd_x with the MERGED
                lse / out: the block's additive share of the gradient), and after sp hops a block's sums are home.

    Blocks behind the own one hold nothing a causal rank sees: ranks late in the sequence compute more rectangles than early ones (a zig-zag token
    order would balance that; the contiguous order is the reference's data layout, modules/embedding.py:52-60, and is kept).  All buffers are
    allocated once.  `be.exchange` = one batch of send + receive per hop (RCCL: on c10d's stream, `wait()` orders the compute stream behind it;
    staged gloo in the tests)."""

    def __init__(self, seqpar, hq, hkv, d, Tl, device, softmax_scale=None):
        import torch

        self.sp, self.j, self.be, self.ranks = seqpar.sp, seqpar.sp_rank, seqpar.be, seqpar.ranks
        self.hq, self.hkv, self.d, self.Tl, self.dev, self.scale = hq, hkv, d, Tl, device, softmax_scale
        bf = dict(dtype=torch.bfloat16, device=device)
        f32 = dict(dtype=torch.float32, device=device)
        self.kv_ring = [torch.empty(Tl, 2, hkv, d, **bf) for _ in range(2)]
        self.out_p = torch.empty(Tl, hq, d, **bf)
        self.lse_p = torch.empty(hq * Tl, **f32)
        self.acc = torch.empty(Tl, hq, d, **f32)
        self.lse_acc = torch.empty(hq * Tl, **f32)
        self.dq_p = torch.empty(Tl * hq * d, **bf)
        self.dkv_p = torch.empty(Tl, 2, hkv, d, **bf)
        self.dq_acc = torch.empty(Tl * hq * d, **f32)
        self.dkv_acc = [torch.empty(Tl, 2, hkv, d, **f32) for _ in range(3)]
        self._plans = {}

    def plan(self, cu_host):
        """ring_plan of a micro-batch's boundaries + the device tensors the kernels read (cached per boundary set)."""
        import torch

        key = tuple(int(c) for c in cu_host)
        pl = self._plans.get(key)
        if pl is None:
            if len(self._plans) > 64:
                self._plans.clear()
            pl = ring_plan(key, self.sp, self.j)
            if pl["Tl"] != self.Tl:
                raise ValueError(f"ring attention: {key[-1]} tokens over {self.sp} ranks are not the {self.Tl} rows the buffers hold")
            i32 = lambda v: torch.tensor(v, dtype=torch.int32).to(self.dev, non_blocking=True)  # noqa: E731
            pl["cu_local_dev"] = i32(pl["cu_local"])
            pl["cu_q_dev"] = i32([0, pl["Lq"]])
            pl["cu_k_dev"] = [i32([koff, koff + Lk]) if Lk else None for _, koff, Lk in pl["steps"]]
            self._plans[key] = pl
        return pl

    def _hop(self, send, recv):
        nxt, prv = self.ranks[(self.j + 1) % self.sp], self.ranks[(self.j - 1) % self.sp]
        return self.be.exchange([(send, nxt)], [(recv, prv)])

    def forward(self, q, kv, pl, out, lse):
        """q [Tl, hq, d], kv [Tl, 2, hkv, d] (this rank's tokens, all heads) -> out [Tl, hq, d] bf16, lse [hq, Tl] fp32 (of the whole rows)."""
        Tl, hq, d, Lq = self.Tl, self.hq, self.d, pl["Lq"]
        # hop 1 leaves BEFORE the own block's kernel is launched (the exchange is ordered behind what the stream held when it was issued, not behind what
        # follows): it travels under the own block; hop s + 1 is issued the moment block s is here, and travels under block s's rectangle
        w = self._hop(kv, self.kv_ring[1]) if self.sp > 1 else None
        K.flash_attn_fwd(q, kv[:, 0], kv[:, 1], pl["cu_local_dev"], pl["max_local"], self.scale, True, out, lse)
        merged = False
        for s in range(1, self.sp):
            w.wait()
            held = self.kv_ring[s % 2]                        # the block of rank j - s
            if s + 1 < self.sp:                               # (into the buffer of block s - 1, whose rectangle was launched in the previous turn)
                w = self._hop(held, self.kv_ring[(s + 1) % 2])
            if pl["steps"][s - 1][2]:
                merged = self._fold(q, held, pl, s - 1, out, lse, merged)
        if merged:
            K.cast(self.acc[:Lq], out.dtype, out[:Lq])
            lse[:, :Lq].copy_(self.lse_acc[: hq * Lq].view(hq, Lq))
        return out, lse

    def _fold(self, q, blk, pl, i, out, lse, merged):
        hq, Lq = self.hq, pl["Lq"]
        if not merged:                                        # the own block's result of the spanning sequence's rows, in fp32
            K.cast(out[:Lq], self.acc.dtype, self.acc[:Lq])
            self.lse_acc[: hq * Lq].view(hq, Lq).copy_(lse[:, :Lq])
        lse_p = self.lse_p[: hq * Lq].view(hq, Lq)
        K.flash_attn_fwd_x(q[:Lq], blk[:, 0], blk[:, 1], pl["cu_q_dev"], pl["cu_k_dev"][i], Lq, self.scale, self.out_p[:Lq], lse_p)
        K.attn_merge(self.acc[:Lq], self.lse_acc[: hq * Lq].view(hq, Lq), self.out_p[:Lq], lse_p, Lq)
        return True

    def backward(self, dout, q, kv, out, lse, pl, dq, dkv, delta_ws=None):
        """dout, q, out [Tl, hq, d]; kv [Tl, 2, hkv, d]; lse [hq, Tl] (forward's results) -> dq [Tl, hq, d], dkv [Tl, 2, hkv, d] (bf16, overwritten)."""
        Tl, hq, hkv, d, Lq, sp = self.Tl, self.hq, self.hkv, self.d, pl["Lq"], self.sp
        # Two rings: the K / V blocks (as forward: hop 1 leaves before the own block's kernels are launched, hop s + 1 the moment block s is here) and, one
        # rectangle behind them, the blocks' fp32 dK / dV sums -- a block's sums can only move on once this rank has added its share, so hop s + 1 of the sums
        # is issued behind rectangle s's accumulation and waited for only where rectangle s + 1's accumulation needs it: both exchanges run under rectangles.
        wk = self._hop(kv, self.kv_ring[1]) if sp > 1 else None
        K.flash_attn_bwd(dout, q, kv[:, 0], kv[:, 1], out, lse, pl["cu_local_dev"], pl["max_local"], self.scale, True, dq, dkv[:, 0], dkv[:, 1], delta_ws)
        if sp == 1:
            return dq, dkv
        home = self.dkv_acc[2]
        K.cast(dkv, home.dtype, home)                          # the own block's sums start with the own queries' share
        wa = self._hop(home, self.dkv_acc[1])
        any_rect = any(Lk for _, _, Lk in pl["steps"])
        if any_rect:
            dq_acc = self.dq_acc[: Lq * hq * d]
            K.cast(dq[:Lq].reshape(-1), dq_acc.dtype, dq_acc)
            lse_sub = lse[:, :Lq].contiguous()
            dq_p = self.dq_p[: Lq * hq * d].view(Lq, hq, d)
        for s in range(1, sp):
            wk.wait()
            held_kv = self.kv_ring[s % 2]
            if s + 1 < sp:
                wk = self._hop(held_kv, self.kv_ring[(s + 1) % 2])
            _, koff, Lk = pl["steps"][s - 1]
            if Lk:
                K.flash_attn_bwd_x(dout[:Lq], q[:Lq], held_kv[:, 0], held_kv[:, 1], out[:Lq], lse_sub, pl["cu_q_dev"], pl["cu_k_dev"][s - 1], Lq, Lk, self.scale,
                                   dq_p, self.dkv_p[:, 0], self.dkv_p[:, 1], delta_ws)
                K.acc_bf16(dq_acc, dq_p.reshape(-1))
            wa.wait()                                          # block s's sums, with the shares of the ranks they have passed
            held_acc = self.dkv_acc[s % 2]
            if Lk:
                K.acc_bf16(held_acc[koff:].reshape(-1), self.dkv_p[koff:].reshape(-1))
            wa = self._hop(held_acc, self.dkv_acc[(s + 1) % 2] if s + 1 < sp else home)   # (the last hop brings every block's sums home)
        wa.wait()
        K.cast(home, dkv.dtype, dkv)
        if any_rect:
            K.cast(dq_acc, dq.dtype, dq[:Lq].reshape(-1))
        return dq, dkv
