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
