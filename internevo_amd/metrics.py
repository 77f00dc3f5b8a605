"""Training metrics of the hot loop, fused into the cross-entropy kernel.

Host-side mirror of the reference's `AccPerplex` + `LossWithTypeId` (internlm/model/metrics.py:56-339): same
constructor meaning (device, data-parallel group, dataset_types), same accumulators with the same dtypes, same
`get_metric(reset)` keys and rounding.  What changes is where the work happens: the reference's update() re-reads the
[T, vocab] fp32 logits (1.5 GB at 7B) for max, argmax, exp-sum and one more CE forward per micro-batch and issues 6+ small
all-reduces; here the CE forward kernel emits the row argmax and plain NLL while it streams the logits anyway, one
single-block kernel folds them into device-resident accumulators, and get_metric() does ONE all-reduce of the packed
accumulators over the data-parallel group.
"""
import torch
import torch.distributed as dist

from . import kernels as K


class AccPerplex:
    def __init__(self, device, dp_pg=None, dataset_types=None, tokenizer=None, dp_world_size=1):
        self.device = device
        self.dp_pg = dp_pg
        self.dp_world = dp_world_size
        self.dataset_types = dataset_types
        self.tokenizer = tokenizer
        n = len(dataset_types) if dataset_types is not None else 0
        self.ntypes = n
        # packed so that get_metric() needs a single fp32 and a single int64 all-reduce
        self._f = torch.zeros(6 + 2 * n, dtype=torch.float32, device=device)   # right,total,total_log_probs,loss,token_num,total_bytes | ds_loss | ds_token_num
        self._i = torch.zeros(max(2 * n, 1), dtype=torch.int64, device=device)  # ds_right | ds_tokens
        self.right, self.total, self.total_log_probs = self._f[0:1], self._f[1:2], self._f[2:3]
        self.loss, self.token_num, self.total_bytes = self._f[3:4], self._f[4:5], self._f[5:6]
        self.ds_loss, self.ds_token_num = self._f[6 : 6 + n], self._f[6 + n : 6 + 2 * n]
        self.ds_right, self.ds_tokens = self._i[0:n], self._i[n : 2 * n]
        self.type_ids = None
        self.type_ids_local = None  # (lo, hi): this rank's token range of a micro-batch under sequence parallelism
        self.batch_shift = 0

    def set_current_type_ids(self, type_ids):
        """metrics.py:97-99: type ids of the whole batch [micro_num, T]; update() consumes one micro-batch row at a time."""
        self.batch_shift = 0
        self.type_ids = type_ids.to(self.device, non_blocking=True)

    def update_fused(self, nll_rows, argmax_rows, labels, host_labels=None):
        """Fold one micro-batch (per-row outputs of K.ce_fwd(..., argmax_rows, nll_rows)) into the accumulators."""
        tid = None
        if self.ntypes:
            if self.type_ids is None:
                raise RuntimeError("dataset_types given but set_current_type_ids() was not called for this batch")
            tid = self.type_ids[self.batch_shift].reshape(-1)
            if self.type_ids_local is not None:
                tid = tid[self.type_ids_local[0] : self.type_ids_local[1]].contiguous()
            self.batch_shift += 1
        if self.tokenizer is not None:  # bits per byte needs the decoded byte count (metrics.py:128-130), host work
            ids = (host_labels if host_labels is not None else labels.cpu()).reshape(1, -1).tolist()
            self.total_bytes += sum(len(x.encode("utf-8")) for x in self.tokenizer.decode_ids(ids))
        if self.ntypes:
            K.metric_accumulate(nll_rows, argmax_rows, labels, tid, self._f, self.ds_right, self.ds_tokens, self.ds_loss, self.ds_token_num)
        else:
            K.metric_accumulate(nll_rows, argmax_rows, labels, None, self._f)

    def get_metric(self, reset=True):
        """metrics.py:201-247 + :312-339 (keys, order, rounding identical)."""
        if self.dp_pg is not None or (self.dp_world > 1 and dist.is_initialized()):
            from .comm import backend_for

            be = backend_for(self.dp_pg)
            works = [be.all_reduce(t, self.dp_pg) for t in (self._f, self._i)]
            for w in works:
                w.wait()
        f, i = self._f.cpu(), self._i.cpu()
        n = self.ntypes
        right, total, tlp, loss, token_num, total_bytes = (f[k : k + 1] for k in range(6))
        res = {"acc": round((right / total).item(), 4), "perplexity": round(torch.exp(tlp / total).item(), 4)}
        if self.tokenizer is not None:
            res["BPB"] = round((tlp / total_bytes).item(), 4)
        for k in range(n):
            res[f"acc/{self.dataset_types[k]}"] = round((i[k].float() / (i[n + k].float() + 1e-5)).item(), 4)
        for k in range(n):
            res[f"tokens/{self.dataset_types[k]}"] = i[n + k].item()
        res["loss_from_metric"] = round((loss / token_num).item(), 4)
        for k in range(n):
            res[f"loss/{self.dataset_types[k]}"] = round((f[6 + k] / f[6 + n + k]).item(), 4)
        if reset:
            self._f.zero_()
            self._i.zero_()
        return res
