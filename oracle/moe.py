"""CPU oracle of the GShard MoE layer's routing arithmetic -- TEST INFRASTRUCTURE ONLY, prepared ahead
of the HIP path (no product code imports it; round 1 ships no MoE kernels).

Restates internlm/model/moe/gshard_layer.py in INDEX form -- what a gather / scatter implementation works with -- instead of the
reference's one-hot masks and O(S*E*C*M) einsums:

    top2gating   :217-285   softmax over experts, first expert = argmax of the gates, second = argmax of (logits + Gumbel noise) with
                            the first masked out, slot of a token in its expert's capacity buffer = its rank among the tokens that
                            chose the expert (second choices queue behind ALL first choices of that expert), tokens whose slot is
                            past the capacity are dropped, the two gate values are renormalised over what survived, the auxiliary
                            loss l_aux = mean_e(mean_s gates * mean_s first-choice mask) * E^2 uses the masks BEFORE the drop.
    _capacity    :113-122   ceil(S / E * capacity_factor [* 2 for top-2]) clamped from below by min_capacity.
    dispatch     :446-448   dispatched[e, c] = the token that holds slot c of expert e (zeros where nobody does).
    combine      :482-486   out[s] = sum over the token's surviving choices of gate weight * expert_output[e, slot].

The Gumbel noise of the second choice comes from the device RNG in the reference (:233-238): callers pass it in, so that oracle,
reference (patched `gumbel_rsample`) and a future kernel see the same numbers.  Pinned by tests/golden/moe.npz
(tests/golden/make_golden.py --moe runs the real top2gating and the reference's dispatch / combine einsums)."""
import math

import torch


def capacity(num_tokens, num_experts, capacity_factor, min_capacity):
    """gshard_layer.py:113-122 (torch: a float32 division, then ceil)."""
    c = int(torch.ceil(torch.tensor(num_tokens / num_experts) * torch.tensor(capacity_factor)).to(torch.int64))
    return max(c, int(min_capacity))


def top2gating(logits, capacity_factor, min_capacity, noise, force_expert=None):
    """logits, noise: fp32 [S, E].  -> dict with
        l_aux (fp32 scalar), capacity (int), exp_counts int64 [E] (first choices per expert, before the drop),
        expert int64 [2, S], slot int64 [2, S] (-1 = dropped), weight fp32 [2, S] (0 when dropped)."""
    logits = logits.float()
    S, E = logits.shape
    gates = torch.softmax(logits, dim=1)
    cap = capacity(S, E, capacity_factor * 2, min_capacity)
    e1 = torch.argmax(gates, dim=1)
    masked = (logits + noise).clone()
    masked[torch.arange(S), e1] = torch.finfo(logits.dtype).min
    e2 = torch.argmax(masked, dim=1)
    if force_expert is not None:
        # teacher-forced routing (tests only): take the two choices from somewhere else -- the HIP engine's -- so that everything
        # DOWNSTREAM of the discrete decision can be compared tightly; the decision itself flips between machines for tokens whose two
        # best (noisy) logits are within bf16 rounding noise, and is compared separately
        e1, e2 = force_expert[0].clone(), force_expert[1].clone()
    # rank of a token among the tokens that picked the same expert, in token order
    first_counts = torch.bincount(e1, minlength=E)
    pos1 = torch.empty(S, dtype=torch.int64)
    pos2 = torch.empty(S, dtype=torch.int64)
    seen1 = [0] * E
    seen2 = [0] * E
    for s in range(S):
        a, b = int(e1[s]), int(e2[s])
        pos1[s] = seen1[a]
        seen1[a] += 1
        pos2[s] = seen2[b] + int(first_counts[b])
        seen2[b] += 1
    me = gates.mean(dim=0)
    ce = torch.nn.functional.one_hot(e1, E).to(logits.dtype).mean(dim=0)
    l_aux = torch.mean(me * ce) * E * E
    keep1, keep2 = pos1 < cap, pos2 < cap
    g1 = torch.where(keep1, gates[torch.arange(S), e1], torch.zeros(S))
    g2 = torch.where(keep2, gates[torch.arange(S), e2], torch.zeros(S))
    denom = torch.clamp(g1 + g2, min=torch.finfo(logits.dtype).eps)
    return dict(l_aux=l_aux, capacity=cap, exp_counts=first_counts,
                expert=torch.stack([e1, e2]), slot=torch.stack([torch.where(keep1, pos1, torch.full_like(pos1, -1)),
                                                                torch.where(keep2, pos2, torch.full_like(pos2, -1))]),
                weight=torch.stack([g1 / denom, g2 / denom]))


def combine_weights_dense(r, num_experts):
    """The reference's [S, E, C] combine_weights tensor from the index form (for comparison only)."""
    S = r["expert"].shape[1]
    out = torch.zeros(S, num_experts, r["capacity"])
    for k in range(2):
        for s in range(S):
            c = int(r["slot"][k, s])
            if c >= 0:
                out[s, int(r["expert"][k, s]), c] += r["weight"][k, s]
    return out


def dispatch(x, r, num_experts):
    """[S, M] tokens -> [E, C, M] expert buffers (gshard_layer.py:446-448 `sec,sm->ecm` with the boolean dispatch mask).
    The mask is combine_weights != 0: a surviving choice whose renormalised weight is exactly 0 is NOT dispatched."""
    out = torch.zeros(num_experts, r["capacity"], x.shape[1], dtype=x.dtype)
    for k in range(2):
        for s in range(x.shape[0]):
            c = int(r["slot"][k, s])
            if c >= 0 and float(r["weight"][k, s]) != 0.0:
                out[int(r["expert"][k, s]), c] += x[s]
    return out


def combine(expert_out, r):
    """[E, C, M] expert outputs -> [S, M] (gshard_layer.py:482-486 `sec,ecm->sm`, weights cast to the activation dtype)."""
    S = r["expert"].shape[1]
    out = torch.zeros(S, expert_out.shape[2], dtype=torch.float32)
    for k in range(2):
        for s in range(S):
            c = int(r["slot"][k, s])
            if c >= 0:
                w = r["weight"][k, s].to(expert_out.dtype).float()
                out[s] += w * expert_out[int(r["expert"][k, s]), c].float()
    return out.to(expert_out.dtype)


def gumbel_noise(shape, seed):
    """Deterministic stand-in for torch.distributions.Gumbel(0, 1).rsample (gshard_layer.py:63-70), same construction:
    -log(-log(u)) with u uniform on the open interval."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(shape, generator=g).clamp_(min=torch.finfo(torch.float32).tiny, max=1.0 - math.ulp(1.0) / 2)
    return -torch.log(-torch.log(u))


# ---------------------------------------------------------------------------------------------------------------------------------
# The whole GShard MoE layer, forward AND backward (through torch autograd over the index form), with the reference's dtypes:
#   gate      TopKGate.wg is an fp32 module behind NaiveAMP's hooks (naive_amp.py:160-206): its input is cast to fp32, its outputs
#             (combine weights, l_aux) are cast back to the model dtype -- the combine weights reach the einsums rounded to bf16, and
#             their gradient comes back as a bf16 tensor;
#   dispatch  `sec,sm->ecm` with the boolean mask (taken from the fp32 weights) in bf16: a copy of the token row (or zeros);
#   experts   FeedForward (modules/mlp.py:82-86): w2(silu(w1 x) * w3 x), bf16 linears;
#   combine   `sec,ecm->sm` in bf16 (fp32 accumulation of the two surviving terms, one rounding).
# Pinned by tests/golden/moe_layer.npz = the REAL GShardMOELayer run forward and backward on CPU (make_golden.py --moe-layer).
class _RoundGrad(torch.autograd.Function):
    """identity whose gradient is rounded to the model dtype (a bf16 tensor's gradient is a bf16 tensor)"""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).to(g.dtype), None


def moe_layer(x, wg, w1, w3, w2, noise, capacity_factor, min_capacity, force_expert=None):
    """x [S, M] in the model dtype (bf16; fp32 for the fp32 runs), wg fp32 [E, M], w1 / w3 [E, F, M], w2 [E, M, F] in the model dtype,
    noise fp32 [S, E].  -> (out [S, M], l_aux scalar, both in the model dtype; routing dict).  Differentiable w.r.t. x, wg, w1, w3, w2."""
    S, M = x.shape
    E = wg.shape[0]
    dt = x.dtype
    logits = x.float() @ wg.float().t()
    with torch.no_grad():
        r = top2gating(logits.detach(), capacity_factor, min_capacity, noise, force_expert)
        r["logits"], r["noisy"] = logits.detach().clone(), (logits.detach() + noise)
    gates = torch.softmax(logits, dim=1)
    ar = torch.arange(S)
    keep = r["slot"] >= 0
    g1 = torch.where(keep[0], gates[ar, r["expert"][0]], torch.zeros(S))
    g2 = torch.where(keep[1], gates[ar, r["expert"][1]], torch.zeros(S))
    denom = torch.clamp(g1 + g2, min=torch.finfo(torch.float32).eps)
    w = torch.stack([g1 / denom, g2 / denom])                       # fp32 [2, S]
    me = gates.mean(dim=0)
    ce = torch.nn.functional.one_hot(r["expert"][0], E).float().mean(dim=0)
    l_aux = (torch.mean(me * ce) * E * E).to(dt)
    wb = _RoundGrad.apply(w, dt).to(dt)                             # what the einsums see
    C = r["capacity"]
    # dispatch: row e*C + c <- token (mask from the fp32 weights)
    sent = keep & (w.detach() != 0)
    row = r["expert"] * C + torch.clamp(r["slot"], min=0)           # [2, S]
    buf = torch.zeros(E * C, M, dtype=torch.float32)
    for k in range(2):
        idx = torch.nonzero(sent[k]).squeeze(1)
        buf = buf.index_add(0, row[k][idx], x[idx].float())          # unique rows: a copy (index_add keeps it differentiable)
    ein = buf.to(dt).reshape(E, C, M)
    outs = []
    for e in range(E):
        h1 = torch.nn.functional.linear(ein[e], w1[e])
        h3 = torch.nn.functional.linear(ein[e], w3[e])
        outs.append(torch.nn.functional.linear(torch.nn.functional.silu(h1) * h3, w2[e]))
    eo = torch.stack(outs).reshape(E * C, M)
    acc = torch.zeros(S, M, dtype=torch.float32)
    for k in range(2):
        rows = _RoundGrad.apply(eo[row[k]].float(), dt)
        acc = acc + torch.where(keep[k], wb[k].float(), torch.zeros(S))[:, None] * rows * keep[k][:, None]
    r["row"], r["sent"] = row, sent
    return acc.to(dt), l_aux, r
