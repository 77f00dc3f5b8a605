"""Drop-in boundary #2: the native-op import sites of InternEvo, served by libinternevo_hip.so.

    import internevo_amd.plugin as plugin; plugin.install()      # before `import internlm`

registers modules under the exact names the reference imports, with the signatures it calls
(call sites are cited in each shim):

    rotary_emb.apply_rotary                                   <- internlm/model/modules/embedding.py:115-120
    fused_dense_lib.linear_bias_wgrad                         <- internlm/model/utils.py:293-299
    flash_attn.flash_attn_varlen_kvpacked_func                <- internlm/model/modeling_internlm2.py:157,446-468
    flash_attn.flash_attn_interface.FlashAttnVarlenKVPackedFunc <- internlm/model/modeling_llama.py:297
    flash_attn.modules.mha.FlashSelfAttention / FlashCrossAttention <- modeling_internlm2.py:158; multi_head_attention.py:382
    flash_attn.losses.cross_entropy.CrossEntropyLoss          <- internlm/model/losses/ce_loss.py:26-36
    apex.normalization.fused_layer_norm.MixedFusedRMSNorm     <- internlm/model/utils.py:662-675
    amp_C.multi_tensor_l2norm + apex.multi_tensor_apply.multi_tensor_applier <- internlm/solver/optimizer/utils.py:30-37,191-204
    torch_scatter.scatter                                     <- internlm/model/metrics.py:12-15

With these in place `use_flash_attn=True` takes the native route in the unmodified reference; the accelerator
plugin (boundary #1) is the reference's own CUDA_Accelerator, because on PyTorch-ROCm `torch.cuda` IS the HIP
runtime and the "nccl" backend IS RCCL.  Nothing here falls back to eager PyTorch math:
every op either launches a HIP kernel through the C ABI or raises.
"""
import sys
import types

from . import apex_shims, flash_attn_shims, misc_shims


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__internevo_amd__ = True
    return m


def install(force=False):
    """Register the shim modules in sys.modules (idempotent).  Existing real packages are left alone unless force."""
    fa = flash_attn_shims
    mods = {
        "rotary_emb": _mod("rotary_emb", apply_rotary=misc_shims.apply_rotary),
        "fused_dense_lib": _mod("fused_dense_lib", linear_bias_wgrad=misc_shims.linear_bias_wgrad),
        "flash_attn": _mod("flash_attn", flash_attn_varlen_kvpacked_func=fa.flash_attn_varlen_kvpacked_func,
                           flash_attn_varlen_qkvpacked_func=fa.flash_attn_varlen_qkvpacked_func, __version__="2.2.1+internevo_amd", __path__=[]),
        "flash_attn.flash_attn_interface": _mod("flash_attn.flash_attn_interface", FlashAttnVarlenKVPackedFunc=fa.FlashAttnVarlenKVPackedFunc,
                                                flash_attn_varlen_kvpacked_func=fa.flash_attn_varlen_kvpacked_func,
                                                flash_attn_varlen_qkvpacked_func=fa.flash_attn_varlen_qkvpacked_func),
        "flash_attn.modules": _mod("flash_attn.modules", __path__=[]),
        "flash_attn.modules.mha": _mod("flash_attn.modules.mha", FlashSelfAttention=fa.FlashSelfAttention, FlashCrossAttention=fa.FlashCrossAttention),
        "flash_attn.modules.mlp": _mod("flash_attn.modules.mlp", ParallelFusedMLP=fa.ParallelFusedMLP),
        "flash_attn.modules.embedding": _mod("flash_attn.modules.embedding", ParallelGPT2Embeddings=fa.ParallelGPT2Embeddings,
                                             VocabParallelEmbedding=fa.VocabParallelEmbedding),
        "flash_attn.losses": _mod("flash_attn.losses", __path__=[]),
        "flash_attn.losses.cross_entropy": _mod("flash_attn.losses.cross_entropy", CrossEntropyLoss=fa.CrossEntropyLoss),
        "flash_attn.ops": _mod("flash_attn.ops", __path__=[]),
        "flash_attn.ops.layer_norm": _mod("flash_attn.ops.layer_norm", dropout_add_layer_norm=fa.dropout_add_layer_norm),
        "apex": _mod("apex", __path__=[]),
        "apex.normalization": _mod("apex.normalization", __path__=[]),
        "apex.normalization.fused_layer_norm": _mod("apex.normalization.fused_layer_norm", MixedFusedRMSNorm=apex_shims.MixedFusedRMSNorm),
        "apex.multi_tensor_apply": _mod("apex.multi_tensor_apply", multi_tensor_applier=apex_shims.multi_tensor_applier),
        "amp_C": _mod("amp_C", multi_tensor_l2norm=apex_shims.multi_tensor_l2norm),
        "torch_scatter": _mod("torch_scatter", scatter=misc_shims.scatter),
    }
    for name, m in mods.items():
        if force or name not in sys.modules:
            sys.modules[name] = m
    # parent packages expose their children as attributes, as real packages do
    for name, m in mods.items():
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, sys.modules[name])
    return sorted(mods)
