"""One-command check of INTEGRATION.md's claim: the UNMODIFIED reference training loop on an MI355X with this repo's shims installed.

    tools/run_reference_dropin.sh <path to an InternEvo checkout>      (needs a GPU box that holds BOTH trees; this pool has none)

Three steps of the tiny InternLM2 config of tests/golden/train_cfg0_bf16.json (BASELINE configs[0]'s model; the fixture is the reference's own
CPU run of the same model, data and closed-form weights) through `internevo_amd.plugin.install()`: `use_flash_attn=True`, packed data, the
reference's CUDA_Accelerator (= HIP on PyTorch-ROCm), its own NonPipelineScheduler / HybridZeroOptimizer.  Every flash_attn / rotary_emb /
fused_dense_lib / apex / amp_C call of those steps lands in libinternevo_hip.so.  Passes when loss (1e-3 relative) and gradient norm (2e-2)
of every step match the fixture, and when the shims were really the ones called.  Nothing here is imported by the product."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) != 2 or not os.path.isdir(os.path.join(sys.argv[1], "internlm")):
        raise SystemExit("usage: reference_dropin_check.py <InternEvo checkout (the folder that holds internlm/)>")
    ref = os.path.abspath(sys.argv[1])
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X: torch.cuda (HIP) is not available here")
    sys.path.insert(0, ROOT)
    import internevo_amd.plugin as mi355x

    mi355x.install()                       # the two lines a maintainer adds on top of train.py
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as H                # only for tiny_config / NUM_SAMPLES (it puts /root/reference on sys.path: the given checkout goes in front)

    sys.path.insert(0, ref)
    import internlm  # noqa: F401
    import internlm.data.build_dataloader as bdl
    from internlm.core.context import ParallelMode
    from internlm.core.context import global_context as gpc
    from internlm.core.trainer import TrainState
    from internlm.data.tokenized.dummy_dataset import RandomDataset
    from internlm.initialize.launch import args_sanity_check, launch
    from internlm.model.losses import FlashGPTLMLoss
    from internlm.model.metrics import AccPerplex
    from internlm.train import get_scheduler_hooks, initialize_isp_communicator, initialize_model, initialize_optimizer, load_new_batch
    from internlm.utils.common import get_current_device

    from oracle.model import formula_init

    assert os.path.realpath(internlm.__file__).startswith(os.path.realpath(ref)), f"internlm was imported from {internlm.__file__}, not from {ref}"
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "train_cfg0_bf16.json")))
    kw = dict(gold["config"], use_packed=True)                  # the shipped configs' route: packed data + flash attention
    cfg = H.tiny_config("torch.bfloat16", **kw)
    cfg["model"]["use_flash_attn"] = True
    bdl.RandomDataset = lambda num_samples, max_len, fixed_seqlen: RandomDataset(num_samples=gold["num_samples"], max_len=max_len, fixed_seqlen=fixed_seqlen)
    launch(config=cfg, rank=0, world_size=1, host="::1", port=29641, backend="nccl", local_rank=0, seed=1024)
    args_sanity_check()
    model = initialize_model()
    with torch.no_grad():
        for name, p in model.model.named_parameters():
            p.copy_(formula_init(name, tuple(p.shape)).to(p.dtype))
    called = {}
    import fused_dense_lib
    import rotary_emb
    from flash_attn import flash_attn_interface as fai

    for mod, fn in ((rotary_emb, "apply_rotary"), (fused_dense_lib, "linear_bias_wgrad")):
        assert getattr(mod, "__internevo_amd__", False), f"{mod.__name__} is not this repo's shim"
        orig = getattr(mod, fn)

        def wrap(*a, _o=orig, _n=f"{mod.__name__}.{fn}", **k):
            called[_n] = called.get(_n, 0) + 1
            return _o(*a, **k)

        setattr(mod, fn, wrap)
    assert getattr(fai, "__internevo_amd__", False)
    criterion = FlashGPTLMLoss(parallel_output=False, label_smoothing=0)
    train_dl, dataset_types = bdl.build_train_loader_with_data_type()
    train_state = TrainState(gpc.config, train_dl.batch_sampler)
    isp = initialize_isp_communicator(model)
    optimizer, beta2_scheduler, lr_scheduler = initialize_optimizer(model, isp)
    metric = AccPerplex(device=get_current_device(), tp_pg=gpc.get_group(ParallelMode.TENSOR), dp_pg=gpc.get_group(ParallelMode.DATA), dataset_types=dataset_types)
    trainer, train_dl, _, _ = internlm.initialize_trainer(model=model, optimizer=optimizer, criterion=criterion, train_dataloader=train_dl, lr_scheduler=lr_scheduler,
                                                          beta2_scheduler=beta2_scheduler, scheduler_hooks=get_scheduler_hooks(metric, optimizer, isp))
    trainer.train()
    train_iter = iter(train_dl)
    bad = []
    for step in range(3):
        batch, train_iter = load_new_batch(train_dl=train_dl, train_iter=train_iter, train_state=train_state)
        trainer.zero_grad()
        if batch[0].get("type_ids", None) is not None:
            metric.set_current_type_ids(type_ids=batch[0].pop("type_ids", None))
        _, _, loss = trainer.execute_schedule(batch, forward_only=False, return_loss=True, return_output_label=False)
        ok, norms = trainer.step()
        w = gold["steps"][step]
        l, n = float(loss.item()), float(norms["0_default"])
        dl, dn = abs(l - w["loss"]) / w["loss"], abs(n - w["grad_norm"]["0_default"]) / w["grad_norm"]["0_default"]
        print(f"step {step}: reference-on-MI355X loss {l:.6f} grad_norm {n:.5f} | its CPU run {w['loss']:.6f} {w['grad_norm']['0_default']:.5f} | rel {dl:.2e} {dn:.2e}", flush=True)
        if not ok or dl > 1e-3 or dn > 2e-2:
            bad.append(step)
    print("native calls through the shims:", called)
    if bad or not called.get("rotary_emb.apply_rotary") or not called.get("fused_dense_lib.linear_bias_wgrad"):
        raise SystemExit(f"DROP-IN CHECK FAILED (steps {bad}; shim calls {called})")
    print("DROP-IN CHECK PASSED: the unmodified reference trained three steps on the HIP kernels within 1e-3 / 2e-2 of its own CPU run")


if __name__ == "__main__":
    main()
