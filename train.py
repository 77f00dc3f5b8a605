"""Training entry of the MI355X path, same invocation as the reference's train.py:

    python train.py --config configs/7B_internlm2.py --launcher torch
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 train.py --config ... --launcher torch

It reads an unmodified InternEvo config file (internevo_amd/config.py), runs the hot loop of the reference's train.py:196-306 on the
HIP engine -- load batch, forward / backward over the micro-batches, optimizer step, metric, one log line per step with the
reference's keys (internevo_amd/trainlog.py), checkpoints in the reference's format every `ckpt.checkpoint_every` steps -- and
resumes from `ckpt.load_ckpt_info` / `ckpt.load_ckpt_folder` (any data-parallel and tensor-parallel layout).  `data.train_folder` /
`data.valid_folder` may name tokenized `.bin` folders (internevo_amd/data.py); None = the reference's RandomDataset.  Out of scope
(control plane): tensorboard, alerts, remote storage backends.
"""
import argparse
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (dmabuf IPC only on these hosts: RCCL peer mappings need it; before the HIP runtime loads)
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, required=True, help="path to an InternEvo config file")
    ap.add_argument("--launcher", type=str, default="torch", choices=["torch"], help="torch (torchrun env vars; a single process without them)")
    ap.add_argument("--port", type=int, default=8888)  # accepted for command-line compatibility (the rendezvous comes from the environment)
    ap.add_argument("--seed", type=int, default=1024)
    ap.add_argument("--profiling", action="store_true")
    return ap.parse_args(argv)


def _local(path):
    if path is None:
        return None
    if ":" in path:
        backend, p = path.split(":", 1)
        if backend != "local":
            raise NotImplementedError(f"checkpoint backend {backend!r}: only local: folders")
        return p
    return path


def resolve_load(ck, log=lambda m: None):
    """What a run starts from, as CheckpointManager.__init__ decides it (checkpoint_manager.py:296-305, initialize/legacy/launch.py:10-41):
    `auto_resume` (default TRUE -- also when the key is absent and `load_given_ckpt` is not set) overrides `load_ckpt_info` with the LATEST complete checkpoint
    under `save_ckpt_folder` (the folder holding the largest `{step}.step` flag), content "all" -- and with nothing when there is none or when
    `enable_save_ckpt` is off (launch.py then drops the folder): a new run;
    otherwise `load_ckpt_info` = dict(path, content, ckpt_type), or the legacy `load_ckpt_folder` / `load_model_only_folder` keys.
    -> (local folder or None, model_only)."""
    from internevo_amd.checkpoint import latest_checkpoint

    auto = ck.get("auto_resume", None)
    if auto is None:
        auto = not ck["load_given_ckpt"] if ck.get("load_given_ckpt", None) is not None else True
    if auto:
        # (initialize/launch.py:189-225: enable_save_ckpt defaults to True, and with saving disabled save_ckpt_folder is overwritten with None -- the
        # latest-checkpoint query then finds nothing and auto_resume starts a NEW run, whatever an old folder of that name holds)
        saving = bool(ck.get("enable_save_ckpt", True))
        folder, step = latest_checkpoint(_local(ck.get("save_ckpt_folder")) if saving else None)
        log(f"Found latest ckpt {folder if folder else 'None'}, step: {step if folder else -1}...")
        return folder, False
    info = ck.get("load_ckpt_info", None)
    if info is None:   # legacy keys
        if ck.get("load_model_only_folder", None) is not None:
            info = dict(path=ck["load_model_only_folder"], content=("model",), ckpt_type="internlm")
        elif isinstance(ck.get("load_ckpt_folder", None), str):
            info = dict(path=ck["load_ckpt_folder"], content=("model", "sampler", "optimizer"), ckpt_type="internlm")
    if not info or info.get("path") is None:
        return None, False
    if info.get("ckpt_type", "internevo") not in ("internevo", "internlm", "normal"):
        raise NotImplementedError(f"load_ckpt_info.ckpt_type {info.get('ckpt_type')!r}: InternEvo checkpoints only")
    content = tuple(info.get("content", ("all",)))
    folder = _local(info["path"])
    if not os.path.isdir(folder):
        raise FileNotFoundError(f"ckpt.load_ckpt_info.path: {folder} does not exist (auto_resume is off: refusing to start from scratch instead)")
    return folder, "all" not in content and "optimizer" not in content


def evaluate_on_val_dls(eng, val_loaders, step_count, dev, log):
    """eval/evaluation.py:45-147: a forward-only pass over every validation set with a fresh AccPerplex (no dataset types); the
    logging rank reports its own mean batch loss (divided by batches + 1e-6, as the reference does) and the all-reduced accuracy /
    perplexity."""
    from internevo_amd.metrics import AccPerplex

    infos_all = {}
    # (pipeline parallelism: only the last stage sees logits; the metric is summed over the whole job, the other stages adding zeros -- as the training metric)
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    metric = AccPerplex(dev, None if eng.pp > 1 else eng.tpar.dp_group, None, dp_world_size=world if eng.pp > 1 else eng.seqpar.data_world)
    for name, vl in val_loaders.items():
        if len(vl) == 0:
            log(f"Validation dataset: {name} is empty")
            continue
        total = torch.zeros(1, dtype=torch.float32, device=dev)
        n = 0
        for batch, labels in vl:
            total += eng.forward_only(batch["input_ids"], labels, metric)
            n += 1
        res = metric.get_metric()  # all-reduces over the data-parallel group and resets
        infos = {"step": step_count, f"val/{name}_loss": float(total) / (n + 1e-6), f"val/{name}_acc": res["acc"], f"val/{name}_plex": res["perplexity"]}
        log(f"Validation on {name}: " + " ".join(f"{k}={v}" for k, v in infos.items()))
        infos_all.update(infos)
    return infos_all


def _train_moe(cfg, raw, dev, world, rank, args, log):
    """The hot loop for model_type INTERNLM_MoE (configs/7B_MoE4_sft.py) on internevo_amd.moe_engine.MoEEngine: synthetic RandomDataset batches,
    forward / backward with the moe loss, the three-group optimizer step, one log line per step with the reference's loss / moe_loss /
    per-group grad_norm keys (train/pipeline.py:494-530); InternEvo checkpoints (model + expert + optimizer files and the run state: scheduler, sampler,
    context) at any data-parallel and tensor-parallel size.  Validation and tokenized folders are the InternLM2 engine's: refused.  (The dense INTERNLM model runs on
    engine.InternLM2Engine like every other dense family.)"""
    from internevo_amd.data import BatchSkipper, SyntheticLoader
    from internevo_amd.moe_engine import MoEEngine

    data_raw, ck = raw.get("data", {}) or {}, raw.get("ckpt", {}) or {}
    load_folder, model_only = resolve_load(ck, log if rank == 0 else (lambda m: None))
    save_folder = _local(ck.get("save_ckpt_folder")) if ck.get("enable_save_ckpt", True) else None   # (launch.py:189-190: the key defaults to True)
    if model_only:
        raise NotImplementedError("INTERNLM_MoE: load_ckpt_info content = ('model',) (weights-only loads are the dense engine's)")
    if data_raw.get("train_folder") or int(data_raw.get("valid_every", 0) or 0) > 0:
        raise NotImplementedError("INTERNLM_MoE runs: set data.train_folder=None and data.valid_every=0 (validation / tokenized folders are implemented for "
                                  "the dense engine)")
    tc = cfg.train
    eng = MoEEngine(cfg, dev, None, world, rank, seed=args.seed)
    eng.sync_replicas()   # sync_model_param (utils/parallel.py:71-107)
    # (the ranks of a tensor group read the same micro-batches: the data stream is split over the data-parallel ranks)
    loader_obj = SyntheticLoader(tc.seq_len, tc.micro_bsz, tc.micro_num, tc.fixed_random_dataset_seqlen, data_rank=eng.tpar.dp_rank, data_world_size=eng.dp_world)
    first_step, run_state = 0, None
    if load_folder:   # model + optimizer files of the reference / of save_checkpoint
        from internevo_amd.checkpoint import load_run_state

        eng.load_checkpoint(load_folder)
        # the run state as the reference writes it (schedulder.pt / sampler.pt / context.pt).  The batch index comes from context.pt's batch_count, NOT
        # from the count of successful optimizer steps: after a skipped (overflowed) step the two differ, and the data stream must move on exactly as an
        # uninterrupted run's does (TrainState.load_state_dict, core/trainer.py:114-117)
        run_state = load_run_state(load_folder)
        ctx = run_state["context"]
        first_step = ctx["batch_count"] + 1 if ctx else eng.step_count
        if run_state["scheduler"] is not None:
            eng.lr_sched.load_state_dict(run_state["scheduler"])
        log(f"load_ckpt_folder: {load_folder} (resuming at batch {first_step}, step_count {eng.step_count})")
    loader = iter(loader_obj)
    if run_state and run_state["sampler"] is not None:
        loader_obj.sampler.load_state_dict(run_state["sampler"])
    else:
        for _ in range(first_step):
            next(loader)
    ctx = run_state["context"] if run_state else None
    consumed = ctx["num_consumed_tokens"] if ctx else 0
    skipped_before = ctx["inf_nan_skip_batches"] if ctx else 0
    every = int(ck.get("checkpoint_every", 0) or 0)
    out = []
    skipper = BatchSkipper(tc.skip_batches)
    for step in range(first_step, tc.total_steps):
        start = time.time()
        batch, labels = next(loader)
        if skipper(step):   # data.skip_batches (train.py:208-212): the batch is drawn -- sampler and consumed samples move on -- and not trained on
            log(f"Skip batch count:`{step}`...")
            continue
        loss, moe_loss = eng.forward_backward(batch, labels)
        eng.step()
        st = eng.read_state()
        infos = dict(step=step, loss=float(loss), moe_loss=float(moe_loss), grad_norm=dict(st.group_norms), loss_scale=st.loss_scale, lr=eng.lr_sched.lr(),
                     tgs=round(labels.nelement() * eng.dp_world / world / (time.time() - start), 2), inf_nan_skip_batches=skipped_before + st.skipped_total)
        if not st.skip:
            consumed += labels.nelement() * eng.dp_world
        out.append(infos)
        if rank % 8 == 0:
            if st.skip:
                log(f"Warning: skip parameter update at step {step}.")
            log(" ".join(f"{k}={v}" for k, v in infos.items()))
        if save_folder and every > 0 and ((step + 1) % every == 0 or step + 1 == tc.total_steps):
            eng.save_checkpoint(os.path.join(save_folder, str(step + 1)))   # collective: rank r writes the reference's ZeRO partition r
            if rank == 0:
                from internevo_amd.checkpoint import save_run_state

                save_run_state(os.path.join(save_folder, str(step + 1)), eng.lr_sched.state_dict(), loader_obj.sampler.state_dict(), batch_count=step,
                               num_consumed_samples_in_epoch=loader_obj.sampler.consumed, num_consumed_tokens=consumed,
                               inf_nan_skip_batches=skipped_before + st.skipped_total, step_count=st.adam_step)
                log(f"Saving checkpoint to `{os.path.join(save_folder, str(step + 1))}` at batch count:{step + 1}")
    if world > 1:
        torch.distributed.barrier()
    return out


def main(argv=None, log=print):
    args = parse_args(argv)
    from internevo_amd.config import from_reference_dict, run_reference_config
    from internevo_amd.data import BatchSkipper, SyntheticLoader
    from internevo_amd.engine import InternLM2Engine
    from internevo_amd.metrics import AccPerplex
    from internevo_amd.trainlog import TgsStatistic, get_megatron_flops, line, step_infos

    raw = run_reference_config(args.config)   # (also configs that open with `with read_base(): from configs._base_... import *`)
    cfg = from_reference_dict(raw)
    tc, mc = cfg.train, cfg.model
    data_raw = raw.get("data", {}) or {}
    train_folder = _local(data_raw.get("train_folder", None))
    into_one = bool(data_raw.get("pack_sample_into_one", False))
    if into_one and not train_folder:
        raise NotImplementedError("data.pack_sample_into_one needs a tokenized train_folder (the reference's RandomDataset has no type_id for it)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 and not torch.distributed.is_initialized():
        torch.distributed.init_process_group("nccl", device_id=dev)  # RCCL
    if mc.model_type == "INTERNLM_MoE":   # (the dense InternLM-1 model, model_type INTERNLM, is a block variant of the engine below)
        return _train_moe(cfg, raw, dev, world, rank, args, log)
    eng = InternLM2Engine(cfg, dev, None, world, rank, seed=args.seed)
    ck = raw.get("ckpt", {}) or {}
    load_folder, model_only = resolve_load(ck, log if rank == 0 else (lambda m: None))
    first_step, run_state = 0, None
    if load_folder and model_only:   # content = ("model",): the weights of another run; step count, optimizer, schedule and data stream start fresh
        eng.load_checkpoint(load_folder, model_only=True)
        log(f"load_ckpt_info: model weights from {load_folder}")
    elif load_folder:
        from internevo_amd.checkpoint import load_run_state

        eng.load_checkpoint(load_folder)
        run_state = load_run_state(load_folder)  # schedulder.pt / sampler.pt / context.pt when the folder has them
        ctx = run_state["context"]
        # TrainState.load_state_dict (core/trainer.py:114-117): the loop restarts one batch after the last one run
        first_step = ctx["batch_count"] + 1 if ctx else eng.step_count
        if run_state["scheduler"] is not None:
            eng.lr_sched.load_state_dict(run_state["scheduler"])
        log(f"load_ckpt_folder: {load_folder} (resuming at batch {first_step}, step_count {eng.step_count})")
    elif world > 1:
        eng.comm.broadcast_params(eng.params)  # sync_model_param (utils/parallel.py:71-107)
        eng.sync_master_from_params()
    dp_world = eng.seqpar.data_world
    if train_folder:  # tokenized .bin / .meta files (build_dataloader.py:41-50); the metric's types are its sub-folders
        from internevo_amd.data import FolderLoader

        loader_obj = FolderLoader(train_folder, tc.seq_len, tc.micro_bsz, tc.micro_num, data_raw.get("min_length", 0), data_raw.get("min_length_dict", None),
                                  data_rank=eng.seqpar.data_rank, data_world_size=dp_world, seed=data_raw.get("seed", 1024), pack_sample_into_one=into_one)
        dataset_types = loader_obj.dataset_types
    else:
        loader_obj = SyntheticLoader(tc.seq_len, tc.micro_bsz, tc.micro_num, tc.fixed_random_dataset_seqlen,
                                     data_rank=eng.seqpar.data_rank, data_world_size=dp_world)
        dataset_types = ["en", "cn", "code"]  # the dummy dataset's type list (build_dataloader.py:93)
    # pipeline parallelism: only the last stage sees logits; the others hold zero accumulators and the packed all-reduce runs over the
    # whole job, so every rank (the logging rank 0 included) reports the job's metric
    # (pipeline parallelism: summed over the whole job -- only the last stage sees logits, the others add zeros; tensor ranks inside a stage count the same
    # tokens, which leaves the accuracy and perplexity RATIOS unchanged)
    metric = AccPerplex(dev, None if eng.pp > 1 else eng.tpar.dp_group, dataset_types, dp_world_size=world if eng.pp > 1 else dp_world)
    eng.attach_metric(metric)
    loader = iter(loader_obj)
    if run_state and run_state["sampler"] is not None:
        loader_obj.sampler.load_state_dict(run_state["sampler"])  # generator state + position: the same batches as an uninterrupted run
    else:
        for _ in range(first_step):
            next(loader)  # no sampler file: replay the position
    flops = lambda t: get_megatron_flops(t, checkpoint=bool(mc.checkpoint_layers), seq_len=tc.seq_len, hidden_size=mc.hidden_size,  # noqa: E731
                                         num_layers=mc.num_layers, vocab_size=mc.vocab_size, global_batch_size=tc.micro_bsz * tc.micro_num * dp_world,
                                         global_world_size=world, mlp_ratio=mc.mlp_ratio)
    tgs = TgsStatistic()
    # validation sets (build_dataloader.py:67-157): the default 500-document RandomDataset per data-parallel rank, or valid_folder
    valid_every = int(data_raw.get("valid_every", 0) or 0)
    val_loaders = {}
    if valid_every > 0:
        from internevo_amd.data import ValidLoader, valid_datasets

        for name, ds in valid_datasets(tc.seq_len, tc.fixed_random_dataset_seqlen, dp_world, _local(data_raw.get("valid_folder", None))).items():
            vl = ValidLoader(ds, tc.seq_len, tc.micro_bsz, int(data_raw.get("valid_micro_num", 1)), eng.seqpar.data_rank, dp_world)
            if vl.batch_size == 0:
                log(f"skip validate {name}.")
                continue
            val_loaders[name] = vl
    save_folder = _local(ck.get("save_ckpt_folder")) if ck.get("enable_save_ckpt", True) else None   # (launch.py:189-190: the key defaults to True)
    if save_folder:
        # say so at start-up instead of training until the first checkpoint_every step and dying there
        try:
            eng._checkpoint_guard()
        except NotImplementedError as e:
            raise NotImplementedError(f"ckpt.enable_save_ckpt: {e}; set enable_save_ckpt=False for such a run") from None
    if val_loaders and eng.pp > 1 and eng.nch > 1:
        raise NotImplementedError("data.valid_every > 0 with the interleaved pipeline schedule (model.num_chunks > 1): the forward-only pass walks the stages "
                                  "once per micro-batch; set valid_every=0 or num_chunks=1")
    every = int(ck.get("checkpoint_every", 0) or 0)
    ctx = run_state["context"] if run_state else None
    consumed = ctx["num_consumed_tokens"] if ctx else 0
    skipped_before = ctx["inf_nan_skip_batches"] if ctx else 0
    out = []
    skipper = BatchSkipper(tc.skip_batches)
    for step in range(first_step, tc.total_steps):
        start = time.time()
        batch, labels = next(loader)
        if skipper(step):   # data.skip_batches (train.py:208-212): the batch is drawn -- sampler and consumed samples move on -- and not trained on
            log(f"Skip batch count:`{step}`...")
            continue
        t0 = time.time()
        loss = eng.forward_backward(batch, labels)
        eng.step()
        st = eng.read_state()                       # the reference syncs here too (loss.item(), the norm and the scaler)
        fwd_bwd_time = time.time() - t0
        success = st.skip == 0
        if success:
            consumed += labels.nelement() * dp_world
        else:
            log(f"Warning: skip parameter update at step {step}.")
        m = metric.get_metric()
        tk_per_gpu = round(labels.nelement() * dp_world / world, 4)
        infos = step_infos(tflops=flops(time.time() - start), step=step, loss=float(loss), tk_per_gpu=tk_per_gpu, start_time=start, tgs=tgs,
                           lr=eng.lr_sched.lr(),  # read after the scheduler stepped, like optimizer.param_groups[0]["lr"] (pipeline.py:494)
                           loss_scale=st.loss_scale, grad_norm=getattr(st, "group_norms", None) or {"0_default": st.grad_norm}, batch=batch, labels=labels,
                           num_consumed_tokens=consumed, inf_nan_skip_batches=skipped_before + st.skipped_total, adam_beta2=eng.beta2_sched.beta2(),
                           fwd_bwd_time=fwd_bwd_time, metric=m)
        out.append(infos)
        if rank % 8 == 0 and success:
            log(line(infos))
        if val_loaders and st.adam_step % valid_every == 0:  # train.py:279-288: keyed on the count of successful steps
            out[-1].update(evaluate_on_val_dls(eng, val_loaders, st.adam_step, dev, log if rank == 0 else (lambda m: None)))
        if save_folder and every and (step + 1) % every == 0:
            eng.save_checkpoint(os.path.join(save_folder, str(step + 1)))  # collective: every data-parallel rank writes its ZeRO shard
            if rank == 0:
                from internevo_amd.checkpoint import save_run_state

                save_run_state(os.path.join(save_folder, str(step + 1)), eng.lr_sched.state_dict(), loader_obj.sampler.state_dict(), batch_count=step,
                               num_consumed_samples_in_epoch=loader_obj.sampler.consumed, num_consumed_tokens=consumed,
                               inf_nan_skip_batches=skipped_before + st.skipped_total, step_count=st.adam_step)
                log(f"Saving checkpoint to `{os.path.join(save_folder, str(step + 1))}` at batch count:{step + 1}")
    if world > 1:
        torch.distributed.barrier()
    return out


if __name__ == "__main__":
    main()
