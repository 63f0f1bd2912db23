"""Training entrypoint with the reference's shape (train_accel_gpu.py:1-188):

    python train_accel_gpu.py <config.yaml>                       # 1 GPU
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train_accel_gpu.py <config.yaml>

Same YAML keys (utils/config.py), same loop order (forward, zero_grad, backward, clip_grad_norm_, AdamW step, cosine
schedule with warm-up, per-epoch checkpoint, final safetensors model).  What differs: the model step runs in the HIP
kernels of mca-paper_amd, the optimizer is the fused clip+AdamW, data parallelism is mca-paper_amd/dp.py (RCCL), and
logging goes to stdout / <output_dir>/log.jsonl instead of wandb.  `--synthetic STEPS` trains on synthetic batches of
the configured shapes when the HF dataset named in the YAML is not on disk.

Log records (train_accel_gpu.py:126-130,163-181): every step (`--log-every N`: every N-th; each record costs a host sync, as the
reference's `.to("cpu")` calls do) total_loss, the loss terms without '|', param_norm, grad_norm, lr; per eval batch val_step_*;
per epoch val_epoch_*.  Under data parallelism rank 0 logs ITS values, as `accelerator.log` does, the eval set is dealt to the
ranks as Accelerate's prepared loader deals it (data.shard_eval_batches) and the embedding metrics gather every rank's rows.
"""
import importlib
import json
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
P = importlib.import_module("mca-paper_amd")
optim = importlib.import_module("mca-paper_amd.optim")
dpmod = importlib.import_module("mca-paper_amd.dp")
from utils.training import get_grad_norm, get_param_norm  # noqa: E402  (the reference's import, train_accel_gpu.py:15)


def lr_factor(name, step, warmup, total):
    """Multiplier of the base LR at optimizer step `step`: transformers.get_scheduler(name, ...) as called at
    train_accel_gpu.py:81-86 (names used by the reference's YAMLs: cosine, constant_with_warmup; plus linear and constant).
    Unknown names raise, as get_scheduler does."""
    if name == "constant":
        return 1.0
    if name == "constant_with_warmup":
        return step / max(1.0, warmup) if step < warmup else 1.0
    if name == "linear":
        if step < warmup:
            return step / max(1, warmup)
        return max(0.0, (total - step) / max(1, total - warmup))
    if name == "cosine":
        if step < warmup:
            return step / max(1, warmup)
        prog = (step - warmup) / max(1, total - warmup)
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
    raise ValueError(f"{name} is not a valid SchedulerType")


def cosine_with_warmup(step, warmup, total):
    return lr_factor("cosine", step, warmup, total)


def move_to(obj, device):
    if torch.is_tensor(obj):
        return obj.to(device, non_blocking=True)
    if isinstance(obj, dict):
        return {k: move_to(v, device) for k, v in obj.items()}
    if isinstance(obj, list):
        return [move_to(v, device) for v in obj]
    raise TypeError("Invalid type for move_to")


def run_eval(model, eval_batches, model_config, device, step_log=None, sync=False):
    """The reference's evaluation loop (train_accel_gpu.py:137-181) on the native model: mean of the total loss and of every loss
    term without '|' over the eval batches, Wang-Isola uniformity per modality (+ 'fusion' unless EAO) and alignment of every
    modality with the fusion embedding (samples that have the modality), raw and normalised, with the reference's log keys
    (its 'unformity_avg' spelling included).  Returns {key: float}.
    step_log: called with the reference's per-batch record {val_step_total_loss, val_step_<term>} (:163-164).
    sync (data parallelism): `eval_batches` is this rank's share (data.shard_eval_batches) and the forward all-gathers the
    embeddings as in training, so every loss is the global-batch loss of this rank's rows; the sums stay this rank's (the
    reference logs from the main process only) while the metrics gather every rank's rows (torchmetrics' cat states)."""
    was_training = model.training
    model.eval()
    names = list(model_config["encoder_configs"].keys())
    has_fusion = not model_config["eao"]          # the EAO baseline has no fusion token (train_accel_gpu.py:47,155,175)
    uni = {k: P.metrics.Uniformity() for k in names + (["fusion"] if has_fusion else [])}
    ali = {k: P.metrics.Alignment() for k in (names if has_fusion else [])}
    sums, n = {}, 0
    with torch.no_grad():
        for batch in eval_batches:
            out = model(move_to(batch, device))
            n += 1
            sums["total_loss"] = sums.get("total_loss", 0.0) + float(out["loss"])
            for k, v in out["losses"].items():
                sums[k] = sums.get(k, 0.0) + float(v)
            if step_log is not None:
                step_log({"val_step_total_loss": float(out["loss"]),
                          **{"val_step_" + k: float(v) for k, v in out["losses"].items() if "|" not in k}})
            for k in names:
                sm = out["modality_sample_mask"][k]
                uni[k].update(out[k][sm])
                if has_fusion:
                    ali[k].update(out[k][sm], out["fusion"][sm])
            if has_fusion:
                uni["fusion"].update(out["fusion"])
    rec = {f"val_epoch_{k}": v / max(1, n) for k, v in sums.items() if "|" not in k}
    mean = lambda d: sum(d.values()) / max(1, len(d))
    for tag, norm in (("", False), ("norm_", True)):
        u = {f"val_epoch_{tag}uniformity_{k}": float(v.compute(norm=norm, sync=sync)) for k, v in uni.items()}
        rec.update(u); rec[f"val_epoch_{tag}unformity_avg"] = mean(u)
        if has_fusion:
            a = {f"val_epoch_{tag}alignment_{k}": float(v.compute(norm=norm, sync=sync)) for k, v in ali.items()}
            rec.update(a); rec[f"val_epoch_{tag}alignment_avg"] = mean(a)
    if was_training:
        model.train()
    return rec


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    synthetic_steps = int(sys.argv[sys.argv.index("--synthetic") + 1]) if "--synthetic" in sys.argv else 0
    log_every = max(1, int(sys.argv[sys.argv.index("--log-every") + 1])) if "--log-every" in sys.argv else 1
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        # RCCL ("nccl" on ROCm); MCA_DIST_BACKEND=gloo rehearses the multi-rank path on a box with one GPU (tests)
        backend = os.environ.get("MCA_DIST_BACKEND", "nccl")
        torch.distributed.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
    config = P.config.training_config(sys.argv[1], make_output_dir=rank == 0)
    torch.manual_seed(config.seed)
    model_config = P.config.get_model_config(config)
    modality_config = config.get("modality_config", config.get("modality_configs", {}))

    # ---- data
    if synthetic_steps:
        def batches(epoch):
            for i in range(synthetic_steps):
                yield P.data.synthetic_batch(model_config, config.batch_size, seed=1234 + rank + 1000 * (epoch * synthetic_steps + i),
                                             p_drop=max([c.get("dropout", 0.0) for c in modality_config.values()] + [0.0]) if config.get("predrop") else 0.0)
        steps_per_epoch = synthetic_steps
        eval_batches = None
    else:
        from torch.utils.data import DataLoader
        from torch.utils.data.distributed import DistributedSampler
        # train_accel_gpu.py:31-36: predrop / predrop_config come from the YAML; dropped modalities reach the model as
        # fully-padded rows through the collators
        ds = P.data.setup_data(config.dataset, split=config.split, ds_frac=config.ds_frac, ds_seed=config.ds_seed,
                               predrop=bool(config.get("predrop", False)), predrop_config=modality_config)
        collate = P.MultimodalCollator(modality_config)
        sampler = DistributedSampler(ds["train"], world, rank, shuffle=True, drop_last=True) if world > 1 else None
        train_dl = DataLoader(ds["train"], collate_fn=collate, batch_size=config.batch_size, shuffle=sampler is None, sampler=sampler,
                              num_workers=8, prefetch_factor=4, drop_last=True, pin_memory=True)
        # the reference's eval loader keeps the last partial batch (train_accel_gpu.py:71); prepared for W processes Accelerate
        # deals its batches round-robin and completes the last round from the start of the set (train_accel_gpu.py:93)
        if world > 1:
            eval_dl = DataLoader(ds["test"], collate_fn=collate, batch_sampler=P.data.shard_eval_batches(len(ds["test"]), config.batch_size, world, rank))
        else:
            eval_dl = DataLoader(ds["test"], collate_fn=collate, batch_size=config.batch_size)
        steps_per_epoch = len(train_dl)

        def batches(epoch):
            if sampler is not None:
                sampler.set_epoch(epoch)
            yield from train_dl
        eval_batches = eval_dl

    # ---- model, optimizer, schedule
    model = P.build_model(model_config).to(device)          # EAO(**model_config) if model_config['eao'] else MCA(...), train_accel_gpu.py:51-54
    opt = optim.FusedAdamW(model, lr=config.lr)
    dp = dpmod.DataParallelMCA(model) if world > 1 else None
    total_steps = config.epochs * steps_per_epoch
    # Accelerate's prepared scheduler advances num_processes times per optimizer step (AcceleratedScheduler.step with
    # split_batches=False), while num_training_steps counts the per-rank batches: under DP=W the reference's warm-up and
    # cosine run W times faster than the YAML reads.  Reproduced by default; MCA_SCHED_PER_STEP=1 steps once per update.
    sched_stride = 1 if os.environ.get("MCA_SCHED_PER_STEP") == "1" else world
    model.engine.check_finite = "deferred"          # device flag, read without a sync; AdamW skips a flagged step
    if config.restart:
        meta = P.checkpoint.load_state(config.restart, model, opt)           # train_accel_gpu.py:97-99
        if rank == 0 and meta.get("warnings"):
            print("\n".join(meta["warnings"]), flush=True)
    log = open(os.path.join(config.output_dir, "log.jsonl"), "a") if rank == 0 else None
    step = config.start_epoch * steps_per_epoch
    if config.restart and "scheduler_last_epoch" in meta:          # the restored scheduler position, as load_state gives the reference
        step = meta["scheduler_last_epoch"] // sched_stride
    # --graph: the whole step replayed as one hipGraph, or under data parallelism as a chain of graph segments cut at the
    # collectives (graph.GraphedStep; batches have static shapes: the collators pad every modality to its pad_len)
    use_graph, graphed = "--graph" in sys.argv, None
    lr_at = lambda st: config.lr * lr_factor(config.lr_scheduler_type, st * sched_stride, config.num_warmup_steps, total_steps * sched_stride)
    model.train()
    for epoch in range(config.start_epoch, config.epochs):
        t_epoch = time.time()
        # host -> device copies one batch ahead on a copy stream (data.DevicePrefetcher); MCA_PREFETCH=0: the reference's
        # synchronous move_to in the compute stream (train_accel_gpu.py:111)
        feed = batches(epoch)
        if os.environ.get("MCA_PREFETCH", "1") != "0":
            feed = P.data.DevicePrefetcher(feed, device)
        for idb, batch in enumerate(feed):
            batch = move_to(batch, device)          # (no-op for batches the prefetcher already placed)
            for g in opt.param_groups:
                g["lr"] = lr_at(step)
            if use_graph:
                if graphed is None:
                    graphed = importlib.import_module("mca-paper_amd.graph").GraphedStep(model, opt, batch, clip=config.clip or 0.0, dp=dp)
                loss = graphed.step(batch)
                outputs, gnorm = graphed.out, graphed.gnorm
                step += 1
                model.engine.poll_finite()
            else:
                outputs = model(batch)
                opt.zero_grad()
                loss = outputs["loss"]
                loss.backward()
                if dp is not None:
                    dp.finish_backward()
                gnorm = optim.clip_grad_norm_(model, config.clip) if config.clip else None
                opt.step()
                step += 1
                model.engine.poll_finite()          # raises for a completed step that saw non-finite values (no host sync)
            if rank == 0 and (idb % log_every == 0 or idb == steps_per_epoch - 1):
                # train_accel_gpu.py:126-130.  grad_norm there is read AFTER clip_grad_norm_ scaled the gradients in place, i.e.
                # the clipped norm, and (like param_norm) without the first parameter (utils/training.py): here the fused AdamW
                # applies the clip coefficient on the fly, so the same number is norm * min(1, clip / (norm + 1e-6)) of the
                # helper's norm; the unclipped total norm is logged beside it.
                # ONE device-to-host copy per logged step: every logged scalar stacked on the device first (separate float() reads
                # were a dozen blocking round trips per step at the reference's every-step cadence)
                names = [k for k in outputs["losses"] if "|" not in k]
                dev_vals = [loss.detach(), get_grad_norm(model), get_param_norm(model),
                            gnorm if gnorm is not None else torch.zeros((), device=device)] + [outputs["losses"][k].detach() for k in names]
                host = torch.stack([torch.as_tensor(v, device=device).float().reshape(()) for v in dev_vals]).cpu().tolist()
                total_loss, gn_ref, pn, g_all = host[0], host[1], host[2], (host[3] if gnorm is not None else None)
                coef = min(1.0, config.clip / (g_all + 1e-6)) if (config.clip and g_all is not None) else 1.0
                rec = {"epoch": epoch, "step": step, "total_loss": total_loss, "lr": opt.param_groups[0]["lr"],
                       "param_norm": pn, "grad_norm": gn_ref * coef, "grad_norm_unclipped": g_all,
                       **dict(zip(names, host[4:]))}
                print(json.dumps(rec), flush=True)
                log.write(json.dumps(rec) + "\n"); log.flush()
            if config.n_step_checkpoint and idb % config.n_step_checkpoint == 0 and rank == 0:
                P.checkpoint.save_state(config.output_dir, model, opt, step, sched_stride=sched_stride, next_lr=lr_at(step))
        model.engine.assert_finite()
        if rank == 0:
            P.checkpoint.save_state(os.path.join(config.output_dir, str(epoch)), model, opt, step, sched_stride=sched_stride, next_lr=lr_at(step))
            print(f"epoch {epoch} done in {time.time() - t_epoch:.1f}s", flush=True)
        if config.run_eval_loop and eval_batches is not None:
            def step_log(r):
                if rank == 0:
                    log.write(json.dumps({"epoch": epoch, **r}) + "\n")
            rec = run_eval(model, eval_batches, model_config, device, step_log=step_log, sync=world > 1)
            if rank == 0:
                rec = {"epoch": epoch, **rec}
                print(json.dumps(rec), flush=True)
                log.write(json.dumps(rec) + "\n"); log.flush()
    if rank == 0:
        P.checkpoint.save_model(model, config.output_dir, safe_serialization=True)   # train_accel_gpu.py:187
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
