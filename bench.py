"""bench.py — training samples/sec of the native MCA step (fwd + bwd + clip + AdamW) on synthetic CMU-shaped
batches (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     : the dominant kernel of the step, timed live with HIP events on its launch stream; achieved =
                 algorithmic flops per launch / average launch duration (DESIGN.md section "Measurement").
  cpu_baseline : the oracle (CPU restatement of the reference, oracle/mca_oracle.py) timed on this host on a
                 bounded sample of the same workload (N=1, rank 0 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_BF16_PEAK_TFLOPS = 2500.0          # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic GFLOP per sample per step, mask-aware, 2 flop/MAC, bwd = 2 x fwd
STEP_GFLOP = {"cmu_mca": 334.8, "cmu_mma": 337.4}


def pmc_traffic(kernel_key: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 on gfx950 per
    MI355X_MICROARCH.md section HBM, + WRITE_SIZE; separate --pmc passes of this same command, see profiles/README.md).
    None when no PMC summary is committed for that kernel."""
    path = os.path.join(REPO, "profiles", "r01_hbm_traffic_pmc.json")
    names = {"mca_gemm_nt": "gemm_nt_", "mca_gemm_tn_acc": "gemm_tn_", "mca_gemm_tn_acc_group": "gemm_tn_256x256_group_kernel", "mca_attn_fwd/layer": "attn_fwd_kernel",
             "mca_attn_bwd/layer": "attn_bwd_kernel", "mca_gemm_nt_geglu_fwd": "gemm_nt_persist256_kernel<true>",
             "mca_gemm_nt_geglu_bwd": "gemm_nt_persist_kernel<3", "mca_gemm_nt_lnres": "gemm_nt_256_kernel<false, 1, 1, 2>"}
    if not os.path.exists(path) or kernel_key not in names:
        return None
    tot_b, tot_n = 0.0, 0
    for k, v in json.load(open(path)).items():
        if names[kernel_key] in k and not (kernel_key == "mca_gemm_nt" and ("persist_kernel<3" in k or "persist256_kernel<true>" in k or "1, 1, 2>" in k)):
            tot_b += (v["fetch_MB_x2_gfx950"] + v["write_MB_per_launch"]) * 1e6 * v["launches"]
            tot_n += v["launches"]
    return {"bytes_per_launch": round(tot_b / tot_n), "source": "profiles/r01_hbm_traffic_pmc.json"} if tot_n else None


def cpu_baseline(P, cfg, threads: int, b: int = 4):
    from oracle import mca_oracle as O
    torch.set_num_threads(threads)
    S = O.Structure(cfg)
    sd = P.params.init_state_dict(cfg, seed=43)
    batch = P.data.synthetic_batch(cfg, b, seed=1234, lengths="uniform")
    opt = None
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        _, _, _, opt = O.train_step(S, sd, batch, "fp32", lr=1e-4, clip=2.0, opt_state=opt)
        times.append(time.perf_counter() - t0)
    timed = sum(times[1:])
    return {"value": round(2 * b / timed, 4), "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"CMU 4-modality MCA fp32 full step (fwd+bwd+clip+AdamW) at batch {b}, 2 timed steps after 1 warm-up step "
                      f"({timed:.1f} s of CPU work)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE config: 32 on 1 GPU)")
    ap.add_argument("--variant", default="mca", choices=["mca", "mma"])
    ap.add_argument("--lengths", default="full", choices=["full", "uniform"])
    ap.add_argument("--p-drop", type=float, default=0.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--sample-every", type=int, default=10, help="record per-kernel HIP events on every n-th timed step")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one process per GPU; MCA_DIST_BACKEND=gloo rehearses the N>1 code path on a box with fewer GPUs than ranks
    backend = os.environ.get("MCA_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    P = importlib.import_module("mca-paper_amd")
    hip = importlib.import_module("mca-paper_amd.hip")
    optim = importlib.import_module("mca-paper_amd.optim")
    dpmod = importlib.import_module("mca-paper_amd.dp")
    b = args.batch
    cfg = P.config.cmu_model_config(batch_size=b, zorro=args.variant == "mma")
    torch.manual_seed(43)
    model = P.MCA(**cfg).to(dev)
    eng = model.engine
    eng.check_finite = False                   # no host syncs inside the timed region (checked once after it)
    opt = optim.FusedAdamW(model, lr=1e-4)
    dp = dpmod.DataParallelMCA(model) if world > 1 else None
    batch = P.data.synthetic_batch(cfg, b, seed=1234 + rank, p_drop=args.p_drop, lengths=args.lengths, device=dev)

    def step():
        out = model(batch)
        opt.zero_grad()
        out["loss"].backward()
        if dp is not None:
            dp.finish_backward()
        optim.clip_grad_norm_(model, 2.0)
        opt.step()
        return out["loss"]

    for w in range(args.warmup):
        exclusive = w == 0 and not args.no_kernel_timing and args.warmup > 1      # also warm the schedule the sampled steps use
        saved = (eng.overlap_wgrad, eng.micro_batches)
        if exclusive:
            eng.overlap_wgrad, eng.micro_batches = False, 1
        step()
        eng.overlap_wgrad, eng.micro_batches = saved
    timed = ("mca_attn_fwd", "mca_attn_bwd", "mca_gemm_nt", "mca_gemm_nt_lnres", "mca_gemm_nt_geglu_fwd", "mca_gemm_nt_geglu_bwd",
             "mca_gemm_tn_acc", "mca_gemm_tn_acc_group")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if not args.no_kernel_timing:
        hip.profile_start(timed)
    t0 = time.perf_counter()
    sampled = 0
    for i in range(args.steps):
        rec = (not args.no_kernel_timing) and (i % args.sample_every) == 0
        hip.profile_enable(rec)
        sampled += int(rec)
        if rec:
            torch.cuda.synchronize()          # sampled step starts on an empty queue ...
            # ... and runs every kernel alone (no side-stream weight gradients, no half-batch interleave), so that an
            # event pair brackets ONE kernel's own duration; the other steps run the overlapped production schedule
            saved = (eng.overlap_wgrad, eng.micro_batches)
            eng.overlap_wgrad, eng.micro_batches = False, 1
        loss = step()
        if rec:
            hip.profile_collect()             # its timing events are resolved and released right away
            eng.overlap_wgrad, eng.micro_batches = saved
    hip.profile_enable(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = hip.profile_stop() if not args.no_kernel_timing else {}
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    assert bool(torch.isfinite(loss)), "non-finite loss in the timed region"

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * b * args.steps / dt
        wl = f"cmu_{args.variant}"
        line = {
            "metric": "training samples/sec (fwd+bwd+opt), CMU 4-modality MCA", "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"synthetic CMU 4-modality ({'MCA fcl' if args.variant == 'mca' else 'MMA/zorro'}), N=2538 D=512 L=5 H=8 F=88, "
                                   f"lengths={args.lengths}, p_drop={args.p_drop}", "per_gpu_batch": b, "global_batch": b * world,
                       "parallelism": f"dp{world}"},
            "step_tflops": round(value * STEP_GFLOP[wl] / 1e3, 1),
            "step_frac_of_mfma_peak": round(value * STEP_GFLOP[wl] / 1e3 / (MFMA_BF16_PEAK_TFLOPS * world), 4),
        }
        if prof:
            kern = {}
            for key, (n, ms, fl) in prof.items():
                kern[key] = {"launches_per_step": n / sampled, "ms_per_step": round(ms / sampled, 3),
                             "avg_us": round(ms / n * 1e3, 1), "tflops": round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else None}
            dom = max(prof.items(), key=lambda kv: kv[1][1])
            n, ms, fl = dom[1]
            ach = fl / (ms * 1e-3) / 1e12
            line["roofline"] = {"kernel": dom[0], "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                                # HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE of the committed PMC passes) or null
                                "traffic": (pmc_traffic(dom[0]) or {}).get("bytes_per_launch"),
                                "traffic_source": (pmc_traffic(dom[0]) or {}).get("source"),
                                "avg_launch_us": round(ms / n * 1e3, 1), "alg_flops_per_launch": fl / n,
                                "sampled_steps": sampled}
            line["kernels"] = kern
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            line["cpu_baseline"] = cpu_baseline(P, P.config.cmu_model_config(batch_size=4, zorro=args.variant == "mma"), threads)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
