"""bench.py — training samples/sec of the native MCA step (fwd + bwd + clip + AdamW) on synthetic CMU-shaped
batches (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3                    # BASELINE configs[1]: CMU 4-modality MCA, b = 32
    python bench.py --workload long --steps 5 --warmup 2              # BASELINE configs[4]: 4 x 1500 tokens, b = 128
    python bench.py --batch 8                                         # the per-GPU batch of the two DP = 8 configs
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W [--batch 8] [--variant mma --p-drop 0.4]

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     : the dominant kernel of the step, timed live with HIP events on its launch stream; achieved =
                 algorithmic flops per launch / average launch duration (DESIGN.md section "Measurement").
  cpu_baseline : the oracle (CPU restatement of the reference, oracle/mca_oracle.py) timed on this host on a
                 bounded sample of the same workload (N=1, rank 0 only).
and, inside `config`: launch_choice (what --launch auto chose and why; the two-step replay-vs-eager guard under data parallelism),
gpu_state_rank0 (shader clock / board power over the timed region, sysfs) and - single GPU, replayed step - sustained: the rate of
the same replayed step after 30 s of unbroken load, measured AFTER the timed region and never used for `value` (every process
starts in the slower of the chip's two sustained-load operating points, profiles/r04_slow_regime.md; --sustain-seconds 0: off).
--step-times prints the host clock after every timed step (the sampled step's cost is visible there).
The step timed is the one train_accel_gpu.py runs: finite checks ON (device flag, polled without a host sync; the
fused AdamW skips a flagged step), inputs resident in HBM (same batch every step).
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


class GpuStateSampler:
    """amdgpu sysfs (shader clock, board power, junction temperature) of THIS process's GPU every 50 ms while the timed region
    runs: the pool's boxes differ by +-6 % and every box has two sustained-load operating points (profiles/r04_slow_regime.md), so
    a samples/s figure is only comparable with the clock it was measured at.  Read-only files; absent files give nulls."""

    def __init__(self, device_index=0):
        import glob, threading
        self.rows, self._stop, self._thread = [], threading.Event(), None
        self.src = {}
        try:
            pr = torch.cuda.get_device_properties(device_index)
            pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            return
        for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
            dev = os.path.join(card, "device")
            if pci.lower() not in os.path.realpath(dev).lower():
                continue
            hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
            if hw:
                cand = {"sclk_hz": "freq1_input", "power_uW": "power1_input", "power_avg_uW": "power1_average", "junction_mC": "temp1_input"}
                self.src = {k: os.path.join(hw[0], f) for k, f in cand.items() if os.path.exists(os.path.join(hw[0], f))}
            break

    def _run(self):
        while not self._stop.is_set():
            row = {}
            for k, path in self.src.items():
                try:
                    row[k] = int(open(path).read().strip())
                except (OSError, ValueError):
                    pass
            self.rows.append(row)
            self._stop.wait(0.05)

    def start(self):
        import threading
        if self.src:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=1)

        def med(key, scale):
            v = sorted(r[key] * scale for r in self.rows if key in r)
            return round(v[len(v) // 2], 1) if v else None
        return {"sclk_mhz_median": med("sclk_hz", 1e-6), "power_w_median": med("power_uW", 1e-6) or med("power_avg_uW", 1e-6),
                "junction_c_median": med("junction_mC", 1e-3), "samples": len(self.rows), "source": "amdgpu sysfs hwmon, 50 ms, timed region only"}

MFMA_BF16_PEAK_TFLOPS = 2500.0          # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic GFLOP per sample per step, mask-aware, 2 flop/MAC, bwd = 2 x fwd
STEP_GFLOP = {"cmu_mca": 334.8, "cmu_mma": 337.4, "long_mca": 885.7}
PMC_JSON = os.path.join("profiles", "r05_hbm_traffic_pmc.json")
MFMA_BUSY_JSON = os.path.join("profiles", "r05_mfma_busy_pmc.json")          # tools/sq_summary.py: matrix-pipe busy share per kernel


def pmc_mfma_busy(kernel_key: str):
    """Share of SIMD cycles the matrix pipe was executing in the dominant kernel, from the committed counter pass
    (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), tools/sq_summary.py): the COUNTER beside the algorithmic
    `frac` - busy cycles include recomputation, masked-out pairs of visited tiles and the mask product, and run at the throttled
    clock, so mfma_busy >= frac x (2.4 GHz / actual clock).  None when no summary is committed for that kernel."""
    names = {"mca_attn_bwd_onepass/layer": "attn_bwd1p_kernel", "mca_attn_fwd/layer": "attn_fwd4_kernel",
             "mca_attn_bwd_dkv/layer": "attn_bwd_dkv_kernel", "mca_attn_bwd_dq/layer": "attn_bwd_dq_kernel"}
    path = os.path.join(REPO, MFMA_BUSY_JSON)
    if not os.path.exists(path) or kernel_key not in names:
        return None
    return json.load(open(path)).get(names[kernel_key])


def pmc_traffic(kernel_key: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 on gfx950 per
    MI355X_MICROARCH.md section HBM, + WRITE_SIZE; separate --pmc passes of this same command, see profiles/README.md;
    layer launches only: tools/pmc_traffic_json.py keeps the largest grid of a kernel).  None when no PMC summary is
    committed for that kernel."""
    path = os.path.join(REPO, PMC_JSON)
    names = {"mca_gemm_nt": "gemm_nt_", "mca_gemm_tn_acc": "gemm_tn_", "mca_gemm_tn_acc_group": "gemm_tn_256x256_group_kernel",
             "mca_attn_fwd/layer": "attn_fwd_kernel", "mca_attn_bwd_dkv/layer": "attn_bwd_dkv_kernel",
             "mca_attn_bwd_dq/layer": "attn_bwd_dq_kernel", "mca_attn_bwd_onepass/layer": "attn_bwd1p_kernel", "mca_gemm_nt_geglu_fwd": "gemm_nt_persist256_kernel<true>",
             "mca_gemm_nt_geglu_bwd": "gemm_nt_persist_kernel<3", "mca_gemm_nt_lnres": "gemm_nt_256_kernel<false, 1, 1, 2>"}
    if not os.path.exists(path) or kernel_key not in names:
        return None
    tot_b, tot_n = 0.0, 0
    for k, v in json.load(open(path)).items():
        if names[kernel_key] in k and "prep" not in k and not (kernel_key == "mca_gemm_nt" and ("persist_kernel<3" in k or "persist256_kernel<true>" in k or "1, 1, 2>" in k)):
            tot_b += (v["fetch_MB_x2_gfx950"] + v["write_MB_per_launch"]) * 1e6 * v["launches"]
            tot_n += v["launches"]
    return {"bytes_per_launch": round(tot_b / tot_n), "source": PMC_JSON} if tot_n else None


def host_cpu():
    """(model name, physical cores this process may use)."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = min(len(cores), allowed) if cores else allowed
    return model, max(1, n)


def cpu_baseline(P, workload: str, variant: str, protocol: str):
    """the oracle's full step (fwd + bwd + clip + AdamW), fp32, b = 8, the same synthetic CMU-shaped
    batch generator (count of threads and CPU model stated).  protocol 'short' (default, keeps the bench within minutes):
    1 warm-up + 3 timed steps on min(physical cores, 16) threads, and 1 + 2 steps on 8 threads (`value_8_threads`: the figure
    comparable with BASELINE.md section 2's container numbers); 'full': 3 warm-up + 5 timed steps on every physical core
    and on 8 threads.  LONG: b = 1, 1 + 1 steps (N = 6088 dense)."""
    from oracle import mca_oracle as O
    model, phys = host_cpu()
    long_seq = workload == "long"
    b = 1 if long_seq else 8
    if variant == "eao":
        cfg = P.config.cmu_eao_model_config(batch_size=b)
        S = O.EAOStructure(cfg)
    else:
        cfg = P.config.cmu_model_config(batch_size=b, zorro=variant == "mma", long_seq=long_seq)
        S = O.Structure(cfg)
    sd0 = P.params.init_state_dict(cfg, seed=43)
    batch = P.data.synthetic_batch(cfg, b, seed=1234, lengths="uniform")

    def run(threads, warm, timed):
        torch.set_num_threads(threads)
        sd = {k: v.clone() for k, v in sd0.items()}
        opt = None
        ts = []
        for _ in range(warm + timed):
            t0 = time.perf_counter()
            _, _, _, opt = O.train_step(S, sd, batch, "fp32", lr=1e-4, clip=2.0, opt_state=opt)
            ts.append(time.perf_counter() - t0)
        return b * timed / sum(ts[warm:]), sum(ts)

    warm, timed = (1, 1) if long_seq else ((3, 5) if protocol == "full" else (1, 3))
    # default: the 16 host threads that are one GPU's share of the box (measured on the 2 x 64-core EPYC 9575F host: 0.59
    # samples/s on 16 threads, 0.32 on all 128: the oracle's small ops do not scale across sockets); 'full': every physical core
    threads = phys if protocol == "full" else min(phys, 16)
    v, spent = run(threads, warm, timed)
    out = {"value": round(v, 4), "unit": "samples/s", "cores": threads, "physical_cores": phys, "cpu_model": model, "kind": "port",
           "sample": f"{'LONG 4x1500' if long_seq else 'CMU 4-modality'} {variant.upper()} fp32 full step (fwd+bwd+clip+AdamW) "
                     f"at batch {b}, uniform lengths, {timed} timed steps after {warm} warm-up ({spent:.0f} s of CPU work on {threads} threads)"}
    if not long_seq:
        w8, t8 = (3, 5) if protocol == "full" else (1, 2)
        v8, spent8 = run(min(8, phys), w8, t8)
        out["value_8_threads"] = round(v8, 4)
        out["sample"] += f"; 8-thread run: {t8} timed steps after {w8} warm-up, {spent8:.0f} s"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cmu", choices=["cmu", "long"],
                    help="cmu: BASELINE configs[1-3] (N = 2538); long: configs[4] (4 x 1500 tokens, N = 6088, batch 128 per GPU)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: 32 for cmu, 128 for long; the DP = 8 configs use 8)")
    ap.add_argument("--variant", default="mca", choices=["mca", "mma", "eao"], help="eao = the paper's EAO baseline (configs/CMU_config1_EAO.yaml); default batch 8")
    ap.add_argument("--lengths", default="full", choices=["full", "uniform"])
    ap.add_argument("--p-drop", type=float, default=0.0)
    ap.add_argument("--inputs", default="device", choices=["device", "host"],
                    help="host: every step first copies its batch from pinned host memory (the PCIe-inclusive rate; NOT the headline value)")
    ap.add_argument("--attn", default="bf16", choices=["bf16", "fp8"], help="attention operand type (fp8: block-scaled MFMA, BASELINE configs[4])")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (small per-GPU batches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-protocol", default="short", choices=["short", "full"])
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--sustain-seconds", type=float, default=45.0, help="single GPU, replayed step: after the timed region keep replaying for this long and report the rate of the last third in config.sustained (0: off); `value` is never taken from it")
    ap.add_argument("--step-times", action="store_true", help="print the host clock after every timed step to stderr (diagnosis)")
    ap.add_argument("--launch", default="auto", choices=["auto", "eager", "graph"], help="auto: hipGraph replay (one graph on one GPU; graph segments cut at the eager collectives under data parallelism); falls back to the eager loop if the capture fails")
    ap.add_argument("--sample-every", type=int, default=20, help="record per-kernel HIP events on every n-th timed step (a sampled step runs its kernels one at a time and costs ~1.3 steps)")
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
    long_seq = args.workload == "long"
    b = args.batch or (128 if long_seq else (8 if args.variant == "eao" else 32))
    if args.variant == "eao":
        if long_seq:
            raise SystemExit("--variant eao is defined on the CMU workload")
        cfg = P.config.cmu_eao_model_config(batch_size=b)
    else:
        cfg = P.config.cmu_model_config(batch_size=b, zorro=args.variant == "mma", long_seq=long_seq)
    torch.manual_seed(43)
    model = P.build_model(cfg).to(dev)
    eng = model.engine
    eng.check_finite = "deferred"              # as train_accel_gpu.py: finite checks ON, device flag, no host sync in the step
    if args.attn == "fp8":
        eng.set_attention_dtype("fp8")
    opt = optim.FusedAdamW(model, lr=1e-4)
    dp = dpmod.DataParallelMCA(model) if world > 1 else None
    batch = P.data.synthetic_batch(cfg, b, seed=1234 + rank, p_drop=args.p_drop, lengths=args.lengths, device=dev)

    host_batch = None
    if args.inputs == "host":          # what the training loop's move_to(batch, device) costs per step
        host_batch = {k: {kk: vv.cpu().pin_memory() for kk, vv in v.items()} for k, v in batch.items()}

    # --inputs host: the training loop's input pipeline (data.DevicePrefetcher: the pinned batch is copied to the device one step
    # ahead on a copy stream, double-buffered); MCA_PREFETCH=0 = the reference's synchronous move_to in the compute stream
    feed = None
    if host_batch is not None and os.environ.get("MCA_PREFETCH", "1") != "0":
        import itertools
        feed = P.data.DevicePrefetcher(itertools.repeat(host_batch), dev)

    def h2d():
        if feed is not None:
            return next(feed)
        for k, v in host_batch.items():
            for kk, vv in v.items():
                batch[k][kk].copy_(vv, non_blocking=True)
        return batch

    def eager_step():
        cur = h2d() if host_batch is not None else batch
        out = model(cur)
        opt.zero_grad()
        out["loss"].backward()
        if dp is not None:
            dp.finish_backward()
        optim.clip_grad_norm_(model, 2.0)
        opt.step()
        eng.poll_finite()
        return out["loss"]

    step = eager_step
    # launch mode: "auto" = one hipGraph replay per step on a single GPU (the eager step issues ~330 launches from Python: on a
    # loaded host that is longer than the GPU needs for them: 23.8-25.0 ms eager against 21.5-21.9 ms replayed on the same box),
    # also under data parallelism (segments cut at the collectives, which stay eager RCCL calls)
    # under data parallelism the replayed step is a chain of graph segments cut at the collectives (graph.GraphedStep)
    use_graph = args.graph or (args.launch == "graph") or args.launch == "auto"
    graphed = None
    # what was chosen and why goes into the JSON line (config.launch_choice), not only to stderr
    choice = {"requested": "graph" if args.graph else args.launch, "chosen": "eager", "reason": "--launch eager"}
    if use_graph:
        err = None
        try:
            graphed = importlib.import_module("mca-paper_amd.graph").GraphedStep(model, opt, batch, clip=2.0, dp=dp)
        except Exception as e:          # auto mode only: a failed capture must not cost the measurement
            if args.launch != "auto" or args.graph:
                raise
            err = e
        ok = torch.tensor([0 if err is not None else 1], device=dev, dtype=torch.int32)
        if world > 1:                   # every rank takes the same path
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok) and world > 1 and args.launch == "auto" and not args.graph:
            # Guard for the data-parallel replay (graph segments with eager collectives between them): time two replayed against
            # two eager steps (max over ranks) and keep the replay only if it is not slower.  The pool has one GPU per box, so the
            # segmented replay has only met RCCL on a world of one rank; with gloo on one shared GPU it is pathologically slow
            # (5.6 s per step: gloo's host threads synchronise streams while hipGraphLaunch holds the runtime) - a measurement
            # must never pay for such an interaction.
            def timed_pair(eager):
                dist.barrier(); torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(2):
                    graphed.step(eager=eager)
                torch.cuda.synchronize()
                tt = torch.tensor([time.perf_counter() - t], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                return float(tt)
            t_replay, t_eager = timed_pair(False), timed_pair(True)
            choice["guard_ms_per_step"] = {"replay": round(t_replay / 2 * 1e3, 2), "eager": round(t_eager / 2 * 1e3, 2)}
            if t_replay > 1.05 * t_eager:
                if rank == 0:
                    print(f"bench: segmented replay {t_replay / 2 * 1e3:.1f} ms/step against {t_eager / 2 * 1e3:.1f} ms eager: running the eager loop", file=sys.stderr, flush=True)
                ok.zero_()
                choice["reason"] = "segmented replay slower than the eager loop in the two-step guard (max over ranks)"
        if int(ok):
            step = graphed.step
            choice.update(chosen="graph", reason=("one hipGraph per step" if world == 1 else "graph segments cut at the eager collectives" +
                                                  (": not slower than the eager loop in the two-step guard" if "guard_ms_per_step" in choice else " (forced: no guard)")))
        else:
            if choice["reason"] == "--launch eager":
                choice["reason"] = f"capture failed ({type(err).__name__ if err else 'on another rank'}): every rank runs the eager loop"
            if err is not None or graphed is None:
                print(f"bench: hipGraph capture failed on rank {rank} ({type(err).__name__ if err else 'another rank'}: {err}); running the eager loop", file=sys.stderr, flush=True)
            use_graph, graphed = False, None
            opt.hyper_external = False
            eng.overlap_wgrad = eng.dbg["overlap_wgrad"]          # (GraphedStep switched the side stream off)

    for w in range(args.warmup):
        exclusive = w == 0 and not args.no_kernel_timing and args.warmup > 1 and not use_graph      # also warm the schedule the sampled steps use
        saved = eng.overlap_wgrad
        if exclusive:
            eng.overlap_wgrad = False
        step()
        eng.overlap_wgrad = saved
    timed = ("mca_attn_bwd_prep", "mca_attn_bwd_prep_onepass", "mca_attn_bwd_onepass", "mca_attn_vmean_if_needed", "mca_attn_vmean", "mca_layernorm_fwd", "mca_layernorm_bwd", "mca_attn_fwd", "mca_attn_fwd_fp8", "mca_attn_quant_mxfp8", "mca_attn_bwd_dq", "mca_attn_bwd_dkv", "mca_attn_bwd_dq_fp8", "mca_attn_bwd_dkv_fp8", "mca_attn_quant_bwd_mxfp8", "mca_gemm_nt", "mca_gemm_nt_lnres", "mca_gemm_nt_geglu_fwd", "mca_gemm_nt_geglu_bwd",
             "mca_gemm_tn_acc", "mca_gemm_tn_acc_group")
    kernel_timing = not args.no_kernel_timing
    if graphed is not None and kernel_timing:
        # The sampled steps of a replayed loop run their body eagerly with an event pair around every launch.  That exact form is
        # run once here, outside the timed region: in a process that is not the first on its box the first such step stalls
        # ~90-100 ms (b = 32: 120.9 ms for the step instead of 23.2; `--step-times`), which a 20-step measurement reports as
        # 26.1 instead of 21.1 ms per step.
        hip.profile_start(timed)
        torch.cuda.synchronize()
        graphed.step(eager=True)
        hip.profile_stop()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if kernel_timing:
        hip.profile_start(timed)
    gpu_state = GpuStateSampler(dev.index if hasattr(dev, "index") and dev.index is not None else 0) if rank == 0 else None
    if gpu_state is not None:
        gpu_state.start()
    t0 = time.perf_counter()
    sampled = 0
    host_marks = []          # --step-times: host clock after every step's launches (a sampled step ends synchronised)
    for i in range(args.steps):
        rec = kernel_timing and (i % args.sample_every) == 0
        hip.profile_enable(rec)
        sampled += int(rec)
        if rec:
            torch.cuda.synchronize()          # sampled step starts on an empty queue ...
            # ... and runs every kernel alone (no side-stream weight gradients), so that an
            # event pair brackets ONE kernel's own duration; the other steps run the overlapped production schedule
            saved = eng.overlap_wgrad
            eng.overlap_wgrad = False
        if graphed is not None:
            loss = graphed.step(h2d() if host_batch is not None else None, eager=rec)          # (a sampled step of the replayed loop runs its body eagerly)
        else:
            loss = step()
        if rec:
            hip.profile_collect()             # its timing events are resolved and released right away
            eng.overlap_wgrad = saved
        host_marks.append(time.perf_counter() - t0)
    hip.profile_enable(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu_state = gpu_state.stop() if gpu_state is not None else None
    prof = hip.profile_stop() if kernel_timing else {}
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if args.step_times and rank == 0:
        print("bench: host ms after each step: " + " ".join(f"{m * 1e3:.1f}" for m in host_marks) + f" | synchronised end {dt * 1e3:.1f}", file=sys.stderr, flush=True)
    eng.assert_finite()                        # the device flag of the last step (blocking read, outside the timed region)
    assert bool(torch.isfinite(loss)), "non-finite loss in the timed region"
    # a step whose gradients blew up is not a measurement (a mis-ordered memset node in the captured graph once left the loss
    # plausible and every gradient at 1e28: DESIGN.md section 5)
    gmax = float(eng.gflat.abs().max())
    assert gmax == gmax and gmax < 1e6, f"gradients of the last timed step are not sane (max |g| = {gmax})"

    # Steady state, reported BESIDE the contract's figure (never as `value`): every process starts in the slower of the chip's two
    # sustained-load operating points and moves to the faster one after 12-27 s of unbroken load (profiles/r04_slow_regime.md), so
    # the K timed steps after W warm-up steps always measure the first; a training run lives in the second.
    sustained = None
    if args.sustain_seconds > 0 and world == 1 and graphed is not None and host_batch is None:
        try:          # the contract's figure is already measured: nothing in this block may lose it
            window = args.sustain_seconds / 3.0
            t_s = time.perf_counter()
            while time.perf_counter() - t_s < args.sustain_seconds - window:
                for _ in range(50):
                    graphed.step()
                torch.cuda.synchronize()
            sampler = GpuStateSampler(dev.index if dev.index is not None else 0)
            sampler.start()
            t_w, n_w = time.perf_counter(), 0
            while time.perf_counter() - t_w < window:
                for _ in range(50):
                    graphed.step()
                n_w += 50
                torch.cuda.synchronize()
            dt_w = time.perf_counter() - t_w
            st = sampler.stop()
            eng.assert_finite()
            sustained = {"samples_per_s": round(b * n_w / dt_w, 2), "ms_per_step": round(dt_w / n_w * 1e3, 3), "steps": n_w,
                         "after_seconds_of_unbroken_load": round(args.sustain_seconds - window, 1), "sclk_mhz_median": st["sclk_mhz_median"],
                         "power_w_median": st["power_w_median"], "note": "replayed step after the timed region; not `value`"}
        except Exception as exc:          # noqa: BLE001 - recorded in the line instead
            sustained = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * b * args.steps / dt
        wl = f"{args.workload}_{args.variant}"
        gf = STEP_GFLOP.get(wl)
        N = eng.N
        line = {
            "metric": "training samples/sec (fwd+bwd+opt), CMU 4-modality MCA", "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.attn == "bf16" else "bf16 (attention QK^T / PV operands fp8 e4m3)",
            "data": "synthetic",
            "config": {"workload": f"synthetic {'LONG 4 x 1500 tokens' if long_seq else 'CMU 4-modality'} ({ {'mca': 'MCA fcl', 'mma': 'MMA/zorro', 'eao': 'EAO baseline, 10 passes as one block-diagonal sequence'}[args.variant]}), "
                                   f"N={N} D=512 L=5 H=8 F=88, lengths={args.lengths}, p_drop={args.p_drop}",
                       "per_gpu_batch": b, "global_batch": b * world, "parallelism": f"dp{world}",
                       "inputs": "device-resident, same batch every step" if host_batch is None else ("pinned host memory, copied to the device every step (PCIe-inclusive" + (", one step ahead on a copy stream)" if feed is not None else ", in the compute stream)")), "finite_checks": "on (device flag, polled)",
                       "attention_operands": args.attn, "attention_backward": model.engine.backward_form(b), "launch": ("hipGraph replay (" + ("one launch per step" if world == 1 else f"{sum(1 for it in graphed.program if isinstance(it, torch.cuda.CUDAGraph))} graph segments cut at {sum(1 for it in graphed.program if not isinstance(it, torch.cuda.CUDAGraph))} eager collectives per step") + (f"; {sampled} of {args.steps} steps eager for the kernel timing)" if sampled else ")")) if use_graph else "eager",
                       "launch_choice": choice,
                       "collectives": (f"{dist.get_backend()} over {dist.get_world_size()} ranks" if world > 1 else "none")},
        }
        if gpu_state is not None:
            line["config"]["gpu_state_rank0"] = gpu_state
        if sustained is not None:
            line["config"]["sustained"] = sustained
        if gf:
            line["step_tflops"] = round(value * gf / 1e3, 1)
            line["step_frac_of_mfma_peak"] = round(value * gf / 1e3 / (MFMA_BF16_PEAK_TFLOPS * world), 4)
        if prof:
            kern = {}
            for key, (n, ms, fl) in prof.items():
                kern[key] = {"launches_per_step": n / sampled, "ms_per_step": round(ms / sampled, 3),
                             "avg_us": round(ms / n * 1e3, 1), "tflops": round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else None}
            dom = max(prof.items(), key=lambda kv: kv[1][1])
            n, ms, fl = dom[1]
            ach = fl / (ms * 1e-3) / 1e12
            tr = pmc_traffic(dom[0]) if not long_seq and b == 32 else None          # the committed PMC passes are of the default workload
            line["roofline"] = {"kernel": dom[0], "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                                # HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE of the committed PMC passes) or null
                                "traffic": (tr or {}).get("bytes_per_launch"),
                                "traffic_source": (tr or {}).get("source"),
                                "mfma_busy": pmc_mfma_busy(dom[0]) if not long_seq and b == 32 else None,
                                "avg_launch_us": round(ms / n * 1e3, 1), "alg_flops_per_launch": fl / n,
                                "sampled_steps": sampled}
            line["kernels"] = kern
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(P, args.workload, args.variant, args.cpu_protocol)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
