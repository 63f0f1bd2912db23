"""The replayed b = 32 CMU step under a monitor: per-100-step time beside the GPU's clocks / power / temperatures sampled every
100 ms from sysfs (amdgpu hwmon + pp_dpm_*), to name what the 13-20 % "slow regime" of rounds 2-3 is.

usage: soak_monitor.py --seconds 60 --tag fresh [--batch 32] [--workers 0] [--loader]
  --workers N: N busy host processes beside the loop (stand-ins for DataLoader workers: a collator-like numpy loop each)
  --loader: the loop of train_accel_gpu.py --graph instead of one resident batch: torch DataLoader (8 worker processes,
            prefetch_factor 4, pin_memory) over full-length CMU-shaped samples -> MultimodalCollator -> DevicePrefetcher (copy stream,
            one batch ahead) -> GraphedStep.step(batch) -> poll_finite, logging the loss every step (a host read per step, as the
            reference's loop does)
Writes gpurun_out/soak_<tag>.json: the step-time series, the sampled series, and a summary (median / p95 ms per step, sclk /
power / temperature ranges, which sysfs files existed).  Run phases as SEPARATE processes from one shell line, e.g.
  python tools/soak_monitor.py --tag fresh && rocprofv3 --pmc SQ_WAVES -d gpurun_out/pmc_tmp -- python3 bench.py --steps 20 \
      --no-cpu-baseline && python tools/soak_monitor.py --tag after_pmc
"""
import argparse, glob, importlib, json, multiprocessing as mp, os, statistics as st, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def find_sources(pci_bus_id=None):
    """the sysfs files of the GPU torch uses: the card whose PCI address is pci_bus_id ('0000:05:00.0'; a box exposes one of its
    eight GPUs to the process, and card0 is usually another tenant's), else the first card with an amdgpu hwmon: {name: path}"""
    src = {}
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*"))
    if pci_bus_id:
        match = [c for c in cards if pci_bus_id.lower() in os.path.realpath(os.path.join(c, "device")).lower()]
        cards = match or cards
    for card in cards:
        dev = os.path.join(card, "device")
        hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
        if not hw or not os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
            continue
        h = hw[0]
        cand = {"sclk_levels": os.path.join(dev, "pp_dpm_sclk"), "mclk_levels": os.path.join(dev, "pp_dpm_mclk"),
                "fclk_levels": os.path.join(dev, "pp_dpm_fclk"), "perf_level": os.path.join(dev, "power_dpm_force_performance_level"),
                "busy": os.path.join(dev, "gpu_busy_percent"), "power_avg_uW": os.path.join(h, "power1_average"),
                "power_in_uW": os.path.join(h, "power1_input"), "power_cap_uW": os.path.join(h, "power1_cap"),
                "sclk_hz": os.path.join(h, "freq1_input"), "mclk_hz": os.path.join(h, "freq2_input")}
        for t in sorted(glob.glob(os.path.join(h, "temp*_input"))):
            lab = t.replace("_input", "_label")
            name = open(lab).read().strip() if os.path.exists(lab) else os.path.basename(t)
            cand["temp_" + name + "_mC"] = t
        src = {k: v for k, v in cand.items() if os.path.exists(v)}
        src["_card"] = card
        break
    return src


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def cur_level(txt):
    """'0: 132Mhz\\n1: 2100Mhz *' -> 2100 (the starred level), None if unreadable"""
    if not txt:
        return None
    for line in txt.splitlines():
        if line.rstrip().endswith("*"):
            digits = "".join(ch for ch in line.split(":", 1)[1] if ch.isdigit())
            return int(digits) if digits else None
    return None


def sampler(src, stop, out, period=0.1):
    t0 = time.perf_counter()
    while not stop.is_set():
        row = {"t": round(time.perf_counter() - t0, 3)}
        for k, p in src.items():
            if k.startswith("_"):
                continue
            v = read(p)
            if k.endswith("_levels"):
                row[k[:-7] + "_mhz"] = cur_level(v)
            elif k == "perf_level":
                row[k] = v
            else:
                try:
                    row[k] = int(v)
                except (TypeError, ValueError):
                    row[k] = None
        out.append(row)
        time.sleep(period)


def busy_worker(stop):
    import numpy as np
    rng = np.random.default_rng(os.getpid())
    while not stop.is_set():
        a = rng.standard_normal((1500, 74)).astype("float32")
        np.pad(a, ((0, 0), (0, 54))).sum()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--tag", default="run")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--loader", action="store_true")
    ap.add_argument("--log-lag", type=int, default=0, help="--loader: read the loss of the step BEFORE the one just enqueued (0: this step's, a host sync per step)")
    args = ap.parse_args()
    import torch
    P = importlib.import_module("mca-paper_amd"); optim = importlib.import_module("mca-paper_amd.optim"); graph = importlib.import_module("mca-paper_amd.graph")
    pr = torch.cuda.get_device_properties(0)
    pci = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)) if hasattr(pr, "pci_bus_id") else None
    src = find_sources(pci)
    src["_pci"] = pci
    samples, stop = [], threading.Event()
    th = threading.Thread(target=sampler, args=(src, stop, samples), daemon=True)
    th.start()
    wstop = mp.Event()
    workers = [mp.Process(target=busy_worker, args=(wstop,), daemon=True) for _ in range(args.workers)]
    for w in workers:
        w.start()
    cfg = P.config.cmu_model_config(batch_size=args.batch)
    torch.manual_seed(43)
    model = P.build_model(cfg).cuda(); model.engine.check_finite = "deferred"
    opt = optim.FusedAdamW(model, lr=1e-4)
    batch = P.data.synthetic_batch(cfg, args.batch, seed=1234, lengths="full", device="cuda")          # BASELINE configs[1]: full-length sequences
    t_setup = time.perf_counter()
    g = None if args.eager else graph.GraphedStep(model, opt, batch, clip=2.0)

    feed = None
    ring, state = [torch.zeros((), device="cuda"), torch.zeros((), device="cuda")], {"i": 0}
    if args.loader:
        from torch.utils.data import DataLoader, Dataset

        class FullLengthSet(Dataset):          # samples as the HF dataset hands them over (with_format("torch")): {modality: {"data": (tokens, features)}}
            def __init__(self, enc):
                self.enc, self.pool = enc, None

            def __len__(self):
                return 1 << 30

            def __getitem__(self, i):
                if self.pool is None:          # per worker: 16 random samples, handed out in turn (generating 0.6 M normals per sample would bound the loop)
                    gen = torch.Generator().manual_seed(1 + (torch.utils.data.get_worker_info().id if torch.utils.data.get_worker_info() else 0))
                    self.pool = [{n: {"data": torch.randn(e["max_tokens"], e["input_size"], generator=gen)} for n, e in self.enc.items()} for _ in range(16)]
                return self.pool[i % 16]
        mod_cfg = {n: {"type": "embedded_sequence", "pad_len": e["max_tokens"], "embedding_size": e["input_size"], "data_col_name": "data", "dropout": 0.0}
                   for n, e in cfg["encoder_configs"].items()}
        dl = DataLoader(FullLengthSet(cfg["encoder_configs"]), collate_fn=P.MultimodalCollator(mod_cfg), batch_size=args.batch, shuffle=False,
                        num_workers=8, prefetch_factor=4, drop_last=True, pin_memory=True)
        feed = iter(P.data.DevicePrefetcher(iter(dl), torch.device("cuda")))

    def step():
        if feed is not None:
            loss = g.step(next(feed))
            model.engine.poll_finite()
            if args.log_lag:          # a two-slot ring of loss copies: the value read is the previous step's, already computed
                ring[state["i"] & 1].copy_(loss, non_blocking=True)
                state["i"] += 1
                return float(ring[state["i"] & 1]) if state["i"] > 1 else 0.0
            return float(loss)          # the reference's loop logs the loss of every step
        if g is not None:
            return g.step()
        out = model(batch); opt.zero_grad(); out["loss"].backward(); optim.clip_grad_norm_(model, 2.0); opt.step()
        model.engine.poll_finite()
        return out["loss"]
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    series, t0 = [], time.perf_counter()
    while time.perf_counter() - t0 < args.seconds:
        ta = time.perf_counter()
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        series.append({"t": round(time.perf_counter() - t0, 2), "ms_per_step": round((time.perf_counter() - ta) * 10, 4)})
        if len(series) % 10 == 0:          # a progress line every 1,000 steps (a silent GPU job is taken to be hung)
            print(f"t = {series[-1]['t']:.0f} s: {series[-1]['ms_per_step']:.3f} ms / step", flush=True)
    stop.set(); wstop.set(); th.join(timeout=2)
    ms = [r["ms_per_step"] for r in series]

    def rng(key, scale=1.0):
        v = [s[key] * scale for s in samples if s.get(key) is not None]
        return {"min": round(min(v), 1), "median": round(st.median(v), 1), "max": round(max(v), 1), "n": len(v)} if v else None
    summary = {"tag": args.tag, "batch": args.batch, "mode": "eager" if args.eager else ("hipGraph replay fed by DataLoader (8 workers) + collator + prefetcher" if args.loader else "hipGraph replay"), "host_workers": args.workers,
               "steps": 100 * len(series), "ms_per_step": {"first": ms[0], "median": round(st.median(ms), 4), "min": min(ms), "max": max(ms),
                                                             "p95": sorted(ms)[int(0.95 * (len(ms) - 1))], "last": ms[-1]},
               "samples_per_s_median": round(args.batch / st.median(ms) * 1e3, 1),
               "sysfs_sources": {k: v for k, v in src.items()},
               "perf_level": sorted({s.get("perf_level") for s in samples if s.get("perf_level")}),
               "sclk_mhz": rng("sclk_mhz"), "sclk_hwmon_mhz": rng("sclk_hz", 1e-6), "mclk_mhz": rng("mclk_mhz"), "fclk_mhz": rng("fclk_mhz"),
               "power_W": rng("power_avg_uW", 1e-6) or rng("power_in_uW", 1e-6), "power_cap_W": rng("power_cap_uW", 1e-6),
               "busy_percent": rng("busy"), "temps_C": {k[5:-3]: rng(k, 1e-3) for k in (samples[0] if samples else {}) if k.startswith("temp_")}}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"summary": summary, "steps": series, "samples": samples[::5]}, open(os.path.join(ROOT, "gpurun_out", f"soak_{args.tag}.json"), "w"))
    print(json.dumps(summary), flush=True)
    if feed is not None:          # stop the loader's worker processes before leaving
        feed = None
        import gc
        del dl
        gc.collect()
    os._exit(0)          # (daemon worker processes: no join)


if __name__ == "__main__":
    main()
