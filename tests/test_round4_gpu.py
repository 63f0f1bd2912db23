"""Round-4 GPU tests: bench.py's data-parallel code path on two ranks (what the driver launches with --gpus N, rehearsed on one GPU
with gloo)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_reports_its_launch_choice_and_collectives():
    """`bench.py --gpus 2 --batch 8 --steps 3` as the driver starts it (one process per rank, RANK / WORLD_SIZE / MASTER_* from the
    environment), with MCA_DIST_BACKEND=gloo so that both ranks can share this box's one GPU: the JSON line carries the rank count
    and backend (config.collectives), what the launch guard measured and chose (config.launch_choice: the segmented replay is
    kept only if its two guard steps are not slower than two eager ones, max over ranks) and, when the replay is kept, the
    segment count (config.launch).  On RCCL over xGMI the same code path runs with backend 'nccl'."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MCA_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--batch", "8", "--steps", "3", "--warmup", "1",
                                       "--no-kernel-timing"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=REPO))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]          # ONE line, from rank 0
    rec = json.loads(lines[0])
    cfg = rec["config"]
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["value"] > 0
    assert cfg["per_gpu_batch"] == 8 and cfg["global_batch"] == 16 and cfg["parallelism"] == "dp2"
    assert cfg["collectives"] == "gloo over 2 ranks"
    ch = cfg["launch_choice"]
    assert ch["requested"] == "auto" and ch["chosen"] in ("graph", "eager") and ch["reason"]
    assert set(ch["guard_ms_per_step"]) == {"replay", "eager"} and all(v > 0 for v in ch["guard_ms_per_step"].values())
    if ch["chosen"] == "graph":
        # forward | all-gather | loss + pooling backward | 7 buckets ... | finite flag: segments and eager collectives alternate
        assert "graph segments cut at" in cfg["launch"] and ch["guard_ms_per_step"]["replay"] <= 1.05 * ch["guard_ms_per_step"]["eager"]
    else:
        assert cfg["launch"] == "eager" and ch["guard_ms_per_step"]["replay"] > 1.05 * ch["guard_ms_per_step"]["eager"]
    # value = samples of ALL ranks / the slowest rank's time
    assert abs(rec["value"] - 16 * 3 / (rec["ms_per_step"] * 3e-3)) <= 1e-2 * rec["value"]


def test_bench_single_gpu_line_follows_the_contract():
    """`python bench.py --steps 4 --warmup 2` on one GPU (small batch, no CPU baseline: the contract's other fields): ONE JSON
    line with BASELINE's metric and unit, whole-job value consistent with ms_per_step, `roofline` of the dominant launch class
    (achieved / peak = frac, live HIP events of the one eager sampled step), the workload named in `config`, the clock the figure was
    measured at, and the steady-state rate beside - never instead of - `value`."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--batch", "4", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
                        "--sustain-seconds", "3"], capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["metric"].startswith("training samples/sec") and rec["unit"] == "samples/s" and rec["higher_is_better"] is True
    assert rec["n_gpus"] == 1 and rec["steps"] == 4 and rec["warmup"] == 2 and rec["scaling"] == "weak" and rec["vs_baseline"] is None
    assert rec["dtype"] == "bf16" and rec["data"] == "synthetic" and "model" not in rec["config"] and "workload" in rec["config"]
    assert abs(rec["value"] - 4 * 4 / (rec["ms_per_step"] * 4e-3)) <= 1e-2 * rec["value"]
    rf = rec["roofline"]
    assert rf["bound"] in ("mfma", "hbm") and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0 < rf["frac"] < 1 and rf["avg_launch_us"] > 0 and rf["sampled_steps"] == 1
    assert rec["config"]["launch_choice"]["chosen"] == "graph" and "hipGraph replay" in rec["config"]["launch"]
    gs = rec["config"]["gpu_state_rank0"]
    assert gs["samples"] >= 1 and (gs["sclk_mhz_median"] is None or 100 < gs["sclk_mhz_median"] < 3000)
    su = rec["config"]["sustained"]
    assert su["samples_per_s"] > 0 and su["steps"] >= 50 and "not `value`" in su["note"]
    assert "cpu_baseline" not in rec and rec["kernels"]["mca_attn_bwd_dkv/layer"]["launches_per_step"] == 5


def test_logged_norms_from_the_flat_buffers_equal_the_per_tensor_walk(pkg):
    P = pkg
    """utils.training.get_grad_norm / get_param_norm (logged every step, train_accel_gpu.py:126-130) read the engine's flat
    buffers after a native backward: same values as the reference's walk over the tensors (first parameter skipped, its quirk),
    and the full norm equals what clip_grad_norm_ reports."""
    import importlib
    from util_small import small_config, to_device
    from utils.training import get_grad_norm, get_param_norm
    optim = importlib.import_module("mca-paper_amd.optim")
    cfg = small_config("mca")
    model = P.build_model(cfg).cuda()
    batch = to_device(P.data.synthetic_batch(cfg, 6, seed=5, p_drop=0.3), "cuda")
    out = model(batch)
    optim.FusedAdamW(model, lr=1e-3).zero_grad()
    out["loss"].backward()
    total = float(optim.clip_grad_norm_(model, 2.0))
    params = list(model.parameters())
    assert all(p.grad.data_ptr() == model.engine.grad_of(p).data_ptr() for p in params)          # the flat path is the one taken
    walk = lambda ts: float(sum(float(t.double().pow(2).sum()) for t in ts) ** 0.5)
    g_all, g_rest = walk([p.grad for p in params]), walk([p.grad for p in params[1:]])
    assert abs(float(get_grad_norm(model)) - g_rest) <= 1e-5 * g_rest and abs(float(get_grad_norm(model, skip_first=False)) - g_all) <= 1e-5 * g_all
    assert abs(total - g_all) <= 1e-5 * g_all and get_grad_norm(model).dtype == torch.float32 and get_grad_norm(model).shape == (1,)
    p_rest = walk([p.detach() for p in params[1:]])
    assert abs(float(get_param_norm(model)) - p_rest) <= 1e-6 * p_rest
    params[3].grad = params[3].grad.clone()          # a gradient that is not the flat view: the walk is taken, same value
    assert abs(float(get_grad_norm(model)) - g_rest) <= 1e-5 * g_rest
