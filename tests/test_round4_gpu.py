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
