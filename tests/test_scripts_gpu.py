"""GPU tests of the entry scripts end to end: infer_accel_gpu.py, train_accel_gpu.py (synthetic and real data, evaluation loop, data
parallelism) and bench.py's contract line (one GPU, two ranks)."""
import copy
import importlib
import json
import os
import pytest
import socket
import subprocess
import sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, REPO
from util_small import small_config, rel_err, to_device
from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return importlib.import_module("mca-paper_amd")


# ------------------------------------------------------------------------------------------------ f1: real data + eval loop
def _ragged_dataset(path, cfg, n=24, seed=0):
    from datasets import Dataset
    g = torch.Generator().manual_seed(seed)
    samples = []
    for i in range(n):
        s = {"Labels": {"data": [float(i)]}}
        for name, enc in cfg["encoder_configs"].items():
            k = int(torch.randint(1, enc["max_tokens"] + 1, (1,), generator=g))
            s[name] = {"data": torch.randn(k, enc["input_size"], generator=g).tolist()}
        samples.append(s)
    Dataset.from_list(samples).save_to_disk(path)


def test_infer_script_reproduces_reference_embeddings(P, tmp_path):
    """infer_accel_gpu.py end to end: YAML -> HF dataset on disk -> collators -> reference-written checkpoint ->
    {train,eval}_{embeddings,masks,labels}.pt in the reference's format (infer_accel_gpu.py:97-136)."""
    import yaml
    from datasets import Dataset
    io = torch.load(os.path.join(GOLDEN, "ref_state_io.pt"), weights_only=False)
    cfg, eb = io["config"], io["eval_batch"]
    samples = []
    for rep in range(2):
        for i in range(4):
            s = {"Labels": {"data": [float(i)]}}
            for name, enc in cfg["encoder_configs"].items():
                if enc["type"] == "EmbeddedSequenceEncoder":
                    n_valid = int((~eb[name]["attention_mask"][i]).sum())
                    s[name] = {"data": eb[name]["tokens"][i, :n_valid].tolist() if n_valid else None}
                else:
                    dropped = bool(eb[name]["attention_mask"][i].all())
                    s[name] = {"values": None if dropped else eb[name]["values"][i].tolist()}
            samples.append(s)
    ds_path = str(tmp_path / "ds")
    Dataset.from_list(samples).save_to_disk(ds_path)
    mod_cfg = {}
    for name, enc in cfg["encoder_configs"].items():
        if enc["type"] == "EmbeddedSequenceEncoder":
            mod_cfg[name] = {"type": "embedded_sequence", "pad_len": enc["max_tokens"], "embedding_size": enc["input_size"], "data_col_name": "data", "dropout": 0.0}
        else:
            mod_cfg[name] = {"type": "sequence", "pad_len": enc["max_tokens"], "data_col_name": "values", "pad_token": -10000, "dropout": 0.0}
    y = dict(encoder_configs=cfg["encoder_configs"], modality_config=mod_cfg, hidden_size=cfg["dim"], layers=cfg["depth"], heads=cfg["heads"],
             dim_head=cfg["dim_head"], num_fusion_tokens=cfg["num_fusion_tokens"], batch_size=2, fcl=cfg["fcl"], fcl_root=cfg["fcl_root"],
             bimodal_contrastive=cfg["bimodal_contrastive"], non_fusion_fcl=cfg["non_fusion_fcl"], fusion_combos=cfg["fusion_combos"],
             zorro=cfg["zorro"], dataset=ds_path, split=0.25, ds_seed=42, predrop=False, restart=os.path.join(GOLDEN, "ref_state"),
             output_dir=str(tmp_path / "out"), label_col="Labels")
    ypath = tmp_path / "infer.yaml"
    ypath.write_text(yaml.safe_dump(y, sort_keys=False))          # the modality order IS the token order
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(REPO, "infer_accel_gpu.py"), str(ypath)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    n_seen = 0
    for tv in ("train", "eval"):
        emb = torch.load(tmp_path / "out" / f"{tv}_embeddings.pt", weights_only=False)
        masks = torch.load(tmp_path / "out" / f"{tv}_masks.pt", weights_only=False)
        labels = torch.load(tmp_path / "out" / f"{tv}_labels.pt", weights_only=False)
        assert set(masks) == set(cfg["encoder_configs"])
        for row in range(labels.shape[0]):
            i = int(labels[row, 0]); n_seen += 1
            for k, want in io["embeddings"].items():
                key = frozenset(int(x) for x in k.split("|")) if "|" in k else k
                assert rel_err(emb[key][row], want[i]) < 3e-3, (tv, row, k)          # one 128-vector: a single slot of a single sample
            for k, want in io["masks"].items():
                assert bool(masks[k][row]) == bool(want[i])
    assert n_seen == 8


# ------------------------------------------------------------------------------------------------ the training script
@pytest.mark.parametrize("variant,graph", [("mca", False), ("mca", True), ("eao", False)])
def test_train_script_end_to_end(P, tmp_path, variant, graph):
    """train_accel_gpu.py <yaml> --synthetic N (the reference's entry point, train_accel_gpu.py:1-185) as a subprocess: YAML ->
    model (MCA or EAO) -> N optimizer steps -> log + Accelerate-layout state directory.  The replayed loop (--graph) logs the
    same first-step loss as the eager one (same seed, same synthetic batches) and every logged number is finite."""
    import json, subprocess, yaml
    cfg = small_config(variant)
    mod_cfg = {name: {"type": "embedded_sequence", "pad_len": enc["max_tokens"], "embedding_size": enc["input_size"], "data_col_name": "data", "dropout": 0.2}
               for name, enc in cfg["encoder_configs"].items()}

    def run(tag, extra):
        out = tmp_path / tag
        y = dict(encoder_configs=cfg["encoder_configs"], modality_config=mod_cfg, hidden_size=cfg["dim"], layers=cfg["depth"], heads=cfg["heads"],
                 dim_head=cfg["dim_head"], num_fusion_tokens=cfg["num_fusion_tokens"], batch_size=4, fcl=cfg["fcl"], fcl_root=cfg["fcl_root"],
                 bimodal_contrastive=cfg["bimodal_contrastive"], non_fusion_fcl=cfg["non_fusion_fcl"], fusion_combos=cfg["fusion_combos"],
                 zorro=cfg["zorro"], eao=cfg["eao"], no_fusion=cfg["no_fusion"], mean_pool=cfg["mean_pool"], predrop=True, epochs=1, lr=1e-3,
                 lr_scheduler_type="cosine", num_warmup_steps=2, clip=2.0, seed=7, output_dir=str(out), dataset="unused", run_eval_loop=False)
        ypath = tmp_path / f"{tag}.yaml"
        ypath.write_text(yaml.safe_dump(y, sort_keys=False))
        r = subprocess.run([sys.executable, os.path.join(REPO, "train_accel_gpu.py"), str(ypath), "--synthetic", "12"] + extra,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        recs = [json.loads(l) for l in open(out / "log.jsonl")]
        assert len(recs) >= 2 and recs[-1]["step"] == 12
        for rec in recs:
            assert all(v == v and abs(v) < 1e9 for k, v in rec.items() if isinstance(v, float)), rec
        assert os.path.exists(out / "0" / "model.safetensors") and os.path.exists(out / "0" / "optimizer.bin")
        return recs

    recs = run("eager", [])
    assert recs[0]["lr"] < recs[1]["lr"] or recs[0]["step"] > 2          # warm-up: the learning rate comes from the schedule
    if graph:
        recs_g = run("graph", ["--graph"])
        assert abs(recs_g[0]["total_loss"] - recs[0]["total_loss"]) <= 1e-4 * abs(recs[0]["total_loss"])
        assert abs(recs_g[-1]["total_loss"] - recs[-1]["total_loss"]) <= 5e-2 * abs(recs[-1]["total_loss"])
    if variant == "eao":
        assert any(k.startswith("audio_") or k.endswith("_audio") for k in recs[0])          # the pairwise terms of the EAO loss


def test_train_script_real_data_with_eval_loop(P, tmp_path):
    """train_accel_gpu.py <yaml> WITHOUT --synthetic (train_accel_gpu.py:31-39,70-71,137-181 of the reference): HF dataset on
    disk -> setup_data with predrop -> collators -> DataLoader (8 workers) -> prefetcher -> 2 epochs -> eval loop.  The logged
    val_epoch_* numbers of the LAST epoch are recomputed by the oracle (fp32 CPU restatement of the reference) + the metric
    formulas on the same eval split with the weights the script saved; tolerance 1e-2 relative (bf16 kernels vs fp32)."""
    import yaml
    from oracle import mca_oracle as O
    from torch.utils.data import DataLoader
    cfg = small_config("mca")
    ds_path = str(tmp_path / "ds")
    _ragged_dataset(ds_path, cfg)
    mod_cfg = {name: {"type": "embedded_sequence", "pad_len": enc["max_tokens"], "embedding_size": enc["input_size"],
                      "data_col_name": "data", "dropout": 0.3 if name == "video" else 0.0}
               for name, enc in cfg["encoder_configs"].items()}
    out = tmp_path / "out"
    y = dict(encoder_configs=cfg["encoder_configs"], modality_config=mod_cfg, hidden_size=cfg["dim"], layers=cfg["depth"], heads=cfg["heads"],
             dim_head=cfg["dim_head"], num_fusion_tokens=cfg["num_fusion_tokens"], batch_size=4, fcl=cfg["fcl"], fcl_root=cfg["fcl_root"],
             bimodal_contrastive=cfg["bimodal_contrastive"], non_fusion_fcl=cfg["non_fusion_fcl"], fusion_combos=cfg["fusion_combos"],
             zorro=cfg["zorro"], eao=False, no_fusion=False, mean_pool=False, predrop=True, epochs=2, lr=1e-3, lr_scheduler_type="cosine",
             num_warmup_steps=2, clip=2.0, seed=11, output_dir=str(out), dataset=ds_path, split=0.25, ds_seed=42, run_eval_loop=True)
    ypath = tmp_path / "train.yaml"
    ypath.write_text(yaml.safe_dump(y, sort_keys=False))
    r = subprocess.run([sys.executable, os.path.join(REPO, "train_accel_gpu.py"), str(ypath)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    recs = [json.loads(l) for l in open(out / "log.jsonl")]
    evals = [rec for rec in recs if "val_epoch_total_loss" in rec]
    steps = [rec for rec in recs if "step" in rec]
    assert [e["epoch"] for e in evals] == [0, 1] and steps[-1]["step"] == 2 * (18 // 4)          # 18 train samples, drop_last
    # the reference logs EVERY step (train_accel_gpu.py:126-130): total_loss, the loss terms, param_norm, grad_norm, lr ...
    assert [rec["step"] for rec in steps] == list(range(1, 9))
    for rec in steps:
        assert rec["param_norm"] > 0 and rec["lr"] >= 0 and 0 < rec["grad_norm"] <= rec["grad_norm_unclipped"] + 1e-6
        assert rec["grad_norm"] <= 2.0 * (1 + 1e-5)          # read after clip_grad_norm_(2.0) in the reference: the clipped norm
    # ... and every eval batch (:163-164): two batches per epoch here
    vsteps = [rec for rec in recs if "val_step_total_loss" in rec]
    assert [v["epoch"] for v in vsteps] == [0, 0, 1, 1] and all(any(k.startswith("val_step_") and k != "val_step_total_loss" for k in v) for v in vsteps)
    assert abs(sum(v["val_step_total_loss"] for v in vsteps[2:]) / 2 - evals[-1]["val_epoch_total_loss"]) < 1e-4 * abs(evals[-1]["val_epoch_total_loss"])
    got = evals[-1]
    # ---- the same eval split, rebuilt the way the script built it (same seeds -> same predrop draws and split)
    torch.manual_seed(11)
    ds = P.data.setup_data(ds_path, split=0.25, ds_frac=1.0, ds_seed=42, predrop=True, predrop_config=mod_cfg)
    eval_batches = list(DataLoader(ds["test"], collate_fn=P.MultimodalCollator(mod_cfg), batch_size=4))
    assert [next(iter(b.values()))["tokens"].shape[0] for b in eval_batches] == [4, 2]          # the partial batch is kept
    assert any(bool(b["video"]["attention_mask"].all(1).any()) for b in eval_batches + list(DataLoader(ds["train"], collate_fn=P.MultimodalCollator(mod_cfg), batch_size=18)))
    sd = {k: (v.float() if v.is_floating_point() else v) for k, v in P.checkpoint._read(str(out)).items()}
    S = O.Structure(copy.deepcopy(cfg))
    names = S.modalities
    uni = {k: P.metrics.Uniformity() for k in names + ["fusion"]}
    ali = {k: P.metrics.Alignment() for k in names}
    sums, logit_mag = {}, 0.0
    for b in eval_batches:
        o = O.mca_forward(S, sd, b, mode="fp32")
        pd = o["pooled"].double()          # a loss term is a difference of logits T * a.b: its error scales with their magnitude
        logit_mag = max(logit_mag, float(torch.exp(sd["loss.loss_fn.logit_scale"].double())) *
                        max(float((pd[:, i] @ pd[:, j].t()).abs().max()) for i in range(pd.shape[1]) for j in range(pd.shape[1])))
        sums["total_loss"] = sums.get("total_loss", 0.0) + float(o["loss"])
        for k, v in o["losses"].items():
            sums[k] = sums.get(k, 0.0) + float(v)
        for k in names:
            sm = o["modality_sample_mask"][k]
            uni[k].update(o[k][sm]); ali[k].update(o[k][sm], o["fusion"][sm])
        uni["fusion"].update(o["fusion"])
    want = {f"val_epoch_{k}": v / len(eval_batches) for k, v in sums.items() if "|" not in k}
    for tag, norm in (("", False), ("norm_", True)):
        want.update({f"val_epoch_{tag}uniformity_{k}": float(v.compute(norm=norm)) for k, v in uni.items()})
        want.update({f"val_epoch_{tag}alignment_{k}": float(v.compute(norm=norm)) for k, v in ali.items()})
    assert "val_epoch_total_loss" in want and "val_epoch_alignment_audio" in want and "val_epoch_norm_uniformity_fusion" in want
    for k, w in want.items():
        assert k in got, k
        if w != w:                      # a NaN loss term (no valid pair in any eval batch) is NaN on both sides
            assert got[k] != got[k], k
            continue
        # embedding metrics and the total: 1e-2 relative; single loss terms: 1e-2 relative or the bf16 error of a logit
        # (1e-4 x the largest logit, the bound tests/test_step_gpu.py uses), whichever is larger
        tol = 1e-2 * abs(w) + 1e-3
        if "uniformity" not in k and "alignment" not in k:
            tol = max(tol, 1e-4 * logit_mag)
        assert abs(got[k] - w) <= tol, (k, got[k], w, tol)
    for k in ("val_epoch_unformity_avg", "val_epoch_alignment_avg", "val_epoch_norm_unformity_avg", "val_epoch_norm_alignment_avg"):
        assert k in got and got[k] == got[k]
    assert os.path.exists(out / "1" / "model.safetensors") and os.path.exists(out / "model.safetensors")
    # the saved state carries the NEXT step's learning rate (torch / Accelerate convention), not the last one used
    ob = torch.load(out / "1" / "optimizer.bin", weights_only=True)
    sb = torch.load(out / "1" / "scheduler.bin", weights_only=True)
    assert sb["last_epoch"] == 8 and abs(ob["param_groups"][0]["lr"] - sb["_last_lr"][0]) < 1e-12
    assert ob["param_groups"][0]["lr"] <= steps[-1]["lr"]          # cosine decay after the 2-step warm-up


def test_train_script_eval_loop_under_data_parallelism(P, tmp_path):
    """The evaluation loop with TWO ranks (gloo, both on this GPU; RCCL on a node): the reference's prepared eval loader is
    sharded (train_accel_gpu.py:71,93 -> Accelerate's BatchSamplerShard: batches dealt round-robin, the last round completed from
    the start of the set), every forward all-gathers the embeddings, rank 0 logs ITS losses, the torchmetrics states gather every
    rank's rows.  Rank 0's logged val_epoch_* must equal the oracle's global-batch objective of rank 0's rows on those very
    batches + the metric formulas over both ranks' rows (until round 3 every rank evaluated the whole set and the all-gather
    gave each sample W - 1 duplicates as negatives: +ln W on every term)."""
    import yaml
    from oracle import mca_oracle as O
    cfg = small_config("mca")
    ds_path = str(tmp_path / "ds")
    _ragged_dataset(ds_path, cfg, n=28)
    mod_cfg = {name: {"type": "embedded_sequence", "pad_len": enc["max_tokens"], "embedding_size": enc["input_size"], "data_col_name": "data",
                      "dropout": 0.0} for name, enc in cfg["encoder_configs"].items()}
    out = tmp_path / "out"
    y = dict(encoder_configs=cfg["encoder_configs"], modality_config=mod_cfg, hidden_size=cfg["dim"], layers=cfg["depth"], heads=cfg["heads"],
             dim_head=cfg["dim_head"], num_fusion_tokens=cfg["num_fusion_tokens"], batch_size=2, fcl=cfg["fcl"], fcl_root=cfg["fcl_root"],
             bimodal_contrastive=cfg["bimodal_contrastive"], non_fusion_fcl=cfg["non_fusion_fcl"], fusion_combos=cfg["fusion_combos"],
             zorro=cfg["zorro"], eao=False, no_fusion=False, mean_pool=False, predrop=False, epochs=1, lr=1e-3, lr_scheduler_type="cosine",
             num_warmup_steps=2, clip=2.0, seed=11, output_dir=str(out), dataset=ds_path, split=0.25, ds_seed=42, run_eval_loop=True)
    ypath = tmp_path / "train.yaml"
    ypath.write_text(yaml.safe_dump(y, sort_keys=False))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MCA_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "train_accel_gpu.py"), str(ypath)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    recs = [json.loads(l) for l in open(out / "log.jsonl")]
    got = [rec for rec in recs if "val_epoch_total_loss" in rec][-1]
    # ---- the same eval set dealt the same way: 7 test samples, batch 2 -> 4 batches -> 2 rounds; the last round's second
    # batch is [6, 0]: completed from the start
    torch.manual_seed(11)
    ds = P.data.setup_data(ds_path, split=0.25, ds_frac=1.0, ds_seed=42, predrop=False, predrop_config=mod_cfg)
    n_test = len(ds["test"])
    shards = [P.data.shard_eval_batches(n_test, 2, 2, r) for r in range(2)]
    assert n_test == 7 and shards[0] == [[0, 1], [4, 5]] and shards[1] == [[2, 3], [6, 0]]
    collate = P.MultimodalCollator(mod_cfg)
    sd = {k: (v.float() if v.is_floating_point() else v) for k, v in P.checkpoint._read(str(out)).items()}
    S = O.Structure(copy.deepcopy(cfg))
    names = S.modalities
    Pr = O.Prec("fp32")
    rows = {r: {k: [] for k in names + ["fusion"]} for r in range(2)}
    pairs = {r: {k: ([], []) for k in names} for r in range(2)}
    sums, logit_mag = {}, 0.0
    for k_round in range(2):
        batch = collate([ds["test"][i] for r in range(2) for i in shards[r][k_round]])          # rank-major global batch
        tokens, padding, sample_mask = O.encode_and_pack(S, sd, batch, Pr)
        pooled = O.mca_trunk(S, sd, tokens, padding, Pr)
        pd = pooled.double()
        logit_mag = max(logit_mag, float(torch.exp(sd["loss.loss_fn.logit_scale"].double())) *
                        max(float((pd[:, i] @ pd[:, j].t()).abs().max()) for i in range(pd.shape[1]) for j in range(pd.shape[1])))
        for r in range(2):
            sm = {n: sample_mask[n][2 * r:2 * r + 2] for n in names}
            o = O.pretraining_loss(S, pooled[2 * r:2 * r + 2], sm, sd["loss.loss_fn.logit_scale"], pooled_all=pooled, rank=r)
            if r == 0:          # the main process logs its own losses
                sums["total_loss"] = sums.get("total_loss", 0.0) + float(o["loss"])
                for kk, v in o["losses"].items():
                    sums[kk] = sums.get(kk, 0.0) + float(v)
            for n in names:
                rows[r][n].append(o[n][sm[n]]); pairs[r][n][0].append(o[n][sm[n]]); pairs[r][n][1].append(o["fusion"][sm[n]])
            rows[r]["fusion"].append(o["fusion"])
    want = {f"val_epoch_{k}": v / 2 for k, v in sums.items() if "|" not in k}
    cat = lambda parts: torch.cat([torch.cat(parts[r]) for r in range(2)])          # rank 0's rows, then rank 1's
    for tag, norm in (("", False), ("norm_", True)):
        for k in names + ["fusion"]:
            want[f"val_epoch_{tag}uniformity_{k}"] = float(P.metrics.lunif(cat({r: rows[r][k] for r in range(2)}), 2, norm))
        for k in names:
            want[f"val_epoch_{tag}alignment_{k}"] = float(P.metrics.lalign(cat({r: pairs[r][k][0] for r in range(2)}), cat({r: pairs[r][k][1] for r in range(2)}), 2, norm))
    for k, w in want.items():
        assert k in got, k
        if w != w:
            assert got[k] != got[k], k
            continue
        tol = 1e-2 * abs(w) + 1e-3
        if "uniformity" not in k and "alignment" not in k:
            tol = max(tol, 1e-4 * logit_mag)
        assert abs(got[k] - w) <= tol, (k, got[k], w, tol)
    # what the unsharded loop of round 3 would have logged is far outside that tolerance: every term carried ~ln 2 more
    assert want["val_epoch_total_loss"] == want["val_epoch_total_loss"]


@pytest.mark.parametrize("world", [2, 4])
def test_bench_ranks_report_their_launch_choice_and_collectives(world):
    """`bench.py --gpus 2 | 4 --batch 8 --steps 3` as the driver starts it (one process per rank, RANK / WORLD_SIZE / MASTER_* from the
    environment), with MCA_DIST_BACKEND=gloo so that the ranks can share this box's one GPU (four stay within its process limit): the JSON line carries the rank count
    and backend (config.collectives), what the launch guard measured and chose (config.launch_choice: the segmented replay is
    kept only if its two guard steps are not slower than two eager ones, max over ranks) and, when the replay is kept, the
    segment count (config.launch).  On RCCL over xGMI the same code path runs with backend 'nccl'."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MCA_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(world), "--batch", "8", "--steps", "3", "--warmup", "1",
                                       "--no-kernel-timing"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=REPO))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for o in outs[1:] for l in o[0].splitlines() if l.startswith("{")]          # ONE line, from rank 0
    rec = json.loads(lines[0])
    cfg = rec["config"]
    assert rec["n_gpus"] == world and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["value"] > 0
    assert cfg["per_gpu_batch"] == 8 and cfg["global_batch"] == 8 * world and cfg["parallelism"] == f"dp{world}"
    assert cfg["collectives"] == f"gloo over {world} ranks"
    ch = cfg["launch_choice"]
    assert ch["requested"] == "auto" and ch["chosen"] in ("graph", "eager") and ch["reason"]
    assert set(ch["guard_ms_per_step"]) == {"replay", "eager"} and all(v > 0 for v in ch["guard_ms_per_step"].values())
    if ch["chosen"] == "graph":
        # forward | all-gather | loss + pooling backward | 7 buckets ... | finite flag: segments and eager collectives alternate
        assert "graph segments cut at" in cfg["launch"] and ch["guard_ms_per_step"]["replay"] <= 1.05 * ch["guard_ms_per_step"]["eager"]
    else:
        assert cfg["launch"] == "eager" and ch["guard_ms_per_step"]["replay"] > 1.05 * ch["guard_ms_per_step"]["eager"]
    # value = samples of ALL ranks / the slowest rank's time
    assert abs(rec["value"] - 8 * world * 3 / (rec["ms_per_step"] * 3e-3)) <= 1e-2 * rec["value"]


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
