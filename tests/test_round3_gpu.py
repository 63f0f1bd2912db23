"""Round-3 GPU tests: the real-data training path with the evaluation loop, the input prefetcher, bf16
weight copies after graph replays, and the data-parallel step replayed as graph segments cut at the collectives."""
import copy
import importlib
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, REPO
from util_small import small_config, rel_err, to_device

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


# ------------------------------------------------------------------------------------------------ input pipeline
def test_device_prefetcher_keeps_order_and_contents(P):
    """data.DevicePrefetcher: every batch arrives on the device, in order, bit for bit, while later batches are already being
    copied; a batch of another shape (the last partial one) passes through; buffers are recycled only after their consumer
    came back."""
    g = torch.Generator().manual_seed(0)
    host = [{"a": {"tokens": torch.randn(4, 70, 10, generator=g), "attention_mask": torch.rand(4, 70, generator=g) > 0.5},
             "l": [torch.full((3,), float(i))]} for i in range(7)]
    host.append({"a": {"tokens": torch.randn(2, 70, 10, generator=g), "attention_mask": torch.rand(2, 70, generator=g) > 0.5},
                 "l": [torch.full((3,), 7.0)]})
    seen, held = 0, []
    for i, b in enumerate(P.data.DevicePrefetcher(iter(host), "cuda")):
        assert b["a"]["tokens"].is_cuda and b["l"][0].is_cuda
        # a consumer that is slow on the GPU: the buffers of batch i are read by a kernel enqueued now and must not be
        # overwritten by the copy of batch i + 2 before that kernel has run
        torch.cuda._sleep(20_000_000)
        held.append((b["a"]["tokens"].double().sum(), b["a"]["attention_mask"].sum(), b["l"][0][0].clone()))
        seen += 1
    assert seen == len(host)
    torch.cuda.synchronize()
    for i, (s, m, l) in enumerate(held):
        assert float(l) == float(i)
        assert float(s) == float(host[i]["a"]["tokens"].double().sum()) and int(m) == int(host[i]["a"]["attention_mask"].sum())


# ------------------------------------------------------------------------------------------------ weights after replay
def test_eval_after_graph_replays_uses_current_weights(P):
    """GraphedStep replays move the fp32 weights behind torch's version counters; an eval forward between replays must rebuild
    the bf16 GEMM-weight copies: replay, eval, replay, eval == eval after refresh_weights(force=True)."""
    optim = importlib.import_module("mca-paper_amd.optim")
    graph = importlib.import_module("mca-paper_amd.graph")
    cfg = small_config("mca")
    sd = P.params.init_state_dict(cfg, seed=3)
    model = P.build_model(copy.deepcopy(cfg)); model.load_state_dict(sd, strict=False); model = model.cuda()
    opt = optim.FusedAdamW(model, lr=5e-2)          # a large rate: one stale step is far outside the tolerance
    batch = to_device(P.data.synthetic_batch(cfg, 4, seed=5, p_drop=0.2), "cuda")
    g = graph.GraphedStep(model, opt, batch, clip=2.0)

    def eval_pooled():
        model.eval()
        with torch.no_grad():
            o = model(batch, no_loss=True)
        model.train()
        return torch.stack([o[k] for k in model.modality_types], 1).clone()

    for _ in range(2):
        g.step(batch)
        got = eval_pooled()
        model.engine.refresh_weights(force=True)
        want = eval_pooled()
        assert torch.equal(got, want)
    g.step(batch)
    again = eval_pooled()
    assert rel_err(again, want) > 1e-3          # the weights did move: the check above is not vacuous


# ------------------------------------------------------------------------------------------------ DP: graph segments
def _seg_worker(rank, world, port, out, backend, always):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = importlib.import_module("mca-paper_amd")
        dpm = importlib.import_module("mca-paper_amd.dp")
        optim = importlib.import_module("mca-paper_amd.optim")
        graph = importlib.import_module("mca-paper_amd.graph")
        cfg = small_config("mca")
        b = 4
        sd = P.params.init_state_dict(cfg, seed=3)
        batches = []
        for i in range(4):
            full = P.data.synthetic_batch(cfg, b * world, seed=21 + i, p_drop=0.3)
            batches.append(to_device({k: {kk: vv[rank * b:(rank + 1) * b] for kk, vv in v.items()} for k, v in full.items()}, "cuda"))
        res = {}
        for mode in ("eager", "segments"):
            model = P.MCA(**copy.deepcopy(cfg)); model.load_state_dict(sd, strict=False); model = model.cuda()
            model.engine.check_finite = "deferred"
            opt = optim.FusedAdamW(model, lr=1e-3)
            dp = dpm.DataParallelMCA(model, always_collect=always)
            hist = []
            if mode == "eager":
                for bt in batches:
                    o = dp(bt); opt.zero_grad(); o["loss"].backward(); dp.finish_backward()
                    gn = optim.clip_grad_norm_(model, 2.0); opt.step()
                    hist.append((float(o["loss"]), float(gn), torch.stack([o[k] for k in model.modality_types], 1).detach().clone().cpu(),
                                 model.engine.gflat.clone().cpu()))
            else:
                g = graph.GraphedStep(model, opt, batches[0], clip=2.0, dp=dp)
                n_graphs = sum(1 for it in g.program if isinstance(it, torch.cuda.CUDAGraph))
                n_coll = len(g.program) - n_graphs
                for bt in batches:
                    loss = g.step(bt)
                    hist.append((float(loss), float(g.gnorm), torch.stack([g.out[k] for k in model.modality_types], 1).detach().clone().cpu(),
                                 model.engine.gflat.clone().cpu()))
                res["shape"] = (n_graphs, n_coll)
            torch.cuda.synchronize()
            res[mode] = dict(hist=hist, flat=model.engine.flat.clone().cpu())
        torch.save(res, out + f".{rank}")
    finally:
        dist.destroy_process_group()


def _check_segments(res, world):
    n_graphs, n_coll = res["shape"]
    L = 2
    # forward | gather | loss+pool bwd | L layer buckets + encoders | wait -> 1 + (L + 2) + 1 collectives, one more graph than that
    assert n_coll == 1 + (L + 2) + 1 and n_graphs == n_coll + 1, res["shape"]
    errs = [(abs(le - ls) / abs(le), abs(ge - gs) / ge, rel_err(gfs, gfe), rel_err(ps, pe))
            for (le, ge, pe, gfe), (ls, gs, ps, gfs) in zip(res["eager"]["hist"], res["segments"]["hist"])]
    pe, ps = res["eager"]["hist"][0][2], res["segments"]["hist"][0][2]
    assert torch.equal(pe, ps), "first forward (same weights, same kernels) must agree bit for bit"
    # step 1: same weights, same inputs -> the gradients differ only by the order of fp32 atomic adds (side-stream weight
    # gradients in the eager loop, none in the replay); later steps start from weights that already differ by those roundings
    # (Adam's first update is lr * sign(g): a gradient element near zero may flip), so they are compared as trajectories
    assert errs[0][0] <= 1e-6 and errs[0][1] <= 1e-3 and errs[0][2] < 1e-2, errs
    for e in errs[1:]:
        assert e[0] <= 1e-2 and e[1] <= 5e-2 and e[3] < 1e-2, errs
    assert rel_err(res["segments"]["flat"], res["eager"]["flat"]) < 2e-3, errs


def test_dp_step_as_graph_segments_matches_eager_two_ranks(P, tmp_path):
    """Two ranks (gloo, both on this GPU; RCCL on a node) run four optimizer steps (a) in the eager data-parallel loop and (b) as
    graph segments cut at the collectives (graph.GraphedStep(dp=...)): first forward bit for bit, losses / gradient norms /
    all-reduced gradients of every step within 1e-2, both ranks identical."""
    W = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "seg.pt")
    mp.spawn(_seg_worker, args=(W, port, out, "gloo", False), nprocs=W, join=True)
    got = [torch.load(out + f".{r}", weights_only=False) for r in range(W)]
    for r in range(W):
        _check_segments(got[r], W)
    for mode in ("eager", "segments"):
        for i, (a, b) in enumerate(zip(got[0][mode]["hist"], got[1][mode]["hist"])):
            bad = (a[3] != b[3]).nonzero().flatten()          # the all-reduce left identical gradients on both ranks
            assert bad.numel() == 0, (mode, i, bad.numel(), bad[:8].tolist(), bad[-8:].tolist(), a[3][bad[:4]].tolist(), b[3][bad[:4]].tolist())
    # ... and, the gradient norm being summed in a fixed order (mca_grad_sqnorm), identical weights after four optimizer steps
    assert torch.equal(got[0]["segments"]["flat"], got[1]["segments"]["flat"]) and torch.equal(got[0]["eager"]["flat"], got[1]["eager"]["flat"])


def test_dp_graph_segments_with_rccl_collectives_world1(P, tmp_path):
    """The same segmented step with the REAL RCCL backend on a world of one rank (this pool has one GPU per box): the packed
    all-gather, the async bucket all-reduces and the finite-flag MAX are issued through torch.distributed 'nccl' between the
    replayed segments (always_collect)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "seg1.pt")
    mp.spawn(_seg_worker, args=(1, port, out, "nccl", True), nprocs=1, join=True)
    _check_segments(torch.load(out + ".0", weights_only=False), 1)
