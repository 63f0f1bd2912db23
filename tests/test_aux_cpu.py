"""CPU tests of the next-tier pieces: reference-format checkpoints, eval metrics, modality pre-dropout."""
import os

import torch


def test_checkpoint_roundtrip_reference_layout(pkg, tmp_path):
    cfg = pkg.config.cmu_model_config(2)
    torch.manual_seed(1)
    m1 = pkg.MCA(**cfg)
    path = pkg.checkpoint.save_model(m1, str(tmp_path))
    assert os.path.basename(path) == "model.safetensors"
    from safetensors.torch import load_file
    sd = load_file(path)
    assert "layers.0.attn.to_kv.weight" in sd and "loss.loss_fn.logit_scale" in sd and "attn_mask" in sd     # reference key names
    torch.manual_seed(2)
    m2 = pkg.MCA(**cfg)
    assert not torch.equal(m1.layers[0].attn.to_q.weight, m2.layers[0].attn.to_q.weight)
    missing, unexpected = pkg.checkpoint.load_model(m2, str(tmp_path))
    assert not missing and not unexpected
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)
    # a checkpoint of another structure is refused
    m3 = pkg.MCA(**pkg.config.cmu_model_config(2, zorro=True))
    try:
        pkg.checkpoint.load_model(m3, str(tmp_path))
        raise AssertionError("should have refused")
    except (ValueError, KeyError):
        pass


def test_metrics_match_formulas(pkg):
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(40, 16, generator=g), torch.randn(40, 16, generator=g)
    al, un = pkg.metrics.Alignment(), pkg.metrics.Uniformity()
    for i in range(0, 40, 8):
        al.update(x[i:i + 8], y[i:i + 8]); un.update(x[i:i + 8])
    assert torch.allclose(al.compute(), (x - y).norm(dim=1).pow(2).mean())
    xn = torch.nn.functional.normalize(x)
    d2 = torch.cdist(xn, xn).pow(2)[torch.triu(torch.ones(40, 40), 1).bool()]
    assert torch.allclose(un.compute(norm=True), d2.mul(-2).exp().mean().log(), atol=1e-6)
    al.reset(); un.reset()
    assert al.preds == [] and un.preds == []


def test_eval_sharding_is_accelerates(pkg):
    """data.shard_eval_batches (the DP eval loader of train_accel_gpu.py) against accelerate.data_loader.BatchSamplerShard with the
    defaults accelerator.prepare uses (split_batches=False, even_batches=True) over a sequential, drop_last=False loader"""
    import pytest
    acc = pytest.importorskip("accelerate.data_loader")
    from torch.utils.data import BatchSampler, SequentialSampler
    for n in (1, 5, 7, 8, 9, 16, 17, 31, 33, 100):
        for b in (1, 2, 4, 8):
            for W in (1, 2, 3, 4, 8):
                per_rank = []
                for r in range(W):
                    want = list(acc.BatchSamplerShard(BatchSampler(SequentialSampler(range(n)), b, False), W, r))
                    got = pkg.data.shard_eval_batches(n, b, W, r)
                    assert got == want, (n, b, W, r)
                    per_rank.append(got)
                assert len({len(x) for x in per_rank}) == 1 and all(len(bt) == b for x in per_rank for bt in x)          # equal, full local batches
                assert set(range(n)) <= {i for x in per_rank for bt in x for i in bt}


def test_predrop_semantics(pkg):
    torch.manual_seed(0)
    apply = pkg.data.batch_predrop({"a": {"dropout": 1.0, "pad_token": -10000}, "b": {"dropout": 0.0}, "c": {}})
    s = apply({"a": {"data": torch.ones(3, 2)}, "b": {"data": torch.ones(3, 2)}, "c": {"data": torch.ones(1)}, "Labels": {"data": torch.zeros(7)}})
    assert s["a"] == {"data": None} and s["b"]["data"] is not None and s["c"]["data"] is not None
    coll = pkg.MultimodalCollator({"a": {"type": "embedded_sequence", "pad_len": 4, "embedding_size": 2, "data_col_name": "data"}})
    out = coll([s])
    assert out["a"]["attention_mask"].all() and out["a"]["tokens"].abs().sum() == 0          # dropped modality = all-pad row


def _hf_dataset(tmp_path, n=24):
    from datasets import Dataset
    g = torch.Generator().manual_seed(0)
    samples = []
    for i in range(n):
        k = int(torch.randint(1, 7, (1,), generator=g))
        samples.append({"audio": {"data": torch.randn(k, 3, generator=g).tolist()},
                        "tab": {"values": torch.randn(5, generator=g).tolist()},
                        "Labels": {"data": [float(i)] * 2}})
    path = str(tmp_path / "ds")
    Dataset.from_list(samples).save_to_disk(path)
    return path


MOD_CFG = {"audio": {"type": "embedded_sequence", "pad_len": 6, "embedding_size": 3, "data_col_name": "data", "dropout": 0.5},
           "tab": {"type": "sequence", "pad_len": 5, "data_col_name": "values", "pad_token": -10000, "dropout": 0.0}}


def test_setup_data_applies_predrop_before_the_split(pkg, tmp_path):
    """utils/dataset.py:72-84: load -> ds_frac prefix -> predrop (dataset.map) -> train_test_split; the dropped modality
    reaches the model as a fully padded row (the real-data path never applied predrop)."""
    from torch.utils.data import DataLoader
    path = _hf_dataset(tmp_path)
    torch.manual_seed(0)
    d = pkg.data.setup_data(path, split=0.25, ds_frac=0.5, ds_seed=42, predrop=True, predrop_config=MOD_CFG)
    assert len(d["train"]) + len(d["test"]) == 12                       # ds_frac took the first half
    coll = pkg.MultimodalCollator(MOD_CFG, labels="Labels")
    dropped = kept = 0
    for split in ("train", "test"):
        for b in DataLoader(d[split], collate_fn=coll, batch_size=3):
            allpad = b["audio"]["attention_mask"].all(1)
            dropped += int(allpad.sum()); kept += int((~allpad).sum())
            assert not b["tab"]["attention_mask"].bool().all(1).any()      # dropout 0.0: never dropped
    assert dropped > 0 and kept > 0                                      # p = 0.5 over 12 samples, both splits pre-dropped
    # the same call without predrop keeps everything
    d0 = pkg.data.setup_data(path, split=0.25, ds_frac=0.5, ds_seed=42, predrop=False, predrop_config=MOD_CFG)
    for b in DataLoader(d0["train"], collate_fn=coll, batch_size=3):
        assert not b["audio"]["attention_mask"].all(1).any()
    # predrop requested but not honourable -> loud failure
    import pytest
    with pytest.raises(ValueError):
        pkg.data.setup_data(path, predrop=True, predrop_config=None)
    with pytest.raises(KeyError):
        pkg.data.setup_data(path, predrop=True, predrop_config={"audio": {"type": "embedded_sequence"}})


def test_train_script_data_path_calls_setup_data():
    """the entry scripts route the real-data path through setup_data with the YAML's predrop / modality_config"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for script in ("train_accel_gpu.py", "infer_accel_gpu.py"):
        src = open(os.path.join(root, script)).read()
        assert re.search(r"setup_data\(config\.dataset.*predrop=bool\(config\.get\(\"predrop\"", src, re.S), script


def test_reference_written_state_dir_loads(pkg, golden_dir):
    """tests/golden/ref_state was written by the REFERENCE (oracle/make_goldens.py --ckpt: safetensors of its state_dict,
    torch AdamW / cosine scheduler state): every key matches the native module tree, values load bit-exactly."""
    from safetensors.torch import load_file
    io = torch.load(os.path.join(golden_dir, "ref_state_io.pt"), weights_only=False)
    m = pkg.MCA(**io["config"])
    missing, unexpected = pkg.checkpoint.load_model(m, os.path.join(golden_dir, "ref_state"), strict=True)
    assert not missing and not unexpected
    sd = load_file(os.path.join(golden_dir, "ref_state", "model.safetensors"))
    own = m.state_dict()
    for k, v in sd.items():
        if k in own:
            assert torch.equal(own[k], v), k
    opt = torch.load(os.path.join(golden_dir, "ref_state", "optimizer.bin"), weights_only=False)
    assert len(opt["state"]) == len(list(m.parameters()))                # state index i = i-th model.parameters()
    for i, p in enumerate(m.parameters()):
        assert tuple(opt["state"][i]["exp_avg"].shape) == tuple(p.shape)


def test_lr_schedules_match_transformers():
    """train_accel_gpu.py:81-86 uses transformers.get_scheduler; the native loop computes the same multipliers."""
    import importlib.util
    from transformers import get_scheduler
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_native", os.path.join(root, "train_accel_gpu.py"))
    src = open(os.path.join(root, "train_accel_gpu.py")).read()
    ns = {}
    exec(src[src.index("def lr_factor"):src.index("def cosine_with_warmup")], {"math": __import__("math")}, ns)
    lr_factor = ns["lr_factor"]
    for name in ("cosine", "constant_with_warmup", "linear", "constant"):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=1.0)
        sched = get_scheduler(name=name, optimizer=opt, num_warmup_steps=7, num_training_steps=40)
        for step in range(40):
            assert abs(opt.param_groups[0]["lr"] - lr_factor(name, step, 7, 40)) < 1e-6, (name, step)      # fp32 vs fp64
            opt.step(); sched.step()
    import pytest
    with pytest.raises(ValueError):
        lr_factor("polynomial_decay_typo", 0, 1, 2)


def test_every_reference_training_yaml_is_accepted(pkg, golden_dir, tmp_path):
    """the native loader takes each of the reference's 145 training YAMLs (key lists and structural values
    from oracle/make_yaml_census.py).  Every structurally distinct configuration constructs on the CPU."""
    import json
    import yaml
    census = json.load(open(os.path.join(golden_dir, "ref_yaml_census.json")))
    assert len(census) == 145
    import pytest
    seen = set()
    n_built = n_eao = 0
    refused = []
    for name, ent in sorted(census.items()):
        y = dict(ent["settings"]); y["encoder_configs"] = ent["encoder_configs"]; y["modality_config"] = ent["modality_config"]
        y["output_dir"] = str(tmp_path / "out")
        f = tmp_path / "c.yaml"
        f.write_text(yaml.safe_dump(y))
        cfg = pkg.config.training_config(str(f), make_output_dir=False)
        assert set(ent["keys"]) - {"dataset", "restart", "wandb_name"} <= set(cfg.keys()), name          # unknown keys are kept (yacs new_allowed)
        assert cfg.lr_scheduler_type in ("cosine", "constant_with_warmup")
        mc = pkg.config.get_model_config(cfg)
        sig = json.dumps({k: v for k, v in mc.items() if k != "batch_size"}, sort_keys=True, default=str)
        if sig in seen:
            continue
        seen.add(sig)
        # not native: MCA(mean_pool=True) (the `_j*` YAMLs), which the reference itself cannot run:
        # MeanTokenProjectionPool.forward evaluates `if self.token_types` on an N-element tensor (model.py:264; recorded from
        # a run of the reference in tests/golden/ref_unrunnable.json).  The EAO baseline (12 YAMLs) builds natively.
        if mc["mean_pool"] and not mc["eao"]:
            with pytest.raises(NotImplementedError):
                pkg.build_model(mc)
            refused.append(name)
            continue
        model = pkg.build_model(mc)
        if mc["eao"]:
            assert type(model).__name__ == "EAO" and model.structure.n_return == len(mc["encoder_configs"]) + len(model.fusion_combos)
            n_eao += 1
        else:
            assert model.structure.n_tokens == sum(e["max_tokens"] for e in mc["encoder_configs"].values()) + model.structure.num_fusion_tokens
        n_built += 1
    assert n_built >= 9 and n_eao >= 1 and len(refused) >= 1
    unrunnable = json.load(open(os.path.join(golden_dir, "ref_unrunnable.json")))
    assert unrunnable["MCA(mean_pool=True).forward"]["type"] == "RuntimeError"


class _RaggedSamples(torch.utils.data.Dataset):
    """{modality: {"data": (len, emb)}} samples with ragged lengths and a missing modality now and then (module level: workers pickle it)"""

    def __init__(self, n=12):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(100 + i)
        a = torch.randn(int(torch.randint(1, 40, (1,), generator=g)), 7, generator=g)
        b = None if i % 5 == 3 else torch.randn(int(torch.randint(1, 9, (1,), generator=g)), 3, generator=g)
        return {"audio": {"data": a}, "video": {"data": b}}


def test_collators_in_loader_workers_build_the_same_batch_in_shared_memory(pkg):
    """Inside a DataLoader worker the collators fill buffers allocated in shared memory (as torch's default_collate does), so the
    batch is not copied a second time on its way to the training loop; the values are those of the in-process call."""
    from torch.utils.data import DataLoader
    mod_cfg = {"audio": {"type": "embedded_sequence", "pad_len": 32, "embedding_size": 7, "data_col_name": "data"},
               "video": {"type": "embedded_sequence", "pad_len": 8, "embedding_size": 3, "data_col_name": "data"}}
    col = pkg.MultimodalCollator(mod_cfg)
    ds = _RaggedSamples()
    here = list(DataLoader(ds, collate_fn=col, batch_size=4, num_workers=0))
    there = list(DataLoader(ds, collate_fn=col, batch_size=4, num_workers=2))
    assert len(here) == len(there) == 3
    for a, b in zip(here, there):
        for m in mod_cfg:
            for k in ("tokens", "attention_mask"):
                assert a[m][k].dtype == b[m][k].dtype and torch.equal(a[m][k], b[m][k])
                assert b[m][k].is_shared() and not a[m][k].is_shared()
    assert bool(here[0]["video"]["attention_mask"][3].all()) and float(here[0]["video"]["tokens"][3].abs().max()) == 0.0          # the missing sample: an all-pad row
