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


def test_predrop_semantics(pkg):
    torch.manual_seed(0)
    apply = pkg.data.batch_predrop({"a": {"dropout": 1.0, "pad_token": -10000}, "b": {"dropout": 0.0}, "c": {}})
    s = apply({"a": {"data": torch.ones(3, 2)}, "b": {"data": torch.ones(3, 2)}, "c": {"data": torch.ones(1)}, "Labels": {"data": torch.zeros(7)}})
    assert s["a"] == {"data": None} and s["b"]["data"] is not None and s["c"]["data"] is not None
    coll = pkg.MultimodalCollator({"a": {"type": "embedded_sequence", "pad_len": 4, "embedding_size": 2, "data_col_name": "data"}})
    out = coll([s])
    assert out["a"]["attention_mask"].all() and out["a"]["tokens"].abs().sum() == 0          # dropped modality = all-pad row
