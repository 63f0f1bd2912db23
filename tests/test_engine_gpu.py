"""GPU tests of the engine's API behaviour: device-side finite flag, weight-copy invalidation, foreign encoders, the workspace guard,
reference-written state directories (inference + resume), BASELINE configs[4] at size, the input prefetcher, logged norms."""
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


def _pooled(model, out):
    by_slot = {}
    for k, sl in model.output_slots().items():
        by_slot.setdefault(sl, k)
    return torch.stack([out[by_slot[sl]] for sl in sorted(by_slot)], 1)


# ------------------------------------------------------------------------------------------------ reference state dir
def _load_ref_state(P):
    io = torch.load(os.path.join(GOLDEN, "ref_state_io.pt"), weights_only=False)
    optim = importlib.import_module("mca-paper_amd.optim")
    model = P.MCA(**copy.deepcopy(io["config"])).cuda()
    opt = optim.FusedAdamW(model, lr=1e-3)
    meta = P.checkpoint.load_state(os.path.join(GOLDEN, "ref_state"), model, opt)
    return io, model, opt, meta


# ------------------------------------------------------------------------------------------------ finite flag
def test_finite_flag_sync_and_deferred(P):
    """encoders.py:197-213: non-finite encoder inputs raise.  Default mode raises inside the forward (one host read);
    'deferred' (train_accel_gpu.py / bench.py) raises at poll / assert time and the fused AdamW leaves the weights alone."""
    optim = importlib.import_module("mca-paper_amd.optim")
    cfg = small_config("tab")
    torch.manual_seed(0)
    model = P.MCA(**copy.deepcopy(cfg)).cuda()
    eng = model.engine
    opt = optim.FusedAdamW(model, lr=1e-2)
    good = to_device(P.data.synthetic_batch(cfg, 4, seed=2), "cuda")
    bad = copy.deepcopy(good)
    bad["audio"]["tokens"][1, 3, 2] = float("inf")
    with pytest.raises(Exception, match="not finite"):
        model(bad)
    out = model(good)                                   # the flag was cleared by the raise: a good batch passes
    assert torch.isfinite(out["loss"])
    bad2 = copy.deepcopy(good)
    bad2["video"]["values"][0, 1] = float("nan")        # tabular values are checked too
    with pytest.raises(Exception, match="not finite"):
        model(bad2)
    # deferred: nothing raises inside the step, the optimizer step is skipped on the device, the poll raises afterwards
    eng.check_finite = "deferred"
    w0 = eng.flat.clone()
    out = model(bad)
    opt.zero_grad(); out["loss"].backward(); optim.clip_grad_norm_(model, 2.0); opt.step()
    torch.cuda.synchronize()
    assert torch.equal(eng.flat, w0), "a flagged step reached the weights"
    with pytest.raises(Exception, match="not finite"):
        eng.assert_finite()
    out = model(good)
    opt.zero_grad(); out["loss"].backward(); optim.clip_grad_norm_(model, 2.0); opt.step()
    eng.assert_finite()
    assert not torch.equal(eng.flat, w0)


# ------------------------------------------------------------------------------------------------ weight copies
def test_weight_copies_follow_load_state_dict_and_foreign_optimizers(P):
    """the bf16 GEMM-weight copies were keyed on the flat buffer's version only, so load_state_dict after a first
    forward left the GEMMs on the old weights."""
    cfg = small_config("mca")
    batch = to_device(P.data.synthetic_batch(cfg, 4, seed=2, p_drop=0.2), "cuda")
    sd_a, sd_b = P.params.init_state_dict(cfg, seed=3), P.params.init_state_dict(cfg, seed=4)
    fresh = P.MCA(**copy.deepcopy(cfg)); fresh.load_state_dict(sd_b, strict=False); fresh = fresh.cuda()
    with torch.no_grad():
        want = _pooled(fresh, fresh(batch)).clone()
    m = P.MCA(**copy.deepcopy(cfg)); m.load_state_dict(sd_a, strict=False); m = m.cuda()
    with torch.no_grad():
        first = _pooled(m, m(batch)).clone()
        m.load_state_dict({k: v.cuda() for k, v in sd_b.items()}, strict=False)
        got = _pooled(m, m(batch))
    assert not torch.equal(first, want)
    assert torch.equal(got, want), rel_err(got, want)
    # a torch optimizer writes the parameters in place, without touching the flat buffer's version counter
    opt = torch.optim.SGD(m.parameters(), lr=0.5)
    out = m(batch); out["loss"].backward(); opt.step()
    with torch.no_grad():
        after = _pooled(m, m(batch))
    ref = P.MCA(**copy.deepcopy(cfg)); ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()}, strict=False); ref = ref.cuda()
    with torch.no_grad():
        want2 = _pooled(ref, ref(batch))
    assert torch.equal(after, want2)


# ------------------------------------------------------------------------------------------------ foreign encoder
def test_foreign_torch_encoder_receives_gradients(P):
    """A user-registered nn.Module encoder (not a NativeEncoder) runs under autograd and feeds the native trunk; its
    gradients land in the flat buffer (they were overwritten by zeros).  Yardstick: the same weights through the
    native encoder kernels."""
    from torch import nn
    encs = importlib.import_module("mca-paper_amd.encoders")

    class TorchSeqEncoder(nn.Module):
        def __init__(self, input_size=128, embedding_dim=512, max_tokens=1024, dropout=0.0, **kwargs):
            super().__init__()
            self.input_size, self.embedding_dim, self.max_tokens = input_size, embedding_dim, max_tokens
            self.token_encoder = nn.Sequential(nn.LayerNorm(input_size), nn.Linear(input_size, embedding_dim), nn.LayerNorm(embedding_dim))
            self.positional_encoder = encs.PositionalEncoder(embedding_dim, dropout, max_tokens)

        def forward(self, batch):
            m = batch["attention_mask"].bool()
            x = self.token_encoder(batch["tokens"].masked_fill(m[..., None], 0.0)).masked_fill(m[..., None], 0.0)
            return x + self.positional_encoder.pe[: x.shape[1]], batch["attention_mask"]

    P.encoders_dict["TorchSeqEncoder"] = TorchSeqEncoder
    try:
        cfg = small_config("mca")
        cfg_f = copy.deepcopy(cfg); cfg_f["encoder_configs"]["text"]["type"] = "TorchSeqEncoder"
        sd = P.params.init_state_dict(cfg, seed=3)
        batch = to_device(P.data.synthetic_batch(cfg, 4, seed=2, p_drop=0.2), "cuda")
        grads = []
        for c in (cfg, cfg_f):
            m = P.MCA(**copy.deepcopy(c)); m.load_state_dict(sd, strict=False); m = m.cuda()
            optim = importlib.import_module("mca-paper_amd.optim")
            opt = optim.FusedAdamW(m, lr=1e-3)
            out = m(batch); opt.zero_grad(); out["loss"].backward()
            torch.cuda.synchronize()
            grads.append({n: p.grad.detach().clone() for n, p in m.named_parameters()})
            for n, p in m.named_parameters():
                assert p.grad.data_ptr() == m.engine.grad_of(p).data_ptr(), n          # .grad IS the flat view
        nat, frn = grads
        for n in nat:
            if n.startswith("encoders.text."):
                assert float(frn[n].abs().max()) > 0, f"{n}: no gradient reached the foreign encoder"
            # the foreign encoder computes in fp32 torch, the native one with bf16 GEMM operands: the temperature-14 loss turns
            # that into percent-level differences on every gradient (observed up to 3.3 %); zero / garbage would be O(1)
            assert rel_err(frn[n], nat[n]) < 8e-2, (n, rel_err(frn[n], nat[n]))
    finally:
        P.encoders_dict.pop("TorchSeqEncoder", None)


def test_backward_after_another_forward_raises(P):
    cfg = small_config("mca")
    m = P.MCA(**copy.deepcopy(cfg)).cuda()
    b1 = to_device(P.data.synthetic_batch(cfg, 4, seed=2), "cuda")
    b2 = to_device(P.data.synthetic_batch(cfg, 4, seed=3), "cuda")
    o1 = m(b1)
    with torch.no_grad():
        m(b2)                                   # an eval forward of the same batch size reuses the workspace
    with pytest.raises(RuntimeError, match="another forward"):
        o1["loss"].backward()
    o2 = m(b2); o2["loss"].backward()           # the normal order still works


def test_reference_state_dir_inference(P):
    """a state directory written by the REFERENCE (accelerator.save_state layout) loads natively and the
    eval forward reproduces the embeddings / masks the reference's inference loop produced from it."""
    io, model, opt, meta = _load_ref_state(P)
    assert meta["scheduler_last_epoch"] == 2 and opt.step_count == 2
    model.eval()
    with torch.no_grad():
        out = model(to_device(io["eval_batch"], "cuda"))
    got_all, want_all = [], []
    for k, want in io["embeddings"].items():
        key = frozenset(int(x) for x in k.split("|")) if "|" in k else k
        # one (4, 128) slot at a time the bf16 noise of this small model scatters around the 1e-3 of the whole block
        assert rel_err(out[key].cpu(), want) < 2e-3, (k, rel_err(out[key].cpu(), want))
        got_all.append(out[key].cpu()); want_all.append(want)
    assert rel_err(torch.cat(got_all, 1), torch.cat(want_all, 1)) < 1e-3          # north_star: outputs within 1e-3 rel
    for k, want in io["masks"].items():
        assert torch.equal(out["modality_sample_mask"][k].cpu(), want)


def test_reference_state_dir_resume_third_step(P):
    """Resume: the reference's AdamW moments (optimizer.bin) and step count are taken over, so the native third step moves
    the weights as the reference's own third step did."""
    optim = importlib.import_module("mca-paper_amd.optim")
    io, model, opt, meta = _load_ref_state(P)
    before = {n: p.detach().clone().cpu() for n, p in model.named_parameters()}
    opt.param_groups[0]["lr"] = io["lr_step3"]
    out = model(to_device(io["train_batches"][2], "cuda"))
    opt.zero_grad(); out["loss"].backward(); optim.clip_grad_norm_(model, 2.0); opt.step()
    torch.cuda.synchronize()
    # the loss is a difference of temperature-scaled logits of magnitude O(10^2..10^3) here: 0.5 % of its value is ~1e-4 of them
    assert abs(float(out["loss"]) - float(io["loss_step3"])) < 5e-3 * abs(float(io["loss_step3"])) + 1e-3
    worst = 0.0
    for n, p in model.named_parameters():
        d_ref, d_nat = io["state_step3"][n] - before[n], p.detach().cpu() - before[n]
        if float(d_ref.abs().max()) < 1e-9:
            continue
        worst = max(worst, rel_err(d_nat, d_ref))
        # with zeroed moments the third update would be ~lr*sign(g) (several times larger): the restored moments matter
        assert rel_err(d_nat, d_ref) < 0.15, (n, rel_err(d_nat, d_ref))
    assert worst > 0


# ------------------------------------------------------------------------------------------------ BASELINE config 5 at size
def test_long_config_at_batch_128(P):
    """BASELINE configs[4]: 4 x 1500 tokens + 88 fusion tokens (N = 6088), batch 128 on one GPU (779,264 tokens; ~135 GB of
    activations, every row offset beyond 2^31 bytes).  Size-independent properties: repeated forwards agree bit for bit,
    the fp8 form of the forward attention agrees with the bf16 form within the stated tolerance, the loss is finite, every
    parameter gets a finite, non-zero gradient, and the samples of the batch do not interact
    before the loss (rows 0..1 of the b = 128 pass equal a b = 2 pass of the same samples)."""
    optim = importlib.import_module("mca-paper_amd.optim")
    b = 128
    free, _ = torch.cuda.mem_get_info()
    if free < 180e9:
        pytest.skip("needs ~150 GB of free HBM")
    cfg = P.config.cmu_model_config(batch_size=b, long_seq=True)
    torch.manual_seed(43)
    model = P.MCA(**cfg).cuda()
    eng = model.engine
    assert eng.N == 6088
    opt = optim.FusedAdamW(model, lr=1e-4)
    batch = P.data.synthetic_batch(cfg, b, seed=1234, lengths="uniform", p_drop=0.2, device="cuda")
    with torch.no_grad():
        o1 = model(batch); p1 = eng.workspace(b)["pooled"].clone(); l1 = o1["loss"].clone()
        o2 = model(batch); p2 = eng.workspace(b)["pooled"].clone(); l2 = o2["loss"].clone()
    # nothing in the forward is order-dependent (mean(V) of rows with no valid key sums in a fixed order): bit for bit
    R = eng.R
    assert torch.equal(p1, p2) and torch.equal(l1, l2)
    # the fp8 form of the forward attention (configs[4] names it) at full size: STATED TOLERANCE 1e-2 rel-L2 on the pooled
    # embeddings against the bf16 form (5 layers of e4m3 operands; measured value printed), loss within 2 %
    eng.set_attention_dtype("fp8")
    with torch.no_grad():
        o8 = model(batch); p8 = eng.workspace(b)["pooled"].clone(); l8 = o8["loss"].clone()
    eng.set_attention_dtype("bf16")
    e8 = rel_err(p8, p1)
    print("long config: fp8 vs bf16 pooled rel-L2", e8, "loss", float(l8), float(l1))
    assert torch.isfinite(p8).all() and e8 < 1e-2 and abs(float(l8) - float(l1)) <= 2e-2 * abs(float(l1))
    out = model(batch)
    opt.zero_grad(); out["loss"].backward()
    torch.cuda.synchronize()
    assert torch.isfinite(out["loss"])
    for n, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), n
        if n != "return_tokens":
            assert float(p.grad.abs().max()) > 0, n
    g16 = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    del out
    # ---- configs[4] names fp8 attention: the TRAINING step in fp8 at this size (the three fp8 backward kernels and the backward
    # quantisation at b = 128: every row offset of their operands is beyond 2^31 bytes).  Stated tolerances against the bf16
    # step on the same weights and batch: gradient norm within 3 %, every tensor within 30 %, median within 5 % (e4m3 operands
    # in S and dP of five layers; measured: norm 0.13 %, median 2.4 %, worst 22 % on layers.0.attn.to_q.weight, the tensor the
    # noise of all five layers reaches); two fp8 steps give the same dq | dk | dv bits in the layer the backward reaches first
    eng.set_attention_dtype("fp8")
    ws = eng.workspace(b)
    assert eng.fp8_backward_on(ws, eng.N)
    runs = []
    for rep in range(2):
        o8 = model(batch)
        opt.zero_grad(); o8["loss"].backward()
        torch.cuda.synchronize()
        runs.append(ws["layers"][eng.L - 1]["dqkv"].clone())
        assert abs(float(o8["loss"]) - float(l8)) <= 1e-5 * abs(float(l8))          # the fp8 forward of above, again
        del o8
    assert torch.equal(runs[0], runs[1]), "fp8 attention backward at b = 128 is not bitwise repeatable"
    del runs
    errs = []
    for n, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), n
        if float(g16[n].abs().max()) == 0:
            continue
        errs.append((rel_err(p.grad, g16[n]), n))
    n16 = float(torch.sqrt(sum((g.double() ** 2).sum() for g in g16.values())))
    n8 = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters())))
    worst, med = max(errs), sorted(e for e, _ in errs)[len(errs) // 2]
    print("long config: fp8 vs bf16 training step: gradient norm", n8, n16, "worst tensor", worst, "median", med)
    assert abs(n8 - n16) <= 3e-2 * n16 and worst[0] < 0.30 and med < 0.05, (n8, n16, worst, med)
    del g16
    optim.clip_grad_norm_(model, 2.0); opt.step()          # the optimizer step on the fp8 gradients
    torch.cuda.synchronize()
    model.engine.assert_finite() if eng.check_finite == "deferred" else None
    eng.set_attention_dtype("bf16")
    # the same first two samples alone: identical pooled rows (nothing mixes samples before the loss)
    pb = p1.view(b, R, -1)[:2].clone()
    small = {k: {kk: vv[:2].contiguous() for kk, vv in v.items()} for k, v in batch.items()}
    torch.manual_seed(43)
    m2 = P.MCA(**P.config.cmu_model_config(batch_size=2, long_seq=True)).cuda()
    with torch.no_grad():
        m2(small)
    ps = m2.engine.workspace(2)["pooled"].view(2, R, -1)
    pr2 = m2.engine.workspace(2)["present_cur"]
    for i in range(2):
        assert torch.equal(ps[i], pb[i])


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
