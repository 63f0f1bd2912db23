"""GPU tests of the hipGraph step: replay against the eager step (small and CMU size), weights after replays, and the data-parallel
step replayed as graph segments cut at the collectives."""
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

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return importlib.import_module("mca-paper_amd")


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


def test_graphed_step_at_cmu_size_matches_eager(P):
    """The replayed step at the CMU shape (D = 512: fused LayerNorm-residual GEMMs, grouped weight gradients, mask product; the
    captured graph is ONE chain, no side stream) against the eager loop, four optimizer steps: loss and global gradient norm
    of every step.  This is the test that caught a `hipMemsetAsync` inside the captured region (the zeroing of dvmean in
    mca_attn_bwd_prep): as a memset node of a single-chain graph it was not ordered against the kernels around it, and from
    the second replay on the whole backward started from garbage while the loss still looked plausible."""
    optim = importlib.import_module("mca-paper_amd.optim")
    graph = importlib.import_module("mca-paper_amd.graph")
    cfg = P.config.cmu_model_config(batch_size=2)
    batch = P.data.synthetic_batch(cfg, 2, seed=1234, lengths="uniform", p_drop=0.3, device="cuda")
    # Both loops run in lockstep and every step STARTS FROM THE SAME STATE (the eager model's weights, moments and step count are
    # copied into the replayed one): the loss of this model is a difference of O(10^3) logits, so two free-running loops drift
    # apart by ~1 % within three steps from the order of fp32 atomic adds alone (the eager loop against itself does), which
    # would force a tolerance too wide to see a broken replay.
    models = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(43)
        m = P.MCA(**cfg).cuda(); m.engine.check_finite = "deferred"
        opt = optim.FusedAdamW(m, lr=1e-5)
        models[mode] = (m, opt)
    (me, oe), (mg, og) = models["eager"], models["graph"]
    g = graph.GraphedStep(mg, og, batch, clip=2.0)
    hist = []
    for step in range(4):
        mg.engine.flat.copy_(me.engine.flat); og.exp_avg.copy_(oe.exp_avg); og.exp_avg_sq.copy_(oe.exp_avg_sq); og.step_count = oe.step_count
        mg.engine.invalidate_weights()
        out = me(batch); oe.zero_grad(); out["loss"].backward(); gn = optim.clip_grad_norm_(me, 2.0); oe.step()
        loss = g.step(batch)
        torch.cuda.synchronize()
        le, ge, lg, gg = float(out["loss"].detach()), float(gn), float(loss), float(g.gnorm)
        hist.append((le, lg, ge, gg))
        assert bool(torch.isfinite(g.out[mg.modality_types[0]]).all())
        assert bool(torch.isfinite(mg.engine.gflat).all()) and float(mg.engine.gflat.abs().max()) < 1e4
        # same weights, same kernels: the forward is bitwise reproducible; the gradients differ by atomic order only
        assert abs(le - lg) <= 1e-6 * abs(le) and abs(ge - gg) <= 2e-3 * ge, hist
        assert rel_err(mg.engine.gflat, me.engine.gflat) < 5e-3, hist
    me.engine.assert_finite(); mg.engine.assert_finite()
    assert hist[0][0] != hist[3][0]                      # the four steps really trained


# ------------------------------------------------------------------------------------------------ hipGraph replay
def test_graphed_step_matches_eager(P):
    """graph.GraphedStep (the whole step as one hipGraph) against the eager loop on the same batches with a changing learning
    rate: the replay takes THIS step's lr and Adam bias corrections from device memory, new inputs go through the static
    buffers, and the finite flag still stops a bad step."""
    optim = importlib.import_module("mca-paper_amd.optim")
    graph = importlib.import_module("mca-paper_amd.graph")
    cfg = small_config("tab")
    sd = P.params.init_state_dict(cfg, seed=3)
    batches = [to_device(P.data.synthetic_batch(cfg, 4, seed=40 + i, p_drop=0.2), "cuda") for i in range(5)]
    lrs = [1e-3, 2e-3, 0.0, 1e-3, 3e-3]
    runs = []
    for graphed in (False, False, True):
        m = P.MCA(**copy.deepcopy(cfg)); m.load_state_dict(sd, strict=False); m = m.cuda()
        m.engine.check_finite = "deferred"
        opt = optim.FusedAdamW(m, lr=lrs[0], weight_decay=0.0)
        losses, snaps = [], []
        if graphed:
            # the constructor's warm-up steps are undone by the constructor itself: weights, moments, step count as before
            w0 = m.engine.flat.clone()
            g = graph.GraphedStep(m, opt, batches[0], clip=2.0, warmup=2)
            assert torch.equal(m.engine.flat, w0) and opt.step_count == 0 and float(opt.exp_avg.abs().max()) == 0.0
        for i, bt in enumerate(batches):
            opt.param_groups[0]["lr"] = lrs[i]
            if graphed:
                losses.append(float(g.step(bt)))
            else:
                out = m(bt); opt.zero_grad(); out["loss"].backward(); optim.clip_grad_norm_(m, 2.0); opt.step()
                losses.append(float(out["loss"].detach()))
            snaps.append(m.engine.flat.clone())
        torch.cuda.synchronize()
        m.engine.assert_finite()
        runs.append((losses, snaps, m, opt, g if graphed else None))
    (le, se, _, _, _), (le2, se2, _, _, _), (lg, sg, mg, og, g) = runs
    emb = mg.encoders["video"].token_encoder.embedding.weight
    i0 = (emb.data_ptr() - mg.engine.flat.data_ptr()) // 4

    def but_table(flat):
        """the flat parameters without the table encoder's embedding: its FORWARD renormalises the rows it reads in place
        (nn.Embedding(max_norm=1), encoders.py:26-32), whatever the optimizer does afterwards"""
        return torch.cat([flat[:i0], flat[i0 + emb.numel():]])
    # the same kernels on the same data.  The yardstick is the eager loop against ITSELF: the fp32-atomic accumulation order of
    # the weight gradients differs from run to run, and Adam's early updates (~lr * sign(g)) turn a gradient element at the
    # noise level into a whole lr of weight, so two eager runs already drift apart; the replay must stay within 3x that drift
    drift_l = [abs(a - b_) / abs(a) for a, b_ in zip(le, le2)]
    drift_w = [rel_err(a, b_) for a, b_ in zip(se, se2)]
    print("graph test: eager-eager loss drift", drift_l, "weights", drift_w)
    print("graph test: graph-eager loss drift", [abs(a - b_) / abs(a) for a, b_ in zip(le, lg)], "weights", [rel_err(a, b_) for a, b_ in zip(se, sg)])
    assert abs(le[0] - lg[0]) <= 1e-5 * abs(le[0]), (le, lg)
    for i in range(len(le)):
        assert abs(le[i] - lg[i]) <= 3 * drift_l[i] * abs(le[i]) + 2e-3 * abs(le[i]), (i, le, le2, lg)
        assert rel_err(sg[i], se[i]) <= 3 * drift_w[i] + 1e-3, (i, rel_err(sg[i], se[i]), drift_w[i])
    # lr = 0 (weight decay 0) at the third step: the replay read THIS step's learning rate, the weights did not move
    assert torch.equal(but_table(sg[2]), but_table(sg[1])) and not torch.equal(but_table(sg[3]), but_table(sg[2]))
    # a non-finite batch through the graph: the device flag stops the fused AdamW, the next poll raises
    bad = copy.deepcopy(batches[0]); bad["audio"]["tokens"][0, 0, 0] = float("nan")
    before, m_before, v_before = mg.engine.flat.clone(), og.exp_avg.clone(), og.exp_avg_sq.clone()
    og.param_groups[0]["lr"] = 1e-3
    g.step(bad)
    torch.cuda.synchronize()
    assert torch.equal(og.exp_avg, m_before) and torch.equal(og.exp_avg_sq, v_before)
    assert torch.equal(but_table(mg.engine.flat), but_table(before))
    with pytest.raises(Exception, match="not finite"):
        mg.engine.assert_finite()


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
