"""Per-item wall time of the segmented data-parallel step (graph.GraphedStep(dp=...)): each graph segment and each eager
collective, synchronised after every item.  Launch: MCA_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2
--master-addr 127.0.0.1 tools/diag_dp_segments.py [batch]   (both ranks on cuda:0 with gloo; nccl needs one GPU per rank)"""
import importlib, os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
backend = os.environ.get("MCA_DIST_BACKEND", "nccl")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ndev = torch.cuda.device_count()
torch.cuda.set_device(rank % ndev)
dist.init_process_group(backend) if backend != "nccl" else dist.init_process_group("nccl", device_id=torch.device("cuda", rank % ndev))
P = importlib.import_module("mca-paper_amd"); optim = importlib.import_module("mca-paper_amd.optim")
dpm = importlib.import_module("mca-paper_amd.dp"); graph = importlib.import_module("mca-paper_amd.graph")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = P.config.cmu_model_config(batch_size=b)
torch.manual_seed(43)
model = P.build_model(cfg).cuda(); model.engine.check_finite = "deferred"
opt = optim.FusedAdamW(model, lr=1e-4)
dp = dpm.DataParallelMCA(model)
batch = P.data.synthetic_batch(cfg, b, seed=1234 + rank, device="cuda")
g = graph.GraphedStep(model, opt, batch, clip=2.0, dp=dp)
for _ in range(2): g.step()
torch.cuda.synchronize(); dist.barrier()
t0 = time.perf_counter()
for _ in range(5): g.step()
torch.cuda.synchronize(); dist.barrier()
if rank == 0: print(f"replayed step: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms", flush=True)
tot = [0.0] * len(g.program)
for _ in range(3):
    opt.step_count += 1; opt.set_hyper(opt.step_count)
    for i, item in enumerate(g.program):
        torch.cuda.synchronize(); t = time.perf_counter()
        item.replay() if isinstance(item, torch.cuda.CUDAGraph) else item()
        torch.cuda.synchronize(); tot[i] += time.perf_counter() - t
if rank == 0:
    for i, item in enumerate(g.program):
        print(f"  item {i:2d} {'graph     ' if isinstance(item, torch.cuda.CUDAGraph) else 'collective'} {tot[i] / 3 * 1e3:9.3f} ms")
dist.destroy_process_group()
