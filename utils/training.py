"""Host helpers of the reference's step loop (utils/training.py:3-70), same names and return types.

``move_to`` keeps the reference's recursion over dicts / lists and its TypeError; copies are issued ``non_blocking`` (pinned
batches then overlap with compute; pageable ones behave as before).

The two norms reproduce the reference's VALUES, quirk included: both fetch the device with ``next(iter(parameters))`` on the
very generator they then loop over (utils/training.py:50-51,61-62), so the FIRST parameter of ``model.parameters()`` is in
neither norm (checked by running the reference's functions: tests/test_host_cpu.py states the contract).  ``skip_first=False``
gives the norm over every parameter.  ``get_param_norm`` reads the engine's flat buffer when the model runs on the native
engine (one reduction instead of one per tensor), and so does ``get_grad_norm`` when every ``p.grad`` is its view of the flat
gradient buffer (always, after a native backward); otherwise it walks ``p.grad``."""
import torch


def move_to(obj, device):
    if torch.is_tensor(obj):
        return obj.to(device, non_blocking=True)
    if isinstance(obj, dict):
        return {k: move_to(v, device) for k, v in obj.items()}
    if isinstance(obj, list):
        return [move_to(v, device) for v in obj]
    raise TypeError("Invalid type for move_to")


def copy_batch(obj):
    """Detached deep copy of a (nested) batch, the reference's helper of the same name (utils/training.py:19-33): tensors are
    cloned, dicts and lists rebuilt, anything else is a TypeError."""
    if torch.is_tensor(obj):
        return obj.detach().clone()
    if isinstance(obj, dict):
        return {k: copy_batch(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [copy_batch(v) for v in obj]
    raise TypeError("Invalid type for copy_to")


def count_parameters(model, print_summary=False):
    """(embedding parameters, other parameters): a parameter counts as 'embedding' when its name contains that word."""
    emb = other = 0
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if print_summary:
            print(f"{name}:{p.numel() / 10 ** 6}M")
        if "embedding" in name:
            emb += p.numel()
        else:
            other += p.numel()
    return emb, other


def _engine_of(model):
    m = getattr(model, "model", model)          # DataParallelMCA wraps .model
    return getattr(m, "_engine", None)


def get_param_norm(model, norm_type=2.0, skip_first=True):
    """L-p norm over the parameters (all but the first: module docstring), a 1-element float64 tensor on the model's device
    (utils/training.py:48-57)."""
    norm_type = float(norm_type)
    params = list(model.parameters())
    eng = _engine_of(model)
    if eng is not None and norm_type == 2.0:
        sq = eng.flat.double().pow(2).sum()          # alignment padding of the flat buffer is zero
        if skip_first and params:
            sq = sq - params[0].detach().double().pow(2).sum()
        return sq.clamp_min(0).sqrt().reshape(1)
    total = torch.zeros(1, dtype=torch.float64, device=params[0].device)
    for p in params[1 if skip_first else 0:]:
        total += torch.norm(p.detach(), norm_type).double() ** norm_type
    return total ** (1.0 / norm_type)


def get_grad_norm(model, norm_type=2.0, skip_first=True):
    """L-p norm over every gradient that exists (all but the first parameter's: module docstring), a 1-element float32 tensor
    (utils/training.py:59-70)."""
    norm_type = float(norm_type)
    params = list(model.parameters())
    eng = _engine_of(model)
    if eng is not None and norm_type == 2.0 and params and all(
            p.grad is not None and p.grad.data_ptr() == eng.grad_of(p).data_ptr() for p in params):
        # after a native backward every p.grad is its view of the flat gradient buffer: one reduction instead of three small
        # kernels per tensor (the loop logs this every step, train_accel_gpu.py:126-130)
        sq = eng.gflat.double().pow(2).sum()          # alignment padding of the flat buffer stays zero
        if skip_first:
            sq = sq - params[0].grad.detach().double().pow(2).sum()
        return sq.clamp_min(0).sqrt().float().reshape(1)
    total = torch.zeros(1, dtype=torch.float32, device=params[0].device)
    for p in params[1 if skip_first else 0:]:
        if p.grad is not None:
            total += torch.norm(p.grad.detach(), norm_type) ** norm_type
    return total ** (1.0 / norm_type)
