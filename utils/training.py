"""Host helpers of the reference's step loop (utils/training.py:3-70), same names and return types.

``move_to`` keeps the reference's recursion over dicts / lists and its TypeError; copies are issued ``non_blocking`` (pinned
batches then overlap with compute; pageable ones behave as before).  The two norms read the engine's flat buffers when the model
runs on the native engine (one reduction instead of one per tensor) and fall back to the per-parameter loop otherwise."""
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


def get_param_norm(model, norm_type=2.0):
    """L-p norm over every parameter, a 1-element float64 tensor on the model's device (utils/training.py:48-57)."""
    norm_type = float(norm_type)
    eng = _engine_of(model)
    if eng is not None and norm_type == 2.0:
        return eng.flat.double().pow(2).sum().sqrt().reshape(1)          # alignment padding of the flat buffer is zero
    params = list(model.parameters())
    total = torch.zeros(1, dtype=torch.float64, device=params[0].device)
    for p in params:
        total += torch.norm(p.detach(), norm_type).double() ** norm_type
    return total ** (1.0 / norm_type)


def get_grad_norm(model, norm_type=2.0):
    """L-p norm over every gradient that exists, a 1-element float32 tensor (utils/training.py:59-70)."""
    norm_type = float(norm_type)
    params = list(model.parameters())
    total = torch.zeros(1, dtype=torch.float32, device=params[0].device)
    for p in params:
        if p.grad is not None:
            total += torch.norm(p.grad.detach(), norm_type) ** norm_type
    return total ** (1.0 / norm_type)
