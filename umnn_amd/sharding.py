"""Batch sharding across the GPUs of one node: one process per GPU, contiguous row shards, replicated
weights.  The forward integral needs NO collective -- every (sample, dimension) integral is independent given h,
and h depends only on that sample.  Training adds exactly one flattened all-reduce of the parameter gradients
(RCCL over xGMI through torch.distributed's "nccl" backend; "gloo" in the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force_group=False):
    """Join the process group torchrun described in the environment.  Returns (rank, world, device).  ``force_group``: create
    the group even for WORLD_SIZE=1 (a one-rank RCCL group: lets a single-GPU box exercise the collective code paths)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    # one rank per GPU; more ranks than GPUs (a 2-rank smoke test of the launch path on a 1-GPU box) wrap around, which
    # only the gloo backend accepts -- RCCL refuses two ranks on one device
    device = torch.device("cuda", local % torch.cuda.device_count()) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if (world > 1 or force_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # (HSA_ENABLE_IPC_MODE_LEGACY=0, which RCCL needs on this driver stack, is read when the HSA runtime initialises: it is set
        # by `import umnn_amd` / at the top of bench.py, BEFORE torch touches the GPU -- here it would be too late)
        backend = backend or os.environ.get("UMNN_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world,
                                **({"device_id": device} if use_gpu and backend == "nccl" else {}))
    return rank, world, device


def shard_bounds(n_rows, rank, world):
    """Contiguous [lo, hi) of the rows rank owns; shards differ by at most one row."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_rows(t, rank, world):
    lo, hi = shard_bounds(t.shape[0], rank, world)
    return t[lo:hi]


def _all_reduce_sum(flat):
    """all_reduce(SUM) in place; device tensors under the gloo backend (the 2-ranks-on-one-GPU smoke test) are staged through
    the host, RCCL reduces them where they are."""
    if flat.is_cuda and dist.get_backend() == "gloo":
        c = flat.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        flat.copy_(c)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)


def allreduce_gradients(module, world=None, average=True, force=False):
    """One flattened all-reduce(SUM) over every trainable gradient (a few MB: latency-bound on xGMI, so one
    message is the right shape), then scatter back.  Call after backward and BEFORE gradient clipping so that
    clipping sees the same global gradient a single process would (UCIExperiments.py:143).  ``force``: issue the collective
    even in a one-rank group (tests of the RCCL path on a single GPU)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 and not (force and dist.is_initialized()):
        return
    params = [p for p in module.parameters() if p.requires_grad]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    grads = [p.grad for p in params]
    flat = torch._utils._flatten_dense_tensors(grads)           # one message (a few MB: latency-bound on xGMI)
    _all_reduce_sum(flat)
    if average:
        flat /= world
    # the gradients become views into the reduced flat buffer: no second pass of per-parameter copies
    for p, v in zip(params, torch._utils._unflatten_dense_tensors(flat, grads)):
        p.grad = v


def broadcast_parameters(module, src=0, force=False):
    """Make every replica start from rank src's weights (and buffers).  ``force``: also in a one-rank group."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return
    from .made import invalidate_caches
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            v = t.detach()                       # detach() shares the version counter: the in-place receive bumps it
            if v.is_cuda and dist.get_backend() == "gloo":       # (gloo builds without device support: stage through the host)
                c = v.cpu()
                dist.broadcast(c, src)
                v.copy_(c)
            else:
                dist.broadcast(v, src)
    invalidate_caches(module)                    # belt and braces for the conditioner's masked / packed weight caches
