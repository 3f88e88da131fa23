"""N>1 path on CPU: world_size-2 gloo.  Batch shards need no forward collective; training needs exactly one
flattened gradient all-reduce, after which every replica holds the single-process gradient."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import umnn_amd
from umnn_amd import sharding

pytestmark = pytest.mark.filterwarnings("ignore")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model():
    torch.manual_seed(7)
    return umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=3, hidden_derivative=[16, 16], hidden_embedding=[24, 24],
                                embedding_s=4, nb_steps=12, solver="CCParallel")


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    r, w, device = sharding.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and device.type == "cpu"
    model = _model()
    if rank == 1:                                   # replicas start different; broadcast must fix that
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    sharding.broadcast_parameters(model, src=0)
    torch.manual_seed(99)
    x = torch.randn(10, 3)                          # the global batch, identical on every rank
    xs = sharding.shard_rows(x, rank, world)
    model.train()
    ll, z = model.compute_ll(xs)                    # forward: no collective
    (-ll.sum() / x.shape[0]).backward()             # shard's share of the global mean
    sharding.allreduce_gradients(model, world, average=False)
    grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    torch.save({"ll": ll.detach(), "z": z.detach(), "grads": grads, "bounds": sharding.shard_bounds(10, rank, world)},
               os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_match_single_process():
    world = 2
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        parts = [torch.load(os.path.join(out, f"r{r}.pt")) for r in range(world)]
    model = _model()
    torch.manual_seed(99)
    x = torch.randn(10, 3)
    model.train()
    ll, z = model.compute_ll(x)
    (-ll.mean()).backward()
    assert [p["bounds"] for p in parts] == [(0, 5), (5, 10)]
    assert torch.allclose(torch.cat([p["ll"] for p in parts]), ll.detach(), atol=1e-6)
    assert torch.allclose(torch.cat([p["z"] for p in parts]), z.detach(), atol=1e-6)
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        for part in parts:                          # every replica holds the same, full gradient
            assert torch.allclose(part["grads"][k], p.grad, atol=2e-6, rtol=1e-5), k


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 65536, 65537):
        for world in (1, 2, 3, 8):
            b = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _bench_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import bench
    sharding.init_from_env(backend="gloo")
    model = _model()
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.5)
    sharding.broadcast_parameters(model, src=0)
    model.train()
    torch.manual_seed(5)
    x = torch.randn(12, 3) * 3            # large inputs: some raw gradients exceed the clip value
    xs = sharding.shard_rows(x, rank, world).contiguous()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    step = bench.make_train_step(model, opt, xs, None, world, clip_value=0.05)     # bench.py's own step closure
    for _ in range(3):
        step()
    torch.save({k: v.detach().clone() for k, v in model.state_dict().items()}, os.path.join(outdir, f"b{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_train_step_two_ranks_equals_single_process():
    """bench.py's train step (loss -> backward -> ONE flattened all-reduce -> clip AFTER the reduction -> Adam,
    UCIExperiments.py:133-146) on two gloo ranks reproduces single-process training on the whole batch."""
    import bench
    world = 2
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_bench_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        parts = [torch.load(os.path.join(out, f"b{r}.pt")) for r in range(world)]
    model = _model()
    model.train()
    torch.manual_seed(5)
    x = torch.randn(12, 3) * 3
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    step = bench.make_train_step(model, opt, x, None, 1, clip_value=0.05)
    for _ in range(3):
        step()
    ref = model.state_dict()
    for k in ref:
        assert torch.allclose(parts[0][k], parts[1][k], atol=0, rtol=0), k             # replicas stay bit-identical
        assert torch.allclose(parts[0][k], ref[k], atol=2e-6, rtol=1e-5), k


def _uneven_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import bench
    sharding.init_from_env(backend="gloo")
    model = _model()
    sharding.broadcast_parameters(model, src=0)
    model.train()
    torch.manual_seed(9)
    x = torch.randn(10, 3) * 2                                  # 10 rows over 3 ranks: shards of 4, 3, 3 rows
    xs = sharding.shard_rows(x, rank, world).contiguous()
    assert xs.shape[0] == (4 if rank == 0 else 3)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    step = bench.make_train_step(model, opt, xs, None, world, clip_value=0.05)
    for _ in range(3):
        step()
    torch.save({k: v.detach().clone() for k, v in model.state_dict().items()}, os.path.join(outdir, f"u{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_train_step_with_uneven_shards_equals_single_process():
    """10 rows over 3 ranks (4 + 3 + 3).  The reference takes ONE mean over the whole batch
    (UCIExperiments.py:133-146); averaging per-rank means would weight the rows of the short shards 4/3 too much.  bench.py's step
    normalises by the global row count and sums the gradients: three replicas bit-identical and equal to single-process training."""
    import bench
    world = 3
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_uneven_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        parts = [torch.load(os.path.join(out, f"u{r}.pt")) for r in range(world)]
    model = _model()
    model.train()
    torch.manual_seed(9)
    x = torch.randn(10, 3) * 2
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    step = bench.make_train_step(model, opt, x, None, 1, clip_value=0.05)
    for _ in range(3):
        step()
    ref = model.state_dict()
    for k in ref:
        for r in range(1, world):
            assert torch.equal(parts[0][k], parts[r][k]), (k, r)
        assert torch.allclose(parts[0][k], ref[k], atol=3e-6, rtol=1e-5), k


def _one_rank_worker(port, q):
    import os
    os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import umnn_amd
    from umnn_amd import sharding
    rank, world, dev = sharding.init_from_env(backend="gloo", force_group=True)
    assert dist.is_initialized() and dist.get_world_size() == 1
    torch.manual_seed(0)
    model = umnn_amd.UMNNMAFFlow(nb_flow=1, nb_in=3, hidden_derivative=[16, 16], hidden_embedding=[16], embedding_s=4,
                                 nb_steps=10, solver="CCParallel")
    w0 = [p.detach().clone() for p in model.parameters()]
    sharding.broadcast_parameters(model, force=True)
    assert all(torch.equal(a, b) for a, b in zip(w0, model.parameters()))
    ll, _ = model.compute_ll(torch.randn(8, 3))
    (-ll.mean()).backward()
    g0 = [p.grad.clone() for p in model.parameters() if p.requires_grad]
    sharding.allreduce_gradients(model, world)                 # world == 1: early return, gradients untouched
    assert all(p.grad._base is None for p in model.parameters() if p.requires_grad)
    sharding.allreduce_gradients(model, world, force=True)     # the collective path: flattened buffer, views back
    g1 = [p.grad for p in model.parameters() if p.requires_grad]
    assert all(torch.equal(a, b) for a, b in zip(g0, g1)) and all(g._base is not None for g in g1)
    dist.destroy_process_group()
    q.put("ok")


def test_one_rank_group_forced_collectives():
    """sharding's ``force`` switches (used by the one-rank RCCL test on the GPU box) on the gloo backend: a world-size-1
    group, broadcast and the flattened all-reduce really run and leave the gradients as views of one buffer."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_one_rank_worker, args=(29671, q))
    pr.start()
    pr.join(120)
    assert pr.exitcode == 0 and q.get(timeout=5) == "ok"


def _world8_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import bench
    sharding.init_from_env(backend="gloo")
    assert sharding.shard_bounds(65536, rank, world) == (8192 * rank, 8192 * (rank + 1))      # BASELINE C3's batch over 8 GPUs
    model = _model()
    with torch.no_grad():                                       # every replica starts somewhere else; broadcast fixes that
        for p in model.parameters():
            p.add_(0.01 * rank)
    sharding.broadcast_parameters(model, src=0)
    model.train()
    # bench.py's first-contact check of the one collective the design depends on (round 6): sum of rank + 1 over 8 ranks = 36
    rec = bench.collective_self_check(model, rank, world, torch.device("cpu"), reps=2)
    assert rec["allreduce_ok"] and rec["allreduce_bytes"] == 4 * sum(p.numel() for p in model.parameters() if p.requires_grad)
    torch.manual_seed(21)
    x = torch.randn(40, 3) * 2                                  # the global batch (identical on every rank); 5 rows per rank
    xs = sharding.shard_rows(x, rank, world).contiguous()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    step = bench.make_train_step(model, opt, xs, None, world, clip_value=0.05)     # bench.py's own step closure
    for _ in range(2):
        step()
    torch.save({k: v.detach().clone() for k, v in model.state_dict().items()}, os.path.join(outdir, f"w{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_train_step_eight_ranks_replicas_stay_bit_identical():
    """bench.py's train step on EIGHT gloo ranks (the node size the metric is quoted on): the C3 batch's shard
    bounds are 8192-row blocks, after two optimisation steps all eight replicas hold bit-identical weights, and those match
    single-process training on the whole batch (UCIExperiments.py:133-146: one gradient, clipped after the reduction)."""
    import bench
    world = 8
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_world8_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        parts = [torch.load(os.path.join(out, f"w{r}.pt")) for r in range(world)]
    model = _model()
    model.train()
    torch.manual_seed(21)
    x = torch.randn(40, 3) * 2
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    step = bench.make_train_step(model, opt, x, None, 1, clip_value=0.05)
    for _ in range(2):
        step()
    ref = model.state_dict()
    for k in ref:
        for r in range(1, world):
            assert torch.equal(parts[0][k], parts[r][k]), (k, r)
        assert torch.allclose(parts[0][k], ref[k], atol=3e-6, rtol=1e-5), k


def test_ipc_mode_is_set_before_the_gpu_runtime_can_initialise():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 must be in the environment before torch touches the GPU.  `import umnn_amd`
    and `import bench` both set it when the caller has not (an explicit setting wins); sharding.init_from_env no longer does (too late)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    env["PYTHONPATH"] = root
    for mod in ("umnn_amd", "bench"):
        code = ("import os, sys; assert 'torch' not in sys.modules; import %s; "
                "print('IPC', os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))" % mod)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
        assert r.returncode == 0 and "IPC 0" in r.stdout, (mod, r.stdout, r.stderr[-500:])
        r = subprocess.run([sys.executable, "-c", code], env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="1"), capture_output=True,
                           text=True, timeout=300, cwd=root)
        assert "IPC 1" in r.stdout, (mod, r.stdout)
    # the common order `import torch; import umnn_amd`: the guard itself must not touch HIP before the default is
    # written -- it may only read torch.cuda.is_initialized() (Python state); is_available() would call hipGetDeviceCount
    code = ("import os, torch, unittest.mock as m\n"
            "with m.patch.object(torch.cuda, 'is_available', side_effect=AssertionError('HIP probe before the env default')):\n"
            "    import umnn_amd\n"
            "print('IPC', os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0 and "IPC 0" in r.stdout, (r.stdout, r.stderr[-800:])
    import inspect
    assert "setdefault(\"HSA_ENABLE_IPC_MODE_LEGACY\"" not in inspect.getsource(sharding.init_from_env)
