import faulthandler; faulthandler.enable()
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, '/root/repo')
os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29653")
import umnn_amd
from umnn_amd import sharding
def mark(m): print("MARK", m, flush=True)
rank, world, dev = sharding.init_from_env(backend="nccl", force_group=True)
mark("group")
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1 and dev.type == "cuda"
torch.manual_seed(0)
model = umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=6, hidden_derivative=[50] * 4, hidden_embedding=[64, 64], embedding_s=30,
                             nb_steps=20, solver="CCParallel").to(dev).train()
w0 = [p.detach().clone() for p in model.parameters()]
SKIP = os.environ.get("SKIP", "")
if "broadcast" not in SKIP:
    sharding.broadcast_parameters(model, force=True)                      # RCCL broadcast of every parameter and buffer
mark("broadcast")
assert all(torch.equal(a, b) for a, b in zip(w0, model.parameters()))
t = torch.arange(1024., device=dev)
sharding._all_reduce_sum(t)                                           # RCCL all_reduce on a device tensor
torch.cuda.synchronize()
assert torch.equal(t, torch.arange(1024., device=dev))
mark("all_reduce")
x = torch.randn(100, 6, device=dev)
if "backward" not in SKIP:
    ll, _ = model.compute_ll(x)
    (-ll.mean()).backward()
    torch.cuda.synchronize()
    mark("backward")
    g0 = [p.grad.detach().clone() for p in model.parameters() if p.requires_grad]
    sharding.allreduce_gradients(model, world, force=True)                # the flattened all-reduce, not short-circuited
    g1 = [p.grad for p in model.parameters() if p.requires_grad]
    assert all(torch.equal(a, b) for a, b in zip(g0, g1))
    base = g1[0]._base if g1[0]._base is not None else g1[0]
    assert all((g._base is base) for g in g1), "gradients must be views of the one reduced buffer"
    if "dropgrads" in SKIP:
        del g0, g1, base
        model.zero_grad(set_to_none=True)
    mark("allreduce_gradients")
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, capturable=True)
mode = os.environ.get("PROBE_MODE", "split")
hook = lambda m: sharding.allreduce_gradients(m, world, force=True)
if mode == "raw_global":            # torch.cuda.graph default mode, collective inside: what the first contact did (SIGSEGV)
    def one_step():
        opt.zero_grad(set_to_none=True); ll, _ = model.compute_ll(x); loss = -ll.mean(); loss.backward(); hook(model); opt.step(); return loss.detach()
    one_step(); torch.cuda.synchronize(); mark("eager")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = one_step()
    mark("captured global"); g.replay(); torch.cuda.synchronize(); mark("replayed")
elif mode == "ingraph":             # collective inside the graph, thread-local capture rules
    step = umnn_amd.GraphedTrainStep(model, opt, x, clip_value=10.0, grad_hook=hook, hook_in_graph=True)
    mark("captured in-graph"); l1 = float(step()); l2 = float(step()); mark("replayed %f %f" % (l1, l2))
else:                               # the default with a nccl group: graph A -> eager hook -> graph B
    step = umnn_amd.GraphedTrainStep(model, opt, x, clip_value=10.0, grad_hook=hook)
    assert step.split
    mark("captured split"); l1 = float(step()); l2 = float(step()); mark("replayed %f %f" % (l1, l2))
dist.destroy_process_group()
print("RCCL_WORLD1_OK", mode)
