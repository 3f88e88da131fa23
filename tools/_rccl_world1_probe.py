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
sharding.broadcast_parameters(model, force=True)                      # RCCL broadcast of every parameter and buffer
mark("broadcast")
assert all(torch.equal(a, b) for a, b in zip(w0, model.parameters()))
t = torch.arange(1024., device=dev)
sharding._all_reduce_sum(t)                                           # RCCL all_reduce on a device tensor
torch.cuda.synchronize()
assert torch.equal(t, torch.arange(1024., device=dev))
mark("all_reduce")
x = torch.randn(100, 6, device=dev)
out = model.compute_ll(x)
(-out[0].mean()).backward()
del out                   # no autograd graph built on the DEFAULT stream may outlive this point (ll AND z hold it): its
torch.cuda.synchronize()  # AccumulateGrad nodes would run on the legacy stream during the capture below (GraphedTrainStep's docstring)
mark("backward")
g0 = [p.grad.detach().clone() for p in model.parameters() if p.requires_grad]
sharding.allreduce_gradients(model, world, force=True)                # the flattened all-reduce, not short-circuited
g1 = [p.grad for p in model.parameters() if p.requires_grad]
assert all(torch.equal(a, b) for a, b in zip(g0, g1))
base = g1[0]._base if g1[0]._base is not None else g1[0]
assert all((g._base is base) for g in g1), "gradients must be views of the one reduced buffer"
mark("allreduce_gradients")
# one eager data-parallel optimisation step with the collective, on a side stream as a training loop would run it
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, capturable=True)
ref = [p.detach().clone() for p in model.parameters()]
losses = []
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for it in range(3):
        opt.zero_grad(set_to_none=True)
        out = model.compute_ll(x)
        loss = -out[0].mean()
        loss.backward()
        sharding.allreduce_gradients(model, world, force=True)
        torch.nn.utils.clip_grad_value_(list(model.parameters()), 10.0)
        opt.step()
        losses.append(float(loss))
        del out, loss            # (ll AND z carry the autograd graph)
torch.cuda.current_stream().wait_stream(side)
assert losses[2] < losses[0] and any(not torch.equal(a, b) for a, b in zip(ref, model.parameters()))
mark("eager steps %s" % losses)
# graphs: inference capture works beside the process group; a captured training step must REFUSE a collective hook
model.eval()
gl = umnn_amd.GraphedLL(model, x)
with torch.no_grad():
    assert torch.equal(gl()[0], model.compute_ll(x)[0])
mark("GraphedLL beside the group")
model.train()
try:
    umnn_amd.GraphedTrainStep(model, opt, x, clip_value=10.0, grad_hook=lambda m: sharding.allreduce_gradients(m, world, force=True))
    raise SystemExit("GraphedTrainStep accepted a collective hook under a nccl group")
except NotImplementedError:
    mark("graphed train step refuses the collective hook")
model.zero_grad(set_to_none=True)
step = umnn_amd.GraphedTrainStep(model, opt, x, clip_value=10.0)          # without a hook it captures beside the group
mark("GraphedTrainStep without hook beside the group")
l1 = float(step()); l2 = float(step()); l3 = float(step(torch.randn(100, 6, device=dev)))
assert all(map(lambda v: v == v and abs(v) < 1e6, (l1, l2, l3))), (l1, l2, l3)
assert l2 < l1
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK", losses, l1, l2, l3)