"""Which side of a HIP-vs-generic-route gradient difference is the rounding?  A float64 run of the generic (ATen) route as the truth."""
import copy, sys, os, torch
sys.path.insert(0, os.getcwd())
import umnn_amd
from umnn_amd import _lib, integral as I
from tests import _util as U
dev = torch.device("cuda:0")
torch.manual_seed(7)
m = umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=8, hidden_derivative=[50, 50, 50, 50], hidden_embedding=[64, 64], embedding_s=10, nb_steps=20, solver="CCParallel").to(dev).train()
x = (torch.randn(2100, 8, device=dev) * 0.8).requires_grad_()
res = {}
for key in ("ws16", "ws", "swp", "fp32bwd", "fp32all", "generic"):
    m.zero_grad(set_to_none=True); x.grad = None
    if key == "generic":
        with I.force_generic():
            ll, _ = m.compute_ll(x); (-ll.mean()).backward()
    else:
        _lib.set_backward_precision("fp32" if key.startswith("fp32") else "bf16x3")
        umnn_amd.set_forward_precision("fp32" if key == "fp32all" else "bf16x3")
        with _lib.options(bwd_ws=1 if key.startswith("ws") else 0, bwd_ws16=2 if key == "ws16" else 0):
            ll, _ = m.compute_ll(x); (-ll.mean()).backward()
        _lib.set_backward_precision("bf16x3"); umnn_amd.set_forward_precision("bf16x3")
    res[key] = {"x": x.grad.detach().double().clone(), **{k: p.grad.detach().double().clone() for k, p in m.named_parameters() if p.grad is not None}}
for mod in m.modules():                      # (the blocks cache their last embedding with its autograd graph)
    if hasattr(mod, "m_embeding"):
        mod.m_embeding = None
m64 = umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=8, hidden_derivative=[50, 50, 50, 50], hidden_embedding=[64, 64], embedding_s=10, nb_steps=20, solver="CCParallel").to(dev).train()
m64.load_state_dict(m.state_dict())
m64 = m64.double()
x64 = x.detach().double().requires_grad_()
with I.force_generic():
    ll, _ = m64.compute_ll(x64); (-ll.mean()).backward()
truth = {"x": x64.grad.detach().clone(), **{k: p.grad.detach().clone() for k, p in m64.named_parameters() if p.grad is not None}}
for key in res:
    errs = {k: U.scaled_err(res[key][k].cpu().numpy(), truth[k].cpu().numpy()) for k in truth}
    w = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    print(f"{key:8s} vs float64 generic route: " + " ".join(f"{k[-34:]}:{v:.2e}" for k, v in w), flush=True)
