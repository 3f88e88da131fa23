import sys, os, torch
sys.path.insert(0, os.getcwd())
import umnn_amd
from umnn_amd import _lib, integral as I
from tests import _util as U
dev = torch.device("cuda:0")
torch.manual_seed(7)
m = umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=8, hidden_derivative=[50, 50, 50, 50], hidden_embedding=[64, 64], embedding_s=10, nb_steps=20, solver="CCParallel").to(dev).train()
x = (torch.randn(2100, 8, device=dev) * 0.8).requires_grad_()
res = {}
for key in ("ws", "swp", "fp32", "generic"):
    m.zero_grad(set_to_none=True); x.grad = None
    if key == "generic":
        with I.force_generic():
            ll, _ = m.compute_ll(x)
            (-ll.mean()).backward()
    else:
        _lib.set_backward_precision("fp32" if key == "fp32" else "bf16x3")
        with _lib.options(bwd_ws=1 if key == "ws" else 0):
            ll, _ = m.compute_ll(x)
            (-ll.mean()).backward()
        _lib.set_backward_precision("bf16x3")
    res[key] = {"x": x.grad.detach().clone(), **{k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}}
for key in ("ws", "swp", "fp32"):
    errs = {k: U.scaled_err(res[key][k].cpu().numpy(), res["generic"][k].cpu().numpy()) for k in res["generic"]}
    w = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print(key, " ".join(f"{k[-40:]}:{v:.2e}" for k, v in w), flush=True)
