import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from umnn_amd import _lib, integral as I
from umnn_amd.nets import MlpSpec
from tests import _util as U
dev = torch.device("cuda:0")
rng = np.random.RandomState(5)
B, d, E, n = 2100, 8, 6, 11
sizes = [1 + E] + [50, 50, 50, 50] + [1]
lin = []
for i in range(len(sizes) - 1):
    m = torch.nn.Linear(sizes[i], sizes[i + 1])
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy((rng.randn(sizes[i + 1], sizes[i]) * (1.6 / np.sqrt(sizes[i]))).astype(np.float32)))
        m.bias.copy_(torch.from_numpy((rng.randn(sizes[i + 1]) * 0.3).astype(np.float32)))
    lin.append(m.to(dev))
spec = MlpSpec(lin, _lib.ACT_RELU, _lib.OUT_ELU_PLUS_ONE)
for seed in range(40):
    torch.manual_seed(seed)
    x, x0 = torch.randn(B, d, device=dev) * 2, torch.randn(B, d, device=dev) * 0.3
    h, gg, gf = torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev), torch.randn(B, d, device=dev)
    outs = {}
    for key, ws, prec in (("swp", 0, "bf16x3"), ("ws", 1, "bf16x3"), ("fp32", 0, "fp32")):
        _lib.set_backward_precision(prec)
        with _lib.options(bwd_ws=ws):
            outs[key] = I.hip_backward(spec, x0, x, h, gg, gf, n)
    _lib.set_backward_precision("bf16x3")
    errs = []
    for i, nm in enumerate(("dx0", "dx", "dh", "dtheta")):
        a_, b_, r_ = (outs[k][i].cpu().numpy() for k in ("swp", "ws", "fp32"))
        errs.append((nm, U.scaled_err(b_, a_), U.scaled_err(b_, r_), U.scaled_err(a_, r_)))
    bad = [e for e in errs if e[1] > 5e-6 or e[2] > (2e-3 if e[0] == "dh" else 2e-4)]
    print(seed, "BAD" if bad else "ok", " ".join(f"{e[0]}:{e[1]:.1e}/{e[2]:.1e}/{e[3]:.1e}" for e in errs), flush=True)
