import sys, os, torch
sys.path.insert(0, os.getcwd())
import umnn_amd
from umnn_amd import _lib, integral as I
from umnn_amd.nets import mlp_spec
from tests import _util as U
dev = torch.device("cuda:0")
for n in (1, 2, 3, 5):
    for gfx in (False, True):
        B, d, E = 300, 63, 30
        torch.manual_seed(n)
        net = umnn_amd.IntegrandNetwork(d, 1 + E, [50] * 4, 1).to(dev)
        spec = mlp_spec(net)
        x, x0 = torch.randn(B, d, device=dev) * 2, torch.randn(B, d, device=dev) * 0.3
        h, gg = torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev)
        gf = torch.randn(B, d, device=dev) if gfx else None
        outs = {}
        for ws in (0, 1):
            with _lib.options(bwd_ws=ws):
                outs[ws] = I.hip_backward(spec, x0, x, h, gg, gf, n)
                name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
        errs = [U.scaled_err(outs[1][i].cpu().numpy(), outs[0][i].cpu().numpy()) for i in range(4)]
        print(n, gfx, name, " ".join(f"{e:.1e}" for e in errs), flush=True)
