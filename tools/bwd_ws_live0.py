import sys, os, torch
sys.path.insert(0, os.getcwd())
import umnn_amd
from umnn_amd import _lib, integral as I
from umnn_amd.nets import mlp_spec
dev = torch.device("cuda:0")
for hid in ([48, 60, 36, 50], [56, 56, 56, 56], [40, 40, 40, 40]):
    B, d, E, n = 8192, 63, 30, 100
    torch.manual_seed(0)
    net = umnn_amd.IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
    spec = mlp_spec(net)
    x, h, g = torch.randn(B, d, device=dev), torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev)
    gf = torch.randn(B, d, device=dev)
    for ws in (0, 1):
        with _lib.options(bwd_ws=ws):
            I.hip_backward(spec, None, x, h, g, gf, n); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): I.hip_backward(spec, None, x, h, g, gf, n)
            e1.record(); torch.cuda.synchronize()
            print(hid, "ws", ws, f"{e0.elapsed_time(e1)/3:.3f} ms", _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode(), flush=True)
