"""Fixed cost (image staging + per-integral prologue + launch) of a forward launch: the same launch at n = nb_steps and at n = 1."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, torch, umnn_amd
from umnn_amd import _lib, integral as I
from umnn_amd.nets import mlp_spec
from umnn_amd.quadrature import device_tables
dev = torch.device("cuda:0")
for name, B, d, E, hid, n in (("toy", 4096, 2, 10, [100] * 4, 50), ("power", 10000, 6, 30, [50] * 4, 100), ("mnist", 100, 784, 30, [100, 50, 50, 50, 50], 50),
                              ("bsds300", 8192, 63, 30, [50] * 4, 100)):
    net = umnn_amd.IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
    spec = mlp_spec(net)
    x, h = torch.randn(B, d, device=dev), torch.randn(B, E * d, device=dev)
    F, fx, fx0 = (torch.empty_like(x) for _ in range(3))
    desc, keep = I._desc(spec)
    out = []
    for nn in (n, 1):
        w, s = device_tables(nn, dev)
        ms = ctypes.c_float()
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        rc = _lib.lib().umnn_cc_forward_timed(ctypes.byref(desc), None, p(x), p(h), p(w), p(s), nn, B, d, E, p(F), p(fx), p(fx0), 50, ctypes.byref(ms),
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "timed")
        out.append(ms.value)
    print(f"{name:8s} n={n}: {out[0]*1e3:8.1f} us   n=1: {out[1]*1e3:7.1f} us   kernel {_lib.lib().umnn_last_kernel_name().decode()}")
