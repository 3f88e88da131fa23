"""GPU time of one made_linear_kernel launch (umnn_made_linear_forward) over row tiles x output-tile groups, events around 100 launches."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umnn_amd import _lib
from umnn_amd.made import pack_fragments

dev = torch.device("cuda:0")
lib = _lib.lib()
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(10000, 512, 512), (8192, 512, 512), (8192, 512, 1890), (1024, 512, 1920)]
for B, K, N in shapes:
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    x = torch.randn(B, K, device=dev)
    frags = pack_fragments(W)
    out = torch.empty(B, N, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    res = []
    for rt in (0, 1, 2, 4):
        for fg in ((0,) if rt == 0 else (1, 2, 3, 4, 6, 8, 16)):
            def go():
                _lib.check(lib.umnn_made_linear_forward(frags.data_ptr(), b.data_ptr(), K, N, x.data_ptr(), None, 0, B, 1, out.data_ptr(), 0, rt, fg, stream), "ml")
            for _ in range(5):
                go()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                go()
            e1.record()
            torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) * 10, rt, fg, lib.umnn_last_made_kernel_name().decode()))
    print(f"B={B} K={K} N={N}: auto {res[0][0]:.1f} us ({res[0][3]});  best five: " + "  ".join(f"RT={r} G={g}: {t:.1f}" for t, r, g, _ in sorted(res[1:])[:5]))
