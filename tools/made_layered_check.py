"""Per-layer launches of the conditioner kernel (umnn_made_linear_forward) against the split + library GEMM route and the fp32
chain, at the two wide-output conditioner shapes (VAE prior flow, BSDS300); times each route (hipEvents around 200 calls)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import umnn_amd
from umnn_amd import MADE, ConditionnalMADE

dev = torch.device("cuda:0")


def timed(fn, reps=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, made, args in [
    ("vae  (64+320 -> 512 -> 512 -> 1920, 1024 rows)", ConditionnalMADE(64, 320, [512, 512], 384 * 30, num_masks=1, natural_ordering=True).to(dev),
     (torch.randn(1024, 64, device=dev), torch.randn(1024, 320, device=dev))),
    ("bsds (63 -> 512 -> 512 -> 1890, 8192 rows)", MADE(63, [512, 512], 63 * 30, num_masks=1, natural_ordering=True).to(dev),
     (torch.randn(8192, 63, device=dev),)),
]:
    with torch.no_grad():
        umnn_amd.set_made_fast_path(False)
        exact = made.raw(*args)
        umnn_amd.set_made_fast_path(True)
        umnn_amd.set_made_fused(True, layered=False)
        per_layer = made.raw(*args)
        t_pl = timed(lambda: made.raw(*args))
        umnn_amd.set_made_fused(True, layered=True)
        scale = exact.abs().max().item()
        print(f"{name}: split + library GEMM {t_pl:7.1f} us   err {((per_layer - exact).abs().max().item() / scale):.2e}")
        for rt, g in [(0, 0), (1, 0), (2, 0), (4, 0), (2, 4), (2, 8), (2, 16), (4, 4), (4, 8), (4, 16), (4, 30), (1, 4), (1, 8)]:
            os.environ["UMNN_MADE_LINEAR_RT"], os.environ["UMNN_MADE_LINEAR_G"] = str(rt), str(g)
            lay = made.raw(*args)
            err = (lay - exact).abs().max().item() / scale
            t = timed(lambda: made.raw(*args))
            print(f"    layered RT={rt} G={g:2d}: {t:7.1f} us   err {err:.2e}   vs per-layer {((lay - per_layer).abs().max().item() / scale):.2e}")
        os.environ["UMNN_MADE_LINEAR_RT"], os.environ["UMNN_MADE_LINEAR_G"] = "0", "0"
