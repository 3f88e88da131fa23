#!/usr/bin/env python
"""Full-size accuracy of the default arithmetic (bf16x3 quadrature kernels + bf16 K-concatenated conditioner GEMMs) against the
exact-fp32 kernels + fp32 conditioner on the same C3-shaped flow: max |d| / max(|ref|, 1) of z, log_jac and ll."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import umnn_amd  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = umnn_amd.UMNNMAFFlow(nb_flow=5, nb_in=63, hidden_derivative=[50] * 4, hidden_embedding=[512, 512], embedding_s=30,
                             nb_steps=100, device=dev).to(dev)
model.eval()
x = torch.randn(8192, 63, device=dev)


def run(prec, made):
    umnn_amd.set_forward_precision(prec)
    os.environ["UMNN_MADE_BF16X3"] = made
    with torch.no_grad():
        z, lj = model.compute_log_jac_bis(x)
        ll, _ = model.compute_ll(x)
    return z.double(), lj.double(), ll.double()


ref = run("fp32", "0")
for prec, made in (("bf16x3", "1"), ("bf16x3", "0"), ("bf16x6", "1")):
    out = run(prec, made)
    errs = [float(((a - b).abs() / b.abs().clamp_min(1.0)).max()) for a, b in zip(out, ref)]
    print(f"{prec:7s} conditioner {'bf16x3' if made == '1' else 'fp32  '}: z {errs[0]:.2e}  log_jac {errs[1]:.2e}  ll {errs[2]:.2e}")
