"""MADE conditioner on the GPU: the inference fast path (every masked linear as one K-concatenated bf16 GEMM, operands
built by umnn_made_split3) against the fp32 chain, the CPU oracle and the autoregressive property."""
import os

import numpy as np
import pytest
import torch

from oracle import cc_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _both(fn):
    """(fast path result, fp32 chain result) of fn() under no_grad."""
    import umnn_amd
    old = umnn_amd.get_made_fast_path()
    try:
        with torch.no_grad():
            umnn_amd.set_made_fast_path(True)
            fast = fn()
            umnn_amd.set_made_fast_path(False)
            slow = fn()
    finally:
        umnn_amd.set_made_fast_path(old)
    return fast, slow


@pytest.mark.parametrize("nin,hid,E,B", [(63, [512, 512], 30, 257), (6, [100, 100, 100], 30, 1000), (2, [100] * 4, 10, 64),
                                         (5, [33], 3, 19)])
def test_made_fast_path_matches_fp32_chain_and_oracle(dev, nin, hid, E, B):
    from umnn_amd import MADE
    from umnn_amd.made import MaskedLinear, _fast_path_ok
    torch.manual_seed(1)
    made = MADE(nin, hid, nin * E, num_masks=1, natural_ordering=True).to(dev)
    x = torch.randn(B, nin, device=dev) * 2
    with torch.no_grad():
        assert _fast_path_ok(x), "fast path must be active on a GPU box (library + torch.mm(out_dtype=))"
    fast, slow = _both(lambda: made.raw(x))
    scale = slow.abs().max().item()
    assert (fast - slow).abs().max().item() <= 2e-5 * scale
    lin = [m for m in made.net if isinstance(m, MaskedLinear)]
    ref = O.made_forward([m.weight.detach().cpu().numpy().astype(np.float64) for m in lin],
                         [m.bias.detach().cpu().numpy().astype(np.float64) for m in lin],
                         [m.mask.cpu().numpy().astype(np.float64) for m in lin], x.cpu().numpy().astype(np.float64))
    assert np.abs(fast.cpu().numpy() - ref).max() <= 2e-5 * scale
    # training keeps the fp32 chain and its autograd graph
    y = made.raw(x)
    assert y.requires_grad and torch.equal(y.detach(), slow)


def test_made_fast_path_is_exactly_autoregressive(dev):
    """Masked weights are zero in both bf16 pieces, so h for dimension j must not move AT ALL when x[j:] changes."""
    from umnn_amd import MADE
    torch.manual_seed(2)
    nin, E = 9, 4
    made = MADE(nin, [64, 64], nin * E, num_masks=1, natural_ordering=True).to(dev)
    x = torch.randn(50, nin, device=dev)
    with torch.no_grad():
        h = made.raw(x).view(50, E, nin)
        for j in range(nin):
            x2 = x.clone()
            x2[:, j:] = torch.randn(50, nin - j, device=dev) * 3
            h2 = made.raw(x2).view(50, E, nin)
            assert torch.equal(h[:, :, :j + 1], h2[:, :, :j + 1]), j


def test_conditional_made_fast_path(dev):
    from umnn_amd import ConditionnalMADE
    torch.manual_seed(3)
    nin, cond, E = 8, 16, 6
    made = ConditionnalMADE(nin, cond, [96, 96], (nin + cond) * E, num_masks=1, natural_ordering=True).to(dev)
    x, ctx = torch.randn(77, nin, device=dev), torch.randn(77, cond, device=dev)
    fast, slow = _both(lambda: made.raw(x, ctx))
    assert fast.shape == slow.shape == (77, nin * E)
    assert (fast - slow).abs().max().item() <= 2e-5 * slow.abs().max().item()


def test_made_split3_edge_shapes(dev):
    """The operand builder on odd column counts / empty input, against a torch restatement of the same rounding."""
    import ctypes
    from umnn_amd import _lib
    lib = _lib.lib()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rows, cols, relu in [(5, 1, 0), (3, 7, 1), (64, 63, 0), (1, 512, 1), (0, 4, 1)]:
        x = torch.randn(rows, cols, device=dev) * 5
        ld = 3 * cols + 2 + ((-(3 * cols + 2)) % 8)
        out = torch.full((rows, ld), 7.0, dtype=torch.bfloat16, device=dev)
        _lib.check(lib.umnn_made_split3(x.data_ptr(), rows, cols, relu, out.data_ptr(), ld, stream), "split3")
        a = torch.relu(x) if relu else x
        hi = a.bfloat16()
        lo = (a - hi.float()).bfloat16()
        want = torch.cat([hi, lo, hi, torch.ones(rows, 2, dtype=torch.bfloat16, device=dev),
                          torch.zeros(rows, ld - 3 * cols - 2, dtype=torch.bfloat16, device=dev)], 1)
        assert torch.equal(out, want), (rows, cols, relu)
    with pytest.raises(RuntimeError):
        _lib.check(lib.umnn_made_split3(x.data_ptr(), 1, 4, 0, out.data_ptr(), 5, stream), "split3")
