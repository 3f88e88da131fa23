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


@pytest.mark.parametrize("nin,hid,E,B", [(63, [512, 512], 30, 8192), (6, [512, 512], 30, 10000), (2, [100] * 4, 10, 4096),
                                         (5, [33], 3, 19), (3, [16, 8, 24], 2, 1), (7, [48, 500], 5, 257), (20, [512], 1, 130)])
@pytest.mark.parametrize("out_dtype", [None, torch.bfloat16])
def test_fused_conditioner_kernel_matches_per_layer_path_and_oracle(dev, nin, hid, E, B, out_dtype):
    """umnn_made_mlp_forward (the whole masked MLP in one launch: csrc/made_fused.hip) against the per-layer fast path (same
    bf16x3 products, bias there in bf16 pieces, here in fp32), the fp32 chain and the float64 oracle of models/UMNN/made.py:113-119
    -- over widths that are not multiples of 16 / 32, fewer tiles than waves, batches that do not fill a row tile, both
    row-tile variants (RT = 1 / 4), fp32 and bf16 output."""
    import umnn_amd
    from umnn_amd import MADE, _lib
    from umnn_amd.made import MaskedLinear
    torch.manual_seed(nin + B)
    made = MADE(nin, hid, nin * E, num_masks=1, natural_ordering=True).to(dev)
    with torch.no_grad():
        for m in made.net:
            if isinstance(m, MaskedLinear):
                m.bias.uniform_(-1.0, 1.0)
    x = torch.randn(B, nin, device=dev) * 2
    outs = {}
    try:
        with torch.no_grad():
            for fused in (True, False):
                umnn_amd.set_made_fused(fused, wide_out=True)       # (wide outputs too: the kernel's multi-pass output layer)
                launches = _lib.lib().umnn_made_launch_count()
                outs[fused] = made.raw(x, out_dtype=out_dtype)
                assert _lib.lib().umnn_made_launch_count() - launches == (1 if fused else 0)
                if fused:
                    assert "made_fused" in _lib.lib().umnn_last_made_kernel_name().decode()
            umnn_amd.set_made_fast_path(False)
            exact = made.raw(x)
    finally:
        umnn_amd.set_made_fused(True, wide_out=False)
        umnn_amd.set_made_fast_path(True)
    assert outs[True].dtype == (out_dtype or torch.float32) and outs[True].shape == exact.shape
    scale = exact.abs().max().item()
    tol = 2e-5 if out_dtype is None else 6e-3           # bf16 storage: 2^-9 of the value, relative to the largest entry
    assert (outs[True].float() - exact).abs().max().item() <= tol * scale
    assert (outs[True].float() - outs[False].float()).abs().max().item() <= (1e-5 if out_dtype is None else 8e-3) * scale
    lin = [m for m in made.net if isinstance(m, MaskedLinear)]
    ref = O.made_forward([m.weight.detach().cpu().numpy().astype(np.float64) for m in lin],
                         [m.bias.detach().cpu().numpy().astype(np.float64) for m in lin],
                         [m.mask.cpu().numpy().astype(np.float64) for m in lin], x[:64].cpu().numpy().astype(np.float64))
    assert np.abs(outs[True][:64].float().cpu().numpy() - ref).max() <= tol * scale


def test_fused_conditioner_falls_back_beyond_512_and_sees_weight_updates(dev):
    """Hidden layers wider than 512 (MNISTExperiment's [1024]*3) keep the per-layer library GEMMs; the fused path's packed
    fragments follow in-place weight updates (version-keyed cache) and explicit cache invalidation."""
    import umnn_amd
    from umnn_amd import MADE, _lib
    torch.manual_seed(0)
    wide = MADE(8, [1024, 1024], 16, num_masks=1, natural_ordering=True).to(dev)
    x = torch.randn(40, 8, device=dev)
    with torch.no_grad():
        n0 = _lib.lib().umnn_made_launch_count()
        wide.raw(x)
        assert _lib.lib().umnn_made_launch_count() == n0, "widths beyond 512 keep the per-layer path"
        made = MADE(8, [64, 64], 16, num_masks=1, natural_ordering=True).to(dev)
        a = made.raw(x).clone()
        assert _lib.lib().umnn_made_launch_count() == n0 + 1 and "made_fused" in _lib.lib().umnn_last_made_kernel_name().decode()
        made.net[0].weight.mul_(1.5)                      # in place: the version counter moves, the fragments are re-packed
        b = made.raw(x)
        assert not torch.equal(a, b)
        umnn_amd.set_made_fast_path(False)
        try:
            exact = made.raw(x)
        finally:
            umnn_amd.set_made_fast_path(True)
        assert (b - exact).abs().max().item() <= 2e-5 * exact.abs().max().item()


@pytest.mark.parametrize("cond", [0, 40])
def test_hidden_stack_kernel_feeding_the_library_gemm_of_a_wide_output_layer(dev, cond):
    """Conditioners with a wide output (BSDS300: 1890 columns, the VAE prior flow: 1920 after dropping the context columns) run
    their hidden stack in the fused kernel, which writes relu(h_last) straight as the bf16 operand [hi | lo | hi | 1 | 1] of the
    output layer's library GEMM (umnn_made_mlp_forward_ex, out_mode 2): two launches per block, same numbers as the per-layer path."""
    import umnn_amd
    from umnn_amd import MADE, ConditionnalMADE, _lib
    from umnn_amd.made import MaskedLinear, _fused_ok
    torch.manual_seed(4 + cond)
    nin, E, B = 63, 30, 777
    if cond:
        made = ConditionnalMADE(nin, cond, [512, 512], (nin + cond) * E, num_masks=1, natural_ordering=True).to(dev)
        args = (torch.randn(B, nin, device=dev), torch.randn(B, cond, device=dev))
    else:
        made = MADE(nin, [512, 512], nin * E, num_masks=1, natural_ordering=True).to(dev)
        args = (torch.randn(B, nin, device=dev),)
    with torch.no_grad():
        n0 = _lib.lib().umnn_made_launch_count()
        made.raw(*args)
        assert _lib.lib().umnn_made_launch_count() == n0 + 3 and "made_linear" in _lib.lib().umnn_last_made_kernel_name().decode(), \
            "wide outputs default to one launch of made_linear_kernel per masked linear (up to 4096 rows: the output layer too)"
        umnn_amd.set_made_fused(True, hybrid=True)
        n0 = _lib.lib().umnn_made_launch_count()
        fused = made.raw(*args)
        assert _lib.lib().umnn_made_launch_count() == n0 + 1 and "made_fused" in _lib.lib().umnn_last_made_kernel_name().decode()
        fused16 = made.raw(*args, out_dtype=torch.bfloat16)
        umnn_amd.set_made_fused(False, hybrid=False)
        try:
            per_layer = made.raw(*args)
            umnn_amd.set_made_fast_path(False)
            exact = made.raw(*args)
        finally:
            umnn_amd.set_made_fused(True)
            umnn_amd.set_made_fast_path(True)
    assert fused.shape == exact.shape == (B, nin * E)
    scale = exact.abs().max().item()
    assert (fused - exact).abs().max().item() <= 2e-5 * scale
    assert (fused - per_layer).abs().max().item() <= 1e-5 * scale
    assert fused16.dtype == torch.bfloat16 and (fused16.float() - exact).abs().max().item() <= 6e-3 * scale


@pytest.mark.parametrize("nin,cond,hid,E,B", [(64, 320, [512, 512], 30, 1024),       # the VAE prior flow's conditioner (C4)
                                              (63, 0, [512, 512], 30, 777),            # BSDS300's widths, K0 = 63: scalar operand loads
                                              (20, 12, [100, 37], 30, 19),             # widths off every tile size, one ragged row tile
                                              (40, 0, [64], 20, 1), (33, 7, [512, 256, 300], 16, 300),
                                              (63, 0, [512, 512], 30, 5000)])         # > 4096 rows: the wide output layer on hipBLASLt
@pytest.mark.parametrize("out_dtype", [None, torch.bfloat16])
def test_one_launch_per_masked_linear_matches_the_library_route_and_the_oracle(dev, nin, cond, hid, E, B, out_dtype):
    """umnn_made_linear_forward (made_linear_kernel: one masked linear per launch, grid over row groups x output-tile groups,
    ReLU of the previous layer and the bf16 split in the operand load, ConditionnalMADE's two input blocks read without the cat)
    against the split + library GEMM route (same three bf16 products), the fp32 chain and the float64 oracle of
    models/UMNN/made.py:16-27,113-119,165-168."""
    import umnn_amd
    from umnn_amd import MADE, ConditionnalMADE, _lib
    from umnn_amd.made import MaskedLinear
    torch.manual_seed(nin + B)
    if cond:
        made = ConditionnalMADE(nin, cond, hid, (nin + cond) * E, num_masks=1, natural_ordering=True).to(dev)
        args = (torch.randn(B, nin, device=dev) * 2, torch.randn(B, cond, device=dev))
    else:
        made = MADE(nin, hid, nin * E, num_masks=1, natural_ordering=True).to(dev)
        args = (torch.randn(B, nin, device=dev) * 2,)
    lin = [m for m in made.net if isinstance(m, MaskedLinear)]
    with torch.no_grad():
        for m in lin:
            m.bias.uniform_(-1.0, 1.0)
    assert nin * E > 512
    try:
        with torch.no_grad():
            n0 = _lib.lib().umnn_made_launch_count()
            layered = made.raw(*args, out_dtype=out_dtype)
            n_kernel = len(lin) if B <= 4096 else len(lin) - 1     # (large batches: hidden layers only, the output layer as split + GEMM)
            assert _lib.lib().umnn_made_launch_count() - n0 == n_kernel
            assert "made_linear" in _lib.lib().umnn_last_made_kernel_name().decode()
            again = made.raw(*args, out_dtype=out_dtype)
            umnn_amd.set_made_fused(True, layered=False)
            library = made.raw(*args, out_dtype=out_dtype)
            assert _lib.lib().umnn_made_launch_count() - n0 == 2 * n_kernel
            umnn_amd.set_made_fast_path(False)
            exact = made.raw(*args)
    finally:
        umnn_amd.set_made_fused(True, layered=True)
        umnn_amd.set_made_fast_path(True)
    assert torch.equal(layered, again), "bit-reproducible"
    assert layered.dtype == (out_dtype or torch.float32) and layered.shape == exact.shape == (B, nin * E)
    scale = exact.abs().max().item()
    tol = 2e-5 if out_dtype is None else 6e-3
    assert (layered.float() - exact).abs().max().item() <= tol * scale
    assert (layered.float() - library.float()).abs().max().item() <= (1e-5 if out_dtype is None else 8e-3) * scale
    if not cond:
        ref = O.made_forward([m.weight.detach().cpu().numpy().astype(np.float64) for m in lin],
                             [m.bias.detach().cpu().numpy().astype(np.float64) for m in lin],
                             [m.mask.cpu().numpy().astype(np.float64) for m in lin], args[0][:64].cpu().numpy().astype(np.float64))
        assert np.abs(layered[:64].float().cpu().numpy() - ref).max() <= tol * scale


def test_made_linear_entry_point_over_grid_shapes_and_argument_errors(dev):
    """The C entry itself: every row-tile count x output-tile grouping gives the same numbers (several passes per workgroup,
    groups with no tile, more groups than tiles), ReLU on load, two input blocks; argument errors are reported, not launched."""
    import ctypes
    from umnn_amd import _lib
    from umnn_amd.made import pack_fragments
    lib = _lib.lib()
    torch.manual_seed(3)
    B, K, K1, N = 83, 200, 72, 1000
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    x = torch.randn(B, K, device=dev)
    frags = pack_fragments(W)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    want = torch.relu(x).double() @ W.double().t() + b.double()
    scale = want.abs().max().item()
    x1, x2 = x[:, :K1].contiguous(), x[:, K1:].contiguous()
    first = None
    for rt, fg in [(0, 0), (1, 1), (1, 7), (2, 3), (4, 1), (4, 63), (2, 500), (1, 16)]:
        for two in (False, True):
            out = torch.full((B, N), float("nan"), device=dev)
            _lib.check(lib.umnn_made_linear_forward(frags.data_ptr(), b.data_ptr(), K, N, (x1 if two else x).data_ptr(),
                                                    x2.data_ptr() if two else None, K1 if two else 0, B, 1, out.data_ptr(), 0, rt, fg,
                                                    stream), "made_linear")
            assert (out.double() - want).abs().max().item() <= 2e-5 * scale, (rt, fg, two)
            first = out if first is None else first
            assert (out - first).abs().max().item() <= 2e-6 * scale       # (one or three accumulators per tile: summation order)
    out = torch.empty(B, N, device=dev)
    for bad in [dict(K=513), dict(rt=3), dict(fg=-1), dict(K1=K, two=True), dict(B=-1)]:
        a = dict(K=K, rt=0, fg=0, K1=K1, two=False, B=B)
        a.update(bad)
        rc = lib.umnn_made_linear_forward(frags.data_ptr(), b.data_ptr(), a["K"], N, x.data_ptr(), x2.data_ptr() if a["two"] else None,
                                          a["K1"], a["B"], 0, out.data_ptr(), 0, a["rt"], a["fg"], stream)
        assert rc != 0 and lib.umnn_last_error(), bad
    assert lib.umnn_made_linear_forward(frags.data_ptr(), b.data_ptr(), K, N, x.data_ptr(), None, 0, 0, 0, out.data_ptr(), 0, 0, 0, stream) == 0


def test_gather_indices_built_inside_a_capture_are_not_cached(dev):
    """pack_fragments caches its gather indices per (N, K, device).  Inside a stream capture the kernels that fill them are only
    RECORDED, so an index tensor cached from there holds nothing until that graph is replayed -- and a later eager call with the
    same layer shape would gather with garbage (seen as x-independent conditioner outputs when a captured-but-never-replayed
    GraphedLL came first in a process).  The cache must stay empty for shapes first met inside a capture."""
    from umnn_amd import MADE, made as M
    torch.manual_seed(5)
    M._FRAG_INDEX.clear()
    x = torch.randn(50, 7, device=dev)
    first = MADE(7, [96, 96], 21, num_masks=1, natural_ordering=True).to(dev)
    with torch.no_grad():
        assert M._fast_path_ok(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            first.raw(x)                                   # recorded, never replayed
    assert not any(k[0] == 96 and k[1] == 96 for k in M._FRAG_INDEX), "indices built under capture must not be cached"
    second = MADE(7, [96, 96], 21, num_masks=1, natural_ordering=True).to(dev)
    fast, slow = _both(lambda: second.raw(x))
    assert (fast - slow).abs().max().item() <= 2e-5 * slow.abs().max().item()
    assert any(k[0] == 96 and k[1] == 96 for k in M._FRAG_INDEX)      # (the eager call did cache them)
