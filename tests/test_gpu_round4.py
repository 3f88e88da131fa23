"""Round 4: the workgroup-pipeline backward kernels pinned DIRECTLY to the reference (a fixture produced by the reference at a
size those kernels take) and to the oracle, the fp16-piece pipeline (cc_bwd_ws16_kernel.h) against the float64 oracle at the size
from which it is the default, its overflow fallback, and the launch path with HSA_ENABLE_IPC_MODE_LEGACY unset.

Reference lines: models/UMNN/ParallelNeuralIntegral.py:66-94,110-123 (custom backward), UMNNMAF.py:136-139 (the f(x) term whose
cotangent is g_fx here).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import cc_oracle as O
from tests import _util as U
from tests.test_gpu_forward import build_integrand, t

pytestmark = pytest.mark.gpu
TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _oracle_backward_chunked(onet, x0, x, h, n, g, gfx, chunk=64):
    """oracle.integrate_backward (+ integrand_vjp for the g_fx cotangent) over row chunks in float64: per-row outputs are
    concatenated, d_theta summed."""
    net64 = onet.astype(np.float64)
    outs, dth = [], 0.0
    for lo in range(0, x.shape[0], chunk):
        sl = slice(lo, lo + chunk)
        a = [v[sl].astype(np.float64) for v in (x0, x, h, g)]
        dx0, dx, dh, _, _, flat = O.integrate_backward(net64, a[0], a[1], a[2], n, a[3])
        if gfx is not None:
            vx, vh, vflat = O.integrand_vjp(net64, a[1], a[2], gfx[sl].astype(np.float64))
            dx, dh, flat = dx + vx, dh + vh, flat + vflat
        outs.append((dx0, dx, dh))
        dth = dth + flat
    return [np.concatenate([o[i] for o in outs]) for i in range(3)] + [dth]


def _rows_agree_or_sit_on_a_kink(got, ref, onet, x0, x, h, n, noise, max_rows, what):
    """Per-row outputs: every row inside TOL (of the tensor's largest entry), except rows that are KINK-AMBIGUOUS at the rounding
    noise of the arithmetic under test -- some hidden pre-activation of that row, at some node, smaller than ``noise`` times the
    sum of the magnitudes of its terms (float64, tests/_util.kink_margin_rows) -- and at most ``max_rows`` of those.  There the
    backward multiplies one unit's contribution by 1 or by the slope depending on a sign that rounding decides: in the
    reference's float32 run as much as in the kernel (the reference's own two solvers disagree on such rows)."""
    err = np.abs(np.asarray(got, np.float64) - ref).reshape(ref.shape[0], -1).max(axis=1) / np.abs(ref).max()
    bad = np.nonzero(err > TOL)[0]
    assert len(bad) <= max_rows, (what, len(bad), float(err.max()))
    if len(bad):
        margins = U.kink_margin_rows(onet, x0[bad], x[bad], h[bad], n)
        assert (margins < noise).all(), (what, "a row off by more than the tolerance without a kink inside the noise", bad, margins, err[bad])
    assert np.median(err) < 1e-5, (what, float(np.median(err)))
    return len(bad), float(err.max())


@pytest.mark.parametrize("pieces", ["bf16", "f16"])
def test_workgroup_pipeline_backward_matches_the_reference_fixture(pieces, dev):
    """tests/golden/g8_ws_d63.npz: ParallelNeuralIntegral.apply(...).backward(g) of the REFERENCE at 280 x 63 integrals -- a size
    both workgroup pipelines take.  d_x, d_x0 at 1e-4 of the fixture; d_h at 1e-4 on every row that is not kink-ambiguous at the
    arithmetic's noise level (1e-6 / 3e-6 of the terms' magnitudes) -- the reference's float32 run itself sits on the other side of
    such a kink in one row of this fixture.  d_theta: the six-term bf16 pipeline at the 1e-4 of every other golden test against the
    fixture; the fp16-piece pipeline (forced: at 3.7e5 node evaluations it is not the default) is anchored on TRUTH instead of on
    another float32 run (round 4 allowed it 3e-4 against the fixture): against the float64 oracle it must be inside 1e-4, or no
    further from it than 1.5x the reference's own float32 fixture is."""
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import mlp_spec
    G = U.load("g8_ws_d63")
    net = build_integrand(G, dev)
    spec = mlp_spec(net)
    with _lib.options(bwd_ws=1, bwd_ws16=2 if pieces == "f16" else 0):
        dx0, dx, dh, dth = I.hip_backward(spec, t(G["x0"], dev), t(G["x"], dev), t(G["h"], dev), t(G["g"], dev), None, int(G["n"]))
        torch.cuda.synchronize()
        name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
    assert ",WS>" in name and name.startswith("cc_bwd_" + pieces), name
    assert U.rel_err(dx0.cpu().numpy(), G["dx0_par"]) < TOL
    assert U.rel_err(dx.cpu().numpy(), G["dx_par"]) < TOL
    onet = U.net_from_g2(G)
    _rows_agree_or_sit_on_a_kink(dh.cpu().numpy(), G["dh_par"].astype(np.float64), onet, G["x0"], G["x"], G["h"], int(G["n"]),
                                 noise=1e-6 if pieces == "bf16" else 3e-6, max_rows=3 if pieces == "bf16" else 8, what="dh")
    if pieces == "bf16":
        assert U.scaled_err(dth.cpu().numpy(), G["dtheta_par"]) < TOL
    else:
        a64 = [G[k].astype(np.float64) for k in ("x0", "x", "h", "g")]
        truth = O.integrate_backward(U.net_from_g2(G, np.float64), a64[0], a64[1], a64[2], int(G["n"]), a64[3])[5]
        e_hip, e_ref = U.scaled_err(dth.cpu().numpy(), truth), U.scaled_err(G["dtheta_par"], truth)
        assert e_hip < max(TOL, 1.5 * e_ref), (e_hip, e_ref)


def test_bf16_workgroup_pipeline_matches_the_oracle_directly(dev):
    """the weight-stationary kernel against oracle.integrate_backward + oracle.integrand_vjp themselves (not
    against sibling kernels): 300 x 63 integrals, 31-50^4-1, n = 20, g_fx on, weights x 1.7."""
    import umnn_amd
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import mlp_spec
    B, d, E, n = 300, 63, 30, 20
    torch.manual_seed(B * 7 + d)
    net = umnn_amd.IntegrandNetwork(d, 1 + E, [50] * 4, 1)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.mul_(1.7)
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    onet = O.Net([m.weight.detach().numpy() for m in lin], [m.bias.detach().numpy() for m in lin], O.LEAKY, O.ELU1)
    net.to(dev)
    spec = mlp_spec(net)
    x, x0 = torch.randn(B, d) * 2, torch.randn(B, d) * 0.3
    h, gg, gf = torch.randn(B, E * d), torch.randn(B, d), torch.randn(B, d)
    with _lib.options(bwd_ws=1, bwd_ws16=0):
        out = I.hip_backward(spec, x0.to(dev), x.to(dev), h.to(dev), gg.to(dev), gf.to(dev), n)
        name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
    assert ",WS>" in name and name.startswith("cc_bwd_bf16"), name
    ref = _oracle_backward_chunked(onet, x0.numpy(), x.numpy(), h.numpy(), n, gg.numpy(), gf.numpy())
    assert U.rel_err(out[0].cpu().numpy(), ref[0]) < TOL
    # (d_x carries g_fx . df/dx at node 0 and d_h the whole chain: both go through the kinks of their own row)
    _rows_agree_or_sit_on_a_kink(out[1].cpu().numpy(), ref[1], onet, x0.numpy(), x.numpy(), h.numpy(), n, 1e-6, 3, "dx")
    _rows_agree_or_sit_on_a_kink(out[2].cpu().numpy(), ref[2], onet, x0.numpy(), x.numpy(), h.numpy(), n, 1e-6, 3, "dh")
    # d_theta sums over every row: 1e-4 holds unless some row of this launch is kink-ambiguous at float32 noise (then one unit's
    # contribution at one node may carry the other slope: with random-sign cotangents that is up to ~2e-4 of the largest entry
    # at 4e5 node evaluations -- in the reference's own float32 run as well, cf. the g8 fixture)
    margins = U.kink_margin_rows(onet, x0.numpy(), x.numpy(), h.numpy(), n)
    assert U.scaled_err(out[3].cpu().numpy(), ref[3]) < (TOL if margins.min() > 1e-6 else 5e-4), (float(margins.min()),)


def test_fp16_piece_pipeline_matches_the_float64_oracle_where_it_is_the_default(dev):
    """cc_bwd_ws16_kernel.h is the default from 2^21 node evaluations per launch: 672 x 63 integrals x 101 nodes, 31-50^4-1, g_fx on,
    weights x 1.5, non-zero x0.  Against the oracle in float64 (chunked over rows): d_x0 and d_theta inside 1e-4; d_x and d_h inside
    1e-4 on every row that is not kink-ambiguous at 3e-6 (the recompute's noise on a pre-activation, relative to the magnitudes
    of its terms; fp32 arithmetic: ~1e-6) -- one of the 6.4e8 kink decisions going the other way moves ITS row by up to ~1e-3 of
    the largest entry; the exact-fp32 kernels show the same pattern at 0.4x the rate (tools/kink_rows.py)."""
    import umnn_amd
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import mlp_spec
    B, d, E, n = 672, 63, 30, 100
    torch.manual_seed(5)
    net = umnn_amd.IntegrandNetwork(d, 1 + E, [50] * 4, 1)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.mul_(1.5)
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    onet = O.Net([m.weight.detach().numpy() for m in lin], [m.bias.detach().numpy() for m in lin], O.LEAKY, O.ELU1)
    net.to(dev)
    spec = mlp_spec(net)
    x, x0 = torch.randn(B, d) * 2, torch.randn(B, d) * 0.3
    h, gg, gf = torch.randn(B, E * d), torch.randn(B, d), torch.randn(B, d)
    args = (spec, x0.to(dev), x.to(dev), h.to(dev), gg.to(dev), gf.to(dev), n)
    out = I.hip_backward(*args)                       # (library defaults: bwd_ws = 1, bwd_ws16 = 1)
    name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
    assert name == "cc_bwd_f16<L=4,LIVE=13,WS>", name
    again = I.hip_backward(*args)
    assert all(torch.equal(u, v) for u, v in zip(out, again)), "bit-reproducible (no floating-point atomics)"
    ref = _oracle_backward_chunked(onet, x0.numpy(), x.numpy(), h.numpy(), n, gg.numpy(), gf.numpy(), chunk=32)
    assert U.rel_err(out[0].cpu().numpy(), ref[0]) < TOL
    _rows_agree_or_sit_on_a_kink(out[1].cpu().numpy(), ref[1], onet, x0.numpy(), x.numpy(), h.numpy(), n, 3e-6, 6, "dx")
    _rows_agree_or_sit_on_a_kink(out[2].cpu().numpy(), ref[2], onet, x0.numpy(), x.numpy(), h.numpy(), n, 3e-6, 24, "dh")
    assert U.scaled_err(out[3].cpu().numpy(), ref[3]) < TOL, U.scaled_err(out[3].cpu().numpy(), ref[3])
    # the bf16 pipeline on the same launch, for the record of what the switch changes
    with _lib.options(bwd_ws16=0):
        ob = I.hip_backward(*args)
    assert U.scaled_err(ob[3].cpu().numpy(), ref[3]) < TOL
    assert U.scaled_err(out[3].cpu().numpy(), ob[3].cpu().numpy()) < TOL


def test_fp16_piece_pipeline_falls_back_when_a_piece_overflows(dev):
    """Activations beyond the fp16 range (first-layer weights x 3e4: |a_1| ~ 1e5) overflow the leading piece to inf; the kernel
    raises its device flag and the bf16 pipeline queued behind it rewrites every output: the call returns exactly what the bf16
    pipeline alone returns, and finite values."""
    import umnn_amd
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import mlp_spec
    B, d, E, n = 300, 63, 30, 20
    torch.manual_seed(11)
    net = umnn_amd.IntegrandNetwork(d, 1 + E, [50] * 4, 1).to(dev)
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        lin[0].weight.mul_(3e4)
        lin[1].weight.mul_(1e-4)        # (keeps the deeper layers, and f, in range for fp32)
    spec = mlp_spec(net)
    x, h, gg = torch.randn(B, d, device=dev), torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev)
    with _lib.options(bwd_ws=1, bwd_ws16=2):
        a = I.hip_backward(spec, None, x, h, gg, None, n)
        assert _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode().startswith("cc_bwd_f16")
    with _lib.options(bwd_ws=1, bwd_ws16=0):
        b = I.hip_backward(spec, None, x, h, gg, None, n)
    for u, v in zip(a[1:], b[1:]):
        assert torch.isfinite(u).all() and torch.equal(u, v)
    # and a launch in range right after it is not affected by the flag of the one before
    net2 = umnn_amd.IntegrandNetwork(d, 1 + E, [50] * 4, 1).to(dev)
    spec2 = mlp_spec(net2)
    with _lib.options(bwd_ws=1, bwd_ws16=2):
        c = I.hip_backward(spec2, None, x, h, gg, None, n)
    with _lib.options(bwd_ws=1, bwd_ws16=0):
        e = I.hip_backward(spec2, None, x, h, gg, None, n)
    assert not torch.equal(c[3], e[3]) and U.scaled_err(c[3].cpu().numpy(), e[3].cpu().numpy()) < 3e-4


@pytest.mark.parametrize("hid, with_gfx", [([100, 50, 50, 50, 50], True), ([112, 48, 60, 36, 50], False),
                                           # (first layers of 5, 6 and 8 tiles: the other instantiations of stage A on fp16 pieces)
                                           ([72, 50, 50, 50, 50], False), ([90, 52, 50, 44, 50], True), ([120, 50, 50, 50, 50], True)])
def test_fp16_piece_pipeline_as_the_middle_stage_of_the_three_stage_backward(hid, with_gfx, dev):
    """MNISTExperiment's shape (31-100-50^4-1, d = 784): the middle stage of the three-stage backward (cc_backward_front.hip) on
    fp16 pieces -- z_2 from HBM into wave Ca, delta_2 un-scaled back to HBM from wave B1, single-chunk calls.  Forced here (the
    default starts at 2^21 node evaluations); against the bf16 pipeline on the same inputs row by row, and against the exact-fp32
    kernels; bit-reproducible; the overflow fallback rewrites delta_2 before the front-backward kernel reads it."""
    import umnn_amd
    from umnn_amd import _lib
    from umnn_amd import integral as I
    from umnn_amd.nets import mlp_spec
    B, d, E, n = 40, 784, 30, 12
    torch.manual_seed(11)
    net = umnn_amd.IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.mul_(1.5)
    spec = mlp_spec(net)
    x, x0 = torch.randn(B, d, device=dev) * 2, torch.randn(B, d, device=dev) * 0.3
    h, gg = torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev)
    gf = torch.randn(B, d, device=dev) if with_gfx else None
    outs = {}
    for key, w16, prec in (("bf16", 0, "bf16x3"), ("f16", 2, "bf16x3"), ("fp32", 0, "fp32")):
        _lib.set_backward_precision(prec)
        try:
            with _lib.options(bwd_ws=1, bwd_ws16=w16):
                outs[key] = I.hip_backward(spec, x0, x, h, gg, gf, n)
                name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
                if key == "f16":
                    assert name.startswith("cc_bwd_f16") and "WS,FRONT" in name, name
                    again = I.hip_backward(spec, x0, x, h, gg, gf, n)
                    assert all(torch.equal(u, v) for u, v in zip(outs[key], again))
                elif key == "bf16":
                    assert name.startswith("cc_bwd_bf16") and "WS,FRONT" in name, name
        finally:
            _lib.set_backward_precision("bf16x3")
    for i, nm in enumerate(("dx0", "dx", "dh", "dtheta")):
        a_, b_, r_ = (outs[k][i].cpu().numpy() for k in ("bf16", "f16", "fp32"))
        assert np.isfinite(b_).all(), nm
        if nm == "dtheta":
            assert U.scaled_err(b_, a_) < 3e-4 and U.scaled_err(b_, r_) < 3e-4, (U.scaled_err(b_, a_), U.scaled_err(b_, r_))
        else:
            # per integral (d_h: [B][E][d]): the median at rounding level, a handful of the 31360 integrals on the other side of a
            # LeakyReLU kink (the module docstring; each integral decides 13 x 250 of them)
            per = np.abs(b_ - a_).reshape(B, -1, d).max(axis=1).ravel() / np.abs(a_).max()
            assert np.median(per) < 5e-6 and (per > 1e-4).sum() <= (0 if nm == "dx0" else 12), (nm, float(np.median(per)), int((per > 1e-4).sum()))
            assert U.scaled_err(b_, r_) < (3e-3 if nm == "dh" else 2e-4), (nm, U.scaled_err(b_, r_))
    # overflow in the middle stage (|z_2| ~ 1e5 is beyond the fp16 range), and in stage A already (entries of G1 ~ 3e5: their fp16
    # pieces are inf, z_2 comes out non-finite, the middle stage's checks see it): the bf16 builds of both stages rewrite everything
    for wscale in (3e5, 3e6):
        torch.manual_seed(23)
        net2 = umnn_amd.IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
        lin = [m for m in net2.net if isinstance(m, torch.nn.Linear)]
        with torch.no_grad():
            lin[1].weight.mul_(wscale)
            lin[2].weight.mul_(1.0 / wscale)
        spec2 = mlp_spec(net2)
        with _lib.options(bwd_ws=1, bwd_ws16=2):
            a = I.hip_backward(spec2, x0, x, h, gg, gf, n)
            assert _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode().startswith("cc_bwd_f16")
        with _lib.options(bwd_ws=1, bwd_ws16=0):
            b = I.hip_backward(spec2, x0, x, h, gg, gf, n)
        for u, v in zip(a, b):
            assert torch.isfinite(u).all() and torch.equal(u, v), wscale


@pytest.mark.parametrize("hid, with_gfx, n", [([100, 50, 50, 50, 50], True, 12), ([112, 48, 60, 36], False, 9), ([96, 50, 50], True, 6)])
def test_three_stage_backward_under_fp32_precision_is_the_six_term_build(hid, with_gfx, n, dev):
    """set_backward_precision('fp32') keeps nets with a wide first hidden layer on HIP kernels -- the three-stage
    backward compiled with three bf16 pieces / six cross terms in every product (cc_backward_front_p3.hip) -- instead of leaving
    the library for an ATen chain.  Against the oracle in float64 on a few rows, against the default (three-term) build, chunked
    (a small scratch) against un-chunked, and through autograd with backward_path_taken() == 'hip'."""
    import umnn_amd
    from umnn_amd import _lib
    from umnn_amd import integral as I
    from umnn_amd.nets import mlp_spec
    B, d, E = 24, 37, 30
    torch.manual_seed(7 + len(hid))
    net = umnn_amd.IntegrandNetwork(d, 1 + E, hid, 1)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.mul_(1.4)
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    onet = O.Net([m.weight.detach().numpy() for m in lin], [m.bias.detach().numpy() for m in lin], O.LEAKY, O.ELU1)
    net.to(dev)
    spec = mlp_spec(net)
    x, x0 = torch.randn(B, d) * 2, torch.randn(B, d) * 0.3
    h, gg = torch.randn(B, E * d), torch.randn(B, d)
    gf = torch.randn(B, d) if with_gfx else None
    args = (spec, x0.to(dev), x.to(dev), h.to(dev), gg.to(dev), None if gf is None else gf.to(dev), n)
    outs = {}
    for prec in ("bf16x3", "fp32"):
        _lib.set_backward_precision(prec)
        try:
            I._bwd_kind.clear()
            assert I._hip_backward_ok(spec, args[2], args[3])
            outs[prec] = I.hip_backward(*args)
            name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
            assert "FRONT" in name and name.startswith("cc_bwd_bf16x6" if prec == "fp32" else "cc_bwd_bf16<"), name
            again = I.hip_backward(*args)
            assert all(torch.equal(u, v) for u, v in zip(outs[prec], again))
        finally:
            _lib.set_backward_precision("bf16x3")
    ref = _oracle_backward_chunked(onet, x0.numpy(), x.numpy(), h.numpy(), n, gg.numpy(), None if gf is None else gf.numpy(), chunk=8)
    margins = U.kink_margin_rows(onet, x0.numpy(), x.numpy(), h.numpy(), n)
    for i, nm in enumerate(("dx0", "dx", "dh", "dtheta")):
        a_, b_ = outs["bf16x3"][i].cpu().numpy(), outs["fp32"][i].cpu().numpy()
        assert np.isfinite(b_).all(), nm
        if nm in ("dx", "dh"):
            _rows_agree_or_sit_on_a_kink(b_, ref[i], onet, x0.numpy(), x.numpy(), h.numpy(), n, 1e-6, 2, nm)
        else:
            assert U.scaled_err(b_, ref[i]) < (TOL if margins.min() > 1e-6 or nm == "dx0" else 5e-4), (nm, U.scaled_err(b_, ref[i]))
        assert U.scaled_err(b_, a_) < (3e-3 if nm == "dh" else 3e-4), (nm, U.scaled_err(b_, a_))
    # through autograd, fp32 mode: the HIP path, announced by nothing
    import warnings
    _lib.set_backward_precision("fp32")
    try:
        xs = args[2].clone().requires_grad_()
        hs = args[3].clone().requires_grad_()
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            F = umnn_amd.ParallelNeuralIntegral.apply(args[1], xs, net, umnn_amd.flow._flatten(net.parameters()), hs, n)
            F.backward(args[4])
        assert umnn_amd.backward_path_taken() == "hip"
        assert U.scaled_err(hs.grad.cpu().numpy(), outs["fp32"][2].cpu().numpy()) < 1e-6 or gf is not None
    finally:
        _lib.set_backward_precision("bf16x3")


def test_three_stage_backward_at_the_benchmarked_mnist_size_runs_on_fp16_pieces(dev):
    """MNISTExperiment's shape at the script's batch (100 x 784 integrals, n = 50, 31-100-50^4-1: what `bench.py --workload mnist
    --mode train` times): under the library defaults stages A and B of the three-stage backward run on fp16 pieces.  Rows sampled
    against the oracle in float64 (d_h, d_x depend on their own row), kernel names of the three routes, bit-repeatability.  d_theta is
    held to float64 truth in tests/test_gpu_round5.py::test_default_backward_against_float64_truth_at_the_benchmarked_mnist_size (round
    4 compared it with the six-term build at 2e-4 here; against truth all three routes AND a float32 run of the reference's own algorithm
    sit at 2.8e-4 .. 3.8e-4 at this size, profiles/r05/bwd_truth64_mnist.txt)."""
    import umnn_amd
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import mlp_spec
    torch.manual_seed(0)
    B, d, E, n = 100, 784, 30, 50
    net = umnn_amd.IntegrandNetwork(d, 1 + E, [100, 50, 50, 50, 50], 1)
    with torch.no_grad():
        for mod in net.net:
            if isinstance(mod, torch.nn.Linear):
                mod.weight.mul_(1.5)
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    onet = O.Net([m.weight.detach().numpy() for m in lin], [m.bias.detach().numpy() for m in lin], O.LEAKY, O.ELU1)
    net.to(dev)
    spec = mlp_spec(net)
    x, h = torch.randn(B, d), torch.randn(B, E * d)
    g, gf = torch.randn(B, d), torch.randn(B, d) * 0.1
    args = (spec, None, x.to(dev), h.to(dev), g.to(dev), gf.to(dev), n)
    out = I.hip_backward(*args)
    name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
    assert name == "cc_bwd_f16<L=4,LIVE=13,WS,FRONT>", name
    again = I.hip_backward(*args)
    assert all(torch.equal(u, v) for u, v in zip(out[1:], again[1:]))
    rows = np.array([0, 37, 99])
    ref = _oracle_backward_chunked(onet, np.zeros((3, d), np.float32), x[rows].numpy(), h[rows].numpy(), n, g[rows].numpy(), gf[rows].numpy(), chunk=1)
    for i, nm in ((1, "dx"), (2, "dh")):
        got = out[i][rows].cpu().numpy().astype(np.float64)
        per = np.abs(got - ref[i]).reshape(3, -1, d).max(axis=1).ravel() / np.abs(ref[i]).max()      # per integral
        assert np.median(per) < 5e-6 and (per > TOL).sum() <= 4, (nm, float(np.median(per)), int((per > TOL).sum()), float(per.max()))
    with _lib.options(bwd_ws16=0):
        ob = I.hip_backward(*args)
        assert _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode().startswith("cc_bwd_bf16<")
    _lib.set_backward_precision("fp32")
    try:
        o6 = I.hip_backward(*args)
        assert _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode().startswith("cc_bwd_bf16x6<")
    finally:
        _lib.set_backward_precision("bf16x3")
    assert np.isfinite(out[3].cpu().numpy()).all() and np.isfinite(ob[3].cpu().numpy()).all()
    # (d_x carries g_fx . df/dx at node 0: a LeakyReLU kink inside rounding noise moves its own integral)
    per = (out[1] - o6[1]).abs().cpu().numpy().ravel() / float(o6[1].abs().max())
    assert np.median(per) < 5e-6 and (per > TOL).sum() <= 16, (float(np.median(per)), int((per > TOL).sum()), float(per.max()))


def test_graphed_train_step_with_the_fp16_pipeline_inside_the_graph(dev):
    """GraphedTrainStep at a size where the backward is the fp16-piece pipeline (1024 x 64 integrals x 51 nodes >= 2^21): the
    capture then holds the scalar memset, the cotangent-scale pre-pass, the pipeline and the queued conditional bf16 kernel.  Replays
    must reproduce eager training (same kernels, deterministic): losses and parameters after four steps on changing batches."""
    import copy
    import umnn_amd
    from umnn_amd import _lib
    torch.manual_seed(31)
    model_a = umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=64, hidden_derivative=[50, 50, 50, 50], hidden_embedding=[128, 128], embedding_s=30,
                                   nb_steps=50, device=dev).to(dev)
    model_b = copy.deepcopy(model_a)
    xs = [torch.randn(1024, 64, device=dev) for _ in range(4)]
    opt_a = torch.optim.Adam([p for p in model_a.parameters() if p.requires_grad], lr=1e-3, capturable=True)
    opt_b = torch.optim.Adam([p for p in model_b.parameters() if p.requires_grad], lr=1e-3, capturable=True)
    model_a.train(); model_b.train()
    warm = 1
    losses_a = []
    for x in [xs[0]] * warm + xs:
        opt_a.zero_grad(set_to_none=True)
        ll, _ = model_a.compute_ll(x)
        loss = -ll.mean()
        loss.backward()
        opt_a.step()
        losses_a.append(loss.item())
    assert _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode().startswith("cc_bwd_f16<")
    step = umnn_amd.GraphedTrainStep(model_b, opt_b, xs[0], warmup=warm)
    losses_b = [step(x).item() for x in xs]
    assert max(abs(a - b) for a, b in zip(losses_a[warm:], losses_b)) < 1e-5 * max(1.0, abs(losses_a[-1])), (losses_a, losses_b)
    for (n, pa), (_, pb) in zip(model_a.named_parameters(), model_b.named_parameters()):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-6), n


def test_fp16_piece_pipeline_scales_tiny_and_huge_cotangents(dev):
    """The cotangent scale is a per-launch power of two: gradients are homogeneous in g to the last bit."""
    import umnn_amd
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import mlp_spec
    B, d, E, n = 300, 63, 30, 20
    torch.manual_seed(3)
    net = umnn_amd.IntegrandNetwork(d, 1 + E, [50] * 4, 1).to(dev)
    spec = mlp_spec(net)
    x, h, gg = torch.randn(B, d, device=dev), torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev)
    with _lib.options(bwd_ws=1, bwd_ws16=2):
        base = I.hip_backward(spec, None, x, h, gg, None, n)
        for k in (-40, -17, 23):
            sc = I.hip_backward(spec, None, x, h, gg * 2.0 ** k, None, n)
            for u, v in zip(sc[1:], base[1:]):
                assert torch.equal(u, v * 2.0 ** k), k


def test_bench_launch_path_without_the_ipc_variable(dev):
    """`torch.distributed.run --nproc-per-node 1 bench.py --mode train` with HSA_ENABLE_IPC_MODE_LEGACY REMOVED from
    the child's environment and a forced one-rank RCCL group: bench.py sets the variable itself before the HIP runtime initialises."""
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    env.update(UMNN_FORCE_GROUP="1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--mode", "train", "--workload", "power",
           "--steps", "2", "--warmup", "1", "--no-telemetry"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["dist"]["backend"] == "nccl" and line["dist"]["world_size"] == 1
    assert line["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and line["env"]["set_by"] == "bench.py"


@pytest.mark.parametrize("workload", ["toy", "vae"])
def test_eager_calls_after_an_unreplayed_capture_read_valid_caches(workload, dev):
    """GraphedLL(warmup=0) used to fill the conditioner's masked / packed weight caches from INSIDE the capture
    (kernels recorded, not run): an eager compute_ll, or a second capture, before the first replay then found the cache key
    matching and read uninitialised weights.  Now at least one eager run precedes every capture and nothing is stored while a
    capture records; ConditionnalMADE's kept-row indices (a host -> device copy) are primed too (the vae workload has a context)."""
    import copy
    import bench
    import umnn_amd
    cfg = dict(bench.WORKLOADS[workload], rows=300)
    model = bench.build_model(cfg, dev)
    twin = copy.deepcopy(model)                      # same weights, its own (empty) caches: the eager truth
    xa, ca = bench.make_inputs(cfg, 300, dev, 1)
    xb, cb = bench.make_inputs(cfg, 300, dev, 2)
    with torch.no_grad():
        ea = twin.compute_ll(xa, context=ca)[0] if ca is not None else twin.compute_ll(xa)[0]
        eb = twin.compute_ll(xb, context=cb)[0] if cb is not None else twin.compute_ll(xb)[0]
    ga = umnn_amd.GraphedLL(model, xa, context=ca, warmup=0)          # captured, NOT replayed
    with torch.no_grad():
        got = model.compute_ll(xb, context=cb)[0] if cb is not None else model.compute_ll(xb)[0]
    assert torch.equal(got, eb)
    gb = umnn_amd.GraphedLL(model, xb, context=cb, warmup=0)          # a second capture before the first graph ever ran
    assert torch.equal(gb()[0], eb) and torch.equal(ga()[0], ea)


def test_f16x3_forward_is_fp32_level_at_the_cost_of_bf16x3(dev):
    """set_precision("f16x3"): the forward kernels of cc_fwd_bf16_kernel.h compiled on fp16 pieces (cc_forward_f16.hip) -- two 11-bit
    pieces, three cross terms.  On the golden cases and on the benchmarked launch (8192 x 63, n = 100, sampled rows) F and f(x) are
    within 3e-6 of the float64 oracle (bf16x3: up to ~2e-5; exact fp32 kernels: ~5e-7), on the same software-pipelined kernel."""
    import umnn_amd
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import mlp_spec
    old = umnn_amd.get_forward_precision()
    try:
        errs = {}
        for mode in ("f16x3", "bf16x3"):
            umnn_amd.set_forward_precision(mode)
            worst = 0.0
            for name in ("g2_bsds_d63_w2", "g2_power_d6_w2", "g2_vae_d64", "g2_toy_d2_w2"):
                G = U.load(name)
                spec = mlp_spec(build_integrand(G, dev))
                F, fx, _ = I.hip_forward(spec, t(G["x0"], dev), t(G["x"], dev), t(G["h"], dev), int(G["n"]))
                net64 = U.net_from_g2(G, np.float64)
                F64 = O.integrate_parallel(net64, *(G[k].astype(np.float64) for k in ("x0", "x", "h")), int(G["n"]))
                f64 = O.integrand(net64, G["x"].astype(np.float64), G["h"].astype(np.float64))
                worst = max(worst, U.rel_err(F.cpu().numpy(), F64), U.rel_err(fx.cpu().numpy(), f64))
            errs[mode] = worst
        assert errs["f16x3"] < 3e-6 and errs["f16x3"] < 0.5 * errs["bf16x3"], errs
        # the benchmarked launch
        umnn_amd.set_forward_precision("f16x3")
        torch.manual_seed(0)
        B, d, E, n = 8192, 63, 30, 100
        net = umnn_amd.IntegrandNetwork(d, 1 + E, [50] * 4, 1)
        lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
        onet = O.Net([m.weight.detach().numpy().astype(np.float64) for m in lin], [m.bias.detach().numpy().astype(np.float64) for m in lin],
                     O.LEAKY, O.ELU1)
        net.to(dev)
        x, h = torch.randn(B, d), torch.randn(B, E * d)
        F, fx, _ = I.hip_forward(mlp_spec(net), None, x.to(dev), h.to(dev), n)
        assert _lib.lib().umnn_last_kernel_name().decode() == "cc_fwd_f16<T=4,PARTS=2,P=2,EXACT=1,LIVE=13,PIPE>"
        rows = np.random.RandomState(4).choice(B, 24, replace=False)
        xr, hr = x.numpy()[rows].astype(np.float64), h.numpy()[rows].astype(np.float64)
        assert U.rel_err(F.cpu().numpy()[rows], O.integrate_parallel(onet, np.zeros_like(xr), xr, hr, n)) < 3e-6
        assert U.rel_err(fx.cpu().numpy()[rows], O.integrand(onet, xr, hr)) < 3e-6
    finally:
        umnn_amd.set_forward_precision(old)


def test_f16x3_forward_overflow_is_finite_and_equal_to_bf16x3_on_the_overflowing_rows(dev):
    """fp16 pieces have fp16's exponent range: hidden activations beyond +-65504 overflow the leading piece.  Round 4 detected that
    in the output-layer sum and returned NaN (which kept the mode opt-in); since round 5 the affected tile groups are deferred to
    the bf16x3 build queued behind the launch (tests/test_gpu_round5.py has the protocol's own tests): finite everywhere, the
    overflowing rows bit-equal to a bf16x3 launch, every row inside 1e-4 of the exact-fp32 kernels."""
    import umnn_amd
    from umnn_amd import integral as I
    from umnn_amd.nets import mlp_spec
    torch.manual_seed(2)
    net = umnn_amd.IntegrandNetwork(6, 31, [50] * 4, 1).to(dev)
    spec = mlp_spec(net)
    x, h = torch.randn(500, 6, device=dev), torch.randn(500, 180, device=dev)
    x[:12] *= 3e6                      # a dozen rows whose first-layer activations reach ~1e5 .. 1e6 along the quadrature nodes
    old = umnn_amd.get_forward_precision()
    try:
        umnn_amd.set_forward_precision("bf16x3")
        Fb, fb, _ = I.hip_forward(spec, None, x, h, 30)
        umnn_amd.set_forward_precision("fp32")
        Fe, _, _ = I.hip_forward(spec, None, x, h, 30)
        umnn_amd.set_forward_precision("f16x3")
        Ff, ff, _ = I.hip_forward(spec, None, x, h, 30)
    finally:
        umnn_amd.set_forward_precision(old)
    assert torch.isfinite(Fb).all() and torch.isfinite(fb).all() and torch.isfinite(Fe).all()
    assert torch.isfinite(Ff).all() and torch.isfinite(ff).all()
    assert torch.equal(Ff[:12], Fb[:12]) and torch.equal(ff[:12], fb[:12])
    rel = lambda A: float(((A - Fe).abs() / Fe.abs().clamp(min=1.0)).max())
    assert rel(Ff) < 1e-4 and rel(Ff) <= rel(Fb), (rel(Ff), rel(Fb))
