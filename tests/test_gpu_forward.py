"""Parity of the HIP forward path (through the C ABI) against the CPU oracle and the reference's golden vectors.

Tolerance (SURVEY 8d / north_star): max |d|/max(|ref|,1) <= 1e-4 on F and f(x); log-det compared as
|d log f| <= 1e-4 * max(1, |log f|).  Measured noise is ~1e-6 (fp32 MFMA is an exact fmaf chain; the only
differences are summation order and v_exp_f32).
"""
import os

import numpy as np
import pytest
import torch

from oracle import cc_oracle as O
from tests import _util as U

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True, params=["bf16x3", "bf16x6", "fp32", "f16x3"])
def precision(request):
    """Every forward-parity test runs under all four arithmetic modes of the forward kernels (same tolerance)."""
    import umnn_amd
    old = umnn_amd.get_forward_precision()
    umnn_amd.set_forward_precision(request.param)
    yield request.param
    umnn_amd.set_forward_precision(old)


def build_integrand(G, dev):
    from umnn_amd import IntegrandNetwork
    hid = [int(v) for v in G["hidden"]]
    net = IntegrandNetwork(int(G["d"]), 1 + int(G["E"]), hid, 1, act_func=str(G["act"]))
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        for l, m in enumerate(lin):
            m.weight.copy_(torch.from_numpy(G[f"W{l}"]))
            m.bias.copy_(torch.from_numpy(G[f"b{l}"]))
    return net.to(dev)


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("name", U.g2_names())
def test_forward_matches_golden_and_oracle(name, dev, precision):
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import mlp_spec
    G = U.load(name)
    net = build_integrand(G, dev)
    spec = mlp_spec(net)
    assert spec is not None
    n = int(G["n"])
    before = _lib.lib().umnn_launch_count()
    F, fx, fx0 = I.hip_forward(spec, t(G["x0"], dev), t(G["x"], dev), t(G["h"], dev), n)
    torch.cuda.synchronize()
    assert _lib.lib().umnn_launch_count() == before + 1
    assert U.rel_err(F.cpu().numpy(), G["F_par"]) < TOL
    assert U.rel_err(F.cpu().numpy(), G["F_seq"]) < TOL
    assert U.rel_err(fx.cpu().numpy(), G["f_x"]) < TOL
    assert U.rel_err(fx0.cpu().numpy(), G["f_x0"]) < TOL
    lf, lref = np.log(fx.cpu().numpy() + 1e-10), np.log(G["f_x"] + 1e-10)
    assert np.all(np.abs(lf - lref) <= TOL * np.maximum(1.0, np.abs(lref)))
    Finv, _, _ = I.hip_forward(spec, t(G["x0"], dev), t(G["x"], dev), t(G["h"], dev), n, inv_f=True)
    assert U.rel_err(Finv.cpu().numpy(), G["F_inv"]) < TOL
    # fp64 oracle: how far from the exact quadrature of the same weights
    net64 = U.net_from_g2(G, np.float64)
    F64 = O.integrate_parallel(net64, G["x0"].astype(np.float64), G["x"].astype(np.float64),
                               G["h"].astype(np.float64), n)
    assert U.rel_err(F.cpu().numpy(), F64) < (4e-5 if precision == "bf16x3" else 2e-5)
    kname = _lib.lib().umnn_last_kernel_name().decode()
    if precision != "fp32" and len(G["hidden"]) >= 2 and max(int(v) for v in G["hidden"]) <= 63:
        family = "cc_fwd_f16" if precision == "f16x3" else "cc_fwd_bf16"
        assert family in kname and ("PARTS=3" if precision == "bf16x6" else "PARTS=2") in kname
    if precision == "fp32":
        assert "bf16" not in kname


@pytest.mark.parametrize("TAIL", [0, 1])
@pytest.mark.parametrize("P,NS", [(1, 1), (2, 1), (1, 2), (1, 4), (2, 4), (2, 2)])
@pytest.mark.parametrize("name", ["g2_power_d6_w2", "g2_toy_d2_w2", "g2_mnist_mixed_d8", "g2_odd_n_d3", "g2_bsds_d63_w2"])
def test_every_kernel_variant_agrees(name, P, NS, TAIL, dev, opts, precision):
    """Point tiles per wave (P), node-split factor (NS) and the VALU-tail variant are launch choices: all must give
    the same answer."""
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import mlp_spec
    opts(fwd_p=P, fwd_ns=NS, fwd_tail=TAIL)
    G = U.load(name)
    net = build_integrand(G, dev)
    F, fx, fx0 = I.hip_forward(mlp_spec(net), t(G["x0"], dev), t(G["x"], dev), t(G["h"], dev), int(G["n"]))
    kname = _lib.lib().umnn_last_kernel_name().decode()
    assert f"P={P}" in kname
    if precision == "fp32" and ("power" in name or "bsds" in name):
        assert f"TAIL={TAIL}" in kname
    assert U.rel_err(F.cpu().numpy(), G["F_par"]) < TOL
    assert U.rel_err(fx.cpu().numpy(), G["f_x"]) < TOL
    assert U.rel_err(fx0.cpu().numpy(), G["f_x0"]) < TOL


@pytest.mark.parametrize("B,d,E,hid,n,relu", [
    (1, 1, 2, [100, 100, 100], 50, True),      # MonotonicNN shape, a single integral
    (37, 3, 1, [20, 20], 20, False),           # ragged: 111 integrals, not a multiple of 16
    (5, 63, 30, [50] * 4, 100, False),
    (3, 7, 4, [127], 9, False),                # widest supported layer, odd node count
    (9, 2, 10, [31, 63, 15], 33, True),        # mixed widths on tile boundaries
    (4, 5, 2, [8, 8, 8, 8, 8, 8, 8], 12, False),   # deepest supported MLP
])
def test_ragged_shapes_against_oracle(B, d, E, hid, n, relu, dev):
    from umnn_amd import integral as I
    from umnn_amd.nets import MlpSpec
    from umnn_amd import _lib
    rng = np.random.RandomState(B * 131 + d)
    sizes = [1 + E] + hid + [1]
    Ws = [(rng.randn(sizes[i + 1], sizes[i]) * (1.6 / np.sqrt(sizes[i]))).astype(np.float32) for i in range(len(sizes) - 1)]
    bs = [(rng.randn(sizes[i + 1]) * 0.3).astype(np.float32) for i in range(len(sizes) - 1)]
    lin = []
    for W, b in zip(Ws, bs):
        m = torch.nn.Linear(W.shape[1], W.shape[0])
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(W))
            m.bias.copy_(torch.from_numpy(b))
        lin.append(m.to(dev))
    spec = MlpSpec(lin, _lib.ACT_RELU if relu else _lib.ACT_LEAKY_RELU, _lib.OUT_ELU_PLUS_ONE)
    x = (rng.randn(B, d) * 2).astype(np.float32)
    x0 = (rng.randn(B, d) * 0.5).astype(np.float32)
    h = rng.randn(B, E * d).astype(np.float32)
    net = O.Net(Ws, bs, O.RELU if relu else O.LEAKY, O.ELU1)
    F, fx, fx0 = I.hip_forward(spec, t(x0, dev), t(x, dev), t(h, dev), n)
    assert U.rel_err(F.cpu().numpy(), O.integrate_parallel(net, x0, x, h, n)) < TOL
    assert U.rel_err(fx.cpu().numpy(), O.integrand(net, x, h)) < TOL
    assert U.rel_err(fx0.cpu().numpy(), O.integrand(net, x0, h)) < TOL
    # x0 = None means zeros
    F0, _, _ = I.hip_forward(spec, None, t(x, dev), t(h, dev), n)
    assert U.rel_err(F0.cpu().numpy(), O.integrate_parallel(net, np.zeros_like(x), x, h, n)) < TOL


def test_empty_batch(dev):
    from umnn_amd import integral as I, IntegrandNetwork
    from umnn_amd.nets import mlp_spec
    net = IntegrandNetwork(3, 2, [16, 16], 1).to(dev)
    F, fx, fx0 = I.hip_forward(mlp_spec(net), None, torch.zeros(0, 3, device=dev), torch.zeros(0, 3, device=dev), 10)
    assert F.shape == (0, 3)


@pytest.mark.parametrize("name", U.g4_names())
def test_fused_flow_block_epilogue(name, dev):
    """z = exp(s)(F + h_0) and log_jac = log(f+1e-10)+s from the fused entry point, block by block vs the oracle."""
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import MlpSpec
    G = U.load(name)
    blocks = U.blocks_from_g4(G)
    x = G["x"]
    ctx = G.get("context")
    n = int(G["n"])
    for i, blk in enumerate(blocks):
        blk.scaling = (np.linspace(-0.3, 0.4, x.shape[1])).astype(np.float32)      # exercise exp(s), + s
        z_ref, h = O.block_forward(blk, x, n, context=ctx)
        lj_ref = O.block_log_jac(blk, x, h)
        lin = []
        for W, b in zip(blk.net.Ws, blk.net.bs):
            m = torch.nn.Linear(W.shape[1], W.shape[0])
            with torch.no_grad():
                m.weight.copy_(torch.from_numpy(W))
                m.bias.copy_(torch.from_numpy(b))
            lin.append(m.to(dev))
        spec = MlpSpec(lin, _lib.ACT_LEAKY_RELU, _lib.OUT_ELU_PLUS_ONE)
        z, lj, fx, fx0 = I.hip_flow_block(spec, t(x, dev), t(h, dev), t(blk.scaling, dev), n)
        assert U.rel_err(z.cpu().numpy(), z_ref) < TOL
        assert np.all(np.abs(lj.cpu().numpy() - lj_ref) <= TOL * np.maximum(1.0, np.abs(lj_ref)))
        x = z_ref[:, ::-1].copy()


@pytest.mark.parametrize("name", U.g4_names())
def test_flow_module_eval_matches_reference(name, dev):
    """The nn.Module API on the GPU (HIP path) reproduces the reference's compute_ll / forward / log-jac."""
    import umnn_amd
    G = U.load(name)
    m = umnn_amd.UMNNMAFFlow(nb_flow=int(G["nb_flow"]), nb_in=int(G["d"]),
                             hidden_derivative=[int(v) for v in G["hidden_derivative"]],
                             hidden_embedding=[int(v) for v in G["hidden_embedding"]], embedding_s=int(G["E"]),
                             nb_steps=int(G["n"]), solver=str(G["solver"]), cond_in=int(G["cond_in"]))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in U.state_dict_of(G).items()})
    m.to(dev).eval()
    x = t(G["x"], dev)
    ctx = t(G["context"], dev) if "context" in G else None
    with torch.no_grad():
        ll, z = m.compute_ll(x, context=ctx)
        assert umnn_amd.path_taken() == "hip"
        zb, lj = m.compute_log_jac_bis(x, context=ctx)
        fwd = m.forward(x, context=ctx)
        bpp, _, _ = m.compute_bpp(x, context=ctx)
    assert U.rel_err(ll.cpu().numpy(), G["ll_eval"]) < TOL
    assert U.rel_err(z.cpu().numpy(), G["z_eval"]) < TOL
    assert U.rel_err(zb.cpu().numpy(), G["z_bis_eval"]) < TOL
    assert U.rel_err(lj.cpu().numpy(), G["log_jac_bis_eval"]) < TOL
    assert U.rel_err(fwd.cpu().numpy(), G["fwd_eval"]) < TOL
    assert U.rel_err(bpp.cpu().numpy(), G["bpp_eval"]) < TOL


def test_full_size_properties_bsds300_shard(dev):
    """BASELINE config C3 at one GPU's share (8192 x 63, n=100, 31-50^4-1): too big for the oracle in full, so
    check (a) a random sample of rows against the oracle (rows are independent), (b) shard consistency: the
    result of the full batch equals the concatenation of two half batches bit-for-bit, (c) F(x0=x) = 0,
    (d) monotonicity in x, (e) determinism."""
    from umnn_amd import integral as I, IntegrandNetwork
    from umnn_amd.nets import mlp_spec
    torch.manual_seed(0)
    B, d, E, n = 8192, 63, 30, 100
    net = IntegrandNetwork(d, 1 + E, [50] * 4, 1)
    with torch.no_grad():
        for m in net.net:
            if isinstance(m, torch.nn.Linear):
                m.weight.mul_(1.7)
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    onet = O.Net([m.weight.detach().numpy() for m in lin], [m.bias.detach().numpy() for m in lin], O.LEAKY, O.ELU1)
    net.to(dev)
    spec = mlp_spec(net)
    x = torch.randn(B, d)
    h = torch.randn(B, E * d)
    xg, hg = x.to(dev), h.to(dev)
    F, fx, fx0 = I.hip_forward(spec, None, xg, hg, n)
    rows = np.random.RandomState(1).choice(B, 48, replace=False)
    Fo = O.integrate_parallel(onet, np.zeros((48, d), np.float32), x.numpy()[rows], h.numpy()[rows], n)
    assert U.rel_err(F.cpu().numpy()[rows], Fo) < TOL
    assert U.rel_err(fx.cpu().numpy()[rows], O.integrand(onet, x.numpy()[rows], h.numpy()[rows])) < TOL
    Fa, _, _ = I.hip_forward(spec, None, xg[:B // 2].contiguous(), hg[:B // 2].contiguous(), n)
    Fb, _, _ = I.hip_forward(spec, None, xg[B // 2:].contiguous(), hg[B // 2:].contiguous(), n)
    assert torch.equal(torch.cat([Fa, Fb]), F)
    Fz, _, _ = I.hip_forward(spec, xg, xg, hg, n)
    assert float(Fz.abs().max()) == 0.0
    Fup, _, _ = I.hip_forward(spec, None, xg + 0.25, hg, n)
    assert bool((Fup > F - 1e-4).all()) and float((Fup > F).float().mean()) > 0.999   # f > 0 => F increasing in x
    F2, _, _ = I.hip_forward(spec, None, xg, hg, n)
    assert torch.equal(F2, F)


def test_invert_on_gpu_matches_reference(dev):
    """Sampling path: per-dimension bracket search whose integrals run on the HIP kernel (d = 1)."""
    import umnn_amd
    G = U.load("g6_invert")
    m = umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=2, hidden_derivative=[50] * 3, hidden_embedding=[32, 32],
                             embedding_s=10, nb_steps=30, solver="CCParallel")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in U.state_dict_of(G).items()})
    m.to(dev).eval()
    x = t(G["x"], dev)
    with torch.no_grad():
        z = m(x)
        x_inv = m.invert(t(G["z"], dev), iter=5)
    assert umnn_amd.path_taken() == "hip"
    assert U.rel_err(z.cpu().numpy(), G["z"]) < TOL
    tol = 2 * 100. / 9 ** 5
    assert float((x_inv - x).abs().max()) < tol
    assert float((x_inv.cpu() - torch.from_numpy(G["x_inv"])).abs().max()) < tol


@pytest.mark.parametrize("d,hid,E,n,nb_flow,B", [(7, [50] * 4, 30, 50, 2, 33), (2, [100] * 4, 10, 50, 1, 64),
                                                 (5, [100, 50, 50, 50, 50], 8, 30, 1, 20), (3, [40, 33], 4, 20, 2, 17)])
def test_in_kernel_inversion_round_trip(d, hid, E, n, nb_flow, B, dev, precision):
    """UMNNMAFFlow.invert with the whole bracket search of a dimension inside one launch (umnn_flow_invert_dim): exactly
    d launches per block, x -> z -> x round trip within the search's own resolution 100 (2/9)^iter, and agreement with the
    host-driven search (the same algorithm issued round by round through the generic quadrature)."""
    import umnn_amd
    from umnn_amd import _lib, integral as I
    torch.manual_seed(d * 7 + len(hid))
    m = umnn_amd.UMNNMAFFlow(nb_flow=nb_flow, nb_in=d, hidden_derivative=hid, hidden_embedding=[64, 64], embedding_s=E,
                             nb_steps=n, solver="CCParallel").to(dev).eval()
    x = torch.randn(B, d, device=dev) * 1.5
    iters = 8
    tol = 4 * 100.0 * (2.0 / 9.0) ** iters            # two bracket widths per block of slack
    with torch.no_grad():
        z = m(x)
        before = _lib.lib().umnn_launch_count()
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)      # (exact-products modes announce the host-driven search once)
            x_inv = m.invert(z, iter=iters)
        wide = max(hid) > 63
        # one launch per dimension and block in every mode.  Under "exact products" (bf16x6 / fp32) nets of up to four tiles per layer
        # run the search with three bf16 pieces / six cross terms (round 4: PARTS=3 variants), wider ones -- whose three-piece form does
        # not fit the register file -- on two fp16 pieces like f16x3 (the default), every fp16 launch with its bf16x3 build queued as
        # the overflow fallback (round 5; until then those nets fell back to a host-driven search)
        assert _lib.lib().umnn_launch_count() - before == nb_flow * d
        name = _lib.lib().umnn_last_kernel_name().decode()
        on_f16 = precision == "f16x3" or (precision in ("bf16x6", "fp32") and wide)
        assert ("cc_invert_f16" if on_f16 else "cc_invert_bf16") in name, name
        assert ("PARTS=3" in name) == (precision in ("bf16x6", "fp32") and not wide), name
        if hid[0] == 100 and len(hid) == 5:
            assert "T1=7,TREST=4" in name
        assert float((x_inv - x).abs().max()) < tol * nb_flow
        with I.force_generic():                       # host-driven search, ATen integrals
            x_ref = m.invert(z, iter=iters)
        assert float((x_inv - x_ref).abs().max()) < tol * nb_flow
        z2 = m(x_inv)
    assert U.rel_err(z2.cpu().numpy(), z.cpu().numpy()) < 5e-3


def test_large_batch_65536_rows(dev):
    """The un-sharded C3 batch (65536 x 63 = 4.1M integrals) on one GPU: indexing stays in range, results finite,
    first and last rows agree with a small launch on the same rows."""
    from umnn_amd import integral as I, IntegrandNetwork
    from umnn_amd.nets import mlp_spec
    torch.manual_seed(3)
    B, d, E, n = 65536, 63, 30, 20
    net = IntegrandNetwork(d, 1 + E, [50] * 4, 1).to(dev)
    spec = mlp_spec(net)
    x = torch.randn(B, d, device=dev)
    h = torch.randn(B, E * d, device=dev)
    F, fx, fx0 = I.hip_forward(spec, None, x, h, n)
    assert bool(torch.isfinite(F).all()) and bool((fx > 0).all())
    for rows in (slice(0, 64), slice(B - 64, B)):
        Fs, _, _ = I.hip_forward(spec, None, x[rows].contiguous(), h[rows].contiguous(), n)
        assert U.rel_err(Fs.cpu().numpy(), F[rows].cpu().numpy()) < 2e-5


def _random_case(seed):
    rng = np.random.RandomState(seed)
    depth = int(rng.randint(1, 8))                          # 1..7 hidden layers
    hid = [int(rng.choice([rng.randint(1, 128), 50, 16, 48, 63, 64, 100, 127])) for _ in range(depth)]
    d = int(rng.choice([1, 2, 3, 6, 17, 63]))
    E = int(rng.choice([1, 2, 10, 30, 37]))
    n = int(rng.choice([1, 2, 7, 20, 51]))
    B = int(rng.randint(1, 40))
    relu = bool(rng.randint(0, 2))
    sigmoid = bool(rng.randint(0, 4) == 0)
    return depth, hid, d, E, n, B, relu, sigmoid


@pytest.mark.parametrize("seed", range(24))
def test_random_mlp_shapes_forward_and_backward(seed, dev):
    """Random depth / widths / d / E / n / batch: whatever kernel family the launcher picks (exact, generic, bf16 or
    the fp32 fallback when LDS does not fit) must agree with the oracle, forward and backward."""
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import MlpSpec
    depth, hid, d, E, n, B, relu, sigmoid = _random_case(seed)
    rng = np.random.RandomState(1000 + seed)
    sizes = [1 + E] + hid + [1]
    Ws = [(rng.randn(sizes[i + 1], sizes[i]) * (1.5 / np.sqrt(sizes[i]))).astype(np.float32) for i in range(len(sizes) - 1)]
    bs = [(rng.randn(sizes[i + 1]) * 0.3).astype(np.float32) for i in range(len(sizes) - 1)]
    lin = []
    for W, b in zip(Ws, bs):
        m = torch.nn.Linear(W.shape[1], W.shape[0])
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(W))
            m.bias.copy_(torch.from_numpy(b))
        lin.append(m.to(dev))
    spec = MlpSpec(lin, _lib.ACT_RELU if relu else _lib.ACT_LEAKY_RELU,
                   _lib.OUT_SIGMOID if sigmoid else _lib.OUT_ELU_PLUS_ONE)
    onet = O.Net(Ws, bs, O.RELU if relu else O.LEAKY, O.SIGMOID if sigmoid else O.ELU1)
    x = (rng.randn(B, d) * 2).astype(np.float32)
    x0 = (rng.randn(B, d) * 0.5).astype(np.float32)
    h = rng.randn(B, E * d).astype(np.float32)
    g = rng.randn(B, d).astype(np.float32)
    try:
        F, fx, fx0 = I.hip_forward(spec, t(x0, dev), t(x, dev), t(h, dev), n)
    except RuntimeError as e:                               # only "does not fit LDS" may refuse, and it must say so
        assert "LDS" in str(e), str(e)
        return
    assert U.rel_err(F.cpu().numpy(), O.integrate_parallel(onet, x0, x, h, n)) < TOL
    assert U.rel_err(fx.cpu().numpy(), O.integrand(onet, x, h)) < TOL
    assert U.rel_err(fx0.cpu().numpy(), O.integrand(onet, x0, h)) < TOL
    try:
        dx0, dx, dh, dth = I.hip_backward(spec, t(x0, dev), t(x, dev), t(h, dev), t(g, dev), None, n)
    except RuntimeError as e:
        assert "LDS" in str(e), str(e)
        return
    rdx0, rdx, rdh, _, _, rflat = O.integrate_backward(onet, x0, x, h, n, g)
    assert U.rel_err(dx0.cpu().numpy(), rdx0) < TOL and U.rel_err(dx.cpu().numpy(), rdx) < TOL
    # gradients: the stated 1e-4 -- unless some hidden pre-activation of this very case sits inside the rounding noise
    # of its own dot product (kink distance below 5e-7 of sum|terms|, i.e. a few ulps of the fp32 dot product): there the SIGN, hence act'(z), is decided by
    # summation order in the reference as much as here, and one flipped unit at one of the few points of these tiny
    # batches moves the gradient by more than 1e-4.  Only such kink-ambiguous cases get the wider bound.
    tol_g = TOL if U.kink_margin(onet, x0, x, h, n) > 5e-7 else 5e-4
    assert U.scaled_err(dh.cpu().numpy(), rdh) < tol_g
    assert U.scaled_err(dth.cpu().numpy(), rflat) < tol_g


def test_half_precision_callers(dev):
    """bf16 / fp16 tensors (autocast, the bf16 VAE-prior configuration): converted at the boundary, fp32 inside, the
    caller's dtype outside; gradients flow.  Tolerance = the I/O dtype's own resolution."""
    import umnn_amd
    from umnn_amd import integral as I
    from umnn_amd.nets import mlp_spec
    G = U.load("g2_vae_d64")
    net = build_integrand(G, dev)
    spec = mlp_spec(net)
    n = int(G["n"])
    for dt, tol in ((torch.bfloat16, 2e-2), (torch.float16, 3e-3)):
        x = t(G["x"], dev).to(dt)
        h = t(G["h"], dev).to(dt)
        F, fx, fx0 = I.hip_forward(spec, None, x, h, n)
        assert F.dtype == dt and umnn_amd.path_taken() == "hip"
        ref = O.integrate_parallel(U.net_from_g2(G), np.zeros_like(G["x"]), x.float().cpu().numpy(), h.float().cpu().numpy(), n)
        assert U.rel_err(F.float().cpu().numpy(), ref) < tol
        xr = x.clone().requires_grad_()
        hr = h.clone().requires_grad_()
        flat = torch.cat([p.contiguous().view(-1) for p in net.parameters()])
        out = umnn_amd.ParallelNeuralIntegral.apply(torch.zeros_like(xr), xr, net, flat, hr, n)
        out.float().sum().backward()
        assert xr.grad.dtype == dt and hr.grad.dtype == dt and bool(torch.isfinite(hr.grad.float()).all())


def test_pipelined_and_plain_bf16x3_kernels_agree_bit_for_bit(dev, opts):
    """The software-pipelined node loop (default for bf16x3, two point tiles per wave) issues the same MFMAs per
    accumulator in the same order as the plain loop it replaces (option fwd_pipe=0): the two must return identical
    bits."""
    import umnn_amd
    from umnn_amd import integral as I, _lib, IntegrandNetwork
    from umnn_amd.nets import mlp_spec
    if umnn_amd.get_forward_precision() not in ("bf16x3", "f16x3"):
        pytest.skip("the pipelined loop exists for the two-piece modes only (bf16x3, f16x3)")
    torch.manual_seed(5)
    net = IntegrandNetwork(7, 31, [50, 50, 50, 50], 1).to(dev)
    x = torch.randn(300, 7, device=dev) * 2
    h = torch.randn(300, 30 * 7, device=dev)
    out = {}
    for pipe in (1, 0):
        opts(fwd_pipe=pipe, fwd_p=2)
        F, fx, fx0 = I.hip_forward(mlp_spec(net), None, x, h, 100)
        torch.cuda.synchronize()
        out[pipe] = (F.cpu().numpy(), fx.cpu().numpy(), fx0.cpu().numpy(), _lib.lib().umnn_last_kernel_name().decode())
    assert "PIPE" in out[1][3] and "PIPE" not in out[0][3], (out[1][3], out[0][3])
    for a, b in zip(out[1][:3], out[0][:3]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("hid,relu,sigmoid,inv_f,n,NS", [
    ([50, 50], False, False, False, 100, 1),       # one hidden->hidden layer: first section + last section only
    ([50, 50, 50], True, False, False, 51, 2),     # ReLU, odd node count, node range split over two waves
    ([56, 56, 56, 56], False, True, False, 40, 1),  # widths 52..62: all 16 registers live; Sigmoid output
    ([63, 63, 63], False, False, True, 30, 4),     # full tiles (63 + the constant feature), 1/f integrand
    ([48, 49, 50, 51, 50], False, False, False, 20, 1),   # mixed widths inside the 4-tile family, 5 hidden layers
])
def test_pipelined_kernel_shapes_against_oracle(hid, relu, sigmoid, inv_f, n, NS, dev, opts):
    """The software-pipelined bf16x3 loop (two point tiles per wave) over its whole shape family, against the oracle."""
    import umnn_amd
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import MlpSpec
    if umnn_amd.get_forward_precision() not in ("bf16x3", "f16x3"):
        pytest.skip("the pipelined loop exists for the two-piece modes only (bf16x3, f16x3)")
    opts(fwd_p=2, fwd_ns=NS)
    B, d, E = 23, 5, 7
    rng = np.random.RandomState(len(hid) * 17 + n)
    sizes = [1 + E] + hid + [1]
    Ws = [(rng.randn(sizes[i + 1], sizes[i]) * (1.6 / np.sqrt(sizes[i]))).astype(np.float32) for i in range(len(sizes) - 1)]
    bs = [(rng.randn(sizes[i + 1]) * 0.3).astype(np.float32) for i in range(len(sizes) - 1)]
    lin = []
    for W, b in zip(Ws, bs):
        m = torch.nn.Linear(W.shape[1], W.shape[0])
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(W))
            m.bias.copy_(torch.from_numpy(b))
        lin.append(m.to(dev))
    spec = MlpSpec(lin, _lib.ACT_RELU if relu else _lib.ACT_LEAKY_RELU, _lib.OUT_SIGMOID if sigmoid else _lib.OUT_ELU_PLUS_ONE)
    net = O.Net(Ws, bs, O.RELU if relu else O.LEAKY, O.SIGMOID if sigmoid else O.ELU1)
    x = (rng.randn(B, d) * 2).astype(np.float32)
    x0 = (rng.randn(B, d) * 0.5).astype(np.float32)
    h = rng.randn(B, E * d).astype(np.float32)
    F, fx, fx0 = I.hip_forward(spec, t(x0, dev), t(x, dev), t(h, dev), n, inv_f=inv_f)
    kname = _lib.lib().umnn_last_kernel_name().decode()
    assert "PIPE" in kname, kname
    assert U.rel_err(F.cpu().numpy(), O.integrate_parallel(net, x0, x, h, n, inv_f=inv_f)) < TOL
    assert U.rel_err(fx.cpu().numpy(), O.integrand(net, x, h)) < TOL
    assert U.rel_err(fx0.cpu().numpy(), O.integrand(net, x0, h)) < TOL


@pytest.mark.parametrize("hid,relu,sigmoid,inv_f,n", [
    ([50, 50], False, False, False, 100),            # one hidden->hidden layer
    ([50, 50, 50], True, False, False, 51),          # ReLU, odd node count
    ([48, 51, 50, 49], False, True, False, 40),      # the flagship depth, mixed widths 48..51, Sigmoid output
    ([51, 51, 51, 51, 51], False, False, True, 20),  # five hidden layers of the widest member, 1/f integrand
])
def test_mfma_32x32x16_formulation_against_oracle(hid, relu, sigmoid, inv_f, n, dev, opts):
    """The opt-in 32x32x16 formulation of the 48..51-wide shapes (cc_forward_p32.hip: the wave's 32 integrals are the N dimension
    of one matrix instruction, pipelined in 20-slot sections): against the oracle, ragged batch, x0 != 0, and through the flow
    entry point (same epilogue as every other forward kernel)."""
    import umnn_amd
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import MlpSpec
    if umnn_amd.get_forward_precision() != "bf16x3":
        pytest.skip("the 32x32x16 formulation exists for bf16x3 only")
    opts(fwd_pipe=2)
    B, d, E = 37, 5, 7
    rng = np.random.RandomState(len(hid) * 19 + n)
    sizes = [1 + E] + hid + [1]
    Ws = [(rng.randn(sizes[i + 1], sizes[i]) * (1.6 / np.sqrt(sizes[i]))).astype(np.float32) for i in range(len(sizes) - 1)]
    bs = [(rng.randn(sizes[i + 1]) * 0.3).astype(np.float32) for i in range(len(sizes) - 1)]
    lin = []
    for W, b in zip(Ws, bs):
        m = torch.nn.Linear(W.shape[1], W.shape[0])
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(W))
            m.bias.copy_(torch.from_numpy(b))
        lin.append(m.to(dev))
    spec = MlpSpec(lin, _lib.ACT_RELU if relu else _lib.ACT_LEAKY_RELU, _lib.OUT_SIGMOID if sigmoid else _lib.OUT_ELU_PLUS_ONE)
    net = O.Net(Ws, bs, O.RELU if relu else O.LEAKY, O.SIGMOID if sigmoid else O.ELU1)
    x = (rng.randn(B, d) * 2).astype(np.float32)
    x0 = (rng.randn(B, d) * 0.5).astype(np.float32)
    h = rng.randn(B, E * d).astype(np.float32)
    F, fx, fx0 = I.hip_forward(spec, t(x0, dev), t(x, dev), t(h, dev), n, inv_f=inv_f)
    kname = _lib.lib().umnn_last_kernel_name().decode()
    assert "32x32x16" in kname, kname
    assert U.rel_err(F.cpu().numpy(), O.integrate_parallel(net, x0, x, h, n, inv_f=inv_f)) < TOL
    assert U.rel_err(fx.cpu().numpy(), O.integrand(net, x, h)) < TOL
    assert U.rel_err(fx0.cpu().numpy(), O.integrand(net, x0, h)) < TOL
    # flow-block entry (z, log_jac in the kernel's epilogue): same values as the default kernel's to rounding
    if not inv_f and not sigmoid and not relu:
        sc = t(rng.randn(d).astype(np.float32) * 0.1, dev)
        z2, lj2 = I.hip_flow_block(spec, t(x, dev), t(h, dev), sc, n)[:2]
        opts(fwd_pipe=1)
        z1, lj1 = I.hip_flow_block(spec, t(x, dev), t(h, dev), sc, n)[:2]
        assert "32x32x16" not in _lib.lib().umnn_last_kernel_name().decode()
        assert float((z1 - z2).abs().max()) < 2e-5 * max(1.0, float(z1.abs().max()))
        assert float((lj1 - lj2).abs().max()) < 2e-5


def test_stack_block_entry_reverses_z_and_accumulates_log_jac(dev):
    """umnn_flow_stack_block_forward = umnn_flow_block_forward + the glue between the blocks of a flow
    (UMNNMAFFlow.py:109-123): same bits, z stored reversed, log_jac added to the running sum."""
    from umnn_amd import integral as I, IntegrandNetwork
    from umnn_amd.nets import mlp_spec
    torch.manual_seed(11)
    B, d, E = 70, 9, 5
    net = IntegrandNetwork(d, 1 + E, [50, 50, 50], 1).to(dev)
    spec = mlp_spec(net)
    x, h = torch.randn(B, d, device=dev), torch.randn(B, E * d, device=dev)
    scaling = torch.randn(d, device=dev) * 0.1
    run = torch.randn(B, d, device=dev)
    z, lj, fx, fx0 = I.hip_flow_block(spec, x, h, scaling, 30)
    z2, lj2, fx2, fx02 = I.hip_flow_block(spec, x, h, scaling, 30, reverse_z=True, log_jac_in=run)
    assert torch.equal(z2, torch.flip(z, [1])) and torch.equal(lj2, run + lj)
    assert torch.equal(fx, fx2) and torch.equal(fx0, fx02)


def test_graphed_compute_ll_replays_the_same_numbers(dev):
    """hipGraph capture of a whole compute_ll (conditioner GEMMs + quadrature launches through the C ABI): replays
    must reproduce the eager result bit for bit, also on new data copied into the captured buffers."""
    import umnn_amd
    torch.manual_seed(4)
    model = umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=2, hidden_derivative=[100] * 4, hidden_embedding=[100] * 4, embedding_s=10,
                                 nb_steps=50, device=dev).to(dev)
    model.eval()
    x1, x2 = torch.randn(512, 2, device=dev), torch.randn(512, 2, device=dev)
    with torch.no_grad():
        ref1 = [t.clone() for t in model.compute_ll(x1)]
        ref2 = [t.clone() for t in model.compute_ll(x2)]
    g = umnn_amd.GraphedLL(model, x1)
    out = g()
    assert torch.equal(out[0], ref1[0]) and torch.equal(out[1], ref1[1])
    out = g(x2)
    assert torch.equal(out[0], ref2[0]) and torch.equal(out[1], ref2[1])
    # weights move between replays (an optimizer step): the graph must not keep serving the stale conditioner weights it
    # baked in -- it re-captures when a version counter moved
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.01)
        ref3 = [t.clone() for t in model.compute_ll(x2)]
    assert not torch.equal(ref3[0], ref2[0])
    out = g(x2)
    assert g.captures == 2
    assert torch.equal(out[0], ref3[0]) and torch.equal(out[1], ref3[1])
    out = g(x1)
    assert g.captures == 2                                            # unchanged weights: plain replay
    # a user-level capture (no version tracking) must see live weights too: nothing cached is baked in
    with torch.no_grad():
        xs = x1.clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            model.compute_ll(xs)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            cap = model.compute_ll(xs)
        for p in model.parameters():
            p.mul_(0.99)
        ref4 = [t.clone() for t in model.compute_ll(xs)]
        graph.replay()
    assert torch.allclose(cap[0], ref4[0], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("hid,T", [([64, 64, 64], 5), ([70, 79, 64, 66], 5), ([90, 88, 95], 6), ([120, 112, 127], 8), ([100] * 4, 7)])
def test_exact_wide_variants_against_oracle(hid, T, dev):
    """Uniform tile counts above four (widths 64-127) run shape-exact bf16x3 variants (odd counts end in a half K-step
    on the K=16 MFMA); checked against the oracle, on a ragged batch."""
    import umnn_amd
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import MlpSpec
    if umnn_amd.get_forward_precision() not in ("bf16x3", "f16x3"):
        pytest.skip("exact wide variants exist for the two-piece modes (bf16x3, f16x3)")
    B, d, E, n = 21, 3, 6, 25
    rng = np.random.RandomState(T * 7 + len(hid))
    sizes = [1 + E] + hid + [1]
    Ws = [(rng.randn(sizes[i + 1], sizes[i]) * (1.6 / np.sqrt(sizes[i]))).astype(np.float32) for i in range(len(sizes) - 1)]
    bs = [(rng.randn(sizes[i + 1]) * 0.3).astype(np.float32) for i in range(len(sizes) - 1)]
    lin = []
    for W, b in zip(Ws, bs):
        m = torch.nn.Linear(W.shape[1], W.shape[0])
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(W))
            m.bias.copy_(torch.from_numpy(b))
        lin.append(m.to(dev))
    spec = MlpSpec(lin, _lib.ACT_LEAKY_RELU, _lib.OUT_ELU_PLUS_ONE)
    net = O.Net(Ws, bs, O.LEAKY, O.ELU1)
    x = (rng.randn(B, d) * 2).astype(np.float32)
    x0 = (rng.randn(B, d) * 0.5).astype(np.float32)
    h = rng.randn(B, E * d).astype(np.float32)
    F, fx, fx0 = I.hip_forward(spec, t(x0, dev), t(x, dev), t(h, dev), n)
    kname = _lib.lib().umnn_last_kernel_name().decode()
    assert f"T={T}," in kname and "EXACT=1" in kname, kname
    assert U.rel_err(F.cpu().numpy(), O.integrate_parallel(net, x0, x, h, n)) < TOL
    assert U.rel_err(fx.cpu().numpy(), O.integrand(net, x, h)) < TOL
    assert U.rel_err(fx0.cpu().numpy(), O.integrand(net, x0, h)) < TOL


@pytest.mark.parametrize("hid", [[50] * 4, [56, 60, 63], [50, 50], [40, 33, 48]])
def test_pipelined_forward_does_not_depend_on_what_ran_before(hid, dev):
    """The two-tile pipelined loop (matrix-pipe remainders, persistent fragment registers) at a size that selects it:
    identical bits whatever ran before."""
    import umnn_amd
    from umnn_amd import integral as I, IntegrandNetwork, _lib
    from umnn_amd.nets import mlp_spec
    if umnn_amd.get_forward_precision() not in ("bf16x3", "f16x3"):
        pytest.skip("the pipelined loop exists for the two-piece modes only (bf16x3, f16x3)")
    torch.manual_seed(8)
    net = IntegrandNetwork(8, 31, hid, 1).to(dev)
    spec = mlp_spec(net)
    x, h = torch.randn(5000, 8, device=dev), torch.randn(5000, 240, device=dev)
    ref = I.hip_forward(spec, None, x, h, 40)
    assert "PIPE" in _lib.lib().umnn_last_kernel_name().decode()
    for trial in range(4):
        junk = torch.randn(3000, 3000, device=dev) * (10.0 ** trial)
        (junk @ junk).sum().item()
        wide = IntegrandNetwork(3, 11, [100] * 4, 1).to(dev)
        I.hip_forward(mlp_spec(wide), None, torch.randn(3000, 3, device=dev) * 30, torch.randn(3000, 30, device=dev) * 30, 20)
        out = I.hip_forward(spec, None, x, h, 40)
        assert all(torch.equal(a, b) for a, b in zip(out, ref))


def test_mnist_width_d784_against_oracle(dev):
    """BASELINE config C4's dimension count: d = 784 (one embedding row is 784 floats apart from the next), MNIST
    integrand 31-100-50-50-50-50-1, a ragged handful of samples, forward and backward against the oracle."""
    from umnn_amd import integral as I, IntegrandNetwork
    from umnn_amd.nets import mlp_spec
    torch.manual_seed(12)
    B, d, E, n = 3, 784, 30, 50
    net = IntegrandNetwork(d, 1 + E, [100, 50, 50, 50, 50], 1)
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    onet = O.Net([m.weight.detach().numpy() for m in lin], [m.bias.detach().numpy() for m in lin], O.LEAKY, O.ELU1)
    net.to(dev)
    spec = mlp_spec(net)
    x, h, g = torch.randn(B, d), torch.randn(B, E * d), torch.randn(B, d)
    F, fx, _ = I.hip_forward(spec, None, x.to(dev), h.to(dev), n)
    import umnn_amd
    from umnn_amd import _lib
    if umnn_amd.get_forward_precision() == "bf16x3":      # the shape-exact wide-first-layer family, not the guarded T=8 one
        assert "T1=7,TREST=4" in _lib.lib().umnn_last_kernel_name().decode()
    x0 = np.zeros((B, d), np.float32)
    assert U.rel_err(F.cpu().numpy(), O.integrate_parallel(onet, x0, x.numpy(), h.numpy(), n)) < TOL
    assert U.rel_err(fx.cpu().numpy(), O.integrand(onet, x.numpy(), h.numpy())) < TOL
    xr, hr = x.to(dev).requires_grad_(True), h.to(dev).requires_grad_(True)
    Fa = I.ParallelNeuralIntegral.apply(torch.zeros_like(xr), xr, net, I._flatten(net.parameters()), hr, n)
    Fa.backward(g.to(dev))
    ref = O.integrate_backward(onet, x0, x.numpy(), h.numpy(), n, g.numpy())
    # 120 000 quadrature points x 300 hidden units: some pre-activation always sits inside the rounding noise of its own dot
    # product (kink margin 8e-10 here; the fp32 and fp64 ORACLES differ by 8.4e-5 on d_h for this very case), and d_h of one
    # (row, dimension) hangs on that unit's sign -- the kink-ambiguity rule of the random-shape sweep applies
    tol_g = TOL if U.kink_margin(onet, x0, x.numpy(), h.numpy(), n) > 5e-7 else 5e-4
    assert U.scaled_err(hr.grad.cpu().numpy(), ref[2]) < tol_g
    dth = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
    assert U.scaled_err(dth, ref[5]) < TOL


def test_c_abi_from_four_threads_at_once(dev):
    """The library is called from whatever thread PyTorch runs the op on (backward: autograd worker threads).  Four host
    threads launch forward + backward through the C ABI concurrently, each on its own stream (ctypes drops the GIL inside
    the call): every thread must get the bits the serial run produced -- no shared mutable state on the launch path
    (options are atomics read per launch, the LDS-cap / CU-count caches are locked or atomic, errors thread-local)."""
    import threading
    from umnn_amd import integral as I, IntegrandNetwork
    from umnn_amd.nets import mlp_spec
    torch.manual_seed(11)
    nets = [IntegrandNetwork(d, 1 + E, hid, 1).to(dev) for d, E, hid in
            [(6, 30, [50] * 4), (2, 10, [100] * 4), (5, 4, [40, 33]), (3, 8, [64, 64, 64])]]
    data = []
    for net in nets:
        d, E = net.nnets, net.nin - 1
        data.append((torch.randn(200, d, device=dev), torch.randn(200, E * d, device=dev), torch.randn(200, d, device=dev)))
    serial = []
    for net, (x, h, g) in zip(nets, data):
        spec = mlp_spec(net)
        serial.append((I.hip_forward(spec, None, x, h, 30), I.hip_backward(spec, None, x, h, g, None, 30)))
    torch.cuda.synchronize()
    results, errors = [None] * 4, []

    def work(i):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                spec = mlp_spec(nets[i])
                x, h, g = data[i]
                out = None
                for _ in range(25):
                    out = (I.hip_forward(spec, None, x, h, 30), I.hip_backward(spec, None, x, h, g, None, 30))
                st.synchronize()
            results[i] = out
        except Exception as e:           # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for got, ref in zip(results, serial):
        for a, b in zip(got[0] + tuple(got[1]), ref[0] + tuple(ref[1])):
            assert torch.equal(a, b)
