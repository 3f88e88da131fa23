"""Parity of the HIP backward path against the reference's custom-backward gradients (golden) and the oracle.

Gradient tensors span many magnitudes, so d_h and d_theta are compared as max|d| / max|ref| <= 1e-4 (the
reference's own noise between its two solvers is ~1e-6 on these vectors); d_x, d_x0 use the F criterion.
"""
import numpy as np
import pytest
import torch

from oracle import cc_oracle as O
from tests import _util as U
from tests.test_gpu_forward import build_integrand, t

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True, params=["bf16x3", "fp32"])
def bwd_precision(request):
    """Every gradient-parity test runs with both arithmetic modes of the backward kernels (same tolerance)."""
    import umnn_amd
    old = umnn_amd.get_backward_precision()
    umnn_amd.set_backward_precision(request.param)
    yield request.param
    umnn_amd.set_backward_precision(old)


@pytest.mark.parametrize("name", U.g2_names())
def test_backward_matches_golden(name, dev, bwd_precision):
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import mlp_spec
    G = U.load(name)
    net = build_integrand(G, dev)
    spec = mlp_spec(net)
    dx0, dx, dh, dth = I.hip_backward(spec, t(G["x0"], dev), t(G["x"], dev), t(G["h"], dev), t(G["g"], dev), None,
                                      int(G["n"]))
    torch.cuda.synchronize()
    assert "cc_bwd" in _lib.lib().umnn_last_kernel_name().decode()
    assert U.rel_err(dx0.cpu().numpy(), G["dx0_par"]) < TOL
    assert U.rel_err(dx.cpu().numpy(), G["dx_par"]) < TOL
    assert U.scaled_err(dh.cpu().numpy(), G["dh_par"]) < TOL
    assert U.scaled_err(dth.cpu().numpy(), G["dtheta_par"]) < TOL
    assert U.scaled_err(dth.cpu().numpy(), G["dtheta_seq"]) < TOL


@pytest.mark.parametrize("name", U.g7_names())
def test_inverse_integrand_backward_matches_golden(name, dev, bwd_precision):
    """ParallelNeuralIntegral.apply(..., inv_f=True).backward on the HIP kernels (one-pass bf16 / fp32, 100-wide, the three-stage
    family for the 100-50-50-50-50 net) against the reference's own output: d_theta / d_h of 1/f, Leibniz terms of f
    (ParallelNeuralIntegral.py:58-59,70-72,110-123)."""
    import umnn_amd
    from umnn_amd import integral as I, _lib
    G = U.load(name)
    net = build_integrand(G, dev)
    x0 = t(G["x0"], dev).requires_grad_()
    x = t(G["x"], dev).requires_grad_()
    h = t(G["h"], dev).requires_grad_()
    before = _lib.lib().umnn_launch_count()
    out = I.ParallelNeuralIntegral.apply(x0, x, net, I._flatten(net.parameters()), h, int(G["n"]), True)
    out.backward(t(G["g"], dev))
    torch.cuda.synchronize()
    from umnn_amd.nets import mlp_spec
    # (round 4: every fixture's net has HIP kernels in both modes -- fp32 mode runs the 100-50-50-50-50 net on the six-term build
    # of the three-stage kernels, cc_backward_front_p3.hip)
    assert I._hip_backward_ok(mlp_spec(net), x, h)
    assert umnn_amd.path_taken() == "hip" and umnn_amd.backward_path_taken() == "hip"
    assert _lib.lib().umnn_launch_count() >= before + 3
    kname = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
    assert "cc_bwd" in kname
    if "mnist" in name:
        assert "FRONT" in kname and kname.startswith("cc_bwd_bf16x6" if bwd_precision == "fp32" else "cc_bwd_bf16<"), kname
    dth = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert U.rel_err(out.detach().cpu().numpy(), G["F_inv"]) < TOL
    assert U.rel_err(x0.grad.cpu().numpy(), G["dx0"]) < TOL and U.rel_err(x.grad.cpu().numpy(), G["dx"]) < TOL
    assert U.scaled_err(h.grad.cpu().numpy(), G["dh"]) < TOL
    assert U.scaled_err(dth.cpu().numpy(), G["dtheta"]) < TOL
    # the (d_theta, d_h) form of integrate(), which the reference's backward consumes
    n = int(G["n"])
    with torch.no_grad():
        dth2, dh2 = I.integrate(x0.detach(), n, (x.detach() - x0.detach()) / n, net, h.detach(), True, t(G["g"], dev), True)
    assert U.scaled_err(dth2.cpu().numpy(), G["dtheta"]) < TOL and U.scaled_err(dh2.reshape(h.shape).cpu().numpy(), G["dh"]) < TOL


@pytest.mark.parametrize("name", ["g2_power_d6_w2", "g2_toy_d2_w2", "g2_sigmoid_d4", "g2_mnist_mixed_d8", "g2_odd_n_d3"])
def test_backward_with_jacobian_cotangent(name, dev):
    """g_fx (cotangent of the f(x;h) output): VJP at node 0 incl. d f/d x, against the oracle's manual backprop."""
    from umnn_amd import integral as I
    from umnn_amd.nets import mlp_spec
    G = U.load(name)
    net = build_integrand(G, dev)
    onet = U.net_from_g2(G)
    rng = np.random.RandomState(7)
    gfx = rng.randn(*G["x"].shape).astype(np.float32)
    n = int(G["n"])
    dx0, dx, dh, dth = I.hip_backward(mlp_spec(net), t(G["x0"], dev), t(G["x"], dev), t(G["h"], dev), t(G["g"], dev),
                                      t(gfx, dev), n)
    rdx0, rdx, rdh, _, _, rflat = O.integrate_backward(onet, G["x0"], G["x"], G["h"], n, G["g"])
    jdx, jdh, jflat = O.integrand_vjp(onet, G["x"], G["h"], gfx)
    assert U.rel_err(dx0.cpu().numpy(), rdx0) < TOL
    assert U.scaled_err(dx.cpu().numpy(), rdx + jdx) < TOL
    assert U.scaled_err(dh.cpu().numpy(), rdh + jdh) < TOL
    assert U.scaled_err(dth.cpu().numpy(), rflat + jflat) < TOL


def test_backward_ragged_and_deep(dev):
    """Tile-ragged batch (111 integrals), 7 hidden layers (two dW passes), ReLU."""
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import MlpSpec
    rng = np.random.RandomState(3)
    B, d, E, n = 37, 3, 2, 12
    sizes = [1 + E] + [24, 16, 40, 8, 33, 20, 12] + [1]
    Ws = [(rng.randn(sizes[i + 1], sizes[i]) * (1.5 / np.sqrt(sizes[i]))).astype(np.float32) for i in range(len(sizes) - 1)]
    bs = [(rng.randn(sizes[i + 1]) * 0.3).astype(np.float32) for i in range(len(sizes) - 1)]
    lin = []
    for W, b in zip(Ws, bs):
        m = torch.nn.Linear(W.shape[1], W.shape[0])
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(W))
            m.bias.copy_(torch.from_numpy(b))
        lin.append(m.to(dev))
    spec = MlpSpec(lin, _lib.ACT_RELU, _lib.OUT_ELU_PLUS_ONE)
    x = (rng.randn(B, d) * 2).astype(np.float32)
    x0 = (rng.randn(B, d) * 0.5).astype(np.float32)
    h = rng.randn(B, E * d).astype(np.float32)
    g = rng.randn(B, d).astype(np.float32)
    onet = O.Net(Ws, bs, O.RELU, O.ELU1)
    dx0, dx, dh, dth = I.hip_backward(spec, t(x0, dev), t(x, dev), t(h, dev), t(g, dev), None, n)
    rdx0, rdx, rdh, _, _, rflat = O.integrate_backward(onet, x0, x, h, n, g)
    assert U.rel_err(dx0.cpu().numpy(), rdx0) < TOL and U.rel_err(dx.cpu().numpy(), rdx) < TOL
    assert U.scaled_err(dh.cpu().numpy(), rdh) < TOL
    assert U.scaled_err(dth.cpu().numpy(), rflat) < TOL


@pytest.mark.parametrize("hid,E,B,d", [([100, 100], 90, 41, 3), ([50, 50, 50], 30, 700, 5), ([20, 20], 4, 9, 2)])
def test_first_layer_gradient_gemm_shapes(hid, E, B, d, dev):
    """d W1[:,1:] / d b1 leave as a skinny GEMM over all integrals (cc_bwd_dw0_kernel: persistent blocks, fp32 MFMA): an
    embedding wider than one launch's 40 output tiles, several chunks per block, and a batch smaller than one chunk."""
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import MlpSpec
    rng = np.random.RandomState(11)
    n = 16
    sizes = [1 + E] + hid + [1]
    Ws = [(rng.randn(sizes[i + 1], sizes[i]) * (1.2 / np.sqrt(sizes[i]))).astype(np.float32) for i in range(len(sizes) - 1)]
    bs = [(rng.randn(sizes[i + 1]) * 0.3).astype(np.float32) for i in range(len(sizes) - 1)]
    lin = []
    for W, b in zip(Ws, bs):
        m = torch.nn.Linear(W.shape[1], W.shape[0])
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(W))
            m.bias.copy_(torch.from_numpy(b))
        lin.append(m.to(dev))
    spec = MlpSpec(lin, _lib.ACT_LEAKY_RELU, _lib.OUT_ELU_PLUS_ONE)
    x = (rng.randn(B, d) * 2).astype(np.float32)
    h = rng.randn(B, E * d).astype(np.float32)
    g = rng.randn(B, d).astype(np.float32)
    onet = O.Net(Ws, bs, O.LEAKY, O.ELU1)
    dx0, dx, dh, dth = I.hip_backward(spec, None, t(x, dev), t(h, dev), t(g, dev), None, n)
    _, rdx, rdh, _, _, rflat = O.integrate_backward(onet, np.zeros_like(x), x, h, n, g)
    assert U.scaled_err(dh.cpu().numpy(), rdh) < TOL
    got, ref = dth.cpu().numpy(), rflat
    nW0 = sizes[1] * sizes[0]
    assert U.scaled_err(got[:nW0 + sizes[1]], ref[:nW0 + sizes[1]]) < TOL      # W1 (x column and embedding columns) and b1
    assert U.scaled_err(got, ref) < TOL


def test_autograd_function_matches_reference_api(dev):
    """ParallelNeuralIntegral.apply / NeuralIntegral.apply through autograd (reference tests/test_jit.py:12-86 shape)."""
    import umnn_amd
    G = U.load("g2_jit_d5")
    net = build_integrand(G, dev)
    for Fn, extra, tag in ((umnn_amd.ParallelNeuralIntegral, (False,), "par"), (umnn_amd.NeuralIntegral, (), "seq")):
        net.zero_grad()
        x0 = t(G["x0"], dev).requires_grad_()
        x = t(G["x"], dev).requires_grad_()
        h = t(G["h"], dev).requires_grad_()
        flat = torch.cat([p.contiguous().view(-1) for p in net.parameters()])
        out = Fn.apply(x0, x, net, flat, h, int(G["n"]), *extra)
        assert umnn_amd.path_taken() == "hip"
        out.backward(t(G["g"], dev))
        assert U.rel_err(out.detach().cpu().numpy(), G[f"Fapply_{tag}"]) < TOL
        assert U.rel_err(x.grad.cpu().numpy(), G[f"dx_{tag}"]) < TOL
        assert U.rel_err(x0.grad.cpu().numpy(), G[f"dx0_{tag}"]) < TOL
        assert U.scaled_err(h.grad.cpu().numpy(), G[f"dh_{tag}"]) < TOL
        got = torch.cat([p.grad.view(-1) for p in net.parameters()]).cpu().numpy()
        assert U.scaled_err(got, G[f"dtheta_{tag}"]) < TOL


@pytest.mark.parametrize("name", U.g4_names())
def test_flow_training_gradients_match_reference(name, dev):
    """-mean(ll).backward() through the module API on the GPU vs the reference's gradients for every parameter."""
    import umnn_amd
    G = U.load(name)
    m = umnn_amd.UMNNMAFFlow(nb_flow=int(G["nb_flow"]), nb_in=int(G["d"]),
                             hidden_derivative=[int(v) for v in G["hidden_derivative"]],
                             hidden_embedding=[int(v) for v in G["hidden_embedding"]], embedding_s=int(G["E"]),
                             nb_steps=int(G["n"]), solver=str(G["solver"]), cond_in=int(G["cond_in"]))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in U.state_dict_of(G).items()})
    m.to(dev).train()
    x = t(G["x"], dev).requires_grad_()
    ctx = t(G["context"], dev) if "context" in G else None
    ll, z = m.compute_ll(x, context=ctx)
    assert umnn_amd.path_taken() == "hip"
    (-ll.mean()).backward()
    assert U.rel_err(ll.detach().cpu().numpy(), G["ll_train"]) < TOL
    assert U.scaled_err(x.grad.cpu().numpy(), G["grad/x"]) < TOL
    worst = 0.0
    for k, p in m.named_parameters():
        if p.grad is not None and ("grad/" + k) in G:
            worst = max(worst, U.scaled_err(p.grad.cpu().numpy(), G["grad/" + k]))
    assert worst < TOL, worst


@pytest.mark.parametrize("n", [50, 100])
def test_monotonic_nn_matches_reference(n, dev):
    import umnn_amd
    G = U.load(f"g5_monotonic_n{n}")
    m = umnn_amd.MonotonicNN(3, [100, 100, 100], nb_steps=n, dev=dev)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in U.state_dict_of(G).items()})
    m.to(dev)
    x = t(G["x"], dev).requires_grad_()
    y = m(x, t(G["h"], dev))
    assert umnn_amd.path_taken() == "hip"
    (y ** 2).mean().backward()
    assert U.rel_err(y.detach().cpu().numpy(), G["y"]) < TOL
    assert U.scaled_err(x.grad.cpu().numpy(), G["grad/x"]) < TOL
    for k, p in m.named_parameters():
        assert U.scaled_err(p.grad.cpu().numpy(), G["grad/" + k]) < TOL, k


def test_backward_deterministic(dev):
    from umnn_amd import integral as I
    from umnn_amd.nets import mlp_spec
    G = U.load("g2_power_d6_w2")
    net = build_integrand(G, dev)
    args = (mlp_spec(net), t(G["x0"], dev), t(G["x"], dev), t(G["h"], dev), t(G["g"], dev), None, int(G["n"]))
    a = I.hip_backward(*args)
    b = I.hip_backward(*args)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_graphed_train_step_matches_eager_training(dev):
    """A whole optimisation step (flow forward, HIP backward through the autograd wrapper, Adam) captured as one
    hipGraph: after the same number of steps on the same data the parameters must equal the eager run's bit for bit
    (the kernels are deterministic), and the loss must go down."""
    import copy
    import umnn_amd
    torch.manual_seed(21)
    model_a = umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=3, hidden_derivative=[50, 50, 50], hidden_embedding=[64, 64], embedding_s=8,
                                   nb_steps=20, device=dev).to(dev)
    model_b = copy.deepcopy(model_a)
    xs = [torch.randn(100, 3, device=dev) for _ in range(6)]
    opt_a = torch.optim.Adam([p for p in model_a.parameters() if p.requires_grad], lr=1e-3, capturable=True)
    opt_b = torch.optim.Adam([p for p in model_b.parameters() if p.requires_grad], lr=1e-3, capturable=True)
    model_a.train(); model_b.train()
    warm = 2
    losses_a = []
    for i, x in enumerate([xs[0]] * warm + xs):        # eager: the warm-up steps of the capture run on xs[0]
        opt_a.zero_grad(set_to_none=True)
        ll, _ = model_a.compute_ll(x)
        loss = -ll.mean()
        loss.backward()
        opt_a.step()
        losses_a.append(loss.item())
    step = umnn_amd.GraphedTrainStep(model_b, opt_b, xs[0], warmup=warm)      # warm-up + capture (capture does not run)
    losses_b = [step(x).item() for x in xs]
    assert max(abs(a - b) for a, b in zip(losses_a[warm:], losses_b)) < 1e-5, (losses_a, losses_b)
    for (n, pa), (_, pb) in zip(model_a.named_parameters(), model_b.named_parameters()):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-6), n


@pytest.mark.parametrize("hid", [[100, 50, 50], [72, 72], [120, 40, 40, 40], [100, 50, 50, 50, 50]])
def test_mixed_wide_nets_backward_routes_match_the_oracle(hid, dev, monkeypatch, bwd_precision):
    """Nets with hidden layers wider than four tiles and no one-pass shape-exact backward.  With the default (bf16x3)
    arithmetic, nets whose FIRST hidden layer is wide and the rest narrow (MNISTExperiment's 100-50-50-50-50) run the
    three-stage HIP backward of cc_backward_front.hip; other nets up to 103 wide (and those in the fp32 mode) run zero-padded on
    the 5- / 7-tile shape-exact fp32 kernels; the rest is differentiated with the materialised ATen chain on the GPU unless
    UMNN_BWD_WIDE=hip forces the generic HIP kernels.  Every route must match the oracle's restatement of the reference backward."""
    import ctypes
    from umnn_amd import integral as I, IntegrandNetwork, _lib
    from umnn_amd.nets import mlp_spec
    torch.manual_seed(len(hid))
    B, d, E, n = 9, 3, 4, 12
    net = IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
    spec = mlp_spec(net)
    lin = spec.linears
    onet = O.Net([m.weight.detach().cpu().numpy() for m in lin], [m.bias.detach().cpu().numpy() for m in lin], O.LEAKY, O.ELU1)
    x0 = torch.randn(B, d, device=dev) * 0.3
    x = torch.randn(B, d, device=dev) * 2
    h = torch.randn(B, E * d, device=dev)
    g = torch.randn(B, d, device=dev)
    ref = O.integrate_backward(onet, x0.cpu().numpy(), x.cpu().numpy(), h.cpu().numpy(), n, g.cpu().numpy())
    ref_dh, ref_dtheta = ref[2], ref[5]
    # (three-stage kernels: both precisions since round 4 -- fp32 mode runs their six-term build, cc_backward_front_p3.hip)
    staged = hid[0] > 63 and max(hid[1:]) <= 63
    # zero-padded onto the 5- / 7- / 8-tile shape-exact fp32 family (pad_to_exact_family) -- where the padded weight images fit
    # the LDS: the four- and five-hidden-layer nets of this list do not (3 x 68 KB, 4 x 45 KB) and keep the ATen chain
    padded = not staged and len(hid) <= 3
    desc, keep = I._desc(spec)
    assert (_lib.lib().umnn_cc_backward_kind(ctypes.byref(desc), E) >= 0) == (staged or padded)
    for mode in ("", "hip"):
        monkeypatch.setitem(I._BWD_WIDE, "hip", mode == "hip")
        xr, hr = x.clone().requires_grad_(True), h.clone().requires_grad_(True)
        for p in net.parameters():
            p.grad = None
        F = I.ParallelNeuralIntegral.apply(x0, xr, net, I._flatten(net.parameters()), hr, n)
        launches = _lib.lib().umnn_launch_count()
        F.backward(g)
        torch.cuda.synchronize()
        # (autograd runs backward on its own thread: count library launches instead of asking path_taken())
        assert (_lib.lib().umnn_launch_count() > launches) == (mode == "hip" or staged or padded)
        if padded:
            name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
            assert "KS=17" in name or "KS=26" in name or "KS=32" in name, name
        if staged:
            name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
            assert "FRONT" in name and name.startswith("cc_bwd_bf16x6" if bwd_precision == "fp32" else "cc_bwd_bf16<"), name
        dtheta = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
        assert np.abs(dtheta - ref_dtheta).max() <= 1e-4 * np.abs(ref_dtheta).max()
        assert np.abs(hr.grad.cpu().numpy() - ref_dh).max() <= 1e-4 * np.abs(ref_dh).max()
        assert U.rel_err(xr.grad.cpu().numpy(), O.integrand(onet, x.cpu().numpy(), h.cpu().numpy()) * g.cpu().numpy()) < 1e-4
    # the (F, f_x) operator of the flow blocks (cotangents for both outputs: exercises the tangent pass through the wide
    # first layer): the HIP route against the materialised ATen chain
    outs = []
    for route in ("hip", "aten"):
        monkeypatch.setitem(I._BWD_WIDE, "hip", route == "hip")
        xr, hr = x.clone().requires_grad_(True), h.clone().requires_grad_(True)
        for p in net.parameters():
            p.grad = None
        if route == "aten":
            monkeypatch.setattr(I, "_hip_backward_ok", lambda *a, **k: False)
        F, fx = I.IntegralWithJacobian.apply(x0, xr, net, I._flatten(net.parameters()), hr, n)
        (F * g).sum().add((torch.log(fx) * g.flip(0)).sum()).backward()
        outs.append((xr.grad.clone(), hr.grad.clone(), torch.cat([p.grad.reshape(-1) for p in net.parameters()])))
    for a, b in zip(*outs):
        assert (a - b).abs().max() <= 1e-4 * b.abs().max()


def test_staged_backward_chunks_and_large_batch(dev, bwd_precision):
    """The three-stage backward at the MNIST shape (d=784, 31-100-50-50-50-50-1, n=50): rows sampled against the oracle
    (d_h, d_x depend on their own row), d_theta against the sum of two half batches (linearity), chunked scratch (the
    batch of 256 rows needs several 1-GiB chunks) and bit-repeatability."""
    from umnn_amd import integral as I, IntegrandNetwork, _lib
    from umnn_amd.nets import mlp_spec
    if bwd_precision != "bf16x3":
        pytest.skip("the staged kernels are the bf16x3 route")
    torch.manual_seed(0)
    B, d, E, n = 256, 784, 30, 50
    net = IntegrandNetwork(d, 1 + E, [100, 50, 50, 50, 50], 1).to(dev)
    with torch.no_grad():
        for mod in net.net:
            if isinstance(mod, torch.nn.Linear):
                mod.weight.mul_(1.5)
    spec = mlp_spec(net)
    lin = spec.linears
    onet = O.Net([m.weight.detach().cpu().numpy() for m in lin], [m.bias.detach().cpu().numpy() for m in lin], O.LEAKY, O.ELU1)
    x = torch.randn(B, d, device=dev)
    h = torch.randn(B, E * d, device=dev)
    g, gf = torch.randn(B, d, device=dev), torch.randn(B, d, device=dev) * 0.1
    before = _lib.lib().umnn_launch_count()
    dx0, dx, dh, dth = I.hip_backward(spec, None, x, h, g, gf, n)
    torch.cuda.synchronize()
    assert "FRONT" in _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
    assert _lib.lib().umnn_launch_count() > before
    rows = np.array([0, 101, 255])
    xs, hs, gs = x[rows].cpu().numpy(), h[rows].cpu().numpy(), g[rows].cpu().numpy()
    r = O.integrate_backward(onet, np.zeros_like(xs), xs, hs, n, gs)
    dh_nogfx = I.hip_backward(spec, None, x, h, g, None, n)
    assert U.scaled_err(dh_nogfx[2][rows].cpu().numpy(), r[2]) < TOL
    assert U.rel_err(dh_nogfx[1][rows].cpu().numpy(), r[1]) < TOL
    a = I.hip_backward(spec, None, x[:128].contiguous(), h[:128].contiguous(), g[:128].contiguous(), gf[:128].contiguous(), n)
    b = I.hip_backward(spec, None, x[128:].contiguous(), h[128:].contiguous(), g[128:].contiguous(), gf[128:].contiguous(), n)
    assert float((a[3] + b[3] - dth).abs().max()) <= 3e-5 * float(dth.abs().max())
    assert torch.equal(torch.cat([a[2], b[2]]), dh) and torch.equal(torch.cat([a[1], b[1]]), dx)
    again = I.hip_backward(spec, None, x, h, g, gf, n)
    assert all(torch.equal(p, q) for p, q in zip(again[1:], (dx, dh, dth)))


def test_full_size_backward_properties_bsds300_shard(dev, bwd_precision):
    """BASELINE config C3 at one GPU's share (8192 x 63, n=100, 31-50^4-1), backward: (a) dh and dx of a random sample
    of rows against the oracle (they depend on their own row only), (b) shard consistency: dh/dx of two half batches
    concatenate bit-for-bit into the full batch's (the halves fall on tile boundaries) and their d_theta add up to the
    full one, (c) linearity in the cotangent, (d) determinism."""
    from umnn_amd import integral as I, IntegrandNetwork
    from umnn_amd.nets import mlp_spec
    torch.manual_seed(0)
    B, d, E, n = 8192, 63, 30, 100
    net = IntegrandNetwork(d, 1 + E, [50] * 4, 1)
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    onet = O.Net([m.weight.detach().numpy() for m in lin], [m.bias.detach().numpy() for m in lin], O.LEAKY, O.ELU1)
    net.to(dev)
    spec = mlp_spec(net)
    x, h, g = torch.randn(B, d), torch.randn(B, E * d), torch.randn(B, d)
    xg, hg, gg = x.to(dev), h.to(dev), g.to(dev)
    dx0, dx, dh, dth = I.hip_backward(spec, None, xg, hg, gg, None, n)
    # (which kernel this test exercises: the workgroup pipeline on fp16 pieces under the default arithmetic, the exact kernels under fp32)
    from umnn_amd import _lib
    kname = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
    assert kname == ("cc_bwd_f16<L=4,LIVE=13,WS>" if bwd_precision == "bf16x3" else "cc_bwd<T=4,NACC=3,EDGE=1,KS=13>"), kname
    rows = np.random.RandomState(2).choice(B, 24, replace=False)
    ref = O.integrate_backward(onet, np.zeros((24, d), np.float32), x.numpy()[rows], h.numpy()[rows], n, g.numpy()[rows])
    assert U.rel_err(dx.cpu().numpy()[rows], ref[1]) < TOL
    assert U.scaled_err(dh.cpu().numpy()[rows], ref[2]) < TOL
    half = B // 2
    a = I.hip_backward(spec, None, xg[:half].contiguous(), hg[:half].contiguous(), gg[:half].contiguous(), None, n)
    b = I.hip_backward(spec, None, xg[half:].contiguous(), hg[half:].contiguous(), gg[half:].contiguous(), None, n)
    assert torch.equal(torch.cat([a[1], b[1]]), dx) and torch.equal(torch.cat([a[2], b[2]]), dh)
    assert float((a[3] + b[3] - dth).abs().max()) <= 2e-5 * float(dth.abs().max())
    g2 = torch.randn(B, d, device=dev)
    s1 = I.hip_backward(spec, None, xg, hg, g2, None, n)
    s12 = I.hip_backward(spec, None, xg, hg, gg + g2, None, n)
    assert float((s12[3] - dth - s1[3]).abs().max()) <= 2e-5 * float(s12[3].abs().max())
    assert float((s12[2] - dh - s1[2]).abs().max()) <= 2e-5 * float(s12[2].abs().max())
    again = I.hip_backward(spec, None, xg, hg, gg, None, n)
    assert all(torch.equal(p, q) for p, q in zip(again[1:], (dx, dh, dth)))


@pytest.mark.parametrize("hid", [[64, 64], [64, 64, 64, 64], [100, 100, 100]])
def test_exact_wide_backward_families_match_the_oracle(hid, dev):
    """64-wide (5 tiles) and 100-wide (7 tiles) nets have shape-exact fp32 backward kernels: HIP route, oracle parity."""
    import ctypes
    from umnn_amd import integral as I, IntegrandNetwork, _lib
    from umnn_amd.nets import mlp_spec
    torch.manual_seed(len(hid) + hid[0])
    B, d, E, n = 11, 2, 5, 15
    net = IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
    spec = mlp_spec(net)
    desc, keep = I._desc(spec)
    assert _lib.lib().umnn_cc_backward_kind(ctypes.byref(desc), E) == 1
    lin = spec.linears
    onet = O.Net([m.weight.detach().cpu().numpy() for m in lin], [m.bias.detach().cpu().numpy() for m in lin], O.LEAKY, O.ELU1)
    x0, x = torch.randn(B, d, device=dev) * 0.3, torch.randn(B, d, device=dev) * 2
    h, g = torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev)
    ref = O.integrate_backward(onet, x0.cpu().numpy(), x.cpu().numpy(), h.cpu().numpy(), n, g.cpu().numpy())
    dx0, dx, dh, dth = I.hip_backward(spec, x0, x, h, g, None, n)
    kname = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
    ks = {64: "T=5", 100: "T=7"}[hid[0]]
    assert kname.startswith("cc_bwd<") and ks in kname and "KS=0" not in kname, kname      # the shape-exact family ran
    assert U.rel_err(dx0.cpu().numpy(), ref[0]) < TOL and U.rel_err(dx.cpu().numpy(), ref[1]) < TOL
    assert U.scaled_err(dh.cpu().numpy(), ref[2]) < TOL
    assert U.scaled_err(dth.cpu().numpy(), ref[5]) < TOL


def _dirty_the_gpu(dev, trial):
    """Run unrelated work so that stale registers / LDS contents differ between two otherwise identical calls."""
    from umnn_amd import integral as I, IntegrandNetwork
    from umnn_amd.nets import mlp_spec
    junk = torch.randn(2048, 2048, device=dev) * (10.0 ** trial)
    (junk @ junk).sum().item()
    if trial % 2:
        big = IntegrandNetwork(7, 31, [50] * 4, 1).to(dev)
        I.hip_forward(mlp_spec(big), None, torch.randn(2000, 7, device=dev) * 50, torch.randn(2000, 210, device=dev) * 50, 30)


@pytest.mark.parametrize("hid,B,d,E,n", [([50, 50, 50], 100, 3, 8, 20), ([50] * 4, 300, 6, 30, 100), ([40, 40], 64, 5, 4, 30),
                                         ([100] * 3, 40, 2, 2, 25), ([64, 64], 50, 3, 4, 20), ([40, 36, 44], 90, 4, 6, 20),
                                         ([56, 60, 52, 63], 70, 5, 8, 30), ([20, 20], 33, 3, 4, 15)])
def test_results_do_not_depend_on_what_ran_before(hid, B, d, E, n, dev):
    """Every kernel must initialise what it reads: the same forward / backward call, repeated after unrelated kernels
    have left other data in registers and LDS, returns the same bits (this caught an experimental split variant whose
    three-hidden-layer instantiation read stale state)."""
    from umnn_amd import integral as I, IntegrandNetwork
    from umnn_amd.nets import mlp_spec
    torch.manual_seed(3)
    net = IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
    spec = mlp_spec(net)
    x, h, g, gf = (torch.randn(B, d, device=dev), torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev),
                   torch.randn(B, d, device=dev))
    ref_b = I.hip_backward(spec, None, x, h, g, gf, n)
    ref_f = I.hip_forward(spec, None, x, h, n)
    for trial in range(4):
        _dirty_the_gpu(dev, trial)
        out_b = I.hip_backward(spec, None, x, h, g, gf, n)
        _dirty_the_gpu(dev, trial + 1)
        out_f = I.hip_forward(spec, None, x, h, n)
        assert all(torch.equal(a, b) for a, b in zip(out_b, ref_b))
        assert all(torch.equal(a, b) for a, b in zip(out_f, ref_f))


def test_integration_md_ctypes_stub_runs_as_written(dev):
    """INTEGRATION.md section B shows the ctypes binding a maintainer of the reference would paste into
    models/UMNN/ParallelNeuralIntegral.py.  Execute that very code block (only the library path substituted) and compare
    its forward and backward with this package's operator: the documented boundary is the real one."""
    import os
    import re
    import umnn_amd
    from umnn_amd import _lib, IntegrandNetwork
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes, torch.*?)```", text, re.S).group(1)
    code = code.replace('"/path/to/umnn_amd/libumnn_cc.so"', repr(_lib.LIB_PATH))
    ns = {"compute_cc_weights": umnn_amd.compute_cc_weights}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    ns["_lib"].umnn_last_error.restype = __import__("ctypes").c_char_p
    ns["_lib"].umnn_cc_backward_workspace_bytes.restype = __import__("ctypes").c_longlong
    Stub = ns["ParallelNeuralIntegral"]
    torch.manual_seed(2)
    B, d, E, n = 40, 6, 30, 50
    net = IntegrandNetwork(d, 1 + E, [50] * 4, 1).to(dev)
    x0 = torch.zeros(B, d, device=dev)
    x = torch.randn(B, d, device=dev)
    h = torch.randn(B, E * d, device=dev)
    g = torch.randn(B, d, device=dev)
    for inv_f in (False, True):
        outs = []
        for Op in (Stub, umnn_amd.ParallelNeuralIntegral):
            xr, hr = x.clone().requires_grad_(True), h.clone().requires_grad_(True)
            flat = torch.cat([p.reshape(-1) for p in net.parameters()]).detach().requires_grad_(True)
            for p in net.parameters():
                p.grad = None
            F = Op.apply(x0, xr, net, flat, hr, n, inv_f)
            F.backward(g)
            torch.cuda.synchronize()
            # the stub returns d_theta for the flat_params argument (the reference's convention); the package fills both
            dth = flat.grad if flat.grad is not None else torch.cat([p.grad.reshape(-1) for p in net.parameters()])
            outs.append((F.detach(), xr.grad, hr.grad, dth))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
