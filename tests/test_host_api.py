"""Host-side module API on CPU (generic ATen quadrature): reference parity without a GPU.

Covers what the reference's own tests cover (tests/test_numerical_validation.py, tests/test_jit.py) plus the
drop-in surface: state_dict keys, method names (including the aliases the reference's scripts call), mode
dispatch, arbitrary-callable integrands, inversion, runtime-variable step counts.
"""
import math
import sys
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import umnn_amd
from tests import _util as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.filterwarnings("ignore:umnn_amd. MLP integrand on host tensors")


def flow_from_golden(G):
    m = umnn_amd.UMNNMAFFlow(nb_flow=int(G["nb_flow"]), nb_in=int(G["d"]),
                             hidden_derivative=[int(v) for v in G["hidden_derivative"]],
                             hidden_embedding=[int(v) for v in G["hidden_embedding"]], embedding_s=int(G["E"]),
                             nb_steps=int(G["n"]), solver=str(G["solver"]), cond_in=int(G["cond_in"]))
    missing = m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in U.state_dict_of(G).items()})
    assert not missing.missing_keys and not missing.unexpected_keys      # exact state_dict key parity
    return m


@pytest.mark.parametrize("name", U.g4_names())
def test_flow_api_matches_reference(name):
    G = U.load(name)
    m = flow_from_golden(G)
    x = torch.from_numpy(G["x"])
    ctx = torch.from_numpy(G["context"]) if "context" in G else None
    for mode in ("eval", "train"):
        m.train(mode == "train")
        with torch.no_grad():
            ll, z = m.compute_ll(x, context=ctx)
            lj = m.compute_log_jac(x, context=ctx)
            zb, ljb = m.compute_log_jac_bis(x, context=ctx)
            llb, _ = m.compute_ll_bis(x, context=ctx)
            bpp, _, _ = m.compute_bpp(x, context=ctx)
            fwd = m(x, context=ctx)
        assert umnn_amd.path_taken() == "aten"
        assert U.rel_err(ll.numpy(), G[f"ll_{mode}"]) < 1e-5
        assert U.rel_err(z.numpy(), G[f"z_{mode}"]) < 1e-5
        assert U.rel_err(lj.numpy(), G[f"log_jac_{mode}"]) < 1e-5
        assert U.rel_err(zb.numpy(), G[f"z_bis_{mode}"]) < 1e-5
        assert U.rel_err(ljb.numpy(), G[f"log_jac_bis_{mode}"]) < 1e-5
        assert U.rel_err(llb.numpy(), G[f"ll_bis_{mode}"]) < 1e-5
        assert U.rel_err(bpp.numpy(), G[f"bpp_{mode}"]) < 1e-5
        assert U.rel_err(fwd.numpy(), G[f"fwd_{mode}"]) < 1e-5
    m.train()
    m.zero_grad()
    xr = x.clone().requires_grad_()
    ll, _ = m.compute_ll(xr, context=ctx)
    (-ll.mean()).backward()
    assert U.scaled_err(xr.grad.numpy(), G["grad/x"]) < 2e-5
    for k, p in m.named_parameters():
        if p.grad is not None:
            assert U.scaled_err(p.grad.numpy(), G["grad/" + k]) < 2e-5, k


@pytest.mark.parametrize("name", ["g2_jit_d5", "g2_fd_d3", "g2_sigmoid_d4", "g2_odd_n_d3"])
def test_autograd_functions_match_reference(name):
    G = U.load(name)
    hid = [int(v) for v in G["hidden"]]
    net = umnn_amd.IntegrandNetwork(int(G["d"]), 1 + int(G["E"]), hid, 1, act_func=str(G["act"]))
    with torch.no_grad():
        for l, lin in enumerate(m for m in net.net if isinstance(m, nn.Linear)):
            lin.weight.copy_(torch.from_numpy(G[f"W{l}"]))
            lin.bias.copy_(torch.from_numpy(G[f"b{l}"]))
    for Fn, extra, tag in ((umnn_amd.ParallelNeuralIntegral, (False,), "par"), (umnn_amd.NeuralIntegral, (), "seq")):
        net.zero_grad()
        x0, x, h = (torch.from_numpy(G[k]).clone().requires_grad_() for k in ("x0", "x", "h"))
        flat = torch.cat([p.contiguous().view(-1) for p in net.parameters()])
        out = Fn.apply(x0, x, net, flat, h, int(G["n"]), *extra)
        out.backward(torch.from_numpy(G["g"]))
        assert U.rel_err(out.detach().numpy(), G[f"Fapply_{tag}"]) < 1e-5
        assert U.rel_err(x.grad.numpy(), G[f"dx_{tag}"]) < 1e-5
        assert U.rel_err(x0.grad.numpy(), G[f"dx0_{tag}"]) < 1e-5
        assert U.scaled_err(h.grad.numpy(), G[f"dh_{tag}"]) < 2e-5
        got = torch.cat([p.grad.view(-1) for p in net.parameters()]).numpy()
        assert U.scaled_err(got, G[f"dtheta_{tag}"]) < 2e-5
    # integrate() keeps the reference's signature, incl. inv_f and the (d_theta, d_h) form
    with torch.no_grad():
        x0, x, h = (torch.from_numpy(G[k]) for k in ("x0", "x", "h"))
        n = int(G["n"])
        assert U.rel_err(umnn_amd.integrate(x0, n, (x - x0) / n, net, h, False).numpy(), G["F_par"]) < 1e-5
        assert U.rel_err(umnn_amd.integrate(x0, n, (x - x0) / n, net, h, False, None, True).numpy(), G["F_inv"]) < 1e-5
    dth, dh = umnn_amd.integrate(x0, n, (x - x0) / n, net, h, True, torch.from_numpy(G["g"]))
    assert U.scaled_err(dth.numpy(), G["dtheta_par"]) < 2e-5 and U.scaled_err(dh.numpy(), G["dh_par"]) < 2e-5


@pytest.mark.parametrize("n", [50, 100])
def test_monotonic_nn(n):
    G = U.load(f"g5_monotonic_n{n}")
    m = umnn_amd.MonotonicNN(3, [100, 100, 100], nb_steps=n, dev="cpu")
    r = m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in U.state_dict_of(G).items()})
    assert not r.missing_keys and not r.unexpected_keys
    x = torch.from_numpy(G["x"]).requires_grad_()
    y = m(x, torch.from_numpy(G["h"]))
    (y ** 2).mean().backward()
    assert U.rel_err(y.detach().numpy(), G["y"]) < 1e-5
    assert U.scaled_err(x.grad.numpy(), G["grad/x"]) < 2e-5
    for k, p in m.named_parameters():
        assert U.scaled_err(p.grad.numpy(), G["grad/" + k]) < 2e-5, k


def test_arbitrary_callable_integrands_known_answers():
    """reference tests/test_numerical_validation.py:18-97 and :319-402 with the same integrands and tolerances."""
    class OnePlusX2(nn.Module):
        def forward(self, x, h):
            return 1.0 + x ** 2
    x0, x1, h = torch.zeros(5, 1), torch.ones(5, 1) * 2.0, torch.zeros(5, 1)
    errs = []
    for n in (5, 10, 20, 50, 100, 200):
        with torch.no_grad():
            r = umnn_amd.ParallelNeuralIntegral.apply(x0, x1, OnePlusX2(), torch.tensor([]), h, n, False)
        errs.append(abs(r.mean().item() - 14. / 3.))
    assert errs[-1] < 1e-4
    cases = [(lambda x, h: torch.ones_like(x) * 2.0, 0., 3., 6.0), (lambda x, h: x, 0., 2., 2.0),
             (lambda x, h: x ** 2, 1., 3., 26. / 3.), (lambda x, h: torch.exp(x), 0., 1., math.e - 1.)]
    for fn, a, b, true in cases:
        class Wrap(nn.Module):
            def forward(self, x, h, fn=fn):
                return fn(x, h)
        for n in (20, 50, 100):
            with torch.no_grad():
                r = umnn_amd.ParallelNeuralIntegral.apply(torch.ones(1, 1) * a, torch.ones(1, 1) * b, Wrap(),
                                                          torch.tensor([]), torch.zeros(1, 1), n, False)
            assert abs(r.item() - true) < 1e-3
        with torch.no_grad():   # a bare lambda (what UMNNMAF.invert passes, UMNNMAF.py:207) with flat_params=None
            r = umnn_amd.ParallelNeuralIntegral.apply(torch.ones(1, 1) * a, torch.ones(1, 1) * b, fn, None,
                                                      torch.zeros(1, 1), 100)
        assert abs(r.item() - true) < 1e-3


def test_finite_difference_gradient_wrt_x0():
    """reference tests/test_numerical_validation.py:100-179 (informational there; asserted here at its observed level)."""
    torch.manual_seed(42)
    net = umnn_amd.IntegrandNetwork(3, 2, [20, 20], 1)
    x0 = torch.zeros(10, 3, requires_grad=True)
    x = torch.randn(10, 3)
    h = torch.randn(10, 3)
    flat = torch.cat([p.contiguous().view(-1) for p in net.parameters()])
    umnn_amd.ParallelNeuralIntegral.apply(x0, x, net, flat, h, 20, False).sum().backward()
    eps = 1e-3
    with torch.no_grad():
        base = umnn_amd.ParallelNeuralIntegral.apply(x0, x, net, flat, h, 20, False).sum().item()
        for i in range(3):
            for j in range(3):
                xp = x0.detach().clone()
                xp[i, j] += eps
                fd = (umnn_amd.ParallelNeuralIntegral.apply(xp, x, net, flat, h, 20, False).sum().item() - base) / eps
                assert abs(fd - x0.grad[i, j].item()) < 2e-2 * max(1.0, abs(fd))


def test_full_model_smoke_like_reference_test_jit():
    """reference tests/test_jit.py:89-167: EmbeddingNetwork + UMNNMAF, B=32, d=10, n=20, CCParallel."""
    torch.manual_seed(0)
    emb = umnn_amd.EmbeddingNetwork(in_d=10, hiddens_embedding=[50, 50], hiddens_integrand=[50, 50], out_made=1)
    model = umnn_amd.UMNNMAF(net=emb, input_size=10, nb_steps=20, device="cpu", solver="CCParallel")
    x = torch.randn(32, 10, requires_grad=True)
    z = model.forward(x)
    assert z.shape == (32, 10)
    z.sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    with_grad = sum(1 for p in model.parameters() if p.grad is not None)
    assert with_grad == sum(1 for p in model.parameters() if p.requires_grad)
    ll, z2 = model.compute_ll(torch.randn(32, 10))
    assert ll.shape == (32,) and float(z2.abs().max()) <= 10.0      # block-level compute_ll clamps z (UMNNMAF.py:150)
    assert model.forward(x, method=None) is not None
    model.solver = "nope"
    assert model.forward(x) is None                                   # unknown solver returns None (UMNNMAF.py:107-108)


def test_state_dict_key_names_and_aliases():
    m = umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=3, hidden_derivative=[8, 8], hidden_embedding=[16, 16], embedding_s=4,
                             nb_steps=10, solver="CC")
    keys = set(m.state_dict().keys())
    for want in ("pi", "Flow0.scaling", "Flow0.pi", "Flow0.cc_weights", "Flow0.cc_steps",
                 "Flow0.net.made.net.0.weight", "Flow0.net.made.net.0.bias", "Flow0.net.made.net.0.mask",
                 "Flow0.net.made.net.4.mask", "Flow0.net.parallel_nets.net.0.weight",
                 "Flow1.net.parallel_nets.net.4.bias"):
        assert want in keys, want
    assert m.state_dict()["Flow0.cc_weights"].shape == (11, 1)
    assert not m.nets[0].scaling.requires_grad
    assert m.to("cpu") is m and m.nets[0].to("cpu") is m.nets[0]
    # names the reference's scripts call although the reference never defines them
    for name in ("computell", "forcei_lpschitz", "forceLipshitz", "computeLipshitz"):
        assert callable(getattr(m, name))
    assert callable(m.nets[0].computeLL) and callable(m.nets[0].net.parallel_nets.computeLipshitz)
    L = m.compute_lipschitz(5)
    m.force_lipschitz(1.5)
    assert float(m.compute_lipschitz(20)) <= 1.5 ** 6 * 1.2


def test_set_steps_nb_and_random_steps_eval_mode():
    """UCIExperiments.py:130-132 draws a new even n per batch; the reference then crashes in eval+CCParallel
    (stale cc_weights buffer).  Here the tables follow n."""
    torch.manual_seed(1)
    m = umnn_amd.UMNNMAFFlow(nb_flow=1, nb_in=3, hidden_derivative=[16, 16], hidden_embedding=[16], embedding_s=3,
                             nb_steps=20, solver="CCParallel")
    x = torch.randn(7, 3)
    m.eval()
    with torch.no_grad():
        ref, _ = m.compute_ll(x)
        for n in (10, 37, 98):
            m.set_steps_nb(n)
            ll, _ = m.compute_ll(x)
            assert m.nets[0].cc_weights.shape == (21, 1)      # registered tables keep the constructor's shape
            assert torch.allclose(ll, ref, atol=2e-3)
    # a checkpoint written after set_steps_nb round-trips into a freshly built model (UCIExperiments.py:131-153)
    fresh = umnn_amd.UMNNMAFFlow(nb_flow=1, nb_in=3, hidden_derivative=[16, 16], hidden_embedding=[16], embedding_s=3,
                                 nb_steps=20, solver="CCParallel")
    res = fresh.load_state_dict(m.state_dict())
    assert not res.missing_keys and not res.unexpected_keys


def test_nb_steps_zero_means_random_steps():
    """'nb_steps: 0 for random' (UCIExperiments.py / MNISTExperiment.py build the model with nb_steps <= 0 and call
    set_steps_nb per batch): construction must work, integration before set_steps_nb must say what is wrong."""
    torch.manual_seed(2)
    m = umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=3, hidden_derivative=[16, 16], hidden_embedding=[16], embedding_s=3,
                             nb_steps=0, solver="CCParallel")
    assert m.nets[0].cc_weights.shape == (1, 1)            # the reference registers NaN tables of shape [n+1, 1]
    x = torch.randn(5, 3)
    with pytest.raises(ValueError):
        m.compute_ll(x)
    m.set_steps_nb(30)
    ll, _ = m.compute_ll(x)
    assert torch.isfinite(ll).all()
    umnn_amd.UMNNMAFFlow(nb_flow=1, nb_in=2, hidden_derivative=[8], hidden_embedding=[8], embedding_s=2, nb_steps=-1)


def test_eval_mode_stays_differentiable():
    """model.eval(); ll.backward() must give the same parameter gradients as train mode (the reference's eval-mode
    direct integration is plain ATen and therefore differentiable, UMNNMAF.py:89-105)."""
    torch.manual_seed(3)
    m = umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=3, hidden_derivative=[16, 16], hidden_embedding=[16], embedding_s=3,
                             nb_steps=30, solver="CCParallel")
    x = torch.randn(9, 3)
    grads = {}
    for mode in ("train", "eval"):
        m.train(mode == "train")
        m.zero_grad()
        ll, _ = m.compute_ll(x)
        (-ll.mean()).backward()
        grads[mode] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    assert grads["train"].keys() == grads["eval"].keys() and len(grads["eval"]) > 4
    for k in grads["train"]:
        assert U.scaled_err(grads["eval"][k].numpy(), grads["train"][k].numpy()) < 1e-5, k
    # module-level integrate(compute_grad=False) likewise
    net = m.nets[0].net.parallel_nets
    h = torch.randn(9, 9, requires_grad=True)
    F = umnn_amd.integrate(torch.zeros(9, 3), 30, x / 30, net, h, False)
    F.sum().backward()
    assert h.grad is not None and float(h.grad.abs().max()) > 0


def test_double_backward_raises():
    """create_graph=True through the quadrature op is not supported (ParallelNeuralIntegral.py:91 builds one in the
    reference; no caller uses it): it must raise, not silently hand back a detached gradient."""
    torch.manual_seed(4)
    net = umnn_amd.IntegrandNetwork(2, 3, [8, 8], 1)
    x = torch.randn(4, 2, requires_grad=True)
    h = torch.randn(4, 4, requires_grad=True)
    flat = torch.cat([p.contiguous().view(-1) for p in net.parameters()])
    F = umnn_amd.ParallelNeuralIntegral.apply(torch.zeros(4, 2), x, net, flat, h, 20)
    (gx,) = torch.autograd.grad(F.sum(), x, create_graph=True)
    with pytest.raises(RuntimeError):
        gx.sum().backward()


def test_masked_weight_cache_invalidation():
    """Writes through .data do not bump tensor versions; invalidate_caches() (called by broadcast_parameters) must make
    the conditioner see them."""
    torch.manual_seed(5)
    made = umnn_amd.MADE(4, [16, 16], 8, natural_ordering=True)
    x = torch.randn(6, 4)
    with torch.no_grad():
        y0 = made.raw(x).clone()
        for p in made.parameters():
            p.data.mul_(2)
        umnn_amd.invalidate_caches(made)
        y1 = made.raw(x)
        for p in made.parameters():          # version-bumping in-place writes need no explicit call
            p.mul_(0.5)
        y2 = made.raw(x)
    assert not torch.allclose(y0, y1)
    assert torch.allclose(y0, y2, atol=1e-6)


def test_invert_round_trip_matches_reference():
    G = U.load("g6_invert")
    m = umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=2, hidden_derivative=[50] * 3, hidden_embedding=[32, 32],
                             embedding_s=10, nb_steps=30, solver="CCParallel")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in U.state_dict_of(G).items()})
    m.eval()
    x = torch.from_numpy(G["x"])
    with torch.no_grad():
        z = m(x)
        assert U.rel_err(z.numpy(), G["z"]) < 1e-5
        x_inv = m.invert(torch.from_numpy(G["z"]), iter=5)
    tol = 2 * 100. / 9 ** 5          # two candidate spacings of the last round
    assert float((x_inv - x).abs().max()) < tol
    assert float((x_inv - torch.from_numpy(G["x_inv"])).abs().max()) < tol


def test_small_embedding_configs_that_break_the_reference():
    """d=2/E=1 and d=1/E=2 give MADE nout == 2, which the reference treats as a Gaussian MADE (made.py:114-118)."""
    for d, E in ((2, 1), (1, 2)):
        m = umnn_amd.UMNNMAFFlow(nb_flow=1, nb_in=d, hidden_derivative=[8], hidden_embedding=[8], embedding_s=E,
                                 nb_steps=10, solver="CCParallel")
        ll, z = m.compute_ll(torch.randn(4, d))
        assert ll.shape == (4,) and z.shape == (4, d)


def test_reference_style_imports_via_compat_shim():
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    try:
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
            del sys.modules[k]
        from models import UMNNMAFFlow                      # ToyExperiments.py:1, UCIExperiments.py:1
        from models.UMNN import MonotonicNN, IntegrandNN    # MonotonicMLP.py:5
        from models.UMNN.UMNNMAF import IntegrandNetwork, EmbeddingNetwork, UMNNMAF   # tests/test_jit.py:6
        from models.UMNN.NeuralIntegral import NeuralIntegral
        from models.UMNN.ParallelNeuralIntegral import ParallelNeuralIntegral, compute_cc_weights
        assert UMNNMAFFlow is umnn_amd.UMNNMAFFlow and MonotonicNN is umnn_amd.MonotonicNN
        assert compute_cc_weights(20)[0].shape == (21, 1)
    finally:
        sys.path.remove(os.path.join(ROOT, "compat"))
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
            del sys.modules[k]


def test_mlp_spec_recognition():
    from umnn_amd.nets import mlp_spec
    from umnn_amd import _lib
    s = mlp_spec(umnn_amd.IntegrandNetwork(4, 3, [10, 12], 1))
    assert s is not None and s.hidden_act == _lib.ACT_LEAKY_RELU and s.out_act == _lib.OUT_ELU_PLUS_ONE
    s = mlp_spec(umnn_amd.IntegrandNetwork(4, 3, [10], 1, act_func="Sigmoid"))
    assert s.out_act == _lib.OUT_SIGMOID
    s = mlp_spec(umnn_amd.IntegrandNN(3, [7, 7]))
    assert s.hidden_act == _lib.ACT_RELU and s.out_act == _lib.OUT_ELU_PLUS_ONE
    assert mlp_spec(lambda x, h: x) is None
    assert mlp_spec(nn.Linear(3, 1)) is None
    assert mlp_spec(umnn_amd.IntegrandNetwork(4, 3, [200], 1)) is None        # wider than the kernels cover
    with pytest.raises(KeyError):
        umnn_amd.IntegrandNetwork(4, 3, [8], 1, act_func="Tanh")


def test_pack_fragments_layout_matches_the_documented_k_order():
    """made.pack_fragments (host side of umnn_made_mlp_forward): fragment (t, s), lane (g, rho), slot j holds
    W[16t + rho][32s + 16(j>>2) + 4g + (j&3)] split into two bf16 pieces, zeros outside the matrix (include/umnn_cc.h)."""
    from umnn_amd.made import pack_fragments
    torch.manual_seed(0)
    for N, K in ((37, 70), (16, 32), (1, 1), (100, 6)):
        W = torch.randn(N, K)
        P = pack_fragments(W)
        T, S = (N + 15) // 16, (K + 31) // 32
        assert P.shape == (T, S, 2, 64, 8) and P.dtype == torch.bfloat16
        lane = torch.arange(64)
        g, rho = lane >> 4, lane & 15
        j = torch.arange(8)
        for t in range(T):
            for s in range(S):
                n = (16 * t + rho).view(64, 1).expand(64, 8)
                k = (32 * s + 16 * (j >> 2) + (j & 3)).view(1, 8) + 4 * g.view(64, 1)
                ok = (n < N) & (k < K)
                want = torch.where(ok, W[n.clamp(max=N - 1), k.clamp(max=K - 1)], torch.zeros(()))
                hi = want.bfloat16()
                lo = (want - hi.float()).bfloat16()
                assert torch.equal(P[t, s, 0], hi) and torch.equal(P[t, s, 1], lo), (N, K, t, s)


def test_bench_cpu_baseline_protocol_runs_on_a_small_workload():
    """bench.cpu_baseline (the `cpu_baseline` record of the bench line): both reference solvers, thread sweep, warm-up, median
    over two chunks, budget flag, the all-cores probe in a child process with a time limit."""
    import bench
    cfg = dict(bench.WORKLOADS["toy"], rows=64)
    model = bench.build_model(cfg, "cpu")
    rec = bench.cpu_baseline(cfg, model, budget_s=6.0)
    assert rec["kind"] == "port" and rec["unit"] == "evals/s" and rec["value"] > 0 and rec["cores"] >= 1
    assert set(rec["solvers"]) == {"sequential", "parallel"} and rec["solver"] in rec["solvers"]
    for s in rec["solvers"].values():
        assert s["reps"] >= 3 and s["warmup"] >= 1 and s["chunks"] == 2 and isinstance(s["budget_limited"], bool)
        assert s["min_ms"] <= s["median_ms"] <= s["max_ms"]
    assert rec["value"] == max(s["evals_per_s"] for s in rec["solvers"].values())
