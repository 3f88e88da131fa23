"""Round 6 parity additions (VERDICT r05 "next" item 1): the library default arithmetic (f16x3) pinned against the oracle at the launch
shape the eight-wave toy kernel was tuned on.  Reference arithmetic: models/UMNN/UMNNMAFFlow.py:109-119,
models/UMNN/ParallelNeuralIntegral.py:49-65.  Tolerance: the path's 1e-4."""
import numpy as np
import pytest
import torch

import bench
from oracle import cc_oracle as O
from tests import _util as U
from tests.test_gpu_bench_models import _oracle_blocks, _sample_rows

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_toy_flow_at_65536_rows_under_the_default_matches_oracle(dev):
    """ToyExperiments' flow (ToyExperiments.py:126-127: d = 2, 11-100^4-1, n = 50) at 65 536 x 2 -- the launch the eight-wave
    workgroups (cc_fwd_f16<T=7,...,WAVES=8>) were tuned on in round 5 -- under the library default, sampled rows against the oracle;
    plus the size-independent properties: shard consistency bit for bit and repeat determinism."""
    import umnn_amd
    from umnn_amd import _lib
    assert umnn_amd.get_forward_precision() == "f16x3"
    cfg = dict(bench.WORKLOADS["toy"])
    model = bench.build_model(cfg, dev)
    B = 65536
    x, _ = bench.make_inputs(cfg, B, dev, 1000)
    with torch.no_grad():
        ll, z = model.compute_ll(x)
        kname = _lib.lib().umnn_last_kernel_name().decode()
        assert kname.startswith("cc_fwd_f16<T=7") and "WAVES=8" in kname, kname
        ll2, z2 = model.compute_ll(x)
        assert torch.equal(ll, ll2) and torch.equal(z, z2)
        # a shard of the batch gives its rows (no forward collective: rows are independent) -- to summation-order noise, because a
        # smaller launch may split the node range of an integral over more waves (the plan depends on the launch size)
        lo, hi = 8192 * 3, 8192 * 4
        lls, zs = model.compute_ll(x[lo:hi].contiguous())
        assert U.rel_err(zs.cpu().numpy(), z[lo:hi].cpu().numpy()) < 2e-6
        assert U.rel_err(lls.cpu().numpy(), ll[lo:hi].cpu().numpy()) < 2e-6
    assert umnn_amd.path_taken() == "hip"
    rows = _sample_rows(B, 48, seed=6)
    blocks = _oracle_blocks(model, cfg)
    ll_ref, z_ref = O.flow_compute_ll(blocks, x[rows].cpu().numpy(), cfg["n"])
    assert U.rel_err(z.cpu().numpy()[rows], z_ref) < TOL
    assert U.rel_err(ll.cpu().numpy()[rows], ll_ref) < TOL
    assert torch.isfinite(ll).all() and torch.isfinite(z).all()


@pytest.mark.parametrize("scale", [1e-2, 1e-3, 1e-4])
def test_default_forward_at_small_activation_scales_against_float64(scale, dev):
    """the fp16 pieces of the default guard only the TOP of fp16's range; the low piece of a hidden activation below ~1e-3
    is an fp16 subnormal and vanishes below ~6e-5, so uniformly tiny activations carry 11 bits or fewer.  A 31-50^4-1 integrand whose
    first layer (weights and bias) is scaled down so that every hidden activation is O(scale): F and f(x) of the default against the
    float64 oracle under the path's criterion (1e-4 of max(|ref|, 1)) -- and, so that the limitation is a NUMBER and not a
    footnote, the error of the part of f that the hidden layers carry, f - f(0 activations), relative to its own size, beside the
    bf16x3 mode (whose pieces share fp32's exponent range)."""
    import umnn_amd
    from umnn_amd import integral as I, _lib
    from umnn_amd.nets import mlp_spec
    torch.manual_seed(17)
    d, E, n, B = 5, 30, 50, 200
    net = umnn_amd.IntegrandNetwork(d, 1 + E, [50] * 4, 1)
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        lin[0].weight.mul_(scale)
        lin[0].bias.mul_(scale)
        for m in lin[1:-1]:
            m.bias.mul_(scale)
        for m in lin[1:]:
            m.weight.mul_(2.0)                            # (keeps the signal from shrinking further layer by layer)
    net.to(dev)
    spec = mlp_spec(net)
    x, h = torch.randn(B, d, device=dev) * 2, torch.randn(B, E * d, device=dev)
    Ws = [m.weight.detach().cpu().numpy().astype(np.float64) for m in lin]
    bs = [m.bias.detach().cpu().numpy().astype(np.float64) for m in lin]
    net64 = O.Net(Ws, bs, O.LEAKY, O.ELU1)
    x64, h64 = x.cpu().numpy().astype(np.float64), h.cpu().numpy().astype(np.float64)
    F_ref = O.integrate_parallel(net64, np.zeros_like(x64), x64, h64, n)
    f_ref = O.integrand(net64, x64, h64)
    f_base = O.integrand(O.Net([W * (0.0 if l == 0 else 1.0) for l, W in enumerate(Ws)], [b * 0.0 if l < len(bs) - 1 else b for l, b in enumerate(bs)],
                               O.LEAKY, O.ELU1), x64, h64)      # the output with all hidden activations at zero: ELU(b_out) + 1
    rep = {}
    old = umnn_amd.get_forward_precision()
    try:
        for mode in ("f16x3", "bf16x3", "fp32"):
            umnn_amd.set_precision(mode)
            F, fx, _ = I.hip_forward(spec, None, x, h, n)
            torch.cuda.synchronize()
            F, fx = F.cpu().numpy().astype(np.float64), fx.cpu().numpy().astype(np.float64)
            sig = np.abs(f_ref - f_base).max()
            rep[mode] = (U.rel_err(F, F_ref), U.rel_err(fx, f_ref), float(np.abs(fx - f_ref).max() / sig))
    finally:
        umnn_amd.set_precision(old)
    print(f"small-activation scale {scale}: (F, f(x), hidden-signal relative) per mode:", rep)
    for mode, (eF, ef, _) in rep.items():
        assert eF < TOL and ef < TOL, (mode, eF, ef)
    # the default is no worse than the two-piece bf16 mode on the hidden layers' share of the output at any of these scales + float32 floor
    assert rep["f16x3"][2] <= max(rep["bf16x3"][2], 50 * rep["fp32"][2], 1e-3), rep


@pytest.mark.parametrize("cond", [0, 7])
@pytest.mark.parametrize("B", [1, 37, 1000])
def test_conditioner_training_chain_as_one_node_matches_the_autograd_chain(cond, B, dev):
    """The MADE / ConditionnalMADE training path as ONE autograd node (made._MadeTrainChain: fp32 library GEMMs on the cached masked
    weight, in-place ReLU, umnn_made_relu_bwd_bias for ReLU-backward + the bias gradient in a fixed order) against the same chain under
    ordinary autograd -- the reference's composition, models/UMNN/made.py:16-27,113-119,165-168: outputs to fp32 rounding (same GEMMs),
    every gradient (input, context, all weights and biases) to summation-order noise, masked weight entries exactly zero gradient, and
    bit-identical gradients on a repeat."""
    from umnn_amd import made as M
    torch.manual_seed(3 + B)
    nin, E = 6, 5
    if cond:
        net = M.ConditionnalMADE(nin, cond, [64, 48], (nin + cond) * E, natural_ordering=True).to(dev)
    else:
        net = M.MADE(nin, [64, 48], nin * E, natural_ordering=True).to(dev)
    x = torch.randn(B, nin, device=dev)
    ctx = torch.randn(B, cond, device=dev) if cond else None
    w = torch.randn(B, nin * E, device=dev)
    res = {}
    for fused in (True, False, True):
        M._TRAIN_FUSED["enabled"] = fused
        try:
            net.zero_grad(set_to_none=True)
            xr = x.clone().requires_grad_()
            cr = ctx.clone().requires_grad_() if cond else None
            out = net.raw(xr, cr) if cond else net.raw(xr)
            assert (type(out.grad_fn).__name__ == "_MadeTrainChainBackward") == fused, type(out.grad_fn).__name__
            (out * w).sum().backward()
            grads = [xr.grad.clone()] + ([cr.grad.clone()] if cond else []) + [p.grad.clone() for p in net.parameters()]
        finally:
            M._TRAIN_FUSED["enabled"] = True
        key = "fused2" if (fused and "fused" in res) else ("fused" if fused else "plain")
        res[key] = (out.detach().clone(), grads)
    # (the same fp32 library GEMMs; the hidden layers take bias + ReLU in the GEMM epilogue, which may pick another hipBLASLt kernel and
    # with it another summation order: fp32 rounding level, not bit equality)
    assert float((res["fused"][0] - res["plain"][0]).abs().max()) <= 2e-6 * max(1.0, float(res["plain"][0].abs().max()))
    assert torch.equal(res["fused"][0], res["fused2"][0])
    for a, b in zip(res["fused"][1], res["plain"][1]):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max())), float((a - b).abs().max())
    for a, b in zip(res["fused"][1], res["fused2"][1]):
        assert torch.equal(a, b)
    # the gradient of a masked-out weight entry is exactly zero (what keeps those entries where the initialisation left them)
    layers = [l for l in net.net if isinstance(l, M.MaskedLinear)]
    for l in layers:
        assert float((l.weight.grad * (1 - l.mask)).abs().max()) == 0.0


def test_vae_prior_flow_training_gradients_against_float64_truth(dev):
    """BASELINE configuration C4 (models/vae_lib/models/flows.py:305-323: d = 64, cond_in = 320, 4 blocks, n = 50) -- the flow term of the
    VAE loss differentiated through the whole flow.  VERDICT r05 weak #3: the bf16-storage gradients were compared with this repo's own
    fp32 run only.  Truth here is the SAME model in float64 on the generic ATen quadrature (materialised nodes + autograd: the reference's
    algorithm, ParallelNeuralIntegral.py:66-94,110-123, in double precision); against it, per parameter tensor and scaled by its largest
    entry: the default fp32 path (HIP forward + backward, one-node conditioner chain) to the path's 1e-4, and the bf16-embedding storage
    mode to the storage format's stated resolution."""
    import copy
    import umnn_amd
    cfg = dict(bench.WORKLOADS["vae"])
    model = bench.build_model(cfg, dev).train()
    x, ctx = bench.make_inputs(cfg, 100, dev, 9)

    def flow_term(m, xx, cc):
        z, lj = m.compute_log_jac_bis(xx, cc)
        return (0.5 * (z.double() ** 2).sum(1) - lj.double().sum(1)).mean()          # -log p(z_K) - log|det J|

    model64 = copy.deepcopy(model).double()
    flow_term(model64, x.double(), ctx.double()).backward()
    assert umnn_amd.path_taken() != "hip", "float64 tensors must take the generic ATen quadrature (the truth of this test)"
    truth = {k: p.grad.detach().clone() for k, p in model64.named_parameters() if p.grad is not None}
    del model64
    errs = {}
    for mode in ("fp32", "bf16"):
        model.set_embedding_dtype(torch.bfloat16 if mode == "bf16" else None)
        model.zero_grad(set_to_none=True)
        flow_term(model, x, ctx).backward()
        assert umnn_amd.path_taken() == "hip"
        got = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        assert got.keys() == truth.keys() and len(got) > 20
        errs[mode] = {k: float((got[k].double() - truth[k]).abs().max() / truth[k].abs().max().clamp_min(1e-30)) for k in got}
    model.set_embedding_dtype(None)
    worst = {m: max(e.values()) for m, e in errs.items()}
    print("C4 gradients against float64 truth, worst tensor:", worst)
    assert worst["fp32"] < 1e-4, sorted(errs["fp32"].items(), key=lambda kv: -kv[1])[:3]
    # bf16 storage of the embedding h (8 significant bits, relative 2^-9 per value, read by forward AND backward kernels, d_h rounded to
    # bf16 on the way out): stated bound 5e-2 of a tensor's largest entry, against truth
    assert worst["bf16"] < 5e-2, sorted(errs["bf16"].items(), key=lambda kv: -kv[1])[:3]


def test_integrand_network_scripts_and_traces_on_the_gpu(dev, tmp_path):
    """What the reference's own JIT test does (tests/test_jit.py:170-266, device = "cuda" when there is one): torch.jit.script and
    torch.jit.trace of IntegrandNetwork on GPU tensors, outputs equal to eager mode (also at another batch size), and a traced module
    saved and loaded again.  (The quadrature op itself is an autograd.Function in the reference too -- "cannot be directly JIT
    compiled", ibid. :243-246 -- and is not scripted there either.)"""
    import umnn_amd
    torch.manual_seed(0)
    d = 5
    net = umnn_amd.IntegrandNetwork(d, 2, [50, 50], 1, device=str(dev)).to(dev)
    x, h = torch.randn(10, d, device=dev), torch.randn(10, d, device=dev)
    with torch.no_grad():
        eager = net(x, h)
        scripted = torch.jit.script(net)
        assert torch.allclose(scripted(x, h), eager, rtol=1e-5)
        traced = torch.jit.trace(net, (x, h))
        assert torch.allclose(traced(x, h), eager, rtol=1e-5)
        x2, h2 = torch.randn(5, d, device=dev), torch.randn(5, d, device=dev)
        assert torch.allclose(traced(x2, h2), net(x2, h2), rtol=1e-5)
        path = str(tmp_path / "integrand_traced.pt")
        torch.jit.save(traced, path)
        assert torch.allclose(torch.jit.load(path)(x, h), eager, rtol=1e-5)
