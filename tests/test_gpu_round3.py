"""Round-3 parity and robustness tests on the GPU (the gaps VERDICT r02 / ADVICE r02 listed, one test each):

  * the 65 536 x 63, n = 100 batch ``bench.py`` times as ``full_batch_n1``, through ``compute_ll``, against the oracle on
    sampled rows (UMNNMAFFlow.compute_ll, models/UMNN/UMNNMAFFlow.py:109-119);
  * BASELINE config 5 as written: d = 784 with bf16 storage of inputs and embedding (MNISTExperiment.py:33-46 shape);
  * ``force_lipschitz`` (models/UMNN/UMNNMAF.py:289-301, called per step by UCIExperiments.py:145-146) seen by the HIP path
    and by a captured ``GraphedLL``;
  * sampling (``invert``) with a bf16 embedding; hipGraph captures that are built but not replayed before the next capture;
  * a one-rank RCCL group: broadcast, all-reduce and a captured training step whose gradient hook really issues the
    collective;
  * every fallback off the HIP kernels warns once and reports ``path_taken() == "aten"``.
"""
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest
import torch

import bench
from oracle import cc_oracle as O
from tests import _util as U
from tests.test_gpu_bench_models import _oracle_blocks, _sample_rows, BF16_TOL, TOL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_full_65536_row_batch_compute_ll_matches_oracle_on_sampled_rows(dev):
    """What ``full_batch_n1`` times: BASELINE config C3's un-sharded batch (65 536 x 63, n = 100, 5 blocks) through the
    one-pass compute_ll, default arithmetic; 24 sampled rows (first, last, random) against the oracle at 1e-4."""
    import umnn_amd
    from umnn_amd import _lib
    cfg = dict(bench.WORKLOADS["bsds300"])
    model = bench.build_model(cfg, dev)
    x, _ = bench.make_inputs(cfg, bench.FULL_BATCH_ROWS, dev, 4242)          # bench.py's own full-batch inputs
    launches = _lib.lib().umnn_launch_count()
    with torch.no_grad():
        ll, z = model.compute_ll(x)
    assert _lib.lib().umnn_launch_count() - launches == cfg["nb_flow"] and umnn_amd.path_taken() == "hip"
    assert ll.shape == (bench.FULL_BATCH_ROWS,) and torch.isfinite(ll).all()
    rows = _sample_rows(bench.FULL_BATCH_ROWS, 24, seed=11)
    ll_ref, z_ref = O.flow_compute_ll(_oracle_blocks(model, cfg), x[rows].cpu().numpy(), cfg["n"])
    assert U.rel_err(ll.cpu().numpy()[rows], ll_ref) < TOL
    assert U.rel_err(z.cpu().numpy()[rows], z_ref) < TOL


@pytest.mark.parametrize("mode", ["bf16_embedding", "bf16_everything"])
def test_mnist_shape_d784_with_bf16_storage_matches_oracle(mode, dev):
    """BASELINE config 5 read literally: the d = 784 flow (MNISTExperiment.py:33-46: 5 blocks, MADE [1024]*3, integrand
    31-100-50-50-50-50-1, n = 50, batch 100) with bf16 STORAGE -- the [B, 30*784] embedding written and read as bf16, and
    (second case) x and every result in bf16 too; arithmetic stays fp32 inside the kernels.  Stated tolerance vs the fp32
    oracle on the same (bf16-rounded) inputs: BF16_TOL = 2e-2 of max(|ref|, 1) for z; for log_jac summed over the 784
    dimensions and 5 blocks the per-value 2^-9 roundings add up as a random walk -- bound 2e-2 * sqrt(d)."""
    import umnn_amd
    from umnn_amd import _lib
    cfg = dict(bench.WORKLOADS["mnist"])
    model = bench.build_model(cfg, dev)
    x, _ = bench.make_inputs(cfg, cfg["rows"], dev, 77)
    try:
        model.set_embedding_dtype(torch.bfloat16)
        xs = x.bfloat16() if mode == "bf16_everything" else x
        launches = _lib.lib().umnn_launch_count()
        with torch.no_grad():
            z, lj = model.compute_log_jac_bis(xs)
            ll, z2 = model.compute_ll(xs)
        assert umnn_amd.path_taken() == "hip" and _lib.lib().umnn_launch_count() - launches == 2 * cfg["nb_flow"]
        assert model.nets[0].net.m_embeding.dtype == torch.bfloat16
        assert z.dtype == xs.dtype and lj.dtype == xs.dtype
    finally:
        model.set_embedding_dtype(None)
    rows = _sample_rows(cfg["rows"], 6, seed=3)
    blocks = _oracle_blocks(model, cfg)
    xr = xs[rows].float().cpu().numpy()
    z_ref, lj_ref = O.flow_log_jac(blocks, xr, cfg["n"])
    ll_ref, _ = O.flow_compute_ll(blocks, xr, cfg["n"])
    assert U.rel_err(z.float().cpu().numpy()[rows], z_ref) < BF16_TOL
    assert U.rel_err(z2.float().cpu().numpy()[rows], z_ref) < BF16_TOL
    assert np.all(np.abs(lj.float().cpu().numpy()[rows] - lj_ref) <= BF16_TOL * np.maximum(1.0, np.abs(lj_ref)))
    assert U.rel_err(ll.float().cpu().numpy()[rows], ll_ref) < BF16_TOL * np.sqrt(cfg["d"])


def test_force_lipschitz_is_seen_by_the_hip_path_and_by_a_captured_graph(dev):
    """UCIExperiments.py:145-146 calls ``model.forcei_lpschitz(L)`` after every optimizer step: an in-place rescale of the
    integrand's weights (UMNNMAF.py:289-301).  The kernels read weights live through the cached ``umnn_mlp`` descriptor, so
    the next ``compute_ll`` must equal the oracle on the RESCALED weights; a ``GraphedLL`` built before must re-capture."""
    import umnn_amd
    cfg = dict(bench.WORKLOADS["power"], rows=512)
    model = bench.build_model(cfg, dev)
    x, _ = bench.make_inputs(cfg, cfg["rows"], dev, 21)
    graphed = umnn_amd.GraphedLL(model, x)
    with torch.no_grad():
        ll0, _ = model.compute_ll(x)
        ll0 = ll0.clone()
        assert torch.equal(graphed()[0], ll0)
    before = [p.detach().clone() for p in model.nets[0].net.parallel_nets.parameters()]
    L_before = float(model.compute_lipschitz(20))
    model.forcei_lpschitz(0.6)                     # the script's (misspelt) name; default-init layers have norm ~1.1 > 0.6
    after = list(model.nets[0].net.parallel_nets.parameters())
    assert any(not torch.equal(a, b) for a, b in zip(after, before)), "force_lipschitz(0.6) must rescale default-init layers"
    for net in model.nets:
        for layer in net.net.parallel_nets.net:
            if isinstance(layer, torch.nn.Linear):
                # (the rescale divides by a 10-iteration power-iteration estimate from a random start: a few per cent of slack)
                assert float(umnn_amd.compute_lipschitz_linear(layer.weight.detach(), 50)) <= 0.6 * 1.1
    assert float(model.compute_lipschitz(20)) < L_before
    captures = graphed.captures
    with torch.no_grad():
        ll1, z1 = model.compute_ll(x)
        assert umnn_amd.path_taken() == "hip"
        llg, zg = graphed()
    assert graphed.captures == captures + 1, "the in-place weight rescale must trigger a re-capture"
    assert torch.equal(llg, ll1) and torch.equal(zg, z1)
    rows = _sample_rows(cfg["rows"], 32)
    ll_ref, z_ref = O.flow_compute_ll(_oracle_blocks(model, cfg), x[rows].cpu().numpy(), cfg["n"])
    assert U.rel_err(ll1.cpu().numpy()[rows], ll_ref) < TOL and U.rel_err(z1.cpu().numpy()[rows], z_ref) < TOL
    assert not torch.allclose(ll1, ll0)            # and it really was a different model
    # training step after the rescale: the HIP backward reads the same live weights
    model.train()
    model.zero_grad()
    ll_t, _ = model.compute_ll(x[:64])
    (-ll_t.mean()).backward()
    assert umnn_amd.path_taken() == "hip"
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_invert_with_bf16_embedding_matches_fp32_embedding(dev):
    """the fused inversion handed a bf16 embedding to an entry point that reads fp32.  With
    set_embedding_dtype(bfloat16) sampling must agree with the fp32-embedding samples to the embedding's rounding (2^-9 on h,
    i.e. on the offset and the integrand's conditioning), and round-trip through forward."""
    import umnn_amd
    torch.manual_seed(5)
    d = 6
    model = umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=d, hidden_derivative=[50] * 3, hidden_embedding=[64, 64], embedding_s=8,
                                 nb_steps=30, solver="CCParallel").to(dev).eval()
    z = torch.randn(200, d, device=dev)
    with torch.no_grad():
        x32 = model.invert(z, iter=12)
        assert umnn_amd.path_taken() == "hip"
        model.set_embedding_dtype(torch.bfloat16)
        try:
            x16 = model.invert(z, iter=12)
            assert model.nets[0].net.m_embeding.dtype == torch.bfloat16
            z_back = model.forward(x16)
        finally:
            model.set_embedding_dtype(None)
    assert torch.isfinite(x16).all()
    assert float((x16 - x32).abs().max()) < 5e-2 * max(1.0, float(x32.abs().max()))
    assert float((z_back - z).abs().max()) < 5e-2 * max(1.0, float(z.abs().max()))


def test_graph_captures_own_their_counters(dev):
    """the row-arrival counters of the one-pass log-likelihood must not be created-and-cached inside a
    capture.  Two GraphedLL objects built back to back, neither replayed before the other is captured, then a weight change
    that makes the FIRST call of each re-capture: every replay must return the eager result."""
    import umnn_amd
    from umnn_amd import integral as I
    I._row_counters.clear()                         # as in a fresh process: the first capture finds no eager buffer
    cfg = dict(bench.WORKLOADS["toy"], rows=777)
    model = bench.build_model(cfg, dev)
    xa, _ = bench.make_inputs(cfg, 777, dev, 1)
    xb, _ = bench.make_inputs(cfg, 777, dev, 2)
    ga = umnn_amd.GraphedLL(model, xa, warmup=0)
    gb = umnn_amd.GraphedLL(model, xb, warmup=0)
    with torch.no_grad():
        for p in model.nets[0].net.parallel_nets.parameters():
            p.mul_(1.25)                            # versions move: the first call of each graph re-captures
    lb = gb()[0].clone()
    la = ga()[0].clone()
    assert ga.captures == 2 and gb.captures == 2
    with torch.no_grad():
        ea, eb = model.compute_ll(xa)[0], model.compute_ll(xb)[0]
    assert torch.equal(la, ea) and torch.equal(lb, eb)
    for _ in range(3):
        assert torch.equal(ga()[0], ea) and torch.equal(gb()[0], eb)
    for t in I._row_counters.values():
        assert int(t.abs().sum()) == 0


_RCCL_SCRIPT = r"""
import faulthandler; faulthandler.enable()
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29653")
import umnn_amd
from umnn_amd import sharding
def mark(m): print("MARK", m, flush=True)
rank, world, dev = sharding.init_from_env(backend="nccl", force_group=True)
mark("group")
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1 and dev.type == "cuda"
torch.manual_seed(0)
model = umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=6, hidden_derivative=[50] * 4, hidden_embedding=[64, 64], embedding_s=30,
                             nb_steps=20, solver="CCParallel").to(dev).train()
w0 = [p.detach().clone() for p in model.parameters()]
sharding.broadcast_parameters(model, force=True)                      # RCCL broadcast of every parameter and buffer
mark("broadcast")
assert all(torch.equal(a, b) for a, b in zip(w0, model.parameters()))
t = torch.arange(1024., device=dev)
sharding._all_reduce_sum(t)                                           # RCCL all_reduce on a device tensor
torch.cuda.synchronize()
assert torch.equal(t, torch.arange(1024., device=dev))
mark("all_reduce")
x = torch.randn(100, 6, device=dev)
out = model.compute_ll(x)
(-out[0].mean()).backward()
del out                   # no autograd graph built on the DEFAULT stream may outlive this point (ll AND z hold it): its
torch.cuda.synchronize()  # AccumulateGrad nodes would run on the legacy stream during the capture below (GraphedTrainStep's docstring)
mark("backward")
g0 = [p.grad.detach().clone() for p in model.parameters() if p.requires_grad]
sharding.allreduce_gradients(model, world, force=True)                # the flattened all-reduce, not short-circuited
g1 = [p.grad for p in model.parameters() if p.requires_grad]
assert all(torch.equal(a, b) for a, b in zip(g0, g1))
base = g1[0]._base if g1[0]._base is not None else g1[0]
assert all((g._base is base) for g in g1), "gradients must be views of the one reduced buffer"
mark("allreduce_gradients")
# one eager data-parallel optimisation step with the collective, on a side stream as a training loop would run it
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, capturable=True)
ref = [p.detach().clone() for p in model.parameters()]
losses = []
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for it in range(3):
        opt.zero_grad(set_to_none=True)
        out = model.compute_ll(x)
        loss = -out[0].mean()
        loss.backward()
        sharding.allreduce_gradients(model, world, force=True)
        torch.nn.utils.clip_grad_value_(list(model.parameters()), 10.0)
        opt.step()
        losses.append(float(loss))
        del out, loss            # (ll AND z carry the autograd graph)
torch.cuda.current_stream().wait_stream(side)
assert losses[2] < losses[0] and any(not torch.equal(a, b) for a, b in zip(ref, model.parameters()))
mark("eager steps %s" % losses)
# graphs: inference capture works beside the process group; a captured training step must REFUSE a collective hook
model.eval()
gl = umnn_amd.GraphedLL(model, x)
with torch.no_grad():
    assert torch.equal(gl()[0], model.compute_ll(x)[0])
mark("GraphedLL beside the group")
model.train()
try:
    umnn_amd.GraphedTrainStep(model, opt, x, clip_value=10.0, grad_hook=lambda m: sharding.allreduce_gradients(m, world, force=True))
    raise SystemExit("GraphedTrainStep accepted a collective hook under a nccl group")
except NotImplementedError:
    mark("graphed train step refuses the collective hook")
model.zero_grad(set_to_none=True)
step = umnn_amd.GraphedTrainStep(model, opt, x, clip_value=10.0)          # without a hook it captures beside the group
mark("GraphedTrainStep without hook beside the group")
l1 = float(step()); l2 = float(step()); l3 = float(step(torch.randn(100, 6, device=dev)))
assert all(map(lambda v: v == v and abs(v) < 1e6, (l1, l2, l3))), (l1, l2, l3)
assert l2 < l1
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK", losses, l1, l2, l3)
"""


def test_one_rank_rccl_group_broadcast_allreduce_and_captured_train_step(dev, tmp_path):
    """RCCL had never executed for this code.  One GPU allows a one-rank ``nccl`` group: init with
    ``device_id=``, ``broadcast_parameters``, a forced ``_all_reduce_sum`` / ``allreduce_gradients`` on device tensors, eager
    optimisation steps with the collective, hipGraph captures beside the group.  First contact found that capturing an RCCL
    collective in a hipGraph segfaults on this stack: ``GraphedTrainStep`` refuses a gradient hook under a nccl group
    (umnn_amd/graphs.py has the details), which is asserted here."""
    script = tmp_path / "rccl_world1.py"
    script.write_text(_RCCL_SCRIPT.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_fallbacks_off_the_hip_kernels_warn_once_and_report_their_path(dev):
    """(i) a net whose HIP backward only has the spilling generic wide kernels (since the zero-padded route of
    pad_to_exact_family: only deep nets whose padded images do not fit the LDS) differentiates with the ATen
    chain -- with a RuntimeWarning and path_taken() == 'aten'; (ii) in-kernel inversion is bf16x3-only: under
    set_precision('fp32') invert() runs the host-driven search on the fp32 forward kernels -- with a warning; results agree."""
    import umnn_amd
    from umnn_amd import integral as I
    I._warned.clear()
    I._bwd_kind.clear()
    torch.manual_seed(1)
    # five unequal hidden layers above 63 units: zero-padded to the 7-tile family its weight images (4 x 45 KB) exceed the LDS
    net = umnn_amd.IntegrandNetwork(3, 1 + 4, [100, 72, 80, 96, 70], 1).to(dev)
    x = torch.randn(32, 3, device=dev, requires_grad=True)
    h = torch.randn(32, 12, device=dev, requires_grad=True)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        F = umnn_amd.ParallelNeuralIntegral.apply(torch.zeros_like(x), x, net, umnn_amd.flow._flatten(net.parameters()), h, 20)
        assert umnn_amd.path_taken() == "hip"
        F.sum().backward()
        assert umnn_amd.backward_path_taken() == "aten"      # (backward runs on autograd's thread: process-wide record)
        n_first = sum("materialised ATen chain" in str(w.message) for w in rec)
        F2 = umnn_amd.ParallelNeuralIntegral.apply(torch.zeros_like(x), x, net, umnn_amd.flow._flatten(net.parameters()), h, 20)
        F2.sum().backward()
        n_second = sum("materialised ATen chain" in str(w.message) for w in rec)
    assert n_first == 1 and n_second == 1, "the fallback is announced exactly once per net shape"
    # thread-local force_generic: another thread's integrals are not rerouted
    import threading
    seen = {}
    with I.force_generic():
        def other():
            with torch.no_grad():
                umnn_amd.integrate(torch.zeros(8, 3, device=dev), 20, torch.ones(8, 3, device=dev) / 20, net,
                                   torch.randn(8, 12, device=dev))
            seen["path"] = umnn_amd.path_taken()
        th = threading.Thread(target=other)
        th.start()
        th.join()
        with torch.no_grad():
            umnn_amd.integrate(torch.zeros(8, 3, device=dev), 20, torch.ones(8, 3, device=dev) / 20, net,
                               torch.randn(8, 12, device=dev))
        assert umnn_amd.path_taken() == "aten"
    assert seen["path"] == "hip"
    # (ii) inversion under exact-products precision: in-kernel and silent for every net -- a 50-wide one on three bf16 pieces (round
    # 4), a 100-wide one (8 tiles x 3 pieces do not fit the register file) on two fp16 pieces, the fp32-level search of the default
    # arithmetic (round 5; until then: host-driven search, announced)
    from umnn_amd import _lib
    for hidw, three_piece in ((100, False), (50, True)):
        model = umnn_amd.UMNNMAFFlow(nb_flow=1, nb_in=3, hidden_derivative=[hidw] * 3, hidden_embedding=[32, 32], embedding_s=6,
                                     nb_steps=30, solver="CCParallel").to(dev).eval()
        z = torch.randn(64, 3, device=dev)
        with torch.no_grad():
            x_fast = model.invert(z, iter=10)
        old = umnn_amd.get_forward_precision()
        umnn_amd.set_precision("fp32")
        try:
            with warnings.catch_warnings(record=True) as rec:
                warnings.simplefilter("always")
                with torch.no_grad():
                    x_exact = model.invert(z, iter=10)
                assert not any("host-driven bracket search" in str(w.message) for w in rec)
                name = _lib.lib().umnn_last_kernel_name().decode()
                assert ("PARTS=3" in name) == three_piece and name.startswith("cc_invert_bf16<" if three_piece else "cc_invert_f16<"), name
        finally:
            umnn_amd.set_precision(old)
        # both searches end within the bracket resolution 100 * (2/9)^10 ~ 3e-5 of the same root unless a candidate tie broke differently
        assert float((x_fast - x_exact).abs().median()) < 1e-3


def test_set_option_rejects_out_of_range_values(dev):
    from umnn_amd import _lib
    for name, bad in (("fwd_p", 3), ("fwd_ns", 3), ("fwd_precision", 7), ("bwd_precision", 2), ("bwd_ns", 64), ("fwd_pipe", 5)):
        old = _lib.get_option(name)
        with pytest.raises(RuntimeError):
            _lib.set_option(name, bad)
        assert _lib.get_option(name) == old


@pytest.mark.parametrize("shape", [(257, 63, 30, [50] * 4, 100), (100, 6, 30, [50] * 3, 50), (64, 5, 8, [50] * 2, 20),
                                   (33, 7, 12, [40, 56, 33, 48], 30), (5, 3, 4, [48, 60, 36], 7), (2000, 2, 10, [50] * 4, 20)])
@pytest.mark.parametrize("with_gfx", [False, True])
def test_software_pipelined_backward_agrees_with_the_round2_loop(shape, with_gfx, dev):
    """cc_bwd_swp_kernel (F(k+1) overlapped with B(k), a_l through LDS slots, W^T fragments read out of the forward image
    with transposing reads) performs the same MFMAs on the same operands in the same accumulation order as
    cc_bwd_bf16_kernel.  The Leibniz outputs (dx0, dx) and the output layer's gradient must match bit for bit (the forward
    recompute is the same instruction stream); d_h and the hidden layers' d_theta may differ in the last bits -- the
    compiler contracts the split residual ``x - bf16(x)`` with the product that formed x differently in the two code shapes
    (measured 1e-7 of the largest entry) -- so they are held to 2e-6, two orders inside the path tolerance.  Every variant
    (L = 2..4 hidden layers, LIVE = 13 / 0) and the small-batch node-range split are covered."""
    import umnn_amd
    from umnn_amd import _lib
    from umnn_amd import integral as I
    from umnn_amd.nets import mlp_spec
    B, d, E, hid, n = shape
    torch.manual_seed(B * 7 + d)
    net = umnn_amd.IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.mul_(1.7)
    spec = mlp_spec(net)
    x, x0 = torch.randn(B, d, device=dev) * 2, torch.randn(B, d, device=dev) * 0.3
    h, gg = torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev)
    gf = torch.randn(B, d, device=dev) if with_gfx else None
    outs = {}
    for swp in (0, 1):
        with _lib.options(bwd_swp=swp, bwd_ws=0):
            outs[swp] = I.hip_backward(spec, x0, x, h, gg, gf, n)
            name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
            assert ("SWP" in name) == bool(swp), name
            again = I.hip_backward(spec, x0, x, h, gg, gf, n)
            assert all(torch.equal(u, v) for u, v in zip(outs[swp], again)), "each loop is bit-reproducible"
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    n_last = spec.linears[-1].weight.numel() + 1
    assert torch.equal(outs[0][3][-n_last:], outs[1][3][-n_last:])
    for a_, b_ in zip(outs[0][2:], outs[1][2:]):
        assert U.scaled_err(b_.cpu().numpy(), a_.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("case", [
    # B, d, E, hidden, n, g_fx, out_act, x dtype, h dtype, inv_f
    (300, 63, 30, [50] * 4, 20, True, "ELU", torch.float32, torch.float32, False),
    (301, 63, 30, [48, 60, 36, 50], 12, True, "ELU", torch.float32, torch.float32, False),      # LIVE = 0 variant, ragged last tile
    (2100, 8, 10, [50] * 4, 15, False, "Sigmoid", torch.float32, torch.float32, False),
    (1100, 16, 6, [50] * 4, 7, True, "ELU", torch.bfloat16, torch.bfloat16, False),
    (1030, 16, 5, [50] * 4, 9, True, "ELU", torch.float32, torch.float32, True),
    (300, 63, 30, [50] * 4, 1, True, "ELU", torch.float32, torch.float32, False),      # fewer elements per tile than pipeline stages
    (300, 63, 30, [50] * 4, 2, False, "ELU", torch.float32, torch.float32, False),
])
def test_weight_stationary_backward_agrees_with_the_pipelined_loop_and_fp32(case, dev):
    """cc_bwd_ws_kernel (eight role waves per workgroup: the weights of one layer resident in each GEMM wave's registers,
    tile-nodes passed from wave to wave through LDS tiles, dW on 32x32x16 MFMAs) against cc_bwd_swp_kernel and the exact
    fp32 kernels.  Same six-term recompute, three-term delta chain and three-term dW; the order of a few additions differs
    (four partial sums in the output layer's dot product, dW summed per 32x32 tile), so outputs are compared to 5e-6 of their
    largest entry against the pipelined loop and within the bf16x3 path tolerance against fp32.  Covers both register
    variants, a ragged last tile, the tangent element (g_fx), sigmoid outputs, bf16 storage and the 1/f operator."""
    import umnn_amd
    from umnn_amd import _lib
    from umnn_amd import integral as I
    from umnn_amd.nets import mlp_spec
    B, d, E, hid, n, with_gfx, out_act, xdt, hdt, inv_f = case
    torch.manual_seed(B * 7 + d)
    net = umnn_amd.IntegrandNetwork(d, 1 + E, hid, 1, act_func=out_act).to(dev)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.mul_(1.7)
    spec = mlp_spec(net)
    x, x0 = (torch.randn(B, d, device=dev) * 2).to(xdt), (torch.randn(B, d, device=dev) * 0.3).to(xdt)
    h, gg = torch.randn(B, E * d, device=dev).to(hdt), torch.randn(B, d, device=dev).to(xdt)
    gf = torch.randn(B, d, device=dev).to(xdt) if with_gfx else None
    outs = {}
    for key, ws, prec in (("swp", 0, "bf16x3"), ("ws", 1, "bf16x3"), ("fp32", 0, "fp32")):
        _lib.set_backward_precision(prec)
        try:
            with _lib.options(bwd_ws=ws, bwd_ws16=0):
                outs[key] = I.hip_backward(spec, x0, x, h, gg, gf, n, inv_f=inv_f)
                name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
                if key == "ws":
                    assert ",WS>" in name, name
                    again = I.hip_backward(spec, x0, x, h, gg, gf, n, inv_f=inv_f)
                    assert all(torch.equal(u, v) for u, v in zip(outs[key], again)), "bit-reproducible (no atomics)"
                else:
                    assert ",WS>" not in name, name
        finally:
            _lib.set_backward_precision("bf16x3")
    storage = 2e-2 if xdt == torch.bfloat16 else 0.0        # outputs stored as bf16: one rounding of the last bit
    for i, nm in enumerate(("dx0", "dx", "dh", "dtheta")):
        a_, b_, r_ = (outs[k][i].float().cpu().numpy() for k in ("swp", "ws", "fp32"))
        assert np.isfinite(b_).all(), nm
        assert U.scaled_err(b_, a_) < max(5e-6, storage), (nm, U.scaled_err(b_, a_))
        assert U.scaled_err(b_, r_) < max(1e-3 if nm == "dh" else 2e-4, storage), (nm, U.scaled_err(b_, r_))


def test_weight_stationary_backward_is_only_taken_for_large_unsplit_batches(dev):
    """The workgroup pipeline needs a long element stream per workgroup: small batches (fewer than four tiles per workgroup, or a
    node range split over several work items) stay on the software-pipelined loop, as do nets that are not four hidden layers."""
    import umnn_amd
    from umnn_amd import _lib
    from umnn_amd import integral as I
    from umnn_amd.nets import mlp_spec
    for B, d, hid in ((64, 63, [50] * 4), (300, 63, [50] * 3)):
        net = umnn_amd.IntegrandNetwork(d, 31, hid, 1).to(dev)
        spec = mlp_spec(net)
        x, h, gg = torch.randn(B, d, device=dev), torch.randn(B, 30 * d, device=dev), torch.randn(B, d, device=dev)
        with _lib.options(bwd_ws=1):
            I.hip_backward(spec, None, x, h, gg, None, 20)
            name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
        assert "SWP" in name, name


def test_weight_stationary_backward_at_the_benchmarked_size(dev):
    """The configuration ``bench.py --mode train`` times (8192 x 63 integrals per block, n = 100, 50-wide net): every workgroup of
    the pipeline streams ~126 tiles x 102 elements.  The bf16 pipeline against the software-pipelined loop on the same inputs
    (same six-term recompute: agreement to 5e-6, d_h 2e-5), both workgroup pipelines bit-reproducible and on the kernels the
    defaults name.  What the fp16-piece pipeline -- the default at this size -- is held to is FLOAT64 TRUTH, not a sibling kernel:
    tests/test_gpu_round5.py::test_default_backward_against_float64_truth_at_the_benchmarked_c3_size (same shape and seed).  Round
    4 bounded it by 5e-4 against the loop here; measured against truth in round 5 it is the LOOP's arithmetic (six-term recompute,
    three-term bf16 delta / dW) that sits 2.4e-4 from the float64 d_theta, the fp16 pipeline 5.8e-5, the exact-fp32 kernels 3.1e-5,
    a float32 ATen run of the reference's own algorithm 2.7e-5 (profiles/r05/bwd_truth64_c3.txt)."""
    import umnn_amd
    from umnn_amd import _lib
    from umnn_amd import integral as I
    from umnn_amd.nets import mlp_spec
    B, d, E, n = 8192, 63, 30, 100
    torch.manual_seed(3)
    net = umnn_amd.IntegrandNetwork(d, 1 + E, [50] * 4, 1).to(dev)
    spec = mlp_spec(net)
    x, h = torch.randn(B, d, device=dev), torch.randn(B, E * d, device=dev)
    gg, gf = torch.randn(B, d, device=dev), torch.randn(B, d, device=dev)
    outs = {}
    for key, ws, ws16, want in (("swp", 0, 0, "SWP>"), ("ws", 1, 0, "cc_bwd_bf16<L=4,LIVE=13,WS>"), ("ws16", 1, 1, "cc_bwd_f16<L=4,LIVE=13,WS>")):
        with _lib.options(bwd_ws=ws, bwd_ws16=ws16):
            outs[key] = I.hip_backward(spec, None, x, h, gg, gf, n)
            name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
            assert want in name, name
            if key != "swp":
                again = I.hip_backward(spec, None, x, h, gg, gf, n)
                assert all(torch.equal(u, v) for u, v in zip(outs[key][1:], again[1:])), key
    for i, nm in ((1, "dx"), (2, "dh"), (3, "dtheta")):
        a_, b_, c_ = (outs[k][i].cpu().numpy() for k in ("swp", "ws", "ws16"))
        assert np.isfinite(b_).all() and np.isfinite(c_).all(), nm
        # (at this size a handful of the 3e8 kink decisions differ between any two summation orders: dh is compared at 2e-5)
        assert U.scaled_err(b_, a_) < (2e-5 if nm == "dh" else 5e-6), (nm, U.scaled_err(b_, a_))


@pytest.mark.parametrize("hid, with_gfx", [([100, 50, 50, 50, 50], True), ([112, 48, 60, 36, 50], False)])
def test_weight_stationary_middle_stage_of_the_three_stage_backward(hid, with_gfx, dev):
    """Nets with a wide first hidden layer (MNISTExperiment's 31-100-50-50-50-50-1): the middle stage of the three-stage backward
    (cc_backward_front.hip) runs as the workgroup pipeline too -- z_2 fetched from HBM by wave Ca one element ahead, delta_2
    written back by wave B1 -- when a chunk has at least four tiles per workgroup.  Against the one-pass middle kernel of round
    2 on the same inputs and against the exact-fp32 kernels."""
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
    for key, ws, prec in (("mid", 0, "bf16x3"), ("ws", 1, "bf16x3"), ("fp32", 0, "fp32")):
        _lib.set_backward_precision(prec)
        try:
            with _lib.options(bwd_ws=ws):
                outs[key] = I.hip_backward(spec, x0, x, h, gg, gf, n)
                name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
                if key == "ws":
                    assert "WS,FRONT" in name, name
                    again = I.hip_backward(spec, x0, x, h, gg, gf, n)
                    assert all(torch.equal(u, v) for u, v in zip(outs[key], again))
                elif key == "mid":
                    assert "FRONT" in name and "WS" not in name, name
        finally:
            _lib.set_backward_precision("bf16x3")
    for i, nm in enumerate(("dx0", "dx", "dh", "dtheta")):
        a_, b_, r_ = (outs[k][i].cpu().numpy() for k in ("mid", "ws", "fp32"))
        assert np.isfinite(b_).all(), nm
        assert U.scaled_err(b_, a_) < 5e-6, (nm, U.scaled_err(b_, a_))
        # (dh against fp32: the 100-wide first layer at 1.5x weights flips a few more LeakyReLU kinks than the 50-wide nets do, for
        # either bf16x3 kernel alike: 1.1e-3 measured)
        assert U.scaled_err(b_, r_) < (3e-3 if nm == "dh" else 2e-4), (nm, U.scaled_err(b_, r_))


def test_weight_stationary_backward_with_relu_hidden_layers(dev):
    """ReLU hidden layers (slope 0: MonotonicNN's activation) through the workgroup pipeline: the activation derivative is read off
    the sign of the leading bf16 piece (a clamped negative is -0: not positive), the tangent element multiplies by it."""
    from umnn_amd import _lib
    from umnn_amd import integral as I
    from umnn_amd.nets import MlpSpec
    rng = np.random.RandomState(5)
    B, d, E, n = 2100, 8, 6, 11
    sizes = [1 + E] + [50, 50, 50, 50] + [1]
    lin = []
    for i in range(len(sizes) - 1):
        m = torch.nn.Linear(sizes[i], sizes[i + 1])
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy((rng.randn(sizes[i + 1], sizes[i]) * (1.6 / np.sqrt(sizes[i]))).astype(np.float32)))
            m.bias.copy_(torch.from_numpy((rng.randn(sizes[i + 1]) * 0.3).astype(np.float32)))
        lin.append(m.to(dev))
    spec = MlpSpec(lin, _lib.ACT_RELU, _lib.OUT_ELU_PLUS_ONE)
    # (seeded: with slope 0 a pre-activation within rounding of zero switches a whole path on or off, so ANY two arithmetics --
    # the bf16x3 kernels against fp32 included -- can differ by 1e-3..1e-2 on single entries for an unlucky draw; tools/_relu_loop.py
    # sweeps 40 seeds: the two bf16x3 kernels agree to 3e-6 on every one of them, and differ from fp32 identically)
    torch.manual_seed(1)
    x, x0 = torch.randn(B, d, device=dev) * 2, torch.randn(B, d, device=dev) * 0.3
    h, gg, gf = torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev), torch.randn(B, d, device=dev)
    outs = {}
    for key, ws, prec in (("swp", 0, "bf16x3"), ("ws", 1, "bf16x3"), ("fp32", 0, "fp32")):
        _lib.set_backward_precision(prec)
        try:
            with _lib.options(bwd_ws=ws, bwd_ws16=0):
                outs[key] = I.hip_backward(spec, x0, x, h, gg, gf, n)
                name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
                assert (",WS>" in name) == (key == "ws"), name
        finally:
            _lib.set_backward_precision("bf16x3")
    for i, nm in enumerate(("dx0", "dx", "dh", "dtheta")):
        a_, b_, r_ = (outs[k][i].cpu().numpy() for k in ("swp", "ws", "fp32"))
        assert np.isfinite(b_).all(), nm
        assert U.scaled_err(b_, a_) < 5e-6, (nm, U.scaled_err(b_, a_))
        assert U.scaled_err(b_, r_) < (2e-3 if nm == "dh" else 2e-4), (nm, U.scaled_err(b_, r_))


def test_flow_training_gradients_through_the_workgroup_pipeline_match_the_generic_route(dev):
    """-mean(ll).backward() of a UMNN-MAF flow at a batch large enough for the workgroup-pipeline backward, against the generic
    ATen quadrature (the reference's algorithm on torch ops: no HIP kernel involved) run in FLOAT64 as the truth -- every parameter's
    gradient and the input's.  At 16 800 integrals x 21 nodes the float32 run of that same route is itself 1.7e-4 of the largest
    entry away from its float64 run (summation order over 350 k terms), while the HIP path is 1.1e-5 away (bf16x3 backward; 2e-6
    with the exact-fp32 kernels): tools/_gen_cmp.py prints the table.  So the float32 generic route is only held to 5e-4 here, and
    the new kernel to 5e-5 of the float64 truth."""
    import umnn_amd
    from umnn_amd import _lib
    from umnn_amd import integral as I
    kw = dict(nb_flow=2, nb_in=8, hidden_derivative=[50, 50, 50, 50], hidden_embedding=[64, 64], embedding_s=10, nb_steps=20,
              solver="CCParallel")
    torch.manual_seed(7)
    m = umnn_amd.UMNNMAFFlow(**kw).to(dev).train()
    x = (torch.randn(2100, 8, device=dev) * 0.8).requires_grad_()

    def grads_of(model, xin):
        return {"x": xin.grad.detach().double().clone(),
                **{k: p.grad.detach().double().clone() for k, p in model.named_parameters() if p.grad is not None}}

    with _lib.options(bwd_ws=1):
        ll, _ = m.compute_ll(x)
        assert umnn_amd.path_taken() == "hip"
        (-ll.mean()).backward()
        assert umnn_amd.backward_path_taken() == "hip"
        assert ",WS>" in _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
    hip, ll_hip = grads_of(m, x), ll.detach().double().clone()
    m.zero_grad(set_to_none=True)
    x.grad = None
    with I.force_generic():
        ll, _ = m.compute_ll(x)
        (-ll.mean()).backward()
    gen32 = grads_of(m, x)
    m64 = umnn_amd.UMNNMAFFlow(**kw).to(dev).train()
    m64.load_state_dict(m.state_dict())
    m64 = m64.double()
    x64 = x.detach().double().requires_grad_()
    with I.force_generic():
        ll64, _ = m64.compute_ll(x64)
        (-ll64.mean()).backward()
    truth = grads_of(m64, x64)
    assert U.rel_err(ll_hip.cpu().numpy(), ll64.detach().cpu().numpy()) < TOL
    err = lambda a_, b_: max(U.scaled_err(a_[k].cpu().numpy(), b_[k].cpu().numpy()) for k in truth)
    assert err(hip, truth) < 5e-5, err(hip, truth)
    assert err(hip, gen32) < 5e-4, err(hip, gen32)


@pytest.mark.parametrize("hid,E", [([100, 80, 70], 10), ([70, 90], 30), ([64, 100, 64, 100], 5), ([67, 65, 66], 12),
                                   ([120, 112, 116], 10), ([127, 70], 30)])
def test_unequal_wide_integrand_nets_differentiate_on_the_zero_padded_shape_exact_kernels(hid, E, dev):
    """VERDICT r02 #9 / missing #4: integrand nets with several UNEQUAL hidden layers above 63 units used to leave the HIP
    backward (its generic wide variants spill hundreds of registers) for the materialised ATen chain.  They now run the
    shape-exact fp32 kernels of the 5- / 7- / 8-tile family that holds their widest layer, zero-padded virtually (tile and
    K-step counts only; cc_backward.hip pad_to_exact_family): gradients against the float64 ATen chain (the reference's own
    algorithm, ParallelNeuralIntegral.py:66-94,110-123) and the float32 one, no fallback warning, bit-repeatable."""
    import copy
    import umnn_amd
    from umnn_amd import _lib, integral as I
    from umnn_amd.nets import IntegrandNN, mlp_spec
    torch.manual_seed(len(hid) + E)
    NI, n = 64 * 8, 20
    f = IntegrandNN(1 + E, hid).to(dev)
    x0 = torch.zeros(NI, 1, device=dev)
    x = torch.randn(NI, 1, device=dev) * 2
    h = torch.randn(NI, E, device=dev)
    g = torch.randn(NI, 1, device=dev)
    I._warned.clear()
    I._bwd_kind.clear()
    assert I._hip_backward_ok(mlp_spec(f), x, h)

    def grads(module, dtype, generic):
        xs = [t.detach().to(dtype).requires_grad_(True) for t in (x0, x, h)]
        if generic:
            with I.force_generic():
                out = I.ParallelNeuralIntegral.apply(xs[0], xs[1], module, I._flatten(module.parameters()), xs[2], n)
        else:
            out = I.ParallelNeuralIntegral.apply(xs[0], xs[1], module, I._flatten(module.parameters()), xs[2], n)
        gs = torch.autograd.grad(out, xs + list(module.parameters()), g.to(dtype))
        return torch.cat([t.reshape(-1).double() for t in gs])

    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        launches = _lib.lib().umnn_launch_count()
        hip = grads(f, torch.float32, False)
        assert _lib.lib().umnn_launch_count() > launches
        name = _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()
        assert not any("materialised ATen chain" in str(w.message) for w in rec)
    assert any(k in name for k in ("KS=17", "KS=26", "KS=32")), name
    assert torch.equal(hip, grads(f, torch.float32, False)), "bit-repeatable"
    aten32 = grads(f, torch.float32, True)
    truth = grads(copy.deepcopy(f).double(), torch.float64, True)
    scale = truth.abs().max().item()
    assert (hip - truth).abs().max().item() <= 2e-5 * scale
    assert (hip - aten32).abs().max().item() <= 5e-5 * scale
