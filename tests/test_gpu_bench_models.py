"""Parity of the EXACT models bench.py times (BASELINE configs C1-C4), at the benchmarked batch sizes, through the default
arithmetic (f16x3 quadrature kernels -- the library default since round 5 --, K-concatenated bf16 conditioner GEMMs, the one-pass
log-likelihood epilogue), and through bf16x3 / exact fp32, against the pinned CPU oracle on sampled rows (rows are independent; the oracle needs seconds for 32 of them).

Reference arithmetic being matched: UMNNMAFFlow.compute_ll / compute_log_jac_bis (models/UMNN/UMNNMAFFlow.py:109-130),
UMNNMAF.forward / compute_log_jac (models/UMNN/UMNNMAF.py:76-139).  Tolerance: the path's 1e-4 in every mode,
including the default; bf16 *callers* (configuration C4) state their own tolerance below.
"""
import numpy as np
import pytest
import torch

import bench
from oracle import cc_oracle as O
from tests import _util as U

pytestmark = pytest.mark.gpu
TOL = 1e-4
# bf16 callers (C4): x, context and the results are rounded to bf16 (8 significant bits, relative 2^-9 = 2e-3 per value);
# z is a sum of an integral and an offset of similar size, log_jac sums d values.  Stated bound vs the fp32 oracle:
BF16_TOL = 2e-2


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _oracle_blocks(model, cfg):
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    blocks = []
    for i in range(cfg["nb_flow"]):
        mW, mb, mm = U._seq(sd, f"Flow{i}.net.made.net.", np.float32)
        iW, ib, _ = U._seq(sd, f"Flow{i}.net.parallel_nets.net.", np.float32)
        blocks.append(O.Block(mW, mb, mm, O.Net(iW, ib, O.LEAKY, O.ELU1), sd[f"Flow{i}.scaling"].astype(np.float32),
                              cfg.get("cond", 0)))
    return blocks


def _sample_rows(B, k=32, seed=5):
    rows = np.sort(np.random.RandomState(seed).choice(B, min(k, B), replace=False))
    rows[0], rows[-1] = 0, B - 1                       # first and last row: tile / shard edges
    return rows


@pytest.mark.parametrize("precision", ["f16x3", "bf16x3", "fp32"])
@pytest.mark.parametrize("workload", ["bsds300", "power", "toy", "vae", "mnist"])
def test_benchmarked_model_matches_oracle_at_full_batch(workload, precision, dev):
    import umnn_amd
    from umnn_amd import _lib
    cfg = dict(bench.WORKLOADS[workload])
    model = bench.build_model(cfg, dev)
    x, ctx = bench.make_inputs(cfg, cfg["rows"], dev, 1000)          # bench.py's own inputs for rank 0
    old = umnn_amd.get_forward_precision()
    if precision == "f16x3":
        assert old == "f16x3", "f16x3 is the library default: this case pins the arithmetic bench.py times"
    umnn_amd.set_precision(precision)
    try:
        launches = _lib.lib().umnn_launch_count()
        with torch.no_grad():
            ll, z = model.compute_ll(x, context=ctx) if ctx is not None else model.compute_ll(x)
            assert _lib.lib().umnn_launch_count() - launches == cfg["nb_flow"], \
                "compute_ll must be nb_flow x (conditioner + ONE quadrature launch)"
            zb, lj = model.compute_log_jac_bis(x, context=ctx) if ctx is not None else model.compute_log_jac_bis(x)
            llb, _ = model.compute_ll_bis(x, context=ctx) if ctx is not None else model.compute_ll_bis(x)
        assert umnn_amd.path_taken() == "hip"
        kname = _lib.lib().umnn_last_kernel_name().decode()
        if precision == "f16x3":
            assert kname.startswith("cc_fwd_f16<"), kname
            if workload == "toy":                      # 100-wide hidden layers: the eight-wave workgroups of round 5
                assert "T=7" in kname and "WAVES=8" in kname, kname
            if workload in ("bsds300", "power", "vae"):
                assert "LIVE=13" in kname and "PIPE" in kname, kname
            if workload == "mnist":
                assert "T1=7,TREST=4" in kname, kname
        else:
            assert ("bf16" in kname) == (precision != "fp32"), kname
    finally:
        umnn_amd.set_precision(old)
    rows = _sample_rows(cfg["rows"], 8 if cfg["d"] > 128 else 32)       # (d = 784: 8 rows keep the oracle to seconds)
    xs = x[rows].cpu().numpy()
    cs = ctx[rows].cpu().numpy() if ctx is not None else None
    blocks = _oracle_blocks(model, cfg)
    ll_ref, z_ref = O.flow_compute_ll(blocks, xs, cfg["n"], context=cs)
    _, lj_ref = O.flow_log_jac(blocks, xs, cfg["n"], context=cs)
    assert U.rel_err(z.cpu().numpy()[rows], z_ref) < TOL
    assert U.rel_err(ll.cpu().numpy()[rows], ll_ref) < TOL
    assert U.rel_err(zb.cpu().numpy()[rows], z_ref) < TOL
    got_lj = lj.cpu().numpy()[rows]
    assert np.all(np.abs(got_lj - lj_ref) <= TOL * np.maximum(1.0, np.abs(lj_ref)))
    # one-pass ll (in-kernel row sums) == elementwise route summed on the host, to fp32 summation noise
    ll_from_bis = llb.sum(1)
    assert U.rel_err(ll.cpu().numpy(), ll_from_bis.cpu().numpy()) < 2e-6 * max(1, cfg["d"] // 8)


def test_one_pass_ll_is_bit_reproducible_and_matches_elementwise_route(dev):
    """The in-kernel log-likelihood (arrival counters + last-arriver row sums, umnn_flow_ll_block_forward) against the
    elementwise kernels + ATen sums, over shapes whose rows straddle tiles, waves and workgroups in every way; repeated
    launches must return identical bits (one writer per row, fixed order) and leave the counters clean."""
    import umnn_amd
    from umnn_amd import integral as I
    for (B, d, nb_flow, hid, E, n) in [(1, 1, 1, [20, 20], 2, 10), (37, 3, 2, [50] * 4, 5, 20), (257, 63, 3, [50] * 4, 30, 20),
                                       (100, 784, 1, [50, 50], 4, 6), (4096, 2, 1, [100] * 4, 10, 50), (999, 17, 2, [40, 33], 6, 12),
                                       (5000, 6, 5, [50] * 4, 30, 20)]:
        torch.manual_seed(B + d)
        m = umnn_amd.UMNNMAFFlow(nb_flow=nb_flow, nb_in=d, hidden_derivative=hid, hidden_embedding=[64, 64], embedding_s=E,
                                 nb_steps=n, solver="CCParallel").to(dev).eval()
        with torch.no_grad():
            for net in m.nets:
                net.scaling.copy_(torch.linspace(-0.2, 0.3, d))
        x = torch.randn(B, d, device=dev)
        with torch.no_grad():
            ll, z = m.compute_ll(x)
            z2, lj = m.compute_log_jac_bis(x)
            ref = lj.double().sum(1) - 0.5 * (np.log(2 * np.pi) + z2.double() ** 2).sum(1)
            assert torch.equal(z, z2)
            assert float(((ll.double() - ref).abs() / ref.abs().clamp_min(1.0)).max()) < 3e-6, (B, d)
            for _ in range(3):
                ll2, _ = m.compute_ll(x)
                assert torch.equal(ll2, ll), (B, d)
        for t in I._row_counters.values():
            assert int(t.abs().sum()) == 0, "arrival counters must be zero again after every launch"


@pytest.mark.parametrize("mode", ["bf16_embedding", "bf16_everything", "autocast"])
def test_vae_prior_flow_bf16_storage(mode, dev):
    """BASELINE configuration C4 as the VAE calls it (models/vae_lib/models/flows.py:305-323: compute_log_jac_bis(z, h_context),
    d=64, cond_in=320, 4 blocks, n=50) with bf16 STORAGE around fp32 arithmetic -- the kernels load / store bf16 themselves
    (umnn_*_io entry points), nothing is converted on the way:
      bf16_embedding   fp32 x / context, the conditioner writes the [B, E*d] embedding in bf16 (set_embedding_dtype)
      bf16_everything  x, context and every result in bf16 as well
      autocast         torch.autocast(bfloat16) around the call, fp32 inputs
    Stated tolerance vs the fp32 oracle (there is no bf16 reference): every stored value carries a relative rounding of
    2^-9 = 2e-3; z = exp(s)(F + h_0) inherits it from h_0 and F, log_jac from f -- BF16_TOL = 2e-2 on max|d|/max(|ref|,1)."""
    import umnn_amd
    from umnn_amd import _lib
    cfg = dict(bench.WORKLOADS["vae"])
    model = bench.build_model(cfg, dev)
    x, ctx = bench.make_inputs(cfg, 100, dev, 7)                # the script's batch of 100
    blocks = _oracle_blocks(model, cfg)
    try:
        launches = _lib.lib().umnn_launch_count()
        with torch.no_grad():
            if mode == "bf16_embedding":
                model.set_embedding_dtype(torch.bfloat16)
                z, lj = model.compute_log_jac_bis(x, ctx)
                xs, cs = x, ctx
                assert z.dtype == torch.float32 and model.nets[0].net.m_embeding.dtype == torch.bfloat16
            elif mode == "bf16_everything":
                model.set_embedding_dtype(torch.bfloat16)
                xs, cs = x.bfloat16(), ctx.bfloat16()
                z, lj = model.compute_log_jac_bis(xs, cs)
                assert z.dtype == torch.bfloat16 and lj.dtype == torch.bfloat16
            else:
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    z, lj = model.compute_log_jac_bis(x, ctx)
                xs, cs = x, ctx
        assert umnn_amd.path_taken() == "hip" and _lib.lib().umnn_launch_count() - launches == cfg["nb_flow"]
    finally:
        model.set_embedding_dtype(None)
    z_ref, lj_ref = O.flow_log_jac(blocks, xs.float().cpu().numpy(), cfg["n"], context=cs.float().cpu().numpy())
    assert U.rel_err(z.float().cpu().numpy(), z_ref) < BF16_TOL
    assert np.all(np.abs(lj.float().cpu().numpy() - lj_ref) <= BF16_TOL * np.maximum(1.0, np.abs(lj_ref)))


def test_vae_prior_flow_bf16_embedding_training_gradients(dev):
    """Training through the bf16-embedding storage mode (forward AND backward kernels read h as bf16, d_h leaves as bf16):
    parameter gradients against the all-fp32 run of the same model, at the storage format's resolution."""
    cfg = dict(bench.WORKLOADS["vae"])
    model = bench.build_model(cfg, dev).train()
    x, ctx = bench.make_inputs(cfg, 100, dev, 9)
    grads = {}
    for mode in ("fp32", "bf16"):
        model.set_embedding_dtype(torch.bfloat16 if mode == "bf16" else None)
        model.zero_grad()
        z, lj = model.compute_log_jac_bis(x, ctx)
        (0.5 * (z ** 2).sum(1) - lj.sum(1)).mean().backward()          # the VAE's flow term: -log p(z_K) - log|det J|
        grads[mode] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    model.set_embedding_dtype(None)
    assert grads["fp32"].keys() == grads["bf16"].keys() and len(grads["bf16"]) > 20
    for k in grads["fp32"]:
        assert U.scaled_err(grads["bf16"][k].cpu().numpy(), grads["fp32"][k].cpu().numpy()) < 5e-2, k


def I_force_generic():
    from umnn_amd import integral as I
    return I.force_generic()


def test_block_level_compute_ll_and_bis_on_the_hip_path(dev):
    """UMNNMAF.compute_ll / compute_ll_bis (block level, with the in-place clamp of z, UMNNMAF.py:141-162) on the HIP path."""
    import umnn_amd
    G = U.load("g4_flow2_power_cc")
    blocks = U.blocks_from_g4(G)
    m = umnn_amd.UMNNMAFFlow(nb_flow=int(G["nb_flow"]), nb_in=int(G["d"]),
                             hidden_derivative=[int(v) for v in G["hidden_derivative"]],
                             hidden_embedding=[int(v) for v in G["hidden_embedding"]], embedding_s=int(G["E"]),
                             nb_steps=int(G["n"]), solver=str(G["solver"]))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in U.state_dict_of(G).items()})
    m.to(dev).eval()
    x = torch.from_numpy(G["x"]).to(dev) * 4                      # large inputs so that the clamp at +-10 bites
    xs = x.cpu().numpy()
    blk = blocks[0]
    z_ref, h = O.block_forward(blk, xs, int(G["n"]))
    lj_ref = O.block_log_jac(blk, xs, h)
    zc = np.clip(z_ref, -10., 10.)
    ll_ref = lj_ref.sum(1) - .5 * (np.log(2 * np.pi) + zc ** 2).sum(1)
    with torch.no_grad():
        ll, z = m.nets[0].compute_ll(x)
        assert umnn_amd.path_taken() == "hip"
        ljb, zb = m.nets[0].compute_ll_bis(x)
    assert float(z.abs().max()) <= 10.0
    assert U.rel_err(z.cpu().numpy(), zc) < TOL and U.rel_err(zb.cpu().numpy(), zc) < TOL
    assert U.rel_err(ll.cpu().numpy(), ll_ref) < TOL
    assert np.all(np.abs(ljb.cpu().numpy() - lj_ref) <= TOL * np.maximum(1.0, np.abs(lj_ref)))
    # block-level compute_log_jac (UMNNMAF.py:136-139): f(x;h) from a one-step launch, with and without a graph
    with torch.no_grad():
        lj_only = m.nets[0].compute_log_jac(x)
    assert umnn_amd.path_taken() == "hip"
    assert np.all(np.abs(lj_only.cpu().numpy() - lj_ref) <= TOL * np.maximum(1.0, np.abs(lj_ref)))
    xr = x.clone().requires_grad_(True)
    lj_g = m.nets[0].compute_log_jac(xr)
    lj_g.sum().backward()
    # (autograd on: the conditioner runs its fp32 chain, autograd off its bf16x3 GEMMs -- same path tolerance)
    assert torch.allclose(lj_g.detach(), lj_only, rtol=TOL, atol=TOL) and torch.isfinite(xr.grad).all()
    with I_force_generic():
        xa = x.clone().requires_grad_(True)
        m.nets[0].compute_log_jac(xa).sum().backward()
    assert U.scaled_err(xr.grad.cpu().numpy(), xa.grad.cpu().numpy()) < TOL
    # flow-level compute_ll_bis (per-dimension log-likelihood terms, UMNNMAFFlow.py:121-130)
    ll_full_ref, z_full_ref = O.flow_compute_ll(blocks, G["x"], int(G["n"]))
    with torch.no_grad():
        llb, zf = m.compute_ll_bis(torch.from_numpy(G["x"]).to(dev))
    assert U.rel_err(zf.cpu().numpy(), z_full_ref) < TOL
    assert U.rel_err(llb.sum(1).cpu().numpy(), ll_full_ref) < TOL


def test_bench_two_ranks_on_one_gpu_smoke(dev, tmp_path):
    """bench.py's N>1 launch path (torch.distributed.run, barrier + max-over-ranks timing, rank gathering, one JSON line on
    rank 0) on the single GPU of this box: two ranks share cuda:0 over gloo (RCCL refuses two ranks on one device; the
    8-GPU run uses the same code with backend nccl).  value must be about twice the per-rank rate of the same run."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UMNN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "power", "--rows", "2000"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["dist"]["backend"] == "gloo"
    assert {rk["rank"] for rk in line["ranks"]} == {0, 1}
    assert line["value"] > 0 and line["scaling"] == "weak" and "cpu_baseline" not in line
    for mode in ("train",):
        r = subprocess.run(cmd + ["--mode", mode], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["metric"] == "umnn_maf_training_samples_per_s"
