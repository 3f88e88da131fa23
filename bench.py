#!/usr/bin/env python
"""Headline benchmark: UMNN-MAF log-density evals/s (BASELINE.json metric) on MI355X.

  python bench.py [--gpus N --steps K --warmup W --workload bsds300|power|toy|vae|mnist --mode eval|train]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step = one ``UMNNMAFFlow.compute_ll`` pass (MADE conditioner + fused HIP quadrature, all flow blocks) over one
synthetic batch already resident in HBM.  Default workload: BASELINE config C3 -- BSDS300-shaped (d=63), 8192 rows
per GPU (65536 rows sharded over 8 GPUs), n_steps=100, 5 blocks, MADE [512,512], E=30, integrand 31-50^4-1, fp32.
Weak scaling: the per-GPU shard is fixed; no collective on the forward path.  One JSON line on rank 0.  At N=1 the
line also carries ``full_batch_n1`` (the un-sharded 65536-row C3 batch on one GPU: the strong-scaling anchor of the
N=8 point), ``exact_fp32`` (same workload, reference arithmetic) and ``cpu_baseline`` (both reference solvers).
"""
import argparse
import hashlib
import json
import os
import sys
import time

# RCCL / device-tensor sharing between the ranks of a node goes through dmabuf IPC on this driver stack; the legacy mode fails with
# "hipIpcGetMemHandle: invalid argument".  The HSA runtime reads the variable when it initialises, so it has to be in the
# environment BEFORE torch touches the GPU: set here, ahead of `import torch` (an explicit setting wins).
_IPC_SET_BY = "environment" if "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ else "bench.py"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: nb_flow, d, hidden_embedding, hidden_derivative, E, n_steps, rows per GPU (, cond_in)
    "bsds300": dict(nb_flow=5, d=63, he=[512, 512], hd=[50] * 4, E=30, n=100, rows=8192, cond=0,
                    desc="C3 BSDS300-shaped UMNN-MAF compute_ll: d=63, 8192 rows/GPU (65536 over 8), n_steps=100"),
    "power": dict(nb_flow=5, d=6, he=[512, 512], hd=[50] * 4, E=30, n=100, rows=10000, cond=0,
                  desc="C2 POWER-shaped UMNN-MAF compute_ll: d=6, batch 10000, n_steps=100"),
    "toy": dict(nb_flow=1, d=2, he=[100] * 4, hd=[100] * 4, E=10, n=50, rows=4096, cond=0,
                desc="C1 2-moons UMNN-MAF compute_ll: d=2, batch 4096, n_steps=50"),
    "mnist": dict(nb_flow=5, d=784, he=[1024] * 3, hd=[100, 50, 50, 50, 50], E=30, n=50, rows=100, cond=0,
                  desc="MNISTExperiment-shaped UMNN-MAF (the d=784 shape BASELINE config 5 quotes): 5 blocks, MADE [1024]*3, "
                       "integrand 31-100-50-50-50-50-1, n_steps=50, the script's batch of 100 rows/GPU"),
    "vae": dict(nb_flow=4, d=64, he=[512, 512], hd=[50] * 4, E=30, n=50, rows=1024, cond=320,
                desc="C4 TrainVaeFlow prior flow: d=64 latent, cond_in=320, 4 blocks, n_steps=50, 1024 rows/GPU"),
}
FULL_BATCH_ROWS = 65536            # BASELINE config C3's un-sharded batch
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 2:1-sparse figure is never used)
# sources that define the dominant forward kernel: profiles/hbm_traffic.json is only trusted while their hash matches
# (the translation unit and the two headers that hold the kernel's code; cc_common.h / cc_bf16.h also carry helpers of the BACKWARD
# kernels -- hashing them made round 3's record go stale over an unrelated change)
TRAFFIC_SOURCES = ["umnn_amd/csrc/cc_forward_bf16.hip", "umnn_amd/csrc/cc_forward_f16.hip", "umnn_amd/csrc/cc_fwd_bf16_kernel.h",
                   "umnn_amd/csrc/cc_fwd_shared.h", "umnn_amd/csrc/cc_bf16.h", "umnn_amd/csrc/cc_forward.hip"]


def kernel_source_hash():
    h = hashlib.sha256()
    for rel in TRAFFIC_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_model(cfg, device, seed=0):
    import umnn_amd
    torch.manual_seed(seed)
    m = umnn_amd.UMNNMAFFlow(nb_flow=cfg["nb_flow"], nb_in=cfg["d"], hidden_derivative=cfg["hd"],
                             hidden_embedding=cfg["he"], embedding_s=cfg["E"], nb_steps=cfg["n"],
                             solver="CCParallel", cond_in=cfg.get("cond", 0))
    return m.to(device).eval()


def make_inputs(cfg, rows, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(rows, cfg["d"], generator=g).to(device)
    ctx = torch.randn(rows, cfg["cond"], generator=g).to(device) if cfg.get("cond", 0) else None
    return x, ctx


def cpu_baseline(cfg, model, budget_s=60.0):
    """Time the torch port of the reference's compute_ll with BOTH of its quadrature solvers -- the materialised
    ``ParallelNeuralIntegral`` (ParallelNeuralIntegral.py:37-65) and the node-by-node ``NeuralIntegral``
    (NeuralIntegral.py:37-66) -- on the host cores, on a bounded sample of the same workload (row chunks; the un-chunked
    node axis of the parallel solver would need terabytes).  Protocol: per solver a short thread-count sweep
    (torch's all-cores default is pathological on a 256-core host for these skinny GEMMs), then at the best count 3 warm-up
    calls and the median of 10 timed calls alternating between TWO different row chunks -- fewer, with ``budget_limited``
    set, when the solver's share of ``budget_s`` runs out -- and the all-cores setting reported beside it.
    ``value`` is the FASTER solver at its best thread count."""
    from oracle import torch_port as TP
    ncpu = os.cpu_count() or 1
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    blocks = TP.blocks_from_state_dict(sd, cfg["nb_flow"])
    chunk_full = 128 if cfg["d"] > 8 else 1024
    chunks_full = []
    for seed in (123, 456):
        g = torch.Generator().manual_seed(seed)
        chunks_full.append((torch.randn(chunk_full, cfg["d"], generator=g),
                            torch.randn(chunk_full, cfg["cond"], generator=g) if cfg.get("cond", 0) else None))
    solvers = {}
    old_threads = torch.get_num_threads()
    WARMUP, REPS = 3, 10
    with torch.no_grad():
        for name, solver, share in (("sequential", "CC", 0.42), ("parallel", "CCParallel", 0.58)):
            # (the materialised solver is ~2x slower per row: half the rows per call, so that 3 warm-up + 10 timed calls fit its share)
            chunk = chunk_full // 2 if (name == "parallel" and cfg["d"] > 8) else chunk_full
            chunks = [(xx[:chunk], cc[:chunk] if cc is not None else None) for xx, cc in chunks_full]

            def run(i=0):
                xx, cc = chunks[i % 2]
                return TP.flow_compute_ll(blocks, xx, cfg["n"], solver=solver, context=cc)

            def timed(i=0):
                t0 = time.perf_counter()
                run(i)
                return time.perf_counter() - t0
            t_start = time.perf_counter()
            deadline = t_start + share * budget_s
            # 1. thread sweep: one warm-up + one timed call per candidate (stops once a setting is clearly slower)
            best = (float("inf"), 1)
            sweep = {}
            for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
                torch.set_num_threads(th)
                run()
                dt = timed()
                sweep[th] = chunk / dt
                if dt < best[0]:
                    best = (dt, th)
                if dt > 1.5 * best[0] or time.perf_counter() - t_start > 0.3 * share * budget_s:
                    break
            one, threads = best
            # 2. the protocol at the best thread count
            torch.set_num_threads(threads)
            warm = 0
            while warm < WARMUP and (warm < 1 or time.perf_counter() + one < deadline - 4 * one):
                run(warm)
                warm += 1
            times = []
            while len(times) < REPS and (len(times) < 3 or time.perf_counter() + one < deadline - 2.5 * one):
                times.append(timed(len(times)))
            med = sorted(times)[len(times) // 2]
            # 3. all host cores (torch's default), beside it -- in a child process with a hard time limit: on a 256-core host
            # the all-cores setting can take minutes per call for these skinny GEMMs, and a torch call cannot be interrupted
            allc = None
            if threads != ncpu:
                allc = _all_cores_probe(sd, cfg, solver, chunks[0], ncpu, limit_s=8.0)
            solvers[name] = {"evals_per_s": chunk / med, "threads": threads, "chunk_rows": chunk, "chunks": 2,
                             "warmup": warm, "reps": len(times), "budget_limited": len(times) < REPS or warm < WARMUP,
                             "min_ms": 1e3 * min(times), "median_ms": 1e3 * med, "max_ms": 1e3 * max(times),
                             "thread_sweep_evals_per_s": sweep, "all_cores": allc,
                             "integrals_per_s": chunk * cfg["d"] * cfg["nb_flow"] / med,
                             "reference": "models/UMNN/ParallelNeuralIntegral.py:37-65" if name == "parallel"
                             else "models/UMNN/NeuralIntegral.py:37-66"}
    torch.set_num_threads(old_threads)
    fast = max(solvers, key=lambda k: solvers[k]["evals_per_s"])
    chunk = solvers[fast]["chunk_rows"]
    return {"value": solvers[fast]["evals_per_s"], "unit": "evals/s", "cores": solvers[fast]["threads"], "kind": "port",
            "solver": fast, "solvers": solvers, "host_cores": ncpu, "torch_parallel_info": torch.__config__.parallel_info().split("\n")[0:3],
            "sample": f"two {chunk}-row chunks of the same flow through oracle/torch_port.py (torch CPU port of the reference's "
                      f"compute_ll); per solver: thread sweep, {WARMUP} warm-up calls, median of {solvers[fast]['reps']} calls "
                      f"alternating between the chunks at the best thread count (budget_limited says when fewer fitted), "
                      f"all-cores setting beside it; value = the faster solver ({fast})"}


def _all_cores_probe(sd, cfg, solver, chunk, ncpu, limit_s):
    """One warm-up and one timed call of the CPU port at torch.set_num_threads(all cores), in a child process that is killed
    after ``limit_s`` seconds (then only an upper bound on the rate is known)."""
    import subprocess
    import tempfile
    xx, cc = chunk
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "w.pt")
        torch.save({"sd": sd, "x": xx, "c": cc}, path)
        code = ("import sys, time, torch; sys.path.insert(0, %r); from oracle import torch_port as TP; "
                "d = torch.load(%r); torch.set_num_threads(%d); b = TP.blocks_from_state_dict(d['sd'], %d); "
                "f = lambda: TP.flow_compute_ll(b, d['x'], %d, solver=%r, context=d['c']); "
                "torch.set_grad_enabled(False); f(); t = time.perf_counter(); f(); print('ALLCORES', time.perf_counter() - t)"
                % (ROOT, path, ncpu, cfg["nb_flow"], cfg["n"], solver))
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=limit_s)
            dt = float([l for l in r.stdout.splitlines() if l.startswith("ALLCORES")][-1].split()[1])
            return {"threads": ncpu, "evals_per_s": xx.shape[0] / dt, "reps": 1, "warmup": 1, "budget_limited": True}
        except subprocess.TimeoutExpired:
            return {"threads": ncpu, "timed_out_after_s": limit_s, "evals_per_s_upper_bound": xx.shape[0] / (limit_s / 2),
                    "note": "warm-up + one call at all cores did not finish inside the limit"}
        except Exception as e:          # a failed probe must not take the bench line down
            return {"threads": ncpu, "error": f"{type(e).__name__}: {e}"}


def hbm_traffic(workload, live, extra_args):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter per
    pass; gfx950 corrections per MI355X_MICROARCH.md).  The committed profiles/hbm_traffic.json is used while the kernel's
    sources hash to what was profiled; when they do not (or with ``live``) the two passes are collected NOW
    (tools/measure_traffic.py: two short child runs of this script under rocprofv3, ~1 min) -- a stale number is never
    reported, and a missing profiler leaves null with the reason."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    why = "--live-traffic"
    if not live:
        try:
            rec = json.load(open(path)).get(workload, {})
        except Exception:
            rec = {}
        if rec.get("source_sha256") == kernel_source_hash():
            return rec.get("hbm_bytes_per_launch"), f"profiles/hbm_traffic.json (kernel sources {rec.get('source_sha256')} = HEAD's)"
        why = "profiles/hbm_traffic.json is for other kernel sources" if rec else "no committed record for this workload"
    if os.environ.get("UMNN_BENCH_CHILD"):          # (a profiling child of this very function)
        return None, "profiling child run"
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:  # (one profiler per node: the in-run collection belongs to the N = 1 line)
        return None, f"{why}; not collected in a multi-rank run"
    import shutil
    if not shutil.which("rocprofv3"):
        return None, f"{why}; rocprofv3 not on PATH"
    try:
        from tools import measure_traffic
        rec = measure_traffic.measure(workload, extra_args)
        return rec.get("hbm_bytes_per_launch"), f"measured in this run ({why}): rocprofv3 PMC passes, kernel sources {kernel_source_hash()}"
    except Exception as e:      # profiler refused / timed out: report null, never a stale number
        return None, f"{why}; live collection failed: {e}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="bsds300", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the exact_fp32 and full_batch_n1 records (profiling runs)")
    ap.add_argument("--no-fused-adam", action="store_true", help="train mode: torch's foreach Adam instead of the fused one")
    ap.add_argument("--no-telemetry", action="store_true",
                    help="skip the telemetry pass (socket power / shader clock are sampled over a SEPARATE run of the same steps right "
                         "after the timed region, so the polling thread never shares the timed region with the launches)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank owns `rows` rows, the global batch grows with N.  strong: the workload's GLOBAL batch "
                         "(bsds300: BASELINE's 65536 rows) is split over the ranks by sharding.shard_bounds, so N = 1, 2, 4, 8 time the same problem")
    ap.add_argument("--live-traffic", action="store_true",
                    help="measure roofline.traffic now with two rocprofv3 PMC child runs instead of reading profiles/")
    ap.add_argument("--mode", default="eval", choices=["eval", "train"],
                    help="eval: compute_ll forward (the headline metric). train: forward + backward + one flattened "
                         "RCCL gradient all-reduce + Adam step per step (reported as training samples/s)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one captured hipGraph (umnn_amd.GraphedLL / GraphedTrainStep) -- for the "
                         "launch-bound small workloads; the eval roofline record then comes from an eager pass after the "
                         "timed region")
    ap.add_argument("--embedding", default="fp32", choices=["fp32", "bf16"],
                    help="storage of the [B, E*d] embedding between conditioner and quadrature kernels (bf16: configuration C4's "
                         "storage mode -- the kernels load bf16, arithmetic stays fp32; reported in config.embedding_storage)")
    ap.add_argument("--precision", default="", choices=["", "fp32", "bf16x3", "bf16x6", "f16x3"],
                    help="forward arithmetic (default: the library default, f16x3 -- two fp16 pieces, fp32-level, queued bf16x3 overflow fallback)")
    args = ap.parse_args()

    from umnn_amd import _lib, sharding
    import torch.distributed as dist
    rank, world, device = sharding.init_from_env(force_group=bool(os.environ.get("UMNN_FORCE_GROUP")))
    assert torch.cuda.is_available(), "bench.py measures the HIP path: it needs a GPU"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    cfg = dict(WORKLOADS[args.workload])
    if args.rows:
        cfg["rows"] = args.rows
    # strong scaling: the global batch is fixed (C3: BASELINE's 65536 rows; otherwise the workload's batch) and this rank owns
    # rows [lo, hi) of it; weak: `rows` per rank
    global_rows = (FULL_BATCH_ROWS if args.workload == "bsds300" and not args.rows else cfg["rows"]) if args.scaling == "strong" \
        else cfg["rows"] * world
    if args.scaling == "strong":
        lo, hi = sharding.shard_bounds(global_rows, rank, world)
        cfg["rows"] = hi - lo
    lib = _lib.lib()
    if args.precision:
        _lib.set_forward_precision(args.precision)
    precision = _lib.get_forward_precision()

    model = build_model(cfg, device)
    if args.embedding == "bf16":
        model.set_embedding_dtype(torch.bfloat16)
    x, ctx = make_inputs(cfg, cfg["rows"], device, 1000 + rank)    # every rank owns a different shard of the global batch

    def ll_of(xb, cb):
        return model.compute_ll(xb, context=cb) if cb is not None else model.compute_ll(xb)

    if args.mode == "train":
        model.train()
        sharding.broadcast_parameters(model)
        # (fused: one multi-tensor kernel per step instead of ~a dozen foreach passes over the parameters -- the MNIST-shaped
        # model has 134 M of them; same update rule, reference scripts: optim.Adam, UCIExperiments.py:118, MNISTExperiment.py:93)
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, capturable=bool(args.graph),
                               fused=not args.no_fused_adam)
        step = make_train_step(model, opt, x, ctx, world)
        eager_step = step
        graph_note = None
        if args.graph and world > 1 and dist.get_backend() == "nccl":
            # RCCL collectives cannot be captured in a hipGraph on this stack (umnn_amd/graphs.py): eager step instead
            graph_note = "--graph ignored: data-parallel training runs the eager step (RCCL collectives cannot be captured)"
            args.graph = False
        if args.graph:      # the whole step (fwd, HIP bwd, all-reduce hook, clipping, Adam) as one replayed hipGraph
            import umnn_amd
            gstep = umnn_amd.GraphedTrainStep(model, opt, x, context=ctx, clip_value=10.0,
                                              grad_hook=lambda mdl: sharding.allreduce_gradients(mdl, world))

            def step():         # noqa: F811
                return gstep().reshape(1), None
    else:
        def eager_step():
            with torch.no_grad():
                return ll_of(x, ctx)
        step = eager_step
        if args.graph:
            import umnn_amd
            graphed = umnn_amd.GraphedLL(model, x, context=ctx)
            step = lambda: graphed()        # noqa: E731  (x is already in the captured buffer)

    collective = collective_self_check(model, rank, world, device) if world > 1 else None

    for _ in range(args.warmup):
        step()
    lib.umnn_profile_enable(1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ll, _ = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    fwd = _lib.profile_read(_lib.PROF_FORWARD)
    bwd = _lib.profile_read(_lib.PROF_BACKWARD)
    fin = _lib.profile_read(_lib.PROF_FINISH)
    lib.umnn_profile_enable(0)
    if args.graph:      # launches inside a replayed graph carry no events: time them in an eager pass of the same step
        lib.umnn_profile_enable(1)
        for _ in range(max(3, args.steps // 4)):
            eager_step()
        torch.cuda.synchronize()
        fwd = _lib.profile_read(_lib.PROF_FORWARD)
        bwd = _lib.profile_read(_lib.PROF_BACKWARD)
        fin = _lib.profile_read(_lib.PROF_FINISH)
        lib.umnn_profile_enable(0)
    assert torch.isfinite(ll).all()
    kernel_name = lib.umnn_last_kernel_name_of(_lib.PROF_FORWARD if args.mode == "eval" else _lib.PROF_BACKWARD).decode()
    # socket power / shader clock of this workload (rank 0, eval): a 50-Hz polling thread over a SEPARATE repeat of the timed steps
    # (at least ~0.5 s of them), after the timed region -- the thread never competes with the launches that are timed.  Any failure
    # of the telemetry source leaves null fields, never takes the line down.
    telemetry = None
    if rank == 0 and world == 1 and args.mode == "eval" and not args.no_telemetry:
        try:
            from tools.telemetry import Sampler
            sampler = Sampler(device.index or 0)
            reps = max(args.steps, int(0.5 / max(elapsed / args.steps, 1e-6)))
            sampler.start()
            try:
                for _ in range(min(reps, 20000)):
                    step()
                torch.cuda.synchronize()
            finally:
                telemetry = sampler.stop()
            if telemetry is not None:
                telemetry["pass"] = f"separate pass of {min(reps, 20000)} steps after the timed region"
        except Exception as e:
            telemetry = {"error": f"{type(e).__name__}: {e}"}

    extras = world == 1 and args.mode == "eval" and not args.no_extras and args.scaling == "weak"

    def side_record(mode, peak, peak_name, conditioner):
        """The same workload, same steps / warm-up, under another arithmetic of the whole path (N = 1 eval only)."""
        import umnn_amd
        umnn_amd.set_precision(mode)
        try:
            for _ in range(args.warmup):
                eager_step()
            lib.umnn_profile_enable(1)
            torch.cuda.synchronize()
            te = time.perf_counter()
            for _ in range(args.steps):
                eager_step()
            torch.cuda.synchronize()
            te = time.perf_counter() - te
            e_ms, e_n, e_fl = _lib.profile_read(_lib.PROF_FORWARD)
            lib.umnn_profile_enable(0)
            kname = lib.umnn_last_kernel_name().decode()
        finally:
            umnn_amd.set_precision(precision)
        tf = e_fl / max(e_ms, 1e-9) / 1e9
        return {"value": cfg["rows"] * args.steps / te, "unit": "evals/s", "ms_per_step": 1e3 * te / args.steps, "steps": args.steps,
                "warmup": args.warmup, "kernel": kname, "avg_launch_ms": e_ms / max(1, e_n), "achieved": tf, "peak": peak,
                "peak_dtype": peak_name, "frac": tf / peak, "conditioner": conditioner}

    # for the record: the same workload in the reference's own arithmetic (exact fp32 products on the fp32 MFMA kernels, fp32
    # conditioner GEMMs) and in the fp32-accurate middle mode on the bf16 matrix cores (three bf16 pieces, six cross terms: ~4e-7 on F)
    exact = bf16x6 = f16x3 = bf16x3 = None
    if extras and precision != "fp32":
        exact = side_record("fp32", PEAK_FP32_MFMA_TFLOPS, "fp32 MFMA", "fp32 F.linear")
    if extras and precision in ("bf16x3", "f16x3"):
        bf16x6 = side_record("bf16x6", PEAK_BF16_MFMA_TFLOPS, "bf16 dense MFMA", "K-concatenated bf16 GEMMs (three products)")
        # ... and the other two-piece arithmetic: f16x3 (the default since round 5: fp32-level, fp16 pieces + queued bf16x3 overflow
        # fallback) when the run is bf16x3, bf16x3 (the default until round 4: ~6e-6 on F) when the run is f16x3
        if precision == "bf16x3":
            f16x3 = side_record("f16x3", PEAK_BF16_MFMA_TFLOPS, "fp16 dense MFMA (same rate as bf16)", "K-concatenated bf16 GEMMs (three products)")
        else:
            bf16x3 = side_record("bf16x3", PEAK_BF16_MFMA_TFLOPS, "bf16 dense MFMA", "K-concatenated bf16 GEMMs (three products)")
    # the un-sharded C3 batch on ONE GPU (65536 rows): the anchor the 8-GPU point of the sharded run is compared with
    full = None
    if extras and args.workload == "bsds300" and not args.rows and not args.graph:
        xf, cf = make_inputs(cfg, FULL_BATCH_ROWS, device, 4242)
        with torch.no_grad():
            for _ in range(args.warmup):
                ll_of(xf, cf)
            torch.cuda.synchronize()
            tfb = time.perf_counter()
            for _ in range(args.steps):
                llf, _ = ll_of(xf, cf)
            torch.cuda.synchronize()
            tfb = time.perf_counter() - tfb
        assert torch.isfinite(llf).all()
        full = {"rows": FULL_BATCH_ROWS, "value": FULL_BATCH_ROWS * args.steps / tfb, "unit": "evals/s",
                "ms_per_step": 1e3 * tfb / args.steps, "steps": args.steps, "warmup": args.warmup,
                "note": "BASELINE config C3's whole batch on one GPU (strong-scaling anchor for the N=8 run); parity of this "
                        "exact batch: tests/test_gpu_round3.py::test_full_65536_row_batch_compute_ll_matches_oracle_on_sampled_rows"}
        del xf, llf

    ranks = [{"rank": rank, "device": torch.cuda.get_device_name(device),
              "pci_bus_id": getattr(torch.cuda.get_device_properties(device), "pci_bus_id", None),
              "local_rank": int(os.environ.get("LOCAL_RANK", "0"))}]
    if world > 1:
        tmax = torch.tensor([elapsed], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, ranks[0])
        ranks = gathered

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        value = global_rows * args.steps / elapsed
        dom = fwd if args.mode == "eval" else bwd           # (ms, launches, algorithmic FLOPs) of the dominant kernel class
        avg_kernel_ms = dom[0] / max(1, dom[1])
        achieved = dom[2] / max(dom[0], 1e-9) / 1e9                    # TFLOP/s over those launches
        traffic, traffic_src = (None, "eval mode only") if args.mode != "eval" else \
            hbm_traffic(args.workload, args.live_traffic, ["--precision", precision] if args.precision else [])
        on_bf16 = "bf16" in kernel_name or "f16" in kernel_name
        peak = PEAK_BF16_MFMA_TFLOPS if on_bf16 else PEAK_FP32_MFMA_TFLOPS
        # FLOPs the matrix pipe actually executes per launch in the bf16-split FORWARD kernels: per hidden->hidden layer
        # ceil((H_out+1)/16) output tiles x ceil(ceil((H_in+1)/16)/2) K-steps x 3|6 cross terms of 16x16x32 MFMAs
        # (2*16*16*32 FLOPs each) per 16 integrals and node (+ the split-remainder MFMAs of the pipelined loop)
        executed = None
        if on_bf16 and args.mode == "eval":
            hd, terms = cfg["hd"], (3 if "PARTS=2" in kernel_name else 6)
            per_tile_node = sum(-(-(hd[i + 1] + 1) // 16) * -(-(-(-(hd[i] + 1) // 16)) // 2) * terms for i in range(len(hd) - 1))
            if "LIVE=13" in kernel_name and terms == 3:     # widths 48..51: the three terms share FIVE K-steps (merged layout)
                per_tile_node = sum(-(-(hd[i + 1] + 1) // 16) * 5 for i in range(len(hd) - 1))
            if "PIPE" in kernel_name and "cc_fwd_bf16" in kernel_name:     # + one remainder MFMA per fully live tile and split (the bf16
                # build only: since round 5 the fp16 build takes its remainders on the VALU, cc_fwd_bf16_kernel.h)
                per_tile_node += sum((-(-(hd[i] + 1) // 4)) // 4 for i in range(len(hd) - 1))
            tiles = -(-cfg["rows"] * cfg["d"] // 16)
            executed = per_tile_node * 16384.0 * tiles * (cfg["n"] + 1) / max(avg_kernel_ms, 1e-9) / 1e9
        dtype = {"fp32": "f32", "bf16x3": "f32 via bf16x3-split MFMA (fp32 accumulate)",
                 "bf16x6": "f32 via bf16x6-split MFMA (fp32 accumulate)",
                 "f16x3": "f32 via f16x3-split MFMA (fp32 accumulate)"}[precision] if on_bf16 or precision == "fp32" \
            else "f32"
        out = {
            "metric": "umnn_maf_log_density_evals_per_s" if args.mode == "eval" else "umnn_maf_training_samples_per_s",
            "value": value, "unit": "evals/s" if args.mode == "eval" else "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": cfg["desc"], "rows_per_gpu": cfg["rows"], "dim": cfg["d"], "n_steps": cfg["n"],
                       "nb_flow": cfg["nb_flow"], "embedding": cfg["E"], "integrand": cfg["hd"], "made": cfg["he"],
                       "cond_in": cfg.get("cond", 0),
                       "global_rows": global_rows,
                       "sharding": (f"weak: {cfg['rows']} rows on each of {world} rank(s), no forward collective" if args.scaling == "weak" else
                                    f"strong: the global batch of {global_rows} rows split over {world} rank(s) by sharding.shard_bounds "
                                    f"(rank 0 owns {cfg['rows']}), no forward collective"),
                       "embedding_storage": args.embedding,
                       "integrals_per_s": value * cfg["d"] * cfg["nb_flow"]},
            # achieved = ALGORITHMIC fp32 FLOPs (backward = 3 x forward) / kernel time; peak = dense MFMA peak
            # of the dtype the matrix instructions execute.  The bf16-split kernels issue 3 (or 6) bf16 MFMAs per fp32
            # product on tiles padded 50->64, so their algorithmic fraction of the bf16 peak is small by construction; the
            # fraction of the fp32-MFMA peak (what an exact-fp32 kernel could reach at best) is given alongside.
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name, "kernel_class": "forward" if args.mode == "eval" else "backward (main pass)",
                         "avg_launch_ms": avg_kernel_ms, "launches": dom[1],
                         "flops_per_launch": dom[2] / max(1, dom[1]),
                         "kernel_share_of_step": (avg_kernel_ms * cfg["nb_flow"] / ms_step) if args.graph
                         else dom[0] / (1e3 * elapsed),
                         "peak_dtype": "bf16 dense MFMA" if on_bf16 else "fp32 MFMA",
                         "frac_of_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "executed_mfma_tflops": executed,
                         "executed_frac_of_peak": executed / peak if executed else None,
                         # the clock this kernel ran at: the SMU lowers it under dense MFMA work (profiles/r03/power_fwd.json)
                         "sclk_mhz": telemetry.get("sclk_mhz") if telemetry else None,
                         "power_w": telemetry.get("power_w") if telemetry else None,
                         "power_cap_w": telemetry.get("power_cap_w") if telemetry else None,
                         "telemetry": telemetry},
            "ranks_seen": len(ranks), "ranks": ranks,
            "unique_pci_ids": len({r.get("pci_bus_id") for r in ranks}) == len(ranks) if all(r.get("pci_bus_id") is not None for r in ranks) else None,
            "collective_check": collective,
            "dist": {"world_size": dist.get_world_size() if dist.is_initialized() else 1,
                     "backend": dist.get_backend() if dist.is_initialized() else None},
            "env": {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "set_by": _IPC_SET_BY},
        }
        if args.graph:
            out["config"]["graph"] = "step replayed as one hipGraph; roofline timings from an eager pass after the timed region"
        if exact is not None:
            out["exact_fp32"] = exact
        if bf16x6 is not None:
            out["bf16x6"] = bf16x6
        if f16x3 is not None:
            out["f16x3"] = f16x3
        if bf16x3 is not None:
            out["bf16x3"] = bf16x3
        if full is not None:
            out["full_batch_n1"] = full
        if args.mode == "train":
            out["config"]["mode"] = "train: fwd + HIP bwd + flattened gradient all-reduce (RCCL) + value clipping + Adam"
            if graph_note:
                out["config"]["graph"] = graph_note
            out["train_kernels"] = {
                "forward": {"ms": fwd[0], "launches": fwd[1], "avg_launch_ms": fwd[0] / max(1, fwd[1])},
                "backward_main": {"ms": bwd[0], "launches": bwd[1], "avg_launch_ms": bwd[0] / max(1, bwd[1])},
                "backward_finishing": {"ms": fin[0], "launches": fin[1], "avg_ms": fin[0] / max(1, fin[1])},
                "share_of_step": (fwd[0] + bwd[0] + fin[0]) / (1e3 * elapsed)}
        if world == 1 and not args.no_cpu_baseline and args.mode == "eval":
            out["cpu_baseline"] = cpu_baseline(cfg, model)
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            out["cpu_baseline"]["gpu_over_cpu_parallel_solver"] = value / out["cpu_baseline"]["solvers"]["parallel"]["evals_per_s"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def collective_self_check(model, rank, world, device, reps=10):
    """First contact with a multi-GPU node validates itself (the driver's SCALE run is the first place RCCL with N > 1 executes):
    before anything is timed, an all-reduce of rank + 1 (checks the sum every rank must see) and of a buffer the size of the
    flattened gradient -- the ONE collective of the training step (sharding.allreduce_gradients) -- timed over `reps` calls.
    -> the record bench.py prints as "collective_check"; raises on a wrong sum."""
    import torch.distributed as dist
    on_gpu = dist.get_backend() == "nccl"
    cdev = device if on_gpu else torch.device("cpu")
    one = torch.full((1,), float(rank + 1), device=cdev, dtype=torch.float64)
    dist.all_reduce(one)
    nparam = sum(p.numel() for p in model.parameters() if p.requires_grad)
    flat = torch.full((nparam,), float(rank + 1), device=cdev, dtype=torch.float32)
    dist.all_reduce(flat)
    if on_gpu:
        torch.cuda.synchronize()
    tc = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(flat)
    if on_gpu:
        torch.cuda.synchronize()
    tc = (time.perf_counter() - tc) / reps
    want = world * (world + 1) / 2
    scale = float(world) ** reps                          # (the timed calls summed the already-reduced buffer `reps` more times)
    ok = float(one.item()) == want and float(flat[0].item()) == want * scale and float(flat[-1].item()) == want * scale
    rec = {"allreduce_ok": bool(ok), "allreduce_us": 1e6 * tc, "allreduce_bytes": 4 * nparam, "backend": dist.get_backend(),
           "algbw_GBps": 4 * nparam / tc / 1e9}
    if not ok:
        raise RuntimeError(f"all-reduce self-check failed on rank {rank}: got {float(one.item())}, want {want}")
    return rec


def make_train_step(model, opt, x, ctx, world, clip_value=10.0):
    """One data-parallel optimisation step as the reference's scripts do it (UCIExperiments.py:133-146): loss, backward,
    ONE flattened gradient all-reduce, value clipping AFTER the reduction (:143), Adam.  The reference is single-process: ONE mean
    over the whole batch.  With shards of unequal size (sharding.shard_bounds hands them out when world does not divide the rows)
    the average of per-rank means is a different gradient, so every rank normalises its summed log-likelihood by the GLOBAL row
    count (one scalar all-reduce when the step is built) and the gradient all-reduce is a plain SUM."""
    from umnn_amd import sharding
    import torch.distributed as dist
    global_rows = x.shape[0]
    if world > 1:
        cnt = torch.tensor([float(x.shape[0])], dtype=torch.float64,
                           device=x.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        global_rows = int(cnt.item())

    def step():
        opt.zero_grad(set_to_none=True)
        ll, z = model.compute_ll(x, context=ctx) if ctx is not None else model.compute_ll(x)
        if world > 1:
            (-(ll.sum() / global_rows)).backward()
            sharding.allreduce_gradients(model, world, average=False)          # one flattened all-reduce (SUM), before clipping
        else:
            (-ll.mean()).backward()
        torch.nn.utils.clip_grad_value_(model.parameters(), clip_value)
        opt.step()
        return ll.detach(), z
    return step


if __name__ == "__main__":
    main()
