"""hipGraph capture of a launch-bound inference step (small flows: the 2-D toy configuration is ~17 short launches).

``GraphedLL(model, x_example)`` records ``model.compute_ll`` once on static buffers (``torch.cuda.CUDAGraph``; the
quadrature launches of libumnn_cc go to torch's current stream, so they are captured like any ATen kernel) and replays
it per call.  Inference only (autograd off); shapes are fixed at capture.  For the large configurations the step is one
long kernel per block and a graph buys nothing."""
import torch


def _capture_mode():
    """``capture_error_mode`` for torch.cuda.graph: with an initialised process group the backend's watchdog thread polls
    events while we capture; under the default "global" mode that is an illegal call during capture (on this stack the process
    dies with SIGSEGV, measured with a one-rank nccl group) -- "thread_local" confines the capture rules to this thread."""
    import torch.distributed as dist
    return "thread_local" if dist.is_available() and dist.is_initialized() else "global"


def _prime_for_capture(model, x):
    """What must exist BEFORE a capture whatever ``warmup`` is: the library probe of the conditioner's GEMM route (run for the first
    time inside a capture, hipBLASLt would initialise under capture and abort), the quadrature tables of every block (their host
    -> device upload is not capturable) and ConditionnalMADE's kept-row indices (a pageable host -> device copy).  Warm-up runs
    create all of them; ``warmup=0`` in a fresh process does not."""
    from . import made, quadrature
    with torch.no_grad():
        made._fast_path_ok(x if x.dtype == torch.float32 else x.float())
    for mod in model.modules():
        n = getattr(mod, "nb_steps", None)
        if isinstance(n, int) and n >= 1:
            quadrature.device_tables(n, x.device)
        if isinstance(mod, made.ConditionnalMADE):
            mod._kept_rows(x.device)


class GraphedLL:
    """``GraphedLL(model, x_example)(x)`` -> the captured ``(ll, z)`` tensors (overwritten by the next call).

    The conditioner's cached masked / packed weights are baked into the graph as constants, so the capture remembers the
    version counters of every parameter and buffer and RE-CAPTURES when one moved (an optimizer step, load_state_dict,
    force_lipschitz between two replays).  Writes that bypass versioning (``p.data``...) need ``refresh()``."""

    def __init__(self, model, x_example, context=None, warmup=3):
        assert x_example.is_cuda, "hipGraph capture needs device tensors"
        self.model = model
        self.x = x_example.clone()
        self.context = context.clone() if context is not None else None
        self.warmup = warmup
        self.captures = 0
        self._capture()

    def _versions(self):
        return tuple(t._version for t in self.model.parameters()) + tuple(t._version for t in self.model.buffers())

    def _run(self):
        return self.model.compute_ll(self.x, self.context) if self.context is not None else self.model.compute_ll(self.x)

    def _capture(self):
        from . import made
        with torch.no_grad():
            side = torch.cuda.Stream(device=self.x.device)
            side.wait_stream(torch.cuda.current_stream(self.x.device))
            with torch.cuda.stream(side):               # warm every cache (LDS caps, tables, packed weights) first
                # At least ONE eager run, whatever ``warmup`` says: the capture below may bake the conditioner's cached masked /
                # packed weights in (capture_may_cache), and those caches must have been filled by kernels that really RAN -- a
                # cache entry first produced inside the capture would hold nothing until the first replay, yet be found under a
                # matching key by any eager compute_ll (or a second capture) before it (made.MaskedLinear._may_store
                # now also refuses to store from inside any capture).
                for _ in range(max(1, self.warmup)):
                    self._run()
            torch.cuda.current_stream(self.x.device).wait_stream(side)
            _prime_for_capture(self.model, self.x)
            self.graph = torch.cuda.CUDAGraph()
            with made.capture_may_cache(), torch.cuda.graph(self.graph, capture_error_mode=_capture_mode()):
                self.out = self._run()
        self._seen = self._versions()
        self.captures += 1

    def refresh(self):
        """Re-capture unconditionally (after weight writes that do not bump tensor versions)."""
        from .made import invalidate_caches
        invalidate_caches(self.model)
        self._capture()

    def __call__(self, x=None, context=None):
        """Replay on new data (copied into the captured buffers); returns the captured output tensors (overwritten by
        the next call)."""
        if self._versions() != self._seen:
            self._capture()
        if x is not None:
            self.x.copy_(x)
        if context is not None:
            self.context.copy_(context)
        self.graph.replay()
        return self.out


class GraphedTrainStep:
    """One optimisation step (``-compute_ll(x).mean()`` -> backward -> optional gradient hook / value clipping ->
    ``optimizer.step()``) recorded as a single hipGraph and replayed per call.  For the launch-bound regime the
    reference's own scripts train in (100 samples per step: ~280 short launches, the CPU cannot issue them as fast as
    the GPU finishes them).  The optimizer must be capturable (e.g. ``torch.optim.Adam(..., capturable=True)``);
    shapes are fixed at capture; as usual for whole-step capture, the warm-up iterations already update the model.

        step = umnn_amd.GraphedTrainStep(model, opt, x_example)
        for x in loader: loss = step(x)          # loss: 0-dim device tensor, overwritten by the next call

    ``grad_hook(model)`` runs between backward and clipping, inside the graph.  It must not issue RCCL collectives: with an
    initialised ``nccl`` process group this class refuses a hook (NotImplementedError) -- first contact with RCCL on this
    stack (ROCm 7.2 / torch 2.10, one-rank group, tools/_rccl_capture_probe.py) showed that (i) capturing ``all_reduce``
    in a hipGraph segfaults, and (ii) the obvious workaround -- graph A (forward + backward) -> eager all-reduce -> graph B
    (clipping + optimizer) -- trains to different weights depending on where the host synchronises between the pieces.
    Data-parallel training therefore runs the eager step (``bench.py --mode train`` does so by itself when ``--graph`` is
    asked for with more than one rank).

    PyTorch constraint worth knowing (it cost a segfault at ``capture_end`` here): autograd binds a parameter's AccumulateGrad
    node to the stream that was current when the node was created.  If an autograd graph built on the DEFAULT stream over
    these parameters is still alive (e.g. the loss tensor of an earlier eager step kept in a variable), the captured backward
    accumulates on the legacy stream, which a capture cannot wait on.  The constructor drops the graphs the model itself
    holds (the blocks' cached ``m_embeding``); references held by the caller (a kept ``ll`` / ``z`` of an earlier eager step)
    must be dropped by the caller, or the step built before any eager training step."""

    def __init__(self, model, optimizer, x_example, context=None, warmup=3, clip_value=None, grad_hook=None):
        assert x_example.is_cuda, "hipGraph capture needs device tensors"
        import torch.distributed as dist
        if grad_hook is not None and dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
            raise NotImplementedError(
                "GraphedTrainStep: a gradient hook under an initialised nccl process group would put RCCL collectives inside "
                "a hipGraph capture, which segfaults on this stack; run the eager optimisation step for data-parallel training")
        self.x = x_example.clone()
        self.context = context.clone() if context is not None else None
        params = [p for g in optimizer.param_groups for p in g["params"]]
        # The blocks cache their last embedding (``m_embeding``, reference attribute) -- with its autograd graph when it was
        # computed in training mode.  That graph keeps the parameters' AccumulateGrad nodes alive, bound to whatever stream
        # the first training forward ran on; drop it so that the warm-up below re-creates them on the side stream (see the
        # class docstring: on the default stream they make capture_end crash).
        for mod in model.modules():
            if hasattr(mod, "m_embeding"):
                mod.m_embeding = None

        def one_step():
            optimizer.zero_grad(set_to_none=True)
            ll, _ = model.compute_ll(self.x, self.context) if self.context is not None else model.compute_ll(self.x)
            loss = -ll.mean()
            loss.backward()
            if grad_hook is not None:
                grad_hook(model)
            if clip_value is not None:
                torch.nn.utils.clip_grad_value_(params, clip_value)
            optimizer.step()
            return loss.detach()

        side = torch.cuda.Stream(device=x_example.device)
        side.wait_stream(torch.cuda.current_stream(x_example.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                one_step()
        torch.cuda.current_stream(x_example.device).wait_stream(side)
        _prime_for_capture(model, x_example)
        self.graph = torch.cuda.CUDAGraph()
        # (captured on the warm-up's stream: the parameters' AccumulateGrad nodes are bound to it -- on another capture stream
        # autograd warns about the mismatch and synchronises through the legacy stream)
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode=_capture_mode()):
            self.loss = one_step()

    def __call__(self, x=None, context=None):
        if x is not None:
            self.x.copy_(x)
        if context is not None:
            self.context.copy_(context)
        self.graph.replay()
        return self.loss
