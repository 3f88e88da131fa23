"""Which hipGraph captures survive an initialised one-rank nccl process group?  (run: PROBE=t1|t2|t3|t4 python this)"""
import faulthandler, os, sys
faulthandler.enable()
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29654")
import umnn_amd
from umnn_amd import sharding
probe = os.environ.get("PROBE", "t1")
def mark(m): print("MARK", probe, m, flush=True)
if probe != "t0":
    rank, world, dev = sharding.init_from_env(backend="nccl", force_group=True)
    t = torch.ones(8, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
else:
    dev = torch.device("cuda:0")
mark("group")
mode = os.environ.get("CAPMODE", "thread_local")
torch.manual_seed(0)
model = umnn_amd.UMNNMAFFlow(nb_flow=2, nb_in=6, hidden_derivative=[50] * 4, hidden_embedding=[64, 64], embedding_s=30,
                             nb_steps=20, solver="CCParallel").to(dev)
x = torch.randn(100, 6, device=dev)
g = torch.cuda.CUDAGraph()
if probe in ("t0", "t1"):
    y = torch.zeros(8, device=dev)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        z = y + 1
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g, capture_error_mode=mode):
        z = y + 1
    mark("captured trivial")
elif probe == "t2":
    model.eval()
    gl = umnn_amd.GraphedLL(model, x)
    mark("captured GraphedLL"); gl(); torch.cuda.synchronize(); mark("replayed")
elif probe == "t3":
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True)
    st = umnn_amd.GraphedTrainStep(model, opt, x, clip_value=10.0)
    mark("captured train step without hook"); st(); torch.cuda.synchronize(); mark("replayed")
elif probe == "t4":
    dist.destroy_process_group()
    mark("group destroyed")
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True)
    st = umnn_amd.GraphedTrainStep(model, opt, x, clip_value=10.0)
    mark("captured after destroy")
if probe not in ("t5", "t6", "t7", "t8", "t9", "t10"): print("DONE", probe)
if probe == "t5":       # the split step by hand, marks between the pieces
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True)
    params = [p for p in model.parameters()]
    hook = lambda m: sharding.allreduce_gradients(m, 1, force=True)
    def fwd_bwd():
        opt.zero_grad(set_to_none=True); ll, _ = model.compute_ll(x); loss = -ll.mean(); loss.backward(); return loss.detach()
    def finish():
        torch.nn.utils.clip_grad_value_(params, 10.0); opt.step()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fwd_bwd(); hook(model); finish()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize(); mark("warm")
    gA = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gA, capture_error_mode=mode):
        loss = fwd_bwd()
    mark("graph A captured")
    static = [p.grad for p in params]
    hook(model); torch.cuda.synchronize(); mark("hook after A")
    src, dst = [], []
    for p, s_ in zip(params, static):
        if s_ is None or p.grad is s_: continue
        src.append(p.grad); dst.append(s_); p.grad = s_
    torch._foreach_copy_(dst, src); torch.cuda.synchronize(); mark("restored")
    gB = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gB, pool=gA.pool(), capture_error_mode=mode):
        finish()
    mark("graph B captured")
    def restore():
        src, dst = [], []
        for p, s_ in zip(params, static):
            if s_ is None or p.grad is s_: continue
            src.append(p.grad); dst.append(s_); p.grad = s_
        if src: torch._foreach_copy_(dst, src)
    for it in range(4):
        gA.replay(); torch.cuda.synchronize()
        fin_a = all(bool(torch.isfinite(s_).all()) for s_ in static if s_ is not None)
        hook(model); torch.cuda.synchronize()
        fin_h = all(bool(torch.isfinite(p.grad).all()) for p in params if p.grad is not None)
        restore(); gB.replay(); torch.cuda.synchronize()
        fin_p = all(bool(torch.isfinite(p).all()) for p in params)
        mark("iter %d loss %f grads-after-A finite %s after-hook %s params-after-B %s" % (it, float(loss), fin_a, fin_h, fin_p))
    print("DONE t5")
if probe in ("t6", "t7"):       # GraphedTrainStep in split mode, several replays (t7: synchronize between calls)
    model.train()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, capturable=True)
    st = umnn_amd.GraphedTrainStep(model, opt, x, clip_value=10.0, grad_hook=lambda m: sharding.allreduce_gradients(m, 1, force=True))
    assert st.split
    for it in range(4):
        loss = st()
        if probe == "t7": torch.cuda.synchronize()
        mark("iter %d loss %f params finite %s" % (it, float(loss), all(bool(torch.isfinite(p).all()) for p in model.parameters())))
    print("DONE", probe)
if probe == "t8":
    model.train()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, capturable=True)
    st = umnn_amd.GraphedTrainStep(model, opt, x, clip_value=10.0, grad_hook=lambda m: sharding.allreduce_gradients(m, 1, force=True))
    fin = lambda ts: all(bool(torch.isfinite(t_).all()) for t_ in ts if t_ is not None)
    mark("after init: params finite %s" % fin(model.parameters()))
    mark("adam state finite %s; steps %s" % (fin([v for s_ in opt.state.values() for v in s_.values() if torch.is_tensor(v)]),
                                             [float(s_["step"]) for s_ in list(opt.state.values())[:2]]))
    st.graph.replay(); torch.cuda.synchronize()
    mark("A: static grads finite %s, loss %f" % (fin(st._static), float(st.loss)))
    mark("p.grad is static: %s" % all((p.grad is s_) for p, s_ in zip(st.params, st._static) if s_ is not None))
    st.grad_hook(model); torch.cuda.synchronize()
    mark("hook: p.grad finite %s" % fin([p.grad for p in st.params]))
    st._restore(); torch.cuda.synchronize()
    mark("restore: static finite %s" % fin(st._static))
    st.graph_b.replay(); torch.cuda.synchronize()
    mark("B: params finite %s; adam state finite %s" % (fin(model.parameters()), fin([v for s_ in opt.state.values() for v in s_.values() if torch.is_tensor(v)])))
    print("DONE t8")
if probe == "t9":
    model.train()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, capturable=True)
    st = umnn_amd.GraphedTrainStep(model, opt, x, clip_value=10.0, grad_hook=lambda m: sharding.allreduce_gradients(m, 1, force=True))
    pts = os.environ.get("SYNCPTS", "")
    fin = lambda ts: all(bool(torch.isfinite(t_).all()) for t_ in ts if t_ is not None)
    for it in range(3):
        st.graph.replay()
        if "a" in pts: torch.cuda.synchronize()
        st.grad_hook(model)
        if "h" in pts: torch.cuda.synchronize()
        st._restore()
        if "r" in pts: torch.cuda.synchronize()
        st.graph_b.replay()
        if "b" in pts: torch.cuda.synchronize()
    torch.cuda.synchronize()
    mark("SYNCPTS=%r -> params finite %s loss %f" % (pts, fin(model.parameters()), float(st.loss)))
    print("DONE t9")
if probe == "t10":      # eager reference trajectory (same model, same data, same hook): losses of steps 1..7
    model.train()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, capturable=True)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for it in range(7):
            opt.zero_grad(set_to_none=True); ll, _ = model.compute_ll(x); loss = -ll.mean(); loss.backward()
            sharding.allreduce_gradients(model, 1, force=True)
            torch.nn.utils.clip_grad_value_([p for p in model.parameters()], 10.0); opt.step()
            mark("eager step %d loss %f" % (it, float(loss)))
    print("DONE t10")
