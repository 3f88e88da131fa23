import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time, umnn_amd, sys
from umnn_amd import _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = umnn_amd.UMNNMAFFlow(nb_flow=1, nb_in=2, hidden_derivative=[100]*4, hidden_embedding=[100]*4, embedding_s=10, nb_steps=50, solver="CCParallel").to(dev).eval()
z = torch.randn(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 2, device=dev)
with torch.no_grad():
    for _ in range(3): x = m.invert(z, iter=10)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): x = m.invert(z, iter=10)
    torch.cuda.synchronize(); t = (time.perf_counter() - t) / 10
print("invert", z.shape[0], "rows:", round(t * 1e3, 3), "ms", _lib.lib().umnn_last_kernel_name().decode())
