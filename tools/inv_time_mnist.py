import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, umnn_amd
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = umnn_amd.UMNNMAFFlow(nb_flow=1, nb_in=784, hidden_derivative=[100, 50, 50, 50, 50], hidden_embedding=[1024] * 3, embedding_s=30,
                         nb_steps=50, solver="CCParallel").to(dev).eval()
z = torch.randn(100, 784, device=dev) * 0.3
with torch.no_grad():
    x = m.invert(z, iter=10)
    torch.cuda.synchronize(); t = time.perf_counter()
    x = m.invert(z, iter=10)
    torch.cuda.synchronize(); t = time.perf_counter() - t
print("invert 100 x 784, one block, rows =", os.environ.get("UMNN_INVERT_ROWS", "1"), ":", round(t * 1e3, 1), "ms")
