import os, sys, torch
sys.path.insert(0, os.getcwd())
import umnn_amd
from umnn_amd import _lib, integral as I
from umnn_amd.nets import mlp_spec
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = umnn_amd.IntegrandNetwork(63, 31, [50]*4, 1).to(dev)
spec = mlp_spec(net)
B=300
x, h, g = torch.randn(B, 63, device=dev), torch.randn(B, 30*63, device=dev), torch.randn(B, 63, device=dev)
gf = torch.randn(B, 63, device=dev)
o = I.hip_backward(spec, None, x, h, g, gf, 20)
torch.save([t.cpu() for t in o[1:]], sys.argv[1])
