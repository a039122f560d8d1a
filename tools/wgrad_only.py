import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sst_amd.dense import weight_bias_grad
m = 90107
for out, inn in ((256, 128), (128, 128), (128, 256)):
    dy = torch.randn(m, out, device='cuda'); x = torch.randn(m, inn, device='cuda')
    for _ in range(5):
        weight_bias_grad(dy, x, True)
torch.cuda.synchronize()
