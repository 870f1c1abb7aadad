import sys; sys.path.insert(0, '.')
import torch
from monosowa_amd.pointwise import linear_wgrad
gy = torch.randn(8800, 256, device="cuda"); x = torch.randn(8800, 256, device="cuda")
for _ in range(5):
    linear_wgrad(gy, x); torch.cuda.synchronize()
