import torch
g = torch.Generator().manual_seed(5)
x = (torch.randn(1 << 20, generator=g) * 4).cuda()
p = x.sigmoid()
print("p**2.0 != p*p:", int((p ** 2.0 != p * p).sum()))
print("0.75*(p**2) vs 0.75*(p*p):", int(((1 - 0.25) * (p ** 2.0) != 0.75 * (p * p)).sum()))
a = -(1 - p + 1e-8).log()
b = -torch.log((1.0 - p) + 1e-8)
print("neglog forms:", int((a != b).sum()))
neg = (1 - 0.25) * (p ** 2.0) * a
neg2 = (0.75 * (p * p)) * b
print("neg forms:", int((neg != neg2).sum()))
# one fused expression per element on the CPU in float32 for comparison of contraction effects
pc = p.cpu()
import numpy as np
pn = pc.numpy()
negn = (np.float32(0.75) * (pn * pn)) * (-np.log((np.float32(1) - pn) + np.float32(1e-8)))
print("gpu vs numpy f32 neg:", int((neg.cpu().numpy() != negn).sum()))
