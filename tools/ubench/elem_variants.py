"""writes x and torch's exp / log / sigmoid of it for tools/ubench/elem_variants.hip"""
import sys, torch
n = 1 << 20
g = torch.Generator().manual_seed(5)
x = (torch.randn(n, generator=g) * 4).cuda()
d = sys.argv[1]
x.cpu().numpy().tofile(d + "/x.bin")
torch.exp(x).cpu().numpy().tofile(d + "/exp.bin")
torch.log(x.abs() + 1e-8).cpu().numpy().tofile(d + "/log.bin")
torch.sigmoid(x).cpu().numpy().tofile(d + "/sig.bin")
p = x.sigmoid()
neg = (1 - 0.25) * (p ** 2.0) * (-(1 - p + 1e-8).log())
pos = 0.25 * ((1 - p) ** 2.0) * (-(p + 1e-8).log())
(pos - neg).cpu().numpy().tofile(d + "/cc.bin")
(1 - p + 1e-8).cpu().numpy().tofile(d + "/neg.bin")
pos.cpu().numpy().tofile(d + "/pos.bin")
(-(1 - p + 1e-8).log()).cpu().numpy().tofile(d + "/nl.bin")
((1 - 0.25) * (p ** 2.0)).cpu().numpy().tofile(d + "/pp.bin")
print(n)
