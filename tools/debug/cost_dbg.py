import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from monosowa_amd.monodetr.matcher import HungarianMatcher
from monosowa_amd.pointwise import match_cost_blocks
gen = torch.Generator().manual_seed(77)
NL, B, Q, C, T, N = 3, 5, 137, 3, 23, 9
logits = (torch.randn(NL, B, Q, C, generator=gen) * 4).cuda()
boxes = torch.rand(NL, B, Q, 6, generator=gen).cuda()
labels = torch.randint(0, C, (T,), generator=gen).cuda()
tboxes = torch.rand(T, 6, generator=gen).cuda()
cols = torch.randint(0, T, (B, N), generator=gen).cuda()
for name, w in (("class", (1., 0., 0., 0.)), ("3d", (0., 1., 0., 0.)), ("bbox", (0., 0., 1., 0.)), ("giou", (0., 0., 0., 1.)), ("all", (2., 10., 5., 2.))):
    m = HungarianMatcher(cost_class=w[0], cost_3dcenter=w[1], cost_bbox=w[2], cost_giou=w[3])
    want = m.cost_blocks(logits, boxes, labels[cols], tboxes[cols])
    got = match_cost_blocks(logits, boxes, labels, tboxes, cols, *w)
    bad = ~((got == want) | (got.isnan() & want.isnan()))
    print(name, int(bad.sum()), "of", bad.numel(), float((got - want)[bad].abs().max()) if bad.any() else 0.0)
# elementary functions
x = (torch.randn(1 << 20, generator=gen) * 4).cuda()
p = x.sigmoid()
print("sigmoid vs 1/(1+exp(-x)):", int((p != 1.0 / (1.0 + torch.exp(-x))).sum()))
