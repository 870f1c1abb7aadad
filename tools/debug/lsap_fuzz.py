import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from monosowa_amd import pointwise, lsap
rng = np.random.default_rng(12345)
bad = 0
for it in range(300):
    NL = int(rng.integers(1, 4)); B = int(rng.integers(1, 9)); G = int(rng.choice([1, 2, 5, 11])); gq = int(rng.integers(1, 65)); Q = gq * G
    sizes = [int(x) for x in rng.integers(0, 65, B)]
    if max(sizes) == 0: sizes[0] = 1
    N = max(sizes)
    kind = it % 4
    if kind == 0: c = rng.standard_normal((NL, B, Q, N)).astype(np.float32)
    elif kind == 1: c = rng.integers(0, 3, (NL, B, Q, N)).astype(np.float32)
    elif kind == 2: c = (rng.integers(0, 50, (NL, B, Q, N)) / 7).astype(np.float32)
    else: c = np.round(rng.standard_normal((NL, B, Q, N)), 1).astype(np.float32)
    blocks = torch.from_numpy(c).cuda()
    if not pointwise.device_lsap_supported(blocks, sizes, G): continue
    st = torch.zeros((), dtype=torch.int32, device="cuda")
    got = pointwise.device_lsap_match_flat(blocks, sizes, G, st).cpu().numpy()
    want = lsap.match_flat(c, np.asarray(sizes, np.int64), G, padded=True)
    if int(st.item()) != 0 or not np.array_equal(got, want):
        bad += 1; print("MISMATCH", it, NL, B, Q, G, sizes)
print("fuzz done, mismatches:", bad)
