"""host time of the native assignment solve (528 problems of the train step) by thread count"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from monosowa_amd import lsap
rng = np.random.default_rng(0)
NL, B, Q, N = 3, 16, 550, 10
cost = rng.standard_normal((NL, B, Q, N)).astype(np.float32)
sizes = np.asarray(rng.integers(3, N + 1, B), np.int64)
for th in (1, 2, 4, 8, 16):
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); lsap.match_flat(cost, sizes, 11, padded=True, n_threads=th); ts.append(time.perf_counter() - t0)
    print("threads %2d: median %.3f ms" % (th, 1e3 * sorted(ts)[len(ts) // 2]))
