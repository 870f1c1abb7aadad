"""CPU model of the cell scatter's candidate scan (msda_scatter_plan.h / msda_plan.hip): how many candidates / point tests one
(batch, head) plane scans and how many deliver, per level, for the per-(head, level) bounds the plan uses and for
per-(head, level, point) bounds.  numpy only; no GPU.

    python tools/debug/scan_sim.py [kitti|c5] [sigma_px]        # the module's initial pattern + N(0, sigma) drift
    (tools/debug/offset_stats.py feeds it the sampling locations of a real train step)"""
import math
import sys
import numpy as np

M, P, L = 8, 4, 4
CELLS = 128
SIGMAS = 3.0
REACH = 8


def tiling(Hl, Wl):
    th, tw = Hl, Wl
    if (Hl + 1) * (Wl + 1) > CELLS:
        best = None
        for nty in range(1, Hl + 1):
            h = -(-Hl // nty)
            wmax = CELLS // (h + 1) - 1
            if wmax < 1:
                continue
            ntx = -(-Wl // wmax)
            w = -(-Wl // ntx)
            cost = nty * ntx * (h + 8) * (w + 8)
            if best is None or cost < best:
                best, th, tw = cost, h, w
    ntx = -(-Wl // tw); tw = -(-Wl // ntx); nty = -(-Hl // th); th = -(-Hl // nty)
    return th, tw, nty, ntx


def cfloor(c, Nq, N):
    return np.floor_divide((2 * c + 1) * N - Nq, 2 * Nq)


def plan_bounds(d):
    """msda_plan.hip: [min, max] of d cut at mean +- (3 sigma + 0.5), clamped to the reach."""
    if d.size == 0:
        return 0, 0
    mu, sd = d.mean(), d.std()
    lo = max(d.min(), math.floor(mu - (SIGMAS * sd + 0.5)))
    hi = min(d.max(), math.ceil(mu + (SIGMAS * sd + 0.5)))
    return int(min(max(lo, -REACH), REACH)), int(max(min(hi, REACH), -REACH))


def analyse(H, W, cells_of):
    """cells_of(m, l) -> (lq, yq, xq, p, cy, cx) int arrays over the VALID points of head m at level l (one batch element)."""
    tot = np.zeros((L, 7))
    for m in range(M):
        for l in range(L):
            th, tw, nty, ntx = tiling(H[l], W[l])
            lqa, yq, xq, pa, cy, cx = cells_of(m, l)
            cfy = np.empty_like(cy); cfx = np.empty_like(cx)
            for lq in range(L):
                s = lqa == lq
                cfy[s] = cfloor(yq[s], H[lq], H[l]); cfx[s] = cfloor(xq[s], W[lq], W[l])
            dy, dx = cy - cfy, cx - cfx
            ylo, yhi = plan_bounds(dy); xlo, xhi = plan_bounds(dx)
            b_p = []
            for p in range(P):
                s = pa == p
                b_p.append(plan_bounds(dy[s]) + plan_bounds(dx[s]))
            near = (dy >= ylo) & (dy <= yhi) & (dx >= xlo) & (dx <= xhi)
            near_p = np.zeros_like(near)
            for p in range(P):
                a = b_p[p]
                near_p |= (pa == p) & (dy >= a[0]) & (dy <= a[1]) & (dx >= a[2]) & (dx <= a[3])
            n_cand = n_ptc = n_deliv = n_deliv_p = 0
            unit = (lqa * 4096 + yq) * 4096 + xq
            n_dcand = 0
            for ty in range(nty):
                for tx in range(ntx):
                    y0 = ty * th; x0 = tx * tw; rh = min(th, H[l] - y0); rw = min(tw, W[l] - x0)
                    intile = (cy >= y0 - 1) & (cy <= y0 + rh - 1) & (cx >= x0 - 1) & (cx <= x0 + rw - 1)
                    n_deliv += (intile & near).sum(); n_deliv_p += (intile & near_p).sum()
                    n_dcand += np.unique(unit[intile & near]).size
                    for lq in range(L):
                        cy_q = cfloor(np.arange(H[lq]), H[lq], H[l]); cx_q = cfloor(np.arange(W[lq]), W[lq], W[l])
                        n_cand += ((cy_q >= y0 - 1 - yhi) & (cy_q <= y0 + rh - 1 - ylo)).sum() * ((cx_q >= x0 - 1 - xhi) & (cx_q <= x0 + rw - 1 - xlo)).sum()
                        for p in range(P):
                            a = b_p[p]
                            n_ptc += ((cy_q >= y0 - 1 - a[1]) & (cy_q <= y0 + rh - 1 - a[0])).sum() * ((cx_q >= x0 - 1 - a[3]) & (cx_q <= x0 + rw - 1 - a[2])).sum()
            tot[l] += [n_cand, n_dcand, n_ptc, n_deliv, n_deliv_p, len(cy), near.sum()]
    tot /= M
    print("per (batch, head) plane, mean over heads:")
    print("level  candidates  delivering  point-tests  delivered  | per-point: tests  delivered |  points  near(plan)")
    for l in list(range(L)) + ["sum"]:
        t = tot.sum(0) if l == "sum" else tot[l]
        print("%5s  %10d  %9d%%  %11d  %8d%%  | %16d  %8d%% | %7d  %6.1f%%"
              % (l, t[0], 100 * t[1] / max(t[0], 1), 4 * t[0], 100 * t[3] / max(4 * t[0], 1), t[2], 100 * t[4] / max(t[2], 1), t[5], 100 * t[6] / max(t[5], 1)))
    return tot


def synthetic_cells(H, W, sigma, seed=0):
    rng = np.random.default_rng(seed)
    thetas = np.arange(M) * (2 * math.pi / M)
    g = np.stack([np.cos(thetas), np.sin(thetas)], -1)
    g = g / np.abs(g).max(-1, keepdims=True)

    def cells_of(m, l):
        out = [[] for _ in range(6)]
        for lq in range(L):
            yq, xq = np.meshgrid(np.arange(H[lq]), np.arange(W[lq]), indexing="ij")
            yq = yq.ravel(); xq = xq.ravel()
            for p in range(P):
                offx = g[m, 0] * (p + 1) + sigma * rng.standard_normal(yq.shape)
                offy = g[m, 1] * (p + 1) + sigma * rng.standard_normal(yq.shape)
                locx = ((xq + 0.5) / W[lq] + offx / W[l]).astype(np.float32); locy = ((yq + 0.5) / H[lq] + offy / H[l]).astype(np.float32)
                wim = locx * np.float32(W[l]) - np.float32(0.5); him = locy * np.float32(H[l]) - np.float32(0.5)
                ok = (him > -1) & (wim > -1) & (him < H[l]) & (wim < W[l])
                for k, a in enumerate((np.full(ok.sum(), lq), yq[ok], xq[ok], np.full(ok.sum(), p), np.floor(him[ok]).astype(int), np.floor(wim[ok]).astype(int))):
                    out[k].append(a)
        return tuple(np.concatenate(a) for a in out)
    return cells_of


if __name__ == "__main__":
    H, W = [48, 24, 12, 6], [160, 80, 40, 20]
    if len(sys.argv) > 1 and sys.argv[1] == "c5":
        H, W = [160, 80, 40, 20], [240, 120, 60, 30]
    sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    analyse(H, W, synthetic_cells(H, W, sigma))
