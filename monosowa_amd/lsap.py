"""ctypes face of the native assignment solver (csrc/lsap.cpp)."""
import ctypes
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmonosowa_lsap.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(_PATH)
        P, I = ctypes.c_void_p, ctypes.c_int64
        lib.lsap_solve_f64.restype = I
        lib.lsap_solve_f64.argtypes = [P, I, I, P, P]
        lib.lsap_match_groups_f32.restype = I
        lib.lsap_match_groups_f32.argtypes = [P, I, I, I, I, P, I, I, P, P, P]
        lib.lsap_match_flat_f32.restype = I
        lib.lsap_match_flat_f32.argtypes = [P, I, I, I, I, P, I, I, P, I]
        assert lib.lsap_abi_version() == 1
        _lib = lib
    return _lib


def linear_sum_assignment(cost):
    """Same contract as scipy.optimize.linear_sum_assignment (minimisation)."""
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    k = min(nr, nc)
    rows, cols = np.empty(k, np.int64), np.empty(k, np.int64)
    got = _load().lsap_solve_f64(cost.ctypes.data, nr, nc, rows.ctypes.data, cols.ctypes.data)
    if got < 0:
        raise ValueError("cost matrix is infeasible")
    return rows, cols


def match_groups(cost, sizes, group_num, padded=False):
    """cost [NL,B,Q,T] float32 (host) -> per layer, per image (src int64, tgt int64) numpy arrays.
    padded=False: T = sum(sizes), image b's targets at columns [off_b, off_b + sizes[b]);
    padded=True : T >= max(sizes), image b's targets at columns [0, sizes[b])."""
    cost = np.ascontiguousarray(cost, dtype=np.float32)
    NL, B, Q, T = cost.shape
    sizes = np.ascontiguousarray(sizes, dtype=np.int64)
    assert sizes.shape == (B,) and Q % group_num == 0
    assert (int(sizes.max(initial=0)) <= T) if padded else (int(sizes.sum()) == T)
    gq = Q // group_num
    per_layer = int(sum(group_num * min(gq, int(n)) for n in sizes))
    src, tgt = np.empty(NL * per_layer, np.int64), np.empty(NL * per_layer, np.int64)
    count = np.empty(NL * B, np.int64)
    got = _load().lsap_match_groups_f32(cost.ctypes.data, NL, B, Q, T, sizes.ctypes.data, group_num, int(padded),
                                        src.ctypes.data, tgt.ctypes.data, count.ctypes.data)
    if got < 0:
        raise ValueError("cost matrix is infeasible")
    out, pos = [], 0
    for l in range(NL):
        layer = []
        for b in range(B):
            c = int(count[l * B + b])
            layer.append((src[pos:pos + c], tgt[pos:pos + c]))
            pos += c
        out.append(layer)
    return out


def _default_threads():
    """Host threads of the assignment pool: 8 (0.13 ms for the 528 problems of a train step; 4 threads: 0.21), capped by this
    rank's SHARE of the host's cores -- under torch.distributed.run every rank of the node has its own pool (8 ranks x 8 threads
    next to 8 x OMP would oversubscribe the cores the matcher's wait sits on)."""
    cores = os.cpu_count() or 1
    try:
        ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        ranks = 1
    return max(1, min(8, cores // (2 * ranks)))


N_THREADS = _default_threads()


_pinned = {}


def _pinned_idx(n):
    """a reusable pinned int64 buffer of >= n elements as a numpy view (the criterion copies the indices to the device: from
    pageable memory that copy blocks the host until everything queued before it has run; from pinned memory it is queued)"""
    import torch
    buf = _pinned.get("idx")
    if buf is None or buf.numel() < n:
        buf = _pinned["idx"] = torch.empty(max(n, 1 << 14), dtype=torch.int64).pin_memory()
    return buf


def match_flat(cost, sizes, group_num, padded=False, n_threads=None, pinned=False):
    """Same assignments as ``match_groups`` as one int64 array [3, NL, K]: (image, query, target index offset by the
    targets of the images before it), K pairs per layer in (image, group) order -- the index tensor of the criterion's
    flat losses; solved on ``n_threads`` host threads."""
    cost = np.ascontiguousarray(cost, dtype=np.float32)
    NL, B, Q, T = cost.shape
    sizes = np.ascontiguousarray(sizes, dtype=np.int64)
    assert sizes.shape == (B,) and Q % group_num == 0
    assert (int(sizes.max(initial=0)) <= T) if padded else (int(sizes.sum()) == T)
    gq = Q // group_num
    K = int(sum(group_num * min(gq, int(n)) for n in sizes))
    if pinned:                                    # -> a torch tensor view of the pinned buffer (valid until the next call)
        holder = _pinned_idx(3 * NL * K)[:3 * NL * K].view(3, NL, K)
        idx_ptr, idx = holder.data_ptr(), holder
    else:
        idx = np.empty((3, NL, K), np.int64)
        idx_ptr = idx.ctypes.data
    got = _load().lsap_match_flat_f32(cost.ctypes.data, NL, B, Q, T, sizes.ctypes.data, group_num, int(padded),
                                      idx_ptr, N_THREADS if n_threads is None else n_threads)
    if got < 0:
        raise ValueError("cost matrix is infeasible")
    assert got == K
    return idx
