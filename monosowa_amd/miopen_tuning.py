"""Shipped MIOpen find results for the MonoDETR convolutions on MI355X.

MIOpen picks a convolution kernel per problem either from its heuristics (PyTorch's default, `cudnn.benchmark =
False`) or from a measured search ("find", `cudnn.benchmark = True`).  The search takes ~165 s for the ~160
(problem, direction) pairs of a ResNet-50 MonoDETR train step at B = 16, 1280x384 and makes the step 2 % faster.
`miopen_db/` holds the outcome of that search (MIOpen's own user find-db / perf-db text files, produced on an
MI355X by `python bench.py --miopen-find` with MIOPEN_USER_DB_PATH pointing here, plus the kernel cache of the
JIT-compiled winners when present), so a fresh process gets the measured choices without searching.

`use_shipped_db()` must run before the first convolution (MIOpen reads the variables when its handle is created).
`torch.backends.cudnn.benchmark` can stay False: MIOpen's immediate mode consults the find-db first, so the measured
winners are used without any search at start-up (with benchmark = True PyTorch re-runs a 25 s find per process).
Each rank works on its own copy: MIOpen appends to these files.
"""
import glob
import os
import shutil
import tempfile

DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def shipped_db_available():
    return bool(glob.glob(os.path.join(DB_DIR, "*.ufdb.txt")))


def use_shipped_db(rank=0):
    """Point MIOpen at a per-rank copy of the shipped find-db; returns the directory or None when there is none
    (or when the caller already chose a MIOPEN_USER_DB_PATH)."""
    if not shipped_db_available() or "MIOPEN_USER_DB_PATH" in os.environ:
        return None
    dst = os.path.join(tempfile.gettempdir(), "monosowa_miopen_%d_r%d_%d" % (os.getuid(), rank, os.getpid()))
    os.makedirs(dst, exist_ok=True)
    for f in glob.glob(os.path.join(DB_DIR, "*")):
        if os.path.isfile(f) and not f.endswith((".md", ".csv")):
            shutil.copy(f, dst)
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = dst
    use_shipped_gemm_choices(dst)
    return dst


def use_shipped_gemm_choices(scratch):
    """PyTorch TunableOp results (which rocBLAS / hipBLASLt solution per GEMM shape of the train step; produced by
    `PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 python bench.py`, 70 s) are replayed without tuning.
    TunableOp appends the device ordinal to the file name, so the file is laid out once per ordinal; its
    validator lines make PyTorch ignore it when the library versions differ."""
    src = os.path.join(DB_DIR, "tunableop_gfx950.csv")
    if not os.path.exists(src) or "PYTORCH_TUNABLEOP_ENABLED" in os.environ:
        return
    for ordinal in range(8):
        shutil.copy(src, os.path.join(scratch, "tunableop%d.csv" % ordinal))
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
    os.environ["PYTORCH_TUNABLEOP_TUNING"] = "0"
    os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.path.join(scratch, "tunableop.csv")
