"""Builds the in-tree HIP shared libraries for gfx950 with hipcc (cross-compiles without a GPU).

    python -m monosowa_amd.build [--force]

Outputs go to ``monosowa_amd/lib/*.so`` (git-ignored, shipped to the GPU box by gpurun), each with a ``.srchash`` beside it:
the sha256 of the sources and flags it was built from.  A library is recompiled when that hash differs from the tree's -- not
by file times -- and the build says per library whether it compiled or found it up to date.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
ARCH = "gfx950"

# library name -> (translation unit, extra link flags); every file under csrc/ and include/ is part of every library's source hash
LIBS = {
    "libmonosowa_msda.so": ("msda_capi.hip", []),
    "libmonosowa_pointwise.so": ("pointwise.hip", []),
    "libmonosowa_attn.so": ("flash_attn.hip", []),
    "libmonosowa_kitti.so": ("rotate_iou.hip", []),
    "libmonosowa_gemm.so": ("gemm_lt.cpp", ["-lhipblaslt"]),      # host code only: the hipBLASLt shim (include/monosowa_gemm.h)
}

FLAGS = ["-O3", "--offload-arch=" + ARCH, "-munsafe-fp-atomics", "-fPIC", "-shared", "-std=c++17",
         "-fno-gpu-rdc", "-Wall"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X-native kernels cannot be built")
    return exe


def source_hash(extra=()):
    """sha256 over the contents of every file under csrc/ and include/ (sorted by name) and the compiler flags: what a library was
    built FROM.  Kept beside each .so (lib/<name>.srchash), so that a reused binary is provably the committed source -- file times
    say nothing on a fresh checkout or on the GPU box's snapshot."""
    import hashlib
    h = hashlib.sha256()
    inc = os.path.normpath(os.path.join(HERE, "..", "include"))
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(inc, f) for f in os.listdir(inc)]
    for path in sorted(files):
        if os.path.isfile(path):
            h.update(os.path.basename(path).encode() + b"\0")
            with open(path, "rb") as f:
                h.update(f.read())
            h.update(b"\0")
    for x in extra:
        h.update(str(x).encode() + b"\0")
    return h.hexdigest()


def _recorded_hash(out):
    try:
        with open(out + ".srchash") as f:
            return f.read().strip()
    except OSError:
        return None


def _build_one(out, cmd, want, force, verbose):
    """Compiles ``out`` unless the hash recorded beside it equals ``want``; says which of the two happened."""
    name = os.path.basename(out)
    if not force and os.path.exists(out) and _recorded_hash(out) == want:
        if verbose:
            print("%s: up to date (source hash %s)" % (name, want[:16]), flush=True)
        return False
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(out + ".srchash", "w") as f:
        f.write(want + "\n")
    if verbose:
        print("%s: compiled (source hash %s)" % (name, want[:16]), flush=True)
    return True


def build_host_libs(force=False, verbose=False):
    """Host-side native helpers (plain g++, no GPU code)."""
    out = os.path.join(LIBDIR, "libmonosowa_lsap.so")
    src = os.path.join(CSRC, "lsap.cpp")
    flags = ["-O3", "-fPIC", "-shared", "-std=c++17", "-Wall", "-pthread"]
    cmd = [shutil.which("g++") or "g++"] + flags + ["-o", out, src]
    _build_one(out, cmd, source_hash(["g++"] + flags), force, verbose)
    return [out]


def build_all(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    built = build_host_libs(force, verbose)
    want = source_hash(["hipcc"] + FLAGS)
    for name, (tu, link) in LIBS.items():
        out = os.path.join(LIBDIR, name)
        _build_one(out, [hipcc()] + FLAGS + ["-o", out, os.path.join(CSRC, tu)] + link, want, force, verbose)
        built.append(out)
    _stamp_commit()
    return built


def _stamp_commit():
    """lib/BUILD_COMMIT: the commit the tree was at when the libraries were last (re)checked -- the GPU box's snapshot has no .git,
    and bench.py / tools/collect_pmc.sh record which code a measurement belongs to."""
    try:
        root = os.path.dirname(HERE)
        h = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10)
        if h.returncode != 0:
            return
        d = subprocess.run(["git", "-C", root, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True, timeout=10)
        with open(os.path.join(LIBDIR, "BUILD_COMMIT"), "w") as f:
            f.write(h.stdout.strip() + ("-dirty" if d.stdout.strip() else "") + "\n")
    except Exception:
        pass


if __name__ == "__main__":
    for p in build_all(force="--force" in sys.argv, verbose=True):
        print("built", p)
