"""Builds the in-tree HIP shared libraries for gfx950 with hipcc (cross-compiles without a GPU).

    python -m monosowa_amd.build [--force]

Outputs go to ``monosowa_amd/lib/*.so`` (git-ignored, shipped to the GPU box by gpurun).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
ARCH = "gfx950"

# library name -> (translation unit, extra dependencies that trigger a rebuild); every file under
# csrc/ counts as a dependency as well
LIBS = {
    "libmonosowa_msda.so": ("msda_capi.hip", [os.path.join("..", "..", "include", "monosowa_msda.h")]),
    "libmonosowa_pointwise.so": ("pointwise.hip", [os.path.join("..", "..", "include", "monosowa_pointwise.h")]),
    "libmonosowa_attn.so": ("flash_attn.hip", [os.path.join("..", "..", "include", "monosowa_attn.h")]),
    "libmonosowa_kitti.so": ("rotate_iou.hip", [os.path.join("..", "..", "include", "monosowa_kitti.h")]),
}

FLAGS = ["-O3", "--offload-arch=" + ARCH, "-munsafe-fp-atomics", "-fPIC", "-shared", "-std=c++17",
         "-fno-gpu-rdc", "-Wall"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X-native kernels cannot be built")
    return exe


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_host_libs(force=False, verbose=False):
    """Host-side native helpers (plain g++, no GPU code)."""
    out = os.path.join(LIBDIR, "libmonosowa_lsap.so")
    src = os.path.join(CSRC, "lsap.cpp")
    if force or _stale(out, [src]):
        cmd = [shutil.which("g++") or "g++", "-O3", "-fPIC", "-shared", "-std=c++17", "-Wall", "-pthread", "-o", out, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return [out]


def build_all(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    built = build_host_libs(force, verbose)
    for name, (tu, deps) in LIBS.items():
        out = os.path.join(LIBDIR, name)
        src = os.path.join(CSRC, tu)
        alldeps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + \
                  [os.path.normpath(os.path.join(CSRC, d)) for d in deps]
        if force or _stale(out, alldeps):
            cmd = [hipcc()] + FLAGS + ["-o", out, src]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
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
