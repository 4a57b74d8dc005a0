"""Build the gfx950 shared library in-tree (hipcc cross-compiles without a GPU).

    python -m semiclassical_amd.build

produces semiclassical_amd/libsemiclassical_hip.so, the only native artefact of
the package.  It is git-ignored but travels to the GPU box with the snapshot.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libsemiclassical_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    deps = sources() + glob.glob(os.path.join(HERE, "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build(force=False, verbose=True, extra=(), out=None):
    if out is None and not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"),
           "-o", out or LIB] + list(extra) + sources()
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out or LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
