"""Build the gfx950 shared library in-tree (hipcc cross-compiles without a GPU).

    python -m semiclassical_amd.build [--force] [-DMACRO=... other hipcc flags]

produces semiclassical_amd/libsemiclassical_hip.so, the only native artefact of
the package.  It is git-ignored but travels to the GPU box with the snapshot.

Every csrc/*.hip is its own translation unit (no relocatable device code): the
objects are compiled side by side, one hipcc process per file, kept under
csrc/build/ with the hash of their inputs, and linked at the end -- editing one
kernel file costs one compile, not eleven.
"""
import glob
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libsemiclassical_hip.so")
OBJDIR = os.path.join(HERE, "csrc", "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in sources() + headers())


def _digest(src, extra):
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + list(extra)).encode())
    for f in [src] + headers():
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _compile(src, extra, tag, verbose):
    os.makedirs(OBJDIR, exist_ok=True)
    stem = os.path.splitext(os.path.basename(src))[0] + (("." + tag) if tag else "")
    obj, stamp = os.path.join(OBJDIR, stem + ".o"), os.path.join(OBJDIR, stem + ".sha")
    want = _digest(src, extra)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return obj
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-c", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"),
                             "-o", obj] + list(extra) + [src]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as fh:
        fh.write(want)
    return obj


def build(force=False, verbose=True, extra=(), out=None, more_sources=(), jobs=None):
    """``extra``: additional hipcc flags (variant libraries: -DSC_TUNING ...), ``out``: where to write the library,
    ``more_sources``: translation units outside csrc/ (tools/variants)."""
    if out is None and not extra and not force and not needs_build():
        return LIB
    srcs = sources() + list(more_sources)
    tag = hashlib.sha256(" ".join(extra).encode()).hexdigest()[:8] if extra else ""
    jobs = jobs or min(len(srcs), max(1, (os.cpu_count() or 2) - 1))
    with ThreadPoolExecutor(jobs) as pool:
        objs = list(pool.map(lambda s: _compile(s, extra, tag, verbose), srcs))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out or LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out or LIB


if __name__ == "__main__":
    argv = sys.argv[1:]
    target = None
    if "--out" in argv:                      # variant library: python -m semiclassical_amd.build --out var/libsc_x.so -DSC_TUNING ...
        i = argv.index("--out")
        target = argv[i + 1]
        del argv[i:i + 2]
    flags = [a for a in argv if a != "--force"]
    if flags and target is None:
        sys.exit("extra compiler flags make a variant library: name it with --out PATH (the product library is built without)")
    variants = sorted(glob.glob(os.path.join(ROOT, "tools", "variants", "*.hip"))) if "-DSC_TUNING" in flags else []
    print(build(force="--force" in argv, extra=flags, out=target, more_sources=variants))
