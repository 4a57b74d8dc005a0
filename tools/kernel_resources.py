#!/usr/bin/env python
"""Registers, scratch and LDS of the kernels in a compiled object / library (gfx950 code object metadata).

    python tools/kernel_resources.py semiclassical_amd/csrc/build/sc_hk_step_lin.o [name-filter]
"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    path, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    notes = ""
    with tempfile.TemporaryDirectory() as tmp:
        co = os.path.join(tmp, "dev.co")
        fat = os.path.join(tmp, "fat.bin")       # the fat binary sits in section .hip_fatbin of the host object / library
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat])
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)] + [len(blob)]
        for i in range(len(starts) - 1):
            piece = os.path.join(tmp, f"bundle{i}.bin")
            with open(piece, "wb") as fh:
                fh.write(blob[starts[i]:starts[i + 1]])
            listing = subprocess.check_output([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={piece}"], text=True).split()
            for target in (t for t in listing if "gfx950" in t):
                subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={piece}",
                                       f"--targets={target}", f"--output={co}"])
                notes += subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        get = lambda k: (re.search(rf"\.{k}:\s+(\d+)", blk) or [None, "?"])[1]
        demangled = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
        if flt and flt not in demangled:
            continue
        agpr = re.match(r"\s*(\d+)", blk).group(1)
        print(f"{demangled[:110]:110s} vgpr {get('vgpr_count'):>3} agpr {agpr:>3} sgpr {get('sgpr_count'):>3} "
              f"scratch {get('private_segment_fixed_size'):>5} B  lds {get('group_segment_fixed_size'):>6} B  "
              f"spill v {get('vgpr_spill_count')} s {get('sgpr_spill_count')}")


if __name__ == "__main__":
    main()
