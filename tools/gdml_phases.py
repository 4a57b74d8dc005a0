#!/usr/bin/env python
"""Where a geometry's time goes inside gdml_stage_kernel: shader cycles per phase of wave 0 of workgroup 0 (variant library
built with -DGDML_PHASE_CLOCK).

    tools/mkvar.sh gdmlclock -DGDML_PHASE_CLOCK
    SC_LIB_PATH=$PWD/var/libsc_gdmlclock.so python tools/gdml_phases.py [n]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_default_dtype(torch.float64)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    from semiclassical_amd import _lib
    import bench
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.sc_gdml_phase_clock.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    buf = torch.zeros(16, dtype=torch.int64, device=dev)
    assert lib.sc_gdml_phase_clock(ctypes.c_void_p(buf.data_ptr())) == 0
    out = bench.config5(dev, n, 2)
    torch.cuda.synchronize()
    c = buf.cpu().numpy().astype(float)
    names = ["prologue", "copy requests", "row reductions + tail", "gathers J^T xs, J^T A", "barrier 1", "gradient + operand rows",
             "copy wait + barrier 2", "matrix-core phase", "epilogue (gradient, pair terms, stores)"]
    calls = c[10]
    total = c[:9].sum()
    print(f"# gdml_stage_kernel, n = {n}: wave 0 of workgroup 0, {int(calls)} geometries, shader cycles per geometry (mean) and share")
    for name, v in zip(names, c[:9]):
        print(f"{name:42s} {v / calls:10.0f}  {100 * v / total:5.1f} %")
    print(f"{'total':42s} {total / calls:10.0f}")
    print("stage kernel per launch (ms):", out["kernels_ms"]["gdml_stage_kernel (x4 per step)"])


if __name__ == "__main__":
    main()
