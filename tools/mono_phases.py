#!/usr/bin/env python
"""Where the time of dense_mono_mfma_slab_dma2 goes: shader cycles per phase of wave 0 of workgroup 0 (variant library built with
-DGDML_PHASE_CLOCK), config 5 (D = 90).

    tools/mkvar.sh gdmlclock -DGDML_PHASE_CLOCK
    SC_LIB_PATH=$PWD/var/libsc_gdmlclock.so python tools/mono_phases.py [n]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_default_dtype(torch.float64)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    from semiclassical_amd import _lib
    import bench
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.sc_mono_phase_clock.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    buf = torch.zeros(8, dtype=torch.int64, device=dev)
    assert lib.sc_mono_phase_clock(ctypes.c_void_p(buf.data_ptr())) == 0
    out = bench.config5(dev, n, 2)
    torch.cuda.synchronize()
    c = buf.cpu().numpy().astype(float)
    names = ["item bookkeeping", "row loads issued (48 per lane)", "wait for image / rows + barrier (x4)", "product loop, 23 k-slices x 6 MFMA (x4)",
             "RK4 update of the tile (x4)", "row stores issued"]
    items = c[7]
    total = c[:6].sum()
    print(f"# dense_mono_mfma_slab_dma2<6,23>, n = {n}: wave 0 of workgroup 0, last launch: {int(items)} slabs, shader cycles per slab and share")
    for name, v in zip(names, c[:6]):
        print(f"{name:46s} {v / items:10.0f}  {100 * v / total:5.1f} %")
    print(f"{'total':46s} {total / items:10.0f}   (ideal matrix-pipe time of a slab: 4 x 138 MFMA x 64 cycles = 35328)")
    print("dense_mono_step per launch (ms):", out["kernels_ms"]["dense_mono_step (MFMA RK4 + prefactor)"],
          " slab kernel (difference):", out["kernels_ms"]["dense_mono_mfma_slab_kernel (difference)"])


if __name__ == "__main__":
    main()
