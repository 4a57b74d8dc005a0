#!/usr/bin/env python
"""Where a workgroup's time goes inside hk_step_sd_kernel on the headline workload: 100 MHz wall-clock stamps of lane 0 of every
wave of workgroups 0 and 512 at six points of every item (variant library built with -DSD_PHASE_CLOCK).

    tools/mkvar.sh sdclock -DSD_PHASE_CLOCK
    SC_LIB_PATH=$PWD/var/libsc_sdclock.so python tools/sd_phases.py [pairs|single]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_default_dtype(torch.float64)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "pairs"
    import bench
    from semiclassical_amd import _lib, potentials as P, propagators as PR
    from semiclassical_amd._lib import lib, check, ptr
    so = ctypes.CDLL(_lib.LIB_PATH)
    so.sc_sd_phase_clock.argtypes = [ctypes.c_void_p]
    n = 100000
    omega, chi, nac, q0, dt = bench.as60_model(60)
    G = torch.diag(omega)
    prop = PR.HermanKlukPropagator(G, G, device="cuda")
    prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(1))
    pot = P.MorsePotential(omega, chi.clone(), nac)
    desc = prop._potential_descriptor(pot, dt)
    prop._set_mono_layout(_lib.SC_MONO_TILED16)
    prop._launch_step_pair(desc, dt)
    m = prop._multi
    if mode == "pairs":
        launch = lambda: check(lib.sc_hk_step_multi(desc, prop._state, prop._hk, m["ms"], dt, ptr(m["epart"]), prop._stream()))
    else:
        launch = lambda: check(lib.sc_hk_step(desc, prop._state, prop._hk, dt, 0, ptr(prop._epart), prop._stream()))
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    buf = torch.zeros(2 * 4 * 256 * 6, dtype=torch.int64, device="cuda")
    assert so.sc_sd_phase_clock(ctypes.c_void_p(buf.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    torch.cuda.synchronize()
    assert so.sc_sd_phase_clock(None) == 0
    t = buf.cpu().numpy().reshape(2, 4, 256, 6).astype(np.int64)
    print(f"# hk_step_sd_kernel, {mode}, n = {n}, D = 60: launch {e0.elapsed_time(e1):.3f} ms; times in us (100 MHz clock), mean over the items "
          "of a wave")
    names = ["streaming phase (loads, RK4, stores, matrix)", "wait at the reset barrier", "elimination (4 diagonal blocks)",
             "wait at the end barrier", "determinant / sign (thread 0)", "-> next item"]
    for blk in range(2):
        for wave in range(4):
            x = t[blk, wave]
            items = int((x[:, 5] > 0).sum())
            x = x[:items]
            d = np.diff(x, axis=1) / 100.0
            gap = (x[1:, 0] - x[:-1, 5]) / 100.0
            total = (x[-1, 5] - x[0, 0]) / 100.0
            print(f"workgroup {blk * 512} wave {wave}: {items} items, {total / items:.2f} us per item | "
                  + " | ".join(f"{nm.split(' (')[0]} {v:.2f}" for nm, v in zip(names, list(d.mean(axis=0)) + [gap.mean()])))
            if mode == "pairs" and wave == 0:
                for ks in range(2):
                    dd = d[ks::2].mean(axis=0)
                    print(f"    sub-step {ks}: " + " | ".join(f"{v:.2f}" for v in dd))


if __name__ == "__main__":
    main()
