#!/usr/bin/env python
"""Time between consecutive pivot publications in the elimination of hk_step_sd_kernel (workgroup 0, first 64 items), shader clock.

    tools/mkvar.sh luclock -DLU_PIVOT_CLOCK
    SC_LIB_PATH=$PWD/var/libsc_luclock.so python tools/lu_pivot_clock.py [DIM]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_default_dtype(torch.float64)


def main():
    dim = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    import bench
    from semiclassical_amd import _lib, potentials as P, propagators as PR
    from semiclassical_amd._lib import lib, check, ptr
    so = ctypes.CDLL(_lib.LIB_PATH)
    so.sc_lu_pivot_clock.argtypes = [ctypes.c_void_p]
    n = 100000
    omega, chi, nac, q0, dt = bench.as60_model(dim)
    G = torch.diag(omega)
    prop = PR.HermanKlukPropagator(G, G, device="cuda")
    prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(1))
    pot = P.MorsePotential(omega, chi.clone(), nac)
    desc = prop._potential_descriptor(pot, dt)
    prop._set_mono_layout(_lib.SC_MONO_TILED16)
    launch = lambda: check(lib.sc_hk_step(desc, prop._state, prop._hk, dt, 0, ptr(prop._epart), prop._stream()))
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    buf = torch.zeros(64 * 64 * 4, dtype=torch.int64, device="cuda")
    assert so.sc_lu_pivot_clock(ctypes.c_void_p(buf.data_ptr())) == 0
    launch()
    torch.cuda.synchronize()
    assert so.sc_lu_pivot_clock(None) == 0
    full = buf.cpu().numpy().reshape(64, 64, 4).astype(np.int64)[8:56]
    t = full[:, :, 2]                                                  # skip the start-up and the tail of the launch
    print(f"# D = {dim}, n = {n}: shader-clock cycles between consecutive pivot publications, workgroup 0, mean over {len(t)} items")
    nr = (dim + 15) // 16
    for kb in range(nr):
        nk = min(16, dim - 16 * kb)
        order = [kt for kt in range(16) if 4 * (kt & 3) + (kt >> 2) < nk]
        stamps = t[:, [16 * kb + kt for kt in order]]
        d = np.diff(stamps, axis=1)
        if d.size == 0:
            print(f"block {kb} (N = {nr - kb}): one step")
            continue
        print(f"block {kb} (N = {nr - kb}): {len(order)} steps, {d.mean():7.0f} cycles per step (min {d.min()}, median {np.median(d):.0f}, max {d.max()}); "
              f"block total {(stamps[:, -1] - stamps[:, 0]).mean():8.0f}")
        # inside a step: poll of the previous record returned -> first-slot update done -> record published -> next owner's poll returns
        idx = [16 * kb + kt for kt in order[1:]]
        prev = [16 * kb + kt for kt in order[:-1]]
        wait = full[:, idx, 0] - full[:, prev, 2]
        upd = full[:, idx, 1] - full[:, idx, 0]
        pub = full[:, idx, 2] - full[:, idx, 1]
        print(f"    publication -> next owner has the row: {np.median(wait):6.0f} | column fetch + first-slot update: {np.median(upd):6.0f} | "
              f"search, inverse, scaling, publication: {np.median(pub):6.0f}   (medians, cycles)")
        if kb + 1 < nr:
            nxt = t[:, 16 * (kb + 1)]
            print(f"    hand-over to block {kb + 1}: {(nxt - stamps[:, -1]).mean():7.0f} cycles")
    first, last = t[:, 0], t[:, [16 * (nr - 1) + kt for kt in range(16) if 4 * (kt & 3) + (kt >> 2) < min(16, dim - 16 * (nr - 1))][-1]]
    print(f"elimination (first to last publication): {(last - first).mean():8.0f} cycles; item to item: {np.diff(first).mean():8.0f} cycles")


if __name__ == "__main__":
    main()
