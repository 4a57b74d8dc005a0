#!/usr/bin/env python
"""Where does an elimination step spend its cycles?  (diagnostic build with s_memtime stamps, GPU box only)

    python -m semiclassical_amd.build --stamps && python tools/lu_stamps.py [ntraj]

Prints, per wave of workgroup 0, the cycle sums of the last trajectory it processed, split into the steps
in which the wave owned the pivot row and those in which it did not.
Segments: 0 owner work before the barrier | 1 barrier | 2 LDS reads (pivot record + scaled row)
          3 pivot-column fetch | 4 rank-1 update
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SC_LIB_PATH"] = os.path.join(ROOT, "semiclassical_amd", "libsemiclassical_hip_stamps.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from semiclassical_amd import potentials as P, propagators as PR  # noqa: E402
from semiclassical_amd._lib import lib  # noqa: E402

torch.set_default_dtype(torch.float64)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
omega, chi, nac, q0, dt = bench.as60_model()
G = torch.diag(omega)
pot = P.MorsePotential(omega, chi.clone(), nac)
prop = PR.HermanKlukPropagator(G, G, device="cuda")
prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(1))
for _ in range(3):
    prop.step(pot, dt)
prop.synchronize()
buf = (ctypes.c_ulonglong * 64)()
lib.sc_debug_read_stamps.restype = ctypes.c_int
assert lib.sc_debug_read_stamps(buf) == 0
names = ["owner work", "barrier", "LDS reads", "column fetch", "rank-1 update"]
for w in range(4):
    own = [buf[w * 16 + i] for i in range(5)]
    oth = [buf[w * 16 + 8 + i] for i in range(5)]
    print(f"wave {w}: owner steps   " + "  ".join(f"{nm} {v:7d}" for nm, v in zip(names, own)) + f"   total {sum(own)}")
    print(f"        other steps   " + "  ".join(f"{nm} {v:7d}" for nm, v in zip(names, oth)) + f"   total {sum(oth)}")
