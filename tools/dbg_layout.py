import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from semiclassical_amd import _lib, potentials as P, propagators as PR
torch.set_default_dtype(torch.float64)
D, n = 60, 2500
omega, chi, nac, q0, _ = bench.as60_model(D)
G = torch.diag(omega)
pot = P.MorsePotential(omega, chi.clone(), nac)
props = []
for tiled in (True, False):
    prop = PR.HermanKlukPropagator(G, G, device="cuda")
    prop._tiled_fast_path = tiled
    prop.pair_steps = False
    prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(D))
    props.append(prop)
a, b = props
for k in range(2):
    a.step(pot, 4.0); b.step(pot, 4.0)
    torch.cuda.synchronize()
    ya, yb = a.y, b.y
    d = (ya - yb).abs()
    bad = (d > 1e-12 * yb.abs().max()).nonzero()
    print("step", k, "layouts", a._state.mono_layout, b._state.mono_layout, "mismatches", bad.shape[0])
    if bad.shape[0]:
        rows, trs = bad[:, 0].cpu().numpy(), bad[:, 1].cpu().numpy()
        print(" trajectories:", np.unique(trs)[:20], "count", len(np.unique(trs)))
        r = rows - 2 * D
        p, ab = r // (D * D), r % (D * D)
        print(" planes", np.unique(p), "rows a", np.unique(ab // D)[:40], "cols b", np.unique(ab % D)[:40])
        print(" c2 diff", (a._c2 - b._c2).abs().max().item())
        break
if bad.shape[0]:
    tr = int(trs[0])
    for (rr, tt) in list(zip(rows, trs))[:8]:
        print(" row", int(rr), "traj", int(tt), "tiled", ya[rr, tt].item(), "rowmajor", yb[rr, tt].item())
    # all four planes of the corner block of that trajectory
    for p in range(4):
        blk_a = ya[2 * D + p * D * D: 2 * D + (p + 1) * D * D, tr].reshape(D, D)[44:48, 44:48]
        blk_b = yb[2 * D + p * D * D: 2 * D + (p + 1) * D * D, tr].reshape(D, D)[44:48, 44:48]
        print(" plane", p, "tiled diag", torch.diagonal(blk_a).cpu().numpy(), "rowmajor diag", torch.diagonal(blk_b).cpu().numpy(), "offdiag max", (blk_a - torch.diag(torch.diagonal(blk_a))).abs().max().item())
