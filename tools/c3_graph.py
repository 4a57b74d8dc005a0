"""config 3 (WM on methylium, n = 1e5): eager launches against HIP-graph replay of the per-step launch sequence"""
import json, sys, torch
sys.path.insert(0, '.')
import bench
from semiclassical_amd import potentials as P, propagators as PR
dev = torch.device('cuda', 0)
torch.set_default_dtype(torch.float64)
g = bench._load("wm_methylium")
pot = P.MolecularHarmonicPotential.from_arrays(g["pos0"], g["energy0"], g["grad0"], g["hess0"], g["masses"], g["nac0"], origin=float(g["origin"]))
Gi = bench._T(g["Gamma_i"])
out = {}
for n in (100000, 10000):
    prop = PR.WaltonManolopoulosPropagator(Gi, Gi, float(g["alpha"]), float(g["beta"]), device=dev)
    prop.initial_conditions(bench._T(g["q0"]), bench._T(g["p0"]), bench._T(g["Gamma_0"]), ntraj=n, generator=torch.Generator().manual_seed(7))
    dt, E0 = float(g["dt"]), float(g["E0"])
    eager = bench._timed_loop(prop, pot, dt, E0, 30, dev)
    try:
        graph = bench._timed_loop(prop, pot, dt, E0, 30, dev, use_graph=True)
    except Exception as e:
        graph = repr(e)
    out[n] = {"eager_ms_per_step": eager / 30 * 1e3, "graph_ms_per_step": graph / 30 * 1e3 if isinstance(graph, float) else graph}
print(json.dumps(out))
