"""One rank of the driver-level multi-process test (started by semiclassical_amd.distributed.launch_local_ranks).

Every rank runs `semiclassical_amd.driver.run_semiclassical_dynamics` on the reference's methylium example task; the
process group makes the driver share every batch among the ranks (here: two processes on cuda:0, gloo).  The sampled
phase-space points of the reference's own run (tests/golden/driver_methylium.npz) are injected where the driver draws a
batch on the host, so that rank 0's result file can be compared with the reference's.

    python tests/_rank_driver.py HK|WM OUT.npz
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.set_default_dtype(torch.float64)


def main():
    prop, out = sys.argv[1], sys.argv[2]
    import torch.distributed as dist
    from semiclassical_amd import distributed as D, driver, propagators as PR
    from tests import cases
    rank, world, _ = D.init_from_env()
    g = cases.load("driver_methylium")
    fchk = os.path.join(ROOT, "tests", "golden", "fchk")
    task = json.loads(str(g[f"{prop}_task"]))
    task["potential"] = {"type": "harmonic", "ground": os.path.join(fchk, "methylium_s0.fchk"),
                         "excited": os.path.join(fchk, "methylium_s1.fchk"),
                         "coupling": os.path.join(fchk, "methylium_s1.fchk")}
    task["results"] = {"correlations": out}
    task["sampling"] = "host"
    if os.environ.get("SC_TEST_NORM_EVERY"):
        task["calc_norm_every"] = int(os.environ["SC_TEST_NORM_EVERY"])
    zis, probis = g[f"{prop}_zi"], g[f"{prop}_probi"]
    count = {"rep": 0}

    def from_golden(self, q0, p0, Gamma_0, ntraj, generator=None):
        rep = count["rep"]
        count["rep"] += 1
        assert zis[rep].shape[1] == ntraj
        return cases.T(zis[rep]), cases.T(probis[rep])
    PR.HermanKlukPropagator.draw_initial_conditions = from_golden
    driver.run_semiclassical_dynamics(task, device=f"cuda:{os.environ.get('SC_TEST_DEVICE', '0')}")
    assert count["rep"] == 2, count
    assert (rank == 0) or not os.path.exists(out + f".rank{rank}")       # only rank 0 keeps a result file
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
