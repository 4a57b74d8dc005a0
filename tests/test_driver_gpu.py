"""`semi dynamics` task through the HIP engine: npz schema, accumulation over repetitions, rates task."""
import json
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu


def test_dynamics_and_rates_task(tmp_path):
    from semiclassical_amd import driver
    g = cases.load("hk_as5_chi002")
    model = tmp_path / "AS_model.dat"
    rows = np.vstack((g["omega"] * 219474.63, 0.5 * g["omega"] * g["q0"] ** 2 * np.sign(g["q0"]), g["nac"],
                      np.full(5, 0.02))).T
    np.savetxt(model, rows)
    out = tmp_path / "correlations.npz"
    task = {"task": "dynamics", "potential": {"type": "anharmonic AS", "model_file": str(model)},
            "propagator": "HK", "batch_size": 4000, "num_trajectories": 8000, "num_steps": 30, "time_step_fs": 0.04,
            "results": {"correlations": str(out)}, "manual_seed": 0}
    driver.run_semiclassical_dynamics(task, device="cuda")
    d = np.load(out)
    assert set(d.keys()) >= {"propagator", "times", "autocorrelation", "ic_correlation", "adiabatic_gap",
                             "zero_point_energy", "trajectories"}
    assert int(d["trajectories"]) == 8000 and str(d["propagator"]) == "HK"
    assert abs(d["autocorrelation"][0] - 1.0) < 1e-3
    nt, dt = 30, 0.04 / 0.02418884326505
    assert np.allclose(d["times"], np.linspace(0.0, nt * dt, nt))                 # quirk Q3
    # accumulate a third batch without overwriting
    task2 = dict(task, num_trajectories=4000, results={"correlations": str(out), "overwrite": False})
    task2.pop("manual_seed")
    driver.run_semiclassical_dynamics(task2, device="cuda")
    d2 = np.load(out)
    assert int(d2["trajectories"]) == 12000
    assert not np.array_equal(d2["autocorrelation"], d["autocorrelation"])
    driver.calculate_rates({"task": "rates", "correlations": str(out), "rates": str(out)})
    d3 = np.load(out)
    assert "ic_rate" in d3 and d3["ic_rate"].shape == d3["energies"].shape and (d3["energies"] >= 0).all()


def test_calc_norm_every_does_not_change_the_correlations(tmp_path, caplog):
    """cli.py:424-429: the norm diagnostic cuts the fused loop into segments; results must be those of one loop"""
    import logging
    from semiclassical_amd import driver
    g = cases.load("hk_as5_chi002")
    model = tmp_path / "AS_model.dat"
    rows = np.vstack((g["omega"] * 219474.63, 0.5 * g["omega"] * g["q0"] ** 2 * np.sign(g["q0"]), g["nac"],
                      np.full(5, 0.02))).T
    np.savetxt(model, rows)
    res = []
    for every in (0, 7):
        out = tmp_path / f"c{every}.npz"
        task = {"task": "dynamics", "potential": {"type": "anharmonic AS", "model_file": str(model)},
                "propagator": "HK", "batch_size": 500, "num_trajectories": 500, "num_steps": 20, "time_step_fs": 0.04,
                "results": {"correlations": str(out)}, "manual_seed": 3, "calc_norm_every": every}
        with caplog.at_level(logging.INFO, logger="semiclassical_amd.driver"):
            driver.run_semiclassical_dynamics(task, device="cuda")
        res.append(dict(np.load(out)))
    assert np.allclose(res[0]["autocorrelation"], res[1]["autocorrelation"], rtol=1e-13, atol=0)
    assert np.allclose(res[0]["ic_correlation"], res[1]["ic_correlation"], rtol=1e-13, atol=0)
    assert sum("norm=" in r.getMessage() for r in caplog.records) == 3          # t = 0, 7, 14
