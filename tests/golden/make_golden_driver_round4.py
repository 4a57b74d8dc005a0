#!/usr/bin/env python
"""Round-4 addition to the driver goldens, produced by running the REFERENCE's own driver (build container only):

  driver_as60.npz   `semi dynamics` + `semi rates` (cli.py:171-476, 519-570) on an "anharmonic AS" task with the synthetic 60-mode
                    model of BASELINE.json config 2 (written to a model file in the reference's four-column format): HK, two
                    repetitions of 48 trajectories, 24 steps of 0.05 fs.  Stored: the model rows, the task, zi / probi of
                    every repetition, every key of the npz the reference wrote.
  driver_as24_wm.npz  the same with the Walton-Manolopoulos propagator on the first 24 modes (cell_width 500), 2 x 24 trajectories,
                    12 steps.
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_driver as gd         # noqa: E402  (imports the reference with the compatibility aliases)
import make_golden as mg                # noqa: E402


def driver_as60():
    omega_cm, S, nac, chi = mg.synthetic_as60()
    rows = np.vstack((omega_cm, S, nac, chi)).T
    with tempfile.TemporaryDirectory() as tmp:
        model = os.path.join(tmp, "AS_model_60.dat")
        np.savetxt(model, rows)
        task = {"task": "dynamics", "potential": {"type": "anharmonic AS", "model_file": model}, "propagator": "HK",
                "batch_size": 48, "num_trajectories": 96, "num_steps": 24, "time_step_fs": 0.05, "manual_seed": 0}
        rat = {"task": "rates", "broadening": "gaussian", "hwhmG_ev": 0.01}
        data, zis, probis = gd.run_reference_task(task, rat)
    out = {f"res_{k}": v for k, v in data.items()}
    out.update(model_rows=rows, zi=np.stack(zis), probi=np.stack(probis),
               task=json.dumps({k: v for k, v in task.items() if k != "potential"}), rates_task=json.dumps(rat))
    np.savez_compressed(os.path.join(HERE, "driver_as60.npz"), **out)
    print("driver_as60: C(0) =", data["autocorrelation"][0], "trajectories", data["trajectories"], "keys", sorted(data))


def driver_as24_wm():
    """the same through the Walton-Manolopoulos propagator on the first 24 modes (D > 16), cell_width 500"""
    omega_cm, S, nac, chi = mg.synthetic_as60()
    rows = np.vstack((omega_cm, S, nac, chi)).T[:24]
    with tempfile.TemporaryDirectory() as tmp:
        model = os.path.join(tmp, "AS_model_24.dat")
        np.savetxt(model, rows)
        task = {"task": "dynamics", "potential": {"type": "anharmonic AS", "model_file": model}, "propagator": "WM", "cell_width": 500.0,
                "batch_size": 24, "num_trajectories": 48, "num_steps": 12, "time_step_fs": 0.05, "manual_seed": 0}
        rat = {"task": "rates", "broadening": "gaussian", "hwhmG_ev": 0.01}
        data, zis, probis = gd.run_reference_task(task, rat)
    out = {f"res_{k}": v for k, v in data.items()}
    out.update(model_rows=rows, zi=np.stack(zis), probi=np.stack(probis),
               task=json.dumps({k: v for k, v in task.items() if k != "potential"}), rates_task=json.dumps(rat))
    np.savez_compressed(os.path.join(HERE, "driver_as24_wm.npz"), **out)
    print("driver_as24_wm: C(0) =", data["autocorrelation"][0], "trajectories", data["trajectories"])


if __name__ == "__main__":
    driver_as60()
    driver_as24_wm()
