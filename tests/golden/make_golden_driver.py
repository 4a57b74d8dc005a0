#!/usr/bin/env python
"""Golden vectors for rows N3 / N4 of SURVEY.md section 8f, produced by running the REFERENCE itself (build container only).

    python tests/golden/make_golden_driver.py

  readers_ref.npz       what the reference's fchk reader returns for the four fchk files its tests hold (copied as DATA
                        to tests/golden/fchk/): pos0, energy0, grad0, hess0, masses, nac, total energy, and
                        vibrational_groundstate() -> x0, Gamma_0, E_zpt; for coumarin also the `Vib-E2` frequencies the
                        reference's own test pins the reader with (tests/test_readers.py:21-46).
  driver_methylium.npz  the reference's `semi dynamics` + `semi rates` on its own example task
                        tests/DATA/examples/methylium_AH/semi.json (harmonic potential, three fchk files), with a small
                        num_trajectories / num_steps: the initial conditions zi, probi of every repetition, and every key
                        of the correlations.npz it writes (cli.py:171-476, 519-570).
  driver_gdml.npz       the same for a "gdml" task on the coumarin model (cli.py:204-227): the reference's outcome (the
                        exception its Newton minimisation raises, or its results).
`ase` is not installed here: tests/golden/ase_stub.py stands in for the calls the reference makes.  Only arrays and
messages are stored, nothing of the reference's source.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

torch.set_default_dtype(torch.float64)
torch.symeig = lambda A, eigenvectors=True, upper=True: tuple(torch.linalg.eigh(A, UPLO='U' if upper else 'L'))
torch.solve = lambda B, A: (torch.linalg.solve(A, B), None)

import ase_stub  # noqa: E402
ase_stub.install()

import logging  # noqa: E402
logging.disable(logging.CRITICAL)

from semiclassical import cli, readers, propagators  # noqa: E402

FCHK = os.path.join(HERE, "fchk")


def reader_golden():
    out = {}
    for name in ("methylium_s0", "methylium_s1", "coumarin_s0", "coumarin_s1"):
        with open(os.path.join(FCHK, name + ".fchk")) as f:
            fchk = readers.FormattedCheckpointFile(f)
        pos0, energy0, grad0, hess0 = fchk.harmonic_approximation()
        x0, Gamma_0, en_zpt = fchk.vibrational_groundstate()
        out.update({f"{name}_pos0": pos0, f"{name}_energy0": energy0, f"{name}_grad0": grad0, f"{name}_hess0": hess0,
                    f"{name}_masses": fchk.masses(), f"{name}_total_energy": fchk.total_energy(),
                    f"{name}_atomic_numbers": np.asarray(fchk.atomic_numbers()),
                    f"{name}_x0": x0, f"{name}_Gamma_0": Gamma_0, f"{name}_en_zpt": en_zpt})
        try:
            out[f"{name}_nac"] = fchk.nonadiabatic_coupling()
        except Exception:                                   # the S0 files carry no coupling vector
            pass
        if name.startswith("coumarin"):
            nmodes = fchk["Number of Normal Modes"]
            out[f"{name}_vib_e2"] = np.asarray(fchk["Vib-E2"][:nmodes])
    np.savez_compressed(os.path.join(HERE, "readers_ref.npz"), **out)
    print("readers_ref.npz:", len(out), "arrays")


def run_reference_task(task, rates_task=None):
    """run the reference driver on `task`; returns (npz contents, [zi per repetition], [probi per repetition])"""
    zis, probis = [], []
    orig = propagators.HermanKlukPropagator.initial_conditions

    def recording(self, *args, **kwargs):
        r = orig(self, *args, **kwargs)
        zis.append(self.zi.numpy().copy())
        probis.append(self.probi.numpy().copy())
        return r
    propagators.HermanKlukPropagator.initial_conditions = recording
    try:
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "correlations.npz")
            task = json.loads(json.dumps(task))
            task["results"] = {"correlations": out}
            cli.run_semiclassical_dynamics(task, device="cpu")
            if rates_task is not None:
                cli.calculate_rates(dict(rates_task, correlations=out, rates=out))
            data = dict(np.load(out))
    finally:
        propagators.HermanKlukPropagator.initial_conditions = orig
    return data, zis, probis


def driver_methylium():
    with open(os.path.join(REF, "tests/DATA/examples/methylium_AH/semi.json")) as f:
        config = json.load(f)
    dyn, rat = config["semi"]
    dyn["potential"] = {"type": "harmonic", "ground": os.path.join(FCHK, "methylium_s0.fchk"),
                        "excited": os.path.join(FCHK, "methylium_s1.fchk"),
                        "coupling": os.path.join(FCHK, "methylium_s1.fchk")}
    out = {}
    for prop in ("HK", "WM"):
        task = dict(dyn, propagator=prop, batch_size=48, num_trajectories=96, num_steps=30)      # 2 repetitions
        data, zis, probis = run_reference_task(task, rat)
        for k, v in data.items():
            out[f"{prop}_{k}"] = v
        out[f"{prop}_zi"] = np.stack(zis)
        out[f"{prop}_probi"] = np.stack(probis)
        out[f"{prop}_task"] = json.dumps({k: v for k, v in task.items() if k not in ("potential", "results")})
        print(prop, "C(0) =", data["autocorrelation"][0], "trajectories", data["trajectories"], "keys", sorted(data))
    out["rates_task"] = json.dumps(rat)
    np.savez_compressed(os.path.join(HERE, "driver_methylium.npz"), **out)


def driver_gdml():
    model = os.path.join(HERE, "gdml_coumarin_model.npz")
    task = {"task": "dynamics",
            "potential": {"type": "gdml", "ground": model, "excited": os.path.join(FCHK, "coumarin_s1.fchk"),
                          "coupling": os.path.join(FCHK, "coumarin_s1.fchk")},
            "propagator": "HK", "batch_size": 8, "num_trajectories": 8, "num_steps": 3, "time_step_fs": 0.005,
            "manual_seed": 0}
    out = {"task": json.dumps({k: v for k, v in task.items() if k != "potential"})}
    try:
        data, zis, probis = run_reference_task(task)
        out.update({f"res_{k}": v for k, v in data.items()})
        out["zi"], out["probi"] = np.stack(zis), np.stack(probis)
        out["outcome"] = "ok"
    except Exception as err:          # whatever the reference does on this input is what the driver has to do as well
        out["outcome"] = type(err).__name__
        out["message"] = str(err)
    print("gdml task:", out["outcome"], out.get("message", ""))
    np.savez_compressed(os.path.join(HERE, "driver_gdml.npz"), **out)


if __name__ == "__main__":
    reader_golden()
    driver_methylium()
    driver_gdml()
