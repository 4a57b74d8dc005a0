#!/usr/bin/env python
"""HBM traffic of the step kernel from two rocprofv3 PMC passes -> profiles/<tag>_hbm_traffic.json

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o runc -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o runc -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline
    python tools/hbm_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1_hbm_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md, HBM section).  The guide gives FETCH_SIZE corrections only for
the load widths it calibrated; the step kernel's 8-byte-per-lane loads of 128-byte row segments are not among them, so
the fetch counter is calibrated in-run on a launch with exactly known traffic: the prefactor-only launch of
initial_conditions() (STEP = false instantiation), which reads the 4*D*D*8*n bytes of the monodromy planes and writes
almost nothing.  The same factor is then applied to the full step kernel.
"""
import csv
import glob
import json
import os
import sqlite3
import sys


def per_kernel(directory, counter):
    out = {}
    dbs = glob.glob(os.path.join(directory, "**", "*_results.db"), recursive=True)
    for f in dbs:                                   # rocprofv3 (ROCm 7.2) default output: a rocpd sqlite database
        con = sqlite3.connect(f)
        for name, value in con.execute("select kernel_name, value from counters_collection where counter_name = ?",
                                       (counter,)):
            out.setdefault(name, []).append(float(value))
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    assert files or dbs, f"no counter output under {directory}"
    for f in files:                                 # --output-format csv
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out


def pick(d, *needles):
    for k, v in d.items():
        if all(n in k for n in needles):
            return k, v
    raise KeyError(needles)


def main():
    fetch_dir, write_dir, out_path = sys.argv[1:4]
    n, dim = 100000, 60
    fetch = per_kernel(fetch_dir, "FETCH_SIZE")
    write = per_kernel(write_dir, "WRITE_SIZE")
    # template arguments <NR, MINW, STEP, TILED, KS>: the step launches run on the tiled layout, the prefactor-only launch
    # of initial_conditions() on the row-major one; KS = 2 is the two-steps-per-visit kernel of run() (sc_hk_step_multi), whose
    # counters cover TWO time steps per launch -- everything below is per TIME STEP
    try:
        kstep, f_step = pick(fetch, "hk_step_sd_kernel", "true, true, 2>")
        _, w_step = pick(write, "hk_step_sd_kernel", "true, true, 2>")
        _, f_modes = pick(fetch, "hk_modes_multi_kernel")
        _, w_modes = pick(write, "hk_modes_multi_kernel")
        steps_per_launch = 2
    except KeyError:
        kstep, f_step = pick(fetch, "hk_step_sd_kernel", "true, true, 1>")
        _, w_step = pick(write, "hk_step_sd_kernel", "true, true, 1>")
        _, f_modes = pick(fetch, "hk_modes_kernel")
        _, w_modes = pick(write, "hk_modes_kernel")
        steps_per_launch = 1
    _, f_pref = pick(fetch, "hk_step_sd_kernel", "false, false, 1>")
    f_step, w_step, f_modes, w_modes = ([v / steps_per_launch for v in x] for x in (f_step, w_step, f_modes, w_modes))
    mean = lambda x: sum(x) / len(x)
    known = 4 * dim * dim * 8 * n
    factor = known / (mean(f_pref) * 1024.0)
    read_b = mean(f_step) * 1024.0 * factor
    write_b = mean(w_step) * 1024.0
    modes_b = mean(f_modes) * 1024.0 + mean(w_modes) * 1024.0
    alg = (64 * dim * dim + 64 * dim + 72) * n
    res = {
        "workload": {"ntraj": n, "dim": dim, "kernel": kstep.replace("void ", "").replace("(anonymous namespace)::", "").split("(StepArgs")[0],
                     "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two separate passes) -- "
                                "python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline"},
        "steps_per_launch": steps_per_launch,
        "FETCH_SIZE_KiB_per_launch": mean(f_step), "WRITE_SIZE_KiB_per_launch": mean(w_step),
        "launches_averaged": len(f_step),
        "fetch_calibration": {"known_bytes": known, "counter_bytes": mean(f_pref) * 1024.0, "factor": factor,
                              "note": "calibrated on the prefactor-only launch (STEP=false) of initial_conditions, which "
                                      "reads exactly the 4*D*D*8*n bytes of the monodromy planes"},
        "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b,
        "modes_kernel_bytes_per_launch_uncalibrated": modes_b,
        "traffic_bytes_per_launch": read_b + write_b + modes_b,
        "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": (read_b + write_b + modes_b) / alg,
        "note": "step kernel + its modes pre-pass (row propagators: 4*D*8 B written and read back per trajectory).  All 'per_launch' "
                "figures are per TIME STEP: with steps_per_launch = 2 (sc_hk_step_multi) the counters of a launch were halved.  "
                "FETCH_SIZE counts what the L2 requests from the fabric; the second sub-step's reads are served by the memory-side "
                "cache (tools/micro/revisit.hip), so DRAM reads are lower than this figure",
    }
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
