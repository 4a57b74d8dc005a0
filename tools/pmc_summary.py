#!/usr/bin/env python
"""Per-kernel SQ counter summary of rocprofv3 --pmc passes -> profiles/<tag>_pmc.json

    python tools/pmc_summary.py OUT.json "comment" KERNEL_SUBSTRING[,KERNEL_SUBSTRING...] PASS_DIR [PASS_DIR ...]

Every pass directory holds one rocpd database (rocprofv3 --kernel-trace --pmc <counters> -d DIR ...).  For each kernel whose
name contains one of the substrings: launches, the mean of every counter per launch, and the derived figures
    valu_busy      = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   fraction of a wave's resident time with a VALU instruction executing
    active         = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES    ... with any instruction executing
    wait_fraction  = SQ_WAIT_ANY / SQ_WAVE_CYCLES           ... parked at s_waitcnt / barrier
    issue_stall    = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES      ... waiting to issue (dependency, pipe busy)
                     (active + wait + issue_stall ~ 1: the three are disjoint, MI355X_MICROARCH.md "rocprofv3 PMC slots")
    lds_conflict   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  All are PER-WAVE fractions: with w waves resident per SIMD the SIMD's VALU is busy ~ w * valu_busy of the time.
as far as the counters were collected (missing ones give null).  Counter semantics: MI355X_MICROARCH.md, rocprofv3 PMC slots.
"""
import glob
import json
import os
import sqlite3
import sys


def collect(directory):
    out = {}
    for f in glob.glob(os.path.join(directory, "**", "*_results.db"), recursive=True):
        con = sqlite3.connect(f)
        for name, counter, value in con.execute("select kernel_name, counter_name, value from counters_collection"):
            out.setdefault(name, {}).setdefault(counter, []).append(float(value))
    return out


def main():
    out_path, comment, needles = sys.argv[1], sys.argv[2], sys.argv[3].split(",")
    merged = {}
    for d in sys.argv[4:]:
        for k, cs in collect(d).items():
            for c, v in cs.items():
                merged.setdefault(k, {}).setdefault(c, []).extend(v)
    res = {"comment": comment, "kernels": {}}
    for k, cs in merged.items():
        if not any(nd in k for nd in needles):
            continue
        mean = {c: sum(v) / len(v) for c, v in cs.items()}
        g = lambda c: mean.get(c)
        ratio = lambda a, b, f=1.0: (f * g(a) / g(b)) if g(a) is not None and g(b) else None
        short = k.replace("void ", "").replace("(anonymous namespace)::", "")
        short = short.split("(StepArgs")[0].split("(WmArgs")[0].split("((anonymous")[0]
        res["kernels"][short] = {
            "launches": max(len(v) for v in cs.values()),
            "counters_mean_per_launch": mean,
            "valu_busy": ratio("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"),
            "wait_fraction": ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"),
            "issue_stall_fraction": ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
            "active_fraction": ratio("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"),
            "lds_conflict_fraction": ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
            "valu_insts_per_wave": ratio("SQ_INSTS_VALU", "SQ_WAVES"),
            "lds_insts_per_wave": ratio("SQ_INSTS_LDS", "SQ_WAVES"),
            "wave_cycles_per_busy_cycle": ratio("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"),
        }
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    main()
