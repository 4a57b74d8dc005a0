#!/usr/bin/env python
"""`rocprofv3 --kernel-trace --stats` summary (view top_kernels of the rocpd database) as a small CSV for profiles/.

    python tools/kernel_stats.py gpurun_out/prof_dir profiles/rN_name_kernel_stats.csv "comment: the profiled command"
"""
import csv
import glob
import os
import sqlite3
import sys


def main():
    directory, out, comment = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
    db = glob.glob(os.path.join(directory, "**", "*_results.db"), recursive=True)[0]
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(top_kernels)")]
    rows = list(con.execute("select * from top_kernels"))
    name_i = cols.index("name")
    pick = lambda *names: next(cols.index(n) for n in names if n in cols)
    calls_i, total_i, avg_i, pct_i = pick("total_calls", "calls"), pick("total_duration", "total_duration (nsec)"), \
        pick("average", "average (nsec)"), pick("percentage")
    with open(out, "w", newline="") as f:
        f.write(f'"# {comment} (durations in us; view top_kernels of the rocpd database)"\n')
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDuration_us", "Average_us", "Percentage"])
        for r in rows:
            w.writerow([r[name_i], r[calls_i], round(r[total_i], 3), round(r[avg_i], 3), r[pct_i]])       # the view is in us
    print(open(out).read()[:1500])


if __name__ == "__main__":
    main()
