#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) of the kernels of configurations 3 and 5 -> gpurun_out/profiles_TAG/TAG_config{3,5}_hbm.json
# Raw counter means per launch (KiB as rocprofv3 reports them); the calibration of the headline pass (profiles/TAG_hbm_traffic.json:
# FETCH_SIZE x 2.000 on this image) applies to the fetch counter.
set -e
cd "$(dirname "$0")/.."
tag=${1:-r4}
export TMPDIR=/tmp
out=gpurun_out/prof_hbm_$tag
dst=gpurun_out/profiles_$tag
mkdir -p $out $dst
for c in 3 5; do
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/f$c -o run -- python3 bench.py --config $c > $out/f$c.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/w$c -o run -- python3 bench.py --config $c > $out/w$c.log 2>&1
done
python tools/pmc_summary.py $dst/${tag}_config3_hbm.json "config 3 (methylium WM, n = 1e5): FETCH_SIZE and WRITE_SIZE per launch in KiB, separate rocprofv3 --pmc passes; FETCH_SIZE x 2.000 = bytes fetched / 1024 (calibration of ${tag}_hbm_traffic.json)" \
    wm_small_kernel,wm_tail_kernel,hk_step_lin_kernel $out/f3 $out/w3 > /dev/null
python tools/pmc_summary.py $dst/${tag}_config5_hbm.json "config 5 (30-atom sGDML, n = 1e4): FETCH_SIZE and WRITE_SIZE per launch in KiB, separate rocprofv3 --pmc passes; FETCH_SIZE x 2.000 = bytes fetched / 1024" \
    gdml_stage_kernel,dense_mono,dense_prefactor $out/f5 $out/w5 > /dev/null
rm -rf $out
ls -la $dst/
