#!/bin/bash
# One rocprofv3 kernel-stats summary per BASELINE configuration + SQ counter passes for the small-matrix kernels (GPU box).
#   tools/profile_configs.sh TAG       -> gpurun_out/profiles_TAG/TAG_config{1,3,5}_kernel_stats.csv, TAG_wm_pmc.json, ...
# Counters are collected in their own runs (--kernel-trace + --pmc only), as the pool requires.
set -e
cd "$(dirname "$0")/.."
tag=${1:-r4}
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
dst=gpurun_out/profiles_$tag       # summaries (small): gpurun merges gpurun_out/ back, copy them to profiles/ afterwards
mkdir -p $out $dst
for c in 1 3 3hk 5; do
    rocprofv3 --kernel-trace --stats -d $out/c$c -o run -- python3 bench.py --config $c > $out/c$c.log 2>&1
    python tools/kernel_stats.py $out/c$c $dst/${tag}_config${c}_kernel_stats.csv \
        "rocprofv3 --kernel-trace --stats -- python3 bench.py --config $c (ONE configuration per run: averages are not mixed over batch sizes)" > /dev/null
    echo "config $c done"
done
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
B="SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
for c in 1 3 3hk 5; do
    rocprofv3 --kernel-trace --pmc $A -d $out/pmcA$c -o run -- python3 bench.py --config $c > $out/pmcA$c.log 2>&1
    rocprofv3 --kernel-trace --pmc $B -d $out/pmcB$c -o run -- python3 bench.py --config $c > $out/pmcB$c.log 2>&1
    echo "pmc $c done"
done
python tools/pmc_summary.py $dst/${tag}_wm_pmc.json "config 3 (methylium WM, n = 1e5): SQ counters of the WM kernels and the HK step kernel, two rocprofv3 --pmc passes" \
    wm_small_kernel,wm_tail_kernel,hk_step_lin_kernel $out/pmcA3 $out/pmcB3 > /dev/null
python tools/pmc_summary.py $dst/${tag}_config3hk_pmc.json "methylium HK, n = 1e5, run() = ONE launch of hk_run_lin_kernel (whole caller loop in registers): SQ counters, two rocprofv3 --pmc passes" \
    hk_run_lin_kernel,hk_step_lin_kernel,hk_correlate $out/pmcA3hk $out/pmcB3hk > /dev/null
python tools/pmc_summary.py $dst/${tag}_config1_pmc.json "config 1 (5-mode AS, n = 1e5): SQ counters, two rocprofv3 --pmc passes" \
    hk_step_sep16_kernel,hk_correlate_kernel $out/pmcA1 $out/pmcB1 > /dev/null
python tools/pmc_summary.py $dst/${tag}_config5_pmc.json "config 5 (30-atom sGDML, n = 1e4): SQ counters, two rocprofv3 --pmc passes" \
    gdml_stage_kernel,dense_mono,dense_prefactor $out/pmcA5 $out/pmcB5 > /dev/null
rm -rf $out          # raw rocpd databases: tens of MB, not needed once summarised
ls -la $dst/
