#!/bin/bash
# Everything under profiles/ for one round, in one session on the GPU box:  tools/round_profiles.sh r3
set -e
cd "$(dirname "$0")/.."
tag=${1:-r4}
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
dst=gpurun_out/profiles_$tag       # summaries (small): gpurun merges gpurun_out/ back, copy them to profiles/ afterwards
mkdir -p $out $dst
# headline bench line + its kernel stats + HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes)
python3 bench.py --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench_line.err
cp $out/bench_line.json $dst/${tag}_bench_full_line.json
rocprofv3 --kernel-trace --stats -d $out/bench -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $out/bench_prof.log 2>&1
python tools/kernel_stats.py $out/bench $dst/${tag}_bench_kernel_stats.csv \
    "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs (headline configuration only)" > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o run -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o run -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs > $out/pmc_write.log 2>&1
python tools/hbm_traffic.py $out/pmc_fetch $out/pmc_write $dst/${tag}_hbm_traffic.json > /dev/null
# SQ counters of the headline kernels (two passes, as for the other configurations)
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
B="SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
rocprofv3 --kernel-trace --pmc $A -d $out/pmcA_bench -o run -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs > $out/pmcA_bench.log 2>&1
rocprofv3 --kernel-trace --pmc $B -d $out/pmcB_bench -o run -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs > $out/pmcB_bench.log 2>&1
python tools/pmc_summary.py $dst/${tag}_bench_pmc.json "headline configuration (60-mode AS, n = 1e5, HK): SQ counters of the step, modes and correlation kernels, two rocprofv3 --pmc passes" \
    hk_step_sd_kernel,hk_modes_kernel,hk_modes_multi_kernel,hk_correlate_kernel $out/pmcA_bench $out/pmcB_bench > /dev/null
echo "headline done"
tools/profile_configs.sh $tag
