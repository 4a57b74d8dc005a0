set -e
cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/prof_x; dst=gpurun_out/profiles_x
mkdir -p $out $dst
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
B="SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
C="SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F64"
rocprofv3 --kernel-trace --pmc $A -d $out/pmcA3 -o run -- python3 bench.py --config 3 > $out/pmcA3.log 2>&1
rocprofv3 --kernel-trace --pmc $B -d $out/pmcB3 -o run -- python3 bench.py --config 3 > $out/pmcB3.log 2>&1
python tools/pmc_summary.py $dst/x_wm_pmc.json "config 3" wm_small_kernel,wm_tail_kernel,hk_step_lin_kernel $out/pmcA3 $out/pmcB3
rm -rf $out
