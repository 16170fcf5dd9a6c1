#!/bin/bash
# rocprofv3 counters of the kernel BASELINE configs[3] runs on (8192 feedback loops, here 2 s of the 10), through gpurun:
#   tools/profile_cfg4.sh r02
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_${TAG}_cfg4
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/configs_bench.py --only cfg4_loop8192 --scale 0.2 > $O/stats.log 2>&1 || echo "stats pass failed"
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  N=$(echo $P | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $O/pmc_$N -- python3 $R/tools/configs_bench.py --only cfg4_loop8192 --scale 0.2 > $O/pmc_$N.log 2>&1 || echo "pmc pass $N failed"
done
grep -h cfg4 $O/stats.log | tail -1
F=$(ls $O/stats/*/*_kernel_stats.csv 2>/dev/null | tail -1)
[ -n "$F" ] && head -4 "$F" | cut -c1-200
python3 $R/tools/pmc_table.py $O dusp_jit_render 786432000
