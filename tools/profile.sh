#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   tools/profile.sh r01
# writes gpurun_out/profile_<tag>/..., and the judged summaries into profiles/<tag>_*.
# Counters go in their own passes (never combined with trace domains other than kernel-trace).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/profile_$TAG
mkdir -p $O $R/profiles
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --cpu-seconds 0"   # default --steps / --warmup = the judged command minus the CPU leg

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/stats.log 2>&1 || echo "stats pass failed"
for P in "WRITE_SIZE" "FETCH_SIZE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  N=$(echo $P | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $O/pmc_$N -- $BENCH > $O/pmc_$N.log 2>&1 || echo "pmc pass $N failed"
done
# the other BASELINE configs (compiled circuit kernels, sum chain): kernel-trace only
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/configs -- python3 $R/tools/configs_bench.py --rounds 2 > $O/configs.log 2>&1 || echo "configs pass failed"
python3 $R/tools/profile_summary.py $O $R/profiles $TAG   # (on the GPU box this lands in the scratch copy: run it again here on the merged gpurun_out/)
