#!/bin/bash
# rocprofv3 counters for the circuit compiler's kernels (tools/wave_ops.py graphs, 16384 instances x 1 s), through gpurun:
#   tools/profile_wave_ops.sh r02 "osc(k)" "fm: osc(osc*40+220)" "filter(osc)"
# Counters in their own passes (never combined with trace domains other than kernel-trace); the program itself after `--`.
set -u
TAG=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_wave_ops_pmc.md   # (copy into profiles/ after the call: only gpurun_out/ comes back from the GPU box)
echo "# rocprofv3 --pmc, circuit-compiler kernels ($TAG): tools/wave_ops.py, 16384 instances x 48000 samples per launch" > $OUT
for G in "$@"; do
  K=$(echo "$G" | tr -c 'a-zA-Z0-9' '_')
  O=$R/gpurun_out/pmc_${TAG}_$K
  rm -rf $O; mkdir -p $O
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/wave_ops.py "--only=$G" > $O/stats.log 2>&1 || echo "stats pass failed for $G"
  for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
    N=$(echo $P | cut -d" " -f1)
    timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $O/pmc_$N -- python3 $R/tools/wave_ops.py "--only=$G" > $O/pmc_$N.log 2>&1 || echo "pmc pass $N failed for $G"
  done
  echo "" >> $OUT; echo "## $G" >> $OUT; echo "" >> $OUT
  grep -h "$G" $O/stats.log | tail -1 | sed 's/^/    /' >> $OUT; echo "" >> $OUT
  F=$(ls $O/stats/*/*_kernel_stats.csv 2>/dev/null | tail -1)
  [ -n "$F" ] && { echo '```'; head -4 "$F" | cut -c1-200; echo '```'; echo; } >> $OUT
  python3 $R/tools/pmc_table.py $O dusp_jit_render 786432000 >> $OUT
done
cat $OUT
