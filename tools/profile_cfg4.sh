#!/bin/bash
# rocprofv3 evidence for the kernel BASELINE configs[3] runs on (8192 feedback loops x 10 s: `bench.py --config cfg4`, full size), through gpurun:
#   tools/profile_cfg4.sh r04
# kernel-trace --stats first, then the counters in passes of their own (FETCH_SIZE and WRITE_SIZE do not fit one pass); the program itself after `--`.
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_${TAG}_cfg4
rm -rf $O; mkdir -p $O $R/profiles
BENCH="python3 $R/bench.py --config cfg4 --steps 3 --warmup 1 --cpu-seconds 0"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/stats.log 2>&1 || echo "stats pass failed"
for P in "WRITE_SIZE" "FETCH_SIZE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  N=$(echo $P | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $O/pmc_$N -- $BENCH > $O/pmc_$N.log 2>&1 || echo "pmc pass $N failed"
done
F=$(ls $O/stats/*/*_kernel_stats.csv 2>/dev/null | tail -1)
[ -n "$F" ] && cp "$F" $R/gpurun_out/${TAG}_cfg4_kernel_stats.csv
{
  echo "# configs[3] on its compiled kernel ($TAG): \`bench.py --config cfg4 --steps 3 --warmup 1\` under rocprofv3 --pmc, passes of their own (tools/profile_cfg4.sh, tools/pmc_table.py)"
  echo "# per launch: 8192 instances x 480 000 samples = 3 932 160 000 samples = 15 360 000 wavefront-chunks; algorithmic bytes 4 B x that = 15.729 GB written, 0 read"
  echo
  grep -h '"metric"' $O/stats.log | tail -1 | python3 -c "import sys,json; l=json.loads(sys.stdin.read()); print('bench line of the --kernel-trace pass: kernel_ms %.4f, frac %.4f, shape %s' % (l['roofline']['kernel_ms'], l['roofline']['frac'], l['config']['shape']))" 2>/dev/null
  [ -n "$F" ] && { echo; echo '```'; head -3 "$F" | cut -c1-220; echo '```'; }
  echo
  python3 $R/tools/pmc_table.py $O dusp_jit_render 3932160000 $R/gpurun_out/traffic_${TAG}_cfg4.json
} > $R/gpurun_out/${TAG}_cfg4_pmc.md
cat $R/gpurun_out/${TAG}_cfg4_pmc.md
