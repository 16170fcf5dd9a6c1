mkdir -p gpurun_out/r04
python -X faulthandler bench.py --config cfg5 --cpu-seconds 0 --gather > gpurun_out/r04/bench_line_cfg5.json 2> gpurun_out/r04/bench_line_cfg5.err; echo "rc $?"; tail -30 gpurun_out/r04/bench_line_cfg5.err; wc -c gpurun_out/r04/bench_line_cfg5.json
python -X faulthandler bench.py --config cfg5 --cpu-seconds 0 > gpurun_out/r04/bench_line_cfg5_nogather.json 2> gpurun_out/r04/bench_line_cfg5_ng.err; echo "rc $?"; tail -5 gpurun_out/r04/bench_line_cfg5_ng.err; wc -c gpurun_out/r04/bench_line_cfg5_nogather.json
