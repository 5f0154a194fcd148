mkdir -p gpurun_out/sweep_kt
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sweep_kt -- python3 tools/sweep_bench.py 16 > gpurun_out/sweep_kt/log.txt 2>&1 || exit 1
grep prefetch gpurun_out/sweep_kt/log.txt
