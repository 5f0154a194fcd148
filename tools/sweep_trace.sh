mkdir -p gpurun_out/sweep_kt2
export TMPDIR=/tmp
SWEEP_RECAL=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sweep_kt2 -- python3 tools/sweep_bench.py 8 > gpurun_out/sweep_kt2/log.txt 2>&1 || exit 1
grep prefetch gpurun_out/sweep_kt2/log.txt
