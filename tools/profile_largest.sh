# BASELINE configs[2] (Largest subnet, GLOBAL batch 512) per-GPU shares on one MI355X: bench lines + kernel traces for the
# per-GPU batches of 1 / 2 / 4 / 8 ranks (512 / 256 / 128 / 64 utterances).  Output: gpurun_out/r3l; summaries are made by
# tools/kt_md.py and committed under profiles/.
set -e
mkdir -p gpurun_out/r3l
export TMPDIR=/tmp
for G in 512 256 128 64; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --workload largest --global-batch $G --no-cpu-baseline --stages \
    > gpurun_out/r3l/bench_largest_g$G.json 2> gpurun_out/r3l/bench_largest_g$G.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3l/kt_g$G -- \
    python3 bench.py --steps 20 --warmup 5 --workload largest --global-batch $G --no-cpu-baseline --no-pipelined \
    > gpurun_out/r3l/kt_g$G.log 2>&1
done
find gpurun_out/r3l -name "*_kernel_stats.csv"
