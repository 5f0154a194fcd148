set -e
mkdir -p gpurun_out/r3p
export TMPDIR=/tmp
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r3p/bench_base256.json 2> gpurun_out/r3p/bench_base256.err
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --workload largest256 --no-cpu-baseline > gpurun_out/r3p/bench_largest256.json 2>/dev/null
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --workload wide256 --no-cpu-baseline > gpurun_out/r3p/bench_wide256.json 2> gpurun_out/r3p/bench_wide256.err
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --precision fp32 --no-cpu-baseline > gpurun_out/r3p/bench_base256_fp32.json 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3p/kt -- $B > gpurun_out/r3p/kt.log 2>&1
P="python3 bench.py --steps 3 --warmup 1 --preheat 0 --no-cpu-baseline --no-pipelined"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r3p/fetch -- $P > gpurun_out/r3p/fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r3p/write -- $P > gpurun_out/r3p/write.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/r3p/mfma -- $P > gpurun_out/r3p/mfma.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/r3p/mfma_wide -- $P --workload wide256 > gpurun_out/r3p/mfma_wide.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d gpurun_out/r3p/lds -- $P > gpurun_out/r3p/lds.log 2>&1
find gpurun_out/r3p -name "*.csv" | head -40
du -sh gpurun_out/r3p
