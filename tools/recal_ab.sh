mkdir -p gpurun_out/recal
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests/test_recalibration.py -m gpu -x -q --timeout 200 --timeout-method thread -k epilogue > gpurun_out/recal/test.log 2>&1
grep -n "assert\|Error\|passed\|failed" gpurun_out/recal/test.log | tail -12
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/recal/fused -- python3 tools/recal_trace.py > gpurun_out/recal/fused.log 2>&1 || exit 1
SUGAR_RECAL_SEPARATE_STATS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/recal/sep -- python3 tools/recal_trace.py > gpurun_out/recal/sep.log 2>&1 || exit 1
grep train-mode gpurun_out/recal/fused.log gpurun_out/recal/sep.log
