#!/bin/bash
# fp32 verification mode: its tests only, measurements into gpurun_out/measured_errors.json
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fp32_verify" > gpurun_out/verify32_test.log 2>&1; rc=$?
tail -30 gpurun_out/verify32_test.log
cat gpurun_out/measured_errors.json 2>/dev/null | head -80
exit $rc
