#!/bin/bash
# round 3, GPU call C: pipelined GEMM main loop - parity tests, then A/B against the round-2 loop (-DGEMM_PIPE=0)
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py tests/test_gpu_end_to_end.py -x -q > "$OUT/c_tests.log" 2>&1; echo "tests rc=$?"; tail -4 "$OUT/c_tests.log"
ROUNDS=2 WLS="cfg2 reference" bash tools/gpu_ab.sh nopipe tree 2>&1 | tee "$OUT/c_ab.log"
