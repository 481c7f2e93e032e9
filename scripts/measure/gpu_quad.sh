#!/bin/bash
# hand-scheduled GEMM kernels (test-hook ids 5 = duo, 7 = quad): parity + timing against the eight-wave kernel (id 6)
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k "duo" 2>&1 | tail -12 | tee gpurun_out/duo_pytest.txt
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu 2>&1 | tail -12 | tee -a gpurun_out/duo_pytest.txt
timeout 300 python scripts/measure/gpu_gemm_bench.py 20297 6 5 7 6 5 7 2>&1 | grep -v amdgpu | tee gpurun_out/duo_bench.txt
