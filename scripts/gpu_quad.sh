#!/bin/bash
# four-wave GEMM kernel (id 7): parity + timing against the eight-wave kernel (id 6); libmdrhip_qabl1.so = the same without its epilogue
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu 2>&1 | tail -8 | tee gpurun_out/quad_pytest.txt
bash scripts/gpu_gemm_ab.sh 20297 6 7 -- libmdrhip.so libmdrhip_qabl1.so 2>&1 | tee gpurun_out/quad_bench.txt
