#!/bin/bash
# Refresh profiles/pmc_traffic.json for the current csrc/mdr_mips* hash: FETCH_SIZE passes of all four screen kernels at 5M rows.
cd $GRAFT_REPO_ROOT
PMC_FETCH_ONLY=1 PMC_PROFILE_TAG=r03_mips5m_i8_pmc bash scripts/measure/gpu_pmc_screen.sh r3pmc_i8 > gpurun_out/r3pmc_i8.log 2>&1
cp gpurun_out/r3pmc_i8/pmc_traffic.json profiles/pmc_traffic.json
MDR_MIPS_I8=0 PMC_FETCH_ONLY=1 PMC_PROFILE_TAG=r03_mips5m_pmc bash scripts/measure/gpu_pmc_screen.sh r3pmc_f16 > gpurun_out/r3pmc_f16.log 2>&1
cp gpurun_out/r3pmc_f16/pmc_traffic.json gpurun_out/pmc_traffic_final.json
tail -n 5 gpurun_out/r3pmc_i8.log; tail -n 5 gpurun_out/r3pmc_f16.log
