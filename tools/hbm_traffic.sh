#!/bin/bash
# HBM traffic of the encode kernels from PMC counters (separate passes, as MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Run on the GPU box via gpurun:
#   tools/hbm_traffic.sh   -> gpurun_out/pmc_fetch, gpurun_out/pmc_write
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/tools/exp_time.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/tools/exp_time.py > /dev/null 2>&1
