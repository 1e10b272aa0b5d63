#!/bin/bash
# usage (GPU box, via gpurun): tools/pmc_overlap.sh <tag> <script> [args...]   -> gpurun_out/pmc_<tag>_3
# VALU / matrix-pipe overlap counters of the encode (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles,
# SQ_VALU_MFMA_COEXEC_CYCLES the cycles in which vector and matrix instructions execute together).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_3 -- python3 $R/$@ > $R/gpurun_out/pmc_${tag}_3.log 2>&1
