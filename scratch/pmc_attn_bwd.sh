#!/bin/bash
# Counter passes over the attention backward kernels (run on the GPU box from the repo root): bash scratch/pmc_attn_bwd.sh TAG
set -e -o pipefail
TAG=${1:-bwd}
REPO=$(pwd)
export DFW_NO_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
rm -rf gpurun_out/pmc_${TAG}_a gpurun_out/pmc_${TAG}_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_a -o runc -- python3 scratch/pmc_attn_bwd.py > gpurun_out/pmc_${TAG}_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_b -o runc -- python3 scratch/pmc_attn_bwd.py > gpurun_out/pmc_${TAG}_b.log 2>&1
for p in a b; do
  f=$(find gpurun_out/pmc_${TAG}_$p -name "*counter_collection.csv" | head -1)
  python3 scratch/pmc_sum.py "$f" fsa_bwd > gpurun_out/pmc_${TAG}_$p.txt
  cat gpurun_out/pmc_${TAG}_$p.txt
done
