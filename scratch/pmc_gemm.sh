#!/bin/bash
# Counter passes over one GEMM launch shape (GPU box, repo root): bash scratch/pmc_gemm.sh lin|conv
set -e -o pipefail
K=${1:-lin}
REPO=$(pwd)
export DFW_NO_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
rm -rf gpurun_out/pmc_g_a gpurun_out/pmc_g_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d gpurun_out/pmc_g_a -o runc -- python3 scratch/pmc_gemm.py $K > gpurun_out/pmc_g_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d gpurun_out/pmc_g_b -o runc -- python3 scratch/pmc_gemm.py $K > gpurun_out/pmc_g_b.log 2>&1
for p in a b; do
  f=$(find gpurun_out/pmc_g_$p -name "*counter_collection.csv" | head -1)
  python3 scratch/pmc_sum.py "$f" gemm_big
done
