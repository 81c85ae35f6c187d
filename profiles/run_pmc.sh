#!/bin/bash
# Counter passes of one round, run ON THE GPU BOX from the repo root:  bash profiles/run_pmc.sh r03
# Extra arguments go to bench.py, e.g. the training step:  bash profiles/run_pmc.sh r03_train --train --no-graph
# Each rocprofv3 --pmc pass is its own process with --kernel-trace only (never combined with other trace
# domains); FETCH_SIZE and WRITE_SIZE do not fit one pass (MI355X_MICROARCH.md, rocprofv3 PMC slots); the SQ /
# GRBM counters for MFMA utilisation share a third pass.  The program itself follows `--` (no env/bash hop).
set -e -o pipefail
ROUND=${1:-r04}
shift || true
EXTRA="$@"
REPO=$(pwd)
python3 -m diffews_amd.build > /dev/null      # never rebuild under the profiler (DFW_NO_BUILD below)
export DFW_NO_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --no-secondary --inline $EXTRA"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${ROUND}_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${ROUND}_$C -o runc -- python3 $ARGS > gpurun_out/pmc_${ROUND}_$C.log 2>&1
  echo "pass $C done"
done
rm -rf gpurun_out/pmc_${ROUND}_SQ
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
  -d gpurun_out/pmc_${ROUND}_SQ -o runc -- python3 $ARGS > gpurun_out/pmc_${ROUND}_SQ.log 2>&1
echo "pass SQ done"
python3 profiles/collect_pmc.py gpurun_out/pmc_${ROUND}_FETCH_SIZE gpurun_out/pmc_${ROUND}_WRITE_SIZE gpurun_out/pmc_${ROUND}_SQ gpurun_out/pmc_${ROUND}_raw.json
python3 profiles/summarize_pmc.py gpurun_out/pmc_${ROUND}_raw.json gpurun_out/${ROUND}_pmc_traffic.json
