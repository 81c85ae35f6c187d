#!/bin/bash
set -e -o pipefail
python3 -m diffews_amd.build > /dev/null
export DFW_NO_BUILD=1
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
rm -rf gpurun_out/prof_f32
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f32 -o run -- python3 bench.py --inline --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --no-graph --dtype fp16 --residual-dtype fp32 > gpurun_out/prof_f32.json 2> gpurun_out/prof_f32.log
