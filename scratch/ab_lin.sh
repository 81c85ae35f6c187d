#!/bin/bash
# token-GEMM sweep + headline bench on one box
set -e
ONLY=lin timeout -k 10 400 python scratch/bench_stages.py > gpurun_out/r04_stages_b.log 2>&1
timeout -k 10 400 python bench.py --inline --steps 20 --warmup 3 > gpurun_out/r04_s2_bench.json 2> gpurun_out/r04_s2_bench.log
