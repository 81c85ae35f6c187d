#!/bin/bash
# same-box A/B of two library builds on the headline bench (alternating, twice)
for rep in 1 2; do
  for L in "$@"; do
    n=$(basename $L .so)
    DFW_LIB=$PWD/$L DFW_NO_BUILD=1 timeout -k 10 300 python bench.py --inline --steps 20 --warmup 3 > gpurun_out/ab_${n}_$rep.json 2> gpurun_out/ab_${n}_$rep.log || exit 1
    python -c "import json,sys; d=json.load(open('gpurun_out/ab_${n}_$rep.json')); print('$n', $rep, d['ms_per_step'])"
  done
done
