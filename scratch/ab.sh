#!/bin/bash
# Same-box A/B of library builds: bash scratch/ab.sh "python scratch/bench_attn_bwd.py" scratch/ab/libA.so scratch/ab/libB.so ...
# (each build twice, alternating, so that box-to-box and warm-up differences cancel)
CMD=$1; shift
for rep in 1 2; do
  for L in "$@"; do
    echo "== $L (rep $rep)"
    DFW_LIB=$PWD/$L DFW_NO_BUILD=1 timeout -k 10 150 $CMD 2>&1 | grep -v amdgpu.ids
  done
done
