#!/bin/bash
# A/B of library builds on ONE box, interleaved rounds (cdna_hip_programming.md §5.4 rule 24).
# usage: tools/ab_bench.sh libA libB [rounds]   -> per-kernel ms per round
A=$1; B=$2; R=${3:-4}
for r in $(seq 1 $R); do
  for L in $A $B; do
    DDSP_HIP_LIB=$PWD/$L python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L', 'ms/step %.3f' % d['ms_per_step'], {k:round(v,4) for k,v in d['kernel_ms'].items()})"
  done
done
