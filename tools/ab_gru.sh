#!/bin/bash
# Same-box A/B of two libddsp_hip.so builds on the GRU recurrence (interleaved rounds): box-to-box spread is +-10 %
# and the backward's step time moves by more than that with the polling phase, so only same-box numbers compare.
# usage: tools/ab_gru.sh path/to/libA.so path/to/libB.so [rounds]      (run through gpurun; build the two copies first)
A=$1; B=$2; R=${3:-2}
for r in $(seq 1 $R); do
  for L in $A $B; do
    echo "== $L"
    DDSP_HIP_LIB=$PWD/$L timeout -k 10 200 python tools/microbench/gru_time.py 2>/dev/null | grep -E "32, 500, 1024|recurrence"
  done
done
