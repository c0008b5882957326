#!/bin/bash
# Same box, one process per point: the oscillator alone, then oscillator + noise with the hop-128 noise kernel at 8 / 7 / 6 / 5 / 4
# wavefronts per CU (DDSP_NOISE_WAVES tuning hook).  Prints the synth kernel's sustained time, the shader clock and bench.py's step.
export DDSP_TEST_HOOKS=1
echo "== oscillator only"; python tools/microbench/clock_ramp.py osc 2>&1 | grep "mean of"
for w in 8 7 6 5 4; do
  echo "== noise at $w wavefronts per CU"
  DDSP_NOISE_WAVES=$w python tools/microbench/clock_ramp.py noise 2>&1 | grep "mean of"
  DDSP_NOISE_WAVES=$w python bench.py --steps 20 --warmup 5 --no-live-pmc --no-secondary --no-cpu-baseline --no-calibrate 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms/step %.4f from idle %.4f clock %.3f' % (d['ms_per_step'], d['clock_settle']['from_idle']['ms_per_step'], d['clock_ghz']), {k: round(v,4) for k,v in d['kernel_ms'].items()})"
done
