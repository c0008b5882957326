#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats and the HBM-traffic PMC passes of bench.py,
# then tools/summarise_profiles.py condenses them into profiles/<round>_*.  Usage: tools/collect_profiles.sh r03
set -u
ROUND=${1:-r01}
OUT=gpurun_out/profiles_$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-live-pmc > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
# PMC passes: counters only with --kernel-trace (separate runs; FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-live-pmc --no-calibrate > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-live-pmc --no-calibrate > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-live-pmc --no-calibrate > /dev/null 2> "$OUT/pmc_sq.err"
# (summarise first so that the plain run below finds a PMC file stamped with these very sources -> `roofline.traffic` in its line)
python3 tools/summarise_profiles.py "$OUT" "$ROUND" > /dev/null 2>&1
cp "$OUT/${ROUND}_pmc.json" profiles/ 2>/dev/null
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
# the N > 1 path at the full shape, rehearsed on this ONE GPU (two ranks on cuda:0, gloo): correctness of the launch path, not a scaling figure
python3 bench.py --gpus 2 --backend gloo --one-device --steps 10 --warmup 3 > "$OUT/bench_two_ranks_one_gpu.json" 2> "$OUT/bench_two_ranks.err"
python3 tools/summarise_profiles.py "$OUT" "$ROUND" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
