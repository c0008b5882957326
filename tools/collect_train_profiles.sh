#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace of the autocast training step + LDS counters of the spectral-loss scales.
# Usage: tools/collect_train_profiles.sh r03   -> gpurun_out/train_profiles_<round>/{train_amp_kernel_stats.csv, mss_pmc.json}
set -u
ROUND=${1:-r03}
OUT=gpurun_out/train_profiles_$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --mode train --amp bf16 --steps 5 --warmup 2 > "$OUT/train.json" 2> "$OUT/trace.err"
python3 tools/summarise_train_profile.py "$OUT/trace" 7 "rocprofv3 --kernel-trace --stats -- python3 bench.py --mode train --amp bf16 --steps 5 --warmup 2" > "$OUT/train_amp_kernel_stats.csv"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_lds" -- python3 bench.py --mode train --amp bf16 --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_lds.err"
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_lds", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for key in ("mss_wave2048_kernel", "mss_wave_kernel", "stft_frames_bwd_kernel", "gru_fwd", "gru_bwd", "colsum"):
            if key in n:
                short = n[n.index(key):].split("(")[0]
                acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["conflict_share_of_lds_active"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
    res[k] = d
json.dump(res, open(os.path.join(out, "mss_pmc.json"), "w"), indent=1, sort_keys=True)
for k, d in res.items():
    if "mss" in k:
        print(k, {c: round(v, 3) for c, v in d.items()})
PY
head -30 "$OUT/train_amp_kernel_stats.csv" | cut -c1-170
