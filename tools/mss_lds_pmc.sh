#!/bin/bash
# LDS bank-conflict counters of the spectral-loss scale kernels for the library in $DDSP_HIP_LIB (default: the in-tree one).
# Usage (GPU box): tools/mss_lds_pmc.sh <tag>   -> gpurun_out/mss_lds_<tag>/summary.txt
set -u
TAG=${1:-default}
OUT=gpurun_out/mss_lds_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d "$OUT/pmc" -- python3 tools/microbench/mss_scale_time.py > "$OUT/time.json" 2> "$OUT/pmc.err"
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import collections, csv, glob, os, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(os.path.join(out, "pmc", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "mss_wave" in n:
            short = n[n.index("mss_wave"):].split("(")[0]
            acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(acc):
    d = acc[k]
    print(k, "conflict cycles / LDS-active cycles = %.3f" % (d["SQ_LDS_BANK_CONFLICT"] / max(1.0, d["SQ_LDS_IDX_ACTIVE"])),
          "conflicts per LDS instruction = %.2f" % (d["SQ_LDS_BANK_CONFLICT"] / max(1.0, d["SQ_INSTS_LDS"])))
PY
