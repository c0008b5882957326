set -u
OUT=gpurun_out/r04d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for L in 0 128 512; do DDSP_OSC_CHUNK_LEN=$L python3 tools/microbench/osc_only.py chunk 10 >> $OUT/plain.txt 2>&1; done
python3 tools/microbench/osc_only.py frame 10 >> $OUT/plain.txt 2>&1
for P in frame chunk; do
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1_$P -- python3 tools/microbench/osc_only.py $P 3 > /dev/null 2> $OUT/pmc1_$P.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc2_$P -- python3 tools/microbench/osc_only.py $P 3 > /dev/null 2> $OUT/pmc2_$P.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc3_$P -- python3 tools/microbench/osc_only.py $P 3 > /dev/null 2> $OUT/pmc3_$P.err
done
python3 - <<'PY'
import csv, glob, collections, json
out={}
for d in sorted(glob.glob('gpurun_out/r04d/pmc*_*')):
    if d.endswith('.err'): continue
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name']
            if 'osc' not in k: continue
            acc[k.split('(')[0][-60:]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            out.setdefault(d.split('/')[-1].split('_')[1]+' '+k, {}).update({c: sum(x)/len(x) for c,x in v.items()})
json.dump(out, open('gpurun_out/r04d/pmc_summary.json','w'), indent=1)
for k,v in out.items(): print(k, {c: round(x) for c,x in v.items()})
PY
