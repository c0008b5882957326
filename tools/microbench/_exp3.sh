set -u
OUT=gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for P in frame chunk; do
python3 tools/microbench/osc_only.py $P 10 >> $OUT/plain.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1_$P -- python3 tools/microbench/osc_only.py $P 3 > /dev/null 2> $OUT/pmc1_$P.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc2_$P -- python3 tools/microbench/osc_only.py $P 3 > /dev/null 2> $OUT/pmc2_$P.err
done
