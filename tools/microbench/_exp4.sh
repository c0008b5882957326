mkdir -p gpurun_out/r04k
for n in pair14 pair12 base14; do
echo "== $n" >> gpurun_out/r04k/fair.txt
DDSP_HIP_LIB=$PWD/ddsp-pytorch_amd/libddsp_hip_st_$n.so timeout -k 10 200 python tools/microbench/osc_stamps.py 2>&1 | grep -v "histogram\|array\|amdgpu.ids\|^ \|per XCC\|waves per\|waves on" | cut -c 1-100,330-600 >> gpurun_out/r04k/fair.txt
DDSP_HIP_LIB=$PWD/ddsp-pytorch_amd/libddsp_hip_st_$n.so timeout -k 10 200 python tools/microbench/osc_only.py chunk 20 2>&1 | grep kernel_ms | sed 's/.*kernel_ms/kernel_ms/' >> gpurun_out/r04k/fair.txt
done
timeout -k 10 200 python tools/microbench/osc_only.py frame 20 2>&1 | grep kernel_ms | sed 's/.*kernel_ms/frame kernel_ms/' >> gpurun_out/r04k/fair.txt
cat gpurun_out/r04k/fair.txt
