mkdir -p gpurun_out/r03d
./tools/microbench/int_mfma_rates.bin 2>&1 | grep "same wave" > gpurun_out/r03d/mfma_int.txt
for v in "" _abl1 _abl2 _abl3 _abl4; do
  DDSP_HIP_LIB=$PWD/ddsp-pytorch_amd/libddsp_hip$v.so python - <<'PY' >> gpurun_out/r03d/abl.txt 2>/dev/null
import os, sys, json
sys.argv=['x']
sys.path.insert(0, 'tools/microbench')
import noise128_ab as ab, numpy as np, torch
from ddsp_pytorch_amd import synthetic as syn
rng = np.random.default_rng(1)
H = torch.from_numpy(syn.controller_range(rng.standard_normal((512, 500, 65), dtype=np.float32))).cuda()
y = torch.zeros(512, 500 * 128, device="cuda")
r = [round(ab.run(H, y, 0, True, None)[0], 4) for _ in range(3)]
print(os.environ['DDSP_HIP_LIB'].split('/')[-1], r)
PY
done
cat gpurun_out/r03d/mfma_int.txt gpurun_out/r03d/abl.txt
