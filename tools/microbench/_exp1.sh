mkdir -p gpurun_out/r04c
for L in 0 128 256 512 1024 1344 1376 1408; do
  echo "== Lc=$L" >> gpurun_out/r04c/lc.txt
  DDSP_OSC_CHUNK_LEN=$L timeout -k 10 200 python - >> gpurun_out/r04c/lc.txt 2>&1 <<'PY'
import os, sys, json, numpy as np, torch
os.environ["DDSP_TEST_HOOKS"]="1"
sys.path.insert(0, os.getcwd())
import ddsp_pytorch_amd as ddsp
from ddsp_pytorch_amd import synthetic as syn
shape=syn.CFG4_PER_GPU
ctl=syn.make_controls(shape,1004,"all_live")
x={k: torch.from_numpy(v).cuda() for k,v in ctl.items() if k!="H"}
def run(): return ddsp.osc_forward(x["f0"],x["c"],x["a"],shape.hop,shape.sample_rate)[0]
for _ in range(3): run()
ddsp._lib.profile_enable(400); torch.cuda.synchronize()
for _ in range(20): run()
torch.cuda.synchronize()
rec={}
for name,ms in ddsp._lib.profile_read(): rec.setdefault(name,[]).append(ms)
print({k: round(float(np.mean(v)),4) for k,v in rec.items()})
PY
done
