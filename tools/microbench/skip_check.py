import sys, torch, numpy as np
sys.path.insert(0,'.')
import ddsp_pytorch_amd as ddsp
from ddsp_pytorch_amd import synthetic as syn
shape=syn.CFG4_PER_GPU
ctl=syn.make_controls(shape,1004,"all_live")
for f0v in (60.0, 200.0, 1000.0):
    f0=torch.full((512,500,1),f0v,device='cuda')
    c=torch.from_numpy(ctl["c"]).cuda(); a=torch.from_numpy(ctl["a"]).cuda()
    for _ in range(2): ddsp.osc_forward(f0,c,a,128,16000)
    ddsp._lib.profile_enable(64); torch.cuda.synchronize()
    for _ in range(5): ddsp.osc_forward(f0,c,a,128,16000)
    torch.cuda.synchronize()
    rec={}
    for n,ms in ddsp._lib.profile_read(): rec.setdefault(n,[]).append(ms)
    ddsp._lib.profile_enable(0)
    print(f0v, {k: round(float(np.mean(v)),4) for k,v in rec.items()})
