import sys, torch, numpy as np
sys.path.insert(0,'.')
import ddsp_pytorch_amd as ddsp
from ddsp_pytorch_amd import synthetic as syn
shape=syn.CFG4_PER_GPU
ctl=syn.make_controls(shape,1004,"all_live")
f0=torch.full((512,500,1),1000.0,device='cuda')
c=torch.from_numpy(ctl["c"]).cuda(); a=torch.from_numpy(ctl["a"]).cuda()
for _ in range(3): ddsp.osc_forward(f0,c,a,128,16000)
torch.cuda.synchronize()
