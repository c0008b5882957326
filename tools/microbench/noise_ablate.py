import sys, torch, numpy as np
sys.path.insert(0,'.')
import ddsp_pytorch_amd as ddsp
from ddsp_pytorch_amd import synthetic as syn
shape=syn.CFG4_PER_GPU
ctl=syn.make_controls(shape,1,batch=512)
H=torch.from_numpy(ctl['H']).cuda()
L=ddsp._lib.lib()
y=torch.empty(512,64000,device='cuda')
for mode in (0, 1<<8, 2<<8, 3<<8, 4<<8):
    L.ddsp_noise_set_generic(mode)
    for _ in range(3): ddsp.noise_forward(H,128,seed=1,out=y)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ddsp.noise_forward(H,128,seed=1,out=y)
    e1.record(); torch.cuda.synchronize()
    print('ablate',mode,'ms',e0.elapsed_time(e1)/10)
L.ddsp_noise_set_generic(0)

# cfg3 shape
shape=syn.CFG3
ctl=syn.make_controls(shape,1,batch=128)
H=torch.from_numpy(ctl['H']).cuda()
y=torch.empty(128,shape.samples,device='cuda')
for mode in (0, 3<<8, 4<<8):
    L.ddsp_noise_set_generic(mode)
    for _ in range(2): ddsp.noise_forward(H,512,seed=1,out=y)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ddsp.noise_forward(H,512,seed=1,out=y)
    e1.record(); torch.cuda.synchronize()
    print('cfg3 B=128 mode',mode,'ms',e0.elapsed_time(e1)/5)
L.ddsp_noise_set_generic(0)
