import sys, torch, numpy as np
sys.path.insert(0,'.')
import ddsp_pytorch_amd as ddsp
from ddsp_pytorch_amd import synthetic as syn
shape=syn.CFG4_PER_GPU
ctl=syn.make_controls(shape,1,batch=512)
H=torch.from_numpy(ctl['H']).cuda()
L=ddsp._lib.lib()
y=torch.empty(512,64000,device='cuda')
modes=[int(a) for a in sys.argv[1:]] or [0,64]
res={m:[] for m in modes}
for rnd in range(8):
    for mode in modes:
        L.ddsp_noise_set_generic(mode)
        for _ in range(2): ddsp.noise_forward(H,128,seed=1,out=y)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ddsp.noise_forward(H,128,seed=1,out=y)
        e1.record(); torch.cuda.synchronize()
        res[mode].append(e0.elapsed_time(e1)/10)
L.ddsp_noise_set_generic(0)
for m in modes: print('mode',m,'median %.4f min %.4f' % (np.median(res[m]), np.min(res[m])))
