import os, sys, subprocess, numpy as np
# interleaved A/B of two library builds on the noise kernel (bench shape): python noise_ab.py libA libB
code = '''
import sys, torch
sys.path.insert(0,'.')
import ddsp_pytorch_amd as ddsp
from ddsp_pytorch_amd import synthetic as syn
shape=syn.CFG4_PER_GPU
H=torch.from_numpy(syn.make_controls(shape,1,batch=512)['H']).cuda()
y=torch.empty(512,64000,device='cuda')
for _ in range(3): ddsp.noise_forward(H,128,seed=1,out=y)
torch.cuda.synchronize()
ts=[]
for r in range(5):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ddsp.noise_forward(H,128,seed=1,out=y)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/10)
print(min(ts))
'''
res={}
for rnd in range(3):
    for lib in sys.argv[1:]:
        out=subprocess.run([sys.executable,'-c',code],env=dict(os.environ,DDSP_HIP_LIB=os.path.abspath(lib)),capture_output=True,text=True).stdout.strip().splitlines()[-1]
        res.setdefault(lib,[]).append(float(out))
for k,v in res.items(): print(k, ['%.4f'%x for x in v])
