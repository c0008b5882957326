import sys, torch, numpy as np
sys.path.insert(0,'.')
import ddsp_pytorch_amd as ddsp
from ddsp_pytorch_amd import synthetic as syn
shape=syn.SynthShape("b",256,16000,128,500,100,65)
ctl=syn.make_controls(shape,1,"musical")
x={k:torch.from_numpy(v).cuda() for k,v in ctl.items()}
c=x["c"].clone().requires_grad_(); a=x["a"].clone().requires_grad_(); H=x["H"].clone().requires_grad_()
class Conf: n_harmonics,sample_rate,hop_length=100,16000,128
osc=ddsp.OscillatorBank(Conf).cuda(); noise=ddsp.FilteredNoise(Conf,rng="device")
def step():
    y=osc({"f0":x["f0"],"c":c,"a":a})+noise({"H":H})
    y.square().mean().backward()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5): step()
    torch.cuda.synchronize()
rows=[(e.key,e.device_time_total/5/1e3) for e in prof.key_averages()]
for k,t in sorted(rows,key=lambda r:-r[1])[:8]: print("%8.3f ms  %s"%(t,k[:90]))
