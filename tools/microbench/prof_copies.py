"""Where do the device-to-device copies of one training step come from?  (aten::copy_ grouped by Python stack)"""
import sys, torch, numpy as np, collections
sys.path.insert(0, '.')
import ddsp_pytorch_amd as ddsp
from ddsp_pytorch_amd import synthetic as syn
class C:
    n_harmonics, n_noise_filters, sample_rate, hop_length = 100, 65, 16000, 128
    decoder_mlp_units, decoder_mlp_layers, decoder_gru_units, decoder_gru_layers = 512, 3, 512, 1
torch.manual_seed(0)
model = ddsp.Decoder(C, noise_rng="device").cuda(); loss_fn = ddsp.MSSLoss().cuda(); opt = torch.optim.Adam(model.parameters(), lr=1e-3)
rng = np.random.default_rng(0); b, frames = 32, 500
batch = {"normalized_cents": torch.from_numpy(rng.uniform(0,1,(b,frames,1)).astype(np.float32)).cuda(),
         "loudness": torch.from_numpy(rng.uniform(-1,1,(b,frames,1)).astype(np.float32)).cuda(),
         "f0": torch.from_numpy(syn.musical_f0(rng,b,frames)).cuda(),
         "audio": torch.from_numpy((0.1*rng.standard_normal((b,frames*128))).astype(np.float32)).cuda()}
for _ in range(3): ddsp.train_step(model, loss_fn, opt, batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    ddsp.train_step(model, loss_fn, opt, batch)
    torch.cuda.synchronize()
cnt = collections.Counter(); tim = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::contiguous", "aten::clone", "aten::cat"):
        key = (e.name, str(e.input_shapes)[:90])
        cnt[key] += 1; tim[key] += e.device_time_total
for k, v in sorted(cnt.items(), key=lambda kv: -tim[kv[0]])[:40]:
    print(v, round(tim[k], 1), "us", k)
