"""Problem definitions shared by tests/two_rank_worker.py (one fresh process per rank) and tests/test_gpu_two_ranks.py
(the unsharded single-process run the ranks are compared with)."""
import numpy as np
import torch

from ddsp_pytorch_amd import synthetic as syn

SYNTH = syn.SynthShape("two_rank", 7, 16000, 128, 40, 100, 65)       # odd batch: the shards differ in size (4 + 3 rows)
TRAIN_ROWS, TRAIN_FRAMES, TRAIN_STEPS = 4, 32, 2


class TrainConf:
    n_harmonics, n_noise_filters, sample_rate, hop_length = 100, 65, 16000, 128
    decoder_mlp_units, decoder_mlp_layers, decoder_gru_units, decoder_gru_layers = 64, 2, 64, 1


def synth_problem():
    ctl = syn.make_controls(SYNTH, 91, "musical")
    uniform = np.random.default_rng(92).random((SYNTH.batch, SYNTH.frames, SYNTH.hop), dtype=np.float32)
    return ctl, uniform


def synthesize(ddsp, x, uniform):
    class Conf:
        n_harmonics, sample_rate, hop_length = SYNTH.n_harmonics, SYNTH.sample_rate, SYNTH.hop

    y = ddsp.OscillatorBank(Conf).cuda()(x)
    return ddsp.noise_forward(x["H"], SYNTH.hop, uniform=uniform, out=y, accumulate=True)


def train_batch():
    rng = np.random.default_rng(93)
    b, t = TRAIN_ROWS, TRAIN_FRAMES
    return {"normalized_cents": torch.from_numpy(rng.uniform(0, 1, (b, t, 1)).astype(np.float32)),
            "loudness": torch.from_numpy(rng.uniform(-1, 1, (b, t, 1)).astype(np.float32)),
            "f0": torch.from_numpy(rng.uniform(80, 400, (b, t, 1)).astype(np.float32)),
            "audio": torch.from_numpy((0.1 * rng.standard_normal((b, t * 128))).astype(np.float32)),
            # the uniform draw of the noise branch travels with the rows, so a shard sees exactly the draw its rows
            # get in the unsharded run (the modules' own RNG modes restart per process)
            "noise_draw": torch.from_numpy(rng.random((b, t, 128), dtype=np.float32))}


def make_trainer(ddsp):
    """Identical replicas on every rank: same seed -> same initial weights.  Plain SGD so that the updated weights are a
    linear image of the gradients (Adam's g/(|g|+eps) would amplify rounding-level differences of near-zero gradients)."""
    torch.manual_seed(17)
    model = ddsp.Decoder(TrainConf, noise_rng="device").cuda()
    noise = model.noise
    plain_forward = noise.forward

    def forward_with_batch_draw(x, noise=None, out=None):
        return plain_forward(x, noise=model._draw, out=out)

    noise.forward = forward_with_batch_draw
    plain_model_forward = model.forward

    def model_forward(batch):
        model._draw = batch["noise_draw"]
        return plain_model_forward(batch)

    model.forward = model_forward
    with torch.no_grad():
        model.reverb.wet.fill_(-1.0)                                   # an audible reverb tail: its parameters get gradients
    loss_fn = ddsp.MSSLoss().cuda()
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.02)
    return model, loss_fn, opt
