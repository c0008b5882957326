"""A fixed-seed slice of the randomised parity sweep (tests/fuzz_parity.py) in the GPU suite: 150 random shapes of the
oscillator bank and the filtered noise through the C ABI against the CPU oracle -- phases bit-exact, audio <= 1e-5, noise <= 2e-6."""
import pytest

pytestmark = pytest.mark.gpu

import fuzz_parity  # noqa: E402


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_shapes_match_the_oracle(seed):
    bad, worst_audio, worst_noise = fuzz_parity.sweep(50, seed, verbose=False)
    assert bad == 0, (bad, worst_audio, worst_noise)


@pytest.mark.parametrize("seed", [21, 22])
def test_random_shapes_of_the_loss_side_kernels_match_torch(seed):
    """40 random cases per seed: one-kernel spectral-loss scales (value + gradient, any overlap), the framing pair, column sums."""
    assert fuzz_parity.sweep_training_kernels(40, seed, verbose=False) == 0


@pytest.mark.parametrize("seed", [404])
def test_random_shapes_of_the_chunked_oscillator_match_the_oracle(seed):
    bad, worst = fuzz_parity.sweep_chunked(60, seed, verbose=False)
    assert bad == 0 and worst <= 1e-5
