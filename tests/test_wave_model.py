"""CPU: the wave-parallel phase-B formulation (oracle/ht_wave_model.c, which the HIP kernel
transliterates) equals the serial stuffing writers of the oracle on adversarial inputs."""
import numpy as np
import pytest

import oracle as O


@pytest.mark.parametrize("form", (1, 2, 3))
@pytest.mark.parametrize("mode", range(8))
def test_wave_model_matches_oracle(mode, form):
    rng = np.random.default_rng(100 + mode)
    for trial in range(60):
        w, h, kmax = int(rng.integers(1, 65)), int(rng.integers(1, 65)), int(rng.integers(2, 20))
        if trial % 3 == 0:
            w = h = 64
        mag = rng.integers(0, 1 << kmax, size=(h, w))
        if mode == 1: mag = mag >> rng.integers(0, kmax, size=(h, w))
        if mode == 2: mag = np.where(rng.random((h, w)) < 0.9, 0, mag)
        if mode == 3: mag = mag & 3
        if mode == 4: mag = np.full((h, w), (1 << kmax) - 1)              # every MagSgn byte is 0xFF
        if mode == 5: mag = np.where(rng.random((h, w)) < 0.5, (1 << kmax) - 1, mag)
        if mode == 6: mag = np.where(rng.random((h, w)) < 0.97, (1 << kmax) - 1, mag)       # long runs of 0xFF with breaks
        if mode == 7: mag = np.where(rng.random((h, w)) < 0.2, (1 << kmax) - 1, mag >> rng.integers(0, kmax, size=(h, w)))
        sm = O.signmag(mag * np.where(rng.random((h, w)) < 0.5, -1, 1), kmax)
        assert O.ht_wave_model(sm, kmax, form) == O.ht_encode_sm(sm, kmax), (w, h, kmax, mode)
