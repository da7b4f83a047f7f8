"""The benchmark's input generator: the batched ray casting (synth.make_pairs, what bench.py's jobs are generated with) against the
scan-by-scan one (synth.make_pair, what every parity test and every earlier round's numbers were generated with) -- bit for bit."""
import numpy as np
import pytest
import torch

from lv_slam_amd import synth


def _same(ids, naz, nb, dev):
    T, S, dTs = synth.make_pairs(ids, naz, device=dev, n_beams=nb)
    for i, k in enumerate(ids):
        t, s, dT = synth.make_pair(k, naz, device=dev, n_beams=nb)
        assert torch.equal(t.view(torch.int32), T[i].view(torch.int32)), f"pair {k}: target cloud differs"
        assert torch.equal(s.view(torch.int32), S[i].view(torch.int32)), f"pair {k}: source cloud differs"
        assert np.array_equal(dT, dTs[i])


def test_batched_generation_is_bit_identical_cpu():
    _same([0, 3, 7, 100, 1866, 2010, 4540], 64, 32, "cpu")     # pairs 24 m slots apart: different primitive counts in one batch (padding)
    _same([5], 128, 64, "cpu")


def test_draws_may_come_from_the_caller():
    ids = [11, 12]
    draws = [synth.pair_noise(k, 64 * 32) for k in ids]
    T, S, _ = synth.make_pairs(ids, 64, n_beams=32, draws=draws)
    T2, S2, _ = synth.make_pairs(ids, 64, n_beams=32)
    assert torch.equal(T, T2) and torch.equal(S, S2)
    bufs = (torch.empty(2, 2, 64 * 32, dtype=torch.float64), torch.empty(2, 2, 64 * 32, dtype=torch.float64))
    for j, k in enumerate(ids):                              # bench.py's way: unit normals drawn in place, scaled where the cast runs
        synth.pair_noise(k, 64 * 32, noise_sigma=None, out=(bufs[0][j], bufs[1][j]))
    T3, S3, _ = synth.make_pairs(ids, 64, n_beams=32, draws=bufs, unit_noise=True)
    assert torch.equal(T, T3) and torch.equal(S, S3)


@pytest.mark.gpu
def test_batched_generation_is_bit_identical_on_the_device():
    dev = torch.device("cuda:0")
    _same([0, 1, 2, 135, 270, 1866, 2010, 4540], 1024, 64, dev)   # the headline's cloud size
    _same([7, 100], 2048, 64, dev)                                # config 5's
