"""The bit queue of K3's MEL coder (grok_amd/csrc/kernels_ht.hip: mel_first_row, mel_row) holds 64 bits; the kernel relies on two bounds
that follow from the coder's tables alone (ojph_block_encoder.cpp:217-291, MEL_E) -- checked here exhaustively, without a GPU:
  * the first quad row: at most 32 events from the coder's initial state put at most 47 bits into the queue (mel_first_row never drains);
  * a later row: the queue is drained when it holds more than 40 bits, and one trip of mel_row's loop -- up to 64 zero events and a one --
    adds at most 12 + 6 bits."""
from functools import lru_cache

MEL_E = [0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 4, 5]


def step(k, run, one):
    """state after an event and the bits it emits"""
    e = MEL_E[k]
    if one:
        return max(k - 1, 0), 0, e + 1
    run += 1
    if run >> e:
        return min(k + 1, 12), 0, 1
    return k, run, 0


@lru_cache(None)
def most_bits(k, run, events):
    if events == 0:
        return 0
    best = 0
    for one in (0, 1):
        k2, run2, n = step(k, run, one)
        best = max(best, n + most_bits(k2, run2, events - 1))
    return best


def test_first_quad_row_fits_the_queue():
    assert most_bits(0, 0, 32) == 47
    assert most_bits(0, 0, 32) <= 64


def test_a_trip_of_the_row_loop_adds_at_most_eighteen_bits():
    worst = 0
    for k in range(13):
        for run in range(1 << MEL_E[k]):
            kk, rr, bits = k, run, 0
            for _ in range(64):                       # 64 zero events ...
                kk, rr, n = step(kk, rr, 0)
                bits += n
            worst = max(worst, bits + MEL_E[kk] + 1)  # ... and the one behind them
    assert worst <= 18
    assert 40 + worst <= 64                           # drained above 40 bits, so the queue never overflows
