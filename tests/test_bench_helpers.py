"""bench.py's bookkeeping (no GPU): the algorithmic bytes are SURVEY.md 8(d)'s per-unit figures at the storage width of the
launched instances, and no rate can come out above what its bytes and time say."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)          # (defines functions; main() runs only under __main__)
    return m


def test_algorithmic_bytes_at_the_storage_width():
    B = _bench()
    S = 8192 * 8192 * 3
    sig5 = sum(4.0 ** -l for l in range(5))
    assert abs(B.sigma(5) - sig5) < 1e-12
    # the 8-bit headline: pixels 1 B, planes 2 B -- fused level 0 reads the pixels and writes one plane's worth
    assert B.dwt_bytes(S, 1, 2, 5, True) == S * 3 + 4.0 * S * (sig5 - 1.0)
    assert round(B.dwt_bytes(S, 1, 2, 5, True)) == 871366656            # (what the bench line of r03 carries)
    # the reference-width path: SURVEY 8(d)'s 8 S sigma_L for the un-fused family
    assert B.dwt_bytes(S, 1, 4, 5, False) == 8.0 * S * sig5 == B.dwt_bytes_unfused(S, 4, 5)
    assert B.ht_bytes(S, 2, 100) == 2 * S + 100 and B.ht_bytes(S, 4, 0) == 4 * S
    assert B.idwt_bytes(S, 1, 2, 5, True) == 4.0 * S * (sig5 - 1.0) + S * 3
    assert B.idwt_bytes(S, 1, 4, 5, False) == 8.0 * S * sig5


def test_rates_and_labels():
    B = _bench()
    g, f = B.rate(8.0e9, 1.0)          # 8 GB in a millisecond is the peak
    assert g == 8000.0 and f == 1.0
    assert B.rate(1.0, 0) == (None, None) and B.rate(1.0, None) == (None, None)
    assert B.dtype_label(True, 4) == "f32" and B.dtype_label(False, 4) == "int32"
    assert B.dtype_label(False, 2).startswith("int16x2 packed")


def test_defaults_of_the_command_line_contract():
    src = open(os.path.join(ROOT, "bench.py")).read()
    # python bench.py with no flags: one GPU, a K / W that finish within minutes; N > 1: no data-path collective in the headline
    assert '"--gpus", type=int, default=1' in src and '"--steps", type=int, default=20' in src and '"--warmup", type=int, default=3' in src
    assert '"--exchange", default="counts"' in src
