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


def _run_bench(*flags, env_extra=None):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True, timeout=600)


def test_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` (no launcher around it) must come back as TWO ranks that met at a barrier, and the JSON
    line -- rank 0's, the last line on stdout -- must say n_gpus = the size of the process group that really formed."""
    import json
    r = _run_bench("--gpus", "2", "--dry-run", "--steps", "7", "--warmup", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["ranks_at_barrier"] == 2
    assert line["steps"] == 7 and line["warmup"] == 2


def test_no_flags_is_one_rank_and_a_short_run():
    import json
    r = _run_bench("--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["ranks_at_barrier"] == 1 and line["steps"] <= 50 and line["warmup"] <= 10


def test_more_gpus_than_the_box_has_is_refused_loudly():
    r = _run_bench("--gpus", "64")
    assert r.returncode != 0 and "refusing" in r.stderr


def test_a_launcher_with_another_world_size_is_refused():
    r = _run_bench("--gpus", "4", "--dry-run", env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_gather_watchdog_prints_the_counts_figures_when_a_rank_stalls():
    """A rank that never joins the gather's collective (--dry-run-stall, gloo): rank 0's watchdog prints the line with what it
    already holds -- the counts exchange -- and exchange.gather = {"error": ...}; every rank leaves with --gather-timeout-rc, so
    the launcher's status is non-zero exactly when asked for."""
    import json
    r = _run_bench("--gpus", "2", "--dry-run", "--dry-run-stall", "1", "--gather-timeout", "4", "--gather-timeout-rc", "3")
    assert r.returncode != 0, r.stdout[-500:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    line = json.loads(lines[-1])
    assert line["exchange"]["counts"] == {"ranks": 2}
    assert "did not complete within 4 s" in line["exchange"]["gather"]["error"]
    # default exit code: the line carries the failure, the process status stays clean
    r = _run_bench("--gpus", "2", "--dry-run", "--dry-run-stall", "0", "--gather-timeout", "4")
    assert r.returncode == 0, r.stderr[-1500:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and "error" in line["exchange"]["gather"]


def test_gather_watchdog_at_world_8_with_a_stalled_rank():
    """The same at the world size the driver's scaling run ends with: eight ranks meet over gloo, rank 5 never joins the gather's
    collective; the line carries the counts of all eight and the gather's error, every rank leaves."""
    import json
    r = _run_bench("--gpus", "8", "--dry-run", "--dry-run-stall", "5", "--gather-timeout", "6", "--gather-timeout-rc", "3")
    assert r.returncode != 0, r.stdout[-500:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["exchange"]["counts"] == {"ranks": 8}
    assert "did not complete within 6 s" in line["exchange"]["gather"]["error"]
