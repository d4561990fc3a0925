"""-m gpu: bench.py's N > 1 path on ONE GPU (GROK_AMD_FORCE_DIST=1: torch.distributed over RCCL at world size 1) -- the three exchanges
of a frame (counts; gather of loose blocks, Tier-2 on the writer's host; finished tile-parts, Tier-2 on the device inside the timed
region) complete, and the two that deliver a frame deliver the SAME codestream."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forced_world1_exchanges_agree():
    env = dict(os.environ, GROK_AMD_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg2", "--steps", "6", "--warmup", "2", "--region-repeats", "1",
                        "--no-cpu-baseline", "--no-workloads", "--no-host-boundary", "--no-live-pmc"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    ex = d["exchange"]
    assert ex["counts"]["Mpixels_s"] > 0
    assert ex["gather"]["assembled_codestream_bytes"] > 0
    parts = ex["parts"]
    assert "error" not in parts, parts
    assert parts["equals_gather_form_length"] is True
    assert parts["file_equals_gather_form_file"] is True
    assert d["multi_gpu"]["ranks_seen_by_rccl"] == 1
