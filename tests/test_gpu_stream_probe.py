"""-m gpu: the stream probe (context.hip probe_streams; profiles/r06_hw_queues.txt).  Which hardware queue a stream gets depends on
the streams the process made before, so what the probe FINDS differs from process to process; what has to hold everywhere: results
do not depend on it, it can be switched off, the exported pair test answers for a host's own streams and refuses nonsense."""
import numpy as np
import pytest
import torch

import grok_amd as G
import gpuutil as U
import synth
from grok_amd import dist as D

pytestmark = pytest.mark.gpu


def _frames(ctx, p, d_px, n, nblocks):
    ctx.set_pipelining(True)
    out = []
    for _ in range(n):
        ctx.encode_tiles(p, 1, d_px.data_ptr(), True, fetch=False)
        t, tot = ctx.fetch_table(nblocks)
        out.append((t.copy(), ctx.fetch_coded(tot).copy()))
    ctx.set_pipelining(False)
    return out


def test_results_do_not_depend_on_the_probe(monkeypatch):
    px = synth.g2(3, 512, 768, 8, seed=2)
    p = G.TileParams.make(768, 512, 3, 8, 4)
    import ctypes as C
    nb = G.lib().grk_amd_tile_num_blocks(C.byref(p))
    d_px = U.to_dev(px.reshape(-1).view(np.uint8))
    extra = [torch.cuda.Stream() for _ in range(5)]               # (another count of earlier streams than the other tests')
    for s in extra:
        with torch.cuda.stream(s):
            torch.zeros(16, device="cuda").add_(1)
    torch.cuda.synchronize()
    res = {}
    for probe in ("1", "0"):
        monkeypatch.setenv("GRK_AMD_STREAM_PROBE", probe)
        ctx = G.Context(0)
        try:
            res[probe] = _frames(ctx, p, d_px, 3, nb)
            r = int(G.lib().grk_amd_stream_probe_result(ctx._h))
            assert (r >= 0) if probe == "1" else (r == -1)
        finally:
            ctx.close()
    ref_t, ref_c = res["0"][0]
    for probe in res:
        for t, c in res[probe]:
            assert np.array_equal(t["length"], ref_t["length"])
            for i in range(len(t)):
                a, b = int(t["offset"][i]), int(ref_t["offset"][i])
                n = int(t["length"][i])
                assert bytes(c[a:a + n]) == bytes(ref_c[b:b + n])


def test_pair_test_for_a_hosts_own_streams():
    ctx = G.Context(0)
    try:
        assert ctx.probe_streams() >= 0
        mine = [ctx.internal_stream(i) for i in range(3)]
        assert all(mine) and len(set(mine)) == 3
        # the context's own three pass their own test once the probe has run (or it found nothing better in eight tries: then it says so)
        ok = [ctx.streams_side_by_side(mine[i], mine[j]) for i in range(3) for j in range(i + 1, 3)]
        assert all(isinstance(v, bool) for v in ok)
        s = D.independent_stream(ctx, torch.device("cuda", 0))
        assert isinstance(s, torch.cuda.Stream)
        with pytest.raises(RuntimeError):
            ctx.streams_side_by_side(mine[0], mine[0])
        with pytest.raises(RuntimeError):
            ctx.streams_side_by_side(None, mine[0])
    finally:
        ctx.close()
