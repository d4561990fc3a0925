"""-m gpu: K5c, the HT SigProp / MagRef passes on the device, through the C ABI
(grk_amd_set_decode_segments: cleanup segment + refinement segment per block) == the reference's decoder called with
lengths2 != 0 (and == the oracle, which tests/test_oracle_ht_refine.py pins to it on the CPU)."""
import numpy as np
import pytest

import grok_amd as G
import chain
import gpuutil as U
import oracle as O
import refharness as R
from test_oracle_ht_refine import make_block, code_block

pytestmark = pytest.mark.gpu


def _case(W, H, L, C, prec, irrev, seed, passes_of):
    rng = np.random.default_rng(seed)
    p = G.TileParams.make(W, H, C, prec, L, irreversible=irrev)
    blocks, qcd = G.tile_layout(p)
    table = np.zeros(len(blocks), G.capi.CODED_DTYPE)
    chunks, segs, want, off = [], [], [], 0
    npass_count = {1: 0, 2: 0, 3: 0}
    for i, b in enumerate(blocks):
        bw, bh = b.x1 - b.x0, b.y1 - b.y0
        kb = min(b.kmax, 14)
        npasses = passes_of(i)
        mag, sign = make_block(rng, bw, bh, kb, int(rng.integers(0, 5)))
        cup, seg, _ = code_block(mag, sign, b.kmax, npasses) if npasses > 1 else (code_block(mag, sign, b.kmax, 2)[0], b"", 0)
        if i % 11 == 5:
            seg = seg[:len(seg) // 2]                    # truncated refinement segment: zeros are read beyond its end
        mm = b.kmax - 1
        if R.have_ref():
            words = R.ht_decode_block_passes(cup + seg, len(cup), len(seg), npasses, mm, bw, bh)
        else:
            words = O.ht_refine_decode(O.ht_decode_block(cup, mm, bw, bh), mm, seg, npasses)
        assert words is not None
        if irrev:
            want.append(O.ht_dequant_irrev(words, chain.band_scale_dec(prec, qcd[chain.band_index(b)], b.kmax)).view(np.int32))
        else:
            want.append(O.ht_dequant_rev(words, mm))
        data = cup + seg
        table["offset"][i] = off; table["length"][i] = len(data); table["missing_msbs"][i] = mm
        chunks.append(data + b"\0" * (-len(data) % 16)); off += len(chunks[-1])
        segs.append([(len(cup), 1)] + ([(len(seg), npasses - 1)] if seg else []))
        npass_count[npasses if seg else 1] += 1
    coded = b"".join(chunks) + b"\0" * 16
    d_c = U.to_dev(np.frombuffer(coded, np.uint8))
    d_m = U.dev_planes(p, C)
    c = U.ctx()
    c.set_decode_segments(segs)
    try:
        c.stage_ht_decode(p, 1, table, d_c.data_ptr(), len(coded), d_m.data_ptr())
        c.synchronize()
    finally:
        c.set_decode_segments(None)
    got = U.planes_to_numpy(d_m, p, C)
    bad = []
    for i, b in enumerate(blocks):
        bw, bh = b.x1 - b.x0, b.y1 - b.y0
        g = got[b.comp, b.py:b.py + bh, b.px:b.px + bw]
        if not np.array_equal(g, want[i]):
            bad.append((i, bw, bh, len(segs[i]), int((g != want[i]).sum())))
    assert not bad, "blocks differing from the reference decoder (idx, w, h, segments, samples): %s" % bad[:8]
    return npass_count


def test_refinement_passes_irreversible_1024():
    """768 blocks of 64x64 (+ the small ones of the low resolutions), two and three passes mixed with cleanup-only blocks;
    the irreversible dequantisation keeps every refined bit visible"""
    n = _case(1024, 1024, 3, 3, 10, True, 11, lambda i: (i % 3) + 1)
    assert n[2] >= 100 and n[3] >= 150 and n[2] + n[3] >= 300          # (two-pass blocks without a single member have no SigProp bytes)


def test_refinement_passes_reversible_and_ragged():
    _case(200, 120, 2, 3, 8, False, 12, lambda i: 3 if i % 2 else 2)
    _case(130, 67, 4, 1, 12, True, 13, lambda i: 3)
    _case(37, 3, 1, 1, 8, True, 14, lambda i: 2)
    _case(64, 64, 0, 1, 9, True, 15, lambda i: 3)


def test_cleanup_only_blocks_are_untouched_by_the_segment_list():
    """a segment list that gives every block one segment decodes exactly as without one"""
    _case(256, 192, 3, 1, 8, True, 16, lambda i: 1)
