"""K3 against the oracle, block by block, with where the first difference lies (dev tool for kernel rewrites):
    GRK_AMD_LIB=build/abl/<name>/libgrok_amd.so python tools/k3_debug.py
int32 planes (stage entry, random content in the five modes of tests/test_gpu_stages.py) and the 8-bit pixel path (int16 planes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import grok_amd as G, gpuutil as U, oracle as O, synth

def where(got, want):
    n = min(len(got), len(want))
    d = next((i for i in range(n) if got[i] != want[i]), n)
    lw = len(want)
    scup = ((want[lw - 1] << 4) | (want[lw - 2] & 0xF)) if lw >= 2 else 0
    seg = "MagSgn" if d < lw - scup else "MEL/VLC"
    return "first diff at byte %d of %d (gpu len %d), oracle MagSgn bytes %d, Scup %d -> in %s; gpu %s want %s" % (
        d, lw, len(got), lw - scup, scup, seg, bytes(got[max(0, d - 2):d + 6]).hex(), bytes(want[max(0, d - 2):d + 6]).hex())

def planes_case(W, H, L, C, prec, mode, seed):
    rng = np.random.default_rng(seed)
    p = G.TileParams.make(W, H, C, prec, L)
    blocks, _ = G.tile_layout(p)
    planes = np.zeros((C, H, W), np.int32)
    for b in blocks:
        bw, bh = b.x1 - b.x0, b.y1 - b.y0
        mag = rng.integers(0, 1 << b.kmax, size=(bh, bw))
        if mode == 1: mag = mag >> rng.integers(0, b.kmax + 1, size=(bh, bw))
        elif mode == 2: mag = np.where(rng.random((bh, bw)) < 0.93, 0, mag & 7)
        elif mode == 3: mag = np.zeros((bh, bw), np.int64)
        elif mode == 4: mag = np.full((bh, bw), (1 << b.kmax) - 1)
        sign = np.where(rng.random((bh, bw)) < 0.5, -1, 1)
        planes[b.comp, b.py:b.py + bh, b.px:b.px + bw] = (mag * sign).astype(np.int32)
    d_m = U.upload_planes(planes, p)
    c = U.ctx()
    c.stage_ht_encode(p, 1, d_m.data_ptr())
    table, tot = c.fetch_table(len(blocks))
    got = U.split_blocks(table, c.fetch_coded(tot))
    bad = 0
    for i, b in enumerate(blocks):
        bw, bh = b.x1 - b.x0, b.y1 - b.y0
        want = O.ht_encode_sm(O.signmag(planes[b.comp, b.py:b.py + bh, b.px:b.px + bw], b.kmax), b.kmax)
        if got[i] != want:
            bad += 1
            if bad <= 3: print("   block %d (%dx%d kmax %d): %s" % (i, bw, bh, b.kmax, where(got[i], want)))
    print("planes %dx%d L%d prec %d mode %d: %d of %d blocks differ" % (W, H, L, prec, mode, bad, len(blocks)), flush=True)
    return bad

def pixel_case(C, H, W, prec, L, gen="g2"):
    px = getattr(synth, gen)(C, H, W, prec)
    p = G.TileParams.make(W, H, C, prec, L)
    table, coded = U.ctx().encode_host(p, px)
    got = U.split_blocks(table, coded)
    blocks, lens, ocoded = O.encode_tile_rev(px, prec, L)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    bad = 0
    for i in range(len(blocks)):
        want = bytes(ocoded[off[i]:off[i + 1]])
        if got[i] != want:
            bad += 1
            if bad <= 3: print("   block %d: %s" % (i, where(got[i], want)))
    print("pixels %s %dx%dx%d prec %d L%d: %d of %d blocks differ" % (gen, C, H, W, prec, L, bad, len(blocks)), flush=True)
    return bad

if __name__ == "__main__":
    tot = 0
    for mode in (3, 2, 1, 0, 4):
        tot += planes_case(128, 128, 0, 1, 8, mode, 5 + mode)
    tot += planes_case(512, 512, 3, 1, 8, 1, 77)
    tot += planes_case(256, 256, 1, 1, 16, 0, 9)
    for args in ((1, 128, 128, 8, 0), (1, 512, 512, 8, 3), (3, 256, 384, 8, 2), (1, 512, 512, 8, 3, "g0")):
        tot += pixel_case(*args)
    print("TOTAL differing blocks:", tot)
