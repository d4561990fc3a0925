#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: encode Mpixels/s (whole job) for 8K RGB HTJ2K
lossless, with the dominant kernel's HBM roofline and the reference CPU encoder beside it.

One "step" = one pass of the hot path (ingest+DC+RCT -> 5-level 5/3 DWT -> HT cleanup coding of
64x64 code-blocks -> compaction) over one 8192x8192x3 8-bit tile whose pixels are already
resident in HBM.  With N ranks (torchrun, one process per GPU) the job is a sequence of (N*8192)x8192
frames cut into N tiles of 8192x8192: tile t is encoded on rank t (tiles are independent, SURVEY.md
§8e: no data-path collective) and every frame ends with an exchange over RCCL/xGMI.  THREE forms are timed
and all are in the line (top-level `exchange`): "gather" -- the ranks' coded blocks + block tables (exact
sizes) to the frame's writer rank, which rotates with the frame number, `--gather-depth` frames' gathers in
flight at once on communicators of their own (grok_amd.dist.FramePipeline); "counts" -- only the byte
counts travel, every rank writes its own tile-parts (parallel writers); "parts" -- every rank runs Tier-2
on its device inside the timed region (grk_amd_assemble_device_async) and its FINISHED tile-parts travel.
`value` is the faster of gather / counts unless `--exchange` names one (`config.headline_exchange` says
which); "parts" is reported, never the headline.  Per-GPU work is fixed => weak scaling.  "multi_gpu" also carries
the BASELINE configs[3] shape -- 16384x16384 as 256 tiles of 1024x1024 split over the ranks, strong
scaling.  `python bench.py --gpus N` starts its N ranks itself (launch_ranks).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 8k|cfg2|cfg1|cfg3|cfg4tile]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import grok_amd as G  # noqa: E402
import grok_amd.dist as D  # noqa: E402
import synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec

WORKLOADS = {
    # name: (C, W, H, prec, levels, ntiles per rank, description)
    "8k": (3, 8192, 8192, 8, 5, 1,
           "8192x8192x3 8-bit RGB, 1 tile, RCT+5/3 lossless HTJ2K, 5 levels, 64x64 blocks (BASELINE metric shape)"),
    "cfg2": (3, 4096, 4096, 8, 5, 1,
             "4096x4096x3 8-bit RGB, 1 tile, RCT+5/3 lossless HTJ2K, 5 levels (BASELINE configs[1])"),
    "cfg1": (1, 512, 512, 8, 3, 1, "512x512 8-bit mono, 1 tile, 5/3 lossless, 3 levels, HTJ2K (configs[0])"),
    "cfg3": (3, 8192, 8192, 16, 5, 1,
             "8192x8192x3 16-bit RGB, 1 tile, ICT + 9/7 + dead-zone quantiser, HTJ2K, 5 levels (BASELINE configs[2])"),
    "cfg4tile": (3, 1024, 1024, 8, 5, 64,
                 "64 tiles of 1024x1024x3 8-bit per rank, RCT+5/3 lossless HTJ2K, 5 levels (configs[3] tiling)"),
}


# md5 of the codestream Grok 8.0.2's CPU encoder writes for the workload's image (G2; re-derived from
# the reference built under oracle/_ref by the parity tests and by cpu_baseline below)
GOLDEN_MD5 = {"8k": "7e5275ef3d61edd7b95332bef74a986c"}


def sigma(levels):
    return sum(4.0 ** -l for l in range(levels))


# ---- ALGORITHMIC bytes, at the storage width the launched kernels really use.
# SURVEY.md §8(d) states them for int32 planes (4 B per coefficient); 8-bit reversible content runs on int16 planes
# (grk_amd_plane_sample_bytes = 2: exact, range-proven on the host, int32 otherwise), so a plane access counts b_pl bytes.
def dwt_bytes(samples, b_in, b_pl, levels, fused):
    """forward DWT family: level l reads S*4^-l samples and writes as many; the fused level 0 reads the caller's pixels
    (b_in B) instead of a plane (K1 folded in)"""
    if fused:
        return samples * (b_in + b_pl) + 2.0 * b_pl * samples * (sigma(levels) - 1.0)
    return 2.0 * b_pl * samples * sigma(levels)


def dwt_bytes_unfused(samples, b_pl, levels):
    """§8(d)'s DWT-only definition (one plane read + one plane write per sample and level) at the storage width"""
    return 2.0 * b_pl * samples * sigma(levels)


def ht_bytes(samples, b_pl, coded_sum):
    """K3 / K5: every coefficient once + the coded bytes (sum of the block lengths, not the arena's extent)"""
    return b_pl * samples + float(coded_sum)


def idwt_bytes(samples, b_out, b_pl, levels, fused):
    if fused:
        return 2.0 * b_pl * samples * (sigma(levels) - 1.0) + samples * (b_pl + b_out)
    return 2.0 * b_pl * samples * sigma(levels)


def rate(nbytes, ms):
    """(GB/s, fraction of the HBM peak) of nbytes moved in ms"""
    if not ms or ms <= 0:
        return None, None
    g = nbytes / ms / 1e6
    return round(g, 1), round(g / HBM_PEAK_GBPS, 4)


def dtype_label(irrev, b_pl):
    if irrev:
        return "f32"
    return "int16x2 packed (exact: range-proven on the host, int32 fallback)" if b_pl == 2 else "int32"


def cpu_baseline(rank_threads, want_cfg5=True):
    """Grok's own CPU encoder / decoder (oracle/_ref = the real reference built from its sources) on this box's host
    cores, on bounded samples of the same workloads.  Returns (json object, cfg5 stream or None): the classic
    (Part-1) codestream the reference's encoder writes for the configs[4] image is also what the GPU decodes in
    extra_workloads() -- the product has no Part-1 encoder, so the reference is the only source of such a stream."""
    cfg5 = None
    try:
        import refharness as R
        if not R.have_ref():
            raise RuntimeError("oracle/_ref missing")
        R.lib(threads=rank_threads)
        R.encode(synth.g2(3, 1024, 1024, 8), 8, numres=6)          # warm the thread pool

        def enc_sample(S, max_runs, budget_s):
            px = synth.g2(3, S, S, 8)
            times, t_start = [], time.time()
            while len(times) < max_runs and (time.time() - t_start < budget_s or len(times) < 2):
                cs, secs = R.encode(px, 8, numres=6)
                times.append(secs)
            return px, cs, sorted(times)[len(times) // 2], len(times)

        _, cs8, med8, n8 = enc_sample(8192, 3, 12.0)
        px, cs, med4, n4 = enc_sample(4096, 5, 8.0)
        out = {"value": round(8192 * 8192 / med8 / 1e6, 2), "unit": "Mpixels/s", "cores": rank_threads,
               "kind": "reference",
               "sample": "Grok 8.0.2 CPU encoder (oracle/_ref), %d x 8192x8192x3 8-bit G2 RCT+5/3 HTJ2K 5 levels (the GPU line's "
                         "workload), median of compress-call wall times, %d threads" % (n8, rank_threads),
               "file_md5": __import__("hashlib").md5(cs8).hexdigest(),
               "cfg2": {"value": round(4096 * 4096 / med4 / 1e6, 2), "unit": "Mpixels/s",
                        "sample": "%d x 4096x4096x3 8-bit, same settings" % n4,
                        "file_md5": __import__("hashlib").md5(cs).hexdigest()}}
        # BASELINE configs[2] as BASELINE.md section 3 asks for it: grk_compress() over the image's int32 planes (defect D10 rules out
        # grk_compress_tile for 16-bit samples).  The reference's HT 9/7 encoder quantises wrongly (defect D1: its FILE is not a parity
        # target) but it does all of the path's work, so its wall time is the CPU figure for this configuration
        try:
            px3 = synth.g2(3, 8192, 8192, 16)
            t3 = sorted(R.encode(px3, 16, numres=6, mode=1, ht=1, irrev=1)[1] for _ in range(2))
            out["cfg3"] = {"value": round(8192 * 8192 / t3[0] / 1e6, 2), "unit": "Mpixels/s",
                           "sample": "2 x 8192x8192x3 16-bit ICT + 9/7 + quantiser + HTJ2K 5 levels, grk_compress() on int32 planes, best "
                                     "of the compress-call wall times, %d threads (output not a parity target: reference defect D1)" % rank_threads}
            del px3
        except Exception as e:  # noqa: BLE001
            out["cfg3"] = {"value": None, "error": str(e)}
        # BASELINE configs[0]: 512 x 512 mono, 3 levels -- the reference's own test_tile_encoder case (too small for its thread pool)
        try:
            px1 = synth.g2(1, 512, 512, 8)
            t1 = sorted(R.encode(px1, 8, numres=4)[1] for _ in range(9))
            out["cfg1"] = {"value": round(512 * 512 / t1[len(t1) // 2] / 1e6, 2), "unit": "Mpixels/s",
                           "sample": "9 x 512x512x1 8-bit, 5/3 HTJ2K 3 levels (grk_compress_tile), median"}
        except Exception as e:  # noqa: BLE001
            out["cfg1"] = {"value": None, "error": str(e)}
        # BASELINE configs[3] in the two forms BASELINE.md section 3 plans, on a bounded sample of the image: 16 of its 256 tiles
        # (a 4096 x 4096 x 3 image as 1024 x 1024 tiles -- per-tile work and tile-level parallelism are the 16384^2 image's, and the
        # whole image would take a minute of CPU time per run): (a) the grk_compress_tile loop (the reference test's sequence: tiles one
        # after the other, each with the library's thread pool inside), (b) grk_compress() over the image (tiles as pooled tasks,
        # codestream/CodeStreamCompress.cpp:535-603)
        try:
            px4 = synth.g2(3, 4096, 4096, 8)
            forms = {}
            for form, mode in (("grk_compress_tile_loop", 0), ("grk_compress_tile_parallel", 1)):
                ts = sorted(R.encode(px4, 8, TW=1024, TH=1024, numres=6, mode=mode)[1] for _ in range(3))
                forms[form] = {"value": round(4096 * 4096 / ts[1] / 1e6, 2), "unit": "Mpixels/s"}
            out["cfg4"] = dict(forms, sample="3 x 16 tiles of 1024x1024x3 8-bit (a 4096x4096 image: 1/16 of configs[3]'s 256 tiles), "
                                              "RCT+5/3 HTJ2K 5 levels, median, %d threads" % rank_threads)
        except Exception as e:  # noqa: BLE001
            out["cfg4"] = {"value": None, "error": str(e)}
        # the decode direction beside it: grk_decompress of the codestream just produced
        try:
            dts = []
            for _ in range(3):
                t0 = time.time()
                R.decode(cs, 3, 4096, 4096)
                dts.append(time.time() - t0)
            out["decode"] = {"value": round(4096 * 4096 / sorted(dts)[1] / 1e6, 2), "unit": "Mpixels/s",
                             "sample": "grk_decompress of the 4096x4096x3 HT stream, median of 3 wall times "
                                       "(includes codestream parsing and Tier-2)"}
        except Exception as e:  # noqa: BLE001
            out["decode"] = {"value": None, "error": str(e)}
        if want_cfg5:
            try:      # BASELINE configs[4]: 8192^2 x 3 12-bit, Part-1 EBCOT + ICT + 9/7
                S = 8192
                px5 = synth.g2(3, S, S, 12)
                t0 = time.time()
                cs5, _ = R.encode(px5, 12, numres=6, mode=1, ht=0, irrev=1)
                t_enc = time.time() - t0
                dts = []
                for _ in range(2):
                    t0 = time.time()
                    ref5 = R.decode(cs5, 3, S, S)
                    dts.append(time.time() - t0)
                out["cfg5_decode"] = {"value": round(S * S / min(dts) / 1e6, 2), "unit": "Mpixels/s",
                                      "sample": "grk_decompress of an 8192x8192x3 12-bit Part-1 + ICT + 9/7 stream (%d bytes, written "
                                                "by grk_compress in %.1f s), best of 2 wall times" % (len(cs5), t_enc)}
                cfg5 = (cs5, ref5, S)
            except Exception as e:  # noqa: BLE001
                out["cfg5_decode"] = {"value": None, "error": str(e)}
        return out, cfg5
    except Exception as e:  # fall back to the scalar port of the oracle
        import oracle as O
        px = synth.g2(3, 1024, 1024, 8)
        t0 = time.time()
        O.encode_tile_rev(px, 8, 5)
        dt = time.time() - t0
        return {"value": round(1024 * 1024 / dt / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                "sample": "oracle/j2k_oracle.c scalar port, 1 x 1024x1024x3 (reference harness unavailable: %s)" % e}, None


_PMC_LIVE = {}          # workload -> (family -> {hbm_bytes_per_step ...}) measured by THIS run (live_pmc_traffic)
_PMC_SOURCE = {}        # workload -> where _pmc_traffic's numbers for it came from


def live_pmc_traffic(workload, seconds=100):
    """HBM traffic of `workload`'s kernel families measured in THIS run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE --
    separately, as MI355X_MICROARCH.md prescribes; no trace flags beside --pmc) over tools/prof_run.py, two steps of the workload
    with every kernel alone, summarised by profiles/summarize_pmc.py's rules (x1024; FETCH x2 on gfx950).  Rank 0 at N = 1 only;
    leaves the committed summary in place (and says so) when rocprofv3 is missing or a pass fails / takes too long."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return False
    try:
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        import summarize_pmc as SP
        tmp = tempfile.mkdtemp(prefix="grk_pmc_")
        env = dict(os.environ, TMPDIR=tmp, GRK_AMD_OVERLAP="0", PROF_WORKLOAD=workload, PROF_N="2", PROF_DECODE="1")
        paths = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            subprocess.run([exe, "--pmc", counter, "-d", out, "-o", "p", "--output-format", "csv", "--",
                            sys.executable, os.path.join(ROOT, "tools", "prof_run.py")],
                           cwd=tmp, env=env, timeout=seconds, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            hits = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
            if not hits:
                return False
            paths[counter] = hits[0]
        js = os.path.join(tmp, "traffic.json")
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            SP.main(paths["FETCH_SIZE"], paths["WRITE_SIZE"], js)
        _PMC_LIVE[workload] = json.load(open(js))
        shutil.rmtree(tmp, ignore_errors=True)
        return True
    except Exception:
        return False


def _pmc_traffic(workload, fams):
    """HBM bytes per step of the kernel families `fams`: from this run's own rocprofv3 --pmc passes when they were taken
    (live_pmc_traffic), else from the committed passes of the same command (profiles/r*_pmc_traffic_<workload>.json;
    profiles/summarize_pmc.py: FETCH_SIZE x1024 x2 [gfx950 half-count] + WRITE_SIZE x1024); None without either."""
    try:
        if workload in _PMC_LIVE:
            pm = _PMC_LIVE[workload]
            _PMC_SOURCE[workload] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes taken by this run"
        else:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%s.json" % workload)))
            if not cands:
                return None
            pm = json.load(open(cands[-1]))
            _PMC_SOURCE[workload] = "committed " + os.path.relpath(cands[-1], ROOT)
        return int(sum(pm[k]["hbm_bytes_per_step"] for k in fams if k in pm)) or None
    except Exception:
        return None


def _parity_vs_cpu_encode(ctx, params, tile, W, H, prec, levels, ntiles, table, arena_used, irrev, cpu_md5=None):
    """The parity field of a `workloads` entry, measured by THIS run.  Reversible configurations: the md5 of the codestream made of
    the GPU's blocks against the md5 of the file Grok's own CPU encoder (oracle/_ref) writes for the same image -- taken from
    cpu_baseline when that leg encoded this very image, else encoded here.  cfg3 (irreversible HT: the reference's own ENCODER is
    broken there, SURVEY.md defect D1, so no reference bytes exist): the GPU's codestream through the reference's DECODER
    (grk_decompress) -- PSNR / max error against the source -- and the GPU's own decode of the stream against those pixels."""
    import hashlib
    try:
        import refharness as R
        if not R.have_ref():
            return {"error": "oracle/_ref missing"}
        g = int(round(ntiles ** 0.5))
        if g * g != ntiles:
            return {"error": "tile count is not a square grid"}
        if irrev:
            # (the reference's decoder refuses the LL block under plain G2's darkest corner at this bit depth -- defect D5, synth.g2_mid:
            #  the frame that takes this route is G2 clipped into the middle three quarters of the range, encoded here)
            tile = synth.g2_mid(tile.shape[0], H, W, prec)
            table, coded = ctx.encode_host(params, tile)
            cs = G.write_codestream(params, W, H, table, coded)
            ref = R.decode(cs, tile.shape[0], H, W)
            err = np.abs(ref.astype(np.int64) - tile.astype(np.int64))
            mse = float((err.astype(np.float64) ** 2).mean())
            back = ctx.decode_host(params, table, coded)[0].astype(np.int32)
            return {"reference_decoder": {"decoder": "grk_decompress (oracle/_ref) of the codestream made of the GPU's blocks; content: G2 "
                                                     "clipped to [2^prec / 8, 7 * 2^prec / 8] (reference decoder defect D5)",
                                          "max_abs_error": int(err.max()),
                                          "psnr_db": round(10.0 * np.log10(float((1 << prec) - 1) ** 2 / mse), 2) if mse else None,
                                          "bound_in_tests": "max_abs_error <= 8, psnr_db >= 90 (tests/test_gpu_at_size.py, tests/test_oracle_decode.py)"},
                    "gpu_decode_equals_reference_decoder": bool(np.array_equal(back, ref)),
                    "blocks_vs_oracle_chain": "all 49 152 blocks == the oracle chain's bytes: tests/test_gpu_at_size.py::"
                                              "test_cfg3_8k_16bit_ict_dwt97_coefficients_and_blocks (minutes of CPU: not repeated here)"}
        coded = ctx.fetch_coded(arena_used)
        cs = G.write_codestream(params, W * g, H * g, table, coded)
        md5 = hashlib.md5(cs).hexdigest()
        if cpu_md5 is None:
            img = np.ascontiguousarray(np.tile(tile, (1, g, g))) if g > 1 else tile
            want, _ = R.encode(img, prec, TW=W if g > 1 else None, TH=H if g > 1 else None, numres=levels + 1, mode=1 if g > 1 else 0)
            cpu_md5, src = hashlib.md5(want).hexdigest(), "encoded by this run"
        else:
            src = "cpu_baseline's file of the same image, this run"
        return {"file_equals_cpu_encode": md5 == cpu_md5, "codestream_md5": md5, "codestream_bytes": len(cs),
                "cpu_file": "Grok 8.0.2 (oracle/_ref), " + src}
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[-200:]}


def _encode_workload(ctx, dev, stream, steps, Cn, W, H, prec, levels, ntiles, desc, irrev, traffic_key, host=None, parity=False,
                     cpu_md5=None):
    """A few pipelined steps of one encode configuration + its kernel families one at a time (HIP events on the stream each
    kernel is launched on).  Bytes at the storage width of the planes (grk_amd_plane_sample_bytes)."""
    params = G.TileParams.make(W, H, Cn, prec, levels, irreversible=irrev)
    tile = None
    if host is None:
        tile = synth.g2(Cn, H, W, prec)
        host = np.ascontiguousarray(np.broadcast_to(tile.reshape(1, -1), (ntiles, tile.size))).reshape(-1)
    d_px = torch.from_numpy(host.view(np.uint8)).to(dev)
    samples = W * H * ntiles * Cn
    b_in = (prec + 7) // 8
    b_pl, pk_levels = ctx.plane_sample_bytes(params)
    nblocks = G.lib().grk_amd_tile_num_blocks(C.byref(params)) * ntiles
    ctx.set_overlap(True)
    ctx.set_pipelining(True)
    # (a warm stretch first, as before the headline's timed region: clocks and pipeline settle; then 2 x steps frames)
    with torch.cuda.stream(stream):
        for _ in range(max(3, min(20, int(12.0 / max(0.1, samples / 2.0e8 * 0.45))))):
            ctx.encode_tiles(params, ntiles, d_px.data_ptr(), True, fetch=False)
    torch.cuda.synchronize(dev)
    nsteps = 2 * steps
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(nsteps):
            ctx.encode_tiles(params, ntiles, d_px.data_ptr(), True, fetch=False)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / nsteps * 1e3
    ms_hold = None
    if samples <= (16 << 20):
        # small frames (a frame's chain on a stream of the context's own): the same again with grk_amd_set_pixel_hold -- the caller
        # keeps its pixels untouched until it asks (this loop never rewrites them), the context's stream does not wait per call
        ctx.set_pixel_hold(True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for _ in range(nsteps):
                ctx.encode_tiles(params, ntiles, d_px.data_ptr(), True, fetch=False)
        torch.cuda.synchronize(dev)
        ms_hold = (time.perf_counter() - t0) / nsteps * 1e3
        ctx.set_pixel_hold(False)
    ctx.set_pipelining(False)
    ctx.set_overlap(False)
    ctx.enable_timing(True)
    with torch.cuda.stream(stream):
        for _ in range(steps):
            ctx.encode_tiles(params, ntiles, d_px.data_ptr(), True, fetch=False)
    torch.cuda.synchronize(dev)
    dwt_ms = ctx.kernel_ms(1)[0]
    parts = [ctx.kernel_ms(i) for i in (2, 4, 8)]
    n_ht = max([n for _, n in parts] + [1])
    ht_ms = sum(m * n for m, n in parts) / n_ht
    ctx.enable_timing(False)
    ctx.set_overlap(True)
    table, arena_used = ctx.fetch_table(nblocks)
    coded_sum = int(table["length"].astype(np.int64).sum())
    fam_d = "dwt97_5levels" if irrev else "dwt53_5levels"
    db, du, hb = dwt_bytes(samples, b_in, b_pl, levels, True), dwt_bytes_unfused(samples, b_pl, levels), ht_bytes(samples, b_pl, coded_sum)
    t_d = _pmc_traffic(traffic_key, ("dwt_level0_fused", "dwt_levels_1plus")) if traffic_key else None
    t_h = _pmc_traffic(traffic_key, ("ht_encode_kernel",)) if traffic_key else None
    w = {"workload": desc, "ms_per_step": round(ms, 4), "value": round(W * H * ntiles / ms / 1e3, 1), "unit": "Mpixels/s",
         "dtype": dtype_label(irrev, b_pl), "plane_bytes_per_coefficient": b_pl, "packed_dwt_levels": pk_levels,
         "coded_bytes": coded_sum, "arena_bytes_used": int(arena_used), "code_blocks": int(nblocks),
         "kernels": {
             fam_d: {"avg_ms": round(dwt_ms, 4), "algorithmic_bytes": int(db), "algorithmic_GBps": rate(db, dwt_ms)[0],
                     "frac": rate(db, dwt_ms)[1],
                     # SURVEY.md §8(d)'s unfused DWT-only figure (one plane read + one write per sample and level) at this
                     # storage width -- for 4-byte planes the 8*S*sigma_L north_star's ">= 40 % on the 9/7 DWT" is stated on
                     "frac_on_unfused_dwt_only_bytes": rate(du, dwt_ms)[1],
                     "traffic": t_d, "frac_on_traffic": rate(t_d, dwt_ms)[1] if t_d else None},
             "ht_cleanup_encode": {"avg_ms": round(ht_ms, 4), "algorithmic_bytes": int(hb), "algorithmic_GBps": rate(hb, ht_ms)[0],
                                   "frac": rate(hb, ht_ms)[1], "traffic": t_h, "frac_on_traffic": rate(t_h, ht_ms)[1] if t_h else None}},
         "traffic_source": _PMC_SOURCE.get(traffic_key) if traffic_key else None}
    if ms_hold is not None:
        w["ms_per_step_pixel_hold"] = round(ms_hold, 4)
    del d_px
    if parity and tile is not None:
        w["parity"] = _parity_vs_cpu_encode(ctx, params, tile, W, H, prec, levels, ntiles, table, arena_used, irrev, cpu_md5)
    return w


def cfg5_sequence(ctx, dev, p5, table, d_c, ref5, S, flights):
    """ms per frame of the cfg5 (Part-1) stream decoded as a SEQUENCE through one context, by frames in flight; every figure
    with the first and the last output buffer compared to grk_decompress's pixels."""
    res = {}
    outs = []
    try:
        for nfl in flights:
            while len(outs) < nfl:
                outs.append(torch.zeros(3 * S * S, dtype=torch.int16, device=dev))
            torch.cuda.synchronize(dev)                    # (torch's fills have landed before the context's streams write)
            ctx.set_decode_pipelining(nfl)
            for k in range(2 * nfl):
                ctx.decode_device(p5, 1, table, d_c.data_ptr(), d_c.numel(), outs[k % nfl].data_ptr())
            ctx.synchronize()
            t0 = time.perf_counter()
            for k in range(3 * nfl):
                ctx.decode_device(p5, 1, table, d_c.data_ptr(), d_c.numel(), outs[k % nfl].data_ptr())
            ctx.synchronize()
            msn = (time.perf_counter() - t0) / (3 * nfl) * 1e3
            ctx.decode_status()
            res[str(nfl)] = {"frames_in_flight": nfl, "ms_per_frame": round(msn, 3), "value": round(S * S / msn / 1e3, 1), "unit": "Mpixels/s",
                             "pixels_equal_grk_decompress": all(bool(np.array_equal(
                                 o.cpu().numpy().view(np.uint16).reshape(3, S, S).astype(np.int32), ref5)) for o in (outs[0], outs[nfl - 1]))}
    except Exception as e:      # noqa: BLE001
        res["error"] = str(e)
    finally:
        ctx.set_decode_pipelining(0)
    return res


def cfg5_sequence_child(fcs, fref, S, device):
    """`bench.py --cfg5-sequence <codestream> <reference pixels .npy> <S> <device>`: the child process of the cfg5 leg (started with
    GPU_MAX_HW_QUEUES=8 in its environment); prints one JSON object."""
    import j2kparse as J
    with open(fcs, "rb") as fh:
        cs5 = fh.read()
    ref5 = np.load(fref)
    dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    info = J.parse(cs5)
    p5 = G.TileParams.make(S, S, 3, 12, info["levels"], irreversible=True, mct=True, part1=True)
    blocks, _ = G.tile_layout(p5)
    rows, data = J.decode_table(info, blocks, True)
    table = np.array(rows, dtype=G.capi.CODED_DTYPE)
    ctx = G.Context(device)
    ctx.set_decode_qcd([(e << 11) | m for e, m in info["qcd"]])
    d_c = torch.from_numpy(np.frombuffer(data, np.uint8).copy()).to(dev)
    res = cfg5_sequence(ctx, dev, p5, table, d_c, ref5, S, (2, 4, 8))
    res["hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES")
    print(json.dumps(res))


def extra_workloads(ctx, dev, stream, steps, cfg5, cpu_md5s=None):
    """The other BASELINE configurations on the same GPU, a few steps each: cfg2, cfg3 with its
    9/7 DWT family against the HBM roofline (north_star's >= 40 % target), the cfg4 tiling and the whole cfg4 image (256
    tiles) at N = 1, the 8K workload at the reference's int32 width, and the cfg5 decode."""
    out = {}
    for name in ("cfg1", "cfg2", "cfg3", "cfg4tile"):
        Cn, W, H, prec, levels, ntiles, desc = WORKLOADS[name]
        try:
            out[name] = _encode_workload(ctx, dev, stream, steps, Cn, W, H, prec, levels, ntiles, desc, name == "cfg3",
                                         None if name == "cfg1" else name, parity=True, cpu_md5=(cpu_md5s or {}).get(name))
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": str(e)}
    # BASELINE configs[3] whole: 16384 x 16384 as 256 tiles of 1024 x 1024 on ONE GPU (the N = 1 point of its scaling curve)
    try:
        out["cfg4"] = _encode_workload(ctx, dev, stream, max(3, steps // 2), 3, 1024, 1024, 8, 5, 256,
                                       "16384x16384x3 8-bit RGB as 256 tiles of 1024x1024, RCT+5/3 lossless HTJ2K, 5 levels, one GPU "
                                       "(BASELINE configs[3] at N = 1)", False, "cfg4", parity=True)
        out["cfg4"]["value"] = round(16384.0 * 16384.0 / out["cfg4"]["ms_per_step"] / 1e3, 1)
    except Exception as e:  # noqa: BLE001
        out["cfg4"] = {"error": str(e)}
    # the headline workload at the REFERENCE's arithmetic width: int32 planes, no packed kernels (SURVEY.md §8(d) bytes as
    # written: 4 B per coefficient) -- beside the int16 path the headline runs, so that a reader sees both
    try:
        saved = {k: os.environ.get(k) for k in ("GRK_AMD_DWT_PK", "GRK_AMD_PLANES16")}
        os.environ["GRK_AMD_DWT_PK"] = "0"
        os.environ["GRK_AMD_PLANES16"] = "0"
        c32 = G.Context(dev.index or 0)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        c32.set_stream(stream.cuda_stream)
        Cn, W, H, prec, levels, ntiles, desc = WORKLOADS["8k"]
        out["8k_int32"] = _encode_workload(c32, dev, stream, steps, Cn, W, H, prec, levels, ntiles,
                                           desc + " -- int32 planes, unpacked 5/3 kernels (GRK_AMD_PLANES16=0 GRK_AMD_DWT_PK=0)",
                                           False, "8k_int32")
        c32.close()
    except Exception as e:  # noqa: BLE001
        out["8k_int32"] = {"error": str(e)}
    # What the content does to the headline shape: frames with nothing in them (one value: every band but LL empty) and the all-zero
    # frame (LL = -128: MagSgn bytes 0xFF throughout) against the G2 frame of the headline, pipelined as the headline is -- r04 found
    # and removed two cliffs here (empty quads' LDS atomics on one address, one stuffing event per window look: a flat frame cost
    # 0.80 ms, a black one 1.18; profiles/r04_small_frames.txt)
    try:
        Cn, W, H, prec, levels, ntiles, _ = WORKLOADS["8k"]
        p8 = G.TileParams.make(W, H, Cn, prec, levels)
        flat = {}
        for name, val in (("all_zero", 0), ("all_128", 128)):
            d_f = torch.full((Cn * H * W,), val, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize(dev)
            ctx.set_overlap(True); ctx.set_pipelining(True)
            with torch.cuda.stream(stream):
                for _ in range(8):
                    ctx.encode_tiles(p8, 1, d_f.data_ptr(), True, fetch=False)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            with torch.cuda.stream(stream):
                for _ in range(2 * steps):
                    ctx.encode_tiles(p8, 1, d_f.data_ptr(), True, fetch=False)
            torch.cuda.synchronize(dev)
            msf = (time.perf_counter() - t0) / (2 * steps) * 1e3
            ctx.set_pipelining(False)
            tf, _ = ctx.fetch_table(G.lib().grk_amd_tile_num_blocks(C.byref(p8)))
            flat[name] = {"ms_per_step": round(msf, 4), "value": round(W * H / msf / 1e3, 1), "unit": "Mpixels/s",
                          "coded_bytes": int(tf["length"].astype(np.int64).sum())}
            del d_f
        out["flat_8k"] = dict(flat, workload="8192x8192x3 8-bit frames of one value, the headline's settings")
    except Exception as e:  # noqa: BLE001
        out["flat_8k"] = {"error": str(e)}
    if cfg5 is not None:
        try:
            import j2kparse as J
            cs5, ref5, S = cfg5
            info = J.parse(cs5)
            p5 = G.TileParams.make(S, S, 3, 12, info["levels"], irreversible=True, mct=True, part1=True)
            blocks, _ = G.tile_layout(p5)
            rows, data = J.decode_table(info, blocks, True)
            table = np.array(rows, dtype=G.capi.CODED_DTYPE)
            ctx.set_decode_qcd([(e << 11) | m for e, m in info["qcd"]])
            d_c = torch.from_numpy(np.frombuffer(data, np.uint8).copy()).to(dev)
            d_out = torch.zeros(3 * S * S, dtype=torch.int16, device=dev)
            with torch.cuda.stream(stream):
                ctx.decode_device(p5, 1, table, d_c.data_ptr(), d_c.numel(), d_out.data_ptr())
            torch.cuda.synchronize(dev)
            ctx.decode_status()
            ctx.enable_timing(True)
            n5 = 3
            t0 = time.perf_counter()
            with torch.cuda.stream(stream):
                for _ in range(n5):
                    ctx.decode_device(p5, 1, table, d_c.data_ptr(), d_c.numel(), d_out.data_ptr())
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) / n5 * 1e3
            got = d_out.cpu().numpy().view(np.uint16).reshape(3, S, S)
            k8, k6 = ctx.kernel_ms(5)[0], ctx.kernel_ms(6)[0]
            ctx.enable_timing(False)
            # a SEQUENCE of such frames through the one context, several in flight (grk_amd_set_decode_pipelining): the later frames'
            # lane waves and long chains run beside the first's -- chain-bound kernels leave most of the machine's issue slots free.
            # Two, four and eight in flight here (r06: a frame's block decoding is ONE launch on one stream), on the HIP runtime's default 4 hardware queues and beside every other stream this process
            # has made (streams that share a queue run in turn); the same in a child process of this one started with
            # GPU_MAX_HW_QUEUES=8 -- process-wide, read at the runtime's start, and a setting the encode-over-RCCL path does not
            # like, so it is the host's to make (profiles/r04_hw_queues.txt)
            seq5 = cfg5_sequence(ctx, dev, p5, table, d_c, ref5, S, (2, 4, 8))
            seq5["hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES", "4 (runtime default)")
            try:
                import subprocess
                import tempfile
                with tempfile.TemporaryDirectory() as td:
                    fcs, fref = os.path.join(td, "cfg5.j2k"), os.path.join(td, "cfg5_ref.npy")
                    with open(fcs, "wb") as fh:
                        fh.write(cs5)
                    np.save(fref, ref5)
                    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
                    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "GROK_AMD_FORCE_DIST"):
                        env.pop(k, None)
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cfg5-sequence", fcs, fref, str(S), str(dev.index or 0)],
                                       env=env, capture_output=True, text=True, timeout=300)
                    seq5["child_process_GPU_MAX_HW_QUEUES_8"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else \
                        {"error": (r.stderr or r.stdout)[-300:]}
            except Exception as e:      # noqa: BLE001
                seq5["child_process_GPU_MAX_HW_QUEUES_8"] = {"error": str(e)}
            samples = 3.0 * S * S
            out["cfg5"] = {"workload": "8192x8192x3 12-bit decode, Part-1 EBCOT + ICT + 9/7 stream written by grk_compress (BASELINE configs[4])",
                           "ms_per_step": round(ms, 3), "value": round(S * S / ms / 1e3, 1), "unit": "Mpixels/s", "dtype": "f32",
                           "coded_bytes": int(len(data)), "pixels_equal_grk_decompress": bool(np.array_equal(got.astype(np.int32), ref5)),
                           "sequence_mode": seq5,
                           "kernels": {"t1_ebcot_decode": {"avg_ms": round(k8, 3), "algorithmic_bytes": int(4 * samples + len(data)),
                                                           "algorithmic_GBps": rate(4 * samples + len(data), k8)[0],
                                                           "frac": rate(4 * samples + len(data), k8)[1],
                                                           "traffic": _pmc_traffic("cfg5", ("t1_dec_kernel", "t1_lanes_kernel", "t1_recon_kernel")),
                                                           "frac_on_traffic": rate(_pmc_traffic("cfg5", ("t1_dec_kernel", "t1_lanes_kernel", "t1_recon_kernel")) or 0, k8)[1]},
                                       "idwt97_5levels": {"avg_ms": round(k6, 3),
                                                          "algorithmic_bytes": int(idwt_bytes(samples, 2, 4, 5, True)),
                                                          "algorithmic_GBps": rate(idwt_bytes(samples, 2, 4, 5, True), k6)[0],
                                                          "frac": rate(idwt_bytes(samples, 2, 4, 5, True), k6)[1],
                                                          "traffic": _pmc_traffic("cfg5", ("idwt_last_level_fused", "idwt_level_kernel"))}}}
            ctx.set_decode_qcd([])
        except Exception as e:  # noqa: BLE001
            out["cfg5"] = {"error": str(e)}
    return out


def host_boundary(ctx, dev, stream, params, ntiles, host_pixels, nblocks, frames):
    """What a caller that owns HOST buffers gets (never the headline `value`, whose inputs are resident
    in HBM): per frame, pinned pixels -> H2D -> encode -> D2H of the block table and of the coded bytes into pinned memory.
    The three legs run on three streams: uploads double-buffered one frame ahead of the encoder, the download one frame
    behind it (its size -- the bytes used in the coded arena -- passes through the host first), the encoder rotating three
    buffer sets (grk_amd_set_pipelining(ctx, 2)) so that it never waits for that round trip."""
    raw = host_pixels.numel()
    arena_cap = int(raw * 2)
    h_px = host_pixels.pin_memory()
    d_buf = [torch.empty(raw, dtype=torch.uint8, device=dev) for _ in range(2)]
    h_coded = torch.empty(arena_cap, dtype=torch.uint8).pin_memory()
    h_offs = torch.empty(nblocks, dtype=torch.int64).pin_memory()
    h_lens = torch.empty(nblocks, dtype=torch.int32).pin_memory()
    h_used = [torch.zeros(1, dtype=torch.int64).pin_memory() for _ in range(2)]
    up, dl = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    ev_up = [torch.cuda.Event() for _ in range(2)]
    ev_enc = [torch.cuda.Event() for _ in range(2)]          # the encode that read d_buf[b] has finished with it
    ev_used = [torch.cuda.Event() for _ in range(2)]
    ev_dl = {}
    pend = [None]
    got_bytes = [0]

    def download(prev):
        f, used_slot, offs, lens, arena = prev
        ev_used[used_slot].synchronize()                       # (long done: the next frame has been queued meanwhile)
        n = int(h_used[used_slot][0])
        with torch.cuda.stream(dl):
            h_coded[:n].copy_(arena[:n], non_blocking=True)
            h_offs.copy_(offs, non_blocking=True)
            h_lens.copy_(lens, non_blocking=True)
            e = torch.cuda.Event()
            e.record(dl)
            ev_dl[f] = e
        got_bytes[0] = n

    def one(f):
        b = f & 1
        with torch.cuda.stream(up):
            if f >= 2:
                up.wait_event(ev_enc[b])
            d_buf[b].copy_(h_px, non_blocking=True)
            ev_up[b].record(up)
        with torch.cuda.stream(stream):
            stream.wait_event(ev_up[b])
            e = ev_dl.pop(f - 3, None)
            if e is not None:
                stream.wait_event(e)                           # frame f reuses frame f - 3's buffer set
            ctx.encode_tiles(params, ntiles, d_buf[b].data_ptr(), True, fetch=False)
            ev_enc[b].record(stream)
        used = _as_tensor(ctx.table_device_ptr(2), 1, dev, "<i8")
        offs = _as_tensor(ctx.table_device_ptr(0), nblocks, dev, "<i8")
        lens = _as_tensor(ctx.table_device_ptr(1), nblocks, dev, "<i4")
        arena = _as_tensor(ctx.coded_device_ptr(), arena_cap, dev)
        ctx.stream_wait_results(dl.cuda_stream)
        prev, pend[0] = pend[0], None
        if prev is not None:
            download(prev)
        with torch.cuda.stream(dl):
            h_used[b].copy_(used, non_blocking=True)
            ev_used[b].record(dl)
        pend[0] = (f, b, offs, lens, arena)

    def flush():
        if pend[0] is not None:
            download(pend[0])
            pend[0] = None
        torch.cuda.synchronize(dev)

    ctx.set_overlap(True)
    ctx.set_pipelining(2)
    for f in range(3):
        one(f)
    flush()
    t0 = time.perf_counter()
    for f in range(frames):
        one(3 + f)
    flush()
    dt = (time.perf_counter() - t0) / frames
    ctx.set_pipelining(False)
    # what arrived is the encoder's output: the table's byte ranges lie inside the downloaded arena prefix
    ok = bool(((h_offs + h_lens.to(torch.int64)) <= got_bytes[0]).all()) and int(h_lens.sum()) > 0
    px_per_frame = params.tile_w * params.tile_h * ntiles
    return {"ms_per_frame": round(dt * 1e3, 3), "value": round(px_per_frame / dt / 1e6, 1), "unit": "Mpixels/s", "frames": frames,
            "h2d_bytes_per_frame": int(raw), "d2h_bytes_per_frame": int(got_bytes[0] + nblocks * 12),
            "h2d_GBps": round(raw / dt / 1e9, 1), "table_consistent": ok,
            "what": "pinned host pixels -> H2D (double-buffered, one frame ahead) -> encode -> D2H of block table + coded bytes "
                    "into pinned memory (one frame behind); bounded by the PCIe upload"}


def node_native(dev, tile, prec, levels, d_px, cpu_file_md5=None):
    """The native node host (grk_amd_node_*, node.cpp: one context + one host thread per device entry, Tier-2 and the codestream
    written by the host) on THIS GPU, whole codestream per call: one worker and two workers ({d} / {d, d}), the frame as 1 tile
    and as 64 tiles of 1024x1024, pixels in host memory and resident on the device, parallel writers and the gather form.
    What it shows is the host side of that path (Tier-2 + assembling ~100 MB of codestream per 8K frame); the kernels are
    the headline's."""
    Cn, H, W = tile.shape
    res = {}
    d = dev.index or 0
    for name, devices, T in (("1_worker_1_tile", [d], W), ("2_workers_64_tiles", [d, d], 1024)):
        try:
            node = G.Node(devices)
            layout = G.ImageLayout.make(W, H, T, T)
            base = G.TileParams.make(1, 1, Cn, prec, levels)
            row = {}
            out_buf = np.empty(tile.size * tile.itemsize * 2 + (1 << 20), np.uint8)
            out_buf[::4096] = 0                          # (the pages exist: a caller in a loop reuses its output buffer)
            for label, fl in (("parallel_writers", 0), ("gather", G.NODE_GATHER)):
                for src in ("host_pixels", "device_pixels"):
                    def once():
                        if src == "host_pixels":
                            return node.encode_image(layout, base, tile, fl, out=out_buf)
                        return node.encode_image_device(layout, base, d_px.data_ptr(), d_px.numel(), d, fl, out=out_buf)
                    cs = once()
                    t0 = time.perf_counter()
                    for _ in range(2):
                        cs = once()
                    row["%s_%s_ms" % (label, src)] = round((time.perf_counter() - t0) / 2 * 1e3, 2)
                    if T == W and cpu_file_md5:
                        import hashlib
                        row["file_equals_cpu_encode"] = bool(row.get("file_equals_cpu_encode", True) and hashlib.md5(cs).hexdigest() == cpu_file_md5)
            row["codestream_bytes"] = int(len(cs))
            # the same frame into a PINNED output buffer (grk_amd_host_alloc): the assembled tile-parts cross the link straight to
            # their places in the file, no host copy at all
            try:
                pin_ctx = G.Context(d)
                pin = pin_ctx.host_array(out_buf.size)
            except Exception:  # noqa: BLE001
                pin = None
            if pin is not None:
                def once_pin():
                    return node.encode_image_device(layout, base, d_px.data_ptr(), d_px.numel(), d, 0, out=pin)
                cs = once_pin()
                t0 = time.perf_counter()
                for _ in range(3):
                    cs = once_pin()
                row["parallel_writers_device_pixels_pinned_out_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
                if T == W and cpu_file_md5:
                    import hashlib
                    row["file_equals_cpu_encode"] = bool(row.get("file_equals_cpu_encode", True) and hashlib.md5(cs).hexdigest() == cpu_file_md5)
                # ... and as a SEQUENCE: three node objects on this GPU, a host thread each (ctypes releases the GIL): one frame's D2H runs
                # beside the others' encode + Tier-2 -- the route's throughput form (tools/node_two_in_flight.py)
                if T == W:
                    try:
                        import threading
                        extra = [G.Node(devices) for _ in range(2)]
                        nodes3 = [node] + extra
                        pins = [pin] + [pin_ctx.host_array(out_buf.size) for _ in range(2)]

                        def run3(i, n):
                            for _ in range(n):
                                nodes3[i].encode_image_device(layout, base, d_px.data_ptr(), d_px.numel(), d, 0, out=pins[i])
                        for i in range(3):
                            run3(i, 1)
                        th3 = [threading.Thread(target=run3, args=(i, 6)) for i in range(3)]
                        t0 = time.perf_counter()
                        for t3 in th3:
                            t3.start()
                        for t3 in th3:
                            t3.join()
                        row["three_nodes_in_flight_pinned_out_ms_per_frame"] = round((time.perf_counter() - t0) / 18 * 1e3, 2)
                        for n3 in extra:
                            n3.close()
                        del pins
                    except Exception as e:  # noqa: BLE001
                        row["three_nodes_in_flight_pinned_out_ms_per_frame"] = {"error": str(e)}
                del pin, pin_ctx
            row["tier2"] = "host plan (GRK_AMD_NODE_T2=host)" if os.environ.get("GRK_AMD_NODE_T2") == "host" else "device (grk_amd_assemble_device)"
            res[name] = row
            node.close()
        except Exception as e:  # noqa: BLE001
            res[name] = {"error": str(e)}
    return res


def via_grok_plugin(ctx, params, tile_pixels, prec, cpu_file_md5=None):
    """The drop-in route itself, once: the plugin's tile tree (GPU encode + D2H + tree) handed to the real Grok library's
    grk_compress_with_plugin(), which runs its own Tier-2 and writes the file (oracle/_ref = that library, built from the
    reference's sources: here it is the HOST the plugin plugs into)."""
    try:
        import hashlib
        import refharness as R
        if not R.have_ref():
            return {"skipped": "oracle/_ref missing"}
        plug = os.path.join(os.path.dirname(G.lib_path()), "libgrokj2k_plugin.so")
        L = C.CDLL(plug)
        L.grk_amd_plugin_tile_create.restype = C.c_void_p
        L.grk_amd_plugin_tile_create.argtypes = [C.c_void_p, C.POINTER(G.TileParams), C.c_void_p, C.c_int]
        L.grk_amd_plugin_tile_destroy.argtypes = [C.c_void_p]
        # the pixels where the plugin's own loader puts them: pinned host memory (plugin.cpp HostPixels / grk_amd_host_alloc)
        px = ctx.host_array(tile_pixels.size * tile_pixels.itemsize).view(tile_pixels.dtype).reshape(tile_pixels.shape)
        px[...] = tile_pixels
        best = None
        for _ in range(3):                                  # (the first builds the tile tree of this geometry; later frames patch it)
            t0 = time.perf_counter()
            tile = L.grk_amd_plugin_tile_create(ctx._h, C.byref(params), px.ctypes.data, 0)
            t_tile = time.perf_counter() - t0
            if not tile:
                return {"error": "grk_amd_plugin_tile_create failed"}
            try:
                t0 = time.perf_counter()
                cs, _ = R.encode(px, prec, numres=params.num_levels + 1, mode=1, rate_algo=1, plugin_tile=tile)
                t_host = time.perf_counter() - t0
            finally:
                L.grk_amd_plugin_tile_destroy(tile)
            if best is None or t_tile + t_host < best[0] + best[1]:
                best = (t_tile, t_host, cs)
        t_tile, t_host, cs = best
        npx = params.tile_w * params.tile_h
        return {"ms_per_frame": round((t_tile + t_host) * 1e3, 1), "value": round(npx / (t_tile + t_host) / 1e6, 1), "unit": "Mpixels/s",
                "plugin_tile_ms": round(t_tile * 1e3, 1), "host_library_ms": round(t_host * 1e3, 1), "codestream_bytes": len(cs),
                "file_equals_cpu_encode": (hashlib.md5(cs).hexdigest() == cpu_file_md5) if cpu_file_md5 else None,
                "what": "grk_amd_plugin_tile_create (pinned host pixels -> H2D -> GPU encode -> D2H into the tile tree's pinned "
                        "buffer -> the kept grk_plugin_tile tree of this geometry patched) + grk_compress_init/start/"
                        "grk_compress_with_plugin/end of the real Grok 8.0.2 library (image set-up, its own Tier-2, codestream "
                        "write to memory), best of 3"}
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


# (no measured N > 1 run stands behind these: 4 channels per peer at ~15-25 GB/s each saturate one link pair, 8 per communicator
#  bound a gather's workgroups to 8 CUs; reported in multi_gpu.rccl_env so that the first real run can be read against them)
RCCL_ENV_DEFAULTS = {"NCCL_MAX_NCHANNELS": "8", "NCCL_NCHANNELS_PER_PEER": "4"}


def launch_ranks(n, dry_run):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks, one per GPU, under
    torch.distributed.run on 127.0.0.1 (what the driver's own N > 1 command does).  Returns the launcher's exit code; the
    ranks' stdout is this process's stdout, so rank 0's JSON line stays the last line."""
    import socket
    import subprocess
    if not dry_run:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print("bench.py: --gpus %d asked for, %d GPU(s) visible -- refusing to report a %d-GPU number from fewer devices"
                  % (n, have, n), file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def start_gather_watchdog(timeout, out, emit, rc, key="gather"):
    """The gather regions run LAST and under this timer: if a transfer never completes, rank 0 still prints the line -- with the
    counts figures it already holds and exchange.gather = {"error": ...} -- and every rank leaves with exit code `rc`
    (--gather-timeout-rc, default 0: the line carries the failure; non-zero for callers that want the process status to say it)."""
    import threading

    def bail():
        if out is not None:
            out.setdefault("exchange", {})[key] = {"error": "the %s regions did not complete within %.0f s" % (key, timeout)}
            emit(out)
        os._exit(rc)
    dog = threading.Timer(timeout, bail)
    dog.daemon = True
    dog.start()
    return dog


_JSON_FD = None          # the process's original stdout, when fd 1 has been pointed at stderr (_own_stdout)


def _own_stdout():
    """RCCL writes a version banner to stdout through C stdio when its first communicator is made: from here on everything that goes
    to fd 1 lands on stderr, and the JSON line alone is written to the original stdout (_emit) -- ONE line, as the contract says."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def _emit(o):
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    if o is not None:
        line = json.dumps(o) + "\n"
        if _JSON_FD is None:
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stdout.flush()
            os.write(_JSON_FD, line.encode())


def dry_run(world, rank, args):
    """The launch path without kernels: the ranks form a gloo group, meet at the barrier and count each other."""
    arrived = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        arrived = int(t.item())
        n_group = dist.get_world_size()
        dist.barrier()
    else:
        n_group = 1
    line = {"metric": "encode Mpixels/s (whole node), 8K RGB HTJ2K lossless", "value": None, "unit": "Mpixels/s",
            "dry_run": True, "n_gpus": n_group, "ranks_at_barrier": arrived, "steps": args.steps,
            "warmup": args.warmup, "backend": "gloo"}
    if world > 1 and args.dry_run_stall >= 0:
        # the watchdog path without a GPU: the "counts" phase above is done, now a "gather" in which one rank never arrives
        line["exchange"] = {"counts": {"ranks": arrived}, "gather": None}
        start_gather_watchdog(args.gather_timeout, line if rank == 0 else None, _emit, args.gather_timeout_rc)
        if rank == args.dry_run_stall:
            time.sleep(3600)
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)                       # (never completes: a peer is missing)
        time.sleep(3600)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="8k", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-workloads", action="store_true", help="skip the `workloads` object (cfg2, cfg3, cfg4tile, cfg5 decode)")
    ap.add_argument("--no-host-boundary", action="store_true", help="skip the `host_boundary` object (PCIe-inclusive end-to-end "
                    "rate, the route through grk_compress_with_plugin)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not take the HBM-traffic counters in this run (two short rocprofv3 --pmc passes of the headline "
                         "workload, ~20 s); the committed summaries under profiles/ are quoted instead")
    ap.add_argument("--region-repeats", type=int, default=5,
                    help="the timed region (warm-up + steps) is run this many times; ms_per_step is the median region / steps")
    ap.add_argument("--no-overlap", action="store_true",
                    help="every kernel alone on the GPU, also in the timed region (what the rocprofv3 kernel-trace summary "
                         "that the roofline durations are checked against is taken with)")
    ap.add_argument("--exchange", default="best", choices=("best", "gather", "counts"),
                    help="N > 1: which exchange the headline `value` is (both are always timed and reported, `exchange` in the "
                         "line; `config.headline_exchange` says which one the value is): 'best' (default) = whichever of the two is "
                         "faster on this node -- DESIGN.md section 6's arithmetic says the gather is bound by xGMI ingress at N = 8 and "
                         "the parallel writers are not, and no measured run has decided it --; 'gather' = every rank's coded tile-parts (exact sizes) over xGMI to the frame's writer rank, "
                         "which rotates with the frame number, --gather-depth frames in flight; 'counts' = all_gather of the coded "
                         "byte counts only (parallel writers: the bytes leave each GPU over its own PCIe link)")
    ap.add_argument("--gather-depth", type=int, default=4, choices=range(1, 6),
                    help="gathers in flight at once (each on its own communicator and stream; a gather is issued two frames behind "
                         "the encoder, which then rotates depth + 3 buffer sets)")
    ap.add_argument("--gather-timeout", type=float, default=240.0,
                    help="N > 1: seconds the gather regions may take before the line is printed without them (a transfer that "
                         "never completes must not cost the run its counts figure)")
    ap.add_argument("--gather-timeout-rc", type=int, default=0,
                    help="exit code of every rank when the gather watchdog fires (the line is printed first, with the counts figures)")
    ap.add_argument("--dry-run-stall", type=int, default=-1,
                    help="with --dry-run: after the barrier this rank never joins the next collective -- the gather watchdog's path")
    ap.add_argument("--cfg5-sequence", nargs=4, metavar=("CODESTREAM", "REF_NPY", "S", "DEVICE"), default=None,
                    help="(internal) child process of the cfg5 leg: the Part-1 stream as a sequence with 2, 3 and 6 frames in flight")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check without a GPU: the ranks meet over gloo, count each other at the barrier and rank 0 prints "
                         "a line with n_gpus = the ranks that really arrived; no kernels run")
    args = ap.parse_args()
    if args.cfg5_sequence:
        fcs, fref, S5, dv = args.cfg5_sequence
        return cfg5_sequence_child(fcs, fref, int(S5), int(dv))

    # `python bench.py --gpus N` starts its N ranks ITSELF (the reference's analogue is its in-process tile pool,
    # codestream/CodeStreamCompress.cpp:535-603: one command, all workers); under a launcher (WORLD_SIZE set) it is one rank
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, args.dry_run))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if args.dry_run:
        return dry_run(world, rank, args)
    # GROK_AMD_FORCE_DIST=1 exercises the exchange step on a 1-GPU box (world_size 1 over RCCL)
    use_dist = world > 1 or os.environ.get("GROK_AMD_FORCE_DIST") == "1"
    if use_dist:
        _own_stdout()
        # RCCL's footprint, bounded unless the caller says otherwise: a gather is a point-to-point transfer over ONE xGMI link pair, depth
        # of them run side by side on communicators of their own next to a coder that wants every CU -- a channel is a workgroup
        # that holds a CU, and the default (up to 32 per communicator) is sized for ring collectives over all links at once
        for k, v in RCCL_ENV_DEFAULTS.items():
            os.environ.setdefault(k, v)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                      # (a launcher sets it; the forced world-1 run picks a free one)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", rank=rank, world_size=world, pg_options=D.comm_options())
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    Cn, W, H, prec, levels, ntiles, desc = WORKLOADS[args.workload]
    irrev = args.workload == "cfg3"
    params = G.TileParams.make(W, H, Cn, prec, levels, irreversible=irrev)
    ctx = G.Context(local_rank)
    if args.no_overlap:
        ctx.set_overlap(False)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)

    # synthetic tile(s): generator G2 (bounded gradient + LCG noise), same buffer for every tile
    tile = synth.g2(Cn, H, W, prec)
    host = np.ascontiguousarray(np.broadcast_to(tile.reshape(1, -1), (ntiles, tile.size))).reshape(-1)
    d_px = torch.from_numpy(host.view(np.uint8)).to(dev)
    nblocks = G.lib().grk_amd_tile_num_blocks(C.byref(params)) * ntiles
    pixels_per_step = W * H * ntiles
    samples = pixels_per_step * Cn

    # rank 0 owns the header blob; everybody gets it by broadcast (tiny, outside the timed region)
    if use_dist:
        got = D.broadcast_params(params, dev)
        assert bytes(got) == bytes(params)

    # one encode up front: buffers exist
    ctx.encode_tiles(params, ntiles, d_px.data_ptr(), True, fetch=False)
    _, total0 = ctx.fetch_table(nblocks)
    comm = D.independent_stream(ctx, dev) if use_dist else None

    def sync():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def run_frames(prm, nt, d_pixels, nblk, exchange, steps, warmup, gather_depth=None):
        """`warmup` + `steps` frames of `nt` tiles on this rank, each followed by `exchange` (None / "counts" / "gather");
        consecutive frames are pipelined (the next frame's DWT runs while this one's blocks are still being coded and its
        tile-parts travel).  Returns (seconds for `steps` frames, MAX over ranks; the gathered parts of the last frame on its
        writer; that writer's rank)."""
        raw = nt * prm.num_comps * prm.tile_w * prm.tile_h * ((prm.prec + 7) // 8)
        arena_cap = int(raw * 2)                # grk_amd allocates >= 2 x raw (context.hip: make_ht_args)
        pipe = [None]
        cbuf = [None, None]

        depth, lag = max(1, gather_depth or args.gather_depth), 2

        def one(f):
            with torch.cuda.stream(stream):
                if pipe[0] is not None:                        # depth + lag + 1 buffer sets in rotation: this frame reuses frame
                    ev = pipe[0].done_event(f - (depth + lag + 1))   # f - (depth + lag + 1)'s
                    if ev is not None:
                        stream.wait_event(ev)                  # ... once its gather has read them
                src = d_pixels[f % len(d_pixels)] if isinstance(d_pixels, (list, tuple)) else d_pixels
                ctx.encode_tiles(prm, nt, src.data_ptr(), True, fetch=False)
                if exchange is None:
                    return
                used = _as_tensor(ctx.table_device_ptr(2), 1, dev, "<i8")
                if exchange == "counts":
                    if cbuf[f & 1] is None:
                        cbuf[f & 1] = torch.empty(world, dtype=torch.int64, device=dev)
                    ctx.stream_wait_results(comm.cuda_stream)
                    with torch.cuda.stream(comm):
                        dist.all_gather_into_tensor(cbuf[f & 1], used)      # 8 bytes per rank, out of the encoder's own word
                elif exchange == "parts":
                    # Tier-2 on the device, queued on the exchange's stream behind the frame's results: what travels is the rank's FINISHED
                    # tile-parts -- one table row (place, length) per tile-part instead of one per code-block
                    ctx.assemble_device_async(prm, list(range(rank, world * nt, world)), 0, comm.cuda_stream)
                    pipe[0].submit(f, _as_tensor(ctx.assembled_table_ptr(2), 1, dev, "<i8"), _as_tensor(ctx.assembled_table_ptr(0), nt, dev, "<i8"),
                                   _as_tensor(ctx.assembled_table_ptr(1), nt, dev, "<i4"), _as_tensor(ctx.assembled_device_ptr(), arena_cap, dev),
                                   wait_results=None)
                    if args.no_overlap:
                        pipe[0].flush()
                else:
                    offs = _as_tensor(ctx.table_device_ptr(0), nblk, dev, "<i8")
                    lens = _as_tensor(ctx.table_device_ptr(1), nblk, dev, "<i4")
                    arena = _as_tensor(ctx.coded_device_ptr(), arena_cap, dev)
                    pipe[0].submit(f, used, offs, lens, arena, wait_results=ctx.stream_wait_results)
                    if args.no_overlap:
                        pipe[0].flush()                         # one buffer set: the gather cannot lag behind the encoder

        def flush():
            if pipe[0] is not None:
                return pipe[0].flush()
            return None, None

        ctx.set_pipelining(False if args.no_overlap else (depth + lag if exchange in ("gather", "parts") else True))
        if exchange in ("gather", "parts"):
            pipe[0] = D.FramePipeline(dev, (stream, comm), depth=1 if args.no_overlap else depth, lag=1 if args.no_overlap else lag, ctx=ctx)
        for f in range(warmup):
            one(f)
        flush()
        sync()
        ctx.enable_timing(False)         # nothing but the hot path inside the timed region
        t0 = time.perf_counter()
        for f in range(steps):
            one(warmup + f)
        parts, root = flush()
        sync()
        secs = time.perf_counter() - t0
        ctx.set_pipelining(False)
        if use_dist:
            t = torch.tensor([secs], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            secs = float(t.item())
        return secs, parts, root

    gather_md5 = [None]

    def assembled_bytes(prm, nt, nblk, img_w, img_h, parts, root):
        """The frame the last gather delivered really is a codestream: Tier-2 + headers over it on its writer."""
        n = torch.zeros(1, dtype=torch.int64, device=dev)
        if parts is not None and rank == root:
            ft, fc = D.merge_tile_parts(D.parts_to_numpy(parts), world * nt, nblk // nt)
            if nt == 1 or img_w:
                import hashlib
                cs_g = G.write_codestream(prm, img_w, img_h, ft, fc)
                n[0] = len(cs_g)
                gather_md5[0] = hashlib.md5(cs_g).hexdigest()
            else:
                n[0] = int(ft["length"].sum())
        dist.all_reduce(n, op=dist.ReduceOp.MAX)
        return int(n.item())

    def assembled_file(prm, nt, img_w, img_h, parts, root):
        """The "parts" exchange's last frame on its writer: main header + the ranks' finished tile-parts in tile order + EOC.
        Returns (length, md5) -- (sum of the tile-parts' lengths, None) for a layout the header writer is not asked for here."""
        import hashlib
        n = torch.zeros(1, dtype=torch.int64, device=dev)
        md5 = None
        if parts is not None and rank == root:
            ft, fc = D.merge_tile_parts(D.parts_to_numpy(parts), world * nt, 1)
            body = b"".join(bytes(fc[int(r["offset"]):int(r["offset"]) + int(r["length"])]) for r in ft)
            if nt == 1 or img_w:
                cs = G.write_main_header(prm, img_w, img_h, 0, None) + body + b"\xff\xd9"
                n[0] = len(cs)
                md5 = hashlib.md5(cs).hexdigest()
            else:
                n[0] = len(body)
        dist.all_reduce(n, op=dist.ReduceOp.MAX)
        return int(n.item()), md5

    pipelined = not args.no_overlap
    # N > 1: the counts exchange first -- no data-path transfer, nothing that can stall --; the gather regions run LAST (below),
    # under a watchdog, so that a transfer that never completes costs the line its gather figures and nothing else
    exchange = ("counts" if use_dist else None)
    # The GPU's clocks take some ten milliseconds of load to settle (tools measurement, r03: the first region of 20 frames after an
    # idle second 0.458-0.476 ms per frame, every later one 0.435-0.444): the job's steady state is what the metric is about, so a
    # fixed stretch of the same encodes runs before the W warm-up steps (untimed, reported in config.prewarm_steps)
    PREWARM = 40
    # THREE distinct frames in rotation (other noise seeds of the same generator): consecutive timed frames never read the same
    # 201 MB of pixels, so the 256 MiB Infinity Cache cannot be what serves them; the same region over the
    # ONE buffer is timed next to it (config.single_input_buffer_ms_per_step)
    rot = [d_px]
    for sd in (777, 424242):
        t2 = synth.g2(Cn, H, W, prec, seed=sd)
        h2 = np.ascontiguousarray(np.broadcast_to(t2.reshape(1, -1), (ntiles, t2.size))).reshape(-1)
        rot.append(torch.from_numpy(h2.view(np.uint8)).to(dev))
        del t2, h2
    run_frames(params, ntiles, rot, nblocks, exchange, PREWARM, 0)      # (the same frames, the same exchange: every path warm)
    # The timed region (W warm-up + exactly K steps between barriers) is 8 ms long at 8K: it runs REGION_REPEATS times and the
    # line reports the MEDIAN region (config.region_repeats carries every one), so that a 1 % A/B means something
    regions = []
    for _ in range(max(1, args.region_repeats)):
        dt_r, last_parts, last_root = run_frames(params, ntiles, rot, nblocks, exchange, args.steps, args.warmup)
        regions.append(dt_r)
    dt = sorted(regions)[len(regions) // 2]
    dt_single, _, _ = run_frames(params, ntiles, d_px, nblocks, exchange, args.steps, args.warmup)
    multi_gpu = None
    cs_len = 0
    if use_dist:
        seen = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(seen)                      # every rank that is really in the RCCL group adds its one
        multi_gpu = {"world_size": dist.get_world_size(), "ranks_seen_by_rccl": int(seen.item()),
                     "backend": "nccl (RCCL %s)" % ".".join(str(v) for v in torch.cuda.nccl.version()),
                     "gather_depth": args.gather_depth,
                     "rccl_env": {k: os.environ.get(k) for k in sorted(set(RCCL_ENV_DEFAULTS) | {"NCCL_MIN_NCHANNELS", "GPU_MAX_HW_QUEUES"})}}
        multi_gpu["replica_%s" % args.workload] = {
            "shape": "one %dx%d tile per rank and frame (a %dx%d frame), weak scaling" % (W, H, W * world, H) if ntiles == 1 else desc,
            "gather": None,
            "counts": {"ms_per_step": round(dt / args.steps * 1e3, 4), "Mpixels_s": round(pixels_per_step * world * args.steps / dt / 1e6, 1)},
            "note": "gather = coded tile-parts + block tables to the frame's writer (rank f mod N), exact sizes, issued one frame "
                    "behind the encoder, gather_depth frames in flight; counts = all_gather of 8 bytes per rank (parallel writers)"}
        # BASELINE configs[3]: 16384x16384 as 256 tiles of 1024x1024, the tiles split over the ranks (strong scaling)
        cfg4 = None
        if not args.no_workloads and args.workload == "8k" and 256 % world == 0:
            p4 = G.TileParams.make(1024, 1024, 3, 8, 5)
            nt4 = 256 // world
            t4 = synth.g2(3, 1024, 1024, 8)
            h4 = np.ascontiguousarray(np.broadcast_to(t4.reshape(1, -1), (nt4, t4.size))).reshape(-1)
            d4 = torch.from_numpy(h4.view(np.uint8)).to(dev)
            nb4 = G.lib().grk_amd_tile_num_blocks(C.byref(p4)) * nt4
            ctx.encode_tiles(p4, nt4, d4.data_ptr(), True, fetch=False)
            ctx.synchronize()
            st4 = max(3, args.steps // 2)
            d_t, _, _ = run_frames(p4, nt4, d4, nb4, "counts", st4, 2)
            multi_gpu["cfg4_strong"] = {"shape": "16384x16384x3 8-bit as 256 tiles of 1024x1024, %d tiles per rank and frame, "
                                                 "strong scaling (BASELINE configs[3])" % nt4, "gather": None,
                                        "counts": {"ms_per_step": round(d_t / st4 * 1e3, 4), "Mpixels_s": round(16384.0 * 16384.0 * st4 / d_t / 1e6, 1)}}
            cfg4 = (p4, nt4, d4, nb4, st4)
            # back to the headline shape for the per-kernel passes below
            ctx.encode_tiles(params, ntiles, d_px.data_ptr(), True, fetch=False)
            ctx.synchronize()

    if rank == 0 and not use_dist and not args.no_live_pmc:
        torch.cuda.synchronize(dev)
        live_pmc_traffic(args.workload)          # this run's own HBM-traffic counters (encode and decode families of the workload)

    # ---- the decode direction on the blocks just produced (HBM-resident coded bytes -> pixels) ----
    decode = None
    if not use_dist:
        table_d, total_d = ctx.fetch_table(nblocks)
        d_back = torch.empty_like(d_px)
        with torch.cuda.stream(stream):
            for _ in range(2):
                ctx.decode_device(params, ntiles, table_d, ctx.coded_device_ptr(), total_d, d_back.data_ptr())
        torch.cuda.synchronize(dev)
        dsteps = max(3, min(args.steps, 10))
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for _ in range(dsteps):
                ctx.decode_device(params, ntiles, table_d, ctx.coded_device_ptr(), total_d, d_back.data_ptr())
        torch.cuda.synchronize(dev)
        ddt = time.perf_counter() - t0
        # the kernel families one at a time, with HIP events around them (a pass of its own: the events cost the timed calls
        # a few microseconds each, and with the overlap on K5b of the top resolution runs beside the small inverse levels)
        ctx.set_overlap(False)
        ctx.enable_timing(True)
        with torch.cuda.stream(stream):
            for _ in range(dsteps):
                ctx.decode_device(params, ntiles, table_d, ctx.coded_device_ptr(), total_d, d_back.data_ptr())
        torch.cuda.synchronize(dev)
        try:
            ctx.decode_status()
            decode_err = None
        except RuntimeError as e:       # e.g. full-scale content: both decoders reject U_q > missing_msbs (defect D5)
            decode_err = str(e)
        dk = {name: ctx.kernel_ms(idx) for idx, name in ((5, "ht_cleanup_decode"), (6, "idwt53_5levels"), (7, "egress_mct"))}
        ctx.enable_timing(False)
        ctx.set_overlap(not args.no_overlap)
        b_in_d = (prec + 7) // 8
        b_pl_d, _ = ctx.plane_sample_bytes(params, decode=True)
        coded_sum_d = int(table_d["length"].astype(np.int64).sum())
        # K7 fused into the last inverse level (no egress launches): that launch writes the pixels instead of a plane
        dalgo = {"ht_cleanup_decode": ht_bytes(samples, b_pl_d, coded_sum_d),
                 "idwt53_5levels": idwt_bytes(samples, b_in_d, b_pl_d, levels, dk["egress_mct"][1] == 0),
                 "egress_mct": samples * (4.0 + b_in_d)}
        dtraffic = {"ht_cleanup_decode": _pmc_traffic(args.workload, ("ht_dec_vlc_kernel", "ht_dec_ms_kernel")),
                    "idwt53_5levels": _pmc_traffic(args.workload, ("idwt_last_level_fused", "idwt_level_kernel")), "egress_mct": None}
        decode = {"value": round(pixels_per_step * dsteps / ddt / 1e6, 1), "unit": "Mpixels/s",
                  "ms_per_step": round(ddt / dsteps * 1e3, 4), "steps": dsteps, "lossless_round_trip": (bool(torch.equal(d_back, d_px)) if not irrev else None),
                  "max_abs_error": (int((d_back.view(torch.int16 if prec > 8 else torch.uint8).to(torch.int32) -
                                         d_px.view(torch.int16 if prec > 8 else torch.uint8).to(torch.int32)).abs().max().item())
                                    if irrev else 0),
                  "plane_bytes_per_coefficient": b_pl_d,
                  "kernels": {k: {"avg_ms": round(v[0], 4), "launches": v[1], "algorithmic_bytes": int(dalgo[k]),
                                  "algorithmic_GBps": rate(dalgo[k], v[0])[0], "frac": rate(dalgo[k], v[0])[1],
                                  "traffic": dtraffic[k], "frac_on_traffic": rate(dtraffic[k], v[0])[1] if dtraffic[k] else None}
                              for k, v in dk.items()}}
        if decode_err:
            decode["rejected"] = decode_err
            decode["lossless_round_trip"] = None
        elif ntiles == 1 and levels >= 1 and prec <= 16:
            # region (windowed) decode of the same tile: a centred window of 1/64 of the area -- only the code-blocks and
            # the parts of each DWT level the window depends on are computed (grk_amd_decode_region)
            wx0, wy0 = (W // 2 - W // 16) & ~1, (H // 2 - H // 16) & ~1
            wx1, wy1 = wx0 + W // 8, wy0 + H // 8
            d_win = torch.empty((wx1 - wx0) * (wy1 - wy0) * Cn * ((prec + 7) // 8), dtype=torch.uint8, device=dev)
            try:
                with torch.cuda.stream(stream):
                    for _ in range(2):
                        ctx.decode_region_device(params, table_d, ctx.coded_device_ptr(), total_d, wx0, wy0, wx1, wy1, d_win.data_ptr())
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                with torch.cuda.stream(stream):
                    for _ in range(dsteps):
                        ctx.decode_region_device(params, table_d, ctx.coded_device_ptr(), total_d, wx0, wy0, wx1, wy1, d_win.data_ptr())
                torch.cuda.synchronize(dev)
                rdt = (time.perf_counter() - t0) / dsteps
                full = d_back.view(Cn, H, W * ((prec + 7) // 8)) if prec <= 8 else None
                ok = None
                if prec <= 8 and not irrev:
                    ok = bool(torch.equal(d_win.view(Cn, wy1 - wy0, wx1 - wx0), full[:, wy0:wy1, wx0:wx1]))
                decode["region"] = {"window": [wx0, wy0, wx1, wy1], "ms_per_step": round(rdt * 1e3, 4),
                                    "equals_crop_of_full_decode": ok}
            except Exception as e:        # noqa: BLE001
                decode["region"] = {"error": str(e)}

    # ---- the decode direction in its throughput form: THREE frames per call (a batch of tiles of one geometry, what
    #      grk_amd_decode_tiles takes).  K5a is one serial chain per code-block at under two waves per SIMD: its time is the
    #      chain's, not the block count's, so a second and third frame ride along for little -- as consecutive encodes are
    #      pipelined, consecutive frames of a sequence are decoded a few per call
    if decode is not None and decode.get("rejected") is None and ntiles == 1 and args.workload == "8k":
        decode["frames_per_call"] = {}
        for NB in (2, 3):
            try:
                ctx.enable_timing(False)
                d3 = d_px.repeat(NB)
                ctx.encode_tiles(params, NB, d3.data_ptr(), True, fetch=False)
                t3, tot3 = ctx.fetch_table(nblocks * NB)
                back3 = torch.empty_like(d3)
                with torch.cuda.stream(stream):
                    for _ in range(3):
                        ctx.decode_device(params, NB, t3, ctx.coded_device_ptr(), tot3, back3.data_ptr())
                torch.cuda.synchronize(dev)
                n3 = max(4, min(args.steps, 10))
                t0 = time.perf_counter()
                with torch.cuda.stream(stream):
                    for _ in range(n3):
                        ctx.decode_device(params, NB, t3, ctx.coded_device_ptr(), tot3, back3.data_ptr())
                torch.cuda.synchronize(dev)
                dt3 = (time.perf_counter() - t0) / n3
                ctx.decode_status()
                decode["frames_per_call"][str(NB)] = {"ms_per_frame": round(dt3 / NB * 1e3, 4), "value": round(pixels_per_step * NB / dt3 / 1e6, 1),
                                                      "unit": "Mpixels/s", "lossless_round_trip": bool(torch.equal(back3, d3))}
                del d3, back3
            except Exception as e:  # noqa: BLE001
                decode["frames_per_call"][str(NB)] = {"error": str(e)}
        ctx.encode_tiles(params, ntiles, d_px.data_ptr(), True, fetch=False)      # back to the headline shape
        ctx.synchronize()
        # ---- ... or TWO / THREE CONTEXTS fed in turn, one frame per call each (a sequence decoded round-robin): K5a's serial
        #      chains leave most of the machine idle, and another context's kernels take what is free
        decode["contexts_in_turn"] = {}
        try:
            table_d, total_d = ctx.fetch_table(nblocks)      # (the arena was written again: the blocks have new offsets)
            others = [G.Context(dev.index) for _ in range(2)]
            ctxs = [ctx] + others
            backs = [d_back] + [torch.empty_like(d_back) for _ in others]
            # (the others keep the streams they created: kernels of different contexts must not queue behind each other)
            for nc in (2, 3):
                for _ in range(2):
                    for i in range(nc):
                        ctxs[i].decode_device(params, ntiles, table_d, ctx.coded_device_ptr(), total_d, backs[i].data_ptr())
                torch.cuda.synchronize(dev)
                n4 = max(4, min(args.steps, 10))
                t0 = time.perf_counter()
                for _ in range(n4):
                    for i in range(nc):
                        ctxs[i].decode_device(params, ntiles, table_d, ctx.coded_device_ptr(), total_d, backs[i].data_ptr())
                torch.cuda.synchronize(dev)
                dt4 = (time.perf_counter() - t0) / (n4 * nc)
                for i in range(nc):
                    ctxs[i].decode_status()
                decode["contexts_in_turn"][str(nc)] = {"ms_per_frame": round(dt4 * 1e3, 4), "value": round(pixels_per_step / dt4 / 1e6, 1),
                                                       "unit": "Mpixels/s",
                                                       "lossless_round_trip": all(bool(torch.equal(b, d_px)) for b in backs[:nc])}
            del others, ctxs, backs
        except Exception as e:        # noqa: BLE001
            decode["contexts_in_turn"] = {"error": str(e)}

    # ---- ... and the same from ONE context: grk_amd_set_decode_pipelining(ctx, n) -- n internal buffer / stream sets in turn
    if decode is not None and decode.get("rejected") is None and ntiles == 1 and args.workload == "8k":
        decode["sequence_mode"] = {}
        try:
            # (a decode on this context clears the status block the encoder's allocator shares: a fresh encode, then its table)
            ctx.encode_tiles(params, ntiles, d_px.data_ptr(), True, fetch=False)
            table_d, total_d = ctx.fetch_table(nblocks)
            backs = [d_back] + [torch.empty_like(d_back) for _ in range(2)]
            for nfl in (2, 3):
                ctx.set_decode_pipelining(nfl)
                for k in range(2 * nfl):
                    ctx.decode_device(params, ntiles, table_d, ctx.coded_device_ptr(), total_d, backs[k % nfl].data_ptr())
                ctx.synchronize()
                n5 = max(4, min(args.steps, 10)) * nfl
                t0 = time.perf_counter()
                for k in range(n5):
                    ctx.decode_device(params, ntiles, table_d, ctx.coded_device_ptr(), total_d, backs[k % nfl].data_ptr())
                ctx.synchronize()
                dt5 = (time.perf_counter() - t0) / n5
                ctx.decode_status()
                decode["sequence_mode"][str(nfl)] = {"ms_per_frame": round(dt5 * 1e3, 4), "value": round(pixels_per_step / dt5 / 1e6, 1),
                                                     "unit": "Mpixels/s",
                                                     "lossless_round_trip": all(bool(torch.equal(b, d_px)) for b in backs[:nfl])}
            ctx.set_decode_pipelining(0)
            best = min(decode["sequence_mode"].values(), key=lambda v: v["ms_per_frame"])
            # the decode figure of the line: a sequence of frames through one context, as the encode figure is a sequence of frames
            decode["one_frame_at_a_time_ms_per_step"] = decode["ms_per_step"]
            decode["ms_per_step"] = best["ms_per_frame"]
            decode["value"] = best["value"]
            decode["mode"] = "sequence of frames, grk_amd_set_decode_pipelining (best of 2 / 3 frames in flight)"
            del backs
        except Exception as e:        # noqa: BLE001
            decode["sequence_mode"] = {"error": str(e)}
            try:
                ctx.set_decode_pipelining(0)
            except Exception:         # noqa: BLE001
                pass

    # ---- per-kernel-family durations: HIP events on the stream each kernel is launched on, `steps` more encodes.
    # K3 runs up to three times per step -- top resolution on a side stream beside DWT levels >= 1 (timer 4), the rest
    # on the context's stream (2), large-LDS classes on a second side stream (8) -- its time per step is their sum.
    def family_pass(overlap):
        ctx.set_overlap(overlap)
        ctx.enable_timing(True)
        with torch.cuda.stream(stream):
            for _ in range(args.steps):
                ctx.encode_tiles(params, ntiles, d_px.data_ptr(), True, fetch=False)
        torch.cuda.synchronize(dev)
        out = {}
        for idx, name in ((0, "ingest_mct"), (1, "dwt53_5levels")):
            out[name] = ctx.kernel_ms(idx)
        parts = [ctx.kernel_ms(i) for i in (2, 4, 8)]
        n_ht = max([n for _, n in parts] + [0])
        out["ht_cleanup_encode"] = (sum(ms * n for ms, n in parts) / n_ht if n_ht else 0.0, n_ht)
        ctx.enable_timing(False)
        return out
    # as the timed region ran (kernels of the two directions of the pipeline share the GPU and stretch each other) ...
    fam_overlapped = family_pass(not args.no_overlap)
    # ... and one kernel at a time: the durations a kernel's roofline figure is about
    fam = family_pass(False)
    ctx.set_overlap(not args.no_overlap)
    table, total = ctx.fetch_table(nblocks)
    # north_star: "bit-exact lossless HTJ2K encode of an 8K x 8K image at 1 and 8 GPUs" -- every rank turns the blocks of its
    # last encode into the tile's codestream and compares its md5 with that of the file Grok 8.0.2's own encoder writes for
    # this image (the constant the parity tests pin against the reference built here, tests/test_gpu_stages.py)
    bit_exact = None
    if args.workload in GOLDEN_MD5 and ntiles == 1:
        import hashlib
        cs_md5 = hashlib.md5(G.write_codestream(params, W, H, table, ctx.fetch_coded(total))).hexdigest()
        ok = torch.tensor([1 if cs_md5 == GOLDEN_MD5[args.workload] else 0], dtype=torch.int32, device=dev)
        if use_dist:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        bit_exact = {"codestream_md5": cs_md5, "equals_grok_cpu_file_on_all_ranks": bool(ok.item()), "ranks": world}
    b_in = (prec + 7) // 8
    b_pl, pk_levels = ctx.plane_sample_bytes(params)
    coded_sum = int(table["length"].astype(np.int64).sum())      # B_out = the blocks' bytes (the arena's extent `total` carries chunk slack)
    fused_in = fam["ingest_mct"][1] == 0                         # K1 folded into DWT level 0: that launch reads the pixels
    algo = {"ingest_mct": samples * (b_in + 4), "dwt53_5levels": dwt_bytes(samples, b_in, b_pl, levels, fused_in),
            "ht_cleanup_encode": ht_bytes(samples, b_pl, coded_sum)}
    fam_kernels = {"ingest_mct": ("ingest_kernel",), "dwt53_5levels": ("dwt_level0_fused", "dwt_levels_1plus"),
                   "ht_cleanup_encode": ("ht_encode_kernel",)}
    dom = max(("ingest_mct", "dwt53_5levels", "ht_cleanup_encode"), key=lambda k: fam[k][0])
    # HBM traffic per launch of each family: PMC counters cannot be read from inside this process; they come from the
    # committed rocprofv3 --pmc passes of this same command (profiles/summarize_pmc.py: FETCH_SIZE x1024 x2 [gfx950
    # half-count] + WRITE_SIZE x1024).
    traffic = {k: _pmc_traffic(args.workload, v) for k, v in fam_kernels.items()}
    ach, frac = rate(algo[dom], fam[dom][0])
    roofline = {"kernel": dom, "bound": "hbm", "achieved": ach or 0.0, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": frac or 0.0, "traffic": traffic[dom],
                "frac_on_traffic": rate(traffic[dom], fam[dom][0])[1] if traffic[dom] else None,
                "traffic_source": _PMC_SOURCE.get(args.workload),
                "algorithmic_bytes_per_launch": int(algo[dom]), "avg_launch_ms": round(fam[dom][0], 4),
                "launches": fam[dom][1], "plane_bytes_per_coefficient": b_pl,
                "bytes_convention": "SURVEY.md 8(d) per-unit figures at the storage width the launched instances use: %d B per "
                                    "coefficient read (planes are int%d here) + the coded bytes written (sum of the block lengths); "
                                    "workloads.8k_int32 is the same frame on the reference's int32 planes" % (b_pl, 8 * b_pl),
                "measured": "HIP events around the kernel's launches, kernels one at a time (grk_amd_set_overlap(0)); "
                            "the timed region runs K3 of the top resolution beside DWT levels >= 1, see kernels_overlapped"}
    kernels = {k: {"avg_ms": round(v[0], 4), "launches": v[1], "algorithmic_bytes": int(algo[k]) if v[1] else 0,
                   "algorithmic_GBps": rate(algo[k], v[0])[0], "frac": rate(algo[k], v[0])[1],
                   "traffic": traffic[k] if v[1] else None,
                   "frac_on_traffic": rate(traffic[k], v[0])[1] if (traffic[k] and v[1]) else None}
               for k, v in fam.items()}
    if not use_dist:
        parallelism = "1 GPU, consecutive encodes pipelined (grk_amd_set_pipelining)" if pipelined else "1 GPU"
    else:
        parallelism = ("tile-sharded x%d; per frame either the coded tile-parts (exact sizes) gathered over RCCL on the frame's "
                       "writer rank (rotating, %d gathers in flight) or only the byte counts (parallel writers): `exchange`" % (world, args.gather_depth))
    kernels_overlapped = {k: {"avg_ms": round(v[0], 4), "launches": v[1]} for k, v in fam_overlapped.items()}
    # whole-pipeline figure: the families' algorithmic bytes as launched (fused level 0, storage width)
    pipeline_bytes = algo["dwt53_5levels"] + algo["ht_cleanup_encode"] + (algo["ingest_mct"] if not fused_in else 0.0)

    pipe_traffic = _pmc_traffic(args.workload, ("ingest_kernel", "dwt_level0_fused", "dwt_levels_1plus", "ht_encode_kernel"))
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = pixels_per_step * world * args.steps / dt / 1e6
        out = {
            "metric": "encode Mpixels/s (whole node), 8K RGB HTJ2K lossless",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": (dist.get_world_size() if use_dist else 1), "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype_label(irrev, b_pl), "data": "synthetic",
            "config": {"workload": desc, "tiles_per_gpu": ntiles, "code_blocks_per_gpu": int(nblocks),
                       "coded_bytes_per_gpu": coded_sum, "arena_bytes_used_per_gpu": int(total), "packed_dwt_levels": pk_levels,
                       "generator": "G2 (SURVEY.md §8d)", "parallelism": parallelism, "prewarm_steps": PREWARM,
                       "input_frames_in_rotation": len(rot),
                       "region_repeats": {"n": len(regions), "ms_per_step": [round(r / args.steps * 1e3, 4) for r in regions],
                                          "median": round(dt / args.steps * 1e3, 4), "min": round(min(regions) / args.steps * 1e3, 4),
                                          "max": round(max(regions) / args.steps * 1e3, 4)},
                       "single_input_buffer_ms_per_step": round(dt_single / args.steps * 1e3, 4)},
            "roofline": roofline,
            # the whole step against the HBM roofline: algorithmic bytes of its kernel families as launched, and what the PMC
            # counters saw per step (`dram_*`)
            "pipeline": {"algorithmic_bytes_per_step": int(pipeline_bytes),
                         "algorithmic_GBps": rate(pipeline_bytes, ms_per_step)[0],
                         "algorithmic_frac_of_hbm_peak": rate(pipeline_bytes, ms_per_step)[1],
                         "dram_traffic_bytes_per_step": pipe_traffic,
                         "dram_GBps": round(pipe_traffic / (ms_per_step * 1e-3) / 1e9, 1) if pipe_traffic else None,
                         "dram_frac_of_hbm_peak": round(pipe_traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if pipe_traffic else None},
            "bit_exact": bit_exact,
            "kernels": kernels,
            "kernels_overlapped": kernels_overlapped,
            "decode": decode,
        }
        if use_dist:
            out["multi_gpu"] = multi_gpu
        cfg5 = None
        if not args.no_cpu_baseline and world == 1:
            # rank 0 at N = 1 only (at N > 1 the other ranks would wait in the final barrier for it, and the N = 1 line of the same
            # box has it): Grok's CPU encoder on this box's host cores
            out["cpu_baseline"], cfg5 = cpu_baseline(os.cpu_count() or 1,
                                                     want_cfg5=world == 1 and not use_dist and not args.no_workloads and args.workload == "8k")
        else:
            out["cpu_baseline"] = None
        if world == 1 and not use_dist and not args.no_workloads and args.workload == "8k":
            cb = out.get("cpu_baseline") or {}
            out["workloads"] = extra_workloads(ctx, dev, stream, max(3, min(args.steps, 10)), cfg5,
                                               {"cfg2": (cb.get("cfg2") or {}).get("file_md5")})
        if world == 1 and not use_dist and not args.no_host_boundary:
            try:
                hb = {"end_to_end": host_boundary(ctx, dev, stream, params, ntiles, torch.from_numpy(host.view(np.uint8)), nblocks,
                                                  max(3, min(args.steps, 10)))}
            except Exception as e:  # noqa: BLE001
                hb = {"end_to_end": {"error": str(e)}}
            if args.workload == "8k" and not args.no_cpu_baseline:
                hb["via_grok_plugin"] = via_grok_plugin(ctx, params, tile, prec, (out.get("cpu_baseline") or {}).get("file_md5"))
            if args.workload == "8k":
                hb["node_native"] = node_native(dev, tile, prec, levels, d_px, (out.get("cpu_baseline") or {}).get("file_md5"))
            out["host_boundary"] = hb
    else:
        out = None

    emit = _emit

    if use_dist:
        # ---- the gather regions, last and under a watchdog (see above): frame f's tile-parts to rank f mod N, depth frames in flight
        rep = multi_gpu["replica_%s" % args.workload]
        if out is not None:
            out["exchange"] = {"counts": rep["counts"], "gather": None}
            out["config"]["headline_exchange"] = "counts"

        dog = start_gather_watchdog(args.gather_timeout, out, emit, args.gather_timeout_rc)
        # the gather with 1, 2 and 4 frames' transfers in flight (the depth asked for last: its figure is the line's): the first run
        # on real links shows the TREND, not one number
        by_depth = {}
        g8 = None
        for dpt in sorted(set([1, 2, 4]) - {args.gather_depth}) + [args.gather_depth]:
            gw = max(args.warmup, 2 * (dpt + 3) + 2)
            dtg, parts_g, root_g = run_frames(params, ntiles, rot, nblocks, "gather", args.steps, gw, gather_depth=dpt)
            cs_len = assembled_bytes(params, ntiles, nblocks, W * world if ntiles == 1 else 0, H, parts_g, root_g)
            g8 = {"ms_per_step": round(dtg / args.steps * 1e3, 4), "Mpixels_s": round(pixels_per_step * world * args.steps / dtg / 1e6, 1),
                  "assembled_codestream_bytes": cs_len, "gather_depth": dpt}
            by_depth[str(dpt)] = {"ms_per_step": g8["ms_per_step"], "Mpixels_s": g8["Mpixels_s"]}
            if out is not None:
                out["exchange"]["gather_by_depth"] = dict(by_depth)     # (what the watchdog prints if a later depth never completes)
        g8["by_depth"] = by_depth
        gw = max(args.warmup, 2 * (args.gather_depth + 3) + 2)
        g4 = None
        if cfg4 is not None:
            p4, nt4, d4, nb4, st4 = cfg4
            ctx.encode_tiles(p4, nt4, d4.data_ptr(), True, fetch=False)
            ctx.synchronize()
            d_t, parts4, root4 = run_frames(p4, nt4, d4, nb4, "gather", st4, gw)
            g4 = {"ms_per_step": round(d_t / st4 * 1e3, 4), "Mpixels_s": round(16384.0 * 16384.0 * st4 / d_t / 1e6, 1),
                  "assembled_block_bytes": assembled_bytes(p4, nt4, nb4, 0, 0, parts4, root4)}
        if out is not None:
            rep["gather"] = g8
            out["exchange"]["gather"] = g8
            if g4 is not None:
                multi_gpu["cfg4_strong"]["gather"] = g4
            # the headline: the faster of the two complete exchanges unless the caller named one; both stay in `exchange`
            counts_v = (rep.get("counts") or {}).get("Mpixels_s") or 0.0
            gbest = g8
            if args.exchange == "best":            # ... at the depth that did best (every depth's figure is a complete gather)
                dbest = max(by_depth, key=lambda k: by_depth[k]["Mpixels_s"])
                gbest = dict(by_depth[dbest], gather_depth=int(dbest))
            take_gather = args.exchange == "gather" or (args.exchange == "best" and gbest["Mpixels_s"] > counts_v)
            out["config"]["headline_exchange_rule"] = args.exchange
            if take_gather:
                out["value"] = gbest["Mpixels_s"]
                out["ms_per_step"] = gbest["ms_per_step"]
                out["config"]["headline_exchange"] = "gather (depth %d)" % gbest["gather_depth"]
                out["config"]["assembled_codestream_bytes"] = cs_len
        # (the line already carries the gather figures and the headline: a rank that fails or stalls below costs it `exchange.parts` only --
        #  a watchdog of its own)
        dog.cancel()
        dog = start_gather_watchdog(args.gather_timeout, out, emit, args.gather_timeout_rc, key="parts")
        # ---- ... and the exchange of FINISHED tile-parts: Tier-2 runs on every rank's device inside the timed region (grk_amd_assemble_device_async
        # on the exchange's stream), the writer receives tile-parts it only has to put behind the main header.  More work per frame than
        # the gather of loose blocks (whose Tier-2, on the writer's host, is NOT in that form's timed region): reported, never the headline
        parts_fig = None
        try:
            dtp, parts_p, root_p = run_frames(params, ntiles, rot, nblocks, "parts", args.steps, gw, gather_depth=args.gather_depth)
            plen, pmd5 = assembled_file(params, ntiles, W * world if ntiles == 1 else 0, H, parts_p, root_p)
            parts_fig = {"ms_per_step": round(dtp / args.steps * 1e3, 4), "Mpixels_s": round(pixels_per_step * world * args.steps / dtp / 1e6, 1),
                         "codestream_bytes": plen, "equals_gather_form_length": bool(plen == cs_len), "gather_depth": args.gather_depth,
                         "what": "Tier-2 on each rank's device inside the timed region; finished tile-parts (one table row each) to the frame's writer"}
            if world == 1 and pmd5 is not None and gather_md5[0] is not None:      # (one process: the same frame's two files side by side)
                parts_fig["file_equals_gather_form_file"] = bool(pmd5 == gather_md5[0])
                ref = (out or {}).get("cpu_baseline") or {}
                if ref.get("file_md5"):
                    parts_fig["file_equals_cpu_encode"] = bool(ref["file_md5"] == pmd5)
        except Exception as e:  # noqa: BLE001
            parts_fig = {"error": str(e)}
        dog.cancel()
        if out is not None:
            out["exchange"]["parts"] = parts_fig
        dist.barrier()
        dist.destroy_process_group()
    emit(out)


_views = {}


def _as_tensor(ptr, n, dev, typestr="|u1"):
    """Zero-copy view of a raw device pointer (context-owned memory): n elements of `typestr` (cached: the encoder's
    buffer sets alternate between a handful of addresses)."""
    key = (int(ptr), int(n), typestr)
    t = _views.get(key)
    if t is None:
        class _Holder:
            pass
        h = _Holder()
        h.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
        t = _views[key] = torch.as_tensor(h, device=dev)
    return t


if __name__ == "__main__":
    main()
