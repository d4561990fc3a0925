"""CPU: the C-ABI library loads, exports every symbol include/grok_amd.h declares, and its
host-only entry points (geometry, Tier-2 writer) behave; no GPU compute is attempted."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import grok_amd as G
import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(grk_amd_[a-z0-9_]+|plugin_[a-z0-9_]+|minpf_post_load_plugin)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = G.lib()
    syms = declared_symbols("grok_amd.h")
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "libgrok_amd.so does not export %s" % s


def test_missing_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        G.Context(0)


@pytest.mark.parametrize("w,h,c,levels,nblocks", [(512, 512, 1, 3, 64), (4096, 4096, 3, 5, 12288),
                                                   (8192, 8192, 3, 5, 49152), (1024, 1024, 3, 5, 777)])
def test_block_counts_match_survey(w, h, c, levels, nblocks):
    p = G.TileParams.make(w, h, c, 8, levels)
    assert G.lib().grk_amd_tile_num_blocks(C.byref(p)) == nblocks


@pytest.mark.parametrize("w,h,levels", [(512, 512, 3), (1024, 1024, 5), (200, 120, 2), (130, 67, 5), (37, 3, 1), (1, 1, 0)])
def test_layout_matches_oracle_enumeration(w, h, levels):
    p = G.TileParams.make(w, h, 1, 8, levels)
    blocks, qcd = G.tile_layout(p)
    expn = O.rev_exponents(8, levels)
    assert [q >> 3 for q in qcd] == expn.tolist()
    ob = O.enumerate_blocks(w, h, levels, expn)
    assert len(ob) == len(blocks)
    for a, b in zip(blocks, ob):
        assert (a.px, a.py, a.x1 - a.x0, a.y1 - a.y0, a.res, a.band, a.kmax) == (b.x, b.y, b.w, b.h, b.res, b.band, b.kmax)


def test_irreversible_qcd_matches_oracle():
    p = G.TileParams.make(8192, 8192, 3, 16, 5, irreversible=True)
    blocks, qcd = G.tile_layout(p)
    oq, od = O.irrev_stepsizes(16, 5)
    assert qcd == oq.tolist()
    steps = {}
    for b in blocks:
        steps[(b.res, b.band)] = b.stepsize
    idx = 0
    for r in range(6):
        for band in ([0] if r == 0 else [1, 2, 3]):
            assert steps[(r, band)] == pytest.approx(float(od[idx]), rel=0, abs=0)
            idx += 1


def test_unsupported_parameters_are_rejected():
    L = G.lib()
    bad = G.TileParams.make(64, 64, 1, 8, 11)
    assert L.grk_amd_tile_num_blocks(C.byref(bad)) == -2
    bad = G.TileParams.make(0, 64, 1, 8, 1)
    assert L.grk_amd_tile_num_blocks(C.byref(bad)) == -3
    bad = G.TileParams.make(64, 64, 2, 8, 1, mct=True)
    assert L.grk_amd_tile_num_blocks(C.byref(bad)) == -3


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under grok_amd/ or include/ imports, links, dlopens or executes it
    (comments that cite it are fine), and the product libraries carry no dependency on it."""
    import re, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = re.compile(r"^\s*(import|from)\s+oracle\b|liboracle|oracle/_ref|dlopen\([^)]*oracle|CDLL\([^)]*oracle", re.M)
    for sub in ("grok_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(root, sub)):
            for f in fs:
                if not f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                    continue
                text = open(os.path.join(dp, f), errors="replace").read()
                # drop comments before looking for uses
                text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
                text = re.sub(r"//[^\n]*|#[^\n]*", "", text)
                assert not code.search(text), "%s refers to the oracle" % os.path.join(dp, f)
    for lib in ("libgrok_amd.so", "libgrokj2k_plugin.so"):
        path = os.path.join(root, "grok_amd", "lib", lib)
        if os.path.exists(path):
            needed = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
            assert "oracle" not in needed


def _build_node_example(tmp_path):
    import subprocess
    exe = str(tmp_path / "node_example")
    libdir = os.path.dirname(G.lib_path())
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "node_example.c"), "-o", exe, "-L" + libdir, "-lgrok_amd",
                           "-Wl,-rpath," + libdir])
    return exe


def test_c_host_of_the_node_api_compiles_and_fails_loudly_without_a_gpu(tmp_path):
    """INTEGRATION.md 5a's C host (tests/c/node_example.c) is real code: plain C99 against include/grok_amd.h, linked with
    libgrok_amd.so.  Without a GPU grk_amd_node_create reports GRK_AMD_ERR_NO_DEVICE (no CPU fallback)."""
    import subprocess
    import torch
    exe = _build_node_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if torch.cuda.is_available():
        assert r.returncode == 0 and "identical" in r.stdout, r.stdout + r.stderr
    else:
        assert r.returncode == 3 and "no device" in r.stdout, r.stdout + r.stderr


def _build_rccl_host_example(tmp_path):
    import subprocess
    exe = str(tmp_path / "rccl_host_example")
    libdir = os.path.dirname(G.lib_path())
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "rccl_host_example.cpp"), "-o", exe, "-L" + libdir, "-lgrok_amd",
                           "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_native_rccl_host_compiles_and_fails_loudly_without_a_gpu(tmp_path):
    """INTEGRATION.md 5c's native host -- one process per GPU, the C ABI + RCCL (ncclAllGather of the byte counts, exact-size
    ncclSend / ncclRecv of the tile-parts to a rotating writer) -- is real code: it compiles and links here; with a GPU it runs
    its one-rank form (tests/test_gpu_node.py), without one it says so."""
    import subprocess
    import torch
    exe = _build_rccl_host_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        assert r.returncode == 0 and "identical" in r.stdout, r.stdout + r.stderr
    else:
        assert r.returncode == 3 and "no device" in r.stdout, r.stdout + r.stderr


def test_the_loaded_library_was_built_from_this_tree():
    """The shared libraries are prebuilt, git-ignored artefacts that travel to the GPU box as they are: the
    library says which sources it was made of, and that has to be the tree the tests run from."""
    import ctypes as C
    import importlib.util
    spec = importlib.util.spec_from_file_location("graft_entry", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "__graft_entry__.py"))
    ge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ge)
    L = G.lib()
    L.grk_amd_source_stamp.restype = C.c_char_p
    assert L.grk_amd_source_stamp().decode() == ge.source_stamp(), "grok_amd/lib/libgrok_amd.so is stale: run python __graft_entry__.py"
