import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the prebuilt real-reference harness under oracle/_ref")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # a GPU test on a box without a GPU is an error of the invocation, not a skip: fail loudly
    # only when the user selected -m gpu; in mixed runs they are skipped.
    if _gpu_available():
        # on a GPU box the pins on the real reference are mandatory (product and oracle share the generated
        # CxtVLC table, what vouches for it is ojph_* run live): a `needs_ref` skip would hide that oracle/_ref did not travel,
        # so the skip is taken off and the test fails on loading the harness
        if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_harness.so")):
            for it in items:
                if "gpu" in it.keywords:
                    it.own_markers = [m for m in it.own_markers
                                      if not (m.name == "skipif" and "oracle/_ref" in str(m.kwargs.get("reason", "")))]
        return
    sel = config.getoption("-m") or ""
    if "gpu" in sel and "not gpu" not in sel:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _library_matches_the_tree():
    """On the GPU box the product libraries are prebuilt artefacts shipped beside the sources: refuse to vouch for a tree with
    libraries that were made of other sources (the CPU suite checks the same in tests/test_capi_host.py)."""
    if not _gpu_available():
        yield
        return
    import ctypes as C
    import importlib.util
    import grok_amd as G
    spec = importlib.util.spec_from_file_location("graft_entry", os.path.join(ROOT, "__graft_entry__.py"))
    ge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ge)
    L = G.lib()
    L.grk_amd_source_stamp.restype = C.c_char_p
    got, want = L.grk_amd_source_stamp().decode(), ge.source_stamp()
    if got != want:
        pytest.exit("grok_amd/lib/libgrok_amd.so was built from other sources (%s) than this tree's (%s): run python __graft_entry__.py" % (got[:12], want[:12]), returncode=3)
    yield
