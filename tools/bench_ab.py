"""bench.py against another build of the library (timing experiments): AB_LIB=/path/libgrok_amd.so python tools/bench_ab.py [bench args]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grok_amd.capi as _capi
if os.environ.get("AB_LIB"):
    _capi.lib_path = lambda: os.environ["AB_LIB"]
import bench
bench.main()
