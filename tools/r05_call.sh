#!/bin/bash
# scratch: one GPU-box call of round 5 (edited per call)
cd $GRAFT_REPO_ROOT
echo "=== full gpu suite"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 3
echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
echo "=== bench"; timeout 900 python bench.py > gpurun_out/bench_r05d.json 2> gpurun_out/bench_r05d.err; tail -c 200 gpurun_out/bench_r05d.err; python tools/show_bench.py gpurun_out/bench_r05d.json 2>&1 | head -12
echo "=== profile round"; PROF_WLS="8k cfg2 cfg3 cfg5" bash tools/prof_round.sh r05 stats pmc sq decstats 2>&1 | tail -n 4
