#!/bin/bash
# usage: tools/ab_lib.sh "variant1 variant2 ..." [bench args] -- same-box A/B of builds under build/abl/<variant>/ ("head" = the in-tree library)
vals=$1; shift
for v in $vals; do
  lib=$GRAFT_REPO_ROOT/build/abl/$v/libgrok_amd.so; [ "$v" = head ] && lib=$GRAFT_REPO_ROOT/grok_amd/lib/libgrok_amd.so
  AB_LIB=$lib timeout 200 python tools/bench_ab.py --no-cpu-baseline --steps 20 "$@" 2>&1 | tail -1 > /tmp/ab.json
  python3 -c "import json; d=json.load(open('/tmp/ab.json')); e=d.get('decode') or {}; print('$v', d['ms_per_step'], d['value'], 'decode', e.get('ms_per_step'), e.get('lossless_round_trip'))"
done
