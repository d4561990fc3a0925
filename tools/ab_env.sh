#!/bin/bash
# usage: tools/ab_env.sh VAR "v1 v2 ..." [bench args]  -- same-box A/B of an environment knob, ms_per_step of bench.py
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v timeout 200 python bench.py --no-cpu-baseline --steps 20 "$@" 2>&1 | tail -1 > /tmp/ab.json
  python3 -c "import json; d=json.load(open('/tmp/ab.json')); print('$var=$v', d['ms_per_step'], d['value'])"
done
