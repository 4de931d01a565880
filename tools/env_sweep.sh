#!/bin/bash
# Diagnostic (GPU box): bench value / kernel ms under a list of environment settings.  usage: tools/env_sweep.sh "A=1" "A=2 B=3" ...
cd "$(dirname "$0")/.."
W=${W:-c1}
for e in "$@"; do
  for i in 1 2; do
    env $e timeout -k 10 180 python bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['roofline']['kernel_ms_avg'])" || exit 1
  done
done
