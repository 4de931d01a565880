#!/bin/bash
# Diagnostic (GPU box): bench value under a list of environment settings.  usage: bash tools/sweep_env.sh out.txt "A=1 B=2" "C=3" ...
OUT=$1; shift
: > $OUT
for e in "$@"; do
  v=$(env $e timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')
  echo "$e -> $v" | tee -a $OUT
done
