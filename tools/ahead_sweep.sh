#!/bin/bash
# Diagnostic (GPU box): consecutive calls with / without the input-ready promise, by frames per call (LC3PLUS_ENC_AHEAD_MAX lifts the library's limit)
cd "$(dirname "$0")/.."
for F in 12 16 24 32 40 48 64; do for m in "" "--serial-calls"; do
  LC3PLUS_ENC_AHEAD_MAX=1000 timeout -k 10 180 python bench.py --workload c1 --frames $F --steps 40 --warmup 5 --no-cpu-baseline --no-extras $m 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('frames $F $m', d['value'], d['ms_per_step'])" || exit 1
done; done
