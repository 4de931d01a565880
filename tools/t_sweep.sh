#!/bin/bash
# Diagnostic (GPU box): Mframes/s by frames per call (4096 streams), the regime change between the in-kernel writer and the pipelined path
cd "$(dirname "$0")/.."
for F in 1 2 4 6 8 9 10 12 16 20; do
  timeout -k 10 180 python bench.py --workload c1 --frames $F --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('frames $F', d['value'], d['ms_per_step'])" || exit 1
done
