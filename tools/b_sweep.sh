#!/bin/bash
# Diagnostic (GPU box): Mframes/s by streams per GPU (64 frames per call)
cd "$(dirname "$0")/.."
for B in 512 1024 2048 3072 4096 6144 8192 16384; do
  timeout -k 10 240 python bench.py --workload c1 --streams $B --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams $B', d['value'], d['ms_per_step'])" || exit 1
done
