#!/bin/bash
# Diagnostic (GPU box): Mframes/s by frames per call for run lengths of 4, 6 and 8 frames (LC3PLUS_ENC_RUN_FRAMES)
cd "$(dirname "$0")/.."
for F in 5 6 8 12 16 24 32 48 64; do for RF in 4 6 8; do
  LC3PLUS_ENC_RUN_FRAMES=$RF timeout -k 10 180 python bench.py --workload c1 --frames $F --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('frames $F run $RF', d['value'], d['ms_per_step'])" || exit 1
done; done
