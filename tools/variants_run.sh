#!/bin/bash
# Diagnostic (GPU box): the bench line's value / kernel ms for every library under audio_codec_amd/_var/ and for the product build.
cd "$(dirname "$0")/.."
W=${W:-c1}
for lib in audio_codec_amd/liblc3plus_hip.so audio_codec_amd/_var/lib_*.so; do
  for i in 1 2; do
    LC3PLUS_HIP_LIB=$PWD/$lib timeout -k 10 180 python bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-parity 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['roofline']['kernel_ms_avg'])" || exit 1
  done
done
