#!/bin/bash
# Diagnostic (GPU box): marginal cost of each kernel / stage in the co-resident mix, by running it twice (tools/variants.sh dup "-DLC3_DUP" ...)
cd "$(dirname "$0")/.."
run() { env "$@" timeout -k 10 180 python bench.py --workload ${W:-c1} --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['roofline']['kernel_ms_avg'])"; }
run X=0 || exit 1
for l in dupgain dupquant dupnoise; do [ -f audio_codec_amd/_var/lib_$l.so ] && run LC3PLUS_HIP_LIB=$PWD/audio_codec_amd/_var/lib_$l.so; done
for k in "" f v r h p k s; do run LC3PLUS_HIP_LIB=$PWD/audio_codec_amd/_var/lib_dup.so LC3PLUS_ENC_DUP=x$k || exit 1; done
