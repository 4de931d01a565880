#!/bin/bash
# Diagnostic (GPU box): the bench value / kernel ms for every library under audio_codec_amd/_var/ and for the product build, ROUNDS times round-robin
# (run-to-run spread on a box is +-3 %: compare medians).  usage: [W=c1] [ROUNDS=3] [STEPS=150] bash tools/variants_run.sh
cd "$(dirname "$0")/.."
W=${W:-c1}; ROUNDS=${ROUNDS:-3}; STEPS=${STEPS:-150}
for r in $(seq $ROUNDS); do
  for lib in audio_codec_amd/liblc3plus_hip.so audio_codec_amd/_var/lib_*.so; do
    LC3PLUS_HIP_LIB=$PWD/$lib timeout -k 10 180 python bench.py --workload $W --steps $STEPS --warmup 10 --no-cpu-baseline --no-extras --no-parity 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$W', '$lib'.split('/')[-1], d['value'], d['roofline']['kernel_ms_avg'])" || exit 1
  done
done | tee /tmp/variants_$W.txt
python3 - <<PY
import collections, statistics
v = collections.defaultdict(list)
for l in open("/tmp/variants_$W.txt"):
    p = l.split()
    if len(p) == 4: v[p[1]].append(float(p[2]))
base = statistics.median(v.get("liblc3plus_hip.so", [1]))
for k, x in sorted(v.items(), key=lambda kv: -statistics.median(kv[1])):
    print("median %-22s %7.2f  (%+.1f %%)  runs %s" % (k, statistics.median(x), 100 * (statistics.median(x) / base - 1), " ".join("%.1f" % a for a in x)))
PY
