#!/bin/bash
# Diagnostic (GPU box): marginal cost of each kernel in the co-resident mix, by launching it twice (tools/variants.sh dup "-DLC3_DUP"):
# r resampler, h HP50, p pitch, f front, e scale factors, v quantiser, a shape, s rate, k writer.  ROUNDS times round-robin; medians.
cd "$(dirname "$0")/.."
W=${W:-c1}; ROUNDS=${ROUNDS:-3}; STEPS=${STEPS:-100}
run() { env "$@" timeout -k 10 180 python bench.py --workload $W --steps $STEPS --warmup 10 --no-cpu-baseline --no-extras --no-parity 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['roofline']['kernel_ms_avg'])"; }
for r in $(seq $ROUNDS); do
  for k in "" r h p f e v a s k; do run LC3PLUS_HIP_LIB=$PWD/audio_codec_amd/_var/lib_dup.so LC3PLUS_ENC_DUP=x$k || exit 1; done
done | tee /tmp/dup_$W.txt
python3 - <<PY
import collections, statistics
v = collections.defaultdict(list)
for l in open("/tmp/dup_$W.txt"):
    p = l.split()
    if len(p) == 3: v[p[0].split("=")[1]].append(float(p[2]))
base = statistics.median(v["x"])
print("$W: ms per call, one kernel launched twice; base %.3f" % base)
for k, x in sorted(v.items(), key=lambda kv: -statistics.median(kv[1])):
    if k != "x": print("  twice %s  %.3f ms  marginal %+.3f ms  (runs %s)" % (k[1:], statistics.median(x), statistics.median(x) - base, " ".join("%.3f" % a for a in x)))
PY
