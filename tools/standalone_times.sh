#!/bin/bash
# Diagnostic (GPU box): duration of every kernel of a call when it has the GPU to itself (AMD_SERIALIZE_KERNEL=3: HIP waits before and after every launch),
# beside the overlapped durations of the same build.  usage: bash tools/standalone_times.sh <tag> [bench args]
TAG=${1:-sa}; shift
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
AMD_SERIALIZE_KERNEL=3 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/ser -o s --output-format csv -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-parity "$@" > $OUT/ser.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/ovl -o s --output-format csv -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-parity "$@" > $OUT/ovl.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob
def load(d):
    r = {}
    for f in glob.glob("$OUT/%s/*kernel_stats.csv" % d):
        for x in csv.DictReader(open(f)):
            if x["Name"].startswith("lc3_"): r[x["Name"].replace("lc3_enc_", "").replace("_kernel", "")] = float(x["AverageNs"]) / 1e6
    return r
a, b = load("ser"), load("ovl")
tot = [0, 0]
for k in sorted(a, key=lambda k: -a[k]):
    print("%-14s alone %.3f ms   overlapped %.3f ms   x%.2f" % (k, a[k], b.get(k, 0), b.get(k, 0) / a[k] if a[k] else 0)); tot[0] += a[k]; tot[1] += b.get(k, 0)
print("sum            alone %.3f ms   overlapped %.3f ms" % tuple(tot))
PY
