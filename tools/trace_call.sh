#!/bin/bash
# Diagnostic: kernel timeline of the last encode() call of a short bench run (GPU box).  Usage: bash tools/trace_call.sh <tag> [bench args]
TAG=${1:-tr}; shift
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT/tr -o t --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras "$@" > $OUT/run.log 2>&1
cd $ROOT
python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob("$OUT/tr/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("lc3_"): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].replace("lc3_enc_","").replace("_kernel","")[:10]))
rows.sort()
packs=[i for i,r in enumerate(rows) if r[2]=="pack"]
a=packs[-2]+1; b=packs[-1]+1
t0=rows[a][0]
with open("$OUT/timeline.txt","w") as o:
    for s,e,n in rows[a:b]: o.write("%8.3f %8.3f %7.3f %s\n"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,n))
import collections
tot=collections.defaultdict(float)
for s,e,n in rows[a:b]: tot[n]+=(e-s)/1e6
print("call span %.3f ms; summed durations:"%((rows[b-1][1]-t0)/1e6), {k:round(v,3) for k,v in tot.items()})
PY
head -60 $OUT/timeline.txt
