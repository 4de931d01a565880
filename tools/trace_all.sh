#!/bin/bash
# Diagnostic: kernel timeline of the last N ms of a short bench run (GPU box), all overlapping calls.  Usage: bash tools/trace_all.sh <tag> <ms> [bench args]
TAG=${1:-tr}; MS=${2:-8}; shift; shift
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT/tr -o t --output-format csv -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras "$@" > $OUT/run.log 2>&1
cd $ROOT
python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob("$OUT/tr/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("lc3_"): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].replace("lc3_enc_","").replace("_kernel","")[:10], r.get("Queue_Id","")))
rows.sort()
tend=max(r[1] for r in rows); t0=tend-int(float("$MS")*1e6)
with open("$OUT/timeline.txt","w") as o:
    for s,e,n,q in rows:
        if e>=t0: o.write("%8.3f %8.3f %7.3f q%s %s\n"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,q,n))
PY
cat $OUT/timeline.txt
