#!/bin/bash
# VALU instruction mix of the encoder bench launch (fp64 instructions occupy the SIMD twice as long).  Usage (GPU box): bash tools/valu_mix.sh <tag>
TAG=${1:-mix}
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $C -d $OUT/pmc_$N -o p --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $OUT/pmc_$N.log 2>&1
done
cd $ROOT
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "lc3_" in k: acc[(k.split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(acc): print("%s %s %.0f" % (k[0], k[1], sum(acc[k]) / len(acc[k])))
PY
