#!/bin/bash
# Instruction-cache counters of the encoder bench launch.  Usage (GPU box): bash tools/icache_pmc.sh <tag>
TAG=${1:-ic}
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VALU"; do
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
for k in sorted(acc): print("%s %s %.0f (mean of %d)" % (k[0], k[1], sum(acc[k]) / len(acc[k]), len(acc[k])))
PY
