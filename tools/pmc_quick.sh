#!/bin/bash
# usage: pmc_quick.sh <lib> <workload> <tag>
ROOT=$PWD; OUT=$ROOT/gpurun_out/$3; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export LC3PLUS_HIP_LIB=$1
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout -k 10 200 rocprofv3 --pmc $C -d $OUT/pmc_$N -o p --output-format csv -- python3 $ROOT/bench.py --workload $2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-parity > $OUT/pmc_$N.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("lc3_dec_parse"): acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(acc): print("$3", k[0], k[1], round(acc[k] / n[k] / 262144))
PY
