#!/bin/bash
# Decoder counters: rocprofv3 PMC passes over `bench.py --workload d1` (both decoder kernels).  Usage (GPU box): bash tools/dec_pmc.sh <tag>
TAG=${1:-dec}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $ROOT/bench.py --workload d1 --steps 5 --warmup 1 > $OUT/stats.log 2>&1
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $C -d $OUT/pmc_$N -o p --output-format csv -- python3 $ROOT/bench.py --workload d1 --steps 2 --warmup 1 > $OUT/pmc_$N.log 2>&1
done
cd $ROOT
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "lc3_dec" in k: acc[(k.split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("$OUT/pmc_summary.txt", "w") as o:
    for k in sorted(acc): o.write("%s %s %.0f (mean of %d launches)\n" % (k[0], k[1], sum(acc[k]) / len(acc[k]), len(acc[k])))
print(open("$OUT/pmc_summary.txt").read())
PY
cut -c1-140 $OUT/stats/*kernel_stats.csv | head -4
