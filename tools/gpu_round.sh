#!/bin/bash
# One batched GPU-box session: parity tests, bench, rocprofv3 kernel stats and PMC passes.  Usage: bash tools/gpu_round.sh <tag>
set -e
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 $BENCH_ARGS > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/stats -o s --output-format csv -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $ROOT/$OUT/stats.log 2>&1
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $C -d $ROOT/$OUT/pmc_$N -o p --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/pmc_$N.log 2>&1
done
cd $ROOT
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("lc3_enc"): acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("$OUT/pmc_summary.txt", "w") as o:
    for k in sorted(acc): o.write("%s %s %.0f (mean of %d launches)\n" % (k[0], k[1], sum(acc[k]) / len(acc[k]), len(acc[k])))
print(open("$OUT/pmc_summary.txt").read())
PY
head -3 $OUT/stats/*kernel_stats.csv | cut -c1-200
