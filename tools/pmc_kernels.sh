#!/bin/bash
# Diagnostic (GPU box): dynamic instruction counts per kernel and channel-frame of the default bench call (separate --pmc passes, no traces).
# usage: bash tools/pmc_kernels.sh <tag> [bench args]      -> gpurun_out/<tag>/summary.txt
TAG=${1:-pmc}; shift
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
PASSES=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU")
if [ -n "$PMC_ONLY" ]; then PASSES=(); fi
IFS=';' read -ra EXTRA <<< "$PMC_EXTRA"
for C in "${PASSES[@]}" "${EXTRA[@]}"; do
  [ -z "$C" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C -d $OUT/p$i -o p --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/p$i.log 2>&1 || { echo "failed pass $i"; tail -5 $OUT/p$i.log; exit 1; }
  echo "progress: pass $i"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
out = "$OUT"
per = collections.defaultdict(float); calls = collections.Counter()
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("lc3_"): continue
        per[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        if k.startswith("lc3_enc_pack_kernel") or k.startswith("lc3_dec_synth_kernel"): calls[r["Counter_Name"]] += 1
frames = float("${FRAMES_PER_CALL:-262144}")
names = sorted({c for _, c in per}); kernels = sorted({k for k, _ in per})
with open(out + "/summary.txt", "w") as o:
    o.write("per channel-frame (%d frames per call)\n%-28s" % (frames, "kernel") + "".join("%14s" % n.replace("SQ_", "")[:13] for n in names) + "\n")
    tot = collections.defaultdict(float)
    for k in kernels:
        o.write("%-28s" % k.replace("lc3_enc_", "").replace("_kernel", ""))
        for n in names:
            v = per.get((k, n), 0) / max(1, calls[n]) / frames; tot[n] += v
            o.write("%14.1f" % v)
        o.write("\n")
    o.write("%-28s" % "total" + "".join("%14.1f" % tot[n] for n in names) + "\n")
print(open(out + "/summary.txt").read())
PY
