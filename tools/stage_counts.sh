#!/bin/bash
# Diagnostic: dynamic instruction counts per stage.  Builds one variant of the kernel per stage (-DLC3_STOP_AFTER=k: frames end
# after stage k), on the GPU box runs each under rocprofv3 --pmc and prints the differences between consecutive variants.
#   bash tools/stage_counts.sh build     (here, no GPU needed)
#   bash tools/stage_counts.sh run       (on the GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/audio_codec_amd/csrc
VAR=$SRC/variants
ORDER="0 2 3 4 5 1 6 7 8 9 10 11 12 13 14 15 16 17"
if [ "$1" = build ]; then
  mkdir -p $VAR
  build_one() { k=$1; hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -fPIC -Wno-unused-value -Wno-unused-label -DLC3_STOP_AFTER=$k -c $SRC/lc3_kernels.hip -o $VAR/k$k.o && hipcc --offload-arch=gfx950 -shared -fPIC -o $VAR/stop$k.so $VAR/k$k.o $SRC/lc3_host.o -lm && rm $VAR/k$k.o; }
  n=0
  for k in $ORDER; do build_one $k & n=$((n+1)); if [ $((n % 6)) = 0 ]; then wait; fi; done
  wait; ls $VAR
else
  OUT=$ROOT/gpurun_out/stage_counts; mkdir -p $OUT
  cd /tmp; export TMPDIR=/tmp
  for k in $ORDER; do
    LC3PLUS_HIP_LIB=$VAR/stop$k.so timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT/s$k -o p --output-format csv -- python3 $ROOT/tools/stage_counts_run.py > $OUT/s$k.log 2>&1
  done
  python3 - <<PY
import csv, glob
names = {0: "load", 1: "mdct", 2: "resample", 3: "olpa", 4: "ltpf", 5: "attack", 6: "energy_bw", 7: "sns_scf", 8: "sns_vq", 9: "sns_apply", 10: "tns",
         11: "gain_est", 12: "quant1", 13: "gain_adj+quant2", 14: "noise", 15: "residual", 16: "bitstream", 17: "store+slide"}
prev = {}
print("%-18s %9s %9s %9s %9s   (instructions per frame)" % ("stage", "VALU", "SALU", "LDS", "VMEM_RD"))
for k in [int(x) for x in "$ORDER".split()]:
    acc = {}
    for f in glob.glob("$OUT/s%d/*counter_collection.csv" % k):
        for r in csv.DictReader(open(f)):
            if "lc3_encode" in r["Kernel_Name"]: acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    fr = 1024 * 16.0
    cur = {c: acc.get(c, 0) / fr for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD")}
    print("%-18s %9.0f %9.0f %9.0f %9.0f" % (names[k], *[cur[c] - prev.get(c, 0) for c in cur]))
    prev = cur
print("%-18s %9.0f %9.0f %9.0f %9.0f" % ("total", *[prev[c] for c in prev]))
PY
fi
