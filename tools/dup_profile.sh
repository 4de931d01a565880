#!/bin/bash
# Diagnostic (GPU box): dynamic instruction counts of every stage.  Each library under audio_codec_amd/_var/ runs one stage twice
# (tools/variants.sh <stage> "-DDUP_<STAGE>" ...; "base" = unchanged): the difference of the kernel's SQ_INSTS_* to base is the stage.
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/dupprof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in $ROOT/audio_codec_amd/_var/lib_*.so; do
  n=$(basename $lib .so); n=${n#lib_}
  LC3PLUS_HIP_LIB=$lib timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/$n -o p --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/$n.log 2>&1 || { echo "failed $n"; exit 1; }
  echo "progress: $n"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections, os
out = "$OUT"; res = {}
for d in sorted(glob.glob(out + "/*/")):
    n = os.path.basename(d[:-1]); per = collections.defaultdict(float); calls = 0
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("lc3_enc"):
                per[(k, r["Counter_Name"])] += float(r["Counter_Value"])
                if k == "lc3_enc_pack_kernel" and r["Counter_Name"] == "SQ_INSTS_VALU": calls += 1
    res[n] = {k: v / max(1, calls) / (4096 * 64) for k, v in per.items()}
base = res["base"]
with open(out + "/summary.txt", "w") as o:
    o.write("per channel-frame: VALU SALU LDS\n")
    for k in sorted({k for k, _ in base}): o.write("base %-28s %7.0f %7.0f %7.0f\n" % (k, base[(k, "SQ_INSTS_VALU")], base[(k, "SQ_INSTS_SALU")], base[(k, "SQ_INSTS_LDS")]))
    for n in sorted(res):
        if n == "base": continue
        d = {c: sum(res[n][(k, c)] - base[(k, c)] for k in {k for k, _ in base}) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")}
        o.write("stage %-27s %7.0f %7.0f %7.0f\n" % (n, d["SQ_INSTS_VALU"], d["SQ_INSTS_SALU"], d["SQ_INSTS_LDS"]))
print(open(out + "/summary.txt").read())
PY
