#!/bin/bash
# One batched GPU-box session for the record: bench lines of every workload, rocprofv3 kernel stats and separate PMC passes of the
# same bench command (counters and traces never in one run), distilled into profiles/<tag>_*.  Usage: bash tools/gpu_round.sh <tag> [workloads]
#   profiles/<tag>_<w>_kernel_stats.csv   rocprofv3 --kernel-trace --stats: per-kernel launches / average duration with calls overlapped
#   profiles/<tag>_<w>_pmc.txt            per kernel and call: every counter collected for the workload
#   profiles/<tag>_counters.json          what bench.py quotes (traffic, instruction counts, lane utilisation, measured clock), keyed by the source hash
#   profiles/<tag>_bench_<w>.json         the bench line of the workload, taken again behind the counter passes so that it carries them
set -e
TAG=${1:-r04}
WL=${2:-"c1 c3 c4 c5 c96 d1 d5"}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT $ROOT/profiles
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/stats_$w -o s --output-format csv -- python3 $ROOT/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/stats_$w.log 2>&1
  echo "progress: stats $w"
done
# PMC passes, counters in their own runs: the metric's workload and the decoder in full, the others traffic + lane utilisation + wave-cycles
FULL=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM")
LIGHT=("SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE")
for w in $WL; do
  if [ $w = c1 ] || [ $w = d1 ]; then PASSES=("${FULL[@]}"); else PASSES=("${LIGHT[@]}"); fi
  for C in "${PASSES[@]}"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    timeout -k 10 200 rocprofv3 --pmc $C -d $OUT/pmc_${w}_$N -o p --output-format csv -- python3 $ROOT/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_${w}_$N.log 2>&1
    echo "progress: pmc $w $N"
  done
done
cd $ROOT
python3 - <<PY
import csv, glob, collections, json, os, sys
sys.path.insert(0, "$ROOT")
import bench
out, tag = "$OUT", "$TAG"
# ---- kernel stats (avg duration per kernel) ----
for w in "$WL".split():
    fs = glob.glob("%s/stats_%s/**/*kernel_stats.csv" % (out, w), recursive=True)
    if fs:
        rows = [r for r in csv.DictReader(open(fs[0])) if r["Name"].startswith("lc3_")]
        with open("profiles/%s_%s_kernel_stats.csv" % (tag, w), "w") as o:
            o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
            for r in rows: o.write("%s,%s,%s,%s,%s,%s,%s\n" % (r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]))
# ---- PMC ----  a call is several launches of some kernels (runs of frames): per-call value = sum over the launches of a pass / calls in
# that pass, a call being one launch of the kernel that runs once per call (lc3_enc_pack_kernel / lc3_dec_synth_kernel)
summ, clock = {}, {}
for w in "$WL".split():
    acc = collections.defaultdict(list); act = dur = 0.0
    for f in glob.glob("%s/pmc_%s_*/**/*counter_collection.csv" % (out, w), recursive=True):
        per = collections.defaultdict(float); calls = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("lc3_dec" if w[0] == "d" else "lc3_enc"):
                per[(k, r["Counter_Name"])] += float(r["Counter_Value"])
                if k.startswith("lc3_enc_pack_kernel") or k.startswith("lc3_dec_synth_kernel"): calls[r["Counter_Name"]] += 1      # (also lc3_enc_pack_kernel_w5, the _big synthesis kernel)
                # effective clock (MI355X_MICROARCH.md "DVFS give-back"): GRBM_GUI_ACTIVE is summed over the 8 XCDs; dispatches of 0.3 ms and more only
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                    if d >= 3e5: act += float(r["Counter_Value"]); dur += d
        for (k, cn), v in per.items(): acc[(k, cn)].append(v / max(1, calls[cn]))
    if not acc: continue
    with open("profiles/%s_%s_pmc.txt" % (tag, w), "w") as o:
        for k in sorted(acc): o.write("%s %s %.0f (per call, all launches of the call summed)\n" % (k[0], k[1], sum(acc[k]) / len(acc[k])))
    summ[w] = {k: sum(v) / len(v) for k, v in acc.items()}
    if dur: clock[w] = round(act / 8.0 / dur, 3)
def tot(w, c): return sum(v for (k, cn), v in summ[w].items() if cn == c)
ents = []
for w in summ:
    B, T = bench.WORKLOADS[w][7], bench.WORKLOADS[w][6]
    fetch, write = tot(w, "FETCH_SIZE"), tot(w, "WRITE_SIZE")        # KB; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section)
    valu, thr = tot(w, "SQ_INSTS_VALU"), tot(w, "SQ_THREAD_CYCLES_VALU")
    e = {"workload": w, "streams": B, "frames": T, "fetch_size_kb_raw": fetch, "write_size_kb_raw": write,
         "traffic_bytes": int(2 * fetch * 1000 + write * 1000), "valu_insts": int(valu),
         "valu_lane_util": round(thr / (valu * 64), 4) if valu else None,
         "per_kernel": {k: {c: v for (kk, c), v in summ[w].items() if kk == k} for k in sorted({kk for kk, _ in summ[w]})}}
    if w in clock:
        e["clock_ghz"] = clock[w]
        e["clock_source"] = "GRBM_GUI_ACTIVE / 8 / dispatch duration over the lc3_* dispatches of 0.3 ms and more of the counter pass (kernels run one at a time under counter collection)"
    ents.append(e)
if ents:
    ents.sort(key=lambda e: e["workload"] != "c1")
    top = ents[0]; top["more"] = ents[1:]
    top["source_head"] = bench.source_hash()       # the sources these counters were collected on: bench.py quotes them only while it matches
    top["source"] = "tools/gpu_round.sh %s: separate rocprofv3 --pmc passes of python3 bench.py --workload W --steps 2 --warmup 1; all lc3_* kernels of one call summed, mean over launches; FETCH_SIZE doubled per MI355X_MICROARCH.md; lane utilisation = SQ_THREAD_CYCLES_VALU / (SQ_INSTS_VALU x 64): the share of lanes the EXEC mask enables per vector instruction (calibration: 64.0 for lc3_enc_hp50_kernel, whose lanes are all live)" % tag
    json.dump(top, open("profiles/%s_counters.json" % tag, "w"), indent=1)
if os.path.exists("profiles/%s_c1_pmc.txt" % tag): print(open("profiles/%s_c1_pmc.txt" % tag).read()[:2500])
PY
# the bench lines behind the counter passes: they quote the counters of these sources (traffic, issue-rate roofline, measured clock)
for w in $WL; do
  timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 3 > $OUT/bench_$w.json 2> $OUT/bench_$w.err && cp $OUT/bench_$w.json profiles/${TAG}_bench_$w.json
  cut -c1-200 $OUT/bench_$w.json
done
mkdir -p $OUT/profiles && cp profiles/${TAG}_* $OUT/profiles/
ls profiles | grep $TAG
