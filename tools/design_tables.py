"""Prints the numbers DESIGN.md section 5 quotes, from profiles/<tag>_* (the per-kernel table in markdown, the roofline figures, the workload table)."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = lambda n: os.path.join(ROOT, "profiles", "%s_%s" % (tag, n))
cnt = json.load(open(P("counters.json")))
F = cnt["streams"] * cnt["frames"]
st = {r["Name"]: float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(P("c1_kernel_stats.csv")))}
b = json.load(open(P("bench_c1.json")))
print("c1 %.1f Mframes/s, %.3f ms per call; traffic %.2f GB; VALU %.3f G = %.0f per frame; lane util %.2f" % (b["value"], b["ms_per_step"], cnt["traffic_bytes"] / 1e9, cnt["valu_insts"] / 1e9, cnt["valu_insts"] / F, cnt["valu_lane_util"]))
for k in ("roofline", "roofline_valu", "issue", "serial_calls", "t1", "host_io", "cpu_baseline", "parity_sample", "other_workloads"):
    print(k, json.dumps(b.get(k))[:600])
tot = dict(v=0, s=0, l=0, w=0, r=0, wr=0, thr=0)
fmt = lambda x: format(int(round(x)), ",").replace(",", " ")
rows = []
for k, v in cnt["per_kernel"].items():
    va, sa, ld, wc = v.get("SQ_INSTS_VALU", 0) / F, v.get("SQ_INSTS_SALU", 0) / F, v.get("SQ_INSTS_LDS", 0) / F, v.get("SQ_WAVE_CYCLES", 0) / F
    rd, wr = 2 * v.get("FETCH_SIZE", 0) / 1000, v.get("WRITE_SIZE", 0) / 1000
    lu = v.get("SQ_THREAD_CYCLES_VALU", 0) / (v["SQ_INSTS_VALU"] * 64) if v.get("SQ_INSTS_VALU") else 0
    rows.append((wc, "| `%s` | %s | %s | %s | %.2f | %s | %s / %s | %.2f |" % (k, fmt(va), fmt(sa), fmt(ld), lu, fmt(wc), fmt(rd), fmt(wr), st.get(k, 0))))
    tot["v"] += va; tot["s"] += sa; tot["l"] += ld; tot["w"] += wc; tot["r"] += rd; tot["wr"] += wr; tot["thr"] += v.get("SQ_THREAD_CYCLES_VALU", 0)
for _, r in sorted(rows, reverse=True): print(r)
print("| all | %s | %s | %s | %.2f | %s | %s / %s | %.2f per call |" % (fmt(tot["v"]), fmt(tot["s"]), fmt(tot["l"]), cnt["valu_lane_util"], fmt(tot["w"]), fmt(tot["r"]), fmt(tot["wr"]), b["ms_per_step"]))
for w in ("c3", "c4", "c5", "c96", "d1", "d5"):
    try: d = json.load(open(P("bench_%s.json" % w))); print(w, d["value"], d["ms_per_step"])
    except Exception as e: print(w, "missing", e)
