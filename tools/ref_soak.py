"""Parity against the compiled ETSI REFERENCE at scale (GPU box; E/conformance/lc3_conformance.py:126-129,746-786 "encode" mode).

For each BASELINE configuration the same synthetic PCM is encoded by the GPU engine and by the unmodified reference
(oracle/_ref/cpu_bench_ref = R's own lc3_enc_* API driven from C on all host cores, glibc libm) and the frames are compared byte
for byte.  Run-time float libm calls are the one place the device cannot be bit-identical to glibc: every
stream with a differing frame is decoded twice by the reference decoder and the ETSI `mld` tool gives the maximum loudness
difference, the conformance procedure's measure (threshold 4), next to the procedure's default metric (the ETSI `rms` tool at 14-bit
resolution) and its energy difference (lc3_conformance.py:126-130, 542-556, 586-619).  Prints one line per configuration and a JSON summary.

usage: python tools/ref_soak.py [scale]      scale 1.0 = 1.3 M channel-frames in all, 0.02 for the reduced pytest case
TEST INFRASTRUCTURE (reads oracle/_ref)."""
import json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from lc3_harness import synth_pcm, mld_between, rms_between, energy_diff_between, have_ref, MLD_TOOL, RMS_TOOL

REF_BENCH = os.path.join(ROOT, "oracle", "_ref", "cpu_bench_ref")
RATES12 = [16000, 24000, 32000, 48000, 64000, 80000, 96000, 128000, 160000, 192000, 256000, 320000]
# name: fs, ms, hr, channels, N, rates, streams, frames  (BASELINE.json configs[0..4] shapes; c0 = the reference's own mono case on synthetic PCM)
# "bw" configurations (E/conformance/lc3_conformance.py:803-817,850-862 band_limiting / bandwidth_switching): a bandwidth set per stream at the start
# (a third of the streams), switched at the call boundary (a ninth), band-limited input (an eighth); 32 kHz / 5 ms is the second shape of VERDICT r3 item 3
CONFIGS = [
    ("c0", 48000, 10.0, 0, 1, 480, [64000], 64, 1009),
    ("b1", 48000, 10.0, 0, 1, 480, [64000, 92000], 2048, 64),
    ("b2", 32000, 5.0, 0, 1, 160, [64000, 128000], 1024, 64),
    ("c1", 48000, 10.0, 0, 1, 480, [64000], 4096, 64),
    ("c3", 48000, 10.0, 0, 2, 480, [128000], 2048, 64),
    ("c4", 96000, 2.5, 1, 1, 240, [256000], 2048, 128),
    ("c5", 48000, 10.0, 0, 1, 480, RATES12, 4096, 64),
]


def ref_encode(pcm, fs, ms, hr, ch, rate, nbytes_total, plan=None):
    """pcm [S, T, ch, N] int16 -> [S, T, nbytes_total] uint8 by the compiled reference, all host cores."""
    S, T = pcm.shape[:2]
    with tempfile.TemporaryDirectory(prefix="refsoak_") as td:
        pin, pout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        np.ascontiguousarray(pcm).tofile(pin)
        thr = max(1, min(os.cpu_count() or 1, 64))
        env = dict(os.environ)
        if plan is not None and plan.any():
            np.ascontiguousarray(plan, np.int32).tofile(os.path.join(td, "bw.bin")); env["LC3_BENCH_BW_PLAN"] = os.path.join(td, "bw.bin")
        r = subprocess.run([REF_BENCH, "enc", str(fs), str(ms), str(hr), str(ch), str(rate), str(S), str(T), str(thr), pin, pout],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=3000, env=env)
        if r.returncode: raise RuntimeError(r.stderr[-300:])
        return np.fromfile(pout, np.uint8).reshape(S, T, nbytes_total)


def run(scale=1.0, verbose=True):
    import audio_codec_amd
    res = []
    for name, fs, ms, hr, ch, N, rates, S, T in CONFIGS:
        S = max(len(rates), int(S * scale)) if name != "c0" else max(2, int(S * scale))
        S -= S % len(rates)
        pcm = synth_pcm(S * ch, T, N, fs, seed=9000 + len(name) + ord(name[1])).reshape(S, ch, T, N).transpose(0, 2, 1, 3).copy()
        br = [rates[i % len(rates)] for i in range(S)]
        b = audio_codec_amd.Batch(S, fs, ch, ms, hr, br, device=0)
        h = T // 2
        plan = np.zeros((S, T), np.int32)
        if name[0] == "b":
            import soak
            lp = [i for i in range(S) if i % 8 == 5]
            pcm[:, :, 0] = soak.band_limit(pcm[:, :, 0], fs, lp, [(4000, 8000, 12000, 16000)[(i // 8) % 4] for i in lp])
            plan = soak.bandwidth_plan(S, T, fs, hr, [h], 5)
            for i in np.nonzero(plan[:, 0])[0]: assert b.set_bandwidth(int(i), int(plan[i, 0])) in (0, 18)
        g1 = b.encode(pcm[:, :h] if ch > 1 else pcm[:, :h, 0])
        for i in np.nonzero(plan[:, h])[0]: assert b.set_bandwidth(int(i), int(plan[i, h])) in (0, 18)
        got = np.concatenate([g1, b.encode(pcm[:, h:] if ch > 1 else pcm[:, h:, 0])], axis=1)   # two calls: state persists
        nbs = [b.num_bytes(i) for i in range(S)]
        b.close()
        diff = tot = 0; worst = None; nstreams_diff = 0; worst_rms = worst_eng = None; rms_ok = True
        for rate in sorted(set(br)):
            idx = [i for i in range(S) if br[i] == rate]
            nb = nbs[idx[0]]
            want = ref_encode(pcm[idx], fs, ms, hr, ch, rate, nb, plan[idx])
            g = got[idx][:, :, :nb]
            neq = (g != want).any(axis=2)
            diff += int(neq.sum()); tot += neq.size * ch
            for k in np.nonzero(neq.any(axis=1))[0]:
                nstreams_diff += 1
                if os.path.exists(MLD_TOOL):
                    v = mld_between(g[k], want[k], fs, ms, hr, ch)
                    worst = v if worst is None else max(worst, v)
                if os.path.exists(RMS_TOOL):            # the script's default metric (rms at 14 bits) and its energy difference, for the record beside the MLD
                    q = rms_between(g[k], want[k], fs, ms, hr, ch)
                    worst_rms = q["rms_db"] if worst_rms is None else max(worst_rms, q["rms_db"]); rms_ok = rms_ok and q["ok"]
                    e = energy_diff_between(g[k], want[k], fs, ms, hr, ch)
                    worst_eng = e if worst_eng is None else max(worst_eng, e)
        r = {"config": name, "channel_frames": tot, "frames_differ": diff, "streams_differ": nstreams_diff, "worst_mld": worst,
             "worst_rms_db": worst_rms, "rms_14bit_ok": rms_ok if worst_rms is not None else None, "worst_energy_diff_log10": worst_eng}
        res.append(r)
        if verbose: print("ref_soak %s: %d channel-frames, %d stream-frames differ in %d streams, worst MLD %s" % (name, tot, diff, nstreams_diff, worst), flush=True)
    return res


if __name__ == "__main__":
    if not (have_ref() and os.path.exists(REF_BENCH)):
        print("ref_soak: oracle/_ref not built here"); sys.exit(2)
    t0 = time.time()
    res = run(float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
    tot = sum(r["channel_frames"] for r in res); diff = sum(r["frames_differ"] for r in res)
    mlds = [r["worst_mld"] for r in res if r["worst_mld"] is not None]
    print(json.dumps({"channel_frames": tot, "frames_differ": diff, "worst_mld": max(mlds) if mlds else None, "mld_threshold": 4.0,
                      "seconds": round(time.time() - t0, 1), "per_config": res}))
    sys.exit(0 if (not mlds or max(mlds) <= 4.0) else 1)
