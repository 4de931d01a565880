"""GPU bring-up tool (not a pytest): decodes a few streams on the GPU with stage tracing and reports, per stage, the
first disagreement with the CPU oracle decoder (portable-math build).
Usage: python tests/gpu_dec_debug.py [fs ms hr bitrate B T channels]   (DBG_LOSS=p drops frames with probability p)"""
import ctypes as C
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lc3_harness import Oracle, OracleDecoder, DecTrace, synth_pcm
import audio_codec_amd


def fields_cmp(name, a, b):
    a = np.asarray(a); b = np.asarray(b)
    bad = ~((a == b) | (np.isnan(a) & np.isnan(b))) if a.dtype.kind == 'f' else a != b
    if bad.any():
        i = int(np.argmax(bad))
        return "%s MISMATCH n=%d first@%d gpu=%r cpu=%r" % (name, int(bad.sum()), i, a.flat[i], b.flat[i])
    return None


def main():
    args = sys.argv[1:]
    fs = int(args[0]) if len(args) > 0 else 48000
    ms = float(args[1]) if len(args) > 1 else 10.0
    hr = int(args[2]) if len(args) > 2 else 0
    brs = [int(x) for x in args[3].split(",")] if len(args) > 3 else [64000]
    B = int(args[4]) if len(args) > 4 else 4
    brs = (brs * B)[:max(B, len(brs))]; B = len(brs)
    T = int(args[5]) if len(args) > 5 else 6
    CH = int(args[6]) if len(args) > 6 else 1
    loss = float(os.environ.get("DBG_LOSS", "0.2"))
    corrupt = float(os.environ.get("DBG_CORRUPT", "0.1"))
    rng = np.random.default_rng(int(os.environ.get("DBG_SEED", "11")))
    N = int((48000 if fs == 44100 else fs) * ms / 1000)
    pcm = synth_pcm(B * CH, T, N, fs, seed=5).reshape(B, CH, T, N).transpose(0, 2, 1, 3)
    enc = [Oracle(fs, CH, ms, hr, brs[b]) for b in range(B)]
    frames = np.zeros((B, T, 1250), dtype=np.uint8)
    nbs = []
    for b in range(B):
        for t in range(T):
            f = enc[b].encode(pcm[b, t])
            frames[b, t, :f.size] = f
        nbs.append(f.size)
    frames = np.ascontiguousarray(frames[:, :, :max(nbs)])
    bfi = (rng.random((B, T)) < loss).astype(np.uint8)
    for b in range(B):
        for t in range(T):
            if rng.random() < corrupt:
                k = rng.integers(0, nbs[b], size=4)
                frames[b, t, k] ^= rng.integers(1, 256, size=4).astype(np.uint8)
    db = audio_codec_amd.DecBatch(B, fs, CH, ms, hr, nbs, device=0)
    got, status, traces = db.decode_traced(frames, bfi)
    print("kernel ms", db.last_kernel_ms(), "bytes/frame", nbs, "lost", int(bfi.sum()), "concealed", int(status.sum()))
    tot = same = shown = 0
    for b in range(B):
        o = OracleDecoder(fs, CH, ms, hr, portable_math=True)
        tr = o.enable_trace()
        for t in range(T):
            rc, want = o.decode(frames[b, t, :nbs[b]], int(bfi[b, t]))
            ok = (got[b, t] == want).all() and int(status[b, t]) == int(rc == 2)
            tot += 1; same += int(ok)
            msgs = []
            for c in range(CH if rc == 0 else 0):
                g = DecTrace.from_buffer_copy(traces[(b * CH + c) * T + t].tobytes()[:C.sizeof(DecTrace)])
                cc = tr[c]
                for f, _ in DecTrace._fields_:
                    if f in ("bfi", "xq", "q_gain", "scf_q"):      # the first kernel hands over the spectrum after TNS only
                        continue
                    ga, ca = getattr(g, f), getattr(cc, f)
                    if hasattr(ga, "__len__"):
                        n = N if len(ga) == 960 else len(ga)
                        m = fields_cmp("ch%d %s" % (c, f), np.ctypeslib.as_array(ga)[:n], np.ctypeslib.as_array(ca)[:n])
                    else:
                        m = fields_cmp("ch%d %s" % (c, f), [ga], [ca])
                    if m:
                        msgs.append(m)
            if (not ok or msgs) and shown < 12:
                shown += 1
                print("stream %d frame %d pcm_ok=%s bfi=%d status gpu=%d cpu_rc=%d" % (b, t, ok, bfi[b, t], status[b, t], rc))
                for m in msgs[:10]: print("    ", m)
                if not msgs:
                    d = np.argwhere(got[b, t] != want)
                    print("     pcm differs at", d[:6].tolist(), "n", len(d))
    print("sample-identical frames: %d / %d" % (same, tot))


if __name__ == "__main__":
    main()
