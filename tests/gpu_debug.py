"""GPU bring-up tool (not a pytest): encodes a few streams on the GPU with stage tracing and reports, per stage, the
first disagreement with the CPU oracle (portable-math build).  Usage: python tests/gpu_debug.py [fs ms hr bitrate B T]"""
import ctypes as C
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lc3_harness import Oracle, Trace, synth_pcm
import audio_codec_amd

def fields_cmp(name, a, b, tol=0):
    a = np.asarray(a); b = np.asarray(b)
    if a.dtype.kind == 'f':
        bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    else:
        bad = a != b
    if bad.any():
        i = int(np.argmax(bad))
        return "%s MISMATCH n=%d first@%d gpu=%r cpu=%r" % (name, int(bad.sum()), i, a.flat[i], b.flat[i])
    return None

def main():
    args = sys.argv[1:]
    fs = int(args[0]) if len(args) > 0 else 48000
    ms = float(args[1]) if len(args) > 1 else 10.0
    hr = int(args[2]) if len(args) > 2 else 0
    br = int(args[3]) if len(args) > 3 else 64000
    B = int(args[4]) if len(args) > 4 else 4
    T = int(args[5]) if len(args) > 5 else 4
    streams = [int(x) for x in args[6].split(",")] if len(args) > 6 else list(range(B))
    B = len(streams)
    N = int(fs * ms / 1000)
    pcm = synth_pcm(max(streams) + 1, T, N, fs, seed=int(os.environ.get("DBG_SEED", "11")))[streams]
    rates = [br] * B
    bt = audio_codec_amd.Batch(B, fs, 1, ms, hr, rates, device=0)
    got, traces = bt.encode_traced(pcm)
    print("kernel ms", bt.last_kernel_ms())
    nb = bt.num_bytes(0)
    tot = same = 0
    shown = 0
    for b in range(B):
        o = Oracle(fs, 1, ms, hr, br, portable_math=True)
        tr = o.enable_trace()
        for t in range(T):
            want = o.encode(pcm[b, t][None])
            g = Trace.from_buffer_copy(traces[b * T + t].tobytes()[:C.sizeof(Trace)])
            c = tr[0]
            ok = (got[b, t, :nb] == want).all()
            tot += 1; same += int(ok)
            msgs = []
            for f, _ in Trace._fields_:
                ga, ca = getattr(g, f), getattr(c, f)
                if hasattr(ga, "__len__"):
                    n = N if len(ga) == 960 else len(ga)
                    m = fields_cmp(f, np.ctypeslib.as_array(ga)[:n], np.ctypeslib.as_array(ca)[:n])
                else:
                    m = fields_cmp(f, [ga], [ca])
                if m:
                    msgs.append(m)
                    if os.environ.get('DBG_FULL') and hasattr(ga, '__len__') and len(ga) <= 64:
                        msgs.append('   gpu ' + np.array2string(np.ctypeslib.as_array(ga), precision=5, max_line_width=200))
                        msgs.append('   cpu ' + np.array2string(np.ctypeslib.as_array(ca), precision=5, max_line_width=200))
            if (not ok or msgs) and shown < 12:
                shown += 1
                print("stream %d frame %d bytes_ok=%s" % (streams[b], t, ok))
                for m in msgs[:8]: print("    ", m)
                if not ok and not msgs:
                    d = np.where(got[b, t, :nb] != want)[0]
                    print("     bytes differ at", d[:10], "gpu", got[b, t, d[:6]], "cpu", want[d[:6]])
    print("byte-identical frames: %d / %d" % (same, tot))

if __name__ == "__main__":
    main()
