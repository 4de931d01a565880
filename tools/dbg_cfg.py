"""Diagnostic (GPU box): one configuration against the oracle, per stream / frame.  usage: python tools/dbg_cfg.py fs ms hr N rate[,rate...] [T]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import audio_codec_amd
from lc3_harness import synth_pcm, oracle_encode_streams
fs, ms, hr, N = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rates = [int(x) for x in sys.argv[5].split(",")]
T = int(sys.argv[6]) if len(sys.argv) > 6 else 24
B = len(rates)
pcm = synth_pcm(B, T, N, fs, seed=21)
b = audio_codec_amd.Batch(B, fs, 1, ms, hr, rates, device=0)
got = np.concatenate([b.encode(pcm[:, :10]), b.encode(pcm[:, 10:])], axis=1)
want = oracle_encode_streams(pcm, fs, ms, hr, rates, portable_math=True)
for i in range(B):
    nb = b.num_bytes(i)
    bad = [t for t in range(T) if not (got[i, t, :nb] == want[i][t][:nb]).all()]
    print("stream %d rate %d nbytes %d: %d bad frames %s" % (i, rates[i], nb, len(bad), bad[:12]))
    if bad:
        t = bad[0]
        d = np.nonzero(got[i, t, :nb] != want[i][t][:nb])[0]
        print("   first bad frame %d: differing bytes %s" % (t, d[:16]))
