"""Driver of tests/test_oracle_sanitizers.py: runs the CPU restatement built with ASan + UBSan (oracle/_asan, `make -C oracle asan`)
over a spread of operating points.  Started as a child with LD_PRELOAD=libasan:libubsan; any report aborts the process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lc3_harness as H
H.ORACLE_DIR = os.path.join(H.ORACLE_DIR, "_asan")
import numpy as np

n = 0
for fs, ms, hr, rates in ((48000, 10.0, 0, [16000, 64000, 96000, 320000]), (96000, 2.5, 1, [256000, 672000]), (16000, 5.0, 0, [32000, 96000]),
                          (44100, 10.0, 0, [64000]), (8000, 2.5, 0, [64000]), (48000, 5.0, 1, [160000, 600000]), (96000, 10.0, 1, [500000])):
    N = int((48000 if fs == 44100 else fs) * ms / 1000)
    pcm = H.synth_pcm(len(rates), 8, N, fs, seed=fs // 1000)
    pcm[-1, 3:] = 0                                               # a stream that falls silent
    for pm in (False, True):
        fr = H.oracle_encode_streams(pcm, fs, ms, hr, rates, portable_math=pm)
        n += sum(f.shape[0] for f in fr)
    frames, nb, bfi = H.make_dec_case(fs, ms, hr, 1, rates, 8, seed=3, loss=0.2, corrupt=0.2)
    H.oracle_decode_streams(frames, nb, bfi, fs, ms, hr, 1, portable_math=False)
    H.oracle_decode_streams(frames, nb, bfi, fs, ms, hr, 1, bps=24, portable_math=True)
o = H.Oracle(48000, 2, 10.0, 0, 128000)
st = H.synth_pcm(2, 6, 480, 48000, seed=9)
for t in range(6):
    if t == 3: o.set_bitrate(96000)
    o.encode(st[:, t])
print("sanitizers clean over %d encoded frames" % n)
