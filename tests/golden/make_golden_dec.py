"""Generates tests/golden/d1_decoder_operating_points.npz from the UNMODIFIED ETSI reference (oracle/_ref/liblc3_etsi_fl.so,
built by oracle/Makefile from /root/reference).  Run in the build container only:  python tests/golden/make_golden_dec.py
Per operating point the fixture holds reference-encoded frames of seeded synthetic PCM, damaged on purpose (frames marked lost and
frames with flipped bytes), and what the reference DECODER made of them: 16-bit PCM and the per-frame LC3_DECODE_ERROR status.
Data only, no reference code."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from lc3_harness import Ref, RefDecoder, make_dec_case

CFGS = [
    ("fb48k_10", 48000, 10.0, 0, 1, [32000, 64000, 128000]), ("fb48k_10_stereo", 48000, 10.0, 0, 2, [64000, 128000]),
    ("fb48k_5", 48000, 5.0, 0, 1, [64000, 128000]), ("fb48k_2p5", 48000, 2.5, 0, 1, [64000, 128000]),
    ("fb44k_10", 44100, 10.0, 0, 1, [64000, 128000]), ("swb32k_10", 32000, 10.0, 0, 1, [32000, 96000]),
    ("swb32k_2p5", 32000, 2.5, 0, 1, [64000, 128000]), ("sswb24k_5", 24000, 5.0, 0, 1, [32000, 96000]),
    ("wb16k_10", 16000, 10.0, 0, 1, [16000, 64000]), ("wb16k_5", 16000, 5.0, 0, 1, [32000, 64000]),
    ("nb8k_10", 8000, 10.0, 0, 1, [16000, 32000]), ("nb8k_2p5", 8000, 2.5, 0, 1, [64000, 96000]),
    ("hr48k_10", 48000, 10.0, 1, 1, [128000, 400000]), ("hr48k_5", 48000, 5.0, 1, 1, [160000, 320000]),
    ("hr96k_2p5", 96000, 2.5, 1, 1, [256000, 400000]),
    ("hr96k_5", 96000, 5.0, 1, 1, [256000, 400000]), ("hr96k_10", 96000, 10.0, 1, 1, [149600, 400000]),
]
T = 20
out = {}
for i, (tag, fs, ms, hr, ch, rates) in enumerate(CFGS):
    frames, nbytes, bfi = make_dec_case(fs, ms, hr, ch, rates, T, seed=100 + i, loss=0.15, corrupt=0.15, enc_cls=Ref)
    pcm = status = None
    for b in range(len(rates)):
        d = RefDecoder(fs, ch, ms, hr)
        if pcm is None:
            pcm = np.zeros((len(rates), T, ch, d.N), np.int16); status = np.zeros((len(rates), T), np.uint8)
        for t in range(T):
            rc, x = d.decode(frames[b, t, :nbytes[b]], int(bfi[b, t]), 16)
            assert rc in (0, 2)
            pcm[b, t] = x; status[b, t] = rc == 2
    out[tag + "_cfg"] = np.array([fs, int(ms * 10), hr, ch]); out[tag + "_frames"] = frames; out[tag + "_nbytes"] = np.array(nbytes)
    out[tag + "_bfi"] = bfi; out[tag + "_pcm"] = pcm; out[tag + "_status"] = status
    print(tag, frames.shape, "lost", int(bfi.sum()), "concealed", int(status.sum()))
np.savez_compressed(os.path.join(HERE, "d1_decoder_operating_points.npz"), tags=np.array([c[0] for c in CFGS]), **out)
