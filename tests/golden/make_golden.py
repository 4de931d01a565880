"""Generates tests/golden/*.npz from the UNMODIFIED ETSI reference (oracle/_ref/liblc3_etsi_fl.so, built by
oracle/Makefile from /root/reference).  Run in the build container only:  python tests/golden/make_golden.py
Each fixture holds the input PCM and the reference's output frames (data only, no reference code)."""
import os, sys, wave, hashlib
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from lc3_harness import synth_pcm, Ref, ref_encode_streams

def save(name, **kw):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **kw)
    print("wrote", name, {k: getattr(v, "shape", v) for k, v in kw.items()})

def mono(name, fs, ms, hr, N, streams, rates, T):
    pcm = synth_pcm(max(streams) + 1, T, N, fs)[streams]
    outs = ref_encode_streams(pcm, fs, ms, hr, rates)
    stride = max(o.shape[1] for o in outs)
    frames = np.zeros((len(streams), T, stride), np.uint8)
    for b, o in enumerate(outs): frames[b, :, :o.shape[1]] = o
    save(name, fs=fs, frame_ms=ms, hrmode=hr, streams=np.array(streams), bitrates=np.array(rates), pcm=pcm,
         frames=frames, nbytes=np.array([o.shape[1] for o in outs]))

mono("c1_48k_10ms_64k", 48000, 10.0, 0, 480, [0, 1, 5, 62, 63], [64000] * 5, 24)
RATES = [16000, 24000, 32000, 48000, 64000, 80000, 96000, 128000, 160000, 192000, 256000, 320000]
mono("c5_48k_10ms_mixed", 48000, 10.0, 0, 480, list(range(12)), RATES, 16)
mono("c4_96k_2p5ms_hr_256k", 96000, 2.5, 1, 240, [0, 1, 62], [256000] * 3, 48)

# stereo 128 kbps: channel pairs (0,1) and (5,62)
T = 16
pcm = synth_pcm(63, T, 480, 48000)
pairs = [(0, 1), (5, 62)]
frames = np.zeros((len(pairs), T, 160), np.uint8)
for i, (a, b) in enumerate(pairs):
    r = Ref(48000, 2, 10.0, 0, 128000)
    for t in range(T): frames[i, t] = r.encode(np.stack([pcm[a, t], pcm[b, t]]))
save("c3_48k_10ms_stereo_128k", fs=48000, frame_ms=10.0, hrmode=0, pairs=np.array(pairs),
     pcm=np.stack([np.stack([pcm[a], pcm[b]]) for a, b in pairs]), frames=frames)

# ETSI test material excerpt (first 64 frames of testvec/input/thetest48.wav) + digest of the whole file's bitstream
wav = "/root/reference/LC3plus_ETSI_src_v17171_20200723/testvec/input/thetest48.wav"
w = wave.open(wav)
x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
Tall = x.size // 480
allpcm = x[:Tall * 480].reshape(1, Tall, 480)
full = ref_encode_streams(allpcm, 48000, 10.0, 0, [64000])[0]
save("c0_thetest48_64k_first64", fs=48000, frame_ms=10.0, hrmode=0, pcm=allpcm[:, :64].copy(), frames=full[None, :64].copy(),
     full_frames=Tall, full_pcm_md5=hashlib.md5(allpcm.tobytes()).hexdigest(), full_bitstream_md5=hashlib.md5(full.tobytes()).hexdigest())

# the other operating points (one small fixture per sample rate / frame length family outside 48 kHz / 10 ms)
def family(name, cfgs, T=12):
    out = {}
    for tag, fs, ms, hr, N, rates in cfgs:
        pcm = synth_pcm(3, T, N, 48000 if fs == 44100 else fs)[[0, 1, 2][:len(rates)]]
        outs = ref_encode_streams(pcm, fs, ms, hr, rates)
        stride = max(o.shape[1] for o in outs)
        fr = np.zeros((len(rates), T, stride), np.uint8)
        for b, o in enumerate(outs): fr[b, :, :o.shape[1]] = o
        out[tag + "_cfg"] = np.array([fs, int(ms * 10), hr, N]); out[tag + "_rates"] = np.array(rates); out[tag + "_pcm"] = pcm
        out[tag + "_frames"] = fr; out[tag + "_nbytes"] = np.array([o.shape[1] for o in outs])
    save(name, tags=np.array([c[0] for c in cfgs]), **out)

family("c6_other_operating_points", [
    ("nb8k_10", 8000, 10.0, 0, 80, [16000, 32000]), ("nb8k_2p5", 8000, 2.5, 0, 20, [64000, 96000]),
    ("wb16k_10", 16000, 10.0, 0, 160, [32000, 64000]), ("wb16k_5", 16000, 5.0, 0, 80, [32000, 64000]),
    ("sswb24k_5", 24000, 5.0, 0, 120, [32000, 96000]), ("sswb24k_2p5", 24000, 2.5, 0, 60, [64000, 96000]),
    ("swb32k_10", 32000, 10.0, 0, 320, [64000, 128000]), ("swb32k_2p5", 32000, 2.5, 0, 80, [64000, 128000]),
    ("fb44k_10", 44100, 10.0, 0, 480, [64000, 128000]), ("fb48k_2p5", 48000, 2.5, 0, 120, [64000, 128000]),
    ("hr48k_5", 48000, 5.0, 1, 240, [160000, 320000]), ("hr96k_5", 96000, 5.0, 1, 480, [256000, 400000]),
    ("hr96k_10", 96000, 10.0, 1, 960, [149600, 400000]),
])
