"""Diagnostic: per-stage wave latency of the DECODER (shader cycles per frame) from the -DLC3_STAGE_TIMING build.
Usage (GPU box): python tools/dec_stage_timing.py [B T bitrate]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import audio_codec_amd.api as api
api.lib_path = lambda: os.path.join(ROOT, "audio_codec_amd", "liblc3plus_hip_timing.so")
from lc3_harness import synth_pcm
NAMES = ["hand-over + overlap-add", "-", "-", "-", "-", "-", "-", "-", "-", "ltpf", "output"]   # stamps of lc3_dec_synth_kernel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
br = int(sys.argv[3]) if len(sys.argv) > 3 else 64000
pcm = synth_pcm(B, T, 480, 48000, seed=3)
enc = api.Batch(B, 48000, 1, 10.0, 0, [br] * B, device=0)
frames = enc.encode(pcm)
d = api.DecBatch(B, 48000, 1, 10.0, 0, [enc.num_bytes(0)] * B, device=0)
out, status, tr = d.decode_traced(frames)
acc = np.zeros(16)
for s in range(B):
    acc += np.frombuffer(tr[s * T].tobytes()[:16 * 8], dtype=np.int64)
acc = acc[:11] / (B * T)
print("kernel %.3f ms for %d streams x %d frames; mean cycles/frame/wave = %.0f" % (d.last_kernel_ms(), B, T, acc.sum()))
for n, v in zip(NAMES, acc):
    print("  %-32s %10.0f  %5.1f %%" % (n, v, 100 * v / acc.sum()))
