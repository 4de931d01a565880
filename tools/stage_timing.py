"""Diagnostic: per-stage wave latency (shader cycles per frame) of lc3_encode_kernel from the -DLC3_STAGE_TIMING build.
The stamps are collected through the traced entry point, i.e. in the single-kernel path (bitstream stage included).
Usage (GPU box): python tools/stage_timing.py [B T bitrate]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import audio_codec_amd.api as api
api.lib_path = lambda: os.path.join(ROOT, "audio_codec_amd", "liblc3plus_hip_timing.so")
from lc3_harness import synth_pcm
NAMES = ["load", "mdct", "resample", "olpa", "ltpf", "attack", "energy_bw", "sns_scf", "sns_vq", "sns_apply", "tns", "gain_est",
         "quant1", "gain_adj+quant2", "noise", "residual", "bitstream", "store+slide"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
br = int(sys.argv[3]) if len(sys.argv) > 3 else 64000
pcm = synth_pcm(B, T, 480, 48000, seed=3)
b = api.Batch(B, 48000, 1, 10.0, 0, [br] * B, device=0)
out, tr = b.encode_traced(pcm)
tsz = tr.shape[1]
SUBS = {0: "resample: FIR", 1: "resample: biquad products", 2: "resample: biquad chain", 3: "resample: history slide",
        4: "sns_vq: stage 1", 5: "sns_vq: DCT-II", 6: "sns_vq: pulse searches", 7: "sns_vq: gains + IDCT", 8: "sns_vq: errors + select",
        9: "sns_vq: MPVQ index", 10: "gain_est: x_max / regulariser", 11: "gain_est: band energies (log10)", 12: "gain_est: probe (6 levels)",
        13: "gain_est: probe (last 2)", 14: "gain_est: tail", 15: "bitstream: clear + side info", 16: "bitstream: TNS symbols",
        17: "bitstream: tuple prep (parallel)", 18: "bitstream: range coder (serial)", 19: "bitstream: residual bits", 20: "bitstream: finalise",
        21: "tns: sums", 22: "tns: levinson (+weight)", 23: "tns: rc quant", 24: "tns: lattice", 25: "olpa: decimate + slide",
        26: "olpa: autocorr + argmax", 27: "olpa: normcorr + decision", 28: "quant: quantise + lastnz", 29: "quant: bit estimate passes",
        31: "noise: zero-line masks", 32: "noise: sums"}
accall = np.zeros(64)
for s in range(B):
    accall += np.frombuffer(tr[s * T].tobytes()[:64 * 8], dtype=np.int64)
accall = accall / (B * T)
acc = accall[:24]
tot = acc.sum()
print("kernel %.3f ms for %d streams x %d frames; mean cycles/frame/wave = %.0f" % (b.last_kernel_ms(), B, T, tot))
for n, v in zip(NAMES, acc):
    print("  %-16s %10.0f  %5.1f %%" % (n, v, 100 * v / tot))
print("sub-stage stamps (cycles/frame):")
for k in sorted(SUBS):
    print("  %-36s %10.0f" % (SUBS[k], accall[24 + k]))
