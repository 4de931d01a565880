"""Driver of tools/stage_counts.sh: one launch of 1024 streams x 16 frames (C1) through the library named by LC3PLUS_HIP_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import audio_codec_amd.api as api
from lc3_harness import synth_pcm
B, T = 1024, 16
pcm = synth_pcm(B, T, 480, 48000, seed=3)
b = api.Batch(B, 48000, 1, 10.0, 0, [64000] * B, device=0)
out = b.encode(pcm)
print("kernel ms", b.last_kernel_ms())
