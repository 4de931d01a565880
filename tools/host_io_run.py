"""Diagnostic: the host-pointer path of lc3plus_enc_batch_encode on the bench workload, for tracing (rocprofv3 --kernel-trace --memory-copy-trace).
Usage (GPU box): python tools/host_io_run.py [pinned|pageable] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import audio_codec_amd
kind = sys.argv[1] if len(sys.argv) > 1 else "pinned"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B, T, N = 4096, 64, 480
g = torch.Generator(); g.manual_seed(5)
pcm = (torch.randn(B, T, 1, N, generator=g) * 3000).round().clamp(-32768, 32767).to(torch.int16)
out = torch.zeros(B, T, 80, dtype=torch.uint8)
if kind == "pinned":
    pcm, out = pcm.pin_memory(), out.pin_memory()
b = audio_codec_amd.Batch(B, 48000, 1, 10.0, 0, [64000] * B, device=0)
hp, ho = pcm.numpy(), out.numpy()
b.encode_host(hp, ho)
for _ in range(reps):
    t0 = time.perf_counter(); b.encode_host(hp, ho); dt = time.perf_counter() - t0
    print("%s host buffers: %.3f ms per call = %.2f Mframes/s (kernel events %.3f ms)" % (kind, dt * 1e3, B * T / dt / 1e6, b.last_kernel_ms()))
