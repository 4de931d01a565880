"""Diagnostic (GPU box): c1 as G independent sub-batches of 4096/G streams, each on its own launch stream, calls issued round-robin."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, audio_codec_amd, bench
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
B, T, n, fs = 4096 // G, 64, 480, 48000
bs, pcms, outs, sts = [], [], [], []
for g in range(G):
    pcm = bench.synth_pcm_device(torch, B, T, 1, n, fs, dev, seed=1234 + g, first_stream=g * B)
    b = audio_codec_amd.Batch(B, fs, 1, 10.0, 0, [64000] * B, device=0); b.set_input_ready(True)
    bs.append(b); pcms.append(pcm); outs.append(torch.zeros(B, T, b.stride, dtype=torch.uint8, device=dev)); sts.append(torch.cuda.Stream(dev))
def step():
    for g in range(G):
        bs[g].encode_device(pcms[g].data_ptr(), 16, T, outs[g].data_ptr(), bs[g].stride, hip_stream=sts[g].cuda_stream, sync=False)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); K = 20
for _ in range(K): step()
torch.cuda.synchronize()
w = time.perf_counter() - t0
print("G=%d: %.2f Mframes/s, %.3f ms per step" % (G, 4096 * T * K / w / 1e6, w / K * 1e3))
