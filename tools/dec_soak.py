"""Differential soak of the DECODER (GPU box): every operating point x seeds x channel counts.  Bitstreams come from the GPU encoder,
get damaged (frames marked lost, bytes flipped in unmarked frames), are decoded on the GPU in two launches and by the CPU oracle
decoder (same math); PCM and the concealment status must be identical.
Usage: python tools/dec_soak.py [seeds] [streams] [frames]  -> one line per differing configuration and a total; exit code 1 on any difference."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import audio_codec_amd
from lc3_harness import synth_pcm, oracle_decode_streams
import importlib.util
_spec = importlib.util.spec_from_file_location("soak", os.path.join(ROOT, "tools", "soak.py"))
_soak = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_soak)
configurations = _soak.configurations


def run(NS, B, T, verbose=True):
    """returns (channel-frames compared, channel-frames that differ)"""
    tot = bad = 0
    for ci, (fs, ms, hr, rates) in enumerate(configurations()):
        N = int(round((48000 if fs == 44100 else fs) * ms / 1000))
        for seed in range(NS):
            ch = 1 + (ci + seed) % 2
            rng = np.random.default_rng(9000 + 31 * seed + ci)
            br = [int(rates[(i + seed) % len(rates)]) * ch for i in range(B)]
            pcm = synth_pcm(B * ch, T, N, fs, seed=5000 + 13 * seed).reshape(B, ch, T, N).transpose(0, 2, 1, 3)
            enc = audio_codec_amd.Batch(B, fs, ch, ms, hr, br, device=0)
            frames = enc.encode(np.ascontiguousarray(pcm))
            nb = [enc.num_bytes(i) for i in range(B)]
            bfi = (rng.random((B, T)) < 0.12).astype(np.uint8)
            for b in range(B):
                for t in np.nonzero(rng.random(T) < 0.12)[0]:
                    k = rng.integers(0, nb[b], size=3)
                    frames[b, t, k] ^= rng.integers(1, 256, size=3).astype(np.uint8)
            bps = (16, 24, 32)[(ci + seed) % 3]
            dec = audio_codec_amd.DecBatch(B, fs, ch, ms, hr, nb, device=0)
            cut = T // 3
            a, sa = dec.decode(frames[:, :cut], bfi[:, :cut], bps)
            c, sc = dec.decode(frames[:, cut:], bfi[:, cut:], bps)
            got, status = np.concatenate([a, c], axis=1), np.concatenate([sa, sc], axis=1)
            want, wstatus = oracle_decode_streams(frames, nb, bfi, fs, ms, hr, ch, bps)
            d = int((got != want).any(axis=3).sum()) + int((status != wstatus).sum())
            tot += B * T * ch; bad += d
            if d and verbose: print("%6d Hz %4.1f ms hr%d ch%d bps%d seed %d: %d of %d channel-frames differ" % (fs, ms, hr, ch, bps, seed, d, B * T * ch))
    return tot, bad


if __name__ == "__main__":
    NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 48
    t0 = time.time()
    tot, bad = run(NS, B, T)
    print("decoder soak: %d channel-frames over %d configurations x %d seeds, %d differ, %.0f s" % (tot, len(configurations()), NS, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
