"""Differential soak (GPU box): every operating point x several seeds, GPU frames against the oracle's batch entry (same math).
Usage: [SOAK_READY=frames_per_call] python tools/soak.py [seeds] [streams] [frames]   -> prints one line per configuration and a total; exit code 1 on any difference."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import audio_codec_amd
from lc3_harness import synth_pcm, ORACLE_DIR

L = C.CDLL(os.path.join(ORACLE_DIR, "liblc3_oracle_pm.so"))
L.lc3o_encode_batch16.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
L.lc3o_encode_batch16_bw.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
BW = int(os.environ.get("SOAK_BW", "1"))                  # the bandwidth column (E/conformance/lc3_conformance.py:803-817,850-862): band-limited input, set and switched bandwidths
READY = int(os.environ.get("SOAK_READY", "0"))            # frames per call of the overlapped variant (0: host-pointer calls)
HIP = C.CDLL("libamdhip64.so") if READY else None


def band_limit(pcm, fs, streams, fcs):
    """low-pass the given streams (windowed-sinc FIR): the conformance procedure's band-limited items, which the bandwidth DETECTOR has to find"""
    out = pcm.copy()
    for s, fc in zip(streams, fcs):
        n = np.arange(-64, 65)
        h = np.sinc(2.0 * fc / fs * n) * (2.0 * fc / fs) * np.hamming(n.size)
        x = np.convolve(pcm[s].reshape(-1).astype(np.float64), h, mode="same")
        out[s] = np.clip(np.rint(x), -32768, 32767).astype(np.int16).reshape(pcm[s].shape)
    return out


def bandwidth_plan(B, T, fs, hr, cuts, seed):
    """int32 [B, T]: bandwidth to set in front of frame t (0 = keep).  A third of the streams set one at the start, a third of those switch at every
    call boundary (the GPU batch switches between calls, the oracle at the same frames)."""
    plan = np.zeros((B, T), np.int32)
    allowed = [bw for bw in (4000, 8000, 12000, 16000, 20000) if 2 * bw <= min(fs, 40000)]
    if hr or fs < 16000 or not allowed: return plan
    rng = np.random.default_rng(77 + seed)
    for i in range(B):
        if i % 3 == 1: plan[i, 0] = allowed[rng.integers(len(allowed))]
        if i % 9 == 4:
            for t in cuts: plan[i, t] = allowed[rng.integers(len(allowed))]
    return plan


def encode_ready(b, pcm, tc):
    B, T = pcm.shape[:2]
    stride = b.stride
    b.set_input_ready(True)
    ptrs, calls = [], []
    for t0 in range(0, T, tc):
        n = min(tc, T - t0)
        a = np.ascontiguousarray(pcm[:, t0:t0 + n]); pi, po = C.c_void_p(), C.c_void_p()
        assert HIP.hipMalloc(C.byref(pi), C.c_size_t(a.nbytes)) == 0 and HIP.hipMalloc(C.byref(po), C.c_size_t(B * n * stride)) == 0
        assert HIP.hipMemcpy(pi, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), C.c_int(1)) == 0 and HIP.hipMemset(po, 0, C.c_size_t(B * n * stride)) == 0
        ptrs += [pi, po]; calls.append((pi.value, po.value, n))
    assert HIP.hipDeviceSynchronize() == 0
    for pi, po, n in calls: b.encode_device(pi, 16, n, po, stride, hip_stream=None, sync=False)
    outs = []
    for pi, po, n in calls:
        o = np.zeros((B, n, stride), np.uint8)
        assert HIP.hipMemcpy(C.c_void_p(o.ctypes.data), C.c_void_p(po), C.c_size_t(o.nbytes), C.c_int(2)) == 0
        outs.append(o)
    for p in ptrs: HIP.hipFree(p)
    return np.concatenate(outs, axis=1)


def configurations():
    cfg = []
    for fs in (8000, 16000, 24000, 32000, 44100, 48000):
        for ms in (10.0, 5.0, 2.5):
            lo = {10.0: 16000 if fs != 44100 else 32000, 5.0: 32000, 2.5: 64000}[ms]
            cfg.append((fs, ms, 0, [lo, 2 * lo, 3 * lo, 4 * lo, 6 * lo, 320000 if fs != 44100 else 256000]))
    for ms, lo in ((10.0, 124800), (5.0, 148800), (2.5, 172800)):
        cfg.append((48000, ms, 1, [lo, 256000, 400000, 500000]))
    for ms, lo in ((10.0, 149600), (5.0, 174400), (2.5, 198400)):
        cfg.append((96000, ms, 1, [lo, 256000, 400000, 500000]))
    return cfg


def run(NS, B, T, verbose=True):
    """returns (frames compared, frames that differ)"""
    tot = bad = 0
    for fs, ms, hr, rates in configurations():
        N = int(round((48000 if fs == 44100 else fs) * ms / 1000))
        for seed in range(NS):
            br = np.array([rates[(i + seed) % len(rates)] for i in range(B)], np.int32)
            pcm = synth_pcm(B, T, N, fs, seed=4000 + 17 * seed)
            plan = np.zeros((B, T), np.int32)
            if BW and not READY:
                lp = [i for i in range(B) if i % 8 == 5]
                pcm = band_limit(pcm, 48000 if fs == 44100 else fs, lp, [(4000, 8000, 12000, 16000)[(i // 8) % 4] for i in lp])
                plan = bandwidth_plan(B, T, fs, hr, [T // 3], seed)
            b = audio_codec_amd.Batch(B, fs, 1, ms, hr, list(map(int, br)), device=0)
            if READY:     # device pointers, the input-ready promise, calls of READY frames queued back to back (consecutive calls overlap)
                got = encode_ready(b, pcm, READY)
            else:
                for i in np.nonzero(plan[:, 0])[0]: assert b.set_bandwidth(int(i), int(plan[i, 0])) in (0, 18)
                g1 = b.encode(pcm[:, :T // 3])
                for i in np.nonzero(plan[:, T // 3])[0]: assert b.set_bandwidth(int(i), int(plan[i, T // 3])) in (0, 18)
                got = np.concatenate([g1, b.encode(pcm[:, T // 3:])], axis=1)
            want = np.zeros_like(got)
            rc = L.lc3o_encode_batch16_bw(fs, ms, hr, B, T, br.ctypes.data, np.ascontiguousarray(plan).ctypes.data, np.ascontiguousarray(pcm).ctypes.data, want.ctypes.data, b.stride)
            assert rc == 0, rc
            nb = np.array([b.num_bytes(i) for i in range(B)])
            d = sum(int((got[i, :, :nb[i]] != want[i, :, :nb[i]]).any(axis=1).sum()) for i in range(B))
            tot += B * T; bad += d
            if d and verbose: print("%6d Hz %4.1f ms hr%d seed %d: %d of %d frames differ" % (fs, ms, hr, seed, d, B * T))
    return tot, bad


if __name__ == "__main__":
    NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 48
    t0 = time.time()
    tot, bad = run(NS, B, T)
    print("soak: %d frames over %d configurations x %d seeds, %d differ, %.0f s" % (tot, len(configurations()), NS, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
