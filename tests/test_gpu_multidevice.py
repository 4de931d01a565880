"""Two devices from one process (GPU box with >= 2 visible devices; skipped on the one-GPU boxes): a batch on device 0 and one on device 1
run side by side and both match the CPU oracle; a checkpoint taken on one device continues byte for byte on the other (streams
shard with no exchange, so a rank's device index is all that differs)."""
import numpy as np
import pytest

from lc3_harness import synth_pcm, oracle_encode_streams

pytestmark = pytest.mark.gpu


def _ndev():
    import ctypes as C
    try:
        hip = C.CDLL("libamdhip64.so")
        n = C.c_int(0)
        return n.value if hip.hipGetDeviceCount(C.byref(n)) == 0 else 0
    except OSError:
        return 0


@pytest.mark.skipif(_ndev() < 2, reason="needs two visible devices")
def test_two_devices_one_process_and_state_migration():
    import audio_codec_amd
    B, T, N, fs = 24, 20, 480, 48000
    rates = [64000, 32000, 128000] * 8
    pcm = synth_pcm(2 * B, T, N, fs, seed=77)
    b0 = audio_codec_amd.Batch(B, fs, 1, 10.0, 0, rates, device=0)
    b1 = audio_codec_amd.Batch(B, fs, 1, 10.0, 0, rates, device=1)
    g0a, g1a = b0.encode(pcm[:B, :12]), b1.encode(pcm[B:, :12])
    # the state of device 1's streams continues on a fresh batch of device 0 (and the other way round)
    s0, s1 = b0.get_state(), b1.get_state()
    c0 = audio_codec_amd.Batch(B, fs, 1, 10.0, 0, rates, device=0); c0.set_state(s1)
    c1 = audio_codec_amd.Batch(B, fs, 1, 10.0, 0, rates, device=1); c1.set_state(s0)
    g1b, g0b = c0.encode(pcm[B:, 12:]), c1.encode(pcm[:B, 12:])
    want = oracle_encode_streams(pcm, fs, 10.0, 0, rates + rates, portable_math=True)
    got = np.concatenate([np.concatenate([g0a, g0b], axis=1), np.concatenate([g1a, g1b], axis=1)], axis=0)
    bad = sum(int((got[i, :, :want[i].shape[1]] != want[i]).any(axis=1).sum()) for i in range(2 * B))
    assert bad == 0, bad
    for b in (b0, b1, c0, c1): b.close()
