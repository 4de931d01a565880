"""CPU: the oracle restatement (oracle/lc3_oracle.c) must reproduce, byte for byte, the golden vectors that
tests/golden/make_golden.py generated from the unmodified ETSI reference (oracle/_ref)."""
import glob, os
import numpy as np
import pytest
from lc3_harness import Oracle, oracle_encode_streams, synth_pcm

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


@pytest.mark.parametrize("name", ["c1_48k_10ms_64k", "c5_48k_10ms_mixed", "c4_96k_2p5ms_hr_256k", "c0_thetest48_64k_first64"])
@pytest.mark.parametrize("portable", [False, True])
def test_mono_golden(name, portable):
    g = load(name)
    pcm = g["pcm"]
    rates = g["bitrates"] if "bitrates" in g else [64000] * pcm.shape[0]
    outs = oracle_encode_streams(pcm, int(g["fs"]), float(g["frame_ms"]), int(g["hrmode"]), rates, portable_math=portable)
    for b, o in enumerate(outs):
        want = g["frames"][b][:, :o.shape[1]]
        same = (o == want).all(axis=1)
        if portable:
            # libm boundary: (float)f((double)x) vs glibc float libm may flip a rare decision
            assert same.mean() >= 0.97, (name, b, same.mean())
        else:
            assert same.all(), (name, b, np.where(~same)[0][:4])


def test_stereo_golden():
    g = load("c3_48k_10ms_stereo_128k")
    for i in range(g["pcm"].shape[0]):
        o = Oracle(48000, 2, 10.0, 0, 128000)
        for t in range(g["pcm"].shape[2]):
            got = o.encode(g["pcm"][i, :, t])
            assert (got == g["frames"][i, t]).all(), (i, t)


def test_generator_is_deterministic():
    g = load("c1_48k_10ms_64k")
    pcm = synth_pcm(64, 24, 480, 48000)[g["streams"]]
    assert (pcm == g["pcm"]).all()


def _c6():
    g = load("c6_other_operating_points")
    for tag in g["tags"]:
        fs, dms, hr, N = (int(v) for v in g[str(tag) + "_cfg"])
        yield str(tag), fs, dms / 10.0, hr, N, g[str(tag) + "_rates"], g[str(tag) + "_pcm"], g[str(tag) + "_frames"], g[str(tag) + "_nbytes"]


def test_other_operating_points_golden():
    """one small reference vector per (sample rate, frame length, mode) family outside 48 kHz / 10 ms"""
    for tag, fs, ms, hr, N, rates, pcm, frames, nbytes in _c6():
        outs = oracle_encode_streams(pcm, fs, ms, hr, rates)
        for b, o in enumerate(outs):
            assert o.shape[1] == nbytes[b] and (o == frames[b][:, :nbytes[b]]).all(), (tag, b)


def _d1():
    g = load("d1_decoder_operating_points")
    for tag in g["tags"]:
        tag = str(tag)
        fs, dms, hr, ch = (int(v) for v in g[tag + "_cfg"])
        yield tag, fs, dms / 10.0, hr, ch, g[tag + "_frames"], g[tag + "_nbytes"], g[tag + "_bfi"], g[tag + "_pcm"], g[tag + "_status"]


@pytest.mark.parametrize("portable", [False, True])
def test_decoder_golden(portable):
    """the decoder restatement against what the unmodified ETSI decoder made of damaged bitstreams (lost and corrupt frames included):
    PCM sample for sample and the concealment status, on every fixture operating point, with either math build"""
    from lc3_harness import oracle_decode_streams
    for tag, fs, ms, hr, ch, frames, nbytes, bfi, pcm, status in _d1():
        got, st = oracle_decode_streams(frames, nbytes, bfi, fs, ms, hr, ch, portable_math=portable)
        assert (st == status).all(), tag
        assert (got == pcm).all(), (tag, np.argwhere((got != pcm).any(axis=(2, 3)))[:3].tolist())


ETSI_WAV = "/root/reference/LC3plus_ETSI_src_v17171_20200723/testvec/input/thetest48.wav"


@pytest.mark.skipif(not os.path.exists(ETSI_WAV), reason="the reference's test vector is only mounted in the build container")
def test_config0_full_length_bitstream_md5():
    """BASELINE configs[0] / SURVEY 8(d) config 1 at FULL length: E/testvec/input/thetest48.wav (484 191 samples = 1008 whole frames + a partial one, which
    the reference's CLI pads with zeros: 1009 frames) at 64 kbps through the oracle (glibc math, i.e. the build that is byte-identical to the compiled
    reference).  (1) The 1008 whole frames against the digest tests/golden/make_golden.py took from the unmodified ETSI encoder's bitstream of the same
    PCM; (2) all 1009 frames, the zero-padded last one included, byte for byte against the compiled reference run live (oracle/_ref, when built).  The WAV
    itself is not copied into the repo: the fixture holds its first 64 frames plus the md5 of the PCM and of the full bitstream."""
    import hashlib, wave
    from lc3_harness import have_ref, ref_encode_streams
    g = load("c0_thetest48_64k_first64")
    w = wave.open(ETSI_WAV)
    x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
    T = x.size // 480
    assert T == int(g["full_frames"]) == 1008 and x.size % 480 != 0
    padded = np.zeros((T + 1) * 480, np.int16); padded[:x.size] = x
    pcm = padded.reshape(1, T + 1, 480)
    assert hashlib.md5(pcm[:, :T].tobytes()).hexdigest() == str(g["full_pcm_md5"])
    out = oracle_encode_streams(pcm, 48000, 10.0, 0, [64000])[0]
    assert out.shape == (1009, 80)
    assert hashlib.md5(out[:T].tobytes()).hexdigest() == str(g["full_bitstream_md5"])
    assert (out[:64] == g["frames"][0]).all()
    if have_ref():
        ref = ref_encode_streams(pcm, 48000, 10.0, 0, [64000])[0]
        assert (ref == out).all(), np.where((ref != out).any(axis=1))[0][:4]
