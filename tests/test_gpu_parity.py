"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI (liblc3plus_hip.so),
against (1) the committed golden vectors generated from the unmodified ETSI reference and (2) the CPU oracle on the
same seeded inputs.

Tolerance (stated, SURVEY 8c / BASELINE north_star): integer stages are bit-exact by construction.  The reference's
run-time libm calls (log2f/log10f/powf) are evaluated on the device as (float)f((double)x), so against the glibc-math
reference vectors a decision could in principle flip; the conformance procedure's tolerance for that is a maximum
loudness difference of 4 between the two decoded signals (E/conformance/lc3_conformance.py:127).  Observed and
therefore gated: ZERO differing frames against the reference vectors and against the oracle built with the device's
math (oracle/liblc3_oracle_pm.so).  Should a frame ever differ, compare_frames() decodes both bitstreams with the
compiled reference decoder and reports the MLD next to the count, so that the failure says whether it is a boundary
flip (MLD <= 4) or a defect."""
import os
import numpy as np
import pytest

from lc3_harness import Oracle, oracle_encode_streams, synth_pcm, compare_frames, MLD_THRESHOLD

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RATES = [16000, 24000, 32000, 48000, 64000, 80000, 96000, 128000, 160000, 192000, 256000, 320000]


def _amd():
    import audio_codec_amd
    return audio_codec_amd


def _frames_equal(got, want_list):
    tot = same = 0
    for b, w in enumerate(want_list):
        eq = (got[b, :, :w.shape[1]] == w).all(axis=1)
        tot += eq.size; same += int(eq.sum())
    return same, tot


@pytest.mark.parametrize("name", ["c1_48k_10ms_64k", "c5_48k_10ms_mixed", "c4_96k_2p5ms_hr_256k", "c0_thetest48_64k_first64"])
def test_golden_mono(name):
    g = np.load(os.path.join(G, name + ".npz"))
    pcm = g["pcm"]
    rates = g["bitrates"] if "bitrates" in g else [64000] * pcm.shape[0]
    b = _amd().Batch(pcm.shape[0], int(g["fs"]), 1, float(g["frame_ms"]), int(g["hrmode"]), rates, device=0)
    got = b.encode(pcm)
    nb = g["nbytes"] if "nbytes" in g else [g["frames"].shape[2]] * pcm.shape[0]
    diff, tot, mld = compare_frames(got, [g["frames"][i][:, :nb[i]] for i in range(pcm.shape[0])], int(g["fs"]), float(g["frame_ms"]), int(g["hrmode"]))
    assert diff == 0, (name, diff, tot, "worst MLD of the differing streams", mld)


def test_golden_stereo():
    g = np.load(os.path.join(G, "c3_48k_10ms_stereo_128k.npz"))
    pcm = np.ascontiguousarray(g["pcm"].transpose(0, 2, 1, 3))      # [pair, T, ch, N]
    b = _amd().Batch(pcm.shape[0], 48000, 2, 10.0, 0, [128000] * pcm.shape[0], device=0)
    got = b.encode(pcm)
    diff, tot, mld = compare_frames(got, list(g["frames"]), 48000, 10.0, 0, channels=2)
    assert diff == 0, (diff, tot, "worst MLD of the differing streams", mld)


@pytest.mark.parametrize("fs,ms,hr,N,rates", [
    (48000, 10.0, 0, 480, RATES * 6),
    (96000, 2.5, 1, 240, [256000, 198400, 320000, 672000] * 8),
    (48000, 5.0, 0, 240, [32000, 64000, 96000, 128000, 256000, 320000] * 2),
    (24000, 10.0, 0, 240, [16000, 32000, 64000, 128000] * 2),
    (48000, 10.0, 1, 480, [128000, 256000, 400000, 500000] * 2),
    (48000, 5.0, 1, 240, [160000, 320000, 600000] * 2),
    (44100, 10.0, 0, 480, [32000, 64000, 128000, 256000] * 2),
    # the other sample rates / frame lengths (N/2-point DFT by prime factors or 4 x 15): SURVEY 8f-4
    (8000, 10.0, 0, 80, [16000, 24000, 32000, 64000] * 2),
    (16000, 10.0, 0, 160, [16000, 32000, 64000, 128000] * 2),
    (16000, 5.0, 0, 80, [32000, 64000, 96000, 128000] * 2),
    (16000, 2.5, 0, 40, [64000, 96000, 128000, 192000] * 2),
    (8000, 5.0, 0, 40, [32000, 48000, 64000, 96000] * 2),
    (8000, 2.5, 0, 20, [64000, 96000, 128000, 160000] * 2),
    (24000, 5.0, 0, 120, [32000, 64000, 96000, 160000] * 2),
    (24000, 2.5, 0, 60, [64000, 96000, 128000, 256000] * 2),
    (32000, 10.0, 0, 320, [32000, 64000, 96000, 128000, 192000, 320000] * 2),
    (32000, 5.0, 0, 160, [32000, 64000, 96000, 192000] * 2),
    (32000, 2.5, 0, 80, [64000, 96000, 128000, 256000] * 2),
    (48000, 2.5, 0, 120, [64000, 96000, 128000, 320000] * 2),
    (44100, 5.0, 0, 240, [64000, 128000] * 2),
    (48000, 2.5, 1, 120, [172800, 256000, 400000] * 2),
    # the large-layout kernel: N = 960, or an MDCT memory of 360 samples
    (96000, 5.0, 1, 480, [256000, 400000, 600000] * 2),
    (96000, 10.0, 1, 960, [149600, 256000, 400000, 500000] * 2),
])
def test_vs_oracle_same_math(fs, ms, hr, N, rates):
    B, T = len(rates), 24
    pcm = synth_pcm(B, T, N, fs, seed=21)
    b = _amd().Batch(B, fs, 1, ms, hr, rates, device=0)
    got = np.concatenate([b.encode(pcm[:, :10]), b.encode(pcm[:, 10:])], axis=1)   # two launches: state must persist
    want = oracle_encode_streams(pcm, fs, ms, hr, rates, portable_math=True)
    same, tot = _frames_equal(got, want)
    assert same == tot, (same, tot)


def test_single_stream_api_matches_batch_and_oracle():
    amd = _amd()
    pcm = synth_pcm(2, 12, 480, 48000, seed=5)
    e = amd.Encoder(48000, 2, 10.0, 0, 128000)
    o = Oracle(48000, 2, 10.0, 0, 128000, portable_math=True)
    for t in range(12):
        assert (e.encode(pcm[:, t]) == o.encode(pcm[:, t])).all(), t
    # error behaviour of the C API (R/lc3.c:102-208)
    import ctypes as C
    L = amd.load_library()
    buf = C.create_string_buffer(L.lc3_enc_get_size(48000, 1)); p = C.cast(buf, C.c_void_p)
    assert L.lc3_enc_init(p, 12345, 1) == 4 and L.lc3_enc_init(p, 48000, 3) == 5 and L.lc3_enc_init(None, 48000, 1) == 3
    assert L.lc3_enc_init(p, 48000, 1) == 0 and L.lc3_enc_set_frame_ms(p, 7.5) == 9
    assert L.lc3_enc_set_bitrate(p, 8000) == 6 and L.lc3_enc_set_bitrate(p, 64000) == 0
    assert L.lc3_enc_set_frame_ms(p, 5.0) == 13 and L.lc3_enc_set_hrmode(p, 0) == 0
    assert L.lc3_enc_get_num_bytes(p) == 80 and L.lc3_enc_get_input_samples(p) == 480 and L.lc3_enc_get_delay(p) == 120
    assert L.lc3_enc_get_real_bitrate(p) == 64000


def test_bitrate_switch_bandwidth_bitdepth():
    amd = _amd()
    pcm = synth_pcm(3, 30, 480, 48000, seed=9)
    b = amd.Batch(1, 48000, 1, 10.0, 0, [128000], device=0)
    o = Oracle(48000, 1, 10.0, 0, 128000, portable_math=True)
    for t in range(30):
        if t == 10:
            assert b.set_bitrate(0, 32000) == 0 and o.set_bitrate(32000) == 0; o.nbytes = 40
        if t == 20:
            assert b.set_bitrate(0, 192000) == 0 and o.set_bitrate(192000) == 0; o.nbytes = 240
        got = b.encode(pcm[0:1, t:t + 1])[0, 0, :o.nbytes]
        assert (got == o.encode(pcm[0, t][None])).all(), t
    b = amd.Batch(1, 48000, 1, 10.0, 0, [64000], device=0); assert b.set_bandwidth(0, 8000) == 0
    o = Oracle(48000, 1, 10.0, 0, 64000, portable_math=True, bandwidth=8000)
    for t in range(20):
        assert (b.encode(pcm[1:2, t:t + 1])[0, 0, :80] == o.encode(pcm[1, t][None])).all(), t
    for depth, scale in ((24, 200), (32, 60000)):
        p32 = pcm[2].astype(np.int32) * scale + 77
        b = amd.Batch(1, 48000, 1, 10.0, 0, [96000], device=0)
        o = Oracle(48000, 1, 10.0, 0, 96000, portable_math=True)
        got = b.encode(p32[None], bitdepth=depth)
        for t in range(30):
            assert (got[0, t, :120] == o.encode(p32[t][None], depth)).all(), (depth, t)


def test_empty_and_edge_inputs():
    amd = _amd()
    z = np.zeros((2, 5, 480), np.int16)                     # digital silence
    full = np.full((2, 5, 480), 32767, np.int16); full[:, :, ::2] = -32768   # full-scale Nyquist square
    for pcm in (z, full):
        b = amd.Batch(2, 48000, 1, 10.0, 0, [16000, 320000], device=0)
        got = b.encode(pcm)
        want = oracle_encode_streams(pcm, 48000, 10.0, 0, [16000, 320000], portable_math=True)
        same, tot = _frames_equal(got, want)
        assert same == tot


def _write_wav(path, pcm_interleaved, fs, channels, bits):
    import wave
    w = wave.open(str(path), "wb")
    w.setnchannels(channels); w.setsampwidth(bits // 8); w.setframerate(fs)
    if bits == 16:
        w.writeframes(pcm_interleaved.astype("<i2").tobytes())
    else:                                     # 24-bit little endian
        v = pcm_interleaved.astype(np.int32)
        b = np.stack([(v & 0xff), (v >> 8) & 0xff, (v >> 16) & 0xff], axis=-1).astype(np.uint8)
        w.writeframes(b.tobytes())
    w.close()


def _container(frames, fs, bitrate, channels, frame_ms, nsamples, hrmode):
    """The .lc3plus container of R/codec_exe.c:651-661,742-748 around a list of frame payloads."""
    hdr = np.array([0xcc1c, 20, fs // 100, bitrate // 100, channels, int(frame_ms * 100), 0, nsamples & 0xffff, nsamples >> 16, hrmode], np.uint16)
    out = [hdr.tobytes()]
    for f in frames:
        out.append(np.uint16(len(f)).tobytes()); out.append(bytes(f))
    return b"".join(out)


@pytest.mark.parametrize("channels,bits,bitrate", [(1, 16, 64000), (2, 24, 128000)])
def test_cli_front_end_writes_the_reference_container(tmp_path, channels, bits, bitrate):
    """tools/lc3plus_enc_cli (C, on the C ABI) against the ETSI CLI (oracle/_ref/LC3plus, when built) and the oracle."""
    import subprocess
    from lc3_harness import ORACLE_DIR
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "tools", "lc3plus_enc_cli")
    if not os.path.exists(cli):
        subprocess.check_call(["make", "-s", "-C", root, "cli"])
    nsamp = 480 * 37 + 123                                      # partial last frame: zero padded (R/codec_exe.c:329-338)
    pcm = synth_pcm(channels, 38, 480, 48000, seed=31).reshape(channels, -1)[:, :nsamp]
    if bits == 24:
        pcm = pcm.astype(np.int32) * 200 + 7
    wav = tmp_path / "in.wav"
    _write_wav(wav, pcm.T.reshape(-1), 48000, channels, bits)
    ours = tmp_path / "ours.lc3plus"
    subprocess.check_call([cli, "-E", "-q", str(wav), str(ours), str(bitrate)])
    got = open(ours, "rb").read()
    # expected container from the oracle (same math as the device)
    o = Oracle(48000, channels, 10.0, 0, bitrate, portable_math=True)
    padded = np.zeros((channels, 38 * 480), pcm.dtype); padded[:, :nsamp] = pcm
    frames = [o.encode(padded[:, t * 480:(t + 1) * 480], bits) for t in range(38)]
    assert got == _container(frames, 48000, bitrate, channels, 10.0, nsamp, 0)
    ref_cli = os.path.join(ORACLE_DIR, "_ref", "LC3plus")
    if os.path.exists(ref_cli):                                 # the real thing, when it travelled with the snapshot
        theirs = tmp_path / "ref.lc3plus"
        subprocess.check_call([ref_cli, "-E", "-q", str(wav), str(theirs), str(bitrate)], stdout=subprocess.DEVNULL)
        ref = open(theirs, "rb").read()
        assert len(ref) == len(got) and ref[:20] == got[:20]
        fr = 2 + o.nbytes
        same = sum(ref[20 + i * fr:20 + (i + 1) * fr] == got[20 + i * fr:20 + (i + 1) * fr] for i in range(38))
        assert same == 38, same


def test_cli_switching_files(tmp_path):
    """-swf / -bandwidth FILE (one int64 per frame, wrapping: R/codec_exe.c:296-326,858-866) against the oracle, and against the
    ETSI CLI when it travelled with the snapshot."""
    import subprocess
    from lc3_harness import ORACLE_DIR
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "tools", "lc3plus_enc_cli")
    subprocess.check_call(["make", "-s", "-C", root, "cli"])
    nfr = 40
    pcm = synth_pcm(1, nfr, 480, 48000, seed=77).reshape(1, -1)
    wav = tmp_path / "in.wav"
    _write_wav(wav, pcm.T.reshape(-1), 48000, 1, 16)
    rates = [64000] * 7 + [32000] * 5 + [128000] * 3 + [96000] * 4         # 19 entries: wraps twice over 40 frames
    bws = [20000] * 6 + [8000] * 9 + [16000] * 2                            # 17 entries
    swf = tmp_path / "rates.bin"; np.array(rates, np.int64).tofile(swf)
    bwf = tmp_path / "bw.bin"; np.array(bws, np.int64).tofile(bwf)
    ours = tmp_path / "ours.lc3plus"
    subprocess.check_call([cli, "-E", "-q", "-swf", str(swf), "-bandwidth", str(bwf), str(wav), str(ours), "64000"])
    got = open(ours, "rb").read()
    o = Oracle(48000, 1, 10.0, 0, 64000, portable_math=True)
    frames = []
    for t in range(nfr):
        assert o.set_bitrate(rates[t % len(rates)]) == 0
        o.set_bandwidth(bws[t % len(bws)])
        frames.append(o.encode(pcm[:, t * 480:(t + 1) * 480], 16))
    assert got == _container(frames, 48000, 64000, 1, 10.0, nfr * 480, 0)
    ref_cli = os.path.join(ORACLE_DIR, "_ref", "LC3plus")
    if os.path.exists(ref_cli):
        theirs = tmp_path / "ref.lc3plus"
        subprocess.check_call([ref_cli, "-E", "-q", "-swf", str(swf), "-bandwidth", str(bwf), str(wav), str(theirs), "64000"], stdout=subprocess.DEVNULL)
        ref = open(theirs, "rb").read()
        assert len(ref) == len(got) and ref[:20] == got[:20]
        assert ref == got


def test_cli_encoder_g192_format_and_cfg_file(tmp_path):
    """tools/lc3plus_enc_cli -formatG192 [-cfgG192 FILE] (R/codec_exe.c:705-749 write_bitstream_frame_G192: sync word 0x6B21, the frame's length in bits,
    one int16 per bit - 0x0081 for 1, 0x007F for 0 - and the 20-byte header in a side file) against the ETSI CLI where it travelled (whole files, byte for
    byte) and against the oracle's frames re-packed here; then both G.192 files back through the two decoders (tools/lc3plus_dec_cli vs LC3plus -D)."""
    import subprocess
    from lc3_harness import ORACLE_DIR
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "tools", "lc3plus_enc_cli"); dcli = os.path.join(root, "tools", "lc3plus_dec_cli")
    subprocess.check_call(["make", "-s", "-C", root, "cli"])
    nfr, nsamp = 21, 480 * 20 + 77
    pcm = synth_pcm(2, nfr, 480, 48000, seed=192).reshape(2, -1)[:, :nsamp]
    wav = tmp_path / "in.wav"
    _write_wav(wav, pcm.T.reshape(-1), 48000, 2, 16)
    for cfg_opt in (False, True):
        ours = tmp_path / ("ours%d.g192" % cfg_opt); cfg = tmp_path / ("ours%d.side" % cfg_opt)
        args = [cli, "-E", "-q", "-formatG192"] + (["-cfgG192", str(cfg)] if cfg_opt else []) + [str(wav), str(ours), "128000"]
        subprocess.check_call(args)
        cfg_path = cfg if cfg_opt else str(ours) + ".cfg"
        got, got_cfg = open(ours, "rb").read(), open(cfg_path, "rb").read()
        o = Oracle(48000, 2, 10.0, 0, 128000, portable_math=True)
        padded = np.zeros((2, nfr * 480), np.int16); padded[:, :nsamp] = pcm
        want = bytearray()
        for t in range(nfr):
            fr = o.encode(padded[:, t * 480:(t + 1) * 480])
            bits = np.unpackbits(fr, bitorder="little")
            want += np.array([0x6B21, fr.size * 8], "<u2").tobytes() + np.where(bits, 0x0081, 0x007F).astype("<u2").tobytes()
        assert got == bytes(want)
        assert got_cfg == _container([], 48000, 128000, 2, 10.0, nsamp, 0)[:20]
        ref_cli = os.path.join(ORACLE_DIR, "_ref", "LC3plus")
        if os.path.exists(ref_cli):
            theirs = tmp_path / ("ref%d.g192" % cfg_opt); tcfg = tmp_path / ("ref%d.side" % cfg_opt)
            subprocess.check_call([ref_cli, "-E", "-q", "-formatG192"] + (["-cfgG192", str(tcfg)] if cfg_opt else []) + [str(wav), str(theirs), "128000"], stdout=subprocess.DEVNULL)
            assert open(theirs, "rb").read() == got
            assert open(tcfg if cfg_opt else str(theirs) + ".cfg", "rb").read() == got_cfg
            a, b = tmp_path / "a.wav", tmp_path / "b.wav"
            subprocess.check_call([dcli, "-D", "-q", "-formatG192"] + (["-cfgG192", str(cfg)] if cfg_opt else []) + [str(ours), str(a)])
            subprocess.check_call([ref_cli, "-D", "-q", "-formatG192"] + (["-cfgG192", str(tcfg)] if cfg_opt else []) + [str(theirs), str(b)], stdout=subprocess.DEVNULL)
            assert open(a, "rb").read() == open(b, "rb").read()


def _oracle_batch(pcm, fs, ms, hr, rates, stride):
    """All streams through the oracle's C batch entry (same math as the device), [B, T, stride] uint8."""
    import ctypes as C
    from lc3_harness import ORACLE_DIR
    L = C.CDLL(os.path.join(ORACLE_DIR, "liblc3_oracle_pm.so"))
    L.lc3o_encode_batch16.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    B, T, N = pcm.shape
    out = np.zeros((B, T, stride), np.uint8)
    br = np.asarray(rates, np.int32)
    pcm = np.ascontiguousarray(pcm)
    assert L.lc3o_encode_batch16(fs, ms, hr, B, T, br.ctypes.data, pcm.ctypes.data, out.ctypes.data, stride) == 0
    return out


@pytest.mark.parametrize("fs,ms,hr,N,B,T,rates", [
    (48000, 10.0, 0, 480, 256, 48, [64000]),                       # the metric's configuration: 12 288 frames
    (48000, 10.0, 0, 480, 192, 32, RATES),                         # mixed bitrates 16-320 kbps (attack detector, LSB mode, LTPF off, LPC weighting)
    (96000, 2.5, 1, 240, 64, 64, [256000, 198400, 672000, 400000]),
    (16000, 10.0, 0, 160, 96, 32, [16000, 24000, 32000, 64000]),
    (32000, 10.0, 0, 320, 64, 32, [32000, 64000, 96000, 128000]),
])
def test_large_sweep_vs_oracle(fs, ms, hr, N, B, T, rates):
    """Thousands of frames per configuration (1 stream in 64 silent, 1 in 64 full-scale noise): the rare paths - a range coder that
    ends in its carry_count branch, symbol lists longer than one chunk, gain clamps - have to come out identical too."""
    br = [rates[i % len(rates)] for i in range(B)]
    pcm = synth_pcm(B, T, N, fs, seed=1009)
    b = _amd().Batch(B, fs, 1, ms, hr, br, device=0)
    got = np.concatenate([b.encode(pcm[:, :T // 2]), b.encode(pcm[:, T // 2:])], axis=1)
    want = _oracle_batch(pcm, fs, ms, hr, br, b.stride)
    nb = np.array([b.num_bytes(i) for i in range(B)])
    bad = [(i, t) for i in range(B) for t in range(T) if (got[i, t, :nb[i]] != want[i, t, :nb[i]]).any()]
    assert not bad, (len(bad), bad[:8])


def test_soak_every_operating_point():
    """All 24 (sample rate, frame length, mode) families of the reference x 2 seeds x 32 streams x 36 frames against the oracle
    (tools/soak.py; the full-size run, 3.1 M frames, is quoted in DESIGN.md)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("soak", os.path.join(root, "tools", "soak.py"))
    soak = importlib.util.module_from_spec(spec); spec.loader.exec_module(soak)
    tot, bad = soak.run(2, 32, 36, verbose=False)
    assert tot == 24 * 2 * 32 * 36 and bad == 0, (tot, bad)


def test_golden_other_operating_points():
    """the reference's own vectors (tests/golden/c6, generated by make_golden.py) for every family outside 48 kHz / 10 ms"""
    g = np.load(os.path.join(G, "c6_other_operating_points.npz"))
    tot = 0
    for tag in g["tags"]:
        tag = str(tag)
        fs, dms, hr, N = (int(v) for v in g[tag + "_cfg"])
        rates, pcm, frames, nbytes = g[tag + "_rates"], g[tag + "_pcm"], g[tag + "_frames"], g[tag + "_nbytes"]
        b = _amd().Batch(len(rates), fs, 1, dms / 10.0, hr, [int(r) for r in rates], device=0)
        got = b.encode(pcm)
        d, t, mld = compare_frames(got, [frames[i][:, :nbytes[i]] for i in range(len(rates))], fs, dms / 10.0, hr)
        assert d == 0, (tag, d, t, "worst MLD of the differing streams", mld)
        tot += t
    assert tot > 0


@pytest.mark.parametrize("fs,ms,hr,br", [(16000, 10.0, 0, 32000), (8000, 2.5, 0, 64000), (32000, 10.0, 0, 64000), (96000, 10.0, 1, 256000)])
def test_single_stream_api_other_geometries(fs, ms, hr, br):
    amd = _amd()
    N = int(fs * ms / 1000)
    pcm = synth_pcm(1, 10, N, fs, seed=9)
    e = amd.Encoder(fs, 1, ms, hr, br)
    o = Oracle(fs, 1, ms, hr, br, portable_math=True)
    for t in range(10):
        assert (e.encode(pcm[:, t]) == o.encode(pcm[:, t])).all(), t


def _oracle_batch_ch(pcm, fs, ms, hr, channels, rates, stride):
    """[B, T, channels, N] int16 through the oracle's C batch entry (device math) -> [B, T, stride] uint8."""
    import ctypes as C
    from lc3_harness import ORACLE_DIR
    L = C.CDLL(os.path.join(ORACLE_DIR, "liblc3_oracle_pm.so"))
    L.lc3o_encode_batch16_ch.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    B, T = pcm.shape[:2]
    out = np.zeros((B, T, stride), np.uint8)
    br = np.asarray(rates, np.int32)
    pcm = np.ascontiguousarray(pcm)
    assert L.lc3o_encode_batch16_ch(fs, ms, hr, channels, B, T, br.ctypes.data, pcm.ctypes.data, out.ctypes.data, stride) == 0
    return out


@pytest.mark.parametrize("fs,ms,br,nbytes", [
    (48000, 10.0, 80000, 100), (48000, 10.0, 96000, 120), (48000, 10.0, 100000, 125), (48000, 10.0, 102400, 128),
    (32000, 10.0, 96000, 120), (48000, 10.0, 104000, 130), (48000, 10.0, 64000, 80), (48000, 5.0, 192000, 120),
])
def test_uniform_size_batches_through_the_pack_kernel(fs, ms, br, nbytes):
    """Batches in which EVERY stream has the same frame size, at the sizes round 1 never compared on the two-kernel path: 10 ms
    frames of 81 ... 128 bytes (48 kHz 80 / 96 / 100 kbps, 32 kHz 96 kbps: attack detector on, LTPF off, 200 tuples, the longest symbol
    and LSB lists that fit 128 bytes) and the first sizes above.  Two launches; vs the oracle with the device's math.
    (R/enc_entropy.c:13-88, R/ari_codec.c:673-800)"""
    B, T = 128, 24
    N = int(fs * ms / 1000)
    pcm = synth_pcm(B, T, N, fs, seed=4242)
    b = _amd().Batch(B, fs, 1, ms, 0, [br] * B, device=0)
    assert b.num_bytes(0) == nbytes
    got = np.concatenate([b.encode(pcm[:, :T // 2]), b.encode(pcm[:, T // 2:])], axis=1)
    want = _oracle_batch(pcm, fs, ms, 0, [br] * B, b.stride)
    bad = [(i, t) for i in range(B) for t in range(T) if (got[i, t, :nbytes] != want[i, t, :nbytes]).any()]
    assert not bad, (len(bad), bad[:8])


def test_fused_writer_still_matches(monkeypatch):
    """The wave-parallel bitstream writer inside lc3_encode_kernel (st_bitstream) now only runs for traced / diagnostic launches
    (LC3PLUS_ENC_FUSED=1); it has to stay identical to the pack kernel and the oracle."""
    monkeypatch.setenv("LC3PLUS_ENC_FUSED", "1")
    B, T = 48, 16
    rates = [RATES[i % len(RATES)] for i in range(B)]
    pcm = synth_pcm(B, T, 480, 48000, seed=515)
    b = _amd().Batch(B, 48000, 1, 10.0, 0, rates, device=0)
    got = b.encode(pcm)
    monkeypatch.delenv("LC3PLUS_ENC_FUSED")
    b2 = _amd().Batch(B, 48000, 1, 10.0, 0, rates, device=0)
    got2 = b2.encode(pcm)
    want = _oracle_batch(pcm, 48000, 10.0, 0, rates, b.stride)
    assert (got == want).all() and (got2 == want).all()


@pytest.mark.parametrize("B,T,rates", [(1024, 16, [128000]), (96, 24, [64000, 128000, 256000, 61600, 32000, 640000])])
def test_large_stereo_sweep_vs_oracle(B, T, rates):
    """BASELINE configs[2]: stereo streams, 128 kbps per pair (80 bytes per channel) - 1 024 pairs x 16 frames =
    32 768 channel-frames in one batch, the shape the 8-GPU run shards; second case: other stereo rates including an odd per-channel
    split (61.6 kbps -> 77 bytes: 39 + 38, unaligned second payload) and the largest frames (400 bytes per channel)."""
    br = [rates[i % len(rates)] for i in range(B)]
    pcm = synth_pcm(B * 2, T, 480, 48000, seed=2027).reshape(B, 2, T, 480).transpose(0, 2, 1, 3)
    pcm = np.ascontiguousarray(pcm)
    b = _amd().Batch(B, 48000, 2, 10.0, 0, br, device=0)
    got = np.concatenate([b.encode(pcm[:, :T // 2]), b.encode(pcm[:, T // 2:])], axis=1)
    want = _oracle_batch_ch(pcm, 48000, 10.0, 0, 2, br, b.stride)
    nb = np.array([b.num_bytes(i) for i in range(B)])
    bad = [(i, t) for i in range(B) for t in range(T) if (got[i, t, :nb[i]] != want[i, t, :nb[i]]).any()]
    assert not bad, (len(bad), bad[:8])


def test_mld_fallback_machinery():
    """The spectral-distance fallback itself (north_star: 'within a stated spectral-distance tolerance'): GPU
    bitstreams and the reference's golden bitstreams, both decoded by the compiled reference decoder, through the ETSI mld tool:
    0 for identical streams, and the tool is alive (a 32 kbps encode of the same PCM is further away than the threshold)."""
    from lc3_harness import have_ref, mld_between, MLD_TOOL
    if not (have_ref() and os.path.exists(MLD_TOOL)):
        pytest.skip("oracle/_ref (compiled reference decoder + mld) did not travel with this snapshot")
    g = np.load(os.path.join(G, "c1_48k_10ms_64k.npz"))
    pcm = g["pcm"][:2]
    b = _amd().Batch(2, 48000, 1, 10.0, 0, [64000, 32000], device=0)
    got = b.encode(pcm)
    assert mld_between(got[0, :, :80], g["frames"][0][:, :80], 48000, 10.0, 0) <= MLD_THRESHOLD
    assert mld_between(got[0, :, :80], g["frames"][0][:, :80], 48000, 10.0, 0) == 0.0
    b3 = _amd().Batch(1, 48000, 1, 10.0, 0, [32000], device=0)
    low = b3.encode(pcm[:1])
    assert mld_between(low[0, :, :40], g["frames"][0][:, :80], 48000, 10.0, 0) > 0.5


def test_reference_cli_relinked_against_this_library(tmp_path):
    """SURVEY 8b 'Who calls it': the reference's own CLI (R/codec_exe.c, unmodified, compiled against the reference's headers by
    `make relink`) with every codec object replaced by liblc3plus_hip.so.  -E and -D files against the all-reference build
    (oracle/_ref/LC3plus), including a rate-switching file in verbose mode: the CLI reads encoder->bitrate straight out of the
    struct (R/codec_exe.c:298-299), so the printed rates prove the field offsets (R/setup_enc_lc3.h:65-104)."""
    import subprocess
    from lc3_harness import ORACLE_DIR
    ours, theirs = os.path.join(ORACLE_DIR, "_ref", "LC3plus_hip"), os.path.join(ORACLE_DIR, "_ref", "LC3plus")
    if not (os.path.exists(ours) and os.path.exists(theirs)):
        pytest.skip("oracle/_ref/LC3plus_hip (make relink) did not travel with this snapshot")
    for channels, bitrate in ((1, 64000), (2, 128000)):
        nsamp = 480 * 30 + 77
        pcm = synth_pcm(channels, 31, 480, 48000, seed=61).reshape(channels, -1)[:, :nsamp]
        wav = tmp_path / ("in%d.wav" % channels)
        _write_wav(wav, pcm.T.reshape(-1), 48000, channels, 16)
        a, b = tmp_path / ("a%d.lc3plus" % channels), tmp_path / ("b%d.lc3plus" % channels)
        subprocess.check_call([ours, "-E", "-q", str(wav), str(a), str(bitrate)], stdout=subprocess.DEVNULL)
        subprocess.check_call([theirs, "-E", "-q", str(wav), str(b), str(bitrate)], stdout=subprocess.DEVNULL)
        assert open(a, "rb").read() == open(b, "rb").read(), ("encode", channels)
        wa, wb = tmp_path / ("a%d.wav" % channels), tmp_path / ("b%d.wav" % channels)
        subprocess.check_call([ours, "-D", "-q", str(b), str(wa)], stdout=subprocess.DEVNULL)
        subprocess.check_call([theirs, "-D", "-q", str(b), str(wb)], stdout=subprocess.DEVNULL)
        assert open(wa, "rb").read() == open(wb, "rb").read(), ("decode", channels)
    # verbose rate switching: the CLI prints encoder->bitrate before every change
    rates = [64000] * 5 + [32000] * 5 + [96000] * 5
    swf = tmp_path / "rates.bin"; np.array(rates, np.int64).tofile(swf)
    wav = tmp_path / "in1.wav"
    outs = []
    for exe, name in ((ours, "sa"), (theirs, "sb")):
        o = tmp_path / (name + ".lc3plus")
        r = subprocess.run([exe, "-E", "-v", "-swf", str(swf), str(wav), str(o), "64000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-500:]
        import re
        outs.append((re.findall(r"Switching rate from \d+ to \d+", r.stdout), open(o, "rb").read()))
    assert outs[0][0] == outs[1][0] and len(outs[0][0]) >= 2 and outs[0][0][0] == "Switching rate from 64000 to 32000", outs[0][0]
    assert outs[0][1] == outs[1][1]


def _pinned(shape, dtype):
    """numpy array in page-locked host memory (hipHostMalloc through ctypes: the tests do not depend on torch)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(ptr), C.c_size_t(n), C.c_uint(0)) == 0
    buf = (C.c_char * n).from_address(ptr.value)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def test_host_pipeline_matches_device_path():
    """lc3plus_enc_batch_encode with host pointers cuts a large call into runs of frames (H2D of run k+1 beside the kernels of run k, one
    bitstream pass and one D2H at the end): same bytes as the oracle, state carried from run to run and from call to call, for pinned
    and for pageable caller memory; the per-frame status words stay clear."""
    B, T = 2048, 40                                   # 2048 x 40 x 960 B = 75 MB of PCM: cut into three runs
    pcm = synth_pcm(B, T, 480, 48000, seed=808)
    amd = _amd()
    want = _oracle_batch(pcm, 48000, 10.0, 0, [64000] * B, 80)
    for pinned in (True, False):
        alloc = _pinned if pinned else (lambda shape, dtype: np.zeros(shape, dtype))
        hp = alloc((B, T, 480), np.int16); hp[:] = pcm
        ho = alloc((B, T, 80), np.uint8); ho[:] = 0
        b = amd.Batch(B, 48000, 1, 10.0, 0, [64000] * B, device=0)
        b.encode_host(hp, ho)
        assert (ho == want).all(), pinned
        assert not b.last_status(T).any()
        # two calls on a fresh batch: the second continues the streams where the first stopped
        b = amd.Batch(B, 48000, 1, 10.0, 0, [64000] * B, device=0)
        h1 = alloc((B, 24, 480), np.int16); h1[:] = pcm[:, :24]; o1 = alloc((B, 24, 80), np.uint8)
        h2 = alloc((B, T - 24, 480), np.int16); h2[:] = pcm[:, 24:]; o2 = alloc((B, T - 24, 80), np.uint8)
        b.encode_host(h1, o1); b.encode_host(h2, o2)
        assert (o1 == want[:, :24]).all() and (o2 == want[:, 24:]).all(), pinned


def test_attack_detector_on_the_split_path():
    """Streams with attack handling (48 kHz / 10 ms at 100 ... 339 bytes, 32 kHz from 81 bytes: R/setup_enc_lc3.c:288-307) on the path
    where the detector's block energies come from the frame-parallel front kernel, its decision from lc3_enc_attack_kernel and the
    scale-factor smoothing from the quantiser's kernel: transient-heavy PCM, several launches (the detector's state crosses them), and
    a bitrate switch that turns attack handling off and on again (the reset rule)."""
    amd = _amd()
    rng = np.random.default_rng(5)
    B, T = 96, 36
    pcm = synth_pcm(B, T, 480, 48000, seed=1234).astype(np.float64)
    for s in range(B):                                   # bursts: silence then full-scale noise, at random frames
        for t in rng.choice(T, size=5, replace=False):
            a0 = t * 480 + int(rng.integers(0, 400))
            flat = pcm[s].reshape(-1)
            flat[max(0, a0 - 960):a0] *= 0.01
            flat[a0:a0 + 300] = rng.uniform(-30000, 30000, size=flat[a0:a0 + 300].shape)
    pcm = np.clip(np.rint(pcm), -32768, 32767).astype(np.int16)
    rates = [80000, 96000, 128000, 192000, 256000, 64000]
    br = [rates[i % len(rates)] for i in range(B)]
    b = amd.Batch(B, 48000, 1, 10.0, 0, br, device=0)
    got = np.concatenate([b.encode(pcm[:, :9]), b.encode(pcm[:, 9:20]), b.encode(pcm[:, 20:])], axis=1)
    want = _oracle_batch(pcm, 48000, 10.0, 0, br, b.stride)
    nb = np.array([b.num_bytes(i) for i in range(B)])
    bad = [(i, t) for i in range(B) for t in range(T) if (got[i, t, :nb[i]] != want[i, t, :nb[i]]).any()]
    assert not bad, (len(bad), bad[:8])
    # attack handling on -> off -> on across bitrate switches, 6-frame launches
    o = Oracle(48000, 1, 10.0, 0, 96000, portable_math=True)
    b = amd.Batch(1, 48000, 1, 10.0, 0, [96000], device=0)
    sched = {0: 96000, 12: 32000, 24: 128000}
    for t0 in range(0, T, 6):
        if t0 in sched:
            assert b.set_bitrate(0, sched[t0]) == 0 and o.set_bitrate(sched[t0]) == 0
        g = b.encode(pcm[3:4, t0:t0 + 6])
        for k in range(6):
            assert (g[0, k, :o.nbytes] == o.encode(pcm[3, t0 + k][None])).all(), (t0, k)


@pytest.mark.parametrize("fs,ms,hr,N,br", [(48000, 10.0, 0, 480, 64000), (48000, 10.0, 0, 480, 128000), (32000, 10.0, 0, 320, 96000),
                                          (16000, 5.0, 0, 80, 32000), (96000, 2.5, 1, 240, 256000), (48000, 2.5, 0, 120, 160000)])
def test_stage_traces_match_oracle(fs, ms, hr, N, br):
    """Stage by stage, not only the final bytes: every intermediate the traced entry point records per channel-frame (MDCT spectrum,
    12.8 kHz signal, pitch lag and correlation, LTPF parameters, attack flag, band energies, bandwidth, scale factors and their
    indices, shaped and TNS-filtered spectra, TNS orders / indices / bits, both gains, bit counts, last non-zero line, quantised lines,
    noise level, residual bit count, side-information cursor) equals the oracle's trace of the same frame exactly (floats bit for bit)."""
    import ctypes as C
    from lc3_harness import Trace
    amd = _amd()
    B, T = 6, 8
    pcm = synth_pcm(B, T, N, fs, seed=31)
    bt = amd.Batch(B, fs, 1, ms, hr, [br] * B, device=0)
    got, traces = bt.encode_traced(pcm)
    nb = bt.num_bytes(0)
    bad = []
    for b in range(B):
        o = Oracle(fs, 1, ms, hr, br, portable_math=True)
        tr = o.enable_trace()
        for t in range(T):
            want = o.encode(pcm[b, t][None])
            assert (got[b, t, :nb] == want).all(), (b, t)
            g = Trace.from_buffer_copy(traces[b * T + t].tobytes()[:C.sizeof(Trace)])
            for f, _ in Trace._fields_:
                ga, ca = getattr(g, f), getattr(tr[0], f)
                if hasattr(ga, "__len__"):
                    n = N if len(ga) == 960 else len(ga)
                    a, c = np.ctypeslib.as_array(ga)[:n], np.ctypeslib.as_array(ca)[:n]
                else:
                    a, c = np.asarray([ga]), np.asarray([ca])
                same = (a.view(np.uint32) == c.view(np.uint32)) if a.dtype.kind == "f" else (a == c)
                if a.dtype.kind == "f": same = same | ((a == 0) & (c == 0))          # +0 / -0
                if not same.all(): bad.append((b, t, f, int((~same).sum())))
    assert not bad, bad[:10]


def test_checkpoint_resume_state():
    """lc3plus_enc_batch_get_state / set_state: a second batch of the same configuration given the first one's state continues the
    streams byte for byte (MDCT / resampler memory, pitch and LTPF histories, rate-control and attack-detector words); mixed bitrates
    with attack handling, the first part through the pipelined path, the rest through short calls."""
    amd = _amd()
    B, T = 96, 30
    rates = [64000, 96000, 128000, 32000, 256000, 16000]
    br = [rates[i % len(rates)] for i in range(B)]
    pcm = synth_pcm(B, T, 480, 48000, seed=909)
    a = amd.Batch(B, 48000, 1, 10.0, 0, br, device=0)
    g1 = a.encode(pcm[:, :13])
    st = a.get_state()
    assert st.size == a.lib.lc3plus_enc_batch_state_size(a.h) and st.size > 0
    b = amd.Batch(B, 48000, 1, 10.0, 0, br, device=0)
    b.set_state(st)
    g2 = np.concatenate([b.encode(pcm[:, 13:20]), b.encode(pcm[:, 20:])], axis=1)
    want = _oracle_batch(pcm, 48000, 10.0, 0, br, a.stride)
    got = np.concatenate([g1, g2], axis=1)
    nb = [a.num_bytes(i) for i in range(B)]
    bad = [(i, t) for i in range(B) for t in range(T) if (got[i, t, :nb[i]] != want[i, t, :nb[i]]).any()]
    assert not bad, (len(bad), bad[:8])
    with pytest.raises(Exception):
        b.set_state(st[:-4])                                   # a state of another size is refused


class _Dev:
    """Device buffers through ctypes (hipMalloc / hipMemcpy): the tests do not depend on torch."""
    def __init__(self):
        import ctypes as C
        self.C = C; self.hip = C.CDLL("libamdhip64.so"); self.ptrs = []
    def put(self, arr):
        C = self.C; arr = np.ascontiguousarray(arr); p = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(p), C.c_size_t(arr.nbytes)) == 0
        assert self.hip.hipMemcpy(p, C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes), C.c_int(1)) == 0
        self.ptrs.append(p); return p.value
    def zeros(self, nbytes):
        C = self.C; p = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0 and self.hip.hipMemset(p, 0, C.c_size_t(nbytes)) == 0
        self.ptrs.append(p); return p.value
    def get(self, ptr, shape, dtype):
        C = self.C; out = np.zeros(shape, dtype)
        assert self.hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(out.nbytes), C.c_int(2)) == 0
        return out
    def sync(self):
        assert self.hip.hipDeviceSynchronize() == 0
    def free(self):
        for p in self.ptrs: self.hip.hipFree(p)
        self.ptrs = []


@pytest.mark.parametrize("fs,ms,hr,N,B,T,K,rates", [
    (48000, 10.0, 0, 480, 1024, 16, 6, [64000]),
    (48000, 10.0, 0, 480, 512, 12, 5, [64000, 96000, 128000, 32000, 256000]),       # attack handling from 100 bytes: the detector's kernel and state
    (48000, 10.0, 0, 480, 4096, 24, 3, [64000]),
    (32000, 10.0, 0, 320, 256, 16, 4, [96000, 64000, 128000]),                        # attack handling from 81 bytes
    (16000, 5.0, 0, 80, 256, 20, 4, [32000, 64000]),
    (96000, 10.0, 1, 960, 64, 12, 4, [256000, 400000]),                               # the large-layout kernels
    (48000, 2.5, 0, 120, 256, 24, 4, [128000, 64000]),
    (48000, 10.0, 0, 480, 512, 6, 6, [64000, 128000]),                                # short calls: pipelined from 4 frames under the promise
    (48000, 10.0, 0, 480, 512, 5, 4, [64000]),
    (48000, 10.0, 0, 480, 256, 4, 5, [64000, 96000]),
    (48000, 10.0, 0, 480, 256, 40, 3, [64000]),                                       # the longest calls that overlap: five runs of 8 frames
])
def test_consecutive_calls_overlap_with_input_ready(fs, ms, hr, N, B, T, K, rates):
    """lc3plus_enc_batch_set_input_ready: K device-pointer calls queued back to back without a host synchronisation in between - the side
    kernels of call k+1 run beside the sequential tail and the bitstream writer of call k, wait per run for the previous call's rate
    kernel, and take the MDCT memory from the previous call's hand-over - give the bytes of one continuous encode (oracle on a sample
    of streams for the large case).  Then: a call of another length (takes the ordered path), a bitrate switch between overlapped
    calls, and the promise withdrawn again."""
    amd = _amd()
    d = _Dev()
    try:
        TT = T * K + 7 + 2 * T
        pcm = synth_pcm(B, TT, N, fs, seed=2024 + B)
        br = [rates[i % len(rates)] for i in range(B)]
        b = amd.Batch(B, fs, 1, ms, hr, br, device=0)
        stride = b.stride
        b.set_input_ready(True)
        cuts = [T] * K + [7] + [T, T]
        ins, outs, t0 = [], [], 0
        for n in cuts:
            ins.append(d.put(pcm[:, t0:t0 + n])); outs.append(d.zeros(B * n * stride)); t0 += n
        d.sync()                                          # the promise: all PCM is on the device before the first call
        for k in range(K + 1):
            b.encode_device(ins[k], 16, cuts[k], outs[k], stride, hip_stream=None, sync=False)
        b.encode_device(ins[K + 1], 16, T, outs[K + 1], stride, hip_stream=None, sync=False)      # same length as before the odd call: ordered, then
        sw = next(i for i in range(B // 2, B) if len(rates) == 1 or br[i] != min(rates))
        new_br = min(rates) if len(rates) > 1 else 48000
        assert b.set_bitrate(sw, new_br) == 0                                                       # (drains the stream)
        b.set_input_ready(False)
        b.encode_device(ins[K + 2], 16, T, outs[K + 2], stride, hip_stream=None, sync=True)
        got = np.concatenate([d.get(outs[k], (B, cuts[k], stride), np.uint8) for k in range(len(cuts))], axis=1)
        assert not b.last_status(T).any()
        pick = list(range(B)) if B <= 1024 else sorted(set([0, 1, B - 1, sw] + [int(v) for v in np.random.default_rng(3).choice(B, 60, replace=False)]))
        pick = [i for i in pick if i != sw]
        want = _oracle_batch(pcm[pick], fs, ms, hr, [br[i] for i in pick], stride)
        nb = [b.num_bytes(i) for i in pick]
        bad = [(i, t) for k, i in enumerate(pick) for t in range(TT) if (got[i, t, :nb[k]] != want[k, t, :nb[k]]).any()]
        assert not bad, (len(bad), bad[:8])
        # the switched stream: continuous up to the switch, then the oracle with the same switch
        o = Oracle(fs, 1, ms, hr, br[sw], portable_math=True)
        for t in range(TT):
            if t == TT - T: assert o.set_bitrate(new_br) == 0
            w = o.encode(pcm[sw, t][None])
            assert (got[sw, t, :len(w)] == w).all(), t
    finally:
        d.free()


def test_consecutive_stereo_calls_overlap_with_input_ready():
    """The same for stereo pairs (BASELINE configs[2]'s shape: two channel-streams per stream, one frame of both per payload), calls of
    16 frames queued back to back under the promise, odd byte split (61.6 kbps: 77 = 39 + 38 bytes) among the rates."""
    amd = _amd()
    d = _Dev()
    try:
        B, T, K = 384, 16, 5
        rates = [128000, 61600, 96000]
        rng = np.random.default_rng(77)
        mono = synth_pcm(2 * B, T * K, 480, 48000, seed=515)
        pcm = np.ascontiguousarray(mono.reshape(B, 2, T * K, 480).transpose(0, 2, 1, 3))       # [pair, T, ch, N]
        br = [rates[i % len(rates)] for i in range(B)]
        b = amd.Batch(B, 48000, 2, 10.0, 0, br, device=0)
        stride = b.stride
        b.set_input_ready(True)
        ins = [d.put(pcm[:, k * T:(k + 1) * T]) for k in range(K)]
        outs = [d.zeros(B * T * stride) for _ in range(K)]
        d.sync()
        for k in range(K): b.encode_device(ins[k], 16, T, outs[k], stride, hip_stream=None, sync=False)
        got = np.concatenate([d.get(outs[k], (B, T, stride), np.uint8) for k in range(K)], axis=1)
        assert not b.last_status(T).any()
        want = _oracle_batch_ch(pcm, 48000, 10.0, 0, 2, br, stride)
        nb = [b.num_bytes(i) for i in range(B)]
        bad = [(i, t) for i in range(B) for t in range(T * K) if (got[i, t, :nb[i]] != want[i, t, :nb[i]]).any()]
        assert not bad, (len(bad), bad[:8])
    finally:
        d.free()


def test_full_size_baseline_batch_properties():
    """BASELINE configs[1] at its full size (4096 mono streams x 64 frames, 48 kHz / 10 ms / 64 kbps) through size-independent properties:
    (1) streams are independent - 512 distinct streams tiled 8 times in a shuffled order give 8 identical copies of every output;
    (2) a call of 64 frames equals calls of 24 + 40 frames (state crosses calls); (3) 48 streams picked at random equal the CPU oracle
    byte for byte; (4) GPU decode of the GPU bitstreams gives back the input (delay-compensated SNR); (5) no status bit is raised."""
    amd = _amd()
    U, REP, T, N = 512, 8, 64, 480
    B = U * REP
    uniq = synth_pcm(U, T, N, 48000, seed=4096)
    rng = np.random.default_rng(64)
    order = np.concatenate([rng.permutation(U) for _ in range(REP)])
    pcm = np.ascontiguousarray(uniq[order])
    b = amd.Batch(B, 48000, 1, 10.0, 0, [64000] * B, device=0)
    got = b.encode(pcm)
    assert not b.last_status(T).any()
    first = {}
    for i, u in enumerate(order):
        if u in first: assert (got[i] == got[first[u]]).all(), (i, u)
        else: first[int(u)] = i
    b2 = amd.Batch(B, 48000, 1, 10.0, 0, [64000] * B, device=0)
    two = np.concatenate([b2.encode(pcm[:, :24]), b2.encode(pcm[:, 24:])], axis=1)
    assert (two == got).all()
    pick = sorted(int(v) for v in rng.choice(U, size=48, replace=False))
    want = _oracle_batch(uniq[pick], 48000, 10.0, 0, [64000] * len(pick), 80)
    for k, u in enumerate(pick):
        assert (got[first[u]] == want[k]).all(), u
    dec = amd.DecBatch(B, 48000, 1, 10.0, 0, [80] * B, device=0)
    out, status = dec.decode(got)
    assert status.sum() == 0
    x = pcm[:U * 2].reshape(U * 2, -1).astype(np.float64); y = out[:U * 2].reshape(U * 2, -1).astype(np.float64)
    lag = 120                                            # lc3_enc_get_delay + lc3_dec_get_delay at 48 kHz / 10 ms (R/lc3.c)
    e = ((x[:, N:-N - lag] - y[:, N + lag:y.shape[1] - N]) ** 2).sum(axis=1); s = (x[:, N:-N - lag] ** 2).sum(axis=1)
    snr = 10 * np.log10(np.maximum(s, 1e-9) / np.maximum(e, 1e-9))
    best = snr
    for lag in (240, 300, 480):
        e = ((x[:, N:-N - lag] - y[:, N + lag:y.shape[1] - N]) ** 2).sum(axis=1); s = (x[:, N:-N - lag] ** 2).sum(axis=1)
        c = 10 * np.log10(np.maximum(s, 1e-9) / np.maximum(e, 1e-9))
        best = c if c.mean() > best.mean() else best
    assert np.median(best) > 12.0, np.median(best)


@pytest.mark.parametrize("fs,ms,hr,N,br", [(48000, 10.0, 0, 480, 64000), (48000, 10.0, 0, 480, 24000), (32000, 5.0, 0, 160, 64000), (16000, 2.5, 0, 40, 96000),
                                           (96000, 2.5, 1, 240, 256000), (96000, 10.0, 1, 960, 256000)])
def test_pipelined_path_records_match_oracle(fs, ms, hr, N, br):
    """Stage by stage on the PRODUCT path (the pipelined kernels, not the traced whole-encoder kernel): the per-frame records the kernels hand to each other
    (lc3plus_enc_batch_last_records) against the oracle's trace of the same frames - scale factors (lc3_enc_scf_lane_kernel), quantised scale factors and SNS
    indices (lc3_enc_snsvq_kernel), LTPF words (lc3_enc_pitch_kernel), bandwidth, TNS filters / orders / bits / coefficient indices, gain floor
    (lc3_enc_shape_lane_kernel), gain index, gain and bit count of the first quantisation (lc3_enc_rate_kernel); floats bit for bit."""
    amd = _amd()
    FR = dict(SCF=0, SCFQ=16, IDX=32, LTPF=48, TNS=52, GGMIN=72, XZERO=73, BWC=74, RATE=76)
    B, T = 5, 12                                               # 12 frames: the pipelined path
    pcm = synth_pcm(B, T, N, fs, seed=41)
    bt = amd.Batch(B, fs, 1, ms, hr, [br] * B, device=0)
    got = bt.encode(pcm)
    rec = bt.last_records(T)
    ri = rec.view(np.int32)
    nb = bt.num_bytes(0)
    bad = []
    for b in range(B):
        o = Oracle(fs, 1, ms, hr, br, portable_math=True)
        tr = o.enable_trace()
        for t in range(T):
            want = o.encode(pcm[b, t][None])
            assert (got[b, t, :nb] == want).all(), (b, t)
            w = tr[0]
            def chk(name, a, c):
                a, c = np.asarray(a), np.asarray(c)
                same = (a.view(np.uint32) == c.view(np.uint32)) | ((a == 0) & (c == 0)) if a.dtype.kind == "f" else (a == c)
                if not np.all(same): bad.append((b, t, name))
            if not w.attack:                                   # with the attack flag the record keeps the unsmoothed factors (the quantiser's kernel smooths them)
                chk("scf", rec[b, t, FR["SCF"]:FR["SCF"] + 16], np.ctypeslib.as_array(w.scf).astype(np.float32))
            chk("scf_q", rec[b, t, FR["SCFQ"]:FR["SCFQ"] + 16], np.ctypeslib.as_array(w.scf_q).astype(np.float32))
            chk("scf_idx", ri[b, t, FR["IDX"]:FR["IDX"] + 7], np.ctypeslib.as_array(w.scf_idx))
            chk("ltpf", ri[b, t, FR["LTPF"]:FR["LTPF"] + 4], np.array([w.ltpf_param[0], w.ltpf_param[1], w.ltpf_param[2], w.ltpf_bits]))
            chk("bw", ri[b, t, FR["BWC"]:FR["BWC"] + 1], np.array([w.bw_idx]))
            chk("tns", ri[b, t, FR["TNS"]:FR["TNS"] + 4], np.array([w.tns_nfilt, w.tns_order[0], w.tns_order[1], w.tns_bits]))
            chk("tns_idx", ri[b, t, FR["TNS"] + 4:FR["TNS"] + 20], np.ctypeslib.as_array(w.tns_rc_idx))
            chk("gg_min", np.array([int(rec[b, t, FR["GGMIN"]])]), np.array([w.gg_min]))
            chk("gg_idx0", ri[b, t, FR["RATE"]:FR["RATE"] + 1], np.array([w.gg_idx0]))
            chk("gain0", rec[b, t, FR["RATE"] + 1:FR["RATE"] + 2], np.array([w.gain0], np.float32))
            chk("nbits0", ri[b, t, FR["RATE"] + 2:FR["RATE"] + 3], np.array([w.nbits0]))
    assert not bad, bad[:12]
    bt.close()


@pytest.mark.parametrize("B,T,off", [(33, 13, 0), (7, 22, 2), (65, 9, 6)])
def test_odd_stream_counts_and_unaligned_pcm_on_the_pipelined_path(B, T, off):
    """The four-frames-per-wave front kernel and the two-streams-per-wave pitch kernel at their edges: an odd number of channel-streams (the
    last pitch wave's second half shadows the last stream), frame counts that are not multiples of four (a wave's last run is short), and a
    PCM pointer that is not 16-byte aligned (the front and resampler kernels fall back from 128-bit loads), two calls so that state runs on."""
    amd = _amd()
    d = _Dev()
    try:
        rates = [64000, 128000, 32000]
        br = [rates[i % 3] for i in range(B)]
        pcm = synth_pcm(B, 2 * T, 480, 48000, seed=4242 + B)
        b = amd.Batch(B, 48000, 1, 10.0, 0, br, device=0)
        stride = b.stride
        outs = []
        for h in range(2):
            raw = np.zeros(off + B * T * 480 * 2 + 16, np.uint8)
            raw[off:off + B * T * 480 * 2] = np.ascontiguousarray(pcm[:, h * T:(h + 1) * T]).view(np.uint8).reshape(-1)
            pin = d.put(raw); pout = d.zeros(B * T * stride)
            b.encode_device(pin + off, 16, T, pout, stride, hip_stream=None, sync=True)
            outs.append(d.get(pout, (B, T, stride), np.uint8))
        got = np.concatenate(outs, axis=1)
        want = _oracle_batch(pcm, 48000, 10.0, 0, br, stride)
        nb = [b.num_bytes(i) for i in range(B)]
        bad = [(i, t) for i in range(B) for t in range(2 * T) if (got[i, t, :nb[i]] != want[i, t, :nb[i]]).any()]
        assert not bad, (len(bad), bad[:8])
    finally:
        d.free()


@pytest.mark.parametrize("level", [0, 1, 2])
def test_resampler_for_96_khz_at_every_switch_level(level):
    """lc3_enc_resample96_kernel_n240 / _n480 / _n960 (four outputs per lane, 1 920 samples = 8 / 4 / 2 frames per step): LC3PLUS_ENC_RESAMPLE96 = 0 (never),
    1 (the default: 2.5 ms frames only) and 2 (every frame length) give the oracle's bytes at all three frame lengths, mono and stereo, with call lengths
    that leave a step and a workgroup partly filled (11 + 8 + 1 frames: 32 / 16 / 8 frames per workgroup), state running on from call to call."""
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import audio_codec_amd
        from lc3_harness import synth_pcm
        from test_gpu_parity import _oracle_batch_ch
        for ms, N, ch, rates in ((2.5, 240, 1, [256000, 320000, 672000]), (2.5, 240, 2, [512000, 640000]), (5.0, 480, 1, [256000, 400000]), (10.0, 960, 1, [149600, 256000]), (10.0, 960, 2, [512000])):
            B, T, fs = 3 * len(rates), 20, 96000
            br = [rates[i %% len(rates)] for i in range(B)]
            pcm = synth_pcm(B * ch, T, N, fs, seed=77 + N + ch).reshape(B, ch, T, N).transpose(0, 2, 1, 3).copy()      # [B, T, ch, N]
            b = audio_codec_amd.Batch(B, fs, ch, ms, 1, br, device=0)
            x = pcm if ch > 1 else pcm[:, :, 0]
            got = np.concatenate([b.encode(x[:, :11]), b.encode(x[:, 11:19]), b.encode(x[:, 19:])], axis=1)
            want = _oracle_batch_ch(pcm, fs, ms, 1, ch, br, b.stride)
            nb = [b.num_bytes(i) for i in range(B)]
            bad = [(i, t) for i in range(B) for t in range(T) if (got[i, t, :nb[i]] != want[i, t, :nb[i]]).any()]
            assert not bad, (ms, N, ch, bad[:6])
        print("ok")
    """ % (root, os.path.join(root, "tests")))
    env = dict(os.environ, LC3PLUS_ENC_RESAMPLE96=str(level))
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("env", ["LC3PLUS_ENC_FRONT4=0", "LC3PLUS_ENC_PITCH2=0", "LC3PLUS_ENC_RATE_STREAM=1", "LC3PLUS_ENC_PACK_WPG=1", "LC3PLUS_DEC_IMDCT4=0",
                                 # round 4's switches: each alternative is byte-identical to the default
                                 "LC3PLUS_ENC_RESAMPLE48=0", "LC3PLUS_ENC_PACK_W5=1", "LC3PLUS_ENC_PACK_SPLIT=1", "LC3PLUS_ENC_FUSE_VQ=1", "LC3PLUS_ENC_RATE_ON=0",
                                 "LC3PLUS_ENC_RATE_ON=1", "LC3PLUS_ENC_SHAPE_ON_PITCH=1", "LC3PLUS_ENC_PACK_STREAM=1", "LC3PLUS_DEC_PLC_STREAM=0"])
def test_diagnostic_switches_give_the_same_bytes(env):
    """The kernels the defaults replaced (one frame at a time front / IMDCT, one stream per wave pitch chain) and the stream / workgroup switches
    still produce the oracle's bytes: a child process per switch (the library reads them once)."""
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import audio_codec_amd
        from lc3_harness import synth_pcm, oracle_decode_streams
        from test_gpu_parity import _oracle_batch
        B, T, N, fs = 96, 22, 480, 48000
        br = [[64000, 128000, 32000][i %% 3] for i in range(B)]
        pcm = synth_pcm(B, T, N, fs, seed=99)
        b = audio_codec_amd.Batch(B, fs, 1, 10.0, 0, br, device=0)
        got = np.concatenate([b.encode(pcm[:, :9]), b.encode(pcm[:, 9:])], axis=1)
        want = _oracle_batch(pcm, fs, 10.0, 0, br, b.stride)
        nb = [b.num_bytes(i) for i in range(B)]
        bad = [(i, t) for i in range(B) for t in range(T) if (got[i, t, :nb[i]] != want[i, t, :nb[i]]).any()]
        assert not bad, bad[:6]
        d = audio_codec_amd.DecBatch(B, fs, 1, 10.0, 0, nb, device=0)
        out, _ = d.decode(got[:, :, :max(nb)].copy())
        ref, _ = oracle_decode_streams(got[:, :, :max(nb)].copy(), nb, None, fs, 10.0, 0, 1)
        assert (out == ref).all()
        # ... and three overlapped device-pointer calls of 16 frames under the input-ready promise (the stream / writer switches only act there)
        from test_gpu_parity import _Dev
        dv = _Dev()
        T2, K = 16, 3
        pcm2 = synth_pcm(B, T2 * K, N, fs, seed=100)
        b2 = audio_codec_amd.Batch(B, fs, 1, 10.0, 0, br, device=0)
        b2.set_input_ready(True)
        ins = [dv.put(pcm2[:, k * T2:(k + 1) * T2]) for k in range(K)]
        outs = [dv.zeros(B * T2 * b2.stride) for _ in range(K)]
        dv.sync()
        for k in range(K): b2.encode_device(ins[k], 16, T2, outs[k], b2.stride, hip_stream=None, sync=False)
        dv.sync()
        got2 = np.concatenate([dv.get(outs[k], (B, T2, b2.stride), np.uint8) for k in range(K)], axis=1)
        want2 = _oracle_batch(pcm2, fs, 10.0, 0, br, b2.stride)
        bad = [(i, t) for i in range(B) for t in range(T2 * K) if (got2[i, t, :nb[i]] != want2[i, t, :nb[i]]).any()]
        assert not bad, bad[:6]
        dv.free()
        print("ok")
    """ % (root, os.path.join(root, "tests")))
    k, v = env.split("=")
    e = dict(os.environ); e[k] = v
    r = subprocess.run([sys.executable, "-c", code], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (env, r.stdout[-300:], r.stderr[-800:])


@pytest.mark.parametrize("fs,ms,N,bws,ready", [
    (48000, 10.0, 480, [0, 4000, 8000, 16000, 12000, 20000], False),
    (48000, 10.0, 480, [0, 4000, 8000, 16000, 12000, 20000], True),
    (32000, 5.0, 160, [0, 4000, 8000, 16000, 12000], False),
    (32000, 5.0, 160, [0, 4000, 8000, 16000, 12000], True),
    (24000, 10.0, 240, [0, 4000, 8000, 12000], True),
])
def test_bandwidth_controller_on_the_pipelined_path(fs, ms, N, bws, ready):
    """lc3plus_enc_batch_set_bandwidth on the PRODUCT path: calls of 14 frames take the pipelined kernels, so the bandwidth
    controller of lc3_enc_shape_lane_kernel (R/enc_lc3_fl.c:92-96, R/cutoff_bandwidth.c:13-26: fade 0.5 / 0.25 / 0.125 / 0.0625, then zeros) and the
    limit on the detected bandwidth (R/enc_lc3_fl.c:68 -> R/detect_cutoff_warped.c) run one frame per lane.  Bandwidths mixed per stream (0 = none set)
    inside one wave, a bandwidth switch between two calls - queued back to back under the input-ready promise when `ready` - against the oracle
    with the same switches, and the records' bandwidth word (FR_BWC) against the oracle's trace.  The conformance procedure's band_limiting and
    bandwidth_switching cases: E/conformance/lc3_conformance.py:835-862."""
    amd = _amd()
    d = _Dev()
    try:
        B, T, K = 150, 14, 4
        rates = [64000, 32000, 96000, 128000]
        br = [rates[i % len(rates)] for i in range(B)]
        bw0 = [bws[i % len(bws)] for i in range(B)]
        bw1 = [bws[(i // 7 + 1) % len(bws)] if i % 3 == 0 else bw0[i] for i in range(B)]          # a third of the streams switch after call 1
        pcm = synth_pcm(B, T * K, N, fs, seed=808 + N)
        b = amd.Batch(B, fs, 1, ms, 0, br, device=0)
        stride = b.stride
        for i in range(B):
            if bw0[i]: assert b.set_bandwidth(i, bw0[i]) == 0, (i, bw0[i])
        assert b.set_bandwidth(0, fs) == 18                                                         # LC3_BW_WARNING (R/lc3.c:197): above the limit, nothing changes
        b.set_input_ready(ready)
        ins = [d.put(pcm[:, k * T:(k + 1) * T]) for k in range(K)]
        outs = [d.zeros(B * T * stride) for _ in range(K)]
        d.sync()
        recs = []
        for k in range(K):
            if k == 2:
                for i in range(B):
                    if bw1[i] != bw0[i] and bw1[i]: assert b.set_bandwidth(i, bw1[i]) == 0
            b.encode_device(ins[k], 16, T, outs[k], stride, hip_stream=None, sync=not ready)
            if not ready: recs.append(b.last_records(T).view(np.int32)[:, :, 74].copy())
        d.sync()
        got = np.concatenate([d.get(outs[k], (B, T, stride), np.uint8) for k in range(K)], axis=1)
        assert not b.last_status(T).any()
        bad, badbw = [], []
        for i in range(B):
            o = Oracle(fs, 1, ms, 0, br[i], portable_math=True, bandwidth=bw0[i])
            tr = o.enable_trace()
            for t in range(T * K):
                if t == 2 * T and bw1[i] != bw0[i] and bw1[i]: assert o.set_bandwidth(bw1[i]) == 0
                w = o.encode(pcm[i, t][None])
                if (got[i, t, :len(w)] != w).any(): bad.append((i, t))
                if recs and recs[t // T][i, t % T] != tr[0].bw_idx: badbw.append((i, t, int(recs[t // T][i, t % T]), tr[0].bw_idx))
        assert not bad, (len(bad), bad[:8])
        assert not badbw, badbw[:8]
        # the limited streams really are band limited: a 4 kHz stream's frames differ from the unlimited encode of the same PCM
        o = Oracle(fs, 1, ms, 0, br[1], portable_math=True)
        assert any((got[1, t, :o.nbytes] != o.encode(pcm[1, t][None])).any() for t in range(T))
    finally:
        d.free()


def test_rate_chain_stream_choice_switches_between_overlapped_calls():
    """where the rate chain runs is chosen per call (its own stream for calls of up to 32 frames or a mean frame size of 120 bytes and more, the
    caller's stream otherwise), so consecutive overlapped calls can alternate between the two: calls of 32 and 33 frames queued back to back under the
    promise (33, 33, 32, 32, 33, 32: same-length neighbours overlap, the others take the ordered path; the chain crosses streams four times), and a batch
    whose mean frame size sits at the threshold (80- and 160-byte streams: mean 120).  Bytes of one continuous encode."""
    amd = _amd()
    d = _Dev()
    try:
        for rates, cuts in (([64000], [33, 33, 32, 32, 33, 32]), ([64000, 128000], [12, 12, 40, 40, 12])):
            B = 640
            TT = sum(cuts)
            pcm = synth_pcm(B, TT, 480, 48000, seed=3233 + len(rates))
            br = [rates[i % len(rates)] for i in range(B)]
            b = amd.Batch(B, 48000, 1, 10.0, 0, br, device=0)
            stride = b.stride
            b.set_input_ready(True)
            ins, outs, t0 = [], [], 0
            for n in cuts:
                ins.append(d.put(pcm[:, t0:t0 + n])); outs.append(d.zeros(B * n * stride)); t0 += n
            d.sync()
            for k, n in enumerate(cuts): b.encode_device(ins[k], 16, n, outs[k], stride, hip_stream=None, sync=False)
            d.sync()
            got = np.concatenate([d.get(outs[k], (B, n, stride), np.uint8) for k, n in enumerate(cuts)], axis=1)
            pick = list(range(0, B, 5))
            want = _oracle_batch(pcm[pick], 48000, 10.0, 0, [br[i] for i in pick], stride)
            nb = [b.num_bytes(i) for i in pick]
            bad = [(i, t) for k, i in enumerate(pick) for t in range(TT) if (got[i, t, :nb[k]] != want[k, t, :nb[k]]).any()]
            assert not bad, (rates, len(bad), bad[:8])
            b.close()
    finally:
        d.free()


def test_check_ready_debug_aid(tmp_path):
    """LC3PLUS_CHECK_READY=1: with the input-ready promise in force, a device-pointer call made while work of the CALLER is still pending on the stream of the
    call (here: a large copy queued just before on the same stream) is refused with LC3_ERROR; without pending work the call goes through."""
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, ctypes as C, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import audio_codec_amd
        from lc3_harness import synth_pcm
        hip = C.CDLL("libamdhip64.so")
        B, T = 64, 12
        pcm = synth_pcm(B, T, 480, 48000, seed=5)
        b = audio_codec_amd.Batch(B, 48000, 1, 10.0, 0, [64000] * B, device=0)
        st = C.c_void_p(); assert hip.hipStreamCreate(C.byref(st)) == 0
        pin = C.c_void_p(); pout = C.c_void_p(); big = C.c_void_p(); big2 = C.c_void_p()
        assert hip.hipMalloc(C.byref(pin), C.c_size_t(pcm.nbytes)) == 0 and hip.hipMalloc(C.byref(pout), C.c_size_t(B * T * b.stride)) == 0
        assert hip.hipMemcpy(pin, C.c_void_p(pcm.ctypes.data), C.c_size_t(pcm.nbytes), 1) == 0
        n = 1 << 30
        assert hip.hipMalloc(C.byref(big), C.c_size_t(n)) == 0 and hip.hipMalloc(C.byref(big2), C.c_size_t(n)) == 0
        b.set_input_ready(True)
        b.encode_device(pin.value, 16, T, pout.value, b.stride, hip_stream=st.value, sync=True)          # nothing pending: accepted
        for _ in range(8): assert hip.hipMemcpyAsync(big2, big, C.c_size_t(n), 3, st) == 0             # the caller's own work on the stream of the call
        try:
            b.encode_device(pin.value, 16, T, pout.value, b.stride, hip_stream=st.value, sync=True)
            print("accepted")
        except audio_codec_amd.LC3Error:
            print("refused")
        assert hip.hipDeviceSynchronize() == 0
    """ % (root, os.path.join(root, "tests")))
    e = dict(os.environ); e["LC3PLUS_CHECK_READY"] = "1"
    r = subprocess.run([sys.executable, "-c", code], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0 and "refused" in r.stdout and "LC3PLUS_CHECK_READY" in r.stderr, (r.stdout[-300:], r.stderr[-800:])


def test_wave_per_frame_tail_writer_switch(tmp_path):
    """LC3PLUS_ENC_TAILW_BYTES=N: channel-streams with frames of N bytes and more get lc3_enc_tailw_kernel (tail of the encoder + wave-parallel range coder, a frame
    per wave) instead of a lane of lc3_enc_pack_kernel; the two kernels split a mixed batch.  Same bytes as the oracle, and the status bytes stay clear."""
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import audio_codec_amd
        from lc3_harness import synth_pcm
        from test_gpu_parity import _oracle_batch, RATES
        for fs, ms, hr, N, rates in ((48000, 10.0, 0, 480, RATES), (96000, 10.0, 1, 960, [256000, 149600, 400000]), (32000, 5.0, 0, 160, [64000, 192000, 320000])):
            B, T = 96, 14
            br = [rates[i %% len(rates)] for i in range(B)]
            pcm = synth_pcm(B, T, N, fs, seed=1717)
            b = audio_codec_amd.Batch(B, fs, 1, ms, hr, br, device=0)
            got = np.concatenate([b.encode(pcm[:, :T]), b.encode(pcm[:, :T])], axis=1)
            assert not b.last_status(T).any()
            want = _oracle_batch(np.concatenate([pcm, pcm], axis=1), fs, ms, hr, br, b.stride)
            nb = [b.num_bytes(i) for i in range(B)]
            bad = [(i, t) for i in range(B) for t in range(2 * T) if (got[i, t, :nb[i]] != want[i, t, :nb[i]]).any()]
            assert not bad, (fs, ms, bad[:6])
        print("ok")
    """ % (root, os.path.join(root, "tests")))
    e = dict(os.environ); e["LC3PLUS_ENC_TAILW_BYTES"] = "100"
    r = subprocess.run([sys.executable, "-c", code], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-300:], r.stderr[-800:])


def test_device_fastmath_equals_host(tmp_path):
    """lc3_fastmath.h gives the same bits on the device as on the host (which tools/fastmath_check.c pins to glibc for every float argument): half a million
    arguments per function - every binade, the neighbourhood of 1 and of the powers of two, subnormals, the special arguments - through the library's test hook."""
    import ctypes as C
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)))
    import importlib.util
    spec = importlib.util.spec_from_file_location("test_fastmath", os.path.join(sys_path, "test_fastmath.py"))
    tf = importlib.util.module_from_spec(spec); spec.loader.exec_module(tf)
    H = tf.host_lib(tmp_path)
    L = _amd().load_library()
    L.lc3hip_test_fastmath.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_longlong]
    logs, ex = tf.sample_arguments(seed=11)
    sp = np.array([0.0, -0.0, -1.0, np.inf, -np.inf, np.nan, 1e30, -1e30, 2000.0, -2000.0], np.float32)
    for kind, x in ((0, np.concatenate([logs, sp])), (1, np.concatenate([logs, sp])), (2, np.concatenate([ex, sp]))):
        x = np.ascontiguousarray(x)
        dev = np.zeros_like(x); host = np.zeros_like(x)
        assert L.lc3hip_test_fastmath(kind, x.ctypes.data, dev.ctypes.data, x.size) == 0
        H.lc3m_host_eval(kind, x.ctypes.data, host.ctypes.data, x.size)
        same = (dev.view(np.uint32) == host.view(np.uint32)) | (np.isnan(dev) & np.isnan(host))
        assert same.all(), (kind, x[~same][:6], dev[~same][:6], host[~same][:6])
