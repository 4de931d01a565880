"""GPU parity tests of the float DECODER (run with -m gpu on an MI355X): lc3_decode_kernel called through the C ABI
(lc3plus_dec_batch_* / lc3_dec_*) against the CPU oracle decoder (oracle/lc3_oracle_dec.inc, itself pinned
sample-exact to the unmodified ETSI decoder in tests/test_oracle_vs_ref.py) and against committed decoder outputs of
the ETSI reference (tests/golden/d*.npz).

Bar: the output PCM is integer data and must be identical sample for sample, for good, lost (bfi = 1) and corrupt
frames alike, and the per-frame concealment status must agree."""
import os
import numpy as np
import pytest

from lc3_harness import make_dec_case, oracle_decode_streams, Oracle, OracleDecoder, synth_pcm

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _amd():
    import audio_codec_amd
    return audio_codec_amd


CASES = [
    (48000, 10.0, 0, [16000, 32000, 64000, 96000, 128000, 192000, 256000, 320000]),
    (48000, 5.0, 0, [32000, 64000, 96000, 128000, 256000, 320000]),
    (48000, 2.5, 0, [64000, 96000, 128000, 320000]),
    (44100, 10.0, 0, [32000, 64000, 128000, 256000]),
    (44100, 5.0, 0, [64000, 128000]),
    (32000, 10.0, 0, [32000, 64000, 96000, 128000, 192000, 320000]),
    (32000, 5.0, 0, [32000, 64000, 96000, 192000]),
    (32000, 2.5, 0, [64000, 96000, 128000, 256000]),
    (24000, 10.0, 0, [16000, 32000, 64000, 128000]),
    (24000, 5.0, 0, [32000, 64000, 96000, 160000]),
    (24000, 2.5, 0, [64000, 96000, 128000, 256000]),
    (16000, 10.0, 0, [16000, 32000, 64000, 128000]),
    (16000, 5.0, 0, [32000, 64000, 96000, 128000]),
    (16000, 2.5, 0, [64000, 96000, 128000, 192000]),
    (8000, 10.0, 0, [16000, 24000, 32000, 64000]),
    (8000, 5.0, 0, [32000, 48000, 64000, 96000]),
    (8000, 2.5, 0, [64000, 96000, 128000, 160000]),
    (48000, 10.0, 1, [128000, 256000, 400000, 500000]),
    (48000, 5.0, 1, [160000, 320000, 600000]),
    (48000, 2.5, 1, [172800, 256000, 400000]),
    (96000, 2.5, 1, [256000, 198400, 320000, 672000]),
    # the large-layout kernel: N = 960, or an IMDCT memory of 360 samples
    (96000, 5.0, 1, [256000, 400000, 600000]),
    (96000, 10.0, 1, [149600, 256000, 400000, 500000]),
]


@pytest.mark.parametrize("fs,ms,hr,rates", CASES)
def test_vs_oracle(fs, ms, hr, rates):
    T = 30
    frames, nbytes, bfi = make_dec_case(fs, ms, hr, 1, rates, T, seed=fs // 100 + int(ms * 10) + hr)
    d = _amd().DecBatch(len(rates), fs, 1, ms, hr, nbytes, device=0)
    a, sa = d.decode(frames[:, :13], bfi[:, :13])          # two launches: the decoder memories must persist in HBM
    b, sb = d.decode(frames[:, 13:], bfi[:, 13:])
    got, status = np.concatenate([a, b], axis=1), np.concatenate([sa, sb], axis=1)
    want, wstatus = oracle_decode_streams(frames, nbytes, bfi, fs, ms, hr, 1)
    assert (status == wstatus).all()
    bad = np.argwhere((got != want).any(axis=(2, 3)))
    assert len(bad) == 0, ("first differing (stream, frame)", bad[:4].tolist())


@pytest.mark.parametrize("fs,ms,br,nbytes", [(48000, 10.0, 80000, 100), (48000, 10.0, 96000, 120), (48000, 10.0, 102400, 128),
                                             (32000, 10.0, 96000, 120), (48000, 10.0, 104000, 130), (48000, 10.0, 64000, 80)])
def test_uniform_size_batches_lds_parse_path(fs, ms, br, nbytes):
    """lc3_dec_parse_kernel stages frames of up to 128 bytes in LDS and the switch is per batch: batches whose streams ALL have
    the same size, at 10 ms frames of 81 ... 128 bytes (the longest symbol lists that path sees), the first size above it (global
    variant) and the metric's 80 bytes; lost and corrupt frames included, two launches.  R/ari_codec.c:204-509."""
    B, T = 64, 26
    frames, nb, bfi = make_dec_case(fs, ms, 0, 1, [br] * B, T, seed=nbytes)
    assert set(nb) == {nbytes}
    d = _amd().DecBatch(B, fs, 1, ms, 0, nb, device=0)
    a, sa = d.decode(frames[:, :11], bfi[:, :11])
    b, sb = d.decode(frames[:, 11:], bfi[:, 11:])
    got, status = np.concatenate([a, b], axis=1), np.concatenate([sa, sb], axis=1)
    want, wstatus = oracle_decode_streams(frames, nb, bfi, fs, ms, 0, 1)
    assert (status == wstatus).all()
    bad = np.argwhere((got != want).any(axis=(2, 3)))
    assert len(bad) == 0, ("first differing (stream, frame)", bad[:4].tolist())


def test_golden_reference_decoder_output():
    """tests/golden/d1_decoder_operating_points.npz: frames damaged on purpose and the PCM / status the unmodified ETSI decoder
    produced from them (tests/golden/make_golden_dec.py)."""
    g = np.load(os.path.join(G, "d1_decoder_operating_points.npz"))
    for tag in g["tags"]:
        tag = str(tag)
        fs, dms, hr, ch = (int(v) for v in g[tag + "_cfg"])
        frames, nbytes, bfi = g[tag + "_frames"], g[tag + "_nbytes"], g[tag + "_bfi"]
        got, status = _amd().DecBatch(frames.shape[0], fs, ch, dms / 10.0, hr, nbytes, device=0).decode(frames, bfi)
        assert (status == g[tag + "_status"]).all(), tag
        assert (got == g[tag + "_pcm"]).all(), tag


@pytest.mark.parametrize("bps", [24, 32])
def test_output_depths(bps):
    rates = [32000, 64000, 128000, 256000]
    frames, nbytes, bfi = make_dec_case(48000, 10.0, 0, 1, rates, 12, seed=3)
    got, status = _amd().DecBatch(len(rates), 48000, 1, 10.0, 0, nbytes, device=0).decode(frames, bfi, bps)
    want, wstatus = oracle_decode_streams(frames, nbytes, bfi, 48000, 10.0, 0, 1, bps)
    assert (status == wstatus).all() and (got == want).all()


@pytest.mark.parametrize("fs,ms,hr,rates", [(48000, 10.0, 0, [64000, 128000, 256000, 96000]), (32000, 5.0, 0, [64000, 128000]),
                                            (48000, 2.5, 1, [345600, 512000])])
def test_stereo_and_error_propagation(fs, ms, hr, rates):
    """Two channels per stream: the payload split, and a corrupt first channel concealing the second (R/dec_lc3_fl.c:146-160)."""
    frames, nbytes, bfi = make_dec_case(fs, ms, hr, 2, rates, 24, seed=17, loss=0.1, corrupt=0.3)
    got, status = _amd().DecBatch(len(rates), fs, 2, ms, hr, nbytes, device=0).decode(frames, bfi)
    want, wstatus = oracle_decode_streams(frames, nbytes, bfi, fs, ms, hr, 2)
    assert (status == wstatus).all()
    assert (got == want).all()


def test_long_loss_bursts():
    """Concealment over bursts longer than the attenuation ramp (R/plc_noise_substitution.c, R/plc_damping_scrambling.c)."""
    rates = [32000, 96000, 160000]
    frames, nbytes, _ = make_dec_case(48000, 10.0, 0, 1, rates, 40, seed=5, loss=0, corrupt=0)
    bfi = np.zeros((3, 40), dtype=np.uint8)
    bfi[0, 5:25] = 1; bfi[1, 0:4] = 1; bfi[1, 20:23] = 1; bfi[2, 10:40] = 1
    got, status = _amd().DecBatch(3, 48000, 1, 10.0, 0, nbytes, device=0).decode(frames, bfi)
    want, wstatus = oracle_decode_streams(frames, nbytes, bfi, 48000, 10.0, 0, 1)
    assert (status == bfi).all() and (wstatus == bfi).all()
    assert (got == want).all()


@pytest.mark.parametrize("fs,ms,hr,rates", [(48000, 10.0, 0, [64000, 128000]), (32000, 5.0, 0, [64000, 96000]), (96000, 2.5, 1, [256000, 400000])])
def test_stage_traces_match_oracle(fs, ms, hr, rates):
    """Stage by stage: what the traced entry point records per decoded channel-frame (side information, TNS orders and indices, scale
    factor indices, LTPF parameters, noise-filling seed, zero-frame flag, residual count, the spectrum after TNS and after SNS shaping,
    the IMDCT output and the synthesis output) equals the oracle decoder's trace exactly, on good frames of a stream with lost frames in
    between (the fields the first kernel does not hand over are skipped, as in tests/gpu_dec_debug.py)."""
    import ctypes as C
    from lc3_harness import DecTrace, OracleDecoder
    T = 10
    frames, nbytes, bfi = make_dec_case(fs, ms, hr, 1, rates, T, seed=19, loss=0.2, corrupt=0)
    B = len(rates)
    N = int(fs * ms / 1000)
    db = _amd().DecBatch(B, fs, 1, ms, hr, nbytes, device=0)
    got, status, traces = db.decode_traced(frames, bfi)
    bad = []
    for b in range(B):
        o = OracleDecoder(fs, 1, ms, hr, portable_math=True)
        tr = o.enable_trace()
        for t in range(T):
            rc, want = o.decode(frames[b, t, :nbytes[b]], int(bfi[b, t]))
            assert (got[b, t] == want).all() and int(status[b, t]) == int(rc == 2), (b, t)
            if rc != 0: continue
            g = DecTrace.from_buffer_copy(traces[b * T + t].tobytes()[:C.sizeof(DecTrace)])
            for f, _ in DecTrace._fields_:
                if f in ("bfi", "xq", "q_gain", "scf_q"): continue
                ga, ca = getattr(g, f), getattr(tr[0], f)
                if hasattr(ga, "__len__"):
                    n = N if len(ga) == 960 else len(ga)
                    a, c = np.ctypeslib.as_array(ga)[:n], np.ctypeslib.as_array(ca)[:n]
                else:
                    a, c = np.asarray([ga]), np.asarray([ca])
                same = (a.view(np.uint32) == c.view(np.uint32)) | ((a == 0) & (c == 0)) if a.dtype.kind == "f" else (a == c)
                if not same.all(): bad.append((b, t, f, int((~same).sum())))
    assert not bad, bad[:10]


def test_checkpoint_resume_state():
    """lc3plus_dec_batch_get_state / set_state: a second batch of the same configuration given the first one's state continues the
    streams sample for sample - the checkpoint taken in the middle of a loss burst (concealment counters, attenuation, last good
    spectrum, overlap memory and LTPF histories all cross it)."""
    rates = [32000, 64000, 96000, 160000]
    frames, nbytes, _ = make_dec_case(48000, 10.0, 0, 1, rates, 30, seed=77, loss=0.1, corrupt=0.05)
    bfi = np.zeros((4, 30), dtype=np.uint8); bfi[1, 9:14] = 1; bfi[3, 11] = 1
    amd = _amd()
    a = amd.DecBatch(4, 48000, 1, 10.0, 0, nbytes, device=0)
    g1, s1 = a.decode(frames[:, :12], bfi[:, :12])
    st = a.get_state()
    b = amd.DecBatch(4, 48000, 1, 10.0, 0, nbytes, device=0)
    b.set_state(st)
    g2, s2 = b.decode(frames[:, 12:], bfi[:, 12:])
    want, wstatus = oracle_decode_streams(frames, nbytes, bfi, 48000, 10.0, 0, 1)
    assert (np.concatenate([g1, g2], axis=1) == want).all() and (np.concatenate([s1, s2], axis=1) == wstatus).all()
    with pytest.raises(Exception):
        b.set_state(st[:-4])                                   # a state of another size is refused


def test_single_stream_api_with_rate_switch_and_empty_frames():
    """lc3_dec_fl as R/codec_exe.c drives it: frame size changes mid-stream (R/dec_lc3_fl.c:149-155), num_bytes = 0 means lost."""
    amd = _amd()
    N, T = 480, 36
    pcm = synth_pcm(1, T, N, 48000, seed=9)[0]
    enc = Oracle(48000, 1, 10.0, 0, 64000)
    frames = []
    for t in range(T):
        if t == 12: enc.set_bitrate(128000)
        if t == 24: enc.set_bitrate(32000)
        frames.append(enc.encode(pcm[t][None]))
    d = amd.Decoder(48000, 1, 10.0, 0)
    o = OracleDecoder(48000, 1, 10.0, 0, portable_math=True)
    for t in range(T):
        f, bfi = frames[t], 0
        if t in (7, 8, 25): f = np.zeros(0, dtype=np.uint8)
        if t == 15: bfi = 1
        got, rc = d.decode(f.tobytes(), bfi)
        wrc, want = o.decode(f if f.size else np.zeros(1, dtype=np.uint8), bfi, num_bytes=int(f.size))
        assert rc == wrc, (t, rc, wrc)
        assert (got == want).all(), t
    d.close()


def test_roundtrip_with_gpu_encoder_large():
    """Size-independent property at a large batch: GPU encode -> GPU decode reproduces the input closely (delay compensated),
    and a sample of the streams equals the oracle decoder exactly."""
    amd = _amd()
    B, T, N = 2048, 20, 480
    pcm = synth_pcm(B, T, N, 48000, seed=31)
    enc = amd.Batch(B, 48000, 1, 10.0, 0, [128000] * B, device=0)
    frames = enc.encode(pcm)
    nb = enc.num_bytes(0)
    dec = amd.DecBatch(B, 48000, 1, 10.0, 0, [nb] * B, device=0)
    out, status = dec.decode(frames)
    assert status.sum() == 0
    x = pcm.reshape(B, -1).astype(np.float64)
    y = out.reshape(B, -1).astype(np.float64)
    best = None
    for lag in (120, 240, 300, 480):   # codec delay candidates; lc3_enc_get_delay + lc3_dec_get_delay is one of them
        e = ((x[:, N:-N - lag] - y[:, N + lag:y.shape[1] - N]) ** 2).sum(axis=1)
        s = (x[:, N:-N - lag] ** 2).sum(axis=1)
        snr = 10 * np.log10(np.maximum(s, 1e-9) / np.maximum(e, 1e-9))
        best = snr if best is None or snr.mean() > best.mean() else best
    assert np.median(best) > 15.0, np.median(best)
    pick = [0, 1, 777, 2047]
    want, _ = oracle_decode_streams(frames[pick], [nb] * len(pick), None, 48000, 10.0, 0, 1)
    assert (out[pick] == want).all()


def _expected_wav(frames, loss, fs, ms, hr, channels, nsamp, bps, dc=1):
    """what R/codec_exe.c:383-450 + R/tinywaveout_c.h write for these frames, computed with the CPU oracle decoder"""
    import struct
    o = OracleDecoder(fs, channels, ms, hr, portable_math=True)
    N = o.N
    la = {(48000, 10.0): 180, (48000, 5.0): 60, (32000, 10.0): 120}[(fs, ms)]         # MDCT_la_zeroes* (R/constants.c:3037-3049)
    delay = (N - 2 * la) // dc if dc else 0
    data, edf, left = [], [], nsamp
    for t, f in enumerate(frames):
        lost = bool(loss[t % len(loss)])
        rc, pcm = o.decode(f if not lost else np.zeros(1, np.uint8), 0, bps, num_bytes=None if not lost else 0)
        edf.append(int(rc == 2))
        n_out = min(N - delay, left) if left < 2 ** 32 else N - delay
        x = pcm[:, delay:delay + n_out].T.reshape(-1)
        if bps == 16: data.append(x.astype("<i2").tobytes())
        elif bps == 24: data.append(b"".join(struct.pack("<i", int(np.clip(v, -8388608, 8388607)))[:3] for v in x))
        else: data.append(x.astype("<i4").tobytes())
        left = (left - (N - delay)) % 2 ** 32
        delay = 0
    if 0 < left < N:
        data.append(bytes(left * channels * (bps // 8)))
    data = b"".join(data)
    ba = channels * (bps // 8)
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, channels, fs, fs * ba, ba, bps) + b"data" + struct.pack("<I", len(data))
    return hdr + data, np.array(edf, dtype="<i2").tobytes()


@pytest.mark.parametrize("fs,ms,channels,bps,bitrate,g192", [(48000, 10.0, 1, 16, 64000, 0), (48000, 5.0, 2, 24, 128000, 0), (32000, 10.0, 1, 32, 48000, 1)])
def test_cli_decoder_front_end(tmp_path, fs, ms, channels, bps, bitrate, g192):
    """tools/lc3plus_dec_cli (C, on the C ABI) against the ETSI CLI in decode mode (oracle/_ref/LC3plus -D, when it travelled with the
    snapshot) and against the file the oracle decoder implies: container reader, error pattern file, delay compensation, WAV writer."""
    import subprocess, struct
    from lc3_harness import ORACLE_DIR
    from test_gpu_parity import _container
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "tools", "lc3plus_dec_cli")
    if not os.path.exists(cli):
        subprocess.check_call(["make", "-s", "-C", root, "cli"])
    N = int(fs * ms / 1000); T = 40
    nsamp = N * (T - 1) - 77
    pcm = synth_pcm(channels, T, N, fs, seed=41)
    enc = Oracle(fs, channels, ms, 0, bitrate)
    frames = []
    for t in range(T):
        if t == 20: enc.set_bitrate(bitrate * 3 // 2)              # a frame size change inside the file
        frames.append(enc.encode(pcm[:, t]))
    loss = [0, 0, 0, 1, 0, 0, 0, 0, 1, 1, 0]
    bs = tmp_path / "in.lc3plus"
    blob = _container(frames, fs, bitrate, channels, ms, nsamp, 0)
    if g192:
        open(str(bs) + ".cfg", "wb").write(blob[:20])
        with open(bs, "wb") as f:
            for fr in frames:
                f.write(struct.pack("<HH", 0x6B21, fr.size * 8))
                f.write(np.where(np.unpackbits(fr, bitorder="little"), 0x0081, 0x007F).astype("<i2").tobytes())
    else:
        open(bs, "wb").write(blob)
    epf = tmp_path / "loss.dat"; np.array(loss, dtype="<i2").tofile(epf)
    ours, edf = tmp_path / "ours.wav", tmp_path / "ours.edf"
    opts = ["-q", "-bps", str(bps), "-epf", str(epf)] + (["-formatG192"] if g192 else [])
    subprocess.check_call([cli, "-D"] + opts + ["-edf", str(edf), str(bs), str(ours)])
    got, got_edf = open(ours, "rb").read(), open(edf, "rb").read()
    want, want_edf = _expected_wav(frames, loss, fs, ms, 0, channels, nsamp, bps)
    assert got_edf == want_edf
    assert len(got) == len(want) and got[:44] == want[:44]
    assert got == want
    ref_cli = os.path.join(ORACLE_DIR, "_ref", "LC3plus")
    if os.path.exists(ref_cli):
        theirs, tedf = tmp_path / "ref.wav", tmp_path / "ref.edf"
        subprocess.check_call([ref_cli, "-D"] + opts + ["-edf", str(tedf), str(bs), str(theirs)], stdout=subprocess.DEVNULL)
        ref = open(theirs, "rb").read()
        assert open(tedf, "rb").read() == got_edf
        assert len(ref) == len(got) and ref[:44] == got[:44]
        # The reference evaluates powf() with the host libm, the device with (float)pow((double)): a sample whose value sits on a
        # rounding boundary may come out one output LSB apart (libm boundary); everything else is identical.
        def samples(blob):
            raw = np.frombuffer(blob[44:], dtype=np.uint8)
            if bps == 16: return raw.view("<i2").astype(np.int64)
            if bps == 32: return raw.view("<i4").astype(np.int64)
            r = raw.reshape(-1, 3).astype(np.int64)
            v = r[:, 0] | (r[:, 1] << 8) | (r[:, 2] << 16)
            return np.where(v >= 1 << 23, v - (1 << 24), v)
        a, b = samples(ref), samples(got)
        tol = np.maximum(1, np.abs(a) >> 22)
        assert (np.abs(a - b) <= tol).all()
        assert (a == b).mean() > 0.999


def test_soak_every_operating_point():
    """All 24 (sample rate, frame length, mode) families x 2 seeds, mono and stereo, 16/24/32-bit output, lost and corrupt frames,
    two launches per stream, against the oracle decoder (tools/dec_soak.py; the full-size run is quoted in DESIGN.md)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("dec_soak", os.path.join(root, "tools", "dec_soak.py"))
    soak = importlib.util.module_from_spec(spec); spec.loader.exec_module(soak)
    tot, bad = soak.run(2, 12, 30, verbose=False)
    assert tot > 24 * 2 * 12 * 30 and bad == 0, (tot, bad)


@pytest.mark.parametrize("fs,ms,hr,rates,B,T,K", [(48000, 10.0, 0, [64000, 128000, 32000], 192, 12, 5), (16000, 5.0, 0, [32000, 64000], 128, 20, 4)])
def test_consecutive_device_calls_overlap_with_input_ready(fs, ms, hr, rates, B, T, K):
    """lc3plus_dec_batch_set_input_ready: K device-pointer calls queued back to back without a host synchronisation in between - the parser of
    call k+1 runs on its own stream beside the transform and synthesis of call k, into the other set of hand-over buffers - give the samples
    of one continuous decode (the oracle decoder); then an ordered call (host pointers) and the promise withdrawn again."""
    from test_gpu_parity import _Dev
    amd = _amd()
    d = _Dev()
    try:
        TT = T * K + 2 * T
        br = [rates[i % len(rates)] for i in range(B)]
        frames, nbytes, _ = make_dec_case(fs, ms, hr, 1, br, TT, seed=77 + B)
        dec = amd.DecBatch(B, fs, 1, ms, hr, nbytes, device=0)
        stride = frames.shape[2]; N = dec.N
        dec.set_input_ready(True)
        ins = [d.put(frames[:, k * T:(k + 1) * T]) for k in range(K)]
        outs = [d.zeros(B * T * N * 2) for _ in range(K)]
        d.sync()                                          # the promise: all frames are on the device before the first call
        for k in range(K):
            dec.decode_device(ins[k], stride, T, outs[k], 16, hip_stream=None, sync=False)
        d.sync()
        got = [d.get(outs[k], (B, T, 1, N), np.int16) for k in range(K)]
        a, _ = dec.decode(frames[:, K * T:(K + 1) * T])                      # host pointers: the ordered path, first set of buffers
        dec.set_input_ready(False)
        pin = d.put(frames[:, (K + 1) * T:]); pout = d.zeros(B * T * N * 2)
        dec.decode_device(pin, stride, T, pout, 16, hip_stream=None, sync=True)
        got += [a, d.get(pout, (B, T, 1, N), np.int16)]
        got = np.concatenate(got, axis=1)
        want, _ = oracle_decode_streams(frames, nbytes, None, fs, ms, hr, 1)
        bad = np.argwhere((got != want).any(axis=(2, 3)))
        assert len(bad) == 0, ("first differing (stream, frame)", bad[:4].tolist())
    finally:
        d.free()


def test_promise_given_while_an_ordered_call_is_in_flight():
    """an ASYNCHRONOUS ordered device-pointer call (promise off, sync = 0), then lc3plus_dec_batch_set_input_ready(1), then another
    asynchronous call - the first 'ahead' call, whose parser must not overwrite the hand-over buffers the call before is still reading
    (lc3hip_dec_set_input_ready drains the last stream on the 0 -> 1 transition).  Large enough that the first call is still running when the second
    is queued; twice, so that the second set of buffers has been both."""
    from test_gpu_parity import _Dev
    amd = _amd()
    d = _Dev()
    try:
        B, T, K = 2048, 24, 4
        U = 128                                             # distinct streams, tiled: the oracle decodes 128
        fr_u, nb_u, _ = make_dec_case(48000, 10.0, 0, 1, [64000, 96000] * (U // 2), T * K, seed=1234)
        reps = B // U
        frames = np.ascontiguousarray(np.tile(fr_u, (reps, 1, 1))); nbytes = list(nb_u) * reps
        dec = amd.DecBatch(B, 48000, 1, 10.0, 0, nbytes, device=0)
        stride = frames.shape[2]; N = dec.N
        ins = [d.put(frames[:, k * T:(k + 1) * T]) for k in range(K)]
        outs = [d.zeros(B * T * N * 2) for _ in range(K)]
        d.sync()
        for k in range(K):
            dec.set_input_ready(k % 2 == 1)                # off, on, off, on: every 'on' call follows an ordered call that was not waited for
            dec.decode_device(ins[k], stride, T, outs[k], 16, hip_stream=None, sync=False)
        d.sync()
        got = np.concatenate([d.get(outs[k], (B, T, 1, N), np.int16) for k in range(K)], axis=1)
        want, _ = oracle_decode_streams(fr_u, nb_u, None, 48000, 10.0, 0, 1)
        for r in range(reps):
            bad = np.argwhere((got[r * U:(r + 1) * U] != want).any(axis=(2, 3)))
            assert len(bad) == 0, ("copy", r, "first differing (stream, frame)", bad[:4].tolist())
    finally:
        d.free()
