"""Test-side helpers: ctypes bindings for the CPU oracle (oracle/liblc3_oracle*.so), the compiled
ETSI reference (oracle/_ref/liblc3_etsi_fl.so, optional) and the deterministic synthetic PCM generator
(SURVEY.md 8(d) "Synthetic PCM").  TEST INFRASTRUCTURE ONLY -- the product never imports this."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "liblc3_etsi_fl.so")


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "restatement"])


def have_ref():
    return os.path.exists(REF_SO)


# ---------------------------------------------------------------------------------------------
# synthetic PCM (int16): sinusoids + coloured noise + transients; some silent / full-scale streams
# ---------------------------------------------------------------------------------------------
def synth_pcm(n_streams, n_frames, frame_len, fs, seed=1, dtype=np.int16):
    """[n_streams, n_frames, frame_len] deterministic int16 test signal."""
    out = np.zeros((n_streams, n_frames * frame_len), dtype=np.float64)
    t = np.arange(n_frames * frame_len) / fs
    for s in range(n_streams):
        rng = np.random.RandomState((0x9E3779B9 * (s + 1) + seed) & 0x7FFFFFFF)
        kind = s % 64
        if kind == 63:            # digital silence
            continue
        if kind == 62:            # full-scale white noise
            out[s] = rng.uniform(-32768, 32767, size=t.size)
            continue
        x = np.zeros_like(t)
        for _ in range(3):
            f = rng.uniform(80.0, 0.4 * fs)
            a = rng.uniform(0.02, 0.25) * 32767
            x += a * np.sin(2 * np.pi * f * t + rng.uniform(0, 2 * np.pi))
        if kind % 4 == 1:         # strongly periodic (pitch) stream to exercise LTPF
            f0 = rng.uniform(90.0, 380.0)
            for h in range(1, 12):
                x += (0.12 / h) * 32767 * np.sin(2 * np.pi * f0 * h * t + h)
        noise = rng.standard_normal(t.size)
        noise = np.convolve(noise, [0.5, 0.3, 0.15, 0.05], mode="same")   # pink-ish
        x += noise * 32767 * 10 ** (-30 / 20)
        # 20 dB step transient every 37 frames
        env = np.ones_like(t)
        for k in range(0, n_frames, 37):
            a0 = k * frame_len + frame_len // 3
            env[a0:a0 + frame_len // 2] *= 10.0
        x *= env * 0.1
        out[s] = x
    out = np.clip(np.rint(out), -32768, 32767)
    return out.reshape(n_streams, n_frames, frame_len).astype(dtype)


# ---------------------------------------------------------------------------------------------
# oracle restatement
# ---------------------------------------------------------------------------------------------
class Trace(C.Structure):
    _fields_ = [
        ("spec_mdct", C.c_float * 960), ("s12k8", C.c_float * 129), ("T0", C.c_int), ("normcorr", C.c_float),
        ("ltpf_param", C.c_int * 3), ("ltpf_bits", C.c_int), ("attack", C.c_int), ("ener", C.c_float * 64),
        ("bw_idx", C.c_int), ("scf", C.c_float * 16), ("scf_idx", C.c_int * 7), ("scf_q", C.c_float * 16),
        ("spec_shaped", C.c_float * 960), ("tns_nfilt", C.c_int), ("tns_order", C.c_int * 2),
        ("tns_rc_idx", C.c_int * 16), ("tns_bits", C.c_int), ("spec_tns", C.c_float * 960),
        ("target_bits_quant", C.c_int), ("gain0", C.c_float), ("gg_idx0", C.c_int), ("gg_min", C.c_int),
        ("nbits0", C.c_int), ("gain", C.c_float), ("gg_idx", C.c_int), ("gain_change", C.c_int),
        ("nbits", C.c_int), ("nbits2", C.c_int), ("lastnz", C.c_int), ("lsb_mode", C.c_int),
        ("xq", C.c_int * 960), ("fac_ns", C.c_int), ("n_res_bits", C.c_int), ("bp_side", C.c_int),
        ("mask_side", C.c_int),
    ]


class Oracle:
    """One multi-channel encoder instance of the CPU restatement."""

    def __init__(self, fs, channels=1, frame_ms=10.0, hrmode=0, bitrate=64000, portable_math=False, bandwidth=0):
        name = "liblc3_oracle_pm.so" if portable_math else "liblc3_oracle.so"
        path = os.path.join(ORACLE_DIR, name)
        if not os.path.exists(path):
            build_oracle()
        self.lib = C.CDLL(path)
        L = self.lib
        L.lc3o_enc_set_frame_ms.argtypes = [C.c_void_p, C.c_float]
        L.lc3o_enc_frame.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        for f in ("lc3o_enc_init", "lc3o_enc_set_hrmode", "lc3o_enc_set_bitrate", "lc3o_enc_set_bandwidth",
                  "lc3o_enc_get_input_samples", "lc3o_enc_get_num_bytes", "lc3o_enc_get_real_bitrate", "lc3o_enc_get_delay"):
            getattr(L, f).restype = C.c_int
        self.buf = C.create_string_buffer(L.lc3o_enc_sizeof())
        self.p = C.cast(self.buf, C.c_void_p)
        self.channels = channels
        self.err = L.lc3o_enc_init(self.p, fs, channels)
        if not self.err: self.err = L.lc3o_enc_set_frame_ms(self.p, frame_ms)
        if not self.err: self.err = L.lc3o_enc_set_hrmode(self.p, hrmode)
        if not self.err: self.err = L.lc3o_enc_set_bitrate(self.p, bitrate)
        if not self.err and bandwidth: self.err = L.lc3o_enc_set_bandwidth(self.p, bandwidth)
        if self.err:
            raise RuntimeError("oracle setup error %d" % self.err)
        self.N = L.lc3o_enc_get_input_samples(self.p)
        self.nbytes = L.lc3o_enc_get_num_bytes(self.p)
        self.trace = None

    def enable_trace(self):
        self.trace = (Trace * self.channels)()
        self.lib.lc3o_enc_set_trace.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.lc3o_enc_set_trace(self.p, C.cast(self.trace, C.c_void_p))
        return self.trace

    def set_bitrate(self, br):
        rc = self.lib.lc3o_enc_set_bitrate(self.p, br)
        self.nbytes = self.lib.lc3o_enc_get_num_bytes(self.p)
        return rc

    def set_bandwidth(self, bw):
        return self.lib.lc3o_enc_set_bandwidth(self.p, bw)

    def encode(self, planar, bitdepth=16):
        """planar: [channels, N] int16 (bitdepth 16) or int32."""
        planar = np.ascontiguousarray(planar)
        ptrs = (C.c_void_p * self.channels)(*[planar[c].ctypes.data for c in range(self.channels)])
        out = np.zeros(self.nbytes, dtype=np.uint8)
        nb = C.c_int(0)
        rc = self.lib.lc3o_enc_frame(self.p, ptrs, bitdepth, out.ctypes.data, C.byref(nb))
        if rc:
            raise RuntimeError("oracle encode error %d" % rc)
        assert nb.value == self.nbytes
        return out


def oracle_encode_streams(pcm, fs, frame_ms, hrmode, bitrates, portable_math=False):
    """pcm [B,T,N] int16 mono streams -> list of [T, nbytes_b] uint8 arrays."""
    B, T, N = pcm.shape
    outs = []
    for b in range(B):
        o = Oracle(fs, 1, frame_ms, hrmode, int(bitrates[b]), portable_math)
        assert o.N == N
        frames = np.zeros((T, o.nbytes), dtype=np.uint8)
        for t in range(T):
            frames[t] = o.encode(pcm[b, t][None, :])
        outs.append(frames)
    return outs


# ---------------------------------------------------------------------------------------------
# compiled ETSI reference (only where oracle/_ref was built)
# ---------------------------------------------------------------------------------------------
class Ref:
    def __init__(self, fs, channels=1, frame_ms=10.0, hrmode=0, bitrate=64000, bandwidth=0):
        L = self.lib = C.CDLL(REF_SO)
        L.lc3_enc_set_frame_ms.argtypes = [C.c_void_p, C.c_float]
        L.lc3_enc_fl.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        size = L.lc3_enc_get_size(fs, channels)
        if size <= 0:
            raise RuntimeError("ref: unsupported fs/channels")
        self.buf = C.create_string_buffer(size + 8)
        self.p = C.cast(self.buf, C.c_void_p)
        self.channels = channels
        err = L.lc3_enc_init(self.p, fs, channels)
        if not err: err = L.lc3_enc_set_frame_ms(self.p, frame_ms)
        if not err: err = L.lc3_enc_set_hrmode(self.p, hrmode)
        if not err: err = L.lc3_enc_set_bitrate(self.p, bitrate)
        if not err and bandwidth: err = L.lc3_enc_set_bandwidth(self.p, bandwidth)
        self.err = err
        if err:
            raise RuntimeError("ref setup error %d" % err)
        self.N = L.lc3_enc_get_input_samples(self.p)
        self.nbytes = L.lc3_enc_get_num_bytes(self.p)

    def set_bitrate(self, br):
        return self.lib.lc3_enc_set_bitrate(self.p, br)

    def encode(self, planar, bitdepth=16):
        planar = np.ascontiguousarray(planar)
        ptrs = (C.c_void_p * self.channels)(*[planar[c].ctypes.data for c in range(self.channels)])
        out = np.zeros(self.nbytes, dtype=np.uint8)
        nb = C.c_int(self.nbytes)
        rc = self.lib.lc3_enc_fl(self.p, ptrs, bitdepth, out.ctypes.data, C.byref(nb))
        if rc:
            raise RuntimeError("ref encode error %d" % rc)
        return out

    def __del__(self):
        try:
            self.lib.lc3_free_encoder_structs(self.p)
        except Exception:
            pass


def ref_encode_streams(pcm, fs, frame_ms, hrmode, bitrates):
    B, T, N = pcm.shape
    outs = []
    for b in range(B):
        o = Ref(fs, 1, frame_ms, hrmode, int(bitrates[b]))
        assert o.N == N
        frames = np.zeros((T, o.nbytes), dtype=np.uint8)
        for t in range(T):
            frames[t] = o.encode(pcm[b, t][None, :])
        outs.append(frames)
    return outs


class DecTrace(C.Structure):
    """lc3o_dec_trace (oracle/lc3_oracle.h) == lc3d_dec_trace (audio_codec_amd/csrc/lc3_shim.h)"""
    _fields_ = [("bfi", C.c_int), ("bw_idx", C.c_int), ("lastnz", C.c_int), ("lsb_mode", C.c_int), ("gg_idx", C.c_int), ("fac_ns", C.c_int),
                ("nfilt", C.c_int), ("tns_order", C.c_int * 2), ("tns_idx", C.c_int * 16), ("scf_idx", C.c_int * 7), ("ltpf", C.c_int * 3),
                ("nf_seed", C.c_int), ("zero_frame", C.c_int), ("nres", C.c_int), ("xq", C.c_int * 960), ("scf_q", C.c_float * 16),
                ("q_gain", C.c_float * 960), ("q_tns", C.c_float * 960), ("q_shaped", C.c_float * 960), ("x_imdct", C.c_float * 960),
                ("x_out", C.c_float * 960)]


class OracleDecoder:
    """oracle/lc3_oracle_dec.inc through ctypes (same call shape as RefDecoder)."""

    def enable_trace(self):
        self.trace = (DecTrace * self.channels)()
        self.lib.lc3o_dec_set_trace.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.lc3o_dec_set_trace(self.p, C.cast(self.trace, C.c_void_p))
        return self.trace

    def __init__(self, fs, channels=1, frame_ms=10.0, hrmode=0, portable_math=False):
        L = self.lib = C.CDLL(os.path.join(ORACLE_DIR, "liblc3_oracle_pm.so" if portable_math else "liblc3_oracle.so"))
        L.lc3o_dec_set_frame_ms.argtypes = [C.c_void_p, C.c_float]
        L.lc3o_dec_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int]
        self.buf = C.create_string_buffer(L.lc3o_dec_sizeof() + 8)
        self.p = C.cast(self.buf, C.c_void_p)
        self.channels = channels
        err = L.lc3o_dec_init(self.p, fs, channels)
        if not err: err = L.lc3o_dec_set_frame_ms(self.p, frame_ms)
        if not err: err = L.lc3o_dec_set_hrmode(self.p, hrmode)
        if err:
            raise RuntimeError("oracle decoder setup error %d" % err)
        self.N = L.lc3o_dec_get_output_samples(self.p)

    def decode(self, frame_bytes, bfi=0, bps=16, num_bytes=None):
        frame_bytes = np.ascontiguousarray(frame_bytes, dtype=np.uint8)
        out = np.zeros((self.channels, self.N), dtype=np.int16 if bps == 16 else np.int32)
        ptrs = (C.c_void_p * self.channels)(*[out[c].ctypes.data for c in range(self.channels)])
        rc = self.lib.lc3o_dec_frame(self.p, frame_bytes.ctypes.data, int(frame_bytes.size) if num_bytes is None else num_bytes, ptrs, bps, bfi)
        return rc, out


class RefDecoder:
    """ETSI float decoder (tool use only: turns bitstreams back into PCM for distance metrics)."""

    def __init__(self, fs, channels=1, frame_ms=10.0, hrmode=0):
        L = self.lib = C.CDLL(REF_SO)
        L.lc3_dec_set_frame_ms.argtypes = [C.c_void_p, C.c_float]
        size = L.lc3_dec_get_size(fs, channels, 0)
        self.buf = C.create_string_buffer(size + 8)
        self.p = C.cast(self.buf, C.c_void_p)
        self.channels = channels
        err = L.lc3_dec_init(self.p, fs, channels, 0)
        if not err: err = L.lc3_dec_set_frame_ms(self.p, frame_ms)
        if not err: err = L.lc3_dec_set_hrmode(self.p, hrmode)
        if err:
            raise RuntimeError("ref decoder setup error %d" % err)
        self.N = L.lc3_dec_get_output_samples(self.p)
        L.lc3_dec16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int]

    def decode(self, frame_bytes, bfi=0, bps=16):
        frame_bytes = np.ascontiguousarray(frame_bytes, dtype=np.uint8)
        out = np.zeros((self.channels, self.N), dtype=np.int16 if bps == 16 else np.int32)
        ptrs = (C.c_void_p * self.channels)(*[out[c].ctypes.data for c in range(self.channels)])
        self.lib.lc3_dec_fl.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int]
        rc = self.lib.lc3_dec_fl(self.p, frame_bytes.ctypes.data, int(frame_bytes.size), ptrs, bps, bfi)
        return rc, out

    def __del__(self):
        try:
            self.lib.lc3_free_decoder_structs(self.p)
        except Exception:
            pass


def make_dec_case(fs, frame_ms, hrmode, channels, rates, T, seed=7, loss=0.15, corrupt=0.1, enc_cls=None):
    """Encodes seeded synthetic PCM with the CPU oracle (one encoder per stream) and damages the result:
    returns frames uint8 [B, T, stride], nbytes [B], bfi uint8 [B, T].  `loss` marks frames as lost (bfi = 1),
    `corrupt` flips bytes inside frames that are NOT marked, so the decoder has to find the damage itself."""
    rng = np.random.default_rng(seed)
    B = len(rates)
    N = int((48000 if fs == 44100 else fs) * frame_ms / 1000)
    pcm = synth_pcm(B * channels, T, N, fs, seed=seed).reshape(B, channels, T, N).transpose(0, 2, 1, 3)
    per = []
    for b in range(B):
        o = (enc_cls or Oracle)(fs, channels, frame_ms, hrmode, int(rates[b]))
        per.append(np.stack([o.encode(pcm[b, t]) for t in range(T)]))
    nbytes = [p.shape[1] for p in per]
    frames = np.zeros((B, T, max(nbytes)), dtype=np.uint8)
    for b in range(B):
        frames[b, :, :nbytes[b]] = per[b]
    bfi = (rng.random((B, T)) < loss).astype(np.uint8)
    for b in range(B):
        for t in range(T):
            if rng.random() < corrupt:
                k = rng.integers(0, nbytes[b], size=3)
                frames[b, t, k] ^= rng.integers(1, 256, size=3).astype(np.uint8)
    return frames, nbytes, bfi


def oracle_decode_streams(frames, nbytes, bfi, fs, frame_ms, hrmode, channels, bps=16, portable_math=True):
    """CPU oracle decode of make_dec_case() output -> (pcm [B, T, channels, N], status [B, T] (1 = concealed))."""
    B, T = frames.shape[:2]
    out = status = None
    for b in range(B):
        o = OracleDecoder(fs, channels, frame_ms, hrmode, portable_math=portable_math)
        if out is None:
            out = np.zeros((B, T, channels, o.N), dtype=np.int16 if bps == 16 else np.int32)
            status = np.zeros((B, T), dtype=np.uint8)
        for t in range(T):
            rc, pcm = o.decode(frames[b, t, :nbytes[b]], int(bfi[b, t]) if bfi is not None else 0, bps)
            assert rc in (0, 2), rc
            out[b, t] = pcm
            status[b, t] = rc == 2
    return out, status


# ---------------------------------------------------------------------------------------------
# spectral-distance fallback of the conformance procedure (E/conformance/lc3_conformance.py:126-129,572-582):
# decode both bitstreams with the compiled reference decoder, compare the PCM with the ETSI `mld` tool
# ---------------------------------------------------------------------------------------------
MLD_TOOL = os.path.join(ORACLE_DIR, "_ref", "mld")
MLD_THRESHOLD = 4.0     # DEFAULTS_TEST['*_mld_threshold'], lc3_conformance.py:127


def _write_wav16(path, pcm_planar, fs):
    import wave
    w = wave.open(str(path), "wb")
    w.setnchannels(pcm_planar.shape[0]); w.setsampwidth(2); w.setframerate(fs)
    w.writeframes(np.ascontiguousarray(pcm_planar.T).astype("<i2").tobytes())
    w.close()


def ref_decode_stream(frames, fs, frame_ms, hrmode, channels=1):
    """[T, nbytes] uint8 -> [channels, T*N] int16 through the compiled ETSI decoder (oracle/_ref)."""
    d = RefDecoder(fs, channels, frame_ms, hrmode)
    out = [d.decode(f)[1] for f in frames]
    return np.concatenate(out, axis=1)


def mld_between(frames_a, frames_b, fs, frame_ms, hrmode, channels=1, tmpdir=None):
    """Maximum loudness difference (ETSI mld tool) between two bitstreams of one stream, both decoded by the reference
    decoder.  The tool works at 48 kHz (the conformance script resamples); other rates are zero-order-held to 48 kHz
    for 8/16/24 kHz and passed as they are otherwise (a relative check, same treatment on both sides)."""
    import re
    import tempfile
    td = tmpdir or tempfile.mkdtemp(prefix="mld_")
    paths = []
    for tag, fr in (("a", frames_a), ("b", frames_b)):
        pcm = ref_decode_stream(fr, fs, frame_ms, hrmode, channels)
        if 48000 % fs == 0 and fs != 48000:
            pcm = np.repeat(pcm, 48000 // fs, axis=1)
        p = os.path.join(str(td), "mld_%s.wav" % tag)
        _write_wav16(p, pcm, 48000)
        paths.append(p)
    out = subprocess.run([MLD_TOOL, "-d", paths[0], paths[1]], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    m = re.search(r"maximum loudness difference:\s*(\S+)", out)
    if not m:
        raise RuntimeError("mld tool output not understood: %r" % out[-400:])
    return float(m.group(1))


RMS_TOOL = os.path.join(ORACLE_DIR, "_ref", "rms")
RMS_K = 14              # DEFAULTS_TEST['*_rms_threshold'], lc3_conformance.py:129: the comparison is made at 14-bit resolution
ENG_THRESHOLD = 70      # DEFAULTS_TEST['*_eng_threshold'], lc3_conformance.py:126 (log10 of the summed squared sample difference)


def rms_between(frames_a, frames_b, fs, frame_ms, hrmode, channels=1, tmpdir=None, k=RMS_K):
    """The conformance script's default metric (compare_wav, lc3_conformance.py:610-619) between two bitstreams of one
    stream, both decoded by the reference decoder: the ETSI `rms` tool per channel at k-bit resolution.  Returns a dict:
    different samples, the overall RMS in dB and the maximum absolute difference (worst channel), the two thresholds
    the script derives from k and whether both hold.  Both decodes have the same length and no delay between them, so
    the script's cross-correlation alignment has nothing to do here."""
    import math
    import re
    import tempfile
    td = tmpdir or tempfile.mkdtemp(prefix="rms_")
    pcm = [ref_decode_stream(fr, fs, frame_ms, hrmode, channels) for fr in (frames_a, frames_b)]
    ndiff, rms, mx = 0, float("-inf"), 0.0
    for c in range(channels):
        paths = []
        for tag, x in zip("ab", pcm):
            p = os.path.join(str(td), "rms_%s%d.wav" % (tag, c))
            _write_wav16(p, x[c:c + 1], fs)
            paths.append(p)
        out = subprocess.run([RMS_TOOL, paths[0], paths[1], str(k)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
        m = re.search(r"different samples\s+: (\d+)", out)
        if not m:
            raise RuntimeError("rms tool output not understood: %r" % out[-400:])
        if int(m.group(1)):
            ndiff += int(m.group(1))
            rms = max(rms, float(re.search(r"Overall RMS value\s+: (\S+) dB ---", out).group(1)))
            mx = max(mx, float(re.search(r"Maximum difference\s+: (\S+) ---", out).group(1)))
    rms_thr = 20 * math.log10(2.0 ** (-k + 1) / 12 ** 0.5)
    diff_thr = 1.0 / 2 ** (k - 3)
    return {"different_samples": ndiff, "rms_db": rms, "max_abs_diff": mx, "rms_threshold_db": rms_thr, "max_abs_diff_threshold": diff_thr,
            "ok": rms <= rms_thr and mx <= diff_thr}


def energy_diff_between(frames_a, frames_b, fs, frame_ms, hrmode, channels=1):
    """The script's `eng` metric (energy_diff, lc3_conformance.py:586-600): log10 of the summed squared difference of the two
    decodes in 16-bit sample units, -inf when they are equal; the gate is <= ENG_THRESHOLD."""
    import math
    a, b = (ref_decode_stream(fr, fs, frame_ms, hrmode, channels).astype(np.float64) for fr in (frames_a, frames_b))
    e = float(((a - b) ** 2).sum())
    return math.log10(e) if e else float("-inf")


def compare_frames(got, want_list, fs, frame_ms, hrmode, channels=1):
    """got [B, T, stride], want_list[b] [T, nbytes_b].  Returns (differing frames, total, worst MLD over the streams that
    differ or None).  The gates assert the first number against the committed count (0); the MLD says whether a
    difference would still be inside the conformance tolerance (<= MLD_THRESHOLD) or is a real defect."""
    diff = tot = 0
    worst = None
    for b, w in enumerate(want_list):
        eq = (got[b, :, :w.shape[1]] == w).all(axis=1)
        tot += eq.size
        if not eq.all():
            diff += int((~eq).sum())
            if have_ref() and os.path.exists(MLD_TOOL):
                v = mld_between(got[b, :, :w.shape[1]], w, fs, frame_ms, hrmode, channels)
                worst = v if worst is None else max(worst, v)
    return diff, tot, worst
