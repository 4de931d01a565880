"""ctypes binding of the C-ABI library (include/lc3.h, include/lc3plus_batch.h).

Nothing here computes anything: every call goes to liblc3plus_hip.so.  If the library has not been built the
import of a symbol fails loudly -- there is no Python or CPU fallback path."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# every symbol include/lc3.h and include/lc3plus_batch.h declare
EXPORTS = [
    "lc3_version", "lc3_channels_supported", "lc3_samplerate_supported", "lc3_enc_get_size", "lc3_enc_init",
    "lc3_enc_set_frame_ms", "lc3_enc_set_hrmode", "lc3_enc_set_bitrate", "lc3_enc_set_bandwidth",
    "lc3_enc_get_input_samples", "lc3_enc_get_num_bytes", "lc3_enc_get_real_bitrate", "lc3_enc_get_delay",
    "lc3_enc_fl", "lc3_enc16", "lc3_enc24", "lc3_enc32", "lc3_enc_free_memory", "lc3_free_encoder_structs",
    "lc3plus_enc_batch_create", "lc3plus_enc_batch_destroy", "lc3plus_enc_batch_input_samples",
    "lc3plus_enc_batch_num_bytes", "lc3plus_enc_batch_stride", "lc3plus_enc_batch_set_bitrate",
    "lc3plus_enc_batch_set_bandwidth", "lc3plus_enc_batch_encode", "lc3plus_enc_batch_last_kernel_ms", "lc3plus_enc_batch_last_status", "lc3plus_enc_batch_last_records", "lc3plus_enc_batch_record_words", "lc3plus_enc_batch_set_input_ready", "lc3plus_enc_batch_state_size", "lc3plus_enc_batch_get_state", "lc3plus_enc_batch_set_state",
    "lc3plus_dec_batch_state_size", "lc3plus_dec_batch_get_state", "lc3plus_dec_batch_set_state",
    "lc3plus_enc_init", "lc3plus_enc_set_frame_ms", "lc3plus_enc_set_hrmode", "lc3plus_enc_set_bitrate",
    "lc3plus_enc16", "lc3plus_enc_get_size",
    "lc3_dec_get_size", "lc3_dec_init", "lc3_dec_set_frame_ms", "lc3_dec_set_hrmode", "lc3_dec_get_output_samples",
    "lc3_dec_get_delay", "lc3_dec_fl", "lc3_dec16", "lc3_dec24", "lc3_dec32", "lc3_dec_free_memory",
    "lc3_free_decoder_structs",
    "lc3plus_dec_batch_create", "lc3plus_dec_batch_destroy", "lc3plus_dec_batch_output_samples", "lc3plus_dec_batch_delay",
    "lc3plus_dec_batch_num_bytes", "lc3plus_dec_batch_set_num_bytes", "lc3plus_dec_batch_decode",
    "lc3plus_dec_batch_last_kernel_ms", "lc3plus_dec_batch_set_input_ready",
]


class LC3Error(RuntimeError):
    def __init__(self, code, what=""):
        super().__init__("LC3_Error %d %s" % (code, what))
        self.code = code


def lib_path():
    # LC3PLUS_HIP_LIB selects another build of the same library (diagnostic builds: stage timing / stage counts)
    return os.environ.get("LC3PLUS_HIP_LIB") or os.path.join(HERE, "liblc3plus_hip.so")


def load_library():
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise ImportError("liblc3plus_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                              "there is no fallback implementation")
        L = C.CDLL(p)
        L.lc3_enc_set_frame_ms.argtypes = [C.c_void_p, C.c_float]
        L.lc3plus_enc_batch_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                               C.POINTER(C.c_int), C.c_int]
        L.lc3plus_enc_batch_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                               C.c_void_p, C.c_int]
        L.lc3plus_enc_batch_encode_traced.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.lc3plus_enc_batch_last_kernel_ms.restype = C.c_float
        L.lc3plus_enc_batch_last_kernel_ms.argtypes = [C.c_void_p]
        L.lc3plus_enc_batch_last_status.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.lc3plus_enc_batch_last_records.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.lc3plus_enc_batch_set_input_ready.argtypes = [C.c_void_p, C.c_int]
        for nm in ("lc3plus_enc_batch", "lc3plus_dec_batch"):
            getattr(L, nm + "_state_size").argtypes = [C.c_void_p]; getattr(L, nm + "_state_size").restype = C.c_size_t
            getattr(L, nm + "_get_state").argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
            getattr(L, nm + "_set_state").argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        for f in ("lc3plus_enc_batch_destroy", "lc3plus_enc_batch_input_samples", "lc3plus_enc_batch_stride"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.lc3plus_enc_batch_num_bytes.argtypes = [C.c_void_p, C.c_int]
        L.lc3plus_enc_batch_set_bitrate.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.lc3plus_enc_batch_set_bandwidth.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.lc3_enc_fl.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.lc3_dec_set_frame_ms.argtypes = [C.c_void_p, C.c_float]
        L.lc3_dec_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.lc3_dec_set_hrmode.argtypes = [C.c_void_p, C.c_int]
        L.lc3_dec_fl.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int]
        for f in ("lc3_dec_get_output_samples", "lc3_dec_get_delay", "lc3_free_decoder_structs"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.lc3plus_dec_batch_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                               C.POINTER(C.c_int), C.c_int]
        L.lc3plus_dec_batch_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_int]
        L.lc3plus_dec_batch_decode_traced.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                      C.c_void_p, C.c_void_p]
        L.lc3plus_dec_batch_last_kernel_ms.restype = C.c_float
        L.lc3plus_dec_batch_last_kernel_ms.argtypes = [C.c_void_p]
        for f in ("lc3plus_dec_batch_destroy", "lc3plus_dec_batch_output_samples", "lc3plus_dec_batch_delay"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.lc3plus_dec_batch_num_bytes.argtypes = [C.c_void_p, C.c_int]
        L.lc3plus_dec_batch_set_input_ready.argtypes = [C.c_void_p, C.c_int]
        L.lc3plus_dec_batch_set_num_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _LIB = L
    return _LIB


class Batch:
    """n_streams independent encoders (lc3plus_enc_batch_*), state resident on the GPU between encode() calls."""

    def __init__(self, n_streams, samplerate, channels, frame_ms, hrmode, bitrates, device=-1):
        self.lib = load_library()
        br = (C.c_int * n_streams)(*[int(b) for b in bitrates])
        self.h = C.c_void_p()
        rc = self.lib.lc3plus_enc_batch_create(C.byref(self.h), n_streams, samplerate, channels, frame_ms, hrmode, br, device)
        if rc:
            raise LC3Error(rc, "lc3plus_enc_batch_create")
        self.n_streams, self.channels = n_streams, channels
        self.N = self.lib.lc3plus_enc_batch_input_samples(self.h)

    @property
    def stride(self):
        return self.lib.lc3plus_enc_batch_stride(self.h)

    def num_bytes(self, stream):
        return self.lib.lc3plus_enc_batch_num_bytes(self.h, stream)

    def set_bitrate(self, stream, bitrate):
        return self.lib.lc3plus_enc_batch_set_bitrate(self.h, stream, bitrate)

    def set_bandwidth(self, stream, bw):
        return self.lib.lc3plus_enc_batch_set_bandwidth(self.h, stream, bw)

    def encode(self, pcm, bitdepth=16):
        """pcm: host array [n_streams, T, channels, N] (or [n_streams, T, N] for mono) -> uint8 [n_streams, T, stride]."""
        pcm = np.ascontiguousarray(pcm)
        T = pcm.shape[1]
        stride = self.stride
        out = np.zeros((self.n_streams, T, stride), dtype=np.uint8)
        rc = self.lib.lc3plus_enc_batch_encode(self.h, pcm.ctypes.data, 0, bitdepth, T, out.ctypes.data, stride, 0, None, 1)
        if rc:
            raise LC3Error(rc, "lc3plus_enc_batch_encode")
        return out

    def encode_host(self, pcm, out, bitdepth=16):
        """Host buffers supplied by the caller (numpy views of pinned or pageable memory): pcm [n_streams, T, channels, N] ->
        out uint8 [n_streams, T, >= stride] in place; the library overlaps the copies with the kernels."""
        assert pcm.flags.c_contiguous and out.flags.c_contiguous and out.dtype == np.uint8
        T = pcm.shape[1]
        rc = self.lib.lc3plus_enc_batch_encode(self.h, pcm.ctypes.data, 0, bitdepth, T, out.ctypes.data, out.shape[2], 0, None, 1)
        if rc:
            raise LC3Error(rc, "lc3plus_enc_batch_encode(host)")
        return out

    def set_input_ready(self, ready=True):
        """Promise that the PCM of every following device-pointer call is complete when the call is made (include/lc3plus_batch.h):
        consecutive calls then overlap (the next call's frame-parallel kernels beside this call's sequential tail)."""
        rc = self.lib.lc3plus_enc_batch_set_input_ready(self.h, 1 if ready else 0)
        if rc:
            raise LC3Error(rc, "lc3plus_enc_batch_set_input_ready")

    def get_state(self):
        """The cross-frame state of all streams (opaque bytes): checkpoint for set_state() on a batch of the same configuration."""
        st = np.zeros(self.lib.lc3plus_enc_batch_state_size(self.h), dtype=np.uint8)
        rc = self.lib.lc3plus_enc_batch_get_state(self.h, st.ctypes.data, st.size)
        if rc:
            raise LC3Error(rc, "lc3plus_enc_batch_get_state")
        return st

    def set_state(self, st):
        st = np.ascontiguousarray(st, dtype=np.uint8)
        rc = self.lib.lc3plus_enc_batch_set_state(self.h, st.ctypes.data, st.size)
        if rc:
            raise LC3Error(rc, "lc3plus_enc_batch_set_state")

    def last_status(self, T):
        """uint8 [n_streams * channels, T]: LC3D_ENC_ST_* bits of the last call (0 = nothing the reference would assert on)."""
        st = np.zeros((self.n_streams * self.channels, T), dtype=np.uint8)
        n = self.lib.lc3plus_enc_batch_last_status(self.h, st.ctypes.data, st.size)
        if n < 0:
            raise LC3Error(1, "lc3plus_enc_batch_last_status")
        return st

    def last_records(self, T):
        """float32 [n_streams * channels, T, words]: the per-frame records of the last call of the pipelined path (integer fields: .view(np.int32))."""
        w = self.lib.lc3plus_enc_batch_record_words()
        rec = np.zeros((self.n_streams * self.channels, T, w), dtype=np.float32)
        n = self.lib.lc3plus_enc_batch_last_records(self.h, rec.ctypes.data, rec.size)
        if n == 0:
            raise LC3Error(1, "lc3plus_enc_batch_last_records: the last call did not take the pipelined path (it had at most 8 frames - 5 under the input-ready "
                              "promise - or was traced), so it left no records")
        if n != rec.size:
            raise LC3Error(1, "lc3plus_enc_batch_last_records (%d of %d words: T does not match the last call)" % (n, rec.size))
        return rec

    def encode_traced(self, pcm, bitdepth=16):
        pcm = np.ascontiguousarray(pcm)
        T = pcm.shape[1]
        stride = self.stride
        out = np.zeros((self.n_streams, T, stride), dtype=np.uint8)
        tsz = self.lib.lc3plus_trace_sizeof()
        traces = np.zeros((self.n_streams * self.channels * T, tsz), dtype=np.uint8)
        rc = self.lib.lc3plus_enc_batch_encode_traced(self.h, pcm.ctypes.data, bitdepth, T, out.ctypes.data, stride, traces.ctypes.data)
        if rc:
            raise LC3Error(rc, "lc3plus_enc_batch_encode_traced")
        return out, traces

    def encode_device(self, d_pcm_ptr, bitdepth, T, d_out_ptr, out_stride, hip_stream=None, sync=False):
        """Device-resident variant: raw device pointers (e.g. torch tensors' data_ptr())."""
        rc = self.lib.lc3plus_enc_batch_encode(self.h, C.c_void_p(d_pcm_ptr), 1, bitdepth, T, C.c_void_p(d_out_ptr), out_stride, 1,
                                               C.c_void_p(hip_stream) if hip_stream else None, 1 if sync else 0)
        if rc:
            raise LC3Error(rc, "lc3plus_enc_batch_encode(device)")

    def last_kernel_ms(self):
        return float(self.lib.lc3plus_enc_batch_last_kernel_ms(self.h))

    def close(self):
        if self.h:
            self.lib.lc3plus_enc_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Encoder:
    """Single-stream drop-in API (lc3_enc_*), mirrors how R/codec_exe.c:171-199,369-381 drives the reference."""

    def __init__(self, samplerate, channels=1, frame_ms=10.0, hrmode=0, bitrate=64000):
        L = self.lib = load_library()
        size = L.lc3_enc_get_size(samplerate, channels)
        if size <= 0:
            raise LC3Error(4, "lc3_enc_get_size")
        self.buf = C.create_string_buffer(size)
        self.p = C.cast(self.buf, C.c_void_p)
        self.channels = channels
        for rc, what in ((L.lc3_enc_init(self.p, samplerate, channels), "init"), (L.lc3_enc_set_frame_ms(self.p, frame_ms), "frame_ms"),
                         (L.lc3_enc_set_hrmode(self.p, hrmode), "hrmode"), (L.lc3_enc_set_bitrate(self.p, bitrate), "bitrate")):
            if rc:
                raise LC3Error(rc, what)
        self.N = L.lc3_enc_get_input_samples(self.p)
        self.nbytes = L.lc3_enc_get_num_bytes(self.p)

    def encode(self, planar, bitdepth=16):
        planar = np.ascontiguousarray(planar)
        ptrs = (C.c_void_p * self.channels)(*[planar[c].ctypes.data for c in range(self.channels)])
        out = np.zeros(self.nbytes, dtype=np.uint8)
        nb = C.c_int(0)
        rc = self.lib.lc3_enc_fl(self.p, ptrs, bitdepth, out.ctypes.data, C.byref(nb))
        if rc:
            raise LC3Error(rc, "lc3_enc_fl")
        return out[:nb.value]

    def close(self):
        if self.p:
            self.lib.lc3_free_encoder_structs(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DecBatch:
    """n_streams independent decoders (lc3plus_dec_batch_*), state resident on the GPU between decode() calls."""

    def __init__(self, n_streams, samplerate, channels, frame_ms, hrmode, num_bytes, device=-1):
        self.lib = load_library()
        nb = (C.c_int * n_streams)(*[int(b) for b in num_bytes])
        self.h = C.c_void_p()
        rc = self.lib.lc3plus_dec_batch_create(C.byref(self.h), n_streams, samplerate, channels, frame_ms, hrmode, nb, device)
        if rc:
            raise LC3Error(rc, "lc3plus_dec_batch_create")
        self.n_streams, self.channels = n_streams, channels
        self.N = self.lib.lc3plus_dec_batch_output_samples(self.h)

    def num_bytes(self, stream):
        return self.lib.lc3plus_dec_batch_num_bytes(self.h, stream)

    def set_num_bytes(self, stream, nbytes):
        return self.lib.lc3plus_dec_batch_set_num_bytes(self.h, stream, nbytes)

    def get_state(self):
        """The decoders' cross-frame state (opaque bytes): checkpoint for set_state() on a batch of the same configuration."""
        st = np.zeros(self.lib.lc3plus_dec_batch_state_size(self.h), dtype=np.uint8)
        rc = self.lib.lc3plus_dec_batch_get_state(self.h, st.ctypes.data, st.size)
        if rc:
            raise LC3Error(rc, "lc3plus_dec_batch_get_state")
        return st

    def set_state(self, st):
        st = np.ascontiguousarray(st, dtype=np.uint8)
        rc = self.lib.lc3plus_dec_batch_set_state(self.h, st.ctypes.data, st.size)
        if rc:
            raise LC3Error(rc, "lc3plus_dec_batch_set_state")

    def _prep(self, frames, bfi, bps):
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        S, T, stride = frames.shape
        assert S == self.n_streams
        if bfi is not None:
            bfi = np.ascontiguousarray(bfi, dtype=np.uint8)
            assert bfi.shape == (S, T)
        pcm = np.zeros((S, T, self.channels, self.N), dtype=np.int16 if bps == 16 else np.int32)
        status = np.zeros((S, T), dtype=np.uint8)
        return frames, T, stride, bfi, pcm, status

    def decode(self, frames, bfi=None, bps=16):
        """frames: uint8 [n_streams, T, stride]; bfi: optional [n_streams, T] -> (pcm [n_streams, T, channels, N], status)."""
        frames, T, stride, bfi, pcm, status = self._prep(frames, bfi, bps)
        rc = self.lib.lc3plus_dec_batch_decode(self.h, frames.ctypes.data, 0, stride, bfi.ctypes.data if bfi is not None else None, T,
                                               pcm.ctypes.data, 0, bps, status.ctypes.data, None, 1)
        if rc:
            raise LC3Error(rc, "lc3plus_dec_batch_decode")
        return pcm, status

    def decode_traced(self, frames, bfi=None, bps=16):
        frames, T, stride, bfi, pcm, status = self._prep(frames, bfi, bps)
        tsz = self.lib.lc3plus_dec_trace_sizeof()
        traces = np.zeros((self.n_streams * self.channels * T, tsz), dtype=np.uint8)
        rc = self.lib.lc3plus_dec_batch_decode_traced(self.h, frames.ctypes.data, stride, bfi.ctypes.data if bfi is not None else None, T,
                                                      pcm.ctypes.data, bps, status.ctypes.data, traces.ctypes.data)
        if rc:
            raise LC3Error(rc, "lc3plus_dec_batch_decode_traced")
        return pcm, status, traces

    def set_input_ready(self, ready=True):
        """lc3plus_dec_batch_set_input_ready: the frames of every following device-pointer call are complete on the device when the call is made."""
        rc = self.lib.lc3plus_dec_batch_set_input_ready(self.h, 1 if ready else 0)
        if rc:
            raise LC3Error(rc, "lc3plus_dec_batch_set_input_ready")

    def decode_device(self, d_frames_ptr, in_stride, T, d_pcm_ptr, bps=16, hip_stream=None, sync=False):
        """Device-resident variant: raw device pointers, no bad-frame flags."""
        rc = self.lib.lc3plus_dec_batch_decode(self.h, C.c_void_p(d_frames_ptr), 1, in_stride, None, T, C.c_void_p(d_pcm_ptr), 1, bps, None,
                                               C.c_void_p(hip_stream) if hip_stream else None, 1 if sync else 0)
        if rc:
            raise LC3Error(rc, "lc3plus_dec_batch_decode(device)")

    def last_kernel_ms(self):
        return float(self.lib.lc3plus_dec_batch_last_kernel_ms(self.h))

    def close(self):
        if self.h:
            self.lib.lc3plus_dec_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Decoder:
    """Single-stream drop-in API (lc3_dec_*), used exactly as R/codec_exe.c uses the reference decoder."""

    def __init__(self, samplerate, channels=1, frame_ms=10.0, hrmode=0):
        self.lib = load_library()
        size = self.lib.lc3_dec_get_size(samplerate, channels)
        if size <= 0:
            raise LC3Error(1, "lc3_dec_get_size")
        self.buf = C.create_string_buffer(size)
        self.h = C.cast(self.buf, C.c_void_p)
        self.channels = channels
        for rc, what in ((self.lib.lc3_dec_init(self.h, samplerate, channels, 0), "lc3_dec_init"),
                         (self.lib.lc3_dec_set_frame_ms(self.h, frame_ms), "lc3_dec_set_frame_ms"),
                         (self.lib.lc3_dec_set_hrmode(self.h, hrmode), "lc3_dec_set_hrmode")):
            if rc:
                raise LC3Error(rc, what)
        self.N = self.lib.lc3_dec_get_output_samples(self.h)

    def decode(self, frame, bfi_ext=0, bps=16):
        """frame: bytes of one stream-frame -> (planar samples [channels, N], LC3_Error code 0 or 2)."""
        data = np.frombuffer(bytes(frame), dtype=np.uint8).copy() if len(frame) else np.zeros(1, dtype=np.uint8)
        out = np.zeros((self.channels, self.N), dtype=np.int16 if bps == 16 else np.int32)
        ptrs = (C.c_void_p * self.channels)(*[out[c].ctypes.data for c in range(self.channels)])
        rc = self.lib.lc3_dec_fl(self.h, data.ctypes.data, len(frame), ptrs, bps, bfi_ext)
        if rc not in (0, 2):
            raise LC3Error(rc, "lc3_dec_fl")
        return out, rc

    def close(self):
        if self.h:
            self.lib.lc3_free_decoder_structs(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
