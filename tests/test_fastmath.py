"""audio_codec_amd/csrc/lc3_fastmath.h (the device's log2 / log10 / 2^x of a float, evaluated in double) against glibc on the host: the exhaustive run is
tools/fastmath_check.c with stride 1 (6.5e9 evaluations, recorded in profiles/r04_fastmath_check.txt); here a strided sample of every exponent range, the special
arguments, and the committed tables against their generator."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "audio_codec_amd", "csrc")


def test_tables_are_what_the_generator_writes(tmp_path):
    out = tmp_path / "tables.h"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_fastmath.py"), str(out)])
    assert out.read_text() == open(os.path.join(CSRC, "lc3_fastmath_tables.h")).read()


def test_strided_sample_equals_glibc(tmp_path):
    exe = tmp_path / "fastmath_check"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-pthread", "-I" + CSRC, os.path.join(ROOT, "tools", "fastmath_check.c"), "-o", str(exe), "-lm"])
    r = subprocess.run([str(exe), "4", "1021"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)       # every 1021st bit pattern: ~2e6 arguments per function
    assert r.returncode == 0, r.stdout
    lines = [l for l in r.stdout.splitlines() if "arguments" in l]
    assert len(lines) == 4 and all(" 0 differ" in l for l in lines), r.stdout


def host_lib(tmp_path):
    so = tmp_path / "fastmath_host.so"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-shared", "-fPIC", "-I" + CSRC, os.path.join(ROOT, "tools", "fastmath_host.c"), "-o", str(so), "-lm"])
    L = C.CDLL(str(so))
    L.lc3m_host_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_long]
    return L


def sample_arguments(seed=5):
    """floats the encoder's calls see and the edges around them: every binade, values next to 1 and to powers of two, subnormals, scale-factor sized exponents"""
    rng = np.random.default_rng(seed)
    bits = rng.integers(1, 0x7F800000, size=400000, dtype=np.int64).astype(np.uint32)
    pos = bits.view(np.float32)
    near1 = (1.0 + np.arange(-4096, 4096) * 2.0 ** -23).astype(np.float32)
    pw = np.concatenate([np.float32(2.0) ** np.arange(-140, 128).astype(np.float32) * s for s in (np.float32(1), np.nextafter(np.float32(1), np.float32(0)), np.nextafter(np.float32(1), np.float32(2)))]).astype(np.float32)
    logs = np.concatenate([pos, near1, pw, np.array([1e-45, 1.1754944e-38, 3.4028235e38, 2.0 ** -31, 1.1920929e-07], np.float32)])
    ex = np.concatenate([rng.uniform(-150, 130, 300000), rng.uniform(-12, 12, 200000), np.arange(-160, 161), np.arange(-160, 160) + 0.5,
                         np.array([0.0, -0.0, 1e-30, -1e-30, 127.99999, -149.5, 128.0, -126.0])]).astype(np.float32)
    return logs, ex


def test_header_equals_glibc_on_the_sample(tmp_path):
    L = host_lib(tmp_path)
    logs, ex = sample_arguments()
    for kind, x in ((0, logs), (1, logs), (2, ex)):
        a = np.zeros_like(x); b = np.zeros_like(x)
        L.lc3m_host_eval(kind, x.ctypes.data, a.ctypes.data, x.size)
        L.lc3m_host_eval(kind + 3, x.ctypes.data, b.ctypes.data, x.size)
        assert (a.view(np.uint32) == b.view(np.uint32)).all(), (kind, x[a.view(np.uint32) != b.view(np.uint32)][:5])
    # arguments that take the library path on both sides
    sp = np.array([0.0, -0.0, -1.0, np.inf, -np.inf, np.nan, 1e30, -1e30, 2000.0, -2000.0], np.float32)
    for kind in (0, 1, 2):
        a = np.zeros_like(sp); b = np.zeros_like(sp)
        L.lc3m_host_eval(kind, sp.ctypes.data, a.ctypes.data, sp.size); L.lc3m_host_eval(kind + 3, sp.ctypes.data, b.ctypes.data, sp.size)
        assert ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all(), kind
