"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (the reference's makefile has the same switches,
R/makefile:51-72).  CPU only - never on the GPU.  `make -C oracle asan` builds oracle/_asan/liblc3_oracle{,_pm}.so; the driver runs in a
child process with the sanitizer runtimes preloaded (the Python interpreter itself is not instrumented)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_restatement_is_clean_under_asan_and_ubsan():
    asan, ubsan = _lib("libasan.so"), _lib("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes not installed")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_driver.py")], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "sanitizers clean" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


def test_c_baseline_driver_matches_the_oracle(tmp_path):
    """oracle/cpu_bench.c (bench.py's CPU baseline) encodes the PCM file it is given exactly like the per-frame API: its optional output
    file against oracle_encode_streams; both builds (restatement, and the compiled reference where oracle/_ref exists)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from lc3_harness import oracle_encode_streams, synth_pcm
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "restatement"])
    S, T = 6, 10
    pcm = synth_pcm(S, T, 480, 48000, seed=77)
    f = tmp_path / "pcm.bin"; pcm.tofile(f)
    want = oracle_encode_streams(pcm, 48000, 10.0, 0, [64000] * S)
    exes = [os.path.join(ROOT, "oracle", "cpu_bench_port")]
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "cpu_bench_ref")):
        exes.append(os.path.join(ROOT, "oracle", "_ref", "cpu_bench_ref"))
    for exe in exes:
        o = tmp_path / "out.bin"
        r = subprocess.run([exe, "enc", "48000", "10", "0", "1", "64000", str(S), str(T), "3", str(f), str(o)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        frames, sec = r.stdout.split()[:2]
        assert int(frames) == S * T and float(sec) > 0
        got = np.fromfile(o, np.uint8).reshape(S, T, 80)
        assert all((got[i] == want[i]).all() for i in range(S)), exe
