"""The product's host C code (audio_codec_amd/csrc/lc3_host.c: configuration and bit-budget derivation, prime-factor DFT plan builder, batch
bookkeeping, API error paths) under AddressSanitizer + UndefinedBehaviorSanitizer, as the reference's makefile offers for its own sources
(R/makefile:51-72).  CPU only: `make -C audio_codec_amd/csrc asan` links the sanitized host object with the product's kernel objects, and
tests/test_host_api_cpu.py runs against that library in a child process with the gcc sanitizer runtimes preloaded."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_host_api_is_clean_under_asan_and_ubsan():
    asan, ubsan = _lib("libasan.so"), _lib("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes not installed")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "audio_codec_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "audio_codec_amd", "csrc"), "asan"])
    lib = os.path.join(ROOT, "audio_codec_amd", "_asan", "liblc3plus_hip.so")
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               LC3PLUS_HIP_LIB=lib)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_host_api_cpu.py"), "-x", "-q", "-p", "no:cacheprovider"],
                         capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0 and " passed" in out.stdout and "runtime error" not in out.stderr, (out.stdout[-1500:], out.stderr[-3000:])
