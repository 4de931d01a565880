"""Builds liblc3plus_hip.so in-tree (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose=False):
    cmd = ["make", "-C", os.path.join(HERE, "csrc")]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return os.path.join(HERE, "liblc3plus_hip.so")
