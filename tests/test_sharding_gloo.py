"""The N>1 path on CPU: two gloo ranks own disjoint stream blocks (audio_codec_amd.sharding), encode them independently with the
CPU oracle standing in for the GPU step, and the union equals the unsharded result -- there is no exchange step to get wrong.
The ranks are started by bench.py's own launcher (bench.launch_ranks: what `python bench.py --gpus N` does) and time their steps
with bench.py's own protocol (bench.timed_steps: warm-up, barrier + sync on both sides, max over ranks), with the oracle as the
step and gloo as the backend."""
import hashlib
import os
import socket
import subprocess
import sys

import numpy as np

from audio_codec_amd.sharding import owner_of, stream_block

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, hashlib, time
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import bench                                                          # the repo's bench.py: its rank / timing code is what runs here
from audio_codec_amd.sharding import stream_block
from lc3_harness import synth_pcm, oracle_encode_streams
rank, world, local = bench.rank_env()
dist.init_process_group("gloo")
assert (rank, world) == (dist.get_rank(), dist.get_world_size())
TOTAL, T = 10, 6
first, last = stream_block(rank, world, TOTAL)
pcm = synth_pcm(TOTAL, T, 480, 48000, seed=77)[first:last]          # every rank generates only what it owns
frames = []
calls = [0]
def step():
    calls[0] += 1
    frames[:] = oracle_encode_streams(pcm, 48000, 10.0, 0, [64000] * (last - first))
    if rank == 1: time.sleep(0.05)                                    # the slower rank must set the reported time
wall = bench.timed_steps(step, 2, 1, lambda: None, dist)
assert calls[0] == 3
own = torch.tensor([0.1 if rank == 1 else 0.0], dtype=torch.float64)
digests = [None] * world
dist.all_gather_object(digests, [hashlib.md5(f.tobytes()).hexdigest() for f in frames])
if rank == 0:
    print("DIGESTS", ",".join(d for per in digests for d in per), "MAXT", wall >= 0.1)
dist.destroy_process_group()
'''


WORKER_C3 = r'''
import os, sys, hashlib, ctypes as C
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import bench
rank, world, local = bench.rank_env()
dist.init_process_group("gloo")
plan = bench.rank_plan("c3", rank, world, streams=%(streams)d, frames=%(frames)d)          # bench.py's own plan of BASELINE configs[2], reduced
pcm = bench.synth_pcm_device(torch, plan["B"], plan["T"], plan["ch"], plan["n"], plan["fs"], torch.device("cpu"), seed=plan["seed"], first_stream=plan["first"]).numpy()
L = C.CDLL(os.path.join(%(root)r, "oracle", "liblc3_oracle_pm.so"))
L.lc3o_encode_batch16_ch.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
out = np.zeros((plan["B"], plan["T"], 160), np.uint8)
br = np.asarray(plan["br"], np.int32)
def step():
    assert L.lc3o_encode_batch16_ch(plan["fs"], plan["ms"], plan["hr"], plan["ch"], plan["B"], plan["T"], br.ctypes.data, pcm.ctypes.data, out.ctypes.data, 160) == 0
wall = bench.timed_steps(step, 1, 1, lambda: None, dist)
rows = [None] * world
dist.all_gather_object(rows, (plan["first"], plan["last"], plan["br"], hashlib.md5(out.tobytes()).hexdigest(), int(out.any(axis=2).sum())))
if rank == 0: print("C3ROWS", repr(rows))
dist.destroy_process_group()
'''


def test_c3_rank_plan_two_gloo_ranks(tmp_path):
    """BASELINE configs[2] (262 144 stereo frames over 8 GPUs, no collectives) through bench.py's own rank code on two gloo ranks with the oracle as the
    step: bench.rank_plan gives every rank its block of the job's streams, its bitrates and its PCM seed; a rank's output depends on its block only (the
    same block computed alone gives the same bytes); and the full-size plan is 8 x 2048 stereo streams x 16 frames."""
    import ast, ctypes as C
    sys.path.insert(0, ROOT)
    import bench, torch
    full = [bench.rank_plan("c3", r, 8) for r in range(8)]
    assert [p["first"] for p in full] == [2048 * r for r in range(8)] and all(p["B"] == 2048 and p["T"] == 16 and p["ch"] == 2 for p in full)
    assert full[0]["job_streams"] == 16384 and full[0]["job_channel_frames_per_step"] == 2 * 262144 and set(full[3]["br"]) == {128000}
    script = tmp_path / "worker_c3.py"
    script.write_text(WORKER_C3 % {"root": ROOT, "streams": 3, "frames": 5})
    drv = tmp_path / "driver_c3.py"
    drv.write_text("import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.launch_ranks(2, [], script=%r))\n" % (ROOT, str(script)))
    out = subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = ast.literal_eval([l for l in out.stdout.splitlines() if l.startswith("C3ROWS")][0][7:])
    assert [(r[0], r[1]) for r in rows] == [(0, 3), (3, 6)] and all(r[4] == 3 * 5 for r in rows)          # every stream-frame non-empty
    L = C.CDLL(os.path.join(ROOT, "oracle", "liblc3_oracle_pm.so"))
    L.lc3o_encode_batch16_ch.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    for r in range(2):
        p = bench.rank_plan("c3", r, 2, streams=3, frames=5)
        pcm = bench.synth_pcm_device(torch, 3, 5, 2, 480, 48000, torch.device("cpu"), seed=p["seed"], first_stream=p["first"]).numpy()
        o = np.zeros((3, 5, 160), np.uint8); br = np.asarray(p["br"], np.int32)
        assert L.lc3o_encode_batch16_ch(48000, 10.0, 0, 2, 3, 5, br.ctypes.data, pcm.ctypes.data, o.ctypes.data, 160) == 0
        assert hashlib.md5(o.tobytes()).hexdigest() == rows[r][3] and rows[r][2] == p["br"]


def test_stream_block_partition():
    for total in (1, 7, 8, 4096, 262144):
        for world in (1, 2, 3, 8):
            blocks = [stream_block(r, world, total) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
            for s in (0, total // 2, total - 1):
                r = owner_of(s, world, total)
                assert blocks[r][0] <= s < blocks[r][1]


def test_two_rank_gloo_matches_unsharded(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from lc3_harness import oracle_encode_streams, synth_pcm
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    # bench.launch_ranks() is the code path of `python bench.py --gpus 2`; run it in a child so that its output can be captured
    drv = tmp_path / "driver.py"
    drv.write_text("import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.launch_ranks(2, [], script=%r))\n" % (ROOT, str(script)))
    out = subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("DIGESTS")][0]
    got = line.split()[1].split(",")
    pcm = synth_pcm(10, 6, 480, 48000, seed=77)
    want = [hashlib.md5(f.tobytes()).hexdigest() for f in oracle_encode_streams(pcm, 48000, 10.0, 0, [64000] * 10)]
    assert got == want
    assert line.split()[3] == "True"


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 8` must not quietly report n_gpus: 1: without 8 visible devices it exits non-zero before touching the GPU."""
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 3 and "device(s) visible" in out.stderr and not out.stdout.strip(), (out.returncode, out.stderr[-300:])
