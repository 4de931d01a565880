"""The N>1 path on CPU: two gloo ranks own disjoint stream blocks (audio_codec_amd.sharding), encode them independently with the
CPU oracle standing in for the GPU step, and the union equals the unsharded result -- there is no exchange step to get wrong.
Also exercises bench.py's barrier + max-over-ranks timing reduction."""
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
from audio_codec_amd.sharding import stream_block
from lc3_harness import synth_pcm, oracle_encode_streams
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
TOTAL, T = 10, 6
first, last = stream_block(rank, world, TOTAL)
pcm = synth_pcm(TOTAL, T, 480, 48000, seed=77)[first:last]          # every rank generates only what it owns
dist.barrier(); t0 = time.perf_counter()
frames = oracle_encode_streams(pcm, 48000, 10.0, 0, [64000] * (last - first))
dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
dist.barrier()
dist.all_reduce(dt, op=dist.ReduceOp.MAX)                            # bench.py: max over ranks
digests = [None] * world
dist.all_gather_object(digests, [hashlib.md5(f.tobytes()).hexdigest() for f in frames])
if rank == 0:
    print("DIGESTS", ",".join(d for per in digests for d in per), "MAXT", float(dt.item()) > 0)
dist.destroy_process_group()
'''


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
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("DIGESTS")][0]
    got = line.split()[1].split(",")
    pcm = synth_pcm(10, 6, 480, 48000, seed=77)
    want = [hashlib.md5(f.tobytes()).hexdigest() for f in oracle_encode_streams(pcm, 48000, 10.0, 0, [64000] * 10)]
    assert got == want
    assert line.split()[3] == "True"
