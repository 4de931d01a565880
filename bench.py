#!/usr/bin/env python3
"""bench.py -- Mframes/s of the LC3plus encode hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: B independent streams advanced by T frames each.  Default
workload c1 = BASELINE.json configs[1] (48 kHz / 10 ms / 64 kbps mono, 4096 streams per GPU; T = 64 as in SURVEY 8(d)).
PCM is synthetic and already resident in HBM when the timed region starts; bitstreams are written to HBM.

Multi-GPU: streams are independent, so every rank (one process per GPU) owns its own block of streams
(audio_codec_amd.sharding.stream_block) for the whole run; torch.distributed (RCCL) is used only for the barrier and the
max-over-ranks time -- no data-path collective, scaling is weak.  `python bench.py --gpus N` with WORLD_SIZE unset starts
the N ranks itself (a child `python -m torch.distributed.run`, before anything in this process touches the GPU) and exits
non-zero when fewer than N devices are visible; under an external torchrun it is a rank.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RATES12 = [16000, 24000, 32000, 48000, 64000, 80000, 96000, 128000, 160000, 192000, 256000, 320000]
# name: (fs, frame_ms, hrmode, channels, N, per-stream total bitrates (cycled), frames per step, streams per GPU, what)
WORKLOADS = {
    "c1": (48000, 10.0, 0, 1, 480, [64000], 64, 4096, "BASELINE configs[1]: independent mono streams, 48kHz/10ms/64kbps (pipelined kernels: one frame per lane / four frames per wave / one or two channel-streams per wave)"),
    # configs[2]: 262 144 stereo frames over 8 GPUs = 16 384 stereo streams x 16 frames -> 2 048 stereo streams per GPU
    "c3": (48000, 10.0, 0, 2, 480, [128000], 16, 2048, "BASELINE configs[2]: stereo streams, 48kHz/10ms/128kbps (80 B per channel); 8 GPUs x 2048 streams x 16 frames = 262144 stereo frames"),
    "c4": (96000, 2.5, 1, 1, 240, [256000], 256, 4096, "BASELINE configs[3]: 96kHz/2.5ms high-resolution 256kbps mono"),
    "c5": (48000, 10.0, 0, 1, 480, RATES12, 64, 4096, "BASELINE configs[4]: mixed-bitrate batch 16-320 kbps, 48kHz/10ms mono"),
    # the large kernel layout (N = 960): 96 kHz / 10 ms high-resolution
    "c96": (96000, 10.0, 1, 1, 960, [256000], 32, 4096, "96kHz/10ms high-resolution 256kbps mono (large LDS layout of the kernels)"),
    # decoder: the C1 / C5 bitstreams, produced on the GPU just before, decoded back to 16-bit PCM
    "d1": (48000, 10.0, 0, 1, 480, [64000], 64, 4096, "decoder: the c1 bitstreams back to 16-bit PCM"),
    "d5": (48000, 10.0, 0, 1, 480, RATES12, 64, 4096, "decoder: the c5 bitstreams back to 16-bit PCM"),
}
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: 8.0 TB/s spec
PROFILE_ROUND = "r04"


# ----------------------------------------------------------------------------------------------------------------
# the launch / timing protocol (shared with tests/test_sharding_gloo.py, which runs it on gloo with a stub step)
# ----------------------------------------------------------------------------------------------------------------
def launch_ranks(n, argv, script=None, extra_env=None):
    """Start n ranks of `script` (this file by default) under torch.distributed.run on this node and return its exit code.
    Must be called before the calling process has initialised the GPU."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if extra_env: env.update(extra_env)
    return subprocess.call(cmd, env=env)


def timed_steps(step, steps, warmup, sync, dist=None, device=None):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + sync on both sides; returns the MAX over ranks of the wall
    time in seconds.  `sync()` drains the device (a no-op for the CPU stub); `dist` is torch.distributed or None."""
    import torch
    for _ in range(warmup):
        step()
    sync()
    if dist is not None: dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    timed_steps.enqueue = time.perf_counter() - t0        # host time to queue the K steps (the GPU runs behind it)
    sync()
    if dist is not None: dist.barrier()
    sync()
    wall = time.perf_counter() - t0
    timed_steps.local_wall = wall                        # this rank's own time (the line reports all of them)
    if dist is not None:
        tw = torch.tensor([wall], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    return wall


def rank_plan(workload, rank, world, streams=0, frames=0):
    """What one rank of an N-rank run works on (weak scaling: every rank owns B of the B x N streams of the job, a contiguous block; no exchange step):
    the block, its per-stream bitrates (cycled over the GLOBAL stream index) and the seed of its synthetic PCM.  main() and tests/test_sharding_gloo.py
    (two gloo ranks, the oracle as the step) both run from this."""
    from audio_codec_amd.sharding import stream_block
    fs, ms, hr, ch, n, rates, T, B, what = WORKLOADS[workload]
    if frames: T = frames
    if streams: B = streams
    if os.environ.get("LC3_BENCH_RATES"):           # diagnostic sweeps (tools/): the workload's shape at other bitrates; the line's config.bytes_per_frame says what ran
        rates = [int(v) for v in os.environ["LC3_BENCH_RATES"].split(",")]
    first, last = stream_block(rank, world, B * world)
    assert last - first == B
    return {"fs": fs, "ms": ms, "hr": hr, "ch": ch, "n": n, "T": T, "B": B, "what": what, "first": first, "last": last,
            "br": [rates[(first + i) % len(rates)] for i in range(B)], "seed": 1234 + first,
            "job_streams": B * world, "job_channel_frames_per_step": B * world * T * ch}


def rank_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


# ----------------------------------------------------------------------------------------------------------------
def synth_pcm_device(torch, B, T, ch, n, fs, dev, seed, first_stream=0):
    """Deterministic synthetic PCM on the device [B, T, ch, n] int16, SURVEY 8(d)'s recipe as tests/lc3_harness.py::synth_pcm states it:
    per channel-stream 3 sinusoids (80 Hz ... 0.4 fs, amplitude 0.02 ... 0.25 FS) + coloured noise at -30 dBFS, a 20 dB step transient every 37
    frames, the whole at -20 dB; one stream in four strongly periodic (11 harmonics of a 90 ... 380 Hz pitch: the LTPF path); of every 64
    streams one is digital silence and one full-scale white noise (kinds by global stream index, so every rank sees its share)."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    S, m = B * ch, T * n
    t = torch.arange(m, device=dev, dtype=torch.float32) / fs
    x = torch.zeros(S, m, device=dev, dtype=torch.float32)
    for _ in range(3):
        f = 80.0 + torch.rand(S, 1, device=dev, generator=g) * (0.4 * fs - 80.0)
        a = (0.02 + 0.23 * torch.rand(S, 1, device=dev, generator=g)) * 32767.0
        ph = torch.rand(S, 1, device=dev, generator=g) * 6.2831853
        x += a * torch.sin(6.2831853 * f * t[None, :] + ph)
    kind = (torch.arange(S, device=dev) // ch + first_stream) % 64
    per = (kind % 4 == 1).to(torch.float32)[:, None]
    f0 = 90.0 + torch.rand(S, 1, device=dev, generator=g) * 290.0
    for h in range(1, 12):
        x += per * (0.12 / h) * 32767.0 * torch.sin(6.2831853 * f0 * h * t[None, :] + h)
    noise = torch.randn(S, m, device=dev, generator=g)
    noise = 0.5 * noise + 0.3 * torch.roll(noise, 1, 1) + 0.15 * torch.roll(noise, 2, 1) + 0.05 * torch.roll(noise, 3, 1)
    x += noise * (32767.0 * 10 ** (-30 / 20))
    env = torch.ones(m, device=dev)
    for k in range(0, T, 37):
        a0 = k * n + n // 3
        env[a0:a0 + n // 2] = 10.0
    x = x * env[None, :] * 0.1
    full = (torch.rand(S, m, device=dev, generator=g) * 65535.0 - 32768.0)
    x = torch.where((kind == 62)[:, None], full, x)
    x = torch.where((kind == 63)[:, None], torch.zeros_like(x), x)
    x = x.round().clamp(-32768, 32767).to(torch.int16)
    return x.reshape(B, ch, T, n).permute(0, 2, 1, 3).contiguous()


def parity_sample(pcm, out, nbl, br, fs, ms, hr, ch, calls, sample):
    """The timed launches' own bytes against the CPU oracle (checker only): `sample` streams, every call of the run (warm-up and timed: the
    encoder's state runs on from call to call over the same PCM), last call's frames compared byte for byte."""
    import ctypes as C
    import numpy as np
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liblc3_oracle_pm.so"))
    lib.lc3o_encode_batch16_ch.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    T = pcm.shape[1]
    frames = differ = 0
    for i in sample:
        x = pcm[i:i + 1].cpu().numpy()                                      # [1, T, ch, n]
        x = np.ascontiguousarray(np.concatenate([x] * calls, axis=1))
        stride = int(out.shape[2])
        want = np.zeros((1, T * calls, stride), np.uint8)
        rate = np.array([br[i]], np.int32)
        rc = lib.lc3o_encode_batch16_ch(fs, ms, hr, ch, 1, T * calls, rate.ctypes.data, x.ctypes.data, want.ctypes.data, stride)
        if rc: raise RuntimeError("oracle rc %d" % rc)
        got = out[i].cpu().numpy()
        nb = nbl[i]
        d = (got[:, :nb] != want[0, -T:, :nb]).any(axis=1)
        frames += T * ch; differ += int(d.sum()) * ch
    return {"frames": frames, "differ": differ, "streams": len(sample),
            "note": "oracle/liblc3_oracle_pm.so over all %d calls of the run on the sampled streams (state runs on), the last call's frames compared with "
                    "the timed launches' output buffer byte for byte" % calls}


def parity_sample_dec(frames, back, nbl, fs, ms, hr, ch, calls, sample):
    """The decoder's timed launches against the CPU oracle decoder (checker only): `sample` streams, every call of the run decoded in order (the
    decoder's state runs on from call to call over the same frames), the last call's PCM compared sample for sample."""
    import ctypes as C
    import numpy as np
    L = C.CDLL(os.path.join(ROOT, "oracle", "liblc3_oracle_pm.so"))
    L.lc3o_dec_set_frame_ms.argtypes = [C.c_void_p, C.c_float]
    L.lc3o_dec_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int]
    T = frames.shape[1]
    n = differ = 0
    for i in sample:
        buf = C.create_string_buffer(L.lc3o_dec_sizeof() + 8); h = C.cast(buf, C.c_void_p)
        if L.lc3o_dec_init(h, fs, ch) or L.lc3o_dec_set_frame_ms(h, ms) or L.lc3o_dec_set_hrmode(h, hr): raise RuntimeError("oracle decoder setup")
        N = L.lc3o_dec_get_output_samples(h)
        fr = np.ascontiguousarray(frames[i, :, :nbl[i]].cpu().numpy())
        got = back[i].cpu().numpy()                                          # [T, ch, N]
        o = np.zeros((ch, N), np.int16)
        ptrs = (C.c_void_p * ch)(*[o[c].ctypes.data for c in range(ch)])
        for k in range(calls):
            for t in range(T):
                rc = L.lc3o_dec_frame(h, fr[t].ctypes.data, int(nbl[i]), ptrs, 16, 0)
                if rc not in (0, 2): raise RuntimeError("oracle decoder rc %d" % rc)
                if k == calls - 1:
                    n += ch; differ += int((got[t] != o).any(axis=1).sum())
    return {"frames": n, "differ": differ, "streams": len(sample),
            "note": "oracle/liblc3_oracle_pm.so's decoder over all %d calls of the run on the sampled streams (state runs on), the last call's PCM compared with "
                    "the timed launches' output buffer sample for sample" % calls}


def gpu_numa_cpus(local):
    """The CPUs of the NUMA node the local_rank-th AMD GPU hangs off (sysfs only: nothing here touches the GPU), or None.  Devices in PCI address order,
    which is HIP's enumeration order unless HIP_VISIBLE_DEVICES reorders them (then: None)."""
    import glob
    if os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES"): return None
    try:
        devs = sorted(d for d in glob.glob("/sys/bus/pci/drivers/amdgpu/0000:*") if os.path.exists(os.path.join(d, "numa_node")))
        node = int(open(os.path.join(devs[local], "numa_node")).read())
        if node < 0: return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-"); cpus.update(range(int(a), int(b or a) + 1))
        return node, sorted(cpus)
    except (OSError, ValueError, IndexError):
        return None


def cpu_baseline(kind_dir, mode, fs, ms, hr, ch, rate_or_nbytes, sample, S, T, tag):
    """The compiled ETSI reference (oracle/_ref/cpu_bench_ref, kind 'reference') or the C restatement (oracle/cpu_bench_port,
    kind 'port') from a C driver (oracle/cpu_bench.c) over the SAME data the GPU ran on: `sample` = the first S streams x T
    frames of the bench's own device buffer.  Two figures: all host cores (one thread per core over disjoint streams) and
    one thread on a quarter of the sample."""
    ref = os.path.join(ROOT, "oracle", "_ref", "cpu_bench_ref")
    exe, kind = (ref, "reference") if os.path.exists(ref) else (os.path.join(ROOT, "oracle", "cpu_bench_port"), "port")
    cores = max(1, min(os.cpu_count() or 1, 64))
    with tempfile.NamedTemporaryFile(suffix=".bin", dir=kind_dir, delete=False) as f:
        f.write(sample.tobytes()); path = f.name
    try:
        def run(streams, threads):
            out = subprocess.run([exe, mode, str(fs), str(ms), str(hr), str(ch), str(rate_or_nbytes), str(streams), str(T), str(threads), path],
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            if out.returncode: raise RuntimeError(out.stderr[-300:])
            fr, sec = out.stdout.split()[:2]
            return int(fr), float(sec)
        fr, sec = run(S, cores)
        fr1, sec1 = run(max(1, S // 16), 1)
    finally:
        os.unlink(path)
    return {"value": round(fr / sec / 1e6, 6), "unit": "Mframes/s", "cores": cores, "kind": kind,
            "single_thread": {"value": round(fr1 / sec1 / 1e6, 6), "us_per_frame": round(sec1 / fr1 * 1e6, 2), "frames": fr1},
            "sample": "%s: the first %d streams x %d frames of the GPU run's own %s (same bytes), C driver oracle/cpu_bench.c, %d threads over disjoint "
                      "streams, %.2f s wall for the threaded region (%.1f s of CPU work); single thread: %d streams, %.2f s"
                      % (tag, S, T, "PCM" if mode == "enc" else "bitstreams", cores, sec, sec * cores, max(1, S // 16), sec1)}


def source_hash():
    """sha256 over the kernel / host sources of the library: committed counters (profiles/*_counters.json) carry it, and the bench line only
    quotes them when it matches the tree that is running."""
    import glob, hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "audio_codec_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.inc")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.c"))):
        with open(f, "rb") as fh: h.update(os.path.basename(f).encode()); h.update(fh.read())
    return h.hexdigest()[:16]


def committed_profile(workload, B, T):
    """Numbers that come from separate rocprofv3 passes (committed under profiles/, collected with tools/gpu_round.sh on the same
    command): HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE) and wave-level VALU instructions per launch.  Only valid for the
    launch shape they were collected on, otherwise empty."""
    for name in (PROFILE_ROUND + "_counters.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
            for e in [t] + list(t.get("more", [])):
                if e.get("workload") == workload and e.get("streams") == B and e.get("frames") == T:
                    e = dict(e); e["source_head"] = t.get("source_head"); e["stale"] = t.get("source_head") != source_hash()
                    return e
        except (OSError, KeyError, ValueError):
            pass
    return {}


def quick_workload(torch, amd, name, dev, local, steps=10, warmup=4):
    """Another BASELINE configuration on this GPU, a few steps (the default line's `other_workloads`; the full lines: --workload NAME)."""
    fs, ms, hr, ch, n, rates, T, B, what = WORKLOADS[name]
    pcm = synth_pcm_device(torch, B, T, ch, n, fs, dev, seed=4321)
    br = [rates[i % len(rates)] for i in range(B)]
    batch = amd.Batch(B, fs, ch, ms, hr, br, device=local)
    out = torch.zeros(B, T, batch.stride, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    batch.set_input_ready(True)
    step = lambda: batch.encode_device(pcm.data_ptr(), 16, T, out.data_ptr(), batch.stride, hip_stream=stream.cuda_stream, sync=False)
    w = timed_steps(step, steps, warmup, lambda: torch.cuda.synchronize(dev))
    ok = bool(out[:, -1].ne(0).any().item())
    batch.close(); del pcm, out
    torch.cuda.empty_cache()
    return {"value": round(B * T * ch * steps / w / 1e6, 3), "unit": "M channel-frames/s", "ms_per_step": round(w / steps * 1e3, 3), "steps": steps,
            "config": "%d streams x %d frames, %d ch, %d Hz / %g ms%s" % (B, T, ch, fs, ms, " hr" if hr else ""), "nonempty": ok}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--streams", type=int, default=0, help="independent streams per GPU (default: the workload's; c1: 4096 = BASELINE configs[1])")
    ap.add_argument("--frames", type=int, default=0, help="frames per stream per step (default: the workload's; c1: 64, SURVEY 8(d))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the check of the timed launches' bytes against the CPU oracle")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-I/O and T = 1 side measurements of the default line")
    ap.add_argument("--serial-calls", action="store_true", help="encoder: do not declare the PCM ready ahead of the calls (no overlap between consecutive calls)")
    ap.add_argument("--workload", default="c1", choices=sorted(WORKLOADS), help="c1 = BASELINE configs[1] (the metric's configuration)")
    a = ap.parse_args()

    rank, world, local = rank_env()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        import torch                                      # device_count() does not initialise the GPU on this image
        have = torch.cuda.device_count()
        if have < a.gpus:
            print("bench.py: --gpus %d requested but %d device(s) visible" % (a.gpus, have), file=sys.stderr)
            sys.exit(3)
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    if world != a.gpus and "WORLD_SIZE" in os.environ and a.gpus != 1:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world), file=sys.stderr)
        sys.exit(3)

    # this rank's threads (and with them the pinned staging buffers they allocate: first touch) on the NUMA node of its GPU, before anything touches the GPU
    numa = gpu_numa_cpus(local) if world > 1 or os.environ.get("LC3_BENCH_NUMA") == "1" else None
    if numa:
        try: os.sched_setaffinity(0, set(numa[1]) & os.sched_getaffinity(0) or os.sched_getaffinity(0))
        except OSError: numa = None
    import torch
    import audio_codec_amd
    if torch.cuda.device_count() <= local:
        print("bench.py: rank %d needs device %d, %d visible" % (rank, local, torch.cuda.device_count()), file=sys.stderr)
        sys.exit(3)
    dist = None
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    plan = rank_plan(a.workload, rank, world, a.streams, a.frames)          # weak scaling: every rank owns B of the B*world streams
    fs, ms, hr, ch, n, T, B, what, first, br = (plan[k] for k in ("fs", "ms", "hr", "ch", "n", "T", "B", "what", "first", "br"))
    decode = a.workload.startswith("d")
    pcm = synth_pcm_device(torch, B, T, ch, n, fs, dev, seed=plan["seed"], first_stream=first)
    batch = audio_codec_amd.Batch(B, fs, ch, ms, hr, br, device=local)
    stride = batch.stride
    out = torch.zeros(B, T, stride, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    nbl = [batch.num_bytes(i) for i in range(B)]
    if decode:
        batch.encode_device(pcm.data_ptr(), 16, T, out.data_ptr(), stride, hip_stream=stream.cuda_stream, sync=True)
        dec = audio_codec_amd.DecBatch(B, fs, ch, ms, hr, nbl, device=local)
        back = torch.zeros(B, T, ch, n, dtype=torch.int16, device=dev)
        dec.set_input_ready(not a.serial_calls)      # the frames are resident and complete before the timed region (include/lc3plus_batch.h)
        step = lambda: dec.decode_device(out.data_ptr(), stride, T, back.data_ptr(), 16, hip_stream=stream.cuda_stream)
    else:
        # the PCM is resident and complete before the timed region: say so, and consecutive calls overlap (include/lc3plus_batch.h)
        batch.set_input_ready(not a.serial_calls)
        step = lambda: batch.encode_device(pcm.data_ptr(), 16, T, out.data_ptr(), stride, hip_stream=stream.cuda_stream, sync=False)

    # HIP events on the launch stream around the timed region -> average launch duration (all kernels of one call)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    count = [0]

    def timed_step():
        if count[0] == a.warmup: e0.record(stream)
        step()
        count[0] += 1
        if count[0] == a.warmup + a.steps: e1.record(stream)

    wall = timed_steps(timed_step, a.steps, a.warmup, lambda: torch.cuda.synchronize(dev), dist, dev)
    wall_local = timed_steps.local_wall
    kern_ms = e0.elapsed_time(e1) / a.steps
    par = None
    if not decode:
        nz = int(out[:, -1, :min(nbl)].ne(0).any(dim=1).sum().item())
        assert nz > 0.9 * B, "encoder produced empty frames"
        if rank == 0 and not a.no_parity:
            # the bytes of the timed launches themselves against the CPU oracle, before anything else touches the batch
            try:
                smp = sorted({i for i in (0, 1, 2, 5, 21, 62, 63, B // 2, B - 1) if 0 <= i < B})
                par = parity_sample(pcm, out, nbl, br, fs, ms, hr, ch, a.warmup + a.steps, smp)
            except Exception as ex:
                par = {"frames": 0, "differ": None, "note": "failed: %r" % (ex,)}
    enqueue_ms = timed_steps.enqueue / a.steps * 1e3
    if decode and rank == 0 and not a.no_parity:
        try:
            smp = sorted({i for i in (0, 1, 5, 21, 62, 63, B - 1) if 0 <= i < B})
            par = parity_sample_dec(out, back, nbl, fs, ms, hr, ch, a.warmup + a.steps, smp)
        except Exception as ex:
            par = {"frames": 0, "differ": None, "note": "failed: %r" % (ex,)}
    per_rank_ms = [round(wall_local / a.steps * 1e3, 4)]
    per_rank_kernel_ms = [round(kern_ms, 4)]
    if dist is not None:
        tw = torch.zeros(2, world, dtype=torch.float64, device=dev); tw[0, rank] = wall_local; tw[1, rank] = kern_ms
        dist.all_reduce(tw)
        per_rank_ms = [round(float(v) / a.steps * 1e3, 4) for v in tw[0].tolist()]
        per_rank_kernel_ms = [round(float(v), 4) for v in tw[1].tolist()]

    if rank == 0:
        units = B * T * ch                                   # channel-frames per step per GPU
        algo = B * T * (2 * n * ch) + T * sum(nbl)           # PCM + bitstream bytes (either direction), SURVEY 8(d)
        value = units * world * a.steps / wall / 1e6
        achieved = algo / (kern_ms * 1e-3) / 1e9
        prof = committed_profile(a.workload, B, T)
        kernels = ("lc3_dec_parse_kernel + lc3_dec_plc_kernel + lc3_dec_imdct_kernel + lc3_dec_synth_kernel" if decode else
                   "lc3_enc_resample/hp50/pitch2_kernel (pitch chain, two streams per wave) || lc3_enc_front4_kernel (MDCT, four frames per wave) -> lc3_enc_scf_lane/attack/snsvq/shape_lane_kernel "
                   "(one frame per lane) -> lc3_enc_rate_kernel (rate chain, wave per stream) -> lc3_enc_pack_kernel (tail + bitstream writer, one frame per lane); "
                   "three HIP streams, up to three calls in flight")
        res = {
            "metric": "Mframes/s encoded (48kHz/10ms/64kbps)" if a.workload == "c1" else "Mframes/s %s (channel-frames)" % ("decoded" if decode else "encoded"),
            "value": round(value, 4), "unit": "Mframes/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(wall / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s = %s; %d streams x %d frames per step per GPU" % (a.workload, what, B, T),
                       "streams_per_gpu": B, "frames_per_step": T, "channels": ch, "bytes_per_frame": sorted(set(nbl)),
                       "stereo_frames_per_step_all_gpus": B * T * world if ch == 2 else None,
                       "parallelism": "streams sharded over %d GPU(s) by contiguous blocks, no collectives" % world,
                       "input_ready": not a.serial_calls,
                       "numa": {"node": numa[0], "cpus": len(numa[1])} if numa else None,
                       "pcm": "synthetic, SURVEY 8(d) recipe as tests/lc3_harness.py states it (3 sinusoids + coloured noise + 20 dB transients at -20 dB; "
                              "1/4 of the streams strongly periodic, 1/64 digital silence, 1/64 full-scale white noise)"},
            "per_rank_ms": per_rank_ms,
            "host_enqueue_ms_per_step": round(enqueue_ms, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None if prof.get("stale") else prof.get("traffic_bytes"),
                         "kernel": kernels + " (one call = these launches; HIP events on the launch stream around all of them)",
                         "kernel_ms_avg": round(kern_ms, 4), "algorithmic_bytes_per_launch": algo,
                         "note": "not HBM bound (SURVEY 8d): see roofline_valu and DESIGN.md section 5"},
        }
        if world > 1:
            # every rank's own figure (rank 0's `achieved` alone would hide a slow rank): algorithmic GB/s from that rank's HIP-event time
            ach = [algo / (k * 1e-3) / 1e9 for k in per_rank_kernel_ms]
            res["roofline"]["per_rank"] = {"kernel_ms_avg": per_rank_kernel_ms, "achieved_min": round(min(ach), 3), "achieved_max": round(max(ach), 3),
                                           "achieved_sum": round(sum(ach), 3), "frac_of_n_peaks": round(sum(ach) / (HBM_PEAK_GBS * world), 6)}
        if par is not None: res["parity_sample"] = par
        if prof.get("stale"):
            res["roofline"]["traffic_note"] = "profiles/%s_counters.json was collected on other sources (%s, running %s): not quoted" % (PROFILE_ROUND, prof.get("source_head"), source_hash())
            prof = {}
        if prof.get("traffic_bytes"):
            res["roofline"]["traffic_over_algorithmic"] = round(prof["traffic_bytes"] / algo, 2)
            # what the hand-overs between the kernels cost: measured HBM traffic of one call over the call's duration, against the same peak
            tr = prof["traffic_bytes"] / (kern_ms * 1e-3) / 1e9
            res["roofline"]["traffic_gbps"] = round(tr, 1)
            res["roofline"]["traffic_frac"] = round(tr / HBM_PEAK_GBS, 4)
            if decode:
                res["roofline"]["note"] = ("algorithmic bytes are 1.4 %% of the HBM peak, the measured traffic (hand-over rows between the four kernels) "
                                           "%.0f %%: the parse kernel (the longest) is bound by its per-lane instruction stream under divergence "
                                           "(lane utilisation in roofline_valu), the IMDCT and synthesis kernels move 2.6 GB in 1.4 ms; DESIGN.md section 8" % (100 * tr / HBM_PEAK_GBS))
        if prof.get("valu_insts"):
            # second roofline object: wave-level VALU instruction issue.  peak = 1024 SIMDs x clock / 2 cycles per wave64 VALU op
            # (MI355X_MICROARCH.md constants table: v_fma_f32 wave64 2 cycles on a SIMD-32 with co-resident waves)
            clk = prof.get("clock_ghz", 2.4)                  # measured where the counter pass held GRBM_GUI_ACTIVE (tools/gpu_round.sh), else the nominal 2.4
            peak = 1024 * clk / 2.0
            ach = prof["valu_insts"] / (kern_ms * 1e-3) / 1e9
            res["roofline_valu"] = {"bound": "valu_issue", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "G wave-instr/s",
                                    "frac": round(ach / peak, 4), "valu_insts_per_launch": prof["valu_insts"], "clock_ghz": clk,
                                    "clock_source": prof.get("clock_source", "nominal"),
                                    "lane_utilisation": prof.get("valu_lane_util"),
                                    "source": "profiles/%s_counters.json (SQ_INSTS_VALU, separate rocprofv3 --pmc pass of this command)" % PROFILE_ROUND}
            res["roofline_valu"]["source_head"] = prof.get("source_head")
            lu = prof.get("valu_lane_util")
            if lu: res["roofline_valu"]["frac_of_lane_peak"] = round(ach / peak * lu, 4)      # 2-cycle VALU peak x the share of lanes the EXEC mask enables
            pk = prof.get("per_kernel") or {}
            counted = sum(v.get(c, 0) for v in pk.values() for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"))
            if counted and any("SQ_INSTS_SALU" in v for v in pk.values()):
                per_cycle = counted / 1024.0 / (kern_ms * 1e-3 * clk * 1e9)
                res["roofline_valu"]["issue"] = {"insts_per_simd_cycle": round(per_cycle, 3), "counted_insts_per_launch": int(counted),
                                                 "insts_per_frame": round(counted / units, 1),
                                                 "note": "vector, scalar, LDS and vector-memory-read instructions of all kernels of a call per SIMD cycle; branches, waits and "
                                                         "scalar loads are not counted (+ ~10 %)"}
        if not decode and not a.no_extras and not a.serial_calls:
            # the same steps without the input-ready promise: every call waits for the previous one to drain
            batch.set_input_ready(False)
            ws = timed_steps(step, a.steps, a.warmup, lambda: torch.cuda.synchronize(dev))
            batch.set_input_ready(True)
            res["serial_calls"] = {"value": round(units * a.steps / ws / 1e6, 4), "unit": "Mframes/s", "ms_per_step": round(ws / a.steps * 1e3, 4),
                                   "note": "rank 0, lc3plus_enc_batch_set_input_ready(0): the side kernels of a call wait for everything queued before it"}
        if a.workload == "c1" and not a.no_extras:
            # (1) the same workload through host pointers: pinned staging + chunked H2D / kernels / D2H (PCIe-inclusive; never `value`)
            h_pcm = torch.empty(pcm.shape, dtype=torch.int16, pin_memory=True); h_pcm.copy_(pcm)
            h_out = torch.empty(out.shape, dtype=torch.uint8, pin_memory=True)
            hp = h_pcm.numpy(); ho = h_out.numpy()
            batch.encode_host(hp, ho)
            th = time.perf_counter()
            for _ in range(3): batch.encode_host(hp, ho)
            th = (time.perf_counter() - th) / 3
            p_pcm = hp.copy(); p_out = ho.copy()              # pageable buffers: the library stages them through its own pinned ring
            batch.encode_host(p_pcm, p_out)
            tp = time.perf_counter()
            for _ in range(3): batch.encode_host(p_pcm, p_out)
            tp = (time.perf_counter() - tp) / 3
            res["host_io"] = {"value": round(units / th / 1e6, 4), "unit": "Mframes/s", "ms_per_step": round(th * 1e3, 3),
                              "pageable": {"value": round(units / tp / 1e6, 4), "ms_per_step": round(tp * 1e3, 3)},
                              "frac_of_kernel_only": round((units / th / 1e6) / (units / (kern_ms * 1e-3) / 1e6), 3),
                              "note": "rank 0, same workload through lc3plus_enc_batch_encode with HOST pointers: H2D of PCM, kernels and D2H of frames "
                                      "overlapped in stream chunks (pinned caller buffers; 'pageable' = staged through the library's pinned ring)"}
            # (2) T = 1: one frame per stream per call (SURVEY 8d config 2 'also report T = 1'; BASELINE configs[1] '4096 mono frames')
            for tn in (1, 4, 8):
                pc = pcm[:, :tn].contiguous(); oc = torch.zeros(B, tn, stride, dtype=torch.uint8, device=dev)
                tf = lambda: batch.encode_device(pc.data_ptr(), 16, tn, oc.data_ptr(), stride, hip_stream=stream.cuda_stream, sync=False)
                wn = timed_steps(tf, 200, 20, lambda: torch.cuda.synchronize(dev))
                res["t%d" % tn] = {"value": round(B * tn * 200 / wn / 1e6, 4), "unit": "Mframes/s", "ms_per_step": round(wn / 200 * 1e3, 4),
                                   "host_enqueue_ms_per_step": round(timed_steps.enqueue / 200 * 1e3, 4),
                                   "note": "%d streams x %d frame(s) per call, calls queued back to back under the input-ready promise (%s)"
                                           % (B, tn, "one kernel, one channel-stream per wave" if tn <= 5 else "the pipelined kernels")}
                del pc, oc
        if a.workload == "c1" and not a.no_extras and world == 1:
            # the other BASELINE configurations, ten steps each (their own lines with rooflines: --workload c3 / c4 / c5 / c96)
            res["other_workloads"] = {}
            for wn in ("c3", "c4", "c5", "c96"):
                try:
                    res["other_workloads"][wn] = quick_workload(torch, audio_codec_amd, wn, dev, local)
                except Exception as ex:
                    res["other_workloads"][wn] = {"value": None, "note": "failed: %r" % (ex,)}
        if not a.no_cpu_baseline:
            try:
                S = min(B, 2048 if not decode else 2048)
                Tc = min(T, 64)
                if decode:
                    nb0 = nbl[0]
                    same = [i for i in range(S) if nbl[i] == nb0]          # the C driver takes one frame size: the streams of the first size
                    smp = out[same][:, :Tc, :nb0].contiguous().cpu().numpy()
                    res["cpu_baseline"] = cpu_baseline("/tmp", "dec", fs, ms, hr, ch, nb0, smp, len(same), Tc, a.workload)
                else:
                    same = [i for i in range(S) if br[i] == br[0]]
                    smp = pcm[same][:, :Tc].contiguous().cpu().numpy()
                    res["cpu_baseline"] = cpu_baseline("/tmp", "enc", fs, ms, hr, ch, br[0], smp, len(same), Tc, a.workload)
            except Exception as ex:   # the baseline is reported, never required
                res["cpu_baseline"] = {"value": None, "unit": "Mframes/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
