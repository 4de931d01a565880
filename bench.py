#!/usr/bin/env python3
"""bench.py -- Mframes/s of the LC3plus encode hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: B independent mono streams advanced by T frames each
(48 kHz / 10 ms / 64 kbps, BASELINE.json configs[1]: 4096 streams; T = 64 as in SURVEY 8(d)).  PCM is synthetic and
already resident in HBM when the timed region starts; bitstreams are written to HBM.  With --gpus N every rank
(one process per GPU, torch.distributed over RCCL used only for the barrier and the max-over-ranks time) encodes its
own B streams: streams are independent, so there is no data-path collective and scaling is weak.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS, FRAME_MS, N, BITRATE, NBYTES = 48000, 10.0, 480, 64000, 80
ALGO_BYTES_PER_FRAME = 2 * N + NBYTES          # int16 PCM in + bitstream out = 1040
# The other BASELINE.json configs (parity-test cases; selectable for the record with --workload, never the default):
#   name: (fs, frame_ms, hrmode, channels, N, per-stream total bitrates (cycled), frames per step)
WORKLOADS = {
    "c1": (48000, 10.0, 0, 1, 480, [64000], 64),
    "c3": (48000, 10.0, 0, 2, 480, [128000], 16),
    "c4": (96000, 2.5, 1, 1, 240, [256000], 256),
    "c5": (48000, 10.0, 0, 1, 480, [16000, 24000, 32000, 48000, 64000, 80000, 96000, 128000, 160000, 192000, 256000, 320000], 64),
    # decoder: the C1 / C5 bitstreams, produced on the GPU just before, decoded back to 16-bit PCM
    "d1": (48000, 10.0, 0, 1, 480, [64000], 64),
    "d5": (48000, 10.0, 0, 1, 480, [16000, 24000, 32000, 48000, 64000, 80000, 96000, 128000, 160000, 192000, 256000, 320000], 64),
}
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: 8.0 TB/s spec


def synth_pcm_device(torch, B, T, dev, seed):
    """Deterministic synthetic PCM on the device: 3 sinusoids + coloured noise + a 20 dB transient every 37 frames."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    n = T * N
    t = torch.arange(n, device=dev, dtype=torch.float32) / FS
    x = torch.zeros(B, n, device=dev, dtype=torch.float32)
    for _ in range(3):
        f = 80.0 + torch.rand(B, 1, device=dev, generator=g) * (0.4 * FS - 80.0)
        a = (0.02 + 0.23 * torch.rand(B, 1, device=dev, generator=g)) * 32767.0
        ph = torch.rand(B, 1, device=dev, generator=g) * 6.2831853
        x += a * torch.sin(6.2831853 * f * t[None, :] + ph)
    noise = torch.randn(B, n, device=dev, generator=g)
    noise = 0.5 * noise + 0.3 * torch.roll(noise, 1, 1) + 0.15 * torch.roll(noise, 2, 1) + 0.05 * torch.roll(noise, 3, 1)
    x += noise * (32767.0 * 10 ** (-30 / 20))
    env = torch.ones(n, device=dev)
    for k in range(0, T, 37):
        a0 = k * N + N // 3
        env[a0:a0 + N // 2] = 10.0
    x = x * env[None, :] * 0.1
    return x.round().clamp(-32768, 32767).to(torch.int16).reshape(B, T, N).contiguous()


def cpu_baseline(n_streams=2048, T=64):
    """ETSI reference (oracle/_ref, kind 'reference') or the C restatement (kind 'port') timed on the host cores.
    Bounded sample: n_streams x T frames of the same workload, one worker process per core (streams are independent)."""
    import numpy as np
    from concurrent.futures import ProcessPoolExecutor
    cores = max(1, min(os.cpu_count() or 1, 32))
    per = max(1, n_streams // cores)
    kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "liblc3_etsi_fl.so")) else "port"
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=cores) as ex:
        secs = list(ex.map(_cpu_worker, [(kind, per, T, 100 + i) for i in range(cores)]))
    wall = time.time() - t0
    frames = per * T * cores
    return {"value": round(frames / max(secs) / 1e6, 6), "unit": "Mframes/s", "cores": cores, "kind": kind,
            "sample": "%d streams x %d frames (48kHz/10ms/64kbps mono), %d worker processes, %.1f s wall incl. start-up; "
                      "slowest worker %.2f s of encode" % (per * cores, T, cores, wall, max(secs))}


def _cpu_worker(args):
    kind, n_streams, T, seed = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import lc3_harness as H
    rng = np.random.RandomState(seed)
    tt = np.arange(T * N) / FS
    pcm = np.zeros((n_streams, T * N))
    for s in range(n_streams):
        for _ in range(3):
            pcm[s] += rng.uniform(0.02, 0.25) * 3276.7 * np.sin(2 * np.pi * rng.uniform(80, 0.4 * FS) * tt + rng.uniform(0, 6.28))
    pcm += rng.standard_normal(pcm.shape) * 100.0
    pcm = np.clip(np.rint(pcm), -32768, 32767).astype(np.int16).reshape(n_streams, T, N)
    t0 = time.time()
    for s in range(n_streams):
        enc = H.Ref(FS, 1, FRAME_MS, 0, BITRATE) if kind == "reference" else H.Oracle(FS, 1, FRAME_MS, 0, BITRATE)
        for t in range(T):
            enc.encode(pcm[s, t][None, :])
    return time.time() - t0


def cpu_baseline_decode(frames, nbytes, n_streams, T):
    """Reference (or restated) DECODER on the host cores over a bounded sample of the bench bitstreams (rank 0, --workload d*)."""
    from concurrent.futures import ProcessPoolExecutor
    cores = max(1, min(os.cpu_count() or 1, 32))
    per = max(1, n_streams // cores)
    kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "liblc3_etsi_fl.so")) else "port"
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=cores) as ex:
        secs = list(ex.map(_cpu_dec_worker, [(kind, frames[i * per:(i + 1) * per, :T], nbytes[i * per:(i + 1) * per]) for i in range(cores)]))
    wall = time.time() - t0
    return {"value": round(per * T * cores / max(secs) / 1e6, 6), "unit": "Mframes/s", "cores": cores, "kind": kind,
            "sample": "%d streams x %d frames of the bench bitstreams, %d worker processes, %.1f s wall incl. start-up; slowest worker %.2f s of decode"
                      % (per * cores, T, cores, wall, max(secs))}


def _cpu_dec_worker(args):
    kind, frames, nbytes = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lc3_harness as H
    t0 = time.time()
    for s in range(frames.shape[0]):
        dec = H.RefDecoder(FS, 1, FRAME_MS, 0) if kind == "reference" else H.OracleDecoder(FS, 1, FRAME_MS, 0)
        for t in range(frames.shape[1]):
            dec.decode(frames[s, t, :nbytes[s]], 0, 16)
    return time.time() - t0


def other_workload(a):
    """Single-GPU record runs of the other BASELINE configs (same timing protocol; no CPU baseline; informational)."""
    import torch
    import audio_codec_amd
    fs, ms, hr, ch, n, rates, T = WORKLOADS[a.workload]
    if a.frames != 64: T = a.frames
    B = a.streams
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    g = torch.Generator(device=dev); g.manual_seed(99)
    pcm = (torch.randn(B, T, ch, n, device=dev, generator=g) * 3000).round().clamp(-32768, 32767).to(torch.int16)
    pcm += (8000 * torch.sin(torch.arange(n, device=dev) * 0.05)).to(torch.int16)[None, None, None, :]
    br = [rates[i % len(rates)] for i in range(B)]
    batch = audio_codec_amd.Batch(B, fs, ch, ms, hr, br, device=0)
    stride = batch.stride
    out = torch.zeros(B, T, stride, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    if a.workload.startswith("d"):
        batch.encode_device(pcm.data_ptr(), 16, T, out.data_ptr(), stride, hip_stream=stream.cuda_stream, sync=True)
        dec = audio_codec_amd.DecBatch(B, fs, ch, ms, hr, [batch.num_bytes(i) for i in range(B)], device=0)
        back = torch.zeros(B, T, ch, n, dtype=torch.int16, device=dev)
        run = lambda: dec.decode_device(out.data_ptr(), stride, T, back.data_ptr(), 16, hip_stream=stream.cuda_stream)
        what = "decoded"
    else:
        run = lambda: batch.encode_device(pcm.data_ptr(), 16, T, out.data_ptr(), stride, hip_stream=stream.cuda_stream)
        what = "encoded"
    for _ in range(a.warmup): run()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(stream)
    for _ in range(a.steps): run()
    e1.record(stream); torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    nbytes = [batch.num_bytes(i) for i in range(min(B, len(rates)))]
    algo = B * T * (2 * n * ch) + T * sum(batch.num_bytes(i) for i in range(B))
    kern_ms = e0.elapsed_time(e1) / a.steps
    extra = {}
    if a.workload.startswith("d") and not a.no_cpu_baseline:
        nbl = [batch.num_bytes(i) for i in range(B)]
        extra["cpu_baseline"] = cpu_baseline_decode(out[:2048].cpu().numpy(), nbl[:2048], min(B, 2048), min(T, 64))
    line = {"metric": "Mframes/s %s (channel-frames)" % what, "value": round(B * T * ch * a.steps / wall / 1e6, 4), "unit": "Mframes/s",
                      "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(wall / a.steps * 1e3, 4), "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": "%s: %d streams x %d frames, %d Hz / %.1f ms%s, %d ch, bytes/frame %s" % (a.workload, B, T, fs, ms, " hr" if hr else "", ch, nbytes)},
                      "roofline": {"bound": "hbm", "achieved": round(algo / (kern_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(algo / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": measured_traffic(a.workload, B, T), "kernel_ms_avg": round(kern_ms, 4)}}
    line.update(extra)
    print(json.dumps(line))


def measured_traffic(workload, B, T):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_traffic.json);
    only valid for the launch shape it was collected on, otherwise null."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            t = json.load(f)
        for e in [t] + list(t.get("more", [])):
            if e["workload"] == workload and e["streams"] == B and e["frames"] == T:
                return int(e["traffic_bytes"])
    except (OSError, KeyError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=4096, help="independent mono streams per GPU (BASELINE configs[1])")
    ap.add_argument("--frames", type=int, default=64, help="frames per stream per step (SURVEY 8(d): T = 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-io", action="store_true", help="also time encode() on pageable host buffers (PCIe-inclusive; reported as host_io, never as value)")
    ap.add_argument("--workload", default="c1", choices=sorted(WORKLOADS), help="c1 = BASELINE configs[1] (the metric's configuration)")
    a = ap.parse_args()
    if a.workload != "c1":
        return other_workload(a)

    import torch
    import audio_codec_amd

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    B, T = a.streams, a.frames
    from audio_codec_amd.sharding import stream_block
    first, last = stream_block(rank, world, B * world)          # weak scaling: every rank owns B of the B*world streams
    assert last - first == B
    pcm = synth_pcm_device(torch, B, T, dev, seed=1234 + first)
    batch = audio_codec_amd.Batch(B, FS, 1, FRAME_MS, 0, [BITRATE] * B, device=local)
    stride = batch.stride
    out = torch.zeros(B, T, stride, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        batch.encode_device(pcm.data_ptr(), 16, T, out.data_ptr(), stride, hip_stream=stream.cuda_stream, sync=False)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist: dist.barrier()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(a.steps):
        step()
    e1.record(stream)
    torch.cuda.synchronize(dev)
    if dist: dist.barrier()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    kern_ms = e0.elapsed_time(e1) / a.steps          # HIP events on the launch stream, averaged over the timed region
    if dist:
        tw = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    assert int(out[:, -1, :NBYTES].ne(0).any(dim=1).sum().item()) > 0.9 * B, "encoder produced empty frames"

    if rank == 0:
        frames_per_step = B * T * world
        value = frames_per_step * a.steps / wall / 1e6
        achieved = B * T * ALGO_BYTES_PER_FRAME / (kern_ms * 1e-3) / 1e9
        res = {
            "metric": "Mframes/s encoded (48kHz/10ms/64kbps)", "value": round(value, 4), "unit": "Mframes/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(wall / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d independent mono streams x %d frames per step per GPU, "
                                   "48kHz/10ms/64kbps, one channel-stream per wavefront" % (B, T),
                       "streams_per_gpu": B, "frames_per_step": T, "parallelism": "streams sharded over %d GPU(s), no collectives" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": measured_traffic(a.workload, B, T),
                         "kernel": "lc3_enc_resample_kernel + lc3_enc_hp50_kernel + lc3_encode_kernel + lc3_enc_pack_kernel (one encode() call = these four launches; HIP events around all of them)", "kernel_ms_avg": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": B * T * ALGO_BYTES_PER_FRAME,
                         "note": "serial-chain (instruction-issue) bound, not HBM bound: see DESIGN.md"},
        }
        if a.host_io:
            import numpy as np
            h_pcm = pcm.cpu().numpy()
            batch.encode(h_pcm[:, :T])                         # warm-up (first-touch of the staging buffers)
            th = time.perf_counter()
            for _ in range(3): batch.encode(h_pcm[:, :T])
            th = (time.perf_counter() - th) / 3
            res["host_io"] = {"value": round(B * T / th / 1e6, 4), "unit": "Mframes/s", "ms_per_step": round(th * 1e3, 3),
                              "note": "rank 0, same workload through lc3plus_enc_batch_encode with pageable host pointers (H2D + kernel + D2H, synchronous)"}
        if not a.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as ex:   # the baseline is reported, never required
                res["cpu_baseline"] = {"value": None, "unit": "Mframes/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}
        print(json.dumps(res))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
