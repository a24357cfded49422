#!/usr/bin/env python3
"""bench.py — Mrays/s of the render hot path on N MI355X (one process per GPU).

A "step" is one frame of the headline workload: SceneFiles/Teapot/scene2.xml (cyTriMesh
teapot behind a cyBVH, a refractive sphere, a ground plane, point + direct light) at
1920x1080, recipe W (one ray per pixel centre, Shade depth 5) — the bit-exact teapot
gate of BASELINE.md config 4. The scene is the flattened blob committed under
tests/golden/ (the GPU box has no scene files); it is uploaded to HBM once, before the
timed region.

N=1: python bench.py.  N>1: launched by torch.distributed.run, one rank per GPU; the
frame is sharded by interleaved 8-row bands (band b -> rank b % N, scene replicated,
no data-path collective) and every step ends with the RCCL gather of the framebuffer to rank 0
(padded float4 shards, every GPU over its own xGMI link), issued asynchronously so that it
overlaps the next frame's kernels; the timed region ends when every frame is rendered AND gathered.
Scaling is "strong": the frame (total work) is fixed as N grows.

Output: ONE JSON line on rank 0 (see the task contract) with `roofline` (algorithmic
bytes of SURVEY.md §8(d) / HIP-event kernel time / 8 TB/s) and, at N=1, `cpu_baseline`
(the CPU oracle timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WORKLOAD_TAG = "teapot2_1080"
WORKLOAD_NAME = "SceneFiles/Teapot/scene2.xml @1920x1080, recipe W (1 spp, Shade depth 5)"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tag", default=WORKLOAD_TAG, help="golden tag to render (default: the headline workload)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the cpu_baseline leg (CPU share of one GPU)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--max-bounce", type=int, default=5, help="diagnostic only: values other than 5 are not the workload")
    ap.add_argument("--frames-in-flight", type=int, default=-1,
                    help="frames rendered by one launch sequence (rtu_render_frames_device). Default: as many as keep 2^25 pixels in "
                         "flight on a GPU, at most 32 (16 full 1080p frames, 32 shards of a half frame or less). 1: one frame per launch "
                         "sequence — the frame LATENCY configuration")
    ap.add_argument("--samples", type=int, default=0,
                    help="diagnostic only: S >= 1 renders recipe S (S samples per pixel; soft shadows, glossy bounces, depth of field) — "
                         "needed for the tags under tests/golden/ whose meta.json says recipe S; not the headline workload")
    ap.add_argument("--paths", action="store_true",
                    help="diagnostic only, with --samples S: recipe P — recipe S plus the 4-bounce Monte-Carlo gather of config 5")
    ap.add_argument("--size", default="", help="diagnostic only: WxH instead of the tag's own resolution (no golden z check then)")
    ap.add_argument("--coop-threshold", type=int, default=0, help="tuning: ray-list length below which stage 2 is cooperative (0 = library default)")
    ap.add_argument("--gather-float4", action="store_true",
                    help="N>1: gather the float4 {r,g,b,z} shards (16 B per pixel) instead of the packed RenderImage content "
                         "(float z + Color24, 7 B per pixel, converted on the device)")
    ap.add_argument("--allgather", action="store_true", help="N>1: all_gather the framebuffer to every rank instead of gathering it to rank 0")
    ap.add_argument("--rehearse", action="store_true",
                    help="N>1 on a box with ONE GPU: every rank renders its shard on cuda:0 and the gather goes through gloo on host "
                         "copies. Exercises the sharded code path; the number it prints is not a measurement")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    import importlib.util
    spec = importlib.util.spec_from_file_location("rtu_sharding", os.path.join(REPO, "raytracer-utah_amd", "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            local_rank = 0
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if args.rehearse else dev  # where the collectives' tensors live

    gdir = os.path.join(REPO, "tests", "golden", args.tag)
    meta = json.load(open(os.path.join(gdir, "meta.json")))
    W, H = meta["width"], meta["height"]
    scene = pkg.Scene.from_blob_file(os.path.join(gdir, "scene.rtus.gz"))
    if args.size:
        W, H = (int(v) for v in args.size.lower().split("x"))
        scene.set_resolution(W, H)
        meta = dict(meta, sha256_z_f32=None)
    ctx = pkg.Context(local_rank)
    ctx.upload(scene)  # inputs resident in HBM before any timing

    frame = pkg.frame_setup(scene.desc.camera, W, H, shard_rank=rank, shard_count=world, max_bounce=args.max_bounce, samples=args.samples,
                            gather_bounces=4 if args.paths else 0)
    frame.coop_threshold = args.coop_threshold
    rows = pkg.shard_rows(frame)
    max_rows = pkg.hip.rtu_shard_max_rows(H, world)
    # frames in flight: B frames per launch sequence, as [frame][row of the shard][x] in one buffer
    B = args.frames_in_flight
    if args.samples:
        B = 1  # recipe S batches its samples itself
    elif B < 1:
        B = max(1, min(32, (1 << 25) // max(1, max_rows * W)))
    B = max(1, min(B, 32, args.steps))
    # two shard / gather buffers: the RCCL gather of batch i runs while batch i+1 is rendered
    shards = [torch.zeros(B * max_rows * W * 4, dtype=torch.float32, device=dev) for _ in range(2 if world > 1 else 1)]
    root_only = not args.allgather
    # N > 1: what travels to the root is the reference's RenderImage content — float z + Color24, 7 bytes per pixel,
    # converted on the device (rtu_pack_image_device) — unless --gather-float4 asks for the raw float4 shards
    packed = world > 1 and not args.gather_float4
    pbytes = sharding.packed_bytes(B, max_rows, W)
    sends = [torch.zeros(pbytes, dtype=torch.uint8, device=dev) for _ in range(2)] if packed else shards
    gathers = None
    if world > 1 and (rank == 0 or not root_only):
        if packed:
            gathers = [torch.empty(world, pbytes, dtype=torch.uint8, device=cdev) for _ in range(2)]
        else:
            gathers = [torch.empty(world, B * max_rows * W * 4, dtype=torch.float32, device=cdev) for _ in range(2)]
    shard = shards[0]
    stream = torch.cuda.current_stream().cuda_stream

    # -- untimed: ray / traversal counters of this shard (stats kernel variant) --------
    sframe = pkg.frame_setup(scene.desc.camera, W, H, shard_rank=rank, shard_count=world, collect_stats=True,
                             max_bounce=args.max_bounce, samples=args.samples, gather_bounces=4 if args.paths else 0)
    if args.samples:  # settle the frame-record capacities first: the counting pass of a sampled frame does not re-provision
        ctx.render_device(frame, shard.data_ptr(), stream)
        torch.cuda.synchronize()
    ctx.render_device(sframe, shard.data_ptr(), stream)
    torch.cuda.synchronize()
    st = ctx.stats()
    keys = sorted(st)
    tot = torch.tensor([st[k] for k in keys], dtype=torch.int64, device=cdev)
    if dist:
        dist.all_reduce(tot)
    total = dict(zip(keys, [int(v) for v in tot.tolist()]))
    rays_per_frame = pkg.total_rays(total)
    alg_bytes_launch = pkg.algorithmic_bytes(st, rows * W)  # this rank's launch

    pipe = sharding.FramePipeline(sends, gathers, dist, staged=args.rehearse, root_only=root_only) if dist else None

    def step(i, nb, ev=None):
        """One launch sequence: nb frames (steps) in flight."""
        buf = pipe.begin(i) if pipe else shard  # waits (on the GPU) for the gather that last read this buffer
        if packed:
            send, buf = buf, shards[i & 1]
        if ev:
            ev[0].record()
        if nb == 1:
            ctx.render_device(frame, buf.data_ptr(), stream)
        else:
            ctx.render_frames_device([frame] * nb, buf.data_ptr(), stream)
        if packed:  # z of the nb frames, then their Color24 pixels (sharding.assemble_gathered_packed reads this layout)
            ctx.pack_image_device(buf.data_ptr(), nb * rows * W, send.data_ptr(), send.data_ptr() + B * max_rows * W * 4, stream)
        if ev:
            ev[1].record()
        if pipe:
            pipe.gather(i)  # RCCL over xGMI, asynchronous: overlaps the next batch's kernels

    for attempt in range(2 * 6 + 1):
        for j in range(max(1, -(-args.warmup // B))):
            step(j, B)  # warm-up at the full batch size (at least args.warmup frames): buffers are provisioned for B frames in flight
        if pipe:
            pipe.drain()
        settled = 1
        try:
            ctx.frame_status()  # a recursion level needed more frame records than provisioned: the context has grown them, warm up again
        except pkg.RtuError as e:
            if e.code != pkg.RTU_ERR_CAPACITY or attempt == 2 * 6:
                raise
            settled = 0
        if dist:  # every rank repeats the warm-up (and its gathers) if any rank has to
            flag = torch.tensor([settled], dtype=torch.int32, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            settled = int(flag[0])
        if settled:
            break
    batches = [min(B, args.steps - i) for i in range(0, args.steps, B)]  # EXACTLY args.steps frames
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in batches]
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j, nb in enumerate(batches):
        step(j, nb, events[j])
    if pipe:
        pipe.drain()  # every frame of the timed region rendered AND gathered
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gathered = pipe.last_gathered() if pipe else None
    # HIP events on the launch stream, around the full launch sequences (B frames each; a shorter last batch is left out)
    full = [(a.elapsed_time(b), nb) for (a, b), nb in zip(events, batches) if nb == batches[0]]
    kernel_ms = float(np.mean([t for t, _ in full]))
    alg_bytes_launch *= batches[0]  # bytes of one launch sequence = frames in flight x bytes of a frame
    ctx.frame_status()  # raises if a recursion level overflowed its provisioned capacity

    t = torch.tensor([elapsed, kernel_ms, float(alg_bytes_launch)], dtype=torch.float64, device=cdev)
    if dist:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0])
        # roofline of the dominant kernel: the slowest rank's launch
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        slow = max(allt, key=lambda x: float(x[1]))
        kernel_ms, alg_bytes_launch = float(slow[1]), float(slow[2])

    # -- untimed: parity of what was just rendered (z bit-exact vs the reference golden) ---
    img = None
    if dist:
        if rank == 0:
            # the last frame of the last batch (frame j of rank r's chunk starts at j * rows_r * W float4)
            if packed:
                zimg, rgb8 = sharding.assemble_gathered_packed(pkg, gathered.view(world, -1).cpu().numpy(), batches[-1] - 1, B, scene.desc.camera, W, H, world, max_rows)
                img = np.zeros((H, W, 4), np.float32)
                img[..., 3] = zimg
            else:
                img = sharding.assemble_gathered_batch(pkg, gathered.view(world, -1).cpu().numpy(), batches[-1] - 1, scene.desc.camera, W, H, world)
    else:
        j = batches[-1] - 1
        img = shard.view(-1)[j * rows * W * 4:(j + 1) * rows * W * 4].view(rows, W, 4).cpu().numpy()
    if args.rehearse and rank == 0:
        print("[rehearsal: %d ranks on one GPU through gloo — not a measurement]" % world, file=sys.stderr)
    import hashlib
    z_ok = img is not None and args.samples == 0 and hashlib.sha256(np.ascontiguousarray(img[..., 3]).tobytes()).hexdigest() == meta["sha256_z_f32"]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_per_frame * args.steps / elapsed / 1e6
        achieved = alg_bytes_launch / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(REPO, "profiles", "hbm_traffic.json")
        if world == 1 and args.tag == WORKLOAD_TAG and os.path.exists(tfile):
            traffic = json.load(open(tfile)).get("bytes_per_launch_by_frames_in_flight", {}).get(str(batches[0]))
        out = {
            "metric": "Mrays/sec at 1920x1080 (primary + secondary + shadow rays per frame / frame time)",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (WORKLOAD_NAME if args.tag == WORKLOAD_TAG else args.tag) + (" recipe %s, %d samples per pixel" % ("P" if args.paths else "S", args.samples) if args.samples else ""),
                       "width": W, "height": H,
                       "rays_per_frame": rays_per_frame, "primary": total["primary_rays"],
                       "secondary": total["secondary_rays"], "shadow": total["shadow_rays"],
                       "frames_in_flight": batches[0], "frame_latency_ms": round(kernel_ms, 4),
                       "sharding": ("interleaved 8-row bands, RCCL gather of the %s to rank 0, overlapped with the next batch" % ("RenderImage content (float z + Color24, 7 B per pixel, packed on the device)" if packed else "float4 framebuffer")) if world > 1 else "single GPU",
                       "z_bit_exact_vs_reference_golden": bool(z_ok)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "note": "achieved = algorithmic (touched, cache-agnostic) bytes of SURVEY 8d / launch duration: the working set is "
                                 "L2-resident, so this may exceed the HBM peak; traffic = HBM bytes per launch from the PMC counters",
                         "kernel": "render_kernel", "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": int(alg_bytes_launch)},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(g, scene, W, H, rays_per_frame, args.cpu_seconds, args.cpu_threads, args.samples, args.paths)
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist:
        dist.destroy_process_group()


def cpu_baseline(g, scene, W, H, rays_per_frame, budget_s, max_threads, samples=0, paths=False):
    """The CPU oracle (a port: bit-identical restatement of the reference's Trace/Shade,
    see oracle/rtu_oracle.cpp) on this box's host cores, same workload, whole frames
    repeated until ~budget_s of wall time has been spent."""
    orc = g.load_oracle()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, max_threads))
    render = (lambda threads: orc.render_samples(scene, W, H, samples, threads=threads)) if samples else (lambda threads: orc.render(scene, W, H, threads=threads))
    if paths:
        render = lambda threads: orc.render_paths(scene, W, H, samples, threads=threads)
    t0 = time.perf_counter()
    render(1)
    t1 = time.perf_counter() - t0
    frames, t0 = 0, time.perf_counter()
    while True:
        render(cores)
        frames += 1
        el = time.perf_counter() - t0
        if el > max(1.0, budget_s - t1) or frames >= 200:
            break
    return {"value": round(rays_per_frame * frames / el / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d full %dx%d frames of the same workload on %d threads (row-chunk schedule); "
                      "1 thread: %.3f Mrays/s" % (frames, W, H, cores, rays_per_frame / t1 / 1e6),
            "ms_per_frame": round(el / frames * 1e3, 3)}


if __name__ == "__main__":
    main()
