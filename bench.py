#!/usr/bin/env python3
"""bench.py — Mrays/s of the render hot path on N MI355X (one process per GPU).

A "step" is ONE FRAME of the headline workload: SceneFiles/Teapot/scene2.xml (cyTriMesh teapot behind a
cyBVH, a refractive sphere, a ground plane, point + direct light) at 1920x1080, recipe W (one ray per pixel
centre, Shade depth 5) — the bit-exact teapot gate of BASELINE.md config 4. The scene is the reference's
own, as the flattened blob committed under tests/golden/ (the GPU box has no scene files); it is uploaded
to HBM once, before the timed region.

Frames in flight: `frames_in_flight` frames per launch sequence (rtu_render_frames_device), each with ITS
OWN camera — a turntable: frame j of a batch is the scene's camera orbited by j x 2 degrees (frame 0 is the
golden camera, whose z is checked bit for bit against the reference's after the run). `value` counts the
rays of the frames actually rendered (counted per camera by the counting variant, untimed).
`config.single_frame` is the same workload with ONE frame per launch sequence — what BeginRender() of one
image costs — measured right after the main timed region.

Roofline (rocprof names, HIP events): the launch sequence is ~20 kernels; `roofline` is quoted for the
DOMINANT one (longest per launch, found by an untimed probe of every kernel slot), its algorithmic bytes
counted by the fast variant itself (collect_stats == 2: the very kernels that are timed, counting what
they read and write — include/rtu_render.h RtuTouched) divided by its duration, measured with HIP events
around its launches INSIDE the timed region. `roofline.kernels` lists every kernel of the sequence the
same way. `achieved` / `frac` are SURVEY 8d's nominal figure: cache-agnostic touched bytes against the HBM
peak (from the device's own memory clock and bus width, rtu_device_info; the guide's 8 TB/s as a fall-back).
The scene is cache-resident, so that figure says how fast a kernel goes through its working set, NOT what
binds it: `bound` and `hbm_counter_frac` / `valu_issue_frac` / `lanes_active` come from the committed
rocprofv3 PMC profile of the same command (profiles/rNN_per_kernel.json, labelled profile-derived, like
`traffic`), and `l2_frac` (34.5 TB/s) is given beside them.

Timing: the K-step region (exactly --steps frames, barrier + synchronize on both sides) is run `--repeats` R
times back to back (default 50: the driver's --steps 20 is ONE 1.4 ms launch sequence, far too short a
sample on its own); `value` and `ms_per_step` are the MEDIAN region's, config.repeats has min / max / first.

N=1: python bench.py.  N>1: launched by torch.distributed.run, one rank per GPU; the frame is sharded by
interleaved 8-row bands (band b -> rank b % N, scene replicated, no data-path collective) and every batch
ends with the RCCL gather of the RenderImage content to rank 0, issued asynchronously so that it overlaps
the next batch's kernels; the timed region ends when every frame is rendered AND gathered.
Scaling is "strong": the frame (total work) is fixed as N grows.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WORKLOAD_TAG = "teapot2_1080"
WORKLOAD_NAME = "SceneFiles/Teapot/scene2.xml @1920x1080, recipe W (1 spp, Shade depth 5)"
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec) — the fall-back of hbm_peak()
L2_PEAK_GBS = 34500.0   # MI355X_MICROARCH.md: L2 aggregate ~34.5 TB/s


def hbm_peak(pkg, device):
    """(GB/s, where the figure comes from): the device's own memory clock x bus width, double data rate ("confirm on
    the machine, do not hard-code"); the guide's constant when the machine reports something implausible for an MI355X."""
    try:
        info = pkg.device_info(device)
        # HIP reports a quarter of the per-pin data rate of HBM3 / HBM3E (MI300X: 1.3 GHz x 8192 bit x 4 = 5.3 TB/s, its spec;
        # measured here: 2.0 GHz x 8192 bit x 4 = 8.19 TB/s, the guide's "8 TB/s")
        gbs = 4.0 * info["memory_clock_khz"] * 1e3 * info["memory_bus_bits"] / 8 / 1e9
        src = "hipGetDeviceProperties: memoryClockRate %d kHz x memoryBusWidth %d bit x 4 (HBM3E: the reported clock is a quarter of the pin rate)" % (
            info["memory_clock_khz"], info["memory_bus_bits"])
        if 6000.0 <= gbs <= 10000.0:
            return gbs, src
        return HBM_PEAK_GBS, "MI355X_MICROARCH.md (8 TB/s): the device reports %s = %.0f GB/s, implausible for HBM3E x 8 stacks" % (src, gbs)
    except Exception as e:  # noqa: BLE001 — a diagnostic must not take the bench down
        return HBM_PEAK_GBS, "MI355X_MICROARCH.md (8 TB/s): rtu_device_info failed (%s)" % e


def kernel_profile():
    """The newest committed per-kernel PMC summary (tools/profile_summary.py), or {}."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_per_kernel.json")))
    if not files:
        return {}
    prof = json.load(open(files[-1]))
    prof["file"] = os.path.relpath(files[-1], REPO)
    return prof
ORBIT_STEP_DEG = 2.0


def orbit_camera(cam, degrees):
    """The camera orbited about its own up axis through the point of the view axis closest to the world origin
    (the scenes are built around the origin): a turntable step. 0 degrees returns an exact copy."""
    c = type(cam).from_buffer_copy(cam)
    if degrees == 0:
        return c
    pos, d, up = (np.array(list(v), np.float64) for v in (cam.pos, cam.dir, cam.up))
    d /= np.linalg.norm(d)
    k = up / np.linalg.norm(up)
    centre = pos + d * float(np.dot(-pos, d))
    th = math.radians(degrees)

    def rot(v):
        return v * math.cos(th) + np.cross(k, v) * math.sin(th) + k * float(np.dot(k, v)) * (1 - math.cos(th))
    p2, d2 = centre + rot(pos - centre), rot(d)
    for i in range(3):
        c.pos[i], c.dir[i] = float(p2[i]), float(d2[i])
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=320)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--repeats", type=int, default=50,
                    help="the timed K-step region is run this many times back to back; value / ms_per_step are the median region's (min, max and the first in config.repeats)")
    ap.add_argument("--tag", default=WORKLOAD_TAG, help="golden tag to render (default: the headline workload)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the cpu_baseline leg (0 = every core this process may use: its affinity mask, capped by the cgroup CPU quota)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--max-bounce", type=int, default=5, help="diagnostic only: values other than 5 are not the workload")
    ap.add_argument("--frames-in-flight", type=int, default=-1,
                    help="frames rendered by one launch sequence (rtu_render_frames_device). Default: as many as keep 2^26 pixels in "
                         "flight on a GPU, at most 128 (32 full 1080p frames, 128 shards of a quarter of the frame or less). 1: one frame per launch "
                         "sequence — the frame LATENCY configuration")
    ap.add_argument("--rays-per-frame", type=int, default=0,
                    help="diagnostic only: skip the (slow, untimed) counting pass that counts the reference's rays per frame and use this number instead — "
                         "for profiler runs of sampled frames, where the counting variant would drown the timed kernels in the trace")
    ap.add_argument("--contexts", type=int, default=2,
                    help="N=1: launch sequences alternate over this many contexts, each on a stream of its own (default 2; measured 1 / 2 / 3: 49.4 / 62.5 / 60.5 Grays/s): the latency-bound recursion "
                         "levels of one sequence overlap the primary kernels of the next. Only when the timed region holds at least that many full "
                         "batches (the driver's --steps 20 is one batch: one context)")
    ap.add_argument("--same-camera", action="store_true", help="diagnostic only: every frame of a batch from the golden camera (no turntable)")
    ap.add_argument("--samples", type=int, default=0,
                    help="diagnostic only: S >= 1 renders recipe S (S samples per pixel; soft shadows, glossy bounces, depth of field) — "
                         "needed for the tags under tests/golden/ whose meta.json says recipe S; not the headline workload")
    ap.add_argument("--paths", action="store_true",
                    help="diagnostic only, with --samples S: recipe P — recipe S plus the 4-bounce Monte-Carlo gather of config 5")
    ap.add_argument("--size", default="", help="diagnostic only: WxH instead of the tag's own resolution (no golden z check then)")
    ap.add_argument("--coop-threshold", type=int, default=0, help="tuning: ray-list length below which stage 2 is cooperative (0 = library default)")
    ap.add_argument("--gather-float4", action="store_true",
                    help="N>1: gather the float4 {r,g,b,z} shards (16 B per pixel) instead of the output images")
    ap.add_argument("--gather-render-image", action="store_true",
                    help="N>1: gather float z + Color24 (7 B per pixel, the reference's RenderImage arrays) instead of the two output images "
                         "(Color24 + z-image byte, 4 B per pixel — the default: at these frame rates the link to the root is the bottleneck)")
    ap.add_argument("--allgather", action="store_true", help="N>1: all_gather the framebuffer to every rank instead of gathering it to rank 0")
    ap.add_argument("--dbg", type=int, default=0, help="experiment switches (rtu_debug_flags): the images are WRONG with anything but 0")
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
    if args.dbg:
        pkg.hip.rtu_debug_flags(ctx._h, args.dbg)
    textured = scene.desc.n_textures > 0
    sampled = args.samples > 0

    def mkframe(cam, **kw):
        f = pkg.frame_setup(cam, W, H, shard_rank=rank, shard_count=world, max_bounce=args.max_bounce, samples=args.samples,
                            gather_bounces=4 if args.paths else 0, **kw)
        f.coop_threshold = args.coop_threshold
        return f

    frame0 = mkframe(scene.desc.camera)
    rows = pkg.shard_rows(frame0)
    max_rows = pkg.hip.rtu_shard_max_rows(H, world)
    # frames in flight: B frames per launch sequence, as [frame][row of the shard][x] in one buffer
    B = args.frames_in_flight
    if sampled:
        B = 1  # recipe S batches its samples itself
    elif B < 1:
        B = max(1, min(128, (1 << 26) // max(1, max_rows * W)))
    B = max(1, min(B, 128, args.steps))
    # the turntable: camera j of a batch (0: the scene's own camera, the one the golden z belongs to)
    cams = [orbit_camera(scene.desc.camera, 0.0 if (args.same_camera or sampled) else ORBIT_STEP_DEG * j) for j in range(B)]
    frames = [mkframe(c) for c in cams]

    # two shard / gather buffers: the RCCL gather of batch i runs while batch i+1 is rendered
    shards = [torch.zeros(B * max_rows * W * 4, dtype=torch.float32, device=dev) for _ in range(2 if world > 1 else 1)]
    root_only = not args.allgather
    # N > 1: what travels to the root are the two OUTPUT images — Color24 + the z-image byte, 4 bytes per pixel, made on the device
    # (rtu_minmax_z_device, one all-reduce MIN of the per-frame zmin / zmax keys, rtu_pack_output_device): at ~10 000 frames per
    # second the root's xGMI links are the bottleneck, and Result.png / ZBuffer.png are what the reference writes.
    # --gather-render-image: float z + Color24 (7 B per pixel, rtu_pack_image_device); --gather-float4: the raw float4 shards
    out4 = world > 1 and not args.gather_float4 and not args.gather_render_image
    packed = world > 1 and args.gather_render_image and not args.gather_float4
    pbytes = sharding.out4_bytes(B, max_rows, W) if out4 else sharding.packed_bytes(B, max_rows, W)
    sends = [torch.zeros(pbytes, dtype=torch.uint8, device=dev) for _ in range(2)] if (packed or out4) else shards
    minmax = [torch.zeros(2 * B, dtype=torch.int64, device=dev) for _ in range(2)] if out4 else None
    gathers = None
    if world > 1 and (rank == 0 or not root_only):
        if packed or out4:
            gathers = [torch.empty(world, pbytes, dtype=torch.uint8, device=cdev) for _ in range(2)]
        else:
            gathers = [torch.empty(world, B * max_rows * W * 4, dtype=torch.float32, device=cdev) for _ in range(2)]
    shard = shards[0]
    stream = torch.cuda.current_stream().cuda_stream
    # N = 1: C contexts, each with its own stream and image buffer; batch i goes to context i mod C. C launch sequences in flight:
    # the deep recursion levels of one (small, latency-bound kernels) overlap the primary kernels of the other (instruction-bound)
    C = 1
    if world == 1 and not sampled and args.contexts > 1 and args.steps >= args.contexts * B and B > 1:
        C = args.contexts
    ctxs, tstreams = [ctx], [torch.cuda.current_stream()]
    for _ in range(C - 1):
        c2 = pkg.Context(local_rank)
        c2.upload(scene)
        if args.dbg:
            pkg.hip.rtu_debug_flags(c2._h, args.dbg)
        ctxs.append(c2)
        tstreams.append(torch.cuda.Stream(device=dev))
        shards.append(torch.zeros_like(shards[0]))

    for cx in ctxs:  # (a hint: every context sizes its long-running kernels for its share of the machine)
        pkg.hip.rtu_set_sequences_in_flight(cx._h, C)

    def launch(frs, buf, k=0):
        if len(frs) == 1:
            ctxs[k].render_device(frs[0], buf.data_ptr(), tstreams[k].cuda_stream)
        else:
            ctxs[k].render_frames_device(frs, buf.data_ptr(), tstreams[k].cuda_stream)

    # -- untimed: rays of every camera of the turntable (this shard), from the counting variant -------------
    if sampled:  # settle the frame-record capacities first: the counting pass of a sampled frame does not re-provision
        ctx.render_device(frame0, shard.data_ptr(), stream)
        torch.cuda.synchronize()
    keys = None
    rays_cam = []
    for j, c in enumerate(cams):
        if j and (args.same_camera or sampled):
            rays_cam.append(rays_cam[0])
            continue
        if args.rays_per_frame:
            keys = ["primary_rays", "secondary_rays", "shadow_rays"]
            rays_cam.append([args.rays_per_frame // world, 0, 0])
            continue
        ctx.render_device(mkframe(c, collect_stats=True), shard.data_ptr(), stream)
        torch.cuda.synchronize()
        st = ctx.stats()
        keys = keys or sorted(st)
        rays_cam.append([st[k] for k in keys])
    tot = torch.tensor(rays_cam, dtype=torch.int64, device=cdev)
    if dist:
        dist.all_reduce(tot)
    per_cam = [dict(zip(keys, [int(v) for v in row])) for row in tot.tolist()]
    rays_of = [pkg.total_rays(d) for d in per_cam]

    pipe = sharding.FramePipeline(sends, gathers, dist, staged=args.rehearse, root_only=root_only) if dist else None

    pending = [None]          # out4: the batch whose pack + gather are still to be queued (done after the NEXT batch's kernels are)
    mmwork = [None, None]

    def finish():
        """out4: quantise and gather the batch rendered one step ago — by now its zmin / zmax all-reduce has long finished, so the
        stream never idles waiting for it."""
        if pending[0] is None:
            return
        i, nb = pending[0]
        pending[0] = None
        send = pipe.begin(i)  # waits (on the GPU) for the gather that last read this send buffer
        if mmwork[i & 1] is not None:
            mmwork[i & 1].wait()  # the current stream waits for the reduced keys (the host does not)
        ctx.pack_output_device(shards[i & 1].data_ptr(), rows * W, nb, minmax[i & 1].data_ptr(), send.data_ptr(), stream)
        pipe.gather(i)  # RCCL over xGMI, asynchronous

    def step(i, nb, ev=None):
        """One launch sequence: nb frames (steps) in flight."""
        if out4:
            if ev:
                ev[0].record()
            launch(frames[:nb], shards[i & 1])
            ctx.minmax_z_device(shards[i & 1].data_ptr(), rows * W, nb, minmax[i & 1].data_ptr(), stream)  # this shard's zmin / zmax keys per frame
            mmwork[i & 1] = sharding.allreduce_minmax(minmax[i & 1], dist, staged=args.rehearse)          # -> the frames' (MIN over the ranks)
            finish()                 # pack + gather of the PREVIOUS batch, queued behind this batch's kernels
            pending[0] = (i, nb)
            if ev:
                ev[1].record()
            return
        k = i % C
        buf = pipe.begin(i) if pipe else shards[k]  # waits (on the GPU) for the gather that last read this buffer
        if packed:
            send, buf = buf, shards[i & 1]
        if ev:
            ev[0].record(tstreams[k])
        launch(frames[:nb], buf, k)
        if packed:  # z of the nb frames, then their Color24 pixels (sharding.assemble_gathered_packed reads this layout)
            ctx.pack_image_device(buf.data_ptr(), nb * rows * W, send.data_ptr(), send.data_ptr() + B * max_rows * W * 4, stream)
        if ev:
            ev[1].record(tstreams[k])
        if pipe:
            pipe.gather(i)  # RCCL over xGMI, asynchronous: overlaps the next batch's kernels

    def settle(fn):
        """Warm up until no recursion level needs more frame records than provisioned (every rank repeats if any has to)."""
        for attempt in range(2 * 6 + 1):
            fn()
            finish()
            if pipe:
                pipe.drain()
            settled = 1
            for cx in ctxs:
                try:
                    cx.frame_status()
                except pkg.RtuError as e:
                    if e.code != pkg.RTU_ERR_CAPACITY or attempt == 2 * 6:
                        raise
                    settled = 0
            if dist:
                flag = torch.tensor([settled], dtype=torch.int32, device=cdev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                settled = int(flag[0])
            if settled:
                return
    settle(lambda: [step(j, B) for j in range(max(C, -(-args.warmup // B)))])  # warm-up at the full batch size (>= args.warmup frames; every context at least once)

    # -- untimed: what every kernel of the launch sequence touches (the fast variant counting itself) and how long it lasts
    # (a sampled frame is many launch sequences — one per batch of samples, ten per batch for recipe P —: its counters are sums over
    # all launches of a slot, `launches` says how many; bytes / rays / ms below are PER LAUNCH of the slot's kernel)
    kernels = {}
    dominant = None
    if True:
        tframes = [mkframe(c, collect_stats=2) for c in cams]
        launch(tframes, shard)
        torch.cuda.synchronize()
        ctx.frame_status()
        kernels = ctx.touched(textured)
        for name, k in kernels.items():
            nl = max(1, k.get("launches", 1))
            k["bytes_per_frame"], k["launches_per_frame"] = k["bytes"], nl
            k["bytes"], k["rays"] = k["bytes"] // nl, k["rays"] // nl
            ctx.probe_kernel(name)
            for _ in range(1 if sampled else 3):
                launch(frames, shard)
            ms, n = ctx.probe_read()  # (at most 64 launches are kept per read)
            k["ms"] = ms / max(n, 1)
        ctx.probe_kernel(None)
        # the dominant kernel: the one the launch sequences spend most time in (per launch x launches; recipe W: one launch each)
        dominant = max(kernels, key=lambda k: kernels[k]["ms"] * kernels[k]["launches_per_frame"])

    batches = [min(B, args.steps - i) for i in range(0, args.steps, B)]  # EXACTLY args.steps frames
    R = max(1, args.repeats if not sampled else min(args.repeats, 3))
    region_s, seq_ms, dom_sum, dom_n = [], [], 0.0, 0
    for rep in range(R):
        # one timed region: exactly args.steps frames, barrier + synchronize on both sides
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in batches]
        if dominant:
            ctx.probe_kernel(dominant)  # its launches are bracketed by HIP events (at most 64 are kept per read)
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j, nb in enumerate(batches):
            if dominant and nb != batches[0]:
                ctx.probe_kernel(None)  # the probe averages full batches only (a host-side flag: nothing is queued)
            step(j, nb, events[j])
        finish()
        if pipe:
            pipe.drain()  # every frame of the timed region rendered AND gathered
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        region_s.append(time.perf_counter() - t0)
        if dominant:
            ms, n = ctx.probe_read()
            dom_sum += ms
            dom_n += n
            ctx.probe_kernel(None)
        # HIP events on the launch stream, around the full launch sequences (B frames each; a shorter last batch is left out)
        seq_ms += [a.elapsed_time(b) for (a, b), nb in zip(events, batches) if nb == batches[0]]
    if dist:  # every rank's regions -> the slowest rank's time of each region (the barrier makes them nearly equal anyway)
        tr = torch.tensor(region_s, dtype=torch.float64, device=cdev)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        region_s = [float(v) for v in tr.tolist()]
    elapsed = float(np.median(region_s))
    gathered = pipe.last_gathered() if pipe else None
    dom_ms = dom_sum / max(dom_n, 1) if dominant else None
    kernel_ms = float(np.mean(seq_ms))
    for cx in ctxs:
        cx.frame_status()  # raises if a recursion level overflowed its provisioned capacity
    rays_total = sum(rays_of[j] for nb in batches for j in range(nb))

    seq_bytes = float(sum(k["bytes_per_frame"] for k in kernels.values()))  # of one launch sequence (recipe W) / of all sequences of one frame (sampled)
    dom_bytes = float(kernels[dominant]["bytes"]) if dominant else 0.0
    t = torch.tensor([elapsed, kernel_ms, seq_bytes, dom_ms or 0.0, dom_bytes], dtype=torch.float64, device=cdev)
    if dist:
        # the roofline is the slowest rank's
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        slow = max(allt, key=lambda x: float(x[1]))
        kernel_ms, seq_bytes, dom_ms, dom_bytes = (float(slow[i]) for i in (1, 2, 3, 4))

    # -- untimed: parity of what was just rendered (z of the golden camera's frame bit-exact vs the reference golden) ---
    img = None
    if dist:
        if rank == 0:
            if out4:
                rgb8, zimg8 = sharding.assemble_gathered_out4(pkg, gathered.view(world, -1).cpu().numpy(), 0, scene.desc.camera, W, H, world)
            elif packed:
                zimg, rgb8 = sharding.assemble_gathered_packed(pkg, gathered.view(world, -1).cpu().numpy(), 0, B, scene.desc.camera, W, H, world, max_rows)
                img = np.zeros((H, W, 4), np.float32)
                img[..., 3] = zimg
            else:
                img = sharding.assemble_gathered_batch(pkg, gathered.view(world, -1).cpu().numpy(), 0, scene.desc.camera, W, H, world)
    else:
        img = shard.view(-1)[:rows * W * 4].view(rows, W, 4).cpu().numpy()
    if args.rehearse and rank == 0:
        print("[rehearsal: %d ranks on one GPU through gloo — not a measurement]" % world, file=sys.stderr)
    import hashlib
    z_ok = img is not None and not sampled and hashlib.sha256(np.ascontiguousarray(img[..., 3]).tobytes()).hexdigest() == meta["sha256_z_f32"]
    if out4 and rank == 0 and not sampled and meta.get("sha256_zbuffer_u8"):
        # the gathered frame IS the two output images: ZBuffer.png's pixels must be the reference's bit for bit (its z-image is the
        # integer quantisation of the float z, frame-wide zmin / zmax included), Result.png's within one level
        gold = np.load(os.path.join(gdir, "golden.npz"))
        z_ok = (hashlib.sha256(np.ascontiguousarray(zimg8).tobytes()).hexdigest() == meta["sha256_zbuffer_u8"]
                and int(np.abs(rgb8.astype(np.int32) - gold["result_u8"].astype(np.int32)).max()) <= 1)

    # -- the single-frame configuration: one frame per launch sequence (what BeginRender() of one image costs) ----------
    single = None
    if not sampled and world == 1 and B > 1:
        n1 = max(20, min(100, args.steps))
        pkg.hip.rtu_set_sequences_in_flight(ctx._h, 1)  # (one frame at a time on one context)
        settle(lambda: [launch([frame0], shard) for _ in range(5)])  # (the cut level of the tail kernel is learned per launch shape)
        els = []
        for _ in range(5):  # (median of five runs of n1 frames: one run is a 7 - 37 ms sample, a host hiccup away from 15 % off)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(n1):
                launch([frame0], shard)
            torch.cuda.synchronize()
            els.append(time.perf_counter() - ts)
        el1 = float(np.median(els))
        ctx.frame_status()
        single = {"frames_in_flight": 1, "frames": n1, "runs": len(els), "ms_per_frame": round(el1 / n1 * 1e3, 4),
                  "mrays_per_s": round(rays_of[0] * n1 / el1 / 1e6, 1)}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_total / elapsed / 1e6
        peak, peak_src = hbm_peak(pkg, local_rank)
        roof = {"bound": "unprofiled", "achieved": None, "peak": round(peak, 1), "unit": "GB/s", "frac": None, "traffic": None, "peak_source": peak_src}
        if dominant:
            achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
            kname = dominant
            # what the committed PMC profile of this command says about the dominant kernel (profile-derived, not measured in this run)
            prof = kernel_profile() if (world == 1 and args.tag == WORKLOAD_TAG) else {}
            pk = (prof.get("per_kernel", {}).get(kname) or {})
            scale = batches[0] / prof["frames_in_flight"] if prof.get("frames_in_flight") else 1.0  # profiled at another batch size: per frame in flight
            traffic = int(pk["hbm_bytes_per_launch"] * scale) if pk.get("hbm_bytes_per_launch") else None
            hbm_cf = round(pk["hbm_bytes_per_launch"] / (pk["us"] * 1e-6) / 1e9 / peak, 4) if pk.get("hbm_bytes_per_launch") and pk.get("us") else None
            bound = "unprofiled"
            if hbm_cf is not None:
                # the PMC counters decide the label: a kernel that moves less than half of what HBM could in its time is not HBM-bound
                bound = "hbm" if hbm_cf >= 0.5 else "valu-issue/divergence (working set cache-resident: lanes_active, valu_issue_frac)"
            roof = {"bound": bound, "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "GB/s", "frac": round(achieved / peak, 5),
                    "frac_is": "SURVEY 8d's nominal figure: cache-agnostic touched bytes / kernel time / HBM peak — how fast the kernel goes through its working "
                               "set, which is cache-resident; what binds the kernel is in bound / hbm_counter_frac / valu_issue_frac / lanes_active",
                    "peak_source": peak_src,
                    "traffic": traffic,
                    "hbm_counter_frac": hbm_cf, "valu_issue_frac": pk.get("valu_issue_frac"), "lanes_active": pk.get("lanes_active"),
                    "waves_waiting_frac": pk.get("waiting_frac"),
                    "profile_source": ("%s: rocprofv3 --kernel-trace --pmc passes of `%s` at commit %s with %s frames in flight (traffic scaled to this run's %d); "
                                       "profile-derived, not measured in this run" % (prof.get("file"), prof.get("command", "bench.py"), prof.get("commit", "?"),
                                                                                       prof.get("frames_in_flight"), batches[0])) if pk else None,
                    "kernel": kname, "kernel_ms": round(dom_ms, 4), "kernel_launches_timed": dom_n, "algorithmic_bytes_per_launch": int(dom_bytes),
                    "l2_peak": L2_PEAK_GBS, "l2_frac": round(achieved / L2_PEAK_GBS, 5),
                    "bytes_are": "touched by the timed (fast) kernels themselves, counted per kernel by collect_stats=2 (rtu_render.h RtuTouched)",
                    "sequence": {"kernels": len(kernels), "ms": round(kernel_ms, 4), "algorithmic_bytes": int(seq_bytes),
                                 "achieved": round(seq_bytes / (kernel_ms * 1e-3) / 1e9, 2),
                                 "frac": round(seq_bytes / (kernel_ms * 1e-3) / 1e9 / peak, 5)},
                    "kernels": {k: dict({"ms": round(v["ms"], 4), "bytes": v["bytes"], "rays": v["rays"],
                                         "GBps": round(v["bytes"] / max(v["ms"], 1e-6) / 1e6, 1)},
                                        **({"launches_per_frame": v["launches_per_frame"]} if sampled else {}))
                                for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"] * kv[1]["launches_per_frame"])}}
            if sampled:
                roof["per_launch"] = "a frame of %d samples is %d launches of the dominant kernel; kernel_ms, achieved and the bytes are per launch" % (
                    args.samples, kernels[dominant]["launches_per_frame"])
            if achieved > peak:  # touched bytes served from cache faster than HBM could: say so instead of printing a "fraction" > 1
                roof["exceeds_hbm_peak"] = True
                print("roofline: %.0f GB/s of touched bytes exceeds the HBM peak — the dominant kernel is cache-bound, read l2_frac" % achieved, file=sys.stderr)
        out = {
            "metric": "Mrays/sec at 1920x1080 (the reference's primary + secondary + shadow rays of the frames rendered / wall time; batched throughput: "
                      "config.frames_in_flight frames with their own cameras per launch sequence — one frame alone: config.single_frame)",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "the reference's own scene (flattened blob tests/golden/%s), no synthetic inputs" % args.tag,
            "config": {"workload": (WORKLOAD_NAME if args.tag == WORKLOAD_TAG else args.tag) + (" recipe %s, %d samples per pixel" % ("P" if args.paths else "S", args.samples) if sampled else ""),
                       "width": W, "height": H,
                       "rays_per_frame": rays_of[0], "primary": per_cam[0]["primary_rays"],
                       "secondary": per_cam[0]["secondary_rays"], "shadow": per_cam[0]["shadow_rays"],
                       "rays_in_timed_region": rays_total,
                       "frames_in_flight": batches[0],
                       "cameras": "one per frame of a batch: the scene's camera orbited by %g degrees per frame (frame 0 = the golden camera)" % ORBIT_STEP_DEG
                                  if len(set(rays_of)) > 1 or not (args.same_camera or sampled) else "the scene's camera for every frame",
                       "batch_latency_ms": round(kernel_ms, 4),  # HIP-event time of ONE launch sequence of frames_in_flight frames
                       "launch_sequences_in_flight": C,  # contexts / streams the batches alternate over (the deep levels of one sequence overlap the primary kernels of the others)
                       "repeats": {"n": R, "region_ms_median": round(elapsed * 1e3, 4), "region_ms_min": round(min(region_s) * 1e3, 4),
                                   "region_ms_max": round(max(region_s) * 1e3, 4), "region_ms_first": round(region_s[0] * 1e3, 4),
                                   "timed_ms_total": round(sum(region_s) * 1e3, 2),
                                   "is": "the K-step region (exactly `steps` frames, barrier + synchronize on both sides) run n times back to back; value and ms_per_step are the median region's"},
                       "single_frame": single,
                       "sharding": ("interleaved 8-row bands, RCCL gather of the %s to rank 0, overlapped with the next batch" % (
                           "two output images (Color24 + z-image byte, 4 B per pixel, made on the device after one all-reduce of the frames' zmin / zmax)" if out4
                           else "RenderImage arrays (float z + Color24, 7 B per pixel, packed on the device)" if packed else "float4 framebuffer")) if world > 1 else "single GPU",
                       "z_bit_exact_vs_reference_golden": bool(z_ok)},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(g, scene, W, H, rays_of[0], args.cpu_seconds, args.cpu_threads, args.samples, args.paths)
        print(json.dumps(out), flush=True)
    for cx in ctxs:
        cx.close()
    if dist:
        dist.destroy_process_group()


def cpu_baseline(g, scene, W, H, rays_per_frame, budget_s, threads, samples=0, paths=False):
    """The CPU oracle (a port: bit-identical restatement of the reference's Trace/Shade, see oracle/rtu_oracle.cpp) on
    this box's host cores, same workload (the golden camera's frame), whole frames: one on 1 thread, then on every core
    under both work distributions of SURVEY 8d — the reference's PixelIterator (one atomic fetch per pixel,
    PixelIterator.h:25-38) and chunks of rows — about half of the remaining budget each. value = the faster one."""
    orc = g.load_oracle()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # the cores this process may actually USE: a container's CPU quota (cgroup cpu.max) can be far below the threads it sees —
    # the GPU box shows 256 hardware threads and grants the time of 16; more threads than that only contend
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, -(-int(q) // int(per)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, -(-q // per))
        except (OSError, ValueError):
            pass
    usable = min(avail, quota) if quota else avail
    cores = max(1, threads if threads > 0 else usable)

    def timed(fn, budget):
        frames, t0 = 0, time.perf_counter()
        while True:
            fn()
            frames += 1
            el = time.perf_counter() - t0
            if el > budget or frames >= 500:
                return frames, el
    if samples:
        render = (lambda th: orc.render_paths(scene, W, H, samples, threads=th)) if paths else (lambda th: orc.render_samples(scene, W, H, samples, threads=th))
        t0 = time.perf_counter()
        render(1 if not paths else cores)
        t1 = time.perf_counter() - t0
        frames, el = timed(lambda: render(cores), max(1.0, budget_s - t1))
        return {"value": round(rays_per_frame * frames / el / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                "sample": "%d full %dx%d frames on %d threads (row-chunk schedule)" % (frames, W, H, cores), "ms_per_frame": round(el / frames * 1e3, 3)}
    t0 = time.perf_counter()
    orc.render(scene, W, H, threads=1)
    t1 = time.perf_counter() - t0
    left = max(2.0, budget_s - t1)
    fr, elr = timed(lambda: orc.render_scheduled(scene, W, H, cores, False), left / 2)
    fp, elp = timed(lambda: orc.render_scheduled(scene, W, H, cores, True), left / 2)
    rate_r, rate_p = rays_per_frame * fr / elr / 1e6, rays_per_frame * fp / elp / 1e6
    best_ms = min(elr / fr, elp / fp) * 1e3
    return {"value": round(max(rate_r, rate_p), 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "host_threads_visible": avail, "cpu_quota_cores": quota,
            "sample": "full %dx%d frames of the same workload (golden camera): 1 on 1 thread (%.3f Mrays/s), %d on %d threads in row chunks, "
                      "%d on %d threads with the reference's per-pixel atomic schedule (PixelIterator.h:25-38)" % (W, H, rays_per_frame / t1 / 1e6, fr, cores, fp, cores),
            "row_chunk_schedule": {"mrays_per_s": round(rate_r, 3), "ms_per_frame": round(elr / fr * 1e3, 3)},
            "per_pixel_atomic_schedule": {"mrays_per_s": round(rate_p, 3), "ms_per_frame": round(elp / fp * 1e3, 3)},
            "one_thread_mrays_per_s": round(rays_per_frame / t1 / 1e6, 3),
            "ms_per_frame": round(best_ms, 3)}


if __name__ == "__main__":
    main()
