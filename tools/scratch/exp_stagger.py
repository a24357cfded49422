"""Experiment: the driver's 20 frames as S sub-batches alternating over C contexts / streams, each sub-batch's stream waiting
for the PREVIOUS sub-batch's primary kernels (rtu_primary_done_event): software pipelining inside one timed region."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import __graft_entry__ as g
import bench
pkg = g.load_package()
tag = "teapot2_1080"
gdir = os.path.join("tests", "golden", tag)
meta = json.load(open(os.path.join(gdir, "meta.json")))
W, H = meta["width"], meta["height"]
scene = pkg.Scene.from_blob_file(os.path.join(gdir, "scene.rtus.gz"))
rays = 2558987
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cams = [bench.orbit_camera(scene.desc.camera, 2.0 * j) for j in range(N)]
frames = [pkg.frame_setup(c, W, H) for c in cams]
for C, sizes, stagger in [(1, [N], False), (2, [N // 2, N - N // 2], False), (2, [N // 2, N - N // 2], True), (2, [12, 8], True), (2, [14, 6], True),
                          (2, [7, 7, 6], True), (2, [5, 5, 5, 5], True), (3, [7, 7, 6], True), (2, [8, 6, 6], True), (2, [10, 6, 4], True)]:
    if sum(sizes) != N:
        continue
    ctxs = [pkg.Context(0) for _ in range(C)]
    for c in ctxs:
        c.upload(scene)
    streams = [torch.cuda.Stream() for _ in range(C)]
    bufs = [torch.zeros(max(sizes) * W * H * 4, dtype=torch.float32, device="cuda") for _ in range(len(sizes))]
    def run():
        off = 0
        for i, n in enumerate(sizes):
            k = i % C
            if stagger and i > 0:
                pk = (i - 1) % C
                pkg.hip.rtu_stream_wait_event(ctxs[k]._h, streams[k].cuda_stream, pkg.hip.rtu_primary_done_event(ctxs[pk]._h))
            ctxs[k].render_frames_device(frames[off:off + n], bufs[i].data_ptr(), streams[k].cuda_stream)
            off += n
    for attempt in range(6):
        run()
        torch.cuda.synchronize()
        ok = True
        for c in ctxs:
            try:
                c.frame_status()
            except pkg.RtuError:
                ok = False
        if ok and attempt >= 2:
            break
    ts = []
    for rep in range(25):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    for c in ctxs:
        c.frame_status()
    ts = np.array(ts) * 1e3
    print("%d contexts, sub-batches %s, stagger %s: best %.3f ms  median %.3f ms -> %.1f Grays/s" % (C, sizes, stagger, ts.min(), np.median(ts), rays * N / np.median(ts) / 1e6), flush=True)
    for c in ctxs:
        c.close()
    del bufs
    torch.cuda.empty_cache()
