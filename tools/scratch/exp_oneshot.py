"""Experiment: the driver's bench call renders 20 frames once (--steps 20). One launch sequence of 20, or K sequences of 20/K on K
streams (K contexts): wall time from the first launch to the last kernel done, best and median of 15 repetitions."""
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
N = 20
for K in (1, 2, 4, 5):
    B = N // K
    ctxs = [pkg.Context(0) for _ in range(K)]
    for c in ctxs:
        c.upload(scene)
    streams = [torch.cuda.Stream() for _ in range(K)]
    bufs = [torch.zeros(B * W * H * 4, dtype=torch.float32, device="cuda") for _ in range(K)]
    cams = [bench.orbit_camera(scene.desc.camera, 2.0 * j) for j in range(N)]
    frames = [pkg.frame_setup(c, W, H) for c in cams]
    def run():
        for k in range(K):
            ctxs[k].render_frames_device(frames[k * B:(k + 1) * B], bufs[k].data_ptr(), streams[k].cuda_stream)
    for attempt in range(5):
        run()
        torch.cuda.synchronize()
        ok = True
        for c in ctxs:
            try:
                c.frame_status()
            except pkg.RtuError:
                ok = False
        if ok and attempt >= 1:
            break
    ts = []
    for rep in range(15):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    for c in ctxs:
        c.frame_status()
    ts = np.array(ts) * 1e3
    print("%d sequences x %2d frames: best %.3f ms  median %.3f ms  -> %.4f ms/frame %.1f Grays/s (median)" % (
        K, B, ts.min(), np.median(ts), np.median(ts) / N, rays * N / np.median(ts) / 1e6), flush=True)
    for c in ctxs:
        c.close()
    del bufs
    torch.cuda.empty_cache()
