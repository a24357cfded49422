"""Several GPUs behind ONE handle of the C-ABI (include/rtu_render.h "Several GPUs", csrc/rtu_multi.hip): what
SpawnRenderThreads (main.cpp:29-64) does with CPU workers. The one-GPU box exercises it with device_ids = {0, 0, ...}: N contexts
on GPU 0, each rendering its interleaved 8-row bands, collected concurrently and de-interleaved under the ABI. The assembled
frame must be the single-context frame bit for bit; progress arrives band by band; the cancel word ends a sampled frame between
sample batches (StopRender(), main.cpp:70-72; RenderImage::IncrementNumRenderPixel, scene.h:585-588)."""
import ctypes
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,n_ctx", [("teapot2_240x135", 3), ("p4_240x135", 2), ("p13_200x150", 5), ("teapot2_240x135", 24)])
def test_multi_context_frame_equals_single_context(pkg, golden, tag, n_ctx):
    g = golden(tag)
    scene = g.scene(pkg)
    W, H = g.width, g.height
    one = pkg.Context(0)
    one.upload(scene)
    ref, _ = one.render(pkg.frame_setup(scene.desc.camera, W, H))
    one.close()
    m = pkg.MultiContext([0] * n_ctx)  # (24 contexts > the 17 bands of a 135-row image: some shards are empty)
    try:
        assert pkg.hip.rtu_multi_size(m._h) == n_ctx and m.context_handle(0) and not m.context_handle(n_ctx)
        m.upload(scene)
        bands = []
        img = m.render(pkg.frame_setup(scene.desc.camera, W, H), on_rows=lambda row0, n: bands.append((row0, n)))
        assert m.gather_kind() == 2  # several contexts on one GPU: concurrent asynchronous copies
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), "the sharded, gathered frame differs from the single-context frame"
        # every row exactly once, in whole bands
        rows = sorted(r for row0, n in bands for r in range(row0, row0 + n))
        assert rows == list(range(H))
        assert all(row0 % 8 == 0 and (n == 8 or row0 + n == H) for row0, n in bands)
        # shard by shard as the transfers complete: context 0's bands first (0, n_ctx, 2 n_ctx, ...)
        assert [row0 // 8 for row0, _ in bands[:2]] == [0, n_ctx][:len(bands[:2])] or n_ctx * 8 >= H
    finally:
        m.close()


def test_multi_context_single_device_and_forced_rccl(pkg, golden, monkeypatch):
    """One context: gather kind 1. RTU_FORCE_RCCL sends the single context through the RCCL branch — librccl.so found and
    resolved at run time, a communicator of one, an (empty) group opened AND closed — the frame arrives through the same
    copies. (The grouped send / receive between distinct GPUs has not run on hardware: INTEGRATION.md.)"""
    g = golden("teapot2_240x135")
    scene = g.scene(pkg)
    W, H = g.width, g.height
    m = pkg.MultiContext([0])
    try:
        m.upload(scene)
        a = m.render(pkg.frame_setup(scene.desc.camera, W, H))
        assert m.gather_kind() == 1
        monkeypatch.setenv("RTU_FORCE_RCCL", "1")
        b = m.render(pkg.frame_setup(scene.desc.camera, W, H))
        assert m.gather_kind() == 3, "the RCCL path was not taken"
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        z = np.ascontiguousarray(a[..., 3])
        assert np.array_equal(z.view(np.uint32), np.asarray(g.npz["z"], np.float32).reshape(H, W).view(np.uint32)), "float z differs from the reference golden"
    finally:
        m.close()


def test_multi_context_sampled_frame_and_cancel(pkg, golden):
    """A sampled frame (recipe S) on three contexts equals the single-context frame bit for bit (keys use the pixel of the whole
    image); the cancel word raised from another thread ends a long sampled frame with RTU_ERR_CANCELLED between sample batches,
    and the handle renders again afterwards."""
    g = golden("teapot2_240x135")
    scene = g.scene(pkg)
    W, H = g.width, g.height
    one = pkg.Context(0)
    one.upload(scene)
    ref, _ = one.render(pkg.frame_setup(scene.desc.camera, W, H, samples=4))
    one.close()
    m = pkg.MultiContext([0, 0, 0])
    try:
        m.upload(scene)
        img = m.render(pkg.frame_setup(scene.desc.camera, W, H, samples=4))
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
        # already raised: nothing is rendered
        flag = ctypes.c_int(1)
        with pytest.raises(pkg.RtuError) as e:
            m.render(pkg.frame_setup(scene.desc.camera, W, H, samples=4), cancel=flag)
        assert e.value.code == pkg.RTU_ERR_CANCELLED
        # raised while a long frame is being rendered (4096 samples at 1920x1080 would take seconds)
        flag = ctypes.c_int(0)
        scene.set_resolution(1920, 1080)
        m.upload(scene)
        t = threading.Timer(0.3, lambda: setattr(flag, "value", 1))
        t0 = time.perf_counter()
        t.start()
        with pytest.raises(pkg.RtuError) as e:
            m.render(pkg.frame_setup(scene.desc.camera, 1920, 1080, samples=4096), cancel=flag)
        t.cancel()
        assert e.value.code == pkg.RTU_ERR_CANCELLED
        assert time.perf_counter() - t0 < 20.0
        # the handle is still good
        flag.value = 0
        scene.set_resolution(W, H)
        m.upload(scene)
        again = m.render(pkg.frame_setup(scene.desc.camera, W, H, samples=4), cancel=flag)
        assert np.array_equal(again.view(np.uint32), ref.view(np.uint32))
    finally:
        m.close()


def test_device_info_is_what_the_machine_says(pkg):
    info = pkg.device_info(0)
    assert info["arch"].startswith("gfx950") and info["compute_units"] >= 200
    assert info["memory_clock_khz"] > 0 and info["memory_bus_bits"] >= 1024
