"""world_size-2 (and 3) gloo test of the multi-GPU path on CPU: shard arithmetic,
padded all_gather of the float4 framebuffer, de-interleave. The renderer of each rank
is stood in for by the oracle (test infrastructure) rendering exactly the rows the
rank's GPU would render, so the assembled frame must equal the single-process frame."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tag, out_dir):
    import sys
    import torch
    import torch.distributed as dist
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import __graft_entry__ as g
    pkg, orc = g.load_package(), g.load_oracle()
    spec = __import__("importlib.util").util.spec_from_file_location(
        "rtu_sharding", os.path.join(repo, "raytracer-utah_amd", "sharding.py"))
    sharding = __import__("importlib.util").util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    gdir = os.path.join(repo, "tests", "golden", tag)
    scene = pkg.Scene.from_blob_file(os.path.join(gdir, "scene.rtus.gz"))
    W, H = scene.desc.camera.img_width, scene.desc.camera.img_height
    frame = pkg.frame_setup(scene.desc.camera, W, H, shard_rank=rank, shard_count=world)
    rows = pkg.shard_global_rows(frame)
    max_rows = pkg.hip.rtu_shard_max_rows(H, world)
    shard = torch.zeros(max_rows * W * 4, dtype=torch.float32)
    buf = shard.view(max_rows, W, 4).numpy()
    for lr in range(0, len(rows), 8):  # band by band, as the kernel's grid does
        n = min(8, len(rows) - lr)
        part, _ = orc.render(scene, W, H, threads=1, row0=int(rows[lr]), nrows=n)
        buf[lr:lr + n] = part
    gathered = torch.empty(world * max_rows * W * 4, dtype=torch.float32)
    sharding.gather_framebuffer(shard, gathered, dist)
    # the double-buffered loop of bench.py: three frames (the third reuses buffer 0), every gather identical
    for root_only in (True, False):
        shards = [shard.clone(), shard.clone()]
        gathers = None
        if rank == 0 or not root_only:
            gathers = [torch.empty(world, max_rows * W * 4), torch.empty(world, max_rows * W * 4)]
        pipe = sharding.FramePipeline(shards, gathers, dist, root_only=root_only)
        for i in range(3):
            b = pipe.begin(i)
            assert b is shards[i & 1]
            b.copy_(shard)  # "render"
            pipe.gather(i)
        pipe.drain()
        if gathers is not None:
            assert pipe.last_gathered() is gathers[0]
            assert torch.equal(gathers[0].view(-1), gathered) and torch.equal(gathers[1].view(-1), gathered)
        else:
            assert pipe.last_gathered() is None
    # frames in flight (bench.py's default): B shards per rank in one buffer, frame j at j * rows_r * W float4,
    # one gather per batch; frame 1 of the batch carries a marker so that a wrong offset cannot go unnoticed
    B = 3
    n_r = len(rows) * W * 4
    batch = torch.zeros(B * max_rows * W * 4, dtype=torch.float32)
    for j in range(B):
        batch[j * n_r:(j + 1) * n_r] = shard[:n_r] + (1000.0 if j == 1 else 0.0)
    gb = [torch.empty(world, B * max_rows * W * 4), torch.empty(world, B * max_rows * W * 4)] if rank == 0 else None
    pipe = sharding.FramePipeline([batch, batch.clone()], gb, dist, root_only=True)
    pipe.begin(0)
    pipe.gather(0)
    pipe.drain()
    if rank == 0:
        for j in range(B):
            fj = sharding.assemble_gathered_batch(pkg, pipe.last_gathered().numpy(), j, scene.desc.camera, W, H, world)
            np.save(os.path.join(out_dir, "batch%d.npy" % j), fj)
    # what bench.py gathers by default: the packed RenderImage content (float z + Color24, 7 bytes per pixel), B frames
    # in flight; the device's pack kernel is stood in for by the oracle's post-process (same arithmetic on the host)
    rgb8, _, _ = orc.postprocess(buf[:len(rows)])
    pb = sharding.packed_bytes(B, max_rows, W)
    send = np.zeros(pb, np.uint8)
    zview = send[:B * max_rows * W * 4].view(np.float32)
    for j in range(B):
        zview[j * len(rows) * W:(j + 1) * len(rows) * W] = buf[:len(rows), :, 3].reshape(-1) + (j == 2) * 7.0
        o = B * max_rows * W * 4 + j * len(rows) * W * 3
        send[o:o + len(rows) * W * 3] = rgb8.reshape(-1)
    st = torch.from_numpy(send)
    gp = [torch.empty(world, pb, dtype=torch.uint8), torch.empty(world, pb, dtype=torch.uint8)] if rank == 0 else None
    pipe = sharding.FramePipeline([st, st.clone()], gp, dist, root_only=True)
    pipe.begin(0)
    pipe.gather(0)
    pipe.drain()
    if rank == 0:
        for j in (0, 2):
            zj, rj = sharding.assemble_gathered_packed(pkg, pipe.last_gathered().numpy(), j, B, scene.desc.camera, W, H, world, max_rows)
            np.save(os.path.join(out_dir, "packz%d.npy" % j), zj)
            np.save(os.path.join(out_dir, "packrgb%d.npy" % j), rj)
    lo, hi = sharding.global_minmax_z(shard.view(max_rows, W, 4)[:len(rows), :, 3], dist, torch)
    img = sharding.assemble_gathered(pkg, gathered.view(world, max_rows, W, 4).numpy(), scene.desc.camera, W, H, world)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), img)
    np.save(os.path.join(out_dir, "minmax%d.npy" % rank), np.float32([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gather_reassembles_the_frame(pkg, orc, golden, tmp_path, world):
    import torch.multiprocessing as mp
    tag = "teapot2_240x135"
    port = _free_port()
    mp.spawn(_worker, args=(world, port, tag, str(tmp_path)), nprocs=world, join=True)
    g = golden(tag)
    ref, _ = orc.render(g.scene(pkg), g.width, g.height, threads=2)
    z = ref[..., 3]
    for r in range(world):
        img = np.load(tmp_path / ("rank%d.npy" % r))
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), "rank %d assembled a different frame" % r
        lo, hi = np.load(tmp_path / ("minmax%d.npy" % r))
    ref8, _, _ = orc.postprocess(ref)
    for j in (0, 2):
        assert np.array_equal(np.load(tmp_path / ("packz%d.npy" % j)), ref[..., 3] + np.float32(7.0 if j == 2 else 0.0))
        assert np.array_equal(np.load(tmp_path / ("packrgb%d.npy" % j)), ref8)
    for j in range(3):
        fj = np.load(tmp_path / ("batch%d.npy" % j))
        assert np.array_equal(fj, ref + np.float32(1000.0 if j == 1 else 0.0)), "frame %d of the gathered batch" % j
        assert lo == z[z != np.float32(1e30)].min() and hi == z[z != np.float32(1e30)].max()


def test_output_image_chunks_reassemble(tmp_path):
    """sharding.assemble_gathered_out4 (the layout of the default multi-GPU gather: 4 bytes per pixel, frame j of rank r at byte
    offset j * rows_r * W * 4) puts every band of every rank back where it belongs, for band counts that do not divide evenly."""
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import __graft_entry__ as g
    pkg = g.load_package()
    spec = __import__("importlib.util").util.spec_from_file_location("rtu_sharding", os.path.join(repo, "raytracer-utah_amd", "sharding.py"))
    sharding = __import__("importlib.util").util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    scene = pkg.Scene.from_blob_file(os.path.join(repo, "tests", "golden", "teapot2_240x135", "scene.rtus.gz"))
    W, H, B = 240, 135, 3
    rng = np.random.default_rng(5)
    images = rng.integers(0, 256, size=(B, H, W, 4), dtype=np.uint8)
    for world in (1, 2, 3, 8, 24):
        max_rows = pkg.hip.rtu_shard_max_rows(H, world)
        assert sharding.out4_bytes(B, max_rows, W) == B * max_rows * W * 4
        chunks = []
        for r in range(world):
            fr = pkg.frame_setup(scene.desc.camera, W, H, shard_rank=r, shard_count=world)
            rows = pkg.shard_global_rows(fr)
            chunk = np.zeros(sharding.out4_bytes(B, max_rows, W), np.uint8)
            n = len(rows) * W * 4
            for j in range(B):
                chunk[j * n:(j + 1) * n] = images[j][rows].reshape(-1)
            chunks.append(chunk)
        for j in range(B):
            rgb, zimg = sharding.assemble_gathered_out4(pkg, chunks, j, scene.desc.camera, W, H, world)
            assert np.array_equal(rgb, images[j][..., :3]) and np.array_equal(zimg, images[j][..., 3])
