"""Multi-GPU plumbing: one process per GPU, the frame sharded by interleaved 8-row
bands, one gather of the float4 framebuffer per frame over
torch.distributed (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU
tests). No data-path collective besides that gather: the scene (<= 0.5 MB) is replicated."""
import numpy as np


def gather_framebuffer(shard, gathered, dist):
    """All ranks contribute their padded shard [max_rows*W*4]; `gathered` is
    [world*max_rows*W*4] on every rank."""
    dist.all_gather_into_tensor(gathered, shard)
    return gathered


class FramePipeline:
    """Double-buffered frame loop of one rank: frame i is rendered into shard buffer i % 2 and its
    gather TO RANK 0 (every GPU writes its 1/N of the frame to the root over its own
    xGMI link — grouped send/recv in RCCL, not a ring) is issued asynchronously, so the transfer
    overlaps the kernels of frame i+1, which write the other buffer. begin(i) makes the current
    stream wait for the gather that last read buffer i % 2 (frame i-2); drain() waits for everything
    in flight. gathers[b] exists on rank 0 only ([world, n]); root_only=False falls back to
    all_gather (every rank receives the frame). staged=True (rehearsal on gloo): host copies."""

    def __init__(self, shards, gathers, dist, staged=False, root_only=True):
        self.shards, self.gathers, self.dist, self.staged, self.root_only = shards, gathers, dist, staged, root_only
        self.rank = dist.get_rank()
        self.work = [None, None]
        self.last = None

    def begin(self, i):
        b = i & 1
        if self.work[b] is not None:
            self.work[b].wait()
            self.work[b] = None
        return self.shards[b]

    def gather(self, i):
        b = i & 1
        src = self.shards[b].cpu() if self.staged else self.shards[b]
        if self.root_only:
            dst = list(self.gathers[b].unbind(0)) if self.rank == 0 else None
            self.work[b] = self.dist.gather(src, gather_list=dst, dst=0, async_op=True)
        else:
            self.work[b] = self.dist.all_gather_into_tensor(self.gathers[b].view(-1), src, async_op=True)
        self.last = b

    def drain(self):
        for b in (0, 1):
            if self.work[b] is not None:
                self.work[b].wait()
                self.work[b] = None

    def last_gathered(self):
        """[world, n] on rank 0 (on every rank with root_only=False)."""
        return self.gathers[self.last] if self.gathers is not None else None


def assemble_gathered(pkg, gathered_np, camera, width, height, world):
    """De-interleave a gathered [world, max_rows, W, 4] array into the [H, W, 4] frame."""
    frames = [pkg.frame_setup(camera, width, height, shard_rank=r, shard_count=world) for r in range(world)]
    return pkg.assemble([gathered_np[r] for r in range(world)], frames, height)


def assemble_gathered_batch(pkg, chunks_np, j, camera, width, height, world):
    """Frames in flight: rank r's gathered chunk holds its shard of every frame of the batch, frame j at
    float offset j * rows_r * W * 4 (rows_r = the rows of rank r's shard, not the padded maximum).
    chunks_np: [world, >= B * rows_r * W * 4]. Returns frame j as [H, W, 4]."""
    import numpy as np
    frames = [pkg.frame_setup(camera, width, height, shard_rank=r, shard_count=world) for r in range(world)]
    parts = []
    for r in range(world):
        rows_r = pkg.shard_rows(frames[r])
        n = rows_r * width * 4
        parts.append(np.asarray(chunks_np[r][j * n:(j + 1) * n]).reshape(rows_r, width, 4))
    return pkg.assemble(parts, frames, height)


def packed_bytes(B, max_rows, width):
    """Bytes of one rank's gather buffer for B frames in flight of packed images (float z + Color24)."""
    return B * max_rows * width * 7


def assemble_gathered_packed(pkg, chunks_u8, j, B, camera, width, height, world, max_rows):
    """The multi-GPU gather moves the reference's RenderImage content, 7 bytes per pixel: rank r's chunk (uint8) holds
    the float z of its shard of every frame of the batch (frame j at float offset j * rows_r * W) in its first
    B * max_rows * W * 4 bytes, then the Color24 pixels (frame j at byte offset j * rows_r * W * 3).
    Returns frame j as (z [H, W] float32, rgb8 [H, W, 3] uint8)."""
    import numpy as np
    z = np.empty((height, width), np.float32)
    rgb = np.empty((height, width, 3), np.uint8)
    zbytes = B * max_rows * width * 4
    for r in range(world):
        fr = pkg.frame_setup(camera, width, height, shard_rank=r, shard_count=world)
        rows = pkg.shard_global_rows(fr)
        n = len(rows) * width
        c = np.ascontiguousarray(chunks_u8[r])
        z[rows] = c[:zbytes].view(np.float32)[j * n:(j + 1) * n].reshape(len(rows), width)
        rgb[rows] = c[zbytes + j * n * 3:zbytes + (j + 1) * n * 3].reshape(len(rows), width, 3)
    return z, rgb


def global_minmax_z(z_local, dist, torch):
    """z-image normalisation needs the frame-wide zmin/zmax (scene.h:596-601): a 2-float
    all-reduce when the frame is not gathered to one place."""
    hit = z_local[z_local != 1.0e30]
    lo = hit.min() if hit.numel() else torch.tensor(1.0e30, dtype=z_local.dtype, device=z_local.device)
    hi = hit.max() if hit.numel() else torch.tensor(0.0, dtype=z_local.dtype, device=z_local.device)
    lo, hi = lo.clone().reshape(1), hi.clone().reshape(1)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return float(lo), float(hi)


def out4_bytes(B, max_rows, width):
    """Bytes of one rank's gather buffer for B frames in flight of output images (Color24 + z-image byte: 4 per pixel)."""
    return B * max_rows * width * 4


def assemble_gathered_out4(pkg, chunks_u8, j, camera, width, height, world):
    """The default multi-GPU gather moves the two OUTPUT images, 4 bytes per pixel {r, g, b, z-image byte}
    (rtu_pack_output_device): rank r's chunk (uint8) holds its shard of frame j at byte offset j * rows_r * W * 4.
    Returns frame j as (rgb8 [H, W, 3] uint8 = the pixels of Result.png, zimg [H, W] uint8 = those of ZBuffer.png)."""
    import numpy as np
    rgb = np.empty((height, width, 3), np.uint8)
    zimg = np.empty((height, width), np.uint8)
    for r in range(world):
        fr = pkg.frame_setup(camera, width, height, shard_rank=r, shard_count=world)
        rows = pkg.shard_global_rows(fr)
        n = len(rows) * width * 4
        px = np.ascontiguousarray(chunks_u8[r])[j * n:(j + 1) * n].reshape(len(rows), width, 4)
        rgb[rows] = px[..., :3]
        zimg[rows] = px[..., 3]
    return rgb, zimg


def allreduce_minmax(mm, dist, staged=False):
    """Element-wise MIN over the ranks of the int64 zmin / zmax keys of rtu_minmax_z_device (the max is stored complemented), in
    place, asynchronously: returns the work handle (wait() makes the current stream wait, not the host). staged (rehearsal on
    gloo): through a host copy, synchronously."""
    if staged:
        h = mm.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MIN)
        mm.copy_(h)
        return None
    return dist.all_reduce(mm, op=dist.ReduceOp.MIN, async_op=True)
