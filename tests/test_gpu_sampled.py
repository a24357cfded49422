"""GPU: recipe S — stochastic effects (soft shadows, glossy bounces, depth of field,
multi-sample pixels) through the C-ABI against the oracle on the keyed sample streams.

The chain of evidence: reference (rand() wrapped) == oracle[sequential stream, libm trig] bit for bit
(tests/test_oracle.py, goldens); oracle[keyed stream, portable trig] vs the device here — float z bit for
bit, RGB within +-1/255. The two oracle configurations differ by where the integers come from and by a
sin/cos evaluation that is within one ulp of libm's (both tested on the CPU)."""
import ctypes

import numpy as np
import pytest

from conftest import SAMPLED_TAGS

pytestmark = pytest.mark.gpu

RGB8_TOL = 1


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


def render_gpu(pkg, ctx, scene, W, H, spp, shard_count=1, stats=False, coop=None):
    ctx.upload(scene)
    shards, frames, allstats = [], [], None
    for r in range(shard_count):
        fr = pkg.frame_setup(scene.desc.camera, W, H, shard_rank=r, shard_count=shard_count, collect_stats=stats, samples=spp)
        if coop is not None:
            fr.coop_threshold = 1 if not coop else 1 << 30
        buf, st = ctx.render(fr, stats=stats)
        shards.append(buf)
        frames.append(fr)
        if stats:
            allstats = st if allstats is None else {k: allstats[k] + st[k] for k in st}
    return pkg.assemble(shards, frames, H), allstats


def check(gpu, cpu, orc, spp, what):
    zbad = int((gpu[..., 3].view(np.uint32) != cpu[..., 3].view(np.uint32)).sum())
    assert zbad == 0, "%s: %d pixels differ in float z" % (what, zbad)
    g8, _, gz8 = orc.postprocess(gpu)
    c8, _, cz8 = orc.postprocess(cpu)
    assert np.array_equal(gz8, cz8), what + ": z-image differs"
    d8 = np.abs(g8.astype(np.int32) - c8.astype(np.int32))
    assert d8.max() <= RGB8_TOL, "%s: 8-bit RGB differs by %d levels at %d pixels" % (what, d8.max(), (d8 > RGB8_TOL).sum())
    d = np.abs(gpu[..., :3].astype(np.float64) - cpu[..., :3].astype(np.float64))
    # a ray that grazes an edge may fall on the other side with a last-bit difference in its direction
    # (powf / expf feed the colours, not the rays; the sampled directions come from integer hashes and IEEE
    # operations only) — none is expected, and one flipped sample of spp would show as ~1/spp here
    assert d.max() < 2e-4, "%s: linear RGB differs by %.3g" % (what, d.max())
    return int((d8 > 0).sum())


@pytest.mark.parametrize("tag", SAMPLED_TAGS)
def test_sampled_scene_vs_oracle(pkg, orc, ctx, golden, tag):
    g = golden(tag)
    scene = g.scene(pkg)
    W, H, spp = g.width, g.height, g.meta["spp"]
    cpu, cst = orc.render_samples(scene, W, H, spp, stream=orc.STREAM_KEYED, trig=orc.TRIG_PORTABLE, threads=8)
    fast, _ = render_gpu(pkg, ctx, scene, W, H, spp)
    check(fast, cpu, orc, spp, tag + " fast")
    cnt, gst = render_gpu(pkg, ctx, scene, W, H, spp, stats=True)
    assert np.array_equal(cnt.view(np.uint32), fast.view(np.uint32)), "fast and counting variants differ"
    assert gst == cst, "counters differ"


@pytest.mark.parametrize("tag,spp", [("p10_s4_160x120", 9), ("p9_s3_160x120", 5), ("p11x86_s1_120x90", 3)])
def test_sampled_other_sample_counts_and_shards(pkg, orc, ctx, golden, tag, spp):
    """More samples than the golden has, and the image sharded over 3 contexts: the streams are keyed by the
    pixel of the whole image, so shards assemble to the single-GPU image bit for bit."""
    g = golden(tag)
    scene = g.scene(pkg)
    W, H = g.width, g.height
    cpu, _ = orc.render_samples(scene, W, H, spp, stream=orc.STREAM_KEYED, trig=orc.TRIG_PORTABLE, threads=8)
    one, _ = render_gpu(pkg, ctx, scene, W, H, spp)
    check(one, cpu, orc, spp, tag)
    three, _ = render_gpu(pkg, ctx, scene, W, H, spp, shard_count=3)
    assert np.array_equal(one.view(np.uint32), three.view(np.uint32))


@pytest.mark.parametrize("coop", [True, False])
def test_sampled_both_stage2_forms(pkg, orc, ctx, golden, coop):
    g = golden("teapot1_s2_160x90")
    scene = g.scene(pkg)
    W, H, spp = 320, 180, 2
    cpu, _ = orc.render_samples(scene, W, H, spp, stream=orc.STREAM_KEYED, trig=orc.TRIG_PORTABLE, threads=8)
    gpu, _ = render_gpu(pkg, ctx, scene, W, H, spp, coop=coop)
    check(gpu, cpu, orc, spp, "teapot1 coop=%s" % coop)


def test_deterministic_scene_sampled(pkg, orc, ctx, golden):
    """Recipe S on a scene without stochastic features: only the pixel offsets move; also every tail cut level."""
    g = golden("teapot2_240x135")
    scene = g.scene(pkg)
    W, H, spp = g.width, g.height, 4
    cpu, _ = orc.render_samples(scene, W, H, spp, stream=orc.STREAM_KEYED, trig=orc.TRIG_PORTABLE, threads=8)
    gpu, _ = render_gpu(pkg, ctx, scene, W, H, spp)
    check(gpu, cpu, orc, spp, "teapot2 sampled")
    fr = pkg.frame_setup(scene.desc.camera, W, H, samples=spp)
    for level in (1, 2, 3):
        assert pkg.hip.rtu_debug_tail_from(ctx._h, level) == 0
        again, _ = ctx.render(fr)
        assert np.array_equal(again.view(np.uint32), gpu.view(np.uint32)), "tail cut at level %d changes the image" % level


def test_recipe_w_refuses_stochastic_scenes(pkg, ctx, golden):
    """samples == 0 is the deterministic recipe: a scene with stochastic features is an error code there."""
    for tag in ("p9_s3_160x120", "p11gs_s2_160x90"):
        scene = golden(tag).scene(pkg)
        ctx.upload(scene)
        fr = pkg.frame_setup(scene.desc.camera, 32, 32)
        out = np.empty((32, 32, 4), np.float32)
        assert pkg.hip.rtu_render_frame(ctx._h, ctypes.byref(fr), out.ctypes.data, None) == pkg.RTU_ERR_STOCHASTIC
        fr.samples = 1
        assert pkg.hip.rtu_render_frame(ctx._h, ctypes.byref(fr), out.ctypes.data, None) == 0
        fr.samples = -1
        assert pkg.hip.rtu_render_frame(ctx._h, ctypes.byref(fr), out.ctypes.data, None) == pkg.RTU_ERR_ARG


def test_sampled_device_entry_and_timing(pkg, ctx, golden):
    """rtu_render_frame_device / rtu_time_render with samples >= 1 give the image of rtu_render_frame."""
    g = golden("p11gs_s2_160x90")
    scene = g.scene(pkg)
    ctx.upload(scene)
    W, H, spp = g.width, g.height, 3
    fr = pkg.frame_setup(scene.desc.camera, W, H, samples=spp)
    want, _ = ctx.render(fr)
    d = pkg.hip.rtu_device_alloc(ctx._h, W * H * 16)
    ctx.render_device(fr, d)
    ctx.frame_status()
    got = np.empty((H, W, 4), np.float32)
    assert pkg.hip.rtu_copy_to_host(ctx._h, got.ctypes.data, d, W * H * 16) == 0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ms = ctx.time_render(fr, d, None, 3)
    assert ms > 0
    pkg.hip.rtu_device_free(ctx._h, d)


def test_begin_render_sampled_dropin(pkg, orc, golden, tmp_path):
    """The BeginRender()-style entry with samples: the PNGs equal the oracle's post-processed image (z image
    exact, RGB +-1); without samples the same scene is refused with RTU_ERR_STOCHASTIC."""
    from conftest import read_png
    g = golden("p10_s4_160x120")
    scene = g.scene(pkg)
    spp = g.meta["spp"]
    cpu, _ = orc.render_samples(scene, g.width, g.height, spp, stream=orc.STREAM_KEYED, trig=orc.TRIG_PORTABLE, threads=8)
    c8, _, cz8 = orc.postprocess(cpu)
    img = pkg.Image(g.width, g.height)
    devs = (ctypes.c_int * 1)(0)
    rp, zp = str(tmp_path / "Result.png"), str(tmp_path / "ZBuffer.png")
    job = pkg.host.rtu_begin_render_sampled(scene._h, img._h, devs, 1, spp, rp.encode(), zp.encode())
    assert job
    assert pkg.host.rtu_render_wait(job) == 0, pkg.host.rtu_host_last_error()
    pkg.host.rtu_render_job_free(job)
    assert np.array_equal(read_png(zp), cz8)
    assert np.abs(read_png(rp).astype(np.int32) - c8.astype(np.int32)).max() <= RGB8_TOL
    job = pkg.host.rtu_begin_render(scene._h, img._h, devs, 1, None, None)
    assert pkg.host.rtu_render_wait(job) == pkg.RTU_ERR_STOCHASTIC
    pkg.host.rtu_render_job_free(job)


# ---- recipe P (config 5): the Monte-Carlo gather ---------------------------------------------------------
def render_paths_gpu(pkg, ctx, scene, W, H, spp, shard_count=1, coop=None):
    ctx.upload(scene)
    shards, frames = [], []
    for r in range(shard_count):
        fr = pkg.frame_setup(scene.desc.camera, W, H, shard_rank=r, shard_count=shard_count, samples=spp, gather_bounces=4)
        if coop is not None:
            fr.coop_threshold = 1 if not coop else 1 << 30
        buf, _ = ctx.render(fr)
        shards.append(buf)
        frames.append(fr)
    return pkg.assemble(shards, frames, H)


@pytest.mark.parametrize("tag,spp", [("p11_p2_120x68", 2), ("p13_p2_96x72", 2), ("p10_s4_160x120", 3), ("teapot1_s2_160x90", 5), ("p9_s3_160x120", 2)])
def test_paths_vs_oracle(pkg, orc, ctx, golden, tag, spp):
    """MonteCarlo() unrolled over the chain on the device against the recursive oracle, keyed streams: z bit-exact,
    RGB within the bar (the chain multiplies up to five Shade() trees: linear RGB to 1e-3 of the value)."""
    g = golden(tag)
    scene = g.scene(pkg)
    W, H = g.width, g.height
    cpu, _ = orc.render_paths(scene, W, H, spp, stream=orc.STREAM_KEYED, trig=orc.TRIG_PORTABLE, threads=8)
    gpu = render_paths_gpu(pkg, ctx, scene, W, H, spp)
    assert np.array_equal(gpu[..., 3].view(np.uint32), cpu[..., 3].view(np.uint32)), "z differs"
    g8, _, gz8 = orc.postprocess(gpu)
    c8, _, cz8 = orc.postprocess(cpu)
    assert np.array_equal(gz8, cz8)
    d8 = np.abs(g8.astype(np.int32) - c8.astype(np.int32))
    assert d8.max() <= RGB8_TOL, "8-bit RGB differs by %d levels at %d pixels" % (d8.max(), (d8 > RGB8_TOL).sum())
    d = np.abs(gpu[..., :3].astype(np.float64) - cpu[..., :3].astype(np.float64))
    assert (d / np.maximum(np.abs(cpu[..., :3]), 1e-2)).max() < 1e-3
    # the counting variant: same image, traversal and ray counters equal to the oracle's (gather rays are in no ray counter)
    _, cst = orc.render_paths(scene, W, H, spp, stream=orc.STREAM_KEYED, trig=orc.TRIG_PORTABLE, threads=8)
    frs = pkg.frame_setup(scene.desc.camera, W, H, samples=spp, gather_bounces=4, collect_stats=True)
    cnt, gst = ctx.render(frs, stats=True)
    assert np.array_equal(cnt.view(np.uint32), gpu.view(np.uint32)), "fast and counting variants differ"
    assert gst == cst, "counters differ"
    if tag == "p11_p2_120x68":
        three = render_paths_gpu(pkg, ctx, scene, W, H, spp, shard_count=3)
        assert np.array_equal(three.view(np.uint32), gpu.view(np.uint32)), "3 shards differ from one GPU"
        for coop in (True, False):
            again = render_paths_gpu(pkg, ctx, scene, W, H, spp, coop=coop)
            assert np.array_equal(again.view(np.uint32), gpu.view(np.uint32)), "stage-2 form changes the image"


def test_paths_argument_checks(pkg, ctx, golden):
    g = golden("p11_p2_120x68")
    scene = g.scene(pkg)
    ctx.upload(scene)
    out = np.empty((16, 16, 4), np.float32)
    fr = pkg.frame_setup(scene.desc.camera, 16, 16, samples=0, gather_bounces=4)
    assert pkg.hip.rtu_render_frame(ctx._h, ctypes.byref(fr), out.ctypes.data, None) == pkg.RTU_ERR_ARG
    fr = pkg.frame_setup(scene.desc.camera, 16, 16, samples=2, gather_bounces=3)
    assert pkg.hip.rtu_render_frame(ctx._h, ctypes.byref(fr), out.ctypes.data, None) == pkg.RTU_ERR_ARG
    fr = pkg.frame_setup(scene.desc.camera, 16, 16, samples=2, gather_bounces=4)
    assert pkg.hip.rtu_render_frame(ctx._h, ctypes.byref(fr), out.ctypes.data, None) == 0


def test_begin_render_paths_dropin(pkg, orc, golden, tmp_path):
    """HEAD's BeginRender() renders the path-traced mode: the drop-in with the gather, PNGs against the oracle."""
    from conftest import read_png
    g = golden("p11_p2_120x68")
    scene = g.scene(pkg)
    spp = 3
    cpu, _ = orc.render_paths(scene, g.width, g.height, spp, stream=orc.STREAM_KEYED, trig=orc.TRIG_PORTABLE, threads=8)
    c8, _, cz8 = orc.postprocess(cpu)
    img = pkg.Image(g.width, g.height)
    devs = (ctypes.c_int * 1)(0)
    rp, zp = str(tmp_path / "Result.png"), str(tmp_path / "ZBuffer.png")
    job = pkg.host.rtu_begin_render_paths(scene._h, img._h, devs, 1, spp, rp.encode(), zp.encode())
    assert job
    assert pkg.host.rtu_render_wait(job) == 0, pkg.host.rtu_host_last_error()
    pkg.host.rtu_render_job_free(job)
    assert np.array_equal(read_png(zp), cz8)
    png = read_png(rp)
    if png.ndim == 2:
        png = np.repeat(png[..., None], 3, axis=2)
    assert np.abs(png.astype(np.int32) - c8.astype(np.int32)).max() <= RGB8_TOL


def test_config5_at_its_own_size(pkg, orc, ctx, golden):
    """BASELINE config 5 at size: Project11 @1920x1080, recipe P (sample loop + 4-bounce Monte-Carlo gather), 4 samples per
    pixel — 8.3 M chains, 33 M shadow rays. Against the committed fixture (every 8th pixel of the oracle's frame on the keyed
    streams, tests/golden/make_oracle_goldens.py): z bit-exact, 8-bit RGB within one level, linear RGB to 1e-3; the sha256 of
    the whole float z; against the oracle run here on every pixel; the counting variant renders the same bits with the
    oracle's counters; properties that do not depend on any oracle: a second render is bit-identical (keyed streams), and the
    16-spp frame is closer to the 64-spp frame than the 4-spp one (the estimator converges)."""
    import hashlib, json, os
    g = golden("p11_1080")
    scene = g.scene(pkg)
    W, H = g.width, g.height
    fx = golden("p11_p4_1080")
    spp = fx.meta["spp"]
    gpu = render_paths_gpu(pkg, ctx, scene, W, H, spp)
    assert hashlib.sha256(np.ascontiguousarray(gpu[..., 3]).tobytes()).hexdigest() == fx.meta["sha256_z_f32"], "z differs from the fixture"
    assert np.array_equal(gpu[::8, ::8, 3].view(np.uint32), fx.npz["z_sub8"].view(np.uint32))
    ref = fx.npz["rgb_sub8"].astype(np.float64)
    d = np.abs(gpu[::8, ::8, :3].astype(np.float64) - ref)
    assert (d / np.maximum(np.abs(ref), 1e-2)).max() < 1e-3, "linear RGB differs from the fixture"
    again = render_paths_gpu(pkg, ctx, scene, W, H, spp)
    assert np.array_equal(again.view(np.uint32), gpu.view(np.uint32)), "two renders of the same frame differ"
    # every pixel against the oracle run on this machine
    cpu, cst = orc.render_paths(scene, W, H, spp, stream=orc.STREAM_KEYED, trig=orc.TRIG_PORTABLE, threads=min(64, os.cpu_count() or 8))
    assert np.array_equal(gpu[..., 3].view(np.uint32), cpu[..., 3].view(np.uint32)), "z differs"
    g8, _, gz8 = orc.postprocess(gpu)
    c8, _, cz8 = orc.postprocess(cpu)
    assert np.array_equal(gz8, cz8)
    d8 = np.abs(g8.astype(np.int32) - c8.astype(np.int32))
    assert d8.max() <= RGB8_TOL, "8-bit RGB differs by %d levels at %d pixels" % (d8.max(), (d8 > RGB8_TOL).sum())
    frs = pkg.frame_setup(scene.desc.camera, W, H, samples=spp, gather_bounces=4, collect_stats=True)
    cnt, gst = ctx.render(frs, stats=True)
    assert np.array_equal(cnt.view(np.uint32), gpu.view(np.uint32)), "fast and counting variants differ"
    assert gst == cst, "counters differ"
    assert (gst["primary_rays"], gst["shadow_rays"]) == (fx.meta["primary"], fx.meta["shadow"])
    # convergence
    g16 = render_paths_gpu(pkg, ctx, scene, W, H, 16)
    g64 = render_paths_gpu(pkg, ctx, scene, W, H, 64)
    e4 = np.abs(gpu[..., :3] - g64[..., :3]).mean()
    e16 = np.abs(g16[..., :3] - g64[..., :3]).mean()
    assert e16 < 0.75 * e4, (e4, e16)
    assert np.array_equal((g64[..., 3] < 1e29), (gpu[..., 3] < 1e29).__or__(g64[..., 3] < 1e29))  # a pixel hit at 4 spp is hit at 64
