"""CPU-only: the oracle (oracle/rtu_oracle.cpp) against the golden vectors generated
from the compiled reference (tests/golden/make_goldens.py). This is what pins the
oracle on a machine without /root/reference."""
import os

import numpy as np
import pytest

from conftest import PATH_TAGS, REFERENCE, SAMPLED_TAGS, SMALL_TAGS, TEX_TAGS, read_png, sha256


@pytest.mark.parametrize("tag", SMALL_TAGS + TEX_TAGS)
def test_oracle_bit_exact_vs_reference_golden(pkg, orc, golden, tag):
    g = golden(tag)
    scene = g.scene(pkg)
    out, st = orc.render(scene, g.width, g.height, threads=4)
    # full float buffers dumped from the reference's own Trace/Shade
    assert np.array_equal(out[..., 3].view(np.uint32), g.npz["z"].view(np.uint32)), "z differs"
    assert np.array_equal(out[..., :3].view(np.uint32), g.npz["rgb"].view(np.uint32)), "linear RGB differs"
    assert sha256(out[..., 3]) == g.meta["sha256_z_f32"]
    assert sha256(out[..., :3]) == g.meta["sha256_rgb_f32"]
    # ray counters measured on the reference by link-time wrapping of Trace / ShadowTrace
    assert st["primary_rays"] == g.meta["primary"]
    assert st["primary_hits"] == g.meta["primary_hits"]
    assert st["secondary_rays"] == g.meta["secondary"]
    assert st["shadow_rays"] == g.meta["shadow"]


def test_oracle_p3s_full_config(pkg, orc, golden):
    """BASELINE config 2 at its full size (800x600), still a sub-second CPU job."""
    g = golden("p3s_800x600")
    out, st = orc.render(g.scene(pkg), g.width, g.height, threads=8)
    assert np.array_equal(out[..., 3].view(np.uint32), g.npz["z"].view(np.uint32))
    assert np.array_equal(out[..., :3].view(np.uint32), g.npz["rgb"].view(np.uint32))
    assert (st["secondary_rays"], st["shadow_rays"]) == (g.meta["secondary"], g.meta["shadow"])


@pytest.mark.parametrize("tag", ["teapot2_1080", "p4_1080"])
def test_oracle_full_size_hashes(pkg, orc, golden, tag):
    """The two 1920x1080 headline configs: sha256 of the full float buffers + subsample."""
    g = golden(tag)
    out, st = orc.render(g.scene(pkg), g.width, g.height, threads=8)
    assert sha256(out[..., 3]) == g.meta["sha256_z_f32"]
    assert sha256(out[..., :3]) == g.meta["sha256_rgb_f32"]
    assert np.array_equal(out[::8, ::8, 3].view(np.uint32), g.npz["z_sub8"].view(np.uint32))
    assert (st["primary_hits"], st["secondary_rays"], st["shadow_rays"]) == (
        g.meta["primary_hits"], g.meta["secondary"], g.meta["shadow"])


@pytest.mark.parametrize("tag", SMALL_TAGS + TEX_TAGS)
def test_oracle_postprocess_matches_reference_pngs(pkg, orc, golden, tag):
    """gamma + Color24 + ComputeZBufferImage against the PNGs the reference wrote."""
    g = golden(tag)
    out, _ = orc.render(g.scene(pkg), g.width, g.height, threads=4)
    rgb8, z, zimg = orc.postprocess(out)
    assert np.array_equal(rgb8, g.npz["result_u8"])
    assert np.array_equal(zimg, g.npz["zbuffer_u8"])
    png = read_png(os.path.join(g.dir, "Result.png"))
    if png.ndim == 2:  # lodepng auto-converts grey RGB images (SURVEY §2 "PNG codec")
        png = np.repeat(png[..., None], 3, axis=2)
    assert np.array_equal(png, rgb8)
    assert np.array_equal(read_png(os.path.join(g.dir, "ZBuffer.png")), zimg)


def test_oracle_threads_and_row_ranges_agree(pkg, orc, golden):
    g = golden("teapot2_240x135")
    scene = g.scene(pkg)
    a, sa = orc.render(scene, g.width, g.height, threads=1)
    b, sb = orc.render(scene, g.width, g.height, threads=5)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa == sb
    part, _ = orc.render(scene, g.width, g.height, threads=2, row0=40, nrows=17)
    assert np.array_equal(part.view(np.uint32), a[40:57].view(np.uint32))


def test_oracle_rejects_stochastic_scenes(pkg, orc, golden):
    """Soft shadows / glossy bounces / dof make the reference non-deterministic."""
    import ctypes
    g = golden("p3s_800x600")
    scene = g.scene(pkg)
    blob = bytearray(scene.to_blob_bytes())
    desc = scene.desc
    # find the point light in the blob and give it a size (RtuLight.size is the last float)
    lights = (ctypes.c_float * (8 * desc.n_lights)).from_address(desc.lights)
    types = (ctypes.c_int32 * (8 * desc.n_lights)).from_address(desc.lights)
    idx = [i for i in range(desc.n_lights) if types[8 * i] == 2][0]
    lights[8 * idx + 7] = 0.5
    with pytest.raises(Exception) as e:
        orc.render(scene, 16, 16)
    assert e.value.code == orc.ERR_STOCHASTIC
    lights[8 * idx + 7] = 0.0
    orc.render(scene, 16, 16)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree only exists in the authoring container")
def test_oracle_vs_live_reference_build(pkg, orc, tmp_path):
    """Where the reference is present: run oracle/_ref/ref_render (the reference's own
    functions) on a scene/resolution that is NOT among the goldens and compare all bits."""
    import subprocess
    from conftest import REPO
    exe = os.path.join(REPO, "oracle", "_ref", "ref_render")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref not built")
    xml = tmp_path / "scene.xml"
    src = open(os.path.join(REFERENCE, "SceneFiles", "Project5", "scene.xml")).read()
    xml.write_text(src.replace("/Users/Peter/GitRepos/RayTracer-Utah", REFERENCE))
    W, H = 200, 150
    subprocess.check_call([exe, str(xml), str(W), str(H), str(tmp_path), "4"], stdout=subprocess.DEVNULL)
    scene = pkg.Scene.from_blob_file(str(tmp_path / "scene.rtus"))
    out, _ = orc.render(scene, W, H, threads=4)
    z = np.fromfile(tmp_path / "z.f32", np.float32).reshape(H, W)
    rgb = np.fromfile(tmp_path / "rgb.f32", np.float32).reshape(H, W, 3)
    assert np.array_equal(out[..., 3].view(np.uint32), z.view(np.uint32))
    assert np.array_equal(out[..., :3].view(np.uint32), rgb.view(np.uint32))


# ---- recipe S: stochastic effects -------------------------------------------------
@pytest.mark.parametrize("tag", SAMPLED_TAGS)
def test_oracle_recipe_s_bit_exact_vs_reference_golden(pkg, orc, golden, tag):
    """The goldens come from the reference built with rand() wrapped to the sequential sample stream
    (oracle/ref_harness): the restatement of the sample loop, soft shadows, glossy bounces and depth of
    field must reproduce every bit of z and linear RGB, and fire the same rays."""
    g = golden(tag)
    assert g.meta["recipe"] == "S" and g.meta["stream"] == "sequential"
    out, st = orc.render_samples(g.scene(pkg), g.width, g.height, g.meta["spp"], stream=orc.STREAM_SEQUENTIAL,
                                 trig=orc.TRIG_LIBM, threads=4)
    assert np.array_equal(out[..., 3].view(np.uint32), g.npz["z"].view(np.uint32)), "z differs"
    assert np.array_equal(out[..., :3].view(np.uint32), g.npz["rgb"].view(np.uint32)), "linear RGB differs"
    assert (st["primary_rays"], st["primary_hits"], st["secondary_rays"], st["shadow_rays"]) == (
        g.meta["primary"], g.meta["primary_hits"], g.meta["secondary"], g.meta["shadow"])
    rgb8, _, zimg = orc.postprocess(out)
    assert np.array_equal(rgb8, g.npz["result_u8"]) and np.array_equal(zimg, g.npz["zbuffer_u8"])


def test_portable_sincos_within_one_ulp_of_libm(orc):
    """The device cannot call libm's sinf/cosf; oracle and device share a binary64 evaluation instead.
    It must stay within 1 ulp of the float results the reference's calls give (and equal them almost always)."""
    rng = np.random.default_rng(5)
    t = np.concatenate([rng.random(400000, dtype=np.float32) * np.float32(2 * np.pi),
                        np.array([0.0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi, 1e-30, 6.2831855], np.float32)])
    s, c = orc.portable_sincos(t)
    for got, want in ((s, np.sin(t.astype(np.float64))), (c, np.cos(t.astype(np.float64)))):
        ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
        assert np.all(np.abs(got.astype(np.float64) - want) <= 0.5000001 * np.maximum(ulp, 1e-45)), "not correctly rounded"
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.sinf.restype = libm.cosf.restype = ctypes.c_float
    libm.sinf.argtypes = libm.cosf.argtypes = [ctypes.c_float]
    sub = t[:20000]
    ls = np.array([libm.sinf(float(x)) for x in sub], np.float32)
    lc = np.array([libm.cosf(float(x)) for x in sub], np.float32)
    for got, want in ((s[:20000], ls), (c[:20000], lc)):
        d = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
        assert d.max() <= 1 and (d != 0).mean() < 0.02


def test_recipe_s_portable_trig_changes_few_pixels(pkg, orc, golden):
    """Depth of field feeds sin/cos into the primary ray: with the portable evaluation a handful of pixels
    move by an ulp-sized lens offset; everything else is still the reference's image bit for bit."""
    g = golden("p9_s3_160x120")
    out, _ = orc.render_samples(g.scene(pkg), g.width, g.height, g.meta["spp"], stream=orc.STREAM_SEQUENTIAL,
                                trig=orc.TRIG_PORTABLE, threads=4)
    same = out[..., 3].view(np.uint32) == g.npz["z"].view(np.uint32)
    assert same.mean() > 0.99
    assert np.abs(out[..., :3] - g.npz["rgb"]).max() < 1e-3


@pytest.mark.parametrize("tag", ["p10_s4_160x120", "p11gs_s2_160x90"])
def test_recipe_s_keyed_stream_is_the_same_estimator(pkg, orc, golden, tag):
    """The keyed stream (what the device follows) draws other numbers than the sequential one, from the
    same distributions: z is identical without depth of field, and the image means agree to within the noise."""
    g = golden(tag)
    scene = g.scene(pkg)
    spp = 16
    a, sa = orc.render_samples(scene, g.width, g.height, spp, stream=orc.STREAM_SEQUENTIAL, threads=8)
    b, sb = orc.render_samples(scene, g.width, g.height, spp, stream=orc.STREAM_KEYED, threads=8)
    assert np.array_equal(a[..., 3].view(np.uint32), b[..., 3].view(np.uint32))
    assert not np.array_equal(a[..., :3], b[..., :3])
    ma, mb = a[..., :3].mean(), b[..., :3].mean()
    assert abs(ma - mb) < 0.01 * ma
    assert abs(sa["shadow_rays"] - sb["shadow_rays"]) < 0.01 * sa["shadow_rays"]
    # threads and row ranges do not change a keyed image
    c, _ = orc.render_samples(scene, g.width, g.height, spp, stream=orc.STREAM_KEYED, threads=3, row0=17, nrows=9)
    assert np.array_equal(c.view(np.uint32), b[17:26].view(np.uint32))


def test_sample_stream_known_answers(orc):
    """The integer hash of include/rtu_render.h, pinned by value so that oracle, device and documentation
    cannot drift apart silently."""
    def mix32(x):
        x &= 0xFFFFFFFF
        x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF
        x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF
        x ^= x >> 16
        return x
    for key, idx in ((0, 0), (1, 2), (0xdeadbeef, 0x30005), (123456789, 17)):
        want = mix32(key ^ mix32((idx * 0x9e3779b9 + 0x85ebca6b) & 0xFFFFFFFF)) >> 1
        assert orc.lib.rtu_oracle_rand31(key, idx) == want
    for pix, smp in ((0, 0), (1919 + 1920 * 1079, 63)):
        want = mix32(mix32((pix + 0x68bc21eb) & 0xFFFFFFFF) ^ ((smp * 0x9e3779b9 + 1) & 0xFFFFFFFF))
        assert orc.lib.rtu_oracle_sample_key(pix, smp) == want
    for key, slot in ((5, 0), (0xffffffff, 2)):
        assert orc.lib.rtu_oracle_child_key(key, slot) == mix32((key + (slot + 1) * 0x632be5ab) & 0xFFFFFFFF)


# ---- recipe P: the Monte-Carlo gather of config 5 ---------------------------------------------------
@pytest.mark.parametrize("tag", PATH_TAGS)
def test_oracle_recipe_p_bit_exact_vs_reference_golden(pkg, orc, golden, tag):
    """MonteCarlo() (4 bounces, cosine-weighted hemisphere, an AmbientLight per level) as HEAD's Render()
    calls it, against the reference built with rand() wrapped to the sequential stream: every bit of z and RGB."""
    g = golden(tag)
    assert g.meta["recipe"] == "P"
    out, st = orc.render_paths(g.scene(pkg), g.width, g.height, g.meta["spp"], stream=orc.STREAM_SEQUENTIAL,
                               trig=orc.TRIG_LIBM, threads=4)
    assert np.array_equal(out[..., 3].view(np.uint32), g.npz["z"].view(np.uint32)), "z differs"
    assert np.array_equal(out[..., :3].view(np.uint32), g.npz["rgb"].view(np.uint32)), "linear RGB differs"
    assert (st["primary_hits"], st["secondary_rays"], st["shadow_rays"]) == (g.meta["primary_hits"], g.meta["secondary"], g.meta["shadow"])
    # with the portable acos / sincos a few percent of the pixels move in their last bits, nothing more
    port, _ = orc.render_paths(g.scene(pkg), g.width, g.height, g.meta["spp"], stream=orc.STREAM_SEQUENTIAL,
                               trig=orc.TRIG_PORTABLE, threads=4)
    d = np.abs(port[..., :3] - g.npz["rgb"])
    assert (d > 0).any(-1).mean() < 0.15 and np.median(d) == 0


def test_portable_acos_within_one_ulp_of_libm(orc):
    rng = np.random.default_rng(11)
    x = np.concatenate([1 - 2 * rng.random(300000, dtype=np.float32), np.float32([-1, 1, 0, 0.5, -0.5, 1e-20, 0.99999994, -0.99999994])])
    got = orc.portable_acos(x)
    want = np.arccos(x.astype(np.float64))
    ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    assert np.all(np.abs(got.astype(np.float64) - want) <= 0.5000001 * ulp + 1e-300), "not correctly rounded"
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.acosf.restype = ctypes.c_float
    libm.acosf.argtypes = [ctypes.c_float]
    la = np.array([libm.acosf(float(v)) for v in x[:20000]], np.float32)
    dd = np.abs(got[:20000].view(np.int32).astype(np.int64) - la.view(np.int32).astype(np.int64))
    assert dd.max() <= 1 and (dd != 0).mean() < 0.12  # glibc acosf is within 1 ulp, not correctly rounded


def test_recipe_p_keyed_stream_is_the_same_estimator(pkg, orc, golden):
    """Recipe P on the keyed streams (what the device follows) against the sequential ones (what the reference with
    wrapped rand() follows): identical z, image means within the Monte-Carlo noise of 24 samples per pixel."""
    g = golden("p11_p2_120x68")
    scene = g.scene(pkg)
    a, _ = orc.render_paths(scene, g.width, g.height, 24, stream=orc.STREAM_SEQUENTIAL, trig=orc.TRIG_LIBM, threads=8)
    b, _ = orc.render_paths(scene, g.width, g.height, 24, stream=orc.STREAM_KEYED, trig=orc.TRIG_PORTABLE, threads=8)
    assert np.array_equal(a[..., 3].view(np.uint32), b[..., 3].view(np.uint32))
    for c in range(3):
        ma, mb = a[..., c].mean(), b[..., c].mean()
        assert abs(ma - mb) < 0.02 * ma, (c, ma, mb)
    # 8x8 block means agree too (no structured bias): relative L1 difference of the block images
    def blocks(x):
        h, w = x.shape[0] // 8 * 8, x.shape[1] // 8 * 8
        return x[:h, :w, :3].reshape(h // 8, 8, w // 8, 8, 3).mean((1, 3))
    ba, bb = blocks(a), blocks(b)
    assert np.abs(ba - bb).sum() / np.abs(ba).sum() < 0.08


def test_per_pixel_schedule_renders_the_same_image(pkg, orc, golden):
    """bench.py times the CPU port under both work distributions: the reference's PixelIterator
    (one atomic fetch per pixel) and chunks of rows. Scheduling must not change a bit."""
    g = golden("teapot2_240x135")
    scene = g.scene(pkg)
    a, sa = orc.render(scene, g.width, g.height, threads=4)
    for threads in (1, 5):
        b, sb = orc.render_scheduled(scene, g.width, g.height, threads, True)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa == sb
