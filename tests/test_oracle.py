"""CPU-only: the oracle (oracle/rtu_oracle.cpp) against the golden vectors generated
from the compiled reference (tests/golden/make_goldens.py). This is what pins the
oracle on a machine without /root/reference."""
import os

import numpy as np
import pytest

from conftest import REFERENCE, SMALL_TAGS, TEX_TAGS, read_png, sha256


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
