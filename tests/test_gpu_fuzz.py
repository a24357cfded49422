"""Randomised scenes (fixed seeds): spheres, planes and two procedural meshes under random nested
transforms, random Blinn materials (mirror, glass, checker / file textures), random lights and camera.
Each is rendered by the GPU (fast and counting variants, cooperative and wide stage 2) and by the CPU
oracle: z bit-exact, RGB within the bar, all counters equal. Catches what the fixed scenes miss —
grazing rays, rays starting inside spheres, nested non-uniform scales, instanced meshes."""
import math
import random

import numpy as np
import pytest

from test_gpu_parity import _write_uv_mesh, check_against

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


def _xf(rnd, scale=True):
    s = ""
    if scale:
        if rnd.random() < 0.5:
            s += '<scale value="%r"/>' % rnd.uniform(0.5, 3.0)
        else:
            s += '<scale x="%r" y="%r" z="%r"/>' % (rnd.uniform(0.4, 3), rnd.uniform(0.4, 3), rnd.uniform(0.4, 3))
    if rnd.random() < 0.7:
        ax = [rnd.uniform(-1, 1) for _ in range(3)]
        s += '<rotate angle="%r" x="%r" y="%r" z="%r"/>' % (rnd.uniform(-180, 180), ax[0], ax[1], ax[2] + 0.1)
    s += '<translate x="%r" y="%r" z="%r"/>' % (rnd.uniform(-6, 6), rnd.uniform(-6, 6), rnd.uniform(-3, 4))
    return s


def _scene_xml(rnd, d, textured):
    mats = []
    for i in range(5):
        kind = rnd.choice(["diffuse", "mirror", "glass", "mixed"])
        diff = 'r="%r" g="%r" b="%r"' % (rnd.random(), rnd.random(), rnd.random())
        tex = ""
        if textured and rnd.random() < 0.6:
            if rnd.random() < 0.5:
                tex = ' texture="checkerboard"><color1 r="%r" g="0.1" b="0.2"/><color2 r="0.9" g="%r" b="0.8"/><scale value="%r"/>' % (
                    rnd.random(), rnd.random(), rnd.uniform(0.05, 0.5))
            else:
                tex = ' texture="%s/noise.ppm"><scale x="%r" y="%r"/><rotate angle="%r" z="1"/>' % (d, rnd.uniform(0.2, 2), rnd.uniform(0.2, 2), rnd.uniform(0, 90))
        m = '<material type="blinn" name="m%d"><diffuse %s%s</diffuse><specular value="%r"/><glossiness value="%r"/>' % (
            i, diff, tex if tex else ">", rnd.uniform(0, 0.9), rnd.uniform(5, 120))
        if kind in ("mirror", "mixed"):
            m += '<reflection value="%r"/>' % rnd.uniform(0.2, 0.8)
        if kind in ("glass", "mixed"):
            m += '<refraction index="%r" value="%r"/><absorption r="%r" g="%r" b="%r"/>' % (
                rnd.uniform(1.1, 1.8), rnd.uniform(0.3, 0.95), rnd.uniform(0, 0.2), rnd.uniform(0, 0.2), rnd.uniform(0, 0.2))
        mats.append(m + "</material>")

    def obj(depth):
        t = rnd.choice(["sphere", "sphere", "plane", "torus", "blob", "group"] if depth < 3 else ["sphere", "plane", "torus"])
        mat = "m%d" % rnd.randrange(5)
        if t == "group":
            return '<object name="g">%s%s</object>' % (_xf(rnd), "".join(obj(depth + 1) for _ in range(rnd.randrange(1, 4))))
        if t in ("torus", "blob"):
            return '<object type="obj" name="%s/%s.obj" material="%s">%s</object>' % (d, t, mat, _xf(rnd))
        return '<object type="%s" name="o" material="%s">%s</object>' % (t, mat, _xf(rnd))

    objs = "".join(obj(0) for _ in range(rnd.randrange(3, 8)))
    objs += '<object type="plane" name="floor" material="m0"><scale value="40"/><translate z="-5"/></object>'
    lights = '<light type="ambient" name="a"><intensity value="%r"/></light>' % rnd.uniform(0.05, 0.3)
    for i in range(rnd.randrange(1, 4)):
        if rnd.random() < 0.5:
            lights += '<light type="direct" name="d%d"><intensity value="%r"/><direction x="%r" y="%r" z="-1"/></light>' % (
                i, rnd.uniform(0.3, 0.8), rnd.uniform(-1, 1), rnd.uniform(-1, 1))
        else:
            lights += '<light type="point" name="p%d"><intensity value="%r"/><position x="%r" y="%r" z="%r"/></light>' % (
                i, rnd.uniform(0.3, 0.8), rnd.uniform(-10, 10), rnd.uniform(-10, 10), rnd.uniform(5, 15))
    env = ""
    if textured:
        env = ('<background r="1" g="1" b="1" texture="%s/noise.ppm"><scale value="0.5"/></background>'
               '<environment value="0.7" texture="checkerboard"><color1 r="0.2" g="0.3" b="0.4"/><color2 r="1" g="0.9" b="0.8"/><scale value="0.1"/></environment>') % d
    a = rnd.uniform(0, 2 * math.pi)
    cam = ('<camera><position x="%r" y="%r" z="%r"/><target x="%r" y="%r" z="0"/><up x="0" y="0" z="1"/><fov value="%r"/>'
           '<width value="96"/><height value="64"/></camera>') % (16 * math.cos(a), 16 * math.sin(a), rnd.uniform(2, 10), rnd.uniform(-1, 1), rnd.uniform(-1, 1), rnd.uniform(30, 70))
    return "<xml><scene>%s%s%s%s</scene>%s</xml>" % (env, objs, "".join(mats), lights, cam)


@pytest.fixture(scope="module")
def assets(tmp_path_factory):
    d = tmp_path_factory.mktemp("fuzz")
    def torus(u, v):
        a, b = 2 * math.pi * u, 2 * math.pi * v
        return ((2 + 0.7 * math.cos(b)) * math.cos(a), (2 + 0.7 * math.cos(b)) * math.sin(a), 0.7 * math.sin(b))
    def blob(u, v):
        a, b = 2 * math.pi * u, math.pi * (v - 0.5)
        r = 1.5 + 0.3 * math.sin(5 * a) * math.cos(3 * b)
        return (r * math.cos(b) * math.cos(a), r * math.cos(b) * math.sin(a), r * math.sin(b))
    _write_uv_mesh(d / "torus.obj", 24, 10, torus)
    _write_uv_mesh(d / "blob.obj", 20, 10, blob)
    rnd = random.Random(1)
    (d / "noise.ppm").write_bytes(b"P6\n16 16\n255\n" + bytes(rnd.randrange(256) for _ in range(16 * 16 * 3)))
    return d


@pytest.mark.parametrize("seed", range(24))
def test_random_scene(pkg, orc, ctx, assets, seed):
    rnd = random.Random(1000 + seed)
    textured = seed % 3 == 2
    xml = assets / ("s%d.xml" % seed)
    xml.write_text(_scene_xml(rnd, assets, textured))
    scene = pkg.Scene.from_xml(str(xml))
    W, H = 96, 64
    cpu, cst = orc.render(scene, W, H, threads=4)
    ctx.upload(scene)
    frs = pkg.frame_setup(scene.desc.camera, W, H, collect_stats=True)
    cnt, gst = ctx.render(frs, stats=True)
    # deep glass / mirror recursions multiply many powf / expf terms (device library vs glibc): the bar
    # is +-1/255 per channel; the linear-RGB bound is a tighter self-check, loosened here from 2e-5
    check_against(cnt, cpu, orc, rel_tol=1e-4)
    assert gst == cst
    for thr in (10 ** 9, 1):  # stage 2 cooperative / one lane per ray
        fr = pkg.frame_setup(scene.desc.camera, W, H)
        fr.coop_threshold = thr
        fast, _ = ctx.render(fr)
        assert np.array_equal(fast.view(np.uint32), cnt.view(np.uint32)), "fast (threshold %d) and counting variants differ" % thr


def _make_stochastic(xml, rnd):
    """The random scene with the reference's stochastic attributes added: glossy reflection / refraction,
    point lights with a size, a lens (focal distance at the look-at distance)."""
    import re
    xml = re.sub(r'<reflection value="([^"]+)"/>', lambda m: '<reflection value="%s" glossiness="%r"/>' % (m.group(1), rnd.choice([0.0, 0.03, 0.1, 0.3])), xml)
    xml = re.sub(r'(<refraction index="[^"]+" value="[^"]+")/>', lambda m: '%s glossiness="%r"/>' % (m.group(1), rnd.choice([0.0, 0.02, 0.08])), xml)
    xml = re.sub(r'(<light type="point" name="p\d+">)', lambda m: m.group(1) + '<size value="%r"/>' % rnd.choice([0.0, 0.5, 2.0, 5.0]), xml)
    if rnd.random() < 0.6:
        xml = xml.replace('<fov value=', '<focaldist value="%r"/><dof value="%r"/><fov value=' % (rnd.uniform(10, 20), rnd.uniform(0.05, 0.6)))
    return xml


@pytest.mark.parametrize("seed", range(12))
def test_random_scene_sampled(pkg, orc, ctx, assets, seed):
    """Recipe S on random scenes: glossy bounces, soft shadows, depth of field, 1-5 samples per pixel, against
    the oracle on the keyed sample streams — z bit-exact, RGB within the bar, counters equal, both stage-2 forms."""
    rnd = random.Random(7000 + seed)
    textured = seed % 3 == 2
    xml = assets / ("ss%d.xml" % seed)
    xml.write_text(_make_stochastic(_scene_xml(rnd, assets, textured), rnd))
    scene = pkg.Scene.from_xml(str(xml))
    W, H, spp = 96, 64, 1 + seed % 5
    cpu, cst = orc.render_samples(scene, W, H, spp, stream=orc.STREAM_KEYED, trig=orc.TRIG_PORTABLE, threads=4)
    ctx.upload(scene)
    frs = pkg.frame_setup(scene.desc.camera, W, H, collect_stats=True, samples=spp)
    cnt, gst = ctx.render(frs, stats=True)
    check_against(cnt, cpu, orc, rel_tol=3e-4)
    assert gst == cst
    for thr in (10 ** 9, 1):
        fr = pkg.frame_setup(scene.desc.camera, W, H, samples=spp)
        fr.coop_threshold = thr
        fast, _ = ctx.render(fr)
        assert np.array_equal(fast.view(np.uint32), cnt.view(np.uint32)), "fast (threshold %d) and counting variants differ" % thr
