"""CPU-only tests of the host side of the boundary and of the C-ABI's shape:
the libraries load and export every symbol include/*.h declares (no compute calls
without a GPU), the loader reproduces the reference's scene values bit for bit, the
RenderImage mirror post-processes like the reference, shard arithmetic is consistent."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ALL_TAGS, LOCAL_SCENES, REFERENCE, REPO, SAMPLED_TAGS, SMALL_TAGS, TEX_TAGS, instantiate_scene, read_png

MAC_PREFIX = "/Users/Peter/GitRepos/RayTracer-Utah"
SCENES = {
    "p1_256": "Project1Example.xml", "p3s_800x600": "Project3Simple.xml", "p4_1080": "Project4.xml",
    "teapot2_1080": "Teapot/scene2.xml", "p11_1080": "Project11/scene.xml", "p4_240x135": "Project4.xml",
    "teapot2_240x135": "Teapot/scene2.xml", "p11_240x135": "Project11/scene.xml",
    "p1test_200x150": "Project1Test.xml", "p2_200x150": "Project2.xml", "p3box_200x150": "Project3Box.xml",
    "p5_200x150": "Project5/scene.xml", "p5low_200x150": "Project5/scene-low.xml",
    "p11simple_200x150": "Project11/scene_simple.xml", "p13_200x150": "Project13/scene.xml",
    "p7_200x150": "Project7/scene.xml",
    "p10_s4_160x120": "Project10/scene.xml", "p9_s3_160x120": "Project9/scene.xml",
    "p11gs_s2_160x90": "Project11/scene_glossy_soft.xml", "p11x86_s1_120x90": "Project11/scene_86.xml",
    "teapot1_s2_160x90": "Teapot/scene.xml", "p11g_s2_160x90": "Project11/scene_glossy.xml",
}


def declared_symbols(header):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtu_[a-z0-9_]+)\s*\(", text)))


def test_c_abi_exports_every_declared_symbol(pkg):
    hip_syms = declared_symbols("rtu_render.h")
    host_syms = declared_symbols("rtu_host.h")
    assert len(hip_syms) >= 15 and len(host_syms) >= 25
    for s in hip_syms:
        assert hasattr(pkg.hip, s), "librtu_hip.so does not export " + s
    for s in host_syms:
        assert hasattr(pkg.host, s), "librtu_host.so does not export " + s
    assert set(hip_syms) == set(pkg.HIP_SYMBOLS)
    assert set(host_syms) == set(pkg.HOST_SYMBOLS)


def test_struct_layouts_match_the_headers(pkg):
    assert ctypes.sizeof(pkg.RtuFrameDesc) == 112 and pkg.RtuFrameDesc.gather_bounces.offset == 108 and pkg.RtuFrameDesc.samples.offset == 28 and pkg.RtuFrameDesc.dof.offset == 104
    assert ctypes.sizeof(pkg.RtuStats) == 88
    assert ctypes.sizeof(pkg.RtuCamera) == 56
    assert ctypes.sizeof(pkg.RtuEnvColor) == 32


def test_no_gpu_means_error_code_not_crash(pkg):
    if pkg.hip.rtu_device_count() > 0:
        pytest.skip("a GPU is present")
    err = ctypes.c_int(0)
    assert not pkg.hip.rtu_create_context(0, ctypes.byref(err))
    assert err.value == pkg.RTU_ERR_NO_DEVICE
    assert b"GPU" in pkg.hip.rtu_error_string(err.value)


@pytest.mark.parametrize("tag", ALL_TAGS + TEX_TAGS)
def test_blob_round_trip(pkg, golden, tag):
    g = golden(tag)
    s = g.scene(pkg)
    blob = s.to_blob_bytes()
    s2 = pkg.Scene.from_blob_bytes(blob)
    assert s2.to_blob_bytes() == blob
    d = s.desc
    assert (d.camera.img_width, d.camera.img_height) == (g.width, g.height)
    assert d.n_nodes >= 2


def test_blob_rejects_garbage(pkg, golden):
    blob = golden("p1_256").scene(pkg).to_blob_bytes()
    for bad in (b"", b"nonsense" * 10, blob[:100], blob[:-64], b"X" + blob[1:]):
        with pytest.raises(pkg.RtuError):
            pkg.Scene.from_blob_bytes(bad)
    # a count larger than the blob can hold must not allocate / crash
    huge = bytearray(blob)
    huge[8:12] = (0x7FFFFFFF).to_bytes(4, "little")
    with pytest.raises(pkg.RtuError):
        pkg.Scene.from_blob_bytes(bytes(huge))


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="scene files only exist in the authoring container")
@pytest.mark.parametrize("tag", [t for t in ALL_TAGS + TEX_TAGS + SAMPLED_TAGS if t not in LOCAL_SCENES])
def test_loader_matches_reference_scene_values(pkg, golden, tag):
    """Own XML + OBJ reader + BVH build vs the blob dumped from the reference's in-memory
    scene graph after ITS LoadScene(): every float of every node/material/light/camera,
    every mesh array and every BVH node identical."""
    g = golden(tag)
    s = pkg.Scene.from_xml(os.path.join(REFERENCE, "SceneFiles", SCENES[tag]), MAC_PREFIX, REFERENCE)
    s.set_resolution(g.width, g.height)
    assert s.to_blob_bytes() == g.scene(pkg).to_blob_bytes()


@pytest.mark.parametrize("tag", sorted(LOCAL_SCENES))
def test_loader_matches_reference_on_obj_with_its_own_materials(pkg, golden, tag, tmp_path):
    """Row f3: a node without material= whose .obj names materials of a .mtl library (xmlload.cpp:199-243,
    cyTriMesh.h:452-547): faces regrouped material by material, a MultiMtl of one MtlBlinn per material bound to
    the node, map_Kd / map_Ks as the reference wires them — against the blob the COMPILED REFERENCE flattened
    from the same files (tests/golden/make_goldens.py), byte for byte. Runs everywhere: the scene is this
    repository's own (tests/scenes)."""
    g = golden(tag)
    s = pkg.Scene.from_xml(instantiate_scene(LOCAL_SCENES[tag], tmp_path / "scene"))
    s.set_resolution(g.width, g.height)
    assert s.to_blob_bytes() == g.scene(pkg).to_blob_bytes()
    assert s.desc.n_textures == 1 and s.desc.n_meshes == 1


def test_obj_faces_that_point_nowhere_are_an_error_not_a_crash(pkg, tmp_path):
    """The reference indexes its vertex arrays with whatever a face line says (cyTriMesh.h:421-431); here a literal 0,
    an index past the last vertex / texture vertex / normal, or a relative index before the first one fails the load
    before normals, bounds or the BVH are computed from it."""
    xml = """<xml><scene><object type="obj" name="{obj}" material="m"/><material type="blinn" name="m"/></scene>
             <camera><position x="0" y="-5" z="0"/><target x="0" y="0" z="0"/><up x="0" y="0" z="1"/></camera></xml>"""
    good = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\n"
    for face in ("f 0 1 2", "f 1 2 4", "f -4 1 2", "f 1/2 2/1 3/1", "f 1//2 2//1 3//1", "f 1 2 3\nf 1 2 99999999999"):
        (tmp_path / "bad.obj").write_text(good + face + "\n")
        (tmp_path / "s.xml").write_text(xml.format(obj=tmp_path / "bad.obj"))
        with pytest.raises(pkg.RtuError, match="does not exist"):
            pkg.Scene.from_xml(str(tmp_path / "s.xml"))
    (tmp_path / "bad.obj").write_text(good + "f 1/1/1 2/1/1 -1/-1/-1\n")
    s = pkg.Scene.from_xml(str(tmp_path / "s.xml"))
    assert s.desc.n_meshes == 1


def test_loader_error_paths(pkg, tmp_path):
    with pytest.raises(pkg.RtuError, match="Failed to load"):
        pkg.Scene.from_xml(str(tmp_path / "missing.xml"))
    for text, msg in [("<xml><scene/></xml>", "camera"), ("<xml><camera/></xml>", "scene"), ("<foo/>", "xml"),
                      ("<xml><scene><object></scene></xml>", "parse error")]:
        p = tmp_path / "s.xml"
        p.write_text(text)
        with pytest.raises(pkg.RtuError, match=msg):
            pkg.Scene.from_xml(str(p))


def test_loader_defaults_and_transform_order(pkg, tmp_path):
    """Attribute defaults of SURVEY Appendix E on a hand-written scene."""
    p = tmp_path / "s.xml"
    p.write_text("""<xml><scene>
      <!-- comment -->
      <object type="sphere" name="a" material="m"><scale value="2"/><translate x="1"/>
        <object type="plane" name="b"><rotate angle="90" z="1"/></object></object>
      <object name="missing-obj" type="obj"/>
      <material type="blinn" name="m"><specular value="0"/><refraction value="0.5" index="1.5"/></material>
      <material type="phong" name="ignored"/>
      <light type="direct" name="d"><intensity value="2"/></light>
      <light type="point" name="p"><position x="1" y="2" z="3"/></light>
      <background r="0.25"/>
    </scene><camera><position z="10"/><target/><fov value="50"/></camera></xml>""")
    s = pkg.Scene.from_xml(str(p))
    d = s.desc
    assert d.n_nodes == 4 and d.n_materials == 1 and d.n_lights == 2
    nodes = np.ctypeslib.as_array(ctypes.cast(d.nodes, ctypes.POINTER(ctypes.c_float)), (d.n_nodes, 32))
    inodes = nodes.view(np.int32)
    # node 1: scale 2 then translate (1,0,0): tm = 2I, pos = (1,0,0), itm = 0.5 I
    assert np.array_equal(nodes[1, :9], np.diag([2, 2, 2]).astype(np.float32).ravel())
    assert np.array_equal(nodes[1, 9:18], np.diag([.5, .5, .5]).astype(np.float32).ravel())
    assert np.array_equal(nodes[1, 18:21], np.float32([1, 0, 0]))
    assert list(inodes[:, 21]) == [-1, 0, 1, 0]          # parents (pre-order)
    assert list(inodes[:, 22]) == [0, 1, 2, 0]           # types: group, sphere, plane, failed obj -> none
    assert list(inodes[:, 24]) == [-1, 0, -1, -1]        # material ids
    assert list(inodes[:, 26]) == [4, 3, 3, 4]           # subtree_end
    mats = np.ctypeslib.as_array(ctypes.cast(d.materials, ctypes.POINTER(ctypes.c_float)), (1, 24))
    assert np.array_equal(mats[0, :3], np.float32([.5, .5, .5]))     # diffuse default (materials.h:22)
    assert np.array_equal(mats[0, 3:6], np.float32([0, 0, 0]))       # specular value=0 -> (1,1,1)*0
    assert np.array_equal(mats[0, 9:12], np.float32([.5, .5, .5]))   # refraction value=0.5
    assert mats[0, 18] == 20 and mats[0, 19] == 1.5                  # glossiness default, ior
    lights = np.ctypeslib.as_array(ctypes.cast(d.lights, ctypes.POINTER(ctypes.c_float)), (2, 8))
    assert np.array_equal(lights[0, 1:4], np.float32([2, 2, 2]))
    assert np.array_equal(lights[0, 4:7], np.float32([0, 0, 1]))     # DirectLight default direction
    assert np.array_equal(lights[1, 4:7], np.float32([1, 2, 3]))
    assert list(d.background.color) == [0.25, 1.0, 1.0] and d.background.has_map == 0
    assert list(d.environment.color) == [0.0, 0.0, 0.0]
    cam = d.camera
    assert list(cam.pos) == [0, 0, 10] and list(cam.dir) == [0, 0, -1] and list(cam.up) == [0, 1, 0]
    assert (cam.fov, cam.img_width, cam.img_height) == (50, 200, 150)


@pytest.mark.parametrize("tag", SMALL_TAGS + ["teapot2_1080"])
def test_frame_setup_matches_oracle_camera_frame(pkg, orc, golden, tag):
    """Row a3: CalculateImageOrigin / CalculateCurrentPoint hoisted to the host."""
    g = golden(tag)
    cam = g.scene(pkg).desc.camera
    for (W, H) in [(g.width, g.height), (1920, 1080), (61, 45)]:
        f = pkg.frame_setup(cam, W, H)
        mine = np.float32([list(f.cam_pos), list(f.origin), list(f.u), list(f.v)])
        ref = orc.camera_frame(cam, W, H)
        assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32))


def test_shard_arithmetic(pkg, golden):
    cam = golden("p1_256").scene(pkg).desc.camera
    for H in (1, 7, 8, 9, 135, 1080, 1081):
        for G in (1, 2, 3, 4, 8, 200):
            seen = []
            for r in range(G):
                f = pkg.frame_setup(cam, 16, H, shard_rank=r, shard_count=G)
                rows = pkg.shard_global_rows(f)
                assert len(rows) == pkg.shard_rows(f) <= pkg.hip.rtu_shard_max_rows(H, G)
                assert all((row // 8) % G == r for row in rows)
                seen.extend(rows.tolist())
            assert sorted(seen) == list(range(H)), (H, G)


@pytest.mark.parametrize("tag", SMALL_TAGS + TEX_TAGS)
def test_render_image_mirror_matches_reference(pkg, orc, golden, tag, tmp_path):
    """rtu_image_* (gamma, Color24, z-image, PNG) on the oracle's float output must give
    the reference's Result / ZBuffer pixels exactly; bands may arrive in any order."""
    g = golden(tag)
    out, _ = orc.render(g.scene(pkg), g.width, g.height, threads=4)
    img = pkg.Image(g.width, g.height)
    order = list(range(0, g.height, 8))
    for r0 in order[::-1]:
        img.fill(out[r0:r0 + 8], r0)
    assert pkg.host.rtu_image_is_done(img._h)
    assert pkg.host.rtu_image_num_rendered(img._h) == g.width * g.height
    img.compute_zimage()
    assert np.array_equal(img.pixels(), g.npz["result_u8"])
    assert np.array_equal(img.zimage(), g.npz["zbuffer_u8"])
    assert np.array_equal(img.zbuffer().view(np.uint32), out[..., 3].view(np.uint32))
    rp, zp = str(tmp_path / "Result.png"), str(tmp_path / "ZBuffer.png")
    img.save(rp, zp)
    assert np.array_equal(read_png(rp), g.npz["result_u8"])
    assert np.array_equal(read_png(zp), g.npz["zbuffer_u8"])


def test_color24_edge_cases(pkg):
    """int(r*255) clamped; NaN / negative / huge behave as the reference's x86 build."""
    img = pkg.Image(8, 1)
    vals = np.float32([0.0, 1.0, 0.5, -1.0, 1e9, np.nan, 4.0, 1e-9])
    px = np.zeros((1, 8, 4), np.float32)
    px[0, :, 0] = vals
    px[0, :, 3] = [1, 2, 1e30, 4, 5, 6, 7, 8]
    img.fill(px, 0)
    exp = []
    for v in vals:
        gam = np.float32(np.float64(v) ** (1 / 2.2)) if v >= 0 else np.float32(np.nan)
        x = gam * np.float32(255)
        i = -2**31 if not (np.isfinite(x) and -2.0**31 <= x < 2.0**31) else int(x)
        exp.append(min(max(i, 0), 255))
    assert list(img.pixels()[0, :, 0]) == exp
    img.compute_zimage()
    zi = img.zimage()[0]
    assert zi[2] == 0 and zi[7] == 0 and zi[0] == 255  # miss -> 0, farthest -> 0, nearest -> 255


def test_float_thresholds_equal_the_double_literals():
    """The device code compares against 0.001f / 0.00001f in binary32 where the reference promotes
    to double and compares with 0.001 / 0.00001 (rtu_intersect.h ge_001 ...): equivalent for every
    float because neither literal is a float. Check the neighbourhood and a random sample."""
    import numpy as np
    rng = np.random.default_rng(5)
    c, e = np.float32(0.001), np.float32(0.00001)
    assert float(c) > 0.001 and float(np.nextafter(c, np.float32(0))) < 0.001
    assert float(e) < 0.00001 and float(np.nextafter(e, np.float32(1))) > 0.00001
    xs = [c, e]
    for base in (c, e):
        lo = hi = base
        for _ in range(8):
            lo, hi = np.nextafter(lo, np.float32(-1)), np.nextafter(hi, np.float32(1))
            xs += [lo, hi]
    xs = np.concatenate([np.float32(xs), rng.standard_normal(100000).astype(np.float32) * np.float32(0.002),
                         np.float32([0, -0.0, np.inf, -np.inf, np.nan, 1e-30, -1e-30])])
    xd = xs.astype(np.float64)
    with np.errstate(invalid="ignore"):
        assert np.array_equal(xd >= 0.001, xs >= c)
        assert np.array_equal(xd > 0.001, xs >= c)
        assert np.array_equal(xd <= 0.001, xs < c)
        assert np.array_equal(xd > 0.00001, xs > e)


def _png_bytes(w, h, ctype, pixels, palette=None):
    """Minimal PNG writer for the decoder test: 8-bit, colour types 0/2/3/4/6, filter 0..4 cycling."""
    import struct, zlib
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    raw = bytearray()
    prev = bytes(w * ch)
    for y in range(h):
        row = bytes(pixels[y * w * ch:(y + 1) * w * ch])
        f = y % 5
        out = bytearray()
        for x in range(w * ch):
            a = row[x - ch] if x >= ch else 0
            b = prev[x]
            c = prev[x - ch] if x >= ch else 0
            if f == 0: pred = 0
            elif f == 1: pred = a
            elif f == 2: pred = b
            elif f == 3: pred = (a + b) // 2
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out.append((row[x] - pred) & 0xFF)
        raw += bytes([f]) + out
        prev = row
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
    if palette is not None:
        data += chunk(b"PLTE", bytes(palette))
    comp = zlib.compress(bytes(raw))
    data += chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:]) + chunk(b"IEND", b"")
    return data


def test_texture_files_decode(pkg, tmp_path):
    """TextureFile::Load through the host library's own PNG (grey, RGB, palette, grey+alpha, RGBA; all
    five filters; split IDAT) and binary PPM readers: the RtuTexture holds exactly the RGB the
    reference's lodepng::decode(..., LCT_RGB) / LoadPPM would hold."""
    import ctypes, random, struct
    rnd = random.Random(3)
    W, H = 7, 6
    cases = {}
    grey = [rnd.randrange(256) for _ in range(W * H)]
    cases["g.png"] = (_png_bytes(W, H, 0, grey), [v for g_ in grey for v in (g_, g_, g_)])
    rgb = [rnd.randrange(256) for _ in range(W * H * 3)]
    cases["c.png"] = (_png_bytes(W, H, 2, rgb), rgb)
    pal = [rnd.randrange(256) for _ in range(5 * 3)]
    idx = [rnd.randrange(5) for _ in range(W * H)]
    cases["p.png"] = (_png_bytes(W, H, 3, idx, pal), [pal[3 * i + k] for i in idx for k in range(3)])
    ga = [rnd.randrange(256) for _ in range(W * H * 2)]
    cases["ga.png"] = (_png_bytes(W, H, 4, ga), [ga[2 * i] for i in range(W * H) for _ in range(3)])
    rgba = [rnd.randrange(256) for _ in range(W * H * 4)]
    cases["a.png"] = (_png_bytes(W, H, 6, rgba), [rgba[4 * i + k] for i in range(W * H) for k in range(3)])
    cases["x.ppm"] = (b"P6\n# a comment\n%d %d\n255\n" % (W, H) + bytes(rgb), rgb)
    for name, (data, want) in cases.items():
        (tmp_path / name).write_bytes(data)
        xml = tmp_path / "t.xml"
        xml.write_text("""<xml><scene>
          <object type="plane" name="p" material="m"/>
          <material type="blinn" name="m"><diffuse texture="%s"/></material>
        </scene><camera><position x="0" y="0" z="5"/><target x="0" y="0" z="0"/><up x="0" y="1" z="0"/><fov value="40"/>
          <width value="16"/><height value="16"/></camera></xml>""" % (tmp_path / name))
        sc = pkg.Scene.from_xml(str(xml))  # keeps the desc alive
        d = sc.desc
        assert d.n_textures == 1, name
        hdr = ctypes.string_at(d.textures, 56)
        typ, w, h, _ = struct.unpack_from("<4i", hdr, 0)
        ptr = struct.unpack_from("<Q", hdr, 16)[0]
        assert (typ, w, h) == (0, W, H), name
        assert list(ctypes.string_at(ptr, W * H * 3)) == want, name
    # a file that cannot be decoded behaves like a missing one: TextureMap(NULL), the colour turns black
    (tmp_path / "bad.png").write_bytes(b"not a png")
    xml.write_text(xml.read_text().replace(str(tmp_path / name), str(tmp_path / "bad.png")))
    sc = pkg.Scene.from_xml(str(xml))
    d = sc.desc
    assert d.n_textures == 0 and not d.material_maps


def test_validate_rejects_bad_mesh_indices(pkg, golden):
    """rtu_validate_scene (what rtu_upload_scene runs first; pure host code): an index the kernels would follow
    out of its array is an error code on the CPU, never a fault on the GPU — texture-vertex indices included
    (they are read on every accepted triangle hit of a textured scene)."""
    import ctypes
    scene = golden("p7_200x150").scene(pkg)
    err = ctypes.create_string_buffer(256)
    assert pkg.hip.rtu_validate_scene(scene.desc_ptr, err, 256) == 0
    m = ctypes.cast(scene.desc.meshes, ctypes.POINTER(pkg.RtuMesh))[0]
    assert m.nvt > 0 and m.ft and m.vt
    u32 = ctypes.POINTER(ctypes.c_uint32)
    for field, limit, what in (("ft", m.nvt, b"texture-vertex"), ("f", m.nv, b"vertex index"), ("fn", m.nvn, b"normal index")):
        arr = ctypes.cast(getattr(m, field), u32)
        keep = arr[5]
        arr[5] = limit  # one past the last element
        assert pkg.hip.rtu_validate_scene(scene.desc_ptr, err, 256) == pkg.RTU_ERR_ARG
        assert what in err.value, err.value
        arr[5] = keep
    keep = m.vt
    m.vt = None  # texture faces without texture vertices
    assert pkg.hip.rtu_validate_scene(scene.desc_ptr, err, 256) == pkg.RTU_ERR_ARG
    m.vt = keep
    keep = m.nvt
    m.nvt = 0
    assert pkg.hip.rtu_validate_scene(scene.desc_ptr, err, 256) == pkg.RTU_ERR_ARG
    m.nvt = keep
    assert pkg.hip.rtu_validate_scene(scene.desc_ptr, err, 256) == 0
    assert pkg.hip.rtu_validate_scene(None, err, 256) == pkg.RTU_ERR_ARG
