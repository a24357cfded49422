"""The occluder lists of hard shadow rays (rtu_device.h DevLightMask, rtu_capi.hip compute_light_list), checked on the CPU ray by ray.

A shadow ray's direction from its light is fixed by its origin, so the device looks the origin's cell up and tests only the triangles
listed there — an empty cell skips the mesh, and entries whose zmin lies beyond the origin's depth are never reached. That is only
right if the list is CONSERVATIVE: every triangle of the mesh that a shadow ray from origin p towards the light can hit must be in
p's cell with zmin <= depth(p); and an origin outside the grid (or behind a point light's pinhole plane) must not be able to hit
anything. rtu_debug_light_list builds the very list rtu_upload_scene uploads (pure host code); this test fires thousands of shadow
rays — from the mesh's own surface (self-shadowing, where the list is longest), from a plane under it, from random points around
it — brute-forces every world-space triangle in binary64 and compares. Known answers, no GPU."""
import ctypes
import math
import os
import random

import numpy as np
import pytest

from conftest import LOCAL_SCENES  # noqa: F401  (conftest puts the repository on sys.path)


class RtuNode(ctypes.Structure):
    _fields_ = [("tm", ctypes.c_float * 9), ("itm", ctypes.c_float * 9), ("pos", ctypes.c_float * 3), ("parent", ctypes.c_int32), ("obj_type", ctypes.c_int32),
                ("mesh_id", ctypes.c_int32), ("material_id", ctypes.c_int32), ("depth", ctypes.c_int32), ("subtree_end", ctypes.c_int32), ("reserved", ctypes.c_int32 * 5)]


class RtuLight(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int32), ("intensity", ctypes.c_float * 3), ("vec", ctypes.c_float * 3), ("size", ctypes.c_float)]


def world_triangles(pkg, scene, node):
    d = scene.desc
    nodes = ctypes.cast(d.nodes, ctypes.POINTER(RtuNode))
    meshes = ctypes.cast(d.meshes, ctypes.POINTER(pkg.RtuMesh))
    m = meshes[nodes[node].mesh_id]
    v = np.ctypeslib.as_array(ctypes.cast(m.v, ctypes.POINTER(ctypes.c_float)), (m.nv * 3,)).reshape(-1, 3).astype(np.float64)
    f = np.ctypeslib.as_array(ctypes.cast(m.f, ctypes.POINTER(ctypes.c_uint32)), (m.nf * 3,)).reshape(-1, 3)
    p, j = v.copy(), node
    while j >= 0:
        t = nodes[j]
        tm = np.array(list(t.tm), np.float64).reshape(3, 3)  # column-major: p' = p.x * col0 + p.y * col1 + p.z * col2 + pos
        p = p[:, 0:1] * tm[0] + p[:, 1:2] * tm[1] + p[:, 2:3] * tm[2] + np.array(list(t.pos), np.float64)
        j = t.parent
    return p[f]  # [nf, 3, 3]


def hits(tri, o, dirs, tmax):
    """Binary64 Moeller-Trumbore of ONE ray against all triangles: boolean per triangle (0 < t < tmax, inside with a hair of tolerance)."""
    e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    pv = np.cross(dirs, e2)
    det = np.einsum("ij,ij->i", e1, pv)
    ok = np.abs(det) > 1e-300
    inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
    tv = o - tri[:, 0]
    u = np.einsum("ij,ij->i", tv, pv) * inv
    qv = np.cross(tv, e1)
    w = (qv @ dirs) * inv
    t = np.einsum("ij,ij->i", e2, qv) * inv
    return ok & (u > -1e-9) & (w > -1e-9) & (u + w < 1 + 1e-9) & (t > 1e-7) & (t < tmax)


def check_list(pkg, scene, light_slot, cover_slot, rnd, n_rays):
    lst = pkg.light_list(scene, light_slot, cover_slot)
    if lst is None:
        return None
    tri = world_triangles(pkg, scene, lst["node"])
    lights = ctypes.cast(scene.desc.lights, ctypes.POINTER(RtuLight))
    l = lights[lst["light"]]
    lvec = np.array(list(l.vec), np.float64)
    point = lst["point"]
    assert point == (l.type == 2)
    G, off = lst["G"], lst["cell_off"]
    lo, hi = tri.reshape(-1, 3).min(axis=0), tri.reshape(-1, 3).max(axis=0)
    ext = float((hi - lo).max())
    X, Y, Z, L = lst["X"], lst["Y"], lst["Z"], lst["L"]
    checked = listed_rays = 0
    for i in range(n_rays):
        kind = i % 3
        if kind == 0:    # on the mesh's own surface (a hair off it, either side)
            t3 = tri[rnd.randrange(len(tri))]
            a, b = rnd.random(), rnd.random()
            if a + b > 1:
                a, b = 1 - a, 1 - b
            n = np.cross(t3[1] - t3[0], t3[2] - t3[0])
            o = t3[0] + a * (t3[1] - t3[0]) + b * (t3[2] - t3[0]) + n / (np.linalg.norm(n) + 1e-300) * rnd.uniform(-1e-3, 1e-3) * ext
        elif kind == 1:  # on a plane below / beside the mesh: where its shadow falls
            o = np.array([rnd.uniform(lo[0] - ext, hi[0] + ext), rnd.uniform(lo[1] - ext, hi[1] + ext), lo[2] - rnd.uniform(0, 0.3) * ext])
        else:            # anywhere around it
            o = np.array([rnd.uniform(lo[k] - 0.5 * ext, hi[k] + 0.5 * ext) for k in range(3)])
        o = o.astype(np.float32).astype(np.float64)  # a device origin is a float
        if point:
            dvec = lvec - o
            tmax = float(np.linalg.norm(dvec))
            dirs = dvec / tmax
        else:
            dirs, tmax = -lvec / np.linalg.norm(lvec), float("inf")
        h = np.nonzero(hits(tri, o, dirs, tmax))[0]
        v = o - L
        u, w, depth = float(v @ X), float(v @ Y), float(v @ Z)
        if point:
            if not depth > 0:
                assert len(h) == 0, "an origin behind the pinhole plane hits the mesh"
                continue
            u, w = u / depth, w / depth
        tu, tw = (u - lst["u0"]) * lst["su"], (w - lst["v0"]) * lst["sv"]
        checked += 1
        if not (0 <= tu < G and 0 <= tw < G):
            assert len(h) == 0, "an origin outside the grid hits triangles %s" % h[:4]
            continue
        cell = int(tw) * G + int(tu)
        faces = lst["entry_face"][off[cell]:off[cell + 1]]
        zmin = lst["entry_zmin"][off[cell]:off[cell + 1]]
        assert np.all(np.diff(zmin) >= 0), "a cell's entries are not sorted by zmin"
        listed_rays += len(faces) > 0
        for f in h:
            where = np.nonzero(faces == f)[0]
            assert len(where) == 1, "ray %d hits face %d, which cell %d does not list (%d entries)" % (i, f, cell, len(faces))
            assert zmin[where[0]] <= depth, "face %d is listed beyond the origin's depth (%g > %g): the walk would end before it" % (f, zmin[where[0]], depth)
    return checked, listed_rays


@pytest.mark.parametrize("tag,lights", [("teapot2_240x135", 2), ("p11_240x135", 1), ("p13_200x150", None)])
def test_every_triangle_a_shadow_ray_can_hit_is_in_its_origins_cell(pkg, golden, tag, lights):
    scene = golden(tag).scene(pkg)
    rnd = random.Random(7)
    usable = 0
    for ls in range(4):
        try:
            r = check_list(pkg, scene, ls, 0, rnd, 1500)
        except pkg.RtuError:
            break  # no such light / mesh node
        if r is not None:
            usable += 1
            assert r[0] > 500 and r[1] > 100, r  # the rays did exercise the list
    if lights is not None:
        assert usable == lights


def test_lists_of_transformed_instanced_meshes_and_awkward_lights(pkg, tmp_path):
    """Nested non-uniform transformations, two nodes of one mesh, a direct light grazing the mesh, a point light far away and one
    INSIDE the hull (no list from there: usable == 0, and the device then walks the BVH)."""
    from test_gpu_parity import _write_uv_mesh

    def torus(u, v):
        a, b = 2 * math.pi * u, 2 * math.pi * v
        return ((2 + 0.7 * math.cos(b)) * math.cos(a), (2 + 0.7 * math.cos(b)) * math.sin(a), 0.7 * math.sin(b))
    _write_uv_mesh(tmp_path / "torus.obj", 24, 10, torus)
    xml = tmp_path / "s.xml"
    xml.write_text("""<xml><scene>
      <object name="g"><rotate angle="25" x="1" y="0.3" z="0.2"/><translate x="1" y="-2" z="3"/>
        <object type="obj" name="{o}" material="m"><scale x="1.5" y="0.7" z="2.0"/><rotate angle="40" z="1"/><translate x="-2" z="1"/></object></object>
      <object type="obj" name="{o}" material="m"><scale value="0.8"/><translate x="5" y="3" z="1"/></object>
      <object type="plane" name="floor" material="m"><scale value="30"/><translate z="-4"/></object>
      <material type="blinn" name="m"><diffuse r="0.6" g="0.6" b="0.6"/></material>
      <light type="point" name="far"><intensity value="0.5"/><position x="40" y="-60" z="50"/></light>
      <light type="direct" name="grazing"><intensity value="0.3"/><direction x="1" y="0.2" z="-0.05"/></light>
      <light type="point" name="inside"><intensity value="0.4"/><position x="5" y="3" z="1"/></light>
      <light type="direct" name="down"><intensity value="0.3"/><direction x="0" y="0" z="-1"/></light>
    </scene><camera><position x="0" y="-20" z="6"/><target x="0" y="0" z="1"/><up x="0" y="0" z="1"/><fov value="45"/>
      <width value="64"/><height value="48"/></camera></xml>""".format(o=tmp_path / "torus.obj"))
    scene = pkg.Scene.from_xml(str(xml))
    rnd = random.Random(11)
    got = {}
    for ls in range(4):
        for cs in range(2):
            got[(ls, cs)] = check_list(pkg, scene, ls, cs, rnd, 900)
    assert got[(2, 1)] is None, "a light inside the mesh's hull has no list"
    assert all(got[k] is not None for k in [(0, 0), (0, 1), (1, 0), (1, 1), (3, 0), (3, 1)])
