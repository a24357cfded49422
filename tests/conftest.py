import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
REFERENCE = "/root/reference"  # exists only in the authoring container, never on the GPU box

SMALL_TAGS = ["p1_256", "p4_240x135", "teapot2_240x135", "p11_240x135"]
FULL_TAGS = ["p3s_800x600", "p4_1080", "teapot2_1080", "p11_1080"]
EXTRA_TAGS = ["p1test_200x150", "p2_200x150", "p3box_200x150", "p5_200x150", "p5low_200x150", "p11simple_200x150",
              "p13_200x150"]  # the reference's other deterministic scenes
SMALL_TAGS = SMALL_TAGS + EXTRA_TAGS
ALL_TAGS = SMALL_TAGS + FULL_TAGS
TEX_TAGS = ["p7_200x150", "mtl_160x120"]  # textured; mtl: an .obj with its own .mtl materials -> MultiMtl (row f3)
LOCAL_SCENES = {"mtl_160x120": "multimtl/scene.xml"}  # scenes written for this repository (tests/scenes), not the reference's


def instantiate_scene(rel, dst):
    """Copy tests/scenes/<dir of rel> to dst with its @DIR@ placeholders (absolute paths of .obj / texture files,
    as the reference's scene files carry them) pointing at dst; returns the path of the scene XML."""
    src = os.path.join(REPO, "tests", "scenes", os.path.dirname(rel))
    os.makedirs(str(dst), exist_ok=True)
    for name in os.listdir(src):
        data = open(os.path.join(src, name), "rb").read()
        if name.endswith((".xml", ".mtl", ".obj")):
            data = data.replace(b"@DIR@", str(dst).encode())
        open(os.path.join(str(dst), name), "wb").write(data)
    return os.path.join(str(dst), os.path.basename(rel))
# stochastic effects, recipe S: glossy + soft + textured; depth of field; glossy + soft; 12 soft lights + glossy refraction; teapot + soft;
# glossy under a hard light (Project11/scene_glossy.xml: with it every scene file of the reference has a fixture)
SAMPLED_TAGS = ["p10_s4_160x120", "p9_s3_160x120", "p11gs_s2_160x90", "p11x86_s1_120x90", "teapot1_s2_160x90", "p11g_s2_160x90"]
PATH_TAGS = ["p11_p2_120x68", "p13_p2_96x72"]  # recipe P (config 5): + the Monte-Carlo gather


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def orc():
    import __graft_entry__ as g
    return g.load_oracle()


class Golden:
    def __init__(self, tag):
        self.tag = tag
        self.dir = os.path.join(GOLDEN, tag)
        self.meta = json.load(open(os.path.join(self.dir, "meta.json")))
        self.width, self.height = self.meta["width"], self.meta["height"]
        self._npz = None

    @property
    def npz(self):
        if self._npz is None:
            self._npz = np.load(os.path.join(self.dir, "golden.npz"))
        return self._npz

    def scene(self, pkg):
        return pkg.Scene.from_blob_file(os.path.join(self.dir, "scene.rtus.gz"))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(tag):
        if tag not in cache:
            cache[tag] = Golden(tag)
        return cache[tag]
    return get


def sha256(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def read_png(path):
    """Decode an 8-bit PNG with a tiny zlib-based reader (grey / RGB / palette / grey+alpha
    / RGBA, bit depths 1-8, non-interlaced) — enough for lodepng's auto-converted output."""
    import struct
    import zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, plte = 8, b"", None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif typ == b"IDAT":
            idat += body
    assert interlace == 0
    chans = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp_bits = chans * depth
    stride = (w * bpp_bits + 7) // 8
    bpp = max(1, bpp_bits // 8)
    raw = zlib.decompress(idat)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    p = 0
    for y in range(h):
        ft = raw[p]
        line = np.frombuffer(raw, np.uint8, stride, p + 1).astype(np.int32)
        p += 1 + stride
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(stride, np.int32)
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    if depth < 8:
        bits = np.unpackbits(out, axis=1)[:, :w * bpp_bits].reshape(h, w * chans, depth)
        vals = (bits * (1 << np.arange(depth - 1, -1, -1))).sum(axis=2).astype(np.uint8)
        if ctype == 0:
            vals = (vals.astype(np.uint32) * 255 // ((1 << depth) - 1)).astype(np.uint8)
        px = vals.reshape(h, w, chans)
    else:
        px = out.reshape(h, w, chans)
    if ctype == 3:
        return plte[px[..., 0]]
    if ctype == 0:
        return px[..., 0]
    if ctype == 4:
        return px[..., 0]
    if ctype == 6:
        return px[..., :3]
    return px
