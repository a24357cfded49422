#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the COMPILED REFERENCE.

Authoring container only (needs /root/reference and oracle/_ref/ref_render, built
by `make -C oracle`). For every BASELINE config it runs the reference's own
Trace/Shade functions through oracle/ref_harness/driver.cpp ("recipe W",
SURVEY.md §8c; "recipe S" for the scenes with stochastic effects) and stores DATA only:

  scene.rtus.gz   flattened scene (input), serialised from the reference's
                  in-memory scene graph after LoadScene()
  Result.png      written by the reference's RenderImage::SaveImage (lodepng)
  ZBuffer.png     written by RenderImage::SaveZImage after ComputeZBufferImage
  golden.npz      full float z + linear RGB for the small configs; for 1080p an
                  every-8th-pixel subsample (z, rgb) plus the 8-bit images
  meta.json       resolution, ray counters, sha256 of the full float z / rgb /
                  8-bit buffers

No reference source text is copied; the fixtures are inputs and outputs.
"""
import gzip, hashlib, json, os, shutil, subprocess, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
RUN = os.path.join(REPO, "oracle", "ref_harness", "run_ref.sh")

CONFIGS = [
    # tag, scene (relative to SceneFiles), W, H, keep full floats
    ("p1_256", "Project1Example.xml", 256, 256, True),
    ("p3s_800x600", "Project3Simple.xml", 800, 600, True),
    ("p4_1080", "Project4.xml", 1920, 1080, False),
    ("teapot2_1080", "Teapot/scene2.xml", 1920, 1080, False),
    ("p11_1080", "Project11/scene.xml", 1920, 1080, False),
    # small versions of the 1080p configs so CPU-only tests stay fast
    ("p4_240x135", "Project4.xml", 240, 135, True),
    ("teapot2_240x135", "Teapot/scene2.xml", 240, 135, True),
    ("p11_240x135", "Project11/scene.xml", 240, 135, True),
    # the reference's other deterministic, untextured scenes, as extra regression inputs
    ("p1test_200x150", "Project1Test.xml", 200, 150, True),
    ("p2_200x150", "Project2.xml", 200, 150, True),
    ("p3box_200x150", "Project3Box.xml", 200, 150, True),
    ("p5_200x150", "Project5/scene.xml", 200, 150, True),
    ("p5low_200x150", "Project5/scene-low.xml", 200, 150, True),
    ("p11simple_200x150", "Project11/scene_simple.xml", 200, 150, True),
    ("p13_200x150", "Project13/scene.xml", 200, 150, True),
    # textures ("next" row f2): checkerboards, two 1024x1024 PNGs (mesh diffuse, background, environment)
    ("p7_200x150", "Project7/scene.xml", 200, 150, True),
    # stochastic effects ("next" row f1), recipe S: the 7th field is samples per pixel. The reference is built
    # with rand() wrapped to the sequential sample stream (oracle/ref_harness/Makefile), so these are
    # reproducible: glossy reflection + refraction + a size-5 light + textures; depth of field + textures;
    # glossy reflections + a size-5 light; twelve size-1 lights + glossy refraction; the teapot under a size-5 light
    ("p10_s4_160x120", "Project10/scene.xml", 160, 120, True, 4),
    ("p9_s3_160x120", "Project9/scene.xml", 160, 120, True, 3),
    ("p11gs_s2_160x90", "Project11/scene_glossy_soft.xml", 160, 90, True, 2),
    ("p11x86_s1_120x90", "Project11/scene_86.xml", 120, 90, True, 1),
    ("teapot1_s2_160x90", "Teapot/scene.xml", 160, 90, True, 2),
    # the last scene file of the reference without a fixture (round 3): glossy reflections under a hard point light
    ("p11g_s2_160x90", "Project11/scene_glossy.xml", 160, 90, True, 2),
    # row f3: an .obj that brings its own materials (usemtl / .mtl -> MultiMtl, xmlload.cpp:199-243) — a scene written for
    # this repository (tests/scenes/multimtl, "@" = repository path), run through the compiled reference like the others
    ("mtl_160x120", "@tests/scenes/multimtl/scene.xml", 160, 120, True),
    # recipe P (config 5): recipe S plus the 4-bounce Monte-Carlo gather of Render(); 8th field "P"
    ("p11_p2_120x68", "Project11/scene.xml", 120, 68, True, 2, "P"),
    ("p13_p2_96x72", "Project13/scene.xml", 96, 72, True, 2, "P"),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    only = set(sys.argv[1:])
    for cfg in CONFIGS:
        tag, scene, W, H, full = cfg[:5]
        spp = cfg[5] if len(cfg) > 5 else 0
        paths = len(cfg) > 6 and cfg[6] == "P"
        if only and tag not in only:
            continue
        scene_arg = os.path.join(REPO, scene[1:]) if scene.startswith("@") else scene
        subprocess.check_call([RUN, scene_arg, str(W), str(H), tag, "8"] + ([str(spp)] if spp else []) + (["paths"] if paths else []))
        src = os.path.join(REPO, "oracle", "_ref", "out", tag)
        dst = os.path.join(HERE, tag)
        os.makedirs(dst, exist_ok=True)
        with open(os.path.join(src, "scene.rtus"), "rb") as f, gzip.GzipFile(
            os.path.join(dst, "scene.rtus.gz"), "wb", mtime=0) as g:
            g.write(f.read())
        for png in ("Result.png", "ZBuffer.png"):
            shutil.copyfile(os.path.join(src, png), os.path.join(dst, png))
        z = np.fromfile(os.path.join(src, "z.f32"), np.float32).reshape(H, W)
        rgb = np.fromfile(os.path.join(src, "rgb.f32"), np.float32).reshape(H, W, 3)
        res8 = np.fromfile(os.path.join(src, "result.u8"), np.uint8).reshape(H, W, 3)
        z8 = np.fromfile(os.path.join(src, "zbuffer.u8"), np.uint8).reshape(H, W)
        stats = json.load(open(os.path.join(src, "stats.json")))
        meta = {
            "scene": scene, "width": W, "height": H, "recipe": "P" if paths else "S" if spp else "W", "spp": spp,
            "stream": "sequential" if spp else None,
            "primary": stats["primary"], "primary_hits": stats["primary_hits"],
            "secondary": stats["secondary"], "shadow": stats["shadow"],
            "sha256_z_f32": sha(z), "sha256_rgb_f32": sha(rgb),
            "sha256_result_u8": sha(res8), "sha256_zbuffer_u8": sha(z8),
            "sum_result_u8": int(res8.astype(np.uint64).sum()),
            "sum_zbuffer_u8": int(z8.astype(np.uint64).sum()),
            "nonzero_zbuffer_u8": int((z8 != 0).sum()),
        }
        arrays = {"result_u8": res8, "zbuffer_u8": z8}
        if full:
            arrays["z"] = z
            arrays["rgb"] = rgb
        else:
            arrays["z_sub8"] = z[::8, ::8].copy()
            arrays["rgb_sub8"] = rgb[::8, ::8].copy()
        np.savez_compressed(os.path.join(dst, "golden.npz"), **arrays)
        json.dump(meta, open(os.path.join(dst, "meta.json"), "w"), indent=1, sort_keys=True)
        print(tag, meta["primary_hits"], meta["secondary"], meta["shadow"])
        shutil.rmtree(src)  # the raw dumps are large (40 MB per 1080p config); the fixtures are what is kept


if __name__ == "__main__":
    sys.exit(main())
