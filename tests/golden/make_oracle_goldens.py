#!/usr/bin/env python3
"""Fixtures of the sampled recipes at FULL size, from the CPU oracle on the KEYED sample streams (the streams the device
follows; the oracle itself is pinned bit for bit to the compiled reference on the sequential streams by the fixtures of
make_goldens.py, and the two streams are the same estimator: tests/test_oracle.py). Stored: every 8th pixel (z, linear RGB)
and the sha256 of the full float z. Input scene: the blob of the tag's 1080p recipe-W fixture.

  p11_p4_1080   BASELINE config 5's scene and size (Project11 @1920x1080, recipe P = samples + the 4-bounce Monte-Carlo
                gather) at 4 samples per pixel — the size at which tests/test_gpu_sampled.py checks the device."""
import hashlib, json, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
import __graft_entry__ as g

CONFIGS = [("p11_p4_1080", "p11_1080", 4, "P")]


def main():
    pkg, orc = g.load_package(), g.load_oracle()
    for tag, src, spp, recipe in CONFIGS:
        scene = pkg.Scene.from_blob_file(os.path.join(HERE, src, "scene.rtus.gz"))
        meta = json.load(open(os.path.join(HERE, src, "meta.json")))
        W, H = meta["width"], meta["height"]
        fn = orc.render_paths if recipe == "P" else orc.render_samples
        img, st = fn(scene, W, H, spp, stream=orc.STREAM_KEYED, trig=orc.TRIG_PORTABLE, threads=os.cpu_count() or 8)
        dst = os.path.join(HERE, tag)
        os.makedirs(dst, exist_ok=True)
        np.savez_compressed(os.path.join(dst, "golden.npz"), z_sub8=img[::8, ::8, 3].copy(), rgb_sub8=img[::8, ::8, :3].copy())
        json.dump({"scene_blob_of": src, "width": W, "height": H, "recipe": recipe, "spp": spp, "stream": "keyed", "trig": "portable",
                   "sha256_z_f32": hashlib.sha256(np.ascontiguousarray(img[..., 3]).tobytes()).hexdigest(),
                   "primary": st["primary_rays"], "secondary": st["secondary_rays"], "shadow": st["shadow_rays"]},
                  open(os.path.join(dst, "meta.json"), "w"), indent=1, sort_keys=True)
        print(tag, st["primary_rays"], st["secondary_rays"], st["shadow_rays"])


if __name__ == "__main__":
    main()
