import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
tag = sys.argv[1]
sc = pkg.Scene.from_blob_file('tests/golden/%s/scene.rtus.gz' % tag)
import json
meta = json.load(open('tests/golden/%s/meta.json' % tag))
W, H = meta['width'], meta['height']
ctx = pkg.Context(0)
ctx.upload(sc)
cnt, _ = ctx.render(pkg.frame_setup(sc.desc.camera, W, H, collect_stats=True), stats=True)
fast, _ = ctx.render(pkg.frame_setup(sc.desc.camera, W, H))
bad = np.argwhere((cnt.view(np.uint32) != fast.view(np.uint32)).any(axis=2))
print(tag, W, H, 'mismatching pixels', len(bad))
if len(bad):
    ys, xs = bad[:, 0], bad[:, 1]
    print(' y range', ys.min(), ys.max(), 'x range', xs.min(), xs.max())
    tiles = sorted(set((int(y) // 8, int(x) // 8) for y, x in bad))
    print(' tiles', len(tiles), tiles[:40])
    y, x = bad[0]
    print(' first', y, x, cnt[y, x], fast[y, x])
