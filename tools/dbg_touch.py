import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
tag = sys.argv[1] if len(sys.argv) > 1 else "teapot2_1080"
sc = pkg.Scene.from_blob_file('tests/golden/%s/scene.rtus.gz' % tag)
meta = json.load(open('tests/golden/%s/meta.json' % tag))
W, H = meta['width'], meta['height']
ctx = pkg.Context(0)
ctx.upload(sc)
fr = pkg.frame_setup(sc.desc.camera, W, H, collect_stats=2)
fr.coop_threshold = 1
ctx.render(fr)
t = ctx.touched(False)
for k, v in t.items():
    if v.get('rays') or v.get('bytes'):
        print(k, {a: b for a, b in v.items() if b})
