import sys, time, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from parity_util import make_pair, compare
for name, W in [("reflect_test", 128), ("shadowmodel", 128), ("NthTest", 128), ("glass", 96), ("toy", 160), ("toy_on_plane", 128), ("dodgeColorTest", 96), ("bunny", 200)]:
    t0 = time.time()
    fly, orc = make_pair(name, width=W, height=W)
    info = fly.info().as_dict()
    r = compare(fly, orc, W, W)
    r = {k: v for k, v in r.items() if k not in ("got", "ref", "ref_nanfree")}
    print(name, json.dumps(info), flush=True)
    print("   ", json.dumps(r), "%.1fs" % (time.time() - t0), flush=True)
