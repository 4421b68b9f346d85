"""Development check (GPU box): small MSMs through the C ABI vs Python big integers."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from __graft_entry__ import load_package
import pyref
pkg = load_package(); pkg.init(0)

def run(curve_id, group, n, seed, special=False):
    cv = pyref.Curve(curve_id); rng = pyref.splitmix64(seed)
    G = cv.gen(group)
    ks = [pyref.rand_below(rng, 1 << 20) + 1 for _ in range(n)]
    pts = [cv.mul(k, G, group) for k in ks]
    scal = [pyref.rand_below(rng, cv.r) for _ in range(n)]
    if special and n >= 8:
        scal[0] = 0; scal[1] = 1; scal[2] = cv.r - 1; pts[3] = None; ks[3] = 0
        pts[5] = pts[4]; ks[5] = ks[4]; scal[5] = scal[4]          # duplicate base & scalar -> doubling path
        pts[7] = cv.neg(pts[6]); ks[7] = -ks[6]; scal[7] = scal[6]  # P + (-P)
    for p in pts: assert cv.on_curve(p, group)
    aff = np.array([cv.affine_to_words(p, group) for p in pts], dtype=np.uint64)
    sw = np.array([cv.fr_to_words(s) for s in scal], dtype=np.uint64)
    t0 = time.time()
    bs = pkg.BaseSet(curve_id, group, aff)
    out = bs.msm(sw)
    dt = time.time() - t0
    got = cv.projective_from_words([int(v) for v in out], group)
    e = sum(s * k for s, k in zip(scal, ks)) % cv.r
    exp = cv.mul(e, G, group)
    ok = got == exp
    print(f"curve={curve_id} group={group} n={n} special={special}: {'OK' if ok else 'MISMATCH'}  ({dt*1e3:.1f} ms)", flush=True)
    return ok

allok = True
which = sys.argv[1] if len(sys.argv) > 1 else "g1"
if which in ("g1", "all"):
    for n, sp in [(1, False), (2, False), (17, True), (300, True), (3000, False)]:
        allok &= run(0, 1, n, 100 + n, sp)
    allok &= run(1, 1, 200, 7, True)
if which in ("g2", "all"):
    allok &= run(0, 2, 40, 11, True)
    allok &= run(1, 2, 40, 12, True)
print("ALL OK" if allok else "FAILURES")
sys.exit(0 if allok else 1)
