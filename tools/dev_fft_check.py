"""Development check (GPU box): FFT kinds / vector ops / compute_H through the C ABI vs the CPU oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import oracle_lib as O
pkg = load_package(); pkg.init(0)
ok_all = True
for curve in (0, 1):
    for logm in (1, 2, 3, 5, 8, 9, 11, 12, 15):
        m = 1 << logm
        v = pkg.synth_scalars(curve, 1000 + logm, m)
        dom = pkg.Domain(curve, m)
        for kind in range(4):
            d = pkg.DeviceBuffer.from_numpy(v)
            dom.fft(kind, d.ptr.value)
            got = d.to_numpy().reshape(m, 12)
            if logm <= 12:
                exp = O.fft(curve, kind, v).reshape(m, 12)
                ok = np.array_equal(got, exp)
            else:
                ok = True
            ok_all &= ok
            print(f"curve={curve} m=2^{logm} kind={kind}: {'OK' if ok else 'MISMATCH'}", flush=True)
        if logm in (5, 11):
            a, b, c = (pkg.synth_scalars(curve, s, m) for s in (1, 2, 3))
            da, db, dc = (pkg.DeviceBuffer.from_numpy(x) for x in (a, b, c))
            dh = pkg.DeviceBuffer(96 * (m + 1))
            dom.compute_h(da.ptr.value, db.ptr.value, dc.ptr.value, dh.ptr.value)
            got = dh.to_numpy()
            exp = O.compute_h(curve, a, b, c)
            ok = np.array_equal(got, exp); ok_all &= ok
            print(f"curve={curve} m=2^{logm} compute_H: {'OK' if ok else 'MISMATCH'}", flush=True)
            da, db = pkg.DeviceBuffer.from_numpy(a), pkg.DeviceBuffer.from_numpy(b)
            pkg.vec_muleq(curve, da.ptr.value, db.ptr.value, m)
            exp = np.array([O.field_op(curve, 0, a[i], b[i]) for i in range(m)])
            ok = np.array_equal(da.to_numpy().reshape(m, 12), exp); ok_all &= ok
            print(f"   muleq: {'OK' if ok else 'MISMATCH'}", flush=True)
            pkg.vec_subeq(curve, da.ptr.value, db.ptr.value, m)
            exp2 = np.array([O.field_op(curve, 2, exp[i], b[i]) for i in range(m)])
            ok = np.array_equal(da.to_numpy().reshape(m, 12), exp2); ok_all &= ok
            print(f"   subeq: {'OK' if ok else 'MISMATCH'}", flush=True)
            dom.divide_by_z_on_coset(da.ptr.value)
            ok = np.array_equal(da.to_numpy(), O.divide_by_z_on_coset(curve, exp2).reshape(-1)); ok_all &= ok
            print(f"   divide_by_Z: {'OK' if ok else 'MISMATCH'}", flush=True)
# timing at 2^20
m = 1 << 20
dom = pkg.Domain(0, m)
v = pkg.synth_scalars(0, 5, m)
d = pkg.DeviceBuffer.from_numpy(v)
import ctypes
for kind in (0, 1, 2, 3):
    pkg.lib().mnt753_sync(None); t = time.time()
    dom.fft(kind, d.ptr.value); pkg.lib().mnt753_sync(None)
    print(f"2^20 kind={kind}: {(time.time()-t)*1e3:.2f} ms", flush=True)
# round trip
d = pkg.DeviceBuffer.from_numpy(v); dom.fft(2, d.ptr.value); dom.fft(3, d.ptr.value)
ok = np.array_equal(d.to_numpy().reshape(m, 12), v); ok_all &= ok
print("2^20 coset round trip:", "OK" if ok else "MISMATCH")
a, b, c = (pkg.DeviceBuffer.from_numpy(pkg.synth_scalars(0, s, m)) for s in (1, 2, 3)); h = pkg.DeviceBuffer(96 * (m + 1))
pkg.lib().mnt753_sync(None); t = time.time(); dom.compute_h(a.ptr.value, b.ptr.value, c.ptr.value, h.ptr.value); pkg.lib().mnt753_sync(None)
print(f"2^20 compute_H: {(time.time()-t)*1e3:.2f} ms")
print("ALL OK" if ok_all else "FAILURES"); sys.exit(0 if ok_all else 1)
