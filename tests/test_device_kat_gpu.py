"""GPU: known-answer tests of the device layers UNDER the MSM, element by element, against vectors minted by the reference's own
classes (second capture of oracle/mint_golden.cpp -- extfield_<curve>.bin, groupkat_<curve>_g<k>.bin) and against the oracle:

* Fq2 (MNT4753) / Fq3 (MNT6753) -- the coordinate field of G2 -- in both device forms: one lane per element (Karatsuba) and the
  lane-split form the G2 point kernels run (fused two- / three-product multipliers with ds_bpermute exchange); reference
  depends/libff/libff/algebra/fields/fp2.tcc:58-142, fp3.tcc:59-143;
* every form of the group law inside the kernels -- the point VM's addition / mixed addition / doubling, the straight-line mixed and
  full additions, the two-point-lanes addition -- for the four groups and, for G2, both lane configurations; reference
  mnt4753_g1.cpp:134-346, mnt4753_g2.cpp:150-362, mnt6753_g1.cpp, mnt6753_g2.cpp:156-368;
* the affine pair addition of the batched-affine levels (k_pair_level: addition, doubling, cancellation kinds, sign flags) through
  two-point MSMs whose entries meet in one bucket, against the same golden sums and differences.
Through the test hooks mnt753_test_ext_op / mnt753_test_point_op of the C ABI (include/mnt753_hip.h)."""
import numpy as np
import pytest

import golden_io as G
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _deg(curve):
    return 2 if curve == 0 else 3


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("split", [0, 1])
def test_device_extension_field_vs_libff_goldens(gpu, curve, split):
    g = G.extfield(curve)
    w = 12 * _deg(curve)
    a, b = g[:, 0].copy(), g[:, 1].copy()
    T = gpu.api.test_ext_op
    assert np.array_equal(T(curve, split, 0, a, b).reshape(-1, w), g[:, 2]), "a * b"
    assert np.array_equal(T(curve, split, 1, a).reshape(-1, w), g[:, 3]), "a * a vs squared()"
    assert np.array_equal(T(curve, split, 2, a).reshape(-1, w), g[:, 4]), "inverse"
    assert np.array_equal(T(curve, split, 3, a, b).reshape(-1, w), g[:, 5]), "a + b"
    assert np.array_equal(T(curve, split, 4, a, b).reshape(-1, w), g[:, 6]), "a - b"
    # a * a^-1 = 1, and the zero test the kernels branch on: (a == a) -> 1, (a == b) -> 0 unless the golden row has a == b
    one = g[0, 0]
    assert np.array_equal(T(curve, split, 0, a, g[:, 4].copy()).reshape(-1, w), np.tile(one, (len(a), 1)))
    eq = T(curve, split, 6, a, a).reshape(-1, w)
    assert np.array_equal(eq, np.tile(one, (len(a), 1)))
    ne = T(curve, split, 6, a, b).reshape(-1, w)
    for i in range(len(a)):
        assert np.array_equal(ne[i], one if np.array_equal(a[i], b[i]) else np.zeros(w, dtype=np.uint64)), i


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("split", [0, 1])
def test_device_extension_field_ragged_batch_vs_oracle(gpu, curve, split):
    """131 elements: more than one wave, more than one workgroup of the three-lane form (84 elements), a ragged tail; components
    that are zero or p - 1; against the oracle (itself pinned to the same golden file)."""
    deg, n = _deg(curve), 131
    w = 12 * deg
    a = gpu.synth_scalars(1 - curve, 300 + curve, n * deg).reshape(n, w)   # Fq of MNT4753 = Fr of MNT6753 and vice versa
    b = gpu.synth_scalars(1 - curve, 310 + curve, n * deg).reshape(n, w)
    m1 = G.field("B" if curve == 0 else "A")[2, 0]                         # -1 of the coordinate field's base field
    a[3, :12] = 0; a[4, 12:24] = 0; a[5, :12] = m1; b[6, 12:24] = m1; a[7] = 0; a[7, :12] = m1
    T = gpu.api.test_ext_op
    for op in (0, 1, 2, 3, 4, 5):
        want = np.stack([O.ext_op(curve, 0 if op == 1 else op, x, x if op == 1 else y) for x, y in zip(a, b)])
        assert np.array_equal(T(curve, split, op, a, b).reshape(n, w), want), f"op {op}"


def _proj(gpu, curve, group, aff):
    return gpu.point_from_affine(curve, group, aff)


def _cases(gpu, curve, group):
    """(P, Q, expected affine sum, Q is affine-normalised) from the golden records: P + Q, 2P + Q, 2P + 3Q with the doubled /
    tripled operands in whatever projective representative the host addition leaves (Z != 1)."""
    out = []
    for r in G.groupkat(curve, group):
        P, Q = _proj(gpu, curve, group, r["P"]), _proj(gpu, curve, group, r["Q"])
        P2 = gpu.point_add(curve, group, P, P)
        Q3 = gpu.point_add(curve, group, gpu.point_add(curve, group, Q, Q), Q)
        out.append((P, Q, r["sum"], True))
        out.append((P2, Q, r["dbl_madd"], True))
        out.append((P2, Q3, r["dbl_add3"], False))
        out.append((Q3, P2, r["dbl_add3"], False))
    return out


FORMS = {0: "VM addition", 2: "VM mixed addition", 3: "straight-line mixed addition", 4: "two point-lanes per addition", 5: "straight-line addition",
         6: "one group of lanes per addition"}


@pytest.mark.parametrize("curve,group,split", [(0, 1, 0), (1, 1, 0), (0, 2, 0), (0, 2, 1), (1, 2, 0), (1, 2, 1)])
@pytest.mark.parametrize("form", sorted(FORMS))
def test_device_group_law_vs_libff_goldens(gpu, curve, group, split, form):
    if form == 4 and group == 2 and not split:
        pytest.skip("the two-lanes addition exists for base fields and the lane-split fields")
    if form == 6 and split:
        pytest.skip("the lane-group addition is instantiated with the one-lane configuration (its fallback picks the lane-split one itself)")
    cases = _cases(gpu, curve, group)
    if form in (2, 3):
        cases = [c for c in cases if c[3]]     # mixed additions read Q as an affine point
    P = np.stack([c[0] for c in cases]); Q = np.stack([c[1] for c in cases])
    got = gpu.api.test_point_op(curve, group, split, form, P, Q).reshape(len(cases), -1)
    for k, (c, g) in enumerate(zip(cases, got)):
        assert np.array_equal(gpu.point_to_affine(curve, group, g), c[2]), f"{FORMS[form]}: case {k} (record {k // 4}, kind {k % 4})"


@pytest.mark.parametrize("curve,group,split", [(0, 1, 0), (1, 1, 0), (0, 2, 0), (0, 2, 1), (1, 2, 0), (1, 2, 1)])
def test_device_doubling_vs_libff_goldens(gpu, curve, group, split):
    recs = G.groupkat(curve, group)
    P = np.stack([_proj(gpu, curve, group, r["P"]) for r in recs])
    got = gpu.api.test_point_op(curve, group, split, 1, P).reshape(len(recs), -1)
    for r, g in zip(recs, got):
        assert np.array_equal(gpu.point_to_affine(curve, group, g), r["dbl"])
    # doubling a point that is not normalised: 2 (2P) = 4P against the oracle
    P2 = np.stack([gpu.point_add(curve, group, p, p) for p in P])
    got = gpu.api.test_point_op(curve, group, split, 1, P2).reshape(len(recs), -1)
    for r, g in zip(recs, got):
        assert np.array_equal(gpu.point_to_affine(curve, group, g), O.point_op(curve, group, 1, r["dbl"]))


@pytest.mark.parametrize("curve,group,split", [(0, 1, 0), (1, 1, 0), (0, 2, 1), (1, 2, 1)])
def test_device_group_law_ragged_batch_vs_oracle(gpu, curve, group, split):
    """173 random pairs (several waves and workgroups of every lane geometry, a ragged tail) with equal and opposite points
    sprinkled in, every form, against the oracle's operator+."""
    n = 173
    pts = gpu.synth_points(curve, group, 401, 2 * n)
    A, B = pts[:n].copy(), pts[n:].copy()
    B[5] = A[5]; B[64] = A[64]; B[100] = A[100]                      # equal points -> the doubling side path
    neg = O.point_op(curve, group, 2, np.zeros_like(A[7]), A[7])      # O - A = -A
    B[7] = neg
    want = np.stack([O.point_op(curve, group, 0, a, b) for a, b in zip(A, B)])
    P = np.stack([_proj(gpu, curve, group, a) for a in A]); Q = np.stack([_proj(gpu, curve, group, b) for b in B])
    for form in sorted(FORMS):
        got = gpu.api.test_point_op(curve, group, 0 if form == 6 else split, form, P, Q).reshape(n, -1)
        res = np.stack([gpu.point_to_affine(curve, group, g) for g in got])
        assert np.array_equal(res, want), FORMS[form]


@pytest.mark.parametrize("curve,group", [(0, 1), (1, 1), (0, 2), (1, 2)])
def test_lane_group_addition_identities_and_coordinates(gpu, curve, group):
    """The lane-group addition (msm_flow.hip.h) on the cases the golden records do not hold: an identity on either side or both, equal
    points next to ordinary ones in one wave (the fallback runs on a few lanes of a group while its neighbours are done), a count that
    leaves groups of the last wave without an addition -- and the same PROJECTIVE coordinates as the VM's addition, not only the same
    point (both follow operator+ of the reference line by line, so every coordinate agrees mod p)."""
    n = 37
    pts = gpu.synth_points(curve, group, 977, 2 * n)
    A = np.stack([_proj(gpu, curve, group, a) for a in pts[:n]]); B = np.stack([_proj(gpu, curve, group, b) for b in pts[n:]])
    A = np.stack([gpu.point_add(curve, group, a, a) for a in A])                   # Z != 1 on one side
    zero = _proj(gpu, curve, group, np.zeros_like(pts[0]))
    A[3] = zero; B[8] = zero; A[20] = zero; B[20] = zero
    for i in (0, 1, 2, 17, 36): B[i] = A[i]
    split = 1 if group == 2 else 0
    vm = gpu.api.test_point_op(curve, group, split, 0, A, B).reshape(n, -1)
    flow = gpu.api.test_point_op(curve, group, 0, 6, A, B).reshape(n, -1)
    for i in range(n):
        assert np.array_equal(gpu.point_to_affine(curve, group, flow[i]), gpu.point_to_affine(curve, group, vm[i])), i
    ordinary = [i for i in range(n) if i not in (0, 1, 2, 3, 8, 17, 20, 36)]
    assert np.array_equal(flow[ordinary], vm[ordinary])      # the hooks return canonical coordinates
