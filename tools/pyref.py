"""Pure-Python big-integer model of the MNT4753 / MNT6753 objects on the prover hot path.

Development aid and synthetic-input generator (bench.py, tests): wire-format codecs, affine group
arithmetic, a naive MSM and a naive DFT.  It is NOT the oracle (oracle/ holds the C restatement of
the reference's algorithms) and nothing in the product path imports it.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mnt753_params as P  # noqa: E402

R_BITS = 768
R = 1 << R_BITS
MNT4, MNT6 = 0, 1


class Curve:
    def __init__(self, cid):
        self.id = cid
        if cid == MNT4:
            self.q, self.r = P.MOD_B, P.MOD_A
            self.a, self.b = P.MNT4_G1_A, P.MNT4_G1_B
            self.deg, self.nr = 2, P.MNT4_FQ2_NON_RESIDUE
            self.g1 = P.MNT4_G1_ONE
            self.g2 = P.MNT4_G2_ONE
            # twist: a' = (a*nr, 0), b' = (0, b*nr)          mnt4753_init.cpp:122-123
            self.a2 = (self.a * self.nr % self.q, 0)
            self.b2 = (0, self.b * self.nr % self.q)
        else:
            self.q, self.r = P.MOD_A, P.MOD_B
            self.a, self.b = P.MNT6_G1_A, P.MNT6_G1_B
            self.deg, self.nr = 3, P.MNT6_FQ3_NON_RESIDUE
            self.g1 = P.MNT6_G1_ONE
            self.g2 = P.MNT6_G2_ONE
            # twist: a' = (0, 0, a), b' = (b*nr, 0, 0)       mnt6753_init.cpp:133-136
            self.a2 = (0, 0, self.a)
            self.b2 = (self.b * self.nr % self.q, 0, 0)

    # ---- extension field (tuples of ints mod q) -------------------------------------------
    def fe(self, x, deg):
        return x if deg == 1 else tuple(x)

    def f_add(self, x, y):
        if isinstance(x, int):
            return (x + y) % self.q
        return tuple((a + b) % self.q for a, b in zip(x, y))

    def f_sub(self, x, y):
        if isinstance(x, int):
            return (x - y) % self.q
        return tuple((a - b) % self.q for a, b in zip(x, y))

    def f_neg(self, x):
        if isinstance(x, int):
            return (-x) % self.q
        return tuple((-a) % self.q for a in x)

    def f_mul(self, x, y):
        q, nr = self.q, self.nr
        if isinstance(x, int):
            return x * y % q
        if len(x) == 2:
            return ((x[0] * y[0] + nr * x[1] * y[1]) % q, (x[0] * y[1] + x[1] * y[0]) % q)
        a, b, c = x
        A, B, C = y
        return ((a * A + nr * (b * C + c * B)) % q, (a * B + b * A + nr * c * C) % q, (a * C + b * B + c * A) % q)

    def f_zero(self, like):
        return 0 if isinstance(like, int) else tuple(0 for _ in like)

    def f_is_zero(self, x):
        return x == 0 if isinstance(x, int) else all(v == 0 for v in x)

    def f_one(self, like):
        return 1 if isinstance(like, int) else tuple([1] + [0] * (len(like) - 1))

    def f_inv(self, x):
        q, nr = self.q, self.nr
        if isinstance(x, int):
            return pow(x, -1, q)
        if len(x) == 2:
            t = pow((x[0] * x[0] - nr * x[1] * x[1]) % q, -1, q)
            return (x[0] * t % q, (-x[1] * t) % q)
        a, b, c = x
        c0 = (a * a - nr * b * c) % q
        c1 = (nr * c * c - a * b) % q
        c2 = (b * b - a * c) % q
        t = pow((a * c0 + nr * (c * c1 + b * c2)) % q, -1, q)
        return (t * c0 % q, t * c1 % q, t * c2 % q)

    # ---- affine group law; None is the identity ------------------------------------------------
    def coeff_a(self, group):
        return self.a if group == 1 else self.a2

    def coeff_b(self, group):
        return self.b if group == 1 else self.b2

    def on_curve(self, pt, group):
        if pt is None:
            return True
        x, y = pt
        a, b = self.coeff_a(group), self.coeff_b(group)
        lhs = self.f_mul(y, y)
        rhs = self.f_add(self.f_add(self.f_mul(self.f_mul(x, x), x), self.f_mul(a, x)), b)
        return lhs == rhs

    def add(self, p1, p2, group):
        if p1 is None:
            return p2
        if p2 is None:
            return p1
        x1, y1 = p1
        x2, y2 = p2
        if x1 == x2:
            if self.f_is_zero(self.f_add(y1, y2)):
                return None
            three = self.f_add(self.f_add(self.f_mul(x1, x1), self.f_mul(x1, x1)), self.f_mul(x1, x1))
            lam = self.f_mul(self.f_add(three, self.coeff_a(group)), self.f_inv(self.f_add(y1, y1)))
        else:
            lam = self.f_mul(self.f_sub(y2, y1), self.f_inv(self.f_sub(x2, x1)))
        x3 = self.f_sub(self.f_sub(self.f_mul(lam, lam), x1), x2)
        y3 = self.f_sub(self.f_mul(lam, self.f_sub(x1, x3)), y1)
        return (x3, y3)

    def neg(self, p):
        return None if p is None else (p[0], self.f_neg(p[1]))

    def mul(self, k, p, group):
        acc = None
        for bit in bin(k)[2:] if k else "":
            acc = self.add(acc, acc, group)
            if bit == "1":
                acc = self.add(acc, p, group)
        return acc

    def gen(self, group):
        if group == 1:
            return self.g1
        return (tuple(self.g2[0]), tuple(self.g2[1]))

    def msm(self, scalars, pts, group):
        acc = None
        for k, p in zip(scalars, pts):
            if p is not None and k % self.r:
                acc = self.add(acc, self.mul(k % self.r, p, group), group)
        return acc

    # ---- wire format ----------------------------------------------------------------------------
    def fq_to_words(self, x):
        v = x * R % self.q
        return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(12)]

    def fr_to_words(self, x):
        v = x * R % self.r
        return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(12)]

    @staticmethod
    def words_to_int(w):
        return sum(int(v) << (64 * i) for i, v in enumerate(w))

    def fq_from_words(self, w):
        return self.words_to_int(w) * pow(R, -1, self.q) % self.q

    def fr_from_words(self, w):
        return self.words_to_int(w) * pow(R, -1, self.r) % self.r

    def coord_to_words(self, c):
        if isinstance(c, int):
            return self.fq_to_words(c)
        out = []
        for v in c:
            out += self.fq_to_words(v)
        return out

    def coord_from_words(self, w, deg):
        if deg == 1:
            return self.fq_from_words(w[:12])
        return tuple(self.fq_from_words(w[12 * k:12 * k + 12]) for k in range(deg))

    def affine_to_words(self, pt, group):
        deg = 1 if group == 1 else self.deg
        if pt is None:
            return [0] * (24 * deg)
        return self.coord_to_words(pt[0]) + self.coord_to_words(pt[1])

    def affine_from_words(self, w, group):
        deg = 1 if group == 1 else self.deg
        x = self.coord_from_words(w[:12 * deg], deg)
        y = self.coord_from_words(w[12 * deg:24 * deg], deg)
        if self.f_is_zero(y):
            return None
        return (x, y)

    def projective_from_words(self, w, group):
        """X|Y|Z wire words -> affine point (or None)."""
        deg = 1 if group == 1 else self.deg
        X = self.coord_from_words(w[0:12 * deg], deg)
        Y = self.coord_from_words(w[12 * deg:24 * deg], deg)
        Z = self.coord_from_words(w[24 * deg:36 * deg], deg)
        if self.f_is_zero(Z):
            return None
        zi = self.f_inv(Z)
        return (self.f_mul(X, zi), self.f_mul(Y, zi))


def splitmix64(seed):
    state = seed & 0xFFFFFFFFFFFFFFFF

    def nxt():
        nonlocal state
        state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)
    return nxt


def rand_below(rng, bound):
    bits = bound.bit_length()
    while True:
        v = 0
        for i in range((bits + 63) // 64):
            v |= rng() << (64 * i)
        v &= (1 << bits) - 1
        if v < bound:
            return v


def words_array(list_of_wordlists):
    return np.array(list_of_wordlists, dtype=np.uint64).reshape(len(list_of_wordlists), -1)
