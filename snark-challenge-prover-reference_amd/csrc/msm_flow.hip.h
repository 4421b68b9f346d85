// One projective addition spread over the lanes of a GROUP (8 lanes for the base fields, 16 for Fq2 / Fq3), for the launches of an MSM
// that are one addition deep however few additions they hold: the narrow halving steps of the bucket reduction and the levels of the
// edge merge (round 4).  Included at the end of msm_kernels.hip.h.
//
// Why: such a launch costs the LATENCY of one addition, and through the point-operation VM (curve753.hip.h) that is 14 dependent
// products in one lane (78 us on a base field, ~200 us for the three-lane Fq3 whose product is a fused triple).  The 12 products + 2
// squarings of operator+ (depends/libff/libff/algebra/curves/mnt753/mnt4753/mnt4753_g1.cpp:134-207, add-1998-cmo-2; mnt6753_g2.cpp has
// the same body over Fq3) are only FOUR dependency levels deep:
//     level 0   t0 = X1 Z2   t1 = Y1 Z2   t2 = Z1 Z2   t3 = X2 Z1   t4 = Y2 Z1                u = t4 - t1,  v = t3 - t0
//     level 1   uu = u^2     vv = v^2
//     level 2   vvv = v vv   R = vv t0    uuZ = uu t2                                         A = uuZ - vvv - 2R
//     level 3   X3 = v A     Yt = u (R - A)   W = vvv t1   Z3 = vvv t2                        Y3 = Yt - W
// so a group runs them as four rounds of ONE product per lane.  A lane is (product m, component k) of its round: it builds all DEG
// components of both operands from values in LDS (each a short signed sum of earlier values: the table flow_task below), and forms
// component k of the product in Fq^DEG with the fused multiplier of the lane-split fields (fp_mul / fp_mul2 / fp_mul3: one Montgomery
// reduction per component, no Karatsuba bookkeeping).  Values never leave LDS between rounds (FV_COUNT values x DEG components x 28
// words per group); a 64-thread block is one wave, so the barrier between rounds costs nothing.
//
// Same group element AND the same projective coordinates (mod p) as pt_vm<PC_ADD>: identities pass the other operand through, equal
// points -- u = v = 0, the formula would give (0, 0, 0) -- fall back to the VM (an outlined call on LANES lanes of the group: its
// addition turns into its doubling).  P + (-P) needs nothing: v = 0 makes X3 = Z3 = 0, an identity.
#pragma once

namespace mnt753 {

enum : uint32_t {
  FV_X1 = 0, FV_Y1, FV_Z1, FV_X2, FV_Y2, FV_Z2,   // the operands, in the memory layout of two projective points
  FV_T0, FV_T1, FV_T2, FV_T3, FV_T4, FV_UU, FV_VV, FV_VVV, FV_R, FV_UUZ, FV_X3, FV_YT, FV_W, FV_Z3,
  FV_U, FV_V,                                     // operands of level 1, kept for level 3
  FV_COUNT
};
// A task: r[dst] = (sum of the a terms) * (sum of the b terms).  Terms are bytes, low byte first, 0 ends the list: value + 1, bit 7 set
// if the value is subtracted (the first term is always added).  b == FLOW_SAME: a squaring.  save != 0: the a operand is kept as value save - 1.
struct FlowTask {
  uint64_t a, b;
  uint32_t dst, save;
};
constexpr uint64_t FLOW_SAME = ~0ull;
constexpr uint64_t ft_add(uint32_t v) { return (uint64_t)(v + 1u); }
constexpr uint64_t ft_sub(uint32_t v) { return (uint64_t)((v + 1u) | 0x80u); }
constexpr uint64_t ft_terms(uint64_t t0, uint64_t t1 = 0, uint64_t t2 = 0, uint64_t t3 = 0, uint64_t t4 = 0) {
  return t0 | (t1 << 8) | (t2 << 16) | (t3 << 24) | (t4 << 32);
}
constexpr int FLOW_LEVELS = 4;
__host__ __device__ constexpr uint32_t flow_tasks_of(int level) { return level == 0 ? 5u : (level == 1 ? 2u : (level == 2 ? 3u : 4u)); }
__device__ __forceinline__ FlowTask flow_task(int level, uint32_t m) {
  constexpr uint64_t U = ft_terms(ft_add(FV_T4), ft_sub(FV_T1)), V = ft_terms(ft_add(FV_T3), ft_sub(FV_T0));
  switch (level * 8 + (int)m) {
    case 0: return {ft_terms(ft_add(FV_X1)), ft_terms(ft_add(FV_Z2)), FV_T0, 0u};
    case 1: return {ft_terms(ft_add(FV_Y1)), ft_terms(ft_add(FV_Z2)), FV_T1, 0u};
    case 2: return {ft_terms(ft_add(FV_Z1)), ft_terms(ft_add(FV_Z2)), FV_T2, 0u};
    case 3: return {ft_terms(ft_add(FV_X2)), ft_terms(ft_add(FV_Z1)), FV_T3, 0u};
    case 4: return {ft_terms(ft_add(FV_Y2)), ft_terms(ft_add(FV_Z1)), FV_T4, 0u};
    case 8: return {U, FLOW_SAME, FV_UU, FV_U + 1u};
    case 9: return {V, FLOW_SAME, FV_VV, FV_V + 1u};
    case 16: return {ft_terms(ft_add(FV_V)), ft_terms(ft_add(FV_VV)), FV_VVV, 0u};
    case 17: return {ft_terms(ft_add(FV_VV)), ft_terms(ft_add(FV_T0)), FV_R, 0u};
    case 18: return {ft_terms(ft_add(FV_UU)), ft_terms(ft_add(FV_T2)), FV_UUZ, 0u};
    // A = uuZ - vvv - 2R;   R - A = 3R + vvv - uuZ
    case 24: return {ft_terms(ft_add(FV_V)), ft_terms(ft_add(FV_UUZ), ft_sub(FV_VVV), ft_sub(FV_R), ft_sub(FV_R)), FV_X3, 0u};
    case 25: return {ft_terms(ft_add(FV_U)), ft_terms(ft_add(FV_R), ft_add(FV_R), ft_add(FV_R), ft_add(FV_VVV), ft_sub(FV_UUZ)), FV_YT, 0u};
    case 26: return {ft_terms(ft_add(FV_VVV)), ft_terms(ft_add(FV_T1)), FV_W, 0u};
    default: return {ft_terms(ft_add(FV_VVV)), ft_terms(ft_add(FV_T2)), FV_Z3, 0u};
  }
}

// Fq3 in the Karatsuba form (flow_add_k3 below): 32 lanes per addition, one BASE-field product per lane and round (the fused
// three-product form on 16 lanes, which Fq2 uses with its two products, took 55 us per Fq3 addition against ~35)
enum : uint32_t { FV_A = FV_COUNT, FV_RA, FV_Y3, FVK_COUNT };   // values the Karatsuba form materialises between its rounds
template <class C>
struct Flow {
  using F = typename C::F;                            // the ONE-lane field class of the group (FieldFp / FieldFp2 / FieldFp3)
  static_assert(F::LANES == 1, "the flow addition is instantiated with the one-lane configuration; its fallback picks the lane-split one");
  static constexpr int D = F::DEG, M = F::MOD;
  static constexpr bool K3 = D == 3;
  static constexpr uint32_t G = D == 1 ? 8u : (K3 ? 32u : 16u);   // lanes per addition: 5 D products in the widest round (5 x 6 pieces: K3)
  static constexpr uint32_t PER_WAVE = 64u / G;
  static constexpr uint32_t VAL_WORDS = (uint32_t)D * FPS_WORDS;
  static constexpr uint32_t PIECE_WORDS = K3 ? 5u * 6u * FPS_WORDS : 0u;   // K3: the six partial products of up to five products
  static constexpr uint32_t FLAG_WORDS = 32u;         // [level * 8 + task]: bit 0 the a operand is zero, bit 1 the b operand (K3: see there)
  static constexpr uint32_t GROUP_WORDS = (K3 ? FVK_COUNT : FV_COUNT) * VAL_WORDS + PIECE_WORDS + FLAG_WORDS;
  static constexpr uint32_t WAVE_WORDS = PER_WAVE * GROUP_WORDS;
};

template <int M, int D>
__device__ __forceinline__ void flow_operand(Fp<M> (&X)[D], const uint32_t* grp, uint64_t terms) {
  {
    const uint32_t v = ((uint32_t)terms & 0x7fu) - 1u;
#pragma unroll
    for (int c = 0; c < D; ++c) fp_load(X[c], grp + (v * (uint32_t)D + (uint32_t)c) * FPS_WORDS);
  }
  terms >>= 8;
#pragma nounroll
  while ((uint32_t)terms & 0xffu) {
    const uint32_t v = ((uint32_t)terms & 0x7fu) - 1u;
    const bool neg = ((uint32_t)terms & 0x80u) != 0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      Fp<M> y;
      fp_load(y, grp + (v * (uint32_t)D + (uint32_t)c) * FPS_WORDS);
      if (neg) fp_sub(X[c], X[c], y); else fp_add(X[c], X[c], y);
    }
    terms >>= 8;
  }
}
template <int M, int D>
__device__ __forceinline__ void flow_pick(Fp<M>& r, const Fp<M> (&X)[D], uint32_t k) {
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    uint32_t v = X[0].l[i];
    if constexpr (D > 1) v = k == 1u ? X[1].l[i] : v;
    if constexpr (D > 2) v = k == 2u ? X[2].l[i] : v;
    r.l[i] = v;
  }
}
// component k of X * Y in Fq^D (fp2.tcc:79-90, fp3.tcc:83-96 written out per component, as FieldFp2S / FieldFp3S::mul do)
template <class F>
__device__ __forceinline__ void flow_mul(Fp<F::MOD>& r, const Fp<F::MOD> (&X)[F::DEG], const Fp<F::MOD> (&Y)[F::DEG], uint32_t k) {
  constexpr int M = F::MOD;
  if constexpr (F::DEG == 1) {
    fp_mul(r, X[0], Y[0]);
  } else if constexpr (F::DEG == 2) {
    // c0 = x0 y0 + NR x1 y1      c1 = x1 y0 + x0 y1
    Fp<M> nx, a1, a2;
    fp_mul_small(nx, X[1], F::NONRES);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      a1.l[i] = k == 0u ? X[0].l[i] : X[1].l[i];
      a2.l[i] = k == 0u ? nx.l[i] : X[0].l[i];
    }
    fp_mul2(r, a1, Y[0], a2, Y[1]);
  } else {
    // c0 = x0 y0 + NR x1 y2 + NR x2 y1      c1 = x1 y0 + NR x2 y2 + x0 y1      c2 = x2 y0 + x0 y2 + x1 y1
    Fp<M> n1, n2, a1, a2, a3;
    fp_mul_small(n1, X[1], F::NONRES);
    fp_mul_small(n2, X[2], F::NONRES);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      a1.l[i] = k == 0u ? X[0].l[i] : (k == 1u ? X[1].l[i] : X[2].l[i]);
      a2.l[i] = k == 0u ? n1.l[i] : (k == 1u ? n2.l[i] : X[0].l[i]);
      a3.l[i] = k == 0u ? n2.l[i] : (k == 1u ? X[0].l[i] : X[1].l[i]);
    }
    fp_mul3(r, a1, Y[0], a2, Y[2], a3, Y[1]);
  }
}

// ---- Fq3, Karatsuba form ----------------------------------------------------------------------------------------------------------
// The fused form above gives a lane one component of an Fq3 product: three base-field products and a reduction (fp_mul3, ~2.3 x the
// time of fp_mul), four of them deep.  Karatsuba (fp3.tcc:83-96) splits a product into SIX base-field products that need nothing of
// each other: a round is then
//     P  lane (product m, piece j): v_j = A_j B_j,  A_0..5 = a0, a1, a2, a1 + a2, a0 + a1, a0 + a2  (B likewise)       one fp_mul
//     R  lane (product m, component k): c0 = v0 + NR (v3 - v1 - v2),  c1 = v4 - v0 - v1 + NR v2,  c2 = v5 - v0 - v2 + v1   (branch-free:
//        base = X - Y - Z with (X, Y, Z) picked by k, then S + [NR] T)
//     D  the values the next round multiplies, component-wise: u, v after round 0;  A = uuZ - vvv - 2R and R - A after round 2;
//        Y3 = Yt - W after round 3
// with a barrier after each (one wave per block: free).  Every operand of a product is ONE stored value, so P prepares its operands
// with at most one addition per side.  Same values mod p as the fused form and as the VM.
struct FlowK3Prod { uint32_t a, b, dst; };
__device__ __forceinline__ FlowK3Prod flow_k3_product(int level, uint32_t m) {
  switch (level * 8 + (int)m) {
    case 0: return {FV_X1, FV_Z2, FV_T0};
    case 1: return {FV_Y1, FV_Z2, FV_T1};
    case 2: return {FV_Z1, FV_Z2, FV_T2};
    case 3: return {FV_X2, FV_Z1, FV_T3};
    case 4: return {FV_Y2, FV_Z1, FV_T4};
    case 8: return {FV_U, FV_U, FV_UU};
    case 9: return {FV_V, FV_V, FV_VV};
    case 16: return {FV_V, FV_VV, FV_VVV};
    case 17: return {FV_VV, FV_T0, FV_R};
    case 18: return {FV_UU, FV_T2, FV_UUZ};
    case 24: return {FV_V, FV_A, FV_X3};
    case 25: return {FV_U, FV_RA, FV_YT};
    case 26: return {FV_VVV, FV_T1, FV_W};
    default: return {FV_VVV, FV_T2, FV_Z3};
  }
}
// piece j of the value at `val` (three components of 28 words): a0, a1, a2, a1 + a2, a0 + a1, a0 + a2
template <int M>
__device__ __forceinline__ void flow_k3_piece(Fp<M>& r, const uint32_t* val, uint32_t j) {
  const uint32_t c_first = j < 3u ? j : (j == 3u ? 1u : 0u), c_second = j == 4u ? 1u : 2u;
  fp_load(r, val + c_first * FPS_WORDS);
  if (j >= 3u) {
    Fp<M> y;
    fp_load(y, val + c_second * FPS_WORDS);
    fp_add(r, r, y);
  }
}
template <class C>
__device__ __forceinline__ void flow_add_k3(uint32_t* grp, uint32_t l, const uint32_t* srcS, const uint32_t* srcT, bool emptyS, bool emptyT,
                                            uint32_t* dst, bool live) {
  using FL = Flow<C>;
  using F = typename C::F;
  constexpr int M = FL::M;
  constexpr uint32_t VW = FL::VAL_WORDS;
  uint32_t* pieces = grp + FVK_COUNT * VW;
  uint32_t* flags = pieces + FL::PIECE_WORDS;        // [0..2] Z1 component zero, [3..5] Z2, [6..8] u, [9..11] v
  if (l < 18u) {
    const bool second = l >= 9u;
    const uint32_t e = second ? l - 9u : l;
    Fp<M> x;
    if (!live || (second ? emptyT : emptyS)) {
      if (e == 3u) fp_one(x); else fp_zero(x);          // (0 : 1 : 0)
    } else {
      fp_load(x, (second ? srcT : srcS) + e * FPS_WORDS);
    }
    fp_store(grp + l * FPS_WORDS, x);
    if (e >= 6u) flags[(second ? 3u : 0u) + e - 6u] = fp_is_zero(x) ? 1u : 0u;
  }
  __syncthreads();
#pragma nounroll
  for (int level = 0; level < FLOW_LEVELS; ++level) {
    const uint32_t nprod = flow_tasks_of(level);
    {                                                  // P
      const uint32_t m = l / 6u, j = l - m * 6u;
      if (m < nprod) {
        const FlowK3Prod pr = flow_k3_product(level, m);
        Fp<M> a, b, r;
        flow_k3_piece<M>(a, grp + pr.a * VW, j);
        if (pr.b == pr.a) b = a; else flow_k3_piece<M>(b, grp + pr.b * VW, j);
        fp_mul(r, a, b);
        fp_store(pieces + (m * 6u + j) * FPS_WORDS, r);
      }
    }
    __syncthreads();
    {                                                  // R
      const uint32_t m = l / 3u, k = l - m * 3u;
      if (m < nprod) {
        const FlowK3Prod pr = flow_k3_product(level, m);
        const uint32_t* v = pieces + m * 6u * FPS_WORDS;
        // base = X - Y - Z:  k = 0: v3 - v1 - v2;  k = 1: v4 - v0 - v1;  k = 2: v5 - v0 - v2
        const uint32_t ix = 3u + k, iy = k == 0u ? 1u : 0u, iz = k == 1u ? 1u : 2u, iw = k == 0u ? 0u : (k == 1u ? 2u : 1u);
        Fp<M> base, y, w, t, n, sum;
        fp_load(base, v + ix * FPS_WORDS);
        fp_load(y, v + iy * FPS_WORDS);
        fp_sub(base, base, y);
        fp_load(y, v + iz * FPS_WORDS);
        fp_sub(base, base, y);
        fp_load(w, v + iw * FPS_WORDS);
        // c_k = S + [NR] T:  k = 0: v0 + NR base;  k = 1: base + NR v2;  k = 2: base + v1
#pragma unroll
        for (int i = 0; i < NL; ++i) t.l[i] = k == 0u ? base.l[i] : w.l[i];
        fp_mul_small(n, t, F::NONRES);
#pragma unroll
        for (int i = 0; i < NL; ++i) { n.l[i] = k == 2u ? t.l[i] : n.l[i]; t.l[i] = k == 0u ? w.l[i] : base.l[i]; }
        fp_add(sum, t, n);
        fp_store(grp + pr.dst * VW + k * FPS_WORDS, sum);
      }
    }
    __syncthreads();
    if (level != 1) {                                  // D (nothing to derive after the squarings)
      const uint32_t q = l / 3u, k = l - q * 3u;
      if (level == 0 && q < 2u) {                      // u = t4 - t1, v = t3 - t0
        Fp<M> x, y;
        fp_load(x, grp + (q == 0u ? FV_T4 : FV_T3) * VW + k * FPS_WORDS);
        fp_load(y, grp + (q == 0u ? FV_T1 : FV_T0) * VW + k * FPS_WORDS);
        fp_sub(x, x, y);
        fp_store(grp + (q == 0u ? FV_U : FV_V) * VW + k * FPS_WORDS, x);
        flags[6u + q * 3u + k] = fp_is_zero(x) ? 1u : 0u;
      } else if (level == 2 && q == 0u) {              // A = uuZ - vvv - 2R,  R - A
        Fp<M> a, y, rr;
        fp_load(a, grp + FV_UUZ * VW + k * FPS_WORDS);
        fp_load(y, grp + FV_VVV * VW + k * FPS_WORDS);
        fp_sub(a, a, y);
        fp_load(rr, grp + FV_R * VW + k * FPS_WORDS);
        fp_sub(a, a, rr);
        fp_sub(a, a, rr);
        fp_store(grp + FV_A * VW + k * FPS_WORDS, a);
        fp_sub(rr, rr, a);
        fp_store(grp + FV_RA * VW + k * FPS_WORDS, rr);
      } else if (level == 3 && q == 0u) {              // Y3 = Yt - W
        Fp<M> x, y;
        fp_load(x, grp + FV_YT * VW + k * FPS_WORDS);
        fp_load(y, grp + FV_W * VW + k * FPS_WORDS);
        fp_sub(x, x, y);
        fp_store(grp + FV_Y3 * VW + k * FPS_WORDS, x);
      }
      __syncthreads();
    }
  }
  const bool zS = (flags[0] & flags[1] & flags[2]) != 0, zT = (flags[3] & flags[4] & flags[5]) != 0;
  const bool same = (flags[6] & flags[7] & flags[8] & flags[9] & flags[10] & flags[11]) != 0;
  if (!live) return;
  if (zS || zT) {
    if (l < 9u) {
      Fp<M> x;
      fp_load(x, grp + ((zT ? 0u : 9u) + l) * FPS_WORDS);
      fp_store(dst + l * FPS_WORDS, x);
    }
  } else if (same) {
    using V = typename SplitOf<C>::type;
    const uint32_t lane = threadIdx.x & 63u, base = lane - l;
    const uint32_t first = ((base + 2u) / 3u) * 3u;
    if (lane >= first && lane < first + 3u) {
      Proj<V> P, Q;
      proj_load<V>(P, grp);
      proj_load<V>(Q, grp + 3u * VW);
      pt_vm_add_outlined<V>(P, Q, PC_ADD);
      proj_store<V>(dst, P);
    }
  } else if (l < 9u) {
    const uint32_t q = l / 3u, k = l - q * 3u;
    Fp<M> x;
    fp_load(x, grp + (q == 0u ? FV_X3 : (q == 1u ? FV_Y3 : FV_Z3)) * VW + k * FPS_WORDS);
    fp_store(dst + l * FPS_WORDS, x);
  }
}

// dst = S + T by the G lanes of one group.  grp: the group's LDS block; l: this thread's lane in the group.  Every thread of the
// (one-wave) block must call it: there are barriers inside.  live == false: a group without an addition (it computes on identities
// and stores nothing).  emptyS / emptyT: the operand is the identity whatever memory holds (a bucket no entry was sorted into).
// dst may be srcS (the operands are read before anything is written).
template <class C>
__device__ __forceinline__ void flow_add(uint32_t* grp, uint32_t l, const uint32_t* srcS, const uint32_t* srcT, bool emptyS, bool emptyT,
                                         uint32_t* dst, bool live) {
  using FL = Flow<C>;
  using F = typename C::F;
  constexpr int D = FL::D, M = FL::M;
  constexpr uint32_t UD = (uint32_t)D;
  if constexpr (FL::K3) {
    flow_add_k3<C>(grp, l, srcS, srcT, emptyS, emptyT, dst, live);
    return;
  }
  uint32_t* flags = grp + FV_COUNT * FL::VAL_WORDS;
  for (uint32_t f = l; f < 6u * UD; f += FL::G) {
    const bool second = f >= 3u * UD;
    const uint32_t e = second ? f - 3u * UD : f;
    Fp<M> x;
    if (!live || (second ? emptyT : emptyS)) {
      if (e == UD) fp_one(x); else fp_zero(x);          // (0 : 1 : 0)
    } else {
      fp_load(x, (second ? srcT : srcS) + e * FPS_WORDS);
    }
    fp_store(grp + f * FPS_WORDS, x);
  }
  __syncthreads();
  const uint32_t m = l / UD, k = l - m * UD;
#pragma nounroll
  for (int level = 0; level < FLOW_LEVELS; ++level) {
    if (m < flow_tasks_of(level)) {
      const FlowTask tk = flow_task(level, m);
      Fp<M> X[D], Y[D], r;
      flow_operand<M, D>(X, grp, tk.a);
      if (tk.b == FLOW_SAME) {
#pragma unroll
        for (int c = 0; c < D; ++c) Y[c] = X[c];
      } else {
        flow_operand<M, D>(Y, grp, tk.b);
      }
      if (k == 0u) {
        bool za = true, zb = true;
#pragma unroll
        for (int c = 0; c < D; ++c) { za = za && fp_is_zero(X[c]); zb = zb && fp_is_zero(Y[c]); }
        flags[level * 8 + (int)m] = (za ? 1u : 0u) | (zb ? 2u : 0u);
      }
      if (tk.save != 0u) {
        Fp<M> xs;
        flow_pick<M, D>(xs, X, k);
        fp_store(grp + ((tk.save - 1u) * UD + k) * FPS_WORDS, xs);
      }
      flow_mul<F>(r, X, Y, k);
      fp_store(grp + (tk.dst * UD + k) * FPS_WORDS, r);
    }
    __syncthreads();
  }
  const uint32_t fz = flags[0 * 8 + 2], fu = flags[1 * 8 + 0], fv = flags[1 * 8 + 1];
  const bool zS = (fz & 1u) != 0, zT = (fz & 2u) != 0, same = (fu & 1u) != 0 && (fv & 1u) != 0;
  if (!live) return;
  if (zS || zT) {
    // S + 0 = S, 0 + T = T (both identities: S)
    for (uint32_t f = l; f < 3u * UD; f += FL::G) {
      Fp<M> x;
      fp_load(x, grp + ((zT ? 0u : 3u * UD) + f) * FPS_WORDS);
      fp_store(dst + f * FPS_WORDS, x);
    }
  } else if (same) {
    // equal points: the VM's addition (it turns into its doubling), on the lanes of the group that form one logical lane of the VM's
    // configuration -- a pair starts on an even lane, a triple on a multiple of three (curve753.hip.h)
    using V = std::conditional_t<std::is_void<typename SplitOf<C>::type>::value, C, typename SplitOf<C>::type>;
    constexpr uint32_t VL = (uint32_t)V::F::LANES;
    const uint32_t lane = threadIdx.x & 63u, base = lane - l;
    const uint32_t first = VL == 3u ? ((base + 2u) / 3u) * 3u : base;
    if (lane >= first && lane < first + VL) {
      Proj<V> P, Q;
      proj_load<V>(P, grp);
      proj_load<V>(Q, grp + 3u * FL::VAL_WORDS);
      pt_vm_add_outlined<V>(P, Q, PC_ADD);
      proj_store<V>(dst, P);
    }
  } else {
    for (uint32_t f = l; f < 3u * UD; f += FL::G) {
      const uint32_t q = f / UD, c = f - q * UD;
      Fp<M> x;
      if (q == 1u) {
        Fp<M> w;
        fp_load(x, grp + (FV_YT * UD + c) * FPS_WORDS);
        fp_load(w, grp + (FV_W * UD + c) * FPS_WORDS);
        fp_sub(x, x, w);
      } else {
        fp_load(x, grp + ((q == 0u ? FV_X3 : FV_Z3) * UD + c) * FPS_WORDS);
      }
      fp_store(dst + f * FPS_WORDS, x);
    }
  }
}

// the halving step of the bucket reduction (k_reduce_step) with one GROUP per addition
template <class C>
__global__ void __launch_bounds__(64) k_reduce_step_flow(const uint32_t* __restrict__ buckets, const uint32_t* __restrict__ offsets,
                                                        uint32_t* __restrict__ A, uint32_t* __restrict__ G, uint32_t n_sets, uint32_t k, uint32_t s) {
  using FL = Flow<C>;
  __shared__ __attribute__((aligned(16))) uint32_t lds[FL::WAVE_WORDS];
  const uint32_t gi = threadIdx.x / FL::G, l = threadIdx.x - gi * FL::G;
  const uint32_t per_set = red_items(k, s), n_items = n_sets * per_set;
  const uint32_t item = blockIdx.x * FL::PER_WAVE + gi;
  const bool live = item < n_items;
  const uint32_t t = live ? item : n_items - 1u;
  constexpr int PW = proj_words<C>();
  const uint32_t set = t / per_set, r0 = t - set * per_set;
  const uint32_t nb = 1u << k, nh = 1u << (k - s - 1);
  const uint32_t* a_src = s == 0 ? buckets + (size_t)set * nb * PW : A + ((size_t)set * nb + red_off_a(k, s)) * PW;
  uint32_t i0, i1;
  const uint32_t* src;
  uint32_t* dst;
  if (r0 < nh) {                                   // halving: A_{s+1}[j] = A_s[2j] + A_s[2j+1]
    i0 = 2u * r0; i1 = i0 + 1u; src = a_src;
    dst = A + ((size_t)set * nb + red_off_a(k, s + 1) + r0) * PW;
  } else {
    const uint32_t sh = k - s - 2, r = r0 - nh, lt = r >> sh, j = r & ((1u << sh) - 1u);
    uint32_t* g = G + ((size_t)set * nb + red_off_g(k, lt)) * PW;
    const uint32_t half = 1u << (k - lt - 2);
    if (lt == s) {                                 // first step of tree s: odd elements of A_s
      i0 = 4u * j + 1u; i1 = i0 + 2u; src = a_src;
      dst = g + (size_t)j * PW;
    } else {                                       // tree lt < s: one more level, ping-pong
      i0 = 2u * j; i1 = i0 + 1u; src = g + (size_t)(((s - lt - 1u) & 1u) * half) * PW;
      dst = g + (size_t)(((s - lt) & 1u) * half + j) * PW;
    }
  }
  // buckets no entry was sorted into hold stale data: they count as the identity (only A_0 = the bucket array has them)
  const bool from_buckets = s == 0 && src == a_src;
  const bool e0 = from_buckets && offsets[(size_t)set * nb + i0 + 1] == offsets[(size_t)set * nb + i0];
  const bool e1 = from_buckets && offsets[(size_t)set * nb + i1 + 1] == offsets[(size_t)set * nb + i1];
  flow_add<C>(lds + gi * FL::GROUP_WORDS, l, src + (size_t)i0 * PW, src + (size_t)i1 * PW, e0, e1, dst, live);
}

// The levels of the edge merge tree (k_edge_tree_level) on lane groups.  A group per SLOT would leave nearly every wave with one live
// group among idle ones (the nodes of a level are a fifth of the slots or far fewer): measured on the MNT6753 G2 2^15 MSM, 533 us for a
// level the VM does in 305.  So the nodes of a level are a LIST: k_edge_nodes writes the list of the first level that runs on groups
// (one thread per slot, no arithmetic), every level appends the nodes of the next one as its own finish -- node i with i even becomes
// node i / 2, same destination, the partner 2 S pieces on -- and a launch is sized for the most nodes its level can have; the blocks
// past the end of the list leave after one scalar load.  Entry: (destination slot, partner slot, bucket, node index).
template <class C>
__global__ void __launch_bounds__(256) k_edge_nodes(const uint32_t* __restrict__ edge_bucket, const uint32_t* __restrict__ offsets, uint32_t n_buckets, uint32_t T_arg,
                                                   uint32_t n_lanes, uint32_t blocked, uint32_t stride, const uint32_t* __restrict__ flags, uint32_t level,
                                                   uint4* __restrict__ list, uint32_t* __restrict__ count) {
  if (level > 0 && flags[level - 1] == 0) return;   // the levels before this one (slot-driven) found no bucket that needs it
  const uint32_t sidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (sidx >= 2u * n_lanes) return;
  const uint32_t b = edge_bucket[sidx];
  if (b == EDGE_NONE) return;
  const uint32_t T = acc_entries_per_lane(T_arg, n_lanes, offsets[n_buckets], blocked != 0);
  const uint32_t o0 = offsets[b], o1 = offsets[b + 1];
  const uint32_t t = sidx >> 1, t_lo = o0 / T, t_hi = (o1 - 1u) / T;
  if (t_hi == t_lo || edge_piece_slot(t, t_lo, o0, T) != sidx) return;
  const uint32_t i = t - t_lo, kk = t_hi - t_lo + 1u;
  const uint64_t first = (uint64_t)i * stride * EDGE_TREE_K;
  if (first + stride >= kk) return;
  const uint32_t pos = atomicAdd(count, 1u);
  list[pos] = make_uint4(edge_piece_slot(t_lo + (uint32_t)first, t_lo, o0, T), edge_piece_slot(t_lo + (uint32_t)(first + stride), t_lo, o0, T), b, i);
}
// node i of a level (i even) becomes node i / 2 of the next one: same destination, the partner 2 S pieces on -- if the bucket reaches
// that far.  Called by ONE thread per node, after the node's sum is stored.
__device__ __forceinline__ void edge_emit_next(const uint4 node, const uint32_t* __restrict__ offsets, uint32_t n_buckets, uint32_t T_arg, uint32_t n_lanes,
                                               uint32_t blocked, uint32_t stride, uint4* __restrict__ list_out, uint32_t* __restrict__ count_out) {
  if ((node.w & 1u) != 0u) return;
  const uint32_t T = acc_entries_per_lane(T_arg, n_lanes, offsets[n_buckets], blocked != 0);
  const uint32_t o0 = offsets[node.z], o1 = offsets[node.z + 1];
  const uint32_t t_lo = o0 / T, kk = (o1 - 1u) / T - t_lo + 1u;
  const uint64_t first = (uint64_t)node.w * stride * 2u, partner = first + 2ull * stride;
  if (partner >= kk) return;
  const uint32_t pos = atomicAdd(count_out, 1u);
  list_out[pos] = make_uint4(node.x, edge_piece_slot(t_lo + (uint32_t)partner, t_lo, o0, T), node.z, node.w >> 1);
}
template <class C>
__global__ void __launch_bounds__(64) k_edge_tree_level_list(uint32_t* __restrict__ edges, const uint32_t* __restrict__ offsets, uint32_t n_buckets, uint32_t T_arg,
                                                            uint32_t n_lanes, uint32_t blocked, uint32_t stride, const uint4* __restrict__ list_in,
                                                            const uint32_t* __restrict__ count_in, uint4* __restrict__ list_out, uint32_t* __restrict__ count_out) {
  using FL = Flow<C>;
  __shared__ __attribute__((aligned(16))) uint32_t lds[FL::WAVE_WORDS];
  static_assert(EDGE_TREE_K == 2, "one partner per node");
  const uint32_t n = *count_in;
  if (blockIdx.x * FL::PER_WAVE >= n) return;       // past the end of the list (an empty list: a level nothing needs)
  const uint32_t gi = threadIdx.x / FL::G, l = threadIdx.x - gi * FL::G;
  const uint32_t item = blockIdx.x * FL::PER_WAVE + gi;
  const bool act = item < n;
  const uint4 node = act ? list_in[item] : make_uint4(0u, 0u, 0u, 1u);
  constexpr int PW = proj_words<C>();
  flow_add<C>(lds + gi * FL::GROUP_WORDS, l, edges + (size_t)node.x * PW, edges + (size_t)node.y * PW, false, false, edges + (size_t)node.x * PW, act);
  if (!act || l != 0u) return;
  edge_emit_next(node, offsets, n_buckets, T_arg, n_lanes, blocked, stride, list_out, count_out);
}
// The same level with one VM addition per (logical) lane, for the levels that hold too many nodes for the lane groups to win: dense
// waves over the list instead of one node in two slots (the slot-driven k_edge_tree_level: 2048 half-empty waves for 65536 lanes).
template <class C>
__global__ void __launch_bounds__(256, vm_waves<C>()) k_edge_tree_level_vmlist(uint32_t* __restrict__ edges, const uint32_t* __restrict__ offsets, uint32_t n_buckets,
                                                                  uint32_t T_arg, uint32_t n_lanes, uint32_t blocked, uint32_t stride,
                                                                  const uint4* __restrict__ list_in, const uint32_t* __restrict__ count_in,
                                                                  uint4* __restrict__ list_out, uint32_t* __restrict__ count_out) {
  const uint32_t j = logical_lane<typename C::F>();     // 0xffffffff for the idle 64th lane of three-lane fields
  if (j >= *count_in) return;
  const uint4 node = list_in[j];
  Proj<C> acc, Q;
  proj_load<C>(acc, edges + (size_t)node.x * proj_words<C>());
  proj_load<C>(Q, edges + (size_t)node.y * proj_words<C>());
  const int pc = add_pc<C>(acc, Q);
  pt_vm_add_outlined<C>(acc, Q, pc);
  proj_store<C>(edges + (size_t)node.x * proj_words<C>(), acc);
  if (lane_comp<typename C::F>() != 0u) return;
  edge_emit_next(node, offsets, n_buckets, T_arg, n_lanes, blocked, stride, list_out, count_out);
}

// a list of independent additions out[i] = p[i] + q[i] (points in the device layout): the test hook's view of flow_add
template <class C>
__global__ void __launch_bounds__(64) k_flow_add_list(const uint32_t* __restrict__ p, const uint32_t* __restrict__ q, uint32_t* __restrict__ out, uint32_t n) {
  using FL = Flow<C>;
  __shared__ __attribute__((aligned(16))) uint32_t lds[FL::WAVE_WORDS];
  const uint32_t gi = threadIdx.x / FL::G, l = threadIdx.x - gi * FL::G;
  const uint32_t item = blockIdx.x * FL::PER_WAVE + gi;
  const bool live = item < n;
  const uint32_t i = live ? item : n - 1u;
  constexpr int PW = proj_words<C>();
  flow_add<C>(lds + gi * FL::GROUP_WORDS, l, p + (size_t)i * PW, q + (size_t)i * PW, false, false, out + (size_t)i * PW, live);
}

}  // namespace mnt753
