// EXPERIMENT (round 5, NOT part of the library): the batched-affine pairing level at TWO waves per SIMD, for the levels whose operands
// are the previous level's own blocked planes (everything but the first level) of a base field.  `git apply tools/experiments/pair2w/wiring.patch`
// and a copy of this file into csrc/ put it behind MNT753_EXP_PAIR2W (1: k_pair_level2w, 2: k_pair_level1w at the end of the file, 3: k_pair_level2w<.., PAIRED = false>).
// Measured on MI355X, alternating with the shipped kernel on one box (profiles/r05/levels_two_waves_per_simd_pairs_on_one_simd.txt,
// levels_no_image_register_prefetch.txt, sq_levels_*.txt), MNT4753 G1 2^20, per MSM:
//     levels 2 + 3:        shipped 7.29-7.33 ms   two waves 7.20-7.25 ms (-1 %)   one wave, register prefetch 8.22 ms (+12 %)
//     last irregular level: shipped 0.89-0.90 ms   two waves 0.92-0.93 ms (+3 %)   one wave, register prefetch 0.97-0.99 ms
//     whole MSM:            23.94-24.23 ms         23.97-24.17 ms
//     two INDEPENDENT waves per SIMD (PAIRED = false, MNT753_EXP_PAIR2W=3: a batch and an inversion per wave, twice the lanes):
//                           levels 2 + 3 7.29-7.33 ms against 7.19-7.25 shipped on that box (+1 %), irregular levels +8 % / +24 %; the VALU is
//                           saturated (0.53 active per wave-cycle, two waves) and the busy cycles fall 3 % while the time rises 1 %
//                           (profiles/r05/levels_two_independent_waves_per_simd.txt, sq_levels_two_independent_waves.txt)
//     one wave, register prefetch, REPAIRED (operands raw until their slot, unconditional loads: the first form waited for its loads
//     where it issued them -- a flag taken out of a loaded quad, a load under a condition, each ends in `s_waitcnt` on the spot):
//                           levels 2 + 3 7.57-7.61 ms against 7.37-7.42 shipped on that box (+2.7 %), first irregular level 1.41-1.42
//                           against 1.52-1.53 (-7 %), last 0.89 against 0.91 (-2 %); 8.6 % fewer VALU instructions, VALU active 0.74,
//                           waits 0.15.  What is left of the waits is the compiler's: with loads AND stores outstanding on gfx9's one
//                           vmcnt it waits for everything (completion order between the two kinds is not defined), so the fourteen
//                           stores of a slot are sat out at the top of the next whatever their place in the source (stores moved to
//                           the top of the next slot, the form in this file: +6 % / -7 % / 0).  The shipped kernel's LDS-DMA image
//                           with hand-placed waits is the answer to exactly that.
//                           (profiles/r05/levels_no_image_register_prefetch_repaired.txt, ..._stores_at_top.txt)
// 10 % fewer VALU instructions and 0.93 of a SIMD's issue slots while its two waves are resident, and no gain: the SIMDs run at a lower
// clock under it (the power bound of DESIGN.md 4.3), and the pair's rendezvous leaves SIMDs idle at the ends.  Not adopted.
//
// k_pair_level runs one wave per SIMD: its step loop keeps 376 registers per lane and stages every operand of a slot through a 36 KB
// per-wave LDS image that is filled one slot ahead.  At one wave per SIMD the multiplier issues at 85 % of what two waves reach
// (profiles/r03/mul_variants_mi355x.txt: 18.7 against 22.5 G products/s) and every wait of the wave is idle time of its SIMD.  This form
// gives the second wave what it needs:
//   * at most 256 registers: the five products of a slot written out in the order of their data dependences, every operand loaded
//     where it is first used (pre, x1, x2 at the top of a slot; y1, y2 behind the second product) and dropped where it dies -- six
//     elements live at the widest point instead of the step loop's nine;
//   * no LDS image: a level's inputs are its predecessor's blocked planes, so a plain global_load_dwordx4 per lane is a contiguous KiB
//     per wave instruction; the latency it exposes is what the partner wave on the SIMD is for;
//   * one inversion per PAIR of waves: twice the waves would be twice the inversions (one per lane and level, 94 product-equivalents
//     whatever the batch length).  A batch (the "lane" of k_pair_level: slots t, t + NLe, t + 2 NLe ...) belongs to lane L of BOTH waves
//     of a pair; wave h takes its iterations with it % 2 == h and keeps its own chain of prefix products; behind the forward sweeps the
//     two running products meet in LDS, wave 0 inverts their product, and each wave leaves with the inverse of ITS chain:
//     1 / run_h = (1 / (run_0 run_1)) run_(1-h).  Two products and two workgroup barriers per level.
// Same slots, same prefix-product layout, same planes in and out, same side paths (doubling, cancellation, odd leftover, empty slot),
// same results as k_pair_level<C, false, last, IRR>.
#pragma once

namespace mnt753 {

// PAIRED = false: the two waves of a SIMD are independent -- a batch and an inversion each (twice the lanes of the one-wave kernel,
// batches half as long): while one wave runs the 32-bit division steps of its inversion the other multiplies.
template <class C, bool last, bool IRR, bool PAIRED = true>
__global__ void __launch_bounds__(512, 1) k_pair_level2w(const uint4* __restrict__ src_planes, size_t src_stride, const uint32_t* __restrict__ offsG,
                                                        uint32_t n_buckets, uint32_t shift, uint32_t* __restrict__ out_sorted, uint4* __restrict__ out_planes,
                                                        size_t out_stride, uint4* __restrict__ prefix_ws, uint32_t min_B, uint32_t n_lanes,
                                                        const uint32_t* __restrict__ gen, uint32_t* __restrict__ fix_count, const uint32_t* __restrict__ irr_src) {
  using F = typename C::F;
  using E = typename F::E;
  constexpr int M = F::MOD;
  static_assert(F::LANES == 1 && F::DEG == 1 && has_lazy<F>::value, "base fields only");
  constexpr int EW = FPS_WORDS;
  // A workgroup of EIGHT waves, two per SIMD: waves w and w + 4 sit on the same SIMD (waves go to the SIMDs round-robin) and form a pair,
  // so that every SIMD holds exactly one wave that inverts (with four-wave workgroups and pairs (0,1), (2,3) the two inverting waves of
  // the two workgroups of a CU met on SIMDs 0 and 2 and the inversion phase took twice as long: measured, 3 % slower than one wave).
  __shared__ __attribute__((aligned(16))) uint32_t xch[8 * 64 * FPS_WORDS];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, h = PAIRED ? wave >> 2 : 0u;
  constexpr uint32_t STEP = PAIRED ? 2u : 1u;
  const uint32_t S = offsG[n_buckets] << shift;
  const uint32_t B = max(min_B, (S + n_lanes - 1u) / n_lanes);
  const uint32_t NLe = S ? (S + B - 1u) / B : 1u;
  const uint32_t t0w = PAIRED ? (blockIdx.x * 4u + (wave & 3u)) * 64u : (blockIdx.x * 8u + wave) * 64u;   // first batch of the wave (pair)
  const bool wave_on = S != 0 && t0w < NLe;
  const uint32_t t = t0w + lane;
  const bool lane_on = wave_on && t < NLe;
  const uint32_t n_it = wave_on ? (S - t0w + NLe - 1u) / NLe : 0u;     // iterations of the pair (its first batch has the most slots)
  uint32_t* my_x = xch + ((size_t)wave * 64u + lane) * FPS_WORDS;
  uint32_t* partner_x = xch + ((size_t)(wave ^ 4u) * 64u + lane) * FPS_WORDS;

  // addresses (uint4 units) of the two inputs of slot o: planes x-even | x-odd | y-even | y-odd, blocked by element
  auto in_index = [=](uint32_t o, uint32_t& ia, uint32_t& ib, uint32_t& leftover) __attribute__((always_inline)) {
    if constexpr (IRR) {
      const uint32_t sw = irr_src[o];
      const uint32_t s1 = sw & 0x7fffffffu, s2 = s1 + ((sw >> 31) ^ 1u);
      leftover = sw >> 31;
      ia = (s1 & 1u) * (uint32_t)src_stride + (uint32_t)blk_index(s1 >> 1);
      ib = (s2 & 1u) * (uint32_t)src_stride + (uint32_t)blk_index(s2 >> 1);
    } else {
      leftover = 0u;
      ia = (uint32_t)blk_index(o);
      ib = (uint32_t)src_stride + ia;
    }
  };
  auto load_q = [=](E& r, const uint4* p) __attribute__((always_inline)) -> uint32_t {   // 7 quads 64 uint4 apart; returns the pad word
    uint32_t flag = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const uint4 v = p[(size_t)i * 64];
      r.l[4 * i] = v.x; r.l[4 * i + 1] = v.y; r.l[4 * i + 2] = v.z;
      if (4 * i + 3 < NL) r.l[4 * i + 3] = v.w; else flag = v.w;
    }
    return flag;
  };

  E run;
  F::one(run);
  // ---- forward: this wave's iterations, its own chain of prefix products
  for (uint32_t it = h; it < n_it; it += STEP) {
    const uint32_t o = it * NLe + t;
    const bool on = lane_on && o < S;
    const uint32_t oc = on ? o : (S - 1u);
    uint32_t ia, ib, leftover;
    in_index(oc, ia, ib, leftover);
    E x1, x2, den, tmp;
    uint32_t f0 = load_q(x1, src_planes + ia);
    uint32_t f1 = load_q(x2, src_planes + ib);
    if (leftover) f1 = PF_EMPTY;
    uint32_t kind;
    if (!on || (f0 & PF_EMPTY)) kind = PK_EMPTY;
    else if (f1 & PF_EMPTY) kind = PK_SINGLE;
    else {
      kind = PK_ADD;
      F::sub_raw(den, x2, x1);
      bool same_x = F::raw_maybe_zero(den);
      if (same_x) { E du; F::sub(du, x2, x1); same_x = F::is_zero(du); }
      if (same_x) {   // equal points (doubling, denominator 2y) or opposite points (cancellation, take 1): rare
        E y1, y2;
        (void)load_q(y1, src_planes + 2 * src_stride + ia);
        (void)load_q(y2, src_planes + 2 * src_stride + ib);
        fp_addsub<M>(den, y1, y2, ((f0 ^ f1) & PF_NEG) != 0);
        if (F::is_zero(den)) { F::one(den); kind = PK_CANCEL; } else kind = PK_DBL;
      }
    }
    if (on) fp_store_blk(prefix_ws, o, run, kind);
    if (kind <= PK_CANCEL) { F::mul_s(tmp, run, den); run = tmp; }
  }
  // ---- the two chains of a batch meet: one inversion per pair of waves
  E inv;
  if constexpr (!PAIRED) {
    E tmp;
    F::norm(tmp, run);
    F::inv(inv, tmp);
  } else {
    E tmp;
    F::norm(tmp, run);
    fp_store(my_x, tmp);
    __syncthreads();
    if (h == 0u) {
      E other, prod;
      fp_load(other, partner_x);
      F::mul_s(prod, tmp, other);
      F::norm(other, prod);
      F::inv(prod, other);          // 1 / (run_0 run_1)
      E r1;
      fp_load(r1, partner_x);       // run_1 once more (`other` was normalised over)
      F::mul_s(inv, prod, r1);      // 1 / run_0 = (1 / (run_0 run_1)) run_1
      // the partner finds 1 / (run_0 run_1) in ITS OWN slot (nobody needs run_1 any more); run_0 stays where it is for the partner
      fp_store(partner_x, prod);
    }
    __syncthreads();
    if (h == 1u) {
      E total_inv, r0;
      fp_load(total_inv, my_x);     // 1 / (run_0 run_1), left here by wave 0
      fp_load(r0, partner_x);       // run_0 (wave 0 never overwrote its own slot)
      F::mul_s(inv, total_inv, r0); // 1 / run_1
    }
  }
  // ---- backward: individual inverses and the sums, this wave's iterations from the last one down
  if (n_it > h) {
    const uint32_t it_last = PAIRED ? h + ((n_it - 1u - h) & ~1u) : n_it - 1u;
    for (uint32_t it = it_last;; it -= STEP) {
      const uint32_t o = it * NLe + t;
      const bool on = lane_on && o < S;
      const uint32_t oc = on ? o : (S - 1u);
      uint32_t ia, ib, leftover;
      in_index(oc, ia, ib, leftover);
      E pre, x1, x2, den, y1, num;
      const uint32_t kflag = load_q(pre, prefix_ws + blk_index(oc));
      uint32_t f0 = load_q(x1, src_planes + ia);
      uint32_t f1 = load_q(x2, src_planes + ib);
      if (leftover) f1 = PF_EMPTY;
      const uint32_t kind = on ? kflag : (uint32_t)PK_EMPTY;
      const bool flip = ((f0 ^ f1) & PF_NEG) != 0;
      uint32_t out_flag = PF_EMPTY;
      // denominators first: the two products of the inversion chain need nothing of y
      if (kind == PK_ADD) {
        out_flag = f1 & PF_NEG;
        F::sub_raw(den, x2, x1);
      } else if (kind == PK_DBL) {
        E y2;
        (void)load_q(y1, src_planes + 2 * src_stride + ia);
        (void)load_q(y2, src_planes + 2 * src_stride + ib);
        fp_addsub<M>(den, y1, y2, flip);
        out_flag = f0 & PF_NEG;
      } else {
        F::one(den);
      }
      {
        E res;
        F::mul_s(res, inv, pre); pre = res;       // 1 / den
        F::mul_s(res, inv, den); inv = res;       // the inverse of the shorter chain
      }
      // numerator
      if (kind == PK_ADD) {
        E y2;
        (void)load_q(y1, src_planes + 2 * src_stride + ia);
        (void)load_q(y2, src_planes + 2 * src_stride + ib);
        F::addsub_raw(num, y2, y1, !flip);
      } else if (kind == PK_DBL) {
        E a, sq;
        F::mul(sq, x1, x1);
        F::add(num, sq, sq); F::add(num, num, sq);
        C::coeff_a(a);
        F::add(num, num, a);
      } else {
        (void)load_q(y1, src_planes + 2 * src_stride + ia);   // an odd leftover hands (x1, y1) on
        F::one(num);
      }
      E lam, x3;
      F::mul_s(lam, num, pre);                    // lambda
      {
        E res;
        F::sqr_s(res, lam);
        F::sub_raw(res, res, x1);
        F::sub_raw(res, res, x2);
        F::norm(x3, res);                         // x3 = lambda^2 - x1 - x2
        F::sub_raw(num, x1, x3);
        F::mul_s(res, lam, num);
        F::addsub_raw(res, res, y1, !(kind == PK_ADD && flip));
        if (kind <= PK_DBL) F::norm(y1, res);     // y3' = lambda (x1 - x3) -+ y1; other kinds keep y1
      }
      if (kind == PK_SINGLE) { out_flag = f0 & PF_NEG; x3 = x1; }
      if (kind == PK_CANCEL) {                    // P + (-P): emit D, remember to take it out of the bucket again
        fp_load(x3, gen);
        fp_load(y1, gen + EW);
        out_flag = 0;
        const uint32_t f = o >> shift;            // final slot -> bucket: largest b with offsG[b] <= f
        uint32_t lo = 0, hi = n_buckets - 1u;
        while (lo < hi) {
          const uint32_t mid = (lo + hi + 1u) >> 1;
          if (offsG[mid] <= f) lo = mid; else hi = mid - 1u;
        }
        atomicAdd(&fix_count[lo], 1u);
      }
      if (on) {
        uint4* px = out_planes + (size_t)(o & 1u) * out_stride;
        fp_store_blk(px, o >> 1, x3, out_flag);
        fp_store_blk(px + 2 * out_stride, o >> 1, y1, 0u);
        if constexpr (last) out_sorted[o] = (out_flag & PF_EMPTY) ? ENTRY_EMPTY : ((o << 1) | ((out_flag & PF_NEG) ? 1u : 0u));
      }
      if (it < STEP) break;
    }
  }
}

// The same level at ONE wave per SIMD, still without the LDS image: every operand of the NEXT slot is loaded into registers of its own
// (five elements: 135 of the 512 a lone wave has) while the current slot's products run -- the prefetch distance the image gave, without
// its LDS-DMA instructions, its address arithmetic and its ds_reads; the five products written out as above.  No pairing of waves.
template <class C, bool last, bool IRR>
__global__ void __launch_bounds__(256, 1) k_pair_level1w(const uint4* __restrict__ src_planes, size_t src_stride, const uint32_t* __restrict__ offsG,
                                                        uint32_t n_buckets, uint32_t shift, uint32_t* __restrict__ out_sorted, uint4* __restrict__ out_planes,
                                                        size_t out_stride, uint4* __restrict__ prefix_ws, uint32_t min_B, uint32_t n_lanes,
                                                        const uint32_t* __restrict__ gen, uint32_t* __restrict__ fix_count, const uint32_t* __restrict__ irr_src) {
  using F = typename C::F;
  using E = typename F::E;
  constexpr int M = F::MOD;
  static_assert(F::LANES == 1 && F::DEG == 1 && has_lazy<F>::value, "base fields only");
  constexpr int EW = FPS_WORDS;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t S = offsG[n_buckets] << shift;
  const uint32_t B = max(min_B, (S + n_lanes - 1u) / n_lanes);
  const uint32_t NLe = S ? (S + B - 1u) / B : 1u;
  const uint32_t t0w = (blockIdx.x * 4u + wave) * 64u;
  if (S == 0 || t0w >= NLe) return;
  const uint32_t t = t0w + lane;
  const bool lane_on = t < NLe;
  const uint32_t n_it = (S - t0w + NLe - 1u) / NLe;
  auto in_index = [=](uint32_t o, uint32_t& ia, uint32_t& ib, uint32_t& leftover) __attribute__((always_inline)) {
    if constexpr (IRR) {
      const uint32_t sw = irr_src[o];
      const uint32_t s1 = sw & 0x7fffffffu, s2 = s1 + ((sw >> 31) ^ 1u);
      leftover = sw >> 31;
      ia = (s1 & 1u) * (uint32_t)src_stride + (uint32_t)blk_index(s1 >> 1);
      ib = (s2 & 1u) * (uint32_t)src_stride + (uint32_t)blk_index(s2 >> 1);
    } else {
      leftover = 0u;
      ia = (uint32_t)blk_index(o);
      ib = (uint32_t)src_stride + ia;
    }
  };
  auto load_q = [=](E& r, const uint4* p) __attribute__((always_inline)) -> uint32_t {
    uint32_t flag = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const uint4 v = p[(size_t)i * 64];
      r.l[4 * i] = v.x; r.l[4 * i + 1] = v.y; r.l[4 * i + 2] = v.z;
      if (4 * i + 3 < NL) r.l[4 * i + 3] = v.w; else flag = v.w;
    }
    return flag;
  };
  auto slot_of = [=](uint32_t it, bool& on) __attribute__((always_inline)) -> uint32_t {
    const uint32_t o = it * NLe + t;
    on = lane_on && o < S;
    return on ? o : (S - 1u);
  };

  // The prefetched operands stay RAW (seven quads per element, flag word included) until the slot that uses them: a flag taken out of
  // the loaded quad where the load is issued makes the compiler wait for that load on the spot (first form of this kernel: s_waitcnt
  // vmcnt(28) / (14) / (4) right behind the 35 loads -- the whole prefetch distance gone; VALU active 66 %).
  struct Raw { uint4 q[7]; };
  auto load_raw = [=](Raw& r, const uint4* p) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 7; ++i) r.q[i] = p[(size_t)i * 64];
  };
  auto unpack = [=](E& e, const Raw& r) __attribute__((always_inline)) -> uint32_t {
    uint32_t flag = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      e.l[4 * i] = r.q[i].x; e.l[4 * i + 1] = r.q[i].y; e.l[4 * i + 2] = r.q[i].z;
      if (4 * i + 3 < NL) e.l[4 * i + 3] = r.q[i].w; else flag = r.q[i].w;
    }
    return flag;
  };
  // ---- forward, the x coordinates of the next slot in flight
  E run;
  F::one(run);
  {
    Raw r1, r2;
    uint32_t ia, ib, leftover;
    bool on;
    uint32_t oc = slot_of(0u, on);
    in_index(oc, ia, ib, leftover);
    load_raw(r1, src_planes + ia);
    load_raw(r2, src_planes + ib);
    // The stores of a slot are issued at the TOP of the next one, ahead of its loads: the compiler waits for EVERYTHING outstanding
    // whenever loads and stores are pending together (one counter, completion order unknown between the two kinds), so a store issued
    // at the end of a slot would be waited for at the top of the next -- its whole write latency, every slot.  At the top it has a
    // slot's products to land in.
    E st_run;
    uint32_t st_o = 0, st_kind = 0;
    bool st_on = false;
    F::one(st_run);
    for (uint32_t it = 0; it < n_it; ++it) {
      E x1, x2;
      uint32_t f0 = unpack(x1, r1);
      uint32_t f1 = unpack(x2, r2);
      if (leftover) f1 = PF_EMPTY;
      if (st_on) fp_store_blk(prefix_ws, st_o, st_run, st_kind);
      const uint32_t o = oc;
      const bool on_c = on;
      const uint32_t ia_c = ia, ib_c = ib;
      {   // unconditional (the last iteration loads its own slot again): a load under a condition ends in register copies behind it
        // -- and a wait for the loads they copy from, on the spot
        oc = slot_of(min(it + 1u, n_it - 1u), on);
        in_index(oc, ia, ib, leftover);
        load_raw(r1, src_planes + ia);
        load_raw(r2, src_planes + ib);
      }
      E den, tmp;
      uint32_t kind;
      if (!on_c || (f0 & PF_EMPTY)) kind = PK_EMPTY;
      else if (f1 & PF_EMPTY) kind = PK_SINGLE;
      else {
        kind = PK_ADD;
        F::sub_raw(den, x2, x1);
        bool same_x = F::raw_maybe_zero(den);
        if (same_x) { E du; F::sub(du, x2, x1); same_x = F::is_zero(du); }
        if (same_x) {
          E y1, y2;
          (void)load_q(y1, src_planes + 2 * src_stride + ia_c);
          (void)load_q(y2, src_planes + 2 * src_stride + ib_c);
          fp_addsub<M>(den, y1, y2, ((f0 ^ f1) & PF_NEG) != 0);
          if (F::is_zero(den)) { F::one(den); kind = PK_CANCEL; } else kind = PK_DBL;
        }
      }
      st_run = run; st_o = o; st_kind = kind; st_on = on_c;
      if (kind <= PK_CANCEL) { F::mul_s(tmp, run, den); run = tmp; }
    }
    if (st_on) fp_store_blk(prefix_ws, st_o, st_run, st_kind);
  }
  E inv;
  { E tmp; F::norm(tmp, run); F::inv(inv, tmp); }
  // ---- backward, all five operands of the next slot in flight
  {
    Raw rp, r1, r2, ry1, ry2;
    uint32_t ia, ib, leftover;
    bool on;
    uint32_t oc = slot_of(n_it - 1u, on);
    in_index(oc, ia, ib, leftover);
    load_raw(rp, prefix_ws + blk_index(oc));
    load_raw(r1, src_planes + ia);
    load_raw(r2, src_planes + ib);
    load_raw(ry1, src_planes + 2 * src_stride + ia);
    load_raw(ry2, src_planes + 2 * src_stride + ib);
    E st_x, st_y;
    uint32_t st_o = 0, st_flag = 0;
    bool st_on = false;
    F::one(st_x); st_y = st_x;
    auto store_result = [=](uint32_t o, const E& x, const E& y, uint32_t flag) __attribute__((always_inline)) {
      uint4* px = out_planes + (size_t)(o & 1u) * out_stride;
      fp_store_blk(px, o >> 1, x, flag);
      fp_store_blk(px + 2 * out_stride, o >> 1, y, 0u);
      if constexpr (last) out_sorted[o] = (flag & PF_EMPTY) ? ENTRY_EMPTY : ((o << 1) | ((flag & PF_NEG) ? 1u : 0u));
    };
    for (uint32_t it = n_it - 1u;; --it) {
      E pre, x1, x2, y1, y2;
      const uint32_t kflag = unpack(pre, rp);
      uint32_t f0 = unpack(x1, r1);
      uint32_t f1 = unpack(x2, r2);
      (void)unpack(y1, ry1);
      (void)unpack(y2, ry2);
      if (leftover) f1 = PF_EMPTY;
      if (st_on) store_result(st_o, st_x, st_y, st_flag);
      const uint32_t o = oc;
      const uint32_t kind = on ? kflag : (uint32_t)PK_EMPTY;
      const bool on_c = on;
      {
        oc = slot_of(it > 0u ? it - 1u : 0u, on);
        in_index(oc, ia, ib, leftover);
        load_raw(rp, prefix_ws + blk_index(oc));
        load_raw(r1, src_planes + ia);
        load_raw(r2, src_planes + ib);
        load_raw(ry1, src_planes + 2 * src_stride + ia);
        load_raw(ry2, src_planes + 2 * src_stride + ib);
      }
      const bool flip = ((f0 ^ f1) & PF_NEG) != 0;
      uint32_t out_flag = PF_EMPTY;
      E den, num;
      if (kind == PK_ADD) {
        out_flag = f1 & PF_NEG;
        F::sub_raw(den, x2, x1);
        F::addsub_raw(num, y2, y1, !flip);
      } else if (kind == PK_DBL) {
        E a, sq;
        fp_addsub<M>(den, y1, y2, flip);
        F::mul(sq, x1, x1);
        F::add(num, sq, sq); F::add(num, num, sq);
        C::coeff_a(a);
        F::add(num, num, a);
        out_flag = f0 & PF_NEG;
      } else {
        F::one(den);
        num = den;
      }
      E res, lam, x3;
      F::mul_s(res, inv, pre); pre = res;         // 1 / den
      F::mul_s(res, inv, den); inv = res;         // the inverse of the shorter chain
      F::mul_s(lam, num, pre);                    // lambda
      F::sqr_s(res, lam);
      F::sub_raw(res, res, x1);
      F::sub_raw(res, res, x2);
      F::norm(x3, res);                           // x3 = lambda^2 - x1 - x2
      F::sub_raw(num, x1, x3);
      F::mul_s(res, lam, num);
      F::addsub_raw(res, res, y1, !(kind == PK_ADD && flip));
      if (kind <= PK_DBL) F::norm(y1, res);       // y3' = lambda (x1 - x3) -+ y1; other kinds keep y1
      if (kind == PK_SINGLE) { out_flag = f0 & PF_NEG; x3 = x1; }
      if (kind == PK_CANCEL) {
        fp_load(x3, gen);
        fp_load(y1, gen + EW);
        out_flag = 0;
        const uint32_t f = o >> shift;
        uint32_t lo = 0, hi = n_buckets - 1u;
        while (lo < hi) {
          const uint32_t mid = (lo + hi + 1u) >> 1;
          if (offsG[mid] <= f) lo = mid; else hi = mid - 1u;
        }
        atomicAdd(&fix_count[lo], 1u);
      }
      st_x = x3; st_y = y1; st_o = o; st_flag = out_flag; st_on = on_c;
      if (it == 0u) break;
    }
    if (st_on) store_result(st_o, st_x, st_y, st_flag);
  }
}

}  // namespace mnt753
