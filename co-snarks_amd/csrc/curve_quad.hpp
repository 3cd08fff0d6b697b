// Quad-distributed XYZZ arithmetic for the latency-bound tail of the MSM (bucket merge, window reduction, fold tree).
//
// Those stages are a dependent chain of ~45 point operations per window with only ~0.5 wave per SIMD in flight: a lone
// wave64 issues one VALU instruction per ~4-5 cycles whatever its occupancy, so the chain runs at the latency of a
// lane-serial point addition (12M + 2S ~ 3200 instructions ~ 6.5 us, profiles/archive/r02_a_msm_stages.log: 0.55-0.59 ms of the
// 2.1 ms BN254 G1 2^20 step). The only lever is parallelism INSIDE one point operation. Here four adjacent lanes (a DPP
// quad) own one point: lane role 0 holds X, 1 holds Y, 2 holds ZZ, 3 holds ZZZ. The independent field products of the
// EFD formulas run on different lanes of the quad in the same instruction slot; operands move between the lanes with
// v_mov_b32_dpp quad_perm (one VALU instruction per limb, no LDS):
//
//   add  (EFD add-2008-s, 12M + 2S + fused)  ->  5 product rounds:  {u1, s1, zz1 zz2, zzz1 zzz2} {u2, s2} {pp, rr} {q, zz3, ppp}
//                                                {y3 = r (q - x3) - s1 ppp  |  zzz3}           ~ 5.5 product times instead of ~14
//   dbl  (EFD dbl-2008-s-1, 6M + 3S)         ->  4 product rounds:  {xx, V} {S, W, zz3} {M^2, zzz3} {y3}   ~ 4.5 instead of ~8.6
//
// The arithmetic is the same signed lazy field (field29.hpp) with the same operand bounds as curve_lazy.hpp: every product
// below has operands of the classes lazy_add_inl / lazy_dbl_inl already feed to the same routines. Lanes whose role has
// nothing to do in a round compute a harmless product of valid operands (SIMD: the instruction is issued anyway).
// Stored points keep the XYZZLazy<L> memory layout: role r loads / stores member r of the struct.
#pragma once
#include <stddef.h>

#include "curve_lazy.hpp"

namespace csh {

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

// quad_perm control word: lane r of every quad reads lane P_r of the same quad
template <int P0, int P1, int P2, int P3>
struct QuadCtrl {
  static constexpr int value = P0 | (P1 << 2) | (P2 << 4) | (P3 << 6);
};

template <int CTRL>
__device__ __forceinline__ int32_t quad_mov(int32_t v) {
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}

// limb-wise helpers, generic over FpS and Fp2S
template <int CTRL, class LP, class F32>
__device__ __forceinline__ FpS<LP, F32> quad_perm(const FpS<LP, F32>& a) {
  FpS<LP, F32> r;
#pragma unroll
  for (int i = 0; i < FpS<LP, F32>::NL; ++i) r.l[i] = quad_mov<CTRL>(a.l[i]);
  return r;
}
template <int CTRL, class LF, class F2>
__device__ __forceinline__ Fp2S<LF, F2> quad_perm(const Fp2S<LF, F2>& a) {
  return {quad_perm<CTRL>(a.c0), quad_perm<CTRL>(a.c1)};
}
template <class LP, class F32>
__device__ __forceinline__ FpS<LP, F32> lane_select(bool c, const FpS<LP, F32>& a, const FpS<LP, F32>& b) {  // c ? a : b
  FpS<LP, F32> r;
#pragma unroll
  for (int i = 0; i < FpS<LP, F32>::NL; ++i) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}
template <class LF, class F2>
__device__ __forceinline__ Fp2S<LF, F2> lane_select(bool c, const Fp2S<LF, F2>& a, const Fp2S<LF, F2>& b) {
  return {lane_select(c, a.c0, b.c0), lane_select(c, a.c1, b.c1)};
}

using QBcast0 = QuadCtrl<0, 0, 0, 0>;
using QBcast1 = QuadCtrl<1, 1, 1, 1>;
using QBcast3 = QuadCtrl<3, 3, 3, 3>;
using QZzOwn = QuadCtrl<2, 3, 2, 3>;  // roles 0/1 read the ZZ / ZZZ lanes, roles 2/3 read themselves

// One quarter of an XYZZ point: the member selected by the lane's role, plus the (quad-uniform) empty flag.
template <class L>
struct QPt {
  L v;
  bool empty;
};

template <class L>
__device__ __forceinline__ QPt<L> qpt_inf() {
  return {L::zero(), true};
}
template <class L>
__device__ __forceinline__ QPt<L> qpt_load(const XYZZLazy<L>* p, int role) {
  QPt<L> r;
  r.empty = p->empty;
  r.v = (&p->x)[role];  // members x, y, zz, zzz are consecutive L objects
  return r;
}
template <class L>
__device__ __forceinline__ void qpt_store(XYZZLazy<L>* p, int role, const QPt<L>& q) {
  (&p->x)[role] = q.v;
  if (role == 0) p->empty = q.empty;
}
static_assert(offsetof(XYZZLazy<Fq29s>, y) == sizeof(Fq29s) && offsetof(XYZZLazy<Fq29s>, zzz) == 3 * sizeof(Fq29s), "XYZZLazy members must be contiguous");

// broadcast a per-lane predicate from the lane with role SRC to its whole quad
template <int SRC>
__device__ __forceinline__ bool quad_bcast_flag(bool f) {
  return quad_mov<QuadCtrl<SRC, SRC, SRC, SRC>::value>((int32_t)f) != 0;
}

// acc = 2 * acc. 4 product rounds.
template <class L>
__device__ __forceinline__ void qdbl(QPt<L>& acc, int role) {
  if (acc.empty) return;
  // 2-torsion (y = 0) cannot occur on these curves; kept for completeness (the filter is 3 instructions)
  if (quad_bcast_flag<1>(acc.v.maybe_zero())) {
    if (quad_bcast_flag<1>(acc.v.is_zero_slow())) {
      acc = qpt_inf<L>();
      return;
    }
  }
  const L U = L::add(acc.v, acc.v).normalized();          // role 1: 2 y
  const L A = lane_select(role == 1, U, acc.v);            // role 0: x, 1: U, 2: zz, 3: zzz
  const L SQ = L::sqr(A);                                  // role 0: xx, role 1: V = U^2
  const L V = quad_perm<QBcast1::value>(SQ);
  const L R2 = L::mul(A, V);                               // role 0: S = x V, 1: W = U V, 2: zz3 = zz V
  const L M = L::add(L::add(SQ, SQ), SQ).normalized();     // role 0: 3 xx
  const L W = quad_perm<QBcast1::value>(R2);
  const L R3 = L::mul(lane_select(role == 0, M, acc.v), lane_select(role == 0, M, W));  // role 0: M^2, role 3: zzz3 = zzz W
  const L x3 = L::sub(R3, L::add(R2, R2)).normalized();    // role 0: M^2 - 2 S
  const L T = L::sub(R2, x3);                              // role 0: S - x3
  const L Mb = quad_perm<QBcast0::value>(M);
  const L Tb = quad_perm<QBcast0::value>(T);
  const L y3 = L::mul_sub(Mb, Tb, R2, acc.v);              // role 1: M (S - x3) - W y1
  acc.v = role == 0 ? x3 : (role == 1 ? y3 : (role == 2 ? R2 : R3));
}

// acc += p. 5 product rounds; the doubling / cancellation case (rare) is detected on the X and Y lanes and broadcast.
template <class L>
__device__ __forceinline__ void qadd(QPt<L>& acc, const QPt<L>& p, int role) {
  if (p.empty) return;
  if (acc.empty) {
    acc = p;
    return;
  }
  const L R1 = L::mul(acc.v, quad_perm<QZzOwn::value>(p.v));   // role 0: u1 = x1 zz2, 1: s1 = y1 zzz2, 2: zz1 zz2, 3: zzz1 zzz2
  const L R2 = L::mul(p.v, quad_perm<QZzOwn::value>(acc.v));   // role 0: u2 = x2 zz1, 1: s2 = y2 zzz1
  const L D = L::sub(R2, R1);                                  // role 0: p = u2 - u1, role 1: r = s2 - s1
  if (quad_bcast_flag<0>(D.maybe_zero())) {
    if (quad_bcast_flag<0>(D.is_zero_slow())) {
      if (quad_bcast_flag<1>(D.is_zero())) {
        qdbl<L>(acc, role);
      } else {
        acc = qpt_inf<L>();  // P + (-P)
      }
      return;
    }
  }
  const L SQ = L::sqr(D);                                      // role 0: pp, role 1: rr
  const L pp = quad_perm<QBcast0::value>(SQ);
  const L pb = quad_perm<QBcast0::value>(D);
  const L R4 = L::mul(lane_select(role == 3, pb, R1), pp);     // role 0: q = u1 pp, 2: zz3 = (zz1 zz2) pp, 3: ppp = p pp
  const L ppp = quad_perm<QBcast3::value>(R4);
  const L rr = quad_perm<QBcast1::value>(SQ);
  const L x3 = L::sub(L::sub(rr, ppp), L::add(R4, R4)).normalized();   // role 0: rr - ppp - 2 q
  const L T = quad_perm<QBcast0::value>(L::sub(R4, x3));       // (q - x3) of role 0, everywhere
  // role 1: y3 = r (q - x3) - s1 ppp ; role 3: zzz3 = (zzz1 zzz2) ppp - 0
  const L a = lane_select(role == 1, D, R1);
  const L b = lane_select(role == 1, T, R4);
  const L c = lane_select(role == 1, R1, L::zero());
  const L R5 = L::mul_sub(a, b, c, ppp);
  acc.v = role == 0 ? x3 : (role == 2 ? R4 : R5);
}

// k * P for a small unsigned k (double-and-add, k is quad-uniform)
template <class L>
__device__ __forceinline__ QPt<L> qmul_small(const QPt<L>& p, uint32_t k, int role) {
  QPt<L> r = qpt_inf<L>();
  if (k == 0 || p.empty) return r;
  int top = 31;
  while (!((k >> top) & 1)) --top;
  for (int b = top; b >= 0; --b) {
    qdbl<L>(r, role);
    if ((k >> b) & 1) qadd<L>(r, p, role);
  }
  return r;
}

#endif  // device

}  // namespace csh
