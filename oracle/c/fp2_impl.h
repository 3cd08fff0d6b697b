/* Fp2 = Fp[i]/(i^2 + F2_NR) over the base field BF(name). Included once per curve with F2(name), BF(name) and optionally F2_NR
 * (default 1: BN254, BLS12-381; 5: BLS12-377, ark-bls12-377 Fq2Config::NONRESIDUE = -5). ORACLE ONLY. */
#ifndef F2_NR
#define F2_NR 1
#endif
static inline void F2(times_nr)(BF(t)* r, const BF(t)* a) {   /* r = F2_NR * a by additions */
  BF(t) acc = *a;
  for (int i = 1; i < F2_NR; i++) BF(add)(&acc, &acc, a);
  *r = acc;
}
typedef struct { BF(t) c0, c1; } F2(t);
static inline void F2(set_zero)(F2(t)* r) { BF(set_zero)(&r->c0); BF(set_zero)(&r->c1); }
static inline void F2(set_one)(F2(t)* r) { BF(set_one)(&r->c0); BF(set_zero)(&r->c1); }
static inline int F2(is_zero)(const F2(t)* a) { return BF(is_zero)(&a->c0) && BF(is_zero)(&a->c1); }
static inline int F2(eq)(const F2(t)* a, const F2(t)* b) { return BF(eq)(&a->c0, &b->c0) && BF(eq)(&a->c1, &b->c1); }
static inline void F2(add)(F2(t)* r, const F2(t)* a, const F2(t)* b) { BF(add)(&r->c0, &a->c0, &b->c0); BF(add)(&r->c1, &a->c1, &b->c1); }
static inline void F2(sub)(F2(t)* r, const F2(t)* a, const F2(t)* b) { BF(sub)(&r->c0, &a->c0, &b->c0); BF(sub)(&r->c1, &a->c1, &b->c1); }
static inline void F2(neg)(F2(t)* r, const F2(t)* a) { BF(neg)(&r->c0, &a->c0); BF(neg)(&r->c1, &a->c1); }
static inline void F2(dbl)(F2(t)* r, const F2(t)* a) { F2(add)(r, a, a); }
static inline void F2(mul)(F2(t)* r, const F2(t)* a, const F2(t)* b) {
  BF(t) t0, t1, t2, t3;
  BF(mul)(&t0, &a->c0, &b->c0);
  BF(mul)(&t1, &a->c1, &b->c1);
  BF(mul)(&t2, &a->c0, &b->c1);
  BF(mul)(&t3, &a->c1, &b->c0);
  F2(times_nr)(&t1, &t1);
  BF(sub)(&r->c0, &t0, &t1);
  BF(add)(&r->c1, &t2, &t3);
}
static inline void F2(sqr)(F2(t)* r, const F2(t)* a) { F2(mul)(r, a, a); }
static inline void F2(inv)(F2(t)* r, const F2(t)* a) {
  BF(t) n, t, i;
  BF(sqr)(&n, &a->c0); BF(sqr)(&t, &a->c1); F2(times_nr)(&t, &t); BF(add)(&n, &n, &t); BF(inv)(&i, &n);
  BF(mul)(&r->c0, &a->c0, &i);
  BF(mul)(&t, &a->c1, &i); BF(neg)(&r->c1, &t);
}
