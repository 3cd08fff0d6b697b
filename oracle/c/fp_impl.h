/* Montgomery prime field on FP_N 64-bit limbs (CIOS, unsigned __int128). Included once per field with
 * FP_N, FP(name), FP_P, FP_R, FP_R2, FP_INV defined. ORACLE / TEST INFRASTRUCTURE ONLY.
 * Restates ark-ff 0.6.0 `Fp<MontBackend<_, N>, N>` semantics (element = x*2^(64N) mod p in N LE limbs). */
typedef struct { uint64_t l[FP_N]; } FP(t);

static inline void FP(set_zero)(FP(t)* r) { for (int i = 0; i < FP_N; i++) r->l[i] = 0; }
static inline void FP(set_one)(FP(t)* r) { for (int i = 0; i < FP_N; i++) r->l[i] = FP_R[i]; }
static inline int FP(is_zero)(const FP(t)* a) { uint64_t o = 0; for (int i = 0; i < FP_N; i++) o |= a->l[i]; return o == 0; }
static inline int FP(eq)(const FP(t)* a, const FP(t)* b) { uint64_t o = 0; for (int i = 0; i < FP_N; i++) o |= a->l[i] ^ b->l[i]; return o == 0; }

static inline uint64_t FP(sub_raw)(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t borrow = 0;
  for (int i = 0; i < FP_N; i++) {
    unsigned __int128 d = (unsigned __int128)a[i] - b[i] - borrow;
    r[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}
static inline uint64_t FP(add_raw)(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t carry = 0;
  for (int i = 0; i < FP_N; i++) {
    unsigned __int128 s = (unsigned __int128)a[i] + b[i] + carry;
    r[i] = (uint64_t)s;
    carry = (uint64_t)(s >> 64);
  }
  return carry;
}
static inline void FP(reduce)(FP(t)* a, uint64_t carry) {
  uint64_t t[FP_N];
  uint64_t borrow = FP(sub_raw)(t, a->l, FP_P);
  if (carry || !borrow) for (int i = 0; i < FP_N; i++) a->l[i] = t[i];
}
static inline void FP(add)(FP(t)* r, const FP(t)* a, const FP(t)* b) {
  uint64_t c = FP(add_raw)(r->l, a->l, b->l);
  FP(reduce)(r, c);
}
static inline void FP(sub)(FP(t)* r, const FP(t)* a, const FP(t)* b) {
  uint64_t t[FP_N];
  uint64_t borrow = FP(sub_raw)(t, a->l, b->l);
  if (borrow) FP(add_raw)(t, t, FP_P);
  for (int i = 0; i < FP_N; i++) r->l[i] = t[i];
}
static inline void FP(neg)(FP(t)* r, const FP(t)* a) {
  if (FP(is_zero)(a)) { *r = *a; return; }
  FP(sub_raw)(r->l, FP_P, a->l);
}
static inline void FP(dbl)(FP(t)* r, const FP(t)* a) { FP(add)(r, a, a); }

static inline void FP(mul)(FP(t)* r, const FP(t)* a, const FP(t)* b) {
  uint64_t t[FP_N + 2];
  for (int i = 0; i < FP_N + 2; i++) t[i] = 0;
  for (int i = 0; i < FP_N; i++) {
    unsigned __int128 acc;
    uint64_t carry = 0;
    for (int j = 0; j < FP_N; j++) {
      acc = (unsigned __int128)a->l[j] * b->l[i] + t[j] + carry;
      t[j] = (uint64_t)acc;
      carry = (uint64_t)(acc >> 64);
    }
    acc = (unsigned __int128)t[FP_N] + carry;
    t[FP_N] = (uint64_t)acc;
    t[FP_N + 1] = (uint64_t)(acc >> 64);
    uint64_t m = t[0] * FP_INV;
    acc = (unsigned __int128)m * FP_P[0] + t[0];
    carry = (uint64_t)(acc >> 64);
    for (int j = 1; j < FP_N; j++) {
      acc = (unsigned __int128)m * FP_P[j] + t[j] + carry;
      t[j - 1] = (uint64_t)acc;
      carry = (uint64_t)(acc >> 64);
    }
    acc = (unsigned __int128)t[FP_N] + carry;
    t[FP_N - 1] = (uint64_t)acc;
    t[FP_N] = t[FP_N + 1] + (uint64_t)(acc >> 64);
  }
  FP(t) o;
  for (int i = 0; i < FP_N; i++) o.l[i] = t[i];
  FP(reduce)(&o, t[FP_N]);
  *r = o;
}
static inline void FP(sqr)(FP(t)* r, const FP(t)* a) { FP(mul)(r, a, a); }
static inline void FP(from_mont)(FP(t)* r, const FP(t)* a) {
  FP(t) one; FP(set_zero)(&one); one.l[0] = 1;
  FP(mul)(r, a, &one);
}
static inline void FP(to_mont)(FP(t)* r, const FP(t)* a) {
  FP(t) r2; for (int i = 0; i < FP_N; i++) r2.l[i] = FP_R2[i];
  FP(mul)(r, a, &r2);
}
static inline void FP(from_u64)(FP(t)* r, uint64_t v) { FP(t) t; FP(set_zero)(&t); t.l[0] = v; FP(to_mont)(r, &t); }
/* a^e, e = canonical little-endian limbs */
static inline void FP(pow)(FP(t)* r, const FP(t)* a, const uint64_t* e, int nl) {
  FP(t) acc; FP(set_one)(&acc);
  for (int i = nl - 1; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      FP(sqr)(&acc, &acc);
      if ((e[i] >> b) & 1) FP(mul)(&acc, &acc, a);
    }
  *r = acc;
}
static inline void FP(inv)(FP(t)* r, const FP(t)* a) {
  uint64_t e[FP_N];
  for (int i = 0; i < FP_N; i++) e[i] = FP_P[i];
  e[0] -= 2; /* p odd and > 2: no borrow */
  FP(pow)(r, a, e, FP_N);
}
