/* Short-Weierstrass (a = 0) group in Jacobian coordinates + Pippenger MSM, generic over the base field FE(name)
 * and scalar field SF(name). Included once per group with EC(name), FE(name), SF(name), SF_BITS. ORACLE ONLY.
 *
 * Restates what the reference gets from ark-ec 0.6.0 (Jacobian `Projective`, mixed addition) and the published
 * arkworks VariableBaseMSM shape behind taceo_ark_algebra::msm::msm_unchecked / msm_bigint (call sites:
 * co-circom/co-groth16/src/groth16.rs:193-194, mpc/plain.rs:66-74, mpc/rep3.rs:124-132): window
 * c = ln(n)+2 (3 below 32 points), unsigned c-bit digits, one bucket array per window, running-sum bucket
 * reduction, Horner over windows; parallel over (window, point-chunk) tasks. Independent of the product code
 * (which uses XYZZ coordinates, signed digits and a sort-based bucket layout). */
typedef struct { FE(t) x, y; } EC(aff);          /* infinity = all-zero */
typedef struct { FE(t) x, y, z; } EC(jac);       /* infinity: z = 0 */

static inline int EC(aff_is_inf)(const EC(aff)* p) { return FE(is_zero)(&p->x) && FE(is_zero)(&p->y); }
static inline void EC(jac_set_inf)(EC(jac)* p) { FE(set_one)(&p->x); FE(set_one)(&p->y); FE(set_zero)(&p->z); }
static inline int EC(jac_is_inf)(const EC(jac)* p) { return FE(is_zero)(&p->z); }

static void EC(jac_dbl)(EC(jac)* r, const EC(jac)* p) { /* dbl-2009-l */
  if (EC(jac_is_inf)(p) || FE(is_zero)(&p->y)) { EC(jac_set_inf)(r); return; }
  FE(t) A, B, C, D, E, F, t;
  FE(sqr)(&A, &p->x); FE(sqr)(&B, &p->y); FE(sqr)(&C, &B);
  FE(add)(&t, &p->x, &B); FE(sqr)(&t, &t); FE(sub)(&t, &t, &A); FE(sub)(&t, &t, &C); FE(dbl)(&D, &t);
  FE(dbl)(&E, &A); FE(add)(&E, &E, &A);
  FE(sqr)(&F, &E);
  FE(t) x3, y3, z3;
  FE(dbl)(&t, &D); FE(sub)(&x3, &F, &t);
  FE(mul)(&z3, &p->y, &p->z); FE(dbl)(&z3, &z3);
  FE(sub)(&t, &D, &x3); FE(mul)(&y3, &E, &t);
  FE(dbl)(&t, &C); FE(dbl)(&t, &t); FE(dbl)(&t, &t); FE(sub)(&y3, &y3, &t);
  r->x = x3; r->y = y3; r->z = z3;
}

static void EC(jac_add_mixed)(EC(jac)* r, const EC(jac)* p, const EC(aff)* q) { /* madd-2007-bl */
  if (EC(aff_is_inf)(q)) { *r = *p; return; }
  if (EC(jac_is_inf)(p)) { r->x = q->x; r->y = q->y; FE(set_one)(&r->z); return; }
  FE(t) z1z1, u2, s2, h, hh, i, j, rr, v, t;
  FE(sqr)(&z1z1, &p->z);
  FE(mul)(&u2, &q->x, &z1z1);
  FE(mul)(&s2, &q->y, &p->z); FE(mul)(&s2, &s2, &z1z1);
  if (FE(eq)(&u2, &p->x)) {
    if (FE(eq)(&s2, &p->y)) { EC(jac_dbl)(r, p); return; }
    EC(jac_set_inf)(r); return;
  }
  FE(sub)(&h, &u2, &p->x);
  FE(sqr)(&hh, &h);
  FE(dbl)(&i, &hh); FE(dbl)(&i, &i);
  FE(mul)(&j, &h, &i);
  FE(sub)(&rr, &s2, &p->y); FE(dbl)(&rr, &rr);
  FE(mul)(&v, &p->x, &i);
  FE(t) x3, y3, z3;
  FE(sqr)(&x3, &rr); FE(sub)(&x3, &x3, &j); FE(dbl)(&t, &v); FE(sub)(&x3, &x3, &t);
  FE(sub)(&t, &v, &x3); FE(mul)(&y3, &rr, &t);
  FE(mul)(&t, &p->y, &j); FE(dbl)(&t, &t); FE(sub)(&y3, &y3, &t);
  FE(add)(&z3, &p->z, &h); FE(sqr)(&z3, &z3); FE(sub)(&z3, &z3, &z1z1); FE(sub)(&z3, &z3, &hh);
  r->x = x3; r->y = y3; r->z = z3;
}

static void EC(jac_add)(EC(jac)* r, const EC(jac)* p, const EC(jac)* q) { /* add-2007-bl */
  if (EC(jac_is_inf)(q)) { *r = *p; return; }
  if (EC(jac_is_inf)(p)) { *r = *q; return; }
  FE(t) z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t;
  FE(sqr)(&z1z1, &p->z); FE(sqr)(&z2z2, &q->z);
  FE(mul)(&u1, &p->x, &z2z2); FE(mul)(&u2, &q->x, &z1z1);
  FE(mul)(&s1, &p->y, &q->z); FE(mul)(&s1, &s1, &z2z2);
  FE(mul)(&s2, &q->y, &p->z); FE(mul)(&s2, &s2, &z1z1);
  if (FE(eq)(&u1, &u2)) {
    if (FE(eq)(&s1, &s2)) { EC(jac_dbl)(r, p); return; }
    EC(jac_set_inf)(r); return;
  }
  FE(sub)(&h, &u2, &u1);
  FE(dbl)(&i, &h); FE(sqr)(&i, &i);
  FE(mul)(&j, &h, &i);
  FE(sub)(&rr, &s2, &s1); FE(dbl)(&rr, &rr);
  FE(mul)(&v, &u1, &i);
  FE(t) x3, y3, z3;
  FE(sqr)(&x3, &rr); FE(sub)(&x3, &x3, &j); FE(dbl)(&t, &v); FE(sub)(&x3, &x3, &t);
  FE(sub)(&t, &v, &x3); FE(mul)(&y3, &rr, &t);
  FE(mul)(&t, &s1, &j); FE(dbl)(&t, &t); FE(sub)(&y3, &y3, &t);
  FE(add)(&z3, &p->z, &q->z); FE(sqr)(&z3, &z3); FE(sub)(&z3, &z3, &z1z1); FE(sub)(&z3, &z3, &z2z2); FE(mul)(&z3, &z3, &h);
  r->x = x3; r->y = y3; r->z = z3;
}

static void EC(jac_to_aff)(EC(aff)* r, const EC(jac)* p) {
  if (EC(jac_is_inf)(p)) { FE(set_zero)(&r->x); FE(set_zero)(&r->y); return; }
  FE(t) zi, zi2, zi3;
  FE(inv)(&zi, &p->z); FE(sqr)(&zi2, &zi); FE(mul)(&zi3, &zi2, &zi);
  FE(mul)(&r->x, &p->x, &zi2); FE(mul)(&r->y, &p->y, &zi3);
}

/* k * P, k a 64-bit integer */
static void EC(mul_u64)(EC(jac)* r, const EC(aff)* p, uint64_t k) {
  EC(jac) acc; EC(jac_set_inf)(&acc);
  for (int b = 63; b >= 0; b--) {
    EC(jac_dbl)(&acc, &acc);
    if ((k >> b) & 1) EC(jac_add_mixed)(&acc, &acc, p);
  }
  *r = acc;
}

static inline uint32_t EC(digit)(const uint64_t* s, int bit, int c) {
  int limb = bit >> 6, off = bit & 63;
  if (limb >= SF_N) return 0;
  unsigned __int128 two = s[limb];
  if (limb + 1 < SF_N) two |= (unsigned __int128)s[limb + 1] << 64;
  return (uint32_t)((uint64_t)(two >> off) & ((1ull << c) - 1));
}

/* out (affine) = sum scalars[i] * points[i]; scalars Montgomery (msm_unchecked) or canonical (msm_bigint) */
static void EC(msm)(EC(aff)* out, const EC(aff)* points, const uint64_t* scalars, size_t n, int mont, int nthreads) {
  EC(jac) total; EC(jac_set_inf)(&total);
  if (n == 0) { EC(jac_to_aff)(out, &total); return; }
  const int dbg = getenv("OC_DEBUG") != NULL;
  double t_0 = omp_get_wtime();
  uint64_t* sc = (uint64_t*)malloc(n * SF_N * 8);
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (size_t i = 0; i < n; i++) {
    SF(t) v; memcpy(&v, scalars + i * SF_N, SF_N * 8);
    if (mont) SF(from_mont)(&v, &v);
    memcpy(sc + i * SF_N, &v, SF_N * 8);
  }
  /* arkworks picks c = ln(n) + 2 and parallelises over the W windows only; with many more threads than windows the
   * points are also cut into chunks, each (window, chunk) task owning a private bucket set. The window width is then
   * the one minimising a task's cost  n/chunks mixed additions + 2 * 2^c full additions (~1.4x a mixed one)  subject
   * to W * chunks <= threads -- for few threads this returns arkworks' own choice. */
  int c = n < 32 ? 3 : (int)(log((double)n)) + 2;
  int W = (SF_BITS + c - 1) / c;
  int chunks = nthreads / W;
  if (chunks < 1) chunks = 1;
  if (nthreads > W && n >= 1024) {
    double best = 1e300;
    for (int cc = 4; cc <= 20; cc++) {
      int ww = (SF_BITS + cc - 1) / cc;
      int ch = nthreads / ww;
      if (ch < 1) ch = 1;
      double rounds = ceil((double)ww * ch / nthreads);
      double cost = rounds * ((double)n / ch + 2.8 * (double)((size_t)1 << cc));
      if (cost < best) { best = cost; c = cc; W = ww; chunks = ch; }
    }
  }
  if ((size_t)chunks > n) chunks = (int)n;
  if (getenv("OC_DEBUG")) fprintf(stderr, "[oc_msm] n=%zu threads=%d c=%d W=%d chunks=%d\n", n, nthreads, c, W, chunks);
  size_t per = (n + chunks - 1) / chunks;
  size_t nb = ((size_t)1 << c) - 1;
  double t_1 = omp_get_wtime();
  EC(jac)* wsum = (EC(jac)*)malloc(sizeof(EC(jac)) * W * chunks);
  /* one allocation for every task's bucket set: per-task malloc/free of MB-sized blocks serialises the threads on the
   * process's mmap lock (measured: negative scaling beyond 32 threads) */
  EC(jac)* all_buckets = (EC(jac)*)malloc(sizeof(EC(jac)) * nb * (size_t)W * chunks);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
  for (int task = 0; task < W * chunks; task++) {
    int w = task / chunks, ch = task % chunks;
    size_t lo = (size_t)ch * per, hi = lo + per; if (hi > n) hi = n;
    EC(jac)* buckets = all_buckets + (size_t)task * nb;
    for (size_t b = 0; b < nb; b++) EC(jac_set_inf)(&buckets[b]);
    for (size_t i = lo; i < hi; i++) {
      uint32_t d = EC(digit)(sc + i * SF_N, w * c, c);
      if (d) EC(jac_add_mixed)(&buckets[d - 1], &buckets[d - 1], &points[i]);
    }
    EC(jac) running, acc; EC(jac_set_inf)(&running); EC(jac_set_inf)(&acc);
    for (size_t b = nb; b-- > 0;) {
      EC(jac_add)(&running, &running, &buckets[b]);
      EC(jac_add)(&acc, &acc, &running);
    }
    wsum[task] = acc;
  }
  double t_2 = omp_get_wtime();
  for (int w = W - 1; w >= 0; w--) {
    for (int k = 0; k < c; k++) EC(jac_dbl)(&total, &total);
    for (int ch = 0; ch < chunks; ch++) EC(jac_add)(&total, &total, &wsum[w * chunks + ch]);
  }
  EC(jac_to_aff)(out, &total);
  if (dbg) fprintf(stderr, "[oc_msm] convert %.1f ms, bucket tasks %.1f ms, fold %.1f ms\n", (t_1 - t_0) * 1e3, (t_2 - t_1) * 1e3, (omp_get_wtime() - t_2) * 1e3);
  free(all_buckets); free(wsum); free(sc);
}

/* ---- second, faster restatement: the CPU baseline ("port") and the full-size checker ---------------------------------
 * Same function (msm_unchecked / msm_bigint), built the way a tuned CPU Pippenger is: Booth-recoded signed c-bit digits
 * (2^(c-1) buckets per window, no carry chain between windows), XYZZ bucket accumulators with mixed additions (8M + 2S,
 * EFD madd-2008-s), thread-private bucket arrays allocated and first-touched by the thread that uses them, dynamic
 * (window, point-chunk) tasks, software prefetch of the next bucket / point. Deliberately different from both the
 * product (sort-based, lane runs, lazy 29-bit limbs, carry-propagating recoding) and EC(msm) above (Jacobian, unsigned
 * digits), so the three agree only if all three are right. Not arkworks: a restatement of the published algorithm shape. */
typedef struct { FE(t) x, y, zz, zzz; } EC(xyzz); /* infinity: zz = 0 */
static inline void EC(xyzz_set_inf)(EC(xyzz)* p) { FE(set_zero)(&p->x); FE(set_zero)(&p->y); FE(set_zero)(&p->zz); FE(set_zero)(&p->zzz); }
static inline int EC(xyzz_is_inf)(const EC(xyzz)* p) { return FE(is_zero)(&p->zz); }

static void EC(xyzz_dbl_aff)(EC(xyzz)* r, const FE(t)* x, const FE(t)* y) { /* mdbl-2008-s-1, a = 0 */
  FE(t) u, v, w, s, m, t;
  FE(dbl)(&u, y); FE(sqr)(&v, &u); FE(mul)(&w, &u, &v); FE(mul)(&s, x, &v);
  FE(sqr)(&m, x); FE(dbl)(&t, &m); FE(add)(&m, &t, &m);
  FE(sqr)(&r->x, &m); FE(dbl)(&t, &s); FE(sub)(&r->x, &r->x, &t);
  FE(sub)(&t, &s, &r->x); FE(mul)(&t, &m, &t); FE(mul)(&u, &w, y); FE(sub)(&r->y, &t, &u);
  r->zz = v; r->zzz = w;
}
/* a += (qx, +-qy) */
static inline void EC(xyzz_madd)(EC(xyzz)* a, const EC(aff)* q, int negate) {
  if (EC(aff_is_inf)(q)) return;
  FE(t) qy = q->y;
  if (negate) FE(neg)(&qy, &qy);
  if (EC(xyzz_is_inf)(a)) { a->x = q->x; a->y = qy; FE(set_one)(&a->zz); FE(set_one)(&a->zzz); return; }
  FE(t) u2, s2, p, r, pp, ppp, qq, t;
  FE(mul)(&u2, &q->x, &a->zz); FE(mul)(&s2, &qy, &a->zzz);
  FE(sub)(&p, &u2, &a->x); FE(sub)(&r, &s2, &a->y);
  if (FE(is_zero)(&p)) {
    if (FE(is_zero)(&r)) EC(xyzz_dbl_aff)(a, &q->x, &qy); else EC(xyzz_set_inf)(a);
    return;
  }
  FE(sqr)(&pp, &p); FE(mul)(&ppp, &p, &pp); FE(mul)(&qq, &a->x, &pp);
  FE(sqr)(&t, &r); FE(sub)(&t, &t, &ppp); FE(sub)(&t, &t, &qq); FE(sub)(&t, &t, &qq);
  FE(t) y3; FE(sub)(&y3, &qq, &t); FE(mul)(&y3, &r, &y3); FE(mul)(&u2, &a->y, &ppp); FE(sub)(&a->y, &y3, &u2);
  a->x = t;
  FE(mul)(&a->zz, &a->zz, &pp); FE(mul)(&a->zzz, &a->zzz, &ppp);
}
static void EC(xyzz_to_jac)(EC(jac)* r, const EC(xyzz)* p) { /* (X, Y, ZZ, ZZZ) -> Jacobian (X ZZZ^2 ZZ, Y ZZZ^3 ZZ^3 ... ) via z = ZZZ/ZZ */
  if (EC(xyzz_is_inf)(p)) { EC(jac_set_inf)(r); return; }
  /* z = zzz / zz satisfies z^2 = zz, z^3 = zzz. Avoid the inversion: scale to z' = zz * zzz (z'^2 = zz^2 zzz^2 = zz * zz^3... ) --
   * simpler and exact: x/zz, y/zzz are the affine coordinates; Jacobian with Z = zz*zzz: X = x * zz * zzz^2, Y = y * zz^3 * zzz^2 */
  FE(t) z, z2, t;
  FE(mul)(&z, &p->zz, &p->zzz);            /* Z = zz zzz, Z^2 = zz^2 zzz^2, Z^3 = zz^3 zzz^3 */
  FE(sqr)(&z2, &p->zzz);                   /* zzz^2 */
  FE(mul)(&t, &p->x, &p->zz); FE(mul)(&r->x, &t, &z2);                 /* x/zz * Z^2 = x zz zzz^2 */
  FE(sqr)(&t, &p->zz); FE(mul)(&t, &t, &p->zz); FE(mul)(&t, &t, &z2); /* zz^3 zzz^2 */
  FE(mul)(&r->y, &p->y, &t);                                          /* y/zzz * Z^3 */
  r->z = z;
}

/* Booth digit of window w (c bits): d in [-2^(c-1), 2^(c-1)]; sum_w d_w 2^(wc) = s when W*c > bit length of s */
static inline int32_t EC(booth)(const uint64_t* s, int w, int c) {
  const int bit = w * c - 1; /* window [bit, bit + c] */
  uint64_t v;
  if (bit < 0) {
    v = (s[0] << 1) & ((1ull << (c + 1)) - 1);
  } else {
    const int limb = bit >> 6, off = bit & 63;
    unsigned __int128 two = limb < SF_N ? s[limb] : 0;
    if (limb + 1 < SF_N) two |= (unsigned __int128)s[limb + 1] << 64;
    v = (uint64_t)(two >> off) & ((1ull << (c + 1)) - 1);
  }
  const int32_t d = (int32_t)((v + 1) >> 1);
  return (v >> c) ? d - (int32_t)(1u << c) : d;
}

static void EC(msm_fast)(EC(aff)* out, const EC(aff)* points, const uint64_t* scalars, size_t n, int mont, int nthreads, int force_c, double* stage_s) {
  EC(jac) total; EC(jac_set_inf)(&total);
  if (n == 0) { EC(jac_to_aff)(out, &total); return; }
  const double t_0 = omp_get_wtime();
  uint64_t* sc = (uint64_t*)scalars;
  uint64_t* sc_own = NULL;
  if (mont) {
    sc_own = (uint64_t*)malloc(n * SF_N * 8);
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (size_t i = 0; i < n; i++) {
      SF(t) v; memcpy(&v, scalars + i * SF_N, SF_N * 8);
      SF(from_mont)(&v, &v);
      memcpy(sc_own + i * SF_N, &v, SF_N * 8);
    }
    sc = sc_own;
  }
  /* window width: minimise per-thread work  W * (n + reduce(c)) / T  with the bucket array (2^(c-1) XYZZ) preferably
   * inside a core's L2 (<= 1 MiB); tasks = W x chunks >= 4 T for balance under dynamic scheduling */
  int c = force_c;
  if (c < 2 || c > 20) {
    double best = 1e300; c = 8;
    for (int cc = 4; cc <= 18; cc++) {
      const int ww = (SF_BITS + cc) / cc;                   /* W*c > bits: Booth needs the spare top bit */
      const double nb = (double)((size_t)1 << (cc - 1));
      const int ch = nthreads == 1 ? 1 : (4 * nthreads + ww - 1) / ww;
      double cost = (double)ww * ((double)n + 3.0 * nb * ch);       /* bucket reduction: 2 full additions ~ 3 mixed per bucket */
      if (nb * sizeof(EC(xyzz)) > (1 << 20)) cost *= 1.25;          /* L2 misses on every bucket touch */
      if (cost < best) { best = cost; c = cc; }
    }
    if (n < 32) c = 3;
  }
  const int W = (SF_BITS + c) / c;
  int chunks = nthreads == 1 ? 1 : (4 * nthreads + W - 1) / W;
  if ((size_t)chunks * 256 > n) chunks = (int)(n / 256);
  if (chunks < 1) chunks = 1;
  const size_t per = (n + chunks - 1) / chunks;
  const size_t nb = (size_t)1 << (c - 1);
  EC(jac)* wsum = (EC(jac)*)malloc(sizeof(EC(jac)) * (size_t)W * chunks);
  const int ntasks = W * chunks;
  const double t_1 = omp_get_wtime();
#pragma omp parallel num_threads(nthreads)
  {
    EC(xyzz)* buckets = (EC(xyzz)*)malloc(sizeof(EC(xyzz)) * nb); /* private, first-touched here */
#pragma omp for schedule(dynamic, 1)
    for (int task = 0; task < ntasks; task++) {
      const int w = task / chunks, ch = task % chunks;
      size_t lo = (size_t)ch * per, hi = lo + per; if (hi > n) hi = n; if (lo > hi) lo = hi;
      for (size_t b = 0; b < nb; b++) FE(set_zero)(&buckets[b].zz);
      const size_t PF = 8;
      for (size_t i = lo; i < hi; i++) {
        if (i + PF < hi) {
          const int32_t dn = EC(booth)(sc + (i + PF) * SF_N, w, c);
          if (dn) __builtin_prefetch(&buckets[(dn < 0 ? -dn : dn) - 1], 1, 1);
          __builtin_prefetch(&points[i + PF], 0, 0);
        }
        const int32_t d = EC(booth)(sc + i * SF_N, w, c);
        if (d > 0) EC(xyzz_madd)(&buckets[d - 1], &points[i], 0);
        else if (d < 0) EC(xyzz_madd)(&buckets[-d - 1], &points[i], 1);
      }
      EC(jac) running, acc, bj; EC(jac_set_inf)(&running); EC(jac_set_inf)(&acc);
      for (size_t b = nb; b-- > 0;) {
        if (!EC(xyzz_is_inf)(&buckets[b])) { EC(xyzz_to_jac)(&bj, &buckets[b]); EC(jac_add)(&running, &running, &bj); }
        EC(jac_add)(&acc, &acc, &running);
      }
      wsum[task] = acc;
    }
    free(buckets);
  }
  const double t_2 = omp_get_wtime();
  for (int w = W - 1; w >= 0; w--) {
    for (int k = 0; k < c; k++) EC(jac_dbl)(&total, &total);
    for (int ch = 0; ch < chunks; ch++) EC(jac_add)(&total, &total, &wsum[w * chunks + ch]);
  }
  EC(jac_to_aff)(out, &total);
  if (stage_s) { stage_s[0] = t_1 - t_0; stage_s[1] = t_2 - t_1; stage_s[2] = omp_get_wtime() - t_2; stage_s[3] = (double)c; stage_s[4] = (double)W; stage_s[5] = (double)chunks; }
  free(wsum); free(sc_own);
}

/* k * P for a 256-bit k (4 little-endian limbs) */
static void EC(mul_u256)(EC(jac)* r, const EC(aff)* p, const uint64_t k[4]) {
  EC(jac) acc; EC(jac_set_inf)(&acc);
  for (int l = 3; l >= 0; l--)
    for (int b = 63; b >= 0; b--) {
      EC(jac_dbl)(&acc, &acc);
      if ((k[l] >> b) & 1) EC(jac_add_mixed)(&acc, &acc, p);
    }
  *r = acc;
}
/* out[i] = k_i * gen for canonical 256-bit scalars k_i (the query points of a synthetic Groth16 key with known toxic waste: the
 * restated trusted setup of oracle/groth16.py) */
static void EC(mul_batch)(EC(aff)* out, const EC(aff)* gen, const uint64_t* scalars, size_t n, int nthreads) {
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
  for (size_t i = 0; i < n; i++) {
    const uint64_t* k = scalars + 4 * i;
    if ((k[0] | k[1] | k[2] | k[3]) == 0) { FE(set_zero)(&out[i].x); FE(set_zero)(&out[i].y); continue; }
    EC(jac) j; EC(mul_u256)(&j, gen, k);
    EC(jac_to_aff)(&out[i], &j);
  }
}
/* Full-range bases for direct (non closed-form) parity at BASELINE sizes: bases[i] = k_i * G with a 253-bit k_i drawn from
 * four splitmix64 outputs -- generic points with full-width coordinates, every fourth-thousandth one the point at infinity
 * (the zkey queries contain such points). SURVEY 8d config 2, family (i) in spirit: no structure the MSM could exploit. */
static void EC(gen_bases_wide)(EC(aff)* out, const EC(aff)* gen, uint64_t seed, size_t n, int nthreads) {
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
  for (size_t i = 0; i < n; i++) {
    uint64_t k[4];
    for (int l = 0; l < 4; l++) {
      uint64_t x = seed + 4 * i + l + 0x9E3779B97F4A7C15ull;
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
      k[l] = x ^ (x >> 31);
    }
    k[3] >>= 3;
    if ((k[0] & 4095) == 0) { FE(set_zero)(&out[i].x); FE(set_zero)(&out[i].y); continue; }
    EC(jac) j; EC(mul_u256)(&j, gen, k);
    EC(jac_to_aff)(&out[i], &j);
  }
}

/* Full-range bases, cheap enough for 2^20 points on G2: blocks of PROG_BLK consecutive points S_b + j D_b with S_b = s_b G and
 * D_b = d_b G for 253-bit s_b, d_b drawn per block (so the discrete logs are s_b + j d_b: full width, a different progression per
 * block, nothing an MSM could exploit) -- one mixed Jacobian addition per point instead of a 253-bit scalar multiplication, and
 * one field inversion per block (Montgomery's trick) for the affine forms. */
#define PROG_BLK 256
static void EC(gen_bases_progression)(EC(aff)* out, const EC(aff)* gen, uint64_t seed, size_t n, int nthreads) {
  const size_t nblk = (n + PROG_BLK - 1) / PROG_BLK;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
  for (size_t b = 0; b < nblk; b++) {
    uint64_t k[8];
    for (int l = 0; l < 8; l++) {
      uint64_t x = seed + 8 * b + l + 0x9E3779B97F4A7C15ull;
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
      k[l] = x ^ (x >> 31);
    }
    k[3] >>= 3;
    k[7] >>= 3;
    k[4] |= 1;
    EC(jac) s, d; EC(aff) da;
    EC(mul_u256)(&s, gen, k);
    EC(mul_u256)(&d, gen, k + 4);
    EC(jac_to_aff)(&da, &d);
    const size_t lo = b * PROG_BLK, cnt = (lo + PROG_BLK <= n ? PROG_BLK : n - lo);
    EC(jac) pts[PROG_BLK];
    FE(t) pref[PROG_BLK], inv, t;
    for (size_t j = 0; j < cnt; j++) {     /* pts[j] = S + j D */
      pts[j] = s;
      EC(jac) nx; EC(jac_add_mixed)(&nx, &s, &da); s = nx;
    }
    /* batch inversion of the z coordinates (a point at infinity -- z = 0 -- takes no part) */
    FE(set_one)(&t);
    for (size_t j = 0; j < cnt; j++) { pref[j] = t; if (!EC(jac_is_inf)(&pts[j])) FE(mul)(&t, &t, &pts[j].z); }
    FE(inv)(&inv, &t);
    for (size_t j = cnt; j-- > 0;) {
      if (EC(jac_is_inf)(&pts[j])) { FE(set_zero)(&out[lo + j].x); FE(set_zero)(&out[lo + j].y); continue; }
      FE(t) zi, zi2, zi3;
      FE(mul)(&zi, &inv, &pref[j]);
      FE(mul)(&inv, &inv, &pts[j].z);
      FE(sqr)(&zi2, &zi); FE(mul)(&zi3, &zi2, &zi);
      FE(mul)(&out[lo + j].x, &pts[j].x, &zi2);
      FE(mul)(&out[lo + j].y, &pts[j].y, &zi3);
    }
  }
}

/* bases[i] = (splitmix64(seed+i)|1) * G (the product's csh_util_generate_bases_dev family) */
static void EC(gen_bases)(EC(aff)* out, const EC(aff)* gen, uint64_t seed, size_t n, int nthreads) {
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (size_t i = 0; i < n; i++) {
    uint64_t x = seed + i + 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x = (x ^ (x >> 31)) | 1ull;
    EC(jac) j; EC(mul_u64)(&j, gen, x);
    EC(jac_to_aff)(&out[i], &j);
  }
}
