/*
 * pynqs_oracle.c -- CPU restatement of the PyNQS determinant hot path (see pynqs_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP product, never the product.
 * Parity status: PINNED against the compiled reference (tests/golden/, oracle/_ref/).
 *
 * Plain C99 + OpenMP over walkers.  The floating-point operation ORDER of the reference is kept
 * (sequential accumulation, same visiting order of occupied orbitals), so f64 and f32 results
 * are bit-identical to the reference CPU path, not merely close.
 */
#include "pynqs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_SORB 192 /* 3 words; cpp_src/common/default.h:3 allows MAX_SORB_LEN 1..3 */

static inline int words_of(int sorb) { return (sorb - 1) / 64 + 1; }
static inline int bit_test(const uint64_t *w, int n) { return (int)((w[n >> 6] >> (n & 63)) & 1u); }
static inline void bit_toggle(uint64_t *w, int n) { w[n >> 6] ^= (1ULL << (n & 63)); }

/* (-1)^(number of set bits strictly below position n).  cpp_src/cpu/onstate.cpp:22-32 */
static inline int sign_below(const uint64_t *w, int n) {
  int par = 0;
  int full = n >> 6, rem = n & 63;
  for (int k = 0; k < full; ++k) par ^= __builtin_parityll(w[k]);
  if (rem) par ^= __builtin_parityll(w[full] & ((1ULL << rem) - 1ULL));
  return par ? -1 : 1;
}

/* cpp_src/cpu/excitation.cpp:8-16 */
int64_t orc_num_sd(int sorb, int noA, int noB) {
  int k = sorb / 2;
  int nvA = k - noA, nvB = k - noB;
  /* the reference evaluates this in 32-bit int; the largest supported case (sorb 192) fits */
  int nSa = noA * nvA, nSb = noB * nvB;
  int nDaa = noA * (noA - 1) * nvA * (nvA - 1) / 4;
  int nDbb = noB * (noB - 1) * nvB * (nvB - 1) / 4;
  int nDab = noA * noB * nvA * nvB;
  return (int64_t)(nSa + nSb + nDaa + nDbb + nDab);
}

/* cpp_src/cpu/onstate.cpp:147-193.  One pass over occupied bits then one over empty bits, the
 * alpha (even orbital) and beta (odd orbital) slot counters running through both passes. */
void orc_merged(const uint64_t *bra, int len, int sorb, int32_t *merged) {
  int na = 0, nb = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int w = 0; w < len; ++w) {
      uint64_t bits = pass == 0 ? bra[w] : ~bra[w];
      if (pass == 1 && w == len - 1) {
        int tail = sorb % 64 == 0 ? 64 : sorb % 64;
        bits &= (tail == 64) ? ~0ULL : ((1ULL << tail) - 1ULL);
      }
      while (bits) {
        int b = __builtin_ctzll(bits);
        int orb = w * 64 + b;
        int slot;
        if (orb & 1) { slot = 2 * nb + 1; ++nb; }
        else         { slot = 2 * na;     ++na; }
        merged[slot] = orb;
        bits &= bits - 1;
      }
    }
  }
}

/* cpp_src/cpu/excitation.h:6-11 : triangular pair rank -> (hi > lo), via double sqrt as the reference. */
static inline void pair_unrank(int q, int *hi, int *lo) {
  int i = (int)(sqrt((double)((q + 1) * 2)) + 0.5);
  *hi = i;
  *lo = q - i * (i - 1) / 2;
}

/* cpp_src/cpu/excitation.cpp:18-110.  Block order [Sa, Sb, Daa, Dbb, Dab]; note the same-spin
 * blocks take the hole pair from the GLOBAL rank modulo the pair count (:63, :79). */
void orc_unpack_sd(int sorb, int noA, int noB, int idx, int32_t out[5]) {
  int k = sorb / 2;
  int nvA = k - noA, nvB = k - noB;
  int noAA = noA * (noA - 1) / 2, noBB = noB * (noB - 1) / 2;
  int nvAA = nvA * (nvA - 1) / 2, nvBB = nvB * (nvB - 1) / 2;
  int e0 = noA * nvA;
  int e1 = e0 + noB * nvB;
  int e2 = e1 + noAA * nvAA;
  int e3 = e2 + noBB * nvBB;
  int i = -1, a = -1, j = -1, b = -1, dbl = 0;
  if (idx < e0) {
    i = 2 * (idx % noA);
    a = 2 * (idx / noA + noA);
    j = b = 0;
  } else if (idx < e1) {
    int t = idx - e0;
    i = 2 * (t % noB) + 1;
    a = 2 * (t / noB + noB) + 1;
    j = b = 0;
  } else if (idx < e2) {
    int t = idx - e1, h1, h0, v1, v0;
    pair_unrank(idx % noAA, &h1, &h0);
    pair_unrank(t / noAA, &v1, &v0);
    i = 2 * h1; j = 2 * h0;
    a = 2 * (v1 + noA); b = 2 * (v0 + noA);
    dbl = 1;
  } else if (idx < e3) {
    int t = idx - e2, h1, h0, v1, v0;
    pair_unrank(idx % noBB, &h1, &h0);
    pair_unrank(t / noBB, &v1, &v0);
    i = 2 * h1 + 1; j = 2 * h0 + 1;
    a = 2 * (v1 + noB) + 1; b = 2 * (v0 + noB) + 1;
    dbl = 1;
  } else {
    int t = idx - e3;
    int ia = t % (noA * nvA), jb = t / (noA * nvA);
    i = 2 * (ia % noA);
    a = 2 * (ia / noA + noA);
    j = 2 * (jb % noB) + 1;
    b = 2 * (jb / noB + noB) + 1;
    dbl = 1;
  }
  out[0] = i; out[1] = a; out[2] = j; out[3] = b; out[4] = dbl;
}

/* cpp_src/tensor/cpu_tensor.cpp:164-218 + cpp_src/cpu/excitation.cpp:112-121,171-181 */
int orc_comb(const uint64_t *bra, int64_t n, int sorb, int noA, int noB, uint64_t *comb,
             double *comb_pm1, int nthreads) {
  if (sorb < 1 || sorb > ORC_MAX_SORB) return -1;
  const int len = words_of(sorb);
  const int64_t ncomb = orc_num_sd(sorb, noA, noB) + 1;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
  for (int64_t w = 0; w < n; ++w) {
    int32_t merged[ORC_MAX_SORB];
    const uint64_t *x = bra + w * len;
    orc_merged(x, len, sorb, merged);
    for (int64_t r = 0; r < ncomb; ++r) {
      uint64_t *out = comb + (w * ncomb + r) * len;
      for (int k = 0; k < len; ++k) out[k] = x[k];
      double *pm = comb_pm1 ? comb_pm1 + (w * ncomb + r) * (int64_t)sorb : 0;
      if (pm) for (int o = 0; o < sorb; ++o) pm[o] = bit_test(x, o) ? 1.0 : -1.0;
      if (r == 0) continue;
      int32_t s[5];
      orc_unpack_sd(sorb, noA, noB, (int)(r - 1), s);
      for (int t = 0; t < 4; ++t) {
        int orb = merged[s[t]];
        bit_toggle(out, orb);
        if (pm) pm[orb] *= -1.0;
      }
    }
  }
  return 0;
}

/* cpp_src/cpu/onstate.h:45-63 */
void orc_onv_to_pm1_f64(const uint64_t *bra, int64_t n, int sorb, double *out) {
  const int len = words_of(sorb);
  for (int64_t w = 0; w < n; ++w)
    for (int o = 0; o < sorb; ++o) out[w * sorb + o] = bit_test(bra + w * len, o) ? 1.0 : -1.0;
}
void orc_onv_to_pm1_f32(const uint64_t *bra, int64_t n, int sorb, float *out) {
  const int len = words_of(sorb);
  for (int64_t w = 0; w < n; ++w)
    for (int o = 0; o < sorb; ++o) out[w * sorb + o] = bit_test(bra + w * len, o) ? 1.0f : -1.0f;
}

/* cpp_src/tensor/cpu_tensor.cpp:8-44 : a byte sets its bit only when it is exactly 1. */
void orc_pm01_to_onv(const uint8_t *occ, int64_t n, int sorb, uint64_t *out) {
  const int len = words_of(sorb);
  memset(out, 0, (size_t)n * len * sizeof(uint64_t));
  for (int64_t w = 0; w < n; ++w)
    for (int o = 0; o < sorb; ++o)
      if (occ[w * sorb + o] == 1) bit_toggle(out + w * len, o);
}

/* cpp_src/tensor/integral.cpp:6-60.  All (i,j,k,l) are visited in lexicographic order and the last
 * writer of a packed slot wins, exactly as in the reference. */
void orc_compress_h1e_h2e(const double *h1e2d, const double *h2e4d, int sorb, double *h1e, double *h2e) {
  const int64_t s = sorb;
  const int64_t pair = s * (s - 1) / 2;
  memcpy(h1e, h1e2d, (size_t)(s * s) * sizeof(double));
  memset(h2e, 0, (size_t)(pair * (pair + 1) / 2) * sizeof(double));
  for (int64_t i = 0; i < s; ++i)
    for (int64_t j = 0; j < s; ++j) {
      if (i == j) continue;
      int64_t ij = i > j ? i * (i - 1) / 2 + j : j * (j - 1) / 2 + i;
      double sij = i > j ? 1.0 : -1.0;
      for (int64_t k = 0; k < s; ++k)
        for (int64_t l = 0; l < s; ++l) {
          if (k == l) continue;
          int64_t kl = k > l ? k * (k - 1) / 2 + l : l * (l - 1) / 2 + k;
          double sg = k > l ? sij : -sij;
          int64_t P = ij >= kl ? ij : kl, Q = ij >= kl ? kl : ij;
          h2e[P * (P + 1) / 2 + Q] = sg * h2e4d[((i * s + j) * s + k) * s + l];
        }
    }
}

/* cpp_src/tensor/integral.cpp:62-125 */
void orc_decompress_h1e_h2e(const double *h1e, const double *h2e, int sorb, double *h1e2d, double *h2e4d) {
  const int64_t s = sorb;
  memcpy(h1e2d, h1e, (size_t)(s * s) * sizeof(double));
  memset(h2e4d, 0, (size_t)(s * s * s * s) * sizeof(double));
  for (int64_t i = 0; i < s; ++i)
    for (int64_t j = 0; j < s; ++j) {
      if (i == j) continue;
      int64_t ij = i > j ? i * (i - 1) / 2 + j : j * (j - 1) / 2 + i;
      double sij = i > j ? 1.0 : -1.0;
      for (int64_t k = 0; k < s; ++k)
        for (int64_t l = 0; l < s; ++l) {
          if (k == l) continue;
          int64_t kl = k > l ? k * (k - 1) / 2 + l : l * (l - 1) / 2 + k;
          double sg = k > l ? sij : -sij;
          int64_t P = ij >= kl ? ij : kl, Q = ij >= kl ? kl : ij;
          h2e4d[((i * s + j) * s + k) * s + l] = h2e[P * (P + 1) / 2 + Q] * sg;
        }
    }
}

/* cpp_src/tensor/cpu_tensor.cpp:589-688 (little_endian=True branch: compare the last word first). */
static inline int key_cmp(const uint64_t *a, const uint64_t *b, int len) {
  for (int k = len - 1; k >= 0; --k) {
    if (a[k] < b[k]) return -1;
    if (a[k] > b[k]) return 1;
  }
  return 0;
}
static inline int64_t key_search(const uint64_t *keys, int64_t nkeys, const uint64_t *q, int len) {
  int64_t lo = 0, hi = nkeys - 1;
  while (lo <= hi) {
    int64_t mid = lo + (hi - lo) / 2;
    int c = key_cmp(keys + mid * len, q, len);
    if (c == 0) return mid;
    if (c < 0) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}
void orc_wavefunction_lut(const uint64_t *keys, int64_t nkeys, const uint64_t *onv, int64_t n, int len,
                          int64_t *idx, uint8_t *mask) {
  for (int64_t i = 0; i < n; ++i) {
    int64_t p = key_search(keys, nkeys, onv + i * len, len);
    idx[i] = p;
    mask[i] = p >= 0;
  }
}

/* ---- dtype-generic part: instantiated for double and float ------------------------------------ */
#define REAL double
#define SUF(name) name##_f64
#include "pynqs_oracle_tmpl.inc"
#undef REAL
#undef SUF
#define REAL float
#define SUF(name) name##_f32
#include "pynqs_oracle_tmpl.inc"
#undef REAL
#undef SUF

/* vmc/ansatz/rbm/rbm.py:186-211, rbm_type "real": ax = exp(x . a); amp = prod_h 2 cosh(W x + b). */
static double rbm_real_psi_one(const uint64_t *x, int sorb, int nhid, const double *W, const double *hb,
                               const double *vb) {
  double ax = 0.0;
  for (int o = 0; o < sorb; ++o) ax += (bit_test(x, o) ? 1.0 : -1.0) * vb[o];
  double amp = 1.0;
  for (int h = 0; h < nhid; ++h) {
    double th = 0.0;
    const double *Wh = W + (int64_t)h * sorb;
    for (int o = 0; o < sorb; ++o) th += (bit_test(x, o) ? 1.0 : -1.0) * Wh[o];
    th += hb[h];
    amp *= 2.0 * cosh(th);
  }
  return exp(ax) * amp;
}

void orc_rbm_real_psi(const uint64_t *onv, int64_t n, int sorb, int nhid, const double *W, const double *hb,
                      const double *vb, double *psi) {
  const int len = words_of(sorb);
  for (int64_t i = 0; i < n; ++i) psi[i] = rbm_real_psi_one(onv + i * len, sorb, nhid, W, hb, vb);
}

/* vmc/energy/eloc.py:134-203 : eloc = sum_k (psi_k / psi_0) * H_k, over the full S+D list. */
int orc_eloc_simple_rbm(const uint64_t *bra, int64_t n, int sorb, int nele, int noA, int noB,
                        const double *h1e, const double *h2e, int nhid, const double *W, const double *hb,
                        const double *vb, double *eloc, double *psi0, int nthreads) {
  if (sorb < 1 || sorb > ORC_MAX_SORB) return -1;
  const int len = words_of(sorb);
  const int64_t ncomb = orc_num_sd(sorb, noA, noB) + 1;
  int rc = 0;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
  for (int64_t w = 0; w < n; ++w) {
    uint64_t *comb = (uint64_t *)malloc((size_t)ncomb * len * sizeof(uint64_t));
    double *hm = (double *)malloc((size_t)ncomb * sizeof(double));
    if (!comb || !hm) { rc = -2; free(comb); free(hm); continue; }
    orc_comb_hij_fused_f64(bra + w * len, 1, sorb, nele, noA, noB, h1e, h2e, comb, hm, 1);
    double p0 = rbm_real_psi_one(comb, sorb, nhid, W, hb, vb);
    double acc = 0.0;
    for (int64_t k = 0; k < ncomb; ++k) {
      double pk = k == 0 ? p0 : rbm_real_psi_one(comb + k * len, sorb, nhid, W, hb, vb);
      acc += (pk / p0) * hm[k];
    }
    eloc[w] = acc;
    psi0[w] = p0;
    free(comb); free(hm);
  }
  return rc;
}

/* vmc/energy/eloc.py:326-401 : psi(x') looked up in the sorted sample table, zero when absent. */
int orc_eloc_sample_space(const uint64_t *bra, int64_t n, int sorb, int nele, int noA, int noB,
                          const double *h1e, const double *h2e, const uint64_t *keys, int64_t nkeys,
                          const double *wf, int wf_is_complex, double *eloc, double *psi0, int nthreads) {
  if (sorb < 1 || sorb > ORC_MAX_SORB) return -1;
  const int len = words_of(sorb);
  const int64_t ncomb = orc_num_sd(sorb, noA, noB) + 1;
  const int cs = wf_is_complex ? 2 : 1;
  int rc = 0;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
  for (int64_t w = 0; w < n; ++w) {
    uint64_t *comb = (uint64_t *)malloc((size_t)ncomb * len * sizeof(uint64_t));
    double *hm = (double *)malloc((size_t)ncomb * sizeof(double));
    if (!comb || !hm) { rc = -2; free(comb); free(hm); continue; }
    orc_comb_hij_fused_f64(bra + w * len, 1, sorb, nele, noA, noB, h1e, h2e, comb, hm, 1);
    /* row 0 of comb is x itself: psi(x) */
    int64_t pz = key_search(keys, nkeys, comb, len);
    double p0r = pz >= 0 ? wf[pz * cs] : 0.0;
    double p0i = (pz >= 0 && wf_is_complex) ? wf[pz * cs + 1] : 0.0;
    double d = p0r * p0r + p0i * p0i;
    double re = 0.0, im = 0.0;
    for (int64_t k = 0; k < ncomb; ++k) {
      int64_t p = key_search(keys, nkeys, comb + k * len, len);
      if (p < 0) continue; /* psi = 0 outside the sample space */
      double vr = wf[p * cs];
      double vi = wf_is_complex ? wf[p * cs + 1] : 0.0;
      if (wf_is_complex) {
        /* (vr + i vi) / (p0r + i p0i) * H */
        re += ((vr * p0r + vi * p0i) / d) * hm[k];
        im += ((vi * p0r - vr * p0i) / d) * hm[k];
      } else {
        re += (vr / p0r) * hm[k];
      }
    }
    if (wf_is_complex) {
      eloc[2 * w] = re; eloc[2 * w + 1] = im;
      psi0[2 * w] = p0r; psi0[2 * w + 1] = p0i;
    } else {
      eloc[w] = re;
      psi0[w] = p0r;
    }
    free(comb); free(hm);
  }
  return rc;
}
