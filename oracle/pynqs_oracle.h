/*
 * pynqs_oracle.h -- CPU restatement of the PyNQS determinant hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the *checker* for the HIP product in
 * pynqs_amd/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Nothing under pynqs_amd/ imports, links or executes it.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit (comb, merged, unpack
 * tables, Hmat in f64 and f32) against the compiled reference C_extension
 * (cpp_src/{common,cpu,tensor}, MAX_SORB_LEN = 1, 2, 3) through the golden vectors in
 * tests/golden/ (generator: tests/golden/make_golden.py) and, when oracle/_ref/ is built,
 * against the reference module itself (tests/test_oracle_vs_ref.py).
 *
 * Each function cites the reference file:line (relative to /root/reference) it restates.
 */
#ifndef PYNQS_ORACLE_H
#define PYNQS_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cpp_src/cpu/excitation.cpp:8-16 : number of singles+doubles (identity NOT included). */
int64_t orc_num_sd(int sorb, int noA, int noB);

/* cpp_src/cpu/onstate.cpp:147-193 : occupied(abab) then virtual(abab) slot list. */
void orc_merged(const uint64_t *bra, int len, int sorb, int32_t *merged);

/* cpp_src/cpu/excitation.cpp:18-110 : rank -> (i, a, j, b, type) slot indices. */
void orc_unpack_sd(int sorb, int noA, int noB, int idx, int32_t out[5]);

/* cpp_src/tensor/cpu_tensor.cpp:164-218 : enumerate only.  comb[n][ncomb][len] (row 0 = bra);
 * comb_pm1 (nullable) = double[n][ncomb][sorb], the flag_bit=True output. */
int orc_comb(const uint64_t *bra, int64_t n, int sorb, int noA, int noB, uint64_t *comb,
             double *comb_pm1, int nthreads);

/* cpp_src/tensor/cpu_tensor.cpp:220-272 : fused enumerate + <x|H|x'>. */
int orc_comb_hij_fused_f64(const uint64_t *bra, int64_t n, int sorb, int nele, int noA, int noB,
                           const double *h1e, const double *h2e, uint64_t *comb, double *hmat,
                           int nthreads);
int orc_comb_hij_fused_f32(const uint64_t *bra, int64_t n, int sorb, int nele, int noA, int noB,
                           const float *h1e, const float *h2e, uint64_t *comb, float *hmat,
                           int nthreads);

/* cpp_src/tensor/cpu_tensor.cpp:274-325 : generic pairs.  ket_is_3d: ket[n][m][len] else ket[m][len]. */
int orc_hij_f64(const uint64_t *bra, int64_t n, const uint64_t *ket, int64_t m, int ket_is_3d,
                const double *h1e, const double *h2e, int sorb, int nele, double *hmat, int nthreads);
int orc_hij_f32(const uint64_t *bra, int64_t n, const uint64_t *ket, int64_t m, int ket_is_3d,
                const float *h1e, const float *h2e, int sorb, int nele, float *hmat, int nthreads);

/* cpp_src/cpu/onstate.h:45-63, cpp_src/tensor/cpu_tensor.cpp:46-88 : bit -> +1/-1. */
void orc_onv_to_pm1_f64(const uint64_t *bra, int64_t n, int sorb, double *out);
void orc_onv_to_pm1_f32(const uint64_t *bra, int64_t n, int sorb, float *out);

/* cpp_src/tensor/cpu_tensor.cpp:8-44 : 0/1 bytes -> packed words (only bytes == 1 set a bit). */
void orc_pm01_to_onv(const uint8_t *occ, int64_t n, int sorb, uint64_t *out);

/* cpp_src/tensor/integral.cpp:6-60 / :62-125 : integral layout (host). */
void orc_compress_h1e_h2e(const double *h1e2d, const double *h2e4d, int sorb, double *h1e, double *h2e);
void orc_decompress_h1e_h2e(const double *h1e, const double *h2e, int sorb, double *h1e2d, double *h2e4d);

/* cpp_src/tensor/cpu_tensor.cpp:589-688 : sorted multi-word key binary search (little endian:
 * most significant word last).  idx[i] = position or -1, mask[i] = found. */
void orc_wavefunction_lut(const uint64_t *keys, int64_t nkeys, const uint64_t *onv, int64_t n, int len,
                          int64_t *idx, uint8_t *mask);

/* vmc/ansatz/rbm/rbm.py:186-211 ("real" type): psi(x) = exp(a.x) * prod_h 2cosh(W x + b)_h, x = +-1. */
void orc_rbm_real_psi(const uint64_t *onv, int64_t n, int sorb, int nhid, const double *W /*[nhid][sorb]*/,
                      const double *hb, const double *vb, double *psi);

/* vmc/energy/eloc.py:134-203 (_simple) with the RBM above: eloc[i] = sum_k H[i,k] psi(comb[i,k]) / psi(comb[i,0]). */
int orc_eloc_simple_rbm(const uint64_t *bra, int64_t n, int sorb, int nele, int noA, int noB,
                        const double *h1e, const double *h2e, int nhid, const double *W, const double *hb,
                        const double *vb, double *eloc, double *psi0, int nthreads);

/* vmc/energy/eloc.py:326-401 (_only_sample_space): psi(x') from a sorted LUT, 0 if absent.
 * wf is complex when wf_is_complex (interleaved re,im); eloc is written with the same layout. */
int orc_eloc_sample_space(const uint64_t *bra, int64_t n, int sorb, int nele, int noA, int noB,
                          const double *h1e, const double *h2e, const uint64_t *keys, int64_t nkeys,
                          const double *wf, int wf_is_complex, double *eloc, double *psi0, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
