/*
 * pynqs_amd.h -- C ABI of libpynqs_amd.so, the MI355X (gfx950) determinant / local-energy engine.
 *
 * This is the drop-in boundary for PyNQS' `libs.C_extension` hot path.  The reference has no C ABI:
 * its boundary is a pybind11/LibTorch module (cpp_src/tensor/bind.cpp:317-391) whose functions take
 * at::Tensor.  Each entry point below names the reference interface it replaces (file:line relative
 * to the PyNQS tree); the Python shim pynqs_amd/C_extension.py re-creates the reference's Python
 * signatures on top of these calls (see INTEGRATION.md for the binding a maintainer would add).
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer (HIP, same device as `stream`) unless marked [host].
 *  - ONVs are little-endian 64-bit words, `len = (sorb-1)/64 + 1` words per determinant; orbital j is
 *    bit j%64 of word j/64, even j = alpha, odd j = beta (cpp_src/tensor/cpu_tensor.cpp:8-44).
 *    The reference's uint8[n, 8*len] tensors are these words viewed as bytes.
 *  - `dtype`: PYNQS_F32 or PYNQS_F64 = element type of h1e / h2e / hmat (the reference dispatches on
 *    h1e's dtype, cpp_src/tensor/cpu_tensor.cpp:249,298).
 *  - h1e: T[sorb*sorb] row-major; h2e: T[pair*(pair+1)/2], pair = sorb*(sorb-1)/2, antisymmetrised
 *    packed triangle (cpp_src/tensor/integral.cpp:6-60).
 *  - The caller owns all buffers.  Calls enqueue work on `stream` (a hipStream_t; NULL = default
 *    stream) and return without synchronising, like the reference's CUDA path (cuda/kernel.cu:266-277).
 *  - Return value: PYNQS_OK or a negative error code; pynqs_last_error() gives a thread-local message.
 *    No exceptions cross the ABI.  n == 0 is a valid no-op.
 *  - Thread-safe and re-entrant: no global mutable state besides the thread-local error string.
 */
#ifndef PYNQS_AMD_H
#define PYNQS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYNQS_ABI_VERSION 1

#define PYNQS_OK 0
#define PYNQS_EINVAL (-1)  /* bad argument: sorb/nele out of range, null pointer, bad dtype        */
#define PYNQS_ELAUNCH (-2) /* HIP launch failure (message carries hipGetErrorString)               */
#define PYNQS_ELENGTH (-3) /* check_sorb: sorb needs more words than PYNQS_MAX_SORB_LEN            */
#define PYNQS_EOVERFLOW (-4) /* check_sorb: too many electrons / virtual orbitals                 */

#define PYNQS_F32 0
#define PYNQS_F64 1

/* Limits.  The reference fixes MAX_SORB_LEN at compile time (cpp_src/common/default.h:3-10); here the
 * word count is a run-time dispatch over 1..3 and the limits are those of its MAX_SORB_LEN=3 build. */
#define PYNQS_MAX_SORB_LEN 3
#define PYNQS_MAX_SORB 192
#define PYNQS_MAX_NELE 120
#define PYNQS_MAX_NVIR 120

int pynqs_abi_version(void);
const char *pynqs_last_error(void);

/* [host] cpp_src/cpu/excitation.cpp:8-16 (get_Num_SinglesDoubles): singles+doubles, identity excluded. */
int64_t pynqs_num_sd(int sorb, int noA, int noB);

/* [host] cpp_src/tensor/bind.cpp:282-301 (check_sorb): PYNQS_OK, PYNQS_ELENGTH or PYNQS_EOVERFLOW. */
int pynqs_check_sorb(int sorb, int nele);

/* cpp_src/tensor/bind.cpp:239-250 (get_comb_hij_fused) -> cpu_tensor.cpp:220-272 / cuda kernel.cu:224-277.
 * comb [nbatch][ncomb][len] (row 0 = bra itself; may be NULL to skip the write), hmat T[nbatch][ncomb]
 * (column 0 = <x|H|x>), ncomb = pynqs_num_sd()+1. */
int pynqs_comb_hij_fused(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                         const void *h1e, const void *h2e, int dtype, uint64_t *comb, void *hmat,
                         void *stream);

/* cpp_src/tensor/bind.cpp:66-83 (get_comb_tensor) -> cpu_tensor.cpp:164-218.  comb as above;
 * comb_pm1 (may be NULL) = double[nbatch][ncomb][sorb], the flag_bit=True +-1 expansion (the reference
 * hard-codes double there, cpu_tensor.cpp:191-195). */
int pynqs_comb(const uint64_t *bra, int64_t nbatch, int sorb, int noA, int noB, uint64_t *comb,
               double *comb_pm1, void *stream);

/* cpp_src/tensor/bind.cpp:39-64 (get_hij_torch) -> cpu_tensor.cpp:274-325.  ket_is_3d: ket[n][m][len]
 * (local-energy mode) else ket[m][len] (full matrix).  hmat T[n][m]; excitation degree > 2 gives 0. */
int pynqs_hij(const uint64_t *bra, int64_t n, const uint64_t *ket, int64_t m, int ket_is_3d,
              const void *h1e, const void *h2e, int dtype, int sorb, int nele, void *hmat, void *stream);

/* cpp_src/tensor/bind.cpp:24-37 (onv_to_tensor) -> cpu_tensor.cpp:46-88: out T[n][sorb] = +1 / -1. */
int pynqs_onv_to_pm1(const uint64_t *bra, int64_t n, int sorb, int dtype, void *out, void *stream);

/* cpp_src/tensor/bind.cpp:9-22 (tensor_to_onv) -> cpu_tensor.cpp:8-44: occ uint8[n][sorb] (1 = occupied,
 * anything else = empty) -> out uint64[n][len]. */
int pynqs_pm01_to_onv(const uint8_t *occ, int64_t n, int sorb, uint64_t *out, void *stream);

/* cpp_src/tensor/bind.cpp:216-236 (wavefunction_lut, little_endian=True) -> cpu_tensor.cpp:589-688 /
 * cuda kernel.cu:608-680.  keys uint64[nkeys][len] sorted ascending as multi-word integers (most
 * significant word last); idx[i] = position of onv[i] or -1, mask[i] = found. */
int pynqs_wavefunction_lut(const uint64_t *keys, int64_t nkeys, const uint64_t *onv, int64_t n, int sorb,
                           int64_t *idx, uint8_t *mask, void *stream);

/* cpp_src/tensor/bind.cpp:303-314 (spin_flip_rand) -> cpu_tensor.cpp:90-137 / cuda kernel.cu:691-716: one random
 * single/double move per walker: r0 uniform on [0, nsd] (both ends), r0 == 0 keeps the walker, else excitation
 * rank r0-1 is applied.  out may alias bra (in place).  The stream is a counter-based hash of
 * (seed, offset + walker index): pass a different `offset` (e.g. a running sample count) on every call. */
int pynqs_spin_flip_rand(const uint64_t *bra, int64_t n, int sorb, int noA, int noB, uint64_t seed,
                         uint64_t offset, uint64_t *out, void *stream);

/* ---- integral plan: the fast path ---------------------------------------------------------------
 * The reference keeps h2e as one packed triangle over all spin-orbital pairs (integral.cpp:6-60); random
 * gathers from it are bound by the CU's vector L1.  A plan is a spin-blocked dense re-layout of the SAME
 * values (pynqs_amd/csrc/plan.h) in caller-owned device memory; it is built once per (h1e, h2e) and then
 * replaces the two pointers.  Needs an even sorb.  Results are bit-identical to the direct entry points.
 *   pynqs_plan_bytes : [host] size of the plan buffer in bytes, or -1 if unsupported
 *   pynqs_plan_build : fill `plan` from the reference-layout h1e / h2e (one kernel, no sync)          */
int64_t pynqs_plan_bytes(int sorb, int dtype);
int pynqs_plan_build(const void *h1e, const void *h2e, int sorb, int dtype, void *plan, void *stream);

/* get_comb_hij_fused (bind.cpp:239-250) on a plan.  Same outputs as pynqs_comb_hij_fused. */
int pynqs_comb_hij_fused_plan(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                              const void *plan, int dtype, uint64_t *comb, void *hmat, void *stream);

/* ---- fused local energy (no comb / Hmat materialisation), on a plan, f64 ---------------------------
 * SAMPLE_SPACE method: vmc/energy/eloc.py:326-401 (_only_sample_space) = get_comb_hij_fused +
 * WavefunctionLUT.lookup (utils/public_function.py:817-838, cpu_tensor.cpp:589-688) + the contraction
 * eloc.py:395-396, in one pass.  keys uint64[nkeys][len] sorted as for pynqs_wavefunction_lut; wf is
 * double[nkeys] or interleaved complex double[nkeys][2]; eloc / psi0 have the same element type ([nbatch] or
 * [nbatch][2]).  psi(x') = 0 for x' outside the table; psi0 = psi(x) (0 if x itself is not in the table,
 * eloc is then inf/nan exactly like the reference's division).  The sum is taken as (sum_k H_k psi_k) / psi_0. */
int pynqs_eloc_sample_space(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                            const void *plan, const uint64_t *keys, int64_t nkeys, const double *wf,
                            int wf_is_complex, double *eloc, double *psi0, void *stream);

/* ---- hash table over the sorted sample keys (the reference's optional GPU table: cuda/hashTable.cu,
 * cuda_tensor.cpp:489-559 hash_build / hash_lookup, off by default: utils/public_function.py:23 USE_HASH).
 * Open addressing in caller-owned memory; values are the positions in the SORTED key array, so a lookup returns
 * exactly what pynqs_wavefunction_lut returns.
 *   pynqs_hash_bytes  : [host] table size in bytes for nkeys keys
 *   pynqs_hash_build  : fill `table` from keys uint64[nkeys][len] (any order, distinct)
 *   pynqs_hash_lookup : idx[i] = position of onv[i] in the key array or -1, mask[i] = found
 *   pynqs_eloc_sample_space_hash : pynqs_eloc_sample_space with the table instead of the sorted keys        */
int64_t pynqs_hash_bytes(int64_t nkeys, int sorb);
int pynqs_hash_build(const uint64_t *keys, int64_t nkeys, int sorb, void *table, void *stream);
int pynqs_hash_lookup(const void *table, int64_t nkeys, const uint64_t *onv, int64_t n, int sorb, int64_t *idx,
                      uint8_t *mask, void *stream);
int pynqs_eloc_sample_space_hash(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                                 const void *plan, const void *table, int64_t nkeys, const double *wf,
                                 int wf_is_complex, double *eloc, double *psi0, void *stream);

/* ---- spin-flip-projected SAMPLE_SPACE: vmc/energy/flip.py:322-418 (_only_sample_space_flip).
 *   E_loc(x) = [ sum_k H_k psi(x'_k) + eta * sum_k H_k eta_m(x'_k) psi(flip(x'_k)) ] / (extra_norm^2 psi(x))
 * with flip = alpha <-> beta exchange and eta_m = (-1)^(doubly occupied orbitals of x') (utils/public_function.py:966-1007).
 * The first sum / psi(x) is pynqs_eloc_sample_space[_hash]; these entries return the second one:
 *   out[x] = sum_k H_k eta_m(x'_k) psi(flip(x'_k)) / psi0[x],   psi0 = psi(x) from that call (an INPUT here),
 * one more pass of the same kernel (the filters are asked with the partner orbitals' Zobrist values).              */
int pynqs_eloc_sample_space_flip(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                                 const void *plan, const uint64_t *keys, int64_t nkeys, const double *wf,
                                 int wf_is_complex, const double *psi0, double *out, void *stream);
int pynqs_eloc_sample_space_hash_flip(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                                      const void *plan, const void *table, int64_t nkeys, const double *wf,
                                      int wf_is_complex, const double *psi0, double *out, void *stream);

/* ---- SAMPLE_SPACE local energy, key-major (kernels_eloc_keys.hip): the same sum as pynqs_eloc_sample_space -- vmc/energy/eloc.py:326-401,
 * psi(x') from the table of the sample space, 0 outside it -- computed by walking the TABLE: every key within a double excitation of
 * the walker (popcount(x ^ key) <= 4) contributes <x|H|key> psi(key), evaluated from the two bit patterns.  Work per walker is nkeys
 * instead of ncomb: the entry for large orbital spaces (ncomb ~ sorb^4) and for small tables.
 *   keys uint64[nkeys][len] in ANY order (no sorting, no hash table), nkeys < 2^27; wf double[nkeys] or complex double[nkeys][2].
 *   flip = 0: eloc[x] = sum / psi(x), psi0[x] (OUTPUT) = psi(x) = the table value of the key equal to x, 0 if there is none;
 *   flip = 1: the projected form's partner sum (flip.py:322-418): eloc[x] = sum_x' <x|H|x'> eta_m(x') psi(flip x') / psi0[x], psi0 INPUT. */
int pynqs_eloc_sample_space_keys(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                                 const uint64_t *keys, int64_t nkeys, const double *wf, int wf_is_complex, int flip, double *eloc,
                                 double *psi0, void *stream);

/* ---- the same behind a BLOCK INDEX of the keys (round 3; kernels_keys_index.hip, kernels_eloc_keys.hip INDEXED): the sorb bits are cut
 * into 5 blocks; a determinant within a double excitation of x agrees with x in at least one whole block, so only the keys that share a
 * block value with the walker are loaded and compared (binary searches in the per-block sorted lists), each counted in the first block it
 * agrees in.  Work per walker ~ the number of such keys (tens for a table of samples at sorb >= 56) instead of nkeys.
 *   pynqs_keys_index_bytes     : [host] size of the index (60 bytes per key), -1 on bad arguments (sorb even, nkeys < 2^27)
 *   pynqs_keys_index_workspace : [host] scratch bytes for the build
 *   pynqs_keys_index_build     : index <- keys uint64[nkeys][len] (any order, distinct); five stable radix sorts: the index, and
 *                                with it the order of the kernel's additions, is a function of the key array alone
 *   pynqs_keys_index_density   : *sum_sq (device, uint64) = sum over the runs of equal block values of length^2: a key of the table used
 *                                as a walker meets sum_sq / nkeys keys through the index (streamed: nkeys).  CAS-like tables share blocks
 *                                among thousands of keys (Fe2S2: 21.5e3 per walker of 18.5e3 keys) -- the streamed or the column-major
 *                                form is the one to use there; tables of samples at sorb >= 56 do not (13 of 65.5e3 at sorb 120 / 184)
 *   pynqs_eloc_sample_space_indexed : pynqs_eloc_sample_space_keys with the index (same arguments, same results up to the order
 *                                of the additions; no atomics: bit-reproducible)                                                          */
int64_t pynqs_keys_index_bytes(int64_t nkeys, int sorb);
int64_t pynqs_keys_index_workspace(int64_t nkeys, int sorb);
int pynqs_keys_index_build(const uint64_t *keys, int64_t nkeys, int sorb, void *index, void *workspace, void *stream);
int pynqs_keys_index_density(const void *index, int64_t nkeys, int sorb, uint64_t *sum_sq, void *stream);
int pynqs_eloc_sample_space_indexed(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                                    const uint64_t *keys, int64_t nkeys, const void *index, const double *wf, int wf_is_complex,
                                    int flip, double *eloc, double *psi0, void *stream);

/* ---- duplicates among determinants, without a sort (`Func`, vmc/energy/flip.py:44-50: torch.unique(dim=0,
 * return_inverse=True) on the x' that reach the ansatz).  first[i] = smallest j with onv[j] == onv[i] (int32[n]);
 * the rows with first[i] == i are the distinct determinants in order of first appearance.  Deterministic.
 *   pynqs_unique_workspace : [host] bytes of scratch for n rows (-1: n out of range, n < 2^30)             */
int64_t pynqs_unique_workspace(int64_t n);
int pynqs_unique_first(const uint64_t *onv, int64_t n, int sorb, void *workspace, int32_t *first, void *stream);

/* ---- SIMPLE method with a real RBM amplitude, fully fused, f64 -------------------------------------------
 * vmc/energy/eloc.py:121-203 (_simple: get_comb_hij_fused + ansatz on all nbatch*ncomb kets + contraction) for
 * the reference's RBMWavefunction with rbm_type "real" (vmc/ansatz/rbm/rbm.py:186-211):
 *   psi(x) = exp(visible_bias . x) * prod_h 2 cosh(hidden_bias[h] + sum_o weights[h][o] x_o),  x_o = +1 / -1.
 * The parameters are re-laid out once per parameter update into a caller-owned "RBM table" (pynqs_amd/csrc/rbm.h):
 *   pynqs_eloc_rbm_supported : [host] 1 if exp(+-4 W) of all (orbital, hidden unit) pairs of this problem fits the
 *                           CU's LDS next to the walker tables (the kernel's working set), else 0
 *   pynqs_rbm_table_bytes : [host] size of the table in bytes, or -1 for bad sizes
 *   pynqs_rbm_table_build : weights double[nhidden][sorb] (row-major, the reference's parameter shape),
 *                           hidden_bias double[nhidden], visible_bias double[sorb] or NULL (= 0)
 *   pynqs_eloc_rbm        : eloc[nbatch] = sum_x' <x|H|x'> psi(x')/psi(x) over x' = x and all singles/doubles;
 *                           psi (may be NULL) receives psi(x).  The amplitude ratios are evaluated from the
 *                           2 or 4 flipped orbitals (no overflow for any theta); results agree with the
 *                           materialised path to rounding (tests: 1e-8 Ha). */
int pynqs_eloc_rbm_supported(int sorb, int nele, int noA, int noB, int nhidden);
int64_t pynqs_rbm_table_bytes(int sorb, int nhidden);
int pynqs_rbm_table_build(const double *weights, const double *hidden_bias, const double *visible_bias, int sorb,
                          int nhidden, void *table, void *stream);
int pynqs_eloc_rbm(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                   const void *rbm_table, int nhidden, double *eloc, double *psi, void *stream);
/* The reference's other RBM amplitudes that share the hidden-unit product (rbm.py:199-211), from the same table:
 *   PYNQS_RBM_REAL  psi = exp(a.x)  prod_h 2cosh(theta_h)                    eloc, psi double[nbatch]  (= pynqs_eloc_rbm)
 *   PYNQS_RBM_TANH  psi = tanh(a.x) prod_h 2cosh(theta_h)   (rbm_type "tanh") eloc, psi double[nbatch]
 *   PYNQS_RBM_PHASE psi = exp(i (a.x + sum_h ln 2cosh(theta_h)))  ("pRBM")   eloc, psi double[nbatch][2] = (re, im)
 * ("cos" and "complex" are not fused: they run through the module path of pynqs_amd.energy.local_energy.) */
#define PYNQS_RBM_REAL 0
#define PYNQS_RBM_TANH 1
#define PYNQS_RBM_PHASE 2
int pynqs_eloc_rbm_flavour(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                           const void *rbm_table, int nhidden, int flavour, double *eloc, double *psi, void *stream);

/* ---- SIMPLE method with an RBM with COMPLEX parameters, fully fused (kernels_rbm_complex.hip) -----------------------------
 * psi(x) = exp(a.x) prod_h 2 cosh(b_h + sum_o W[h][o] x_o) with complex a, b, W: rbm.py:199-211, rbm_type "complex"
 * (weights double[nhidden][sorb][2], hidden_bias double[nhidden][2], visible_bias double[sorb][2] or NULL: the reference's
 * params_weights / params_hidden_bias / params_visible_bias, (re, im) pairs).  rbm_type "cos" (prod_h cos(theta_h), real
 * parameters) is the same function of i*W, i*b up to the constant 2^nhidden: pass log_scale = nhidden * ln 2.
 *   pynqs_eloc_crbm_supported : [host] 1 if the complex rows of all (orbital, hidden unit) pairs fit the CU's LDS
 *   pynqs_crbm_table_bytes / _build : re-laid-out parameters in caller-owned memory (rebuild after every update)
 *   pynqs_eloc_crbm : eloc double[nbatch][2] = sum_x' <x|H|x'> psi(x')/psi(x);  psi double[nbatch][2] (may be NULL) =
 *                     psi(x) exp(-log_scale). */
int pynqs_eloc_crbm_supported(int sorb, int nele, int noA, int noB, int nhidden);
int64_t pynqs_crbm_table_bytes(int sorb, int nhidden);
int pynqs_crbm_table_build(const double *weights, const double *hidden_bias, const double *visible_bias, int sorb, int nhidden,
                           void *table, void *stream);
int pynqs_eloc_crbm(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                    const void *crbm_table, int nhidden, double log_scale, double *eloc, double *psi, void *stream);

/* ---- Green's-function Monte-Carlo move: gfmc/walker.py:260-279 (sample_update) in one kernel ---------------
 * green double[n][ncomb] (the fixed-node Green's function row of each walker, >= 0), rand_num double[n] in [0, 1),
 * comb uint64[n][ncomb][len] (get_comb_hij_fused's first output).  Per walker: beta = sum_k green[k];
 * index = first k with (green[0] + ... + green[k]) >= rand_num * beta  (the reference's
 * searchsorted(cumsum / beta, rand_num, right=False), clamped to ncomb - 1); x_new = comb[index]. */
int pynqs_gfmc_sample(const double *green, int64_t n, int64_t ncomb, const double *rand_num, const uint64_t *comb,
                      int sorb, int64_t *index, double *beta, uint64_t *x_new, void *stream);

/* ---- the whole Green's-function row of gfmc/walker.py:167-235 (_calculate_green_kernel) in ONE kernel for a trial function
 * that is an RBM with real parameters (flavour PYNQS_RBM_REAL or PYNQS_RBM_TANH; the complex-valued phase flavour has no fixed node):
 *   r_k = psi(x'_k)/psi(x), h_k = <x|H|x'_k>;  for k >= 1: green[k] = -h_k r_k if h_k r_k < 0 (sign-preserving move) else 0;
 *   v_sf = sum of the other h_k r_k (sign-flip potential);  green[0] = max(0, lambda - h_0 - v_sf), clamped[x] = 1 if that was < 0;
 *   eloc[x] = sum_k h_k r_k (unchanged by the fixed-node construction), psi (may be NULL) = psi(x).
 * green double[nbatch][ncomb] in the reference's column order; neither comb nor the amplitude batch is materialised.
 *   pynqs_gfmc_sample_rank : pynqs_gfmc_sample for such a row: x_new = the excitation of rank index - 1 of bra[x] (index 0: bra[x]). */
int pynqs_green_rbm(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                    const void *rbm_table, int nhidden, int flavour, double lambda, double *eloc, double *psi, double *green,
                    uint8_t *clamped, void *stream);
int pynqs_gfmc_sample_rank(const double *green, int64_t n, const double *rand_num, const uint64_t *bra, int sorb, int nele,
                           int noA, int noB, int64_t *index, double *beta, uint64_t *x_new, void *stream);

/* ---- statistics: utils/stats/dist_stats.py:18-79 needs sum p O, sum p |O|^2 (and sum p) before its all-reduce -----
 *   pynqs_moments_workspace : [host] bytes of the workspace (device memory, ZERO it once before the first call)
 *   pynqs_weighted_moments  : workspace[0..3] (doubles) = sum_i p_i Re x_i, sum_i p_i Im x_i, sum_i p_i |x_i|^2,
 *                             sum_i p_i; x is double[n] or interleaved complex double[n][2]; fixed order of additions. */
int64_t pynqs_moments_workspace(void);
/* closing arithmetic of dist_stats.py:59-79 on the summed moments of all ranks (moments[0..3] as above, SUMMED over
 * the ranks; inv_world = 1 / world_size as in comm.py:62-67): out5 = mean_re, mean_im, var, sd, se = sd / sqrt(counts). */
int pynqs_stats_finish(const double *moments, double inv_world, double counts, double *out5, void *stream);
int pynqs_weighted_moments(const double *x, int is_complex, const double *prob, int64_t n, void *workspace, void *stream);

/* REDUCE method front end: vmc/energy/eloc.py:205-324 with eps_sample == 0 keeps the columns with
 * |<x|H|x'>| >= eps (eloc.py:297-298; column 0 is treated like any other).  Two passes, nothing materialised, no
 * atomics.  A walker's row is visited in tiles; T = pynqs_reduce_tiles(...) tiles per walker:
 *   pynqs_reduce_tiles : [host] T for this batch size and system (-1 on bad arguments)
 *   pynqs_reduce_count : tile_counts uint32[nbatch][T] = kept columns per tile (0 for unused tiles)
 *   pynqs_reduce_emit  : tile_offsets int64[nbatch][T] = exclusive prefix sum of tile_counts over the whole flattened
 *                        array (caller) -> kept_col int32[total], kept_onv uint64[total][len], kept_h T[total].
 *                        Walker w's records are the contiguous range that its tiles span; inside it they come tile by
 *                        tile in a reproducible order (diagonal, singles, doubles), NOT in ascending column order --
 *                        kept_col names the column of each record. */
int64_t pynqs_reduce_tiles(int64_t nbatch, int sorb, int nele, int noA, int noB);
int pynqs_reduce_count(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                       const void *plan, int dtype, double eps, uint32_t *tile_counts, void *stream);
int pynqs_reduce_emit(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                      const void *plan, int dtype, double eps, const int64_t *tile_offsets, int32_t *kept_col,
                      uint64_t *kept_onv, void *kept_h, void *stream);

/* Semi-stochastic REDUCE (eloc.py:257-296, eps_sample > 0; the Fe2S2 example uses eps = 1e-2, eps_sample = 1000):
 * columns with |H| >= eps are kept (the three calls above); from the others N columns are drawn with replacement,
 * p_m = |H_m| / S, and a column drawn c times carries the weight (c / N) sign(H_m) S.  The multinomial is drawn
 * hierarchically -- over the tiles on the host, inside a tile on the GPU -- which is the same distribution:
 *   pynqs_reduce_count_sums : like pynqs_reduce_count, plus tile_sums double[nbatch][T] = sum of the sub-eps |H| per
 *                             tile (eps = +inf: nothing is kept, every column can be drawn)
 *   pynqs_reduce_sample     : tile_draws int32[nbatch][T] = draws per tile (host: multinomial over tile_sums),
 *                             sample_offsets int64[nbatch][T] = exclusive prefix of tile_draws over the flattened
 *                             array, walker_scale double[nbatch] = S / N, seed for the counter-based generator.
 *                             Tile t writes one record per DISTINCT drawn column into s_col / s_onv / s_h starting at
 *                             sample_offsets[t] (at most tile_draws[t] of them; pre-fill s_col with -1 to tell the
 *                             unused slots): s_h = sign(H) * hits * walker_scale. */
int pynqs_reduce_count_sums(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                            const void *plan, int dtype, double eps, uint32_t *tile_counts, double *tile_sums,
                            void *stream);
int pynqs_reduce_sample(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                        int dtype, double eps, const int32_t *tile_draws, const int64_t *sample_offsets,
                        const double *walker_scale, uint64_t seed, int32_t *s_col, uint64_t *s_onv, void *s_h,
                        void *stream);

/* ---- REDUCE front end in ONE launch (kernels_reduce_onepass.hip): vmc/energy/eloc.py:205-324 (_reduce_psi) + `Func`
 * (vmc/energy/flip.py:29-63) + onv_to_tensor, i.e. everything between the walkers and the ansatz' forward:
 *   - every column of a walker's row is enumerated once; |<x|H|x'>| >= eps is kept (eloc.py:297-298, 258-262);
 *   - eps_sample > 0 (eloc.py:263-296): the sub-eps |H| are summed per tile in LDS, the N = eps_sample draws of
 *     torch.multinomial are drawn over the tiles inside the kernel (counter-based generator, `seed`), and only the tiles
 *     that received draws are enumerated a second time for the draws inside them; a column drawn c times carries the
 *     weight (c / N) sign(H) S, S = sum of the row's sub-eps |H| (the same distribution as the reference's);
 *   - every record's x' is looked up in the wave-function table (if one is given) and otherwise inserted into a
 *     de-duplication hash table in the same pass; the first occurrence of a determinant gets the next row of the
 *     DISTINCT list and writes its +1/-1 row (the ansatz' input, onv_to_tensor's layout) there.
 * No host round trip: output space is a fixed-capacity segment per (walker, chunk) -- `fixed` slots with a fixed
 * position per column for column 0, the singles and the unpaired doubles (slot = -1 in rec_col when not kept), then the
 * kept doubles compacted in tile order (decoupled look-back in LDS, reproducible), then N slots per walker for the draws.
 * Capacities are the caller's; what was NEEDED comes back in seg_count / counters, so that an overflow is detected (and
 * the call repeated with larger buffers) without anything having been read back in between.
 *
 *   pynqs_reduce_onepass_geometry : [host] out[0] = segments (nbatch * chunks), out[1] = fixed slots per segment,
 *                                   out[2] = bytes of the de-duplication table for `dedup_slots` slots of this sorb
 *                                   (dedup_slots passed in out[2] on entry), out[3] = 1 if the fused form exists for
 *                                   this system (LDS budget; with draws on rows of more than 65536 columns: provided
 *                                   io->tile_scratch is given), else 0
 *   pynqs_reduce_onepass_list_capacity : [host] the largest io->cap_doubles with which pynqs_reduce_onepass keeps a segment's
 *                                   records in an LDS list -- its LIST form (one list per segment) or, without draws, the flushing
 *                                   form (the list is emptied as it fills: 2^30 - 1 = any capacity on rows of more than 65536
 *                                   columns and whenever there is no de-duplication table, else a tenth of a segment's columns);
 *                                   -1: never.  Beyond it the look-back form runs, which long rows should avoid (the multi-pass entry
 *                                   points pynqs_reduce_count / _emit are 2-4 x faster there).  with_row_cache / without_table: whether
 *                                   io->row_cache will be given / io->dedup_table will be NULL
 *   pynqs_reduce_onepass          : the launch (memsets of the de-duplication table and counters included)
 *   pynqs_reduce_contract         : eloc[x] = sum_records w A(x') / A(x) (divide = 1) or sum_records w A(x') (divide = 0) and
 *                                   psi_x[x] = A(x), from the records (io->rec_w / srec_w: any weights in the records' slot
 *                                   layout) and the values A of the distinct list (psi_unique) and of the table (psi_table);
 *                                   one wave per walker, fixed order of additions.
 * Record link: >= 2^30 : row (link - 2^30) of the distinct list (the row was known when the record was written: the record's own new
 * determinant, or one whose winner had already published its row); 0 .. 2^30 - 1 : slot of the de-duplication table (its row number is
 * stored in the slot: the contraction looks it up); <= -2 : position -(link + 2) of the wave-function table; -1 : no amplitude
 * (capacity overflow).  Which of the first two forms a record gets depends on the launch's timing; both lead to the same row. */
typedef struct pynqs_reduce_io {
  int64_t cap_doubles;  /* in: compacted slots per segment after the fixed ones */
  int64_t cap_unique;   /* in: rows of uniq_onv / uniq_pm1 */
  int64_t dedup_slots;  /* in: power of two >= 2 * cap_unique, <= 2^30 */
  int32_t *rec_col;     /* [segments][fixed + cap_doubles] column of the record, -1 = empty slot */
  void *rec_w;          /* T, same shape: <x|H|x'> */
  uint64_t *rec_onv;    /* same shape x len (may be NULL) */
  int32_t *rec_link;    /* same shape */
  int32_t *seg_count;   /* [segments] kept doubles of the segment (may exceed cap_doubles: overflow) */
  int32_t *srec_col;    /* [nbatch][eps_sample]: drawn records (eps_sample > 0), -1 = unused slot */
  void *srec_w;         /* T: (c / N) sign(H) S */
  uint64_t *srec_onv;   /* may be NULL */
  int32_t *srec_link;
  double *row_sum;      /* [nbatch] S (may be NULL) */
  void *dedup_table;    /* out[2] bytes from the geometry call; NULL = no de-duplication: every record that the wave-function table
                           does not answer gets a row of its own in uniq_onv (cap_unique >= the number of records; the distinct-list
                           counter counts records).  E_loc is the same; worth it when nearly all x' are distinct anyway and the table
                           would be gigabytes (random probes there cost 10x the enumeration).  LIST forms only: cap_doubles <=
                           pynqs_reduce_onepass_list_capacity */
  uint64_t *uniq_onv;   /* [cap_unique][len] distinct determinants, order of first insertion */
  void *uniq_pm1;       /* [cap_unique][sorb] +1/-1 rows, element type pm1_dtype (may be NULL) */
  int32_t pm1_dtype;    /* PYNQS_F32 / PYNQS_F64 */
  int32_t lut_is_hash;  /* reserved, must be 1 */
  const void *lut_table;  /* pynqs_hash_build table of the wave-function keys, or NULL */
  int64_t lut_nkeys;
  int32_t *counters;    /* [4] out: distinct determinants needed, overflow bits (1 doubles, 2 table, 4 distinct list),
                           largest seg_count of an overflowing segment, reserved */
  const uint64_t *seed_dev; /* optional: a seed in DEVICE memory, added to `seed` (a captured HIP graph replays the launch with
                               the same arguments: the caller bumps this word between replays) */
  void *row_cache;      /* optional, eps_sample > 0: T[nbatch][ncomb] scratch.  The enumeration stores every matrix element there and the
                           draws read the row back (L2) instead of visiting the drawn tiles a second time: worth it when the draws are
                           dense in the row (1000 draws over Fe2S2's 7876 columns hit every tile); leave NULL for long rows */
  int32_t *uniq_parent; /* optional, [cap_unique]: the walker whose record put the row on the distinct list -- the row is that walker or a
                           single / double excitation of it, which lets an amplitude with cheap updates start from the walker's
                           intermediate values (pynqs_rbm_forward_children) */
  void *tile_scratch;   /* optional, eps_sample > 0 without row_cache: pynqs_reduce_onepass_tile_scratch_bytes bytes.  The per-tile sums of
                           the sub-eps |H| and the tiles' draw counts then live there instead of the LDS (12 bytes per tile: 57 KB per
                           workgroup at sorb 120, which leaves one workgroup per CU); for rows of more than ~65536 columns */
  int64_t tile_scratch_bytes;
  float *row_f32;       /* optional, eps_sample > 0: float[nbatch][stride] scratch, stride = ncomb rounded up to a multiple of 16, 64-byte aligned
                           (pynqs_reduce_onepass_row_f32_elements).  With it -- and where pynqs_reduce_onepass_wants_row_f32 says 1: rows of up to
                           8192 columns, Fe2S2 -- the enumeration leaves the row's sub-eps matrix elements there as float32 (kept columns: 0) and
                           the same workgroup then locates the N draws in it: segments of 16 columns, their sums in float64, one lane per draw
                           (binary search over the segments, 64 bytes of the row read back), hit counts by the rank of a column's bit in a bitmap.
                           P(column j) = float32(|H_j|) / sum of those (relative 6e-8 of the reference's |H_j| / S); the weight of a drawn record
                           is the reference's (c / N) sign(H_j) S with S summed in float64.  The drawn records fill the first slots of
                           srec_* in ascending column order.  row_cache / tile_scratch are not used then. */
} pynqs_reduce_io;
int pynqs_reduce_onepass_geometry(int64_t nbatch, int sorb, int nele, int noA, int noB, int eps_sample, int64_t *out4);
int64_t pynqs_reduce_onepass_tile_scratch_bytes(int64_t nbatch, int sorb, int nele, int noA, int noB, int eps_sample);
int pynqs_reduce_onepass_list_capacity(int64_t nbatch, int sorb, int nele, int noA, int noB, int dtype, int eps_sample,
                                       int with_row_cache, int without_table, int64_t *cap_doubles);
/* 1 when a call with these arguments takes the round-4 semi-stochastic form if io->row_f32 is given (rows of up to 8192 columns, at most
 * 16383 draws, kept records within the list: Fe2S2); 2 when it takes the flushing form (rows of any length, kept records beyond the list),
 * whose draws then read the drawn tiles back from io->row_f32 instead of enumerating them a second time (4 bytes per column and walker:
 * the caller decides whether that is affordable; io->tile_scratch is still wanted on long rows); 0 when the buffer would not be used
 * (leave row_f32 NULL then), -1 on bad arguments;
 * ..._row_f32_elements: floats io->row_f32 must hold for nbatch walkers (-1 on bad arguments). */
int pynqs_reduce_onepass_wants_row_f32(int64_t nbatch, int sorb, int nele, int noA, int noB, int dtype, int eps_sample, int64_t cap_doubles,
                                       int with_tile_scratch /* whether the call will also pass io->tile_scratch */,
                                       int without_table /* whether io->dedup_table will be NULL */);
int64_t pynqs_reduce_onepass_row_f32_elements(int64_t nbatch, int sorb, int nele, int noA, int noB);
int pynqs_reduce_onepass(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                         int dtype, double eps, int eps_sample, uint64_t seed, const pynqs_reduce_io *io, void *stream);
int pynqs_reduce_contract(int64_t nbatch, int sorb, int nele, int noA, int noB, int dtype, int eps_sample,
                          const pynqs_reduce_io *io, const double *psi_unique, const double *psi_table, int psi_is_complex,
                          int divide, double *eloc, double *psi_x, void *stream);

/* ---- the RBM amplitudes themselves on a LIST of determinants (kernels_rbm_forward.hip): psi(x) of vmc/ansatz/rbm/rbm.py:186-211 from the
 * packed bits, one lane per determinant, nothing but the result written.  The amplitude forward of the REDUCE local energy (`Func` on the
 * distinct x', vmc/energy/flip.py:44-50) when the ansatz is an RBM: the PyTorch module's GEMM + element-wise kernels on [n, nhidden] arrays
 * cost 3.5 ms per 1.5 M determinants, this kernel 0.3-0.5 ms.
 *   flavour PYNQS_RBM_REAL / _TANH: weights double[nhidden][sorb], hidden_bias double[nhidden], visible_bias double[sorb] or NULL; psi double[n]
 *   flavour PYNQS_RBM_PHASE       : same parameters; psi double[n][2] = exp(i (a.x + sum_h ln 2cosh theta_h))
 *   flavour PYNQS_RBM_COMPLEX     : weights double[nhidden][sorb][2], hidden_bias double[nhidden][2], visible_bias double[sorb][2] or NULL
 *                                   (the reference's params_* layout, rbm_type "complex"); psi double[n][2] */
#define PYNQS_RBM_COMPLEX 4
int pynqs_rbm_forward(const uint64_t *onv, int64_t n, int sorb, const double *weights, const double *hidden_bias,
                      const double *visible_bias, int nhidden, int flavour, double *psi, void *stream);

/* ---- the energy-gradient estimator for RBM amplitudes, analytically (kernels_rbm_grad.hip; vmc/grad/energy_grad.py:118-184, "AD" method:
 *   loss = 2 Re sum_n p_n conj(ln psi(x_n)) (E_loc(x_n) - <E> c_n); the reference calls loss.backward() on it).
 * For psi = exp(a.x) prod_h 2cosh(theta_h), theta = b + W x (vmc/ansatz/rbm/rbm.py:186-211):
 *   d loss / d theta_k = 2 Re G_k (real parameters) or (2 Re G_k, -2 Im G_k) for a complex parameter stored as (re, im),
 *   G_k = sum_n conj(f_n) O_k(x_n),  f_n = p_n (E_loc(x_n) - <E> c_n),  O = (x_o, tanh theta_h, tanh theta_h x_o).
 *   flavour: PYNQS_RBM_REAL (parameters double[nhidden][sorb], [nhidden], [sorb]) or PYNQS_RBM_COMPLEX (the same with a trailing [2]);
 *   prob double[n]; eloc double[n] or (eloc_is_complex) double[n][2]; e_total: DEVICE pointer to <E> (1 or 2 doubles, as eloc);
 *   pow: c_n double[n] (extra_psi_pow) or NULL for 1; visible_bias / grad_visible_bias may be NULL.
 *   grad_*: same shapes as the parameters (overwritten); loss (may be NULL): the loss above, ln psi on torch.log's principal branch.
 *   workspace: pynqs_rbm_grad_workspace(n, sorb, nhidden, flavour) bytes.  Sums run in a fixed order: bit-reproducible.               */
/* ---- RBM amplitudes of the DISTINCT x' of a REDUCE front end, from their parents (kernels_rbm_forward.hip): row r of the distinct list is
 * walker parent[r] with at most four orbitals flipped, and  prod_h 2cosh(theta_h) = exp(sum_h theta_h) prod_h (1 + q_h),  q_h = exp(-2 theta_h):
 * flipping orbital o to x'_o = +-1 multiplies q_h by the table entry exp(-+4 W_ho).  A child costs 4 table multiplications per hidden unit:
 * no exponential, no sine, no loop over the orbitals (Fe2S2, 1.5 M rows x 40 complex hidden units: 0.60 ms from scratch).
 *   pynqs_rbm_children_table_bytes : [host] size of the table for nwalkers parents (-1 on bad arguments)
 *   pynqs_rbm_children_prepare     : table <- the parents' q_h, sum_h theta_h, a.x and the parameters' factor table
 *   pynqs_rbm_forward_children     : psi[r] for r < min(*count_dev, n) (count_dev may be NULL: all n rows; rows past the count are left alone)
 *   pynqs_rbm_forward_children_supported : 1 for every valid (sorb, nhidden, flavour), else 0.  A factor table ((2 sorb + 1) x (nhidden + 2)
 *                                    entries) of at most 64 KB is kept in LDS, a thread per row; a larger one (sorb x nhidden above
 *                                    ~64 x 64) is read from the L2 by one wave per row, the lanes over the hidden units
 * Flavours and parameter layouts as pynqs_rbm_forward.  exp(-2 theta_h) is formed for the parents: if some Re theta_h < -340 the prepare
 * step raises a flag in the table and pynqs_rbm_forward_children computes every row from scratch instead (pynqs_rbm_forward's
 * algorithm inside the same kernel).  Values agree with pynqs_rbm_forward to rounding (typically
 * 1e-14 relative; a factor 2cosh(theta_h) near zero amplifies it).                                                                      */
int64_t pynqs_rbm_children_table_bytes(int64_t nwalkers, int sorb, int nhidden, int flavour);
int pynqs_rbm_children_prepare(const uint64_t *walkers, int64_t nwalkers, int sorb, const double *weights, const double *hidden_bias,
                               const double *visible_bias, int nhidden, int flavour, void *table, void *stream);
int pynqs_rbm_forward_children(const uint64_t *onv, int64_t n, const int32_t *count_dev, const int32_t *parent,
                               const uint64_t *walkers, int64_t nwalkers, const void *table, int sorb, const double *weights,
                               const double *hidden_bias, const double *visible_bias, int nhidden, int flavour, double *psi,
                               void *stream);
int pynqs_rbm_forward_children_supported(int sorb, int nhidden, int flavour);

int64_t pynqs_rbm_grad_workspace(int64_t n, int sorb, int nhidden, int flavour);
int pynqs_rbm_grad(const uint64_t *onv, int64_t n, int sorb, const double *weights, const double *hidden_bias,
                   const double *visible_bias, int nhidden, int flavour, const double *prob, const double *eloc,
                   int eloc_is_complex, const double *e_total, const double *pow, double *grad_weights,
                   double *grad_hidden_bias, double *grad_visible_bias, double *loss, void *workspace, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PYNQS_AMD_H */
