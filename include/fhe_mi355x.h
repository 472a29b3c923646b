/*
 * fhe_mi355x.h -- C ABI of libfhe_mi355x.so, the MI355X (gfx950) negacyclic-NTT /
 * RNS polynomial-arithmetic engine.
 *
 * Plain C: opaque handles, raw device addresses, sizes; every function returns an
 * int status (0 = FHE_OK) and never throws across the boundary; fhe_last_error()
 * gives the text of the last failure on the calling thread.  All work is enqueued
 * on the stream passed in (a hipStream_t as void*; NULL = the context's own
 * stream); the caller synchronises (fhe_sync), exactly as the reference's harness
 * does around the Phantom calls (reliability_test/ntt_test.cu:88-102).
 *
 * Each entry point cites the reference interface it stands in for.  Paths are
 * relative to the Stardust-lf/fhe-reliability-gpu tree.  "Phantom" symbols are the
 * ones the reference's binaries import from the (absent) libPhantom.so
 * (nm -D reliability_test/build/ntt_test).
 *
 * Data layout: residue polynomials are limb-major, [n_poly][limbs][N] uint64_t,
 * contiguous (h_data[i*dim + j], reliability_test/ntt_test.cu:79).
 * Semantics: every 64-bit input word is first taken modulo its limb's modulus, all
 * outputs are canonical residues in [0, q).
 */
#ifndef FHE_MI355X_H
#define FHE_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FHE_OK 0
#define FHE_ERR_INVALID 1     /* bad argument */
#define FHE_ERR_HIP 2         /* HIP runtime error, see fhe_last_error() */
#define FHE_ERR_UNSUPPORTED 3 /* size / modulus outside what the kernels handle */
#define FHE_ERR_NOMEM 4

#define FHE_PATH_F64 0 /* q < 2^50: exact FP64 arithmetic (reference prime size, ntt_test.cu:44) */
#define FHE_PATH_U64 1 /* q < 2^61: 64-bit Shoup / Harvey arithmetic */

typedef struct fhe_ctx fhe_ctx;               /* device + stream (phantom::util::cuda_stream_wrapper, ntt_test.cu:40-41) */
typedef struct fhe_ntt_tables fhe_ntt_tables; /* DModulus[] + DNTTTable (ntt_test.cu:47-69) */
typedef struct fhe_baseconv fhe_baseconv;     /* base-conversion plan (rfhe_framewk/src/baseConv.py:14-18) */
typedef struct fhe_fourstep fhe_fourstep;     /* four-step plan (reliability_test/four_step_ntt_prot.py:71-79) */
typedef struct fhe_abft fhe_abft;             /* checksum weights of the ECC detector (rfhe_framewk/src/negaclic_ntt.py:130-149) */
typedef struct fhe_keyswitch fhe_keyswitch;   /* key-switch plan (shape of profile_framewk/build/data/ckks/16384_4:466-539) */

int fhe_version(void);
const char *fhe_last_error(void);

/* ---- context, memory, streams ------------------------------------------- */
/* cuda_stream_wrapper{ctor,get_stream} (ntt_test.cu:40-41).  device = HIP ordinal. */
int fhe_ctx_create(int device, fhe_ctx **out);
int fhe_ctx_destroy(fhe_ctx *ctx);
int fhe_ctx_stream(fhe_ctx *ctx, void **stream_out);
/* Tuning knobs (no reference counterpart): "ntt_mode" 0 = two launches per transform
 * (default), 1 = fused launch with the first-pass -> second-pass hand-off inside one XCD
 * (experimental, sizes 2^13..2^17); "fused_dist" pipeline distance between a limb's first
 * and second pass; "fused_wgs" persistent workgroups launched; "fused_variant" hand-off
 * load flavour; "ntt_resident" 1 = sizes 2^13 and 2^14 run as one LDS-resident pass
 * (their limb fits a CU's 160 KiB of LDS; experimental), 0 (default) = two launches like the larger sizes;
 * "ntt_packed" 1 = the forward 2^16 transform of FP64 limbs hands its intermediate over as packed 50-bit residues
 * (fewer bytes, more arithmetic: measured slower, experimental), 0 (default) = 8-byte words in place;
 * "ntt_chunk_mib" sub-batch size of two-launch transforms of batches above "ntt_chunk_floor_mib" (192) (default 96: a sub-batch's second launch finds
 * the first one's output in the 256 MiB Infinity Cache; 0 = one launch pair for the whole batch); "ntt_split" 1 / 0 / -1 = the
 * sub-batches of one call alternate between the caller's stream and a side stream the context owns, forked and joined by events
 * (one sub-batch's row pass runs under the next one's column pass; not inside a stream capture), -1 = default = on; "ntt_stream" 1 / 0 / -1 = non-temporal
 * loads / stores on the external side of the two launches (always / never / for sub-batched calls: the default); "ntt_pingpong" 1 / 0 / -1 = the two launches hand
 * over through a per-stream scratch buffer of one sub-batch, so that both run out of place (always / never / for calls that are
 * sub-batched: the default); "ks_fused" -1 / 0 / 1 = key-switch inner product
 * fused with the extended limbs' row pass by shape / never / always (hoisted batches of two or more elements transform the shared digits
 * once and use the plain inner product unless this is set); "hmult_fused_rescale" 1 (default) / 0 = fhe_hmult with rescale runs the mod-down and
 * the rescale behind one forward transform where the shape allows (two-launch sizes, K >= 2, no plain modulus) / as two steps (FHE_HMULT_FUSED_RESCALE);
 * "tile_geo" column-tile geometry of the two-launch path; "ntt_only_pass" 0 / 1 = launch only the
 * first / second pass of a two-pass size (timing of the individual kernels; -1 = whole transform).  Environment overrides at context creation: FHE_NTT_MODE=twopass|fused,
 * FHE_FUSED_DIST, FHE_FUSED_WGS.  Results are identical in every setting. */
int fhe_ctx_set_option(fhe_ctx *ctx, const char *name, long value);
/* Synchronises the streams used so far and reports whether any fused-NTT launch hit its
 * bounded-wait limit (never expected; the kernel then stops instead of hanging). */
int fhe_ctx_check(fhe_ctx *ctx);
/* cudaStreamSynchronize (ntt_test.cu:102,151) */
int fhe_sync(fhe_ctx *ctx, void *stream);
/* make_cuda_auto_ptr<uint64_t>(n, stream) / its destructor (ntt_test.cu:88) */
int fhe_alloc(fhe_ctx *ctx, size_t bytes, void **dptr);
int fhe_free(fhe_ctx *ctx, void *dptr);
/* cudaMemcpyAsync H2D / D2H / D2D (ntt_test.cu:89-101) */
int fhe_h2d(fhe_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream);
int fhe_d2h(fhe_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream);
int fhe_d2d(fhe_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream);
int fhe_memset(fhe_ctx *ctx, void *dst, int byte, size_t bytes, void *stream);

/* ---- moduli and host tables (a10) ---------------------------------------- */
/* phantom::arith::CoeffModulus::Create(N, {bits...}) (ntt_test.cu:44): for each bit
 * size the largest primes p < 2^bits with p = 1 mod 2N, handed out smallest first. */
int fhe_moduli_create(uint64_t N, const int *bits, int count, uint64_t *out_q);
/* Modulus::const_ratio() (ntt_test.cu:51-52): out = {floor(2^128/q) lo, hi, 2^128 mod q} */
int fhe_modulus_const_ratio(uint64_t q, uint64_t out[3]);
/* minimal primitive `order`-th root of unity mod q (what phantom::arith::NTT picks) */
int fhe_min_primitive_root(uint64_t q, uint64_t order, uint64_t *out);
/* NTT::get_from_root_powers() / get_from_root_powers_shoup() (ntt_test.cu:60-64):
 * rp[bitrev(i)] = psi^i, shoup[k] = floor(rp[k] 2^64 / q); either pointer may be NULL */
int fhe_root_powers(uint64_t q, int log_n, uint64_t *rp, uint64_t *rp_shoup);

/* ---- NTT tables (device) -------------------------------------------------- */
/* DModulus::set + DNTTTable::init/set for `count` limbs (ntt_test.cu:47-69) with the
 * tables of phantom::arith::NTT(log_n, q[i]).  Limb i uses FHE_PATH_F64 when
 * q[i] < 2^50, else FHE_PATH_U64 (q[i] < 2^61). */
int fhe_ntt_tables_create(fhe_ctx *ctx, int log_n, const uint64_t *q, int count, fhe_ntt_tables **out);
/* Same, from caller-supplied forward root powers (count x N, entry k = psi^bitrev(k)):
 * the DNTTTable::set(..., twiddle, twiddle_shoup, ...) path of ntt_test.cu:61-69.
 * force_path: -1 = choose by modulus size, else FHE_PATH_*. */
int fhe_ntt_tables_create_from_roots(fhe_ctx *ctx, int log_n, const uint64_t *q, int count,
                                     const uint64_t *root_powers, int force_path, fhe_ntt_tables **out);
int fhe_ntt_tables_destroy(fhe_ntt_tables *t);
/* out_path[i] = arithmetic path of limb i; out_psi[i] = its 2N-th root (either may be NULL) */
int fhe_ntt_tables_info(const fhe_ntt_tables *t, int *log_n, int *count, int *out_path, uint64_t *out_psi);

/* ---- transforms (a2, a3) --------------------------------------------------- */
/* nwt_2d_radix8_forward_inplace(uint64_t*, const DNTTTable&, size_t coeff_modulus_size,
 * size_t start_modulus_idx, const cudaStream_t&) (ntt_test.cu:95,144; ntt_real_test.cu:89,126):
 * `limbs` forward negacyclic NTTs in place, limb i with modulus start_idx + i;
 * natural-order input, bit-reversed output, results in [0,q). */
int fhe_ntt_forward_inplace(fhe_ctx *ctx, uint64_t *d_data, const fhe_ntt_tables *t, size_t limbs, size_t start_idx,
                            void *stream);
/* Phantom's inverse (nwt_2d_radix8_backward_inplace, used inside multiply/decrypt,
 * reliability_test/dotprod_test.cu:113,119): bit-reversed input, natural output, times N^-1;
 * equals rfhe_framewk/src/negaclic_ntt.py:102-109 composed with the bit reversal. */
int fhe_ntt_inverse_inplace(fhe_ctx *ctx, uint64_t *d_data, const fhe_ntt_tables *t, size_t limbs, size_t start_idx,
                            void *stream);
/* Batched forms: d_data = [n_poly][limbs][N]; polynomial p, limb i uses modulus start_idx + i. */
int fhe_ntt_forward_batch(fhe_ctx *ctx, uint64_t *d_data, const fhe_ntt_tables *t, size_t n_poly, size_t limbs,
                          size_t start_idx, void *stream);
int fhe_ntt_inverse_batch(fhe_ctx *ctx, uint64_t *d_data, const fhe_ntt_tables *t, size_t n_poly, size_t limbs,
                          size_t start_idx, void *stream);

/* d_dst[v][i] = d_src[v][bitrev(i, log_n)] for n_vec vectors; d_dst != d_src.  Converts between
 * the bit-reversed order of the Phantom-style transforms and the natural order the
 * reference's Python returns (rfhe_framewk/src/negaclic_ntt.py:16-21,42). */
int fhe_bitrev_permute(fhe_ctx *ctx, uint64_t *d_dst, const uint64_t *d_src, int log_n, size_t n_vec, void *stream);

/* ---- cyclic transform (a1) --------------------------------------------------- */
/* ntt(a, mod, root) of motivation/ntt.py:8-32 (twins motivation/bsgs.py:5-29,
 * rfhe_framewk/src/ntt.py:38-55): natural order in and out, `root` a generator
 * (wlen = root^((mod-1)/len)); any modulus < 2^61, composite included.
 * convention 1 selects rfhe_framewk/src/negaclic_ntt.py:38-57 (root is a primitive
 * n-th root, wlen = root^(n/len)).  inverse != 0 gives intt(): transform with
 * root^(mod-2), then times n^(mod-2) (motivation/bsgs.py:31-36).
 * d_data: n_vec vectors of 2^log_n words, in place; d_scratch: same size. */
int fhe_ntt_cyclic(fhe_ctx *ctx, uint64_t *d_data, uint64_t *d_scratch, int log_n, size_t n_vec, uint64_t mod,
                   uint64_t root, int convention, int inverse, void *stream);

/* ---- four-step transform (a6) -------------------------------------------------- */
/* four_step_ntt(a, N) of reliability_test/four_step_ntt_prot.py:71-109 with N = n1*n2
 * (both powers of two, n1 != n2 allowed): column transforms, twiddle w^(k2 t1), row
 * transforms, transposed output; equals ntt_direct (:49-58).  g = generator (G=3, :17). */
/* Range: 2 <= n1, n2 <= 2^20, n1 * n2 <= 2^26, mod < 2^61 (FHE_ERR_INVALID / FHE_ERR_UNSUPPORTED beyond).  Up to N = 2^20 the whole flow
 * is ONE natural-order transform of the engine (two launches, no transpose pass); from 2^21 to 2^26 (the reference's default modulus
 * 998244353 admits N up to 2^23) it is the reference's composition itself: transpose, n1 transforms of length n2, the twiddle on the
 * way through the second transpose, n2 transforms of length n1, transpose -- five sweeps, plan-owned buffers of two batches.
 * A plan's hand-off buffers are reused by every call on it: one plan, one stream at a time (see fhe_hmult). */
int fhe_fourstep_create(fhe_ctx *ctx, uint64_t n1, uint64_t n2, uint64_t mod, uint64_t g, fhe_fourstep **out);
int fhe_fourstep_destroy(fhe_fourstep *p);
int fhe_fourstep_ntt(fhe_ctx *ctx, uint64_t *d_dst, const uint64_t *d_src, fhe_fourstep *p, void *stream);
/* n_vec vectors of n1*n2 words each, contiguous; d_dst may equal d_src.  Two launches for the whole batch: the column
 * transforms read the input's columns directly (no transpose pass), the row transforms carry the twiddle step in their
 * butterflies and write natural order. */
int fhe_fourstep_ntt_batch(fhe_ctx *ctx, uint64_t *d_dst, const uint64_t *d_src, fhe_fourstep *p, size_t n_vec, void *stream);

/* ---- coefficient-wise products (a4, a5) ------------------------------------------ */
/* C_hat[i] = A_hat[i] * B_hat[i] % mod (rfhe_framewk/src/negaclic_ntt.py:126), per limb;
 * the NTT-domain core of phantom::multiply (dotprod_test.cu:113).  c may alias a or b. */
int fhe_modmul(fhe_ctx *ctx, uint64_t *d_c, const uint64_t *d_a, const uint64_t *d_b, const fhe_ntt_tables *t,
               size_t n_poly, size_t limbs, size_t start_idx, void *stream);
/* c = (c + a*b) mod q: keyswitch / BSGS inner-product accumulate (motivation/bsgs.py:50) */
int fhe_modmul_acc(fhe_ctx *ctx, uint64_t *d_c, const uint64_t *d_a, const uint64_t *d_b, const fhe_ntt_tables *t,
                   size_t n_poly, size_t limbs, size_t start_idx, void *stream);
/* c = (a + b) mod q per limb (phantom::add_inplace, reliability_test/dotprod_test.cu:147) */
int fhe_modadd(fhe_ctx *ctx, uint64_t *d_c, const uint64_t *d_a, const uint64_t *d_b, const fhe_ntt_tables *t, size_t n_poly,
               size_t limbs, size_t start_idx, void *stream);
int fhe_modsub(fhe_ctx *ctx, uint64_t *d_c, const uint64_t *d_a, const uint64_t *d_b, const fhe_ntt_tables *t, size_t n_poly,
               size_t limbs, size_t start_idx, void *stream);
/* c = a * mul[l] + add[l] mod q_l with one host-side scalar pair per limb (either array may be NULL:
 * mul defaults to 1, add to 0); limbs <= 64.  Scalar steps of mod-switching and BGV decryption
 * (phantom::mod_switch_to_next_inplace / PhantomSecretKey::decrypt, dotprod_test.cu:115,119). */
int fhe_scalar_affine(fhe_ctx *ctx, uint64_t *d_c, const uint64_t *d_a, const uint64_t *mul, const uint64_t *add,
                      const fhe_ntt_tables *t, size_t n_poly, size_t limbs, size_t start_idx, void *stream);
/* poly_mul_negacyclic_ntt (rfhe_framewk/src/negaclic_ntt.py:123-127): c = a * b mod (x^N + 1, q).
 * a and b are scratch: their contents are unspecified afterwards (partly transformed); c may alias
 * either.  One read of each factor tile and one write of the product tile between the column passes. */
int fhe_polymul(fhe_ctx *ctx, uint64_t *d_c, uint64_t *d_a, uint64_t *d_b, const fhe_ntt_tables *t, size_t n_poly,
                size_t limbs, size_t start_idx, void *stream);

/* ---- base conversion / CRT (a8) ------------------------------------------------- */
int fhe_baseconv_create(fhe_ctx *ctx, const uint64_t *mod_in, int m, const uint64_t *mod_out, int k,
                        fhe_baseconv **out);
int fhe_baseconv_destroy(fhe_baseconv *p);
/* base_conv_fixed (motivation/baseConv.py:67-83): exact; in m x N, out k x N (limb-major) */
int fhe_baseconv_exact(fhe_ctx *ctx, uint64_t *d_out, const uint64_t *d_in, const fhe_baseconv *p, size_t N,
                       void *stream);
/* bConv (rfhe_framewk/src/baseConv.py:10-40): sum_j ((r_j Phat_j inv_j) mod q_k), NOT reduced;
 * out k x N limb-major (the reference returns the transposed [i][k] list).  Requires
 * m * max(q_k) < 2^64. */
int fhe_baseconv_fast(fhe_ctx *ctx, uint64_t *d_out, const uint64_t *d_in, const fhe_baseconv *p, size_t N,
                      void *stream);
/* crt_kernel (rfhe_framewk/src/baseConv.cu:85-120, launch :187-193): Garner CRT of m <= 16
 * limbs into a 128-bit integer (lo, hi) per coefficient, 128-bit wrap-around as the
 * kernel; residues m x N row-major. */
int fhe_crt_garner(fhe_ctx *ctx, uint64_t *d_x_lo, uint64_t *d_x_hi, const uint64_t *d_residues,
                   const uint64_t *moduli, int m, size_t N, void *stream);

/* ---- BSGS block-diagonal Hadamard (a9) --------------------------------------------- */
/* diag_block_hadamard_matvec (motivation/bsgs.py:39-52): y_i = sum_j M[(j-i) mod k] (.) v_j,
 * k blocks of `block_size`.  mod == 0: int64 wrap-around, no reduction (the reference's
 * NumPy arithmetic); otherwise every product and the sum are reduced mod `mod`. */
int fhe_bsgs_hadamard(fhe_ctx *ctx, uint64_t *d_y, const uint64_t *d_M_blocks, const uint64_t *d_v, int k,
                      int block_size, uint64_t mod, void *stream);

/* ---- rotation / key switching ----------------------------------- */
/* Galois automorphism x -> x^galois_elt (odd) -- the index map behind phantom::rotate_inplace
 * (reliability_test/dotprod_test.cu:146) -- on coefficient-domain limbs: dst[(i k) mod N] = +-src[i].
 * d_dst != d_src; layout [n_poly][limbs][N]. */
int fhe_automorphism(fhe_ctx *ctx, uint64_t *d_dst, const uint64_t *d_src, const fhe_ntt_tables *t, uint32_t galois_elt,
                     size_t n_poly, size_t limbs, size_t start_idx, void *stream);
/* The same map on NTT-domain limbs (bit-reversed order): a pure permutation of the slots. */
int fhe_automorphism_ntt(fhe_ctx *ctx, uint64_t *d_dst, const uint64_t *d_src, int log_n, uint32_t galois_elt, size_t n_units,
                         void *stream);
/* Hybrid RNS key switching with the operation sequence the reference's SEAL trace shows for one
 * KEYSWITCH + MODSWITCH (profile_framewk/build/data/ckks/16384_4:466-539; summary in
 * profile_framewk/build/sum_trace.py:10-94): INTT of the input limbs, per digit base extension to every
 * other prime ("MODREDUCTION"), NTT, multiply-accumulate with the evaluation key ("MULTEVALK"), then
 * mod-down by the special primes (INTT, conversion, NTT, subtract, times P^-1).
 * `t` holds L ciphertext primes followed by K special primes; the L primes are cut into `dnum` digits of
 * ceil(L/dnum) consecutive limbs (SEAL: dnum = L, one prime per digit; draw_dnum_rot_mul.py:64 sweeps dnum).
 *   d_c   : L x N, NTT domain                       d_evk : dnum x 2 x (L+K) x N, NTT domain (b_d, a_d)
 *   d_out0, d_out1 : L x N, NTT domain; out0 + out1*s ~ c*s' when evk encrypts P*Qhat_d*[Qhat_d^-1]_{Q_d}*s'.
 * The plan owns the intermediate buffers (extended digits, accumulators): one call at a time per plan. */
int fhe_keyswitch_create(fhe_ctx *ctx, const fhe_ntt_tables *t, int L, int K, int dnum, fhe_keyswitch **out);
/* The same key switch with the RNS limbs sharded over `world` ranks, one process per GPU (BASELINE configs 4-5; the reference
 * has no multi-device code).  Rank r owns the ciphertext limbs [clo, clo+cn) and the special limbs
 * [slo, slo+sn) that fhe_keyswitch_shard_layout reports (out = {clo, cn, slo, sn, cmax, smax}; pure host function, no
 * device needed): its rows of the input (cn x N), of every key digit (dnum x 2 x (cn+sn) x N, ciphertext rows first) and of
 * the result.  The only exchanges are two all-gathers, issued by the host between the three phases (RCCL over xGMI when
 * the host uses torch.distributed's "nccl" backend):
 *   fhe_keyswitch_shard_begin   INTT of the owned input limbs into rows [rank*cmax, rank*cmax+cn) of d_gather1
 *   -- all-gather of d_gather1 ([world][cmax][N] words, in place) --
 *   fhe_keyswitch_shard_inner   digit extension to the owned limbs, NTT, inner product with the owned key rows, INTT of
 *                               the owned special limbs of both halves into slot `rank` of d_gather2 ([world][2][smax][N])
 *   -- all-gather of d_gather2 (in place) --
 *   fhe_keyswitch_shard_finish  mod-down to the owned ciphertext limbs (+ optional addends, as a rotation / relinearisation needs)
 * The gather buffers belong to the caller and are fixed at creation.  world = 1 gives fhe_keyswitch_create's plan.
 * d_bcast (optional, 3 x N words): broadcast buffer of the sharded rescale below; NULL = the plan does not rescale. */
int fhe_keyswitch_shard_layout(int L, int K, int world, int rank, int out[6]);
int fhe_keyswitch_create_sharded(fhe_ctx *ctx, const fhe_ntt_tables *t, int L, int K, int dnum, int world, int rank,
                                 uint64_t *d_gather1, uint64_t *d_gather2, uint64_t *d_bcast, fhe_keyswitch **out);
int fhe_keyswitch_shard_begin(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_c_local, void *stream);
int fhe_keyswitch_shard_inner(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_c_local, const uint64_t *d_evk_local, void *stream);
int fhe_keyswitch_shard_finish(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0_local, uint64_t *d_out1_local,
                               const uint64_t *d_add0_local, const uint64_t *d_add1_local, void *stream);
/* BGV form of the mod-down: with plaintext modulus `plain_modulus` (0 = off, the CKKS-style flooring
 * above) the removed part delta satisfies delta = acc mod P and delta = 0 mod plain_modulus, so the
 * plaintext is preserved exactly (scheme of reliability_test/dotprod_test.cu:199-204). */
int fhe_keyswitch_set_plain_modulus(fhe_keyswitch *p, uint64_t plain_modulus);
int fhe_keyswitch_destroy(fhe_keyswitch *p);
int fhe_keyswitch_apply(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_c,
                        const uint64_t *d_evk, void *stream);

/* Rotation of a two-part ciphertext (c0, c1), NTT domain, L limbs each -- phantom::rotate_inplace
 * (dotprod_test.cu:146), frontend "ROTATE" of the reference's traces (16384_4:466-539): apply
 * x -> x^galois_elt to both parts, key-switch the second part with the Galois key, add.
 * out0 + out1*s ~ sigma(c0 + c1*s). */
/* OUT OF PLACE: the parts are read through the Galois permutation while the outputs are written -- neither output may be one of the
 * input parts (FHE_ERR_INVALID); the same holds for fhe_rotate_shard_finish's d_c0_local. */
int fhe_rotate(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_c0, const uint64_t *d_c1,
               uint32_t galois_elt, const uint64_t *d_galois_key, void *stream);
/* Hoisted rotations: n_rot rotations of ONE ciphertext (the baby steps of a BSGS matrix-vector product,
 * profile_framewk/src/matmul_ckks.cpp:45-113; rotate-and-sum, reliability_test/dotprod_test.cu:143-148).  The decomposition of c1 --
 * INTT, digit extension, the extended limbs' forward transform -- is shared; per Galois element only the inner product with the key
 * and the mod-down run (sigma is taken on the mod-down's loads).  Keys are given in the UN-ROTATED frame:
 * d_prepared_keys[r] = fhe_galois_key_prepare(key of galois_elts[r]) = sigma^-1 of every key row, computed once per key (layout
 * unchanged, [dnum][2][L+K][N]).  d_out0 / d_out1 / d_prepared_keys are HOST arrays of n_rot device pointers.
 * Each result is a rotation of (c0, c1) by its element -- out0 + out1 s ~ sigma(c0 + c1 s) -- with ext_d = sigma(extension of c1's
 * digit); fhe_rotate extends sigma(c1)'s digit instead, so the two differ word by word (by multiples of the digit moduli where sigma
 * flips a sign) while decrypting to the same plaintext.  One device, N >= 2^5, out of place. */
int fhe_galois_key_prepare(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_key_out, const uint64_t *d_key_in, uint32_t galois_elt, void *stream);
int fhe_rotate_hoisted(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *const *d_out0, uint64_t *const *d_out1, const uint64_t *d_c0,
                       const uint64_t *d_c1, const uint32_t *galois_elts, const uint64_t *const *d_prepared_keys, size_t n_rot, void *stream);
/* Hoisted rotations with the limbs sharded (plans of fhe_keyswitch_create_sharded): gather 1 and the digit extension are shared by all
 * elements, so a rotation costs ONE all-gather (the special limbs') instead of two.  Per ciphertext: _begin (INTT of the owned limbs of the
 * un-rotated c1 into d_gather1), all-gather of d_gather1, _extend; per Galois element: _inner (inner product with the owned rows of the
 * prepared key -- fhe_galois_key_prepare on the rank's rows --, INTT of sigma(owned special limbs) into d_gather2), all-gather of d_gather2,
 * _finish (mod-down to the owned limbs; sums and c0 read through the Galois map).  Values as fhe_rotate_hoisted. */
int fhe_rotate_hoisted_shard_begin(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_c1_local, void *stream);
int fhe_rotate_hoisted_shard_extend(fhe_ctx *ctx, fhe_keyswitch *p, void *stream);
int fhe_rotate_hoisted_shard_inner(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_c1_local, const uint64_t *d_prepared_key_local,
                                   uint32_t galois_elt, void *stream);
int fhe_rotate_hoisted_shard_finish(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0_local, uint64_t *d_out1_local, const uint64_t *d_c0_local,
                                    uint32_t galois_elt, void *stream);
/* Baby-step / giant-step matrix-vector product on a two-part ciphertext x = (c0, c1) (profile_framewk/src/matmul_ckks.cpp:45-113 --
 * rotate, multiply_plain with a diagonal, add -- in the arrangement with n1 + n2 - 2 rotations; plaintext block form:
 * motivation/bsgs.py:39-52):
 *     y = sum_{g < n2} sigma_{giant_elts[g-1]}( sum_{b < n1} diag[g][b] (.) sigma_{baby_elts[b-1]}(x) ),   g = 0 and b = 0: no rotation.
 * d_diags = [n2][n1][L][N] NTT-form plaintext polynomials (the caller pre-rotates the diagonals of giant step g by -g n1, as BSGS
 * requires); baby keys in the un-rotated frame (fhe_galois_key_prepare), giant keys as fhe_rotate takes them; baby_elts / the key
 * arrays have n1 - 1 resp. n2 - 1 entries (host arrays).  The n1 - 1 baby rotations share one decomposition of x
 * (fhe_rotate_hoisted), each inner sum is one launch, the giant rotations are fhe_rotate calls accumulated into (d_out0, d_out1).
 * No rescale inside (apply fhe_rescale to the result).  Out of place; one device. */
int fhe_bsgs_matvec(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_c0, const uint64_t *d_c1,
                    const uint64_t *d_diags, size_t n1, size_t n2, const uint32_t *baby_elts, const uint64_t *const *d_baby_keys_prepared,
                    const uint32_t *giant_elts, const uint64_t *const *d_giant_keys, void *stream);
/* The same rotation on a limb-sharded plan (fhe_keyswitch_create_sharded): the automorphism permutes slots inside a limb, so each
 * rank applies it to its own rows -- on the loads of the launches below, there is no permuted copy of c0 and no launch for it.
 * Phases and joins as for the sharded key switch:
 *   fhe_rotate_shard_begin   INTT of sigma(c1)'s owned limbs into this rank's rows of d_gather1   -- all-gather of d_gather1 --
 *   fhe_rotate_shard_inner   extension, NTT, inner product with the owned Galois-key rows, INTT of the owned special limbs
 *                            -- all-gather of d_gather2 --
 *   fhe_rotate_shard_finish  mod-down to the owned limbs, sigma(c0) added to the first part */
int fhe_rotate_shard_begin(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_c1_local, uint32_t galois_elt, void *stream);
int fhe_rotate_shard_inner(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_galois_key_local, void *stream);
int fhe_rotate_shard_finish(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0_local, uint64_t *d_out1_local, const uint64_t *d_c0_local,
                            uint32_t galois_elt, void *stream);
/* ---- homomorphic multiply: tensor product, relinearisation, rescale (BASELINE config 4) -------------- */
/* phantom::multiply (reliability_test/dotprod_test.cu:113; frontend MULTIPLY_CKKS of the SEAL traces,
 * profile_framewk/build/data/ckks/16384_4:388-389): per limb d0 = a0 b0, d1 = a0 b1 + a1 b0, d2 = a1 b1 on NTT-form
 * parts of `limbs` limbs each, one pass over the seven operands. */
int fhe_tensor_product(fhe_ctx *ctx, uint64_t *d_d0, uint64_t *d_d1, uint64_t *d_d2, const uint64_t *d_a0, const uint64_t *d_a1,
                       const uint64_t *d_b0, const uint64_t *d_b1, const fhe_ntt_tables *t, size_t limbs, size_t start_idx, void *stream);
/* phantom::relinearize_inplace (dotprod_test.cu:114; frontend RELIN, 16384_4:390-451): (d0, d1, d2) -> (d0 + ks0, d1 + ks1)
 * with (ks0, ks1) = key switch of d2 under the relinearisation key (layout of fhe_keyswitch_apply's d_evk); the two
 * additions ride on the key switch's last launch. */
int fhe_relinearize(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_d0, const uint64_t *d_d1,
                    const uint64_t *d_d2, const uint64_t *d_relin_key, void *stream);
/* phantom::mod_switch_to_next_inplace (dotprod_test.cu:115) / CKKS rescale: drop the plan's last ciphertext prime,
 * c' = (c - [c]_{q_last}) / q_last on each of n_parts (1..3) parts; d_in = [n_parts][L][N], d_out = [n_parts][L-1][N],
 * NTT domain.  With a plain modulus set on the plan (BGV) the removed part is t [c t^-1]_{q_last}, so the plaintext is
 * kept up to the factor q_last^-1 mod t. */
/* OUT OF PLACE: input parts are L rows apart, output parts L-1 -- d_out must not overlap d_in (FHE_ERR_INVALID). */
int fhe_rescale(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out, const uint64_t *d_in, size_t n_parts, void *stream);
/* fhe_rescale with the limbs sharded (plans of fhe_keyswitch_create_sharded with a d_bcast buffer): d_in_local = [n_parts][cn][N], the
 * rows this rank owns.  _begin: the owner of limb L-1 (fhe_rescale_shard_info: owns_last) turns the last limb of every part to
 * coefficient form into d_bcast; the host broadcasts d_bcast from that rank (n_parts x N words, the one exchange); _finish: every
 * rank forms (c - delta) / q_last on the out_rows owned limbs below L-1, d_out_local = [n_parts][out_rows][N]. */
int fhe_rescale_shard_info(const fhe_keyswitch *p, int *owns_last, int *out_rows);
int fhe_rescale_shard_begin(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_in_local, size_t n_parts, void *stream);
int fhe_rescale_shard_finish(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out_local, const uint64_t *d_in_local, size_t n_parts, void *stream);
/* The three steps above in one call (multiply -> relinearize -> mod_switch, dotprod_test.cu:113-115) on two two-part
 * ciphertexts of L limbs; rescale != 0: outputs have L-1 limbs, else L.  The plan owns the intermediates. */
/* The operands are consumed by the first launch (tensor product into plan-owned buffers), so an output may reuse an operand's
 * buffer; d_out0 and d_out1 must be distinct.
 * ONE PLAN, ONE STREAM AT A TIME: a plan (fhe_keyswitch, fhe_fourstep, fhe_abft) owns scratch buffers -- extended digits, sums,
 * converted limbs, the tensor product, rescale residues, the four-step hand-off, checksum partials -- that every call on it
 * reuses.  Calls on the SAME plan must be ordered by one stream (or by the caller's events); different plans, and the plan-less
 * transforms / products, may run on different streams concurrently. */
int fhe_hmult(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_a0, const uint64_t *d_a1,
              const uint64_t *d_b0, const uint64_t *d_b1, const uint64_t *d_relin_key, int rescale, void *stream);
/* multiply -> relinearize -> mod_switch with the limbs sharded, the mod-down and the rescale behind ONE forward transform (what
 * fhe_hmult does on one device where fhe_hmult_shard_fusable() / the shape allow: two-launch sizes, K >= 2, CKKS form): after
 * fhe_tensor_product, fhe_keyswitch_shard_begin / _inner and their two all-gathers,
 *   _finish_begin: converts the gathered special limbs to this rank's rows; the owner of limb L-1 writes
 *                  y = INTT(v_{L-1}) of both parts into d_bcast ([2][N] of it), d_add*_local = this rank's rows of d0 / d1;
 *   the host broadcasts d_bcast from that rank (the one exchange of the rescale);
 *   _finish_end:   d_out*_local = [out_rows][N] (fhe_rescale_shard_info), the owned limbs below L-1 of the rescaled parts.
 * Same words as fhe_keyswitch_shard_finish + fhe_rescale_shard_begin / _finish (reliability_test/dotprod_test.cu:114-115). */
int fhe_hmult_shard_fusable(const fhe_ctx *ctx, const fhe_keyswitch *p);
int fhe_hmult_shard_finish_begin(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_add0_local, const uint64_t *d_add1_local, void *stream);
int fhe_hmult_shard_finish_end(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0_local, uint64_t *d_out1_local, const uint64_t *d_add0_local,
                               const uint64_t *d_add1_local, void *stream);

/* Operation trace in the line format the reference's tools consume
 * (profile_framewk/build/analyze_trace.py:16-19, sum_trace.py:16-19): "frontend: ROTATE",
 * "[NTT] total cost <n> us" per transform launch, "[MODREDUCTION]/[MULTEVK]/[KEYSWITCH]/[MODSWITCH] total
 * cost <n> us" (inclusive of the transforms before them, as SEAL's timers are), "frontend: ROTATE[<n>
 * microseconds]".  While enabled every traced step synchronises the stream (timing tool, not a fast
 * path).  fhe_ctx_trace_read copies the text collected so far (NUL-terminated, truncated to `cap`) and
 * returns its full length through *len. */
int fhe_ctx_trace(fhe_ctx *ctx, int enable);
int fhe_ctx_trace_read(fhe_ctx *ctx, char *buf, size_t cap, size_t *len);

/* ---- ABFT detector around the forward NTT ------------------------- */
/* Weighted-checksum ECC of rfhe_framewk/src/negaclic_ntt.py:130-149: with weights w (generate_weights,
 * :7-13: w[i] = (i % p + 1) + (i / p + 1), p = 2^floor(log_n / 2)) and w_hat = V^-T w, a correct
 * transform satisfies sum_i w_i a_i = sum_j w_hat_j a_hat_j (mod q).  w_hat is obtained on the device
 * from the forward transform itself: w_hat = N^-1 * NTT(w_0, -w_{N-1}, ..., -w_1), already in the
 * engine's bit-reversed order.  One weight set per limb of `t`. */
int fhe_abft_create(fhe_ctx *ctx, const fhe_ntt_tables *t, fhe_abft **out);
int fhe_abft_destroy(fhe_abft *a);
/* side 0: d_out[unit] = sum_i w_i x_i (input side); side 1: N^-1 sum_j w'_j x_j (output side) */
int fhe_abft_checksum(fhe_ctx *ctx, const fhe_abft *a, int side, const uint64_t *d_data, uint64_t *d_out, size_t n_poly,
                      size_t limbs, size_t start_idx, void *stream);
/* Forward NTT with the detector around it: d_flags[unit] (uint32) = 1 where the two checksums differ,
 * i.e. where a fault hit the transform of that limb-polynomial (faults already present in the input
 * are, by construction, not flagged).  For N >= 32 the two checksums are accumulated inside the
 * transform's own passes (from the registers just loaded / the words about to be stored): no extra
 * sweep over the data.  The detector object owns scratch: one call at a time per fhe_abft. */
int fhe_ntt_forward_checked(fhe_ctx *ctx, uint64_t *d_data, const fhe_ntt_tables *t, const fhe_abft *a, size_t n_poly,
                            size_t limbs, size_t start_idx, uint32_t *d_flags, void *stream);
/* The same detector phase by phase, where the reference checks its four-step flow: batch_check of the column transforms,
 * check_inter around the twiddle step, batch_check of the row transforms (rfhe_framewk/src/ntt_test/relia_ntt_sim.cpp:235-292,
 * 331-355; the per-multiply equality of reliability_test/four_step_ntt_prot.py:185-194).  The engine's two launches are that
 * flow (column transforms; row transforms with the twiddle folded into their butterflies), so three flags per
 * limb-polynomial, d_flags[3*unit + k]:
 *   k = 0  column pass : sum w x over the words it loaded  !=  sum u y over the words it stored       (u = P1^-T w)
 *   k = 1  hand-off    : sum u y as stored by the column pass  !=  as loaded by the row pass (corruption between the launches)
 *   k = 2  row pass    : sum u y over the words it loaded  !=  sum w^ X over the words it stored
 * A fault raises the flag of the phase it hit and no other.  Two-launch sizes (N >= 2^13); smaller transforms are one
 * launch = one phase: fhe_ntt_forward_checked. */
int fhe_ntt_forward_checked_phases(fhe_ctx *ctx, uint64_t *d_data, const fhe_ntt_tables *t, const fhe_abft *a, size_t n_poly,
                                   size_t limbs, size_t start_idx, uint32_t *d_flags, void *stream);
/* Test hook: a soft error INSIDE a pass of the next fhe_ntt_forward_checked_phases call -- XOR bit `bit` of word `lds_word`
 * (modulo the image size) of workgroup `workgroup`'s LDS image between the pass's first two register steps; pass 0 = column
 * pass, 1 = row pass, < 0 clears it.  One shot.  (The reference injects into butterfly results, relia_ntt_sim.cpp:189-194.) */
int fhe_ctx_inject_fault_in_pass(fhe_ctx *ctx, int pass, uint32_t workgroup, uint32_t lds_word, int bit);
/* Test hook for the detector: XOR bit `bit` of word `idx` of the buffer BETWEEN the two launches of the
 * next two-pass forward/inverse transform issued on this context (one shot; idx < 0 clears it).  This is
 * the in-flight analogue of the host-side flips of reliability_test/ntt_test.cu:104-135. */
int fhe_ctx_inject_fault(fhe_ctx *ctx, long long idx, int bit);

/* ---- fault injection ---------------------------------------------------------------- */
/* _flip_bit_kernel<<<1,1>>> (reliability_test/dotprod_test.cu:31-33,55): data[idx] ^= 1 << bit */
int fhe_flip_bit(fhe_ctx *ctx, uint64_t *d_data, uint64_t idx, int bit, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FHE_MI355X_H */
