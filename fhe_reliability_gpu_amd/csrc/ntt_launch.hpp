// ntt_launch.hpp -- host-visible launch entry points of the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "ntt_plan.hpp"

namespace fhe {

// Batched in-place transform of a.units limbs of 2^logn points, all on `path`.
// geo: 1 = 16-column tiles (default), 0 = widest column tile (kept for 2^16 tuning runs)
// which: -1 whole transform; 0 / 1 = only the first / second launch of a two-pass size
// resident (opt-in): sizes whose limb fits a CU's LDS (2^13, 2^14) run as ONE LDS-resident pass when the whole transform is asked for
hipError_t launch_ntt(hipStream_t st, const PassArgs &a, int logn, bool inverse, int path, int geo = 1, int which = -1, bool resident = false);

// Natural-order transform (cyclic NTT of motivation/ntt.py:8-32 / the four-step flow): a.src = natural-order input,
// a.data = natural-order output (may alias), tmp = hand-off buffer of the same layout for the two-launch sizes.  The limbs'
// INVERSE table slot must hold the cyclic table (capi.cpp build_tables, GS mode).  2^5 .. 2^20.
bool ntt_gs_supported(int logn);
hipError_t launch_ntt_gs(hipStream_t st, const PassArgs &a, u64 *tmp, int logn, int path);

// Forward transform of a.data = [parts <= 3][limbs][N] whose last pass writes out_h[l] = (a_h[l] - NTT(data_h[l])) * scal[l]
// (+ add_h[l]) mod q_l instead of the transformed words (mod-down / rescale tail fused into the row pass); a_h = a + h * a_stride
// words, out / add per part, scal per limb of the run (ntt_kernels.hip k_ntt_row_subscale).  a.data is scratch afterwards.
struct RowEpiArgs {
    u64 *out[3];
    const u64 *add[3];
    const u64 *a;
    u64 a_stride;
    const u64 *scal;
    const u64 *pre = nullptr;   // optional per-limb factor applied to the transformed words first (t of the BGV forms)
    u32 galois = 0;             // != 0: the addends are sigma_k(add[h]), read through the NTT-domain Galois map (a rotation's sigma(c0))
    u32 galois_a = 0;           // != 0 (with galois): `a` is read through the same map too -- a hoisted rotation's sums, formed in the un-rotated frame
    const u64 *scal2 = nullptr; // optional second per-limb factor applied LAST: out = ((a - X) scal + add) scal2 (mod-down and rescale in one tail)
};
// Optional second input of the transform (two-launch sizes): the column pass loads data_h[l] + w[l] * y_h (mod q_l) instead of
// data_h[l]; y_h = y + h * y_stride words is ONE limb per part (residues of some other prime, any 64-bit words), w = per TABLE limb
// factors in the limb's twiddle encoding.  By linearity NTT(data + w y) = NTT(data) + w NTT(y): two subtractions share one transform.
struct ColAddSrc {
    const u64 *y;
    u64 y_stride;
    const Tw *w;
};
bool ntt_subscale_supported(int logn);
hipError_t launch_ntt_subscale(hipStream_t st, const PassArgs &a, const RowEpiArgs &ep, int logn, int path, const ColAddSrc *add_src = nullptr);

// c = a * b mod (x^N + 1, q_l): forward column passes, one launch that finishes both forward transforms,
// multiplies and starts the inverse, inverse column pass.  a and b are scratch afterwards.
bool polymul_fused_supported(int logn);
hipError_t launch_polymul(hipStream_t st, const PassArgs &a, u64 *b, u64 *c, int logn, int path);

// forward transform with the ABFT checksums fused into the passes (ntt_kernels.hip): sum_in / sum_out are
// [units][tin] / [units][tout] partial sums (units in [poly][limb] order; tin, tout from ntt_checked_tiles);
// win / wout = [table limbs][N] weights in twiddle encoding (used by ArithU64 limbs), wout8 = output-side weights as
// residues (ArithF64 limbs; their input-side weights are computed in place).  `which` as in launch_ntt.
bool ntt_checked_supported(int logn);
void ntt_checked_tiles(int logn, u32 *tin, u32 *tout);
hipError_t launch_ntt_checked(hipStream_t st, const PassArgs &a, const Tw *win, const Tw *wout, const u64 *wout8, u64 *sum_in, u64 *sum_out,
                              int logn, int path, int which = -1);

// per-phase detector (two-launch sizes): the column pass accumulates sum w x over what it loads and sum u y over what it
// stores, the row pass sum u y over what it loads and sum w^ X over what it stores (weights: capi_abft.cpp).  One PhaseArgs
// per launch; sum_a / sum_b = that launch's [units][tiles] partial sums.  fault_*: test hook, a bit flip in the LDS image of
// workgroup fault_block of pass fault_pass between its first two phases (fault_pass < 0: off).
struct PhaseArgs {
    const Tw *win, *umid, *wout;
    const u64 *umid8, *wout8;
    int logp;
    u64 *sum_a, *sum_b;
    int fault_pass;
    u32 fault_block, fault_word;
    int fault_bit;
};
bool ntt_phases_supported(int logn);
int ntt_column_stages(int logn);   // PC of the size's plan (0 for single-launch sizes)
hipError_t launch_ntt_phases(hipStream_t st, const PassArgs &a, const PhaseArgs &p1, const PhaseArgs &p2, int logn, int path, int which = -1);
// flags[unit*3 + {0,1,2}] = column pass / hand-off / row pass checks failed; s_in, s_mid1 have `tc` partial sums per unit,
// s_mid2, s_out `tr`
hipError_t launch_compare_phases(hipStream_t st, u32 *flags, const u64 *s_in, const u64 *s_mid1, u32 tc, const u64 *s_mid2, const u64 *s_out, u32 tr,
                                 const LimbParams *lp, u32 limb0, u32 limbs, u32 units);

// packed hand-off between the two launches (forward 2^16, FP64 limbs): when PassArgs::scratch is set (ntt_packed_scratch_words()
// 64-bit words per unit) the intermediate travels as 50-bit residues in 16x16 blocks instead of 8-byte words in place
bool ntt_packed_supported(int logn, bool inverse, int path);
size_t ntt_packed_scratch_words();

// ---- ntt_fused.hip: single-launch variant for two-pass sizes --------------------
bool fused_supported(int logn);
size_t fused_ctl_bytes(u32 units);
// variant = handoff * 2 + stream_hint, handoff: 1 sc1 loads, 2 nt loads, 3 acquire + plain loads
hipError_t launch_ntt_fused(hipStream_t st, const PassArgs &a, int logn, bool inverse, int path, u32 *ctl, u32 dist, u32 wgs,
                            int variant, u32 skip_teams);

// ---- aux_kernels.hip --------------------------------------------------------
struct PointwiseArgs {
    u64 *c;
    const u64 *a;
    const u64 *b;
    const LimbParams *lp;
    u32 limb0, limbs, units, poly_stride;
    int logn;
};
// c = a*b mod q (accumulate = false) or c = (c + a*b) mod q (accumulate = true), per limb
hipError_t launch_modmul(hipStream_t st, const PointwiseArgs &p, bool accumulate);
// tensor product of two two-part ciphertexts, NTT domain, per limb: d0 = a0 b0, d1 = a0 b1 + a1 b0, d2 = a1 b1
// (the coefficient-wise core of phantom::multiply, reliability_test/dotprod_test.cu:113); every operand [limbs][N]
struct TensorArgs {
    u64 *d0, *d1, *d2;
    const u64 *a0, *a1, *b0, *b1;
    const LimbParams *lp;
    u32 limb0, limbs;
    int logn;
};
hipError_t launch_tensor(hipStream_t st, const TensorArgs &p);
// inner sum of a BSGS matrix-vector product: out_h = sum_b diag[b] * R_b,h per limb (aux_kernels.hip k_diag_mac); diag = [n1][limbs][N],
// R_0 = (x0, x1), R_b = rot + (b - 1) * 2 * limbs * N ([n1 - 1][2][limbs][N])
struct DiagMacArgs {
    u64 *out0, *out1;
    const u64 *diag, *x0, *x1, *rot;
    const LimbParams *lp;
    u32 limb0, limbs, n1;
    int logn;
};
hipError_t launch_diag_mac(hipStream_t st, const DiagMacArgs &a);
hipError_t launch_modadd(hipStream_t st, const PointwiseArgs &p);
hipError_t launch_modsub(hipStream_t st, const PointwiseArgs &p);
constexpr int SCALAR_MAX_LIMBS = 64;
struct ScalarVec {
    u64 v[SCALAR_MAX_LIMBS];
};
// c = a * mul[l] + add[l] mod q_l (per-limb scalars, reduced by the caller)
hipError_t launch_scalar_affine(hipStream_t st, const PointwiseArgs &p, const ScalarVec &mul, const ScalarVec &add);
// data[idx] ^= 1 << bit  (reliability_test/dotprod_test.cu:31-33)
hipError_t launch_flip_bit(hipStream_t st, u64 *data, u64 idx, int bit);
// modulus + floor(2^128/q) for the Barrett helpers, passed by value
struct ModConst {
    u64 q, r0, r1;
};
// per unit of 2^logn words: dst[i] = src[bitrev(i)] (* scale mod q when do_scale).  dst != src.
hipError_t launch_bitrev_scale(hipStream_t st, u64 *dst, const u64 *src, int logn, u32 units, const ModConst &mc, u64 scale,
                               bool do_scale);
// per vector of rows x cols words: dst[c][r] = src[r][c] (* w^(r c) mod q when tlo != nullptr, w^e = tlo[e mod 2^lo_bits] * thi[e >> lo_bits])
hipError_t launch_transpose_tw(hipStream_t st, u64 *dst, const u64 *src, u32 rows, u32 cols, u32 n_vec, const ModConst &mc, const u64 *tlo, const u64 *thi,
                               int lo_bits);
// out_h = (a_h - b_h) * scal[l] (+ add_h) mod q_l over `limbs` limbs from table index limb0, for one half
// (out1 == nullptr) or both halves of a key switch in one launch; a_h = a + h * a_stride, b_h = b + h * b_stride (in words);
// a == nullptr / b == nullptr: that operand is zero
struct SubScaleArgs {
    u64 *out0, *out1;
    const u64 *a, *b, *add0, *scal;
    u64 a_stride, b_stride;
    const LimbParams *lp;
    u32 limb0, limbs;
    int logn;
    const u64 *add1 = nullptr;
};
hipError_t launch_sub_scale(hipStream_t st, const SubScaleArgs &p);
// out[unit] = sum_i w[limb][i] * x[unit][i] mod q_limb (one workgroup per limb-polynomial): the weighted
// checksum of the reference's ECC (rfhe_framewk/src/negaclic_ntt.py:130-149); scal[limb] multiplies the sum
hipError_t launch_weighted_checksum(hipStream_t st, u64 *out, const u64 *x, const u64 *w, const u64 *scal, const LimbParams *lp,
                                    u32 limb0, u32 limbs, u32 units, u32 poly_stride, int logn);
// flags[unit] = a[unit] != b[unit]
hipError_t launch_compare_flags(hipStream_t st, u32 *flags, const u64 *a, const u64 *b, u32 units);
// flags[unit] = (sum of a[unit][0..ta) mod q_l) != (sum of b[unit][0..tb) mod q_l)  (unit = poly * limbs + l)
hipError_t launch_compare_sums(hipStream_t st, u32 *flags, const u64 *a, u32 ta, const u64 *b, u32 tb, const LimbParams *lp, u32 limb0,
                               u32 limbs, u32 units);
// Galois automorphism x -> x^k: coefficient domain (sign-aware scatter) and NTT domain (gather)
hipError_t launch_automorphism(hipStream_t st, u64 *dst, const u64 *src, const LimbParams *lp, u32 limb0, u32 limbs, u32 units,
                               int logn, u32 k);
// optional second (destination, source) pair of the same shape in the same launch
hipError_t launch_automorphism_ntt(hipStream_t st, u64 *dst, const u64 *src, u32 units, int logn, u32 k, u64 *dst_b = nullptr,
                                   const u64 *src_b = nullptr);

// ---- baseconv_kernels.hip ------------------------------------------------------
struct BaseConvPlanDev {
    int m, k;
    const u64 *mod_in;        // m
    const u64 *mod_out;       // k
    // exact conversion, mixed-radix digits c_j = r_j*A_j - sum_{l<j} c_l*D_lj (mod p_j) and x = sum_l c_l*E_lo (mod q_o):
    const Tw *dig;            // m x m   [j*m+j] = A_j = (p_0..p_{j-1})^-1 mod p_j;  [l*m+j], l<j = D_lj = (p_l..p_{j-1})^-1 mod p_j
    const Tw *hor;            // m x k   E_lo = p_0..p_{l-1} mod q_o
    const Tw *fp_in, *fp_out; // m, k    {bits((double)q), bits(1.0/q)} per modulus (ArithF64 context)
    int f64;                  // 1: every modulus < 2^50, constants encoded for ArithF64; 0: Shoup pairs (ArithU64)
    const u64 *fast_coef;     // m x k   (Phat_j * inv_j) mod q_o   (rfhe_framewk/src/baseConv.py:17-29)
    const u64 *fast_coef_shoup;
    // m <= 16: the same constants laid out as the fixed-size kernels keep them in LDS, so that a workgroup stages them with ONE batch
    // of 16-byte loads: img_head = dig [m][m], fp_in [m], mod_in [m] (two per entry); img_out = per output o: hor[0..m-1][o], fp_out[o],
    // {mod_out[o], 0} -- m + 2 entries per output, a slice of outputs is a contiguous range
    const Tw *img_head;
    const Tw *img_out;
    const u32 *rows_identity; // 0, 1, 2, ... (64 entries): the row map of a job that has none
};
// output limb o is written at limb index o (o < gap_at) or o + gap: lets a key-switch digit's extension land
// in the [M][N] layout with the digit's own limbs skipped
struct BcJob;
hipError_t launch_baseconv_exact(hipStream_t st, u64 *out, const u64 *in, const BaseConvPlanDev &pl, u64 N, u32 gap_at = 0xFFFFFFFFu,
                                 u32 gap = 0, const u32 *in_rows = nullptr);
struct BcJob {
    BaseConvPlanDev pl;
    const u64 *in;
    u64 *out;
    u32 gap_at, gap;
    const u32 *in_rows = nullptr;   // optional: input limb j sits at row in_rows[j] of `in` (rows of N words) instead of row j --
                                    // the all-gathered slabs of a sharded key switch are padded per rank
};
// m_mask: bit (m - 1) set for every input size m <= 16 among the jobs (one straight-line launch per size); 0 = the runtime-m kernels
hipError_t launch_baseconv_exact_jobs(hipStream_t st, const BcJob *dev_jobs, u32 n_jobs, int max_m, int max_k, bool f64, u64 N, u32 m_mask = 0);
// key-switch inner product over all digits and both key halves (aux_kernels.hip k_ks_mac).  Rows are the limbs this
// rank owns: cn ciphertext limbs (table limbs clo ..) followed by the owned special limbs (table limb = row + sp_shift);
// one device owns everything: M = L + K, cn = L, clo = 0, sp_shift = 0.
struct KsMacArgs {
    u64 *acc;            // [2][M][N]
    const u64 *ext;      // [dnum][M][N]  extended digits, NTT form (the digit's own limbs unused)
    const u64 *c;        // [cn][N]       owned input limbs, NTT form
    const u64 *evk;      // [dnum][2][M][N]
    const LimbParams *lp;
    u32 L, M, dnum, alpha;
    int logn;
    u32 cn, clo, sp_shift;
};
hipError_t launch_ks_mac(hipStream_t st, const KsMacArgs &a);
// n <= 4 keys on the same digits: acc[r] = sum_d ext_d * evk[r][d] (a.acc / a.evk unused)
struct KsMacMultiArgs {
    KsMacArgs a;
    u32 n;
    const u64 *evk[4];
    u64 *acc[4];
};
hipError_t launch_ks_mac_multi(hipStream_t st, const KsMacMultiArgs &m);
// The same inner product with the LAST pass of the extended limbs' forward transform fused in (ntt_kernels.hip k_ks_rowmac):
// `ext` then holds what the transform's first launch left (the column pass's lazy words; for single-pass sizes the
// base-extension output itself), one workgroup per (owned limb, row tile) runs the row pass of every digit's limb in LDS,
// multiplies by the two key halves and keeps the sums in registers: the transformed digits are never written or re-read.
bool ks_rowmac_supported(int logn);
hipError_t launch_ks_rowmac(hipStream_t st, const KsMacArgs &a, bool has_f64, bool has_u64);
hipError_t launch_bconv_fast(hipStream_t st, u64 *out, const u64 *in, const BaseConvPlanDev &pl, u64 N);
hipError_t launch_crt_garner(hipStream_t st, u64 *x_lo, u64 *x_hi, const u64 *residues, const u64 *moduli,
                             const u64 *ratios, const u64 *pref_lo, const u64 *pref_hi, const u64 *inv_pref, int m, u64 N);
hipError_t launch_bsgs_hadamard(hipStream_t st, u64 *y, const u64 *M_blocks, const u64 *v, int k, int bs,
                                const ModConst *mc /* nullptr: int64 wrap-around like the reference */);

} // namespace fhe
