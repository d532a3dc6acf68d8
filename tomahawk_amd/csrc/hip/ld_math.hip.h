// Pair statistics on the device: plane products -> contingency cells -> D, D',
// r, r2, Fisher's exact P, chi-squared, flags -> filters -> compacted records.
//
// Device counterpart of twk_ld_engine::PhasedMath / UnphasedMath /
// ChiSquaredUnphasedTable / ChooseF11Calculate (lib/ld/ld_engine.cpp:1162-1740)
// and kt_fisher_exact (lib/fisher_math.cpp:183-267).  All floating point is
// FP64 and this translation unit is compiled with -ffp-contract=off so that
// products and sums round exactly where the reference's SSE4.2 build rounds
// them (filter decisions on r2 and D' are then bit-identical; lgamma / exp /
// pow / acos / cos come from the device math library and agree with glibc to
// a few ulp, well inside the 1e-6 relative budget).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../../include/twk_hip.h"

namespace twk {

// lib/ld/ld_engine.h:33-37
#define TWK_D_LOW_AC        5
#define TWK_D_INVALID_HWE   1e-4
#define TWK_D_LONG_RANGE    500e3
#define TWK_D_MIN_ALLELES   5
#define TWK_D_ROUNDING_ERR  0.00001

struct VariantMeta {       // SoA view of twk_hip_variant_meta on the device
	const uint32_t* ac;
	const uint32_t* an;
	const uint32_t* pos;
	const uint32_t* rid;
	const uint32_t* missing;
	const double*   hwe;
};

// ---- Fisher's exact test: fisher_math.cpp:183-267 ------------------------------
// lbinom (fisher_math.cpp:183-187) is three lgamma calls, and the test evaluates it nine times per
// re-synchronisation of its recurrence (every 11th term) and for every verified starting point.  The
// arguments are integers <= 2N + 1, so the values come from a table lf[i] = lgamma(i + 1) filled once per
// problem with the same lgamma (k_build_lfact): the same bits, three loads instead of three evaluations.
struct LFact { const double* lf; int n; };       // lf[0..n)
// lgamma(i + 1); arguments outside the table (the wrapped counts of TWK_HIP_OPT_REF_COMPAT narrow to negative ints) take lgamma itself
__device__ inline double d_lgamma1(const LFact& t, int i) { return (unsigned)i < (unsigned)t.n ? t.lf[i] : lgamma((double)(i + 1)); }
__device__ inline double d_lbinom(const LFact& t, int n, int k) {
	if (k == 0 || n == k) return 0;
	return d_lgamma1(t, n) - d_lgamma1(t, k) - d_lgamma1(t, n - k);
}
__device__ inline double d_hypergeo(const LFact& t, int n11, int n1_, int n_1, int n) {
	return exp(d_lbinom(t, n1_, n11) + d_lbinom(t, n - n1_, n_1 - n11) - d_lbinom(t, n, n_1));
}
__global__ void k_build_lfact(double* __restrict__ lf, int n) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) lf[i] = lgamma((double)i + 1.0);
}
// The pmf along one margin-fixed family of 2x2 tables, as kt_fisher_exact walks it (fisher_math.cpp:206-229):
// the table is fixed by its margins (row1, col1, total) and the walk moves its upper-left cell k one step at a
// time.  Neighbouring cells are reached by the ratio pmf(k+1)/pmf(k) = (row1-k)(col1-k) / ((k+1)(k+1+total-row1-col1)),
// but every 11th cell - and the cell where that last factor would vanish - is evaluated from the log-binomials again, so
// rounding never accumulates over more than ten ratios.  The order of the multiplications and divisions below is the
// reference's: P then agrees with it to the last bits, including which terms its stopping rule lets in.
struct TableWalk {
	int k, row1, col1, total;          // current upper-left cell and the margins
	double pmf;                        // probability of the current table
	__device__ inline double reset(const LFact& t, int k0, int r1, int c1, int n) {      // fix the margins, evaluate cell k0 in full
		k = k0; row1 = r1; col1 = c1; total = n;
		return pmf = d_hypergeo(t, k, row1, col1, total);
	}
	__device__ inline double move_to(const LFact& t, int to) {                            // pmf of cell `to` (same margins)
		const int slack = total - row1 - col1;                                            // lower-right cell = to + slack
		if (to % 11 != 0 && to + slack != 0) {
			if (to == k + 1) {
				pmf *= (double)(row1 - k) / to * (col1 - k) / (to + slack);
				k = to;
				return pmf;
			}
			if (to == k - 1) {
				pmf *= (double)k / (row1 - to) * (k + slack) / (col1 - to);
				k = to;
				return pmf;
			}
		}
		k = to;
		return pmf = d_hypergeo(t, k, row1, col1, total);
	}
};
// How far below q (in e-folds) a tail walk may start: everything further out is at most (max - min) terms, each below
// q e^-K.  K = 40: the part of P that is not summed is below 4e-18 (max - min) q, < 1e-10 of P at 2e7 haplotypes.  (A
// support-dependent K = 17.5 + ln(max - min) - 2.5e-8 of P, a third fewer terms - was measured: 5 % of the kernel, and
// in the underflow band, where the reference's result hangs on the rounding of its recurrence (k_ld_fisher_group below),
// the nearer starting point no longer reproduced it bit for bit: 9 of 1.2 M records.  Not worth it.)
__device__ inline double d_fisher_skip_exponent(int) { return 40.0; }
// Two-sided P only (left / right tails are not stored in the record).
__device__ inline double d_fisher_two(const LFact& t, int n11, int n12, int n21, int n22) {
	int i, j, max, min;
	double p, q, left, right;
	TableWalk w;
	const int n1_ = n11 + n12, n_1 = n11 + n21, n = n11 + n12 + n21 + n22;
	max = (n_1 < n1_) ? n_1 : n1_;
	min = n1_ + n_1 - n;
	if (min < 0) min = 0;
	if (min == max) return 1.;
	q = w.reset(t, n11, n1_, n_1, n);
	// The reference walks both tails from the ends of the support (min, max) inwards until the
	// terms reach q: up to min(n1_, n_1) steps per record, almost all of them over terms that are
	// zero or tens of orders of magnitude below q.  Start each walk closer in instead, at a point
	// that is *verified* (one log-pmf evaluation) to lie below q by a factor e^-40: the pmf is
	// monotone out there, so everything skipped sums to < (max - min) * 4e-18 * q, i.e. < 1e-10
	// of the result (P >= q) even at 2e7 haplotypes -- far inside the 1e-6 bar; the walk itself, its
	// re-synchronisation every 11th step and its stopping rule are unchanged.  The candidate point
	// comes from the normal approximation of the log-pmf, -(s - mean)^2 / (2 sd^2) relative to the
	// mode: a term e^-K below q lies sqrt(dev^2 + 2 K sd^2) from the mean (dev = |n11 - mean|), a
	// few sd beyond n11's own distance for a significant table instead of a fixed 12 sd; the
	// approximation only proposes, the exact log-pmf decides, and a point that fails moves outwards.
	int i0 = min, j0 = max;
	if (q > 0 && max - min > 64) {
		const double lq = log(q), nn = (double)n;
		const double mean = (double)n1_ * (double)n_1 / nn;
		const double sd = sqrt(mean * ((nn - n1_) / nn) * ((nn - n_1) / (nn - 1.0)));
		const double lden = d_lbinom(t, n, n_1);
		const double dev = fabs((double)n11 - mean);
		const double K = d_fisher_skip_exponent(max - min);
		const double D0 = sqrt(dev * dev + 2.0 * (K + 8.0) * sd * sd) + 4.0;      // K + 8 proposed, K required
		double D = D0;
		for (int k = 0; k < 4; ++k, D = D * 1.5 + 8.0) {
			const double sf = floor(mean - D);
			if (sf <= (double)min) break;
			const int s = (int)sf;
			if (d_lbinom(t, n1_, s) + d_lbinom(t, n - n1_, n_1 - s) - lden <= lq - K) { i0 = s; break; }
		}
		D = D0;
		for (int k = 0; k < 4; ++k, D = D * 1.5 + 8.0) {
			const double sf = ceil(mean + D);
			if (sf >= (double)max) break;
			const int s = (int)sf;
			if (d_lbinom(t, n1_, s) + d_lbinom(t, n - n1_, n_1 - s) - lden <= lq - K) { j0 = s; break; }
		}
	}
	p = w.move_to(t, i0);
	for (left = 0., i = i0 + 1; p < 0.99999999 * q && i <= max; ++i)
		left += p, p = w.move_to(t, i);
	if (p < 1.00000001 * q) left += p;
	p = w.move_to(t, j0);
	for (right = 0., j = j0 - 1; p < 0.99999999 * q && j >= 0; --j)
		right += p, p = w.move_to(t, j);
	if (p < 1.00000001 * q) right += p;
	double two = left + right;
	if (two > 1.) two = 1.;
	return two;
}

// ---- flags shared by both maths: ld_engine.cpp:1244-1255 / 1674-1684 -----------------
__device__ inline uint32_t d_common_flags(const VariantMeta& vm, uint32_t A, uint32_t B,
                                          const double cnt[4], double R2) {
	uint32_t c = 0;
	if (vm.ac[A] < TWK_D_LOW_AC) c |= 1u << 10;
	if (vm.ac[B] < TWK_D_LOW_AC) c |= 1u << 11;
	if (cnt[0] < 1 || cnt[1] < 1 || cnt[2] < 1 || cnt[3] < 1) c |= 1u << 3;
	if (R2 > 0.99) c |= 1u << 4;
	if (vm.an[A]) c |= 1u << 8;
	if (vm.an[B]) c |= 1u << 9;
	const int32_t diff = (int32_t)vm.pos[A] - (int32_t)vm.pos[B];
	const bool same = vm.rid[A] == vm.rid[B];
	if (abs(diff) > TWK_D_LONG_RANGE && same) c |= 1u << 2;
	if (same) c |= 1u << 1;
	if (vm.hwe[A] < TWK_D_INVALID_HWE) c |= 1u << 12;
	if (vm.hwe[B] < TWK_D_INVALID_HWE) c |= 1u << 13;
	return c;
}

#define TWK_N11_IN_PAD 0x80000000u     // internal: rec->_pad holds Fisher's n11 (see d_phased_math)

// ---- PhasedMath: ld_engine.cpp:1162-1310 ------------------------------------------------
// c0,c1,c4,c5 = alleleCounts[0],[1],[4],[5].  Returns true if the pair survives.
__device__ inline bool d_phased_math(uint64_t c0, uint64_t c1, uint64_t c4, uint64_t c5,
                                     const VariantMeta& vm, uint32_t A, uint32_t B,
                                     const twk_hip_filters& f, twk_hip_record* rec) {
	const uint64_t total = c0 + c4 + c1 + c5;
	if (total < TWK_D_MIN_ALLELES) return false;
	if (c0 < c5) { if (c4 + c1 + c0 < 5) return false; }
	else         { if (c5 + c4 + c1 < 5) return false; }
	// Screen before the nine divisions below: r2 = (c0 c5 - c1 c4)^2 / ((c0+c4)(c1+c5)(c0+c1)(c4+c5)).  The
	// products and sums below are evaluated in FP64: c0*c5 and c1*c4 are exact while the cells stay below
	// 2^26 (N < 16.7 M haplotypes), the denominator is not, and at the largest sample counts set_problem
	// accepts neither is - each operation carries <= 1 ulp, so dn^2 and den are good to ~1e-15 relative.
	// The screen only rejects a pair whose r2 computed this way lies more than a part in 1e6 below the
	// cut-off: nine orders of magnitude more than the rounding error of either evaluation, so such a pair
	// cannot pass the reference's (rounded) test further down, and nothing else below has an effect.  Pairs
	// inside the band minR2 * (1 +- 1e-6) always go through the reference's formula
	// (tests: test_r2_screen_agrees_at_the_cutoff).
	if (f.minR2 > 1e-6 && c0 < (1ull << 62)) {       // (a wrapped REFREF of TWK_HIP_OPT_REF_COMPAT goes through the reference's own arithmetic)
		const double dn = (double)c0 * (double)c5 - (double)c1 * (double)c4;
		const double den = ((double)c0 + (double)c4) * ((double)c1 + (double)c5) * (((double)c0 + (double)c1) * ((double)c4 + (double)c5));
		if (dn * dn < f.minR2 * (1.0 - 1e-6) * den) return false;
	}
	const double T = (double)total;
	const double pA = (double)c0 / T, qA = (double)c1 / T, pB = (double)c4 / T, qB = (double)c5 / T;
	const double D = pA * qB - qA * pB;
	if (D == 0) return false;
	const double g0 = ((double)c0 + (double)c4) / T;
	const double g1 = ((double)c1 + (double)c5) / T;
	const double h0 = ((double)c0 + (double)c1) / T;
	const double h1 = ((double)c4 + (double)c5) / T;
	const double R2 = D * D / (g0 * g1 * h0 * h1);
	if (R2 < f.minR2 || R2 > f.maxR2) return false;
	double dmax;
	if (D >= 0) dmax = g0 * h1 < h0 * g1 ? g0 * h1 : h0 * g1;
	else        dmax = g0 * g1 < h0 * h1 ? -g0 * g1 : -h0 * h1;
	const double Dprime = D / dmax;
	if (Dprime < f.minDprime || Dprime > f.maxDprime) return false;
	// Fisher's exact test (:1221-1231) runs in k_ld_fisher on the compacted survivors
	rec->idxA = A; rec->idxB = B; rec->_pad = 0;
	rec->cnt[0] = (double)c0; rec->cnt[1] = (double)c1; rec->cnt[2] = (double)c4; rec->cnt[3] = (double)c5;
	rec->D = D; rec->Dprime = Dprime; rec->R = sqrt(R2); rec->R2 = R2; rec->P = 0;
	rec->ChiSqFisher = T * R2; rec->ChiSqModel = 0;
	rec->flags = d_common_flags(vm, A, B, rec->cnt, R2) | 1u;
	// A REFREF count that wrapped (TWK_HIP_OPT_REF_COMPAT) no longer fits the double: the reference hands
	// Fisher's test the uint64 narrowed to int (ld_engine.cpp:1222, fisher_math.cpp:231), i.e. its low 32
	// bits; they ride in _pad to k_ld_fisher, which clears the marker bit again.
	if (c0 >= (1ull << 53)) { rec->_pad = (uint32_t)c0; rec->flags |= TWK_N11_IN_PAD; }
	return true;
}

// ---- ChiSquaredUnphasedTable: ld_engine.cpp:1562-1588 -------------------------------------
// o = {0, 1+4, 5, 16+64, hets, 21+69, 80, 81+84, 85}
__device__ inline double d_chisq_unphased(const double o[9], double total, double target, double p, double q) {
	const double f12 = p - target;
	const double f21 = q - target;
	const double f22 = 1 - (target + f12 + f21);
	double e[9];
	e[0] = total * (target * target);
	e[1] = 2 * total * target * f12;
	e[2] = total * (f12 * f12);
	e[3] = 2 * total * target * f21;
	e[4] = 2 * total * f12 * f21 + 2 * total * target * f22;
	e[5] = 2 * total * f12 * f22;
	e[6] = total * (f21 * f21);
	e[7] = 2 * total * f21 * f22;
	e[8] = total * (f22 * f22);
	double s = 0;
#pragma unroll
	for (int k = 0; k < 9; ++k) {
		const double d = o[k] - e[k];
		s += e[k] > 0 ? (d * d) / e[k] : 0;
	}
	return s;
}

// x^3 correctly rounded (double-double through FMA).  The reference writes pow(x, 3.0); glibc's pow is
// correctly rounded in all but ~1e-5 of the cases, the device's pow() is only good to an ulp - and the sign
// of yN^2 - h2 (one or three real roots) is decided at that level when the cubic has a double root, which
// real haplotype-block data produces all the time (identical or complementary variants).
__device__ inline double d_cube(double x) {
	const double hi = x * x;
	const double lo = fma(x, x, -hi);            // x^2 = hi + lo exactly
	const double p = hi * x;
	const double e = fma(hi, x, -p);             // hi * x = p + e exactly
	return p + (e + lo * x);
}

// ---- ChooseF11Calculate: ld_engine.cpp:1590-1740 ---------------------------------------------
__device__ inline bool d_choose_f11(double total, double target, double p, double q, uint32_t pre_flags,
                                    const VariantMeta& vm, uint32_t A, uint32_t B,
                                    const twk_hip_filters& f, twk_hip_record* rec) {
	const double f11 = target;
	const double f12 = p - f11;
	const double f21 = q - f11;
	const double f22 = 1 - (f11 + f12 + f21);
	const double D = (f11 * f22) - (f12 * f21);
	const double R2 = (D * D) / (p * (1 - p) * q * (1 - q));
	if (R2 < f.minR2 || R2 > f.maxR2) return false;
	double cnt[4];
	cnt[0] = f11 * 2 * total;
	cnt[2] = f12 * 2 * total;
	cnt[1] = f21 * 2 * total;
	cnt[3] = f22 * 2 * total;
	if (cnt[0] < cnt[3]) { if (cnt[2] + cnt[1] + cnt[0] < 5) return false; }
	else                 { if (cnt[3] + cnt[2] + cnt[1] < 5) return false; }
	double dmax;
	if (D >= 0) dmax = p * (1.0 - q) < q * (1.0 - p) ? p * (1.0 - q) : q * (1.0 - p);
	else        dmax = p * q < (1 - p) * (1 - q) ? -p * q : -(1 - p) * (1 - q);
	const double Dprime = D / dmax;
	if (Dprime < f.minDprime || Dprime > f.maxDprime) return false;
	// Fisher's exact test on the rounded counts (:1655-1664) runs in k_ld_fisher
	rec->idxA = A; rec->idxB = B; rec->_pad = 0;
	rec->cnt[0] = cnt[0]; rec->cnt[1] = cnt[1]; rec->cnt[2] = cnt[2]; rec->cnt[3] = cnt[3];
	rec->D = D; rec->Dprime = Dprime; rec->R = sqrt(R2); rec->R2 = R2; rec->P = 0;
	rec->ChiSqModel = 0;
	rec->ChiSqFisher = (cnt[0] + cnt[2] + cnt[1] + cnt[3]) * R2;
	rec->flags = pre_flags | d_common_flags(vm, A, B, cnt, R2);
	return true;
}

// ---- UnphasedMath: ld_engine.cpp:1312-1560 ------------------------------------------------------
// cell order: {0, 1+4, 5, 16+64, hets, 21+69, 80, 81+84, 85}
__device__ inline bool d_unphased_math(const uint64_t c[9], const VariantMeta& vm, uint32_t A, uint32_t B,
                                       const twk_hip_filters& f, twk_hip_record* rec) {
	const uint64_t a0 = c[0], a14 = c[1], a5 = c[2], a1664 = c[3], hets = c[4], a2169 = c[5],
	               a80 = c[6], a8184 = c[7], a85 = c[8];
	const uint64_t total_u = a0 + a14 + a5 + a1664 + hets + a2169 + a80 + a8184 + a85;
	if (total_u < TWK_D_MIN_ALLELES) return false;
	if (hets == 0)   // no phase uncertainty: collapse to the 2x2 table (:1334-1348)
		return d_phased_math(2 * a0 + a14 + a1664, 2 * a80 + a1664 + a8184, 2 * a5 + a14 + a2169,
		                     2 * a85 + a8184 + a2169, vm, A, B, f, rec);

	const double total = (double)total_u;
	const double dh = (double)hets;
	const double P = ((double)(a0 + a14 + a5) * 2.0 + (double)(a1664 + hets + a2169)) / (2.0 * total);
	const double Q = ((double)(a0 + a1664 + a80) * 2.0 + (double)(a14 + hets + a8184)) / (2.0 * total);
	const double n11 = (double)(2 * a0 + a14 + a1664);
	const double minhap = n11 / (2.0 * total);
	const double maxhap = (n11 + dh) / (2.0 * total);
	// Screen before the cubic.  Whatever root the reference ends up with, it only keeps one inside
	// [minhap - 1e-5, maxhap + 1e-5] (the admissibility tests below, applied to the computed values), and from it
	// D = f11 f22 - f12 f21, which is f11 - P Q written out (f12 = P - f11, f21 = Q - f11, f22 = 1 - f11 - f12 - f21), and
	// r2 = D^2 / (P (1 - P) Q (1 - Q)).  (x - P Q)^2 over that interval is largest at an end, so no admissible root can
	// reach the cut-off when both ends fail it - with the same 1e-6 margin as the phased screen against the rounding of
	// either evaluation.  For unlinked variants the interval is centred on P Q with half width P (1 - P) Q (1 - Q)
	// (a quarter of the double-het frequency), so the screen rejects them whenever that product is below the cut-off:
	// always at the default r2 >= 0.1 (the product never exceeds 1/16).
	if (f.minR2 > 1e-6) {
		const double pq = P * Q, v = (P * (1.0 - P)) * (Q * (1.0 - Q));
		const double d_lo = (minhap - TWK_D_ROUNDING_ERR) - pq, d_hi = (maxhap + TWK_D_ROUNDING_ERR) - pq;
		const double bound = f.minR2 * (1.0 - 1e-6) * v;
		if (d_lo * d_lo < bound && d_hi * d_hi < bound) return false;
	}
	const double dee = -n11 * P * Q;
	const double cc = -n11 * (1.0 - 2.0 * P - 2.0 * Q) - dh * (1.0 - P - Q) + (2.0 * total * P * Q);
	const double b = 2.0 * total * (1.0 - 2.0 * P - 2.0 * Q) - 2.0 * n11 - dh;
	const double a = 4.0 * total;

	const double xN  = -b / (3.0 * a);
	const double d2  = ((b * b) - 3.0 * a * cc) / (9 * (a * a));
	const double yN  = a * d_cube(xN) + b * (xN * xN) + cc * xN + dee;
	const double yN2 = yN * yN;
	const double h2  = 4 * (a * a) * d_cube(d2);
	const double diff = yN2 - h2;
	const double lo = minhap - TWK_D_ROUNDING_ERR, hi = maxhap + TWK_D_ROUNDING_ERR;
	const double o[9] = { (double)a0, (double)a14, (double)a5, (double)a1664, dh, (double)a2169,
	                      (double)a80, (double)a8184, (double)a85 };

	// Double roots (small tables in perfect LD) make yN^2 and h2 agree to the last bit, and the sign of
	// `diff` - three real roots or one, and the one-root formula cannot see a double root - is then
	// decided by the rounding of pow(d2, 3.0).  glibc's pow is within 0.52 ulp, i.e. correctly rounded
	// except when the exact cube lies within ~0.02 ulp of a rounding midpoint; d_cube is correctly
	// rounded always, so the two agree on all but a few per cent of these knife-edge pairs (measured:
	// one table in 2e5 hostile records).  Treating the knife edge as "three roots" instead was tried and
	// is worse: the reference drops most such pairs (diff >= 0 by a hair, simple root inadmissible), and
	// following its sign reproduces that.
	if (diff < 0) {
		const double h = sqrt(h2);                  // pow(h2, 0.5): correctly rounded either way
		const double theta = acos(-yN / h) / 3.0;
		const double delta = sqrt(d2);
		const double alpha = xN + 2.0 * delta * cos(theta);
		const double beta  = xN + 2.0 * delta * cos(2.0 * M_PI / 3.0 + theta);
		const double gamma = xN + 2.0 * delta * cos(4.0 * M_PI / 3.0 + theta);
		int possible = 0;
		double best = 1.7976931348623157e308, chosen = alpha;
		if (alpha >= lo && alpha <= hi) { ++possible; best = d_chisq_unphased(o, total, alpha, P, Q); }
		if (beta >= lo && beta <= hi) {
			++possible;
			const double x = d_chisq_unphased(o, total, beta, P, Q);
			if (x < best) { chosen = beta; best = x; }
		}
		if (gamma >= lo && gamma <= hi) {
			++possible;
			const double x = d_chisq_unphased(o, total, gamma, P, Q);
			if (x < best) { chosen = gamma; best = x; }
		}
		if (possible == 0) return false;
		return d_choose_f11(total, chosen, P, Q, possible > 1 ? (1u << 5) : 0u, vm, A, B, f, rec);
	} else if (diff > 0) {
		const double sq = sqrt(yN2 - h2);
		const double t1 = 1.0 / (2.0 * a) * (-yN + sq);
		const double t2 = 1.0 / (2.0 * a) * (-yN - sq);
		const double number1 = t1 < 0 ? -pow(-t1, 1.0 / 3.0) : pow(t1, 1.0 / 3.0);
		const double number2 = t2 < 0 ? -pow(-t2, 1.0 / 3.0) : pow(t2, 1.0 / 3.0);
		const double alpha = xN + number1 + number2;
		if (!(alpha >= lo && alpha <= hi)) return false;
		return d_choose_f11(total, alpha, P, Q, 0u, vm, A, B, f, rec);
	} else {
		const double delta = pow((yN / 2.0 * a), (1.0 / 3.0));
		const double alpha = xN + delta;
		const double gamma = xN - 2.0 * delta;
		if (isnan(alpha) || isnan(gamma)) return false;
		int possible = 0;
		double best = 1.7976931348623157e308, chosen = alpha;
		if (alpha >= lo && alpha <= hi) { ++possible; best = d_chisq_unphased(o, total, alpha, P, Q); }
		if (gamma >= lo && gamma <= hi) {
			++possible;
			const double x = d_chisq_unphased(o, total, gamma, P, Q);
			if (x < best) { chosen = gamma; best = x; }
		}
		if (possible == 0) return false;
		return d_choose_f11(total, chosen, P, Q, 0u, vm, A, B, f, rec);
	}
}

// ---- plane products -> contingency cells ---------------------------------------------------
// How the rows of one variant are laid out and what the table is made from.
enum PlaneKind : int {
	PK_PHASED        = 0,  // 1 row : a                 (2N bits)
	PK_PHASED_MASKED = 1,  // 2 rows: a & ~m, m
	PK_UNPHASED      = 2,  // 2 rows: H, Q              (N bits)
	PK_UNPHASED_MASKED = 3 // 3 rows: H & ~M, Q & ~M, M
};
__host__ __device__ inline int planes_per_variant(int kind) {
	return kind == PK_PHASED ? 1 : (kind == PK_UNPHASED_MASKED ? 3 : 2);
}

struct TileView {
	const uint32_t* C;        // plane-product counts of the super-tile
	uint32_t ldc;
	const uint32_t* rowpop;   // popcount of every plane row (global row index)
	int kind;
	uint32_t n_samples;
	uint32_t a0, b0;          // first variant of the tile rows / cols (index into the plane set)
	const uint32_t* ids;      // plane-set index -> variant id; null: the plane set is in file order
};

// 2x2 table (alleleCounts[0],[1],[4],[5]) of local pair (i,j).
// PhasedListVector derivation (ld_engine.cpp:244-246) when nothing is masked;
// PhasedVectorized masking (~(mA|mB), ld_engine.h:139-143) otherwise.
__device__ inline void d_cells_phased(const TileView& t, uint32_t i, uint32_t j, uint64_t c[4]) {
	const uint64_t twoN = 2ull * t.n_samples;
	if (t.kind == PK_PHASED) {
		const uint64_t AA = t.C[(size_t)i * t.ldc + j];
		const uint64_t acA = t.rowpop[t.a0 + i], acB = t.rowpop[t.b0 + j];
		c[3] = AA; c[1] = acA - AA; c[2] = acB - AA; c[0] = twoN - ((acA + acB) - AA);
	} else {
		const uint32_t* r0 = t.C + (size_t)(2 * i) * t.ldc + 2 * j;
		const uint32_t* r1 = r0 + t.ldc;
		const uint64_t AA = r0[0], AM = r0[1], MA = r1[0], MM = r1[1];
		const uint64_t acA = t.rowpop[2 * (t.a0 + i)], nmA = t.rowpop[2 * (t.a0 + i) + 1];
		const uint64_t acB = t.rowpop[2 * (t.b0 + j)], nmB = t.rowpop[2 * (t.b0 + j) + 1];
		const uint64_t valid = twoN - nmA - nmB + MM;
		c[3] = AA; c[1] = acA - AA - AM; c[2] = acB - AA - MA; c[0] = valid - c[1] - c[2] - c[3];
	}
}

// The same 2x2 table for the plain phased planes from a count that arrives by value (the candidate list of the fused
// count kernel, ld_count.hip.h): sA / sB are positions in the plane set.
__device__ inline void d_cells_phased_aa(const TileView& t, uint32_t sA, uint32_t sB, uint64_t AA, uint64_t c[4]) {
	const uint64_t twoN = 2ull * t.n_samples;
	const uint64_t acA = t.rowpop[sA], acB = t.rowpop[sB];
	c[3] = AA; c[1] = acA - AA; c[2] = acB - AA; c[0] = twoN - ((acA + acB) - AA);
}

// The 3x3 table of the plain unphased planes from products that arrive by value (the candidate list of the fused count
// kernel's unphased form): the arithmetic of d_cells_unphased's PK_UNPHASED branch.
__device__ inline void d_cells_unphased_v(const TileView& t, uint32_t sA, uint32_t sB, uint64_t HH, uint64_t HQ, uint64_t QH, uint64_t QQ, uint64_t c[9]) {
	const uint64_t nhA = t.rowpop[2 * sA], nqA = t.rowpop[2 * sA + 1], nhB = t.rowpop[2 * sB], nqB = t.rowpop[2 * sB + 1];
	const uint64_t nvalid = t.n_samples;
	c[4] = HH; c[5] = HQ; c[7] = QH; c[8] = QQ;
	c[3] = nhA - HH - HQ; c[6] = nqA - QH - QQ; c[1] = nhB - HH - QH; c[2] = nqB - HQ - QQ;
	c[0] = nvalid - (c[1] + c[2] + c[3] + c[4] + c[5] + c[6] + c[7] + c[8]);
}

// 3x3 table as the nine sums UnphasedMath reads.
__device__ inline void d_cells_unphased(const TileView& t, uint32_t i, uint32_t j, uint64_t c[9]) {
	uint64_t HH, HQ, QH, QQ, nhA, nqA, nhB, nqB, nvalid;
	if (t.kind == PK_UNPHASED) {
		const uint32_t* r0 = t.C + (size_t)(2 * i) * t.ldc + 2 * j;
		const uint32_t* r1 = r0 + t.ldc;
		HH = r0[0]; HQ = r0[1]; QH = r1[0]; QQ = r1[1];
		nhA = t.rowpop[2 * (t.a0 + i)]; nqA = t.rowpop[2 * (t.a0 + i) + 1];
		nhB = t.rowpop[2 * (t.b0 + j)]; nqB = t.rowpop[2 * (t.b0 + j) + 1];
		nvalid = t.n_samples;
	} else {
		const uint32_t* r0 = t.C + (size_t)(3 * i) * t.ldc + 3 * j;
		const uint32_t* r1 = r0 + t.ldc;
		const uint32_t* r2 = r1 + t.ldc;
		HH = r0[0]; HQ = r0[1]; QH = r1[0]; QQ = r1[1];
		const uint64_t HM = r0[2], QM = r1[2], MH = r2[0], MQ = r2[1], MM = r2[2];
		const uint32_t ra = 3 * (t.a0 + i), rb = 3 * (t.b0 + j);
		nhA = t.rowpop[ra] - HM; nqA = t.rowpop[ra + 1] - QM;
		nhB = t.rowpop[rb] - MH; nqB = t.rowpop[rb + 1] - MQ;
		nvalid = (uint64_t)t.n_samples - t.rowpop[ra + 2] - t.rowpop[rb + 2] + MM;
	}
	c[4] = HH;                 // (het, het)   cells 17+20+65+68
	c[5] = HQ;                 // (het, 1/1)   cells 21+69
	c[7] = QH;                 // (1/1, het)   cells 81+84
	c[8] = QQ;                 // (1/1, 1/1)   cell  85
	c[3] = nhA - HH - HQ;      // (het, 0/0)   cells 16+64
	c[6] = nqA - QH - QQ;      // (1/1, 0/0)   cell  80
	c[1] = nhB - HH - QH;      // (0/0, het)   cells 1+4
	c[2] = nqB - HQ - QQ;      // (0/0, 1/1)   cell  5
	c[0] = nvalid - (c[1] + c[2] + c[3] + c[4] + c[5] + c[6] + c[7] + c[8]);
}

// TWK_HIP_OPT_REF_COMPAT: turn the correct masked 2x2 table c = {REFREF, A alt/B ref, A ref/B alt, ALTALT} of
// file-order variants A, B into what the reference's PhasedVectorized returns (ld_engine.cpp:513-634 as
// compiled; oracle: orc_count_phased_k3_as_is).  Its SIMD body over the 128-bit lanes [0, 2N/128) is right;
// its scalar tail over the remaining one or two 64-bit words adds popcnt(A ref, B alt) to REFREF instead of
// the REFREF count, and feeds (A alt, B ref) / (A ref, B alt) to the opposite counters; REFREF is then
// reduced by (ceil(2N/64)*64 - 2N)/2.  The tail is at most 128 bits per row: recomputed here from the raw rows.
__device__ inline void d_k3_as_compiled(const uint32_t* __restrict__ raw, const uint32_t* __restrict__ rawmask, uint32_t Wp,
                                        uint32_t n_samples, uint32_t A, uint32_t B, uint64_t c[4]) {
	const uint64_t two_n = 2ull * n_samples;
	const uint32_t byte_width = (uint32_t)((two_n + 63) / 64);              // 64-bit words per row
	const uint32_t byte_aligned_end = (uint32_t)(two_n / 128) * 2;
	const uint64_t adjustment = ((uint64_t)byte_width * 64 - two_n) / 2;
	uint64_t t_ar = 0, t_ra = 0, t_rr = 0;                                   // tail: A alt/B ref, A ref/B alt, REFREF (real alleles)
	for (uint32_t k = byte_aligned_end; k < byte_width; ++k) {
		const size_t wa = (size_t)A * Wp + 2 * k, wb = (size_t)B * Wp + 2 * k;
		const uint64_t x = (uint64_t)raw[wa] | (uint64_t)raw[wa + 1] << 32, y = (uint64_t)raw[wb] | (uint64_t)raw[wb + 1] << 32;
		const uint64_t ma = (uint64_t)rawmask[wa] | (uint64_t)rawmask[wa + 1] << 32, mb = (uint64_t)rawmask[wb] | (uint64_t)rawmask[wb + 1] << 32;
		const uint64_t m = ~(ma | mb);
		const uint64_t bit0 = (uint64_t)k * 64;
		const uint64_t live = two_n - bit0 >= 64 ? ~0ull : ((1ull << (two_n - bit0)) - 1);      // bits that are alleles
		t_ar += (uint64_t)__popcll((x ^ y) & x & m);
		t_ra += (uint64_t)__popcll((x ^ y) & y & m);
		t_rr += (uint64_t)__popcll(~x & ~y & m & live);
	}
	const uint64_t c0 = c[0], c1 = c[1], c2 = c[2];
	c[0] = (c0 - t_rr) + t_ra - adjustment;          // uint64 like the reference: wraps when the adjustment exceeds it
	c[1] = (c1 - t_ar) + t_ra;
	c[2] = (c2 - t_ra) + t_ar;
}

// ---- the math / filter / compaction kernel ---------------------------------------------------
struct StatsParams {
	TileView tv;
	VariantMeta vm;
	const uint32_t* raw; const uint32_t* rawmask; uint32_t Wp;   // raw rows (file order), for TWK_HIP_OPT_REF_COMPAT
	const uint32_t* col_hi; uint32_t hi_a0, hi_b0;               // r2 screen: the row at set position a reaches the columns
	                                                             // below hi_b0 + col_hi[a - hi_a0] only (others were not contracted)
	uint32_t list_zone;       // pairs with both set positions below it belong to the carrier-list pass (ld_list.hip.h)
	uint32_t nA, nB;          // variants in the tile
	uint32_t n_variants;      // total (pairs beyond it do not exist)
	int diag;                 // keep only col > row (global indices)
	int phased_math;          // 1: PhasedMath on the 2x2 table, 0: UnphasedMath on the 3x3
	int auto_select;          // 0: all pairs; 1: only pairs with an_A == 0 && an_B == 0;
	                          // 2: only pairs with an_A != 0 || an_B != 0   
	int window; uint32_t l_window;   // window: TWK_HIP_OPT_* bits
	twk_hip_filters filt;
	twk_hip_record* out;
	unsigned long long capacity;
	unsigned long long* n_out; // device counter
};

// One pair of the super-tile: skips, window, cells -> math -> filters.  sA / sB: positions in the plane set; i / j the
// same relative to the tile (C is indexed by them); aa: the pair's count when it arrives by value (plain phased planes,
// from the fused count kernel's candidate list) instead of through C.  Returns true if the pair survives (rec filled).
enum { SRC_MATRIX = 0, SRC_PHASED_VALUE = 1, SRC_UNPHASED_VALUES = 2 };
template <int SRC>
__device__ __forceinline__ bool d_pair(const StatsParams& p, uint32_t sA, uint32_t sB, uint32_t i, uint32_t j, uint64_t aa, twk_hip_record* rec,
                                       uint32_t hq = 0, uint32_t qh = 0, uint32_t qq = 0) {
	constexpr bool BY_VALUE = SRC != SRC_MATRIX;
	bool todo = sA < p.n_variants && sB < p.n_variants && (!p.diag || sB > sA);
	if (BY_VALUE && (sA - p.tv.a0 >= p.nA || sB - p.tv.b0 >= p.nB)) todo = false;       // (a candidate is a pair of the tile's own variants: checked again here)
	if (todo && p.col_hi && sB >= p.hi_b0 + p.col_hi[sA - p.hi_a0]) todo = false;
	if (!BY_VALUE && sA < p.list_zone && sB < p.list_zone) todo = false;       // (the list pass hands its own pairs over by value)
	// A regrouped plane set (ids != null) can meet a pair in either order; the record always
	// has the variant that comes first in the file as A, like the reference's i < j loops.
	uint32_t A = sA, B = sB;
	if (todo && p.tv.ids) { A = p.tv.ids[sA]; B = p.tv.ids[sB]; }
	const bool flip = A > B;
	if (flip) { const uint32_t x = A; A = B; B = x; }
	// ld_engine.cpp:1918 / 2033: nothing to learn from two singletons
	if (todo && !(p.window & TWK_HIP_OPT_KEEP_LOW_AC) && p.vm.ac[A] + p.vm.ac[B] <= 2) todo = false;
	if (todo && p.auto_select) {
		const bool anymiss = p.vm.an[A] || p.vm.an[B];
		if ((p.auto_select == 1) == anymiss) todo = false;
	}
	if (todo && (p.window & TWK_HIP_OPT_WINDOW)) {   // exact window: same contig, |dpos| <= w
		const int64_t d = (int64_t)p.vm.pos[A] - (int64_t)p.vm.pos[B];
		if (p.vm.rid[A] != p.vm.rid[B] || (d < 0 ? -d : d) > (int64_t)p.l_window) todo = false;
	}
	if (!todo) return false;
	if (SRC == SRC_PHASED_VALUE || (SRC == SRC_MATRIX && p.phased_math)) {          // (the phased list kernel carries no cubic, the unphased one no PhasedMath front end)
		uint64_t c[4];
		if (SRC == SRC_PHASED_VALUE) d_cells_phased_aa(p.tv, sA, sB, aa, c);
		else d_cells_phased(p.tv, i, j, c);
		if (flip) { const uint64_t x = c[1]; c[1] = c[2]; c[2] = x; }
		// Which of the two off-diagonal counts is stored in cnt[1] depends on the CPU kernel the
		// reference picks for the pair: its run-length kernel - taken when either
		// variant has missing genotypes and ac_A + ac_B is below a sample-count dependent
		// threshold (ld_engine.cpp:1910, :1925-1926) - has (A ref, B alt) there, the vector
		// kernels (A alt, B ref).  The statistics are symmetric in the two; mirror the slot.
		if ((p.vm.missing[A] || p.vm.missing[B]) &&
		    p.vm.ac[A] + p.vm.ac[B] < (uint32_t)(0.0047 * p.tv.n_samples + 5.2913)) {
			const uint64_t x = c[1]; c[1] = c[2]; c[2] = x;
		}
		else if ((p.window & TWK_HIP_OPT_REF_COMPAT) && (p.vm.missing[A] || p.vm.missing[B]) && p.rawmask)
			d_k3_as_compiled(p.raw, p.rawmask, p.Wp, p.tv.n_samples, A, B, c);
		return d_phased_math(c[0], c[1], c[2], c[3], p.vm, A, B, p.filt, rec);
	}
	uint64_t c[9];
	if (SRC == SRC_UNPHASED_VALUES) d_cells_unphased_v(p.tv, sA, sB, aa, hq, qh, qq, c);
	else d_cells_unphased(p.tv, i, j, c);
	if (flip) {          // transpose the 3x3 table
		uint64_t x;
		x = c[1]; c[1] = c[3]; c[3] = x;
		x = c[2]; c[2] = c[6]; c[6] = x;
		x = c[5]; c[5] = c[7]; c[7] = x;
	}
	return d_unphased_math(c, p.vm, A, B, p.filt, rec);
}

// wave-level compaction of the survivors: one atomic per wave
__device__ __forceinline__ void d_append_survivor(const StatsParams& p, bool keep, const twk_hip_record& rec) {
	const unsigned long long ballot = __ballot(keep);
	if (ballot) {
		const int lane = threadIdx.x & 63;
		const int leader = __ffsll((long long)ballot) - 1;
		unsigned long long base = 0;
		if (lane == leader) base = atomicAdd(p.n_out, (unsigned long long)__popcll(ballot));
		base = __shfl(base, leader);
		if (keep) {
			const unsigned long long slot = base + __popcll(ballot & ((1ull << lane) - 1));
			if (slot < p.capacity) p.out[slot] = rec;
		}
	}
}

__global__ __launch_bounds__(256)
void k_ld_stats(const StatsParams p) {
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t i = blockIdx.y;
	bool keep = false;
	twk_hip_record rec;
	if (i < p.nA && j < p.nB) keep = d_pair<SRC_MATRIX>(p, p.tv.a0 + i, p.tv.b0 + j, i, j, 0, &rec);
	d_append_survivor(p, keep, rec);
}

// The math on the candidate list of the fused count kernel (k_count_screen_t): one candidate (set position A, set
// position B, AA) per lane, grid-stride; n_cand is the device counter the count kernel left (it may have run past
// `cap`: the host then redoes the tile through C).  Everything the plain kernel tests is tested again here - the count
// kernel's screen only decides what is worth looking at.
__global__ __launch_bounds__(256)
void k_ld_stats_list(const StatsParams* pp, const uint32_t* __restrict__ cand, const unsigned long long* __restrict__ n_cand,
                     unsigned long long cap) {
	unsigned long long n = *n_cand;
	if (n > cap) n = cap;
	const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
	const unsigned long long n_up = (n + 63) / 64 * 64;          // whole waves stay together for the ballot
#pragma unroll 1
	for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < n_up; k += stride) {
		// the parameter block is read from memory inside the loop (the fence keeps the loads from being hoisted: held
		// across the loop they cost ~170 scalar registers, spilled into vector registers, and half the occupancy)
		asm volatile("" ::: "memory");
		const StatsParams& p = *pp;
		bool keep = false;
		twk_hip_record rec;
		if (k < n) {
			const uint32_t sA = cand[3 * k], sB = cand[3 * k + 1];
			keep = d_pair<SRC_PHASED_VALUE>(p, sA, sB, 0, 0, cand[3 * k + 2], &rec);
		}
		d_append_survivor(p, keep, rec);
	}
}

// One candidate of the unphased form; out of line, so that the cubic's registers are the callee's and not held across the
// candidate loop (inlined into it the kernel needs 212 VGPRs: two waves per SIMD).
__device__ __noinline__ void d_list_item_unphased(const StatsParams* pp, const uint32_t* e, bool valid) {
	const StatsParams& p = *pp;
	bool keep = false;
	twk_hip_record rec;
	if (valid) keep = d_pair<SRC_UNPHASED_VALUES>(p, e[0], e[1], 0, 0, e[2], &rec, e[3], e[4], e[5]);
	d_append_survivor(p, keep, rec);
}
// The same over the unphased form's candidates: (set position A, set position B, HH, HQ, QH, QQ), the cubic and all.
__global__ __launch_bounds__(256)
void k_ld_stats_list_unphased(const StatsParams* pp, const uint32_t* __restrict__ cand, const unsigned long long* __restrict__ n_cand,
                              unsigned long long cap) {
	unsigned long long n = *n_cand;
	if (n > cap) n = cap;
	const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
	const unsigned long long n_up = (n + 63) / 64 * 64;
#pragma unroll 1
	for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < n_up; k += stride)
		d_list_item_unphased(pp, cand + 6 * k, k < n);
}

// Second stage of the math: Fisher's exact test on the compacted survivors (one thread per
// record, grid-stride).  Inside k_ld_stats the test's data-dependent loops (up to min(n1_, n_1)
// iterations) would run with one live lane per wave while 63 wait; here every lane has a record.
// Arguments as at ld_engine.cpp:1222-1226 (integer cells) / :1656-1658 (round() of the expected
// haplotype counts): n11 = cnt[0], n12 = cnt[2] (REFALT slot), n21 = cnt[1], n22 = cnt[3].
// Records with P > minP are dropped (:1228, :1661): marked idxA = 0xFFFFFFFF for the host.
#define TWK_DROPPED_RECORD 0xFFFFFFFFu
#define TWK_FISHER_RECURRENCE_BELOW 1e-270   // observed-table probabilities below this take the reference's own recurrence (k_ld_fisher)
__global__ __launch_bounds__(256)
void k_ld_fisher(twk_hip_record* __restrict__ recs, unsigned long long* __restrict__ n_out,
                 unsigned long long capacity, double minP, const LFact lfact, const uint32_t* __restrict__ deferred) {
	// deferred == null: every record; else (behind k_ld_fisher_group) the n_out[3] records that kernel listed
	unsigned long long n = deferred ? n_out[3] : n_out[0];
	if (!deferred && n > capacity) n = capacity;
	uint32_t dropped = 0;
	for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < n;
	     k += (unsigned long long)gridDim.x * blockDim.x) {
		twk_hip_record* r = recs + (deferred ? (unsigned long long)deferred[k] : k);
		int n11 = (int)round(r->cnt[0]);
		if (r->flags & TWK_N11_IN_PAD) { n11 = (int)r->_pad; r->flags &= ~TWK_N11_IN_PAD; r->_pad = 0; }
		const double both = d_fisher_two(lfact, n11, (int)round(r->cnt[2]), (int)round(r->cnt[1]), (int)round(r->cnt[3]));
		r->P = both;
		if (both > minP) { r->idxA = TWK_DROPPED_RECORD; ++dropped; }
	}
	// how many were dropped (n_out[1]): the host cuts them off behind the sort without looking at the records
	for (int o = 32; o > 0; o >>= 1) dropped += __shfl_xor(dropped, o);
	if ((threadIdx.x & 63) == 0 && dropped) atomicAdd(n_out + 1, (unsigned long long)dropped);
}

// ---- Fisher's test, sixteen lanes per record ------------------------------------------------------------
// k_ld_fisher above is the reference's walk as it stands: one record per lane, one term after the other through the
// ratio recurrence (two FP64 divisions per term, the log-binomials again every 11th term).  A wave then runs as long
// as its longest record, every term waits for the one before it, and the divisions alone are ~300 cycles per term:
// 69 ms for the 33 M survivors of the 2,504-sample run.  The terms of a tail do not depend on each other, though:
//     pmf(s) = exp( lbinom(row1, s) + lbinom(total - row1, col1 - s) - lbinom(total, col1) )
// is what the reference itself evaluates at every 11th step, and with the log-factorial table it is four loads and one
// exp.  So a group of 16 lanes takes one record and evaluates 16 consecutive terms of a tail at once - coalesced
// loads, no divisions, no dependency chain - and the reference's stopping rule becomes a ballot: the walk ends at
// the first term (from the outside) that is not below 0.99999999 q; the terms before it are summed, that term is
// added if it is below 1.00000001 q (fisher_math.cpp:249-258).  Each term is the value the reference's recurrence
// would be re-synchronised to at that point, so the two differ by the rounding the recurrence accumulates over at
// most ten ratios (~1e-15 relative): the same stop decisions, P equal to ~1e-14.  The verified starting points of
// d_fisher_two are kept, their (up to four) proposals per side evaluated by eight lanes at once.
constexpr int FISHER_GROUP = 16;
// What a walk needs to know about its record: the margins and the three table values that do not depend on the term.
struct FisherMargins {
	int row1, col1, total;             // n1_, n_1, n
	double lf_row1, lf_rest, lb_all;   // lf[row1], lf[total - row1], lbinom(total, col1)
};
// log pmf(s) for fixed margins, bit for bit the exponent of d_hypergeo(t, s, row1, col1, total) - the same subtractions and
// additions in the same order - but with the table entries loaded side by side: d_hypergeo reaches every entry through
// its own range check (a branch per load), which strings nine memory round trips together.  Every index lies inside the
// table: the group kernel only takes records whose margins do (the others go to k_ld_fisher), and min <= s <= max.
__device__ __forceinline__ double d_pmf_logterm(const LFact& t, const FisherMargins& m, int s) {
	const int i1 = s, i2 = m.row1 - s, i3 = m.col1 - s, i4 = m.total - m.row1 - m.col1 + s;
	const double a1 = t.lf[i1], a2 = t.lf[i2], a3 = t.lf[i3], a4 = t.lf[i4];
	const double lb1 = (i1 == 0 || i2 == 0) ? 0. : m.lf_row1 - a1 - a2;          // d_lbinom(row1, s)
	const double lb2 = (i3 == 0 || i4 == 0) ? 0. : m.lf_rest - a3 - a4;          // d_lbinom(total - row1, col1 - s)
	return lb1 + lb2 - m.lb_all;
}
// Two terms at once (one of each tail): eight loads in flight, then two exps.  A term that is not wanted comes back as 0
// and touches nothing.
__device__ __forceinline__ void d_pmf_term2(const LFact& t, const FisherMargins& m, int sL, bool wantL, int sR, bool wantR, double& pL, double& pR) {
	const int l1 = sL, l2 = m.row1 - sL, l3 = m.col1 - sL, l4 = m.total - m.row1 - m.col1 + sL;
	const int r1 = sR, r2 = m.row1 - sR, r3 = m.col1 - sR, r4 = m.total - m.row1 - m.col1 + sR;
	double a1 = 0, a2 = 0, a3 = 0, a4 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0;
	if (wantL) { a1 = t.lf[l1]; a2 = t.lf[l2]; a3 = t.lf[l3]; a4 = t.lf[l4]; }
	if (wantR) { b1 = t.lf[r1]; b2 = t.lf[r2]; b3 = t.lf[r3]; b4 = t.lf[r4]; }
	pL = 0.; pR = 0.;
	if (wantL) {
		const double lb1 = (l1 == 0 || l2 == 0) ? 0. : m.lf_row1 - a1 - a2;
		const double lb2 = (l3 == 0 || l4 == 0) ? 0. : m.lf_rest - a3 - a4;
		pL = exp(lb1 + lb2 - m.lb_all);
	}
	if (wantR) {
		const double lb1 = (r1 == 0 || r2 == 0) ? 0. : m.lf_row1 - b1 - b2;
		const double lb2 = (r3 == 0 || r4 == 0) ? 0. : m.lf_rest - b3 - b4;
		pR = exp(lb1 + lb2 - m.lb_all);
	}
}
__device__ __forceinline__ uint32_t d_group_ballot(bool pred, int g0) { return (uint32_t)(__ballot(pred) >> g0) & 0xFFFFu; }
__device__ __forceinline__ double d_group_sum(double v) {
#pragma unroll
	for (int o = FISHER_GROUP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, FISHER_GROUP);
	return v;
}

__global__ __launch_bounds__(256)
void k_ld_fisher_group(twk_hip_record* __restrict__ recs, unsigned long long* __restrict__ n_out,
                       unsigned long long capacity, double minP, const LFact lfact, uint32_t* __restrict__ deferred) {
	unsigned long long n_recs = n_out[0];
	if (n_recs > capacity) n_recs = capacity;
	const int lane = threadIdx.x & 63, l = lane & (FISHER_GROUP - 1), g0 = lane & ~(FISHER_GROUP - 1);
	const unsigned long long n_groups = (unsigned long long)gridDim.x * blockDim.x / FISHER_GROUP;
	uint32_t dropped = 0;
	for (unsigned long long rec_i = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) / FISHER_GROUP; rec_i < n_recs; rec_i += n_groups) {
		twk_hip_record* r = recs + rec_i;
		// arguments as in k_ld_fisher (every lane of the group reads the same words)
		int n11 = (int)round(r->cnt[0]);
		const uint32_t flags = r->flags;
		if (flags & TWK_N11_IN_PAD) n11 = (int)r->_pad;
		const int n12 = (int)round(r->cnt[2]), n21 = (int)round(r->cnt[1]), n22 = (int)round(r->cnt[3]);
		FisherMargins m;
		m.row1 = n11 + n12; m.col1 = n11 + n21; m.total = n11 + n12 + n21 + n22;
		const int n1_ = m.row1, n_1 = m.col1, n = m.total;
		int max = (n_1 < n1_) ? n_1 : n1_;
		int min = n1_ + n_1 - n;
		if (min < 0) min = 0;
		double two = 1.;
		if (min != max) {
			// the record's constants and q, the observed table's own probability: nine table entries side by side
			double q;
			{
				const int rest = n - n1_, i2 = n1_ - n11, i3 = n_1 - n11, i4 = n - n1_ - n_1 + n11, k5 = n - n_1;
				// every index is one of the table's cells or margins: none negative and the total inside the table means all are
				if ((n | n1_ | rest | n_1 | k5 | n11 | i2 | i3 | i4) < 0 || n >= lfact.n) {
					// a count beyond the log-factorial table (the wrapped counts of TWK_HIP_OPT_REF_COMPAT, sample counts past the
					// table's limit): left to k_ld_fisher, which runs behind this kernel over the records listed here
					if (l == 0) deferred[atomicAdd(n_out + 3, 1ull)] = (uint32_t)rec_i;
					continue;
				}
				const double c1 = lfact.lf[n1_], c2 = lfact.lf[rest], c3 = lfact.lf[n], c4 = lfact.lf[n_1], c5 = lfact.lf[k5];
				const double a1 = lfact.lf[n11], a2 = lfact.lf[i2], a3 = lfact.lf[i3], a4 = lfact.lf[i4];
				m.lf_row1 = c1; m.lf_rest = c2;
				m.lb_all = (n_1 == 0 || k5 == 0) ? 0. : c3 - c4 - c5;                   // d_lbinom(n, n_1)
				const double lb1 = (n11 == 0 || i2 == 0) ? 0. : c1 - a1 - a2;
				const double lb2 = (i3 == 0 || i4 == 0) ? 0. : c2 - a3 - a4;
				q = exp(lb1 + lb2 - m.lb_all);
			}
			// Where the walk would start on denormal terms (q e^-K below ~1e-308) the reference's recurrence carries a value
			// with a few dozen significant bits to the observed table, and whether that lands inside its 1e-8 stopping band -
			// i.e. whether the observed table's own probability is counted in P at all - is decided by that rounding
			// (fisher_math.cpp:249-258; e.g. the table (3741, 794, 8, 465): q = 1.103e-296, the sum of all terms <= q is
			// 1.107e-296, the reference returns 4.0e-299).  Evaluating every term exactly does not reproduce that; the
			// one-lane walk, which runs the same recurrence on the same values, does: such records are left to it.
			if (q > 0 && q < TWK_FISHER_RECURRENCE_BELOW) {
				if (l == 0) deferred[atomicAdd(n_out + 3, 1ull)] = (uint32_t)rec_i;
				continue;
			}
			const double thr = 0.99999999 * q, tie = 1.00000001 * q;
			int i0 = min, j0 = max;
			if (q > 0 && max - min > 64) {
				// verified starting points (see d_fisher_two): lanes 0-3 try the left proposals k = 0..3, lanes 4-7 the right ones
				const double lq = log(q), nn = (double)n;
				const double mean = (double)n1_ * (double)n_1 / nn;
				const double sd = sqrt(mean * ((nn - n1_) / nn) * ((nn - n_1) / (nn - 1.0)));
				const double dev = fabs((double)n11 - mean);
				const double K = d_fisher_skip_exponent(max - min);
				double D = sqrt(dev * dev + 2.0 * (K + 8.0) * sd * sd) + 4.0;
				for (int k = 0; k < (l & 3); ++k) D = D * 1.5 + 8.0;
				const bool right = (l & 4) != 0;
				const double sf = right ? ceil(mean + D) : floor(mean - D);
				const bool out = right ? sf >= (double)max : sf <= (double)min;       // the proposal left the support: stay at its end
				const int s = out ? (right ? max : min) : (int)sf;
				bool hit = out;
				if (!out && l < 8) hit = d_pmf_logterm(lfact, m, s) <= lq - K;
				const uint32_t hits = d_group_ballot(hit && l < 8, g0);
				const uint32_t hl = hits & 0xFu, hr = (hits >> 4) & 0xFu;
				if (hl) i0 = __shfl(s, g0 + (__ffs(hl) - 1));
				if (hr) j0 = __shfl(s, g0 + 4 + (__ffs(hr) - 1));
			}
			// Both tails at once, 16 terms of each per round: the left walk takes i0, i0 + 1, ... and the right walk j0, j0 - 1, ...,
			// each up to its first term that is not below thr (or the end of the support); a walk that is done no longer loads
			// anything.
			double left = 0., right = 0.;
			bool moreL = true, moreR = true;
			for (int baseL = i0, baseR = j0; moreL || moreR; baseL += FISHER_GROUP, baseR -= FISHER_GROUP) {
				const int sL = baseL + l, sR = baseR - l;
				// (the reference's right walk is bounded by 0, but it ends at the observed table at the latest, and n11 >= min: lanes
				// beyond min would index the table with a negative cell)
				const bool validL = moreL && sL <= max, validR = moreR && sR >= min;
				double pL, pR;
				d_pmf_term2(lfact, m, sL, validL, sR, validR, pL, pR);
				if (moreL) {
					const uint32_t stop = d_group_ballot(validL && !(pL < thr), g0);
					const int first = stop ? __ffs(stop) - 1 : FISHER_GROUP;
					left += d_group_sum((validL && l < first) ? pL : 0.);
					if (stop) { const double ps = __shfl(pL, g0 + first); if (ps < tie) left += ps; moreL = false; }
					else if (baseL + FISHER_GROUP > max) moreL = false;
				}
				if (moreR) {
					const uint32_t stop = d_group_ballot(validR && !(pR < thr), g0);
					const int first = stop ? __ffs(stop) - 1 : FISHER_GROUP;
					right += d_group_sum((validR && l < first) ? pR : 0.);
					if (stop) { const double ps = __shfl(pR, g0 + first); if (ps < tie) right += ps; moreR = false; }
					else if (baseR - FISHER_GROUP < min) moreR = false;
				}
			}
			two = left + right;
			if (two > 1.) two = 1.;
		}
		if (l == 0) {
			r->P = two;
			if (flags & TWK_N11_IN_PAD) { r->flags = flags & ~TWK_N11_IN_PAD; r->_pad = 0; }
			if (two > minP) { r->idxA = TWK_DROPPED_RECORD; ++dropped; }
		}
	}
	for (int o = 32; o > 0; o >>= 1) dropped += __shfl_xor(dropped, o);
	if ((threadIdx.x & 63) == 0 && dropped) atomicAdd(n_out + 1, (unsigned long long)dropped);
}

// Raw cells for parity tests: out[(i*nB + j)*ncell + k] (uint64).
__global__ void k_ld_cells(const TileView tv, uint32_t nA, uint32_t nB, uint32_t n_variants, int diag,
                           int phased, unsigned long long* __restrict__ out) {
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t i = blockIdx.y;
	if (i >= nA || j >= nB) return;
	const uint32_t A = tv.a0 + i, B = tv.b0 + j;
	const int ncell = phased ? 4 : 9;
	unsigned long long* o = out + ((size_t)i * nB + j) * ncell;
	const bool todo = A < n_variants && B < n_variants && (!diag || B > A);
	uint64_t c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	if (todo) { if (phased) d_cells_phased(tv, i, j, c); else d_cells_unphased(tv, i, j, c); }
	for (int k = 0; k < ncell; ++k) o[k] = c[k];
}

}  // namespace twk
