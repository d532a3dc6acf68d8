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
// (kept out of line: inlined, libm's lgamma - three copies per log-binomial - made hipcc leave d_lbinom and d_fisher_two
// as real calls with the table passed through scratch memory)
__device__ __noinline__ double d_lgamma1_beyond_table(int i) { return lgamma((double)(i + 1)); }
__device__ __forceinline__ double d_lgamma1(const LFact& t, int i) { return (unsigned)i < (unsigned)t.n ? t.lf[i] : d_lgamma1_beyond_table(i); }
__device__ __forceinline__ double d_lbinom(const LFact& t, int n, int k) {
	if (k == 0 || n == k) return 0;
	return d_lgamma1(t, n) - d_lgamma1(t, k) - d_lgamma1(t, n - k);
}
__device__ __forceinline__ double d_hypergeo(const LFact& t, int n11, int n1_, int n_1, int n) {
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
	__device__ __forceinline__ double reset(const LFact& t, int k0, int r1, int c1, int n) {      // fix the margins, evaluate cell k0 in full
		k = k0; row1 = r1; col1 = c1; total = n;
		return pmf = d_hypergeo(t, k, row1, col1, total);
	}
	__device__ __forceinline__ double move_to(const LFact& t, int to) {                            // pmf of cell `to` (same margins)
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
// support-dependent K = 17.5 + ln(max - min) - 2.5e-8 of P, a third fewer terms - was measured before the starts were
// moved onto the reference's re-synchronisation cells: 5 % of the kernel then, and in the underflow band, where the
// reference's result hangs on the rounding of its recurrence, 9 of 1.2 M records no longer bit for bit.  Not worth it.)
__device__ inline double d_fisher_skip_exponent(int) { return 40.0; }
// ---- where the two tail walks start ------------------------------------------------------------------------
// The reference walks both tails from the ends of the support (min, max) inwards until the terms reach q: up to
// min(n1_, n_1) steps per record, almost all of them over terms that are zero or tens of orders of magnitude below q.
// Each walk starts closer in instead, at a point that is *verified* (one log-pmf evaluation from the table) to lie
// below q by a factor e^-40: the pmf is monotone out there, so everything skipped sums to < (max - min) * 4e-18 * q,
// i.e. < 1e-10 of the result (P >= q) even at 2e7 haplotypes - far inside the 1e-6 bar; the walk itself, its
// re-synchronisation every 11th step and its stopping rule are unchanged.
// Finding the point (round 3; before: up to four proposals at 1, 1.5, 2.25, ... times the normal approximation's
// distance, which on skewed tables - rare x common variants - failed once or twice and then started ten times too far
// out: 46 terms a record on a 2,504-sample run, 208 for the longest record of a wave):
//   1. the normal approximation's e^-40 point, sqrt(dev^2 + 2 K sd^2) from the mean;
//   2. if the pmf there is still above q e^-40: one step outwards of (excess / |ln ratio there|) cells - the ratios
//      only get steeper further out (the pmf is log-concave), so that point is below the target;
//   3. one step back in of (margin / |ln ratio towards the inside|) cells - every ratio further in is flatter, so the
//      point stays below the target.
// The ratios are the recurrence's own ((row1-k)(col1-k) / ((k+1)(k+1+slack))), taken in float with the rounding
// pushed to the safe side; every point is verified in FP64 against the table and a point that fails is not used
// (the walk then starts where the reference does).  21 terms a record on the same run (the best possible start: 20.7),
// before d_fisher_starts moves the point out to a multiple of 11.
__device__ __forceinline__ double d_fisher_lpmf(const LFact& t, int s, int n1_, int n_1, int n, double lden) {
	return d_lbinom(t, n1_, s) + d_lbinom(t, n - n1_, n_1 - s) - lden;
}
// |ln(pmf(s + dir) / pmf(s))| (float): the steepness of the pmf between cell s and its neighbour on the dir side
__device__ __forceinline__ float d_fisher_steepness(int s, int dir, int n1_, int n_1, int slack) {
	const float up = dir > 0 ? ((float)(n1_ - s) * (float)(n_1 - s)) / ((float)(s + 1) * (float)(s + 1 + slack))
	                         : ((float)s * (float)(s + slack)) / ((float)(n1_ - s + 1) * (float)(n_1 - s + 1));
	return -__logf(up);
}
__device__ __forceinline__ int d_fisher_start(const LFact& t, int dir, int n11, int n1_, int n_1, int n, int min, int max,
                                              double mean, double reach, double target, double lden) {
	const int end = dir < 0 ? min : max;
	const double sf = dir < 0 ? floor(mean - reach) : ceil(mean + reach);
	if (dir < 0 ? sf <= (double)min : sf >= (double)max) return end;
	const int slack = n - n1_ - n_1;
	int s = (int)sf;
	double v = d_fisher_lpmf(t, s, n1_, n_1, n, lden);
	if (v > target) {                                     // 2. outwards
		const float steep = d_fisher_steepness(s, dir, n1_, n_1, slack) * 0.999f;
		if (!(steep > 0.f)) return end;
		const float step = ceilf((float)(v - target) / steep) + 1.f;
		if (!(step < 1e9f)) return end;
		const long long s1 = (long long)s + (long long)dir * (long long)step;
		if (dir < 0 ? s1 <= (long long)min : s1 >= (long long)max) return end;
		s = (int)s1;
		v = d_fisher_lpmf(t, s, n1_, n_1, n, lden);
		if (v > target) return end;
	}
	{	                                                   // 3. back in
		const float steep = d_fisher_steepness(s - dir, dir, n1_, n_1, slack) * 1.001f;
		const float step = floorf((float)(target - 0.5 - v) / steep);
		if (steep > 0.f && step >= 1.f && step < 1e9f) {
			const long long s1 = (long long)s - (long long)dir * (long long)step;
			if (dir < 0 ? (s1 > (long long)min && s1 < (long long)n11) : (s1 < (long long)max && s1 > (long long)n11)) {
				if (d_fisher_lpmf(t, (int)s1, n1_, n_1, n, lden) <= target) s = (int)s1;
			}
		}
	}
	return s;
}
// The margins, the support and the observed table's own probability q; the walk object is left on cell n11.
struct FisherSetup { int n1_, n_1, n, min, max; double q; };
__device__ __forceinline__ bool d_fisher_setup(const LFact& t, TableWalk& w, int n11, int n12, int n21, int n22, FisherSetup& f) {
	f.n1_ = n11 + n12; f.n_1 = n11 + n21; f.n = n11 + n12 + n21 + n22;
	f.max = (f.n_1 < f.n1_) ? f.n_1 : f.n1_;
	f.min = f.n1_ + f.n_1 - f.n;
	if (f.min < 0) f.min = 0;
	if (f.min == f.max) return false;        // one table only: P = 1
	f.q = w.reset(t, n11, f.n1_, f.n_1, f.n);
	return true;
}
__device__ __forceinline__ void d_fisher_starts(const LFact& t, int n11, const FisherSetup& f, int& i0, int& j0) {
	i0 = f.min; j0 = f.max;
	if (f.q > 0 && f.max - f.min > 64) {
		const double lq = log(f.q), nn = (double)f.n;
		const double mean = (double)f.n1_ * (double)f.n_1 / nn;
		const double sd = sqrt(mean * ((nn - f.n1_) / nn) * ((nn - f.n_1) / (nn - 1.0)));
		const double lden = d_lbinom(t, f.n, f.n_1);
		const double dev = fabs((double)n11 - mean);
		const double K = d_fisher_skip_exponent(f.max - f.min);
		const double reach = sqrt(dev * dev + 2.0 * K * sd * sd) + 1.0;
		i0 = d_fisher_start(t, -1, n11, f.n1_, f.n_1, f.n, f.min, f.max, mean, reach, lq - K, lden);
		j0 = d_fisher_start(t, +1, n11, f.n1_, f.n_1, f.n, f.min, f.max, mean, reach, lq - K, lden);
		// ... and from there out to the next cell whose index is a multiple of 11 (or the end of the support): the cells
		// where the reference's own walk evaluates the pmf in full.  The walk's first value is then the very number the
		// reference has at that cell, and so is every term after it - including the ones its stopping rule compares with
		// q, however short the walk (a start in between carries its own rounding until the next multiple of 11: on steep
		// tails, where a walk is a handful of terms, that decided now and then whether a term equal to q was counted).
		i0 -= i0 % 11;
		if (i0 < f.min) i0 = f.min;
		j0 += (11 - j0 % 11) % 11;
		if (j0 > f.max) j0 = f.max;
	}
}
// The two walks, term for term the reference's (fisher_math.cpp:249-258) - but the lanes of a wave are lined up on the
// recurrence's re-synchronisation first.  move_to evaluates the cell in full (nine table look-ups and an exp, several
// times the cost of a ratio step) whenever the cell index is a multiple of 11; lanes start at unrelated cells, so in a
// wave walking in step some lane is at a multiple of 11 at almost every step (1 - (10/11)^64) and the whole wave pays
// for the full evaluation every time.  The starting points of d_fisher_starts are multiples of 11 themselves; a walk
// that starts at the end of the support (cell i0, any residue) sits out its first (i0 mod 11) rounds: from then on its
// cell index is congruent to the round number, and all lanes re-synchronise in the same rounds, one in eleven.  A lane's
// own sequence of operations is unchanged; it waits at most ten cheap rounds.
__device__ __forceinline__ double d_fisher_walks(const LFact& t, TableWalk& w, const FisherSetup& f, int i0, int j0) {
	const double q = f.q;
	double left, right;
	double p = w.move_to(t, i0);
	{
		left = 0.;
		int i = i0 + 1;
		bool more = p < 0.99999999 * q && i <= f.max;
		const int wait = i0 % 11;
		for (int round = 1; __ballot(more) != 0; ++round)
			if (more && round > wait) { left += p; p = w.move_to(t, i); ++i; more = p < 0.99999999 * q && i <= f.max; }
	}
	if (p < 1.00000001 * q) left += p;
	p = w.move_to(t, j0);
	{
		right = 0.;
		int j = j0 - 1;
		bool more = p < 0.99999999 * q && j >= 0;
		const int wait = (11 - j0 % 11) % 11;          // walking down: cell index = -round (mod 11)
		for (int round = 1; __ballot(more) != 0; ++round)
			if (more && round > wait) { right += p; p = w.move_to(t, j); --j; more = p < 0.99999999 * q && j >= 0; }
	}
	if (p < 1.00000001 * q) right += p;
	double two = left + right;
	if (two > 1.) two = 1.;
	return two;
}
// Two-sided P only (left / right tails are not stored in the record).  All lanes of the wave that are active at the
// call take part in the walks' ballots.
__device__ __forceinline__ double d_fisher_two(const LFact& t, int n11, int n12, int n21, int n22) {
	TableWalk w;
	FisherSetup f;
	const bool walk = d_fisher_setup(t, w, n11, n12, n21, n22, f);
	int i0 = 0, j0 = 0;
	if (walk) d_fisher_starts(t, n11, f, i0, j0);
	double two = 1.;
	if (walk) two = d_fisher_walks(t, w, f, i0, j0);
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
	// Fisher's exact test (:1221-1231) runs in k_ld_fisher_t on the compacted survivors
	rec->idxA = A; rec->idxB = B; rec->_pad = 0;
	rec->cnt[0] = (double)c0; rec->cnt[1] = (double)c1; rec->cnt[2] = (double)c4; rec->cnt[3] = (double)c5;
	rec->D = D; rec->Dprime = Dprime; rec->R = sqrt(R2); rec->R2 = R2; rec->P = 0;
	rec->ChiSqFisher = T * R2; rec->ChiSqModel = 0;
	rec->flags = d_common_flags(vm, A, B, rec->cnt, R2) | 1u;
	// A REFREF count that wrapped (TWK_HIP_OPT_REF_COMPAT) no longer fits the double: the reference hands
	// Fisher's test the uint64 narrowed to int (ld_engine.cpp:1222, fisher_math.cpp:231), i.e. its low 32
	// bits; they ride in _pad to k_ld_fisher_t, which clears the marker bit again.
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
	// Fisher's exact test on the rounded counts (:1655-1664) runs in k_ld_fisher_t
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
	uint32_t list_zone;       // pairs with both set positions below it belong to the carrier-list merge pass (ld_list.hip.h),
	uint32_t probe_zone;      // and every other pair of a row below this (<= list_zone) to the probe pass
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
	// The survivors leave the device in (idxA, idxB) order (twk_hip.hip sort_records): the 64-bit sort key idxA << key_shift |
	// idxB and the record's position are written next to the record where it is appended (null: not wanted).
	unsigned long long* keys; uint32_t* vals; uint32_t key_shift;
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
	if (!BY_VALUE && ((sA < p.list_zone && sB < p.list_zone) || sA < p.probe_zone)) todo = false;       // (the list pass hands its own pairs over by value)
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

// Compaction of the survivors: one atomic per *block*.  (One per wave was the kernels' bound in a survivor-rich run: the
// counter is one address, an L2 channel takes ~12 ns per atomic on it, and 574 k waves with a survivor are the 7.2 ms
// k_ld_stats_list took for the 36 M candidates of the 2,504-sample window run - the same figure as the candidate counter
// of the fused count kernel, ld_count.hip.h.)  Every thread of the block calls this the same number of times; blocks are
// 256 threads.
constexpr int STATS_THREADS = 256;
__device__ __forceinline__ void d_append_survivor(const StatsParams& p, bool keep, const twk_hip_record& rec) {
	__shared__ uint32_t wave_keep[STATS_THREADS / 64];
	__shared__ unsigned long long block_base;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned long long ballot = __ballot(keep);
	if (lane == 0) wave_keep[wave] = (uint32_t)__popcll(ballot);
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t total = 0;
		for (int w = 0; w < STATS_THREADS / 64; ++w) total += wave_keep[w];
		block_base = total ? atomicAdd(p.n_out, (unsigned long long)total) : 0ull;
	}
	__syncthreads();
	if (keep) {
		unsigned long long slot = block_base;
		for (int w = 0; w < wave; ++w) slot += wave_keep[w];
		slot += (unsigned long long)__popcll(ballot & ((1ull << lane) - 1));
		if (slot < p.capacity) {
			p.out[slot] = rec;
			if (p.keys) { p.keys[slot] = (unsigned long long)rec.idxA << p.key_shift | rec.idxB; p.vals[slot] = (uint32_t)slot; }
		}
	}
	__syncthreads();                              // (the next call overwrites the counts)
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
// position B, AA) per lane, grid-stride; n_cand is the device counter of the slots the count kernel handed out (it may
// have run past `cap`: the host then redoes the tile through C; slots handed out and not used are marked).  Everything the plain kernel tests is tested again here - the count
// kernel's screen only decides what is worth looking at.
__global__ __launch_bounds__(256)
void k_ld_stats_list(const StatsParams* pp, const uint32_t* __restrict__ cand, const unsigned long long* __restrict__ n_cand,
                     unsigned long long cap) {
	unsigned long long n = *n_cand;
	if (n > cap) n = cap;
	const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
	const unsigned long long n_up = (n + STATS_THREADS - 1) / STATS_THREADS * STATS_THREADS;          // whole blocks stay together for the append
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
			if (sA != 0xFFFFFFFFu)             // (a slot a wave reserved and did not use: CAND_UNUSED, ld_count.hip.h)
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
	if (valid && e[0] != 0xFFFFFFFFu)      // (CAND_UNUSED: a slot a wave reserved and did not use, ld_count.hip.h)
		keep = d_pair<SRC_UNPHASED_VALUES>(p, e[0], e[1], 0, 0, e[2], &rec, e[3], e[4], e[5]);
	d_append_survivor(p, keep, rec);
}
// The same over the unphased form's candidates: (set position A, set position B, HH, HQ, QH, QQ), the cubic and all.
__global__ __launch_bounds__(256)
void k_ld_stats_list_unphased(const StatsParams* pp, const uint32_t* __restrict__ cand, const unsigned long long* __restrict__ n_cand,
                              unsigned long long cap) {
	const unsigned long long n = *n_cand;
	if (n > cap) return;                                    // (an overflowed list: the host redoes the launch another way and keeps nothing of this one)
	const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
	const unsigned long long n_up = (n + STATS_THREADS - 1) / STATS_THREADS * STATS_THREADS;
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
// The four integer arguments of the test as the reference passes them (see above).
__device__ __forceinline__ void d_fisher_args(const twk_hip_record* r, int& n11, int& n12, int& n21, int& n22) {
	n11 = (r->flags & TWK_N11_IN_PAD) ? (int)r->_pad : (int)round(r->cnt[0]);
	n12 = (int)round(r->cnt[2]); n21 = (int)round(r->cnt[1]); n22 = (int)round(r->cnt[3]);
}

// ---- the test in three passes: prepare, order, walk -------------------------------------------------------------
// A wave of the walk kernel runs as long as its longest walk, and neighbouring survivors (one row variant, adjacent
// columns) have walks of very different lengths.  So the work is split:
//   k_fisher_prepare   per record: q and the two starting points (uniform work: a handful of table look-ups), parked in
//                      the record's P field - which the test is about to fill - together with the bin of the walk's
//                      length, now known almost exactly: start to crossing on either side (the crossing on the observed
//                      side is n11 itself, the other one its mirror image about the mean); bins of one term up to 64
//                      terms, of 1/8 octave beyond (exponent and three mantissa bits of the estimate as a float); the
//                      bins' histogram
//   k_fisher_scatter   counting sort: index[bin start + rank] = record position, any order inside a bin
//   k_ld_fisher_t      lane k walks the k-th record in bin order
// Only the order of evaluation changes with the bins.  Measured on the 10 M survivors of a 2,504-sample run
// (profiles/r03_fisher_order.txt).
constexpr int FISHER_BINS = 256;
constexpr unsigned long long FISHER_AT_END = (1ull << 26) - 1;         // packed start: "the end of the support"
__device__ __forceinline__ unsigned long long d_fisher_pack(int n11, int i0, int j0, const FisherSetup& f, bool walk) {
	if (!walk) return ~0ull;                                               // P = 1 without a walk
	const unsigned long long a = (i0 == f.min || (unsigned)(n11 - i0) >= FISHER_AT_END) ? FISHER_AT_END : (unsigned long long)(n11 - i0);
	const unsigned long long b = (j0 == f.max || (unsigned)(j0 - n11) >= FISHER_AT_END) ? FISHER_AT_END : (unsigned long long)(j0 - n11);
	// length of the two walks
	float est = 0.f;
	if (f.q > 0) {
		const float mean = (float)f.n1_ * ((float)f.n_1 / (float)f.n), dev = fabsf((float)n11 - mean);
		const float s0 = (float)(a == FISHER_AT_END ? f.min : i0), s1 = (float)(b == FISHER_AT_END ? f.max : j0);
		est = fmaxf(mean - dev - s0, 0.f) + fmaxf(s1 - (mean + dev), 0.f);
	}
	uint32_t bin;
	if (est < 64.f) bin = (uint32_t)est;
	else { const int key = 64 + (int)(__float_as_uint(est) >> 20) - ((127 + 6) << 3); bin = (uint32_t)(key >= FISHER_BINS ? FISHER_BINS - 1 : key); }
	return a << 38 | b << 12 | bin;
}
__device__ __forceinline__ void d_fisher_unpack(unsigned long long v, int n11, const FisherSetup& f, int& i0, int& j0) {
	const unsigned long long a = v >> 38, b = (v >> 12) & FISHER_AT_END;
	i0 = a == FISHER_AT_END ? f.min : n11 - (int)a;
	j0 = b == FISHER_AT_END ? f.max : n11 + (int)b;
}
constexpr int FISHER_LDS_TABLE_MAX = 8064;       // 63 KiB of doubles (k_fisher_prepare keeps 1 KiB of bins beside it): N <= 4,024 samples
// LDS_TABLE: the log-factorial table is copied into LDS first (it must fit: FISHER_LDS_TABLE_MAX entries) - through the
// vector memory path every look-up of a wave is up to 64 separate cache-line requests.
// bins[0..256): records per bin (zeroed before the launch; only the first min(n, index_limit) records are counted).
template <bool LDS_TABLE>
__global__ __launch_bounds__(LDS_TABLE ? 1024 : 256)
void k_fisher_prepare(twk_hip_record* __restrict__ recs, const unsigned long long* __restrict__ n_out, unsigned long long capacity,
                      const LFact lfact_in, unsigned long long index_limit, uint32_t* __restrict__ bins) {
	extern __shared__ double lf_lds[];
	__shared__ uint32_t h[FISHER_BINS];
	LFact lfact = lfact_in;
	if (LDS_TABLE) {
		for (int i = threadIdx.x; i < lfact_in.n; i += blockDim.x) lf_lds[i] = lfact_in.lf[i];
		lfact.lf = lf_lds;
	}
	if (threadIdx.x < FISHER_BINS) h[threadIdx.x] = 0;
	__syncthreads();
	unsigned long long n = n_out[0];
	if (n > capacity) n = capacity;
	for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (unsigned long long)gridDim.x * blockDim.x) {
		twk_hip_record* r = recs + k;
		int n11, n12, n21, n22;
		d_fisher_args(r, n11, n12, n21, n22);
		TableWalk w;
		FisherSetup f;
		const bool walk = d_fisher_setup(lfact, w, n11, n12, n21, n22, f);
		int i0 = 0, j0 = 0;
		if (walk) d_fisher_starts(lfact, n11, f, i0, j0);
		const unsigned long long v = d_fisher_pack(n11, i0, j0, f, walk);
		reinterpret_cast<unsigned long long*>(&r->P)[0] = v;
		if (k < index_limit) atomicAdd(&h[v & 0xFFu], 1u);
	}
	__syncthreads();
	if (threadIdx.x < FISHER_BINS && h[threadIdx.x]) atomicAdd(bins + threadIdx.x, h[threadIdx.x]);
}
// index[bin start + rank] = record position.  bins[256..512): the bins' fill cursors (zeroed before the launch).  A block
// takes 1024 records at a time: ranks inside the block through LDS, one global atomic per bin the batch touches.
__global__ __launch_bounds__(256)
void k_fisher_scatter(const twk_hip_record* __restrict__ recs, const unsigned long long* __restrict__ n_out, unsigned long long capacity,
                      unsigned long long index_limit, uint32_t* __restrict__ bins, uint32_t* __restrict__ index) {
	__shared__ uint32_t start[FISHER_BINS], cnt[FISHER_BINS], base[FISHER_BINS], wsum[4];
	unsigned long long n = n_out[0];
	if (n > capacity) n = capacity;
	if (n > index_limit) n = index_limit;
	{	// exclusive scan of the 256 bin sizes
		const uint32_t v = bins[threadIdx.x];
		uint32_t incl = v;
		for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if ((int)(threadIdx.x & 63) >= o) incl += t; }
		if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
		__syncthreads();
		uint32_t before = 0;
		for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wsum[w];
		start[threadIdx.x] = before + incl - v;
	}
	for (unsigned long long k0 = (unsigned long long)blockIdx.x * 1024; k0 < n; k0 += (unsigned long long)gridDim.x * 1024) {
		cnt[threadIdx.x] = 0;
		__syncthreads();
		uint32_t bin[4], rank[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const unsigned long long k = k0 + (unsigned)j * 256u + threadIdx.x;
			bin[j] = 0; rank[j] = 0;
			if (k < n) { bin[j] = (uint32_t)(reinterpret_cast<const unsigned long long*>(&recs[k].P)[0] & 0xFFu); rank[j] = atomicAdd(&cnt[bin[j]], 1u); }
		}
		__syncthreads();
		if (cnt[threadIdx.x]) base[threadIdx.x] = start[threadIdx.x] + atomicAdd(bins + FISHER_BINS + threadIdx.x, cnt[threadIdx.x]);
		__syncthreads();
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const unsigned long long k = k0 + (unsigned)j * 256u + threadIdx.x;
			if (k < n) index[base[bin[j]] + rank[j]] = (uint32_t)k;
		}
		__syncthreads();
	}
}
// The walks over every record, in the order of `index` for the first min(n, index_limit) of them (k_fisher_scatter: a
// permutation of those positions).  prepared: the starting points are in the record (k_fisher_prepare); else they are
// found here.
template <bool LDS_TABLE>
__global__ __launch_bounds__(LDS_TABLE ? 1024 : 256)
void k_ld_fisher_t(twk_hip_record* __restrict__ recs, unsigned long long* __restrict__ n_out,
                   unsigned long long capacity, double minP, const LFact lfact_in, const uint32_t* __restrict__ index,
                   unsigned long long index_limit, int prepared, unsigned long long* __restrict__ keys) {
	extern __shared__ double lf_lds[];
	LFact lfact = lfact_in;
	if (LDS_TABLE) {
		for (int i = threadIdx.x; i < lfact_in.n; i += blockDim.x) lf_lds[i] = lfact_in.lf[i];
		__syncthreads();
		lfact.lf = lf_lds;
	}
	unsigned long long n = n_out[0];
	if (n > capacity) n = capacity;
	const unsigned long long n_indexed = index ? (n < index_limit ? n : index_limit) : 0;
	uint32_t dropped = 0;
	for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < n;
	     k += (unsigned long long)gridDim.x * blockDim.x) {
		twk_hip_record* r = recs + (k < n_indexed ? (unsigned long long)index[k] : k);
		int n11, n12, n21, n22;
		d_fisher_args(r, n11, n12, n21, n22);
		if (r->flags & TWK_N11_IN_PAD) { r->flags &= ~TWK_N11_IN_PAD; r->_pad = 0; }
		TableWalk w;
		FisherSetup f;
		const bool walk = d_fisher_setup(lfact, w, n11, n12, n21, n22, f);
		int i0 = 0, j0 = 0;
		if (walk) {
			if (prepared) d_fisher_unpack(reinterpret_cast<const unsigned long long*>(&r->P)[0], n11, f, i0, j0);
			else d_fisher_starts(lfact, n11, f, i0, j0);
		}
		double both = 1.;
		if (walk) both = d_fisher_walks(lfact, w, f, i0, j0);
		r->P = both;
		if (both > minP) {
			r->idxA = TWK_DROPPED_RECORD; ++dropped;
			if (keys) keys[r - recs] = ~0ull;          // sorts behind every kept record
		}
	}
	// how many were dropped (n_out[1]): the host cuts them off behind the sort without looking at the records
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
