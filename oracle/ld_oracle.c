/*
 * TEST INFRASTRUCTURE ONLY -- see ld_oracle.h.
 *
 * Scalar C restatement of the reference LD hot path.  Floating point follows
 * the reference expression by expression (same association, same libm calls)
 * and is compiled with -ffp-contract=off like the reference's -msse4.2 build
 * (no FMA), so that results are bit-identical to oracle/_ref on this machine.
 */
#include "ld_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <float.h>

/* lib/ld/ld_engine.h:33-37 */
#define ORC_LOW_AC_THRESHOLD        5
#define ORC_INVALID_HWE_THRESHOLD   1e-4
#define ORC_LONG_RANGE_THRESHOLD    500e3
#define ORC_MINIMUM_ALLOWED_ALLELES 5
#define ORC_ALLOWED_ROUNDING_ERROR  0.00001

void orc_default_settings(orc_settings* s) { /* lib/core.cpp:297-306 */
	s->minR2 = 0.1; s->maxR2 = 100; s->minDprime = 0; s->maxDprime = 100; s->minP = 1;
	s->force_phased = 0; s->forced_unphased = 0; s->keep_low_ac = 0;
}

uint32_t orc_words64(uint32_t n_samples) { /* ld_engine.cpp:58, core.cpp:353 */
	return (uint32_t)((2ull * n_samples + 63) / 64);
}

/* ---- T1: twk_igt_vec::Build, lib/core.cpp:365-391 ---------------------- */
int orc_build_bitvector(const uint32_t* run_len, const uint8_t* run_a, const uint8_t* run_b,
                        uint32_t n_runs, uint32_t n_samples, uint64_t* data, uint64_t* mask) {
	const uint32_t n = orc_words64(n_samples);
	memset(data, 0, n * sizeof(uint64_t));
	if (mask) memset(mask, 0, n * sizeof(uint64_t));
	uint64_t cumpos = 0;
	for (uint32_t i = 0; i < n_runs; ++i) {
		const uint32_t len = run_len[i];
		const uint8_t refA = run_a[i], refB = run_b[i];
		if (cumpos + 2ull * len > 2ull * n_samples) return -1;
		if (refA == 0 && refB == 0) { cumpos += 2ull * len; continue; } /* core.cpp:371-374 */
		for (uint64_t j = 0; j < 2ull * len; j += 2) {                  /* core.cpp:376-381 */
			const uint64_t p0 = cumpos + j, p1 = cumpos + j + 1;
			if (refA == 1) data[p0 / 64] |= 1ull << (p0 % 64);
			if (refB == 1) data[p1 / 64] |= 1ull << (p1 % 64);
			if ((refA == 2 || refB == 2) && mask) {
				mask[p0 / 64] |= 1ull << (p0 % 64);
				mask[p1 / 64] |= 1ull << (p1 % 64);
			}
		}
		cumpos += 2ull * len;
	}
	return cumpos == 2ull * n_samples ? 0 : -1; /* core.cpp:391 */
}

static inline int bit(const uint64_t* v, uint64_t p) { return (int)((v[p / 64] >> (p % 64)) & 1); }

void orc_genotype(const uint64_t* data, const uint64_t* mask, uint32_t s, int* a, int* b) {
	if (mask && (bit(mask, 2ull * s) || bit(mask, 2ull * s + 1))) { *a = 2; *b = 2; return; }
	*a = bit(data, 2ull * s); *b = bit(data, 2ull * s + 1);
}

/* ---- K1/K2/K3: vector kernels' 2x2 table -------------------------------
 * PhasedVectorized masks with ~(maskA | maskB) (ld_engine.h:139-143,
 * ld_engine.cpp:555-581); PhasedListVector / NoMissing derive the other
 * cells from ac (ld_engine.cpp:244-246, 682-685) which is the same table when
 * nothing is missing.  Slot order of `out`: c[0], c[1], c[4], c[5] with
 * c[1] = TWK_LD_ALTREF = (A alt, B ref) as assigned at ld_engine.cpp:606-607. */
void orc_count_phased(const uint64_t* a, const uint64_t* ma, const uint64_t* b, const uint64_t* mb,
                      uint32_t n_samples, uint64_t out[4]) {
	out[0] = out[1] = out[2] = out[3] = 0;
	for (uint64_t p = 0; p < 2ull * n_samples; ++p) {
		if ((ma && bit(ma, p)) || (mb && bit(mb, p))) continue;
		const int x = bit(a, p), y = bit(b, p);
		if (x && y) ++out[3];
		else if (x && !y) ++out[1];
		else if (!x && y) ++out[2];
		else ++out[0];
	}
}

/* ---- K3 as compiled: PhasedVectorized, ld_engine.cpp:513-634 -----------------
 * The SIMD body (128-bit lanes [0, vector_cycles), :555-581) counts the four masked cells correctly;
 * the lanes it skips at the front / tail are all-zero in data and mask of both variants
 * (core.cpp:394-435) and are credited to REFREF (:609), which is what they hold.  The scalar tail
 * over the 64-bit words [byte_aligned_end, byte_width) (:596-603) adds popcnt(b_refalt) to REFREF
 * instead of popcnt(b_refref), and its b_altref / b_refalt are (A alt, B ref) / (A ref, B alt) -
 * the other way round from the body's PHASED_ALTREF / PHASED_REFALT (ld_engine.h:135-138) - while
 * both feed the same counters.  Finally REFREF is reduced by phased_unbalanced_adjustment
 * = (byte_width*64 - 2N)/2 (:61, :609).  All arithmetic is uint64 like the reference's (it wraps if
 * the adjustment exceeds the count).  Only reached when either variant has missing genotypes
 * (:515-517) and ac_A + ac_B is at or above the run-length threshold (:1925-1928). */
void orc_count_phased_k3_as_is(const uint64_t* a, const uint64_t* ma, const uint64_t* b, const uint64_t* mb,
                               uint32_t n_samples, uint64_t out[4]) {
	const uint32_t byte_width = orc_words64(n_samples);                     /* :58 */
	const uint32_t vector_cycles = 2 * n_samples / 128;                      /* :59 */
	const uint32_t byte_aligned_end = vector_cycles * 2;                     /* :60 */
	const uint64_t adjustment = ((uint64_t)byte_width * 64 - 2ull * n_samples) / 2;   /* :61 */
	uint64_t refref = 0, simd_altref = 0, simd_refalt = 0, altalt = 0;
	for (uint32_t k = 0; k < byte_width; ++k) {
		const uint64_t m = ~((ma ? ma[k] : 0) | (mb ? mb[k] : 0));
		const uint64_t x = a[k], y = b[k];
		if (k < byte_aligned_end) {                                         /* body, ld_engine.h:131-142 */
			refref      += (uint64_t)__builtin_popcountll(~x & ~y & m);
			simd_altref += (uint64_t)__builtin_popcountll((x ^ y) & y & m);
			simd_refalt += (uint64_t)__builtin_popcountll((x ^ y) & x & m);
			altalt      += (uint64_t)__builtin_popcountll(x & y & m);
		} else {                                                            /* scalar tail, :596-603 */
			const uint64_t b_altref = (x ^ y) & x & m, b_refalt = (x ^ y) & y & m;
			refref      += (uint64_t)__builtin_popcountll(b_refalt);
			simd_refalt += (uint64_t)__builtin_popcountll(b_refalt);
			simd_altref += (uint64_t)__builtin_popcountll(b_altref);
			altalt      += (uint64_t)__builtin_popcountll(x & y & m);
		}
	}
	out[0] = refref - adjustment;      /* alleleCounts[TWK_LD_REFREF], :609 */
	out[1] = simd_refalt;              /* alleleCounts[TWK_LD_ALTREF] = counters[TWK_LD_SIMD_REFALT], :607 */
	out[2] = simd_altref;              /* alleleCounts[TWK_LD_REFALT] = counters[TWK_LD_SIMD_ALTREF], :606 */
	out[3] = altalt;
}

/* ---- K4: PhasedRunlength, ld_engine.cpp:1011-1091 ----------------------
 * index = (alleleA << 2) | alleleB per haplotype, alleles 0/1/2; only indices
 * 0,1,4,5 are read by PhasedMath.  A bitvector cannot tell 1|. from 1|1 with a
 * mask, so missing is per sample here too; what differs from the vector kernels
 * is the orientation: index 1 = (A ref, B alt). */
void orc_count_phased_rle(const uint64_t* a, const uint64_t* ma, const uint64_t* b, const uint64_t* mb,
                          uint32_t n_samples, uint64_t out[4]) {
	uint64_t v[4];
	orc_count_phased(a, ma, b, mb, n_samples, v);
	out[0] = v[0]; out[1] = v[2]; out[2] = v[1]; out[3] = v[3];
}

/* ---- K5/K6/K7: 3x3 table ------------------------------------------------
 * UnphasedRunlength indexes alleleCounts[(A1<<6)|(A2<<4)|(B1<<2)|B2]
 * (ld_engine.cpp:1105,1137); UnphasedMath only reads the nine sums below
 * (ld_engine.cpp:1314-1375).  The vector kernels fill the same sums
 * (ld_engine.cpp:835-844, 976-986). */
void orc_count_unphased(const uint64_t* a, const uint64_t* ma, const uint64_t* b, const uint64_t* mb,
                        uint32_t n_samples, uint64_t out[9]) {
	for (int i = 0; i < 9; ++i) out[i] = 0;
	for (uint32_t s = 0; s < n_samples; ++s) {
		int a1, a2, b1, b2;
		orc_genotype(a, ma, s, &a1, &a2);
		orc_genotype(b, mb, s, &b1, &b2);
		if (a1 == 2 || b1 == 2) continue;
		const int gA = a1 + a2, gB = b1 + b2; /* 0 hom-ref, 1 het, 2 hom-alt */
		++out[gA * 3 + gB];
	}
}

/* ---- M3: Fisher, lib/fisher_math.cpp:183-267 --------------------------- */
static double lbinom(int n, int k) { /* :183-187 */
	if (k == 0 || n == k) return 0;
	return lgamma(n + 1) - lgamma(k + 1) - lgamma(n - k + 1);
}
static double hypergeo(int n11, int n1_, int n_1, int n) { /* :195-198 */
	return exp(lbinom(n1_, n11) + lbinom(n - n1_, n_1 - n11) - lbinom(n, n_1));
}
typedef struct { int n11, n1_, n_1, n; double p; } hgacc_t;
static double hypergeo_acc(int n11, int n1_, int n_1, int n, hgacc_t* aux) { /* :206-229 */
	if (n1_ || n_1 || n) {
		aux->n11 = n11; aux->n1_ = n1_; aux->n_1 = n_1; aux->n = n;
	} else {
		if (n11 % 11 && n11 + aux->n - aux->n1_ - aux->n_1) {
			if (n11 == aux->n11 + 1) {
				aux->p *= (double)(aux->n1_ - aux->n11) / n11
				        * (aux->n_1 - aux->n11) / (n11 + aux->n - aux->n1_ - aux->n_1);
				aux->n11 = n11;
				return aux->p;
			}
			if (n11 == aux->n11 - 1) {
				aux->p *= (double)aux->n11 / (aux->n1_ - n11)
				        * (aux->n11 + aux->n - aux->n1_ - aux->n_1) / (aux->n_1 - n11);
				aux->n11 = n11;
				return aux->p;
			}
		}
		aux->n11 = n11;
	}
	aux->p = hypergeo(aux->n11, aux->n1_, aux->n_1, aux->n);
	return aux->p;
}
double orc_fisher_exact(int n11, int n12, int n21, int n22, double* _left, double* _right, double* two) { /* :231-267 */
	int i, j, max, min;
	double p, q, left, right;
	hgacc_t aux;
	int n1_, n_1, n;
	n1_ = n11 + n12; n_1 = n11 + n21; n = n11 + n12 + n21 + n22;
	max = (n_1 < n1_) ? n_1 : n1_;
	min = n1_ + n_1 - n;
	if (min < 0) min = 0;
	*two = *_left = *_right = 1.;
	if (min == max) return 1.;
	q = hypergeo_acc(n11, n1_, n_1, n, &aux);
	p = hypergeo_acc(min, 0, 0, 0, &aux);
	for (left = 0., i = min + 1; p < 0.99999999 * q && i <= max; ++i)
		left += p, p = hypergeo_acc(i, 0, 0, 0, &aux);
	--i;
	if (p < 1.00000001 * q) left += p;
	else --i;
	p = hypergeo_acc(max, 0, 0, 0, &aux);
	for (right = 0., j = max - 1; p < 0.99999999 * q && j >= 0; --j)
		right += p, p = hypergeo_acc(j, 0, 0, 0, &aux);
	++j;
	if (p < 1.00000001 * q) right += p;
	else ++j;
	*two = left + right;
	if (*two > 1.) *two = 1.;
	if (abs(i - n11) < abs(j - n11)) right = 1. - left + q;
	else left = 1.0 - right + q;
	*_left = left; *_right = right;
	return q;
}

/* ---- flags common to both maths: ld_engine.cpp:1244-1255 / 1674-1684 ---- */
static uint32_t common_flags(const orc_variant* A, const orc_variant* B, const double cnt[4], double R2) {
	uint32_t c = 0;
	if (A->ac < ORC_LOW_AC_THRESHOLD) c |= 1u << 10;
	if (B->ac < ORC_LOW_AC_THRESHOLD) c |= 1u << 11;
	if (cnt[0] < 1 || cnt[1] < 1 || cnt[2] < 1 || cnt[3] < 1) c |= 1u << 3;
	if (R2 > 0.99) c |= 1u << 4;
	if (A->an) c |= 1u << 8;
	if (B->an) c |= 1u << 9;
	const int32_t diff = (int32_t)A->pos - (int32_t)B->pos;
	if (abs(diff) > ORC_LONG_RANGE_THRESHOLD && A->rid == B->rid) c |= 1u << 2;
	if (A->rid == B->rid) c |= 1u << 1;
	if (A->hwe < ORC_INVALID_HWE_THRESHOLD) c |= 1u << 12;
	if (B->hwe < ORC_INVALID_HWE_THRESHOLD) c |= 1u << 13;
	return c;
}

/* ---- M1: PhasedMath, ld_engine.cpp:1162-1310 ---------------------------
 * c = {alleleCounts[0], [1], [4], [5]}. */
int orc_phased_math(const uint64_t c[4], const orc_variant* A, const orc_variant* B,
                    const orc_settings* st, orc_record* rec) {
	const uint64_t c0 = c[0], c1 = c[1], c4 = c[2], c5 = c[3];
	const uint64_t total = c0 + c4 + c1 + c5;                              /* :1164-1165 */
	if (total < ORC_MINIMUM_ALLOWED_ALLELES) return 0;                     /* :1168 */
	if (c0 < c5) { if (c4 + c1 + c0 < 5) return 0; }                       /* :1174-1186 */
	else         { if (c5 + c4 + c1 < 5) return 0; }

	const double pA = (double)c0 / total, qA = (double)c1 / total;        /* :1189-1192 */
	const double pB = (double)c4 / total, qB = (double)c5 / total;
	if (pA * qB - qA * pB == 0) return 0;                                  /* :1194 */
	const double g0 = ((double)c0 + c4) / total;                           /* :1197-1200 */
	const double g1 = ((double)c1 + c5) / total;
	const double h0 = ((double)c0 + c1) / total;
	const double h1 = ((double)c4 + c5) / total;

	const double D  = pA * qB - qA * pB;                                   /* :1202 */
	const double R2 = D * D / (g0 * g1 * h0 * h1);                         /* :1203 */
	if (R2 < st->minR2 || R2 > st->maxR2) return 0;                        /* :1204 */
	double dmax = 0;                                                       /* :1209-1211 */
	if (D >= 0) dmax = g0 * h1 < h0 * g1 ? g0 * h1 : h0 * g1;
	else        dmax = g0 * g1 < h0 * h1 ? -g0 * g1 : -h0 * h1;
	const double Dprime = D / dmax;                                        /* :1213 */
	if (Dprime < st->minDprime || Dprime > st->maxDprime) return 0;        /* :1215 */

	double left, right, both;                                              /* :1221-1226: int narrowing */
	orc_fisher_exact((int)c0, (int)c4, (int)c1, (int)c5, &left, &right, &both);
	if (both > st->minP) return 0;                                         /* :1228 */

	memset(rec, 0, sizeof(*rec));
	rec->P = both; rec->R = sqrt(R2); rec->R2 = R2; rec->D = D; rec->Dprime = Dprime;
	rec->Apos = A->pos; rec->Bpos = B->pos; rec->ridA = A->rid; rec->ridB = B->rid;
	rec->cnt[0] = (double)c0;  /* cur_rcd[SIMD_REFREF=0] = c[0]   :1239 */
	rec->cnt[2] = (double)c4;  /* cur_rcd[SIMD_REFALT=2] = c[4]   :1240 */
	rec->cnt[1] = (double)c1;  /* cur_rcd[SIMD_ALTREF=1] = c[1]   :1241 */
	rec->cnt[3] = (double)c5;  /*                                 :1242 */
	rec->controller = common_flags(A, B, rec->cnt, R2) | 1u;              /* :1244-1255, bit0 = phased math */
	rec->ChiSqModel = 0;                                                   /* :1258 */
	rec->ChiSqFisher = total * R2;                                         /* :1259 */
	return 1;
}

/* ---- ChiSquaredUnphasedTable, ld_engine.cpp:1562-1588 ------------------
 * o = observed {0, 1+4, 5, 16+64, hets, 21+69, 80, 81+84, 85}. */
static double chisq_unphased(const uint64_t o[9], double total, double target, double p, double q) {
	const double f12 = p - target;
	const double f21 = q - target;
	const double f22 = 1 - (target + f12 + f21);
	const double e1111 = total * pow(target, 2);
	const double e1112 = 2 * total * target * f12;
	const double e1122 = total * pow(f12, 2);
	const double e1211 = 2 * total * target * f21;
	const double e1212 = 2 * total * f12 * f21 + 2 * total * target * f22;
	const double e1222 = 2 * total * f12 * f22;
	const double e2211 = total * pow(f21, 2);
	const double e2212 = 2 * total * f21 * f22;
	const double e2222 = total * pow(f22, 2);
	const double c1111 = e1111 > 0 ? pow((double)o[0] - e1111, 2) / e1111 : 0,
	             c1112 = e1112 > 0 ? pow((double)o[1] - e1112, 2) / e1112 : 0,
	             c1122 = e1122 > 0 ? pow((double)o[2] - e1122, 2) / e1122 : 0,
	             c1211 = e1211 > 0 ? pow((double)o[3] - e1211, 2) / e1211 : 0,
	             c1212 = e1212 > 0 ? pow((double)o[4] - e1212, 2) / e1212 : 0,
	             c1222 = e1222 > 0 ? pow((double)o[5] - e1222, 2) / e1222 : 0,
	             c2211 = e2211 > 0 ? pow((double)o[6] - e2211, 2) / e2211 : 0,
	             c2212 = e2212 > 0 ? pow((double)o[7] - e2212, 2) / e2212 : 0,
	             c2222 = e2222 > 0 ? pow((double)o[8] - e2222, 2) / e2222 : 0;
	return c1111 + c1112 + c1122 + c1211 + c1212 + c1222 + c2211 + c2212 + c2222;
}

/* ---- ChooseF11Calculate, ld_engine.cpp:1590-1740 ------------------------ */
static int choose_f11(double total, double target, double p, double q, uint32_t pre_flags,
                      const orc_variant* A, const orc_variant* B, const orc_settings* st, orc_record* rec) {
	const double f11 = target;
	const double f12 = p - f11;
	const double f21 = q - f11;
	const double f22 = 1 - (f11 + f12 + f21);
	const double D = (f11 * f22) - (f12 * f21);
	const double R2 = (D * D) / (p * (1 - p) * q * (1 - q));               /* :1609 */
	if (R2 < st->minR2 || R2 > st->maxR2) return 0;                        /* :1617 */

	double cnt[4];
	cnt[0] = f11 * 2 * total;                                              /* :1624-1627 */
	cnt[2] = f12 * 2 * total;   /* SIMD_REFALT = 2 */
	cnt[1] = f21 * 2 * total;   /* SIMD_ALTREF = 1 */
	cnt[3] = f22 * 2 * total;
	if (cnt[0] < cnt[3]) { if (cnt[2] + cnt[1] + cnt[0] < 5) return 0; }  /* :1631-1643 */
	else                 { if (cnt[3] + cnt[2] + cnt[1] < 5) return 0; }

	double dmax = 0;                                                       /* :1645-1647 */
	if (D >= 0) dmax = p * (1.0 - q) < q * (1.0 - p) ? p * (1.0 - q) : q * (1.0 - p);
	else        dmax = p * q < (1 - p) * (1 - q) ? -p * q : -(1 - p) * (1 - q);
	const double Dprime = D / dmax;
	if (Dprime < st->minDprime || Dprime > st->maxDprime) return 0;        /* :1650 */

	double left, right, both;                                              /* :1655-1658 */
	orc_fisher_exact((int)round(cnt[0]), (int)round(cnt[2]), (int)round(cnt[1]), (int)round(cnt[3]),
	                 &left, &right, &both);
	if (both > st->minP) return 0;                                         /* :1661 */

	memset(rec, 0, sizeof(*rec));
	rec->D = D; rec->R2 = R2; rec->R = sqrt(R2); rec->Dprime = Dprime; rec->P = both;
	rec->cnt[0] = cnt[0]; rec->cnt[1] = cnt[1]; rec->cnt[2] = cnt[2]; rec->cnt[3] = cnt[3];
	rec->Apos = A->pos; rec->Bpos = B->pos; rec->ridA = A->rid; rec->ridB = B->rid;
	rec->ChiSqModel = 0;                                                   /* :1670 (sic) */
	rec->ChiSqFisher = (cnt[0] + cnt[2] + cnt[1] + cnt[3]) * R2;           /* :1671 */
	rec->controller = pre_flags | common_flags(A, B, cnt, R2);             /* :1674-1684, bit0 clear */
	return 1;
}

/* ---- M2: UnphasedMath, ld_engine.cpp:1312-1560 ------------------------- */
int orc_unphased_math(const uint64_t o[9], const orc_variant* A, const orc_variant* B,
                      const orc_settings* st, orc_record* rec) {
	const uint64_t a0 = o[0], a14 = o[1], a5 = o[2], a1664 = o[3], hets = o[4],
	               a2169 = o[5], a80 = o[6], a8184 = o[7], a85 = o[8];
	const uint64_t total_u = a0 + a14 + a5 + a1664 + hets + a2169 + a80 + a8184 + a85; /* :1314-1318 */
	if (total_u < ORC_MINIMUM_ALLOWED_ALLELES) return 0;                   /* :1321 */

	if (hets == 0) {                                                       /* :1334-1348 */
		uint64_t c[4];
		c[0] = 2 * a0  + a14   + a1664;          /* alleleCounts[0]  :1335 */
		c[2] = 2 * a5  + a14   + a2169;          /* alleleCounts[4]  :1336 */
		c[1] = 2 * a80 + a1664 + a8184;          /* alleleCounts[1]  :1337 */
		c[3] = 2 * a85 + a8184 + a2169;          /* alleleCounts[5]  :1338 */
		return orc_phased_math(c, A, B, st, rec);
	}

	const double total = (double)total_u;
	const double P = ((a0 + a14 + a5) * 2.0 + (a1664 + hets + a2169)) / (2.0 * total);     /* :1363 */
	const double Q = ((a0 + a1664 + a80) * 2.0 + (a14 + hets + a8184)) / (2.0 * total);     /* :1364 */
	const double n11 = (2.0 * a0 + a14 + a1664);                                             /* :1365 */
	const double minhap = n11 / (2.0 * total);                                               /* :1369 */
	const double maxhap = (n11 + hets) / (2.0 * total);                                      /* :1370 */
	const double dee = -n11 * P * Q;                                                         /* :1372 */
	const double c = -n11 * (1.0 - 2.0 * P - 2.0 * Q) - hets * (1.0 - P - Q) + (2.0 * total * P * Q); /* :1373 */
	const double b = 2.0 * total * (1.0 - 2.0 * P - 2.0 * Q) - 2.0 * n11 - hets;             /* :1374 */
	const double a = 4.0 * total;                                                            /* :1375 */

	const double xN  = -b / (3.0 * a);                                     /* :1388-1392 */
	const double d2  = (pow(b, 2) - 3.0 * a * c) / (9 * pow(a, 2));
	const double yN  = a * pow(xN, 3) + b * pow(xN, 2) + c * xN + dee;
	const double yN2 = pow(yN, 2);
	const double h2  = 4 * pow(a, 2) * pow(d2, 3);
	const double diff = yN2 - h2;                                          /* :1429 */
	const double lo = minhap - ORC_ALLOWED_ROUNDING_ERROR, hi = maxhap + ORC_ALLOWED_ROUNDING_ERROR;

	if (diff < 0) {                                                        /* :1438-1496 */
		const double h = pow(h2, 0.5);
		const double theta = ((acos(-yN / h)) / 3.0);
		const double delta = pow(d2, 0.5);
		const double alpha = xN + 2.0 * delta * cos(theta);
		const double beta  = xN + 2.0 * delta * cos(2.0 * M_PI / 3.0 + theta);
		const double gamma = xN + 2.0 * delta * cos(4.0 * M_PI / 3.0 + theta);
		int possible = 0;
		double best = DBL_MAX, chosen = alpha;
		if (alpha >= lo && alpha <= hi) { ++possible; best = chisq_unphased(o, total, alpha, P, Q); }
		if (beta >= lo && beta <= hi) {
			++possible;
			if (chisq_unphased(o, total, beta, P, Q) < best) { chosen = beta; best = chisq_unphased(o, total, beta, P, Q); }
		}
		if (gamma >= lo && gamma <= hi) {
			++possible;
			if (chisq_unphased(o, total, gamma, P, Q) < best) { chosen = gamma; best = chisq_unphased(o, total, gamma, P, Q); }
		}
		if (possible == 0) return 0;
		return choose_f11(total, chosen, P, Q, possible > 1 ? (1u << 5) : 0, A, B, st, rec);
	} else if (diff > 0) {                                                 /* :1498-1519 */
		double number1, number2;
		if ((1.0 / (2.0 * a) * (-yN + pow((yN2 - h2), 0.5))) < 0)
			number1 = -pow(-(1.0 / (2.0 * a) * (-yN + pow((yN2 - h2), 0.5))), 1.0 / 3.0);
		else number1 = pow((1.0 / (2.0 * a) * (-yN + pow((yN2 - h2), 0.5))), 1.0 / 3.0);
		if ((1.0 / (2.0 * a) * (-yN - pow((yN2 - h2), 0.5))) < 0)
			number2 = -pow(-(1.0 / (2.0 * a) * (-yN - pow((yN2 - h2), 0.5))), 1.0 / 3.0);
		else number2 = pow((1.0 / (2.0 * a) * (-yN - pow((yN2 - h2), 0.5))), 1.0 / 3.0);
		const double alpha = xN + number1 + number2;
		if (!(alpha >= lo && alpha <= hi)) return 0;
		return choose_f11(total, alpha, P, Q, 0, A, B, st, rec);
	} else {                                                               /* :1521-1558 */
		const double delta = pow((yN / 2.0 * a), (1.0 / 3.0));
		const double alpha = xN + delta;
		const double gamma = xN - 2.0 * delta;
		if (isnan(alpha) || isnan(gamma)) return 0;
		int possible = 0;
		double best = DBL_MAX, chosen = alpha;
		if (alpha >= lo && alpha <= hi) { ++possible; best = chisq_unphased(o, total, alpha, P, Q); }
		if (gamma >= lo && gamma <= hi) {
			++possible;
			if (chisq_unphased(o, total, gamma, P, Q) < best) { chosen = gamma; best = chisq_unphased(o, total, gamma, P, Q); }
		}
		if (possible == 0) return 0;
		return choose_f11(total, chosen, P, Q, 0, A, B, st, rec);
	}
}

/* ---- S1: pair treatment, ld_engine.cpp:1898-2188 (forced) / 2740-2838 (default) */
int orc_pair(const uint64_t* a, const uint64_t* ma, const orc_variant* A,
             const uint64_t* b, const uint64_t* mb, const orc_variant* B,
             uint32_t n_samples, const orc_settings* st, int vector_only, orc_record* rec) {
	if (!st->keep_low_ac && A->ac + B->ac <= 2) return 0;                  /* :1918, :2033; not in :2267 */
	const uint64_t* mA = A->gt_missing ? ma : NULL;
	const uint64_t* mB = B->gt_missing ? mb : NULL;
	int phased;
	if (st->force_phased) phased = 1;
	else if (st->forced_unphased) phased = 0;
	else phased = !(A->an || B->an);                                       /* :2775,2803 */

	if (phased) {
		uint64_t c[4];
		const uint32_t thresh_miss = (uint32_t)(0.0047 * n_samples + 5.2913); /* :1910 */
		if (!vector_only && (A->gt_missing || B->gt_missing) && (A->ac + B->ac < thresh_miss))
			orc_count_phased_rle(a, mA, b, mB, n_samples, c);              /* :1925-1926 */
		else if (st->ref_compat && (A->gt_missing || B->gt_missing))
			orc_count_phased_k3_as_is(a, mA, b, mB, n_samples, c);         /* :1928, as compiled */
		else
			orc_count_phased(a, mA, b, mB, n_samples, c);                  /* :1922-1923, :1928 */
		return orc_phased_math(c, A, B, st, rec);
	}
	uint64_t o[9];
	orc_count_unphased(a, mA, b, mB, n_samples, o);
	return orc_unphased_math(o, A, B, st, rec);
}

/* rows [row0, row1) of the pair triangle, every row against the variants behind it; records in pair order.
   No state outside the arguments: tests/ call disjoint row ranges from several threads (oracle.py). */
uint64_t orc_all_pairs_rows(const uint64_t* data, const uint64_t* mask, const orc_variant* vars,
                            uint32_t n_variants, uint32_t n_samples, const orc_settings* st,
                            int vector_only, uint32_t row0, uint32_t row1, orc_record* recs) {
	const uint32_t w = orc_words64(n_samples);
	uint64_t n = 0;
	for (uint32_t i = row0; i < row1 && i < n_variants; ++i)
		for (uint32_t j = i + 1; j < n_variants; ++j)
			if (orc_pair(data + (size_t)i * w, mask ? mask + (size_t)i * w : NULL, &vars[i],
			             data + (size_t)j * w, mask ? mask + (size_t)j * w : NULL, &vars[j],
			             n_samples, st, vector_only, &recs[n]))
				++n;
	return n;
}

uint64_t orc_all_pairs(const uint64_t* data, const uint64_t* mask, const orc_variant* vars,
                       uint32_t n_variants, uint32_t n_samples, const orc_settings* st,
                       int vector_only, orc_record* recs) {
	return orc_all_pairs_rows(data, mask, vars, n_variants, n_samples, st, vector_only, 0, n_variants, recs);
}

/* ---- O1: lib/core.cpp:470-490 ------------------------------------------ */
void orc_pack_record(const orc_record* r, uint8_t out[106]) {
	uint8_t* p = out;
	const uint16_t ctrl = (uint16_t)r->controller;
	const uint32_t packA = r->Apos << 2, packB = r->Bpos << 2; /* Aphased/Amiss never set */
	memcpy(p, &ctrl, 2); p += 2;
	memcpy(p, &r->ridA, 4); p += 4;
	memcpy(p, &r->ridB, 4); p += 4;
	memcpy(p, &packA, 4); p += 4;
	memcpy(p, &packB, 4); p += 4;
	memcpy(p, r->cnt, 32); p += 32;
	const double tail[7] = { r->D, r->Dprime, r->R, r->R2, r->P, r->ChiSqFisher, r->ChiSqModel };
	memcpy(p, tail, 56);
}
