// Rare variants as sorted carrier lists: the device counterpart of the reference's twk_igt_list (include/core.h:368-692,
// built at :517-672) and of PhasedListVector (lib/ld/ld_engine.cpp:185-267), for very large sample counts.
//
// The dense contraction decides a pair in time proportional to the row length W (62,528 words at N = 1 M: 2.55 ns), a
// merge of two sorted carrier lists in time proportional to the carriers (about 2.5 ps per merge step,
// profiles/r03_t2_list_vs_dense.txt): at N = 1 M the lists win below ~400 carriers per variant (5x at 100, 100x at 10),
// at N = 2,504 they never do (a dense pair costs 7 ps).  So this path exists for long rows only: the variants of the
// allele-count-sorted plane set (the r2 screen's order, twk_hip.hip) whose minor allele has at most L carriers - they
// lead that set, as the "list zone" - get a list of the haplotypes that carry their *minor* allele, and the pairs inside
// the zone (and inside the r2 band) are intersected instead of contracted:
//     X = |minor_A & minor_B|   ->   ALTALT = X,  ac_B - X,  ac_A - X  or  2N - (mac_A + mac_B - X)
// depending on which of the two has REF as its minor allele.  ALTALT is the one count the plain phased planes yield per
// pair, so from here on the pair goes the way of the fused count kernel's candidates (ld_count.hip.h): the same division-
// free r2 screen, the candidate list, k_ld_stats_list - records bit for bit those of the dense path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ld_count.hip.h"      // wave_scan_inclusive

namespace twk {

constexpr uint32_t LIST_END = 0xFFFFFFFFu;      // sentinel behind a variant's last carrier

// rows: the allele-count-sorted phased planes (one row of W words per variant, bit h = haplotype h carries ALT);
// variant v < n_list gets lists[v * stride + k], k < mac[v]: the haplotypes carrying its minor allele, ascending,
// then LIST_END.  flip[v] != 0: the minor allele is REF (the row has more than N ones).  One wave per variant.
__global__ __launch_bounds__(256)
void k_build_lists(const uint32_t* __restrict__ rows, uint32_t W, uint32_t W_live, uint64_t two_n, const uint32_t* __restrict__ rowpop,
                   uint32_t n_list, uint32_t stride, uint32_t* __restrict__ lists, uint32_t* __restrict__ mac, uint32_t* __restrict__ flip) {
	const uint32_t v = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (v >= n_list) return;
	const int lane = threadIdx.x & 63;
	const uint32_t ac = rowpop[v];
	const bool inv = (uint64_t)ac * 2 > two_n;
	const uint32_t m = inv ? (uint32_t)(two_n - ac) : ac;
	const uint32_t* row = rows + (size_t)v * W;
	uint32_t* out = lists + (size_t)v * stride;
	uint32_t base = 0;                                     // carriers written so far (wave-uniform)
	for (uint32_t k0 = 0; k0 < W_live; k0 += 64) {
		const uint32_t k = k0 + lane;
		uint32_t x = k < W_live ? row[k] : 0u;
		if (inv && k < W_live) {
			x = ~x;
			const uint64_t bit0 = (uint64_t)k * 32;          // the padding bits of the last word are not haplotypes
			if (two_n - bit0 < 32) x &= (1u << (two_n - bit0)) - 1u;
		}
		const unsigned long long any = __ballot(x != 0);
		if (!any) continue;
		const uint32_t cnt = __popc(x);
		uint32_t incl = cnt;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if (lane >= o) incl += y; }
		uint32_t at = base + incl - cnt;
		while (x) {                                         // this lane's carriers, ascending
			const int b = __ffs(x) - 1;
			x &= x - 1;
			if (at < stride - 1) out[at] = k * 32 + b;
			++at;
		}
		base += __shfl(incl, 63);
	}
	if (lane == 0) { out[m < stride - 1 ? m : stride - 1] = LIST_END; mac[v] = m; flip[v] = inv ? 1u : 0u; }
}

struct ListWork {
	const uint32_t* lists; uint32_t stride;        // [n_list][stride], LIST_END-terminated
	const uint32_t* mac; const uint32_t* flip;     // carriers of the minor allele; minor allele is REF
	const uint32_t* rowpop;                        // ALT alleles per set position
	uint32_t n_list;                               // the list zone: set positions [0, n_list)
	uint32_t row0, n_rows;                         // rows of this launch
	const uint32_t* col_hi; uint32_t hi_a0, hi_b0; // r2 band: row a reaches the columns below hi_b0 + col_hi[a - hi_a0]
	double two_n, cut;                             // 2N; minR2 * (1 - 1e-6)
	uint32_t* cand; unsigned long long cap; unsigned long long* n_cand;      // as in ScreenWork (ld_count.hip.h)
};

// One pair per lane: row i = row0 + blockIdx.y against the columns i + 1 + (64 consecutive per wave); neighbours in the
// allele-count order have lists of similar length, so the lanes of a wave run about equally long.
__global__ __launch_bounds__(256)
void k_list_screen(const ListWork w) {
	const uint32_t i = w.row0 + blockIdx.y;
	const uint32_t j = i + 1 + blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t limit = w.col_hi ? w.hi_b0 + w.col_hi[i - w.hi_a0] : w.n_list;
	if (limit > w.n_list) limit = w.n_list;
	bool keep = false;
	uint32_t aa = 0;
	if (j < limit) {
		const uint32_t* a = w.lists + (size_t)i * w.stride;
		const uint32_t* b = w.lists + (size_t)j * w.stride;
		const uint32_t na = w.mac[i], nb = w.mac[j];
		uint32_t ia = 0, ib = 0, x = 0, va = a[0], vb = b[0];
		while (ia < na && ib < nb) {           // merge step: count a match, advance the smaller head (both on a match)
			x += va == vb;
			const bool fa = va <= vb, fb = vb <= va;
			ia += fa; ib += fb;
			if (fa) va = a[ia];
			if (fb) vb = b[ib];
		}
		const uint32_t acA = w.rowpop[i], acB = w.rowpop[j];
		const bool fA = w.flip[i] != 0, fB = w.flip[j] != 0;
		// ALT carriers of both: the lists hold the minor allele's carriers
		if (!fA && !fB) aa = x;
		else if (fA && !fB) aa = nb - x;               // ALT_A = everyone but minor_A
		else if (!fA && fB) aa = na - x;
		else aa = (uint32_t)((uint64_t)w.two_n - ((uint64_t)na + nb - x));
		const double da = (double)acA, db = (double)acB;
		const double dn = w.two_n * (double)aa - da * db;
		keep = dn != 0.0 && dn * dn >= w.cut * (da * (w.two_n - da)) * (db * (w.two_n - db));
	}
	const unsigned long long ballot = __ballot(keep);
	if (ballot) {
		const int lane = threadIdx.x & 63;
		const int leader = __ffsll((long long)ballot) - 1;
		unsigned long long base = 0;
		if (lane == leader) base = atomicAdd(w.n_cand, (unsigned long long)__popcll(ballot));
		base = __shfl(base, leader);
		if (keep) {
			const unsigned long long slot = base + __popcll(ballot & ((1ull << lane) - 1));
			if (slot < w.cap) { uint32_t* e = w.cand + slot * 3; e[0] = i; e[1] = j; e[2] = aa; }
		}
	}
}

// ---- the same for UnphasedMath: lists of samples with their genotype ------------------------------------------
// rows: the allele-count-sorted unphased planes (per variant a row H = het and a row Q = hom-alt, one bit per sample).
// A rare variant's list holds the samples that are not homozygous for its major allele, ascending, as
// (sample << 1) | g with g = 0 for a het and g = 1 for the rare homozygote (hom-alt; hom-ref when flip[v], i.e. when ALT is
// the major allele), then LIST_END.  len[v] <= the minor allele count.  One wave per variant.
__global__ __launch_bounds__(256)
void k_build_lists_unphased(const uint32_t* __restrict__ rows, uint32_t W, uint32_t W_live, uint32_t n_samples, const uint32_t* __restrict__ rowpop,
                            uint32_t n_list, uint32_t stride, uint32_t* __restrict__ lists, uint32_t* __restrict__ len, uint32_t* __restrict__ flip) {
	const uint32_t v = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (v >= n_list) return;
	const int lane = threadIdx.x & 63;
	const uint32_t nh = rowpop[2 * v], nq = rowpop[2 * v + 1];
	const bool inv = (uint64_t)nh + 2ull * nq > (uint64_t)n_samples;          // more ALT than REF alleles: the rare homozygote is 0/0
	const uint32_t* H = rows + (size_t)(2 * v) * W;
	const uint32_t* Q = H + W;
	uint32_t* out = lists + (size_t)v * stride;
	uint32_t base = 0;
	for (uint32_t k0 = 0; k0 < W_live; k0 += 64) {
		const uint32_t k = k0 + lane;
		uint32_t h = 0, r = 0;
		if (k < W_live) {
			h = H[k];
			const uint32_t q = Q[k];
			if (inv) {
				r = ~(h | q);
				const uint32_t s0 = k * 32;                                     // samples beyond N do not exist
				if (n_samples - s0 < 32) r &= (1u << (n_samples - s0)) - 1u;
			} else r = q;
		}
		uint32_t x = h | r;
		const unsigned long long any = __ballot(x != 0);
		if (!any) continue;
		const uint32_t cnt = __popc(x);
		uint32_t incl = cnt;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if (lane >= o) incl += y; }
		uint32_t at = base + incl - cnt;
		while (x) {
			const int b = __ffs(x) - 1;
			x &= x - 1;
			if (at < stride - 1) out[at] = ((k * 32 + b) << 1) | ((r >> b) & 1u);
			++at;
		}
		base += __shfl(incl, 63);
	}
	const uint32_t total = inv ? nh + (n_samples - nh - nq) : nh + nq;
	if (lane == 0) { out[total < stride - 1 ? total : stride - 1] = LIST_END; len[v] = total; flip[v] = inv ? 1u : 0u; }
}

// One pair per lane, as k_list_screen: the merge runs over sample ids and sorts every common sample into one of four
// counters by the two genotype bits (packed 16 bits each: a list has at most a few hundred entries); the four products
// the unphased planes would have yielded follow from them and the variants' own counts -
//     HH = x[het][het];  a rare homozygote that is hom-alt counts as Q directly, one that is hom-ref turns its row / column
//     into "everyone else": HQ = h_A - x[het][het] - x[het][rare_B], QQ = N - |list_A u list_B| for two such variants, ...
// - then the interval screen of the fused unphased epilogue (ld_count.hip.h), and the six-word candidate
// (set position A, set position B, HH, HQ, QH, QQ) for k_ld_stats_list_unphased.
__global__ __launch_bounds__(256)
void k_list_screen_unphased(const ListWork w, uint32_t n_samples) {
	const uint32_t i = w.row0 + blockIdx.y;
	const uint32_t j = i + 1 + blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t limit = w.col_hi ? w.hi_b0 + w.col_hi[i - w.hi_a0] : w.n_list;
	if (limit > w.n_list) limit = w.n_list;
	bool keep = false;
	uint32_t HH = 0, HQ = 0, QH = 0, QQ = 0;
	if (j < limit) {
		const uint32_t* a = w.lists + (size_t)i * w.stride;
		const uint32_t* b = w.lists + (size_t)j * w.stride;
		const uint32_t na = w.mac[i], nb = w.mac[j];
		uint32_t ia = 0, ib = 0, va = a[0], vb = b[0];
		uint32_t xa0 = 0, xa1 = 0;                     // four 16-bit counters [gA][gB], two to a word (no 64-bit shifts: see k_probe_screen_unphased_t)
		while (ia < na && ib < nb) {
			const uint32_t sa = va >> 1, sb = vb >> 1;
			const uint32_t inc = (sa == sb) ? (1u << (16u * (vb & 1u))) : 0u;
			xa0 += (va & 1u) ? 0u : inc;
			xa1 += (va & 1u) ? inc : 0u;
			const bool fa = sa <= sb, fb = sb <= sa;
			ia += fa; ib += fb;
			if (fa) va = a[ia];
			if (fb) vb = b[ib];
		}
		const uint32_t x00 = xa0 & 0xFFFFu, x01 = xa0 >> 16, x10 = xa1 & 0xFFFFu, x11 = xa1 >> 16;
		const uint32_t hA = w.rowpop[2 * i], qA = w.rowpop[2 * i + 1], hB = w.rowpop[2 * j], qB = w.rowpop[2 * j + 1];
		const bool fA = w.flip[i] != 0, fB = w.flip[j] != 0;
		const uint32_t rA = fA ? n_samples - hA - qA : qA, rB = fB ? n_samples - hB - qB : qB;      // rare homozygotes (listed with g = 1)
		HH = x00;
		HQ = !fB ? x01 : hA - x00 - x01;               // het_A with hom-alt_B: listed, or everyone of het_A that is not listed in B
		QH = !fA ? x10 : hB - x00 - x10;
		if (!fA && !fB) QQ = x11;
		else if (!fA && fB) QQ = rA - x10 - x11;       // hom-alt_A (listed) with hom-alt_B (not listed)
		else if (fA && !fB) QQ = rB - x01 - x11;
		else QQ = n_samples - ((hA + rA) + (hB + rB) - (x00 + x01 + x10 + x11));
		// the interval screen (see ScreenCountsUnphased): T = 2N, ALT / REF allele counts, no division
		const double T = w.two_n, eps = 1e-5 * (T * T);
		const double da = (double)(hA + 2u * qA), db = (double)(hB + 2u * qB), ra = T - da, rb = T - db;
		const double n11 = (ra - db) + (double)(QH + HQ + 2u * QQ);
		const double e_lo = (n11 * T - ra * rb) - eps, e_hi = ((n11 + (double)HH) * T - ra * rb) + eps;
		const double bound = (w.cut * (da * ra)) * (db * rb);
		keep = !(e_lo * e_lo < bound && e_hi * e_hi < bound);
	}
	const unsigned long long ballot = __ballot(keep);
	if (ballot) {
		const int lane = threadIdx.x & 63;
		const int leader = __ffsll((long long)ballot) - 1;
		unsigned long long base = 0;
		if (lane == leader) base = atomicAdd(w.n_cand, (unsigned long long)__popcll(ballot));
		base = __shfl(base, leader);
		if (keep) {
			const unsigned long long slot = base + __popcll(ballot & ((1ull << lane) - 1));
			if (slot < w.cap) { uint32_t* e = w.cand + slot * 6; e[0] = i; e[1] = j; e[2] = HH; e[3] = HQ; e[4] = QH; e[5] = QQ; }
		}
	}
}


// ---- rare x common: a rare variant's carriers probing a common variant's row -----------------------------------------
// The reference's list kernel walks the *shorter* of the two carrier lists and tests the partner's bitvector
// (PhasedListVector, lib/ld/ld_engine.cpp:230-242: O(min carriers) per pair).  The merge kernels above need a list on both
// sides; a pair of a zone variant with a variant *outside* the zone (more carriers than a list holds) was contracted densely
// until round 4.  Measured (csrc/tools/probe_vs_dense.hip, profiles/r04_probe_vs_dense.txt): at 2N = 2,000,000 a probe of
// AC carriers into a 250 KB row costs 31 / 126 / 293 / 660 / 1,485 ps per pair for AC = 2 / 10 / 30 / 100 / 300 against
// 2,614 ps for the dense pair - it wins up to ~550 carriers, i.e. for every row of the zone (lists are kept up to W / 128 =
// 488 carriers) - and at 2N = 131,072 up to ~80 (lists: up to 32).  Layout as measured: a wave is 64 consecutive zone rows
// against ONE column - every lane walks its own list and reads the column's word h / 32 - and the blocks of one column run
// next to each other (block id = column x row blocks + row block), so the column's row stays in L2 while the zone works
// through it.  From the count on, the pair goes the way of the merge kernels: screen, candidate, list math.
struct ProbeWork {
	ListWork lw;                                   // lists, band, screen, candidate list (n_list: the zone; row0 / n_rows: the zone rows of this launch)
	const uint32_t* rows; uint32_t W;              // the sorted plane set's rows (phased: one per variant; unphased: H and Q)
	uint32_t col0, n_cols;                         // the columns of this launch: set positions [col0, col0 + n_cols); of a column inside the zone only the rows above it count
	uint32_t n_row_blocks;                         // ceil(n_rows / 256)
};

__device__ __forceinline__ void append_candidate3(const ListWork& w, bool keep, uint32_t i, uint32_t j, uint32_t aa) {
	const unsigned long long ballot = __ballot(keep);
	if (!ballot) return;
	const int lane = threadIdx.x & 63;
	const int leader = __ffsll((long long)ballot) - 1;
	unsigned long long base = 0;
	if (lane == leader) base = atomicAdd(w.n_cand, (unsigned long long)__popcll(ballot));
	base = __shfl(base, leader);
	if (keep) {
		const unsigned long long slot = base + __popcll(ballot & ((1ull << lane) - 1));
		if (slot < w.cap) { uint32_t* e = w.cand + slot * 3; e[0] = i; e[1] = j; e[2] = aa; }
	}
}

__global__ __launch_bounds__(256)
void k_probe_screen(const ProbeWork p) {
	const ListWork& w = p.lw;
	const uint32_t cb = blockIdx.x / p.n_row_blocks, rb = blockIdx.x - cb * p.n_row_blocks;
	const uint32_t i = w.row0 + rb * 256 + threadIdx.x, j = p.col0 + cb;
	bool keep = false;
	uint32_t aa = 0;
	if (i < w.row0 + w.n_rows) {
		const uint32_t limit = w.col_hi ? w.hi_b0 + w.col_hi[i - w.hi_a0] : 0xFFFFFFFFu;
		if (j < limit && j > i) {                              // (j > i: columns inside the zone - the triangle's other half belongs to row j)
			const uint32_t* a = w.lists + (size_t)i * w.stride;
			const uint32_t* row = p.rows + (size_t)j * p.W;
			const uint32_t na = w.mac[i];
			uint32_t x = 0;
			uint32_t k = 0;
			for (; k + 4 <= na; k += 4) {                  // four entries' loads in flight together (see k_probe_screen_unphased_t)
				const uint32_t h0 = a[k], h1 = a[k + 1], h2 = a[k + 2], h3 = a[k + 3];
				const uint32_t r0 = row[h0 >> 5], r1 = row[h1 >> 5], r2 = row[h2 >> 5], r3 = row[h3 >> 5];
				x += ((r0 >> (h0 & 31u)) & 1u) + ((r1 >> (h1 & 31u)) & 1u) + ((r2 >> (h2 & 31u)) & 1u) + ((r3 >> (h3 & 31u)) & 1u);
			}
			for (; k < na; ++k) { const uint32_t h = a[k]; x += (row[h >> 5] >> (h & 31u)) & 1u; }
			const uint32_t acA = w.rowpop[i], acB = w.rowpop[j];
			aa = w.flip[i] ? acB - x : x;                  // the list holds the carriers of A's minor allele: ALT, or (flip) REF - then ALT_A & ALT_B = ALT_B minus those
			const double da = (double)acA, db = (double)acB;
			const double dn = w.two_n * (double)aa - da * db;
			keep = dn != 0.0 && dn * dn >= w.cut * (da * (w.two_n - da)) * (db * (w.two_n - db));
		}
	}
	append_candidate3(w, keep, i, j, aa);
}

// UnphasedMath: the list holds A's samples that are not homozygous for its major allele as (sample << 1) | g (g = 0: het,
// g = 1: the rare homozygote - hom-alt, or hom-ref when flip); the column is its H (het) and Q (hom-alt) plane rows.
template <int UNROLL>
__global__ __launch_bounds__(256)
void k_probe_screen_unphased_t(const ProbeWork p, uint32_t n_samples) {
	const ListWork& w = p.lw;
	const uint32_t cb = blockIdx.x / p.n_row_blocks, rb = blockIdx.x - cb * p.n_row_blocks;
	const uint32_t i = w.row0 + rb * 256 + threadIdx.x, j = p.col0 + cb;
	bool keep = false;
	uint32_t HH = 0, HQ = 0, QH = 0, QQ = 0;
	if (i < w.row0 + w.n_rows) {
		const uint32_t limit = w.col_hi ? w.hi_b0 + w.col_hi[i - w.hi_a0] : 0xFFFFFFFFu;
		if (j < limit && j > i) {
			const uint32_t* a = w.lists + (size_t)i * w.stride;
			const uint32_t* H = p.rows + (size_t)(2 * j) * p.W;
			const uint32_t* Q = H + p.W;
			const uint32_t na = w.mac[i];
			// Four 16-bit counters [gA][class of B: 0 het, 1 hom-alt], two to a 32-bit word: x0 for A's hets, x1 for its rare
			// homozygotes.  (Not one 64-bit word shifted by 16 * (2 gA + class): a v_lshlrev_b64 whose shift amount the register
			// allocator happens to put into the *last VGPR the wave owns* gives wrong results on the MI355X boxes of this pool - the
			// "shift64 high register" erratum LLVM works around for gfx90a only.  Found when an unrolled form of this loop moved the
			// amount into v31 of 32 / v47 of 48 and one group of rows per run came out wrong; csrc/tools/shift64_probe.hip reproduces
			// it with three instructions, tests/test_build.py scans every kernel of the library for the pattern.)
			uint32_t x0 = 0, x1 = 0;
			auto tally = [&](uint32_t e, uint32_t hw, uint32_t qw) {
				const uint32_t sm = (e >> 1) & 31u;
				const uint32_t add = ((hw >> sm) & 1u) | (((qw >> sm) & 1u) << 16);
				x0 += (e & 1u) ? 0u : add;
				x1 += (e & 1u) ? add : 0u;
			};
			// UNROLL list entries at a time: their probes do not depend on each other, and with the loads of four entries in flight
			// a lane waits for memory once where it waited four times (the probes of the 1 M x 50 k cohort run: 314 -> 230 ms)
			uint32_t k = 0;
			if (UNROLL > 1) {
				for (; k + UNROLL <= na; k += UNROLL) {
					uint32_t e[UNROLL], hw[UNROLL], qw[UNROLL];
#pragma unroll
					for (int u = 0; u < UNROLL; ++u) e[u] = a[k + u];
#pragma unroll
					for (int u = 0; u < UNROLL; ++u) { hw[u] = H[e[u] >> 6]; qw[u] = Q[e[u] >> 6]; }
#pragma unroll
					for (int u = 0; u < UNROLL; ++u) tally(e[u], hw[u], qw[u]);
				}
			}
			for (; k < na; ++k) { const uint32_t e = a[k]; tally(e, H[e >> 6], Q[e >> 6]); }
			const uint32_t x0h = x0 & 0xFFFFu, x0q = x0 >> 16, x1h = x1 & 0xFFFFu, x1q = x1 >> 16;
			const uint32_t hA = w.rowpop[2 * i], qA = w.rowpop[2 * i + 1], hB = w.rowpop[2 * j], qB = w.rowpop[2 * j + 1];
			HH = x0h; HQ = x0q;
			if (!w.flip[i]) { QH = x1h; QQ = x1q; }                                   // A's hom-alt samples are the listed rare homozygotes
			else { QH = hB - x0h - x1h; QQ = qB - x0q - x1q; }                          // A's hom-alt samples are everyone *not* listed
			const double T = w.two_n, eps = 1e-5 * (T * T);
			const double da = (double)(hA + 2u * qA), db = (double)(hB + 2u * qB), ra = T - da, rbb = T - db;
			const double n11 = (ra - db) + (double)(QH + HQ + 2u * QQ);
			const double e_lo = (n11 * T - ra * rbb) - eps, e_hi = ((n11 + (double)HH) * T - ra * rbb) + eps;
			const double bound = (w.cut * (da * ra)) * (db * rbb);
			keep = !(e_lo * e_lo < bound && e_hi * e_hi < bound);
			(void)n_samples;
		}
	}
	const unsigned long long ballot = __ballot(keep);
	if (ballot) {
		const int lane = threadIdx.x & 63;
		const int leader = __ffsll((long long)ballot) - 1;
		unsigned long long base = 0;
		if (lane == leader) base = atomicAdd(w.n_cand, (unsigned long long)__popcll(ballot));
		base = __shfl(base, leader);
		if (keep) {
			const unsigned long long slot = base + __popcll(ballot & ((1ull << lane) - 1));
			if (slot < w.cap) { uint32_t* e = w.cand + slot * 6; e[0] = i; e[1] = j; e[2] = HH; e[3] = HQ; e[4] = QH; e[5] = QQ; }
		}
	}
}

// ---- probes through LDS (round 5) ---------------------------------------------------------------------------------------
// What bounds the probe kernels above is not latency but requests: every probe is a gather of one word from a 250 KB row -
// its own cache line, its own request to L2 - and every list entry read is another (a block's lists do not fit the CU's L1):
// two L2 requests per (pair, carrier), 2.4e11 a second on the whole chip (round 5's strip kernels - a block against a strip of columns, 16
// gathers in flight per lane instead of 4 - were slower at every width and are gone: profiles/r05_probe_kernels.txt).  Here the column rows come to the probes instead: a block is
// PROBE_ROWS zone rows against C columns, and walks the columns' rows segment by segment - SEGW words of each of the C rows are
// copied into LDS with coalesced 16-byte loads (1/2 KB per pair instead of one line per carrier), and every lane tests the
// carriers of its (ascending) list that fall into the segment against all C rows *in LDS*: C ds_read_b32 per carrier, no L2
// request but the list entry itself, read four at a time and once per C columns.  Lists end in LIST_END, which lies behind every
// segment.  One candidate reservation per wave and strip; counts, screen and candidates as in the kernels above.
constexpr int PROBE_ROWS = 512;          // zone rows (threads) of a block
constexpr int PROBE_SEGW = 4096;         // words of a column row staged at a time (16 KiB per row)

// The lane's next four list entries (a[k .. k + 3]; LIST_END behind the list's end).
__device__ __forceinline__ void probe_next4(const uint32_t* __restrict__ a, uint32_t k, uint32_t na, uint32_t (&q)[4]) {
#pragma unroll
	for (int u = 0; u < 4; ++u) q[u] = k + u < na ? a[k + u] : LIST_END;      // (nothing is read behind the list's own entries)
}

template <int C>
__global__ __launch_bounds__(PROBE_ROWS)
void k_probe_lds_t(const ProbeWork p) {
	__shared__ uint32_t seg[C * PROBE_SEGW];
	const ListWork& w = p.lw;
	const uint32_t sb = blockIdx.x / p.n_row_blocks, rb = blockIdx.x - sb * p.n_row_blocks;
	const uint32_t i = w.row0 + rb * PROBE_ROWS + threadIdx.x, j0 = p.col0 + sb * C;
	const uint32_t col_end = p.col0 + p.n_cols;
	const bool row_ok = i < w.row0 + w.n_rows;
	const uint32_t limit0 = row_ok ? (w.col_hi ? w.hi_b0 + w.col_hi[i - w.hi_a0] : 0xFFFFFFFFu) : 0u;
	const uint32_t limit = limit0 < col_end ? limit0 : col_end;          // first column the row does not take
	const bool any = row_ok && j0 < limit && j0 + C > i + 1;              // some column of the strip lies in (i, limit)
	if (!__syncthreads_or(any ? 1 : 0)) return;                           // (a strip no row of the block reaches: below the diagonal, beyond the band)
	const uint32_t na = any ? w.mac[i] : 0u;
	const uint32_t* a = w.lists + (size_t)i * w.stride;
	const uint32_t* base = p.rows + (size_t)j0 * p.W;                     // (uniform) row of the strip's first column
	uint32_t x[C];
#pragma unroll
	for (int c = 0; c < C; ++c) x[c] = 0;
	uint32_t k = 0, q[4];
	probe_next4(a, 0, na, q);
	const uint32_t nseg = (p.W + PROBE_SEGW - 1) / PROBE_SEGW;
	for (uint32_t sg = 0; sg < nseg; ++sg) {
		const uint32_t w0 = sg * PROBE_SEGW, w1 = w0 + PROBE_SEGW;
		__syncthreads();                                                  // everyone is done with the previous segment
#pragma unroll
		for (int r = 0; r < C * PROBE_SEGW / 4 / PROBE_ROWS; ++r) {         // C x SEGW words, 16 bytes per thread and turn
			const uint32_t v4 = (uint32_t)r * PROBE_ROWS + threadIdx.x;     // 16-byte piece of the staged block
			const uint32_t c = v4 / (PROBE_SEGW / 4), wq = v4 - c * (PROBE_SEGW / 4);
			const uint32_t word = w0 + 4 * wq;
			uint4 v = make_uint4(0, 0, 0, 0);
			if (word < p.W) v = *reinterpret_cast<const uint4*>(base + (size_t)c * p.W + word);      // (W is a multiple of 32 words: whole pieces)
			*reinterpret_cast<uint4*>(&seg[c * PROBE_SEGW + 4 * wq]) = v;
		}
		__syncthreads();
		for (;;) {                                                        // the lane's carriers inside [w0, w1): four at a time
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const uint32_t wd = q[u] >> 5;
				if (wd >= w0 && wd < w1) {
					const uint32_t off = wd - w0, sh = q[u] & 31u;
#pragma unroll
					for (int c = 0; c < C; ++c) x[c] += (seg[c * PROBE_SEGW + off] >> sh) & 1u;
				}
			}
			if ((q[3] >> 5) >= w1) break;                                  // the batch reaches into the next segment (or the list is at its end): keep it
			k += 4;
			probe_next4(a, k, na, q);
		}
	}
	uint32_t keepmask = 0;
	if (any) {
		const uint32_t acA = w.rowpop[i];
		const bool flip = w.flip[i] != 0;
		const double da = (double)acA, fA = w.cut * (da * (w.two_n - da));
#pragma unroll
		for (int c = 0; c < C; ++c) {
			const uint32_t j = j0 + c;
			if (j < limit && j > i) {                                  // (j > i: columns inside the zone - the triangle's other half belongs to row j)
				const uint32_t acB = w.rowpop[j];
				x[c] = flip ? acB - x[c] : x[c];
				const double db = (double)acB;
				const double dn = w.two_n * (double)x[c] - da * db;
				if (dn != 0.0 && dn * dn >= fA * (db * (w.two_n - db))) keepmask |= 1u << c;
			}
		}
	}
	if (__ballot(keepmask != 0)) {
		const int lane = threadIdx.x & 63;
		const uint32_t cnt = __popc(keepmask);
		const uint32_t incl = wave_scan_inclusive(cnt);
		const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		unsigned long long first = 0;
		if (lane == 0) first = atomicAdd(w.n_cand, (unsigned long long)total);
		first = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)first) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(first >> 32)) << 32;
		unsigned long long slot = first + (incl - cnt);
#pragma unroll
		for (int c = 0; c < C; ++c)
			if ((keepmask >> c) & 1u) {
				if (slot < w.cap) { uint32_t* e = w.cand + slot * 3; e[0] = i; e[1] = j0 + c; e[2] = x[c]; }
				++slot;
			}
	}
}

// UnphasedMath: C columns = 2 C plane rows (H, Q) of SEGW words each; entries (sample << 1) | g.
template <int C>
__global__ __launch_bounds__(PROBE_ROWS)
void k_probe_lds_unphased_t(const ProbeWork p) {
	__shared__ uint32_t seg[2 * C * PROBE_SEGW];
	const ListWork& w = p.lw;
	const uint32_t sb = blockIdx.x / p.n_row_blocks, rb = blockIdx.x - sb * p.n_row_blocks;
	const uint32_t i = w.row0 + rb * PROBE_ROWS + threadIdx.x, j0 = p.col0 + sb * C;
	const uint32_t col_end = p.col0 + p.n_cols;
	const bool row_ok = i < w.row0 + w.n_rows;
	const uint32_t limit0 = row_ok ? (w.col_hi ? w.hi_b0 + w.col_hi[i - w.hi_a0] : 0xFFFFFFFFu) : 0u;
	const uint32_t limit = limit0 < col_end ? limit0 : col_end;
	const bool any = row_ok && j0 < limit && j0 + C > i + 1;
	if (!__syncthreads_or(any ? 1 : 0)) return;
	const uint32_t na = any ? w.mac[i] : 0u;
	const uint32_t* a = w.lists + (size_t)i * w.stride;
	const uint32_t* base = p.rows + (size_t)(2 * j0) * p.W;               // (uniform) H row of the strip's first column; its Q row follows, then the next column's H row
	uint32_t x0[C], x1[C];                                                // per column [gA][class of B] in 16-bit halves, as in k_probe_screen_unphased_t
#pragma unroll
	for (int c = 0; c < C; ++c) { x0[c] = 0; x1[c] = 0; }
	uint32_t k = 0, q[4];
	probe_next4(a, 0, na, q);
	const uint32_t nseg = (p.W + PROBE_SEGW - 1) / PROBE_SEGW;
	for (uint32_t sg = 0; sg < nseg; ++sg) {
		const uint32_t w0 = sg * PROBE_SEGW, w1 = w0 + PROBE_SEGW;
		__syncthreads();
#pragma unroll
		for (int r = 0; r < 2 * C * PROBE_SEGW / 4 / PROBE_ROWS; ++r) {
			const uint32_t v4 = (uint32_t)r * PROBE_ROWS + threadIdx.x;
			const uint32_t c = v4 / (PROBE_SEGW / 4), wq = v4 - c * (PROBE_SEGW / 4);      // c: plane row of the strip (2 per column)
			const uint32_t word = w0 + 4 * wq;
			uint4 v = make_uint4(0, 0, 0, 0);
			if (word < p.W) v = *reinterpret_cast<const uint4*>(base + (size_t)c * p.W + word);
			*reinterpret_cast<uint4*>(&seg[c * PROBE_SEGW + 4 * wq]) = v;
		}
		__syncthreads();
		for (;;) {
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const uint32_t wd = q[u] >> 6;
				if (wd >= w0 && wd < w1) {
					const uint32_t off = wd - w0, sm = (q[u] >> 1) & 31u;
					const bool rare = (q[u] & 1u) != 0;
#pragma unroll
					for (int c = 0; c < C; ++c) {
						const uint32_t add = ((seg[(2 * c) * PROBE_SEGW + off] >> sm) & 1u) | (((seg[(2 * c + 1) * PROBE_SEGW + off] >> sm) & 1u) << 16);
						x0[c] += rare ? 0u : add;
						x1[c] += rare ? add : 0u;
					}
				}
			}
			if ((q[3] >> 6) >= w1) break;
			k += 4;
			probe_next4(a, k, na, q);
		}
	}
	uint32_t keepmask = 0;
	uint32_t HQ_[C], QH_[C], QQ_[C];
	if (any) {
		const uint32_t hA = w.rowpop[2 * i], qA = w.rowpop[2 * i + 1];
		const bool flip = w.flip[i] != 0;
		const double T = w.two_n, eps = 1e-5 * (T * T);
		const double da = (double)(hA + 2u * qA), ra = T - da, fA = w.cut * (da * ra);
#pragma unroll
		for (int c = 0; c < C; ++c) {
			const uint32_t j = j0 + c;
			HQ_[c] = 0; QH_[c] = 0; QQ_[c] = 0;
			if (j < limit && j > i) {
				const uint32_t hB = w.rowpop[2 * j], qB = w.rowpop[2 * j + 1];
				const uint32_t x0h = x0[c] & 0xFFFFu, x0q = x0[c] >> 16, x1h = x1[c] & 0xFFFFu, x1q = x1[c] >> 16;
				const uint32_t HH = x0h, HQ = x0q;
				uint32_t QH, QQ;
				if (!flip) { QH = x1h; QQ = x1q; }
				else { QH = hB - x0h - x1h; QQ = qB - x0q - x1q; }
				const double db = (double)(hB + 2u * qB), rbb = T - db;
				const double n11 = (ra - db) + (double)(QH + HQ + 2u * QQ);
				const double e_lo = (n11 * T - ra * rbb) - eps, e_hi = ((n11 + (double)HH) * T - ra * rbb) + eps;
				const double bound = fA * (db * rbb);
				if (!(e_lo * e_lo < bound && e_hi * e_hi < bound)) { keepmask |= 1u << c; x0[c] = HH; HQ_[c] = HQ; QH_[c] = QH; QQ_[c] = QQ; }
			}
		}
	}
	if (__ballot(keepmask != 0)) {
		const int lane = threadIdx.x & 63;
		const uint32_t cnt = __popc(keepmask);
		const uint32_t incl = wave_scan_inclusive(cnt);
		const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		unsigned long long first = 0;
		if (lane == 0) first = atomicAdd(w.n_cand, (unsigned long long)total);
		first = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)first) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(first >> 32)) << 32;
		unsigned long long slot = first + (incl - cnt);
#pragma unroll
		for (int c = 0; c < C; ++c)
			if ((keepmask >> c) & 1u) {
				if (slot < w.cap) { uint32_t* o = w.cand + slot * 6; o[0] = i; o[1] = j0 + c; o[2] = x0[c]; o[3] = HQ_[c]; o[4] = QH_[c]; o[5] = QQ_[c]; }
				++slot;
			}
	}
}

}  // namespace twk
