// The three-product form of UnphasedMath's contraction (ld_count.hip.h: contract3_half, StoreCounts3, ScreenCountsUnphased<TB, true>)
// leaves (HH, S) per variant pair.  This header holds what follows it:
//   k_screen3_pairs     the r2 screen over an (HH, S) count matrix (long rows, where tiles are split along K)
//   k_recount_unphased  the four products HH, HQ, QH, QQ of the pairs that passed, counted afresh from their plane rows
// The candidates then go through k_ld_stats_list_unphased (ld_math.hip.h) like those of the four-product fused form.
// Reference shape: UnphasedMath reads the 3 x 3 table only through n11, the double hets and the margins before its cubic
// (lib/ld/ld_engine.cpp:1334-1375); PhasedListVector counts one cell and derives the rest (ld_engine.cpp:244-246).
#pragma once
#include "ld_count.hip.h"
#include "ld_math.hip.h"

namespace twk {

// ---- the three-product form's screen over a count matrix (long rows: tiles split along K, counts added into C) ----
// One thread per variant pair of the super-tile: (HH, S) from the matrix StoreCounts3 wrote, the row margins, and the test of
// ScreenCountsUnphased word for word (same doubles, same order); a pair that passes becomes a candidate (A, B, HH, S, -, -) in the
// launch's list, appended with one atomic per block.  Structural tests as in the fused epilogue (the list math checks them again).
constexpr int SCREEN3_THREADS = 256;
__global__ __launch_bounds__(SCREEN3_THREADS)
void k_screen3_pairs(const ScreenWork* __restrict__ sp, const StatsParams* __restrict__ pp, const uint32_t* __restrict__ C, uint32_t ldc) {
	const ScreenWork& s = *sp;
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
	const uint32_t vA = s.a0 + i, vB = s.b0 + j;
	bool ok = i < s.nA && j < s.nB && vA < s.n_variants && vB < s.n_variants && (!s.diag || vB > vA);
	if (ok && s.col_hi) { const uint32_t k = vA - s.hi_a0; ok = vB < s.hi_b0 + s.col_hi[k < s.hi_n ? k : s.hi_n - 1]; }
	if (ok && ((vA < s.list_zone && vB < s.list_zone) || vA < s.probe_zone)) ok = false;
	if (ok && (pp->window & TWK_HIP_OPT_WINDOW)) {      // window mode: only the tiles some row can reach were contracted - the exact test of d_pair, so that
		const StatsParams& p = *pp;                     // nothing is read from a tile that holds no counts
		const uint32_t A = p.tv.ids ? p.tv.ids[vA] : vA, B = p.tv.ids ? p.tv.ids[vB] : vB;
		const int64_t d = (int64_t)p.vm.pos[A] - (int64_t)p.vm.pos[B];
		if (p.vm.rid[A] != p.vm.rid[B] || (d < 0 ? -d : d) > (int64_t)p.l_window) ok = false;
	}
	uint32_t hh = 0, s_sum = 0;
	if (ok) {
		const uint2 hs = *reinterpret_cast<const uint2*>(C + (size_t)i * ldc + 2 * j);
		hh = hs.x; s_sum = hs.y;
		const uint32_t hA = s.rowpop[2 * vA], qA = s.rowpop[2 * vA + 1], hB = s.rowpop[2 * vB], qB = s.rowpop[2 * vB + 1];
		const double T2n = s.two_n, eps = 1e-5 * (T2n * T2n);
		const double da = (double)(hA + 2u * qA), ra = T2n - da, fA = s.cut * (da * ra);
		const double db = (double)(hB + 2u * qB), rb = T2n - db, fB = db * rb;
		const double n11 = (ra - db) + (double)s_sum;
		const double e_lo = (n11 * T2n - ra * rb) - eps;
		const double e_hi = ((n11 + (double)hh) * T2n - ra * rb) + eps;
		const double bound = fA * fB;
		ok = !(e_lo * e_lo < bound && e_hi * e_hi < bound);
	}
	__shared__ uint32_t wave_n[SCREEN3_THREADS / 64];
	__shared__ unsigned long long block_base;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned long long ballot = __ballot(ok);
	if (lane == 0) wave_n[wave] = (uint32_t)__popcll(ballot);
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t total = 0;
		for (int k = 0; k < SCREEN3_THREADS / 64; ++k) total += wave_n[k];
		block_base = total ? atomicAdd(s.n_cand, (unsigned long long)total) : 0ull;
	}
	__syncthreads();
	if (ok) {
		unsigned long long slot = block_base;
		for (int k = 0; k < wave; ++k) slot += wave_n[k];
		slot += (unsigned long long)__popcll(ballot & ((1ull << lane) - 1));
		if (slot < s.cap) {
			uint32_t* e = s.cand + slot * 6;
			e[0] = vA; e[1] = vB; e[2] = hh; e[3] = 0; e[4] = 0; e[5] = s_sum;
		}
	}
}

// ---- the candidates' four products --------------------------------------------------------------------------------------
// The three-product forms hand over (A, B, HH, S); UnphasedMath needs HH, HQ, QH and QQ.  LANES lanes per candidate (64 for
// long rows, 16 - one DPP row - for rows of a few hundred words) stream the four plane rows of the pair with 16-byte loads
// and count all four products afresh - nothing of the three-product contraction is reused, so a record is exactly what
// the four-product form would have made it.  The few candidates of a default run (r2 >= 0.1: pairs in real LD) cost nothing
// next to the contraction; the host stops using the three-product forms when a launch's candidates are too many for that to
// hold (twk_hip.hip: three_ok).  The recount also proves the contraction, always: a candidate whose (HH, S) disagree with its
// own four products is counted in *mismatches, and the host fails the call on it.
template <int LANES>
__global__ __launch_bounds__(256)
void k_recount_unphased(const uint32_t* __restrict__ rows, uint32_t W, uint32_t W_live, uint32_t* __restrict__ cand,
                        const unsigned long long* __restrict__ n_cand, unsigned long long cap, unsigned long long* __restrict__ mismatches) {
	static_assert(LANES == 64 || LANES == 16, "a wave or a DPP row per candidate");
	const unsigned long long n = *n_cand;
	if (n > cap) return;                                    // the list overflowed: the host redoes the launch with four products, nothing of this one is kept
	const uint32_t lane = threadIdx.x & (LANES - 1);
	const unsigned long long group = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
	const unsigned long long n_groups = (unsigned long long)gridDim.x * blockDim.x / LANES;
	const uint32_t W4 = (W_live + 3) / 4;                 // 16-byte pieces of a row that carry data (rows are padded with zeros to W, a multiple of 32 words)
	for (unsigned long long k = group; k < n; k += n_groups) {
		uint32_t* e = cand + 6 * k;
		const uint32_t sA = e[0], sB = e[1];
		if (sA == CAND_UNUSED) continue;                    // (uniform over the candidate's lanes)
		const uint4* hA = reinterpret_cast<const uint4*>(rows + (size_t)(2 * sA) * W);
		const uint4* qA = reinterpret_cast<const uint4*>(rows + (size_t)(2 * sA + 1) * W);
		const uint4* hB = reinterpret_cast<const uint4*>(rows + (size_t)(2 * sB) * W);
		const uint4* qB = reinterpret_cast<const uint4*>(rows + (size_t)(2 * sB + 1) * W);
		uint32_t hh = 0, hq = 0, qh = 0, qq = 0;
		for (uint32_t p = lane; p < W4; p += LANES) {
			const uint4 a = hA[p], b = qA[p], x = hB[p], y = qB[p];
			hh += __popc(a.x & x.x) + __popc(a.y & x.y) + __popc(a.z & x.z) + __popc(a.w & x.w);
			hq += __popc(a.x & y.x) + __popc(a.y & y.y) + __popc(a.z & y.z) + __popc(a.w & y.w);
			qh += __popc(b.x & x.x) + __popc(b.y & x.y) + __popc(b.z & x.z) + __popc(b.w & x.w);
			qq += __popc(b.x & y.x) + __popc(b.y & y.y) + __popc(b.z & y.z) + __popc(b.w & y.w);
		}
#pragma unroll
		for (int d = 1; d < LANES; d <<= 1) {
			hh += __shfl_xor(hh, d, LANES); hq += __shfl_xor(hq, d, LANES); qh += __shfl_xor(qh, d, LANES); qq += __shfl_xor(qq, d, LANES);
		}
		if (lane == 0) {
			if (mismatches && (e[2] != hh || e[5] != qh + hq + 2u * qq)) atomicAdd(mismatches, 1ull);      // the contraction's (HH, S) against the recount
			e[2] = hh; e[3] = hq; e[4] = qh; e[5] = qq;
		}
	}
}

}  // namespace twk
