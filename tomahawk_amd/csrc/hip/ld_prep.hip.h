// Input preparation kernels: reference bitvector layout -> contraction planes.
//
// The reference keeps, per variant, one bitvector with two bits per sample
// (bit 2s / 2s+1 = first / second allele is ALT) and an optional mask with both
// bits of a sample set when either allele is missing (twk_igt_vec,
// include/core.h:724-753; built by lib/core.cpp:349-391).  The count kernel
// wants rows whose AND-popcounts are the cells of the contingency table:
//
//   phased, no missing   1 row  / variant: the raw bitvector (2N bits)
//   phased, missing      2 rows / variant: a' = a & ~m,  m            (2N bits)
//   unphased, no missing 2 rows / variant: H = a0 ^ a1 (het), Q = a0 & a1 (hom-alt)
//                                          de-interleaved to one bit per sample (N bits)
//   unphased, missing    3 rows / variant: H & ~M, Q & ~M, M  (M = sample missing)
//
// All rows are uint32 words, pitch a multiple of KC (=32) words, zero padded;
// the row count is padded to a multiple of 128 with zero rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace twk {

// splitmix64 finaliser; shared with the host twin in twk_synth.cpp.
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
// ALT-allele threshold of variant v: p_v ~ U(0.05, 0.5) as a 32-bit fraction.
__host__ __device__ __forceinline__ uint32_t synth_threshold(uint64_t seed, uint32_t v) {
	const uint64_t u = mix64(seed + 0x632BE59BD9B4E019ull * (uint64_t)(v + 1)) >> 11;    // 53 bits
	const double p = 0.05 + 0.45 * ((double)u * (1.0 / 9007199254740992.0));
	return (uint32_t)(p * 4294967296.0);
}
// Two alleles of sample s of variant v: low / high half of one 64-bit draw.
__host__ __device__ __forceinline__ uint32_t synth_sample_bits(uint64_t seed, uint32_t v, uint32_t s, uint32_t thr) {
	const uint64_t key = mix64(seed ^ (0xD1342543DE82EF95ull * (uint64_t)(v + 1)));
	const uint64_t x = mix64(key + s);
	return ((uint32_t)x < thr ? 1u : 0u) | ((uint32_t)(x >> 32) < thr ? 2u : 0u);
}

// Synthetic genotypes straight into the raw layout: one thread per 32-bit word
// (16 samples).  raw[v * Wp + w].
__global__ void k_synth(uint32_t* __restrict__ raw, uint32_t Wp, uint32_t n_samples, uint32_t n_variants,
                        uint64_t seed, uint32_t first_variant) {
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= Wp) return;
	for (uint32_t v = blockIdx.y; v < n_variants; v += gridDim.y) {     // grid.y is capped at 65535
		const uint32_t gv = first_variant + v;                           // global variant id
		const uint32_t thr = synth_threshold(seed, gv);
		uint32_t word = 0;
		const uint32_t s0 = w * 16;
		for (uint32_t i = 0; i < 16; ++i) {
			const uint32_t s = s0 + i;
			if (s < n_samples) word |= synth_sample_bits(seed, gv, s, thr) << (2 * i);
		}
		raw[(size_t)v * Wp + w] = word;
	}
}

// Population count of every row: out[r] = sum_k popc(rows[r][k]).  One wave per row.
__global__ void k_row_popcount(const uint32_t* __restrict__ rows, uint32_t W, uint32_t n_rows,
                               uint32_t* __restrict__ out) {
	const uint32_t r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (r >= n_rows) return;
	const int lane = threadIdx.x & 63;
	const uint4* p = reinterpret_cast<const uint4*>(rows + (size_t)r * W);
	uint32_t c = 0;
	for (uint32_t k = lane; k < W / 4; k += 64) {
		const uint4 x = p[k];
		c += __builtin_popcount(x.x) + __builtin_popcount(x.y) + __builtin_popcount(x.z) + __builtin_popcount(x.w);
	}
	for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
	if (lane == 0) out[r] = c;
}

// Keep the even bits of a 64-bit word, packed into 32 bits.
__device__ __forceinline__ uint32_t compress_even(uint64_t x) {
	x &= 0x5555555555555555ull;
	x = (x | (x >> 1)) & 0x3333333333333333ull;
	x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
	x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
	x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
	x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
	return (uint32_t)x;
}

// raw (2 bits / sample) -> unphased planes.  P = 2 (H,Q) or 3 (H,Q,M).
// One thread per output word (32 samples) of one variant.
__global__ void k_build_unphased(const uint32_t* __restrict__ raw, const uint32_t* __restrict__ rawmask,
                                 uint32_t Wp, uint32_t n_samples, uint32_t n_variants,
                                 uint32_t* __restrict__ planes, uint32_t Wu, int P,
                                 const uint32_t* __restrict__ ids) {
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= Wu) return;
	for (uint32_t slot = blockIdx.y; slot < n_variants; slot += gridDim.y) {
	const uint32_t v = ids ? ids[slot] : slot;     // ids: plane-set position -> variant (regrouped set)
	uint64_t x = 0, m = 0;
	if (2 * w + 1 < Wp || 2 * w < Wp) {
		const uint32_t lo = (2 * w < Wp) ? raw[(size_t)v * Wp + 2 * w] : 0;
		const uint32_t hi = (2 * w + 1 < Wp) ? raw[(size_t)v * Wp + 2 * w + 1] : 0;
		x = (uint64_t)lo | ((uint64_t)hi << 32);
		if (rawmask) {
			const uint32_t mlo = (2 * w < Wp) ? rawmask[(size_t)v * Wp + 2 * w] : 0;
			const uint32_t mhi = (2 * w + 1 < Wp) ? rawmask[(size_t)v * Wp + 2 * w + 1] : 0;
			m = (uint64_t)mlo | ((uint64_t)mhi << 32);
		}
	}
	const uint32_t a0 = compress_even(x), a1 = compress_even(x >> 1);
	uint32_t ms = compress_even(m) | compress_even(m >> 1);
	// samples beyond N do not exist
	const uint32_t s0 = w * 32;
	uint32_t valid = 0xFFFFFFFFu;
	if (s0 >= n_samples) valid = 0;
	else if (n_samples - s0 < 32) valid = (1u << (n_samples - s0)) - 1;
	ms &= valid;
	const uint32_t keep = valid & ~ms;
	planes[((size_t)slot * P + 0) * Wu + w] = (a0 ^ a1) & keep;
	planes[((size_t)slot * P + 1) * Wu + w] = (a0 & a1) & keep;
	if (P == 3) planes[((size_t)slot * P + 2) * Wu + w] = ms;
	}
}

// raw + mask -> phased-with-missing planes (a & ~m, m), 2 rows per variant.
__global__ void k_build_phased_masked(const uint32_t* __restrict__ raw, const uint32_t* __restrict__ rawmask,
                                      uint32_t Wp, uint32_t n_variants, uint32_t* __restrict__ planes) {
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= Wp) return;
	for (uint32_t v = blockIdx.y; v < n_variants; v += gridDim.y) {
		const uint32_t a = raw[(size_t)v * Wp + w], m = rawmask[(size_t)v * Wp + w];
		planes[((size_t)v * 2 + 0) * Wp + w] = a & ~m;
		planes[((size_t)v * 2 + 1) * Wp + w] = m;
	}
}

// out[slot] = raw[ids[slot]]: a permuted copy of the raw rows (the allele-count-sorted phased plane set).
__global__ void k_permute_rows(const uint32_t* __restrict__ raw, uint32_t Wp, uint32_t n_variants,
                               uint32_t* __restrict__ out, const uint32_t* __restrict__ ids) {
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= Wp) return;
	for (uint32_t slot = blockIdx.y; slot < n_variants; slot += gridDim.y) out[(size_t)slot * Wp + w] = raw[(size_t)ids[slot] * Wp + w];
}

// Zero the bits of the raw layout that lie beyond allele 2N (defensive: the
// reference guarantees it, core.cpp:361).
__global__ void k_clear_tail(uint32_t* __restrict__ raw, uint32_t Wp, uint32_t n_samples, uint32_t n_variants) {
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= Wp) return;
	const uint64_t bit0 = (uint64_t)w * 32, nb = 2ull * n_samples;
	if (bit0 + 32 <= nb) return;
	const uint32_t keep = bit0 >= nb ? 0u : ((1u << (uint32_t)(nb - bit0)) - 1);
	for (uint32_t v = blockIdx.y; v < n_variants; v += gridDim.y) raw[(size_t)v * Wp + w] &= keep;
}

// ---- T1 on the device: run-length genotypes -> bitvector + mask ---------------------------------
// Device counterpart of twk_igt_vec::Build (lib/core.cpp:349-391): the variant's run words exactly as
// they sit in a .twk block (twk1_t::gt, 1 / 2 / 4 bytes per run, little endian; lib/core.h:195-205,
// lib/genotype_encoder.h:277-343) are expanded to the raw layout (bit 2s / 2s+1 = first / second allele
// of sample s is ALT) and, for variants with missing genotypes, the mask (both bits of a sample set when
// either allele is missing, core.cpp:379-380).  A run word is  length << (2 + 2m) | A << (1 + m) | B
// with m = 1 and two bits per allele (0 ref, 1 alt, 2 missing) when the variant has missing genotypes,
// m = 0 and one bit per allele otherwise.
//
// One 256-thread block per variant, two phases:
//   1. every thread sums the lengths of groups of 32 runs, a block scan turns the sums into the first
//      sample of every group (scratch: one u32 per group);
//   2. every thread produces spans of 8 output words (128 samples): binary search for the group that
//      holds the span's first sample, walk the runs, OR the allele patterns in, store.
// Output-centric, so no atomics, every row word is written exactly once (zero where nothing is ALT) and
// the padding beyond 2N bits is zero.  status[0] is set to 1 if any variant's runs do not add up to
// n_samples (corrupt input).
struct RleDesc {
	unsigned long long off;     // byte offset of the variant's first run word in `bytes`
	uint32_t n_runs;
	uint32_t width_missing;     // bits 0-7: bytes per run word (1, 2, 4); bit 8: variant has missing genotypes
};
constexpr int RLE_GROUP = 32;          // runs per scan group
constexpr int RLE_SPAN_WORDS = 8;      // 32-bit output words per paint step of a thread

__device__ __forceinline__ uint32_t rle_word(const uint8_t* __restrict__ p, uint32_t i, uint32_t width) {
	p += (size_t)i * width;
	if (width == 1) return p[0];
	if (width == 2) return (uint32_t)p[0] | (uint32_t)p[1] << 8;
	return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}

__global__ __launch_bounds__(256)
void k_inflate_rle(const uint8_t* __restrict__ bytes, const RleDesc* __restrict__ desc,
                   const unsigned long long* __restrict__ group_base,   // first scratch slot of every variant
                   uint32_t* __restrict__ scratch, uint32_t* __restrict__ raw, uint32_t* __restrict__ rawmask,
                   uint32_t Wp, uint32_t n_samples, uint32_t first_row, int* __restrict__ status) {
	const uint32_t v = blockIdx.x;
	const RleDesc d = desc[v];
	const uint32_t width = d.width_missing & 0xFFu, m = (d.width_missing >> 8) & 1u;
	const uint32_t shift = 2 + 2 * m, amask = (1u << (1 + m)) - 1;
	const uint8_t* runs = bytes + d.off;
	const uint32_t n_groups = (d.n_runs + RLE_GROUP - 1) / RLE_GROUP;
	uint32_t* gstart = scratch + group_base[v];                  // first sample of every group
	__shared__ uint32_t warp_sum[4];
	__shared__ uint32_t carry_s;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	if (tid == 0) carry_s = 0;
	__syncthreads();
	// ---- phase 1: exclusive scan of the group sums -------------------------------------------
	for (uint32_t g0 = 0; g0 < n_groups; g0 += 256) {
		const uint32_t g = g0 + tid;
		uint32_t sum = 0;
		if (g < n_groups) {
			const uint32_t r1 = min(d.n_runs, (g + 1) * RLE_GROUP);
			for (uint32_t r = g * RLE_GROUP; r < r1; ++r) sum += rle_word(runs, r, width) >> shift;
		}
		uint32_t inc = sum;                                     // inclusive scan within the wave
		for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
		if (lane == 63) warp_sum[wv] = inc;
		__syncthreads();
		uint32_t before = carry_s;
		for (int k = 0; k < wv; ++k) before += warp_sum[k];
		if (g < n_groups) gstart[g] = before + inc - sum;
		__syncthreads();
		if (tid == 255) carry_s = before + inc;
		__syncthreads();
	}
	const uint32_t total = carry_s;
	if (total != n_samples) { if (tid == 0) status[0] = 1; }
	// ---- phase 2: paint -----------------------------------------------------------------------------
	uint32_t* row = raw + (size_t)(first_row + v) * Wp;
	uint32_t* mrow = rawmask ? rawmask + (size_t)(first_row + v) * Wp : nullptr;
	const uint32_t n_spans = (Wp + RLE_SPAN_WORDS - 1) / RLE_SPAN_WORDS;
	const uint32_t live = min(total, n_samples);                // samples the runs actually describe
	for (uint32_t sp = tid; sp < n_spans; sp += 256) {
		uint32_t out[RLE_SPAN_WORDS], outm[RLE_SPAN_WORDS];
#pragma unroll
		for (int k = 0; k < RLE_SPAN_WORDS; ++k) { out[k] = 0; outm[k] = 0; }
		const uint32_t s0 = sp * (RLE_SPAN_WORDS * 16);          // first sample of the span
		const uint32_t s1 = min(live, s0 + RLE_SPAN_WORDS * 16);
		if (s0 < s1 && n_groups) {
			// largest group g with gstart[g] <= s0
			uint32_t lo = 0, hi = n_groups;                     // invariant: gstart[lo] <= s0 (gstart[0] = 0)
			while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (gstart[mid] <= s0) lo = mid; else hi = mid; }
			uint32_t r = lo * RLE_GROUP, cur = gstart[lo];      // run r starts at sample cur
			while (r < d.n_runs && cur < s1) {
				const uint32_t wd = rle_word(runs, r, width);
				const uint32_t len = wd >> shift, a = (wd >> (1 + m)) & amask, b = wd & amask;
				const uint32_t e = cur + len;
				if (e > s0 && (a | b)) {
					const uint32_t p0 = max(cur, s0) - s0, p1 = min(e, s1) - s0;       // samples [p0, p1) of the span
					const uint32_t pat = (a == 1 ? 0x55555555u : 0u) | (b == 1 ? 0xAAAAAAAAu : 0u);
					const uint32_t miss = (a == 2 || b == 2) ? 0xFFFFFFFFu : 0u;
#pragma unroll
					for (int k = 0; k < RLE_SPAN_WORDS; ++k) {
						const uint32_t w0 = k * 16, w1 = w0 + 16;
						if (p1 > w0 && p0 < w1) {
							const uint32_t lo_s = max(p0, w0) - w0, hi_s = min(p1, w1) - w0;       // samples [lo_s, hi_s) of word k
							const uint32_t bits = (hi_s - lo_s) * 2;
							const uint32_t rng = (bits == 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u)) << (lo_s * 2);
							out[k] |= pat & rng; outm[k] |= miss & rng;
						}
					}
				}
				cur = e; ++r;
			}
		}
#pragma unroll
		for (int k = 0; k < RLE_SPAN_WORDS; ++k) {
			const uint32_t w = sp * RLE_SPAN_WORDS + k;
			if (w < Wp) { row[w] = out[k]; if (mrow) mrow[w] = outm[k]; }
		}
	}
}

}  // namespace twk
