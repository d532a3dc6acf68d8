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

// Planted LD (twk_hip_plant, include/twk_hip.h): odd variant 2k + 1, k < n_planted, is not drawn by itself but is a noisy copy of
// the even variant 2 ((k mult + offset) mod half) - every allele of the source flipped with probability eps_k = max_eps u_k,
// u_k ~ U[0, 1) from the seed.  Sources are never copies themselves and (mult coprime with half) no two copies share one, so the
// data set holds exactly n_planted variant pairs in LD - from r2 = 1 (eps 0) down to (1 - 2 eps)^2 x a ratio of the heterozygosities -
// in a sea of unlinked variants.  Shared by the device generator and its host twin.
struct SynthPlant { uint32_t n_planted, half, mult, offset; uint32_t eps_scale; };      // eps_scale: max_eps as a 32-bit fraction
// -> true when global variant gv is a copy: its source and the 32-bit threshold of its flips
__host__ __device__ __forceinline__ bool synth_plant_source(uint64_t seed, const SynthPlant& pl, uint32_t gv, uint32_t& src, uint32_t& eps_thr) {
	if (!(gv & 1u) || (gv >> 1) >= pl.n_planted || pl.half == 0) return false;
	const uint32_t k = gv >> 1;
	src = 2u * (uint32_t)(((uint64_t)k * pl.mult + pl.offset) % pl.half);
	const uint64_t u = mix64(seed ^ (0xA24BAED4963EE407ull * (uint64_t)(k + 1))) >> 32;      // 32 bits
	eps_thr = (uint32_t)((u * (uint64_t)pl.eps_scale) >> 32);
	return true;
}
// the flips of sample s of copy gv: low / high half of one 64-bit draw, like the alleles themselves
__host__ __device__ __forceinline__ uint32_t synth_flip_bits(uint64_t seed, uint32_t gv, uint32_t s, uint32_t eps_thr) {
	const uint64_t key = mix64(seed ^ (0x9FB21C651E98DF25ull * (uint64_t)(gv + 1)));
	const uint64_t x = mix64(key + s);
	return ((uint32_t)x < eps_thr ? 1u : 0u) | ((uint32_t)(x >> 32) < eps_thr ? 2u : 0u);
}

// Synthetic genotypes straight into the raw layout: one thread per 32-bit word
// (16 samples).  raw[v * Wp + w].
__global__ void k_synth(uint32_t* __restrict__ raw, uint32_t Wp, uint32_t n_samples, uint32_t n_variants,
                        uint64_t seed, uint32_t first_variant, SynthPlant plant) {
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= Wp) return;
	for (uint32_t v = blockIdx.y; v < n_variants; v += gridDim.y) {     // grid.y is capped at 65535
		const uint32_t gv = first_variant + v;                           // global variant id
		uint32_t src = gv, eps_thr = 0;
		const bool copy = synth_plant_source(seed, plant, gv, src, eps_thr);
		const uint32_t thr = synth_threshold(seed, src);
		uint32_t word = 0;
		const uint32_t s0 = w * 16;
		for (uint32_t i = 0; i < 16; ++i) {
			const uint32_t s = s0 + i;
			if (s < n_samples) word |= (synth_sample_bits(seed, src, s, thr) ^ (copy ? synth_flip_bits(seed, gv, s, eps_thr) : 0u)) << (2 * i);
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
// Run-centric, a variant's runs cut into chunks of RLE_CHUNK_STEPS steps of 4096 bytes (16 bytes per
// thread: 16 / 8 / 4 runs of 1 / 2 / 4 bytes), one 256-thread block per chunk, so a common variant with
// half a million short runs is expanded by a hundred blocks and a rare one by one:
//   k_rle_chunk_sums   the samples every chunk covers (sum of its run lengths);
//   k_inflate_rle      a block adds up the chunks before its own (its first sample), then walks its
//                      steps: dword loads from the aligned address below the thread's first run +
//                      v_alignbit, the step after the current one already in flight; a block scan of
//                      the run lengths gives every run its first sample.
// The rows are zero when the kernel starts (the host clears them on the same stream), so only the runs
// that carry an ALT or a missing allele are painted: ORed into an LDS window of RLE_WIN row words that
// starts at the step's first word (ds_or), and the window is then stored to the row - plain coalesced
// stores, except the step's first and last word, which it shares with its neighbours (atomic OR; the
// neighbour may be another block).  Words of a step beyond the window (runs longer than 32 samples on
// average: few of them carry anything) are ORed straight into the row.  A run that covers 64 words or
// more is painted by its whole wave.  status[0] is set to 1 if a variant's runs do not add up to
// n_samples (corrupt input); nothing is ever written outside the variant's row.
struct RleDesc {
	unsigned long long off;     // byte offset of the variant's first run word in `bytes`
	uint32_t n_runs;
	uint32_t width_missing;     // bits 0-7: bytes per run word (1, 2, 4); bit 8: variant has missing genotypes
};
constexpr int RLE_STEP_BYTES = 16;         // run bytes per thread and step
constexpr uint32_t RLE_CHUNK_STEPS = 4;    // steps per block
constexpr uint32_t RLE_CHUNK_BYTES = RLE_CHUNK_STEPS * 256 * RLE_STEP_BYTES;
constexpr uint32_t RLE_WIN = 4096;         // row words per LDS window (65,536 samples)

// The part of samples [s, e) that lies in row word w gets `pat` (and all-ones in the mask row when miss):
// in the LDS window when the word is inside it, in the row itself otherwise.
__device__ __forceinline__ void rle_paint_word(uint32_t* __restrict__ row, uint32_t* __restrict__ mrow,
                                               uint32_t* win, uint32_t* winm, uint32_t w_base, uint32_t w,
                                               uint32_t s, uint32_t e, uint32_t pat, bool miss) {
	const uint32_t w_s = w << 4;
	const uint32_t lo = s > w_s ? s - w_s : 0u, hi = min(e - w_s, 16u);
	const uint32_t bits = (hi - lo) * 2;
	const uint32_t rng = (bits == 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u)) << (lo * 2);
	const uint32_t i = w - w_base;
	if (i < RLE_WIN) {
		if (pat) atomicOr(win + i, pat & rng);
		if (miss) atomicOr(winm + i, rng);
	} else {
		if (pat) atomicOr(row + w, pat & rng);
		if (miss) atomicOr(mrow + w, rng);
	}
}

struct RleRegs { uint32_t q[5]; };
__device__ __forceinline__ void rle_load(RleRegs& r, const uint8_t* __restrict__ runs, uint32_t first_run, uint32_t n_runs, uint32_t width) {
	if (first_run < n_runs) {
		const uintptr_t a = reinterpret_cast<uintptr_t>(runs) + (size_t)first_run * width;
		const uint32_t* q = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
#pragma unroll
		for (int i = 0; i < 5; ++i) r.q[i] = q[i];                 // up to 19 bytes past the last run: the buffer is padded
	} else {
#pragma unroll
		for (int i = 0; i < 5; ++i) r.q[i] = 0;
	}
}
// The thread's R runs of a step (0 beyond the variant's last run) and the samples they cover.
template <int WIDTH>
__device__ __forceinline__ unsigned long long rle_unpack(const RleRegs& regs, uint32_t sh, uint32_t r_first, uint32_t n_runs,
                                                         uint32_t shift, uint32_t (&run)[RLE_STEP_BYTES / WIDTH]) {
	constexpr int R = RLE_STEP_BYTES / WIDTH;
	uint32_t w[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) w[i] = __builtin_amdgcn_alignbit(regs.q[i + 1], regs.q[i], sh);
	const uint32_t nv = r_first < n_runs ? min((uint32_t)R, n_runs - r_first) : 0u;
	unsigned long long sum = 0;
#pragma unroll
	for (int j = 0; j < R; ++j) {
		uint32_t x;
		if (WIDTH == 4) x = w[j];
		else if (WIDTH == 2) x = (w[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
		else x = (w[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
		run[j] = (uint32_t)j < nv ? x : 0u;
		sum += run[j] >> shift;
	}
	return sum;
}
__device__ __forceinline__ unsigned long long rle_block_sum(unsigned long long x, unsigned long long* wave_tot) {
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
	if (lane == 0) wave_tot[wv] = x;
	__syncthreads();
	const unsigned long long t = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
	__syncthreads();
	return t;
}
// block -> (variant, chunk of the variant): chunk_base[v] = first block of variant v, ascending, chunk_base[count] = grid
__device__ __forceinline__ uint32_t rle_variant_of_block(const uint32_t* __restrict__ chunk_base, uint32_t count, uint32_t blk) {
	uint32_t lo = 0, hi = count;                                   // invariant: chunk_base[lo] <= blk < chunk_base[hi]
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (chunk_base[mid] <= blk) lo = mid; else hi = mid; }
	return lo;
}

template <int WIDTH>
__device__ __forceinline__ unsigned long long rle_chunk_sum(const uint8_t* __restrict__ runs, uint32_t n_runs, uint32_t m, uint32_t chunk) {
	constexpr int R = RLE_STEP_BYTES / WIDTH;
	const uint32_t shift = 2 + 2 * m, tid = threadIdx.x;
	const uint32_t sh = (uint32_t)((reinterpret_cast<uintptr_t>(runs) + (size_t)tid * RLE_STEP_BYTES) & 3) * 8;
	unsigned long long sum = 0;
	RleRegs regs[RLE_CHUNK_STEPS];
#pragma unroll
	for (uint32_t k = 0; k < RLE_CHUNK_STEPS; ++k) rle_load(regs[k], runs, (chunk * RLE_CHUNK_STEPS + k) * 256 * R + tid * R, n_runs, WIDTH);
#pragma unroll
	for (uint32_t k = 0; k < RLE_CHUNK_STEPS; ++k) {
		uint32_t run[R];
		sum += rle_unpack<WIDTH>(regs[k], sh, (chunk * RLE_CHUNK_STEPS + k) * 256 * R + tid * R, n_runs, shift, run);
	}
	return sum;
}

__global__ __launch_bounds__(256)
void k_rle_chunk_sums(const uint8_t* __restrict__ bytes, const RleDesc* __restrict__ desc, const uint32_t* __restrict__ chunk_base,
                      uint32_t count, unsigned long long* __restrict__ chunk_sum) {
	__shared__ unsigned long long wave_tot[4];
	const uint32_t v = rle_variant_of_block(chunk_base, count, blockIdx.x), chunk = blockIdx.x - chunk_base[v];
	const RleDesc d = desc[v];
	const uint32_t width = d.width_missing & 0xFFu, m = (d.width_missing >> 8) & 1u;
	const uint8_t* runs = bytes + d.off;
	unsigned long long s;
	if (width == 2) s = rle_chunk_sum<2>(runs, d.n_runs, m, chunk);
	else if (width == 1) s = rle_chunk_sum<1>(runs, d.n_runs, m, chunk);
	else s = rle_chunk_sum<4>(runs, d.n_runs, m, chunk);
	s = rle_block_sum(s, wave_tot);
	if (threadIdx.x == 0) chunk_sum[blockIdx.x] = s;
}

template <int WIDTH>
__device__ __forceinline__ void rle_inflate_chunk(const uint8_t* __restrict__ runs, const uint32_t n_runs, const uint32_t m,
                                                  const uint32_t chunk, unsigned long long pos, const bool last_chunk,
                                                  uint32_t* __restrict__ row, uint32_t* __restrict__ mrow,
                                                  const uint32_t n_samples, int* __restrict__ status, unsigned long long* wave_tot,
                                                  uint32_t* win, uint32_t* winm) {
	constexpr int R = RLE_STEP_BYTES / WIDTH;                      // runs per thread and step
	const uint32_t shift = 2 + 2 * m, amask = (1u << (1 + m)) - 1;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t sh = (uint32_t)((reinterpret_cast<uintptr_t>(runs) + (size_t)tid * RLE_STEP_BYTES) & 3) * 8;   // same for every step: a step is 4096 bytes
	const uint32_t r_begin = chunk * RLE_CHUNK_STEPS * 256 * R;
	const uint32_t r_end = min(n_runs, r_begin + RLE_CHUNK_STEPS * 256 * R);
	RleRegs cur_regs, next_regs;                                   // pos: first sample of the step (uniform)
	rle_load(cur_regs, runs, r_begin + (uint32_t)tid * R, n_runs, WIDTH);
	for (uint32_t r0 = r_begin; r0 < r_end; r0 += 256 * R) {
		const uint32_t r_first = r0 + (uint32_t)tid * R;
		if (r0 + 256 * R < r_end) rle_load(next_regs, runs, r_first + 256 * R, n_runs, WIDTH);
		uint32_t run[R];
		const unsigned long long sum = rle_unpack<WIDTH>(cur_regs, sh, r_first, n_runs, shift, run);
		unsigned long long inc = sum;                              // inclusive scan within the wave
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(inc, o); if (lane >= o) inc += t; }
		if (lane == 63) wave_tot[wv] = inc;
		__syncthreads();
		unsigned long long cur = pos + inc - sum, step_total = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) { const unsigned long long t = wave_tot[k]; if (k < wv) cur += t; step_total += t; }
		__syncthreads();                                           // wave_tot is rewritten by the next step
		// the step's words: [w_base, w_base + n_words)
		const unsigned long long step_end = pos + step_total < n_samples ? pos + step_total : n_samples;
		const uint32_t w_base = (uint32_t)(pos < n_samples ? pos : n_samples) >> 4;
		const uint32_t n_words = step_end > pos ? (uint32_t)((step_end - 1) >> 4) - w_base + 1 : 0u;
#pragma unroll
		for (int j = 0; j < R; ++j) {
			const uint32_t len = run[j] >> shift, a = (run[j] >> (1 + m)) & amask, b = run[j] & amask;
			const unsigned long long e64 = cur + len;
			const bool live = len != 0 && (a | b) != 0 && cur < n_samples;
			const uint32_t s = (uint32_t)cur, e = (uint32_t)(e64 < n_samples ? e64 : n_samples);
			const uint32_t pat = (a == 1 ? 0x55555555u : 0u) | (b == 1 ? 0xAAAAAAAAu : 0u);
			const bool miss = mrow != nullptr && (a == 2 || b == 2);
			const uint32_t wa = s >> 4, wb = live ? (e - 1) >> 4 : 0u;
			const bool wide = live && wb - wa >= 64;
			if (live && !wide) for (uint32_t x = wa; x <= wb; ++x) rle_paint_word(row, mrow, win, winm, w_base, x, s, e, pat, miss);
			unsigned long long todo = __ballot(wide);              // long runs: the whole wave paints
			while (todo) {
				const int src = __ffsll((long long)todo) - 1;
				todo &= todo - 1;
				const uint32_t s2 = __shfl(s, src), e2 = __shfl(e, src), pat2 = __shfl(pat, src);
				const bool miss2 = __shfl((int)miss, src) != 0;
				for (uint32_t x = (s2 >> 4) + lane; x <= (e2 - 1) >> 4; x += 64) rle_paint_word(row, mrow, win, winm, w_base, x, s2, e2, pat2, miss2);
			}
			cur = e64;
		}
		__syncthreads();
		// window -> row, and the window is zero again
		for (uint32_t i = tid; i < min(n_words, RLE_WIN); i += 256) {
			const bool shared_word = i == 0 || i == n_words - 1;       // also holds bits of the previous / next step
			const uint32_t x = win[i];
			if (x) { if (shared_word) atomicOr(row + w_base + i, x); else row[w_base + i] = x; win[i] = 0; }
			if (mrow) {
				const uint32_t y = winm[i];
				if (y) { if (shared_word) atomicOr(mrow + w_base + i, y); else mrow[w_base + i] = y; winm[i] = 0; }
			}
		}
		__syncthreads();
		pos += step_total;
		cur_regs = next_regs;
	}
	if (last_chunk && pos != n_samples && tid == 0) status[0] = 1;
}

__global__ __launch_bounds__(256)
void k_inflate_rle(const uint8_t* __restrict__ bytes, const RleDesc* __restrict__ desc, const uint32_t* __restrict__ chunk_base,
                   uint32_t count, const unsigned long long* __restrict__ chunk_sum,
                   uint32_t* __restrict__ raw, uint32_t* __restrict__ rawmask,
                   uint32_t Wp, uint32_t n_samples, uint32_t first_row, int* __restrict__ status) {
	__shared__ unsigned long long wave_tot[4];
	__shared__ uint32_t win[RLE_WIN], winm[RLE_WIN];
	const uint32_t v = rle_variant_of_block(chunk_base, count, blockIdx.x);
	const uint32_t cb = chunk_base[v], chunk = blockIdx.x - cb, n_chunks = chunk_base[v + 1] - cb;
	const RleDesc d = desc[v];
	const uint32_t width = d.width_missing & 0xFFu, m = (d.width_missing >> 8) & 1u;
	const uint8_t* runs = bytes + d.off;
	uint32_t* row = raw + (size_t)(first_row + v) * Wp;
	uint32_t* mrow = rawmask ? rawmask + (size_t)(first_row + v) * Wp : nullptr;
	for (uint32_t i = threadIdx.x; i < RLE_WIN; i += 256) { win[i] = 0; winm[i] = 0; }
	unsigned long long before = 0;                                 // samples of the chunks before this one
	for (uint32_t k = threadIdx.x; k < chunk; k += 256) before += chunk_sum[cb + k];
	const unsigned long long pos = rle_block_sum(before, wave_tot);                 // (also the barrier after the window's zeroing)
	const bool last = chunk + 1 == n_chunks;
	if (d.n_runs == 0) { if (n_samples != 0 && threadIdx.x == 0) status[0] = 1; return; }
	if (width == 2) rle_inflate_chunk<2>(runs, d.n_runs, m, chunk, pos, last, row, mrow, n_samples, status, wave_tot, win, winm);
	else if (width == 1) rle_inflate_chunk<1>(runs, d.n_runs, m, chunk, pos, last, row, mrow, n_samples, status, wave_tot, win, winm);
	else rle_inflate_chunk<4>(runs, d.n_runs, m, chunk, pos, last, row, mrow, n_samples, status, wave_tot, win, winm);
}

}  // namespace twk
