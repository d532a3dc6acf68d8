// K1's rare x common path, measured before built (reference lib/ld/ld_engine.cpp:230-242: PhasedListVector walks the *shorter*
// carrier list and tests the partner's bitvector, O(min carriers) per pair; include/core.h:601-641 twk_igt_list).  What does a
// pair (rare variant with AC carriers) x (common variant as a bitvector row) cost on the device as AC bit probes into the row,
// against the dense AND+popcount contraction of the same pair?  Same variants both ways, every count compared.
//   probe/row   k_probe_rows   a wave = 64 rare variants against ONE common row: lane l walks rare variant i0 + l's carrier list and
//                              reads the row's word h / 32 for every carrier h (the row - 250 KB at 2N = 2 M - stays hot in L2 while
//                              the blocks of a grid row work through the rare variants)
//   probe/pair  k_probe_pairs  a wave = one rare variant against 64 consecutive common rows: lane l reads word h / 32 of row j0 + l
//                              (64 rows touched per carrier: no reuse between lanes)
//   dense       k_count_list_t the production kernel over the rare x common rectangle
// usage: probe_vs_dense <haplotypes 2N> <carriers per rare variant AC> [rare variants = 4096] [common variants = 4096] [reps = 3]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../hip/ld_count.hip.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP %s @%d: %s\n",#x,__LINE__,hipGetErrorString(e)); exit(1);} }while(0)

// out[i * Mc + j] = number of carriers of rare variant i whose bit is set in common row j
__global__ __launch_bounds__(256)
void k_probe_rows(const uint32_t* __restrict__ lists, uint32_t stride, uint32_t ac, uint32_t Mr, const uint32_t* __restrict__ rows, uint32_t W,
                  uint32_t Mc, uint32_t* __restrict__ out) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
	if (i >= Mr) return;
	const uint32_t* a = lists + (size_t)i * stride;
	const uint32_t* row = rows + (size_t)j * W;
	uint32_t n = 0;
	for (uint32_t k = 0; k < ac; ++k) { const uint32_t h = a[k]; n += (row[h >> 5] >> (h & 31)) & 1u; }
	out[(size_t)i * Mc + j] = n;
}
__global__ __launch_bounds__(256)
void k_probe_pairs(const uint32_t* __restrict__ lists, uint32_t stride, uint32_t ac, const uint32_t* __restrict__ rows, uint32_t W,
                   uint32_t Mc, uint32_t* __restrict__ out) {
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
	if (j >= Mc) return;
	const uint32_t* a = lists + (size_t)i * stride;
	const uint32_t* row = rows + (size_t)j * W;
	uint32_t n = 0;
	for (uint32_t k = 0; k < ac; ++k) { const uint32_t h = a[k]; n += (row[h >> 5] >> (h & 31)) & 1u; }      // (a[k]: the same address in every lane - a broadcast)
	out[(size_t)i * Mc + j] = n;
}

int main(int argc, char** argv) {
	if (argc < 3) { fprintf(stderr, "usage: probe_vs_dense <haplotypes> <carriers> [rare=4096] [common=4096] [reps=3]\n"); return 2; }
	const uint64_t H = strtoull(argv[1], nullptr, 10);
	const uint32_t AC = (uint32_t)atoi(argv[2]);
	const uint32_t Mr = argc > 3 ? (uint32_t)atoi(argv[3]) / 128 * 128 : 4096, Mc = argc > 4 ? (uint32_t)atoi(argv[4]) / 128 * 128 : 4096;
	const int reps = argc > 5 ? atoi(argv[5]) : 3;
	if (AC == 0 || AC > H || Mr < 128 || Mc < 128) { fprintf(stderr, "bad arguments\n"); return 2; }
	const uint32_t W = (uint32_t)((H + 31) / 32 + 31) / 32 * 32;
	const uint32_t stride = AC + 1, M = Mr + Mc;
	std::mt19937_64 rng(H * 7919ull + AC);
	std::vector<uint32_t> lists((size_t)Mr * stride), rows((size_t)M * W, 0);
	for (uint32_t v = 0; v < Mr; ++v) {         // rare variants: AC distinct haplotypes, sorted
		std::vector<uint32_t> c;
		while (c.size() < AC) { c.push_back((uint32_t)(rng() % H)); if (c.size() == AC) { std::sort(c.begin(), c.end()); c.erase(std::unique(c.begin(), c.end()), c.end()); } }
		for (uint32_t k = 0; k < AC; ++k) { lists[(size_t)v * stride + k] = c[k]; rows[(size_t)v * W + c[k] / 32] |= 1u << (c[k] % 32); }
		lists[(size_t)v * stride + AC] = 0xFFFFFFFFu;
	}
	for (uint32_t v = Mr; v < M; ++v) {         // common variants: ALT frequency 50, 25 or 12.5 % (random words, ANDed once or twice)
		const int ands = (int)(v % 3);
		for (uint64_t k = 0; k < (H + 31) / 32; ++k) {
			uint64_t r = rng(); uint32_t x = (uint32_t)r;
			if (ands >= 1) x &= (uint32_t)(r >> 32);
			if (ands >= 2) x &= (uint32_t)rng();
			if (k == (H - 1) / 32 && H % 32) x &= (1u << (H % 32)) - 1u;
			rows[(size_t)v * W + k] = x;
		}
	}
	uint32_t *d_lists, *d_rows, *d_out, *d_out2, *d_C, *d_tiles, *tick; twk::CountUnit* d_units;
	CK(hipMalloc(&d_lists, lists.size() * 4)); CK(hipMalloc(&d_rows, rows.size() * 4));
	CK(hipMalloc(&d_out, (size_t)Mr * Mc * 4)); CK(hipMalloc(&d_out2, (size_t)Mr * Mc * 4)); CK(hipMalloc(&d_C, (size_t)Mr * Mc * 4)); CK(hipMalloc(&tick, 32));
	CK(hipMemcpy(d_lists, lists.data(), lists.size() * 4, hipMemcpyHostToDevice));
	CK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
	std::vector<uint32_t> tl;
	for (uint32_t y = 0; y < Mr / 128; ++y) for (uint32_t x = 0; x < Mc / 128; ++x) tl.push_back(y << 16 | x);
	std::vector<twk::CountUnit> units;
	const uint32_t first_split = twk::build_count_units((uint32_t)tl.size(), W / twk::KC, 512, 8, units);
	twk::fill_unit_tiles(units, tl.data());
	CK(hipMalloc(&d_tiles, tl.size() * 4)); CK(hipMalloc(&d_units, units.size() * sizeof(twk::CountUnit)));
	CK(hipMemcpy(d_tiles, tl.data(), tl.size() * 4, hipMemcpyHostToDevice));
	CK(hipMemcpy(d_units, units.data(), units.size() * sizeof(twk::CountUnit), hipMemcpyHostToDevice));
	twk::CountWork w{}; w.rows = d_rows; w.W = W; w.rowA0 = 0; w.rowB0 = Mr; w.tiles = d_tiles; w.units = d_units; w.n_units = (uint32_t)units.size(); w.C = d_C; w.ldc = Mc;
	w.ticket = tick; w.n_queues = 1; w.queue_begin[0] = 0; w.queue_begin[1] = w.n_units;
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto time_it = [&](auto&& launch) { float best = 1e30f; for (int r = 0; r <= reps; ++r) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms; } return best; };
	const uint32_t* d_common = d_rows + (size_t)Mr * W;
	const float ms_rows = time_it([&] { hipLaunchKernelGGL(k_probe_rows, dim3((Mr + 255) / 256, Mc), dim3(256), 0, 0, d_lists, stride, AC, Mr, d_common, W, Mc, d_out); });
	const float ms_pairs = time_it([&] { hipLaunchKernelGGL(k_probe_pairs, dim3((Mc + 255) / 256, Mr), dim3(256), 0, 0, d_lists, stride, AC, d_common, W, Mc, d_out2); });
	const float ms_dense = time_it([&] {
		CK(hipMemsetAsync(tick, 0, 32, 0));
		if (first_split < tl.size()) hipLaunchKernelGGL(twk::k_zero_tiles, dim3((uint32_t)tl.size() - first_split), dim3(256), 0, 0, w.tiles, first_split, d_C, Mc);
		hipLaunchKernelGGL((twk::k_count_list_t<twk::COUNT_NW>), dim3(std::min<uint32_t>(512, w.n_units)), dim3(twk::COUNT_THREADS), 0, 0, w);
	});
	CK(hipDeviceSynchronize());
	std::vector<uint32_t> a((size_t)Mr * Mc), a2((size_t)Mr * Mc), b((size_t)Mr * Mc);
	CK(hipMemcpy(a.data(), d_out, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(a2.data(), d_out2, a2.size() * 4, hipMemcpyDeviceToHost));
	CK(hipMemcpy(b.data(), d_C, b.size() * 4, hipMemcpyDeviceToHost));
	size_t bad = 0; uint64_t sum = 0;
	for (size_t k = 0; k < a.size(); ++k) { bad += a[k] != b[k]; bad += a2[k] != b[k]; sum += b[k]; }
	const double pairs = (double)Mr * Mc;
	printf("2N=%llu AC=%u rare=%u common=%u (W=%u words/row): probe, 64 rare x one row per wave %9.3f ms = %8.1f ps/pair (%.2e probes/s) | probe, one rare x 64 rows per wave %9.3f ms = %8.1f ps/pair "
	       "(%.2e probes/s) | dense %9.3f ms = %8.1f ps/pair (%4.1f %% of the and+bcnt ceiling) | best probe / dense %7.3f | mismatches %zu, mean ALTALT %.2f\n",
	       (unsigned long long)H, AC, Mr, Mc, W, ms_rows, ms_rows * 1e9 / pairs, pairs * AC / (ms_rows * 1e-3), ms_pairs, ms_pairs * 1e9 / pairs, pairs * AC / (ms_pairs * 1e-3),
	       ms_dense, ms_dense * 1e9 / pairs, pairs * W / (ms_dense * 1e-3) / 2.6214e13 * 100, std::min(ms_rows, ms_pairs) / ms_dense, bad, (double)sum / pairs);
	return bad != 0;
}
