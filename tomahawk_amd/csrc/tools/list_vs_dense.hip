// T2 evidence (reference include/core.h:517-672 twk_igt_list, lib/ld/ld_engine.cpp:185-267 PhasedListVector):
// what a rare pair costs as an intersection of two sorted carrier lists against what it costs in the dense
// AND+popcount contraction, on the same variants, self-checked (both must give the same ALTALT count for every pair).
//   list   k_list_pairs   one pair per lane, branch-free merge of two sorted uint32 lists (haplotype ids); a wave holds
//                         64 consecutive partners of one variant
//   dense  k_count_list_t the production kernel over the same M x M rectangle of bitvector rows
// usage: list_vs_dense <haplotypes 2N> <carriers per variant AC> [variants M = 1024] [reps = 3]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../hip/ld_count.hip.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP %s @%d: %s\n",#x,__LINE__,hipGetErrorString(e)); exit(1);} }while(0)

// lists[v * stride + k], k < ac sorted ascending, lists[v * stride + ac] = 0xFFFFFFFF (sentinel).
// out[i * M + j] = |list_i & list_j| for j > i (the reference's loop order), lanes along j.
__global__ __launch_bounds__(256)
void k_list_pairs(const uint32_t* __restrict__ lists, uint32_t stride, uint32_t ac, uint32_t M, uint32_t* __restrict__ out) {
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
	if (j >= M || j <= i) return;
	const uint32_t* a = lists + (size_t)i * stride;
	const uint32_t* b = lists + (size_t)j * stride;
	uint32_t ia = 0, ib = 0, n = 0, va = a[0], vb = b[0];
	while (ia < ac && ib < ac) {          // one step of the merge: count a match, advance the smaller head (both on a match)
		n += va == vb;
		const bool fa = va <= vb, fb = vb <= va;
		ia += fa; ib += fb;
		if (fa) va = a[ia];
		if (fb) vb = b[ib];
	}
	out[(size_t)i * M + j] = n;
}

int main(int argc, char** argv) {
	if (argc < 3) { fprintf(stderr, "usage: list_vs_dense <haplotypes> <carriers> [variants=1024] [reps=3]\n"); return 2; }
	const uint64_t H = strtoull(argv[1], nullptr, 10);
	const uint32_t AC = (uint32_t)atoi(argv[2]);
	const uint32_t M = argc > 3 ? (uint32_t)atoi(argv[3]) / 128 * 128 : 1024;
	const int reps = argc > 4 ? atoi(argv[4]) : 3;
	if (AC == 0 || AC > H || M < 128) { fprintf(stderr, "bad arguments\n"); return 2; }
	const uint32_t W = (uint32_t)((H + 31) / 32 + 31) / 32 * 32;          // words per row, padded to the K chunk
	const uint32_t stride = AC + 1;
	std::mt19937_64 rng(H * 1000003ull + AC);
	std::vector<uint32_t> lists((size_t)M * stride), rows((size_t)M * W, 0);
	// carriers cluster a little (a third of each list is copied from the previous variant) so that counts are not all ~0
	std::vector<uint32_t> cur;
	for (uint32_t v = 0; v < M; ++v) {
		std::vector<uint32_t> next;
		if (v) for (uint32_t k = 0; k < AC / 3; ++k) next.push_back(cur[rng() % cur.size()]);
		while (true) {
			std::sort(next.begin(), next.end()); next.erase(std::unique(next.begin(), next.end()), next.end());
			if (next.size() >= AC) break;
			const size_t need = AC - next.size();
			for (size_t k = 0; k < need; ++k) next.push_back((uint32_t)(rng() % H));
		}
		next.resize(AC);
		cur = next;
		for (uint32_t k = 0; k < AC; ++k) { lists[(size_t)v * stride + k] = cur[k]; rows[(size_t)v * W + cur[k] / 32] |= 1u << (cur[k] % 32); }
		lists[(size_t)v * stride + AC] = 0xFFFFFFFFu;
	}
	uint32_t *d_lists, *d_rows, *d_out, *d_C, *d_tiles, *tick; twk::CountUnit* d_units;
	CK(hipMalloc(&d_lists, lists.size() * 4)); CK(hipMalloc(&d_rows, rows.size() * 4));
	CK(hipMalloc(&d_out, (size_t)M * M * 4)); CK(hipMalloc(&d_C, (size_t)M * M * 4)); CK(hipMalloc(&tick, 32));
	CK(hipMemcpy(d_lists, lists.data(), lists.size() * 4, hipMemcpyHostToDevice));
	CK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
	CK(hipMemset(d_out, 0, (size_t)M * M * 4)); CK(hipMemset(d_C, 0, (size_t)M * M * 4));
	// dense: the tiles on and above the diagonal, whole-K units, 512 persistent blocks (as the engine launches it)
	std::vector<uint32_t> tl; const uint32_t g = M / 128;
	for (uint32_t y = 0; y < g; ++y) for (uint32_t x = y; x < g; ++x) tl.push_back(y << 16 | x);
	std::vector<twk::CountUnit> units;
	const uint32_t first_split = twk::build_count_units((uint32_t)tl.size(), W / twk::KC, 512, 8, units);
	twk::fill_unit_tiles(units, tl.data());
	CK(hipMalloc(&d_tiles, tl.size() * 4)); CK(hipMalloc(&d_units, units.size() * sizeof(twk::CountUnit)));
	CK(hipMemcpy(d_tiles, tl.data(), tl.size() * 4, hipMemcpyHostToDevice));
	CK(hipMemcpy(d_units, units.data(), units.size() * sizeof(twk::CountUnit), hipMemcpyHostToDevice));
	twk::CountWork w{}; w.rows = d_rows; w.W = W; w.tiles = d_tiles; w.units = d_units; w.n_units = (uint32_t)units.size(); w.C = d_C; w.ldc = M;
	w.ticket = tick; w.n_queues = 1; w.queue_begin[0] = 0; w.queue_begin[1] = w.n_units;
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto time_it = [&](auto&& launch) { float best = 1e30f; for (int r = 0; r <= reps; ++r) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms; } return best; };
	const float ms_list = time_it([&] { hipLaunchKernelGGL(k_list_pairs, dim3((M + 255) / 256, M), dim3(256), 0, 0, d_lists, stride, AC, M, d_out); });
	const float ms_dense = time_it([&] {
		CK(hipMemsetAsync(tick, 0, 32, 0));
		if (first_split < tl.size()) hipLaunchKernelGGL(twk::k_zero_tiles, dim3((uint32_t)tl.size() - first_split), dim3(256), 0, 0, w.tiles, first_split, d_C, M);
		hipLaunchKernelGGL((twk::k_count_list_t<twk::COUNT_NW>), dim3(std::min<uint32_t>(512, w.n_units)), dim3(twk::COUNT_THREADS), 0, 0, w);
	});
	CK(hipDeviceSynchronize());
	std::vector<uint32_t> a((size_t)M * M), b((size_t)M * M);
	CK(hipMemcpy(a.data(), d_out, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_C, b.size() * 4, hipMemcpyDeviceToHost));
	size_t bad = 0; uint64_t sum = 0;
	for (uint32_t i = 0; i < M; ++i) for (uint32_t j = i + 1; j < M; ++j) { bad += a[(size_t)i * M + j] != b[(size_t)i * M + j]; sum += a[(size_t)i * M + j]; }
	const double pairs = (double)M * (M - 1) / 2, dense_pairs = (double)tl.size() * 128 * 128;
	printf("2N=%llu AC=%u M=%u (W=%u words/row): list merge %9.3f ms = %9.1f ps/pair (%.2e merge steps/s) | dense %9.3f ms = %9.1f ps/pair (%4.1f %% of the and+bcnt ceiling) | list/dense %7.3f | mismatches %zu, mean ALTALT %.2f\n",
	       (unsigned long long)H, AC, M, W, ms_list, ms_list * 1e9 / pairs, pairs * 2.0 * AC / (ms_list * 1e-3), ms_dense, ms_dense * 1e9 / dense_pairs,
	       dense_pairs * W / (ms_dense * 1e-3) / 2.6214e13 * 100, (ms_list / pairs) / (ms_dense / dense_pairs), bad, (double)sum / pairs);
	return bad != 0;
}
