// Development tool: device read bandwidth of the box ("confirm the 8 TB/s datasheet
// peak with a read microbenchmark and report both").  Every thread streams uint4 loads over a
// buffer much larger than the 256 MB of L2/MALL and folds them into one word so that nothing is elided.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ p, size_t n, uint32_t* __restrict__ sink) {
	uint32_t acc = 0;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	for (; i + 3 * stride < n; i += 4 * stride) {
		const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
		acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
	}
	for (; i < n; i += stride) { const uint4 a = p[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
	if (acc == 0x12345678u) *sink = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
	const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 16;
	const size_t bytes = gib << 30, n = bytes / sizeof(uint4);
	uint4* buf = nullptr; uint32_t* sink = nullptr;
	CK(hipMalloc((void**)&buf, bytes)); CK(hipMalloc((void**)&sink, 4));
	CK(hipMemset(buf, 1, bytes));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int blocks_per_cu : {4, 8, 16, 32}) {
		const int grid = 256 * blocks_per_cu;
		hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, buf, n, sink);   // warm-up
		CK(hipDeviceSynchronize());
		float best = 1e30f;
		for (int rep = 0; rep < 5; ++rep) {
			CK(hipEventRecord(e0));
			hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, buf, n, sink);
			CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
			float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
			if (ms < best) best = ms;
		}
		printf("read %zu GiB, %d blocks/CU: %.3f ms -> %.1f GB/s\n", gib, blocks_per_cu, best, bytes / (best * 1e-3) / 1e9);
		fflush(stdout);
	}
	return 0;
}
