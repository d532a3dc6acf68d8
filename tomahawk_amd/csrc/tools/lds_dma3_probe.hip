// Dev tool: where does global_load_lds_dwordx3 put a wave's 64 x 12 bytes?  (The count kernels stage with dwordx4: lane-linear, 16 bytes
// per lane.  A 96-byte row - 24 words of each of three planes - would be eight 12-byte pieces.)  Source word i holds i; every lane
// loads the three words at 12 * lane; the first 256 words of LDS are printed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint32_t* g, uint32_t* out) {
	__shared__ uint32_t lds[1024];
	for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xDEADu;
	__syncthreads();
	uint32_t keep; const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
	const uint32_t voff = threadIdx.x * 12;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(g), "s"(base) : "memory");
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
	std::vector<uint32_t> h(4096); for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)i;
	uint32_t *d, *o; hipMalloc(&d, h.size() * 4); hipMalloc(&o, 4096); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
	std::vector<uint32_t> r(1024); hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
	int linear = 1; for (int i = 0; i < 192; ++i) if (r[i] != (uint32_t)i) linear = 0;
	printf("lane-linear 12-byte pieces (LDS word i = source word i for i < 192): %s\n", linear ? "yes" : "NO");
	for (int i = 0; i < 256; ++i) printf("%s%5x", (i % 16) ? " " : "\n", r[i]);
	printf("\n");
	return 0;
}
