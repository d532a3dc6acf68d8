// Does a 64-bit shift whose amount sits in the last VGPR a wave owns give wrong results on this GPU?  (The "shift64 high
// register" erratum LLVM works around for gfx90a - GCNHazardRecognizer::fixShift64HighRegBug - and not for gfx950)
// Three kernels do the same thing - out = value << amount, per lane, with v_lshlrev_b64 - and differ only in where the amount
// lives and how many VGPRs the kernel owns (inline asm with fixed registers; the clobber list sets the allocation):
//   A: amount in v31, kernel owns 32 VGPRs  -> the amount is in the last owned register
//   B: amount in v30, kernel owns 32 VGPRs
//   C: amount in v31, kernel owns 40 VGPRs
// Every lane's result is checked against the host's; many waves are resident at once and fill their registers with different
// values (a neighbouring wave's registers are what a read beyond the allocation would see).
//   hipcc --offload-arch=gfx950 -O2 shift64_probe.hip -o shift64_probe && ./shift64_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define BODY(AMT)                                                                                                  \
	unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32), olo, ohi;                                                   \
	asm volatile("v_mov_b32 " AMT ", %4\n\tv_mov_b32 v28, %2\n\tv_mov_b32 v29, %3\n\ts_nop 4\n\t"                      \
	             "v_lshlrev_b64 v[28:29], " AMT ", v[28:29]\n\ts_nop 4\n\tv_mov_b32 %0, v28\n\tv_mov_b32 %1, v29"      \
	             : "=v"(olo), "=v"(ohi) : "v"(lo), "v"(hi), "v"(a)

__global__ void k_a(const uint64_t* val, const uint32_t* amt, uint64_t* out, int rounds) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t v = val[i]; const uint32_t a = amt[i]; uint64_t acc = 0;
	for (int r = 0; r < rounds; ++r) { BODY("v31") : "v28", "v29", "v31"); acc ^= ((uint64_t)ohi << 32 | olo) + r; v = v * 6364136223846793005ull + 1442695040888963407ull; }
	out[i] = acc;
}
__global__ void k_b(const uint64_t* val, const uint32_t* amt, uint64_t* out, int rounds) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t v = val[i]; const uint32_t a = amt[i]; uint64_t acc = 0;
	for (int r = 0; r < rounds; ++r) { BODY("v30") : "v28", "v29", "v30", "v31"); acc ^= ((uint64_t)ohi << 32 | olo) + r; v = v * 6364136223846793005ull + 1442695040888963407ull; }
	out[i] = acc;
}
__global__ void k_c(const uint64_t* val, const uint32_t* amt, uint64_t* out, int rounds) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t v = val[i]; const uint32_t a = amt[i]; uint64_t acc = 0;
	for (int r = 0; r < rounds; ++r) { BODY("v31") : "v28", "v29", "v31", "v39"); acc ^= ((uint64_t)ohi << 32 | olo) + r; v = v * 6364136223846793005ull + 1442695040888963407ull; }
	out[i] = acc;
}
// a kernel that only occupies registers: 64 VGPRs per lane full of a pattern, resident next to the ones under test
__global__ void k_noise(uint32_t* sink, int spin) {
	uint32_t r[48];
	for (int k = 0; k < 48; ++k) r[k] = 0xFFFFFFC0u | (threadIdx.x + k);
	for (int s = 0; s < spin; ++s) for (int k = 0; k < 48; ++k) r[k] = r[k] * 1664525u + r[(k + 7) % 48];
	uint32_t x = 0; for (int k = 0; k < 48; ++k) x ^= r[k];
	if (x == 12345u) sink[0] = x;
}

int main() {
	const size_t n = 256 * 256 * 16;
	const int rounds = 64;
	std::vector<uint64_t> val(n), want(n), got(n); std::vector<uint32_t> amt(n);
	uint64_t s = 88172645463325252ull;
	for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; val[i] = s; amt[i] = (uint32_t)(s >> 40) & 63u; }
	for (size_t i = 0; i < n; ++i) {
		uint64_t v = val[i], acc = 0;
		for (int r = 0; r < rounds; ++r) { acc ^= (v << amt[i]) + r; v = v * 6364136223846793005ull + 1442695040888963407ull; }
		want[i] = acc;
	}
	uint64_t *dv, *dout; uint32_t *da, *dsink;
	hipMalloc(&dv, n * 8); hipMalloc(&dout, n * 8); hipMalloc(&da, n * 4); hipMalloc(&dsink, 64);
	hipMemcpy(dv, val.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(da, amt.data(), n * 4, hipMemcpyHostToDevice);
	hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
	const char* names[3] = {"A: amount in v31, 32 VGPRs owned (last owned register)", "B: amount in v30, 32 VGPRs owned", "C: amount in v31, 40 VGPRs owned"};
	for (int pass = 0; pass < 3; ++pass)
		for (int which = 0; which < 3; ++which) {
			hipMemset(dout, 0, n * 8);
			hipLaunchKernelGGL(k_noise, dim3(4096), dim3(256), 0, s2, dsink, 40);
			if (which == 0) hipLaunchKernelGGL(k_a, dim3(n / 256), dim3(256), 0, s1, dv, da, dout, rounds);
			if (which == 1) hipLaunchKernelGGL(k_b, dim3(n / 256), dim3(256), 0, s1, dv, da, dout, rounds);
			if (which == 2) hipLaunchKernelGGL(k_c, dim3(n / 256), dim3(256), 0, s1, dv, da, dout, rounds);
			hipDeviceSynchronize();
			hipMemcpy(got.data(), dout, n * 8, hipMemcpyDeviceToHost);
			size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += got[i] != want[i];
			printf("pass %d  %-58s %zu of %zu lanes wrong\n", pass, names[which], bad, n);
		}
	return 0;
}
