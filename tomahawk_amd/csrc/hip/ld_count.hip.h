// Count kernel: the AND+popcount contraction over the sample axis.
//
//   C[r][c] = sum_k popc( rows[rowA0 + r][k] & rows[rowB0 + c][k] )
//
// `rows` is a row-major matrix of bit-planes (uint32 words, row pitch W words,
// W a multiple of KC, zero padded).  One row is one plane of one variant; what
// the planes mean (haplotype bits, het / hom-alt indicator, missing mask) is
// the caller's business (ld_prep.hip.h builds them, ld_math.hip.h turns the
// plane products back into the reference's contingency cells).  This replaces
// the word loops of the reference kernels PhasedListVector / PhasedVectorized /
// UnphasedVectorized(NoMissing) (lib/ld/ld_engine.cpp:230-242, 583-585,
// 668, 941-943): same AND, same popcount, but as a tiled contraction instead
// of one pair at a time.
//
// CDNA4 mapping (gfx950, wave64, 4 x SIMD32 per CU) - the production kernel is k_count_list_t below:
//   * block = 512 threads = 8 waves as a 2 x 4 wave grid over a 128 x 128 tile of row pairs; lane (li,lj)
//     of the 8 x 8 lane grid owns the 8 x 4 pairs { wr*64 + li + 8t } x { wc*32 + lj + 8u }: 32 u32
//     accumulators; <= 128 VGPRs, 2 blocks/CU = 4 waves/SIMD.
//   * K is walked in chunks of KC = 32 words (128 B = one cache line per row).  A chunk of the A tile and
//     of the B tile (2 x 16 KiB) is DMA'd HBM -> LDS with global_load_lds_dwordx4 (no VGPR staging), double
//     buffered: the loads of chunk c+1 are in flight while chunk c is contracted; one barrier per chunk.
//   * LDS image is lane-linear per wave-instruction (8 rows x 128 B), so the bank swizzle lives on the
//     *source* address: 16-byte slot q of row r is stored at slot q ^ ((r >> 1) & 7).  The LDS reads of a
//     lane group then see 8 consecutive rows at 8 distinct (row parity, slot) bank positions: conflict free.
//   * the loop is VALU-issue bound by construction: one v_and_b32 + one v_bcnt_u32_b32 per 32-bit word pair
//     (2 + 4 cycles per wave64 on a SIMD), LDS pipe at ~25 %.
//   * no MFMA: this is integer popcount.
// k_count_tile_t (one block per tile on a 2-D grid) is the round-1 kernel, kept as the baseline of the
// dev tool csrc/tools/count_microbench.hip; the engine launches k_count_list_t only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace twk {

constexpr int TILE = 128;       // rows per block tile edge
constexpr int KC   = 32;        // 32-bit words of K per chunk (128 B per row)
constexpr int LDS_TILE_BYTES = TILE * KC * 4;   // 16 KiB: one operand, one chunk

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// One LDS-DMA: 64 lanes x 16 B, HBM -> LDS at (wave-uniform lds_byte + 16*lane).
// Issued through asm so that hipcc does not see it: with the builtin it drains
// vmcnt(0) in front of the next ds_read (it cannot prove the DMA does not alias
// the reads), which serialises staging with the contraction.  We count it
// ourselves: one `s_waitcnt vmcnt(0)` per chunk, right before the barrier
// (cdna_hip_programming.md 5.7: M0 is written in the same statement).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_byte) {
	uint32_t keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
	             "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}

// HBM -> LDS for one chunk of one 128-row operand tile: 16 wave-instructions
// of 1 KiB; this wave issues `n` of them starting at segment `seg0`.
// lds_tile_byte: LDS byte address of the operand tile (wave-uniform).
__device__ __forceinline__ void stage_rows(const uint32_t* __restrict__ rows, size_t W,
                                           uint32_t row0, uint32_t chunk, uint32_t lds_tile_byte,
                                           int seg0, int n, int lane) {
	const int lr = lane >> 3;          // row within the 8-row segment
	const int ls = lane & 7;           // 16-byte slot written by this lane
#pragma unroll
	for (int i = 0; i < n; ++i) {
		const int seg = seg0 + i;
		const int r = seg * 8 + lr;                    // row within the tile
		const int src_slot = ls ^ ((r >> 1) & 7);      // swizzle on the source
		const uint32_t* g = rows + (size_t)(row0 + r) * W + (size_t)chunk * KC + src_slot * 4;
		glds16(g, lds_tile_byte + seg * 8 * KC * 4);
	}
}

// Eight independent (AND, popcount-accumulate) pairs against one B word:
//     acc[t] += popc(a[t] & b),  t = 0..7
// v_bcnt_u32_b32 D = popc(S0) + S1 folds the accumulate into the popcount, so
// a word pair costs exactly one v_and_b32 (full rate) + one v_bcnt_u32_b32
// (half rate on gfx950: measured 3.7e13 vs 7.2e13 lane-ops/s).  Left to itself
// hipcc (a) reassociates the adds into bcnt(x,0) + v_add3 (+25 % VALU) and
// (b) serialises everything through one temporary, so every instruction waits
// for the previous one.  The asm block pins the accumulate form and the issue
// order (see the comment inside).
__device__ __forceinline__ void and_bcnt8(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t& c4,
                                          uint32_t& c5, uint32_t& c6, uint32_t& c7, uint32_t a0, uint32_t a1,
                                          uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5,
                                          uint32_t a6, uint32_t a7, uint32_t b) {
	uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
	// gfx950 issue quirk (csrc/tools/issue_test2.hip): a v_bcnt_u32_b32 issued directly
	// behind another VALU op of the same wave costs 6 cycles instead of 4 (and+bcnt streams run
	// at 1.84e13 word pairs/s); with a non-VALU instruction in the slot before it the pair costs
	// the ideal 2 + 4 cycles (2.52e13).  The s_nop itself is free: the SIMD issues another
	// wave's VALU op in that slot.  Order used: (AND, s_nop, BCNT) x 8 -- pattern I of the tool,
	// the fastest of the nine orders tried there and in this kernel (2.48e13 here, 98 % of the
	// register-only stream).
	asm("v_and_b32 %8, %16, %24\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %0, %8, %0\n\t"
	    "v_and_b32 %9, %17, %24\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %1, %9, %1\n\t"
	    "v_and_b32 %10, %18, %24\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %2, %10, %2\n\t"
	    "v_and_b32 %11, %19, %24\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %3, %11, %3\n\t"
	    "v_and_b32 %12, %20, %24\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %4, %12, %4\n\t"
	    "v_and_b32 %13, %21, %24\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %5, %13, %5\n\t"
	    "v_and_b32 %14, %22, %24\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %6, %14, %6\n\t"
	    "v_and_b32 %15, %23, %24\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %7, %15, %7"
	    : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7),
	      "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
	    : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(b));
}

// Same, volatile (stays where it is written relative to the other volatile asm of the hand-scheduled
// loop) and with two alternating temporaries instead of eight: the loop is at the 128-VGPR limit of
// 4 waves/SIMD, and in-order issue makes the reuse safe (the bcnt that reads a temporary is issued
// before the next v_and that overwrites it).
__device__ __forceinline__ void and_bcnt8v(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t& c4,
                                           uint32_t& c5, uint32_t& c6, uint32_t& c7, uint32_t a0, uint32_t a1,
                                           uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5,
                                           uint32_t a6, uint32_t a7, uint32_t b) {
	uint32_t t0, t1;
	asm volatile("v_and_b32 %8, %10, %18\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %0, %8, %0\n\t"
	    "v_and_b32 %9, %11, %18\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %1, %9, %1\n\t"
	    "v_and_b32 %8, %12, %18\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %2, %8, %2\n\t"
	    "v_and_b32 %9, %13, %18\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %3, %9, %3\n\t"
	    "v_and_b32 %8, %14, %18\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %4, %8, %4\n\t"
	    "v_and_b32 %9, %15, %18\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %5, %9, %5\n\t"
	    "v_and_b32 %8, %16, %18\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %6, %8, %6\n\t"
	    "v_and_b32 %9, %17, %18\n\t"
	    "s_nop 0\n\t"
	    "v_bcnt_u32_b32 %7, %9, %7"
	    : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), "=&v"(t0), "=&v"(t1)
	    : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(b));
}

// acc[t][u] += popc(a[t] & b) over the four words of a 16-byte slot, t = 0..7.
template <int TB>
__device__ __forceinline__ void contract_slot(uint32_t (&acc)[8][TB], int u, const uint4 (&a)[8], const uint4& b) {
	and_bcnt8(acc[0][u], acc[1][u], acc[2][u], acc[3][u], acc[4][u], acc[5][u], acc[6][u], acc[7][u],
	          a[0].x, a[1].x, a[2].x, a[3].x, a[4].x, a[5].x, a[6].x, a[7].x, b.x);
	and_bcnt8(acc[0][u], acc[1][u], acc[2][u], acc[3][u], acc[4][u], acc[5][u], acc[6][u], acc[7][u],
	          a[0].y, a[1].y, a[2].y, a[3].y, a[4].y, a[5].y, a[6].y, a[7].y, b.y);
	and_bcnt8(acc[0][u], acc[1][u], acc[2][u], acc[3][u], acc[4][u], acc[5][u], acc[6][u], acc[7][u],
	          a[0].z, a[1].z, a[2].z, a[3].z, a[4].z, a[5].z, a[6].z, a[7].z, b.z);
	and_bcnt8(acc[0][u], acc[1][u], acc[2][u], acc[3][u], acc[4][u], acc[5][u], acc[6][u], acc[7][u],
	          a[0].w, a[1].w, a[2].w, a[3].w, a[4].w, a[5].w, a[6].w, a[7].w, b.w);
}

// ---- round-1 kernel (baseline of the micro-benchmark only) -----------------------------------
// grid: x = column tiles, y = row tiles of the super-tile.  C is the count
// matrix of the super-tile: C[(by*128 + r) * ldc + bx*128 + c].
// diag != 0: the super-tile sits on the diagonal (rowA0 == rowB0); tiles with
// bx < by are not needed and exit at once.
//
// NW waves per block share one 128 x 128 tile as a 2 x (NW/2) wave grid; a
// lane owns 8 x TB pairs, TB = 16 / NW * 2:
//   NW = 4: lane tile 8 x 8, ~210 VGPR, 2 blocks/CU = 2 waves/SIMD
//   NW = 8: lane tile 8 x 4, <=128 VGPR, 2 blocks/CU = 4 waves/SIMD
// A lone wave issues a VALU op only every other slot on gfx950 (measured: one
// wave/SIMD reaches half the and/bcnt rate of two), so stalls of one wave are
// only covered when >= 2 others are runnable: NW = 8 is the production shape.
template <int NW>
__global__ __launch_bounds__(NW * 64, NW / 2)
void k_count_tile_t(const uint32_t* __restrict__ rows, uint32_t W, uint32_t rowA0, uint32_t rowB0,
                    int diag, uint32_t* __restrict__ C, uint32_t ldc) {
	constexpr int WC = NW / 2;            // wave columns (2 wave rows)
	constexpr int TB = 16 / WC;           // B rows per lane: 8 (NW=4) or 4 (NW=8)
	constexpr int NSEG = 32 / NW;         // DMA instructions per wave per chunk
	__shared__ __attribute__((aligned(16))) uint32_t lds[2 * 2 * TILE * KC];   // [buf][A|B] 64 KiB

	const uint32_t bx = blockIdx.x, by = blockIdx.y;
	if (diag && bx < by) return;

	const int tid  = threadIdx.x;
	const int lane = tid & 63;
	const int wave = tid >> 6;
	const int wr = wave / WC, wc = wave % WC;
	const int li = lane >> 3, lj = lane & 7;

	const uint32_t tileA0 = rowA0 + by * TILE;
	const uint32_t tileB0 = rowB0 + bx * TILE;
	const uint32_t nchunks = W / KC;

	uint32_t acc[8][TB];
#pragma unroll
	for (int t = 0; t < 8; ++t)
#pragma unroll
		for (int u = 0; u < TB; ++u) acc[t][u] = 0;

	// Per-lane LDS byte offsets.  Row (base + l + 8t), base a multiple of 16:
	// (row >> 1) & 7 = ((l >> 1) + 4t) & 7 = (l >> 1) ^ ((t & 1) << 2), so slot q
	// of that row lives at slot q ^ (l >> 1) ^ ((t & 1) << 2).  offX[k] = row
	// base + 16 * ((l >> 1) ^ k); the reads below pick k = q ^ ((t & 1) << 2)
	// and add the compile-time row stride 8*t*128.
	uint32_t offA[8], offB[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		offA[k] = (uint32_t)((wr * 64 + li) * (KC * 4) + (((li >> 1) ^ k) << 4));
		offB[k] = (uint32_t)(LDS_TILE_BYTES + (wc * 8 * TB + lj) * (KC * 4) + (((lj >> 1) ^ k) << 4));
	}
	const char* lds_b = reinterpret_cast<const char*>(lds);

	// Staging split: 32 wave-instructions per chunk (16 A + 16 B), NSEG per wave:
	// the first half of the waves stages A, the second half B.
	const int wave_u = __builtin_amdgcn_readfirstlane(wave);
	const bool st_isB = wave_u >= NW / 2;
	const uint32_t st_row0 = st_isB ? tileB0 : tileA0;
	const int st_seg0 = (wave_u % (NW / 2)) * NSEG;
	const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t*)lds;                    // LDS byte address
	const uint32_t st_lds = lds_base + (st_isB ? (uint32_t)LDS_TILE_BYTES : 0u);

	stage_rows(rows, W, st_row0, 0, st_lds, st_seg0, NSEG, lane);

	for (uint32_t c = 0; c < nchunks; ++c) {
		const int buf = c & 1;
		// Chunk c was issued one contraction ago: drain this wave's DMAs, then
		// the barrier makes every wave's part visible and proves every wave is
		// done reading the other buffer (chunk c-1).
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (c + 1 < nchunks)
			stage_rows(rows, W, st_row0, c + 1, st_lds + (buf ^ 1) * (2 * LDS_TILE_BYTES), st_seg0, NSEG, lane);

		const char* base = lds_b + buf * (2 * LDS_TILE_BYTES);
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			uint4 a[8];
#pragma unroll
			for (int t = 0; t < 8; ++t)
				a[t] = *reinterpret_cast<const uint4*>(base + offA[q ^ ((t & 1) << 2)] + t * 8 * (KC * 4));
#pragma unroll
			for (int u = 0; u < TB; ++u) {
				const uint4 b = *reinterpret_cast<const uint4*>(base + offB[q ^ ((u & 1) << 2)] + u * 8 * (KC * 4));
				contract_slot<TB>(acc, u, a, b);
			}
		}
	}

	// Epilogue: each lane stores its 8 x TB counts.  For fixed (t,u) the 8 lanes
	// of one li write 8 consecutive u32 (32 B); small next to the K loop.
	uint32_t* Cblk = C + (size_t)(by * TILE + wr * 64 + li) * ldc + bx * TILE + wc * 8 * TB + lj;
#pragma unroll
	for (int t = 0; t < 8; ++t)
#pragma unroll
		for (int u = 0; u < TB; ++u) Cblk[(size_t)(8 * t) * ldc + 8 * u] = acc[t][u];
}

// LDS-DMA with a scalar base: HBM (sbase + 32-bit per-lane byte offset voff) -> LDS (lds_byte + 16 * lane).
// The lane part of the address is the same for every segment of the same parity, so the list kernel
// keeps two offset VGPRs for all its DMAs and does the rest of the address arithmetic on the scalar unit.
__device__ __forceinline__ void glds16s(const void* sbase, uint32_t voff, uint32_t lds_byte) {
	uint32_t keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
	             "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}
// One dword per lane, HBM (per-lane address) -> LDS (lds_byte + 4 * lane).
__device__ __forceinline__ void glds4(const void* gsrc, uint32_t lds_byte) {
	uint32_t keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
	             "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}
// One chunk of one 128-row operand tile, this wave's `n` segments from `seg0` on.  voff_even / voff_odd:
// (lr * W + src_slot * 4) * 4 for even / odd segments, lr = lane >> 3, src_slot = (lane & 7) ^ (lr >> 1)
// [^ 4 for odd segments]: the source-side swizzle of stage_rows, (r >> 1) & 7 with r = seg * 8 + lr.
__device__ __forceinline__ void stage_rows_s(const uint32_t* __restrict__ rows, size_t W, uint32_t row0, uint32_t chunk,
                                             uint32_t lds_tile_byte, int seg0, int n, uint32_t voff_even, uint32_t voff_odd) {
	const uint32_t* base = rows + (size_t)row0 * W + (size_t)chunk * KC;
#pragma unroll
	for (int i = 0; i < n; ++i) {
		const int seg = seg0 + i;
		glds16s(base + (size_t)(seg * 8) * W, (seg & 1) ? voff_odd : voff_even, lds_tile_byte + seg * 8 * KC * 4);
	}
}

// One ds_read_b64 through asm (see k_count_list_t for why): 64 lanes x 8 B from LDS byte address
// `addr` + OFF.
template <int OFF>
__device__ __forceinline__ uint2 lds_read8(uint32_t addr) {
	uint2 v;
	asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
	return v;
}
// The 8 + TB reads of one half-slot.  addrA0 / addrA1: LDS byte address of this lane's A row for even / odd
// t (the slot swizzle differs by 4 between them, see the offA table); likewise addrB0 / addrB1; row t is
// 8 * t rows = t * 1024 bytes further on; HALF selects the upper 8 bytes of the 16-byte slot.
// PAIRED (the unphased fused kernel): a lane's rows are not li + 8t but 2 li + (t & 1) + 16 (t >> 1) - the H and the Q plane
// of the same four variants - so that all four products of a variant pair end up in one lane (ScreenCountsUnphased); the slot
// swizzle ((row >> 1) & 7 = li) is then the same for every t, and both addresses are the same.
template <bool PAIRED, int T>
constexpr int lane_row_offset() { return PAIRED ? ((T & 1) + 16 * (T >> 1)) * (KC * 4) : T * 8 * (KC * 4); }
template <int TB, bool PAIRED, int T = 0>
__device__ __forceinline__ void read_half_a(uint2 (&ra)[8], uint32_t addrA0, uint32_t addrA1, int half) {
	if constexpr (T < 8) {
		ra[T] = half ? lds_read8<lane_row_offset<PAIRED, T>() + 8>((T & 1) ? addrA1 : addrA0) : lds_read8<lane_row_offset<PAIRED, T>()>((T & 1) ? addrA1 : addrA0);
		read_half_a<TB, PAIRED, T + 1>(ra, addrA0, addrA1, half);
	}
}
template <int TB, bool PAIRED, int U = 0>
__device__ __forceinline__ void read_half_b(uint2 (&rb)[TB], uint32_t addrB0, uint32_t addrB1, int half) {
	if constexpr (U < TB) {
		rb[U] = half ? lds_read8<lane_row_offset<PAIRED, U>() + 8>((U & 1) ? addrB1 : addrB0) : lds_read8<lane_row_offset<PAIRED, U>()>((U & 1) ? addrB1 : addrB0);
		read_half_b<TB, PAIRED, U + 1>(rb, addrB0, addrB1, half);
	}
}
template <int TB, bool PAIRED = false>
__device__ __forceinline__ void read_half(uint2 (&ra)[8], uint2 (&rb)[TB], uint32_t addrA0, uint32_t addrA1,
                                          uint32_t addrB0, uint32_t addrB1, int half) {
	read_half_a<TB, PAIRED>(ra, addrA0, addrA1, half);
	read_half_b<TB, PAIRED>(rb, addrB0, addrB1, half);
}
// acc[t][u] += popc(a[t] & b) over the two words of a half-slot, t = 0..7 (volatile: keeps its place
// between the hand-placed LDS reads and waits).
template <int TB>
__device__ __forceinline__ void contract_half(uint32_t (&acc)[8][TB], int u, const uint2 (&a)[8], const uint2& b) {
	and_bcnt8v(acc[0][u], acc[1][u], acc[2][u], acc[3][u], acc[4][u], acc[5][u], acc[6][u], acc[7][u],
	           a[0].x, a[1].x, a[2].x, a[3].x, a[4].x, a[5].x, a[6].x, a[7].x, b.x);
	and_bcnt8v(acc[0][u], acc[1][u], acc[2][u], acc[3][u], acc[4][u], acc[5][u], acc[6][u], acc[7][u],
	           a[0].y, a[1].y, a[2].y, a[3].y, a[4].y, a[5].y, a[6].y, a[7].y, b.y);
}

// ---- three-product form for the plain unphased planes (rows H, Q of a variant; PAIRED lane rows) ----------------
// What UnphasedMath's r2 screen reads of a pair is HH = popc(H_A & H_B) and S = QH + HQ + 2 QQ (ScreenCountsUnphased below,
// d_unphased_math), not the four products one by one, and with C = H | Q (the carriers; H and Q are disjoint)
//     S = popc(Q_A & C_B) + popc(C_A & Q_B)            (Q_A & C_B = QH + QQ, C_A & Q_B = HQ + QQ)
// - three AND+popcounts per word and variant pair into two accumulators (v_bcnt adds, so both halves of S land in one
// register) instead of four into four.  The reference does the like in its list kernel: one popcount, the other cells from
// the margins (ld_engine.cpp:244-246).  Only pairs that pass the screen need HQ, QH and QQ themselves: k_recount_unphased
// (ld_three.hip.h) counts those few from their rows.
// Round 6: the carriers are never formed.  gfx950 has a three-input boolean instruction, v_bitop3_b32 (truth table in the
// instruction: 0xE0 = a & (b | c)), that issues at the rate of v_and_b32 (csrc/tools/bitop3_probe.hip: 24 x (v_bitop3, s_nop, v_bcnt)
// 2.527e13 products/s against 2.518e13 for v_and - and the three-product mix at 96.1 % of the and+bcnt ceiling where the v_or of
// rounds 5 and 6a left 85.4 / 88.7 %): Q_A & (H_B | Q_B) and Q_B & (H_A | Q_A) are one instruction each, so a product of the
// three-product form costs what a product of the four-product form costs, and there are three of them - PROVIDED no v_bitop3 reads
// a VGPR bank twice, which is what the rest of this section is about (profiles/r06_three_bitop3.txt has the numbers of every step).
//
// A whole chunk (8 slots of 4 words) of the three-product form with its operand registers placed by hand.  What csrc/tools/bitop3_probe.hip found:
//  * a VGPR's bank is its number mod 4 and register tuples start on even registers, so word x of EVERY ds_read_b64 pair lies in bank 0 or 2 and
//    word y in bank 1 or 3: any v_bitop3 of three same-numbered words has two sources in one bank, and that costs - the probe's half-slots run at
//    78.6 % of the and+bcnt ceiling in compiler-like registers, at 85.2 % with the four words of a product group (hA qA hB qB) in four banks, at
//    85.3 % with v_and in the place of every v_bitop3.  So the B pairs are read by ds_read2_b32 offset0:1 offset1:0 - the odd word into the even
//    register - H tuples start on registers = 0 and Q tuples on registers = 2 (mod 4), and no v_bitop3 (nor v_and) reads a bank twice;
//  * an LDS read costs the wave's VALU stream about the same whatever its width, and reads in one batch cost less than the same reads spread
//    through the products: the A variants' H and Q rows are read a whole slot at a time (ds_read_b128: 8 + 4 + 4 reads per 96 products where the
//    half-slot form had 24).
// hipcc cannot be told any of this, hence one asm statement with the operands in v36..v123 (the count kernel's other values fit below, above and in
// the gaps: 125 VGPRs, no scratch): A sets P and R (a slot each: H tuples of the lane's four A variants v36.. / v72.., their Q tuples v54.. / v90..), B sets
// BX = v108..v115 (lower half of a slot: the pairs hB0 qB0 hB1 qB1, words swapped) and BY = v116..v123 (upper half).  Order: the B pairs of a slot's upper half
// are read in front of its lower half's products, all of the next slot's A tuples and lower B pairs in front of its upper half's; LDS returns in
// order, so lgkmcnt(n) = "everything but the n reads just issued has arrived".  bA / bB: the lane's A / B row in the chunk's buffer; slot q
// lies at b ^ (q << 4) (the swizzle is in address bits 4..6); the B variants' second pair of rows is 2048 bytes on, beyond ds_read2_b32's
// offsets, and gets its own address.
// ((op, s_nop 0, v_bcnt) as in and_bcnt8v: without the s_nop behind the v_bitop3 the kernel runs at 82.8 % instead of 90.3 % of the and+bcnt ceiling, without
// any at 76.0 %; s_setprio around the read batches, either way round, costs 0.3 - 0.7 points - profiles/r06_three_bitop3.txt)
#define TWK_PR_AND(T, A, B, ACC) "v_and_b32 %[" #T "], v" #A ", v" #B "\n\t" "s_nop 0\n\tv_bcnt_u32_b32 %[" #ACC "], %[" #T "], %[" #ACC "]\n\t"
#define TWK_PR_BIT(T, A, B, C, ACC) "v_bitop3_b32 %[" #T "], v" #A ", v" #B ", v" #C " bitop3:0xe0\n\t" "s_nop 0\n\tv_bcnt_u32_b32 %[" #ACC "], %[" #T "], %[" #ACC "]\n\t"
// one B variant V against the lane's four A variants, one word: HH[s][V] += popc(hA[s] & hB), S[s][V] += popc(qA[s] & (hB | qB)) + popc(qB & (hA[s] | qA[s]))
#define TWK_G12(H0, H1, H2, H3, Q0, Q1, Q2, Q3, HB, QB, V) \
	TWK_PR_AND(t0, H0, HB, h0##V) TWK_PR_AND(t1, H1, HB, h1##V) TWK_PR_AND(t0, H2, HB, h2##V) TWK_PR_AND(t1, H3, HB, h3##V) \
	TWK_PR_BIT(t0, Q0, HB, QB, s0##V) TWK_PR_BIT(t1, Q1, HB, QB, s1##V) TWK_PR_BIT(t0, Q2, HB, QB, s2##V) TWK_PR_BIT(t1, Q3, HB, QB, s3##V) \
	TWK_PR_BIT(t0, QB, H0, Q0, s0##V) TWK_PR_BIT(t1, QB, H1, Q1, s1##V) TWK_PR_BIT(t0, QB, H2, Q2, s2##V) TWK_PR_BIT(t1, QB, H3, Q3, s3##V)
#define TWK_HALF0_P TWK_G12(36, 40, 44, 48, 54, 58, 62, 66, 109, 111, 0) TWK_G12(36, 40, 44, 48, 54, 58, 62, 66, 113, 115, 1) \
                    TWK_G12(37, 41, 45, 49, 55, 59, 63, 67, 108, 110, 0) TWK_G12(37, 41, 45, 49, 55, 59, 63, 67, 112, 114, 1)
#define TWK_HALF1_P TWK_G12(38, 42, 46, 50, 56, 60, 64, 68, 117, 119, 0) TWK_G12(38, 42, 46, 50, 56, 60, 64, 68, 121, 123, 1) \
                    TWK_G12(39, 43, 47, 51, 57, 61, 65, 69, 116, 118, 0) TWK_G12(39, 43, 47, 51, 57, 61, 65, 69, 120, 122, 1)
#define TWK_HALF0_R TWK_G12(72, 76, 80, 84, 90, 94, 98, 102, 109, 111, 0) TWK_G12(72, 76, 80, 84, 90, 94, 98, 102, 113, 115, 1) \
                    TWK_G12(73, 77, 81, 85, 91, 95, 99, 103, 108, 110, 0) TWK_G12(73, 77, 81, 85, 91, 95, 99, 103, 112, 114, 1)
#define TWK_HALF1_R TWK_G12(74, 78, 82, 86, 92, 96, 100, 104, 117, 119, 0) TWK_G12(74, 78, 82, 86, 92, 96, 100, 104, 121, 123, 1) \
                    TWK_G12(75, 79, 83, 87, 93, 97, 101, 105, 116, 118, 0) TWK_G12(75, 79, 83, 87, 93, 97, 101, 105, 120, 122, 1)
#define TWK_RD_A(LO, HI, OFF) "ds_read_b128 v[" #LO ":" #HI "], %[aA] offset:" #OFF "\n\t"
#define TWK_RD_B(LO, HI, ADDR, O0, O1) "ds_read2_b32 v[" #LO ":" #HI "], %[" #ADDR "] offset0:" #O0 " offset1:" #O1 "\n\t"
#define TWK_READ_P TWK_RD_A(36, 39, 0) TWK_RD_A(54, 57, 128) TWK_RD_A(40, 43, 2048) TWK_RD_A(58, 61, 2176) TWK_RD_A(44, 47, 4096) TWK_RD_A(62, 65, 4224) TWK_RD_A(48, 51, 6144) TWK_RD_A(66, 69, 6272)
#define TWK_READ_R TWK_RD_A(72, 75, 0) TWK_RD_A(90, 93, 128) TWK_RD_A(76, 79, 2048) TWK_RD_A(94, 97, 2176) TWK_RD_A(80, 83, 4096) TWK_RD_A(98, 101, 4224) TWK_RD_A(84, 87, 6144) TWK_RD_A(102, 105, 6272)
#define TWK_READ_BX TWK_RD_B(108, 109, aB, 1, 0) TWK_RD_B(110, 111, aB, 33, 32) TWK_RD_B(112, 113, aB2, 1, 0) TWK_RD_B(114, 115, aB2, 33, 32)
#define TWK_READ_BY TWK_RD_B(116, 117, aB, 3, 2) TWK_RD_B(118, 119, aB, 35, 34) TWK_RD_B(120, 121, aB2, 3, 2) TWK_RD_B(122, 123, aB2, 35, 34)
#define TWK_CLOBBER_P "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69"
#define TWK_CLOBBER_R "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105"
#define TWK_CLOBBER_B "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123"
#define TWK_SLOT_ADDR(Q16) "v_xor_b32 %[aA], " #Q16 ", %[bA]\n\tv_xor_b32 %[aB], " #Q16 ", %[bB]\n\tv_add_u32 %[aB2], 0x800, %[aB]\n\t"
// slot q out of A set CUR: the B pairs of its upper half are read in front of its lower half's products, the whole of slot q + 1 (A set NXT, the
// lower B pairs) in front of its upper half's
#define TWK_SLOT(CUR, NXT, NEXT_Q16) TWK_READ_BY "s_waitcnt lgkmcnt(4)\n\t" TWK_HALF0_##CUR TWK_SLOT_ADDR(NEXT_Q16) TWK_READ_##NXT TWK_READ_BX "s_waitcnt lgkmcnt(12)\n\t" TWK_HALF1_##CUR
__device__ __forceinline__ void contract3_chunk(uint32_t (&acc)[8][4], uint32_t bA, uint32_t bB) {
	static_assert(KC * 4 == 128 && lane_row_offset<true, 1>() == 128 && lane_row_offset<true, 2>() == 2048 && lane_row_offset<true, 7>() == 6272, "the offsets in TWK_READ_P / _R / _BX / _BY");
	uint32_t t0, t1, aA, aB, aB2;
	asm volatile(TWK_SLOT_ADDR(0) TWK_READ_P TWK_READ_BX
	             TWK_SLOT(P, R, 16) TWK_SLOT(R, P, 32) TWK_SLOT(P, R, 48) TWK_SLOT(R, P, 64) TWK_SLOT(P, R, 80) TWK_SLOT(R, P, 96) TWK_SLOT(P, R, 112)
	             TWK_READ_BY "s_waitcnt lgkmcnt(4)\n\t" TWK_HALF0_R "s_waitcnt lgkmcnt(0)\n\t" TWK_HALF1_R
	             : [h00] "+v"(acc[0][0]), [h01] "+v"(acc[0][2]), [h10] "+v"(acc[2][0]), [h11] "+v"(acc[2][2]), [h20] "+v"(acc[4][0]), [h21] "+v"(acc[4][2]),
	               [h30] "+v"(acc[6][0]), [h31] "+v"(acc[6][2]), [s00] "+v"(acc[1][1]), [s01] "+v"(acc[1][3]), [s10] "+v"(acc[3][1]), [s11] "+v"(acc[3][3]),
	               [s20] "+v"(acc[5][1]), [s21] "+v"(acc[5][3]), [s30] "+v"(acc[7][1]), [s31] "+v"(acc[7][3]),
	               [t0] "=&v"(t0), [t1] "=&v"(t1), [aA] "=&v"(aA), [aB] "=&v"(aB), [aB2] "=&v"(aB2)
	             : [bA] "v"(bA), [bB] "v"(bB)
	             : "memory", TWK_CLOBBER_P, TWK_CLOBBER_R, TWK_CLOBBER_B);
}

// One slot the plain way (reads, wait, products): the last chunk of rows that end inside it (CountWork::last_halves), one such chunk per unit.
__device__ __forceinline__ void contract3_slot(uint32_t (&acc)[8][4], uint32_t aA, uint32_t aB) {
	uint32_t t0, t1, aB2;
	asm volatile("v_add_u32 %[aB2], 0x800, %[aB]\n\t" TWK_READ_P TWK_READ_BX TWK_READ_BY "s_waitcnt lgkmcnt(0)\n\t" TWK_HALF0_P TWK_HALF1_P
	             : [h00] "+v"(acc[0][0]), [h01] "+v"(acc[0][2]), [h10] "+v"(acc[2][0]), [h11] "+v"(acc[2][2]), [h20] "+v"(acc[4][0]), [h21] "+v"(acc[4][2]),
	               [h30] "+v"(acc[6][0]), [h31] "+v"(acc[6][2]), [s00] "+v"(acc[1][1]), [s01] "+v"(acc[1][3]), [s10] "+v"(acc[3][1]), [s11] "+v"(acc[3][3]),
	               [s20] "+v"(acc[5][1]), [s21] "+v"(acc[5][3]), [s30] "+v"(acc[7][1]), [s31] "+v"(acc[7][3]),
	               [t0] "=&v"(t0), [t1] "=&v"(t1), [aB2] "=&v"(aB2)
	             : [aA] "v"(aA), [aB] "v"(aB)
	             : "memory", TWK_CLOBBER_P, TWK_CLOBBER_B);
}

// ---- persistent work-list form of the same contraction ------------------------------------
// One launch = a list of 128 x 128 tiles of one super-tile (only the tiles that hold wanted pairs:
// on/above the diagonal, inside the window band, ...) run by P persistent blocks, P = the number
// of blocks the chip holds at once (2 per CU), which pull *units* of work from an atomic ticket:
//   * a unit is a K-range [c0, c1) of one tile; whole tiles (full K range) are stored plainly, parts
//     of a tile are added with atomics into a tile that a small kernel zeroed beforehand;
//   * the host (build_count_units) hands out whole tiles while more than ~2 rounds of work remain
//     and ever shorter K-ranges after that, so the launch ends within a few short units of balance.
// Why not a plain 2-D grid: T tiles take ceil(T / P) rounds and the partial last round is pure loss
// (12 % of a configs[1]-sized launch, more for thin window-mode row blocks).  Why not a static equal
// split (stream-K): the two blocks of a CU do NOT run at the same speed - the issue arbiter favours
// the older waves, so the block that arrived first runs ~1.5x faster than its mate (measured: with
// equal static shares half of the blocks finish at 55 % of the kernel time, and their mates then
// run alone at 78 % of the CU's rate).  With tickets the fast block simply takes more units.
// The software pipeline (LDS-DMA of chunk i+1 behind the contraction of chunk i) runs across units;
// the next unit's ticket is drawn by thread 0 at the first chunk of a unit and handed to the other
// waves through a two-slot LDS mailbox behind the next chunk barrier, so no extra synchronisation
// is paid (units of a single chunk excepted).
struct CountUnit { uint32_t tile, c0, c1, yx; };       // chunks [c0, c1) of tiles[tile]; yx = tiles[tile], so that a block learns everything about its
                                                       // next unit from one 16-byte scalar load (fill_unit_tiles) instead of two dependent ones
struct CountWork {
	const uint32_t* rows; uint32_t W;      // plane rows, row pitch in words (multiple of KC)
	uint32_t rowA0, rowB0;                 // first plane row of the super-tile's row / column axis
	const uint32_t* tiles;                 // [n_tiles]: (tile row << 16) | tile column, within the super-tile
	const CountUnit* units;                // [n_units], in the order they are handed out
	uint32_t n_units;
	uint32_t* C; uint32_t ldc;             // counts of the super-tile
	uint32_t* ticket;                      // [n_queues], zeroed before the launch
	// One queue: units[0, n_units) in order.  Several (long rows, one per XCD): queue q is units[queue_begin[q],
	// queue_begin[q + 1]); a block starts on the queue of the XCD it runs on and moves on to the next when that is empty.
	uint32_t n_queues;
	uint32_t queue_begin[9];
	// Half-slots (8 bytes per row) of a row's LAST chunk that carry data, 1..16; 0 = all 16.  The padding behind
	// them is zero in every plane, so the contraction of the last chunk stops there: at 2 504 samples an unphased
	// row is 79 live words in 3 chunks of 32, and the last 8 half-slots of every tile are skipped (-17 %).
	uint32_t last_halves;
	// [2 + 2 + 8] (may be null): [4 + x]: the latest tick of the 100 MHz counter at which a block on XCD x finished (how far apart
	// the XCDs end a launch: the outlier watch of twk_hip.hip); [0], [1]: every block adds the shader cycles (s_memtime) and the ticks of the constant 100 MHz counter
	// (s_memrealtime) it lived for: their ratio is the clock the launch really ran at.  The and+bcnt ceiling is quoted at
	// 2.4 GHz; a launch that starts on an idle chip runs its first ~20 ms below that (1.9 GHz for a 2 ms launch,
	// profiles/r04_clock_probe.txt), which is most of what short-row runs lose against long ones.
	unsigned long long* clocks;
};

// The unit table of a launch (host side; shared by the engine and the dev tools).  Guided self-scheduling:
// a unit is never longer than 1/share_div of an even share of the work still to be handed out, so whole
// tiles go out only while more than tail_rounds (= share_div) rounds of work remain, and after that
// K-ranges that shrink with the remainder down to `min_chunks` chunks.  The launch then ends within a few
// of the shortest units of perfect balance however unequal the blocks' speeds are (measured: finish
// times of the 512 blocks within 0.14 ms of each other on a 21 ms launch).  Returns the index of the
// first tile that is split (tiles from there on must be zeroed before the launch); n_tiles if none.
// Long rows, segmented walk (seg_chunks != 0, rows of at least two segments, patch_end = the list index where each
// patch of tiles ends): the tiles of a patch - which share their row and column tiles - are not contracted one whole
// tile per block, each block at its own pace, but K segment by K segment: all tiles of the patch over chunks
// [0, seg), then all over [seg, 2 seg), ...  The blocks working on a patch are then always within a segment of each
// other, so a chunk of a row tile fetched for one block is still in the MALL (and often in L2) when the other blocks
// that need it come by: HBM sees every (row tile, segment) about once per patch instead of once per tile.  The price:
// every unit is a part of its tile's K range (counts added with atomics into zeroed tiles - 32 adds per lane per
// segment, nothing next to the segment's 2048 x seg VALU ops).
template <class Vec>
inline uint32_t build_count_units(uint32_t n_tiles, uint32_t nchunks, uint32_t n_blocks, uint32_t min_chunks, Vec& units,
                                  uint32_t share_div = 8, uint32_t tail_rounds = 8,
                                  const uint32_t* patch_end = nullptr, uint32_t n_patches = 0, uint32_t seg_chunks = 0) {
	units.clear();
	const uint32_t tail = (uint32_t)(n_tiles < (unsigned long long)tail_rounds * n_blocks ? n_tiles : (unsigned long long)tail_rounds * n_blocks);
	uint32_t first_split = n_tiles - tail;
	if (nchunks < 2 * min_chunks) first_split = n_tiles;             // rows too short to be worth splitting
	const bool segmented = seg_chunks && patch_end && n_patches && nchunks >= 2 * seg_chunks && nchunks >= 2 * min_chunks;
	if (segmented) {
		const uint32_t nseg = (nchunks + seg_chunks - 1) / seg_chunks;
		uint32_t p0 = 0;
		for (uint32_t p = 0; p < n_patches && p0 < first_split; ++p) {
			const uint32_t p1 = patch_end[p] < first_split ? patch_end[p] : first_split;
			for (uint32_t sgm = 0; sgm < nseg; ++sgm)
				for (uint32_t t = p0; t < p1; ++t)
					units.push_back(CountUnit{t, (uint32_t)((unsigned long long)nchunks * sgm / nseg), (uint32_t)((unsigned long long)nchunks * (sgm + 1) / nseg), 0});
			p0 = p1;
		}
	} else {
		for (uint32_t t = 0; t < first_split; ++t) units.push_back(CountUnit{t, 0, nchunks, 0});
	}
	for (uint32_t t = first_split; t < n_tiles; ++t) {
		const unsigned long long rem = (unsigned long long)(n_tiles - t) * nchunks;      // chunks left, this tile included
		unsigned long long target = rem / ((unsigned long long)share_div * n_blocks);
		if (target < min_chunks) target = min_chunks;
		if (segmented && target > seg_chunks) target = seg_chunks;
		uint32_t S = (uint32_t)((nchunks + target - 1) / target);
		if (S < 1) S = 1;
		if (S > nchunks) S = nchunks;
		for (uint32_t k = 0; k < S; ++k)
			units.push_back(CountUnit{t, (uint32_t)((unsigned long long)nchunks * k / S), (uint32_t)((unsigned long long)nchunks * (k + 1) / S), 0});
	}
	return segmented ? 0 : first_split;
}

template <class Vec>
inline void fill_unit_tiles(Vec& units, const uint32_t* tiles) { for (auto& u : units) u.yx = tiles[u.tile]; }

// What a block does with the 8 x TB counts each of its lanes holds when a unit ends.  StoreCounts is the plain form:
// the counts go to the super-tile's C matrix (stored for a whole tile, added for a part of its K range).
// What a wave keeps between the units it runs (fused forms: the candidate slots it has reserved and not yet used): three
// words of LDS per wave - the first free slot (64 bits) and how many are left; in registers they were spilled.
struct SlotWindow { uint32_t* w; uint2* q; };       // q: the wave's queue of pairs that passed the prefilter (ScreenCounts: QUEUE entries; else null)
constexpr int SCREEN_QUEUE = 128;     // two entries per lane of the wave: the exact test of a tile's flagged pairs is one or two passes of the wave

template <int TB>
struct StoreCounts {
	static constexpr int META_WORDS = 0;       // nothing to stage for the epilogue
	static constexpr bool PAIRED_ROWS = false; // a lane's rows are li + 8t (see read_half)
	static constexpr bool THREE_PRODUCTS = false;
	static constexpr bool QUEUED = false;
	uint32_t* C; uint32_t ldc;
	__device__ __forceinline__ const uint32_t* meta_src(uint32_t, uint32_t) const { return nullptr; }
	__device__ __forceinline__ void finish(SlotWindow&, int) const {}
	__device__ __forceinline__ void operator()(uint32_t (&acc)[8][TB], uint32_t yx, int wr, int wc, int li, int lj, int lane, bool whole, const uint32_t*, SlotWindow&) const {
		uint32_t* Cblk = C + (size_t)((yx >> 16) * TILE + wr * 64 + li) * ldc + (yx & 0xFFFFu) * TILE + wc * 8 * TB + lj;
		if (whole) {
#pragma unroll
			for (int t = 0; t < 8; ++t)
#pragma unroll
				for (int u = 0; u < TB; ++u) { Cblk[(size_t)(8 * t) * ldc + 8 * u] = acc[t][u]; acc[t][u] = 0; }
		} else {
#pragma unroll
			for (int t = 0; t < 8; ++t)
#pragma unroll
				for (int u = 0; u < TB; ++u) { atomicAdd(&Cblk[(size_t)(8 * t) * ldc + 8 * u], acc[t][u]); acc[t][u] = 0; }
		}
	}
};

// The three-product form's plain epilogue: (HH, S) of the variant pair (A, B) of the super-tile go to C[A * ldc + 2 B + {0, 1}]
// - the count matrix of the four-product form is rows x rows words, this one half of that: variants x 2 variants (ldc = plane
// rows of the super-tile's column axis, as before).  A tile is 64 x 64 variant pairs = 64 rows of 128 words.
template <int TB>
struct StoreCounts3 {
	static constexpr int META_WORDS = 0;
	static constexpr bool PAIRED_ROWS = true;
	static constexpr bool THREE_PRODUCTS = true;
	static constexpr bool QUEUED = false;
	uint32_t* C; uint32_t ldc;
	__device__ __forceinline__ const uint32_t* meta_src(uint32_t, uint32_t) const { return nullptr; }
	__device__ __forceinline__ void finish(SlotWindow&, int) const {}
	__device__ __forceinline__ void operator()(uint32_t (&acc)[8][TB], uint32_t yx, int wr, int wc, int li, int lj, int lane, bool whole, const uint32_t*, SlotWindow&) const {
		static_assert(TB == 4, "two column variants per lane");
		// the lane's variant pairs: rows vA + 8 s (s = 0..3), columns vB + 8 v (v = 0, 1), as in ScreenCountsUnphased
		uint32_t* Cblk = C + (size_t)((yx >> 16) * (TILE / 2) + wr * 32 + li) * ldc + 2 * ((yx & 0xFFFFu) * (TILE / 2) + wc * 16 + lj);
#pragma unroll
		for (int s = 0; s < 4; ++s)
#pragma unroll
			for (int v = 0; v < 2; ++v) {
				uint32_t* e = Cblk + (size_t)(8 * s) * ldc + 16 * v;
				if (whole) *reinterpret_cast<uint2*>(e) = make_uint2(acc[2 * s][2 * v], acc[2 * s + 1][2 * v + 1]);
				else { atomicAdd(e, acc[2 * s][2 * v]); atomicAdd(e + 1, acc[2 * s + 1][2 * v + 1]); }
				acc[2 * s][2 * v] = 0; acc[2 * s + 1][2 * v + 1] = 0;
			}
	}
};

template <int NW, int EXPERIMENT, class Epilogue>      // EXPERIMENT == 5: the dev tool's finish-time probe (overwrites C)
__device__ __forceinline__ void count_list_body(const CountWork& w, const Epilogue& epilogue) {
	constexpr int WC = NW / 2;
	constexpr int TB = 16 / WC;
	constexpr int NSEG = 32 / NW;
	__shared__ __attribute__((aligned(16))) uint32_t lds[2 * 2 * TILE * KC];
	__shared__ uint32_t mbox[2];
	// What a fused epilogue reads per row and column of its tile (allele counts, band limits): fetched by LDS-DMA when the
	// unit starts - one dword per lane, no registers held - and published by the chunk barriers like the operand chunks, so
	// that the epilogue finds it in LDS instead of waiting a memory round trip per tile.
	constexpr int META = Epilogue::META_WORDS;
	static_assert(META % 64 == 0 && META <= 2 * NW * 64, "whole waves, at most two dwords per thread");
	__shared__ uint32_t meta[META ? META : 1];

	const int tid  = threadIdx.x;
	const int lane = tid & 63;
	const int wave = tid >> 6;
	const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // (wave-uniform: lives in a scalar register, and so do the wave's tile coordinates)
	const int wr = wave_u / WC, wc = wave_u % WC;
	const uint32_t nchunks = w.W / KC;
	const uint32_t n_units = w.n_units;
	const unsigned long long probe_wall0 = wall_clock64(), probe_clk0 = clock64();      // the block's start on the constant 100 MHz clock and on the shader clock (scalar registers)

	// unit id -> (tile, first chunk, end chunk); wave-uniform (scalar loads)
	auto decode = [&](uint32_t u, uint32_t& tl, uint32_t& c0, uint32_t& c1, uint32_t& yx) {
		const CountUnit cu = w.units[u];
		tl = __builtin_amdgcn_readfirstlane(cu.tile); c0 = __builtin_amdgcn_readfirstlane(cu.c0); c1 = __builtin_amdgcn_readfirstlane(cu.c1);
		yx = __builtin_amdgcn_readfirstlane(cu.yx);
	};

	// Drawing a unit (thread 0): the next ticket of the block's queue; with one queue per XCD the queue of the XCD this
	// block runs on - whose L2 then holds the row and column tiles its 64 blocks share - and, once that is empty, the
	// next one that still has work.
	uint32_t my_q = 0;
	if (w.n_queues > 1) {
		uint32_t xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		my_q = (xcc & 0xFu) % w.n_queues;
	}
	auto draw = [&]() -> uint32_t {
		if (w.n_queues <= 1) return atomicAdd(w.ticket, 1u);
		for (uint32_t k = 0; k < w.n_queues; ++k) {
			const uint32_t q = my_q + k < w.n_queues ? my_q + k : my_q + k - w.n_queues;
			const uint32_t t = atomicAdd(w.ticket + q, 1u);
			if (t < w.queue_begin[q + 1] - w.queue_begin[q]) { my_q = q; return w.queue_begin[q] + t; }
		}
		return 0xFFFFFFFFu;
	};
	// the first ticket of the block
	uint32_t fetched = 0;                       // thread 0: the ticket in flight (the next unit)
	if (tid == 0) mbox[0] = draw();
	__syncthreads();
	uint32_t unit = __builtin_amdgcn_readfirstlane(mbox[0]);
	if (unit >= n_units) return;
	uint32_t unit_next = 0xFFFFFFFFu;
	uint32_t n_started = 0;                     // units this block has started (mailbox slot parity)
	bool unit_start = true;                     // this iteration is the first of its unit
	bool want_next = false;                     // unit_next is to be picked up from the mailbox behind the next barrier

	uint32_t tile, c, c_end, tile_yx;
	decode(unit, tile, c, c_end, tile_yx);

	uint32_t acc[8][TB];
#pragma unroll
	for (int t = 0; t < 8; ++t)
#pragma unroll
		for (int u = 0; u < TB; ++u) acc[t][u] = 0;

	constexpr bool PAIRED = Epilogue::PAIRED_ROWS;
	constexpr bool THREE = Epilogue::THREE_PRODUCTS;      // the three-product form of the plain unphased planes (contract3_chunk)
	static_assert(!THREE || (PAIRED && TB == 4), "the three-product form needs a variant's H and Q rows in one lane");
	// The per-lane LDS read offsets and DMA source offsets of the K loop.  For the fused epilogues they are recomputed when a unit ends
	// instead of being held through the epilogue: the epilogue is where the kernels' register demand peaks, and what the allocator evicts
	// there it reloads from scratch inside the K loop (round 5: sixteen spilled offsets, scratch loads between the half-slots).  The lane
	// id comes fresh from the hardware (volatile asm), so the recomputation is not hoisted back in front of the loop.
	// (One offset per operand: slot k of the lane's rows lies at off ^ (k << 4) - the swizzle lives in address bits 4..6 - and the buffer's
	// base, a multiple of 32 KiB, can be added first: (bufbase + off) ^ (k << 4).  A table of all eight per operand cost 14 more registers
	// for the same number of VALU operations per read address.)
	uint32_t offA, offB, voff_even, voff_odd;
	auto lane_offsets = [&](bool fresh) {
		uint32_t ln = (uint32_t)lane;
		if (fresh) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
		const uint32_t li_ = ln >> 3, lj_ = ln & 7u;
		if (PAIRED) {      // rows 2 li + (t & 1) + 16 (t >> 1): (row >> 1) & 7 = li for every t
			offA = (uint32_t)((wr * 64 + 2 * li_) * (KC * 4) + (li_ << 4));
			offB = (uint32_t)(LDS_TILE_BYTES + (wc * 8 * TB + 2 * lj_) * (KC * 4) + (lj_ << 4));
		} else {
			offA = (uint32_t)((wr * 64 + li_) * (KC * 4) + ((li_ >> 1) << 4));
			offB = (uint32_t)(LDS_TILE_BYTES + (wc * 8 * TB + lj_) * (KC * 4) + ((lj_ >> 1) << 4));
		}
		// DMA source offsets: lane's row within an 8-row segment and its (swizzled) 16-byte slot, see stage_rows_s
		voff_even = (li_ * w.W + ((lj_ ^ (li_ >> 1)) << 2)) << 2;
		voff_odd  = (li_ * w.W + ((lj_ ^ (li_ >> 1) ^ 4u) << 2)) << 2;
	};
	lane_offsets(false);
	constexpr int ODD = PAIRED ? 0 : 4;          // what the slot index of an odd t / u is XORed with (see the offA table)

	// The lane id, fresh from the hardware: what the loop needs of it per unit (thread 0's ticket, the epilogue's rows and columns) is
	// derived where it is used instead of being held in registers through the K loop.
	auto fresh_lane = [&]() -> uint32_t { uint32_t ln; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln)); return ln; };
	auto thread0 = [&]() -> bool { return wave_u == 0 && fresh_lane() == 0; };
	const bool st_isB = wave_u >= NW / 2;
	const int st_seg0 = (wave_u % (NW / 2)) * NSEG;
	const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t*)lds;
	const uint32_t st_lds = lds_base + (st_isB ? (uint32_t)LDS_TILE_BYTES : 0u);
	// first plane row this wave stages for a tile: the A rows (first half of the waves) or the B rows
	auto stage_row0 = [&](uint32_t yx) -> uint32_t {       // (yx: wave-uniform, lives in an SGPR)
		return st_isB ? w.rowB0 + (yx & 0xFFFFu) * TILE : w.rowA0 + (yx >> 16) * TILE;
	};

	uint32_t st_row0 = stage_row0(tile_yx);
	stage_rows_s(w.rows, w.W, st_row0, c, st_lds, st_seg0, NSEG, voff_even, voff_odd);
	uint32_t seg_c0 = c;
	__shared__ uint32_t window_words[NW][8];
	constexpr bool QUEUED = Epilogue::QUEUED;
	__shared__ uint2 queue_words[QUEUED ? NW : 1][QUEUED ? SCREEN_QUEUE : 1];
	SlotWindow window{&window_words[wave_u][0], QUEUED ? &queue_words[wave_u][0] : nullptr};
	if (lane < 8) window.w[lane] = 0;           // (wave-private: no barrier)
	int buf = 0;
	for (;;) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		// the ticket fetched one iteration ago has arrived with everything else: publish it
		if (want_next && thread0()) mbox[n_started & 1u] = fetched;
		if (EXPERIMENT != 7 || unit_start || want_next) __syncthreads();      // (7: the dev tool's timing WITHOUT the chunk barrier - wrong counts: what the barrier costs)
		if (want_next) { unit_next = __builtin_amdgcn_readfirstlane(mbox[n_started & 1u]); want_next = false; }
		if (unit_start) {
			// First chunk of a unit: thread 0 draws the ticket of the next unit.  It is needed when the
			// last chunk of this unit is contracted (to stage the next unit's first chunk behind it); with
			// two or more chunks it travels through the mailbox at the next barrier for free, a one-chunk
			// unit has to wait for it here.  A block thus never holds more than its current and next unit,
			// which keeps the slow block of a CU from sitting on big units drawn long ago.
			++n_started;
			if (thread0()) fetched = draw();
			if (META) {
#pragma unroll
				for (int round = 0; round < (META + NW * 64 - 1) / (NW * 64); ++round) {
					const int mw = wave_u + round * NW;          // the 64 dwords this wave fetches in this round
					if (mw < META / 64) {
						const uint32_t yx_m = tile_yx;
						// (the lane id is taken afresh from the hardware, by volatile asm: derived from threadIdx.x the compiler computed
						// wave * 64 + lane once in front of the loop and kept it in scratch - the kernels' one spilled register, and the
						// only reason they needed scratch memory at all)
						uint32_t ln;
						asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
						const uint32_t mi = (uint32_t)mw * 64u + ln;
						glds4(epilogue.meta_src(yx_m, mi), (uint32_t)(uintptr_t)(lptr_t*)meta + (uint32_t)mw * 256u);
					}
				}
				if (c + 1 == c_end) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // a one-chunk unit needs it in this very iteration: the barrier below publishes it
			}
			if (c + 1 == c_end) {
				if (thread0()) mbox[n_started & 1u] = fetched;
				__syncthreads();
				unit_next = __builtin_amdgcn_readfirstlane(mbox[n_started & 1u]);
			} else {
				want_next = true;
			}
			unit_start = false;
		}
		// prefetch the chunk after this one: same unit, or the first chunk of the next unit
		uint32_t n_tile = tile, n_c = c + 1, n_end = c_end, n_yx = tile_yx;
		bool more = true;
		if (n_c == c_end) {
			more = unit_next < n_units;
			if (more) { decode(unit_next, n_tile, n_c, n_end, n_yx); st_row0 = stage_row0(n_yx); }
		}
		// (issued here, in front of the chunk's LDS reads; behind the first or the third half-slot's contraction it is neither faster nor
		// slower - 18.31 / 18.31 / 18.26 ms on the 1 M-sample microbenchmark, round 5)
		if (more && (EXPERIMENT != 8 || n_c == 0 || n_c + 1 >= n_end)) stage_rows_s(w.rows, w.W, st_row0, n_c, st_lds + (buf ^ 1) * (2 * LDS_TILE_BYTES), st_seg0, NSEG, voff_even, voff_odd);      // (8: the dev tool's timing WITHOUT the operand staging)
		// The chunk in 16 half-slots of 8 bytes per row.  The 12 LDS reads of half-slot h + 1 (8 A rows,
		// TB B rows, ds_read_b64) are issued before the contraction of half-slot h, into the other
		// register set, so the contraction never waits for LDS except at the first half-slot of a chunk.
		// hipcc will not keep such a schedule by itself (it sinks the reads to their uses to save
		// registers: -10 %), so reads, waits and contraction are volatile asm in program order and the
		// outstanding-read count is tracked by hand (LDS returns in order: lgkmcnt(12) = "everything but
		// the 12 reads just issued has arrived").
		const uint32_t bufbase = lds_base + (uint32_t)buf * (2 * LDS_TILE_BYTES);
		const int h_end = (c + 1 == nchunks && w.last_halves) ? (int)w.last_halves : 16;     // wave-uniform
		if constexpr (THREE) {
			// (operand registers placed by hand, see contract3_chunk; a row's last, partly filled chunk slot by slot - the half-slot of padding that may
			// come with it is zeros)
			if (h_end == 16) contract3_chunk(acc, bufbase + offA, bufbase + offB);
			else {
#pragma unroll 1
				for (uint32_t q = 0; q < (uint32_t)(h_end + 1) >> 1; ++q) contract3_slot(acc, bufbase + (offA ^ (q << 4)), bufbase + (offB ^ (q << 4)));
			}
		} else if (h_end == 16) {
			uint2 ra[2][8], rb[2][TB];
			const uint32_t baseA = bufbase + offA, baseB = bufbase + offB;
			read_half<TB, PAIRED>(ra[0], rb[0], baseA, baseA ^ (uint32_t)(ODD << 4), baseB, baseB ^ (uint32_t)(ODD << 4), 0);
#pragma unroll
			for (int h = 0; h < 16; ++h) {
				if (h + 1 < 16) {
					const int q = (h + 1) >> 1;
					read_half<TB, PAIRED>(ra[(h + 1) & 1], rb[(h + 1) & 1], baseA ^ (uint32_t)(q << 4), baseA ^ (uint32_t)((q ^ ODD) << 4), baseB ^ (uint32_t)(q << 4),
					              baseB ^ (uint32_t)((q ^ ODD) << 4), (h + 1) & 1);
					asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
				} else {
					asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				}
#pragma unroll
				for (int u = 0; u < TB; ++u) contract_half<TB>(acc, u, ra[h & 1], rb[h & 1][u]);
			}
		} else {
			// The last chunk of a row whose data ends before the chunk does (CountWork::last_halves): the zero padding is not contracted.
			// A rolled loop - one such chunk per tile.  Slot k of the lane's rows lies at off ^ (k << 4): the swizzle lives in address bits 4..6.
			auto rd = [&](uint2 (&a)[8], uint2 (&b)[TB], int h) {
				const uint32_t q = (uint32_t)(h >> 1) << 4, hb = (uint32_t)(h & 1) << 3;
				read_half<TB, PAIRED>(a, b, bufbase + (offA ^ q) + hb, bufbase + (offA ^ q ^ (uint32_t)(ODD << 4)) + hb, bufbase + (offB ^ q) + hb,
				              bufbase + (offB ^ q ^ (uint32_t)(ODD << 4)) + hb, 0);
			};
			// (Pipelined in pairs of half-slots - the next half-slot's reads behind this one's contraction, two register sets - it was slower, not
			// faster: 240.7 -> 246.8 ms on the 2,504-sample -u run, the three-product kernels at 124 registers instead of 100; round 5.)
#pragma unroll 1
			for (int h = 0; h < h_end; ++h) {
				uint2 ra[8], rb[TB];
				rd(ra, rb, h);
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
				for (int u = 0; u < TB; ++u) contract_half<TB>(acc, u, ra, rb[u]);
			}
		}

		if (c + 1 == c_end) {          // unit done: write (whole tile) or add (part of a tile's K range) - or screen (fused form)
			const uint32_t yx = tile_yx;
			if (EXPERIMENT != 6) {      // (6: the dev tool's no-epilogue timing)
				const uint32_t ln = META ? fresh_lane() : (uint32_t)lane;
				epilogue(acc, yx, wr, wc, (int)(ln >> 3), (int)(ln & 7u), (int)ln, seg_c0 == 0 && c_end == nchunks, meta, window);
			}
			if (META) lane_offsets(true);
			if (!more) {
				epilogue.finish(window, (int)fresh_lane());
				if (thread0() && w.clocks) {
					typedef __attribute__((address_space(1))) unsigned long long g_u64;
					__hip_atomic_fetch_add((g_u64*)w.clocks, clock64() - probe_clk0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_fetch_add((g_u64*)w.clocks + 1, wall_clock64() - probe_wall0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					uint32_t xcc_id;
					asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
					__hip_atomic_fetch_max((g_u64*)w.clocks + 4 + (xcc_id & 7u), wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				if (EXPERIMENT == 5 && thread0()) {      // probe: when did this block finish, and on which XCD / CU?
					uint32_t xcc, hwid;
					asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
					asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
					unsigned long long* o = reinterpret_cast<unsigned long long*>(w.C) + 4 * (size_t)blockIdx.x;
					const unsigned long long wall1 = wall_clock64(), clk1 = clock64();
					o[0] = wall1; o[1] = ((unsigned long long)xcc << 32) | hwid;
					o[2] = wall1 - probe_wall0; o[3] = clk1 - probe_clk0;       // the block's life on both clocks: their ratio is the shader clock it ran at
				}
				break;
			}
			unit = unit_next; seg_c0 = n_c; unit_start = true;          // the next unit starts
		}
		tile = n_tile; c = n_c; c_end = n_end; tile_yx = n_yx;
		buf ^= 1;
	}
}

template <int NW, int EXPERIMENT = 0>
__global__ __launch_bounds__(NW * 64, NW / 2)
void k_count_list_t(const CountWork w) {
	count_list_body<NW, EXPERIMENT>(w, StoreCounts<16 / (NW / 2)>{w.C, w.ldc});
}

template <int NW, int EXPERIMENT = 0>
__global__ __launch_bounds__(NW * 64, NW / 2)
void k_count3_list_t(const CountWork w) {
	count_list_body<NW, EXPERIMENT>(w, StoreCounts3<16 / (NW / 2)>{w.C, w.ldc});
}

// Inclusive prefix sum over the 64 lanes of a wave (all active) without LDS: Hillis-Steele inside each row of 16 lanes
// (row_shr 1, 2, 4, 8; lanes shifted in from outside a row contribute 0), then row 0's total into row 1 and row 2's into row
// 3 (row_bcast:15), then lane 31's into rows 2 and 3 (row_bcast:31).  __shfl_up goes through ds_bpermute: six dependent LDS
// round trips where a wave with candidates can least afford them, behind its last contraction.
__device__ __forceinline__ uint32_t wave_scan_inclusive(uint32_t x) {
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
	return x;
}

// ---- fused form for short rows: count -> r2 screen -> candidate list --------------------------------------
// With a few thousand samples a pair's whole contraction is a few hundred word pairs, and writing its count
// to C (4 B), reading it back in the math kernel and running that kernel's FP64 front end on it - one thread
// per pair, almost all of them rejected by the r2 cut-off - costs as much as counting it.  The reference has
// no such round trip: its per-pair loop goes count -> math in registers (ld_engine.cpp:1898-2015).  Here the
// block that counted a tile also screens it: phased planes only (one count per pair, in the lane's own
// accumulator), and only the part of PhasedMath that needs no division (ld_engine.cpp:1162-1310):
//     r2 = D^2 / (pA qA pB qB) = (2N AA - acA acB)^2 / (acA (2N - acA) acB (2N - acB))
// (AA = the pair's count, acA / acB = the rows' popcounts, 2N haplotypes, no missing data in these planes), so
//     a pair can pass r2 >= minR2 only if  (2N AA - acA acB)^2 >= minR2 (1 - 1e-6) acA (2N - acA) acB (2N - acB).
// All integers involved are exact in FP64 for 2N < 2^26; the factor 1 - 1e-6 is six orders of magnitude more
// than the rounding of either side, so no pair the reference's rounded formula would keep is lost (the same
// screen stands at the head of d_phased_math).  Pairs that pass - with the structural tests: inside the
// super-tile, above the diagonal, inside the r2 band / the window - are *candidates*: (set position A, set
// position B, AA) appended to a list with one atomic per wave, and k_ld_stats_list (ld_math.hip.h) runs the
// whole of the reference's math on exactly those, one candidate per lane.  Nothing is written for the rest.
struct ScreenWork {
	const uint32_t* rowpop;            // ALT alleles per position of the plane set (row popcounts, P = 1)
	uint32_t a0, b0, nA, nB;           // set positions of the super-tile's first row / column, variants on each axis
	uint32_t n_variants; int diag;
	const uint32_t* col_hi; uint32_t hi_a0, hi_b0;       // r2 band (twk_hip.hip region_impl): row at set position a reaches columns < hi_b0 + col_hi[a - hi_a0]
	uint32_t hi_n;                     // entries of col_hi (the region's rows)
	uint32_t list_zone;                // pairs with both set positions below it belong to the carrier-list merge pass (ld_list.hip.h)
	uint32_t probe_zone;               // <= list_zone: every other pair of a row below it belongs to the probe pass
	double two_n, cut;                 // 2N; minR2 * (1 - 1e-6)
	uint32_t* cand; unsigned long long cap;            // [cap][3]: set position A, set position B, AA  (unphased form: [cap][6]: ..., HH, HQ, QH, QQ)
	unsigned long long* n_cand;        // device counter of the slots handed out (may run past cap: the host then redoes the tile the plain way)
	uint32_t chunk;                    // slots a wave reserves at a time (0: exactly what a tile needs, one atomic per wave and tile)
	// The FP32 prefilter's per-variant terms, computed once per run (k_screen_terms) instead of per lane and tile: terms[pos] = {a / T, sqrt(cut
	// a (T - a)) / T, b, sqrt(b (T - b)) (1 - 2^-16)} for the variant at set position pos in its role as a row / as a column (a = b = its
	// ALT allele count; UnphasedMath: its dosage h + 2 q); +inf in the square roots' place where a (T - a) = 0 - a variant that is fixed, or
	// a padding row: no pair of it can pass the exact test (its D is exactly 0), and +inf keeps it from passing this one.
	const float4* terms;
	float slack;                       // 0.5 + T 2^-20 counts: four times the worst rounding of the prefilter's left side
};
// -> terms[] of ScreenWork for n_pos set positions (P = 1: rowpop[pos]; P = 2: rowpop[2 pos] + 2 rowpop[2 pos + 1])
__global__ void k_screen_terms(const uint32_t* __restrict__ rowpop, uint32_t n_pos, int P, double two_n, double cut, float4* __restrict__ terms) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_pos) return;
	const uint32_t a = P == 1 ? rowpop[i] : rowpop[2 * i] + 2u * rowpop[2 * i + 1];
	const float T = (float)two_n, af = (float)a, w = af * (T - af);
	const float inf = __builtin_inff();
	float4 t;
	t.x = af / T;
	t.y = w > 0.0f ? sqrtf((float)cut * w) / T * (1.0f - 1.0f / 65536.0f) : inf;
	t.z = af;
	t.w = w > 0.0f ? sqrtf(w) * (1.0f - 1.0f / 65536.0f) : inf;
	terms[i] = t;
}
// Slots are handed out through one counter.  One atomic per wave and tile is 3.9 M atomics on one address in the unphased
// 2,504-sample window run - at the ~12 ns an L2 channel takes for each, as long as the whole kernel (48 ms where the same
// tiles with a cut-off few pairs pass take 33).  So a wave reserves `chunk` slots at a time and numbers its candidates
// from that window (SlotWindow, kept across the units it runs); what is left of its last window when the kernel ends is
// marked CAND_UNUSED in the first word, and the list kernels skip such slots.
constexpr uint32_t CAND_UNUSED = 0xFFFFFFFFu;
// -> first slot of `total` consecutive ones for this wave (wave-uniform; all lanes active).  W = words per slot.
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
template <int W>
__device__ __forceinline__ void mark_unused(const ScreenWork& s, unsigned long long first, uint32_t n, int lane) {
	typedef __attribute__((address_space(1))) uint32_t g_u32;
	g_u32* const cand = (g_u32*)s.cand;
	for (uint32_t i = (uint32_t)lane; i < n; i += 64) { const unsigned long long k = first + i; if (k < s.cap) cand[k * W] = CAND_UNUSED; }
}
// A tile's candidates go into what is left of the wave's window first and into a new reservation behind that, so nothing
// is given up when a window runs out; a reservation is four tiles' worth at the current density, `chunk` at most.
struct SlotRange {
	unsigned long long first, second; uint32_t n_first;      // candidate i of the wave: first + i for i < n_first, else second + (i - n_first)
	__device__ __forceinline__ unsigned long long operator[](uint32_t i) const { return i < n_first ? first + i : second + (i - n_first); }
};
__device__ __forceinline__ SlotRange reserve_slots(const ScreenWork& s, SlotWindow& win, uint32_t total, int lane) {
	typedef __attribute__((address_space(1))) unsigned long long g_u64;
	const unsigned long long next = (unsigned long long)uniform(win.w[0]) | (unsigned long long)uniform(win.w[1]) << 32;
	const uint32_t left = uniform(win.w[2]);
	SlotRange r{next, 0, total};
	unsigned long long new_next = next + total; uint32_t new_left = left - total;
	if (total > left) {
		const uint32_t need = total - left;
		const uint32_t want = 4 * total < s.chunk ? 4 * total : s.chunk;
		const uint32_t take = want > need ? want : need;
		unsigned long long b = 0;
		if (lane == 0) b = __hip_atomic_fetch_add((g_u64*)s.n_cand, (unsigned long long)take, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		r.second = (unsigned long long)uniform((uint32_t)b) | (unsigned long long)uniform((uint32_t)(b >> 32)) << 32;
		r.n_first = left;
		new_next = r.second + need; new_left = take - need;
	}
	if (lane == 0) { win.w[0] = (uint32_t)new_next; win.w[1] = (uint32_t)(new_next >> 32); win.w[2] = new_left; }
	return r;
}
template <int W>
__device__ __forceinline__ void release_slots(const ScreenWork& s, SlotWindow& win, int lane) {
	const unsigned long long next = (unsigned long long)uniform(win.w[0]) | (unsigned long long)uniform(win.w[1]) << 32;
	mark_unused<W>(s, next, uniform(win.w[2]), lane);
}
template <int TB>
struct ScreenCounts {
	static constexpr bool PAIRED_ROWS = false;
	static constexpr bool THREE_PRODUCTS = false;
	static constexpr bool QUEUED = true;       // the wave's queue of prefilter-passing pairs (SlotWindow::q)
	// staged per tile: [0, 2 TILE) the prefilter's (a / T, sA) of its 128 rows, [2 TILE, 4 TILE) (b, sB) of its 128 columns (ScreenWork::terms),
	// then - for the exact test - the allele counts of the rows, of the columns, and the rows' band limits
	static constexpr int META_WORDS = 7 * TILE;
	static constexpr int M_COLF = 2 * TILE, M_ROWS = 4 * TILE, M_COLS = 5 * TILE, M_REACH = 6 * TILE;
	const ScreenWork* sp;              // in device memory: read where it is needed, not held in registers through the K loop
	// where word i of the staged block comes from (always a readable address; what lies outside the region is masked in the epilogue)
	__device__ __forceinline__ const uint32_t* meta_src(uint32_t yx, uint32_t i) const {
		const ScreenWork& s = *sp;
		const uint32_t tA = s.a0 + (yx >> 16) * TILE, tB = s.b0 + (yx & 0xFFFFu) * TILE;
		if (i < (uint32_t)M_COLF) return reinterpret_cast<const uint32_t*>(s.terms + tA + (i >> 1)) + (i & 1u);
		if (i < (uint32_t)M_ROWS) return reinterpret_cast<const uint32_t*>(s.terms + tB + ((i - M_COLF) >> 1)) + 2 + (i & 1u);
		const uint32_t r = i & (TILE - 1);
		if (i < (uint32_t)M_COLS) return s.rowpop + tA + r;
		if (i < (uint32_t)M_REACH) return s.rowpop + tB + r;
		if (!s.col_hi) return s.rowpop;
		const uint32_t k = tA + r - s.hi_a0;
		return s.col_hi + (k < s.hi_n ? k : s.hi_n - 1);
	}
	__device__ __forceinline__ void finish(SlotWindow& win, int lane) const { release_slots<3>(*sp, win, lane); }
	__device__ __forceinline__ void operator()(uint32_t (&acc)[8][TB], uint32_t yx, int wr, int wc, int li, int lj, int lane, bool, const uint32_t* meta, SlotWindow& win) const {
		const ScreenWork& s = *sp;
		// Round 6: the FP32 prefilter in front of everything, on terms computed once per run (ScreenWork::terms) - per pair a convert, two
		// fused multiply-adds and a compare:
		//     |AA - a' b| >= sA sB - slack          a' = a / T, sA = sqrt(cut a (T - a)) / T, sB = sqrt(b (T - b)) (1 - 2^-16)
		// (AA, a, b <= 2N exact in FP32 below 2^24; slack = 0.5 + T 2^-20 counts is four times the worst rounding of the left side: no pair the
		// exact test below passes fails this one).  It holds for ANY tile - what the structural tests take away they take away from the pairs
		// that pass - and rows beyond the matrix or of fixed variants carry sA = +inf.  No lane of the wave with such a pair - the rule for
		// unlinked variants - and the wave is done: no structural test, no mask, no parameter block read (round 5 took this way out for
		// interior tiles only, and spent a third of it on the square roots that are now a table).
		// The wave has pairs to look at - where variants are in LD, a handful among its 2,048.  Round 5 ran the structural tests and the FP64
		// screen for all thirty-two pairs of every lane then; slot by slot behind wave-uniform branches it costs the same (a wave executes a
		// slot's test for one lane's sake).  Instead the lanes QUEUE their flagged pairs - (lane, slot, count), one LDS atomic each, in the wave's
		// own 64-entry queue - and the wave then tests the queue's entries one per lane: one pass of the exact test per tile instead of thirty-two.
		// A tile with more flagged pairs than the queue holds (dense LD next to the diagonal) takes round 5's way.
		const uint32_t r0 = (yx >> 16) * TILE + wr * 64 + li;             // this lane's rows: r0 + 8t
		const uint32_t c0 = (yx & 0xFFFFu) * TILE + wc * 8 * TB + lj;     // and columns: c0 + 8u
		const double two_n = s.two_n, cut = s.cut;
		const uint32_t a0 = s.a0, b0 = s.b0;
		const bool diag = s.diag != 0, banded = s.col_hi != nullptr;
		const uint32_t zone = s.list_zone, pzone = s.probe_zone, nA = s.nA, nB = s.nB, n_variants = s.n_variants, hi_b0 = s.hi_b0;
		typedef __attribute__((address_space(3))) uint32_t l_u32;
		// (win.w[5]: the wave's previous tile overflowed the queue - dense LD comes in runs of tiles, and queueing that fails is paid on top of
		// the full test; the full test clears the flag again when it finds a tile the queue would have held)
		const bool try_queue = uniform(win.w[5]) == 0;
		uint32_t n_q = SCREEN_QUEUE + 1;
		if (try_queue) {
			if (lane == 0) win.w[4] = 0;
			const float slack = s.slack;
#pragma unroll
			for (int t = 0; t < 8; ++t) {
				const float2 fa = *reinterpret_cast<const float2*>(meta + 2 * (wr * 64 + li + 8 * t));
#pragma unroll
				for (int u = 0; u < TB; ++u) {
					const float2 fb = *reinterpret_cast<const float2*>(meta + M_COLF + 2 * (wc * 8 * TB + lj + 8 * u));
					if (__builtin_fabsf(__builtin_fmaf(-fa.x, fb.x, (float)acc[t][u])) >= __builtin_fmaf(fa.y, fb.y, -slack)) {
						const uint32_t at = __hip_atomic_fetch_add((l_u32*)&win.w[4], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
						if (at < (uint32_t)SCREEN_QUEUE) win.q[at] = make_uint2(((uint32_t)lane << 8) | (uint32_t)(4 * t + u), acc[t][u]);
					}
				}
			}
			n_q = uniform(win.w[4]);
			if (n_q > (uint32_t)SCREEN_QUEUE && lane == 0) win.w[5] = 1;
		}
		if (n_q <= (uint32_t)SCREEN_QUEUE) {
			bool any_ok = false;
			for (uint32_t base = 0; base < n_q; base += 64) {          // (wave-uniform: one pass, two for a tile with more than 64 flagged pairs)
				bool ok = false; uint32_t eA = 0, eB = 0, eN = 0;
				if (base + (uint32_t)lane < n_q) {
					const uint2 e = win.q[base + lane];
					const uint32_t src = e.x >> 8, slot = e.x & 31u, t = slot >> 2, u = slot & 3u;
					const uint32_t tr = (uint32_t)wr * 64 + (src >> 3) + 8 * t, tc = (uint32_t)wc * 8 * TB + (src & 7u) + 8 * u;       // row / column within the tile
					const uint32_t rt = (yx >> 16) * TILE + tr, cu = (yx & 0xFFFFu) * TILE + tc;
					const uint32_t sA = a0 + rt, sB = b0 + cu;
					eA = sA; eB = sB; eN = e.y;
					if (rt < nA && sA < n_variants && cu < nB && sB < n_variants) {
						uint32_t h = b0 + nB < n_variants ? b0 + nB : n_variants;
						if (banded) { const uint32_t hb = hi_b0 + meta[M_REACH + tr]; h = hb < h ? hb : h; }
						const double a = (double)meta[M_ROWS + tr], b = (double)meta[M_COLS + tc];
						const double dn = two_n * (double)e.y - a * b;
						ok = (!diag || sB > sA) && sB < h && !((sA < zone && sB < zone) || sA < pzone) && dn != 0.0 && dn * dn >= (cut * (a * (two_n - a))) * (b * (two_n - b));
					}
				}
				const unsigned long long votes = __ballot(ok);
				if (votes) {
					typedef __attribute__((address_space(1))) uint32_t g_u32;
					any_ok = true;
					const uint32_t total = (uint32_t)__popcll(votes);
					const SlotRange slots = reserve_slots(s, win, total, lane);
					if (ok) {
						const unsigned long long slot = slots[(uint32_t)__popcll(votes & ((1ull << lane) - 1ull))];
						if (slot < s.cap) { g_u32* o = (g_u32*)s.cand + slot * 3; o[0] = eA; o[1] = eB; o[2] = eN; }
					}
				}
			}
			if (lane == 0) win.w[3] = any_ok ? 1u : 0u;
#pragma unroll
			for (int t = 0; t < 8; ++t)
#pragma unroll
				for (int u = 0; u < TB; ++u) acc[t][u] = 0;
			return;
		}
		// allele counts of the lane's rows and columns; hiA: first column the row does not reach (0 for a row outside the
		// tile's variants: reaches nothing) - the columns outside the tile's variants lie beyond every row's reach
		uint32_t acB[TB], acA[8], hiA[8];
#pragma unroll
		for (int u = 0; u < TB; ++u) {
			const uint32_t cu = c0 + 8 * u, sB = b0 + cu;
			acB[u] = (cu < nB && sB < n_variants) ? meta[M_COLS + wc * 8 * TB + lj + 8 * u] : 0u;
		}
#pragma unroll
		for (int t = 0; t < 8; ++t) {
			const uint32_t rt = r0 + 8 * t, sA = a0 + rt;
			const bool okr = rt < nA && sA < n_variants;
			acA[t] = okr ? meta[M_ROWS + wr * 64 + li + 8 * t] : 0u;
			// first column the row does not reach: the end of the tile's columns, of the matrix, of the row's r2 band
			uint32_t h = b0 + nB < n_variants ? b0 + nB : n_variants;
			if (banded) { const uint32_t hb = hi_b0 + meta[M_REACH + wr * 64 + li + 8 * t]; h = hb < h ? hb : h; }
			hiA[t] = okr ? h : 0u;
		}
		uint32_t m = 0;                  // bit 4t + u: pair (t, u) is a candidate
#pragma unroll
		for (int t = 0; t < 8; ++t) {
			const uint32_t sA = a0 + r0 + 8 * t;
			const double a = (double)acA[t];
			const double fA = cut * (a * (two_n - a));
#pragma unroll
			for (int u = 0; u < TB; ++u) {
				const uint32_t sB = b0 + c0 + 8 * u;
				const double b = (double)acB[u];
				const double dn = two_n * (double)acc[t][u] - a * b;
				const bool okp = (!diag || sB > sA) && sB < hiA[t] && !((sA < zone && sB < zone) || sA < pzone) && dn != 0.0 && dn * dn >= fA * (b * (two_n - b));
				m |= (okp ? 1u : 0u) << (4 * t + u);
			}
		}
		const bool wave_has = __ballot(m != 0) != 0;
		if (lane == 0) win.w[3] = wave_has ? 1u : 0u;
		{
			const uint32_t cnt = __popc(m);
			const uint32_t incl = wave_scan_inclusive(cnt);
			const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
			if (lane == 0) win.w[5] = total > (uint32_t)(SCREEN_QUEUE / 4) ? 1u : 0u;      // (the prefilter flags a few times the pairs that pass: back to the queue where the tile held few)
			if (wave_has) {
			// (the pointers come out of the parameter block as generic addresses: say that they are global memory, or
			// every candidate store becomes a flat store that also waits on the LDS queue)
			typedef __attribute__((address_space(1))) uint32_t g_u32;
			const SlotRange slots = reserve_slots(s, win, total, lane);
			uint32_t mine = incl - cnt;                  // this lane's next candidate, counted over the wave
			g_u32* const cand = (g_u32*)s.cand; const unsigned long long cap = s.cap;
#pragma unroll
			for (int t = 0; t < 8; ++t)
#pragma unroll
				for (int u = 0; u < TB; ++u)
					if ((m >> (4 * t + u)) & 1u) {
						const unsigned long long slot = slots[mine];
						if (slot < cap) {
							g_u32* e = cand + slot * 3;
							e[0] = a0 + r0 + 8 * t; e[1] = b0 + c0 + 8 * u; e[2] = acc[t][u];
						}
						++mine;
					}
			}
		}
#pragma unroll
		for (int t = 0; t < 8; ++t)
#pragma unroll
			for (int u = 0; u < TB; ++u) acc[t][u] = 0;
	}
};
static_assert(16 / (8 / 2) == 4, "ScreenCounts packs (t, u) as 4t + u: TB = 4");

// ---- the same for the unphased planes (two rows per variant: H = het, Q = hom-alt; no missing data) -----------
// A variant pair has four products HH, HQ, QH, QQ.  With the plain row assignment (a lane's rows li + 8t) they sit in four
// lanes, and round 3 brought them together with three DPP moves per slot - 96 moves and as many selects per lane and tile,
// twice when the tile had candidates: 9 % of a 3-chunk tile before a single candidate was stored.  Round 4 gives this kernel
// its own row assignment instead (PAIRED_ROWS, see read_half): a lane's A rows are 2 li + (t & 1) + 16 (t >> 1) and its B
// rows 2 lj + (u & 1) + 16 (u >> 1) - the H and the Q plane of the same four row variants li + 8 s and the same two column
// variants lj + 8 v - so that
//     HH = acc[2s][2v],  HQ = acc[2s][2v + 1],  QH = acc[2s + 1][2v],  QQ = acc[2s + 1][2v + 1]
// are the lane's own registers: no DPP, no selects, eight variant pairs per lane as before.  (The LDS image does not change;
// the slot swizzle of such a row is li for every t, still eight distinct bank positions per read.)
//
// The screen (UnphasedMath, ld_engine.cpp:1312-1560; d_unphased_math has the same test in front of its cubic): every root the
// reference may keep lies in [minhap - 1e-5, maxhap + 1e-5], minhap = n11 / 2N, maxhap = (n11 + HH) / 2N with
// n11 = 2 (0/0,0/0) + (0/0,het) + (het,0/0) = (2N - h_A - 2 q_A) - (h_B + 2 q_B) + QH + HQ + 2 QQ, and D = f11 - P Q,
// r2 = D^2 / (P (1 - P) Q (1 - Q)) with the REF frequencies P = 1 - (h_A + 2 q_A) / 2N, Q likewise: the pair can pass only if
// one end of that interval reaches the cut-off.  For unlinked variants the interval is centred on P Q with half width
// P (1 - P) Q (1 - Q), which is below any cut-off above 1/16: at the default r2 >= 0.1 only pairs in LD are candidates.
// THREE: the three-product contraction (contract3_chunk) - the lane holds HH and S = QH + HQ + 2 QQ of a pair, which is all the
// screen reads; a candidate's entry then carries (A, B, HH, S, -, -) and k_recount_unphased fills in the four products.
template <int TB, bool THREE = false>
struct ScreenCountsUnphased {
	static constexpr bool PAIRED_ROWS = true;
	static constexpr bool THREE_PRODUCTS = THREE;
	static constexpr bool QUEUED = false;
	// staged per tile: the prefilter's (da / T, sA) of its 64 row variants and (db, sB) of its 64 column variants (ScreenWork::terms), then - for the
	// exact test - the H / Q counts of its 128 plane rows and of its 128 plane columns, and the band limits of its 64 row variants
	static constexpr int U_COLF = TILE, U_ROWS = 2 * TILE, U_COLS = 3 * TILE, U_REACH = 4 * TILE;
	static constexpr int META_WORDS = 4 * TILE + TILE / 2;
	const ScreenWork* sp;
	__device__ __forceinline__ const uint32_t* meta_src(uint32_t yx, uint32_t i) const {
		const ScreenWork& s = *sp;
		const uint32_t vA = s.a0 + (yx >> 16) * (TILE / 2), vB = s.b0 + (yx & 0xFFFFu) * (TILE / 2);       // the tile's first row / column variant
		if (i < (uint32_t)U_COLF) return reinterpret_cast<const uint32_t*>(s.terms + vA + (i >> 1)) + (i & 1u);
		if (i < (uint32_t)U_ROWS) return reinterpret_cast<const uint32_t*>(s.terms + vB + ((i - U_COLF) >> 1)) + 2 + (i & 1u);
		if (i < (uint32_t)U_COLS) return s.rowpop + 2 * vA + (i - U_ROWS);
		if (i < (uint32_t)U_REACH) return s.rowpop + 2 * vB + (i - U_COLS);
		if (!s.col_hi) return s.rowpop;
		const uint32_t k = vA + (i - U_REACH) - s.hi_a0;
		return s.col_hi + (k < s.hi_n ? k : s.hi_n - 1);
	}
	__device__ __forceinline__ void finish(SlotWindow& win, int lane) const { release_slots<6>(*sp, win, lane); }
	__device__ __forceinline__ void operator()(uint32_t (&acc)[8][TB], uint32_t yx, int wr, int wc, int li, int lj, int lane, bool, const uint32_t* meta, SlotWindow& win) const {
		static_assert(TB == 4, "two column variants per lane, two planes each");
		const ScreenWork& s = *sp;
		// set positions (variants) of the lane's pairs: rows vA0 + 8 sI (sI = 0..3), columns vB0 + 8 v (v = 0, 1)
		const uint32_t vA0 = s.a0 + (yx >> 16) * (TILE / 2) + wr * 32 + li;
		const uint32_t vB0 = s.b0 + (yx & 0xFFFFu) * (TILE / 2) + wc * (4 * TB) + lj;
		// The screen without a division: with T = 2N, a / b the ALT allele counts and ra = T - a, rb = T - b the REF counts,
		//     (f11 - P Q) T^2 = n11 T - ra rb   at f11 = minhap,   (n11 + HH) T - ra rb   at f11 = maxhap
		// (integers, exact in FP64), the admissibility slack 1e-5 becomes 1e-5 T^2, and r2 >= cut reads
		//     ((f11 - P Q) T^2)^2 >= cut a ra b rb.
		const double T2n = s.two_n, cut = s.cut, eps = 1e-5 * (T2n * T2n);
		const bool diag = s.diag != 0;
		// Prefilter in FP32 (round 5; round 6: on terms computed once per run, ScreenWork::terms).  Divided by T the two ends of the interval read
		//     e_lo / T = S - da db / T - eps',   e_hi / T = e_lo / T + HH + 2 eps'      (eps' = 1e-5 T; n11 - ra rb / T = S - da db / T),
		// and the pair can pass only if max(e_hi, -e_lo) / T >= sqrt(cut da ra db rb) / T = sA sB: a convert or two, two fused multiply-adds, two adds,
		// a max and a compare per pair.  S, HH, da, db <= 2N are exact in FP32 below 2^24; the rest is covered by `slack` as in ScreenCounts.
		// Only pairs the prefilter lets through reach the exact FP64 test, so the candidates are those of the exact test alone.
		const float Tf = (float)T2n, epsf = 1e-5f * Tf, slack = s.slack;
		float2 fa[4], fb[2];
#pragma unroll
		for (int sI = 0; sI < 4; ++sI) fa[sI] = *reinterpret_cast<const float2*>(meta + 2 * (wr * 32 + li + 8 * sI));
#pragma unroll
		for (int v = 0; v < 2; ++v) fb[v] = *reinterpret_cast<const float2*>(meta + U_COLF + 2 * (wc * (4 * TB) + lj + 8 * v));
		// Way out, for any tile (round 5: interior tiles only - the structural tests can only take pairs away, and variants beyond the matrix or
		// fixed ones carry sA = +inf): one FP32 test per pair tells whether the lane has a candidate at all, and a wave without one is done.
		if (uniform(win.w[3]) == 0) {      // (see ScreenCounts: not behind a tile that had candidates)
			bool any = false;
#pragma unroll
			for (int sI = 0; sI < 4; ++sI)
#pragma unroll
				for (int v = 0; v < 2; ++v) {
					const uint32_t hh = acc[2 * sI][2 * v];
					const uint32_t s_sum = THREE ? acc[2 * sI + 1][2 * v + 1] : acc[2 * sI + 1][2 * v] + acc[2 * sI][2 * v + 1] + 2u * acc[2 * sI + 1][2 * v + 1];
					const float q = __builtin_fmaf(-fa[sI].x, fb[v].x, (float)s_sum);
					any |= __builtin_fmaxf(q + ((float)hh + epsf), epsf - q) >= __builtin_fmaf(fa[sI].y, fb[v].y, -slack);
				}
			if (!__ballot(any)) {
#pragma unroll
				for (int t = 0; t < 8; ++t)
#pragma unroll
					for (int u = 0; u < TB; ++u) acc[t][u] = 0;
				return;
			}
		}
		// everything else the lane needs of the staged block in one go
		uint32_t rawA[4][2], rawH[4];
#pragma unroll
		for (int sI = 0; sI < 4; ++sI) { const int rowA = U_ROWS + wr * 64 + 2 * li + 16 * sI; rawA[sI][0] = meta[rowA]; rawA[sI][1] = meta[rowA + 1]; rawH[sI] = meta[U_REACH + wr * 32 + li + 8 * sI]; }
		const bool banded = s.col_hi != nullptr;
		const uint32_t hi_b0 = s.hi_b0, endA = s.a0 + s.nA, endB = s.b0 + s.nB, n_variants = s.n_variants, list_zone = s.list_zone, probe_zone = s.probe_zone;
		uint32_t vB[2];
#pragma unroll
		for (int v = 0; v < 2; ++v) {
			vB[v] = vB0 + 8 * v;
			if (!(vB[v] < endB && vB[v] < n_variants)) vB[v] = 0xFFFFFFFFu;              // (beyond every row's reach)
		}
		uint32_t mp = 0;                 // bit 2 sI + v: the pair passes the prefilter and the structural tests
#pragma unroll
		for (int sI = 0; sI < 4; ++sI) {
			const uint32_t vA = vA0 + 8 * sI;
			const bool okA = vA < endA && vA < n_variants;
			uint32_t hi = banded ? hi_b0 + rawH[sI] : 0xFFFFFFFFu;
			if (!okA || vA < probe_zone) hi = 0;
			uint32_t lo = diag ? vA + 1 : 0u;
			if (vA < list_zone && lo < list_zone) lo = list_zone;
			const uint32_t width = hi > lo ? hi - lo : 0u;
#pragma unroll
			for (int v = 0; v < 2; ++v) {
				const uint32_t hh = acc[2 * sI][2 * v];
				const uint32_t s_sum = THREE ? acc[2 * sI + 1][2 * v + 1] : acc[2 * sI + 1][2 * v] + acc[2 * sI][2 * v + 1] + 2u * acc[2 * sI + 1][2 * v + 1];      // QH + HQ + 2 QQ
				const float q = __builtin_fmaf(-fa[sI].x, fb[v].x, (float)s_sum);      // S - da db / T
				const float e = __builtin_fmaxf(q + ((float)hh + epsf), epsf - q);
				const bool ok = (vB[v] - lo) < width && e >= __builtin_fmaf(fa[sI].y, fb[v].y, -slack);
				mp |= (ok ? 1u : 0u) << (2 * sI + v);
			}
		}
		uint32_t m = 0;                  // bit 2 sI + v: the pair (row variant sI, column variant v) is a candidate
		if (__ballot(mp != 0)) {
#pragma unroll
			for (int sI = 0; sI < 4; ++sI) {
				if (!((mp >> (2 * sI)) & 3u)) continue;
				const double da = (double)(rawA[sI][0] + 2u * rawA[sI][1]), ra = T2n - da;
				const double fA = cut * (da * ra);
#pragma unroll
				for (int v = 0; v < 2; ++v) {
					if (!((mp >> (2 * sI + v)) & 1u)) continue;
					const uint32_t hh = acc[2 * sI][2 * v];
					const uint32_t s_sum = THREE ? acc[2 * sI + 1][2 * v + 1] : acc[2 * sI + 1][2 * v] + acc[2 * sI][2 * v + 1] + 2u * acc[2 * sI + 1][2 * v + 1];
					// n11 = ra - b + (QH + HQ + 2 QQ): the (REF, REF) haplotypes that are certain
					const double dbv = (double)fb[v].x, rbv = T2n - dbv;
					const double n11 = (ra - dbv) + (double)s_sum;
					const double e_lo = (n11 * T2n - ra * rbv) - eps;
					const double e_hi = ((n11 + (double)hh) * T2n - ra * rbv) + eps;
					const double bound = fA * (dbv * rbv);
					const bool ok = !(e_lo * e_lo < bound && e_hi * e_hi < bound);
					m |= (ok ? 1u : 0u) << (2 * sI + v);
				}
			}
		}
		const bool wave_has = __ballot(m != 0) != 0;
		if (lane == 0) win.w[3] = wave_has ? 1u : 0u;
		if (wave_has) {
			typedef __attribute__((address_space(1))) uint32_t g_u32;
			const uint32_t cnt = __popc(m);
			const uint32_t incl = wave_scan_inclusive(cnt);
			const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
			const SlotRange slots = reserve_slots(s, win, total, lane);
			uint32_t mine = incl - cnt;
			g_u32* const cand = (g_u32*)s.cand; const unsigned long long cap = s.cap;
#pragma unroll
			for (int sI = 0; sI < 4; ++sI)
#pragma unroll
				for (int v = 0; v < 2; ++v)
					if ((m >> (2 * sI + v)) & 1u) {
						const unsigned long long slot = slots[mine];
						if (slot < cap) {
							g_u32* e = cand + slot * 6;
							e[0] = vA0 + 8 * sI; e[1] = vB[v];
							e[2] = acc[2 * sI][2 * v]; e[3] = acc[2 * sI][2 * v + 1]; e[4] = acc[2 * sI + 1][2 * v]; e[5] = acc[2 * sI + 1][2 * v + 1];      // (THREE: HH, 0, 0, S - replaced by k_recount_unphased)
						}
						++mine;
					}
		}
#pragma unroll
		for (int t = 0; t < 8; ++t)
#pragma unroll
			for (int u = 0; u < TB; ++u) acc[t][u] = 0;
	}
};

// The phased form copies the screen's parameter block into LDS first.  It lives in device memory (in SGPRs through the K
// loop it cost 72 spills), and read from there - by vector loads, again after every hand-written asm block, with a memory
// round trip in front of the meta staging of every unit and of every epilogue - it cost the survivor-rich 2,504-sample run
// 4 % of the kernel (19.4 -> 18.6 ms).  The unphased form is at its register limit: there the copy's addresses cost three
// spills that land in the candidate loop (48 -> 115 ms on the same run), so it reads the block where it is.
static_assert(sizeof(ScreenWork) % 4 == 0 && sizeof(ScreenWork) / 4 <= 64, "copied by the first wave, one dword per lane");
template <int NW, int EXPERIMENT = 0>      // (EXPERIMENT == 6: the dev tool's no-epilogue timing)
__global__ __launch_bounds__(NW * 64, NW / 2)
void k_count_screen_t(const CountWork w, const ScreenWork* sw) {
	__shared__ ScreenWork sw_lds;      // (re-measured in round 5 against reading the block where it lies: 84.6 against 84.1 % at five chunks a tile, 76.9 against 75.5 % at three)
	if (threadIdx.x < sizeof(ScreenWork) / 4) reinterpret_cast<uint32_t*>(&sw_lds)[threadIdx.x] = reinterpret_cast<const uint32_t*>(sw)[threadIdx.x];
	__syncthreads();
	count_list_body<NW, EXPERIMENT>(w, ScreenCounts<16 / (NW / 2)>{&sw_lds});
}
template <int NW, int EXPERIMENT = 0>
__global__ __launch_bounds__(NW * 64, NW / 2)
void k_count_screen_unphased_t(const CountWork w, const ScreenWork* sw) {
	count_list_body<NW, EXPERIMENT>(w, ScreenCountsUnphased<16 / (NW / 2)>{sw});
}

template <int NW, int EXPERIMENT = 0>
__global__ __launch_bounds__(NW * 64, NW / 2)
void k_count3_screen_unphased_t(const CountWork w, const ScreenWork* sw) {
	count_list_body<NW, EXPERIMENT>(w, ScreenCountsUnphased<16 / (NW / 2), true>{sw});
}

// Zero the tiles [first, n_tiles) of the list (the ones whose K range is split into several units).  tile_rows: rows of C a
// tile covers - TILE, or TILE / 2 in the three-product form's matrix (StoreCounts3).
__global__ __launch_bounds__(256)
void k_zero_tiles(const uint32_t* __restrict__ tiles, uint32_t first, uint32_t* __restrict__ C, uint32_t ldc, uint32_t tile_rows = TILE) {
	const uint32_t yx = tiles[first + blockIdx.x];
	uint32_t* Cblk = C + (size_t)((yx >> 16) * tile_rows) * ldc + (yx & 0xFFFFu) * TILE;
	const uint4 z = make_uint4(0, 0, 0, 0);
	for (int i = threadIdx.x; i < (int)tile_rows * TILE / 4; i += 256) {
		const int r = i / (TILE / 4), q = i % (TILE / 4);
		*reinterpret_cast<uint4*>(Cblk + (size_t)r * ldc + q * 4) = z;
	}
}

#ifndef TWK_COUNT_NW
#define TWK_COUNT_NW 8
#endif
constexpr int COUNT_NW = TWK_COUNT_NW;
constexpr int COUNT_THREADS = COUNT_NW * 64;

}  // namespace twk
