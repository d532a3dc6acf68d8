// C ABI of the MI355X pairwise-LD engine (include/twk_hip.h).
//
// Host-side orchestration of the three device stages
//   prep  (ld_prep.hip.h)  reference bitvectors -> contraction planes
//   count (ld_count.hip.h) LDS-tiled AND+popcount contraction  [dominant kernel]
//   math  (ld_math.hip.h)  cells -> D/D'/r2/Fisher -> filters -> compaction
//   (ld_three.hip.h: screen + recount behind the three-product form of the unphased contraction)
// over super-tiles of the variant-pair triangle.  Device memory lives in the
// ctx; nothing here falls back to the CPU.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <chrono>
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rccl/rccl.h>          // types only: the library is opened on first use (twk_hip_gather_records), never linked
#include <dlfcn.h>
#include <map>

#include "../../../include/twk_hip.h"
#include "ld_count.hip.h"
#include "ld_prep.hip.h"
#include "ld_math.hip.h"
#include "ld_list.hip.h"
#include "ld_three.hip.h"
#include "ld_plan.h"
#include "twk_delivery.h"

using namespace twk;

namespace {

inline uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }

struct PlaneSet {
	uint32_t* rows = nullptr;     // [rows_alloc][W]
	uint32_t* rowpop = nullptr;   // [rows_alloc]
	uint32_t  W = 0;              // words per row (padded to KC)
	uint32_t  W_live = 0;         // words per row that carry data
	uint32_t  rows_alloc = 0;
	bool      built = false;
	bool      owns_rows = false;
	// regrouped set only: position in the set -> variant id (device + host), and how many
	// variants with missing genotypes lead the set
	uint32_t* ids = nullptr;
	std::vector<uint32_t> h_ids;
	uint32_t  n_front = 0;
	// allele-count-sorted phased set only (ld_list.hip.h): carrier lists of the variants that lead the set with at most
	// list_max carriers of their minor allele - positions [0, n_list), n_list a multiple of the tile edge
	uint32_t* lists = nullptr; uint32_t* list_mac = nullptr; uint32_t* list_flip = nullptr;
	uint32_t  n_list = 0, list_max = 0;
	uint32_t  n_probe = 0;         // <= n_list: the leading variants whose lists are short enough for probing to beat the dense pair (ld_list.hip.h)
	// fused screen kernels: the prefilter's per-variant terms at the cut-off of the run that built them (ScreenWork::terms, k_screen_terms)
	float4*   terms = nullptr; double terms_cut = -1.0;
};

// Plane sets a context can hold: one per PlaneKind in file order, plus the masked unphased planes
// regrouped so that the variants with missing genotypes come first (both groups in file order).
// Default-mode runs (reference: phased math unless either variant has missing data)
// then need the expensive 3-plane products only for the pairs that involve the leading group:
// a triangle over it plus its rectangle against the rest.
// ... and (TWK_HIP_OPT_R2_SCREEN) the plain phased / unphased planes in order of minor allele count, variants with
// missing genotypes last: in that order the pairs whose r2 can reach the cut-off at all - a bound from the two
// allele counts alone - are a band along the diagonal, and whole tiles outside it are never contracted.
enum { PS_GROUPED = 4, PS_SORTED_P = 5, PS_SORTED_U = 6, N_PLANE_SETS = 7 };
inline int set_kind(int set) {
	return set == PS_GROUPED ? (int)PK_UNPHASED_MASKED : set == PS_SORTED_P ? (int)PK_PHASED : set == PS_SORTED_U ? (int)PK_UNPHASED : set;
}
// internal modes of the two stages of such a run
enum { MODE_INT_AUTO_CLEAN = 0x10,     // phased math on the plain planes; pairs without missing data only
       MODE_INT_GROUPED    = 0x11,     // unphased math on the regrouped planes; every pair of the tile
       MODE_INT_SORTED_P   = 0x12,     // phased math on the allele-count-sorted planes (variants without missing data)
       MODE_INT_SORTED_U   = 0x13 };   // unphased math on the allele-count-sorted planes

constexpr int N_SLOT_COUNTERS = 16;      // (see Slot::n_out)
// Launches of a region call in flight.  Three, not two: the survivors of launch t are sorted on the copy stream, where they
// wait for a CU until the persistent count kernel of launch t + 1 lets go; with only t + 1 enqueued the device then idled
// until the host had sorted, copied and handed over launch t and come back with launch t + 2 (2,504 x 531,500, all pairs:
// 39 launches of 13 ms took 1.25 s).  With t + 2 already queued the count kernels run back to back.
constexpr int PIPE_SLOTS = 3, SYNC_SLOT = PIPE_SLOTS;
struct Slot {                      // one in-flight tile (double buffered)
	uint32_t* C = nullptr; size_t C_words = 0;
	twk_hip_record* out = nullptr; unsigned long long capacity = 0;      // survivor buffer and its size (grow-only)
	unsigned long long* keys = nullptr; uint32_t* vals = nullptr;        // [capacity]: sort key and position of every survivor, written where it is appended
	unsigned long long cap_use = 0;               // ... of which the current launch may use this many (what the caller asked for)
	unsigned long long* n_out = nullptr;          // device counters: [0] survivors appended, [1] of those dropped by the Fisher cut-off,
	                                              // [2] candidates of the fused count kernel, [3] three-product candidates whose recount disagrees, [4] shader cycles and
	                                              // [5] 100 MHz ticks the count kernel's blocks lived for (summed over the blocks), [6], [7] spare, [8 + x] the 100 MHz
	                                              // tick at which the last block on XCD x finished
	unsigned long long* h_n_out = nullptr;        // pinned host copy of all of them
	hipEvent_t ev_c0 = nullptr, ev_c1 = nullptr, ev_s1 = nullptr, ev_c0b = nullptr, ev_c1b = nullptr;
	bool two_pass = false;
	bool is_list = false;                         // the launch was a carrier-list pass (ld_list.hip.h): C holds its candidate list
	bool is_probe = false;                        // ... of the probe kind (zone rows x columns outside the zone)
	// a band launch (region_impl): its pair math is enqueued once its candidate count is known (enqueue_band_math)
	bool deferred = false; bool deferred_unphased = false;
	bool was_deferred = false;                    // (this slot's launch was one: its math runs between ev_c0b and ev_s1, not right behind the count kernel)
	StatsParams* d_stats_dev = nullptr;           // the math kernel's parameter block on the device (behind the launch's tile list)
	StatsParams stats_host;                       // ... and the host's copy, patched with the survivor buffer before it is sent again
	twk_hip_record* sorted = nullptr; unsigned long long sorted_cap = 0;      // band launches: the survivors in (idxA, idxB) order, sorted on the
	bool presorted = false;                                                   // compute stream right behind Fisher's test (enqueue_band_math)
	bool fused = false;                           // first launch ran the fused count -> screen kernel: C holds the candidate list
	bool three = false;                           // ... in the three-product form (HH + S; the candidates' four products are recounted): fused, or
	bool three_plain = false;                     // through a count matrix (long rows): C holds the (HH, S) matrix and, behind it, the candidate list
	uint32_t* cand = nullptr;                     // the candidate list of the launch (in C)
	int plane_set = 0;                            // the plane set the launch contracted
	unsigned long long cand_cap = 0;              // ... of this many entries; n_out[2] counts them
	bool cand_overflow = false;                   // set by finish_tile: the list did not hold them all
	bool band_too_big = false;                    // a band launch whose candidates need more survivor / sort buffers than it may have (or could get): finish_tile reports
	                                              // it as an overflow and the launch's rows are redone as matrix-sized tiles
	double minP = 1.0;
	uint64_t row_pairs = 0, row_pairs_b = 0;
	// work lists of the (up to two) count launches of the tile: pinned host copy + device copy
	uint32_t* h_tiles[2] = {nullptr, nullptr}; uint32_t* d_tiles[2] = {nullptr, nullptr}; size_t tiles_cap[2] = {0, 0};
};

// Window mode: row variant a0 + r of a region reaches the columns [b0 + lo[r], b0 + hi[r]).
struct ColRange { const uint32_t* lo = nullptr; const uint32_t* hi = nullptr; uint32_t a0 = 0, b0 = 0;
                  const uint32_t* d_hi = nullptr; uint32_t n_hi = 0;      // d_hi: device copy of hi, n_hi entries (r2 screen: the math kernel skips what was not contracted)
                  uint32_t list_zone = 0;              // pairs with both set positions below it are intersected as carrier lists (ld_list.hip.h), not contracted
                  uint32_t probe_zone = 0; };          // <= list_zone: every other pair of a row below it is decided by probing the column's row with the row's carriers (k_probe_screen)

}  // namespace

namespace {
// Measurement and test switches (twk_hip_set_option): what used to be TWK_HIP_* environment variables.  The library
// reads no environment variable; a host that embeds it gets the defaults below unless it says otherwise.
// ONE table - X(key, default, lowest, highest, drops the derived plane sets, meaning): the Options struct, the key table of
// twk_hip_set_option / twk_hip_get_option and - through twk_hip_option_describe - the table of INTEGRATION.md 1 all come from it
// (tests/test_docs_consistency.py holds the document to it).
#define TWK_HIP_OPTIONS(X) \
	X(fused, 1, 0, 2, false, "fused count -> r2 screen kernel (no count matrix; DESIGN 3.2a): 0 never, 1 rows of <= 128 K chunks, 2 always") \
	X(three, 1, 0, 2, false, "UnphasedMath on planes without missing genotypes, r2 cut-off > 1e-6: contract three products a pair (HH and S = QH + HQ + 2 QQ: all the screen reads) and recount the four products of the pairs that pass (DESIGN 3.1a); 0: four products for every pair; 2: keep to three whatever a launch's candidate density (1 samples every launch first)") \
	X(count_min_chunks, 8, 1, 1 << 20, false, "shortest K range a tile of the count kernel is split into at the end of a launch (test hook: 1 splits short rows too)") \
	X(patch_rows, 8, 1, 4096, false, "rows of a patch of tiles in the count kernel's work order (DESIGN 3.1, profiles/r03_patch_pmc.txt)") \
	X(patch_cols, 8, 1, 4096, false, "... and its columns") \
	X(seg, 0, 0, 1 << 20, false, "walk a patch in K segments of this many chunks (0: whole tiles)") \
	X(xcd_queues, 0, 0, 8, false, "one unit queue per XCD (2..8; 0: one queue): -42 % fabric traffic at +0.5 % kernel time") \
	X(skip_pad, 1, 0, 1, false, "leave the zero padding behind a row's last live 8 bytes uncontracted (0: contract it)") \
	X(cand_chunk, -1, -1, 1 << 20, false, "candidate slots a wave of the fused kernels reserves at a time (-1: sized from the list)") \
	X(lists, 1, 0, 2, true, "carrier lists for the rare head of the allele-count-sorted plane sets (DESIGN 3.5): 0 never, 1 rows of >= 4096 words, 2 always (lists of >= 8 carriers)") \
	X(list_max, 0, 0, 60000, true, "longest carrier list kept (0: row words / 128, / 64 for UnphasedMath)") \
	X(probe, 1, 0, 1, false, "pairs of a listed variant with one that keeps no list: probes of its carriers into the partner's row (0: the dense contraction)") \
	X(probe_zone, 1, 0, 1, false, "rows with a list short enough to probe take every column behind them that way, the list zone's own included (0: pairs inside the zone are merges of two lists)") \
	X(probe_lds, 1, 0, 1, false, "probe kernels: the column rows are staged into LDS segment by segment and the carriers tested there (512 zone rows x 4 columns a block; 2 columns of unphased planes); 0: gathers from L2, one column a block (the twin the LDS form is tested against)") \
	X(band_launch, 1, 0, 1, false, "fused runs: launches sized by their work - a band of rows over all the columns it reaches - instead of by a count matrix; 0: matrix-sized tiles only") \
	X(band_work_log2, 19, 0, 40, false, "... of at least 2^n tile-chunks each (19: about 5 ms of contraction)") \
	X(band_max_launches, 8, 1, 64, false, "... and at most this many a region") \
	X(band_list_entries, 0, 0, 1ll << 32, false, "candidate slots of such a launch (0: 1/32 of its pairs, 4 M .. 256 M); a launch that outgrows them, or its survivor buffer, is redone as matrix-sized tiles") \
	X(band_reverse, 1, 0, 1, false, "allele-count-sorted runs: the last band (commonest variants, most survivors) first") \
	X(fisher_order, 1, 0, 1, false, "Fisher's walks binned by their length (DESIGN 3.2)") \
	X(fisher_lds, 1, 0, 1, false, "log-factorial table in LDS while it fits") \
	X(async_delivery, 1, 0, 1, false, "region calls with a sink: a finished launch's sorted survivors are copied aside on the device and a second thread of the engine takes them to the host and calls the sink - in the launches' order, one call at a time - while the calling thread goes on enqueueing launches (twk_delivery.h); 0: the calling thread does both") \
	X(deliver_buffers, 3, 1, 64, false, "staging buffers that thread may hold at a time: a launch whose survivors find none free waits for one (back-pressure)") \
	X(record_cap, 0, 0, 1ll << 40, false, "test hook: cap on a launch's survivor buffer in records (0: none) - forces the overflow paths: a matrix-sized tile is redone in row strips, a band launch as matrix-sized tiles") \
	X(deliver_fail_alloc_at, 0, 0, 1 << 30, false, "test hook: the n-th staging allocation of a region call fails (the calling thread then delivers that launch itself)") \
	X(deliver_fail_copy_at, 0, 0, 1 << 30, false, "test hook: the n-th copy aside of a region call fails (the call fails)") \
	X(timeline, 0, 0, 1, false, "1: the host's steps through a region's launch pipeline, with times, and the outlier watch's findings on stderr")
struct Options {
#define X(key, dflt, lo, hi, rebuilds, doc) long long key = dflt;
	TWK_HIP_OPTIONS(X)
#undef X
};
struct OptionKey { const char* name; long long Options::* field; long long dflt, lo, hi; bool rebuilds_planes; const char* doc; };
const OptionKey OPTION_KEYS[] = {
#define X(key, dflt, lo, hi, rebuilds, doc) {#key, &Options::key, dflt, lo, hi, rebuilds, doc},
	TWK_HIP_OPTIONS(X)
#undef X
};
}  // namespace

// The delivery thread of a region call (twk_delivery.h): the queue's device operations.
struct twk_hip_ctx;
struct HipDeliveryOps {
	twk_hip_ctx* c = nullptr;
	long long n_alloc = 0, n_copy = 0;       // of the current call (test hooks deliver_fail_alloc_at / deliver_fail_copy_at)
	void* alloc(size_t bytes);
	void release(void* p);
	int copy_aside(void* dst, const void* src, size_t bytes);
	int deliver(const void* recs, uint64_t n, twk_hip_record_sink sink, void* user, char* err, size_t err_len);
	void thread_begin();
};
typedef twk::DeliveryQueue<HipDeliveryOps, twk_hip_record_sink> Delivery;

struct twk_hip_ctx {
	int device = 0;
	Options opt;
	hipStream_t s_compute = nullptr, s_copy = nullptr, s_deliver = nullptr;      // s_deliver: the delivery thread's copies to the host
	HipDeliveryOps dl_ops;
	Delivery dl{sizeof(twk_hip_record)};
	bool deliver_warm = false;       // s_deliver has carried a copy (HipDeliveryOps::thread_begin)
	uint32_t N = 0, M = 0, M_alloc = 0;
	uint32_t Wp = 0, Wu = 0;       // padded words per row: raw (2N bits) / unphased planes (N bits)
	uint32_t* raw = nullptr;       // [M_alloc][Wp]
	uint32_t* rawmask = nullptr;   // [M_alloc][Wp] or null
	bool any_missing = false;
	// metadata (device SoA) + host mirror
	uint32_t *d_ac = nullptr, *d_an = nullptr, *d_pos = nullptr, *d_rid = nullptr, *d_missing = nullptr;
	double* d_hwe = nullptr;
	double* d_lfact = nullptr; int lfact_n = 0;    // lgamma(i + 1), i <= 2N: Fisher's log-binomials (ld_math.hip.h)
	uint32_t* d_fisher_bins = nullptr;             // [2 * FISHER_BINS]: bin sizes and fill cursors of the walk-length order (s_compute only)
	std::vector<twk_hip_variant_meta> h_meta;
	std::vector<uint32_t> h_popc;  // ALT alleles per variant as counted on the device (r2 screen; empty until needed, dropped on upload)
	PlaneSet planes[N_PLANE_SETS];
	Slot slot[PIPE_SLOTS + 1];     // [0 .. PIPE_SLOTS): the pipeline of region calls; [SYNC_SLOT]: synchronous single-tile calls
	twk_hip_record* h_recs = nullptr; unsigned long long h_recs_cap = 0;   // pinned staging
	// twk_hip_set_device_sink: the survivors of region calls stay on the device, appended here tile by tile
	StatsParams* d_list_stats = nullptr;          // parameter block of the list pass's math kernel (device copy)
	bool fused_ok = true;           // cleared for the rest of a call when a fused tile's candidate list overflowed
	bool three_ok = true;           // cleared for the rest of a call when a three-product launch had too many candidates for the recount to stay cheap
	bool sampling = false;          // the launch being enqueued is a density sample (RegionRun::decide_three_by_samples): its count kernel runs under its own name
	                                // (k_count3_list_t<.., 1>), so that a kernel trace's statistics of the real launches are not diluted by half-millisecond ones
	bool sorted_keeps_no_lists[2] = {false, false};      // [phased, unphased]: the allele-count-sorted set was built once for a run below the band's cut-off and
	                                                     // kept no carrier lists: such runs go the file-order way without building it again (cleared with the planes)
	bool device_sink = false;
	twk_hip_record* d_keep = nullptr; unsigned long long d_keep_n = 0, d_keep_cap = 0;
	// the survivors of a tile leave in (idxA, idxB) order: sort keys / permutation (double-buffered), the
	// reordered records, rocprim's scratch (grow-only)
	unsigned long long* d_sort_keys = nullptr; uint32_t* d_sort_vals = nullptr; twk_hip_record* d_sorted = nullptr;
	unsigned long long sort_cap = 0;
	void* d_sort_tmp = nullptr; size_t sort_tmp_bytes = 0;
	// the same for the sorts of band launches, which run on the compute stream (one at a time, in stream order) while a sort of
	// the other kind may be running on the copy stream
	unsigned long long* d_band_keys = nullptr; uint32_t* d_band_vals = nullptr; unsigned long long band_sort_cap = 0;
	void* d_band_tmp = nullptr; size_t band_tmp_bytes = 0;
	std::vector<void*> host_graveyard; // the same for page-locked host buffers
	std::vector<void*> graveyard;      // device buffers outgrown while launches were in flight: hipFree waits for the device, so they are freed when the call ends
	twk_hip_timing timing{};
	std::vector<twk_hip_launch_stat> launch_ring; uint64_t launches_seen = 0;      // the outlier watch's log (twk_hip_launch_log): the last LAUNCH_RING count launches
	twk_hip_progress_cb progress_cb = nullptr; void* progress_user = nullptr;
	bool progress_muted = false;       // second stage of a default-mode run: its pairs were already counted
	uint32_t resident_blocks = 512;   // count-kernel blocks the chip holds at once (2 per CU)
	uint32_t* tickets = nullptr;      // [2 * (PIPE_SLOTS + 1)][8] work tickets of the count launches: one set of queues per (slot, launch)
	// staging of twk_hip_upload_rle (grow-only): run bytes, descriptors, status word
	uint8_t* d_rle = nullptr; size_t d_rle_cap = 0;
	uint8_t* d_rle_desc = nullptr; size_t d_rle_desc_cap = 0;
	int* d_status = nullptr;
	uint32_t* d_col_hi = nullptr; size_t d_col_hi_cap = 0;   // r2 screen: per-row column limit of the current region
	char err[512] = {0};
};

namespace {

#define HIPCHK(ctx, call)                                                                         \
	do {                                                                                          \
		hipError_t e__ = (call);                                                                  \
		if (e__ != hipSuccess) {                                                                  \
			snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call,              \
			         hipGetErrorString(e__), __FILE__, __LINE__);                                 \
			return e__ == hipErrorOutOfMemory ? TWK_HIP_E_NOMEM : TWK_HIP_E_DEVICE;               \
		}                                                                                         \
	} while (0)

void free_planes(twk_hip_ctx* c) {
	c->h_popc.clear();
	c->sorted_keeps_no_lists[0] = c->sorted_keeps_no_lists[1] = false;
	for (auto& p : c->planes) {
		if (p.owns_rows && p.rows) (void)hipFree(p.rows);
		if (p.rowpop) (void)hipFree(p.rowpop);
		if (p.ids) (void)hipFree(p.ids);
		if (p.lists) (void)hipFree(p.lists);
		if (p.list_mac) (void)hipFree(p.list_mac);
		if (p.list_flip) (void)hipFree(p.list_flip);
		if (p.terms) (void)hipFree(p.terms);
		p = PlaneSet();
	}
}
void free_slots(twk_hip_ctx* c) {
	for (auto& s : c->slot) {
		if (s.C) (void)hipFree(s.C);
		if (s.out) (void)hipFree(s.out);
		if (s.keys) (void)hipFree(s.keys);
		if (s.vals) (void)hipFree(s.vals);
		if (s.sorted) (void)hipFree(s.sorted);
		s.C = nullptr; s.C_words = 0; s.out = nullptr; s.keys = nullptr; s.vals = nullptr; s.capacity = 0; s.sorted = nullptr; s.sorted_cap = 0;
		for (int k = 0; k < 2; ++k) {
			if (s.h_tiles[k]) (void)hipHostFree(s.h_tiles[k]);
			if (s.d_tiles[k]) (void)hipFree(s.d_tiles[k]);
			s.h_tiles[k] = s.d_tiles[k] = nullptr; s.tiles_cap[k] = 0;
		}
	}
}
void free_problem(twk_hip_ctx* c) {
	free_planes(c);
	free_slots(c);
	void* ptrs[] = {c->raw, c->rawmask, c->d_ac, c->d_an, c->d_pos, c->d_rid, c->d_missing, c->d_hwe, c->d_lfact};
	for (void* p : ptrs) if (p) (void)hipFree(p);
	c->raw = c->rawmask = nullptr; c->d_lfact = nullptr; c->lfact_n = 0;
	c->d_ac = c->d_an = c->d_pos = c->d_rid = c->d_missing = nullptr; c->d_hwe = nullptr;
	c->h_meta.clear();
	c->N = c->M = c->M_alloc = 0; c->any_missing = false;
}

int plane_kind_for(const twk_hip_ctx* c, bool phased) {
	if (phased) return c->any_missing ? PK_PHASED_MASKED : PK_PHASED;
	return c->any_missing ? PK_UNPHASED_MASKED : PK_UNPHASED;
}

// The r2 screen's bound is a statement about the margins of the table the kernels count, so it takes the
// allele counts from the bits on the device, not from the `ac` field of the file (a .twk whose header
// disagrees with its genotypes must not lose records to the screen).
int ensure_popcounts(twk_hip_ctx* c) {
	if (c->h_popc.size() == c->M) return TWK_HIP_OK;
	uint32_t* d = nullptr;
	HIPCHK(c, hipMalloc((void**)&d, (size_t)c->M * 4));
	hipLaunchKernelGGL(k_row_popcount, dim3((c->M + 3) / 4), dim3(256), 0, c->s_compute, c->raw, c->Wp, c->M, d);
	hipError_t e = hipGetLastError();
	c->h_popc.assign(c->M, 0);
	if (e == hipSuccess) e = hipMemcpyAsync(c->h_popc.data(), d, (size_t)c->M * 4, hipMemcpyDeviceToHost, c->s_compute);
	if (e == hipSuccess) e = hipStreamSynchronize(c->s_compute);
	(void)hipFree(d);
	if (e != hipSuccess) c->h_popc.clear();
	HIPCHK(c, e);
	return TWK_HIP_OK;
}

int ensure_planes(twk_hip_ctx* c, int set) {
	PlaneSet& ps = c->planes[set];
	if (ps.built) return TWK_HIP_OK;
	const int kind = set_kind(set);
	const int P = planes_per_variant(kind);
	const bool wide = (kind == PK_PHASED || kind == PK_PHASED_MASKED);
	ps.W = wide ? c->Wp : c->Wu;
	ps.W_live = wide ? (uint32_t)((2ull * c->N + 31) / 32) : (c->N + 31) / 32;
	ps.rows_alloc = round_up(c->M * P, TILE) + TILE;
	if ((kind == PK_PHASED_MASKED || kind == PK_UNPHASED_MASKED) && !c->rawmask) return TWK_HIP_E_STATE;
	const bool sorted = set == PS_SORTED_P || set == PS_SORTED_U;
	if (sorted) {
		// order: minor allele count ascending (ties in file order), variants with missing genotypes last (file order)
		const uint64_t T2 = 2ull * c->N;
		{ const int rc = ensure_popcounts(c); if (rc) return rc; }
		ps.h_ids.resize(c->M);
		for (uint32_t v = 0; v < c->M; ++v) ps.h_ids[v] = v;
		auto key = [&](uint32_t v) -> uint64_t {
			const twk_hip_variant_meta& m = c->h_meta[v];
			if (m.missing || m.an) return ~0ull;
			const uint64_t ac = std::min<uint64_t>(c->h_popc[v], T2);
			return std::min(ac, T2 - ac);
		};
		std::stable_sort(ps.h_ids.begin(), ps.h_ids.end(), [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
		ps.n_front = 0;                                 // number of variants without missing genotypes (they lead the set)
		for (uint32_t v = 0; v < c->M; ++v) if (key(ps.h_ids[v]) != ~0ull) ++ps.n_front;
	}
	if (kind == PK_PHASED && !sorted) {
		ps.rows = c->raw; ps.owns_rows = false;       // the raw layout *is* the phased plane
	} else {
		const size_t bytes = (size_t)ps.rows_alloc * ps.W * 4;
		HIPCHK(c, hipMalloc((void**)&ps.rows, bytes));
		ps.owns_rows = true;
		HIPCHK(c, hipMemsetAsync(ps.rows, 0, bytes, c->s_compute));
		if (sorted) {
			HIPCHK(c, hipMalloc((void**)&ps.ids, (size_t)c->M * 4));
			HIPCHK(c, hipMemcpyAsync(ps.ids, ps.h_ids.data(), (size_t)c->M * 4, hipMemcpyHostToDevice, c->s_compute));
		}
		if (set == PS_GROUPED) {
			ps.h_ids.clear(); ps.h_ids.reserve(c->M);
			for (uint32_t v = 0; v < c->M; ++v) if (c->h_meta[v].an) ps.h_ids.push_back(v);
			ps.n_front = (uint32_t)ps.h_ids.size();
			for (uint32_t v = 0; v < c->M; ++v) if (!c->h_meta[v].an) ps.h_ids.push_back(v);
			HIPCHK(c, hipMalloc((void**)&ps.ids, (size_t)c->M * 4));
			HIPCHK(c, hipMemcpyAsync(ps.ids, ps.h_ids.data(), (size_t)c->M * 4, hipMemcpyHostToDevice, c->s_compute));
		}
		const dim3 blk(256), grd((ps.W + 255) / 256, std::min<uint32_t>(c->M, 65535u));
		if (set == PS_SORTED_P)
			hipLaunchKernelGGL(k_permute_rows, grd, blk, 0, c->s_compute, c->raw, c->Wp, c->M, ps.rows, (const uint32_t*)ps.ids);
		else if (kind == PK_PHASED_MASKED)
			hipLaunchKernelGGL(k_build_phased_masked, grd, blk, 0, c->s_compute, c->raw, c->rawmask, c->Wp, c->M, ps.rows);
		else
			hipLaunchKernelGGL(k_build_unphased, grd, blk, 0, c->s_compute, c->raw,
			                   kind == PK_UNPHASED_MASKED ? c->rawmask : (const uint32_t*)nullptr,
			                   c->Wp, c->N, c->M, ps.rows, ps.W, P, (const uint32_t*)ps.ids);
		HIPCHK(c, hipGetLastError());
	}
	HIPCHK(c, hipMalloc((void**)&ps.rowpop, (size_t)ps.rows_alloc * 4));
	HIPCHK(c, hipMemsetAsync(ps.rowpop, 0, (size_t)ps.rows_alloc * 4, c->s_compute));
	const uint32_t live_rows = c->M * P;
	hipLaunchKernelGGL(k_row_popcount, dim3((live_rows + 3) / 4), dim3(256), 0, c->s_compute, ps.rows, ps.W, live_rows, ps.rowpop);
	HIPCHK(c, hipGetLastError());
	if (set == PS_SORTED_P || set == PS_SORTED_U) {
		// Carrier lists for the head of the set (ld_list.hip.h): worth it where a dense pair costs more than a merge of
		// two lists, i.e. for long rows only.  list_max = (phased row words) / 128 carriers for PhasedMath (the measured
		// break-even is ~W / 150 merge steps per side, profiles/r03_t2_list_vs_dense.txt) and twice that for UnphasedMath (a
		// dense unphased pair costs twice a phased one, a merge over samples about the same), and not below 32 - rows
		// shorter than 4096 words (N < 65,536) keep no lists.  Option "lists" = 0: never; 2: always, with at least 8 carriers
		// (test hook); "list_max" = n: the limit itself (measurement hook).
		const int lists_env = (int)c->opt.lists;
		uint32_t lmax = c->Wp / (set == PS_SORTED_U ? 64 : 128);      // measured optimum at N = 1 M: 488 / 976 carriers (profiles/r03_list_max_sweep.txt)
		if (lists_env == 2) lmax = std::max<uint32_t>(lmax, 8);
		if (c->opt.list_max >= 8) lmax = (uint32_t)c->opt.list_max;      // measurement hook (<= 60000: the unphased merge counts in 16 bits)
		if (lists_env != 0 && (c->Wp / 128 >= 32 || lists_env == 2)) {
			const uint64_t T2 = 2ull * c->N;
			uint32_t n = 0;                                  // variants of the missing-free head with a minor allele count <= lmax
			while (n < ps.n_front) {
				const uint64_t ac = std::min<uint64_t>(c->h_popc[ps.h_ids[n]], T2);
				if (std::min(ac, T2 - ac) > lmax) break;
				++n;
			}
			n = n / TILE * TILE;                             // whole tiles: a tile is either intersected or contracted
			if (n >= 2 * TILE) {
				ps.n_list = n; ps.list_max = lmax;
				// Probing (k_probe_screen) beats the dense pair up to ~W / 114 carriers in isolation (csrc/tools/probe_vs_dense.hip) and
				// up to about half that in a whole run, where the rows that reach beyond the zone are the ones with the longest lists and
				// the columns are thousands of 250 KB rows (1 M x 50,000: 10-11 ps per carrier against 5; with every list probing, the
				// default-cut-off run was 23 ms slower than without probes): PhasedMath probes lists of up to W / 256 carriers,
				// UnphasedMath - two reads per listed sample, a dense pair of twice the cost - of up to W / 160 entries.
				const uint32_t pmax = set == PS_SORTED_U ? std::max<uint32_t>(c->Wp / 160, 8) : std::max<uint32_t>(c->Wp / 256, 8);
				uint32_t np = 0;
				while (np < n) {
					const uint64_t ac = std::min<uint64_t>(c->h_popc[ps.h_ids[np]], T2);
					if (std::min(ac, T2 - ac) > pmax) break;
					++np;
				}
				ps.n_probe = np / TILE * TILE;
				HIPCHK(c, hipMalloc((void**)&ps.lists, (size_t)n * (lmax + 1) * 4));
				HIPCHK(c, hipMalloc((void**)&ps.list_mac, (size_t)n * 4));
				HIPCHK(c, hipMalloc((void**)&ps.list_flip, (size_t)n * 4));
				if (set == PS_SORTED_P)
					hipLaunchKernelGGL(k_build_lists, dim3((n + 3) / 4), dim3(256), 0, c->s_compute, (const uint32_t*)ps.rows, ps.W, ps.W_live, (uint64_t)T2,
					                   (const uint32_t*)ps.rowpop, n, lmax + 1, ps.lists, ps.list_mac, ps.list_flip);
				else
					hipLaunchKernelGGL(k_build_lists_unphased, dim3((n + 3) / 4), dim3(256), 0, c->s_compute, (const uint32_t*)ps.rows, ps.W, ps.W_live, c->N,
					                   (const uint32_t*)ps.rowpop, n, lmax + 1, ps.lists, ps.list_mac, ps.list_flip);
				HIPCHK(c, hipGetLastError());
			}
		}
	}
	HIPCHK(c, hipStreamSynchronize(c->s_compute));
	ps.built = true;
	return TWK_HIP_OK;
}

// (Outgrown buffers go to the graveyard - freed when the region call ends, or with the context: hipFree waits for the whole device,
// every stream, and in the middle of a region's pipeline that was a stall of 50-250 ms a time: profiles/r05_delivery_thread.txt.)
// hipMalloc for the buffers of a running call: out of memory -> what the delivery queue holds idle and what the call has outgrown is
// given back (reclaim_device_memory, below) and the allocation tried once more, before the call fails.
bool reclaim_device_memory(twk_hip_ctx* c);
hipError_t dev_malloc(twk_hip_ctx* c, void** p, size_t bytes) {
	hipError_t e = hipMalloc(p, bytes);
	if (e == hipErrorOutOfMemory && reclaim_device_memory(c)) e = hipMalloc(p, bytes);
	return e;
}
int ensure_slot(twk_hip_ctx* c, Slot& s, size_t C_words, unsigned long long capacity) {
	if (s.C_words < C_words) {
		if (s.C) c->graveyard.push_back(s.C);
		s.C = nullptr; s.C_words = 0;
		HIPCHK(c, dev_malloc(c, (void**)&s.C, C_words * 4));
		s.C_words = C_words;
	}
	if (s.capacity < capacity) {
		if (s.out) c->graveyard.push_back(s.out);
		if (s.keys) c->graveyard.push_back(s.keys);
		if (s.vals) c->graveyard.push_back(s.vals);
		s.out = nullptr; s.keys = nullptr; s.vals = nullptr; s.capacity = 0;
		HIPCHK(c, dev_malloc(c, (void**)&s.out, (size_t)capacity * sizeof(twk_hip_record)));
		HIPCHK(c, dev_malloc(c, (void**)&s.keys, (size_t)capacity * sizeof(unsigned long long)));
		HIPCHK(c, dev_malloc(c, (void**)&s.vals, (size_t)capacity * sizeof(uint32_t)));
		s.capacity = capacity;
	}
	s.cap_use = capacity;
	return TWK_HIP_OK;
}

// (err: where a failure's text goes - c->err on the calling thread, the delivery queue's own buffer on its thread)
#define HIPCHK_E(err, err_len, call)                                                              \
	do {                                                                                          \
		hipError_t e__ = (call);                                                                  \
		if (e__ != hipSuccess) {                                                                  \
			snprintf((err), (err_len), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
			return e__ == hipErrorOutOfMemory ? TWK_HIP_E_NOMEM : TWK_HIP_E_DEVICE;               \
		}                                                                                         \
	} while (0)

int ensure_host_records(twk_hip_ctx* c, unsigned long long n, char* err = nullptr, size_t err_len = 0) {
	if (!err) { err = c->err; err_len = sizeof(c->err); }
	if (c->h_recs_cap >= n) return TWK_HIP_OK;
	if (c->h_recs) (void)hipHostFree(c->h_recs);
	c->h_recs = nullptr; c->h_recs_cap = 0;
	const unsigned long long cap = std::max<unsigned long long>(n, 1ull << 16);
	HIPCHK_E(err, err_len, hipHostMalloc((void**)&c->h_recs, (size_t)cap * sizeof(twk_hip_record), hipHostMallocDefault));
	c->h_recs_cap = cap;
	return TWK_HIP_OK;
}

// Rows of at most this many K chunks (32 words each: N <= 65,536 phased, N <= 131,072 unphased) take the fused
// count -> screen kernel.  Up to 16 chunks a tile is never worth splitting, and the C round trip plus the
// one-thread-per-pair math front end cost as much as the counting itself.  Beyond that a fused launch ends less evenly
// (a block must hold a pair's whole count to screen it, so the last tiles cannot be cut along K: +2..6 % count kernel at
// 40-118 chunks) but still saves several times that in the math kernel (tests/sweeps/fused_mid_n.sh,
// profiles/r03_fused_mid_n.txt: N = 60,000 -p, 4e4 variants: 46 + 15 ms -> 49 + 2.5 ms); the two meet near 220 chunks.
constexpr uint32_t FUSED_MAX_CHUNKS = 128;

struct Geometry { uint32_t rowsA, rowsB, gx, gy, ldc; };
Geometry tile_geometry(int P, const twk_hip_tile_desc& t) {
	Geometry g;
	g.rowsA = round_up(t.nA * P, TILE); g.rowsB = round_up(t.nB * P, TILE);
	g.gy = g.rowsA / TILE; g.gx = g.rowsB / TILE; g.ldc = g.rowsB;
	return g;
}

// The 128 x 128 tiles of a super-tile that hold wanted pairs, as (tile row << 16 | tile column):
// on or above the diagonal (diag), and in window mode only those some row of the tile can reach.
// Order: 8 x 8 patches of tiles, patch by patch - the blocks pull consecutive tickets, so the ~P tiles in
// flight at any time are a few neighbouring patches (shared row / column tiles meet in L2 and the MALL).
void build_tile_list(const twk_hip_tile_desc& t, int P, const Geometry& g, bool diag, const ColRange* cr,
                     std::vector<uint32_t>& out, std::vector<uint32_t>* patch_end = nullptr, uint32_t PR = 8, uint32_t PC = 8) {
	std::vector<uint32_t> x0(g.gy, 0), x1(g.gy, g.gx);
	for (uint32_t by = 0; by < g.gy; ++by) {
		if (diag) x0[by] = by;
		if (cr && cr->lo) {
			const uint32_t v0 = t.rowA0 + (by * TILE) / P;
			const uint32_t v1 = std::min<uint64_t>((uint64_t)t.rowA0 + t.nA, (uint64_t)t.rowA0 + ((uint64_t)(by + 1) * TILE + P - 1) / P);   // exclusive
			if (v0 >= v1) { x1[by] = x0[by]; continue; }
			const uint64_t c_lo = (uint64_t)cr->b0 + cr->lo[v0 - cr->a0], c_hi = (uint64_t)cr->b0 + cr->hi[v1 - 1 - cr->a0];   // variants [c_lo, c_hi)
			const uint64_t lo = std::max<uint64_t>(c_lo, t.rowB0), hi = std::min<uint64_t>(c_hi, (uint64_t)t.rowB0 + t.nB);
			if (hi <= lo) { x1[by] = x0[by]; continue; }
			x0[by] = std::max<uint32_t>(x0[by], (uint32_t)(((lo - t.rowB0) * P) / TILE));
			x1[by] = std::min<uint32_t>(x1[by], (uint32_t)(((hi - t.rowB0) * P + TILE - 1) / TILE));
			if (x1[by] < x0[by]) x1[by] = x0[by];
		}
		if (cr && cr->list_zone) {      // a row of tiles that lies wholly inside the list zone starts at the zone's last column tile
			const uint64_t v1 = std::min<uint64_t>((uint64_t)t.rowA0 + t.nA, (uint64_t)t.rowA0 + ((uint64_t)(by + 1) * TILE + P - 1) / P);
			if (v1 <= cr->probe_zone) x1[by] = x0[by];                    // every pair of these rows is a list merge or a probe
			else if (v1 <= cr->list_zone && cr->list_zone > t.rowB0) {
				x0[by] = std::max<uint32_t>(x0[by], (uint32_t)((((uint64_t)cr->list_zone - t.rowB0) * P) / TILE));
				if (x1[by] < x0[by]) x1[by] = x0[by];
			}
		}
	}
	std::vector<uint32_t> seq;
	seq.reserve((size_t)g.gx * g.gy);
	// Patch shape 8 x 8.  Measured (profiles/r03_patch_pmc.txt, FETCH_SIZE = what the L2s ask the fabric for, 16,384 variants
	// at N = 1 M, 1.05 TB of tile-level demand per launch): 8 x 8 patches 258 GB per launch; wider patches are *worse*
	// (16 x 32: 309 GB), and so is walking a patch K segment by K segment (TWK_HIP_SEG=64: 306 GB) - the tiles in flight
	// are spread over eight L2s by the dispatcher, so fewer distinct row tiles in flight buys nothing per L2.  What does
	// help is giving each XCD its own patches (TWK_HIP_XCD_QUEUES=8 with 64-chunk segments: 149 GB, -42 %), at +0.5 % kernel
	// time for the adds into C; since the kernel is VALU-bound and the fabric sees < 10 % of its rate either way, that
	// trade is not taken by default.  (Options "patch_rows" / "patch_cols" / "seg" / "xcd_queues" are the hooks of that measurement.)
	if (patch_end) patch_end->clear();
	for (uint32_t py = 0; py < g.gy; py += PR)
		for (uint32_t px = 0; px < g.gx; px += PC) {
			for (uint32_t y = py; y < std::min(py + PR, g.gy); ++y)
				for (uint32_t x = std::max(px, x0[y]); x < std::min(px + PC, x1[y]); ++x) seq.push_back(y << 16 | x);
			if (patch_end && (patch_end->empty() ? !seq.empty() : seq.size() > patch_end->back())) patch_end->push_back((uint32_t)seq.size());
		}
	out.swap(seq);
}

// Launch the count kernel for one tile on the compute stream (which: first or second launch of the slot).
// screen != null: the fused form (k_count_screen_t) if the launch qualifies - rows short enough that no tile's K range
// is split (or TWK_HIP_FUSED=2: never split) - in which case *fused is set and the slot's C buffer holds the candidate
// list instead of counts.
// The two parameter blocks of a fused launch travel to the device behind the tile list and the unit table (one copy
// per launch as before): the kernels read them from memory where they need them instead of holding ~60 more scalar
// registers through the contraction loop / the candidate loop.
// three: the three-product form (fa->unphased) - fused where the launch fuses, else k_count3_list_t into an (HH, S) matrix; the parameter
// blocks travel in both cases (*d_screen / *d_stats: where they landed).
struct FusedArgs { ScreenWork screen; StatsParams stats; int unphased; };
int launch_count(twk_hip_ctx* c, int set, const twk_hip_tile_desc& t, Slot& s, int which, hipEvent_t e0, hipEvent_t e1,
                 uint64_t* row_pairs, const ColRange* cr = nullptr, const FusedArgs* fa = nullptr, bool* fused = nullptr,
                 const StatsParams** d_stats = nullptr, bool three = false, const ScreenWork** d_screen = nullptr) {
	const PlaneSet& ps = c->planes[set];
	const int P = planes_per_variant(set_kind(set));
	const Geometry g = tile_geometry(P, t);
	if ((uint64_t)t.rowA0 * P + g.rowsA > ps.rows_alloc || (uint64_t)t.rowB0 * P + g.rowsB > ps.rows_alloc) return TWK_HIP_E_INVALID;
	if (g.gx > 0xFFFFu || g.gy > 0xFFFFu) return TWK_HIP_E_INVALID;
	const bool diag = t.diag && t.rowA0 == t.rowB0;
	const uint32_t n_blocks = c->resident_blocks;
	std::vector<uint32_t> list, patch_end;
	build_tile_list(t, P, g, diag, cr, list, &patch_end, (uint32_t)c->opt.patch_rows, (uint32_t)c->opt.patch_cols);
	const size_t T = list.size();
	// units of work: see build_count_units (ld_count.hip.h)
	const uint32_t nchunks = ps.W / KC;
	uint32_t min_chunks = (uint32_t)c->opt.count_min_chunks;      // (test hook: 1 splits short rows too)
	bool fuse = false;
	if (fa) {
		const int fused_env = (int)c->opt.fused;      // 0: never; 1 (default): rows of <= FUSED_MAX_CHUNKS chunks; 2: always (test hook)
		fuse = fused_env == 2 || (fused_env == 1 && nchunks <= FUSED_MAX_CHUNKS);
		if (fuse) min_chunks = nchunks + 1;          // whole tiles only: a block must hold a pair's whole count to screen it
	}
	if (fused) *fused = fuse;
	const bool with_args = fuse || (three && fa);       // the parameter blocks go to the device behind the unit table
	std::vector<CountUnit> units;
	uint32_t seg_chunks = (uint32_t)c->opt.seg;    // 0: whole tiles (see build_tile_list for the measurement behind that)
	if (fuse) seg_chunks = 0;
	uint32_t first_split = 0, n_queues = 1, queue_begin[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	const uint32_t xcd_queues = (uint32_t)c->opt.xcd_queues;    // measurement hook (profiles/): one unit queue per XCD
	if (T && !fuse && xcd_queues > 1 && xcd_queues <= 8 && nchunks >= 128 && !patch_end.empty()) {
		// patches dealt round robin to the queues; within a queue patch after patch, each K segment by K segment
		const uint32_t nseg = seg_chunks ? (nchunks + seg_chunks - 1) / seg_chunks : 1;
		std::vector<std::vector<CountUnit>> qs(xcd_queues);
		uint32_t p0 = 0;
		for (size_t pi = 0; pi < patch_end.size(); ++pi) {
			auto& qv = qs[pi % xcd_queues];
			for (uint32_t sgm = 0; sgm < nseg; ++sgm)
				for (uint32_t tl = p0; tl < patch_end[pi]; ++tl)
					qv.push_back(CountUnit{tl, (uint32_t)((unsigned long long)nchunks * sgm / nseg), (uint32_t)((unsigned long long)nchunks * (sgm + 1) / nseg), 0});
			p0 = patch_end[pi];
		}
		n_queues = xcd_queues;
		for (uint32_t q = 0; q < n_queues; ++q) { queue_begin[q] = (uint32_t)units.size(); units.insert(units.end(), qs[q].begin(), qs[q].end()); }
		for (uint32_t q = n_queues; q <= 8; ++q) queue_begin[q] = (uint32_t)units.size();
		first_split = nseg > 1 ? 0 : (uint32_t)T;
	} else {
		first_split = T ? build_count_units((uint32_t)T, nchunks, n_blocks, min_chunks, units, 8, 8, patch_end.data(), (uint32_t)patch_end.size(), seg_chunks) : 0;
		queue_begin[1] = (uint32_t)units.size();
	}
	fill_unit_tiles(units, list.data());
	const size_t T4 = (T + 3) / 4 * 4, words_units = T4 + units.size() * 4;       // [tiles | pad | units], units 16-byte aligned
	const size_t fa_words = (sizeof(FusedArgs) + 15) / 16 * 4;
	const size_t words = words_units + (with_args ? fa_words : 0);               // [... | FusedArgs] for a fused or three-product launch
	if (s.tiles_cap[which] < words) {
		if (s.h_tiles[which]) c->host_graveyard.push_back(s.h_tiles[which]);      // (the device copy of the previous launch's table may still be travelling)
		if (s.d_tiles[which]) c->graveyard.push_back(s.d_tiles[which]);
		s.h_tiles[which] = s.d_tiles[which] = nullptr; s.tiles_cap[which] = 0;
		const size_t cap = std::max<size_t>(words, 16384);
		HIPCHK(c, hipHostMalloc((void**)&s.h_tiles[which], cap * 4, hipHostMallocDefault));
		HIPCHK(c, hipMalloc((void**)&s.d_tiles[which], cap * 4));
		s.tiles_cap[which] = cap;
	}
	if (T) {
		std::memcpy(s.h_tiles[which], list.data(), T * 4);
		std::memcpy(s.h_tiles[which] + T4, units.data(), units.size() * sizeof(CountUnit));
		if (with_args) std::memcpy(s.h_tiles[which] + words_units, fa, sizeof(FusedArgs));
		HIPCHK(c, hipMemcpyAsync(s.d_tiles[which], s.h_tiles[which], words * 4, hipMemcpyHostToDevice, c->s_compute));
	}
	HIPCHK(c, hipEventRecord(e0, c->s_compute));
	if (T) {
		CountWork w{};
		w.rows = ps.rows; w.W = ps.W; w.rowA0 = t.rowA0 * P; w.rowB0 = t.rowB0 * P;
		w.tiles = s.d_tiles[which];
		w.units = reinterpret_cast<const CountUnit*>(s.d_tiles[which] + T4); w.n_units = (uint32_t)units.size();
		w.C = s.C; w.ldc = g.ldc;
		w.ticket = c->tickets + ((&s - c->slot) * 2 + which) * 8;
		w.n_queues = n_queues;
		w.clocks = s.n_out + 4;
		for (int q = 0; q < 9; ++q) w.queue_begin[q] = queue_begin[q];
		{
			const bool skip_pad = c->opt.skip_pad != 0;      // measurement hook: 0 = contract the zero padding too
			const uint32_t live_last = ps.W_live - (ps.W / KC - 1) * KC;      // live words of the last chunk, 1..KC (W = W_live rounded up to KC)
			w.last_halves = (skip_pad && ps.W_live && ps.W_live <= ps.W && ps.W - ps.W_live < KC) ? (live_last + 1) / 2 : 0;
			// The shortened chunk runs a plain rolled loop (no reads in flight behind the contraction): worth it when it drops a
			// quarter of the chunk or more, not for a half-slot or two (2,504 samples phased: 157 live words of 160, 15 half-slots
			// of the fifth chunk - there the pipelined loop over all 16 is the faster one).
			if (w.last_halves > 12) w.last_halves = 0;
		}
		HIPCHK(c, hipMemsetAsync(w.ticket, 0, 8 * 4, c->s_compute));
		if (first_split < T) {
			hipLaunchKernelGGL(k_zero_tiles, dim3((uint32_t)T - first_split), dim3(256), 0, c->s_compute, w.tiles, first_split, w.C, w.ldc, (uint32_t)(three ? TILE / 2 : TILE));
			HIPCHK(c, hipGetLastError());
		}
		const FusedArgs* d_fa = reinterpret_cast<const FusedArgs*>(s.d_tiles[which] + words_units);
		if (with_args && d_stats) *d_stats = &d_fa->stats;
		if (with_args && d_screen) *d_screen = &d_fa->screen;
		if (fuse && fa->unphased && three && c->sampling) hipLaunchKernelGGL((k_count3_screen_unphased_t<COUNT_NW, 1>), dim3(std::min(n_blocks, w.n_units)), dim3(COUNT_THREADS), 0, c->s_compute, w, &d_fa->screen);
		else if (fuse && fa->unphased && three) hipLaunchKernelGGL((k_count3_screen_unphased_t<COUNT_NW>), dim3(std::min(n_blocks, w.n_units)), dim3(COUNT_THREADS), 0, c->s_compute, w, &d_fa->screen);
		else if (three && c->sampling) hipLaunchKernelGGL((k_count3_list_t<COUNT_NW, 1>), dim3(std::min(n_blocks, w.n_units)), dim3(COUNT_THREADS), 0, c->s_compute, w);
		else if (three) hipLaunchKernelGGL((k_count3_list_t<COUNT_NW>), dim3(std::min(n_blocks, w.n_units)), dim3(COUNT_THREADS), 0, c->s_compute, w);
		else if (fuse && fa->unphased) hipLaunchKernelGGL((k_count_screen_unphased_t<COUNT_NW>), dim3(std::min(n_blocks, w.n_units)), dim3(COUNT_THREADS), 0, c->s_compute, w, &d_fa->screen);
		else if (fuse) hipLaunchKernelGGL((k_count_screen_t<COUNT_NW>), dim3(std::min(n_blocks, w.n_units)), dim3(COUNT_THREADS), 0, c->s_compute, w, &d_fa->screen);
		else hipLaunchKernelGGL((k_count_list_t<COUNT_NW>), dim3(std::min(n_blocks, w.n_units)), dim3(COUNT_THREADS), 0, c->s_compute, w);
		HIPCHK(c, hipGetLastError());
	}
	HIPCHK(c, hipEventRecord(e1, c->s_compute));
	*row_pairs = (uint64_t)T * TILE * TILE;
	return TWK_HIP_OK;
}

// Bits of idxB in the sort key idxA << shift | idxB of a survivor.
uint32_t key_shift_for(uint32_t n_variants) { uint32_t b = 1; while (b < 32 && (1ull << b) < n_variants) ++b; return b; }

StatsParams make_stats(twk_hip_ctx* c, int set, const twk_hip_tile_desc& t, const Slot& s, bool phased_math,
                       int auto_select, const twk_hip_filters& f, const ColRange* cr = nullptr) {
	const PlaneSet& ps = c->planes[set];
	const int kind = set_kind(set);
	const int P = planes_per_variant(kind);
	StatsParams p;
	p.tv.C = s.C; p.tv.ldc = tile_geometry(P, t).ldc; p.tv.rowpop = ps.rowpop; p.tv.kind = kind;
	p.tv.n_samples = c->N; p.tv.a0 = t.rowA0; p.tv.b0 = t.rowB0; p.tv.ids = ps.ids;
	p.vm = VariantMeta{c->d_ac, c->d_an, c->d_pos, c->d_rid, c->d_missing, c->d_hwe};
	p.raw = c->raw; p.rawmask = c->rawmask; p.Wp = c->Wp;
	p.col_hi = cr ? cr->d_hi : nullptr; p.hi_a0 = cr ? cr->a0 : 0; p.hi_b0 = cr ? cr->b0 : 0;
	p.list_zone = cr ? cr->list_zone : 0; p.probe_zone = cr ? cr->probe_zone : 0;
	p.nA = t.nA; p.nB = t.nB; p.n_variants = c->M;
	p.diag = (t.diag && t.rowA0 == t.rowB0) ? 1 : 0;
	p.phased_math = phased_math ? 1 : 0; p.auto_select = auto_select;
	p.window = t.window; p.l_window = t.l_window;
	p.filt = f; p.out = s.out; p.capacity = s.cap_use; p.n_out = s.n_out;
	p.keys = s.keys; p.vals = s.vals; p.key_shift = key_shift_for(c->M);
	return p;
}

uint64_t pairs_in_tile(const twk_hip_ctx* c, const twk_hip_tile_desc& t) {
	const uint64_t nA = t.rowA0 >= c->M ? 0 : std::min<uint64_t>(t.nA, c->M - t.rowA0);
	const uint64_t nB = t.rowB0 >= c->M ? 0 : std::min<uint64_t>(t.nB, c->M - t.rowB0);
	if (t.diag && t.rowA0 == t.rowB0) { const uint64_t n = std::min(nA, nB); return n * (n - 1) / 2 + (nB > n ? n * (nB - n) : 0); }
	return nA * nB;
}

// Enqueue everything for one tile into slot s (count [+ second pass], math, counter copy).
// What a mode runs on a tile: one or two (plane set, math, pair selection) passes.
struct TilePlan { int set1, set2; bool phased1; int select1; int Pmax; };
TilePlan plan_for(const twk_hip_ctx* c, int mode) {
	TilePlan p{PK_PHASED, -1, true, 0, 1};
	switch (mode) {
	case TWK_HIP_MODE_PHASED:   p.set1 = plane_kind_for(c, true); break;
	case TWK_HIP_MODE_UNPHASED: p.set1 = plane_kind_for(c, false); p.phased1 = false; break;
	case MODE_INT_AUTO_CLEAN:   p.select1 = 1; break;
	case MODE_INT_GROUPED:      p.set1 = PS_GROUPED; p.phased1 = false; break;
	case MODE_INT_SORTED_P:     p.set1 = PS_SORTED_P; break;
	case MODE_INT_SORTED_U:     p.set1 = PS_SORTED_U; p.phased1 = false; break;
	default:                    // AUTO on one tile: plain phased (pairs without missing) then masked unphased
		if (c->any_missing) { p.select1 = 1; p.set2 = PK_UNPHASED_MASKED; }
		break;
	}
	p.Pmax = planes_per_variant(set_kind(p.set1));
	if (p.set2 >= 0) p.Pmax = std::max(p.Pmax, planes_per_variant(set_kind(p.set2)));
	return p;
}

// Fisher's exact test, one record per lane, on recs[0, min(*n_out, cap)): starting points (k_fisher_prepare), then the
// walks (k_ld_fisher_t) - in the order of their length (k_fisher_scatter) when `scratch` (scratch_words uint32, free at
// this point of the stream) is given; records beyond the scratch keep their place.  ld_math.hip.h.
// Option "fisher_order" = 0: walks in the order the records were appended; "fisher_lds" = 0: log-factorial table read
// from global memory also when it would fit LDS (measurement hooks).
int launch_fisher(twk_hip_ctx* c, twk_hip_record* recs, unsigned long long* n_out, unsigned long long cap, double minP,
                  uint32_t* scratch, size_t scratch_words, unsigned long long* keys = nullptr) {
	const LFact lf{c->d_lfact, c->lfact_n};
	const bool ordered = c->opt.fisher_order != 0, lds_ok = c->opt.fisher_lds != 0;
	const bool lds_table = lds_ok && c->lfact_n <= FISHER_LDS_TABLE_MAX;
	const size_t lds_bytes = lds_table ? (size_t)c->lfact_n * sizeof(double) : 0;
	if (!c->d_fisher_bins) HIPCHK(c, hipMalloc((void**)&c->d_fisher_bins, 2 * FISHER_BINS * sizeof(uint32_t)));
	const unsigned long long limit = (ordered && scratch && scratch_words >= 4096) ? std::min<unsigned long long>(scratch_words, 0xFFFFFFFFull) : 0;
	HIPCHK(c, hipMemsetAsync(c->d_fisher_bins, 0, 2 * FISHER_BINS * sizeof(uint32_t), c->s_compute));
	if (lds_table) hipLaunchKernelGGL(k_fisher_prepare<true>, dim3(c->resident_blocks), dim3(1024), lds_bytes, c->s_compute, recs, (const unsigned long long*)n_out, cap, lf, limit, c->d_fisher_bins);
	else hipLaunchKernelGGL(k_fisher_prepare<false>, dim3(c->resident_blocks * 2), dim3(256), 0, c->s_compute, recs, (const unsigned long long*)n_out, cap, lf, limit, c->d_fisher_bins);
	if (limit) {
		const unsigned long long bound = std::min(cap, limit);
		const unsigned blocks = (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>(c->resident_blocks * 4ull, (bound + 1023) / 1024));
		hipLaunchKernelGGL(k_fisher_scatter, dim3(blocks), dim3(256), 0, c->s_compute, (const twk_hip_record*)recs, (const unsigned long long*)n_out, cap, limit, c->d_fisher_bins, scratch);
	}
	if (lds_table) hipLaunchKernelGGL(k_ld_fisher_t<true>, dim3(c->resident_blocks), dim3(1024), lds_bytes, c->s_compute, recs, n_out, cap, minP, lf,
	                                  (const uint32_t*)(limit ? scratch : nullptr), limit, 1, keys);
	else hipLaunchKernelGGL(k_ld_fisher_t<false>, dim3(c->resident_blocks * 2), dim3(256), 0, c->s_compute, recs, n_out, cap, minP, lf,
	                        (const uint32_t*)(limit ? scratch : nullptr), limit, 1, keys);
	HIPCHK(c, hipGetLastError());
	return TWK_HIP_OK;
}

// Would a launch of this mode run the fused count -> screen form (and nothing else: one pass)?  The same tests as
// enqueue_tile / launch_count make, for the callers that size a launch by it.
bool fused_form_applies(twk_hip_ctx* c, int mode, const twk_hip_filters& f) {
	const TilePlan pl = plan_for(c, mode);
	if (pl.set2 >= 0 || !c->fused_ok || c->opt.fused == 0) return false;
	const int k = set_kind(pl.set1);
	if (!((pl.phased1 && k == PK_PHASED) || (!pl.phased1 && k == PK_UNPHASED))) return false;
	if (!(f.minR2 > 1e-6 && f.minR2 <= 1.0)) return false;
	if (ensure_planes(c, pl.set1) != TWK_HIP_OK) return false;
	return c->opt.fused == 2 || c->planes[pl.set1].W / KC <= FUSED_MAX_CHUNKS;
}

// The four products of the candidates of a three-product launch (k_recount_unphased), on the compute stream: a wave per candidate for
// long rows, a DPP row of 16 lanes for rows of up to 1024 words.  The recount also checks the contraction: a candidate whose (HH, S)
// disagrees with its own four products is counted in n_out[3], and finish_tile fails the call on it.
int launch_recount(twk_hip_ctx* c, int set, Slot& s) {
	const PlaneSet& ps = c->planes[set];
	if (ps.W_live <= 1024)
		hipLaunchKernelGGL(k_recount_unphased<16>, dim3(c->resident_blocks * 4), dim3(256), 0, c->s_compute, (const uint32_t*)ps.rows, ps.W, ps.W_live, s.cand,
		                   (const unsigned long long*)(s.n_out + 2), s.cand_cap, s.n_out + 3);
	else
		hipLaunchKernelGGL(k_recount_unphased<64>, dim3(c->resident_blocks * 4), dim3(256), 0, c->s_compute, (const uint32_t*)ps.rows, ps.W, ps.W_live, s.cand,
		                   (const unsigned long long*)(s.n_out + 2), s.cand_cap, s.n_out + 3);
	HIPCHK(c, hipGetLastError());
	return TWK_HIP_OK;
}

// list_words != 0: a band launch (region_impl) - the fused form with a candidate list of that many words and no count
// matrix at all (its rectangle may be far beyond what a matrix could hold); it is an error if the launch does not fuse.
int enqueue_tile(twk_hip_ctx* c, int mode, const twk_hip_tile_desc& t, const twk_hip_filters& f, Slot& s,
                 unsigned long long capacity, const ColRange* cr = nullptr, size_t list_words = 0) {
	const TilePlan pl = plan_for(c, mode);
	const bool two_pass = pl.set2 >= 0;
	const bool phased = pl.phased1;
	const int kind1 = pl.set1, kind2 = pl.set2;
	int rc = ensure_planes(c, kind1); if (rc) return rc;
	if (two_pass) { rc = ensure_planes(c, kind2); if (rc) return rc; }
	const Geometry g = tile_geometry(pl.Pmax, t);
	if (list_words && !fused_form_applies(c, mode, f)) return TWK_HIP_E_STATE;
	const auto tl0 = std::chrono::steady_clock::now();
	auto tl = [&](const char* what) { if (c->opt.timeline) fprintf(stderr, "[timeline]     enqueue_tile: %s at +%.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count()); };
	rc = ensure_slot(c, s, list_words ? list_words : (size_t)g.rowsA * g.rowsB, capacity); if (rc) return rc;
	tl("slot buffers");
	s.two_pass = two_pass;

	HIPCHK(c, hipMemsetAsync(s.n_out, 0, N_SLOT_COUNTERS * sizeof(unsigned long long), c->s_compute));
	// The fused form: plain phased planes (one count per pair) with PhasedMath, or plain unphased planes (four products per
	// pair, gathered in the epilogue) with UnphasedMath, and an r2 cut-off the screen can use.
	const bool fused_u = !phased && set_kind(kind1) == PK_UNPHASED;
	const bool want_fused = c->fused_ok && ((phased && set_kind(kind1) == PK_PHASED) || fused_u) && f.minR2 > 1e-6 && f.minR2 <= 1.0;
	FusedArgs fa{};
	fa.unphased = fused_u ? 1 : 0;
	ScreenWork& sw = fa.screen;
	s.fused = false; s.is_list = false; s.is_probe = false; s.cand_overflow = false; s.cand_cap = (list_words ? list_words : s.C_words) / (fused_u ? 6 : 3);
	s.three = false; s.three_plain = false; s.cand = s.C; s.band_too_big = false;
	// The three-product form (ld_count.hip.h): UnphasedMath on the plain unphased planes with a cut-off the screen can use.  Where the launch
	// does not fuse (long rows: tiles are split along K) the (HH, S) matrix takes the first half of the slot's count buffer and the candidate
	// list the room behind it - at most 1/128 of the tile's pairs: a candidate's recount streams its four rows once more, ~25 pairs' worth of
	// contraction, so a launch with more candidates than that is cheaper in the four-product form (overflow -> three_ok = false -> redone).
	const bool want_three = want_fused && fused_u && c->three_ok && c->opt.three != 0;
	const bool fuses = want_fused && (c->opt.fused == 2 || (c->opt.fused == 1 && c->planes[kind1].W / KC <= FUSED_MAX_CHUNKS));      // (launch_count's own test)
	if (want_three && !fuses) {
		const size_t c2_words = (size_t)(g.rowsA / 2) * g.rowsB;
		const unsigned long long room = s.C_words > c2_words ? (s.C_words - c2_words) / 6 : 0;
		const unsigned long long pairs = (unsigned long long)t.nA * t.nB;
		s.cand = s.C + c2_words;
		s.cand_cap = std::min<unsigned long long>(room, c->opt.three == 2 ? room : std::max<unsigned long long>(pairs / 128, 4096));
	}
	if (want_fused) {
		PlaneSet& ps = c->planes[kind1];
		sw.cut = f.minR2 * (1.0 - 1e-6); sw.two_n = 2.0 * (double)c->N;
		if (fuses && (!ps.terms || ps.terms_cut != sw.cut)) {      // the prefilter's per-variant terms: once per plane set and cut-off, on the stream the kernels follow
			const uint32_t P1 = (uint32_t)planes_per_variant(set_kind(kind1)), n_pos = ps.rows_alloc / P1;
			if (!ps.terms) HIPCHK(c, hipMalloc((void**)&ps.terms, (size_t)n_pos * sizeof(float4)));
			hipLaunchKernelGGL(k_screen_terms, dim3((n_pos + 255) / 256), dim3(256), 0, c->s_compute, (const uint32_t*)ps.rowpop, n_pos, (int)P1, sw.two_n, sw.cut, ps.terms);
			HIPCHK(c, hipGetLastError());
			ps.terms_cut = sw.cut;
		}
		sw.terms = ps.terms; sw.slack = 0.5f + (float)sw.two_n * (1.0f / 1048576.0f);
		fa.stats = make_stats(c, kind1, t, s, phased, pl.select1, f, cr);
		sw.rowpop = ps.rowpop; sw.a0 = t.rowA0; sw.b0 = t.rowB0; sw.nA = t.nA; sw.nB = t.nB;
		sw.n_variants = c->M; sw.diag = (t.diag && t.rowA0 == t.rowB0) ? 1 : 0;
		sw.col_hi = cr ? cr->d_hi : nullptr; sw.hi_a0 = cr ? cr->a0 : 0; sw.hi_b0 = cr ? cr->b0 : 0; sw.hi_n = cr ? cr->n_hi : 0;
		sw.list_zone = cr ? cr->list_zone : 0; sw.probe_zone = cr ? cr->probe_zone : 0;
		sw.two_n = 2.0 * (double)c->N; sw.cut = f.minR2 * (1.0 - 1e-6);
		sw.cand = s.cand; sw.cap = s.cand_cap; sw.n_cand = s.n_out + 2;
		{	// slots a wave reserves at a time: what it cannot use is lost to the list, so at most an eighth of the list's
			// capacity may be tied up in the waves' windows (small tiles: 0, i.e. one atomic per wave and tile)
			const unsigned long long per_wave = s.cand_cap / (8ull * c->resident_blocks * (COUNT_THREADS / 64));
			sw.chunk = c->opt.cand_chunk >= 0 ? (uint32_t)c->opt.cand_chunk      // measurement hook
			                                  : (per_wave >= 64 ? (uint32_t)std::min<unsigned long long>(per_wave, 128) : 0u);
		}
	}
	const StatsParams* d_stats = nullptr; const ScreenWork* d_screen = nullptr;
	s.deferred = false; s.was_deferred = false; s.presorted = false;
	rc = launch_count(c, kind1, t, s, 0, s.ev_c0, s.ev_c1, &s.row_pairs, cr, want_fused ? &fa : nullptr, &s.fused, &d_stats, want_three, &d_screen); if (rc) return rc;
	s.three = want_three; s.three_plain = want_three && !s.fused; s.plane_set = kind1;
	tl("count kernel enqueued");
	if (list_words) {
		// A band launch stops here for now: how many survivors it can have is how many candidates it found, and only the count
		// kernel knows.  The counters travel to the host behind it; enqueue_band_math sizes the survivor buffer by them and
		// enqueues the rest (sizing it by a guess - 1/32 of the launch's pairs - meant gigabytes of allocation per slot, a
		// tenth of a second each, for launches that then kept a few thousand records).
		if (!s.fused) return TWK_HIP_E_STATE;
		s.deferred = true; s.was_deferred = true; s.deferred_unphased = fused_u; s.d_stats_dev = const_cast<StatsParams*>(d_stats); s.stats_host = fa.stats; s.minP = f.minP;
		HIPCHK(c, hipMemcpyAsync(s.h_n_out, s.n_out, N_SLOT_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->s_compute));
		HIPCHK(c, hipEventRecord(s.ev_c1b, c->s_compute));
		return TWK_HIP_OK;
	}
	if (s.three_plain && d_stats) {
		// (HH, S) matrix -> screen -> candidates -> their four products -> the list math
		hipLaunchKernelGGL(k_screen3_pairs, dim3((t.nB + SCREEN3_THREADS - 1) / SCREEN3_THREADS, t.nA), dim3(SCREEN3_THREADS), 0, c->s_compute, d_screen, d_stats, (const uint32_t*)s.C, g.ldc);
		HIPCHK(c, hipGetLastError());
		rc = launch_recount(c, kind1, s); if (rc) return rc;
		hipLaunchKernelGGL(k_ld_stats_list_unphased, dim3(c->resident_blocks * 4), dim3(256), 0, c->s_compute, d_stats, (const uint32_t*)s.cand,
		                   (const unsigned long long*)(s.n_out + 2), s.cand_cap);
	} else if (s.three_plain) {
		// (no tiles, no launch, no candidates)
	} else if (s.fused) {
		if (d_stats && s.three) { rc = launch_recount(c, kind1, s); if (rc) return rc; }
		if (d_stats && fused_u)
			hipLaunchKernelGGL(k_ld_stats_list_unphased, dim3(c->resident_blocks * 4), dim3(256), 0, c->s_compute, d_stats, (const uint32_t*)s.C,
			                   (const unsigned long long*)(s.n_out + 2), s.cand_cap);
		else if (d_stats)     // (no tiles, no launch, no candidates)
			hipLaunchKernelGGL(k_ld_stats_list, dim3(c->resident_blocks * 4), dim3(256), 0, c->s_compute, d_stats, (const uint32_t*)s.C,
			                   (const unsigned long long*)(s.n_out + 2), s.cand_cap);
	} else {
		const StatsParams p = make_stats(c, kind1, t, s, phased, pl.select1, f, cr);
		hipLaunchKernelGGL(k_ld_stats, dim3((t.nB + 255) / 256, t.nA), dim3(256), 0, c->s_compute, p);
	}
	HIPCHK(c, hipGetLastError());
	if (two_pass) {
		rc = launch_count(c, kind2, t, s, 1, s.ev_c0b, s.ev_c1b, &s.row_pairs_b, cr); if (rc) return rc;
		const StatsParams p = make_stats(c, kind2, t, s, false, 2, f);
		hipLaunchKernelGGL(k_ld_stats, dim3((t.nB + 255) / 256, t.nA), dim3(256), 0, c->s_compute, p);
		HIPCHK(c, hipGetLastError());
	}
	// Fisher's exact test on the compacted survivors (the slot's count / candidate buffer is free by now - the math
	// kernels in front are done with it - and holds the walk-length order)
	rc = launch_fisher(c, s.out, s.n_out, s.cap_use, f.minP, s.C, s.C_words, s.keys); if (rc) return rc;
	HIPCHK(c, hipGetLastError());
	s.minP = f.minP;
	HIPCHK(c, hipMemcpyAsync(s.h_n_out, s.n_out, N_SLOT_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->s_compute));
	HIPCHK(c, hipEventRecord(s.ev_s1, c->s_compute));
	return TWK_HIP_OK;
}

// Survivors are appended with an atomic counter, in no order.  They leave the device in (idxA, idxB) order
// - the order the writer puts them in the file, which makes a one-GPU run's output deterministic - by a key
// sort of (idxA << bits | idxB, position) and a gather.  Key and position are written by the math kernels where the
// record is appended (d_append_survivor); records the Fisher cut-off drops get the all-ones key from the Fisher kernel
// and sort behind the rest.
__global__ void k_gather_records(const twk_hip_record* __restrict__ recs, const uint32_t* __restrict__ order, unsigned long long n,
                                 twk_hip_record* __restrict__ out) {
	constexpr uint32_t W = sizeof(twk_hip_record) / 8;              // 13 eight-byte words per record
	const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n * W) return;
	const unsigned long long r = i / W; const uint32_t w = (uint32_t)(i % W);
	reinterpret_cast<unsigned long long*>(out)[i] = reinterpret_cast<const unsigned long long*>(recs + order[r])[w];
}

// Replace *p (cap items of `item` bytes) by a buffer of at least `need` items without waiting for the device: the old buffer goes to
// the graveyard (freed when the call ends).  Contents are not kept.
int regrow(twk_hip_ctx* c, void** p, unsigned long long* cap, unsigned long long need, size_t item) {
	if (*cap >= need) return TWK_HIP_OK;
	const unsigned long long want = need + need / 4;
	void* q = nullptr;
	HIPCHK(c, dev_malloc(c, &q, (size_t)want * item));
	if (*p) c->graveyard.push_back(*p);
	*p = q; *cap = want;
	return TWK_HIP_OK;
}

// The second half of a band launch (see enqueue_tile): wait for its count kernel, size the survivor buffers by the candidates it
// found (no pair that was not a candidate can survive), enqueue the list math, Fisher's test, the sort of the survivors and the
// counters' copy - all on the compute stream.  The sort is over as many slots as there were candidates (unused slots carry the
// all-ones key, like records the Fisher cut-off drops: they sort behind the survivors): it does not need the survivor count, so it
// need not wait for the host - on the copy stream (finish_tile's sort_records) it waited for a CU until the *next* launch's
// persistent count kernel was through, 60 ms per launch of the 2,504 x 531,500 run.
int enqueue_band_math(twk_hip_ctx* c, Slot& s) {
	if (!s.deferred) return TWK_HIP_E_STATE;
	s.deferred = false;
	HIPCHK(c, hipEventSynchronize(s.ev_c1b));
	const unsigned long long cand = s.h_n_out[2];
	// Every candidate may survive: the survivor, key and sort buffers are sized by the candidates (220 bytes each).  Beyond 2^26 of them -
	// 15 GB per slot - or when the device cannot give the memory, the launch is treated like one whose list overflowed: its rows are redone
	// as matrix-sized tiles, which split further on their own overflow (the round-4 code returned E_NOMEM and took the run down).
	constexpr unsigned long long BAND_MAX_SURVIVORS = 1ull << 26;
	s.band_too_big = false;
	bool overflow = cand > s.cand_cap;                       // finish_tile reports it; nothing to compute here
	if (!overflow && cand > BAND_MAX_SURVIVORS) { s.band_too_big = true; overflow = true; }
	unsigned long long need = overflow ? 1 : std::max<unsigned long long>(cand, 1);
	if (c->opt.record_cap > 0) need = std::min<unsigned long long>(need, (unsigned long long)c->opt.record_cap);      // (test hook: forces the overflow path)
	auto grow_all = [&]() -> int {
		if (s.capacity < need) {       // (with some room: the next launch of the region will be about as rich)
			unsigned long long c1 = s.capacity, c2 = s.capacity, c3 = s.capacity;
			int rc = regrow(c, (void**)&s.out, &c1, need, sizeof(twk_hip_record)); if (rc) return rc;
			rc = regrow(c, (void**)&s.keys, &c2, need, sizeof(unsigned long long)); if (rc) return rc;
			rc = regrow(c, (void**)&s.vals, &c3, need, sizeof(uint32_t)); if (rc) return rc;
			s.capacity = std::min(c1, std::min(c2, c3));
		}
		{ const int rc = regrow(c, (void**)&s.sorted, &s.sorted_cap, need, sizeof(twk_hip_record)); if (rc) return rc; }
		if (c->band_sort_cap < need) {
			unsigned long long c1 = c->band_sort_cap, c2 = c->band_sort_cap;
			int rc = regrow(c, (void**)&c->d_band_keys, &c1, need, sizeof(unsigned long long)); if (rc) return rc;
			rc = regrow(c, (void**)&c->d_band_vals, &c2, need, sizeof(uint32_t)); if (rc) return rc;
			c->band_sort_cap = std::min(c1, c2);
		}
		size_t tmp = 0;
		HIPCHK(c, rocprim::radix_sort_pairs(nullptr, tmp, s.keys, c->d_band_keys, s.vals, c->d_band_vals, (size_t)need, 0u, 64u, c->s_compute));
		if (!c->d_band_tmp || tmp > c->band_tmp_bytes) {
			unsigned long long cap = c->band_tmp_bytes;
			const int rc = regrow(c, &c->d_band_tmp, &cap, std::max<size_t>(tmp, 4096), 1); if (rc) return rc;
			c->band_tmp_bytes = (size_t)cap;
		}
		return TWK_HIP_OK;
	};
	{
		int rc = grow_all();
		if (rc == TWK_HIP_E_NOMEM && !overflow) {            // no room for this many survivors: the fallback's tiles need far less at a time
			(void)hipGetLastError();
			s.band_too_big = true; overflow = true; need = 1;
			rc = grow_all();
		}
		if (rc) return rc;
	}
	s.cap_use = need;
	HIPCHK(c, hipEventRecord(s.ev_c0b, c->s_compute));
	if (!overflow) {
		HIPCHK(c, hipMemsetAsync(s.keys, 0xFF, (size_t)need * sizeof(unsigned long long), c->s_compute));
		HIPCHK(c, hipMemsetAsync(s.vals, 0, (size_t)need * sizeof(uint32_t), c->s_compute));
	}
	if (s.d_stats_dev && !overflow) {
		s.stats_host.out = s.out; s.stats_host.capacity = s.cap_use; s.stats_host.n_out = s.n_out;
		s.stats_host.keys = s.keys; s.stats_host.vals = s.vals;
		HIPCHK(c, hipMemcpyAsync(s.d_stats_dev, &s.stats_host, sizeof(StatsParams), hipMemcpyHostToDevice, c->s_compute));
		if (s.three) { const int rc = launch_recount(c, s.plane_set, s); if (rc) return rc; }
		if (s.deferred_unphased)
			hipLaunchKernelGGL(k_ld_stats_list_unphased, dim3(c->resident_blocks * 4), dim3(256), 0, c->s_compute, (const StatsParams*)s.d_stats_dev, (const uint32_t*)s.C,
			                   (const unsigned long long*)(s.n_out + 2), s.cand_cap);
		else
			hipLaunchKernelGGL(k_ld_stats_list, dim3(c->resident_blocks * 4), dim3(256), 0, c->s_compute, (const StatsParams*)s.d_stats_dev, (const uint32_t*)s.C,
			                   (const unsigned long long*)(s.n_out + 2), s.cand_cap);
		HIPCHK(c, hipGetLastError());
	}
	{ const int rc = launch_fisher(c, s.out, s.n_out, s.cap_use, s.minP, s.C, s.C_words, s.keys); if (rc) return rc; }
	s.presorted = false;
	if (!overflow) {
		size_t bytes = c->band_tmp_bytes;
		HIPCHK(c, rocprim::radix_sort_pairs(c->d_band_tmp, bytes, s.keys, c->d_band_keys, s.vals, c->d_band_vals, (size_t)need, 0u, 64u, c->s_compute));
		const unsigned long long words = need * (sizeof(twk_hip_record) / 8);
		hipLaunchKernelGGL(k_gather_records, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, c->s_compute, (const twk_hip_record*)s.out, (const uint32_t*)c->d_band_vals, need, s.sorted);
		HIPCHK(c, hipGetLastError());
		s.presorted = true;
	}
	HIPCHK(c, hipMemcpyAsync(s.h_n_out, s.n_out, N_SLOT_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->s_compute));
	HIPCHK(c, hipEventRecord(s.ev_s1, c->s_compute));
	return TWK_HIP_OK;
}

static_assert(sizeof(twk_hip_record) % 8 == 0, "record gather copies 8-byte words");

// recs[0..n) on the device, with their keys and positions -> c->d_sorted in (idxA, idxB) order, on stream st.
int sort_records(twk_hip_ctx* c, const twk_hip_record* recs, unsigned long long* keys_in, uint32_t* vals_in, unsigned long long n, bool any_dropped, hipStream_t st) {
	if (n > 0xFFFFFFFFull) return TWK_HIP_E_INVALID;
	if (c->sort_cap < n) {
		if (c->d_sort_keys) c->graveyard.push_back(c->d_sort_keys);       // (not hipFree: it waits for every stream of the device)
		if (c->d_sort_vals) c->graveyard.push_back(c->d_sort_vals);
		if (c->d_sorted) c->graveyard.push_back(c->d_sorted);
		c->d_sort_keys = nullptr; c->d_sort_vals = nullptr; c->d_sorted = nullptr; c->sort_cap = 0;
		const unsigned long long cap = std::max<unsigned long long>(n + n / 4, 1ull << 16);
		HIPCHK(c, dev_malloc(c, (void**)&c->d_sort_keys, (size_t)cap * sizeof(unsigned long long)));
		HIPCHK(c, dev_malloc(c, (void**)&c->d_sort_vals, (size_t)cap * sizeof(uint32_t)));
		HIPCHK(c, dev_malloc(c, (void**)&c->d_sorted, (size_t)cap * sizeof(twk_hip_record)));
		c->sort_cap = cap;
	}
	unsigned long long* keys_out = c->d_sort_keys;
	uint32_t* vals_out = c->d_sort_vals;
	const uint32_t bits_b = key_shift_for(c->M);
	// dropped records (Fisher cut-off) carry the all-ones key: sort every bit when there can be any, else only the
	// 2 * bits_b bits a (idxA, idxB) key uses
	const unsigned end_bit = any_dropped ? 64u : 2u * bits_b;
	size_t tmp = 0;
	HIPCHK(c, rocprim::radix_sort_pairs(nullptr, tmp, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, end_bit, st));
	if (!c->d_sort_tmp || tmp > c->sort_tmp_bytes) {          // (a null scratch pointer would turn the sort into another size query)
		if (c->d_sort_tmp) c->graveyard.push_back(c->d_sort_tmp);
		c->d_sort_tmp = nullptr; c->sort_tmp_bytes = 0;
		const size_t want = std::max<size_t>(tmp + tmp / 4, 4096);
		HIPCHK(c, dev_malloc(c, &c->d_sort_tmp, want));
		c->sort_tmp_bytes = want;
	}
	tmp = c->sort_tmp_bytes;
	HIPCHK(c, rocprim::radix_sort_pairs(c->d_sort_tmp, tmp, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, end_bit, st));
	const unsigned long long words = n * (sizeof(twk_hip_record) / 8);
	hipLaunchKernelGGL(k_gather_records, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, recs, vals_out, n, c->d_sorted);
	HIPCHK(c, hipGetLastError());
	return TWK_HIP_OK;
}

// Room for n more records behind the ones the device sink holds (grow-only, contents kept).
int ensure_device_keep(twk_hip_ctx* c, unsigned long long n_more) {
	const unsigned long long need = c->d_keep_n + n_more;
	if (need <= c->d_keep_cap) return TWK_HIP_OK;
	const unsigned long long cap = std::max<unsigned long long>(need + need / 2, 1ull << 16);
	twk_hip_record* p = nullptr;
	HIPCHK(c, dev_malloc(c, (void**)&p, (size_t)cap * sizeof(twk_hip_record)));
	if (c->d_keep_n) {
		const hipError_t e = hipMemcpyAsync(p, c->d_keep, (size_t)c->d_keep_n * sizeof(twk_hip_record), hipMemcpyDeviceToDevice, c->s_copy);
		if (e != hipSuccess) { (void)hipFree(p); HIPCHK(c, e); }
		HIPCHK(c, hipStreamSynchronize(c->s_copy));
	}
	if (c->d_keep) (void)hipFree(c->d_keep);
	c->d_keep = p; c->d_keep_cap = cap;
	return TWK_HIP_OK;
}

// The outlier watch: every count launch goes into a ring with its time per unit of work, and is compared with the median of the launches
// of its own kind and row length before it.  (Round 4 saw the same run take 1.7 x as long now and then, cause unknown; the watch names
// the launch, the clock its blocks ran at and how far apart the XCDs finished.)
constexpr size_t LAUNCH_RING = 4096;
void watch_launch(twk_hip_ctx* c, const Slot& s, float ms, const twk_hip_tile_desc& t) {
	twk_hip_launch_stat st{};
	st.ms = ms; st.row_pairs = s.row_pairs; st.candidates = (s.fused || s.three) ? s.h_n_out[2] : 0;
	st.shader_mhz = s.h_n_out[5] ? (double)s.h_n_out[4] / (double)s.h_n_out[5] * 100.0 : 0.0;
	st.words_per_row = c->planes[s.plane_set].W_live;
	st.kind = s.fused ? (s.three ? 4u : (c->planes[s.plane_set].rows && set_kind(s.plane_set) == PK_UNPHASED ? 3u : 2u)) : (s.three_plain ? 1u : 0u);
	unsigned long long lo = ~0ull, hi = 0;
	for (int x = 0; x < 8; ++x) if (s.h_n_out[8 + x]) { lo = std::min(lo, s.h_n_out[8 + x]); hi = std::max(hi, s.h_n_out[8 + x]); }
	st.xcd_finish_spread_us = hi ? (double)(hi - lo) / 100.0 : 0.0;
	auto cost = [](const twk_hip_launch_stat& x) { return x.row_pairs ? x.ms / ((double)x.row_pairs * (double)x.words_per_row * ((x.kind == 1 || x.kind == 4) ? 0.75 : 1.0)) : 0.0; };
	if (st.ms >= 0.3 && st.row_pairs) {
		std::vector<double> peers;
		for (const auto& x : c->launch_ring)
			if (x.kind == st.kind && x.words_per_row == st.words_per_row && x.ms >= 0.3 && !x.outlier) peers.push_back(cost(x));
		if (peers.size() >= 8) {
			std::nth_element(peers.begin(), peers.begin() + peers.size() / 2, peers.end());
			const double med = peers[peers.size() / 2];
			if (cost(st) > 1.4 * med) {
				st.outlier = 1; c->timing.outlier_launches += 1;
				if (c->opt.timeline) fprintf(stderr, "[outlier] count launch #%llu (kind %u, rows %u+%u x cols %u+%u, %llu row pairs of %u words): %.3f ms = %.2f x the median cost of its %zu peers; "
				                             "blocks ran at %.0f MHz; XCDs finished %.1f us apart; %llu candidates\n", (unsigned long long)c->launches_seen, st.kind, t.rowA0, t.nA, t.rowB0, t.nB,
				                             (unsigned long long)st.row_pairs, st.words_per_row, st.ms, cost(st) / med, peers.size(), st.shader_mhz, st.xcd_finish_spread_us, (unsigned long long)st.candidates);
			}
		}
	}
	if (c->launch_ring.size() < LAUNCH_RING) c->launch_ring.push_back(st);
	else c->launch_ring[c->launches_seen % LAUNCH_RING] = st;
	c->launches_seen += 1;
}

// kept sorted records on the device -> the host, through the pinned staging buffer, handed to `sink` in pieces of HOST_CHUNK records, each
// piece while the next is being copied (a launch may hold tens of millions of survivors: page-locking a buffer for all of them would cost
// more than the copy).  Called by the thread that finishes the launch, or by the delivery thread.
constexpr unsigned long long HOST_CHUNK = 1ull << 20;       // records per piece (109 MB)
int deliver_records(twk_hip_ctx* c, const twk_hip_record* sorted, unsigned long long kept, twk_hip_record_sink sink, void* user, hipStream_t st, double tl_wait,
                    char* err = nullptr, size_t err_len = 0) {
	if (!err) { err = c->err; err_len = sizeof(c->err); }
	auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
	double tl_copy = 0, tl_sink = 0;
	int rc;
	if (kept <= HOST_CHUNK) {
		rc = ensure_host_records(c, kept, err, err_len); if (rc) return rc;
		if (kept) HIPCHK_E(err, err_len, hipMemcpyAsync(c->h_recs, sorted, (size_t)kept * sizeof(twk_hip_record), hipMemcpyDeviceToHost, st));
		HIPCHK_E(err, err_len, hipStreamSynchronize(st));
		if (kept && sink(user, c->h_recs, kept)) { snprintf(err, err_len, "the record sink failed"); return TWK_HIP_E_INVALID; }
		return TWK_HIP_OK;
	}
	rc = ensure_host_records(c, 2 * HOST_CHUNK, err, err_len); if (rc) return rc;
	auto copy_piece = [&](unsigned long long first) -> hipError_t {
		const unsigned long long m = std::min(HOST_CHUNK, kept - first);
		return hipMemcpyAsync(c->h_recs + ((first / HOST_CHUNK) & 1) * HOST_CHUNK, sorted + first, (size_t)m * sizeof(twk_hip_record), hipMemcpyDeviceToHost, st);
	};
	HIPCHK_E(err, err_len, copy_piece(0));
	for (unsigned long long first = 0; first < kept; first += HOST_CHUNK) {
		auto t1 = std::chrono::steady_clock::now();
		HIPCHK_E(err, err_len, hipStreamSynchronize(st));                                          // piece `first` has arrived
		if (first + HOST_CHUNK < kept) HIPCHK_E(err, err_len, copy_piece(first + HOST_CHUNK));     // the next one travels while the sink works (into the other half)
		tl_copy += since(t1); t1 = std::chrono::steady_clock::now();
		if (sink(user, c->h_recs + ((first / HOST_CHUNK) & 1) * HOST_CHUNK, std::min(HOST_CHUNK, kept - first))) {
			(void)hipStreamSynchronize(st);
			snprintf(err, err_len, "the record sink failed");
			return TWK_HIP_E_INVALID;
		}
		tl_sink += since(t1);
	}
	if (c->opt.timeline) fprintf(stderr, "[timeline]   inside: waited %.3f ms for the launch, %.3f ms for copies, %.3f ms in the sink (%llu records)\n", tl_wait, tl_copy, tl_sink, kept);
	return TWK_HIP_OK;
}

// ---- the delivery thread (option async_delivery; the queue itself: twk_delivery.h) -----------------------------------------------
// During a region call with a sink a finished launch's sorted survivors are copied aside on the device (a millisecond a gigabyte) into
// one of at most `deliver_buffers` staging buffers and queued; a second thread takes the queue to the host, in order, through
// deliver_records.  The sink is then called from that thread - one call at a time, in the order of the launches, as before.
int discard_records(void*, const twk_hip_record*, uint64_t);
void delivery_begin(twk_hip_ctx* c, twk_hip_record_sink sink) {
	if (c->dl.active() || !sink || c->device_sink || !c->opt.async_delivery) return;
	c->dl_ops.c = c; c->dl_ops.n_alloc = 0; c->dl_ops.n_copy = 0;
	(void)c->dl.begin(&c->dl_ops, (size_t)c->opt.deliver_buffers);      // (no thread to be had: the caller's thread delivers, as with the option off)
}
// -> the delivery thread's result once everything queued has reached the sink; its error text into c->err
int delivery_end(twk_hip_ctx* c) {
	if (!c->dl.active()) return TWK_HIP_OK;
	const int rc = c->dl.end();
	if (rc && c->dl.error()[0]) snprintf(c->err, sizeof(c->err), "%s", c->dl.error());
	return rc;
}
// everything queued so far has reached its sink (the caller may then hand records over itself, in order: finish_tile does for launches with
// few survivors, which are not worth a staging copy and a thread hand-off)
int delivery_drain(twk_hip_ctx* c) { return c->dl.drain(); }
int stage_for_delivery(twk_hip_ctx* c, const twk_hip_record* sorted, unsigned long long kept, twk_hip_record_sink sink, void* user) {
	if (!kept || sink == discard_records) return TWK_HIP_OK;
	const int rc = c->dl.stage(sorted, kept, sink, user);
	if (rc == Delivery::STAGE_DELIVER_YOURSELF) {
		// no room for a copy on the device: this thread hands the records over itself, behind what is queued (as with the option off)
		(void)hipGetLastError();
		const int drc = delivery_drain(c); if (drc) return drc;
		return deliver_records(c, sorted, kept, sink, user, c->s_copy, 0.0);
	}
	return rc;
}
// The calling thread is out of device memory (ensure_slot, regrow, sort_records): everything queued goes to the sink, the idle staging
// buffers and what the call has outgrown so far are freed -> true when anything was given back (the allocation is then tried once more).
bool reclaim_device_memory(twk_hip_ctx* c) {
	(void)hipGetLastError();
	size_t bytes = c->dl.active() ? c->dl.reclaim() : 0;
	if (!c->graveyard.empty()) {
		(void)hipDeviceSynchronize();            // (outgrown buffers may still be read by launches in flight)
		for (void* p : c->graveyard) { (void)hipFree(p); ++bytes; }
		c->graveyard.clear();
	}
	return bytes != 0;
}

// Wait for slot s, account timing, put its records in (idxA, idxB) order and hand them on: appended to the device sink
// (!to_host), or through the pinned staging buffer to the host - left there whole (sink == null: c->h_recs, for the
// single-tile entry point), or handed to `sink` in pieces of HOST_CHUNK records, each piece while the next is being copied
// (a launch may hold tens of millions of survivors: page-locking a buffer for all of them would cost more than the copy).
int finish_tile(twk_hip_ctx* c, Slot& s, const twk_hip_tile_desc& t, unsigned long long* n_out, bool to_host = true,
                twk_hip_record_sink sink = nullptr, void* user = nullptr) {
	const auto tl0 = std::chrono::steady_clock::now();
	auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
	HIPCHK(c, hipEventSynchronize(s.ev_s1));
	const double tl_wait = since(tl0);
	struct FinishClock {          // timing.finish_ms: from here to whichever return hands the records on
		twk_hip_ctx* c; std::chrono::steady_clock::time_point t0;
		~FinishClock() { c->timing.finish_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
	} finish_clock{c, std::chrono::steady_clock::now()};
	float ms = 0;
	HIPCHK(c, hipEventElapsedTime(&ms, s.ev_c0, s.ev_c1));
	if (!s.is_list) watch_launch(c, s, ms, t);
	if (s.is_list && s.is_probe) { c->timing.probe_ms += ms; c->timing.probe_launches += 1; c->timing.probe_pairs += s.row_pairs; c->timing.candidates += s.h_n_out[2]; }
	else if (s.is_list) { c->timing.list_ms += ms; c->timing.list_launches += 1; c->timing.list_pairs += s.row_pairs; c->timing.candidates += s.h_n_out[2]; }
	else { c->timing.count_ms += ms; c->timing.count_launches += 1; c->timing.row_pairs += s.row_pairs;
	       c->timing.count_shader_cycles += s.h_n_out[4]; c->timing.count_wall_ticks += s.h_n_out[5]; }
	if (s.fused) { c->timing.fused_launches += 1; c->timing.candidates += s.h_n_out[2]; }
	if (s.three) {
		c->timing.three_launches += 1; c->timing.three_row_pairs += s.row_pairs; c->timing.recount_candidates += std::min<unsigned long long>(s.h_n_out[2], s.cand_cap);
		if (s.three_plain) c->timing.candidates += s.h_n_out[2];
		if (s.h_n_out[3]) {      // the recount disagrees with the contraction: never to be papered over
			snprintf(c->err, sizeof(c->err), "three-product contraction: %llu candidates whose (HH, S) differ from their recounted products (tile rows %u+%u, cols %u+%u)", s.h_n_out[3], t.rowA0, t.nA, t.rowB0, t.nB);
			return TWK_HIP_E_DEVICE;
		}
		// A launch this rich in candidates pays more for their recount than the fourth product costs: the rest of the call in the four-product
		// forms.  Measured at 2,504 samples, -u -w 1000000 (2.9 % of the pairs candidates): count kernel 32.1 -> 27.3 ms, but 58 M recounts at
		// 0.16 ns each (four 320-byte rows from L2 per candidate: 8 TB/s) add 9.3 ms to the math; the two meet near 1.2 %.
		// (against the variant pairs of the tiles the launch contracted - four plane-row pairs each - not the launch's rectangle: a window
		// launch's rectangle is mostly outside the window)
		if (c->opt.three != 2 && s.h_n_out[2] > std::max<unsigned long long>(s.row_pairs / 4 / 128, 4096)) c->three_ok = false;
	}
	float ms_all = 0;
	if (s.two_pass) {
		HIPCHK(c, hipEventElapsedTime(&ms, s.ev_c0b, s.ev_c1b));
		c->timing.count_ms += ms; c->timing.count_launches += 1; c->timing.row_pairs += s.row_pairs_b;
		float m1 = 0, m2 = 0;
		HIPCHK(c, hipEventElapsedTime(&m1, s.ev_c1, s.ev_c0b));
		HIPCHK(c, hipEventElapsedTime(&m2, s.ev_c1b, s.ev_s1));
		ms_all = m1 + m2; c->timing.stats_launches += 2;
	} else {
		HIPCHK(c, hipEventElapsedTime(&ms_all, s.was_deferred ? s.ev_c0b : s.ev_c1, s.ev_s1));
		c->timing.stats_launches += 1;
	}
	c->timing.stats_ms += ms_all;
	if (!s.is_list) c->timing.variant_pairs += pairs_in_tile(c, t);      // (the dense tiles that cover the list zone count its pairs)
	const unsigned long long n = *s.h_n_out;
	*n_out = n;
	if (s.was_deferred && s.band_too_big) {          // (enqueue_band_math: more candidates than a band launch may keep survivors for)
		snprintf(c->err, sizeof(c->err), "%llu candidates: beyond the survivor buffers of a band launch (tile rows %u+%u, cols %u+%u)", s.h_n_out[2], t.rowA0, t.nA, t.rowB0, t.nB);
		return TWK_HIP_E_OVERFLOW;
	}
	if ((s.fused || s.is_list || s.three_plain) && s.h_n_out[2] > s.cand_cap) {       // more candidates than the list holds
		s.cand_overflow = true;
		snprintf(c->err, sizeof(c->err), "%llu candidates for a list of %llu (tile rows %u+%u, cols %u+%u)", s.h_n_out[2], s.cand_cap, t.rowA0, t.nA, t.rowB0, t.nB);
		return TWK_HIP_E_OVERFLOW;
	}
	if (n > s.cap_use) {
		snprintf(c->err, sizeof(c->err), "%llu survivors for a buffer of %llu (tile rows %u+%u, cols %u+%u%s)", n, s.cap_use, t.rowA0, t.nA, t.rowB0, t.nB, s.is_list ? ", list pass" : "");
		return TWK_HIP_E_OVERFLOW;
	}
	if (!n) return TWK_HIP_OK;
	// records that failed the Fisher cut-off were only marked on the device (and counted): they sort behind the rest
	const unsigned long long dropped = std::min(s.h_n_out[1], n), kept = n - dropped;
	int rc = TWK_HIP_OK;
	if (!s.presorted) { rc = sort_records(c, s.out, s.keys, s.vals, n, dropped != 0, c->s_copy); if (rc) return rc; }
	const twk_hip_record* sorted = s.presorted ? s.sorted : c->d_sorted;      // (a band launch sorted its own behind Fisher's test: enqueue_band_math)
	*n_out = kept;
	if (to_host && sink && c->dl.active()) {
		// many survivors: the delivery thread takes them to the host (deliver_records) while this thread goes on with the launches; few
		// (a hand-over of a millisecond or two): this thread does, behind whatever is queued
		if (kept >= HOST_CHUNK / 4 || sink == discard_records) return stage_for_delivery(c, sorted, kept, sink, user);
		rc = delivery_drain(c); if (rc) return rc;
	}
	if (!to_host) {
		rc = ensure_device_keep(c, kept); if (rc) return rc;
		if (kept) HIPCHK(c, hipMemcpyAsync(c->d_keep + c->d_keep_n, sorted, (size_t)kept * sizeof(twk_hip_record), hipMemcpyDeviceToDevice, c->s_copy));
		c->d_keep_n += kept;
		HIPCHK(c, hipStreamSynchronize(c->s_copy));
		return TWK_HIP_OK;
	}
	if (!sink) {
		rc = ensure_host_records(c, kept); if (rc) return rc;
		if (kept) HIPCHK(c, hipMemcpyAsync(c->h_recs, sorted, (size_t)kept * sizeof(twk_hip_record), hipMemcpyDeviceToHost, c->s_copy));
		HIPCHK(c, hipStreamSynchronize(c->s_copy));
		return TWK_HIP_OK;
	}
	return deliver_records(c, sorted, kept, sink, user, c->s_copy, tl_wait);
}

// Rows [row0, row0 + n_rows) of the list zone of the allele-count-sorted phased set, synchronously on the spare slot: every
// pair (i, j), i < j < zone, inside the r2 band, as an intersection of two carrier lists (ld_list.hip.h) -> candidates ->
// the list math kernel -> Fisher -> sorted survivors (c->h_recs or the device sink, like a tile).
int run_list_block(twk_hip_ctx* c, const twk_hip_filters& f, bool unphased, uint32_t row0, uint32_t n_rows, uint32_t zone, int32_t window, uint32_t l_window,
                   const ColRange& cr, unsigned long long capacity, unsigned long long* n_out, bool to_host,
                   twk_hip_record_sink sink = nullptr, void* user = nullptr) {
	const int set = unphased ? PS_SORTED_U : PS_SORTED_P;
	const PlaneSet& ps = c->planes[set];
	Slot& s = c->slot[SYNC_SLOT];
	const uint64_t pairs_max = (uint64_t)n_rows * (zone - row0);
	const unsigned cand_words = unphased ? 6 : 3;              // (A, B, ALTALT) or (A, B, HH, HQ, QH, QQ)
	int rc = ensure_slot(c, s, (size_t)std::max<uint64_t>(cand_words * pairs_max, 1024), capacity); if (rc) return rc;
	if (!c->d_list_stats) HIPCHK(c, hipMalloc((void**)&c->d_list_stats, sizeof(StatsParams)));
	s.two_pass = false; s.fused = false; s.is_list = true; s.is_probe = false; s.was_deferred = false; s.presorted = false; s.cand_overflow = false; s.cand_cap = s.C_words / cand_words; s.minP = f.minP;
	twk_hip_tile_desc t{};
	t.rowA0 = row0; t.nA = n_rows; t.rowB0 = row0; t.nB = zone - row0; t.diag = 1; t.window = window; t.l_window = l_window;
	const StatsParams sp = make_stats(c, set, t, s, !unphased, 0, f, &cr);
	HIPCHK(c, hipMemsetAsync(s.n_out, 0, N_SLOT_COUNTERS * sizeof(unsigned long long), c->s_compute));
	HIPCHK(c, hipMemcpyAsync(c->d_list_stats, &sp, sizeof(sp), hipMemcpyHostToDevice, c->s_compute));
	ListWork w{};
	w.lists = ps.lists; w.stride = ps.list_max + 1; w.mac = ps.list_mac; w.flip = ps.list_flip; w.rowpop = ps.rowpop;
	w.n_list = zone; w.row0 = row0; w.n_rows = n_rows;
	w.col_hi = cr.d_hi; w.hi_a0 = cr.a0; w.hi_b0 = cr.b0;
	w.two_n = 2.0 * (double)c->N; w.cut = f.minR2 * (1.0 - 1e-6);
	w.cand = s.C; w.cap = s.cand_cap; w.n_cand = s.n_out + 2;
	// the widest row of the block decides the grid; lanes beyond a row's own reach leave at once
	uint32_t width = 0;
	for (uint32_t i = row0; i < row0 + n_rows; ++i) {
		const uint32_t lim = std::min<uint32_t>(zone, cr.hi ? cr.b0 + cr.hi[i - cr.a0] : zone);
		if (lim > i + 1) width = std::max(width, lim - i - 1);
	}
	s.row_pairs = 0;
	for (uint32_t i = row0; i < row0 + n_rows; ++i) {
		const uint32_t lim = std::min<uint32_t>(zone, cr.hi ? cr.b0 + cr.hi[i - cr.a0] : zone);
		if (lim > i + 1) s.row_pairs += lim - i - 1;              // pairs intersected (accounting only)
	}
	HIPCHK(c, hipEventRecord(s.ev_c0, c->s_compute));
	if (width) {
		if (unphased) hipLaunchKernelGGL(k_list_screen_unphased, dim3((width + 255) / 256, n_rows), dim3(256), 0, c->s_compute, w, c->N);
		else hipLaunchKernelGGL(k_list_screen, dim3((width + 255) / 256, n_rows), dim3(256), 0, c->s_compute, w);
		HIPCHK(c, hipGetLastError());
	}
	HIPCHK(c, hipEventRecord(s.ev_c1, c->s_compute));
	if (width) {
		if (unphased) hipLaunchKernelGGL(k_ld_stats_list_unphased, dim3(c->resident_blocks * 4), dim3(256), 0, c->s_compute, (const StatsParams*)c->d_list_stats,
		                                 (const uint32_t*)s.C, (const unsigned long long*)(s.n_out + 2), s.cand_cap);
		else hipLaunchKernelGGL(k_ld_stats_list, dim3(c->resident_blocks * 4), dim3(256), 0, c->s_compute, (const StatsParams*)c->d_list_stats, (const uint32_t*)s.C,
		                        (const unsigned long long*)(s.n_out + 2), s.cand_cap);
		HIPCHK(c, hipGetLastError());
	}
	{ const int rc = launch_fisher(c, s.out, s.n_out, s.cap_use, f.minP, s.C, s.C_words, s.keys); if (rc) return rc; }
	HIPCHK(c, hipMemcpyAsync(s.h_n_out, s.n_out, N_SLOT_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->s_compute));
	HIPCHK(c, hipEventRecord(s.ev_s1, c->s_compute));
	return finish_tile(c, s, t, n_out, to_host, sink, user);
}

// Zone rows [row0, row0 + n_rows) against the columns [col0, col0 + n_cols) outside the zone (col0 >= zone): every pair inside the
// r2 band as a probe of the row variant's carrier list into the column variant's plane row(s) (k_probe_screen, ld_list.hip.h) ->
// candidates -> the list math kernel -> Fisher -> sorted survivors, like a list block.
int run_probe_block(twk_hip_ctx* c, const twk_hip_filters& f, bool unphased, uint32_t row0, uint32_t n_rows, uint32_t zone, uint32_t col0, uint32_t n_cols,
                    int32_t window, uint32_t l_window, const ColRange& cr, unsigned long long capacity, unsigned long long* n_out, bool to_host,
                    twk_hip_record_sink sink, void* user) {
	const int set = unphased ? PS_SORTED_U : PS_SORTED_P;
	const PlaneSet& ps = c->planes[set];
	Slot& s = c->slot[SYNC_SLOT];
	const uint64_t pairs_max = (uint64_t)n_rows * n_cols;
	const unsigned cand_words = unphased ? 6 : 3;
	int rc = ensure_slot(c, s, (size_t)std::max<uint64_t>(cand_words * pairs_max, 1024), capacity); if (rc) return rc;
	if (!c->d_list_stats) HIPCHK(c, hipMalloc((void**)&c->d_list_stats, sizeof(StatsParams)));
	s.two_pass = false; s.fused = false; s.is_list = true; s.is_probe = true; s.was_deferred = false; s.presorted = false; s.cand_overflow = false; s.cand_cap = s.C_words / cand_words; s.minP = f.minP;
	twk_hip_tile_desc t{};
	t.rowA0 = row0; t.nA = n_rows; t.rowB0 = col0; t.nB = n_cols; t.diag = 0; t.window = window; t.l_window = l_window;
	const StatsParams sp = make_stats(c, set, t, s, !unphased, 0, f, &cr);
	HIPCHK(c, hipMemsetAsync(s.n_out, 0, N_SLOT_COUNTERS * sizeof(unsigned long long), c->s_compute));
	HIPCHK(c, hipMemcpyAsync(c->d_list_stats, &sp, sizeof(sp), hipMemcpyHostToDevice, c->s_compute));
	ProbeWork p{};
	ListWork& w = p.lw;
	w.lists = ps.lists; w.stride = ps.list_max + 1; w.mac = ps.list_mac; w.flip = ps.list_flip; w.rowpop = ps.rowpop;
	w.n_list = zone; w.row0 = row0; w.n_rows = n_rows;
	w.col_hi = cr.d_hi; w.hi_a0 = cr.a0; w.hi_b0 = cr.b0;
	w.two_n = 2.0 * (double)c->N; w.cut = f.minR2 * (1.0 - 1e-6);
	w.cand = s.C; w.cap = s.cand_cap; w.n_cand = s.n_out + 2;
	// through LDS (k_probe_lds_t: 512 rows x 4 columns a block, 2 columns of unphased planes); option probe_lds = 0: gathers straight from L2, one
	// column a block (the round-4 kernels: kept as the twin the LDS form is tested against)
	const bool via_lds = c->opt.probe_lds != 0;
	p.rows = ps.rows; p.W = ps.W; p.col0 = col0; p.n_cols = n_cols; p.n_row_blocks = via_lds ? (n_rows + PROBE_ROWS - 1) / PROBE_ROWS : (n_rows + 255) / 256;
	s.row_pairs = 0;
	for (uint32_t i = row0; i < row0 + n_rows; ++i) {
		const uint32_t lim = std::min<uint64_t>((uint64_t)col0 + n_cols, cr.hi ? (uint64_t)cr.b0 + cr.hi[i - cr.a0] : (uint64_t)col0 + n_cols);
		const uint32_t first = std::max(col0, i + 1);             // (columns inside the zone: those behind the row)
		if (lim > first) s.row_pairs += lim - first;              // pairs probed (accounting only)
	}
	constexpr uint32_t LDS_COLS_P = 4, LDS_COLS_U = 2;
	const uint32_t strip_cols = via_lds ? (unphased ? LDS_COLS_U : LDS_COLS_P) : 1;
	const uint64_t n_blocks = (uint64_t)p.n_row_blocks * ((n_cols + strip_cols - 1) / strip_cols);
	if (n_blocks > 0x7FFFFFFFull) return TWK_HIP_E_INVALID;
	HIPCHK(c, hipEventRecord(s.ev_c0, c->s_compute));
	if (s.row_pairs) {
		const dim3 grid((uint32_t)n_blocks);
		if (via_lds && unphased) hipLaunchKernelGGL(k_probe_lds_unphased_t<LDS_COLS_U>, grid, dim3(PROBE_ROWS), 0, c->s_compute, p);
		else if (via_lds) hipLaunchKernelGGL(k_probe_lds_t<LDS_COLS_P>, grid, dim3(PROBE_ROWS), 0, c->s_compute, p);
		else if (unphased) hipLaunchKernelGGL(k_probe_screen_unphased_t<4>, grid, dim3(256), 0, c->s_compute, p, c->N);
		else hipLaunchKernelGGL(k_probe_screen, grid, dim3(256), 0, c->s_compute, p);
		HIPCHK(c, hipGetLastError());
	}
	HIPCHK(c, hipEventRecord(s.ev_c1, c->s_compute));
	if (s.row_pairs) {
		if (unphased) hipLaunchKernelGGL(k_ld_stats_list_unphased, dim3(c->resident_blocks * 4), dim3(256), 0, c->s_compute, (const StatsParams*)c->d_list_stats,
		                                 (const uint32_t*)s.C, (const unsigned long long*)(s.n_out + 2), s.cand_cap);
		else hipLaunchKernelGGL(k_ld_stats_list, dim3(c->resident_blocks * 4), dim3(256), 0, c->s_compute, (const StatsParams*)c->d_list_stats, (const uint32_t*)s.C,
		                        (const unsigned long long*)(s.n_out + 2), s.cand_cap);
		HIPCHK(c, hipGetLastError());
	}
	{ const int rc2 = launch_fisher(c, s.out, s.n_out, s.cap_use, f.minP, s.C, s.C_words, s.keys); if (rc2) return rc2; }
	HIPCHK(c, hipMemcpyAsync(s.h_n_out, s.n_out, N_SLOT_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->s_compute));
	HIPCHK(c, hipEventRecord(s.ev_s1, c->s_compute));
	return finish_tile(c, s, t, n_out, to_host, sink, user);
}

// One tile, synchronously, on the spare slot; survivors end up in c->h_recs (sink == null), with `sink`, or in the device sink.
int run_tile_sync(twk_hip_ctx* c, int mode, const twk_hip_tile_desc& t, const twk_hip_filters& f,
                  unsigned long long capacity, unsigned long long* n_out, bool to_host = true, const ColRange* cr = nullptr,
                  twk_hip_record_sink sink = nullptr, void* user = nullptr) {
	Slot& s = c->slot[SYNC_SLOT];
	int rc = enqueue_tile(c, mode, t, f, s, capacity, cr); if (rc) return rc;
	rc = finish_tile(c, s, t, n_out, to_host, sink, user);
	if (rc == TWK_HIP_E_OVERFLOW && s.cand_overflow) {       // too many candidates for the fused / three-product form: through C, four products, for the rest of this call
		c->fused_ok = false; c->three_ok = false;
		rc = enqueue_tile(c, mode, t, f, s, capacity, cr); if (rc) return rc;
		rc = finish_tile(c, s, t, n_out, to_host, sink, user);
	}
	return rc;
}

int discard_records(void*, const twk_hip_record*, uint64_t) { return 0; }     // the sink of a caller that passed none

// A tile whose survivors overflowed the device buffer: redo it in row strips
// that cannot overflow (strip_rows * cols <= capacity).
// (cr: the region's column ranges - window, r2 band, list zone - so that a strip decides exactly the pairs its tile would have)
int redo_tile_in_strips(twk_hip_ctx* c, int mode, const twk_hip_tile_desc& t, const twk_hip_filters& f,
                        unsigned long long capacity, twk_hip_record_sink sink, void* user, uint64_t* n_recs, const ColRange* cr = nullptr) {
	const uint32_t strip = (uint32_t)std::max<unsigned long long>(1, capacity / std::max<uint32_t>(t.nB, 1));
	const bool diag = t.diag && t.rowA0 == t.rowB0;
	for (uint32_t r0 = 0; r0 < t.nA; r0 += strip) {
		const uint32_t nr = std::min(strip, t.nA - r0);
		twk_hip_tile_desc parts[2]; int np = 0;
		if (diag) {      // rows [r0, r0+nr): the diagonal sub-block, then the rectangle to its right
			twk_hip_tile_desc d = t; d.rowA0 = t.rowA0 + r0; d.nA = nr; d.rowB0 = d.rowA0; d.nB = nr; d.diag = 1;
			parts[np++] = d;
			if (r0 + nr < t.nB) {
				twk_hip_tile_desc r = t; r.rowA0 = t.rowA0 + r0; r.nA = nr; r.rowB0 = t.rowB0 + r0 + nr; r.nB = t.nB - (r0 + nr); r.diag = 0;
				parts[np++] = r;
			}
		} else {
			twk_hip_tile_desc r = t; r.rowA0 = t.rowA0 + r0; r.nA = nr; r.diag = 0;
			parts[np++] = r;
		}
		for (int k = 0; k < np; ++k) {
			unsigned long long n = 0;
			int rc = run_tile_sync(c, mode, parts[k], f, (unsigned long long)parts[k].nA * parts[k].nB, &n, !c->device_sink, cr, sink ? sink : discard_records, user);
			if (rc) return rc;
			*n_recs += n;
		}
	}
	return TWK_HIP_OK;
}

// Per-variant metadata of variants [first, first + count): device SoA + host mirror (synchronous copies).
int upload_meta(twk_hip_ctx* c, uint32_t first, uint32_t count, const twk_hip_variant_meta* meta) {
	std::vector<uint32_t> ac(count), an(count), pos(count), rid(count), miss(count);
	std::vector<double> hwe(count);
	for (uint32_t i = 0; i < count; ++i) {
		ac[i] = meta[i].ac; an[i] = meta[i].an; pos[i] = meta[i].pos; rid[i] = meta[i].rid;
		miss[i] = meta[i].missing ? 1 : 0; hwe[i] = meta[i].hwe;
		c->h_meta[first + i] = meta[i];
		if (meta[i].missing) c->any_missing = true;
	}
	HIPCHK(c, hipMemcpy(c->d_ac + first, ac.data(), (size_t)count * 4, hipMemcpyHostToDevice));
	HIPCHK(c, hipMemcpy(c->d_an + first, an.data(), (size_t)count * 4, hipMemcpyHostToDevice));
	HIPCHK(c, hipMemcpy(c->d_pos + first, pos.data(), (size_t)count * 4, hipMemcpyHostToDevice));
	HIPCHK(c, hipMemcpy(c->d_rid + first, rid.data(), (size_t)count * 4, hipMemcpyHostToDevice));
	HIPCHK(c, hipMemcpy(c->d_missing + first, miss.data(), (size_t)count * 4, hipMemcpyHostToDevice));
	HIPCHK(c, hipMemcpy(c->d_hwe + first, hwe.data(), (size_t)count * 8, hipMemcpyHostToDevice));
	return TWK_HIP_OK;
}

bool valid_mode(int m) { return m == TWK_HIP_MODE_PHASED || m == TWK_HIP_MODE_UNPHASED || m == TWK_HIP_MODE_AUTO; }
bool valid_tile(const twk_hip_ctx* c, const twk_hip_tile_desc* t) {
	if (!t || t->nA == 0 || t->nB == 0 || t->nA > 32768 || t->nB > 32768) return false;
	if ((uint64_t)t->rowA0 + t->nA > c->M || (uint64_t)t->rowB0 + t->nB > c->M) return false;
	if (t->diag && (t->rowA0 != t->rowB0 || t->nB < t->nA)) return false;
	return true;
}

}  // namespace

void* HipDeliveryOps::alloc(size_t bytes) {
	if (c->opt.deliver_fail_alloc_at && ++n_alloc == c->opt.deliver_fail_alloc_at) return nullptr;      // (test hook)
	void* p = nullptr;
	if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
	return p;
}
void HipDeliveryOps::release(void* p) { (void)hipFree(p); }
int HipDeliveryOps::copy_aside(void* dst, const void* src, size_t bytes) {
	if (c->opt.deliver_fail_copy_at && ++n_copy == c->opt.deliver_fail_copy_at) {      // (test hook)
		snprintf(c->err, sizeof(c->err), "staging copy of a launch's survivors failed (option deliver_fail_copy_at)");
		return TWK_HIP_E_DEVICE;
	}
	HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->s_copy));
	HIPCHK(c, hipStreamSynchronize(c->s_copy));
	return TWK_HIP_OK;
}
int HipDeliveryOps::deliver(const void* recs, uint64_t n, twk_hip_record_sink sink, void* user, char* err, size_t err_len) {
	return deliver_records(c, static_cast<const twk_hip_record*>(recs), n, sink, user, c->s_deliver, 0.0, err, err_len);
}
void HipDeliveryOps::thread_begin() {
	(void)hipSetDevice(c->device);
	if (!c->deliver_warm) {          // this thread's and this stream's first copy sets up a queue (tens of milliseconds): now, behind the first launches
		uint32_t x = 0;
		if (c->tickets && hipMemcpyAsync(&x, c->tickets, 4, hipMemcpyDeviceToHost, c->s_deliver) == hipSuccess) (void)hipStreamSynchronize(c->s_deliver);
		c->deliver_warm = true;
	}
}

// ================================ C ABI =======================================================
extern "C" {

int twk_hip_abi_version(void) { return TWK_HIP_ABI_VERSION; }

int twk_hip_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

const char* twk_hip_strerror(int code) {
	switch (code) {
	case TWK_HIP_OK: return "ok";
	case TWK_HIP_E_INVALID: return "invalid argument";
	case TWK_HIP_E_NOMEM: return "out of memory";
	case TWK_HIP_E_DEVICE: return "HIP device error";
	case TWK_HIP_E_OVERFLOW: return "record buffer too small";
	case TWK_HIP_E_STATE: return "call sequence error";
	default: return "unknown error";
	}
}

const char* twk_hip_last_error(const twk_hip_ctx* ctx) { return ctx ? ctx->err : ""; }

int twk_hip_ctx_create(int device, twk_hip_ctx** out) {
	if (!out) return TWK_HIP_E_INVALID;
	*out = nullptr;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return TWK_HIP_E_DEVICE;
	if (device < 0 || device >= n) return TWK_HIP_E_INVALID;
	twk_hip_ctx* c = new (std::nothrow) twk_hip_ctx();
	if (!c) return TWK_HIP_E_NOMEM;
	c->device = device;
	auto fail = [&](int code) { twk_hip_ctx_destroy(c); return code; };
	if (hipSetDevice(device) != hipSuccess) return fail(TWK_HIP_E_DEVICE);
	{
		int cus = 0;
		if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->resident_blocks = 2u * (uint32_t)cus;
	}
	if (hipStreamCreateWithFlags(&c->s_compute, hipStreamNonBlocking) != hipSuccess) return fail(TWK_HIP_E_DEVICE);
	{	// The copy stream carries the sort of a launch's survivors and their copy to the host, beside the next launch's persistent
		// count kernel on the compute stream.  At normal priority each of the sort's half-dozen dependent kernels had to wait for
		// the CUs the previous one freed, and lost them to the count kernel's waiting blocks every time - one sort kernel per gap
		// between count launches; at the highest priority its blocks are placed first.
		int lo = 0, hi = 0;
		if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; }
		if (hipStreamCreateWithPriority(&c->s_copy, hipStreamNonBlocking, hi) != hipSuccess) return fail(TWK_HIP_E_DEVICE);
		if (hipStreamCreateWithPriority(&c->s_deliver, hipStreamNonBlocking, hi) != hipSuccess) return fail(TWK_HIP_E_DEVICE);
	}
	if (hipMalloc((void**)&c->tickets, 2 * (PIPE_SLOTS + 1) * 8 * sizeof(uint32_t)) != hipSuccess) return fail(TWK_HIP_E_NOMEM);
	for (auto& s : c->slot) {
		hipEvent_t* evs[] = {&s.ev_c0, &s.ev_c1, &s.ev_s1, &s.ev_c0b, &s.ev_c1b};
		for (auto* e : evs) if (hipEventCreate(e) != hipSuccess) return fail(TWK_HIP_E_DEVICE);
		if (hipMalloc((void**)&s.n_out, N_SLOT_COUNTERS * sizeof(unsigned long long)) != hipSuccess) return fail(TWK_HIP_E_NOMEM);
		if (hipHostMalloc((void**)&s.h_n_out, N_SLOT_COUNTERS * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) return fail(TWK_HIP_E_NOMEM);
		std::memset(s.h_n_out, 0, N_SLOT_COUNTERS * sizeof(unsigned long long));
	}
	*out = c;
	return TWK_HIP_OK;
}

int twk_hip_ctx_destroy(twk_hip_ctx* c) {
	if (!c) return TWK_HIP_OK;
	(void)hipSetDevice(c->device);
	(void)hipDeviceSynchronize();
	free_problem(c);
	for (auto& s : c->slot) {
		hipEvent_t evs[] = {s.ev_c0, s.ev_c1, s.ev_s1, s.ev_c0b, s.ev_c1b};
		for (auto e : evs) if (e) (void)hipEventDestroy(e);
		if (s.n_out) (void)hipFree(s.n_out);
		if (s.h_n_out) (void)hipHostFree(s.h_n_out);
	}
	if (c->h_recs) (void)hipHostFree(c->h_recs);
	if (c->d_keep) (void)hipFree(c->d_keep);
	if (c->d_list_stats) (void)hipFree(c->d_list_stats);
	if (c->d_fisher_bins) (void)hipFree(c->d_fisher_bins);
	if (c->d_sort_keys) (void)hipFree(c->d_sort_keys);
	if (c->d_sort_vals) (void)hipFree(c->d_sort_vals);
	if (c->d_sorted) (void)hipFree(c->d_sorted);
	if (c->d_sort_tmp) (void)hipFree(c->d_sort_tmp);
	if (c->d_band_keys) (void)hipFree(c->d_band_keys);
	if (c->d_band_vals) (void)hipFree(c->d_band_vals);
	if (c->d_band_tmp) (void)hipFree(c->d_band_tmp);
	for (void* p : c->graveyard) (void)hipFree(p);
	c->graveyard.clear();
	for (void* p : c->host_graveyard) (void)hipHostFree(p);
	c->host_graveyard.clear();
	if (c->tickets) (void)hipFree(c->tickets);
	if (c->d_rle) (void)hipFree(c->d_rle);
	if (c->d_rle_desc) (void)hipFree(c->d_rle_desc);
	if (c->d_status) (void)hipFree(c->d_status);
	if (c->d_col_hi) (void)hipFree(c->d_col_hi);
	if (c->s_compute) (void)hipStreamDestroy(c->s_compute);
	if (c->s_copy) (void)hipStreamDestroy(c->s_copy);
	if (c->s_deliver) (void)hipStreamDestroy(c->s_deliver);
	delete c;
	return TWK_HIP_OK;
}

int twk_hip_host_alloc(size_t bytes, void** out) {
	if (!out || bytes == 0) return TWK_HIP_E_INVALID;
	*out = nullptr;
	return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? TWK_HIP_OK : TWK_HIP_E_NOMEM;
}
int twk_hip_host_free(void* p) {
	if (!p) return TWK_HIP_OK;
	return hipHostFree(p) == hipSuccess ? TWK_HIP_OK : TWK_HIP_E_DEVICE;
}

int twk_hip_set_problem(twk_hip_ctx* c, uint32_t n_samples, uint32_t n_variants) {
	if (!c || n_samples == 0 || n_variants == 0) return TWK_HIP_E_INVALID;
	if (n_samples > (1u << 30)) return TWK_HIP_E_INVALID;       // counts are u32 on the device
	HIPCHK(c, hipSetDevice(c->device));
	free_problem(c);
	c->N = n_samples; c->M = n_variants;
	c->M_alloc = round_up(n_variants, TILE) + TILE;
	c->Wp = round_up((uint32_t)((2ull * n_samples + 31) / 32), KC);
	c->Wu = round_up((n_samples + 31) / 32, KC);
	const size_t raw_bytes = (size_t)c->M_alloc * c->Wp * 4;
	HIPCHK(c, hipMalloc((void**)&c->raw, raw_bytes));
	HIPCHK(c, hipMemset(c->raw, 0, raw_bytes));
	const size_t m4 = (size_t)c->M_alloc * 4;
	HIPCHK(c, hipMalloc((void**)&c->d_ac, m4)); HIPCHK(c, hipMalloc((void**)&c->d_an, m4));
	HIPCHK(c, hipMalloc((void**)&c->d_pos, m4)); HIPCHK(c, hipMalloc((void**)&c->d_rid, m4));
	HIPCHK(c, hipMalloc((void**)&c->d_missing, m4)); HIPCHK(c, hipMalloc((void**)&c->d_hwe, (size_t)c->M_alloc * 8));
	HIPCHK(c, hipMemset(c->d_ac, 0, m4)); HIPCHK(c, hipMemset(c->d_an, 0, m4)); HIPCHK(c, hipMemset(c->d_pos, 0, m4));
	HIPCHK(c, hipMemset(c->d_rid, 0, m4)); HIPCHK(c, hipMemset(c->d_missing, 0, m4)); HIPCHK(c, hipMemset(c->d_hwe, 0, (size_t)c->M_alloc * 8));
	c->h_meta.assign(n_variants, twk_hip_variant_meta{});
	c->timing.words_per_row = 0;
	{	// every count Fisher's test sees is <= 2N (+ rounding of the unphased expected counts); beyond 2^26 entries lgamma itself is used
		const unsigned long long want = std::min<unsigned long long>(2ull * n_samples + 16, 1ull << 26);
		HIPCHK(c, hipMalloc((void**)&c->d_lfact, (size_t)want * sizeof(double)));
		c->lfact_n = (int)want;
		hipLaunchKernelGGL(k_build_lfact, dim3((unsigned)((want + 255) / 256)), dim3(256), 0, c->s_compute, c->d_lfact, c->lfact_n);
		HIPCHK(c, hipGetLastError());
		HIPCHK(c, hipStreamSynchronize(c->s_compute));
	}
	return TWK_HIP_OK;
}

int twk_hip_upload_bitvectors(twk_hip_ctx* c, uint32_t first, uint32_t count, const uint64_t* data,
                              const uint64_t* mask, size_t stride64, const twk_hip_variant_meta* meta) {
	if (!c || !data || !meta || count == 0) return TWK_HIP_E_INVALID;
	if (!c->raw) return TWK_HIP_E_STATE;
	const size_t w64 = ((size_t)2 * c->N + 63) / 64;
	if (stride64 < w64 || (uint64_t)first + count > c->M) return TWK_HIP_E_INVALID;
	// Validate everything before any device or context state changes.  A variant's missing-data
	// flags must agree: the reference picks the pair math from `an` (ld_engine.cpp:2775) and builds
	// the mask from `gt_missing` (core.cpp:356-358); both come from the same genotypes in any .twk
	// its importer writes, and the plane selection here relies on that.
	for (uint32_t i = 0; i < count; ++i) {
		if (meta[i].missing && !mask) return TWK_HIP_E_INVALID;
		if ((meta[i].an != 0) != (meta[i].missing != 0)) return TWK_HIP_E_INVALID;
	}
	HIPCHK(c, hipSetDevice(c->device));
	free_planes(c);                                             // derived planes are stale now
	HIPCHK(c, hipMemcpy2D(c->raw + (size_t)first * c->Wp, (size_t)c->Wp * 4, data, stride64 * 8, w64 * 8, count, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_clear_tail, dim3((c->Wp + 255) / 256, std::min<uint32_t>(count, 65535u)), dim3(256), 0, c->s_compute,
	                   c->raw + (size_t)first * c->Wp, c->Wp, c->N, count);
	HIPCHK(c, hipGetLastError());
	if (mask) {
		if (!c->rawmask) {
			const size_t raw_bytes = (size_t)c->M_alloc * c->Wp * 4;
			HIPCHK(c, hipMalloc((void**)&c->rawmask, raw_bytes));
			HIPCHK(c, hipMemset(c->rawmask, 0, raw_bytes));
		}
		HIPCHK(c, hipMemcpy2D(c->rawmask + (size_t)first * c->Wp, (size_t)c->Wp * 4, mask, stride64 * 8, w64 * 8, count, hipMemcpyHostToDevice));
		hipLaunchKernelGGL(k_clear_tail, dim3((c->Wp + 255) / 256, std::min<uint32_t>(count, 65535u)), dim3(256), 0, c->s_compute,
		                   c->rawmask + (size_t)first * c->Wp, c->Wp, c->N, count);
		HIPCHK(c, hipGetLastError());
	}
	int rc = upload_meta(c, first, count, meta); if (rc) return rc;
	HIPCHK(c, hipStreamSynchronize(c->s_compute));
	return TWK_HIP_OK;
}

int twk_hip_upload_rle(twk_hip_ctx* c, uint32_t first, uint32_t count, const void* bytes, size_t n_bytes,
                       const twk_hip_rle_desc* desc, const twk_hip_variant_meta* meta) {
	if (!c || !bytes || !desc || !meta || count == 0) return TWK_HIP_E_INVALID;
	if (!c->raw) return TWK_HIP_E_STATE;
	if ((uint64_t)first + count > c->M) return TWK_HIP_E_INVALID;
	// validate before anything changes (same rules as twk_hip_upload_bitvectors)
	bool any_mask = false;
	std::vector<RleDesc> dd(count);
	std::vector<uint32_t> chunk_base(count + 1);      // one block per chunk of RLE_CHUNK_BYTES run bytes, at least one per variant
	uint64_t n_chunks = 0;
	for (uint32_t i = 0; i < count; ++i) {
		const twk_hip_rle_desc& d = desc[i];
		if (d.width != 1 && d.width != 2 && d.width != 4) return TWK_HIP_E_INVALID;
		if (d.offset > n_bytes || (uint64_t)d.n_runs * d.width > n_bytes - d.offset) return TWK_HIP_E_INVALID;
		if ((meta[i].missing != 0) != (d.missing != 0)) return TWK_HIP_E_INVALID;
		if ((meta[i].an != 0) != (meta[i].missing != 0)) return TWK_HIP_E_INVALID;
		if (d.missing) any_mask = true;
		dd[i].off = d.offset; dd[i].n_runs = d.n_runs; dd[i].width_missing = (uint32_t)d.width | (d.missing ? 256u : 0u);
		chunk_base[i] = (uint32_t)n_chunks;
		n_chunks += std::max<uint64_t>(1, ((uint64_t)d.n_runs * d.width + RLE_CHUNK_BYTES - 1) / RLE_CHUNK_BYTES);
	}
	if (n_chunks > 0x7FFFFFFFull) return TWK_HIP_E_INVALID;
	chunk_base[count] = (uint32_t)n_chunks;
	HIPCHK(c, hipSetDevice(c->device));
	free_planes(c);                                             // derived planes are stale now
	auto grow = [&](void** p, size_t* cap, size_t need) -> hipError_t {
		if (*cap >= need) return hipSuccess;
		if (*p) (void)hipFree(*p);
		*p = nullptr; *cap = 0;
		const size_t want = need + need / 4;
		const hipError_t e = hipMalloc(p, want);
		if (e == hipSuccess) *cap = want;
		return e;
	};
	// descriptors | chunk sums (u64 per block) | first block of every variant
	const size_t desc_bytes = (size_t)count * sizeof(RleDesc) + (size_t)n_chunks * 8 + ((size_t)count + 1) * 4;
	HIPCHK(c, grow((void**)&c->d_rle, &c->d_rle_cap, n_bytes + 32));        // the kernel's dword loads run up to 19 bytes past the last run
	HIPCHK(c, grow((void**)&c->d_rle_desc, &c->d_rle_desc_cap, desc_bytes));
	if (!c->d_status) HIPCHK(c, hipMalloc((void**)&c->d_status, sizeof(int)));
	if (any_mask && !c->rawmask) {
		const size_t raw_bytes = (size_t)c->M_alloc * c->Wp * 4;
		HIPCHK(c, hipMalloc((void**)&c->rawmask, raw_bytes));
		HIPCHK(c, hipMemsetAsync(c->rawmask, 0, raw_bytes, c->s_compute));
	}
	RleDesc* d_desc = reinterpret_cast<RleDesc*>(c->d_rle_desc);
	unsigned long long* d_chunk_sum = reinterpret_cast<unsigned long long*>(c->d_rle_desc + (size_t)count * sizeof(RleDesc));
	uint32_t* d_chunk_base = reinterpret_cast<uint32_t*>(c->d_rle_desc + (size_t)count * sizeof(RleDesc) + (size_t)n_chunks * 8);
	HIPCHK(c, hipMemsetAsync(c->d_status, 0, sizeof(int), c->s_compute));
	HIPCHK(c, hipMemcpyAsync(c->d_rle, bytes, n_bytes, hipMemcpyHostToDevice, c->s_compute));
	HIPCHK(c, hipMemcpyAsync(d_desc, dd.data(), (size_t)count * sizeof(RleDesc), hipMemcpyHostToDevice, c->s_compute));
	// the kernel ORs the ALT / missing runs into rows of zeros
	HIPCHK(c, hipMemsetAsync(c->raw + (size_t)first * c->Wp, 0, (size_t)count * c->Wp * 4, c->s_compute));
	if (c->rawmask) HIPCHK(c, hipMemsetAsync(c->rawmask + (size_t)first * c->Wp, 0, (size_t)count * c->Wp * 4, c->s_compute));
	HIPCHK(c, hipMemcpyAsync(d_chunk_base, chunk_base.data(), ((size_t)count + 1) * 4, hipMemcpyHostToDevice, c->s_compute));
	hipLaunchKernelGGL(k_rle_chunk_sums, dim3((uint32_t)n_chunks), dim3(256), 0, c->s_compute, c->d_rle, d_desc, d_chunk_base, count, d_chunk_sum);
	HIPCHK(c, hipGetLastError());
	hipLaunchKernelGGL(k_inflate_rle, dim3((uint32_t)n_chunks), dim3(256), 0, c->s_compute, c->d_rle, d_desc, d_chunk_base, count, d_chunk_sum,
	                   c->raw, c->rawmask, c->Wp, c->N, first, c->d_status);
	HIPCHK(c, hipGetLastError());
	int status = 0;
	HIPCHK(c, hipMemcpyAsync(&status, c->d_status, sizeof(int), hipMemcpyDeviceToHost, c->s_compute));
	int rc = upload_meta(c, first, count, meta); if (rc) return rc;
	HIPCHK(c, hipStreamSynchronize(c->s_compute));
	if (status) { snprintf(c->err, sizeof(c->err), "run lengths of an uploaded variant do not add up to %u samples", c->N); return TWK_HIP_E_INVALID; }
	return TWK_HIP_OK;
}

int twk_hip_download_bitvectors(twk_hip_ctx* c, uint32_t first, uint32_t count, uint64_t* data, uint64_t* mask, size_t stride64) {
	if (!c || !data || count == 0) return TWK_HIP_E_INVALID;
	if (!c->raw) return TWK_HIP_E_STATE;
	const size_t w64 = ((size_t)2 * c->N + 63) / 64;
	if (stride64 < w64 || (uint64_t)first + count > c->M) return TWK_HIP_E_INVALID;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->s_compute));
	HIPCHK(c, hipMemcpy2D(data, stride64 * 8, c->raw + (size_t)first * c->Wp, (size_t)c->Wp * 4, w64 * 8, count, hipMemcpyDeviceToHost));
	if (mask) {
		if (c->rawmask) HIPCHK(c, hipMemcpy2D(mask, stride64 * 8, c->rawmask + (size_t)first * c->Wp, (size_t)c->Wp * 4, w64 * 8, count, hipMemcpyDeviceToHost));
		else for (uint32_t i = 0; i < count; ++i) std::memset(mask + (size_t)i * stride64, 0, w64 * 8);
	}
	return TWK_HIP_OK;
}

int twk_hip_generate_synthetic(twk_hip_ctx* c, uint64_t seed) { return twk_hip_generate_synthetic_planted(c, seed, 0, nullptr); }
int twk_hip_generate_synthetic_range(twk_hip_ctx* c, uint64_t seed, uint32_t first_variant) { return twk_hip_generate_synthetic_planted(c, seed, first_variant, nullptr); }

// twk_hip_plant -> the generator's own form (max_eps as a 32-bit fraction); false: out of range
static bool plant_of(const twk_hip_plant* p, SynthPlant& out) {
	out = SynthPlant{0, 0, 1, 0, 0};
	if (!p || p->n_planted == 0) return true;
	if (p->n_planted > p->half || p->mult == 0 || !(p->max_eps >= 0.0 && p->max_eps <= 0.5)) return false;
	out.n_planted = p->n_planted; out.half = p->half; out.mult = p->mult; out.offset = p->offset;
	out.eps_scale = (uint32_t)(p->max_eps * 4294967296.0);
	return true;
}

int twk_hip_generate_synthetic_planted(twk_hip_ctx* c, uint64_t seed, uint32_t first_variant, const twk_hip_plant* plant) {
	if (!c) return TWK_HIP_E_INVALID;
	if ((uint64_t)first_variant + (c ? c->M : 0) > 0xFFFFFFFFull) return TWK_HIP_E_INVALID;
	SynthPlant pl;
	if (!plant_of(plant, pl)) return TWK_HIP_E_INVALID;
	if (!c->raw) return TWK_HIP_E_STATE;
	HIPCHK(c, hipSetDevice(c->device));
	free_planes(c);
	if (c->rawmask) { (void)hipFree(c->rawmask); c->rawmask = nullptr; }
	c->any_missing = false;
	hipLaunchKernelGGL(k_synth, dim3((c->Wp + 255) / 256, std::min<uint32_t>(c->M, 65535u)), dim3(256), 0, c->s_compute, c->raw, c->Wp, c->N, c->M, seed, first_variant, pl);
	HIPCHK(c, hipGetLastError());
	// metadata: ac = popcount, pos = 1000 + 100 v, one contig, hwe = 1
	hipLaunchKernelGGL(k_row_popcount, dim3((c->M + 3) / 4), dim3(256), 0, c->s_compute, c->raw, c->Wp, c->M, c->d_ac);
	HIPCHK(c, hipGetLastError());
	std::vector<uint32_t> pos(c->M), zero(c->M, 0), ac(c->M);
	std::vector<double> hwe(c->M, 1.0);
	for (uint32_t v = 0; v < c->M; ++v) pos[v] = 1000u + 100u * (first_variant + v);
	HIPCHK(c, hipMemcpyAsync(c->d_pos, pos.data(), (size_t)c->M * 4, hipMemcpyHostToDevice, c->s_compute));
	HIPCHK(c, hipMemcpyAsync(c->d_an, zero.data(), (size_t)c->M * 4, hipMemcpyHostToDevice, c->s_compute));
	HIPCHK(c, hipMemcpyAsync(c->d_rid, zero.data(), (size_t)c->M * 4, hipMemcpyHostToDevice, c->s_compute));
	HIPCHK(c, hipMemcpyAsync(c->d_missing, zero.data(), (size_t)c->M * 4, hipMemcpyHostToDevice, c->s_compute));
	HIPCHK(c, hipMemcpyAsync(c->d_hwe, hwe.data(), (size_t)c->M * 8, hipMemcpyHostToDevice, c->s_compute));
	HIPCHK(c, hipMemcpyAsync(ac.data(), c->d_ac, (size_t)c->M * 4, hipMemcpyDeviceToHost, c->s_compute));
	HIPCHK(c, hipStreamSynchronize(c->s_compute));
	for (uint32_t v = 0; v < c->M; ++v) {
		twk_hip_variant_meta m{}; m.ac = ac[v]; m.pos = pos[v]; m.hwe = 1.0;
		c->h_meta[v] = m;
	}
	return TWK_HIP_OK;
}

uint32_t twk_synth_bitvector(uint64_t seed, uint32_t n_samples, uint32_t v, uint64_t* out_words) {
	return twk_synth_planted_bitvector(seed, n_samples, v, nullptr, out_words);
}

uint32_t twk_synth_planted_bitvector(uint64_t seed, uint32_t n_samples, uint32_t v, const twk_hip_plant* plant, uint64_t* out_words) {
	const size_t w64 = ((size_t)2 * n_samples + 63) / 64;
	std::memset(out_words, 0, w64 * 8);
	SynthPlant pl;
	if (!plant_of(plant, pl)) return 0;
	uint32_t src = v, eps_thr = 0;
	const bool copy = synth_plant_source(seed, pl, v, src, eps_thr);
	const uint32_t thr = synth_threshold(seed, src);
	uint32_t ac = 0;
	for (uint32_t s = 0; s < n_samples; ++s) {
		const uint64_t bits = synth_sample_bits(seed, src, s, thr) ^ (copy ? synth_flip_bits(seed, v, s, eps_thr) : 0u);
		out_words[(2ull * s) >> 6] |= bits << ((2ull * s) & 63);
		ac += (uint32_t)(bits & 1) + (uint32_t)(bits >> 1);
	}
	return ac;
}

int twk_synth_plant_source(uint64_t seed, const twk_hip_plant* plant, uint32_t v, uint32_t* src, double* eps) {
	SynthPlant pl;
	uint32_t s = v, thr = 0;
	const bool copy = plant_of(plant, pl) && synth_plant_source(seed, pl, v, s, thr);
	if (src) *src = copy ? s : v;
	if (eps) *eps = copy ? (double)thr / 4294967296.0 : 0.0;
	return copy ? 1 : 0;
}

int twk_hip_get_marginals(twk_hip_ctx* c, uint32_t* ac, uint32_t* n_het, uint32_t* n_hom, uint32_t* n_miss) {
	if (!c) return TWK_HIP_E_INVALID;
	if (!c->raw) return TWK_HIP_E_STATE;
	HIPCHK(c, hipSetDevice(c->device));
	if (ac) {
		const int kind = plane_kind_for(c, true);
		int rc = ensure_planes(c, kind); if (rc) return rc;
		const int P = planes_per_variant(kind);
		std::vector<uint32_t> rp((size_t)c->M * P);
		HIPCHK(c, hipMemcpy(rp.data(), c->planes[kind].rowpop, rp.size() * 4, hipMemcpyDeviceToHost));
		for (uint32_t v = 0; v < c->M; ++v) ac[v] = rp[(size_t)v * P];
	}
	if (n_het || n_hom || n_miss) {
		const int kind = plane_kind_for(c, false);
		int rc = ensure_planes(c, kind); if (rc) return rc;
		const int P = planes_per_variant(kind);
		std::vector<uint32_t> rp((size_t)c->M * P);
		HIPCHK(c, hipMemcpy(rp.data(), c->planes[kind].rowpop, rp.size() * 4, hipMemcpyDeviceToHost));
		for (uint32_t v = 0; v < c->M; ++v) {
			if (n_het) n_het[v] = rp[(size_t)v * P];
			if (n_hom) n_hom[v] = rp[(size_t)v * P + 1];
			if (n_miss) n_miss[v] = P == 3 ? rp[(size_t)v * P + 2] : 0;
		}
	}
	return TWK_HIP_OK;
}

// Buffers outgrown during a call (regrow, ensure_slot, sort_records): freed once nothing is in flight any more.
static void flush_graveyard(twk_hip_ctx* c) {
	if (c->graveyard.empty() && c->host_graveyard.empty()) return;
	(void)hipSetDevice(c->device);
	(void)hipDeviceSynchronize();
	for (void* p : c->graveyard) (void)hipFree(p);
	c->graveyard.clear();
	for (void* p : c->host_graveyard) (void)hipHostFree(p);
	c->host_graveyard.clear();
}

int twk_hip_count_tile(twk_hip_ctx* c, int mode, const twk_hip_tile_desc* t, uint64_t* out) {
	if (!c || !out || (mode != TWK_HIP_MODE_PHASED && mode != TWK_HIP_MODE_UNPHASED)) return TWK_HIP_E_INVALID;
	if (!c->raw) return TWK_HIP_E_STATE;
	if (!valid_tile(c, t)) return TWK_HIP_E_INVALID;
	HIPCHK(c, hipSetDevice(c->device));
	const bool phased = mode == TWK_HIP_MODE_PHASED;
	const int kind = plane_kind_for(c, phased);
	int rc = ensure_planes(c, kind); if (rc) return rc;
	Slot& s = c->slot[SYNC_SLOT];
	const Geometry g = tile_geometry(planes_per_variant(kind), *t);
	rc = ensure_slot(c, s, (size_t)g.rowsA * g.rowsB, 1); if (rc) return rc;
	uint64_t rp = 0;
	rc = launch_count(c, kind, *t, s, 0, s.ev_c0, s.ev_c1, &rp); if (rc) return rc;
	const int ncell = phased ? 4 : 9;
	const size_t n = (size_t)t->nA * t->nB * ncell;
	unsigned long long* d_cells = nullptr;
	HIPCHK(c, hipMalloc((void**)&d_cells, n * 8));
	StatsParams p = make_stats(c, kind, *t, s, phased, 0, twk_hip_filters{});
	hipLaunchKernelGGL(k_ld_cells, dim3((t->nB + 255) / 256, t->nA), dim3(256), 0, c->s_compute, p.tv, t->nA, t->nB, c->M, p.diag, phased ? 1 : 0, d_cells);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipMemcpyAsync(out, d_cells, n * 8, hipMemcpyDeviceToHost, c->s_compute);
	if (e == hipSuccess) e = hipStreamSynchronize(c->s_compute);
	(void)hipFree(d_cells);
	flush_graveyard(c);
	HIPCHK(c, e);
	return TWK_HIP_OK;
}

int twk_hip_ld_tile(twk_hip_ctx* c, int mode, const twk_hip_tile_desc* t, const twk_hip_filters* f,
                    twk_hip_record* out, uint64_t capacity, uint64_t* n_out, uint64_t* n_pairs) {
	if (!c || !f || !n_out || !valid_mode(mode) || (capacity && !out)) return TWK_HIP_E_INVALID;
	if (!c->raw) return TWK_HIP_E_STATE;
	if (!valid_tile(c, t)) return TWK_HIP_E_INVALID;
	HIPCHK(c, hipSetDevice(c->device));
	c->fused_ok = true; c->three_ok = true;
	unsigned long long n = 0;
	int rc = run_tile_sync(c, mode, *t, *f, std::max<unsigned long long>(capacity, 1), &n);
	flush_graveyard(c);
	*n_out = n;
	if (n_pairs) *n_pairs = pairs_in_tile(c, *t);
	if (rc == TWK_HIP_OK && n > capacity) rc = TWK_HIP_E_OVERFLOW;
	if (rc) return rc;
	if (n) std::memcpy(out, c->h_recs, (size_t)n * sizeof(twk_hip_record));
	return TWK_HIP_OK;
}

int twk_hip_ld_all(twk_hip_ctx* c, int mode, const twk_hip_filters* f, uint32_t part, uint32_t n_parts,
                   uint32_t tile_variants, int32_t window, uint32_t l_window, twk_hip_record_sink sink,
                   void* user, uint64_t* n_pairs, uint64_t* n_records) {
	if (!c) return TWK_HIP_E_INVALID;
	return twk_hip_ld_region(c, mode, f, 0, c->M, 0, c->M, 1, part, n_parts, tile_variants, window, l_window,
	                         sink, user, n_pairs, n_records);
}

// ---- one region in one index space ------------------------------------------------------------------------------------------
// (the file order, or - modes MODE_INT_GROUPED / MODE_INT_SORTED_* - the order of a regrouped / sorted plane set; a0 / b0 / nA / nB and
// the tiles are positions in that space).  region_impl asks the planner (ld_plan.h: shard, reach of every row, launches - host-only
// arithmetic, tested without a GPU) and hands the plan to a RegionRun, which executes it: zone passes, the decision between three and
// four products, the launch pipeline with its fallbacks.
namespace {

struct RegionRun {
	twk_hip_ctx* c; int mode; const twk_hip_filters* f;
	const PlanGeom& g; const PlanEnv& env; RegionPlan& plan;
	twk_hip_record_sink sink; void* user;
	std::chrono::steady_clock::time_point t_origin = std::chrono::steady_clock::now();
	ColRange col_range;
	unsigned long long cap_default = 1ull << 24;
	uint64_t tot_pairs = 0, tot_recs = 0;
	std::vector<char> want_three;             // per launch: the three-product form may be used (decide_three_by_samples)

	RegionRun(twk_hip_ctx* c_, int mode_, const twk_hip_filters* f_, const PlanGeom& g_, const PlanEnv& e_, RegionPlan& p_, twk_hip_record_sink sink_, void* user_)
		: c(c_), mode(mode_), f(f_), g(g_), env(e_), plan(p_), sink(sink_ ? sink_ : discard_records), user(user_) {}

	void mark(const char* what, size_t i, unsigned long long x = 0) const {       // "timeline" option: where the host's time goes
		if (!c->opt.timeline) return;
		fprintf(stderr, "[timeline] %9.3f ms  %s %zu  %llu\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_origin).count(), what, i, x);
	}
	const ColRange* cr() const { return plan.windowed ? &col_range : nullptr; }

	// Survivor buffer of a launch (worst case every pair of a tile survives: capped, split on overflow) and the reach of the rows on the device.
	int prepare() {
		unsigned long long worst = 0;
		for (const auto& t : plan.mine) worst = std::max<unsigned long long>(worst, (unsigned long long)t.nA * t.nB);
		cap_default = std::min<unsigned long long>(worst ? worst : 1, 1ull << 24);
		if (c->opt.record_cap > 0) cap_default = std::min<unsigned long long>(cap_default, (unsigned long long)c->opt.record_cap);   // test hook: force the overflow / strip path
		if (plan.windowed) { col_range.lo = plan.lo.data(); col_range.hi = plan.hi.data(); col_range.a0 = g.a0; col_range.b0 = g.b0; }
		if (env.screen && g.nA) {
			if (c->d_col_hi_cap < g.nA) {
				if (c->d_col_hi) (void)hipFree(c->d_col_hi);
				c->d_col_hi = nullptr; c->d_col_hi_cap = 0;
				HIPCHK(c, hipMalloc((void**)&c->d_col_hi, (size_t)g.nA * 4));
				c->d_col_hi_cap = g.nA;
			}
			HIPCHK(c, hipMemcpy(c->d_col_hi, plan.hi.data(), (size_t)g.nA * 4, hipMemcpyHostToDevice));
			col_range.d_hi = c->d_col_hi; col_range.n_hi = g.nA;
		}
		return TWK_HIP_OK;
	}

	// The list zone of an allele-count-sorted set (long rows only): its pairs are intersected as carrier lists, block of rows after
	// block of rows, and the rows with the shortest lists probe every column behind them, before the tiles that are left are contracted.
	int run_zone_passes() {
		if (!(env.screen && g.a0 == 0 && g.b0 == 0)) return TWK_HIP_OK;
		const bool unphased = env.screen == 2;
		const PlaneSet& ps = c->planes[unphased ? PS_SORTED_U : PS_SORTED_P];
		const uint32_t zone = std::min(ps.n_list, g.nA), r0 = plan.r0, r1 = plan.r1;
		if (!(zone >= 2 && ps.lists)) return TWK_HIP_OK;
		col_range.list_zone = zone;
		// rows whose list is short enough to probe with (the first n_probe of the zone) take every column behind them as probes,
		// the zone's own included ("probe_zone": a probe walks one list and tests bits of the partner's row, a merge walks two
		// lists in step - 0.28 against 1.05 ns a pair on the unphased zone of the 1 M x 50 k cohort run); the merges are left
		// with the zone's last rows
		const uint32_t pz_first = (c->opt.probe && c->opt.probe_zone) ? std::min(ps.n_probe, zone) : 0;
		const uint32_t lr0 = std::min(std::max(r0, pz_first), zone), lr1 = std::min(r1, zone);
		uint32_t rows_per = (uint32_t)std::max<uint64_t>(64, std::min<uint64_t>(32768, (1ull << 25) / zone));
		unsigned long long cap_list = cap_default;
		for (uint32_t row = lr0; row < lr1;) {
			const uint32_t nr = std::min(rows_per, lr1 - row);
			unsigned long long nrec = 0;
			const int rc = run_list_block(c, *f, unphased, row, nr, zone, g.window, g.l_window, col_range, cap_list, &nrec, !c->device_sink, sink, user);
			if (rc == TWK_HIP_E_OVERFLOW && nr > 1) { rows_per = std::max<uint32_t>(1, nr / 2); continue; }      // more survivors than the buffer holds: fewer rows
			if (rc == TWK_HIP_E_OVERFLOW && cap_list < zone) { cap_list = zone; continue; }                      // one row: it cannot have more than `zone` partners
			if (rc) return rc;
			tot_recs += nrec;
			row += nr;
		}
		mark("carrier-list pass done, zone", zone, tot_recs);
		// ... and the zone's rows against the columns beyond the zone: probes of the row variant's carriers into the column
		// variant's row (K1's asymmetric path, ld_engine.cpp:230-242; measured to win for every list the zone keeps,
		// ld_list.hip.h), so that no tile row inside the zone is contracted at all.
		const uint32_t pzone = c->opt.probe ? std::min(ps.n_probe, zone) : 0;
		const bool zone_cols = c->opt.probe_zone != 0;             // the probes' columns start behind the row, not behind the zone
		if (!(pzone && (zone < g.nB || zone_cols))) return TWK_HIP_OK;
		col_range.probe_zone = pzone;
		unsigned long long cap_probe = cap_default;
		uint32_t rows_cap = 32768;                                   // halved when a block's survivors outgrow the buffer
		const uint32_t pr0 = std::min(r0, pzone), pr1 = std::min(r1, pzone);
		// the columns a block's rows reach (the band limit never decreases along the rows)
		auto reach = [&](uint32_t last_row) -> uint32_t { return (uint32_t)std::min<uint64_t>(g.nB, col_range.hi ? (uint64_t)col_range.b0 + col_range.hi[last_row - col_range.a0] : g.nB); };
		for (uint32_t row = pr0; row < pr1;) {
			uint32_t nr = std::min<uint32_t>(pr1 - row, rows_cap);
			const uint32_t first = zone_cols ? row + 1 : zone;       // first column of the block (a row's own start at row + 1: the kernel's j > i)
			while (nr > 256 && (uint64_t)nr * (reach(row + nr - 1) > first ? reach(row + nr - 1) - first : 0) > (1ull << 25)) nr = std::max<uint32_t>(256, nr / 2);
			const uint32_t lim = reach(row + nr - 1);
			if (lim <= first) { row += nr; continue; }
			unsigned long long nrec = 0;
			const int rc = run_probe_block(c, *f, unphased, row, nr, zone, first, lim - first, g.window, g.l_window, col_range, cap_probe, &nrec, !c->device_sink, sink, user);
			if (rc == TWK_HIP_E_OVERFLOW && nr > 1) { rows_cap = std::max<uint32_t>(1, nr / 2); continue; }
			if (rc == TWK_HIP_E_OVERFLOW && cap_probe < (unsigned long long)(lim - first)) { cap_probe = lim - first; continue; }
			if (rc) return rc;
			tot_recs += nrec;
			row += nr;
		}
		mark("probe pass done, zone", pzone, tot_recs);
		return TWK_HIP_OK;
	}

	// The three-product form through a count matrix (long rows) pays only where candidates are few: a launch whose list overflows is redone
	// with four products, three-product launches already in flight behind it included (1 M x 50,000 cohort run, calc -u: 977 -> 1,426 ms of
	// count kernel with three such launches wasted), and how many pairs are candidates differs from launch to launch - in allele-count order
	// the launches over the common variants hold nearly all of them.  So every launch is decided by a sample of its own, taken before the
	// pipeline starts: a sub-tile of at most 384 x 384 variants from its middle through the same kernels (half a millisecond), three
	// products if at most 1 pair in 256 of the sample is a candidate (1 in 128 for fused launches, whose recount reads short rows from L2).  (Regions of more than 64 launches sample every k-th one; the others
	// follow their nearest sampled neighbour.)
	int decide_three_by_samples() {
		const std::vector<twk_hip_tile_desc>& mine = plan.mine;
		const size_t n = mine.size();
		want_three.assign(n, 1);
		if (mine.empty() || !c->three_ok || c->opt.three != 1) return TWK_HIP_OK;
		const TilePlan pl = plan_for(c, mode);
		const bool eligible = !pl.phased1 && pl.set2 < 0 && set_kind(pl.set1) == PK_UNPHASED && f->minR2 > 1e-6 && f->minR2 <= 1.0;
		if (!eligible) return TWK_HIP_OK;
		// (fused launches - short rows - are sampled as well: they keep their candidates whatever their number, but past ~1.2 % of the pairs
		// the recount costs more than the third product saves: 2,504 samples, -u -w 1000000, 2.9 % candidates: 27.4 + 30.3 ms of count kernel and
		// math with three products in the first of two launches, 30 + 21 ms with four in both)
		const unsigned long long dense_one_in = env.fused ? 128 : 256;
		const size_t step = (n + 63) / 64;
		for (size_t i0 = 0; i0 < n; i0 += step) {
			const size_t pick = std::min(n - 1, i0 + step / 2);
			const twk_hip_tile_desc& t0 = mine[pick];
			twk_hip_tile_desc st = t0;
			const uint32_t h = std::min<uint32_t>(384, t0.nA), row_c = t0.rowA0 + (t0.nA - h) / 2;
			st.rowA0 = row_c; st.nA = h;
			if (t0.diag && t0.rowA0 == t0.rowB0) { st.rowB0 = row_c; st.nB = std::min<uint32_t>(h, t0.rowB0 + t0.nB - row_c); st.diag = 1; }
			else { const uint32_t wv = std::min<uint32_t>(384, t0.nB); st.rowB0 = t0.rowB0 + (t0.nB - wv) / 2; st.nB = wv; st.diag = 0; }
			const twk_hip_timing keep_timing = c->timing;
			const size_t keep_ring = c->launch_ring.size(); const uint64_t keep_seen = c->launches_seen;
			Slot& ss = c->slot[SYNC_SLOT];
			unsigned long long nrec = 0;
			c->sampling = true;
			int rc = enqueue_tile(c, mode, st, *f, ss, std::max<unsigned long long>((unsigned long long)st.nA * st.nB, 1), cr());
			c->sampling = false;
			if (rc == TWK_HIP_OK) rc = finish_tile(c, ss, st, &nrec, true, discard_records, nullptr);
			const unsigned long long cand = ss.h_n_out[2], sample_pairs = std::max<uint64_t>(pairs_in_tile(c, st), 1);
			c->timing = keep_timing;                 // (a sample is not a launch of the run: neither in the timing nor in the launch log)
			if (c->launch_ring.size() > keep_ring && c->launches_seen == keep_seen + 1 && keep_seen < LAUNCH_RING) c->launch_ring.pop_back();
			c->launches_seen = keep_seen;
			c->three_ok = true;                      // (a sample's own overflow decides its launch, not the call)
			const bool dense = rc == TWK_HIP_E_OVERFLOW || (rc == TWK_HIP_OK && cand * dense_one_in > sample_pairs);
			if (rc != TWK_HIP_OK && rc != TWK_HIP_E_OVERFLOW) return rc;
			for (size_t i = i0; i < std::min(n, i0 + step); ++i) want_three[i] = dense ? 0 : 1;
			mark("three-product sample of launch: candidates", pick, cand);
		}
		return TWK_HIP_OK;
	}

	// One matrix-sized tile, synchronously, with its fallbacks: the fused form's candidate list overflowed -> through C (and
	// the rest of the call as well); more survivors than the buffer holds -> row strips.
	int run_tile_with_fallbacks(const twk_hip_tile_desc& t) {
		unsigned long long nrec = 0;
		int r = run_tile_sync(c, mode, t, *f, cap_default, &nrec, !c->device_sink, cr(), sink, user);
		if (r == TWK_HIP_E_OVERFLOW) {
			uint64_t nr = 0;
			r = redo_tile_in_strips(c, mode, t, *f, c->slot[SYNC_SLOT].cap_use, sink, user, &nr, cr());
			nrec = nr;
		}
		if (r == TWK_HIP_OK) tot_recs += nrec;
		return r;
	}
	// a band launch that cannot run (or has overflowed) as the matrix-sized tiles of its rows
	int run_band_as_matrix_tiles(const BandLaunch& b) {
		std::vector<twk_hip_tile_desc> sub;
		plan_matrix_tiles(env, g, plan, b.xa, b.xb, sub);
		for (const auto& t : sub) { const int r = run_tile_with_fallbacks(t); if (r) return r; }
		return TWK_HIP_OK;
	}

	// The second half of band launch i (its pair math: once the count kernel's candidate count is known).
	int band_math(size_t i, const std::vector<char>& skipped) {
		if (i < plan.bands.size() && !skipped[i]) {
			mark("math: wait for count of launch", i);
			const int r = enqueue_band_math(c, c->slot[i % PIPE_SLOTS]);
			mark("math enqueued for launch", i, c->slot[i % PIPE_SLOTS].h_n_out[2]);
			return r;
		}
		return TWK_HIP_OK;
	}

	// Launch `done` has been waited for: deliver its records, or run its fallbacks.
	int finish_launch(size_t done, const std::vector<char>& skipped) {
		const std::vector<twk_hip_tile_desc>& mine = plan.mine;
		Slot& s = c->slot[done % PIPE_SLOTS];
		unsigned long long nrec = 0;
		const BandLaunch* b = done < plan.bands.size() ? &plan.bands[done] : nullptr;
		mark("finish: wait for launch", done);
		int rc;
		if (b) {
			rc = skipped[done] ? TWK_HIP_E_OVERFLOW : finish_tile(c, s, mine[done], &nrec, !c->device_sink, sink, user);
			mark("finished (records delivered) launch", done, nrec);
			if (rc == TWK_HIP_E_OVERFLOW) rc = run_band_as_matrix_tiles(*b);      // candidates or survivors beyond the launch's buffers
			else if (rc == TWK_HIP_OK) tot_recs += nrec;
			return rc;
		}
		rc = finish_tile(c, s, mine[done], &nrec, !c->device_sink, sink, user);
		if (rc == TWK_HIP_E_OVERFLOW && s.cand_overflow) {     // the fused / three-product form's candidate list overflowed: this tile again through C, four products
			c->fused_ok = false; c->three_ok = false;            // (and the tiles not yet enqueued as well)
			rc = run_tile_sync(c, mode, mine[done], *f, cap_default, &nrec, !c->device_sink, cr(), sink, user);
		}
		if (rc == TWK_HIP_E_OVERFLOW) {
			uint64_t nr = 0;
			rc = redo_tile_in_strips(c, mode, mine[done], *f, s.cap_use, sink, user, &nr, cr());
			if (rc) return rc;
			tot_recs += nr;
		} else if (rc) {
			return rc;
		} else {
			tot_recs += nrec;
		}
		mark("finished (records delivered) launch", done, tot_recs);
		return TWK_HIP_OK;
	}

	// Software pipeline over the launches of this shard, PIPE_SLOTS deep.  The pair math of a band launch follows once its count kernel is
	// done - with the next launch's count kernel already queued behind it, so that the device has work while the host waits for the
	// candidate count.
	int run_pipeline() {
		const std::vector<twk_hip_tile_desc>& mine = plan.mine;
		const size_t n = mine.size();
		std::vector<char> skipped(n, 0);
		size_t issued = 0, done = 0, math_issued = 0;      // band launches [0, math_issued) have had the second half of their work enqueued
		int rc = TWK_HIP_OK;
		auto issue_next = [&]() -> int {
			const BandLaunch* b = issued < plan.bands.size() ? &plan.bands[issued] : nullptr;
			if (b && !c->fused_ok) skipped[issued] = 1;                 // an earlier launch gave the fused form up: this one goes the matrix way when its turn comes
			else {
				mark("enqueue launch", issued);
				const bool three_was = c->three_ok;
				c->three_ok = three_was && want_three[issued] != 0;       // (the launch's own sample)
				const int r = enqueue_tile(c, mode, mine[issued], *f, c->slot[issued % PIPE_SLOTS], b ? 1 : cap_default, cr(), b ? b->list_words : 0);
				c->three_ok = three_was;
				if (r) return r;
				mark("enqueued launch", issued);
			}
			++issued;
			return TWK_HIP_OK;
		};
		while (done < n) {
			while (issued < n && issued < done + PIPE_SLOTS) {
				rc = issue_next(); if (rc) return rc;
				while (math_issued + 1 < issued) { rc = band_math(math_issued, skipped); if (rc) return rc; ++math_issued; }
			}
			while (math_issued <= done && math_issued < issued) { rc = band_math(math_issued, skipped); if (rc) return rc; ++math_issued; }
			rc = finish_launch(done, skipped);
			if (rc) return rc;
			tot_pairs += pairs_in_tile(c, mine[done]);
			++done;
			if (c->progress_cb && !c->progress_muted) c->progress_cb(c->progress_user, tot_pairs, (uint32_t)done, (uint32_t)n);
		}
		return TWK_HIP_OK;
	}
};

// What the planner needs to know of the context for this mode.
int plan_env_for(twk_hip_ctx* c, int mode, const twk_hip_filters* f, PlanEnv& env) {
	env = PlanEnv();
	if (mode == MODE_INT_GROUPED || mode == MODE_INT_SORTED_P || mode == MODE_INT_SORTED_U) {
		const int set = mode == MODE_INT_GROUPED ? PS_GROUPED : mode == MODE_INT_SORTED_P ? PS_SORTED_P : PS_SORTED_U;
		const int rc = ensure_planes(c, set); if (rc) return rc;
		env.ids = c->planes[set].h_ids.data();
	}
	// r2 screen (TWK_HIP_OPT_R2_SCREEN): the region is a triangle over the leading, missing-free part of an allele-count-sorted set
	env.screen = (mode == MODE_INT_SORTED_P) ? 1 : (mode == MODE_INT_SORTED_U) ? 2 : 0;
	if (env.screen) { const int rc = ensure_popcounts(c); if (rc) return rc; env.popc = c->h_popc.data(); }
	const TilePlan pl = plan_for(c, mode);
	env.n_samples = c->N; env.Pmax = pl.Pmax; env.resident_blocks = c->resident_blocks; env.minR2 = f->minR2; env.phased_math = pl.phased1;
	env.meta = c->h_meta.data();
	env.fused = fused_form_applies(c, mode, *f);
	if (env.fused) env.nchunks = c->planes[pl.set1].W / KC;
	env.band_launch = c->opt.band_launch != 0; env.band_reverse = c->opt.band_reverse != 0;
	env.band_work_log2 = c->opt.band_work_log2; env.band_max_launches = c->opt.band_max_launches; env.band_list_entries = c->opt.band_list_entries;
	return TWK_HIP_OK;
}

}  // namespace

static int region_impl(twk_hip_ctx* c, int mode, const twk_hip_filters* f, uint32_t a0, uint32_t nA,
                       uint32_t b0, uint32_t nB, int32_t triangle, uint32_t part, uint32_t n_parts,
                       uint32_t tile_variants, int32_t window, uint32_t l_window, twk_hip_record_sink sink,
                       void* user, uint64_t* n_pairs, uint64_t* n_records) {
	const PlanGeom g{a0, nA, b0, nB, triangle, part, n_parts, tile_variants, window, l_window};
	PlanEnv env;
	int rc = plan_env_for(c, mode, f, env); if (rc) return rc;
	RegionPlan plan;
	plan_region(env, g, plan);
	RegionRun run(c, mode, f, g, env, plan, sink, user);
	run.mark("region: mode", (size_t)mode, nA);
	rc = run.prepare(); if (rc) return rc;
	rc = run.run_zone_passes(); if (rc) return rc;
	run.mark("tiles listed", plan.mine.size());
	rc = run.decide_three_by_samples(); if (rc) return rc;
	rc = run.run_pipeline(); if (rc) return rc;
	if (plan.windowed) run.tot_pairs = plan.pairs;      // screen: every pair of the band is decided; window: the pairs inside it, the ones the math evaluates
	if (n_pairs) *n_pairs = run.tot_pairs;
	if (n_records) *n_records = run.tot_recs;
	return TWK_HIP_OK;
}

static int region_dispatch(twk_hip_ctx* c, int mode, const twk_hip_filters* f, uint32_t a0, uint32_t nA,
                           uint32_t b0, uint32_t nB, int32_t triangle, uint32_t part, uint32_t n_parts,
                           uint32_t tile_variants, int32_t window, uint32_t l_window, twk_hip_record_sink sink,
                           void* user, uint64_t* n_pairs, uint64_t* n_records);

int twk_hip_ld_region(twk_hip_ctx* c, int mode, const twk_hip_filters* f, uint32_t a0, uint32_t nA,
                      uint32_t b0, uint32_t nB, int32_t triangle, uint32_t part, uint32_t n_parts,
                      uint32_t tile_variants, int32_t window, uint32_t l_window, twk_hip_record_sink sink,
                      void* user, uint64_t* n_pairs, uint64_t* n_records) {
	if (c) delivery_begin(c, sink);
	int rc = region_dispatch(c, mode, f, a0, nA, b0, nB, triangle, part, n_parts, tile_variants, window, l_window, sink, user, n_pairs, n_records);
	if (c) {                                // every record staged so far reaches the sink before the call returns, whatever the call's own result
		(void)hipSetDevice(c->device);
		const int drc = delivery_end(c);
		if (rc == TWK_HIP_OK && drc) rc = drc;            // (its text is in c->err: delivery_end)
	}
	if (c) flush_graveyard(c);               // buffers outgrown during the call: nothing is in flight any more
	return rc;
}

static int region_dispatch(twk_hip_ctx* c, int mode, const twk_hip_filters* f, uint32_t a0, uint32_t nA,
                           uint32_t b0, uint32_t nB, int32_t triangle, uint32_t part, uint32_t n_parts,
                           uint32_t tile_variants, int32_t window, uint32_t l_window, twk_hip_record_sink sink,
                           void* user, uint64_t* n_pairs, uint64_t* n_records) {
	if (!c || !f || !valid_mode(mode) || n_parts == 0 || part >= n_parts) return TWK_HIP_E_INVALID;
	if (!c->raw) return TWK_HIP_E_STATE;
	if (nA == 0 || nB == 0 || (uint64_t)a0 + nA > c->M || (uint64_t)b0 + nB > c->M) return TWK_HIP_E_INVALID;
	if (triangle && (a0 != b0 || nB < nA)) return TWK_HIP_E_INVALID;
	HIPCHK(c, hipSetDevice(c->device));
	c->fused_ok = true; c->three_ok = true;
	const bool whole = triangle && a0 == 0 && nA == c->M && nB == c->M;
	// TWK_HIP_OPT_R2_SCREEN: whole-triangle runs with an r2 cut-off worth the name, outside window mode (which
	// already prunes by position, in an order the allele-count sort would destroy)
	// ... or, at any cut-off above zero, rows long enough to keep carrier lists: the band is then (nearly) everything, but the
	// allele-count order still puts the rare variants in a zone whose pairs are list merges and probes instead of contractions
	const bool lists_pay = f->minR2 > 0 && c->opt.lists != 0 && (c->Wp / 128 >= 32 || c->opt.lists == 2);
	const bool screen = (window & TWK_HIP_OPT_R2_SCREEN) && whole && !(window & TWK_HIP_OPT_WINDOW) && (f->minR2 >= 1e-3 || lists_pay) && c->M >= 2;
	if (screen && !c->any_missing && (mode == TWK_HIP_MODE_PHASED || mode == TWK_HIP_MODE_AUTO || mode == TWK_HIP_MODE_UNPHASED)) {
		// Below the cut-off that makes a band worth having the sorted order is only taken for its carrier lists: when the set turns out to
		// keep none (too few rare variants), the sorted copy of the planes is dropped again and the run goes the file-order way (it used to
		// run the plain matrix path in sorted order with a band that covers everything: twice the plane memory for nothing).
		const int sset = mode == TWK_HIP_MODE_UNPHASED ? PS_SORTED_U : PS_SORTED_P;
		bool keep_sorted = true;
		bool& known_useless = c->sorted_keeps_no_lists[sset == PS_SORTED_U ? 1 : 0];      // (every region call of a multi-step run used to build and drop the set again)
		if (f->minR2 < 1e-3 && known_useless) keep_sorted = false;
		else if (f->minR2 < 1e-3) {
			const int rc = ensure_planes(c, sset); if (rc) return rc;
			if (c->planes[sset].n_list < 2) {
				keep_sorted = false; known_useless = true;
				HIPCHK(c, hipDeviceSynchronize());
				PlaneSet& ps = c->planes[sset];
				if (ps.owns_rows && ps.rows) (void)hipFree(ps.rows);
				if (ps.rowpop) (void)hipFree(ps.rowpop);
				if (ps.ids) (void)hipFree(ps.ids);
				if (ps.lists) (void)hipFree(ps.lists);
				if (ps.list_mac) (void)hipFree(ps.list_mac);
				if (ps.list_flip) (void)hipFree(ps.list_flip);
				if (ps.terms) (void)hipFree(ps.terms);
				ps = PlaneSet();
			}
		}
		if (keep_sorted)
			return region_impl(c, mode == TWK_HIP_MODE_UNPHASED ? MODE_INT_SORTED_U : MODE_INT_SORTED_P, f, 0, c->M, 0, c->M, 1, part, n_parts,
			                   tile_variants, window, l_window, sink, user, n_pairs, n_records);
	}
	if (!(mode == TWK_HIP_MODE_AUTO && c->any_missing && whole))
		return region_impl(c, mode, f, a0, nA, b0, nB, triangle, part, n_parts, tile_variants, window, l_window,
		                   sink, user, n_pairs, n_records);
	// Default mode over the whole triangle with missing genotypes somewhere: every pair goes through
	// the cheap one-plane phased products, and only the pairs that involve a variant with missing
	// data (the leading group G of the regrouped set) through the 3-plane unphased ones:
	// G x G (triangle) and G x rest (rectangle).  Each stage is sharded on its own.
	int rc = ensure_planes(c, PS_GROUPED); if (rc) return rc;
	const uint32_t nG = c->planes[PS_GROUPED].n_front;
	uint64_t pairs = 0, recs = 0, p2 = 0, r2 = 0;
	if (screen) {
		// the pairs without missing data: a screened triangle over the missing-free head of the sorted set;
		// the stages below then cover every pair that involves a variant with missing data
		rc = ensure_planes(c, PS_SORTED_P); if (rc) return rc;
		const uint32_t nC = c->planes[PS_SORTED_P].n_front;
		if (nC >= 2) rc = region_impl(c, MODE_INT_SORTED_P, f, 0, nC, 0, nC, 1, part, n_parts, tile_variants, window, l_window,
		                              sink, user, &pairs, &recs);
	} else
	rc = region_impl(c, MODE_INT_AUTO_CLEAN, f, 0, c->M, 0, c->M, 1, part, n_parts, tile_variants, window, l_window,
	                 sink, user, &pairs, &recs);
	struct Mute { twk_hip_ctx* c; ~Mute() { c->progress_muted = false; } } mute{c};
	c->progress_muted = true;
	if (rc == TWK_HIP_OK && nG >= 2) {
		rc = region_impl(c, MODE_INT_GROUPED, f, 0, nG, 0, nG, 1, part, n_parts, tile_variants, window, l_window,
		                 sink, user, &p2, &r2);
		recs += r2;
		if (screen) pairs += p2;               // the screened stage only counted the pairs without missing data
	}
	if (rc == TWK_HIP_OK && nG >= 1 && nG < c->M) {
		rc = region_impl(c, MODE_INT_GROUPED, f, 0, nG, nG, c->M - nG, 0, part, n_parts, tile_variants, window, l_window,
		                 sink, user, &p2, &r2);
		recs += r2;
		if (screen) pairs += p2;
	}
	if (n_pairs) *n_pairs = pairs;          // every pair of the shard is evaluated exactly once
	if (n_records) *n_records = recs;
	return rc;
}

int twk_hip_shard_rows(uint32_t n_rows, uint32_t n_cols, int32_t triangle, uint32_t part, uint32_t n_parts,
                       uint32_t* row_begin, uint32_t* row_end, uint64_t* n_pairs) {
	if (n_parts == 0 || part >= n_parts || n_rows == 0 || n_cols == 0 || (triangle && n_rows != n_cols)) return TWK_HIP_E_INVALID;
	const uint32_t r0 = band_boundary(part, n_parts, n_rows, n_cols, triangle != 0);
	const uint32_t r1 = band_boundary(part + 1, n_parts, n_rows, n_cols, triangle != 0);
	if (row_begin) *row_begin = r0;
	if (row_end) *row_end = r1;
	if (n_pairs) *n_pairs = band_pairs_before(r1, n_rows, n_cols, triangle != 0) - band_pairs_before(r0, n_rows, n_cols, triangle != 0);
	return TWK_HIP_OK;
}

int twk_hip_plan_region(const twk_hip_plan_env* pe, const twk_hip_variant_meta* meta, const uint32_t* popc, uint32_t n_variants,
                        uint32_t a0, uint32_t nA, uint32_t b0, uint32_t nB, int32_t triangle, uint32_t part, uint32_t n_parts,
                        uint32_t tile_variants, int32_t window, uint32_t l_window,
                        twk_hip_tile_desc* tiles, uint32_t capacity, uint32_t* n_tiles, uint32_t* n_band_launches,
                        uint32_t* row_begin, uint32_t* row_end, uint64_t* n_pairs, uint32_t* lo, uint32_t* hi) {
	if (!pe || !meta || !n_tiles || (capacity && !tiles) || n_parts == 0 || part >= n_parts) return TWK_HIP_E_INVALID;
	if (nA == 0 || nB == 0 || (uint64_t)a0 + nA > n_variants || (uint64_t)b0 + nB > n_variants) return TWK_HIP_E_INVALID;
	if (triangle && (a0 != b0 || nB < nA)) return TWK_HIP_E_INVALID;
	if (pe->screen && (!popc || !triangle)) return TWK_HIP_E_INVALID;
	if (pe->planes_per_variant < 1 || pe->planes_per_variant > 3 || pe->n_samples == 0) return TWK_HIP_E_INVALID;
	PlanEnv env;
	env.n_samples = pe->n_samples; env.Pmax = pe->planes_per_variant; env.nchunks = std::max<uint32_t>(pe->k_chunks, 1); env.resident_blocks = std::max<uint32_t>(pe->resident_blocks, 1);
	env.screen = pe->screen; env.minR2 = pe->minR2; env.fused = pe->fused != 0; env.phased_math = pe->phased_math != 0;
	env.meta = meta; env.ids = nullptr; env.popc = popc;
	env.band_launch = pe->band_launch != 0; env.band_reverse = pe->band_reverse != 0; env.band_work_log2 = pe->band_work_log2; env.band_max_launches = std::max<long long>(pe->band_max_launches, 1);
	const PlanGeom g{a0, nA, b0, nB, triangle, part, n_parts, tile_variants, window, l_window};
	RegionPlan plan;
	plan_region(env, g, plan);
	*n_tiles = (uint32_t)plan.mine.size();
	if (n_band_launches) *n_band_launches = (uint32_t)plan.bands.size();
	if (row_begin) *row_begin = plan.r0;
	if (row_end) *row_end = plan.r1;
	if (n_pairs) *n_pairs = plan.pairs;
	if (plan.windowed) {
		if (lo) std::copy(plan.lo.begin(), plan.lo.end(), lo);
		if (hi) std::copy(plan.hi.begin(), plan.hi.end(), hi);
	}
	if (plan.mine.size() > capacity) return TWK_HIP_E_OVERFLOW;
	std::copy(plan.mine.begin(), plan.mine.end(), tiles);
	return TWK_HIP_OK;
}

// Fisher's exact test on caller-supplied tables, through the same kernels the pair math uses.
__global__ void k_tables_to_records(const int32_t* __restrict__ t, unsigned long long n, twk_hip_record* __restrict__ r) {
	const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	twk_hip_record x{};
	x.idxA = 0; x.idxB = 1;
	x.cnt[0] = t[4 * i]; x.cnt[2] = t[4 * i + 1]; x.cnt[1] = t[4 * i + 2]; x.cnt[3] = t[4 * i + 3];      // n12 rides in the REFALT slot (ld_engine.cpp:1222-1226)
	r[i] = x;
}
__global__ void k_records_to_p(const twk_hip_record* __restrict__ r, unsigned long long n, double* __restrict__ p) {
	const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = r[i].P;
}

int twk_hip_fisher_exact(twk_hip_ctx* c, const int32_t* tables, uint64_t n, double* p_two_sided, int32_t in_given_order, float* kernel_ms) {
	if (!c || !tables || !p_two_sided || n == 0 || n > (1ull << 28)) return TWK_HIP_E_INVALID;
	if (!c->d_lfact) return TWK_HIP_E_STATE;
	HIPCHK(c, hipSetDevice(c->device));
	int32_t* d_t = nullptr; twk_hip_record* d_r = nullptr; double* d_p = nullptr; unsigned long long* d_n = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	auto cleanup = [&] {
		if (d_t) (void)hipFree(d_t);
		if (d_r) (void)hipFree(d_r);
		if (d_p) (void)hipFree(d_p);
		if (d_n) (void)hipFree(d_n);
		if (e0) (void)hipEventDestroy(e0);
		if (e1) (void)hipEventDestroy(e1);
	};
	hipError_t e = hipMalloc((void**)&d_t, (size_t)n * 16);
	if (e == hipSuccess) e = hipMalloc((void**)&d_r, (size_t)n * sizeof(twk_hip_record));
	if (e == hipSuccess) e = hipMalloc((void**)&d_p, (size_t)n * 8);
	if (e == hipSuccess) e = hipMalloc((void**)&d_n, 4 * sizeof(unsigned long long));
	if (e == hipSuccess) e = hipEventCreate(&e0);
	if (e == hipSuccess) e = hipEventCreate(&e1);
	const unsigned long long counters[4] = {n, 0, 0, 0};
	if (e == hipSuccess) e = hipMemcpyAsync(d_t, tables, (size_t)n * 16, hipMemcpyHostToDevice, c->s_compute);
	if (e == hipSuccess) e = hipMemcpyAsync(d_n, counters, sizeof(counters), hipMemcpyHostToDevice, c->s_compute);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_tables_to_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->s_compute, d_t, (unsigned long long)n, d_r);
		e = hipEventRecord(e0, c->s_compute);
	}
	if (e == hipSuccess) {
		// as the engine runs it (walk-length order; the table buffer is free: the tables are in the records by now), or in the order given
		const int rc = launch_fisher(c, d_r, d_n, (unsigned long long)n, 2.0, in_given_order ? nullptr : (uint32_t*)d_t, (size_t)n * 4);
		if (rc) { cleanup(); return rc; }
		e = hipEventRecord(e1, c->s_compute);
	}
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_records_to_p, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->s_compute, d_r, (unsigned long long)n, d_p);
		e = hipMemcpyAsync(p_two_sided, d_p, (size_t)n * 8, hipMemcpyDeviceToHost, c->s_compute);
	}
	if (e == hipSuccess) e = hipStreamSynchronize(c->s_compute);
	if (e == hipSuccess) e = hipGetLastError();
	if (e == hipSuccess && kernel_ms) e = hipEventElapsedTime(kernel_ms, e0, e1);
	cleanup();
	HIPCHK(c, e);
	return TWK_HIP_OK;
}

int twk_hip_set_device_sink(twk_hip_ctx* c, int on) {
	if (!c) return TWK_HIP_E_INVALID;
	c->device_sink = on != 0;
	c->d_keep_n = 0;
	return TWK_HIP_OK;
}

int twk_hip_device_records(twk_hip_ctx* c, const twk_hip_record** records, uint64_t* n) {
	if (!c || !records || !n) return TWK_HIP_E_INVALID;
	if (!c->device_sink) return TWK_HIP_E_STATE;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->s_copy));
	*records = c->d_keep_n ? c->d_keep : nullptr;
	*n = c->d_keep_n;
	return TWK_HIP_OK;
}

// ---- the gather of a one-process multi-GPU run: device sink -> device sink over RCCL ------------------------------------------------
// (north star: "a final RCCL gather of .two output blocks over xGMI"; the reference's counterpart is every slave's flush of its output
// block into the shared writer, lib/ld/ld_engine.cpp:1742-1802).  librccl is opened when the first gather asks for it - a process that
// never gathers, or a box without RCCL, loses nothing - and one communicator clique per set of devices is kept for the life of the process.
extern "C++" {
namespace {
struct Rccl {
	void* lib = nullptr; bool tried = false; char why[256] = {0}; int version = 0;
	ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GetVersion)(int*) = nullptr;
	const char* (*GetErrorString)(ncclResult_t) = nullptr;
	std::map<std::vector<int>, std::vector<ncclComm_t>> cliques;
	std::mutex mu;
	bool open() {
		if (tried) return lib != nullptr;
		tried = true;
		// RCCL 2.27 prints a banner (its version, HIP's, the host name, its own path) to STDOUT when the first communicator comes up, unless
		// NCCL_DEBUG says NONE: a `tomahawk calc` whose stdout is somebody's pipe must not grow five lines.  Left alone if the caller set it.
		(void)setenv("NCCL_DEBUG", "NONE", 0);
		for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
		if (!lib) { snprintf(why, sizeof(why), "librccl.so.1 cannot be opened: %s", dlerror()); return false; }
		auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) snprintf(why, sizeof(why), "librccl lacks %s", n); return p; };
		CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
		GroupStart = (decltype(GroupStart))sym("ncclGroupStart"); GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
		Send = (decltype(Send))sym("ncclSend"); Recv = (decltype(Recv))sym("ncclRecv");
		GetVersion = (decltype(GetVersion))sym("ncclGetVersion"); GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
		if (!(CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetVersion && GetErrorString)) { dlclose(lib); lib = nullptr; return false; }
		(void)GetVersion(&version);
		return true;
	}
};
Rccl& rccl() { static Rccl r; return r; }
}  // namespace
}  // extern "C++"

const char* twk_hip_gather_backend(void) {
	Rccl& r = rccl();
	std::lock_guard<std::mutex> lk(r.mu);
	static char text[320];
	if (r.open()) snprintf(text, sizeof(text), "rccl %d", r.version);
	else snprintf(text, sizeof(text), "unavailable (%s)", r.why);
	return text;
}

int twk_hip_gather_records(twk_hip_ctx* const* ctxs, uint32_t n, uint32_t dst, int32_t flags, uint64_t* n_records, double* transfer_ms) {
	if (!ctxs || n == 0 || dst >= n) return TWK_HIP_E_INVALID;
	for (uint32_t r = 0; r < n; ++r) {
		if (!ctxs[r]) return TWK_HIP_E_INVALID;
		if (!ctxs[r]->device_sink) return TWK_HIP_E_STATE;
		for (uint32_t q = 0; q < r; ++q) if (ctxs[q]->device == ctxs[r]->device) return TWK_HIP_E_INVALID;      // one context per GPU: RCCL refuses a device twice
	}
	twk_hip_ctx* d = ctxs[dst];
	const bool loop = n == 1 && (flags & TWK_HIP_GATHER_SELF_LOOP);
	if (n_records) *n_records = d->d_keep_n;
	if (transfer_ms) *transfer_ms = 0.0;
	if (n == 1 && !loop) return TWK_HIP_OK;
	Rccl& rc = rccl();
	std::lock_guard<std::mutex> lk(rc.mu);
	if (!rc.open()) { snprintf(d->err, sizeof(d->err), "RCCL gather: %s", rc.why); return TWK_HIP_E_DEVICE; }
	auto NCHK = [&](ncclResult_t e, const char* what) -> int {
		if (e == ncclSuccess) return TWK_HIP_OK;
		snprintf(d->err, sizeof(d->err), "RCCL gather: %s failed: %s", what, rc.GetErrorString(e));
		return TWK_HIP_E_DEVICE;
	};
	std::vector<int> devs(n);
	for (uint32_t r = 0; r < n; ++r) devs[r] = ctxs[r]->device;
	auto it = rc.cliques.find(devs);
	if (it == rc.cliques.end()) {
		std::vector<ncclComm_t> comms(n);
		if (const int e = NCHK(rc.CommInitAll(comms.data(), (int)n, devs.data()), "ncclCommInitAll")) return e;
		it = rc.cliques.emplace(devs, std::move(comms)).first;
	}
	const std::vector<ncclComm_t>& comm = it->second;
	// every context's survivors are final (their copies ran on the contexts' copy streams): counts on the host, room on the destination
	unsigned long long total = 0;
	std::vector<unsigned long long> cnt(n), off(n);
	for (uint32_t r = 0; r < n; ++r) { HIPCHK(d, hipSetDevice(ctxs[r]->device)); HIPCHK(d, hipStreamSynchronize(ctxs[r]->s_copy)); cnt[r] = ctxs[r]->d_keep_n; }
	off[dst] = 0; total = cnt[dst];
	for (uint32_t r = 0; r < n; ++r) if (r != dst) { off[r] = total; total += cnt[r]; }
	HIPCHK(d, hipSetDevice(d->device));
	twk_hip_record* loop_buf = nullptr;
	if (loop) {          // one GPU: the same group of ncclSend / ncclRecv, from the sink to itself through a second buffer (what a one-GPU box can show of the path)
		if (cnt[0]) HIPCHK(d, hipMalloc((void**)&loop_buf, (size_t)cnt[0] * sizeof(twk_hip_record)));
	} else {
		const unsigned long long own = d->d_keep_n;
		d->d_keep_n = own;
		const int e = ensure_device_keep(d, total - own); if (e) return e;         // (keeps the destination's own records, at the front)
	}
	struct Events {          // (destroyed on every way out)
		hipEvent_t e0 = nullptr, e1 = nullptr;
		~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
	} ev;
	hipEvent_t& e0 = ev.e0; hipEvent_t& e1 = ev.e1;
	struct LoopBuf { twk_hip_record** p; ~LoopBuf() { if (*p) (void)hipFree(*p); } } loop_guard{&loop_buf};      // (freed unless it became the sink)
	HIPCHK(d, hipEventCreate(&e0)); HIPCHK(d, hipEventCreate(&e1));
	HIPCHK(d, hipEventRecord(e0, d->s_copy));
	int rcode = NCHK(rc.GroupStart(), "ncclGroupStart");
	for (uint32_t r = 0; r < n && !rcode; ++r) {
		if (!loop && r == dst) continue;
		const size_t bytes = (size_t)cnt[r] * sizeof(twk_hip_record);
		if (!bytes) continue;
		if (hipSetDevice(ctxs[r]->device) != hipSuccess) { rcode = TWK_HIP_E_DEVICE; break; }
		rcode = NCHK(rc.Send(ctxs[r]->d_keep, bytes, ncclUint8, (int)dst, comm[r], ctxs[r]->s_copy), "ncclSend");
		if (rcode) break;
		if (hipSetDevice(d->device) != hipSuccess) { rcode = TWK_HIP_E_DEVICE; break; }
		rcode = NCHK(rc.Recv(loop ? loop_buf : d->d_keep + off[r], bytes, ncclUint8, (int)r, comm[dst], d->s_copy), "ncclRecv");
	}
	{ const int e = NCHK(rc.GroupEnd(), "ncclGroupEnd"); if (!rcode) rcode = e; }
	(void)hipSetDevice(d->device);
	if (!rcode && hipEventRecord(e1, d->s_copy) != hipSuccess) rcode = TWK_HIP_E_DEVICE;
	for (uint32_t r = 0; r < n && !rcode; ++r) {
		if (hipSetDevice(ctxs[r]->device) != hipSuccess || hipStreamSynchronize(ctxs[r]->s_copy) != hipSuccess) rcode = TWK_HIP_E_DEVICE;
	}
	(void)hipSetDevice(d->device);
	float ms = 0;
	if (!rcode && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && transfer_ms) *transfer_ms = ms;
	if (rcode) { if (!d->err[0]) snprintf(d->err, sizeof(d->err), "RCCL gather: a HIP call failed"); return rcode; }
	if (loop) {
		if (loop_buf) {        // the records that went round are the sink's content from here on
			if (d->d_keep) (void)hipFree(d->d_keep);
			d->d_keep = loop_buf; d->d_keep_cap = cnt[0];
			loop_buf = nullptr;
		}
	} else {
		d->d_keep_n = total;
		for (uint32_t r = 0; r < n; ++r) if (r != dst) ctxs[r]->d_keep_n = 0;
	}
	if (n_records) *n_records = d->d_keep_n;
	return TWK_HIP_OK;
}

int twk_hip_drain_device_sink(twk_hip_ctx* c, twk_hip_record_sink sink, void* user, uint64_t* n_records) {
	if (!c || !sink) return TWK_HIP_E_INVALID;
	if (!c->device_sink) return TWK_HIP_E_STATE;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->s_copy));
	const unsigned long long n = c->d_keep_n;
	if (n_records) *n_records = n;
	const int rc = n ? deliver_records(c, c->d_keep, n, sink, user, c->s_copy, 0.0) : TWK_HIP_OK;
	c->d_keep_n = 0;
	return rc;
}

int twk_hip_set_option(twk_hip_ctx* c, const char* key, int64_t value) {
	if (!c || !key) return TWK_HIP_E_INVALID;
	for (const OptionKey& k : OPTION_KEYS) {
		if (std::strcmp(k.name, key) != 0) continue;
		if (value < k.lo || value > k.hi) { snprintf(c->err, sizeof(c->err), "option %s: %lld is outside [%lld, %lld]", key, (long long)value, k.lo, k.hi); return TWK_HIP_E_INVALID; }
		if (c->opt.*(k.field) == value) return TWK_HIP_OK;
		c->opt.*(k.field) = value;
		if (k.rebuilds_planes && c->raw) {       // the carrier lists belong to the allele-count-sorted sets: rebuilt on next use
			HIPCHK(c, hipSetDevice(c->device));
			HIPCHK(c, hipDeviceSynchronize());
			free_planes(c);
		}
		return TWK_HIP_OK;
	}
	snprintf(c->err, sizeof(c->err), "unknown option %s", key);
	return TWK_HIP_E_INVALID;
}

int twk_hip_option_describe(uint32_t index, const char** key, int64_t* dflt, int64_t* lo, int64_t* hi, const char** meaning) {
	if (index >= sizeof(OPTION_KEYS) / sizeof(OPTION_KEYS[0])) return TWK_HIP_E_INVALID;
	const OptionKey& k = OPTION_KEYS[index];
	if (key) *key = k.name;
	if (dflt) *dflt = k.dflt;
	if (lo) *lo = k.lo;
	if (hi) *hi = k.hi;
	if (meaning) *meaning = k.doc;
	return TWK_HIP_OK;
}

int twk_hip_get_option(const twk_hip_ctx* c, const char* key, int64_t* value) {
	if (!c || !key || !value) return TWK_HIP_E_INVALID;
	for (const OptionKey& k : OPTION_KEYS)
		if (std::strcmp(k.name, key) == 0) { *value = c->opt.*(k.field); return TWK_HIP_OK; }
	return TWK_HIP_E_INVALID;
}

int twk_hip_set_progress(twk_hip_ctx* c, twk_hip_progress_cb cb, void* user) {
	if (!c) return TWK_HIP_E_INVALID;
	c->progress_cb = cb; c->progress_user = user;
	return TWK_HIP_OK;
}

int twk_hip_launch_log(twk_hip_ctx* c, twk_hip_launch_stat* out, uint32_t capacity, uint32_t* n_copied, uint64_t* n_total) {
	if (!c || (capacity && !out)) return TWK_HIP_E_INVALID;
	const size_t have = c->launch_ring.size(), n = std::min<size_t>(have, capacity);
	// oldest first: the ring's write position is launches_seen % LAUNCH_RING once it has wrapped
	const size_t start = have < LAUNCH_RING ? 0 : (size_t)(c->launches_seen % LAUNCH_RING);
	for (size_t k = 0; k < n; ++k) out[k] = c->launch_ring[(start + (have - n) + k) % have];
	if (n_copied) *n_copied = (uint32_t)n;
	if (n_total) *n_total = c->launches_seen;
	return TWK_HIP_OK;
}

int twk_hip_timing_reset(twk_hip_ctx* c) {
	if (!c) return TWK_HIP_E_INVALID;
	c->launch_ring.clear(); c->launches_seen = 0;
	const uint64_t w = c->timing.words_per_row;
	c->timing = twk_hip_timing{};
	c->timing.words_per_row = w;
	return TWK_HIP_OK;
}

int twk_hip_timing_get(twk_hip_ctx* c, twk_hip_timing* out) {
	if (!c || !out) return TWK_HIP_E_INVALID;
	*out = c->timing;
	// words contracted per row pair: live words of whichever plane set was used last
	for (const auto& p : c->planes) if (p.built) out->words_per_row = p.W_live;
	return TWK_HIP_OK;
}

}  // extern "C"
