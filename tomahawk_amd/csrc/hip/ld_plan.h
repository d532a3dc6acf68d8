// The plan of one region call: which rows this shard owns, which columns every row reaches, the launches that cover them.
// Host-only arithmetic (no HIP): twk_hip.hip's region_impl builds a PlanEnv from the context, calls plan_region() and executes
// the result; twk_hip_plan_region() (include/twk_hip.h) runs the same planner on caller-supplied arrays so that it can be
// tested without a GPU (tests/test_plan.py).  Replaces the reference's chunk partition and block-pair ticker
// (lib/ld/ld_balancing.h:23-80, 176-233) for one GPU's share of the pair space.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <vector>
#include "../../../include/twk_hip.h"

namespace twk {

constexpr uint32_t PLAN_TILE = 128;                 // plane rows per block tile edge (ld_count.hip.h TILE)

struct PlanGeom {                                   // the caller's region and shard
	uint32_t a0, nA, b0, nB; int32_t triangle;
	uint32_t part, n_parts, tile_variants;
	int32_t window; uint32_t l_window;
};
struct PlanEnv {                                    // what the planner needs to know of the problem and the engine
	uint32_t n_samples = 0;
	int Pmax = 1;                                   // plane rows per variant of the widest plane set of the mode
	uint32_t nchunks = 1;                           // K chunks of a row of the first plane set
	uint32_t resident_blocks = 512;
	int screen = 0;                                 // 0: none; 1 / 2: r2 band over an allele-count-sorted set (PhasedMath / UnphasedMath)
	double minR2 = 0;
	bool fused = false;                             // launches of this mode run the fused count -> screen form (band launches possible)
	bool phased_math = true;                        // (candidate entries: 3 words, else 6)
	// per position of the region's index space (file order, or a regrouped / sorted set through `ids`)
	const twk_hip_variant_meta* meta = nullptr; const uint32_t* ids = nullptr; const uint32_t* popc = nullptr;
	// options
	bool band_launch = true, band_reverse = true; long long band_work_log2 = 19, band_max_launches = 8, band_list_entries = 0;
	const twk_hip_variant_meta& at(uint32_t i) const { return meta[ids ? ids[i] : i]; }
};
struct BandLaunch { uint32_t xa, xb; size_t list_words; size_t tile_index; };
struct RegionPlan {
	bool windowed = false;
	std::vector<uint32_t> lo, hi;                   // windowed: row a0 + r reaches the columns [b0 + lo[r], b0 + hi[r])
	std::vector<uint64_t> cum;                      // cum[r] = pairs in reach of rows [0, r)
	uint32_t r0 = 0, r1 = 0;                        // the shard's rows
	uint32_t S = 0;                                 // super-tile edge in variants
	std::vector<twk_hip_tile_desc> mine;            // the launches, in order; the first bands.size() of them are band launches
	std::vector<BandLaunch> bands;
	uint64_t pairs = 0;                             // pairs the shard decides
};

inline uint32_t plan_round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }

// Rows [0, r) of a triangle (or trapezoid: nB >= nA columns, col > row) hold r*nB - r(r+1)/2 pairs; of an nA x nB rectangle r*nB.
inline uint64_t band_pairs_before(uint64_t r, uint64_t nA, uint64_t nB, bool triangle) {
	(void)nA;
	return triangle ? r * nB - r * (r + 1) / 2 : r * nB;      // triangle: row i pairs with cols (i, nB)
}
// First row of shard k: equal-area bands, boundaries on multiples of 64 variants.
inline uint32_t band_boundary(uint32_t k, uint32_t n_parts, uint32_t nA, uint32_t nB, bool triangle) {
	if (k == 0) return 0;
	if (k >= n_parts) return nA;
	const long double target = (long double)band_pairs_before(nA, nA, nB, triangle) * k / n_parts;
	uint32_t lo = 0, hi = nA;
	while (lo < hi) {
		const uint32_t mid = lo + (hi - lo) / 2;
		if ((long double)band_pairs_before(mid, nA, nB, triangle) < target) lo = mid + 1; else hi = mid;
	}
	return std::min(nA, (lo + 32) / 64 * 64);
}

// ---- the columns every row reaches ---------------------------------------------------------------------------------------
// r2 band (screen): with a and b the minor allele frequencies of two variants, a <= b, no 2x2 table with those margins has r2 above
// a(1-b) / ((1-a)b): |D| <= a(1-b), r2 = D^2 / (a(1-a)b(1-b)).  In order of minor allele count the pairs that can reach the cut-off
// are therefore a band above the diagonal: row r needs the columns (r, hi[r]) only, hi non-decreasing.  UnphasedMath estimates the
// haplotype frequency from genotypes and admits roots up to 1e-5 outside [minhap, maxhap] (ld_engine.h:37, ld_engine.cpp:1429-1558),
// so its |D| is bounded by a(1-b) + 1e-5.  The cut-off is lowered by a part in 1e6 against rounding in the reference's formula;
// pairs inside the band still go through that formula, so the survivors are the same.
inline void plan_reach_screen(const PlanEnv& e, const PlanGeom& g, RegionPlan& p) {
	const uint32_t nA = g.nA, nB = g.nB;
	p.lo.resize(nA); p.hi.resize(nA); p.cum.assign((size_t)nA + 1, 0);
	const long double T2 = 2.0L * e.n_samples, cut = (long double)e.minR2 * (1.0L - 1e-6L);
	auto mac = [&](uint32_t i) -> long double {      // minor allele count as the device counts it (twk_hip.hip ensure_popcounts)
		const long double ac = std::min<long double>(e.popc[e.ids ? e.ids[i] : i], T2);
		return std::min(ac, T2 - ac);
	};
	auto reach = [&](long double ma, long double mb) -> bool {        // can a pair with these minor counts (ma <= mb) pass?
		if (ma <= 0 || mb <= 0) return e.screen == 2;                   // a monomorphic site: PhasedMath drops it (D == 0)
		if (e.screen == 1) return ma * (T2 - mb) >= cut * (T2 - ma) * mb;
		const long double a = ma / T2, b = mb / T2, d = a * (1 - b) + 1e-5L;
		return d * d >= cut * a * (1 - a) * b * (1 - b);
	};
	uint32_t h = 0;
	for (uint32_t r = 0; r < nA; ++r) {
		const long double mr = mac(g.a0 + r);
		if (h < r + 1) h = std::min(r + 1, nB);
		while (h < nB && reach(mr, mac(g.b0 + h))) ++h;
		p.lo[r] = std::min(r + 1, nB);
		p.hi[r] = std::max(h, p.lo[r]);
		p.cum[r + 1] = p.cum[r] + (p.hi[r] - p.lo[r]);
	}
}
// Window mode: same contig, |dpos| <= l_window; variants are sorted by (rid, pos) like every .twk, so both ends only move forward.
// The ranges drive the shard boundaries (equal in-window pairs), the tile edge and the column range of every row block; the exact
// test itself stays in the math kernel.
inline void plan_reach_window(const PlanEnv& e, const PlanGeom& g, RegionPlan& p) {
	const uint32_t nA = g.nA, nB = g.nB;
	p.lo.resize(nA); p.hi.resize(nA); p.cum.assign((size_t)nA + 1, 0);
	uint32_t l = 0, h = 0;
	for (uint32_t r = 0; r < nA; ++r) {
		const twk_hip_variant_meta& R = e.at(g.a0 + r);
		auto before = [&](uint32_t j) { const twk_hip_variant_meta& B = e.at(g.b0 + j);
			return B.rid < R.rid || (B.rid == R.rid && (uint64_t)B.pos + g.l_window < R.pos); };
		auto within = [&](uint32_t j) { const twk_hip_variant_meta& B = e.at(g.b0 + j);
			return B.rid < R.rid || (B.rid == R.rid && B.pos <= (uint64_t)R.pos + g.l_window); };
		while (l < nB && before(l)) ++l;
		if (h < l) h = l;
		while (h < nB && within(h)) ++h;
		p.lo[r] = g.triangle ? std::min(std::max(l, r + 1), nB) : l;
		p.hi[r] = std::max(h, p.lo[r]);
		p.cum[r + 1] = p.cum[r] + (p.hi[r] - p.lo[r]);
	}
}

// ---- shard: a contiguous band of rows holding 1/n_parts of the region's pairs --------------------------------------------
// Row i of a triangle has nA-1-i pairs, of a rectangle nB; in window / band mode what the row reaches.  Equal-area bands,
// boundaries on multiples of 64 variants, derived identically (and without communication) by every rank.
inline void plan_shard(const PlanGeom& g, RegionPlan& p) {
	auto window_boundary = [&](uint32_t k) -> uint32_t {
		if (k == 0) return 0;
		if (k >= g.n_parts) return g.nA;
		const long double target = (long double)p.cum[g.nA] * k / g.n_parts;
		const uint32_t r = (uint32_t)(std::lower_bound(p.cum.begin(), p.cum.end(), (uint64_t)target) - p.cum.begin());
		return std::min(g.nA, (r + 32) / 64 * 64);
	};
	p.r0 = p.windowed ? window_boundary(g.part) : band_boundary(g.part, g.n_parts, g.nA, g.nB, g.triangle != 0);
	p.r1 = p.windowed ? window_boundary(g.part + 1) : band_boundary(g.part + 1, g.n_parts, g.nA, g.nB, g.triangle != 0);
}

// The block tiles build_tile_list (twk_hip.hip) will list for rows [x, x + h) over all the columns they reach.
inline uint64_t plan_tiles_of_rows(const PlanEnv& e, const PlanGeom& g, const RegionPlan& p, uint32_t x, uint32_t h) {
	const uint64_t P = (uint64_t)e.Pmax;
	const uint32_t col0 = g.triangle ? x : (p.windowed ? p.lo[x] : 0);
	uint64_t tiles = 0;
	for (uint64_t by = 0, gy = (h * P + PLAN_TILE - 1) / PLAN_TILE; by < gy; ++by) {
		const uint32_t v0 = x + (uint32_t)((by * PLAN_TILE) / P);
		const uint32_t v1 = (uint32_t)std::min<uint64_t>((uint64_t)x + h, (uint64_t)x + ((by + 1) * PLAN_TILE + P - 1) / P);
		if (v0 >= v1) continue;
		const uint32_t reach = p.windowed ? p.hi[v1 - 1] : g.nB;
		if (reach <= col0) continue;
		uint64_t c_lo = (p.windowed && p.lo[v0] > col0) ? ((uint64_t)(p.lo[v0] - col0) * P) / PLAN_TILE : 0;
		if (g.triangle) c_lo = std::max<uint64_t>(c_lo, by);
		const uint64_t c_hi = ((uint64_t)(reach - col0) * P + PLAN_TILE - 1) / PLAN_TILE;
		if (c_hi > c_lo) tiles += c_hi - c_lo;
	}
	return tiles;
}

// ---- super-tile edge -------------------------------------------------------------------------------------------------------
// Edge S in variants (multiple of 128).  Default: ~16384 plane rows per tile edge so that a launch holds >= 16 rounds of resident
// blocks and the partial last round costs < 3 %.  Window mode: a search over the row-block height.  A launch over rows [x, x + h)
// holds, per row of tiles, the tiles from the diagonal (or the first column its rows reach) to the last column they reach and costs
// ceil(tiles / resident blocks) rounds plus about half a round of launch, ramp-up and tail; the sum over the band's row blocks is
// minimised.  The plain path's math kernel visits every pair of the launch's rectangle, so there the block height stays near the
// window width; the fused path (count -> screen in the same kernel) visits the listed tiles only.
inline uint32_t plan_tile_edge(const PlanEnv& e, const PlanGeom& g, const RegionPlan& p) {
	uint32_t S = g.tile_variants ? g.tile_variants : (16384u / (uint32_t)e.Pmax);
	if (p.windowed && !g.tile_variants && p.r1 > p.r0) {
		const uint64_t wv = std::max<uint64_t>(1, (p.cum[p.r1] - p.cum[p.r0]) / (p.r1 - p.r0));   // mean partners per row
		const uint32_t s_hi = e.fused ? S : std::min<uint32_t>(S, std::max<uint32_t>(512u, plan_round_up((uint32_t)std::min<uint64_t>(wv, 1u << 20), 64)));
		const uint32_t s_lo = std::max<uint32_t>(128u, std::min<uint32_t>(s_hi, plan_round_up((uint32_t)std::min<uint64_t>(wv / 8, 1u << 20), 64)));
		const uint64_t R = e.resident_blocks;
		uint32_t best = s_lo; uint64_t best_cost = ~0ull;
		for (uint32_t cand = s_lo; cand <= s_hi; cand += 64) {
			uint64_t cost = 0;                    // in half rounds
			for (uint32_t x = p.r0; x < p.r1; x += cand) {
				const uint64_t tiles = plan_tiles_of_rows(e, g, p, x, std::min(cand, p.r1 - x));
				if (tiles) cost += 2 * ((tiles + R - 1) / R) + 1;
			}
			if (cost < best_cost || (cost == best_cost && cand > best)) { best_cost = cost; best = cand; }
		}
		S = best;
	}
	S = std::max<uint32_t>(PLAN_TILE, std::min<uint32_t>(S / PLAN_TILE * PLAN_TILE, 32768u));
	return std::min(S, plan_round_up(std::max(g.nA, g.nB), PLAN_TILE));
}

// One launch (tile descriptor) of the region, unless window mode proves that none of its pairs is wanted.
inline void plan_push_tile(const PlanEnv& e, const PlanGeom& g, std::vector<twk_hip_tile_desc>& out, uint32_t ra, uint32_t na, uint32_t cb, uint32_t nb_, int diag) {
	twk_hip_tile_desc t{};
	t.rowA0 = g.a0 + ra; t.nA = na; t.rowB0 = g.b0 + cb; t.nB = nb_; t.diag = diag; t.window = g.window; t.l_window = g.l_window;
	if ((g.window & TWK_HIP_OPT_WINDOW) && !diag) {
		// Each axis of a tile is sorted by (rid, pos) (file order, or one group of the regrouped set), but the two axes are in no
		// particular order relative to each other (regrouped rectangle: the rows are the variants with missing data, the columns the
		// rest).  A tile can only be skipped when both of its axes lie on one contig each and either the contigs differ or the
		// position intervals are more than the window apart, in whichever direction (the reference's ticker skips the rest of a row
		// on the same grounds, ld_balancing.h:191).
		const twk_hip_variant_meta& firstA = e.at(t.rowA0);
		const twk_hip_variant_meta& lastA  = e.at(t.rowA0 + t.nA - 1);
		const twk_hip_variant_meta& firstB = e.at(t.rowB0);
		const twk_hip_variant_meta& lastB  = e.at(t.rowB0 + t.nB - 1);
		if (firstA.rid == lastA.rid && firstB.rid == lastB.rid) {
			if (firstA.rid != firstB.rid) return;
			if ((uint64_t)firstB.pos > (uint64_t)lastA.pos + g.l_window) return;      // columns wholly after the rows' reach
			if ((uint64_t)firstA.pos > (uint64_t)lastB.pos + g.l_window) return;      // columns wholly before it
		}
	}
	out.push_back(t);
}

// Super-tiles sized by their count matrix, for the rows [xa, xb) of the band (appended to `out`).  Column step per row block: a
// launch of B blocks takes ceil(B / resident) rounds of (equal length) blocks, so the partial last round is pure loss; with the
// default tiling the column step is chosen, per row block, to minimise the total number of rounds (ties: fewer launches) - it matters
// for the thin bands of a multi-GPU shard.  C stays <= 2 GiB per tile.
inline void plan_matrix_tiles(const PlanEnv& e, const PlanGeom& g, const RegionPlan& p, uint32_t xa, uint32_t xb, std::vector<twk_hip_tile_desc>& out) {
	const uint32_t S = p.S, nB = g.nB;
	const uint64_t P = (uint64_t)e.Pmax;
	auto rows_of = [&](uint32_t nv) -> uint64_t { return ((uint64_t)nv * P + PLAN_TILE - 1) / PLAN_TILE; };
	auto blocks_of = [&](uint32_t h, uint32_t w, bool diag) -> uint64_t {      // a diagonal tile only runs the blocks on and above its diagonal
		const uint64_t ra = rows_of(h), rb = rows_of(w);
		return diag ? ra * (ra + 1) / 2 + ra * (rb - ra) : ra * rb;
	};
	auto choose_col_step = [&](uint32_t h, uint32_t col0, bool first_is_diag) -> uint32_t {
		if (g.tile_variants || col0 >= nB) return std::max(S, h);
		const uint64_t R = e.resident_blocks, ra = rows_of(h);
		const uint64_t max_rows_b = std::min<uint64_t>(((2ull << 30) / 4) / (ra * PLAN_TILE), 32768ull * P / PLAN_TILE);   // blocks
		// window mode: the column range of a row block is already cut to what it can reach: one launch (or as few as the 2 GiB bound on C allows)
		if (p.windowed) return std::max<uint32_t>(h, (uint32_t)std::min<uint64_t>(32768ull, max_rows_b * PLAN_TILE / P / 64 * 64));
		uint32_t best = std::max(S, h); uint64_t best_cost = ~0ull;
		for (uint32_t sc = plan_round_up(h, 64); sc <= 32768; sc += 64) {
			if (rows_of(sc) > max_rows_b) break;
			if (sc * 4 < S) continue;                               // keep launches reasonably large
			uint64_t cost = 0; bool diag = first_is_diag;
			for (uint32_t col = col0; col < nB; col += sc, diag = false)
				cost += (blocks_of(h, std::min(sc, nB - col), diag) + R - 1) / R;
			if (cost < best_cost || (cost == best_cost && sc > best)) { best_cost = cost; best = sc; }
		}
		return best;
	};
	for (uint32_t x = xa; x < xb; x += S) {
		const uint32_t h = std::min(S, xb - x);
		// triangle: the first tile of the row block starts on the diagonal (rows [x,x+h) x cols [x,x+w), w >= h, only col > row) and
		// continues into the rectangle to its right in the same launch
		uint32_t col = g.triangle ? x : 0, col_end = nB;
		if (p.windowed) {       // only the columns some row of the block can reach
			if (!g.triangle) col = p.lo[x];
			col_end = p.hi[x + h - 1];
			if (col_end <= col) continue;
		}
		const uint32_t sc = choose_col_step(h, col, g.triangle != 0);
		bool diag = g.triangle != 0;
		for (; col < col_end; col += sc, diag = false) {
			uint32_t w = std::min(sc, col_end - col);
			if (diag && w < h) w = std::min(h, nB - col);            // the diagonal tile must span its own rows
			plan_push_tile(e, g, out, x, h, col, w, diag ? 1 : 0);
			if (diag && w > sc) col += w - sc;
		}
	}
}

// Band launches.  A fused launch keeps no count matrix - what it leaves behind is the list of its candidates - so nothing ties its
// extent to the 2 GiB a matrix may take: it is sized by its *work*.  The rows of the band are cut into launches of at least
// ~2^band_work_log2 tile-chunks (19: about 5 ms of contraction), at most band_max_launches per region, every one over all the columns
// its rows reach: one ramp-up and one tail per launch instead of per row block of 16,384 plane rows, and a handful of sorts, copies and
// hand-overs per region on the host instead of dozens.  More than one launch when there is work for it: the host's writer gets its first
// records while the device still counts.  A launch whose candidates or survivors outgrow their buffers is redone as matrix-sized tiles.
// Fills p.mine / p.bands; leaves both empty when the region does not qualify.
inline void plan_band_launches(const PlanEnv& e, const PlanGeom& g, RegionPlan& p) {
	const uint32_t step = PLAN_TILE;                                  // rows are cut on multiples of 128 variants
	const uint64_t P = (uint64_t)e.Pmax;
	auto rb = [&](uint64_t nv) -> uint64_t { return (nv * P + PLAN_TILE - 1) / PLAN_TILE; };
	std::vector<uint64_t> cum_tiles(1, 0);
	for (uint32_t x = p.r0; x < p.r1; x += step) cum_tiles.push_back(cum_tiles.back() + plan_tiles_of_rows(e, g, p, x, std::min(step, p.r1 - x)));
	const uint64_t total = cum_tiles.back();
	const uint64_t n_launch = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)e.band_max_launches, total * e.nchunks >> e.band_work_log2));
	const uint64_t pairs_per_tile = (uint64_t)(PLAN_TILE / P) * (PLAN_TILE / P);
	const unsigned words_per_entry = e.phased_math ? 3 : 6;
	size_t k0 = 0;
	for (uint64_t l = 0; l < n_launch && k0 + 1 < cum_tiles.size(); ++l) {
		size_t k1 = cum_tiles.size() - 1;
		if (l + 1 < n_launch) {
			const uint64_t target = total * (l + 1) / n_launch;
			k1 = (size_t)(std::lower_bound(cum_tiles.begin() + k0 + 1, cum_tiles.end(), target) - cum_tiles.begin());
			k1 = std::min(k1, cum_tiles.size() - 1);
		}
		if (k1 <= k0) continue;
		const uint32_t xa = p.r0 + (uint32_t)k0 * step, xb = (uint32_t)std::min<uint64_t>(p.r1, (uint64_t)p.r0 + (uint64_t)k1 * step);
		const uint64_t tiles = cum_tiles[k1] - cum_tiles[k0];
		k0 = k1;
		if (!tiles) continue;
		const uint32_t col0 = g.triangle ? xa : (p.windowed ? p.lo[xa] : 0), col_end = p.windowed ? p.hi[xb - 1] : g.nB;
		if (col_end <= col0) continue;
		uint32_t w = col_end - col0;
		if (g.triangle && w < xb - xa) w = std::min(xb - xa, g.nB - col0);
		if (rb(xb - xa) > 0xFFFFu || rb(w) > 0xFFFFu) { p.bands.clear(); p.mine.clear(); return; }      // beyond a tile list's 16-bit coordinates: matrix tiles
		// candidate slots: 1/32 of the launch's pairs (a survivor-rich window run has 2 % candidates), 4 M at least, 256 M at most; the
		// survivor buffer is sized once the candidates are counted (enqueue_band_math)
		uint64_t entries = std::min<uint64_t>(std::max<uint64_t>(tiles * pairs_per_tile / 32, 1ull << 22), 1ull << 28);
		entries = std::min<uint64_t>(entries, std::max<uint64_t>(tiles * pairs_per_tile / 3, 1024));      // (never more than a matrix tile would get)
		if (e.band_list_entries) entries = (uint64_t)e.band_list_entries;      // (test / measurement: exactly this many)
		const BandLaunch b{xa, xb, (size_t)entries * words_per_entry, p.mine.size()};
		const size_t before = p.mine.size();
		plan_push_tile(e, g, p.mine, xa, xb - xa, col0, w, g.triangle ? 1 : 0);
		if (p.mine.size() > before) p.bands.push_back(b);
	}
	if (p.bands.empty()) { p.mine.clear(); return; }
	if (e.screen && e.band_reverse) {
		// Allele-count order: the survivors of a run concentrate in the last bands (common variants).  Last band first, so that the host
		// compresses those while the device counts the poor ones, instead of after it has finished (profiles/r04_band_timeline.txt).
		std::reverse(p.bands.begin(), p.bands.end());
		std::reverse(p.mine.begin(), p.mine.begin() + (ptrdiff_t)p.bands.size());
		for (size_t i = 0; i < p.bands.size(); ++i) p.bands[i].tile_index = i;
	}
}

// The whole plan of a region call.
inline void plan_region(const PlanEnv& e, const PlanGeom& g, RegionPlan& p) {
	p = RegionPlan();
	p.windowed = (g.window & TWK_HIP_OPT_WINDOW) != 0 || e.screen != 0;      // rows reach a column range only
	if (e.screen) plan_reach_screen(e, g, p);
	else if (p.windowed) plan_reach_window(e, g, p);
	plan_shard(g, p);
	p.S = plan_tile_edge(e, g, p);
	if (!g.tile_variants && e.band_launch && p.r1 > p.r0 && e.fused) plan_band_launches(e, g, p);
	if (p.bands.empty()) plan_matrix_tiles(e, g, p, p.r0, p.r1, p.mine);
	if (e.screen) p.pairs = band_pairs_before(p.r1, g.nA, g.nB, true) - band_pairs_before(p.r0, g.nA, g.nB, true);   // every pair of the band is decided
	else if (p.windowed) p.pairs = p.cum[p.r1] - p.cum[p.r0];       // pairs inside the window: the ones the math evaluates
	else p.pairs = band_pairs_before(p.r1, g.nA, g.nB, g.triangle != 0) - band_pairs_before(p.r0, g.nA, g.nB, g.triangle != 0);
}

}  // namespace twk
