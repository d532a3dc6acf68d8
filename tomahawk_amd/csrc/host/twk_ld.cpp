// tomahawk::twk_ld on the MI355X engine.
//
// Host orchestration that the reference does in lib/ld/ld.cpp:477-671 (Compute):
// open the .twk, pick the chunk of the block triangle (-c/-C), unpack the blocks
// to bitvectors on worker threads, hand them to the device through the C ABI
// (include/twk_hip.h), stream the surviving pairs back and write the .two file
// (forward + reverse copies in separate blocks, ld_engine.cpp:1270-1309).
#include "twk_ld.h"

#include <algorithm>
#include <memory>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <ctime>
#include <iomanip>
#include <iostream>
#include <regex>
#include <sstream>
#include <thread>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <sys/time.h>
#include <cstdio>

#include "twk_format.h"
#include "twk_hip.h"
#include "twk_parallel.h"
#include "twk_util.h"

#ifndef TWK_AMD_VERSION
#define TWK_AMD_VERSION "0.7.0-mi355x"
#endif

namespace tomahawk {

using namespace util;

namespace {
// twk_ld_balancer::Build (lib/ld/ld_balancing.h:23-80), including its chunk
// arithmetic (last chunk of a row is `chunk_size` blocks ending at n_blocks).
struct Balancer {
	bool diag = true;
	uint32_t fromL = 0, toL = 0, fromR = 0, toR = 0;
	bool build(uint32_t n_blocks, uint32_t parts, uint32_t chosen) {
		if (chosen >= parts) { std::cerr << stamp("ERROR", "BALANCER") << "Illegal chosen block: " << chosen << " >= " << parts << std::endl; return false; }
		if (parts > n_blocks) { std::cerr << stamp("ERROR", "BALANCER") << "Illegal desired number of blocks! You are asking for more subproblems than there are blocks available (" << parts << ">" << n_blocks << ")..." << std::endl; return false; }
		if (parts == 1) { fromL = 0; toL = n_blocks; fromR = 0; toR = n_blocks; diag = true; return true; }
		uint32_t factor = 0;
		for (uint32_t i = 1; i < parts; ++i) if ((((i * i) - i) / 2) + i == parts) { factor = i; break; }
		if (factor == 0) { std::cerr << stamp("ERROR", "BALANCER") << "Could not partition into " << parts << " number of subproblems. This number is not a function of x!2 + x..." << std::endl; return false; }
		const uint32_t chunk = n_blocks / factor;
		for (uint32_t i = 0, k = 0; i < factor; ++i) {
			for (uint32_t j = i; j < factor; ++j, ++k) {
				if (k != chosen) continue;
				toR = (j + 1 == factor ? n_blocks : chunk * (j + 1)); fromR = toR - chunk;
				toL = (i + 1 == factor ? n_blocks : chunk * (i + 1)); fromL = toL - chunk;
				diag = (i == j);
				return true;
			}
		}
		return true;
	}
};
}  // namespace

// ---- settings (lib/core.cpp:297-332) -----------------------------------------------------
twk_ld_settings::twk_ld_settings()
    : square(true), window(false), low_memory(false), bitmaps(false), single(false), force_phased(false),
      forced_unphased(false), force_cross_intervals(false), c_level(1), bl_size(500), b_size(10000),
      l_window(1000000), n_threads((int32_t)std::thread::hardware_concurrency()), cycle_threshold(0),
      ldd_load_type(TWK_LDD_ALL), l_surrounding(500000), out("-"), minP(1), minR2(0.1), maxR2(100),
      minDprime(0), maxDprime(100), n_chunks(1), c_chunk(0) {}

std::string twk_ld_settings::GetString() const {
	auto tf = [](bool b) { return std::string(b ? "TRUE" : "FALSE"); };
	return "square=" + tf(square) + ",window=" + tf(window) + ",low_memory=" + tf(low_memory) + ",bitmaps=" + tf(bitmaps)
	     + ",single=" + tf(single) + ",force_phased=" + tf(force_phased) + ",force_unphased=" + tf(forced_unphased)
	     + ",compression_level=" + std::to_string(c_level) + ",block_size=" + std::to_string(bl_size)
	     + ",output_block_size=" + std::to_string(b_size) + (window ? ",window_size=" + std::to_string(l_window) : "")
	     + ",l_surrounding=" + std::to_string(l_surrounding) + ",minP=" + std::to_string(minP) + ",minR2=" + std::to_string(minR2)
	     + ",maxR2=" + std::to_string(maxR2) + ",minDprime=" + std::to_string(minDprime) + ",maxDprime=" + std::to_string(maxDprime)
	     + ",n_chunks=" + std::to_string(n_chunks) + ",c_chunk=" + std::to_string(c_chunk) + ",n_threads=" + std::to_string(n_threads)
	     + ",ldd_type=" + std::to_string((int)ldd_load_type) + ",cycle_threshold=" + std::to_string(cycle_threshold);
}

// ---- implementation ---------------------------------------------------------------------------
class twk_ld::twk_ld_impl {
public:
	uint64_t n_pairs = 0, n_records = 0;

	// per-variant (rid,pos) of the uploaded selection
	std::vector<uint32_t> rid, pos;

	// Output state.  The reference keeps one forward and one reverse block per thread
	// (ld_engine.h:321) and flushes both when the forward block is full or the contig pair of
	// the next record differs from the block's first (ld_engine.cpp:1270-1281).  Here the
	// survivors of one tile arrive together: they are put in (row, col) order with a parallel
	// key sort, cut into blocks by that same rule, and the blocks are expanded to forward +
	// reverse records and compressed on worker threads while this thread writes them in order.
	// The last, still open block carries over to the next tile.
	TwoWriter writer;
	uint32_t b_size = 10000;
	int c_level = 1, n_workers = 1;
	bool write_failed = false;
	std::vector<twk_hip_record> carry;            // records of the open block (< b_size), in order

	void expand(const twk_hip_record& r, TwoRecord& f, TwoRecord& v) const {
		f.controller = (uint16_t)r.flags;
		f.ridA = rid[r.idxA]; f.ridB = rid[r.idxB];
		f.packA = pos[r.idxA] << 2; f.packB = pos[r.idxB] << 2;
		std::memcpy(f.cnt, r.cnt, sizeof(f.cnt));
		f.D = r.D; f.Dprime = r.Dprime; f.R = r.R; f.R2 = r.R2; f.P = r.P;
		f.ChiSqFisher = r.ChiSqFisher; f.ChiSqModel = r.ChiSqModel;
		v = f;                           // reverse copy swaps (rid,pos) only; cnt is NOT transposed (:1292-1298)
		std::swap(v.ridA, v.ridB); std::swap(v.packA, v.packB);
	}

	// Write the survivors `recs[0..n)` (any order) behind the carried block; final: close the open block too.
	bool emit(const twk_hip_record* recs, uint64_t n, bool final) {
		// (row, col) order: the file is deterministic (the reference's order is thread-timing dependent)
		par::Raw<par::SortKey> keys;
		keys.alloc(n);
		{
			const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_workers, n / 65536 + 1));
			std::vector<std::thread> th;
			for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
				for (uint64_t i = n * (uint64_t)t / T, e = n * (uint64_t)(t + 1) / T; i < e; ++i)
					keys[i] = par::SortKey{0, (uint64_t)recs[i].idxA << 32 | recs[i].idxB, (uint32_t)i};
			});
			for (auto& x : th) x.join();
		}
		par::parallel_sort(keys, n_workers);
		// the sequence is carry[0..nc) followed by recs[keys[.].idx]
		const uint64_t nc = carry.size(), total = nc + n;
		auto at = [&](uint64_t i) -> const twk_hip_record& { return i < nc ? carry[i] : recs[keys[i - nc].idx]; };
		// cuts by the flush rule: a block ends when it holds b_size records or the next record's
		// (ridA, ridB) differs from its first record's
		std::vector<uint64_t> cut{0};
		if (total) {
			uint32_t fa = rid[at(0).idxA], fb = rid[at(0).idxB];
			for (uint64_t i = 1; i < total; ++i) {
				const twk_hip_record& r = at(i);
				const uint32_t ra = rid[r.idxA], rb = rid[r.idxB];
				if (i - cut.back() == b_size || ra != fa || rb != fb) { cut.push_back(i); fa = ra; fb = rb; }
			}
		}
		// the block after the last cut stays open unless this is the end
		const size_t n_closed = total ? (final ? cut.size() : cut.size() - 1) : 0;
		if (final && total) cut.push_back(total);
		struct Slot { std::vector<TwoRecord> f, v; TwoWriter::Packed pf, pv; };
		const int level = c_level;
		std::function<bool(size_t, Slot&)> produce = [&](size_t b, Slot& s) -> bool {
			const uint64_t lo = cut[b], hi = cut[b + 1];
			s.f.resize(hi - lo); s.v.resize(hi - lo);
			for (uint64_t i = lo; i < hi; ++i) expand(at(i), s.f[i - lo], s.v[i - lo]);
			return TwoWriter::pack(s.f.data(), (uint32_t)s.f.size(), level, s.pf) && TwoWriter::pack(s.v.data(), (uint32_t)s.v.size(), level, s.pv);
		};
		std::function<bool(size_t, Slot&)> consume = [&](size_t, Slot& s) -> bool {   // CompressBlock (:1804-1810): forward, then reverse
			return writer.write_packed(s.pf) && writer.write_packed(s.pv);
		};
		if (n_closed && !par::ordered_parallel<Slot>(n_closed, n_workers, produce, consume)) return false;
		std::vector<twk_hip_record> next;
		if (!final && total) { next.reserve(total - cut.back()); for (uint64_t i = cut.back(); i < total; ++i) next.push_back(at(i)); }
		carry.swap(next);
		return true;
	}

	bool run(twk_ld_settings& settings, const Header& hdr, twk_hip_ctx* ctx, uint32_t n_samples, const void* spec);

	static int sink(void* user, const twk_hip_record* recs, uint64_t n) {
		auto* self = static_cast<twk_ld_impl*>(user);
		if (!self->emit(recs, n, false)) { self->write_failed = true; return 1; }
		self->n_records += 2 * n;
		return 0;
	}
};

twk_ld::twk_ld() : mImpl(new twk_ld_impl) {}
twk_ld::~twk_ld() { delete mImpl; }
uint64_t twk_ld::n_pairs() const { return mImpl->n_pairs; }
uint64_t twk_ld::n_records() const { return mImpl->n_records; }

bool twk_ld::Compute(const twk_ld_settings& s) { settings = s; return Compute(); }
bool twk_ld::ComputeSingle(const twk_ld_settings& s, bool verbose, bool progress) { settings = s; return ComputeSingle(verbose, progress); }
bool twk_ld::ComputePerformance() {
	std::cerr << stamp("ERROR") << "ComputePerformance is a compile-time debug harness of the CPU reference; not available." << std::endl;
	return false;
}
namespace {
struct DeviceCtx {
	twk_hip_ctx* ctx = nullptr;
	~DeviceCtx() { if (ctx) twk_hip_ctx_destroy(ctx); }
};
bool hip_ok(twk_hip_ctx* ctx, int rc, const char* what) {
	if (rc == TWK_HIP_OK) return true;
	std::cerr << stamp("ERROR", "HIP") << what << ": " << twk_hip_strerror(rc);
	if (ctx && twk_hip_last_error(ctx)[0]) std::cerr << " (" << twk_hip_last_error(ctx) << ")";
	std::cerr << std::endl;
	return false;
}
}  // namespace

// ---- interval strings (lib/intervals.cpp:96-139; regexes include/tomahawk.h:57-59) -------------
struct Interval { int32_t rid; uint32_t from, to; };
static bool parse_interval(const std::string& str, const Header& hdr, Interval& out) {
	static const std::regex re_range("^[A-Za-z0-9\\-_]+\\:[0-9]+([\\.]{1}[0-9]+){0,1}([eE]{1}[0-9]{1})?\\-[0-9]+([\\.]{1}[0-9]+){0,1}([eE]{1}[0-9]{1})?$");
	static const std::regex re_pos("^[A-Za-z0-9\\-_]+\\:[0-9]+([\\.]{1}[0-9]+){0,1}([eE]{1}[0-9]{1})?$");
	static const std::regex re_contig("^[A-Za-z0-9\\-_]+$");
	auto contig = [&](const std::string& name) -> int {
		const int id = hdr.contig_id(name);
		if (id < 0) std::cerr << stamp("ERROR", "INTERVAL") << "Contig does not exist in string " << str << std::endl;
		return id;
	};
	if (std::regex_match(str, re_range)) {
		const size_t c = str.find(':'), d = str.find('-', c);
		const int id = contig(str.substr(0, c));
		if (id < 0) return false;
		out = {hdr.contigs[id].idx == (uint32_t)id ? id : (int32_t)hdr.contigs[id].idx,
		       (uint32_t)std::atof(str.substr(c + 1, d - c - 1).c_str()), (uint32_t)std::atof(str.substr(d + 1).c_str())};
		return true;
	}
	if (std::regex_match(str, re_pos)) {
		const size_t c = str.find(':');
		const int id = contig(str.substr(0, c));
		if (id < 0) return false;
		const uint32_t p = (uint32_t)std::atof(str.substr(c + 1).c_str());
		out = {(int32_t)hdr.contigs[id].idx, p, p + 1};
		return true;
	}
	if (std::regex_match(str, re_contig)) {
		const int id = contig(str);
		if (id < 0) return false;
		out = {(int32_t)hdr.contigs[id].idx, 0, (uint32_t)hdr.contigs[id].n_bases};
		return true;
	}
	std::cerr << stamp("ERROR", "INTERVAL") << "Illegal format: " << str << std::endl;
	return false;
}
// Index::FindOverlap (lib/index.cpp:124-133): blocks with rid match and [minpos,maxpos] meeting [from,to].
static void overlapping_blocks(const TwkIndex& idx, const Interval& iv, std::vector<uint32_t>& out) {
	for (size_t i = 0; i < idx.ent.size(); ++i)
		if (idx.ent[i].rid == iv.rid && idx.ent[i].minpos <= iv.to && idx.ent[i].maxpos >= iv.from) out.push_back((uint32_t)i);
}

namespace {
// Everything after the variants are in HBM: output file, compute, final statistics.
struct RunSpec {
	uint32_t nA = 0, nB = 0;     // variants [0,nA) are the row set, [nA,nA+nB) the column set (nB = 0: one set)
	bool triangleA = true;       // all pairs inside the row set
	bool rectAB = false;         // row set x column set
	int options = 0;             // TWK_HIP_OPT_*
	uint32_t l_window = 0;
};
}  // namespace

static bool open_output(twk_ld_settings& settings, const Header& in_hdr, TwoWriter& writer) {   // ld.cpp:583-618
	std::string out = settings.out;
	if (out.empty() || out == "-") {
		std::cerr << stamp("LOG", "WRITER") << "Writing to stdout..." << std::endl;
		out = "-";
	} else {
		const std::string ext = extension(out);
		if (!(ext.size() == 3 && strncasecmp(ext.c_str(), "two", 3) == 0)) {
			const std::string bp = base_path(out);
			out = (bp.size() ? bp + "/" : "") + base_name(out) + ".two";
		}
		std::cerr << stamp("LOG", "WRITER") << "Opening " << out << "..." << std::endl;
	}
	settings.out = out;
	Header hdr = in_hdr;
	hdr.literals += "\n##tomahawk_calcVersion=" + std::string(TWK_AMD_VERSION) + "\n";
	hdr.literals += "##tomahawk_calcCommand=" + command_line() + "; Date=" + datetime() + "\n";
	if (!writer.open(out, hdr, settings.c_level)) { std::cerr << stamp("ERROR", "WRITER") << "Failed to open file: " << out << "..." << std::endl; return false; }
	return true;
}

// Unpack `sel` blocks (ld.cpp:370-465, ld_unpacker.h) in batches on T threads and upload them in file order.
static bool load_blocks(const std::string& path, const TwkReader& reader, const std::vector<uint32_t>& sel,
                        uint32_t n_samples, uint32_t T, twk_hip_ctx* ctx, std::vector<uint32_t>& rid, std::vector<uint32_t>& pos) {
	const size_t w64 = ((size_t)2 * n_samples + 63) / 64;
	std::vector<uint32_t> first(sel.size() + 1, 0);
	for (size_t k = 0; k < sel.size(); ++k) first[k + 1] = first[k] + reader.index.ent[sel[k]].n;
	rid.assign(first.back(), 0); pos.assign(first.back(), 0);
	const size_t batch_bytes = (size_t)512 << 20;
	size_t k0 = 0;
	while (k0 < sel.size()) {
		size_t k1 = k0; uint32_t nv = 0;
		while (k1 < sel.size() && (nv == 0 || (size_t)(nv + reader.index.ent[sel[k1]].n) * w64 * 8 <= batch_bytes)) nv += reader.index.ent[sel[k1++]].n;
		std::vector<uint64_t> data((size_t)nv * w64), mask;
		std::vector<twk_hip_variant_meta> meta(nv);
		std::vector<uint8_t> has_mask(k1 - k0, 0);
		std::vector<Block> blocks(k1 - k0);
		std::atomic<size_t> next(k0);
		std::atomic<bool> failed(false);
		auto reader_job = [&]() {
			TwkReader rd;
			if (!rd.open(path)) { failed = true; return; }
			for (size_t k = next++; k < k1; k = next++) {
				Block& blk = blocks[k - k0];
				if (!rd.read_block(sel[k], blk) || blk.rcds.size() != reader.index.ent[sel[k]].n) { failed = true; return; }
				for (const auto& v : blk.rcds) if (v.gt_missing) has_mask[k - k0] = 1;
			}
		};
		const uint32_t nt = std::min<uint32_t>(T, (uint32_t)(k1 - k0));
		{ std::vector<std::thread> th; for (uint32_t t = 0; t < nt; ++t) th.emplace_back(reader_job); for (auto& t : th) t.join(); }
		if (failed) { std::cerr << stamp("ERROR") << "Failed to load blocks " << k0 << "-" << k1 << "!" << std::endl; return false; }
		const bool any_mask = std::any_of(has_mask.begin(), has_mask.end(), [](uint8_t x) { return x != 0; });
		if (any_mask) mask.assign((size_t)nv * w64, 0);
		next = k0;
		auto build_job = [&]() {
			for (size_t k = next++; k < k1; k = next++) {
				const Block& blk = blocks[k - k0];
				const uint32_t base = first[k] - first[k0];
				for (size_t i = 0; i < blk.rcds.size(); ++i) {
					const Variant& v = blk.rcds[i];
					uint64_t* d = &data[(size_t)(base + i) * w64];
					uint64_t* m = (any_mask && v.gt_missing) ? &mask[(size_t)(base + i) * w64] : nullptr;
					if (!v.build_bitvector(n_samples, d, m)) { failed = true; return; }
					twk_hip_variant_meta& mm = meta[base + i];
					mm.ac = v.ac; mm.an = v.an; mm.pos = v.pos; mm.rid = v.rid; mm.missing = v.gt_missing ? 1 : 0; mm._pad = 0; mm.hwe = v.hwe;
				}
			}
		};
		{ std::vector<std::thread> th; for (uint32_t t = 0; t < nt; ++t) th.emplace_back(build_job); for (auto& t : th) t.join(); }
		if (failed) { std::cerr << stamp("ERROR") << "Corrupt genotype runs in blocks " << k0 << "-" << k1 << "!" << std::endl; return false; }
		for (uint32_t i = 0; i < nv; ++i) { rid[first[k0] + i] = meta[i].rid; pos[first[k0] + i] = meta[i].pos; }
		if (!hip_ok(ctx, twk_hip_upload_bitvectors(ctx, first[k0], nv, data.data(), any_mask ? mask.data() : nullptr, w64, meta.data()),
		            "twk_hip_upload_bitvectors")) return false;
		k0 = k1;
	}
	return true;
}

static bool create_device(DeviceCtx& dc) {
	const char* dev_env = std::getenv("TWK_HIP_FORCE_DEVICE");        // testing: several workers on one GPU
	if (!dev_env) dev_env = std::getenv("TWK_HIP_DEVICE");
	const int device = dev_env ? std::atoi(dev_env) : 0;
	if (twk_hip_device_count() <= 0) { std::cerr << stamp("ERROR", "HIP") << "No HIP device available (this build has no CPU path)." << std::endl; return false; }
	return hip_ok(nullptr, twk_hip_ctx_create(device, &dc.ctx), "twk_hip_ctx_create");
}

// compute + write + final log lines (ld.cpp:620-668, ld_progress.h:89-96)
bool twk_ld::twk_ld_impl::run(twk_ld_settings& settings, const Header& hdr, twk_hip_ctx* ctx, uint32_t n_samples, const void* spec_) {
	using clock = std::chrono::steady_clock;
	const RunSpec& spec = *static_cast<const RunSpec*>(spec_);
	if (!open_output(settings, hdr, writer)) return false;
	b_size = (uint32_t)std::max(2, settings.b_size);
	carry.clear(); write_failed = false; n_records = 0; n_pairs = 0;
	c_level = settings.c_level; n_workers = std::min(std::max(1, settings.n_threads), 64);
	const int mode = settings.single ? TWK_HIP_MODE_AUTO
	               : settings.force_phased ? TWK_HIP_MODE_PHASED : (settings.forced_unphased ? TWK_HIP_MODE_UNPHASED : TWK_HIP_MODE_AUTO);
	twk_hip_filters f{settings.minR2, settings.maxR2, settings.minDprime, settings.maxDprime, settings.minP};
	// One process per GPU: shard k of n of every region (TWK_HIP_PART=k/n, set by the multi-GPU launcher).
	uint32_t part = 0, n_parts = 1;
	if (const char* e = std::getenv("TWK_HIP_PART")) {
		unsigned k = 0, n = 1;
		if (sscanf(e, "%u/%u", &k, &n) == 2 && n >= 1 && k < n) { part = k; n_parts = n; }
		else { std::cerr << stamp("ERROR") << "Bad TWK_HIP_PART (want k/n): " << e << std::endl; return false; }
	}
	const auto t0 = clock::now();
	uint64_t np = 0, nr = 0;
	int rc = TWK_HIP_OK;
	// Progress lines like the reference's ticker (ld_progress.h:40-86), every 30 s, driven by the
	// engine's per-tile callback instead of a polling thread.
	struct Progress {
		twk_ld_impl* self; clock::time_point t0, last; uint64_t base = 0, total = 0; uint32_t n_s = 0; bool header = false;
		double every = 30.0;
		static void cb(void* u, uint64_t pairs_done, uint32_t, uint32_t) {
			Progress& p = *static_cast<Progress*>(u);
			const auto now = clock::now();
			if (std::chrono::duration<double>(now - p.last).count() < p.every) return;
			p.last = now;
			if (!p.header) {
				std::cerr << stamp("PROGRESS") << std::setw(12) << "Time elapsed" << std::setw(15) << "Variants" << std::setw(20) << "Genotypes"
				          << std::setw(15) << "Output" << std::setw(10) << "Progress" << "\tEst. Time left" << std::endl;
				p.header = true;
			}
			const double sec = std::chrono::duration<double>(now - p.t0).count();
			const uint64_t done = p.base + pairs_done;
			const double frac = p.total ? (double)done / (double)p.total : 0.0;
			std::cerr << stamp("PROGRESS") << std::setw(12) << elapsed_string(sec) << std::setw(15) << pretty(done)
			          << std::setw(20) << pretty(done * p.n_s) << std::setw(15) << pretty(p.self->n_records)
			          << std::setw(10) << frac * 100 << "%\t" << (frac > 0 ? elapsed_string(sec * (1 - frac) / frac) : std::string("-")) << std::endl;
		}
	} progress;
	progress.self = this; progress.t0 = progress.last = t0; progress.n_s = n_samples;
	if (const char* e = std::getenv("TWK_HIP_PROGRESS_SECONDS")) progress.every = std::atof(e);      // test hook
	progress.total = (spec.triangleA && spec.nA > 1 ? (uint64_t)spec.nA * (spec.nA - 1) / 2 : 0) + (spec.rectAB ? (uint64_t)spec.nA * spec.nB : 0);
	if (n_parts > 1) progress.total /= n_parts;
	twk_hip_set_progress(ctx, &Progress::cb, &progress);
	struct ProgressOff { twk_hip_ctx* c; ~ProgressOff() { twk_hip_set_progress(c, nullptr, nullptr); } } progress_off{ctx};
	if (spec.triangleA && spec.nA > 1) {
		rc = twk_hip_ld_region(ctx, mode, &f, 0, spec.nA, 0, spec.nA, 1, part, n_parts, 0, spec.options, spec.l_window, sink, this, &np, &nr);
		n_pairs += np;
		progress.base = n_pairs;
	}
	if (rc == TWK_HIP_OK && spec.rectAB && spec.nA && spec.nB) {
		rc = twk_hip_ld_region(ctx, mode, &f, 0, spec.nA, spec.nA, spec.nB, 0, part, n_parts, 0, spec.options, spec.l_window, sink, this, &np, &nr);
		n_pairs += np;
	}
	if (write_failed) { std::cerr << stamp("ERROR", "WRITER") << "Failed to write output block!" << std::endl; return false; }
	if (!hip_ok(ctx, rc, "twk_hip_ld_region")) return false;
	if (!emit(nullptr, 0, true)) { std::cerr << stamp("ERROR", "WRITER") << "Failed to write output block!" << std::endl; return false; }
	const double sec = std::chrono::duration<double>(clock::now() - t0).count();
	std::cerr << stamp("PROGRESS") << "Finished in " << elapsed_string(sec) << ". Variants: " << pretty(n_pairs) << ", genotypes: "
	          << pretty(n_pairs * n_samples) << ", output: " << pretty(n_records) << std::endl;
	std::cerr << stamp("PROGRESS") << pretty((uint64_t)(n_pairs / std::max(sec, 1e-9))) << " variants/s and "
	          << pretty((uint64_t)((double)n_pairs * n_samples / std::max(sec, 1e-9))) << " genotypes/s" << std::endl;
	twk_hip_timing tm;
	if (twk_hip_timing_get(ctx, &tm) == TWK_HIP_OK)
		std::cerr << stamp("LOG", "HIP") << "count kernel " << tm.count_ms << " ms in " << tm.count_launches << " launches, math kernel "
		          << tm.stats_ms << " ms" << std::endl;
	if (!writer.close()) { std::cerr << stamp("ERROR", "WRITER") << "Failed to write final block!" << std::endl; return false; }
	return true;
}

bool twk_ld::Compute() {
	using clock = std::chrono::steady_clock;
	mImpl->n_pairs = mImpl->n_records = 0;
	if (settings.in.empty()) { std::cerr << stamp("ERROR") << "No file-name provided..." << std::endl; return false; }
	if (settings.window && settings.n_chunks != 1) { std::cerr << stamp("ERROR") << "Cannot use chunking in window mode!" << std::endl; return false; }
	if (settings.bitmaps || settings.low_memory)
		std::cerr << stamp("LOG") << "Note: -m/-M are CPU memory-saving modes; the GPU engine keeps dense bit-planes in HBM." << std::endl;

	std::cerr << stamp("LOG", "READER") << "Opening " << settings.in << "..." << std::endl;
	TwkReader reader;
	if (!reader.open(settings.in)) { std::cerr << stamp("ERROR") << "Failed to open file: " << settings.in << "... (" << reader.error << ")" << std::endl; return false; }
	const uint32_t n_samples = (uint32_t)reader.hdr.samples.size();
	std::cerr << stamp("LOG") << "Samples: " << pretty(n_samples) << "..." << std::endl;

	// The universe of blocks: all of them, or (-I) those overlapping the intervals
	// (ld.cpp:523-527, 257-277: whole blocks are loaded, like the reference).
	std::vector<uint32_t> universe;
	if (settings.ival_strings.empty()) {
		for (uint32_t b = 0; b < reader.index.ent.size(); ++b) universe.push_back(b);
	} else {
		for (const auto& str : settings.ival_strings) {
			Interval iv;
			if (!parse_interval(str, reader.hdr, iv)) return false;
			overlapping_blocks(reader.index, iv, universe);
		}
		std::sort(universe.begin(), universe.end());
		universe.erase(std::unique(universe.begin(), universe.end()), universe.end());
		if (universe.empty()) { std::cerr << stamp("ERROR", "INTERVAL") << "Found no blocks overlapping the provided range(s)..." << std::endl; return false; }
	}
	const uint32_t n_blocks = (uint32_t)universe.size();
	if (n_blocks == 0 || n_samples == 0) { std::cerr << stamp("ERROR") << "No valid data available..." << std::endl; return true; }

	if (settings.window) settings.c_chunk = 0;
	Balancer bal;
	if (!bal.build(n_blocks, (uint32_t)settings.n_chunks, (uint32_t)settings.c_chunk)) return false;
	std::cerr << stamp("LOG", "BALANCING") << "Using ranges [" << bal.fromL << "-" << bal.toL << "," << bal.fromR << "-" << bal.toR
	          << "] in " << (settings.window ? "window mode" : "square mode") << "..." << std::endl;

	// Selected blocks: the L range, then (square chunk only) the R range.
	std::vector<uint32_t> sel;
	uint32_t nL = 0, nR = 0;
	for (uint32_t b = bal.fromL; b < bal.toL; ++b) { sel.push_back(universe[b]); nL += reader.index.ent[universe[b]].n; }
	if (!bal.diag) for (uint32_t b = bal.fromR; b < bal.toR; ++b) { sel.push_back(universe[b]); nR += reader.index.ent[universe[b]].n; }
	const uint32_t M = nL + nR;
	const uint64_t n_cmp = bal.diag ? (uint64_t)M * (M - 1) / 2 : (uint64_t)nL * nR;
	std::cerr << stamp("LOG") << pretty(M) << " variants from " << pretty(sel.size()) << " blocks..." << std::endl;
	std::cerr << stamp("LOG", "PARAMS") << settings.GetString() << std::endl;
	std::cerr << stamp("LOG") << "Performing: " << pretty(n_cmp) << " variant comparisons..." << std::endl;

	DeviceCtx dc;
	if (!create_device(dc)) return false;
	if (!hip_ok(dc.ctx, twk_hip_set_problem(dc.ctx, n_samples, M), "twk_hip_set_problem")) return false;
	const auto t_load = clock::now();
	const uint32_t T = (uint32_t)std::max(1, settings.n_threads);
	std::cerr << stamp("LOG", "THREAD") << "Unpacking using " << T << " threads..." << std::endl;
	if (!load_blocks(settings.in, reader, sel, n_samples, T, dc.ctx, mImpl->rid, mImpl->pos)) return false;
	std::cerr << stamp("LOG") << "Unpacked and uploaded " << pretty(M) << " variants. "
	          << elapsed_string(std::chrono::duration<double>(clock::now() - t_load).count()) << std::endl;

	RunSpec spec;
	spec.nA = bal.diag ? M : nL; spec.nB = bal.diag ? 0 : nR;
	spec.triangleA = bal.diag; spec.rectAB = !bal.diag;
	spec.options = settings.window ? TWK_HIP_OPT_WINDOW : 0; spec.l_window = (uint32_t)settings.l_window;
	if (!mImpl->run(settings, reader.hdr, dc.ctx, n_samples, &spec)) return false;
	std::cerr << stamp("LOG", "PROGRESS") << "All done..." << elapsed_string(std::chrono::duration<double>(clock::now() - t_load).count()) << "!" << std::endl;
	return true;
}

// scalc: one target site against its neighbourhood (ld.cpp:673-876, LoadTargetSingle :123-255,
// CalculateSingle ld_engine.cpp:2226-2332).
bool twk_ld::ComputeSingle(bool verbose, bool) {
	mImpl->n_pairs = mImpl->n_records = 0;
	if (settings.in.empty()) { std::cerr << stamp("ERROR") << "No file-name provided..." << std::endl; return false; }
	if (settings.n_chunks != 1) { std::cerr << stamp("ERROR") << "Cannot use chunking in single mode!" << std::endl; return false; }
	if (settings.window) { std::cerr << stamp("ERROR") << "Cannot use window mode when running in single mode!" << std::endl; return false; }
	if (settings.ival_strings.size() != 1) { std::cerr << stamp("ERROR") << "Single mode requires exactly one target interval (-I)..." << std::endl; return false; }
	settings.single = true;
	if (verbose) std::cerr << stamp("LOG", "READER") << "Opening " << settings.in << "..." << std::endl;
	TwkReader reader;
	if (!reader.open(settings.in)) { std::cerr << stamp("ERROR") << "Failed to open file: " << settings.in << "... (" << reader.error << ")" << std::endl; return false; }
	const uint32_t n_samples = (uint32_t)reader.hdr.samples.size();
	Interval tgt;
	if (!parse_interval(settings.ival_strings[0], reader.hdr, tgt)) return false;
	// Flanks (ld.cpp:145-153), 1-based inclusive matching of pos+1 (ld.cpp:192):
	//   left  [max(from - l_surrounding, 0), max(from - 1, 0)]    right [to, to + l_surrounding]
	const uint32_t L = (uint32_t)std::max(0, settings.l_surrounding);
	const uint32_t left_lo = tgt.from > L ? tgt.from - L : 0, left_hi = tgt.from > 0 ? tgt.from - 1 : 0;
	const uint32_t right_lo = tgt.to, right_hi = tgt.to + L;
	std::vector<uint32_t> blocks;
	overlapping_blocks(reader.index, Interval{tgt.rid, left_lo, right_hi}, blocks);
	if (blocks.empty()) { std::cerr << stamp("ERROR", "INTERVAL") << "Found no blocks overlapping the provided range(s)..." << std::endl; return false; }

	// Neighbourhoods are small: read the overlapping blocks whole, keep targets first then the rest.
	std::vector<Variant> targets, others;
	for (uint32_t b : blocks) {
		Block blk;
		if (!reader.read_block(b, blk)) { std::cerr << stamp("ERROR") << "Failed to load block " << b << "..." << std::endl; return false; }
		for (auto& v : blk.rcds) {
			if ((int32_t)v.rid != tgt.rid) continue;
			const uint32_t p1 = v.pos + 1;
			if (p1 >= tgt.from && p1 <= tgt.to) targets.push_back(std::move(v));
			else if ((p1 >= left_lo && p1 <= left_hi) || (p1 >= right_lo && p1 <= right_hi)) others.push_back(std::move(v));
		}
	}
	if (targets.empty()) { std::cerr << "no data found for reference" << std::endl; return false; }
	if (others.empty()) { std::cerr << "no surrounding variants" << std::endl; return false; }
	const uint32_t nT = (uint32_t)targets.size(), nO = (uint32_t)others.size(), M = nT + nO;
	if (verbose) std::cerr << stamp("LOG") << pretty(nT) << " target and " << pretty(nO) << " surrounding variants..." << std::endl;

	DeviceCtx dc;
	if (!create_device(dc)) return false;
	if (!hip_ok(dc.ctx, twk_hip_set_problem(dc.ctx, n_samples, M), "twk_hip_set_problem")) return false;
	const size_t w64 = ((size_t)2 * n_samples + 63) / 64;
	std::vector<uint64_t> data((size_t)M * w64), mask;
	std::vector<twk_hip_variant_meta> meta(M);
	bool any_mask = false;
	for (uint32_t i = 0; i < M; ++i) any_mask |= (i < nT ? targets[i] : others[i - nT]).gt_missing;
	if (any_mask) mask.assign((size_t)M * w64, 0);
	mImpl->rid.assign(M, 0); mImpl->pos.assign(M, 0);
	for (uint32_t i = 0; i < M; ++i) {
		const Variant& v = i < nT ? targets[i] : others[i - nT];
		if (!v.build_bitvector(n_samples, &data[(size_t)i * w64], (any_mask && v.gt_missing) ? &mask[(size_t)i * w64] : nullptr)) {
			std::cerr << stamp("ERROR") << "Corrupt genotype runs!" << std::endl; return false;
		}
		twk_hip_variant_meta& mm = meta[i];
		mm.ac = v.ac; mm.an = v.an; mm.pos = v.pos; mm.rid = v.rid; mm.missing = v.gt_missing ? 1 : 0; mm._pad = 0; mm.hwe = v.hwe;
		mImpl->rid[i] = v.rid; mImpl->pos[i] = v.pos;
	}
	if (!hip_ok(dc.ctx, twk_hip_upload_bitvectors(dc.ctx, 0, M, data.data(), any_mask ? mask.data() : nullptr, w64, meta.data()),
	            "twk_hip_upload_bitvectors")) return false;
	RunSpec spec;
	spec.nA = nT; spec.nB = nO; spec.triangleA = true; spec.rectAB = true;
	spec.options = TWK_HIP_OPT_KEEP_LOW_AC;      // the skip is commented out in CalculateSingle (:2267-2269)
	return mImpl->run(settings, reader.hdr, dc.ctx, n_samples, &spec);
}

}  // namespace tomahawk
