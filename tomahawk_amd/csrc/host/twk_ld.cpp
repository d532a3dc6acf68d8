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

#include <fcntl.h>
#include <unistd.h>

#include "twk_format.h"
#include "twk_hip.h"
#include "twk_parallel.h"
#include "twk_record_sink.h"
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
	std::vector<std::pair<std::string, int64_t>> engine_options;     // twk_ld::SetEngineOption, in the order given
	int64_t option(const char* key, int64_t dflt) const {
		for (const auto& kv : engine_options) if (kv.first == key) dflt = kv.second;
		return dflt;
	}

	// per-variant (rid,pos) of the uploaded selection
	std::vector<uint32_t> rid, pos;

	// Output: one shared writer, one emitter (pair of open forward / reverse blocks) per GPU driver
	// thread - the reference's per-thread blocks behind one spinlocked writer (twk_record_sink.h).
	TwoOutput out;

	// TWK_REF_COMPAT=1 with -w: what the reference's window mode really keeps, as a filter on the
	// exact window's records.  blk_first[k] = first variant of the k-th loaded block.
	struct CompatWindow {
		bool on = false, forced = false;             // forced: -p / -u (the slave loops test every pair)
		uint32_t w = 0;
		std::vector<uint32_t> blk_first;             // [n_blocks + 1]
		std::vector<uint32_t> blk_of;                // [n_variants]
	} cw;
	// Reference, block pair (bi, bj) of a window run:
	//  * the ticker issues (i, j), j = i, i+1, ..., and gives up the rest of row i at the first j != i with
	//    pos(first of j) - pos(last of i) > w - in uint32 arithmetic and without looking at the contigs
	//    (ld_balancing.h:176-203);
	//  * -p / -u: the slave walks the pairs (p, q) of the block pair in order and leaves the WHOLE block pair at
	//    the first same-contig pair with pos(q) - pos(p) > w (goto end_cycle, ld_engine.cpp:2553-2560, 2582-2588).
	//    Positions ascend, so that happens in the first row or never: of a block pair that is not wholly inside
	//    the window only the first variant's in-window pairs come out;
	//  * default mode has no window loop at all (ld_engine.cpp:1841-1843: Calculate): every pair of every block
	//    pair the ticker issued is computed.
	bool compat_keep(uint32_t A, uint32_t B) const {
		const uint32_t bi = cw.blk_of[A], bj = cw.blk_of[B];
		const uint32_t last_i = cw.blk_first[bi + 1] - 1;
		for (uint32_t j = bi + 1; j <= bj; ++j)
			if ((uint32_t)(pos[cw.blk_first[j]] - pos[last_i]) > cw.w) return false;
		if (!cw.forced) return true;
		const uint32_t first_i = cw.blk_first[bi], last_j = cw.blk_first[bj + 1] - 1;
		if (rid[first_i] != rid[last_j] || (uint32_t)(pos[last_j] - pos[first_i]) <= cw.w) return true;      // wholly inside
		return A == first_i && rid[A] == rid[B] && (uint32_t)(pos[B] - pos[A]) <= cw.w;
	}

	bool run(twk_ld_settings& settings, const Header& hdr, const std::vector<twk_hip_ctx*>& ctxs, uint32_t n_samples, const void* spec);
};

twk_ld::twk_ld() : mImpl(new twk_ld_impl) {}
twk_ld::~twk_ld() { delete mImpl; }
void twk_ld::SetEngineOption(const std::string& key, int64_t value) { mImpl->engine_options.emplace_back(key, value); }
uint64_t twk_ld::n_pairs() const { return mImpl->n_pairs; }
uint64_t twk_ld::n_records() const { return mImpl->n_records; }

bool twk_ld::Compute(const twk_ld_settings& s) { settings = s; return Compute(); }
bool twk_ld::ComputeSingle(const twk_ld_settings& s, bool verbose, bool progress) { settings = s; return ComputeSingle(verbose, progress); }
// (the reference's own default build answers the same way: TWK_SLAVE_DEBUG_MODE is 0, lib/ld/ld_engine.h:20, lib/ld/ld.cpp:878-882)
bool twk_ld::ComputePerformance() {
	std::cerr << stamp("ERROR") << "ComputePerformance is a compile-time debug harness of the CPU reference; not available." << std::endl;
	return false;
}
namespace {
struct DeviceCtxs {          // one engine context per GPU of the run
	std::vector<twk_hip_ctx*> ctx;
	~DeviceCtxs() { for (auto* c : ctx) if (c) twk_hip_ctx_destroy(c); }
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
	// Window mode on several GPUs: GPU g holds only its band of rows plus the halo its window reaches, as
	// local variants [0, n_local) = global [first, first + n_local); rows [0, n_band) are its own.
	struct Slab { uint32_t first = 0, n_band = 0, n_local = 0; };
	std::vector<Slab> slabs;     // empty: every GPU holds everything and takes shard g of n of every region
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

// ---- input: .twk blocks -> HBM (ld.cpp:370-465, ld_unpacker.h:44-123) ------------------------------------
// The reference's unpack threads decompress every block and expand every variant's run-length genotypes
// into a bitvector in host memory.  Here the host only decompresses: blocks are grouped into batches of
// <= 32 MB (uncompressed), decode threads read (pread) and zstd-decompress a batch each and walk the
// record headers for the per-variant metadata and the place of the run words; the runs are expanded by
// HIP kernels (twk_hip_upload_rle), so PCIe carries the compressed genotypes.  The batches pass through a
// small ring of page-locked slots to one uploader thread per GPU (load_blocks below).
namespace {
bool pread_all(int fd, void* buf, size_t n, uint64_t off) {
	uint8_t* p = static_cast<uint8_t*>(buf);
	while (n) {
		const ssize_t r = pread(fd, p, n, (off_t)off);
		if (r <= 0) return false;
		p += r; n -= (size_t)r; off += (uint64_t)r;
	}
	return true;
}
struct Staging {            // page-locked when the HIP runtime can give it, plain memory otherwise
	uint8_t* p = nullptr; size_t cap = 0; bool pinned = false;
	bool reserve(size_t n) {
		if (cap >= n) return true;
		release();
		void* q = nullptr;
		if (twk_hip_host_alloc(n, &q) == TWK_HIP_OK && q) { p = static_cast<uint8_t*>(q); pinned = true; }
		else { p = static_cast<uint8_t*>(std::malloc(n)); pinned = false; }
		cap = p ? n : 0;
		return p != nullptr;
	}
	void release() { if (p) { if (pinned) twk_hip_host_free(p); else std::free(p); } p = nullptr; cap = 0; }
	~Staging() { release(); }
};
struct LoadBatch {
	size_t k0 = 0, k1 = 0;                      // blocks [k0, k1) of the selection
	uint32_t first = 0, nv = 0;                 // variants [first, first + nv)
	size_t bytes = 0;
	std::vector<size_t> boff;                   // where each block's bytes start in the staging buffer
	std::vector<twk_hip_rle_desc> desc;
	std::vector<twk_hip_variant_meta> meta;
};
// One decompressed block (core.cpp:245-261: u32 n, u32 m, u32 rid, then n records) -> descriptors + metadata
// of its variants; record = 38-byte header (core.cpp:59-73) + n_runs run words.
bool index_block(const uint8_t* base, size_t block_off, size_t block_len, uint32_t n_expected,
                 twk_hip_rle_desc* desc, twk_hip_variant_meta* meta) {
	const uint8_t* p = base + block_off; const uint8_t* end = p + block_len;
	if (block_len < 12) return false;
	uint32_t n; std::memcpy(&n, p, 4);
	if (n != n_expected) return false;
	p += 12;
	for (uint32_t i = 0; i < n; ++i) {
		if ((size_t)(end - p) < 38) return false;
		const uint8_t pack = p[0];
		const uint32_t width = pack >> 3;
		if (width != 1 && width != 2 && width != 4) return false;       // "illegal gt primitive type" core.cpp:95
		twk_hip_variant_meta& m = meta[i];
		std::memcpy(&m.pos, p + 2, 4); std::memcpy(&m.ac, p + 6, 4); std::memcpy(&m.an, p + 10, 4); std::memcpy(&m.rid, p + 14, 4);
		std::memcpy(&m.hwe, p + 26, 8);
		uint32_t n_write; std::memcpy(&n_write, p + 34, 4);
		const uint32_t n_runs = n_write >> 1, missing = n_write & 1;   // the container's own miss bit (= gt_missing in valid files)
		m.missing = missing; m._pad = 0;
		p += 38;
		if ((uint64_t)n_runs * width > (uint64_t)(end - p)) return false;
		desc[i].offset = (uint64_t)(p - base); desc[i].n_runs = n_runs; desc[i].width = (uint8_t)width; desc[i].missing = (uint8_t)missing; desc[i]._pad = 0;
		p += (size_t)n_runs * width;
	}
	return true;
}
}  // namespace

static bool load_blocks(const std::string& path, const TwkReader& reader, const std::vector<uint32_t>& sel,
                        uint32_t T, const std::vector<twk_hip_ctx*>& ctxs, uint32_t n_samples, std::vector<uint32_t>& rid, std::vector<uint32_t>& pos) {
	using lclock = std::chrono::steady_clock;
	const auto t_begin = lclock::now();
	std::vector<uint32_t> first(sel.size() + 1, 0);
	for (size_t k = 0; k < sel.size(); ++k) first[k + 1] = first[k] + reader.index.ent[sel[k]].n;
	rid.assign(first.back(), 0); pos.assign(first.back(), 0);
	if (sel.empty()) {
		for (auto* c : ctxs) if (!hip_ok(c, twk_hip_set_problem(c, n_samples, 0), "twk_hip_set_problem")) return false;
		return true;
	}
	const int fd = ::open(path.c_str(), O_RDONLY);
	if (fd < 0) { std::cerr << stamp("ERROR") << "Failed to open " << path << std::endl; return false; }
	struct Closer { int fd; ~Closer() { ::close(fd); } } closer{fd};
	// Batches of whole blocks, one decode thread each.  A batch holds at least one block and at most 32 MB
	// (the first batch reaches the GPU after ~30 ms); small inputs are cut finer so that every decode thread
	// gets several.
	const uint32_t W = std::max<uint32_t>(1, std::min<uint32_t>(T, 32));           // decode threads
	size_t total_unc = 0, max_block = 0;
	for (uint32_t k : sel) { const size_t u = ((size_t)reader.index.ent[k].b_unc + 15) / 16 * 16; total_unc += u; max_block = std::max(max_block, u); }
	const size_t batch_cap = std::max(max_block, std::min<size_t>((size_t)32 << 20, std::max<size_t>((size_t)2 << 20, total_unc / (4 * (size_t)W))));
	std::vector<LoadBatch> batches;
	for (size_t k = 0; k < sel.size();) {
		LoadBatch b; b.k0 = k; b.first = first[k];
		while (k < sel.size() && (b.bytes == 0 || b.bytes + reader.index.ent[sel[k]].b_unc <= batch_cap)) {
			b.boff.push_back(b.bytes);
			b.bytes += ((size_t)reader.index.ent[sel[k]].b_unc + 15) / 16 * 16;
			++k;
		}
		b.k1 = k; b.nv = first[k] - b.first;
		batches.push_back(std::move(b));
	}
	size_t max_bytes = 0;
	for (const auto& b : batches) max_bytes = std::max(max_bytes, b.bytes);
	const size_t n_batches = batches.size(), n_gpus = ctxs.size();
	const uint32_t n_workers = (uint32_t)std::min<size_t>(W, n_batches);
	// A decode thread takes the next batch in file order, reads and decompresses its blocks into its own buffer
	// and walks the record headers; when its batch's turn comes it copies the bytes into a slot of a small ring of
	// page-locked memory (page-locking costs ~0.2 s per GB here, unlocking ~0.13 s: the ring is 8 slots, not one
	// per thread).  One uploader thread per GPU takes the batches in file order (copy + device inflate); batch b
	// uses slot b mod 8, free again when every GPU has batch b - 8.  The device allocations of the problem are
	// made before the threads start: 10-20 ms for 12.5 GB then, but up to a second next to 32 threads faulting
	// their buffers in.
	const size_t n_slots = std::min<size_t>(n_batches, 8);
	const auto t_alloc0 = lclock::now();
	{
		std::vector<int> rc(n_gpus, TWK_HIP_OK);
		std::vector<std::thread> th;
		for (size_t g = 0; g < n_gpus; ++g) th.emplace_back([&, g] { rc[g] = twk_hip_set_problem(ctxs[g], n_samples, first.back()); });
		for (auto& t : th) t.join();
		for (size_t g = 0; g < n_gpus; ++g) if (!hip_ok(ctxs[g], rc[g], "twk_hip_set_problem")) return false;
	}
	Staging ring;
	if (!ring.reserve(n_slots * max_bytes)) { std::cerr << stamp("ERROR") << "Out of host memory for the upload staging buffers" << std::endl; return false; }
	const double t_alloc = std::chrono::duration<double>(lclock::now() - t_alloc0).count();
	size_t max_cmp = 0;
	for (uint32_t k : sel) max_cmp = std::max<size_t>(max_cmp, reader.index.ent[k].b_cmp);
	std::mutex mu;
	std::condition_variable cv_slot, cv_ready;
	std::vector<char> ready(n_batches, 0);           // under mu: batch is in its slot
	std::vector<uint32_t> pending(n_batches, (uint32_t)n_gpus);
	size_t released = 0;                             // under mu: batches [0, released) have left their slots
	std::atomic<size_t> next_batch(0);
	std::atomic<bool> failed(false);
	std::string fail_msg;                            // under mu, first failure only
	auto fail = [&](const std::string& msg) {
		std::lock_guard<std::mutex> lk(mu);
		if (!failed.exchange(true)) fail_msg = msg;
		cv_slot.notify_all(); cv_ready.notify_all();
	};
	std::vector<double> busy_decode(n_workers, 0.0), busy_upload(n_gpus, 0.0);

	auto decode_batch = [&](LoadBatch& b, uint8_t* buf, std::vector<uint8_t>& z) -> bool {
		b.desc.resize(b.nv); b.meta.resize(b.nv);
		for (size_t k = b.k0; k < b.k1; ++k) {
			const IndexEntry& e = reader.index.ent[sel[k]];
			uint8_t head[9];
			if (!pread_all(fd, head, 9, e.foff)) return false;
			uint32_t unc, cmp; std::memcpy(&unc, head + 1, 4); std::memcpy(&cmp, head + 5, 4);
			if (head[0] != 1 || unc != e.b_unc || cmp != e.b_cmp) return false;      // the header must agree with the index (sizes the buffers)
			z.resize(cmp);
			if (!pread_all(fd, z.data(), cmp, e.foff + 9)) return false;
			if (!zstd_decompress_into(z.data(), cmp, buf + b.boff[k - b.k0], unc)) return false;
			const uint32_t v0 = first[k] - b.first;
			if (!index_block(buf, b.boff[k - b.k0], unc, e.n, &b.desc[v0], &b.meta[v0])) return false;
		}
		for (uint32_t i = 0; i < b.nv; ++i) { rid[b.first + i] = b.meta[i].rid; pos[b.first + i] = b.meta[i].pos; }
		return true;
	};
	auto decoder = [&](uint32_t w) {
		std::vector<uint8_t> z;                          // compressed block, sized once
		z.reserve(max_cmp + 16);
		std::unique_ptr<uint8_t[]> own(new (std::nothrow) uint8_t[max_bytes]);
		if (!own) { fail("Out of host memory for the decode buffers"); return; }
		for (;;) {
			const size_t bi = next_batch++;
			if (bi >= n_batches || failed) return;
			const auto t0 = lclock::now();
			if (!decode_batch(batches[bi], own.get(), z)) {
				fail("Failed to load blocks " + std::to_string(batches[bi].k0) + "-" + std::to_string(batches[bi].k1) + "!");
				return;
			}
			busy_decode[w] += std::chrono::duration<double>(lclock::now() - t0).count();
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_slot.wait(lk, [&] { return failed || bi < released + n_slots; });
				if (failed) return;
			}
			const auto t1 = lclock::now();
			std::memcpy(ring.p + (bi % n_slots) * max_bytes, own.get(), batches[bi].bytes);
			busy_decode[w] += std::chrono::duration<double>(lclock::now() - t1).count();
			{ std::lock_guard<std::mutex> lk(mu); ready[bi] = 1; }
			cv_ready.notify_all();
		}
	};
	auto uploader = [&](size_t g) {
		for (size_t bi = 0; bi < n_batches; ++bi) {
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_ready.wait(lk, [&] { return failed || ready[bi]; });
				if (failed) return;
			}
			const LoadBatch& b = batches[bi];
			const auto t0 = lclock::now();
			const int rc = twk_hip_upload_rle(ctxs[g], b.first, b.nv, ring.p + (bi % n_slots) * max_bytes, b.bytes, b.desc.data(), b.meta.data());
			busy_upload[g] += std::chrono::duration<double>(lclock::now() - t0).count();
			if (rc != TWK_HIP_OK) {
				const char* m = twk_hip_last_error(ctxs[g]);
				fail(std::string("twk_hip_upload_rle: ") + twk_hip_strerror(rc) + (m && m[0] ? std::string(" (") + m + ")" : std::string()));
				return;
			}
			std::lock_guard<std::mutex> lk(mu);
			if (--pending[bi] == 0) {                         // every GPU is in file order, so batches leave in file order too
				while (released < n_batches && pending[released] == 0) ++released;
				cv_slot.notify_all();
			}
		}
	};
	std::vector<std::thread> th;
	for (size_t g = 0; g < n_gpus; ++g) th.emplace_back(uploader, g);
	for (uint32_t w = 0; w < n_workers; ++w) th.emplace_back(decoder, w);
	for (auto& t : th) t.join();
	if (failed) { std::cerr << stamp("ERROR") << fail_msg << std::endl; return false; }
	double dec = 0, up = 0;
	for (double x : busy_decode) dec += x;
	for (double x : busy_upload) up = std::max(up, x);
	std::cerr << stamp("LOG", "UNPACK") << n_batches << " batches, " << total_unc / 1000000 << " MB of run-length genotypes in "
	          << std::chrono::duration<double>(lclock::now() - t_begin).count() << " s: " << n_workers << " decode threads busy " << dec
	          << " s in all (read + decompress + copy to staging), copy + device inflate " << up
	          << " s per GPU, device + staging allocations " << t_alloc << " s, " << n_slots << " staging slots of "
	          << max_bytes / 1000000 << " MB" << (ring.pinned ? "" : " (not page-locked)") << std::endl;
	return true;
}

// TWK_REF_COMPAT=1: reproduce slips of the reference that change output bytes instead of the correct result -
// PhasedVectorized's tail / padding arithmetic (q6/q7; TWK_HIP_OPT_REF_COMPAT, include/twk_hip.h), scalc's dropping
// of the last partial group of 100 neighbours (q15) and, with -w, the reference's window mode as it really
// behaves (q8; twk_ld_impl::compat_keep).  The off-diagonal-chunk slip (q9) is not reproduced.
static bool ref_compat() { const char* e = std::getenv("TWK_REF_COMPAT"); return e && e[0] && e[0] != '0'; }

// The r2 screen (TWK_HIP_OPT_R2_SCREEN) is on unless TWK_HIP_NO_SCREEN=1 (A/B comparisons; the records are the same).
static bool r2_screen() { const char* e = std::getenv("TWK_HIP_NO_SCREEN"); return !(e && e[0] && e[0] != '0'); }

// TWK_HIP_PART=k/n: this process's share of a multi-process run.
static bool part_from_env(uint32_t& part0, uint32_t& n_procs) {
	part0 = 0; n_procs = 1;
	if (const char* e = std::getenv("TWK_HIP_PART")) {
		unsigned k = 0, n = 1;
		if (sscanf(e, "%u/%u", &k, &n) == 2 && n >= 1 && k < n) { part0 = k; n_procs = n; }
		else { std::cerr << stamp("ERROR") << "Bad TWK_HIP_PART (want k/n): " << e << std::endl; return false; }
	}
	return true;
}

static bool create_devices(DeviceCtxs& dc, int n_gpus, const std::vector<std::pair<std::string, int64_t>>& options) {
	int64_t force_device = -1;                                       // option "force_device" (testing): several engine contexts on one GPU
	for (const auto& kv : options) if (kv.first == "force_device") force_device = kv.second;
	const bool force = force_device >= 0;
	const char* dev_env = std::getenv("TWK_HIP_DEVICE");
	if (twk_hip_abi_version() != TWK_HIP_ABI_VERSION) {      // (struct layouts cross this boundary: twk_hip_timing, twk_hip_record)
		std::cerr << stamp("ERROR", "HIP") << "libtwk_hip has ABI version " << twk_hip_abi_version() << ", this library was built against " << TWK_HIP_ABI_VERSION << "." << std::endl;
		return false;
	}
	const int n_dev = twk_hip_device_count();
	if (n_dev <= 0) { std::cerr << stamp("ERROR", "HIP") << "No HIP device available (this build has no CPU path)." << std::endl; return false; }
	if (!force && n_gpus > n_dev) { std::cerr << stamp("ERROR", "HIP") << "TWK_HIP_GPUS=" << n_gpus << " but only " << n_dev << " device(s) are visible." << std::endl; return false; }
	for (int g = 0; g < n_gpus; ++g) {
		const int device = force ? (int)force_device : (n_gpus == 1 && dev_env ? std::atoi(dev_env) : g);
		twk_hip_ctx* c = nullptr;
		if (!hip_ok(nullptr, twk_hip_ctx_create(device, &c), "twk_hip_ctx_create")) return false;
		dc.ctx.push_back(c);
		for (const auto& kv : options) {
			if (kv.first == "force_device" || kv.first == "progress_ms" || kv.first == "map_output" || kv.first == "emit_workers" || kv.first == "emit_backlog_mb" || kv.first == "emit_queue_pieces" || kv.first == "record_codec" || kv.first == "direct_output" || kv.first == "gather") continue;       // this class's own
			if (!hip_ok(c, twk_hip_set_option(c, kv.first.c_str(), kv.second), "twk_hip_set_option")) return false;
		}
	}
	return true;
}
static int gpus_from_env() {
	if (const char* g = std::getenv("TWK_HIP_GPUS")) { const int n = std::atoi(g); if (n >= 1) return std::min(n, 64); }
	return 1;
}

// compute + write + final log lines (ld.cpp:620-668, ld_progress.h:89-96)
bool twk_ld::twk_ld_impl::run(twk_ld_settings& settings, const Header& hdr, const std::vector<twk_hip_ctx*>& ctxs, uint32_t n_samples, const void* spec_) {
	using clock = std::chrono::steady_clock;
	const RunSpec& spec = *static_cast<const RunSpec*>(spec_);
	if (!open_output(settings, hdr, out.writer)) return false;
	// Engine option "map_output" = 1: blocks go into the file through a shared mapping, copied in by the emitter's workers in
	// parallel (twk_format.h).  Built because round 3's single writer thread looked like the bound of survivor-rich runs;
	// measured on the GPU box it is 2-3x *slower* than the stream (66 M records, 4.3 GB: 2.8-3.0 s against 1.0 s,
	// profiles/r04_writer_ab.txt - 32 threads taking page faults in one address space do not scale there), so the stream
	// stays the default and the mapping an option.
	if (option("map_output", 0)) (void)out.writer.map_output();
	// Engine option "direct_output" = 1: the block frames go into the file with one pwritev() each, its space reserved ahead
	// (twk_format.h) - with the record codec the one stream into the file is what a survivor-rich run waits for.
	else if (option("direct_output", 0)) (void)out.writer.direct_output();
	out.b_size = (uint32_t)std::max(2, settings.b_size);
	// The blocks' zstd frames come from the records' own encoder (twk_repcodec.h) instead of libzstd wherever the run asks for the
	// fastest level (-k <= 1, the reference's default, lib/core.cpp:304): what binds a survivor-rich run is libzstd's level 1 itself
	// (2,504 x 531,500 -p -w 4000000, 37 M records: compute + write 0.69 -> 0.42 s for a file 6 % larger).  -k 2 and up asks
	// for ratio and gets libzstd at that level; engine option "record_codec" = 0 / 1 says so either way.
	const bool own_codec = option("record_codec", -1) < 0 ? settings.c_level <= 1 : option("record_codec", -1) != 0;
	out.c_level = own_codec ? (int)RECORD_CODEC_LEVEL + std::max(-999, std::min(settings.c_level, 999)) : settings.c_level;
	out.rid = rid.data(); out.pos = pos.data(); out.n_variants = rid.size(); out.n_records = 0;
	n_records = 0; n_pairs = 0;
	const int n_gpus = (int)ctxs.size();
	// output workers per GPU: 32 at most (beyond that the one placing step and the memory system are the limit)
	// (and never more threads than the process has CPUs - its affinity mask, its container's quota: util::usable_cpus)
	int n_workers = std::max(1, std::min(std::max(1, std::min(settings.n_threads, util::usable_cpus()) / n_gpus), 32));
	if (option("emit_workers", 0) > 0) n_workers = (int)std::min<int64_t>(option("emit_workers", 0), 64);      // (measurement: the emitter's worker threads per GPU)
	const int mode = settings.single ? TWK_HIP_MODE_AUTO
	               : settings.force_phased ? TWK_HIP_MODE_PHASED : (settings.forced_unphased ? TWK_HIP_MODE_UNPHASED : TWK_HIP_MODE_AUTO);
	twk_hip_filters f{settings.minR2, settings.maxR2, settings.minDprime, settings.maxDprime, settings.minP};
	// Shards: GPU g of this process takes part k * n_gpus + g of n * n_gpus of every region, where k/n is
	// this process's share of a multi-node (farm) run: TWK_HIP_PART=k/n (the reference's -c/-C idea,
	// ld_balancing.h:23-80, with equal-area row bands; `concat` merges the processes' outputs).
	uint32_t part0 = 0, n_procs = 1;
	if (!part_from_env(part0, n_procs)) return false;
	const uint32_t n_parts = n_procs * (uint32_t)n_gpus;
	const auto t0 = clock::now();
	// Progress lines like the reference's ticker (ld_progress.h:40-86), every 30 s, driven by the
	// engines' per-tile callbacks instead of a polling thread.
	struct Progress {
		twk_ld_impl* self; clock::time_point t0, last; uint64_t total = 0; uint32_t n_s = 0; bool header = false;
		double every = 30.0;
		std::mutex mu;
		std::vector<uint64_t> done, base;         // per GPU: pairs finished in the current / earlier region calls
		struct PerGpu { Progress* p; int g; };
		static void cb(void* u, uint64_t pairs_done, uint32_t, uint32_t) {
			PerGpu& pg = *static_cast<PerGpu*>(u);
			Progress& p = *pg.p;
			std::lock_guard<std::mutex> lk(p.mu);
			p.done[pg.g] = pairs_done;
			const auto now = clock::now();
			if (std::chrono::duration<double>(now - p.last).count() < p.every) return;
			p.last = now;
			if (!p.header) {
				std::cerr << stamp("PROGRESS") << std::setw(12) << "Time elapsed" << std::setw(15) << "Variants" << std::setw(20) << "Genotypes"
				          << std::setw(15) << "Output" << std::setw(10) << "Progress" << "\tEst. Time left" << std::endl;
				p.header = true;
			}
			const double sec = std::chrono::duration<double>(now - p.t0).count();
			uint64_t done = 0;
			for (size_t g = 0; g < p.done.size(); ++g) done += p.done[g] + p.base[g];
			const double frac = p.total ? (double)done / (double)p.total : 0.0;
			uint64_t n_out;
			{ std::lock_guard<std::mutex> lo(p.self->out.mu); n_out = p.self->out.n_records; }
			std::cerr << stamp("PROGRESS") << std::setw(12) << elapsed_string(sec) << std::setw(15) << pretty(done)
			          << std::setw(20) << pretty(done * p.n_s) << std::setw(15) << pretty(n_out)
			          << std::setw(10) << frac * 100 << "%\t" << (frac > 0 ? elapsed_string(sec * (1 - frac) / frac) : std::string("-")) << std::endl;
		}
	} progress;
	progress.self = this; progress.t0 = progress.last = t0; progress.n_s = n_samples;
	progress.done.assign(n_gpus, 0); progress.base.assign(n_gpus, 0);
	progress.every = (double)option("progress_ms", (int64_t)(progress.every * 1000.0)) / 1000.0;      // (test hook)
	progress.total = (spec.triangleA && spec.nA > 1 ? (uint64_t)spec.nA * (spec.nA - 1) / 2 : 0) + (spec.rectAB ? (uint64_t)spec.nA * spec.nB : 0);
	if (n_procs > 1) progress.total /= n_procs;
	std::vector<Progress::PerGpu> per_gpu(n_gpus);

	// one driver thread per GPU: region calls for its shard, survivors into its own emitter
	struct Driver {
		twk_ld_impl* self;
		std::vector<twk_hip_record> kept;      // declared before the emitter: its workers read it until they are joined
		RecordEmitter emitter; bool write_failed = false; uint64_t pairs = 0; int rc = TWK_HIP_OK;
		RecordHandOff handoff;                 // between the engine's thread (sink below) and the emitter (twk_record_sink.h)
		uint32_t shift = 0;
		Driver(twk_ld_impl* s, int workers, size_t backlog, size_t pieces) : self(s), emitter(s->out, workers, backlog), handoff(emitter, pieces) {}
		bool drain() {
			if (!handoff.drain()) write_failed = true;
			return !write_failed;
		}
		static int sink(void* user, const twk_hip_record* recs, uint64_t n) {
			auto* d = static_cast<Driver*>(user);
			const bool edit = d->shift || d->self->cw.on;
			// slab-local variant indices -> positions in the run's rid / pos arrays; TWK_REF_COMPAT window filter
			auto fix = [d](twk_hip_record& r) -> bool {
				r.idxA += d->shift; r.idxB += d->shift;
				return !d->self->cw.on || d->self->compat_keep(r.idxA, r.idxB);
			};
			if (d->handoff.takes(n)) {
				if (!(edit ? d->handoff.put(recs, n, fix) : d->handoff.put(recs, n))) { d->write_failed = true; return 1; }
				return 0;
			}
			if (!d->drain()) return 1;           // (no queue, or a piece larger than its buffers: in order behind what is queued)
			if (edit) {
				d->kept.clear();
				for (uint64_t i = 0; i < n; ++i) {
					twk_hip_record r = recs[i];
					if (fix(r)) d->kept.push_back(r);
				}
				recs = d->kept.data(); n = d->kept.size();
			}
			if (!d->emitter.emit(recs, n, false, true)) { d->write_failed = true; return 1; }     // (the engine's survivors come sorted)
			return 0;
		}
	};
	std::vector<std::unique_ptr<Driver>> drivers;
	// expanded blocks that may wait in memory for the compressing workers, per GPU (twk_record_sink.h; measurement: none pays)
	const size_t backlog = (size_t)std::max<int64_t>(0, option("emit_backlog_mb", 0)) << 20;
	// buffers of the hand-off between the engine's thread and the emitter, per GPU (Driver above; 0: none, the sink feeds the emitter itself)
	// (8: 2,504 x 531,500 `-p` 1.02 -> 0.86 s, `-u` 1.57 -> 1.42 s of compute + write; 32 and 64 no better; window runs, which wait
	// for the compression whatever the queue holds, within their noise - profiles/r04_band_sort_ab.txt)
	const size_t pieces = (size_t)std::max<int64_t>(0, option("emit_queue_pieces", 8));
	for (int g = 0; g < n_gpus; ++g) drivers.emplace_back(new Driver(this, n_workers, backlog, pieces));
	// Engine option "gather" = 1 (the north star's "final RCCL gather of .two output blocks over xGMI", inside this one process): every GPU keeps
	// its survivors in HBM (twk_hip_set_device_sink), and when all are done they travel GPU to GPU over RCCL into GPU 0's sink
	// (twk_hip_gather_records: one group of exact-size ncclSend / ncclRecv), from where they leave for the host once, into ONE emitter.  Default 0:
	// every GPU's driver thread streams its survivors to the host while it computes (one hop fewer for a file sink, and no bound on the
	// survivors by GPU 0's memory: INTEGRATION 1).  With one GPU the records take the same calls in a loop from the sink to itself.
	const bool gather = option("gather", 0) != 0;
	if (gather) for (int g = 0; g < n_gpus; ++g) if (!hip_ok(ctxs[g], twk_hip_set_device_sink(ctxs[g], 1), "twk_hip_set_device_sink")) return false;
	auto drive = [&](int g) {
		Driver& d = *drivers[g];
		twk_hip_ctx* ctx = ctxs[g];
		const uint32_t part = part0 * (uint32_t)n_gpus + (uint32_t)g;
		per_gpu[g] = Progress::PerGpu{&progress, g};
		twk_hip_set_progress(ctx, &Progress::cb, &per_gpu[g]);
		uint64_t np = 0, nr = 0;
		if (!spec.slabs.empty()) {          // this GPU's own slab: its band of rows against band + halo, nothing to shard
			const RunSpec::Slab& sl = spec.slabs[g];
			d.shift = sl.first;
			if (sl.n_band && sl.n_local > 1)
				d.rc = twk_hip_ld_region(ctx, mode, &f, 0, sl.n_band, 0, sl.n_local, 1, 0, 1, 0, spec.options, spec.l_window, &Driver::sink, &d, &np, &nr);
			d.pairs += np;
		} else
		if (spec.triangleA && spec.nA > 1) {
			d.rc = twk_hip_ld_region(ctx, mode, &f, 0, spec.nA, 0, spec.nA, 1, part, n_parts, 0, spec.options, spec.l_window, &Driver::sink, &d, &np, &nr);
			d.pairs += np;
			std::lock_guard<std::mutex> lk(progress.mu);
			progress.base[g] += np; progress.done[g] = 0;
		}
		if (spec.slabs.empty() && d.rc == TWK_HIP_OK && spec.rectAB && spec.nA && spec.nB) {
			d.rc = twk_hip_ld_region(ctx, mode, &f, 0, spec.nA, spec.nA, spec.nB, 0, part, n_parts, 0, spec.options, spec.l_window, &Driver::sink, &d, &np, &nr);
			d.pairs += np;
		}
		twk_hip_set_progress(ctx, nullptr, nullptr);
		if (gather) return;                 // (the survivors are still in HBM: gathered and written below)
		if (!d.drain()) d.write_failed = true;
		if (d.rc == TWK_HIP_OK && !d.write_failed && !d.emitter.emit(nullptr, 0, true)) d.write_failed = true;     // close the open blocks
	};
	if (n_gpus == 1) drive(0);
	else {
		std::vector<std::thread> th;
		for (int g = 0; g < n_gpus; ++g) th.emplace_back(drive, g);
		for (auto& t : th) t.join();
	}
	if (gather) {
		for (int g = 0; g < n_gpus; ++g) if (!hip_ok(ctxs[g], drivers[g]->rc, "twk_hip_ld_region")) return false;
		// what every GPU holds (its slab's index shift applies to its records wherever they travel), then the gather, then GPU 0 -> host -> emitter 0
		struct Feed {
			Driver* d0; std::vector<uint64_t> end; std::vector<uint32_t> shift; uint64_t seen = 0; std::vector<twk_hip_record> tmp;
			static int sink(void* user, const twk_hip_record* recs, uint64_t n) {
				auto* f = static_cast<Feed*>(user);
				uint64_t i = 0;
				while (i < n) {
					size_t g = 0;
					while (g + 1 < f->end.size() && f->seen >= f->end[g]) ++g;
					const uint64_t m = std::min<uint64_t>(n - i, f->end[g] > f->seen ? f->end[g] - f->seen : n - i);
					f->tmp.clear();
					for (uint64_t k = 0; k < m; ++k) {
						twk_hip_record r = recs[i + k];
						r.idxA += f->shift[g]; r.idxB += f->shift[g];
						if (!f->d0->self->cw.on || f->d0->self->compat_keep(r.idxA, r.idxB)) f->tmp.push_back(r);
					}
					// (a piece of the sink may span launches: each launch's records are in order, the piece need not be - the emitter sorts what is not)
					if (!f->tmp.empty() && !f->d0->emitter.emit(f->tmp.data(), f->tmp.size(), false, false)) { f->d0->write_failed = true; return 1; }
					f->seen += m; i += m;
				}
				return 0;
			}
		} feed;
		feed.d0 = drivers[0].get();
		uint64_t total = 0;
		std::vector<uint64_t> held(n_gpus, 0);
		for (int g = 0; g < n_gpus; ++g) {
			const twk_hip_record* p = nullptr;
			if (!hip_ok(ctxs[g], twk_hip_device_records(ctxs[g], &p, &held[g]), "twk_hip_device_records")) return false;
			total += held[g];
			feed.end.push_back(total); feed.shift.push_back(drivers[g]->shift);
		}
		const auto t_g = clock::now();
		uint64_t got = 0; double xfer_ms = 0;
		int grc = twk_hip_gather_records(ctxs.data(), (uint32_t)n_gpus, 0, n_gpus == 1 ? TWK_HIP_GATHER_SELF_LOOP : 0, &got, &xfer_ms);
		if (grc == TWK_HIP_OK) {
			if (got != total) { std::cerr << stamp("ERROR", "HIP") << "gather: " << got << " records arrived, " << total << " were held." << std::endl; return false; }
			std::cerr << stamp("LOG", "HIP") << "Gathered " << pretty(total) << " records of " << n_gpus << " GPU(s) into GPU 0 over " << twk_hip_gather_backend()
			          << (n_gpus == 1 ? " (one GPU: from its sink to itself)" : "") << ": " << pretty((total - held[0]) * sizeof(twk_hip_record)) << " bytes moved"
			          << (n_gpus == 1 ? " (+ the loop's " + pretty(held[0] * sizeof(twk_hip_record)) + ")" : std::string()) << ", transfers " << xfer_ms << " ms, "
			          << std::chrono::duration<double, std::milli>(clock::now() - t_g).count() << " ms in all." << std::endl;
			uint64_t nd = 0;
			if (!hip_ok(ctxs[0], twk_hip_drain_device_sink(ctxs[0], &Feed::sink, &feed, &nd), "twk_hip_drain_device_sink")) return false;
		} else {
			// no RCCL, or not enough room on GPU 0 for everybody's survivors: every GPU's sink goes to the host by itself, one after the other
			std::cerr << stamp("WARNING", "HIP") << "gather over RCCL not possible (" << twk_hip_last_error(ctxs[0]) << "): the GPUs' survivors go to the host one GPU at a time." << std::endl;
			for (int g = 0; g < n_gpus; ++g) {
				uint64_t nd = 0;
				if (!hip_ok(ctxs[g], twk_hip_drain_device_sink(ctxs[g], &Feed::sink, &feed, &nd), "twk_hip_drain_device_sink")) return false;
			}
		}
		for (int g = 0; g < n_gpus; ++g) (void)twk_hip_set_device_sink(ctxs[g], 0);
		if (!drivers[0]->write_failed && !drivers[0]->emitter.emit(nullptr, 0, true)) drivers[0]->write_failed = true;
	}
	for (int g = 0; g < n_gpus; ++g) {
		if (drivers[g]->write_failed) { std::cerr << stamp("ERROR", "WRITER") << "Failed to write output block!" << std::endl; return false; }
		if (!hip_ok(ctxs[g], drivers[g]->rc, "twk_hip_ld_region")) return false;
		n_pairs += drivers[g]->pairs;
	}
	n_records = out.n_records;
	const double sec = std::chrono::duration<double>(clock::now() - t0).count();
	std::cerr << stamp("PROGRESS") << "Finished in " << elapsed_string(sec) << ". Variants: " << pretty(n_pairs) << ", genotypes: "
	          << pretty(n_pairs * n_samples) << ", output: " << pretty(n_records) << std::endl;
	std::cerr << stamp("PROGRESS") << pretty((uint64_t)(n_pairs / std::max(sec, 1e-9))) << " variants/s and "
	          << pretty((uint64_t)((double)n_pairs * n_samples / std::max(sec, 1e-9))) << " genotypes/s" << std::endl;
	for (int g = 0; g < n_gpus; ++g) {
		twk_hip_timing tm;
		if (twk_hip_timing_get(ctxs[g], &tm) == TWK_HIP_OK)
			std::cerr << stamp("LOG", "HIP") << (n_gpus > 1 ? "GPU " + std::to_string(g) + ": " : std::string()) << "count kernel " << tm.count_ms << " ms in "
			          << tm.count_launches << " launches ("
			          // what the launches issued: one AND+popcount (6 cycles per wave) per word of a plane-row pair - three for every four of them in
			          // the three-product form (v_and / v_bitop3 + v_bcnt: no other instruction since round 6)
			          << (tm.count_ms > 0 ? ((double)(tm.row_pairs - tm.three_row_pairs) + (double)tm.three_row_pairs * 0.75) * (double)tm.words_per_row / (tm.count_ms * 1e-3) / 2.62e13 * 100.0 : 0.0)
			          << " % of the and+bcnt issue ceiling over the tiles it contracted"
			          << (tm.count_wall_ticks ? "; its blocks ran at " + std::to_string((int)((double)tm.count_shader_cycles / (double)tm.count_wall_ticks * 100.0 + 0.5)) + " MHz" : std::string())
			          << "), math kernels " << tm.stats_ms << " ms"
			          << (tm.fused_launches ? "; " + std::to_string(tm.fused_launches) + " launches fused count -> r2 screen, " + pretty(tm.candidates) + " candidate slots" : std::string())
			          << (tm.three_launches ? "; " + std::to_string(tm.three_launches) + " launches in the three-product form (HH + S), " + pretty(tm.recount_candidates) + " candidates recounted" : std::string())
			          << (tm.list_launches ? "; carrier-list kernel " + std::to_string(tm.list_ms) + " ms in " + std::to_string(tm.list_launches) + " launches over " + pretty(tm.list_pairs) + " rare pairs" : std::string())
			          << (tm.probe_launches ? "; probe kernel " + std::to_string(tm.probe_ms) + " ms in " + std::to_string(tm.probe_launches) + " launches over " + pretty(tm.probe_pairs) + " rare x common pairs" : std::string())
			          << std::endl;
	}
	{
		double t_sort = 0, t_blocks = 0;
		for (int g = 0; g < n_gpus; ++g) { t_sort = std::max(t_sort, drivers[g]->emitter.t_sort); t_blocks = std::max(t_blocks, drivers[g]->emitter.t_blocks); }
		std::cerr << stamp("LOG", "WRITER") << pretty(out.n_blocks) << " blocks, " << out.bytes_packed / 1000000 << " MB compressed; the producer spent "
		          << t_sort + t_blocks << " s handing its survivors over (the engine's thread " << drivers[0]->handoff.t_copy << " s copying them out of its buffer); workers: expanding " << drivers[0]->emitter.ns_expand.load() * 1e-9 << " s, compressing "
		          << drivers[0]->emitter.ns_pack.load() * 1e-9 << " s in all; writer thread " << drivers[0]->emitter.ns_write.load() * 1e-9 << " s" << (n_gpus > 1 ? " (slowest GPU's emitter)" : "") << std::endl;
	}
	if (!out.writer.close()) { std::cerr << stamp("ERROR", "WRITER") << "Failed to write final block!" << std::endl; return false; }
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

	const int n_gpus = gpus_from_env();
	DeviceCtxs dc;
	if (!create_devices(dc, n_gpus, mImpl->engine_options)) return false;
	if (n_gpus > 1) std::cerr << stamp("LOG", "HIP") << "Using " << n_gpus << " GPUs: one driver thread each, equal-area row bands of the pair space..." << std::endl;
	uint32_t part0 = 0, n_procs = 1;
	if (!part_from_env(part0, n_procs)) return false;
	const uint32_t n_parts = n_procs * (uint32_t)n_gpus;
	RunSpec spec;
	const auto t_load = clock::now();
	const uint32_t T = (uint32_t)std::max(1, std::min(settings.n_threads, util::usable_cpus()));
	std::cerr << stamp("LOG", "THREAD") << "Unpacking using " << T << " threads..."
	          << (T < (uint32_t)settings.n_threads ? " (of " + std::to_string(settings.n_threads) + " asked for: the process may use " + std::to_string(util::usable_cpus()) + " CPUs)" : std::string()) << std::endl;
	// Window mode on several GPUs (or processes): a GPU needs only the blocks of its band of rows plus the blocks its
	// window reaches beyond it (the reference's ticker prunes block pairs the same way, ld_balancing.h:176-203) -
	// for BASELINE configs[4] that is 75 GB per GPU instead of 500 GB.  Bands are cut at block boundaries with equal
	// estimated in-window pairs, from the index alone (block span and size), identically in every process.
	if (settings.window && bal.diag && n_parts > 1 && !(ref_compat() && !(settings.force_phased || settings.forced_unphased))) {
		const uint64_t w = (uint64_t)std::max(0, settings.l_window);
		const size_t nb = sel.size();
		auto ent = [&](size_t k) -> const IndexEntry& { return reader.index.ent[sel[k]]; };
		std::vector<size_t> reach(nb);                       // first block beyond what block k's window can touch
		std::vector<uint64_t> nvar(nb + 1, 0), cost(nb + 1, 0);
		for (size_t k = 0; k < nb; ++k) nvar[k + 1] = nvar[k] + ent(k).n;
		for (size_t k = 0, e = 0; k < nb; ++k) {
			if (e < k + 1) e = k + 1;
			while (e < nb && ent(e).rid == ent(k).rid && (uint64_t)ent(e).minpos <= (uint64_t)ent(k).maxpos + w) ++e;
			reach[k] = e;
			const uint64_t n = ent(k).n;
			cost[k + 1] = cost[k] + n * (n - 1) / 2 + n * (nvar[e] - nvar[k + 1]);
		}
		auto boundary = [&](uint32_t p) -> size_t {
			if (p == 0) return 0;
			if (p >= n_parts) return nb;
			const long double target = (long double)cost[nb] * p / n_parts;
			return (size_t)(std::lower_bound(cost.begin(), cost.end(), (uint64_t)target) - cost.begin());
		};
		std::vector<std::vector<uint32_t>> sel_g(n_gpus);
		for (int g = 0; g < n_gpus; ++g) {
			const uint32_t p = part0 * (uint32_t)n_gpus + (uint32_t)g;
			const size_t B0 = boundary(p), B1 = std::max(B0, boundary(p + 1));
			const size_t H1 = B1 > B0 ? std::max(B1, reach[B1 - 1]) : B1;
			RunSpec::Slab sl;
			sl.first = (uint32_t)nvar[B0]; sl.n_band = (uint32_t)(nvar[B1] - nvar[B0]); sl.n_local = (uint32_t)(nvar[H1] - nvar[B0]);
			spec.slabs.push_back(sl);
			for (size_t k = B0; k < H1; ++k) sel_g[g].push_back(sel[k]);
			std::cerr << stamp("LOG", "BALANCING") << "GPU " << g << ": rows = variants [" << sl.first << ", " << sl.first + sl.n_band << "), + "
			          << sl.n_local - sl.n_band << " halo variants" << std::endl;
		}
		mImpl->rid.assign(M, 0); mImpl->pos.assign(M, 0);
		std::vector<std::vector<uint32_t>> rid_g(n_gpus), pos_g(n_gpus);
		std::vector<char> ok(n_gpus, 1);
		std::vector<std::thread> th;
		for (int g = 0; g < n_gpus; ++g) th.emplace_back([&, g] {
			if (sel_g[g].empty()) return;
			if (!load_blocks(settings.in, reader, sel_g[g], std::max<uint32_t>(1, T / (uint32_t)n_gpus), {dc.ctx[g]}, n_samples, rid_g[g], pos_g[g])) ok[g] = 0;
		});
		for (auto& t : th) t.join();
		for (int g = 0; g < n_gpus; ++g) {
			if (!ok[g]) return false;
			std::copy(rid_g[g].begin(), rid_g[g].end(), mImpl->rid.begin() + spec.slabs[g].first);
			std::copy(pos_g[g].begin(), pos_g[g].end(), mImpl->pos.begin() + spec.slabs[g].first);
		}
	} else {
		if (!load_blocks(settings.in, reader, sel, T, dc.ctx, n_samples, mImpl->rid, mImpl->pos)) return false;
	}
	std::cerr << stamp("LOG") << "Unpacked and uploaded " << pretty(M) << " variants. "
	          << elapsed_string(std::chrono::duration<double>(clock::now() - t_load).count()) << std::endl;

	spec.nA = bal.diag ? M : nL; spec.nB = bal.diag ? 0 : nR;
	spec.triangleA = bal.diag; spec.rectAB = !bal.diag;
	spec.options = (settings.window ? TWK_HIP_OPT_WINDOW : 0) | (ref_compat() ? TWK_HIP_OPT_REF_COMPAT : 0) | (r2_screen() ? TWK_HIP_OPT_R2_SCREEN : 0);
	spec.l_window = (uint32_t)settings.l_window;
	mImpl->cw = twk_ld_impl::CompatWindow();
	if (ref_compat() && settings.window) {
		auto& cw = mImpl->cw;
		cw.on = true; cw.w = (uint32_t)settings.l_window;
		cw.forced = settings.force_phased || settings.forced_unphased;
		cw.blk_first.assign(1, 0);
		for (uint32_t b : sel) cw.blk_first.push_back(cw.blk_first.back() + reader.index.ent[b].n);
		cw.blk_of.resize(M);
		for (size_t k = 0; k + 1 < cw.blk_first.size(); ++k) for (uint32_t v = cw.blk_first[k]; v < cw.blk_first[k + 1]; ++v) cw.blk_of[v] = (uint32_t)k;
		// default mode: the reference tests no pair against the window, so pairs outside it come out too: compute them all
		if (!cw.forced) spec.options &= ~TWK_HIP_OPT_WINDOW;
	}
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
	if (ref_compat()) others.resize(others.size() / 100 * 100);      // the reference keeps the neighbours in full groups of 100 only (ld.cpp:203-205, 239-244)
	if (others.empty()) { std::cerr << "no surrounding variants" << std::endl; return false; }
	const uint32_t nT = (uint32_t)targets.size(), nO = (uint32_t)others.size(), M = nT + nO;
	if (verbose) std::cerr << stamp("LOG") << pretty(nT) << " target and " << pretty(nO) << " surrounding variants..." << std::endl;

	DeviceCtxs dc;
	if (!create_devices(dc, 1, mImpl->engine_options)) return false;
	twk_hip_ctx* ctx = dc.ctx[0];
	if (!hip_ok(ctx, twk_hip_set_problem(ctx, n_samples, M), "twk_hip_set_problem")) return false;
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
	if (!hip_ok(ctx, twk_hip_upload_bitvectors(ctx, 0, M, data.data(), any_mask ? mask.data() : nullptr, w64, meta.data()),
	            "twk_hip_upload_bitvectors")) return false;
	RunSpec spec;
	spec.nA = nT; spec.nB = nO; spec.triangleA = true; spec.rectAB = true;
	spec.options = TWK_HIP_OPT_KEEP_LOW_AC | (ref_compat() ? TWK_HIP_OPT_REF_COMPAT : 0);      // the skip is commented out in CalculateSingle (:2267-2269)
	return mImpl->run(settings, reader.hdr, dc.ctx, n_samples, &spec);
}

}  // namespace tomahawk
