// `view` and `sort` of .two files -- see twk_two_tools.h.
#include "twk_two_tools.h"

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <memory>
#include <mutex>
#include <queue>
#include <regex>
#include <sstream>
#include <unistd.h>

#include "twk_parallel.h"
#include "twk_util.h"

namespace tomahawk {
using namespace util;
using namespace par;

// ---- record filters (include/two_reader.h:153-191) ----------------------------------------------
bool TwoFilter::pass(const TwoRecord& r) const {
	const uint32_t v = filter_vec;
	if (v == 0) return true;
	auto in = [](double x, double lo, double hi) { return x >= lo && x <= hi; };
	if ((v >> R2 & 1) && !in(r.R2, minR2, maxR2)) return false;
	if ((v >> D & 1) && !in(r.D, minD, maxD)) return false;
	if ((v >> DPRIME & 1) && !in(r.Dprime, minDprime, maxDprime)) return false;
	if ((v >> P & 1) && !in(r.P, minP, maxP)) return false;
	if ((v >> HAPA & 1) && !in(r.cnt[0], hA_min, hA_max)) return false;
	if ((v >> HAPB & 1) && !in(r.cnt[1], hB_min, hB_max)) return false;
	if ((v >> HAPC & 1) && !in(r.cnt[2], hC_min, hC_max)) return false;
	if ((v >> HAPD & 1) && !in(r.cnt[3], hD_min, hD_max)) return false;
	// two_reader.h:165: the R test reads the R2 bounds (minR/maxR are set by -z/-Z but never read)
	if ((v >> R & 1) && !in(r.R, minR2, maxR2)) return false;
	// two_reader.h:166-171: both tests hold only within one contig
	if ((v >> UPPER & 1) && !(r.ridA <= r.ridB && (r.ridA == r.ridB && r.Apos() < r.Bpos()))) return false;
	if ((v >> LOWER & 1) && !(r.ridB <= r.ridA && (r.ridA == r.ridB && r.Bpos() < r.Apos()))) return false;
	if (v >> MHC & 1) {                       // two_reader.h:173-181: everything but the largest cell
		int m = r.cnt[0] > r.cnt[1] ? 0 : 1;
		if (r.cnt[2] > r.cnt[m]) m = 2;
		if (r.cnt[3] > r.cnt[m]) m = 3;
		double c = 0;
		for (int i = 0; i < 4; ++i) c += (i == m ? 0 : r.cnt[i]);
		if (!in(c, mhc_min, mhc_max)) return false;
	}
	if ((v >> FLAGS & 1) && !((r.controller & flag_include) && ((r.controller & flag_exclude) == 0))) return false;
	if ((v >> CHI & 1) && !in(r.ChiSqFisher, minChi, maxChi)) return false;
	if ((v >> CHIMODEL & 1) && !in(r.ChiSqModel, minChiModel, maxChiModel)) return false;
	return true;
}

// ---- intervals (lib/intervals.cpp:157-403) ---------------------------------------------------------
namespace {
std::vector<std::string> split(const std::string& s, char delim) {     // utility.cpp:43-51, empties dropped
	std::vector<std::string> out;
	std::stringstream ss(s);
	std::string item;
	while (std::getline(ss, item, delim)) if (!item.empty()) out.push_back(item);
	return out;
}
const std::regex& re_range() { static const std::regex r("^[A-Za-z0-9\\-_]+\\:[0-9]+([\\.]{1}[0-9]+){0,1}([eE]{1}[0-9]{1})?\\-[0-9]+([\\.]{1}[0-9]+){0,1}([eE]{1}[0-9]{1})?$"); return r; }
const std::regex& re_pos()   { static const std::regex r("^[A-Za-z0-9\\-_]+\\:[0-9]+([\\.]{1}[0-9]+){0,1}([eE]{1}[0-9]{1})?$"); return r; }
const std::regex& re_contig(){ static const std::regex r("^[A-Za-z0-9\\-_]+$"); return r; }
}  // namespace

bool TwoIntervals::parse(const std::string& s, const Header& hdr, std::string& error) {
	if (s.empty()) return false;
	const std::vector<std::string> sides = split(s, ',');
	if (sides.size() > 2 || sides.empty()) { error = "Illegal format: " + s; return false; }
	// one side: (rid, from, to); returns the position of the new interval in its contig's vector
	auto one = [&](const std::string& t, int32_t& rid, uint32_t& off) -> bool {
		uint32_t from = 0, to = 0;
		std::string name;
		if (std::regex_match(t, re_range())) {
			const std::vector<std::string> a = split(t, ':');
			if (a.size() != 2) { error = "Illegal format: " + s; return false; }
			const std::vector<std::string> p = split(a[1], '-');
			if (p.size() < 2) { error = "Illegal format: " + s; return false; }
			name = a[0]; from = (uint32_t)std::atof(p[0].c_str()); to = (uint32_t)std::atof(p[1].c_str());
		} else if (std::regex_match(t, re_pos())) {
			const std::vector<std::string> a = split(t, ':');
			name = a[0]; from = to = (uint32_t)std::atof(a[1].c_str());
		} else if (std::regex_match(t, re_contig())) {
			name = t;
		} else {
			return false;                       // intervals.cpp:279, :339: silently
		}
		const int id = hdr.contig_id(name);
		if (id < 0) { error = "Contig does not exist in string " + s; return false; }
		rid = (int32_t)hdr.contigs[id].idx;
		if (rid < 0 || (size_t)rid >= ivecs_.size()) { error = "Contig does not exist in string " + s; return false; }
		if (name == t) { from = 0; to = (uint32_t)hdr.contigs[id].n_bases; }
		ivecs_[rid].push_back(Ival{from, to, -1, 0, 0});
		off = (uint32_t)ivecs_[rid].size() - 1;
		return true;
	};
	int32_t ridA = -1, ridB = -1; uint32_t offA = 0, offB = 0;
	if (sides.size() == 1) return one(s, ridA, offA);
	if (!one(sides[0], ridA, offA) || !one(sides[1], ridB, offB)) return false;
	Ival& a = ivecs_[ridA][offA];
	a.mate_rid = ridB; a.mate_off = offB; a.mate = 0;          // intervals.cpp:345-346
	Ival& b = ivecs_[ridB][offB];
	b.mate_rid = ridA; b.mate_off = offA; b.mate = 1;
	return true;
}

bool TwoIntervals::build(const std::vector<std::string>& strings, const Header& hdr, const TwoIndex& index, std::string& error) {
	if (strings.empty()) return true;
	if (hdr.contigs.empty()) return false;
	ivecs_.assign(hdr.contigs.size(), {});
	blocks.clear();
	for (const auto& s : strings) if (!parse(s, hdr, error)) return false;
	// Dedupe (intervals.cpp:354-379): sorted copy, neighbours that touch are merged; only used to
	// pick index blocks (Index::FindOverlap, index.cpp:231-240)
	for (size_t rid = 0; rid < ivecs_.size(); ++rid) {
		std::vector<Ival> v = ivecs_[rid];
		if (v.empty()) continue;
		std::sort(v.begin(), v.end(), [](const Ival& x, const Ival& y) { return x.start < y.start || (x.start == y.start && x.stop < y.stop); });
		std::vector<Ival> merged{v[0]};
		for (size_t j = 1; j < v.size(); ++j) {
			if (v[j].start <= merged.back().stop && v[j].stop >= merged.back().start) merged.back().stop = v[j].stop;
			else merged.push_back(v[j]);
		}
		for (const Ival& iv : merged)
			for (size_t i = 0; i < index.ent.size(); ++i)
				if (index.ent[i].rid == (int32_t)rid && index.ent[i].minpos <= iv.stop && index.ent[i].maxpos >= iv.start)
					blocks.push_back((uint32_t)i);
	}
	// Kept in this order and with repeats: the reference visits the blocks of every merged interval
	// in turn (view.h:409-428), so a block that meets two of them is read -- and its records that
	// pass are written -- twice (e.g. both sides of a linked pair inside one block).
	active_ = true;
	if (blocks.empty()) { error = "Found no blocks overlapping the provided range(s)..."; return false; }
	return true;
}

bool TwoIntervals::filtered_out(const TwoRecord& r) const {
	if (r.ridA >= ivecs_.size()) return true;
	const uint32_t a = r.Apos(), b = r.Bpos();
	uint32_t n_linked = 0, matches_F = 0, matches = 0;
	for (const Ival& iv : ivecs_[r.ridA]) {
		if (!(iv.stop >= a && iv.start <= a)) continue;
		if (iv.mate != 0) continue;                 // the B side of a linked pair
		++matches_F;
		if (iv.mate_rid >= 0) {
			++n_linked;
			const Ival& m = ivecs_[iv.mate_rid][iv.mate_off];
			if (b <= m.stop && b >= m.start && r.ridB == (uint32_t)iv.mate_rid) { ++matches; break; }
		}
	}
	if (n_linked) return matches == 0;
	return matches_F == 0;
}

// ---- ordered block-parallel pipeline ---------------------------------------------------------------
namespace {
// twk1_two_t::PrintLD (core.cpp:520-525): default ostream formatting of doubles is %g, i.e.
// std::to_chars(general, precision 6) -- same digits, no locale, no format parsing.
void append_text(std::string& out, const TwoRecord& r, const Header& hdr) {
	char buf[640];
	char* p = buf; char* const end = buf + sizeof(buf);
	auto put_u = [&](uint32_t v) { p = std::to_chars(p, end, v).ptr; *p++ = '\t'; };
	auto put_d = [&](double v, char sep) { p = std::to_chars(p, end, v, std::chars_format::general, 6).ptr; *p++ = sep; };
	auto put_name = [&](uint32_t rid) {
		if (rid < hdr.contigs.size()) {
			const std::string& n = hdr.contigs[rid].name;
			if (n.size() > 128) { out.append(buf, (size_t)(p - buf)); out.append(n); p = buf; }
			else { std::memcpy(p, n.data(), n.size()); p += n.size(); }
		} else *p++ = '.';
		*p++ = '\t';
	};
	put_u(r.controller); put_name(r.ridA); put_u(r.Apos() + 1); put_name(r.ridB); put_u(r.Bpos() + 1);
	put_d(r.cnt[0], '\t'); put_d(r.cnt[1], '\t'); put_d(r.cnt[2], '\t'); put_d(r.cnt[3], '\t');
	put_d(r.D, '\t'); put_d(r.Dprime, '\t'); put_d(r.R, '\t'); put_d(r.R2, '\t'); put_d(r.P, '\t');
	put_d(r.ChiSqFisher, '\t'); put_d(r.ChiSqModel, '\n');
	out.append(buf, (size_t)(p - buf));
}
}  // namespace

// ---- view (lib/view.h:62-459) -----------------------------------------------------------------------
int two_view(two_view_settings& st) {
	if (st.in.empty()) { std::cerr << stamp("ERROR") << "No input value specified..." << std::endl; return 1; }
	const bool to_stdout = st.out.empty() || st.out == "-";
	if (!to_stdout) std::cerr << stamp("LOG") << "Calling view..." << std::endl;
	TwoReader rd;
	if (!rd.open(st.in)) { std::cerr << "failed to open" << std::endl; return 1; }
	TwoIntervals ivals;
	{
		std::string err;
		if (!ivals.build(st.ivals, rd.hdr, rd.index, err)) {
			if (!err.empty()) std::cerr << stamp("ERROR", "INTERVAL") << err << std::endl;
			return 1;
		}
	}
	rd.hdr.literals += "##tomahawk_viewVersion=" + std::string(TWK_AMD_VERSION) + "\n";
	rd.hdr.literals += "##tomahawk_viewCommand=" + command_line() + "; Date=" + datetime() + "\n";
	if (st.mode != 'u' && st.mode != 'b') { std::cerr << "illegal O" << std::endl; return 1; }

	// Text goes where -o points.  (The reference sends the header there and the records to
	// stdout, writer.h:334-343; with the default -o - the two are the same stream.)
	std::ofstream tfile;
	std::ostream* tos = &std::cout;
	TwoWriter bw;
	const bool sorted_in = rd.index.state == 2;
	const bool use_blocks = sorted_in && !st.ivals.empty();          // view.h:405
	const bool sorted_out = sorted_in;                                // view.h:406, :441-442
	if (st.mode == 'u') {
		if (!to_stdout) {
			tfile.open(st.out, std::ios::binary | std::ios::trunc);
			if (!tfile.good()) { std::cerr << "failed to open" << std::endl; return 1; }
			tos = &tfile;
		}
		static const char* cols = "flags\tridA\tposA\tridB\tposB\tHOMHOM\tHOMALT\tALTHOM\tALTALT\tD\tDprime\tR\tR2\tP\tChiSqFisher\tChiSqModel\n";
		if (st.header_only) { *tos << rd.hdr.literals << cols; tos->flush(); return 0; }       // view.h:380-382
		if (st.write_header) *tos << rd.hdr.literals << cols;                                    // writer.h:243-253
		else *tos << "FLAG\tCHROM_A\tPOS_A\tCHROM_B\tPOS_B\tREF_REF\tREF_ALT\tALT_REF\tALT_ALT\tD\tDPrime\tR\tR2\tP\tChiSqModel\tChiSqTable\n";   // view.h:386-388
	} else {
		if (!bw.open(st.out, rd.hdr, 1)) { std::cerr << "failed to open" << std::endl; return 1; }
		if (sorted_out) bw.set_state(2);
	}

	// blocks to visit
	std::vector<uint32_t> blocks;
	if (use_blocks) blocks = ivals.blocks;
	else { blocks.resize(rd.index.ent.size()); for (size_t i = 0; i < blocks.size(); ++i) blocks[i] = (uint32_t)i; }

	struct Slot { std::string text; std::vector<TwoRecord> keep, recs; std::ifstream in; };
	const bool want_ivals = !ivals.empty();
	const Header& hdr = rd.hdr;
	auto produce = [&](size_t i, Slot& s) -> bool {
		if (!s.in.is_open()) { s.in.open(st.in, std::ios::binary); if (!s.in.good()) return false; }
		std::vector<TwoRecord>& recs = s.recs;
		if (!TwoReader::read_block_at(s.in, rd.index.ent[blocks[i]].foff, recs)) return false;
		s.text.clear(); s.keep.clear();
		if (st.mode == 'u') s.text.reserve(recs.size() * 128);
		for (const TwoRecord& r : recs) {
			if (want_ivals && ivals.filtered_out(r)) continue;
			if (!st.filter.pass(r)) continue;
			if (st.mode == 'u') append_text(s.text, r, hdr); else s.keep.push_back(r);
		}
		return true;
	};
	// -O b: blocks of 10000 records (twk_two_writer_t n_blk_lim, writer.h:164), cut by count only
	std::vector<TwoRecord> pending;
	const uint32_t blk_lim = 10000;
	auto flush_pending = [&](bool all) -> bool {
		size_t off = 0;
		while (pending.size() - off >= blk_lim || (all && pending.size() > off)) {
			const uint32_t n = (uint32_t)std::min<size_t>(blk_lim, pending.size() - off);
			TwoWriter::Packed p;
			if (!TwoWriter::pack_generic(pending.data() + off, n, bw.compression_level(), sorted_out, p) || !bw.write_packed(p)) return false;
			off += n;
		}
		pending.erase(pending.begin(), pending.begin() + (std::ptrdiff_t)off);
		return true;
	};
	auto consume = [&](size_t, Slot& s) -> bool {
		if (st.mode == 'u') { tos->write(s.text.data(), (std::streamsize)s.text.size()); return tos->good(); }
		pending.insert(pending.end(), s.keep.begin(), s.keep.end());
		return flush_pending(false);
	};
	if (!ordered_parallel<Slot>(blocks.size(), st.n_threads, produce, consume)) {
		std::cerr << "failed to get next block" << std::endl;
		return 1;
	}
	if (st.mode == 'u') { tos->flush(); return tos->good() ? 0 : 1; }
	if (!flush_pending(true) || !bw.close()) { std::cerr << stamp("ERROR") << "Failed to write..." << std::endl; return 1; }
	return 0;
}

// ---- sort (lib/two_reader.cpp:168-416) -------------------------------------------------------------
namespace {
inline SortKey key_of(const TwoRecord& r, uint32_t idx) {
	return SortKey{(uint64_t)r.ridA << 32 | r.ridB, (uint64_t)r.Apos() << 32 | r.Bpos(), idx};
}

// Records of index blocks [b0, b1) into `recs` (sized), decoded on T threads.
bool load_range(const std::string& path, const TwoIndex& idx, size_t b0, size_t b1, Raw<TwoRecord>& recs, int T) {
	std::vector<uint64_t> off(b1 - b0 + 1, 0);
	for (size_t b = b0; b < b1; ++b) off[b - b0 + 1] = off[b - b0] + idx.ent[b].n;
	recs.alloc(off.back());
	std::atomic<size_t> next{b0};
	std::atomic<bool> ok{true};
	auto worker = [&]() {
		std::ifstream in(path, std::ios::binary);
		std::vector<TwoRecord> tmp;
		for (;;) {
			const size_t b = next.fetch_add(1);
			if (b >= b1 || !ok) return;
			if (!in.good() || !TwoReader::read_block_at(in, idx.ent[b].foff, tmp) || tmp.size() != idx.ent[b].n) { ok = false; return; }
			if (!tmp.empty()) std::memcpy(&recs[off[b - b0]], tmp.data(), tmp.size() * sizeof(TwoRecord));
		}
	};
	std::vector<std::thread> th;
	for (int t = 0; t < std::max(1, std::min<int>(T, (int)(b1 - b0))); ++t) th.emplace_back(worker);
	for (auto& x : th) x.join();
	return ok;
}

// Sequential reader of one sorted run on disk (raw 106-byte records).
struct RunReader {
	std::ifstream in; std::vector<TwoRecord> buf; size_t pos = 0; uint64_t left = 0; size_t chunk = 0;
	bool open(const std::string& path, uint64_t n, size_t chunk_records) {
		in.open(path, std::ios::binary); left = n; chunk = std::max<size_t>(1, chunk_records); return in.good();
	}
	bool next(TwoRecord& r) {
		if (pos == buf.size()) {
			if (left == 0) return false;
			const size_t n = (size_t)std::min<uint64_t>(left, chunk);
			buf.resize(n); pos = 0;
			in.read((char*)buf.data(), (std::streamsize)(n * sizeof(TwoRecord)));
			if (!in.good()) { left = 0; buf.clear(); return false; }
			left -= n;
		}
		r = buf[pos++];
		return true;
	}
};
}  // namespace

bool two_sort(two_sorter_settings& st) {
	using clock = std::chrono::steady_clock;
	if (st.in.empty()) { std::cerr << stamp("ERROR") << "No input value specified..." << std::endl; return false; }
	TwoReader rd;
	if (!rd.open(st.in)) { std::cerr << stamp("ERROR") << "Failed to open \"" << st.in << "\"..." << std::endl; return false; }
	const TwoIndex& idx = rd.index;
	uint64_t b_unc = 0, n_recs = 0;
	for (const auto& e : idx.ent) { b_unc += e.b_unc; n_recs += e.n; }
	std::cerr << stamp("LOG") << "Blocks: " << pretty(idx.ent.size()) << std::endl;
	std::cerr << stamp("LOG") << "Uncompressed size: " << pretty(b_unc) << " b" << std::endl;
	std::cerr << stamp("LOG") << "Sorting " << pretty(n_recs) << " records..." << std::endl;
	if (b_unc == 0) { std::cerr << stamp("ERROR") << "Cannot sort empty file..." << std::endl; return false; }
	const int T = std::max(1, std::min(st.n_threads, util::usable_cpus()));

	// Output (two_reader.cpp:313-340)
	const bool to_stdout = st.out.empty() || st.out == "-";
	if (to_stdout) std::cerr << stamp("LOG", "WRITER") << "Writing to stdout..." << std::endl;
	else {
		if (extension(st.out) != "two") st.out += ".two";
		std::cerr << stamp("LOG", "WRITER") << "Opening \"" << st.out << "\"..." << std::endl;
	}
	Header hdr = rd.hdr;
	hdr.literals += "\n##tomahawk_sortVersion=" + std::string(TWK_AMD_VERSION) + "\n";
	hdr.literals += "##tomahawk_sortCommand=" + command_line() + "; Date=" + datetime() + "\n";

	// Memory plan: the reference gives every thread `memory_limit` GB (two_sorter_structs.cpp);
	// here that product bounds one in-memory run (records + keys).
	uint64_t run_cap = (uint64_t)((double)st.memory_limit * 1e9 * T / (sizeof(TwoRecord) + sizeof(SortKey)));
	run_cap = std::max<uint64_t>(std::min<uint64_t>(run_cap, 0xFFFFFFF0ull), 1000);      // (a tiny -m forces the external path: tests)
	// runs = consecutive index blocks holding <= run_cap records
	std::vector<std::pair<size_t, size_t>> runs;
	for (size_t b = 0; b < idx.ent.size();) {
		size_t e = b; uint64_t n = 0;
		while (e < idx.ent.size() && (e == b || n + idx.ent[e].n <= run_cap)) n += idx.ent[e++].n;
		runs.emplace_back(b, e);
		b = e;
	}
	std::cerr << stamp("LOG") << "Using " << T << " threads, " << runs.size() << (runs.size() == 1 ? " run (in memory)" : " runs (external merge)") << "..." << std::endl;

	TwoWriter w;
	auto open_writer = [&]() -> bool {
		if (!w.open(st.out, hdr, st.c_level)) { std::cerr << stamp("ERROR") << "Failed top open \"" << st.out << "\"..." << std::endl; return false; }
		w.set_state(2);
		return true;
	};
	// Write sorted records given through `get(i)` for i in [0, n): blocks of <= 10000 records cut at
	// every change of ridA (two_reader.cpp:358-365 + writer.h:320-328), packed on T threads.
	const uint32_t blk_lim = 10000;
	struct PSlot { TwoWriter::Packed p; std::vector<TwoRecord> tmp; };

	const auto t0 = clock::now();
	if (runs.size() == 1) {
		Raw<TwoRecord> recs;
		if (!load_range(st.in, idx, 0, idx.ent.size(), recs, T)) { std::cerr << stamp("ERROR") << "Failed to read input blocks..." << std::endl; return false; }
		std::cerr << stamp("LOG") << "Decoded " << pretty(recs.size()) << " records. " << elapsed_string(std::chrono::duration<double>(clock::now() - t0).count()) << std::endl;
		Raw<SortKey> keys; keys.alloc(recs.size());
		{
			std::vector<std::thread> th;
			for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
				for (size_t i = recs.size() * (size_t)t / T, e = recs.size() * (size_t)(t + 1) / T; i < e; ++i) keys[i] = key_of(recs[i], (uint32_t)i);
			});
			for (auto& x : th) x.join();
		}
		parallel_sort(keys, T);
		std::cerr << stamp("LOG") << "Sorted. " << elapsed_string(std::chrono::duration<double>(clock::now() - t0).count()) << std::endl;
		std::vector<size_t> cut{0};
		for (size_t i = 1; i <= keys.size(); ++i)
			if (i == keys.size() || i - cut.back() == blk_lim || (keys[i].hi >> 32) != (keys[i - 1].hi >> 32)) cut.push_back(i);
		if (!open_writer()) return false;
		const int c_level = st.c_level;
		auto produce = [&](size_t b, PSlot& s) -> bool {
			const size_t n = cut[b + 1] - cut[b];
			s.tmp.resize(n);
			for (size_t i = 0; i < n; ++i) s.tmp[i] = recs[keys[cut[b] + i].idx];
			return TwoWriter::pack_generic(s.tmp.data(), (uint32_t)n, c_level, true, s.p);
		};
		auto consume = [&](size_t, PSlot& s) -> bool { return w.write_packed(s.p); };
		if (!ordered_parallel<PSlot>(cut.size() - 1, T, produce, consume)) { std::cerr << stamp("ERROR") << "Failed to flush block..." << std::endl; return false; }
	} else {
		// external: sorted runs as raw records in temporary files next to the output, then a k-way merge
		const std::string tmp_base = (to_stdout ? std::string("/tmp/twk_sort") : st.out) + "_" + std::to_string((unsigned)getpid());
		std::vector<std::string> tmp_names; std::vector<uint64_t> run_n;
		for (size_t r = 0; r < runs.size(); ++r) {
			Raw<TwoRecord> recs;
			if (!load_range(st.in, idx, runs[r].first, runs[r].second, recs, T)) { std::cerr << stamp("ERROR") << "Failed to read input blocks..." << std::endl; return false; }
			Raw<SortKey> keys; keys.alloc(recs.size());
			for (size_t i = 0; i < recs.size(); ++i) keys[i] = key_of(recs[i], (uint32_t)i);
			parallel_sort(keys, T);
			const std::string name = tmp_base + "_run" + std::to_string(r) + ".tmp";
			std::ofstream o(name, std::ios::binary | std::ios::trunc);
			std::vector<TwoRecord> chunk;
			for (size_t i = 0; i < keys.size(); i += 65536) {
				const size_t n = std::min<size_t>(65536, keys.size() - i);
				chunk.resize(n);
				for (size_t j = 0; j < n; ++j) chunk[j] = recs[keys[i + j].idx];
				o.write((const char*)chunk.data(), (std::streamsize)(n * sizeof(TwoRecord)));
			}
			o.close();
			if (!o.good()) { std::cerr << stamp("ERROR") << "Failed to write \"" << name << "\"..." << std::endl; return false; }
			tmp_names.push_back(name); run_n.push_back(recs.size());
			std::cerr << stamp("LOG", "THREAD") << "Run " << r << ": blocks " << runs[r].first << "->" << runs[r].second << "/" << idx.ent.size() << " and name " << name << std::endl;
		}
		std::vector<RunReader> rr(runs.size());
		const size_t chunk_records = (size_t)std::max<uint64_t>(1024, run_cap / (2 * runs.size()));
		struct QE { SortKey k; TwoRecord rec; };
		auto cmp = [](const QE& a, const QE& b) { return b.k < a.k || (!(a.k < b.k) && b.k.idx < a.k.idx); };   // min-heap, ties by run
		std::priority_queue<QE, std::vector<QE>, decltype(cmp)> q(cmp);
		for (size_t r = 0; r < runs.size(); ++r) {
			if (!rr[r].open(tmp_names[r], run_n[r], chunk_records)) { std::cerr << stamp("ERROR") << "Failed open \"" << tmp_names[r] << "\"..." << std::endl; return false; }
			QE e;
			if (rr[r].next(e.rec)) { e.k = key_of(e.rec, (uint32_t)r); q.push(e); }
		}
		if (!open_writer()) return false;
		// merged stream -> batches of whole blocks -> packed on T threads
		std::vector<TwoRecord> batch; std::vector<size_t> cut{0};
		const int c_level = st.c_level;
		auto flush_batch = [&]() -> bool {
			if (cut.size() < 2) return true;
			auto produce = [&](size_t b, PSlot& s) -> bool { return TwoWriter::pack_generic(batch.data() + cut[b], (uint32_t)(cut[b + 1] - cut[b]), c_level, true, s.p); };
			auto consume = [&](size_t, PSlot& s) -> bool { return w.write_packed(s.p); };
			const bool ok = ordered_parallel<PSlot>(cut.size() - 1, T, produce, consume);
			batch.clear(); cut.assign(1, 0);
			return ok;
		};
		const size_t batch_blocks = (size_t)T * 8;
		while (!q.empty()) {
			QE e = q.top(); q.pop();
			const size_t in_block = batch.size() - cut.back();
			if (in_block && (in_block == blk_lim || batch.back().ridA != e.rec.ridA)) {
				cut.push_back(batch.size());
				if (cut.size() - 1 >= batch_blocks && !flush_batch()) { std::cerr << stamp("ERROR") << "Failed to flush block..." << std::endl; return false; }
			}
			batch.push_back(e.rec);
			const uint32_t r = e.k.idx;
			if (rr[r].next(e.rec)) { e.k = key_of(e.rec, r); q.push(e); }
		}
		if (batch.size() > cut.back()) cut.push_back(batch.size());
		if (!flush_batch()) { std::cerr << stamp("ERROR") << "Failed to flush block..." << std::endl; return false; }
		std::cerr << stamp("LOG") << "Deleting temp files..." << std::endl;
		for (const auto& n : tmp_names) {
			if (std::remove(n.c_str()) != 0) std::cerr << stamp("ERROR") << "Error deleting file " << n << "!" << std::endl;
			else std::cerr << stamp("LOG") << "Deleted " << n << std::endl;
		}
	}
	if (!w.close()) { std::cerr << stamp("ERROR") << "Failed to write..." << std::endl; return false; }
	std::cerr << stamp("LOG") << "Finished merging! Time: " << elapsed_string(std::chrono::duration<double>(clock::now() - t0).count()) << std::endl;
	std::cerr << stamp("LOG") << "Finished!" << std::endl;
	return true;
}

}  // namespace tomahawk
