// Downstream tools of the calc path: `view` and `sort` of `.two` files.
//
// Same flags, same record filters and interval semantics, same output bytes as the reference
// (lib/view.h:62-459, include/two_reader.h:39-206, lib/intervals.cpp:142-405,
// lib/two_reader.cpp:168-416, include/writer.h:163-406).  Host code: both tools are bound by
// zstd and text formatting, which run block-parallel on worker threads here.
#pragma once
#include <cstdint>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "twk_format.h"

namespace tomahawk {

// twk_two_filter (include/two_reader.h:39-206): a conjunction of the range tests that were set.
class TwoFilter {
public:
	uint32_t filter_vec = 0;
	double minR2 = 0, maxR2 = 100, minR = -100, maxR = 100, minD = -100, maxD = 100;
	double minDprime = 0, maxDprime = 100, minP = 0, maxP = 1;
	double hA_min = 0, hA_max = 999999999, hB_min = 0, hB_max = 999999999;
	double hC_min = 0, hC_max = 999999999, hD_min = 0, hD_max = 999999999;
	double mhc_min = 0, mhc_max = 999999999;
	double minChi = 0, maxChi = std::numeric_limits<double>::max();
	double minChiModel = 0, maxChiModel = std::numeric_limits<double>::max();
	uint32_t flag_include = std::numeric_limits<uint32_t>::max(), flag_exclude = 0;

	enum Bit { R2 = 0, D, DPRIME, P, HAPA, HAPB, HAPC, HAPD, R, UPPER, LOWER, MHC, FLAGS, CHI, CHIMODEL };
	void set(Bit b) { filter_vec |= 1u << b; }
	// true: the record passes (twk_two_filter::Filter, two_reader.h:144-151)
	bool pass(const TwoRecord& r) const;
};

// twk_intervals_two (include/intervals.h:86-170, lib/intervals.cpp:142-405): plain intervals
// `contig`, `contig:pos`, `contig:from-to` on the A side, or linked pairs `ivalA,ivalB`.
class TwoIntervals {
public:
	struct Ival { uint32_t start, stop; int32_t mate_rid; uint32_t mate_off; uint8_t mate; };
	// Parse, merge and map to index blocks.  false + message on the reference's error paths.
	bool build(const std::vector<std::string>& strings, const Header& hdr, const TwoIndex& index, std::string& error);
	// true: drop the record (twk_intervals_two::FilterInterval, intervals.cpp:381-403)
	bool filtered_out(const TwoRecord& r) const;
	bool empty() const { return !active_; }
	std::vector<uint32_t> blocks;                 // index entries that overlap: per contig, per merged interval (repeats kept)
private:
	bool parse(const std::string& s, const Header& hdr, std::string& error);
	std::vector<std::vector<Ival>> ivecs_;        // per contig, in the order given
	bool active_ = false;
};

struct two_view_settings {
	std::string in, out;
	char mode = 'u';                              // 'u': text, 'b': .two  (writer.mode, view.h:137-144)
	bool write_header = true, header_only = false;
	int n_threads = (int)std::thread::hardware_concurrency();
	std::vector<std::string> ivals;
	TwoFilter filter;
};
// Exit code of `tomahawk view` (0 ok, 1 error).
int two_view(two_view_settings& settings);

struct two_sorter_settings {                      // include/two_reader.h:224-230
	std::string in, out;
	float memory_limit = 0.5f;                    // GB per thread
	int c_level = 1, n_threads = (int)std::thread::hardware_concurrency();
};
// two_reader::Sort (lib/two_reader.cpp:168-416): order (ridA, ridB, Apos, Bpos) (core.cpp:458-468),
// blocks of <= 10000 records cut at every change of ridA, sorted index with per-contig entries.
bool two_sort(two_sorter_settings& settings);

}  // namespace tomahawk
