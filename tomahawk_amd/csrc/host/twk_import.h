// `.twk` producer without htslib: VCF text (plain or gzip/bgzip) or BCF2 -> `.twk`.
//
// Same settings, site filters, counters and output bytes layout as the reference importer
// (include/importer.h:33-58, lib/importer.cpp:25-338, lib/genotype_encoder.h:44-343,
// twk1_t::calculateHardyWeinberg lib/core.cpp:103-201).  The reference reads VCF/BCF through
// htslib; here VCF text is parsed directly, and BCF2 records (BCF2 spec section 6: typed values, the string
// and contig dictionaries of the header, genotypes as (allele + 1) << 1 | phased with vector-end padding)
// are turned into the data line the same site parser reads, so both containers share filters and counters.
#pragma once
#include <cstdint>
#include <string>

namespace tomahawk {

struct twk_vimport_settings {      // include/importer.h:33-42 (same defaults)
	twk_vimport_settings() : remove_univariate(true), flip_major_minor(false), c_level(1), block_size(500),
	                         threshold_miss(0.9f), hwe(0), input("-"), output("-"), n_threads(0) {}
	bool remove_univariate, flip_major_minor;
	uint8_t c_level;
	uint32_t block_size;
	float threshold_miss;
	double hwe;
	std::string input, output;
	int n_threads;                  // parser threads (0: all); not in the reference
};

class twk_variant_importer {       // include/importer.h:49-57
public:
	bool Import(twk_vimport_settings& settings);
	bool Import(void);
	twk_vimport_settings settings;
	// sites dropped per reason, in the reference's order (genotype_encoder.h:25-35) + duplicates
	uint64_t filtered[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	uint64_t n_sites = 0, n_written = 0, n_duplicates = 0;
};

// Exact test of Hardy-Weinberg equilibrium (Wigginton, Cutler & Abecasis 2005) on genotype counts,
// as twk1_t::calculateHardyWeinberg evaluates it (core.cpp:133-200).
double hardy_weinberg_exact(uint64_t obs_hom1, uint64_t obs_hets, uint64_t obs_hom2);

}  // namespace tomahawk
