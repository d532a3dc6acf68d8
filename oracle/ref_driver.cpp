// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
//
// Thin client of the *reference* library (mklarqvist/tomahawk, compiled from
// the sources where they lie under /root/reference by oracle/Makefile into
// oracle/_ref/).  It uses the reference exclusively through its public C++
// API (include/ld.h:40-69 `tomahawk::twk_ld`, include/two_reader.h:235-286
// `tomahawk::two_reader`, include/twk_reader.h `tomahawk::twk_reader`,
// lib/fisher_math.h `kt_fisher_exact`), exactly like the reference's own
// lib/calc.h:237-238 does.  No reference source text is reproduced here.
//
// Sub-commands
//   calc  -i in.twk -o out.two [-p|-u] [-t T] [-r minR2] [-P minP] [-w W]
//         [-c parts -C part] [-k level]      -> twk_ld::Compute (calc.h:56-240)
//   scalc -i in.twk -o out.two -I chr:pos [-w W] [-t T] -> twk_ld::ComputeSingle
//   dump  file.two        -> every record, %.17g doubles, one per line
//   twkinfo file.twk      -> header + per-variant metadata as read by the
//                            reference reader (format parity of our writer)
//   fisher n11 n12 n21 n22 -> kt_fisher_exact left right two (%.17g)
//   view ... / sort ... / concat ... -> the reference's own CLI entry points view() / sort() / concat()
//                            (lib/view.h:62, lib/sort.h:43; header-only, included from
//                            where they lie) with their own option parsing
//   twoinfo file.two      -> index of a .two: state, block entries, per-contig entries
//   hwe hom1 het hom2     -> twk1_t::calculateHardyWeinberg (core.cpp:103-201) on three runs (%.17g)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <getopt.h>

#include "tomahawk.h"
#include "ld.h"
#include "two_reader.h"
#include "twk_reader.h"
#include "fisher_math.h"
#include "view.h"      // int view(int, char**)  -- reference CLI function, lib/view.h:62
#include "sort.h"      // int sort(int, char**)  -- reference CLI function, lib/sort.h:43
#include "concat.h"    // int concat(int, char**) -- reference CLI function, lib/concat.h:63

// The executable owns these globals in the reference too (lib/main.cpp:4-5,
// include/tomahawk.h:30-35); the library reads LITERAL_COMMAND_LINE at
// lib/ld/ld.cpp:611.
int SILENT = 0;
std::string tomahawk::LITERAL_COMMAND_LINE;
std::string tomahawk::INTERPRETED_COMMAND;

// Program banner.  The reference defines it in lib/tomahawk.cpp:13-24, a file
// that only exists to print version strings (it includes htslib/hts.h for
// hts_version()) and is not on the calc path; the one library reference to it
// is from two_reader::Aggregate (off-path, never called here).  An executable
// that links the library without tomahawk.cpp provides its own banner.
namespace tomahawk {
void ProgramMessage(const bool separator) {
	std::cerr << "Program:   tomahawk_ref (reference library " << TOMAHAWK_LIB_VERSION
	          << " behind oracle/ref_driver.cpp)" << std::endl;
	if (separator) std::cerr << "----------" << std::endl;
}
}

static int do_calc(int argc, char** argv, bool single) {
	tomahawk::twk_ld_settings settings;
	int c;
	optind = 1;
	while ((c = getopt(argc, argv, "i:o:t:puP:r:w:I:c:C:k:")) != -1) {
		switch (c) {
		case 'i': settings.in = optarg; break;
		case 'o': settings.out = optarg; break;
		case 't': settings.n_threads = atoi(optarg); break;
		case 'p': settings.force_phased = true; settings.forced_unphased = false; break;
		case 'u': settings.forced_unphased = true; settings.force_phased = false; break;
		case 'P': settings.minP = atof(optarg); break;
		case 'r': settings.minR2 = atof(optarg); break;
		case 'w':
			if (single) settings.l_surrounding = (int32_t)atof(optarg);
			else { settings.window = true; settings.l_window = (int32_t)atof(optarg); }
			break;
		case 'I': settings.ival_strings.push_back(optarg); break;
		case 'c': settings.n_chunks = atoi(optarg); break;
		case 'C': settings.c_chunk = atoi(optarg) - 1; break; // 1-based like calc.h:152-153
		case 'k': settings.c_level = atoi(optarg); break;
		default: fprintf(stderr, "ref_driver: bad option\n"); return 2;
		}
	}
	if (settings.in.empty() || settings.out.empty()) { fprintf(stderr, "ref_driver: need -i and -o\n"); return 2; }
	tomahawk::twk_ld ld;
	if (single) {
		settings.single = true; settings.minR2 = 0; // scalc.h:188-189
		return ld.ComputeSingle(settings, false, false) ? 0 : 1;
	}
	return ld.Compute(settings) ? 0 : 1;
}

static int do_dump(const char* file) {
	tomahawk::two_reader rdr;
	if (!rdr.Open(file)) return 1;
	printf("#n_samples=%zu n_contigs=%zu index_blocks=%llu index_state=%d\n",
	       rdr.hdr.GetNumberSamples(), rdr.hdr.GetNumberContigs(),
	       (unsigned long long)rdr.index.n, (int)rdr.index.state);
	while (rdr.NextRecord()) {
		const tomahawk::twk1_two_t& r = *rdr.it.rcd;
		printf("%u\t%u\t%u\t%u\t%u\t%.17g\t%.17g\t%.17g\t%.17g\t%.17g\t%.17g\t%.17g\t%.17g\t%.17g\t%.17g\t%.17g\n",
		       (unsigned)r.controller, r.ridA, (unsigned)r.Apos, r.ridB, (unsigned)r.Bpos,
		       r.cnt[0], r.cnt[1], r.cnt[2], r.cnt[3],
		       r.D, r.Dprime, r.R, r.R2, r.P, r.ChiSqFisher, r.ChiSqModel);
	}
	return 0;
}

static int do_twkinfo(const char* file) {
	tomahawk::twk_reader rdr;
	if (!rdr.Open(file)) return 1;
	printf("#n_samples=%zu n_contigs=%zu n_blocks=%llu\n", rdr.hdr.GetNumberSamples(),
	       rdr.hdr.GetNumberContigs(), (unsigned long long)rdr.index.n);
	for (uint64_t i = 0; i < rdr.index.n; ++i) {
		const tomahawk::IndexEntry& e = rdr.index.ent[i];
		printf("#block\t%d\t%u\t%u\t%u\t%u\t%u\t%llu\t%llu\n", e.rid, e.n, e.minpos, e.maxpos,
		       e.b_unc, e.b_cmp, (unsigned long long)e.foff, (unsigned long long)e.fend);
	}
	tomahawk::twk1_blk_iterator bit;
	bit.stream = rdr.stream;
	while (bit.NextBlock()) {
		for (uint32_t i = 0; i < bit.blk.n; ++i) {
			const tomahawk::twk1_t& r = bit.blk.rcds[i];
			printf("%u\t%u\t%u\t%u\t%u\t%u\t%d\t%d\t%d\t%.17g\t%u\n", r.rid, r.pos, r.ac, r.an, r.n_het,
			       r.n_hom, (int)r.gt_phase, (int)r.gt_missing, (int)r.gt_ptype, r.hwe, (unsigned)r.gt->n);
		}
	}
	return 0;
}

static int do_twoinfo(const char* file) {
	tomahawk::two_reader rdr;
	if (!rdr.Open(file)) return 1;
	printf("#state=%d n=%llu m_ent=%llu\n", (int)rdr.index.state, (unsigned long long)rdr.index.n,
	       (unsigned long long)rdr.index.m_ent);
	for (uint64_t i = 0; i < rdr.index.n; ++i) {
		const tomahawk::IndexEntryOutput& e = rdr.index.ent[i];
		printf("#block\t%d\t%d\t%u\t%u\t%u\t%u\n", e.rid, e.ridB, e.n, e.minpos, e.maxpos, e.b_unc);
	}
	for (uint64_t i = 0; i < rdr.index.m_ent; ++i) {
		const tomahawk::IndexEntryEntry& e = rdr.index.ent_meta[i];
		printf("#contig\t%d\t%u\t%u\t%u\t%llu\n", e.rid, e.n, e.minpos, e.maxpos, (unsigned long long)e.nn);
	}
	return 0;
}

static int do_hwe(uint32_t hom1, uint32_t het, uint32_t hom2) {
	tomahawk::twk1_t rec;
	tomahawk::twk1_igt_t<uint32_t>* gt = new tomahawk::twk1_igt_t<uint32_t>;
	gt->data = new uint32_t[3];
	gt->n = 3; gt->miss = 0;
	gt->data[0] = hom1 << 2 | 0; gt->data[1] = het << 2 | 1; gt->data[2] = hom2 << 2 | 3;   // RLE words: len << 2 | (a << 1 | b)
	rec.gt = gt;
	rec.calculateHardyWeinberg();
	printf("%.17g\n", rec.hwe);
	return 0;
}

int main(int argc, char** argv) {
	if (argc < 2) { fprintf(stderr, "usage: tomahawk_ref calc|scalc|dump|twkinfo|fisher ...\n"); return 2; }
	tomahawk::LITERAL_COMMAND_LINE = tomahawk::TOMAHAWK_PROGRAM_NAME;
	for (int i = 1; i < argc; ++i) tomahawk::LITERAL_COMMAND_LINE += " " + std::string(argv[i]);
	const std::string cmd = argv[1];
	if (cmd == "calc")  return do_calc(argc - 1, argv + 1, false);
	if (cmd == "scalc") return do_calc(argc - 1, argv + 1, true);
	if (cmd == "dump" && argc == 3) return do_dump(argv[2]);
	if (cmd == "twkinfo" && argc == 3) return do_twkinfo(argv[2]);
	if (cmd == "twoinfo" && argc == 3) return do_twoinfo(argv[2]);
	if (cmd == "hwe" && argc == 5) return do_hwe((uint32_t)atoi(argv[2]), (uint32_t)atoi(argv[3]), (uint32_t)atoi(argv[4]));
	if (cmd == "view") { optind = 1; return view(argc - 1, argv + 1); }
	if (cmd == "sort") { optind = 1; return sort(argc - 1, argv + 1); }
	if (cmd == "concat") { optind = 1; return concat(argc - 1, argv + 1); }
	if (cmd == "fisher" && argc == 6) {
		double l, r, t;
		kt_fisher_exact(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), &l, &r, &t);
		printf("%.17g\t%.17g\t%.17g\n", l, r, t);
		return 0;
	}
	fprintf(stderr, "tomahawk_ref: unknown command\n");
	return 2;
}
