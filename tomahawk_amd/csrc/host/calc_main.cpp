// `tomahawk calc` on the MI355X engine: same flags, defaults, messages-on-stderr
// and exit codes as the reference CLI (lib/main.cpp:19-93, lib/calc.h:28-240).
#include <cstdlib>
#include <cstring>
#include <getopt.h>
#include <iostream>
#include <regex>
#include <string>

#include "twk_ld.h"
#include "twk_format.h"
#include "twk_hip.h"

namespace tomahawk { std::string LITERAL_COMMAND_LINE; }

static void program_message() {
	std::cerr << "Program:   tomahawk-mi355x (pairwise LD on AMD MI355X; `tomahawk calc` compatible)\n"
	          << "Libraries: tomahawk_amd; ZSTD-" << tomahawk::zstd_version() << "; twk_hip ABI " << twk_hip_abi_version() << "\n"
	          << "----------" << std::endl;
}

static void calc_usage() {
	program_message();
	std::cerr <<
	"About:  Calculate linkage disequilibrium\n"
	"        Force phased -p or unphased -u for faster calculations if\n"
	"        all variant sites are guaranteed to have the given phasing.\n\n"
	"Usage:  tomahawk calc [options] -i <in.twk> -o <output.two>\n\n"
	"Options:\n"
	"  -i FILE   input Tomahawk (required)\n"
	"  -o FILE   output file or file prefix (required)\n"
	"  -t INT    number of CPU threads used to unpack the input (default: maximum available)\n"
	"  -c INT    number of subproblems to split compute into (must be in (c!2 + c))\n"
	"  -C INT    chosen part to compute (0 < -C < -c)\n"
	"  -m, -M    accepted for compatibility (CPU low-memory modes; no effect on the GPU engine)\n"
	"  -b        number of records in a block (accepted; unused by calc, as in the reference)\n"
	"  -w INT    sliding window width in bases\n"
	"  -I STRING filter interval <contig>:pos-pos (see manual)\n"
	"  -p        force computations to use phased math\n"
	"  -u        force computations to use unphased math\n"
	"  -P FLOAT  Fisher's exact test / Chi-squared cutoff P-value (default: 1)\n"
	"  -r FLOAT  Pearson's R-squared minimum cut-off value (default: 0.1)\n"
	"  -k INT    compression level to use (default: 1, max = 22).\n"
	"Environment: TWK_HIP_DEVICE=<n> selects the GPU (default 0).\n" << std::endl;
}

static std::string stamp(const char* t) { return std::string("[") + t + "] "; }

static int calc(int argc, char** argv) {
	if (argc < 3) { calc_usage(); return 1; }
	static struct option long_options[] = {
		{"input", required_argument, 0, 'i'}, {"threads", optional_argument, 0, 't'}, {"output", required_argument, 0, 'o'},
		{"interval", optional_argument, 0, 'I'}, {"parts", optional_argument, 0, 'c'}, {"partStart", optional_argument, 0, 'C'},
		{"low-memory", optional_argument, 0, 'm'}, {"block-size", optional_argument, 0, 'b'}, {"bitmaps", optional_argument, 0, 'M'},
		{"compression-level", optional_argument, 0, 'k'}, {"cross-chr-only", no_argument, 0, 'X'}, {"no-cross-chr", no_argument, 0, 'x'},
		{"minP", optional_argument, 0, 'P'}, {"force-phased", no_argument, 0, 'p'}, {"force-unphased", no_argument, 0, 'u'},
		{"samples", optional_argument, 0, 'S'}, {"minR2", optional_argument, 0, 'r'}, {"detailedProgress", no_argument, 0, 'd'},
		{"silent", no_argument, 0, 's'}, {"windowBases", optional_argument, 0, 'w'}, {0, 0, 0, 0}};
	tomahawk::twk_ld_settings settings;
	int c, option_index = 0;
	while ((c = getopt_long(argc, argv, "i:o:t:puP:a:A:r:w:S:I:sdc:C:mMb:xXk:?", long_options, &option_index)) != -1) {
		switch (c) {
		case 'i': settings.in = optarg; break;
		case 'o': settings.out = optarg; break;
		case 'I': settings.ival_strings.push_back(optarg); break;
		case 'm': settings.low_memory = true; break;
		case 'p': settings.force_phased = true; settings.forced_unphased = false; break;
		case 'u': settings.forced_unphased = true; settings.force_phased = false; break;
		case 'M': settings.force_phased = true; settings.low_memory = true; settings.bitmaps = true; break;
		case 't':
			settings.n_threads = atoi(optarg);
			if (settings.n_threads <= 0) { std::cerr << stamp("ERROR") << "Cannot have a non-positive number of worker threads" << std::endl; return 1; }
			break;
		case 'b':
			settings.bl_size = atoi(optarg);
			if (settings.bl_size <= 0) { std::cerr << stamp("ERROR") << "Cannot have a non-positive number of entries in a block!" << std::endl; return 1; }
			break;
		case 'c':
			settings.n_chunks = atoi(optarg);
			if (settings.n_chunks <= 0) { std::cerr << stamp("ERROR") << "Cannot have a negative or zero amount of partitions" << std::endl; return 1; }
			break;
		case 'C':
			settings.c_chunk = atoi(optarg) - 1;   // 1-based on the command line (calc.h:152-153)
			if (settings.c_chunk < 0) { std::cerr << stamp("ERROR") << "Cannot have a non-positive start partition" << std::endl; return 1; }
			break;
		case 'r':
			settings.minR2 = atof(optarg);
			if (settings.minR2 < 0) { std::cerr << stamp("ERROR") << "Cannot have a negative minimum R-squared value" << std::endl; return 1; }
			if (settings.minR2 > 1) { std::cerr << stamp("ERROR") << "Cannot have minimum R-squared value > 1" << std::endl; return 1; }
			break;
		case 'P':
			settings.minP = atof(optarg);
			if (settings.minP < 0) { std::cerr << stamp("ERROR") << "Cannot have a negative cutoff P-value" << std::endl; return 1; }
			if (settings.minP > 1) { std::cerr << stamp("ERROR") << "Cannot have a cutoff P-value > 1" << std::endl; return 1; }
			break;
		case 'w': {
			settings.window = true;
			const std::string a(optarg);
			if (!std::regex_match(a, std::regex("^(([0-9]+)|([0-9]+[eE]{1}[0-9]+))$"))) { std::cerr << "not an integer" << std::endl; return 1; }
			settings.l_window = std::regex_match(a, std::regex("^[0-9]+$")) ? atoi(optarg) : (int32_t)atof(optarg);
			if (settings.l_window <= 0) { std::cerr << stamp("ERROR") << "Cannot have a non-positive window size" << std::endl; return 1; }
			break;
		}
		case 'k': settings.c_level = atoi(optarg); break;
		default:
			std::cerr << stamp("ERROR") << "Unrecognized option: " << (char)c << std::endl;
			return 1;
		}
	}
	if (settings.in.empty()) { std::cerr << stamp("ERROR") << "No input value specified..." << std::endl; return 1; }
	if (settings.out.empty()) { std::cerr << stamp("ERROR") << "No output value specified..." << std::endl; return 1; }
	program_message();
	std::cerr << stamp("LOG") << "Calling calc..." << std::endl;
	tomahawk::twk_ld ld;
	return ld.Compute(settings) ? 0 : 1;
}

// `tomahawk scalc` (lib/scalc.h:50-194): one site against its neighbourhood.
static int scalc(int argc, char** argv) {
	if (argc < 3) {
		program_message();
		std::cerr << "About:  Calculate linkage disequilibrium for a single site\n"
		             "Usage:  tomahawk scalc [options] -i <in.twk> -I <chr:pos> -o <output.two>\n\n"
		             "Options:\n"
		             "  -i FILE   input Tomahawk (required)\n  -o FILE   output file or file prefix (required)\n"
		             "  -I STRING target site <contig>:pos (required)\n  -w INT    flanking width in bases (default: 500000)\n"
		             "  -t INT    number of CPU threads\n  -P FLOAT  Fisher's exact test cutoff P-value (default: 1)\n"
		             "  -k INT    compression level to use (default: 1)\n" << std::endl;
		return 1;
	}
	tomahawk::twk_ld_settings settings;
	int c;
	while ((c = getopt(argc, argv, "i:o:t:I:mMb:r:R:P:k:w:?")) != -1) {
		switch (c) {
		case 'i': settings.in = optarg; break;
		case 'o': settings.out = optarg; break;
		case 'I': settings.ival_strings.push_back(optarg); break;
		case 'm': settings.low_memory = true; break;
		case 'M': settings.force_phased = true; settings.low_memory = true; settings.bitmaps = true; break;
		case 't':
			settings.n_threads = atoi(optarg);
			if (settings.n_threads <= 0) { std::cerr << stamp("ERROR") << "Cannot have a non-positive number of worker threads" << std::endl; return 1; }
			break;
		case 'b':
			settings.bl_size = atoi(optarg);
			if (settings.bl_size <= 0) { std::cerr << stamp("ERROR") << "Cannot have a non-positive number of entries in a block!" << std::endl; return 1; }
			break;
		case 'r':
			settings.minR2 = atof(optarg);
			if (settings.minR2 < 0 || settings.minR2 > 1) { std::cerr << stamp("ERROR") << "Cannot have a minimum R-squared value outside [0,1]" << std::endl; return 1; }
			break;
		case 'R':
			settings.maxR2 = atof(optarg);
			if (settings.maxR2 < 0 || settings.maxR2 > 1) { std::cerr << stamp("ERROR") << "Cannot have a maximum R-squared value outside [0,1]" << std::endl; return 1; }
			break;
		case 'P':
			settings.minP = atof(optarg);
			if (settings.minP < 0 || settings.minP > 1) { std::cerr << stamp("ERROR") << "Cannot have a cutoff P-value outside [0,1]" << std::endl; return 1; }
			break;
		case 'k': settings.c_level = atoi(optarg); break;
		case 'w':
			settings.l_surrounding = atoi(optarg);
			if (settings.l_surrounding < 1) { std::cerr << stamp("ERROR") << "Cannot have a non-positive window size" << std::endl; return 1; }
			break;
		default:
			std::cerr << stamp("ERROR") << "Unrecognized option: " << (char)c << std::endl;
			return 1;
		}
	}
	if (settings.in.empty()) { std::cerr << stamp("ERROR") << "No input value specified..." << std::endl; return 1; }
	if (settings.out.empty()) { std::cerr << stamp("ERROR") << "No output value specified..." << std::endl; return 1; }
	program_message();
	std::cerr << stamp("LOG") << "Calling calc..." << std::endl;
	settings.single = true;
	settings.minR2 = 0;           // scalc.h:188-189: -r is parsed, then overwritten
	tomahawk::twk_ld ld;
	return ld.ComputeSingle(settings, true, true) ? 0 : 1;
}

int main(int argc, char** argv) {
	if (argc == 1) { program_message(); std::cerr << "Usage: tomahawk calc [options] -i <in.twk> -o <output.two>" << std::endl; return 1; }
	tomahawk::LITERAL_COMMAND_LINE = "tomahawk";
	for (int i = 1; i < argc; ++i) tomahawk::LITERAL_COMMAND_LINE += " " + std::string(argv[i]);
	if (strcmp(argv[1], "calc") == 0) return calc(argc, argv);
	if (strcmp(argv[1], "calc-single") == 0 || strcmp(argv[1], "scalc") == 0) return scalc(argc, argv);
	if (strcmp(argv[1], "--version") == 0 || strcmp(argv[1], "version") == 0) { program_message(); return 0; }
	if (strcmp(argv[1], "--help") == 0 || strcmp(argv[1], "help") == 0) { calc_usage(); return 0; }
	program_message();
	std::cerr << stamp("ERROR") << "Illegal command: only `calc` and `scalc` are provided by the MI355X engine (view/sort/concat/... are the reference's)" << std::endl;
	return 1;
}
