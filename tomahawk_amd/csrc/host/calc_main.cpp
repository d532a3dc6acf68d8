// `tomahawk calc` on the MI355X engine: same flags, defaults, messages-on-stderr
// and exit codes as the reference CLI (lib/main.cpp:19-93, lib/calc.h:28-240).
#include <cstdlib>
#include <cstring>
#include <getopt.h>
#include <iostream>
#include <regex>
#include <string>
#include <vector>
#include <fstream>
#include <ctime>
#include <unistd.h>
#include <sys/wait.h>
#include <malloc.h>

#include "twk_ld.h"
#include "twk_format.h"
#include "twk_hip.h"
#include "twk_two_tools.h"
#include "twk_import.h"

namespace tomahawk { std::string LITERAL_COMMAND_LINE; }

static void program_message() {
	std::cerr << "Program:   tomahawk-mi355x (pairwise LD on AMD MI355X; `tomahawk calc` compatible)\n"
	          << "Libraries: tomahawk_amd; ZSTD-" << tomahawk::zstd_version() << "; twk_hip ABI " << twk_hip_abi_version() << "\n"
	          << "----------" << std::endl;
}

static std::vector<std::string> g_argv;     // argv as given (getopt_long permutes the live one)

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
	"  --engine-option KEY=INT  a switch of the GPU engine (twk_hip_set_option, include/twk_hip.h; repeatable)\n"
	"Environment: TWK_HIP_DEVICE=<n> selects the GPU (default 0); TWK_HIP_GPUS=<n> uses GPUs 0..n-1, one\n"
	"             driver thread each (equal-area row bands, one shared output file); TWK_HIP_PART=k/n makes\n"
	"             this process compute share k of n of a multi-node run (merge the outputs with concat).\n" << std::endl;
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
		{"silent", no_argument, 0, 's'}, {"windowBases", optional_argument, 0, 'w'},
		{"engine-option", required_argument, 0, 1000}, {0, 0, 0, 0}};
	tomahawk::twk_ld_settings settings;
	int c, option_index = 0;
	std::vector<std::pair<std::string, long long>> engine_options;
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
		case 1000: {      // --engine-option key=value (not in the reference): twk_ld::SetEngineOption
			const std::string a(optarg);
			const size_t eq = a.find('=');
			if (eq == std::string::npos || eq == 0 || eq + 1 >= a.size()) { std::cerr << stamp("ERROR") << "--engine-option wants key=value" << std::endl; return 1; }
			engine_options.emplace_back(a.substr(0, eq), atoll(a.c_str() + eq + 1));
			break;
		}
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
	for (const auto& kv : engine_options) ld.SetEngineOption(kv.first, kv.second);
	return ld.Compute(settings) ? 0 : 1;
}

// `tomahawk concat` (lib/concat.h:63-251): copy the compressed blocks of several .two files into one.
static int concat(int argc, char** argv) {
	if (argc < 3) {
		program_message();
		std::cerr << "About:  Concatenate two or more TWO files\n\n"
		             "Usage:  tomahawk concat [options] -i <in.two> -i <in.two> -o <out.two>\n\n"
		             "Options:\n  -i FILE    input TWO file specified 1-or-more times (required)\n"
		             "  -I STRING  input file list (required)\n  -o FILE    output file (- for stdout; default: -)\n" << std::endl;
		return 0;
	}
	std::vector<std::string> in_list, lists;
	std::string out = "-";
	int c;
	while ((c = getopt(argc, argv, "i:I:o:?")) != -1) {
		switch (c) {
		case 'i': in_list.push_back(optarg); break;
		case 'I': lists.push_back(optarg); break;
		case 'o': out = optarg; break;
		default: fprintf(stderr, "%s: option `-%c' is invalid: ignored\n", argv[0], optopt); break;
		}
	}
	for (const auto& l : lists) {
		std::ifstream f(l);
		if (!f.good()) { std::cerr << "faield to open list=" << l << std::endl; return 1; }
		std::string line;
		while (std::getline(f, line)) if (!line.empty()) in_list.push_back(line);
	}
	if (in_list.empty()) { std::cerr << stamp("ERROR") << "No input value specified..." << std::endl; return 1; }
	if (in_list.size() == 1) { std::cerr << stamp("ERROR") << "Only one input file provided..." << std::endl; return 1; }
	if (out != "-") {      // extension forced to .two like calc (concat.h:163-171)
		const size_t sl = out.find_last_of("/\\"), dot = out.rfind('.');
		const std::string ext = (dot == std::string::npos || (sl != std::string::npos && dot < sl)) ? "" : out.substr(dot + 1);
		if (!(ext.size() == 3 && strncasecmp(ext.c_str(), "two", 3) == 0)) out += ".two";
	}
	char date[64]; time_t t = time(nullptr); struct tm now; localtime_r(&t, &now);
	strftime(date, sizeof(date), "%Y-%m-%d %H:%M:%S", &now);
	const std::string note = "\n##tomahawk_concatVersion=mi355x\n##tomahawk_concatCommand=" + tomahawk::LITERAL_COMMAND_LINE + "; Date=" + date;
	std::string err;
	std::cerr << stamp("LOG") << "Concatenating " << in_list.size() << " files into " << out << "..." << std::endl;
	if (!tomahawk::two_concat(in_list, out, note, err)) { std::cerr << stamp("ERROR") << err << std::endl; return 1; }
	return 0;
}

// `tomahawk view` (lib/view.h:28-459): same flags; -h prints the header only and -H drops it,
// as the reference's option switch has them (view.h:367-368).
static void view_usage() {
	program_message();
	std::cerr <<
	"About:  Convert binary TWO->LD/TWO, subset and slice TWO data\n\n"
	"Usage:  tomahawk view [options] -i <in.two>\n\n"
	"Options:\n"
	"  -i FILE   input TWO file (required)\n"
	"  -h/H      (twk/two) header only / no header\n"
	"  -I STRING filter interval <contig>:pos-pos (TWK/TWO) or linked interval <contig>:pos-pos,<contig>:pos-pos\n\n"
	"  -o FILE    output file (- for stdout; default: -)\n"
	"  -O <b|u>   b: compressed TWO, u: uncompressed LD\n\n"
	"Filter parameters:\n"
	"  -r,-R --minR2,--maxR2   FLOAT   Pearson's R-squared min/max cut-off value\n"
	"  -z,-Z --minR,--maxR     FLOAT   Pearson's R min/max cut-off value\n"
	"  -p,-P --minP,--maxP     FLOAT   Min/max P-value (default: [0,1])\n"
	"  -d,-D --minD,--maxD     FLOAT   Min/max D value (default: [-1,1])\n"
	"  -b,-B --minDP,--maxDP   FLOAT   Min/max D' value (default: [0,1])\n"
	"  -1,-5 --minP1,--maxP1   FLOAT   Min/max REF_REF count (default: [0,inf])\n"
	"  -2,-6 --minP2,--maxP2   FLOAT   Min/max REF_ALT count (default: [0,inf])\n"
	"  -3,-7 --minQ1,--maxQ1   FLOAT   Min/max ALT_REF count (default: [0,inf])\n"
	"  -4,-8 --minQ2,--maxQ1   FLOAT   Min/max ALT_ALT count (default: [0,inf])\n"
	"  -a,-A --minMHC,--maxMHC FLOAT   Min/max number of non-major haplotype count (default: [0,inf])\n"
	"  -x,-X --minChi,--maxChi FLOAT   Min/max Chi-squared CV of contingency table (default: [0,inf])\n"
	"  -m,-M --minMCV,--maxMCV FLOAT   Min/max Chi-squared CV of unphased model (default: [0,inf])\n"
	"  -f  INT  include FLAG value\n"
	"  -F  INT  exclude FLAG value\n"
	"  -u       output only the upper triangular values\n"
	"  -l       output only the lower triangular values\n"
	"  -t INT   number of worker threads (default: all; not in the reference)\n";
}

static int view(int argc, char** argv) {
	if (argc < 3) { view_usage(); return 0; }
	static struct option long_options[] = {
		{"input", required_argument, 0, 'i'}, {"output", optional_argument, 0, 'o'}, {"output-type", optional_argument, 0, 'O'},
		{"minP", optional_argument, 0, 'p'}, {"maxP", optional_argument, 0, 'P'}, {"minR", optional_argument, 0, 'z'},
		{"maxR", optional_argument, 0, 'Z'}, {"minR2", optional_argument, 0, 'r'}, {"maxR2", optional_argument, 0, 'R'},
		{"minDP", optional_argument, 0, 'b'}, {"maxDP", optional_argument, 0, 'B'}, {"minD", optional_argument, 0, 'd'},
		{"maxD", optional_argument, 0, 'D'}, {"minP1", optional_argument, 0, '1'}, {"minP2", optional_argument, 0, '2'},
		{"minQ1", optional_argument, 0, '3'}, {"minQ2", optional_argument, 0, '4'}, {"maxP1", optional_argument, 0, '5'},
		{"maxP2", optional_argument, 0, '6'}, {"maxQ1", optional_argument, 0, '7'}, {"maxQ2", optional_argument, 0, '8'},
		{"minMHC", optional_argument, 0, 'a'}, {"maxMHC", optional_argument, 0, 'A'}, {"minChi", optional_argument, 0, 'x'},
		{"maxChi", optional_argument, 0, 'X'}, {"minMCV", optional_argument, 0, 'm'}, {"maxMCV", optional_argument, 0, 'M'},
		{"flagInclude", optional_argument, 0, 'f'}, {"flagExclude", optional_argument, 0, 'F'},
		{"upperTriangular", no_argument, 0, 'u'}, {"lowerTriangular", no_argument, 0, 'l'},
		{"headerOnly", no_argument, 0, 'H'}, {"noHeader", no_argument, 0, 'h'}, {"interval", optional_argument, 0, 'I'},
		{"threads", required_argument, 0, 't'}, {0, 0, 0, 0}};
	static const std::regex re_float("^[-+]?[0-9]*\\.?[0-9]+([eE][-+]?[0-9]+)?$"), re_number("^[0-9]+$");   // tomahawk.h:61-62
	tomahawk::two_view_settings st;
	tomahawk::TwoFilter& f = st.filter;
	using F = tomahawk::TwoFilter;
	int c = 0, long_index = 0;
	while ((c = getopt_long(argc, argv, "i:HhI:o:O:r:R:z:Z:p:P:d:D:b:B:1:2:3:4:5:6:7:8:x:X:a:A:m:M:f:F:ult:", long_options, &long_index)) != -1) {
		double* dst = nullptr; F::Bit bit = F::R2;
		switch (c) {
		case ':': fprintf(stderr, "%s: option `-%c' requires an argument\n", argv[0], optopt); continue;
		case '?': default: fprintf(stderr, "%s: option `-%c' is invalid: ignored\n", argv[0], optopt); continue;
		case 'i': st.in = optarg; continue;
		case 'o': st.out = optarg; continue;
		case 'O': if (std::string(optarg).size() != 1) { std::cerr << "illegal O" << std::endl; return 1; } st.mode = optarg[0]; continue;
		case 'u': f.set(F::UPPER); continue;
		case 'l': f.set(F::LOWER); continue;
		case 'h': st.header_only = true; continue;
		case 'H': st.write_header = false; continue;
		case 'I': st.ivals.push_back(optarg); continue;
		case 't': st.n_threads = atoi(optarg); continue;
		case 'f': case 'F':
			if (!std::regex_match(std::string(optarg), re_number)) { std::cerr << "not a valid number" << std::endl; return 1; }
			(c == 'f' ? f.flag_include : f.flag_exclude) = (uint32_t)atof(optarg); f.set(F::FLAGS); continue;
		case 'p': dst = &f.minP; bit = F::P; break;          case 'P': dst = &f.maxP; bit = F::P; break;
		case 'z': dst = &f.minR; bit = F::R; break;          case 'Z': dst = &f.maxR; bit = F::R; break;
		case 'r': dst = &f.minR2; bit = F::R2; break;        case 'R': dst = &f.maxR2; bit = F::R2; break;
		case 'b': dst = &f.minDprime; bit = F::DPRIME; break; case 'B': dst = &f.maxDprime; bit = F::DPRIME; break;
		case 'd': dst = &f.minD; bit = F::D; break;          case 'D': dst = &f.maxD; bit = F::D; break;
		case '1': dst = &f.hA_min; bit = F::HAPA; break;     case '5': dst = &f.hA_max; bit = F::HAPA; break;
		case '2': dst = &f.hB_min; bit = F::HAPB; break;     case '6': dst = &f.hB_max; bit = F::HAPB; break;
		case '3': dst = &f.hC_min; bit = F::HAPC; break;     case '7': dst = &f.hC_max; bit = F::HAPC; break;
		case '4': dst = &f.hD_min; bit = F::HAPD; break;     case '8': dst = &f.hD_max; bit = F::HAPD; break;
		case 'a': dst = &f.mhc_min; bit = F::MHC; break;     case 'A': dst = &f.mhc_max; bit = F::MHC; break;
		case 'x': dst = &f.minChi; bit = F::CHI; break;      case 'X': dst = &f.maxChi; bit = F::CHI; break;
		case 'm': dst = &f.minChiModel; bit = F::CHIMODEL; break; case 'M': dst = &f.maxChiModel; bit = F::CHIMODEL; break;
		}
		if (!std::regex_match(std::string(optarg), re_float)) { std::cerr << "not a valid float" << std::endl; return 1; }
		*dst = atof(optarg); f.set(bit);
	}
	if (st.in.empty()) { std::cerr << stamp("ERROR") << "No input value specified..." << std::endl; return 1; }
	if (!(st.out.empty() || st.out == "-")) program_message();
	if (st.n_threads <= 0) st.n_threads = 1;
	return tomahawk::two_view(st);
}

// `tomahawk import` (lib/import.h:28-131): VCF text (plain / gzip) -> .twk
static int import_cmd(int argc, char** argv) {
	if (argc < 3) {
		program_message();
		std::cerr << "About:  Convert VCF->TWK\n\nUsage:  tomahawk import [options] -i <in.vcf[.gz]> -o <out.twk>\n\nOptions:\n"
		             "  -i FILE  input VCF (plain or gzip/bgzip text) or BCF2, '-' for stdin (required)\n  -o FILE  output file prefix (required)\n"
		             "  -n FLOAT missingness fraction in range [0,1] (default: 0.9)\n  -H FLOAT Hardy-Weinberg P-value cutoff (default: 0)\n"
		             "  -r       do not filter out univariate sites\n  -f       flip reference and alternative alleles when the major allele is the alternative (no effect, as in the reference)\n"
		             "  -b INT   number of variants per block (default: 500)\n  -L INT   compression level 1-20 (default: 1)\n"
		             "  -t INT   parser threads (default: all; not in the reference)\n\n";
		return 1;
	}
	static struct option long_options[] = {{"input", required_argument, 0, 'i'}, {"output", optional_argument, 0, 'o'},
		{"filter-univariate", optional_argument, 0, 'r'}, {"flip", optional_argument, 0, 'f'}, {"missingness", optional_argument, 0, 'n'},
		{"compression-level", optional_argument, 0, 'L'}, {"block-size", optional_argument, 0, 'b'}, {"hwe", optional_argument, 0, 'H'},
		{"threads", required_argument, 0, 't'}, {0, 0, 0, 0}};
	tomahawk::twk_vimport_settings st;
	int c = 0, option_index = 0;
	while ((c = getopt_long(argc, argv, "i:o:rfn:b:L:H:t:?", long_options, &option_index)) != -1) {
		switch (c) {
		case 'i': st.input = optarg; break;
		case 'o': st.output = optarg; break;
		case 'n':
			st.threshold_miss = (float)atof(optarg);
			if (st.threshold_miss < 0) { std::cerr << stamp("ERROR") << "Cannot set missingness filter to < 0..." << std::endl; return 1; }
			if (st.threshold_miss > 1) { std::cerr << stamp("ERROR") << "Cannot set missingness filter to > 1..." << std::endl; return 1; }
			break;
		case 'H':
			st.hwe = atof(optarg);
			if (st.hwe < 0) { std::cerr << stamp("ERROR") << "Cannot set Hardy-Weinberg filter to < 0..." << std::endl; return 1; }
			if (st.hwe > 1) { std::cerr << stamp("ERROR") << "Cannot set Hardy-Weinberg filter to > 1..." << std::endl; return 1; }
			break;
		case 'r': st.remove_univariate = false; break;
		case 'f': st.flip_major_minor = false; break;          // import.h:92-94
		case 'b': st.block_size = (uint32_t)atoi(optarg); break;
		case 'L': st.c_level = (uint8_t)atoi(optarg); break;
		case 't': st.n_threads = atoi(optarg); break;
		default: std::cerr << stamp("ERROR") << "Unrecognized option: " << (char)c << std::endl; return 1;
		}
	}
	if (st.input.empty()) { std::cerr << stamp("ERROR") << "No input value specified..." << std::endl; return 1; }
	if (st.output.empty()) { std::cerr << stamp("ERROR") << "No output value specified..." << std::endl; return 1; }
	if (st.block_size == 0) { std::cerr << stamp("ERROR") << "Cannot set the block size to 0..." << std::endl; return 1; }
	program_message();
	std::cerr << stamp("LOG") << "Calling import..." << std::endl;
	tomahawk::twk_variant_importer importer;
	if (!importer.Import(st)) { std::cerr << "failed import" << std::endl; return 1; }
	return 0;
}

// `tomahawk sort` (lib/sort.h:28-124)
static int sort_cmd(int argc, char** argv) {
	if (argc < 3) {
		program_message();
		std::cerr << "About:  Sort TWO files\n\nUsage:  tomahawk sort [options] -i <in.two>\n\nOptions:\n"
		             "  -i FILE   input TWO file (required)\n  -o FILE   output file (- for stdout; default: -)\n"
		             "  -m FLOAT  maximum memory usage per thread in GB (default: 0.5)\n"
		             "  -c INT    compression level 1-20 (default: 1)\n  -t INT    number of threads (default: maximum available)\n\n";
		return 0;
	}
	static struct option long_options[] = {{"input", required_argument, 0, 'i'}, {"output", optional_argument, 0, 'o'},
		{"memory-usage", optional_argument, 0, 'm'}, {"compression-level", optional_argument, 0, 'c'},
		{"threads", optional_argument, 0, 't'}, {0, 0, 0, 0}};
	tomahawk::two_sorter_settings st;
	int c = 0, long_index = 0;
	while ((c = getopt_long(argc, argv, "i:o:m:c:t:?", long_options, &long_index)) != -1) {
		switch (c) {
		case 'i': st.in = optarg; break;
		case 'o': st.out = optarg; break;
		case 'm': st.memory_limit = (float)atof(optarg); break;
		case 'c': st.c_level = atoi(optarg); break;
		case 't': st.n_threads = atoi(optarg); break;
		default: fprintf(stderr, "%s: option `-%c' is invalid: ignored\n", argv[0], optopt); break;
		}
	}
	if (st.in.empty()) { std::cerr << stamp("ERROR") << "No input value specified..." << std::endl; return 1; }
	if (st.memory_limit <= 0) { std::cerr << stamp("ERROR") << "Cannot set memory limit <= 0..." << std::endl; return 1; }
	if (st.n_threads <= 0) { std::cerr << stamp("ERROR") << "Cannot set number of threads <= 0..." << std::endl; return 1; }
	if (st.c_level <= 0) { std::cerr << stamp("ERROR") << "Cannot set the compression level <= 0..." << std::endl; return 1; }
	program_message();
	std::cerr << stamp("LOG") << "Calling sort..." << std::endl;
	return tomahawk::two_sort(st) ? 0 : 1;
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

static int run_main(int argc, char** argv) {
	if (argc == 1) {           // reference: lib/main.cpp:28-60 lists its commands
		program_message();
		std::cerr << "Usage: tomahawk <command> [options]\n\n"
		             "Commands:\n"
		             "  import   convert VCF text (plain / gzip) to .twk\n"
		             "  calc     calculate linkage disequilibrium: tomahawk calc [options] -i <in.twk> -o <output.two>\n"
		             "  scalc    linkage disequilibrium of one site against its neighbourhood\n"
		             "  sort     sort a .two file\n"
		             "  view     convert, filter and slice .two files\n"
		             "  concat   concatenate .two files from the same set of samples\n" << std::endl;
		return 1;
	}
	// The host tools allocate and free MB-sized block buffers on hundreds of threads; served by
	// mmap/munmap (glibc's default above 128 KiB) that serialises on the address-space lock.
	mallopt(M_MMAP_THRESHOLD, 1 << 30);
	mallopt(M_TRIM_THRESHOLD, 1 << 30);
	g_argv.assign(argv, argv + argc);
	tomahawk::LITERAL_COMMAND_LINE = "tomahawk";
	for (int i = 1; i < argc; ++i) tomahawk::LITERAL_COMMAND_LINE += " " + std::string(argv[i]);
	if (strcmp(argv[1], "calc") == 0) return calc(argc, argv);
	if (strncmp(argv[1], "concat", 6) == 0) return concat(argc, argv);
	if (strcmp(argv[1], "calc-single") == 0 || strcmp(argv[1], "scalc") == 0) return scalc(argc, argv);
	if (strcmp(argv[1], "view") == 0) return view(argc, argv);
	if (strcmp(argv[1], "sort") == 0) return sort_cmd(argc, argv);
	if (strcmp(argv[1], "import") == 0) return import_cmd(argc, argv);
	if (strcmp(argv[1], "--version") == 0 || strcmp(argv[1], "version") == 0) { program_message(); return 0; }
	if (strcmp(argv[1], "--help") == 0 || strcmp(argv[1], "help") == 0) { calc_usage(); return 0; }
	program_message();
	std::cerr << stamp("ERROR") << "Illegal command: only `import`, `calc`, `scalc`, `concat`, `view` and `sort` are provided by the MI355X engine (aggregate/decay/... are the reference's)" << std::endl;
	return 1;
}

int main(int argc, char** argv) {
	try {
		return run_main(argc, argv);
	} catch (const std::bad_alloc&) {
		std::cerr << stamp("ERROR") << "Out of memory (or a corrupt file declaring an absurd size)..." << std::endl;
	} catch (const std::exception& e) {
		std::cerr << stamp("ERROR") << e.what() << std::endl;
	}
	return 1;
}
