// VCF text -> .twk -- see twk_import.h.
#include "twk_import.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <strings.h>
#include <thread>
#include <unordered_map>
#include <vector>
#include <zlib.h>

#include "twk_format.h"
#include "twk_util.h"

namespace tomahawk {
using namespace util;

// ---- Hardy-Weinberg exact test --------------------------------------------------------------------
// Wigginton, Cutler & Abecasis, AJHG 76 (2005): probability of every heterozygote count compatible
// with the allele counts, by recurrence from the mode; P = sum of the probabilities not larger than
// that of the observed count.  Evaluated in the order of core.cpp:133-200 (same rounding).
double hardy_weinberg_exact(uint64_t obs_hom1, uint64_t obs_hets, uint64_t obs_hom2) {
	const int64_t hom_common = (int64_t)std::max(obs_hom1, obs_hom2), hom_rare = (int64_t)std::min(obs_hom1, obs_hom2);
	const int64_t rare = 2 * hom_rare + (int64_t)obs_hets;          // copies of the rare allele
	const int64_t n = (int64_t)obs_hets + hom_common + hom_rare;    // genotypes
	if (n == 0) return 1.0;
	std::vector<double> prob((size_t)rare + 1, 0.0);
	int64_t mid = rare * (2 * n - rare) / (2 * n);                  // expected heterozygotes
	if ((rare & 1) ^ (mid & 1)) ++mid;                              // same parity as the rare-allele count
	prob[mid] = 1.0;
	double sum = 1.0;
	{   // downwards: two heterozygotes -> one rare + one common homozygote
		int64_t hr = (rare - mid) / 2, hc = n - mid - hr;
		for (int64_t h = mid; h > 1; h -= 2) {
			prob[h - 2] = prob[h] * h * (h - 1.0) / (4.0 * (hr + 1.0) * (hc + 1.0));
			sum += prob[h - 2];
			++hr; ++hc;
		}
	}
	{   // upwards
		int64_t hr = (rare - mid) / 2, hc = n - mid - hr;
		for (int64_t h = mid; h <= rare - 2; h += 2) {
			prob[h + 2] = prob[h] * 4.0 * hr * hc / ((h + 2.0) * (h + 1.0));
			sum += prob[h + 2];
			--hr; --hc;
		}
	}
	for (auto& p : prob) p /= sum;
	double p_hwe = 0.0;
	const double p_obs = prob[obs_hets];
	for (int64_t i = 0; i <= rare; ++i) if (!(prob[i] > p_obs)) p_hwe += prob[i];
	return p_hwe > 1.0 ? 1.0 : p_hwe;
}

namespace {

// ---- input ------------------------------------------------------------------------------------------
class LineReader {           // VCF text (plain or gzip/bgzip) or BCF2 (bgzip or raw), "-" = stdin
public:
	bool open(const std::string& path) {
		gz_ = (path == "-") ? gzdopen(0, "rb") : gzopen(path.c_str(), "rb");
		if (!gz_) return false;
		gzbuffer(gz_, 1 << 20);
		buf_.resize(1 << 22);
		fill();
		// BCF2 (htslib's binary VCF; lib/importer.cpp:25-337 reads either through bcf_read): magic "BCF\2\2"
		if (len_ >= 5 && std::memcmp(buf_.data(), "BCF\2\2", 5) == 0) { bcf_ = true; pos_ = 5; return bcf_header(); }
		return true;
	}
	~LineReader() { if (gz_) gzclose(gz_); }
	// Next text line.  For BCF input: the header text line by line, then every record as the VCF data line
	// that carries what the importer reads (CHROM POS . REF ALT . . . FORMAT GT...), so that one site parser and
	// one set of filters serves both containers.
	bool next(std::string& line) {
		line.clear();
		if (bcf_) {
			if (hdr_line_ < hdr_lines_.size()) { line = hdr_lines_[hdr_line_++]; return true; }
			return bcf_record(line);
		}
		for (;;) {
			if (pos_ == len_) { if (!fill()) return !line.empty(); }
			const char* s = buf_.data() + pos_;
			const char* nl = (const char*)std::memchr(s, '\n', len_ - pos_);
			if (nl) {
				line.append(s, (size_t)(nl - s));
				pos_ += (size_t)(nl - s) + 1;
				if (!line.empty() && line.back() == '\r') line.pop_back();
				return true;
			}
			line.append(s, len_ - pos_);
			pos_ = len_;
		}
	}
	bool bad() const { return bad_; }
private:
	gzFile gz_ = nullptr;
	std::vector<char> buf_;
	size_t pos_ = 0, len_ = 0;
	bool bad_ = false, bcf_ = false;
	// BCF state: header lines to hand out, the two dictionaries (header_internal / htslib bcf_hdr: strings of
	// FILTER/INFO/FORMAT IDs with PASS at 0, contigs in header order, both overridable with IDX=)
	std::vector<std::string> hdr_lines_; size_t hdr_line_ = 0;
	std::vector<std::string> dict_str_, dict_ctg_;
	std::vector<uint8_t> rec_;

	bool fill() {
		const int n = gzread(gz_, buf_.data(), (unsigned)buf_.size());
		if (n <= 0) { if (n < 0) bad_ = true; pos_ = len_ = 0; return false; }
		pos_ = 0; len_ = (size_t)n;
		return true;
	}
	bool read_bytes(void* dst, size_t n) {
		uint8_t* d = static_cast<uint8_t*>(dst);
		while (n) {
			if (pos_ == len_ && !fill()) return false;
			const size_t k = std::min(n, len_ - pos_);
			std::memcpy(d, buf_.data() + pos_, k);
			d += k; pos_ += k; n -= k;
		}
		return true;
	}
	static std::string field_of(const std::string& line, const char* key) {      // value of key= inside <...>
		const std::string k = std::string(key) + "=";
		size_t i = line.find('<');
		while (i != std::string::npos) {
			const size_t j = line.find(k, i + 1);
			if (j == std::string::npos) return "";
			if (line[j - 1] == '<' || line[j - 1] == ',') {
				size_t e = j + k.size(), b = e;
				bool q = false;
				while (e < line.size() && (q || (line[e] != ',' && line[e] != '>'))) { if (line[e] == '"') q = !q; ++e; }
				return line.substr(b, e - b);
			}
			i = j;
		}
		return "";
	}
	bool bcf_header() {
		uint32_t l_text = 0;
		if (!read_bytes(&l_text, 4) || l_text > (1u << 30)) { bad_ = true; return false; }
		std::string text(l_text, '\0');
		if (l_text && !read_bytes(&text[0], l_text)) { bad_ = true; return false; }
		while (!text.empty() && text.back() == '\0') text.pop_back();
		size_t b = 0;
		while (b < text.size()) {
			size_t e = text.find('\n', b);
			if (e == std::string::npos) e = text.size();
			if (e > b) hdr_lines_.push_back(text.substr(b, e - b));
			b = e + 1;
		}
		dict_str_.push_back("PASS");
		// an IDX= sizes the dictionary: bounded (1 << 24 entries is far beyond any real header), or a hostile
		// header asks for gigabytes
		bool idx_ok = true;
		auto place = [&idx_ok](std::vector<std::string>& d, const std::string& id, const std::string& idx) {
			if (!idx.empty()) {
				const long long k = std::atoll(idx.c_str());
				if (k < 0 || k > (1ll << 24)) { idx_ok = false; return; }
				if (d.size() <= (size_t)k) d.resize((size_t)k + 1);
				d[(size_t)k] = id; return;
			}
			if (std::find(d.begin(), d.end(), id) == d.end()) d.push_back(id);
		};
		for (const auto& l : hdr_lines_) {
			if (l.compare(0, 10, "##contig=<") == 0) place(dict_ctg_, field_of(l, "ID"), field_of(l, "IDX"));
			else if (l.compare(0, 10, "##FILTER=<") == 0 || l.compare(0, 8, "##INFO=<") == 0 || l.compare(0, 10, "##FORMAT=<") == 0) {
				const std::string id = field_of(l, "ID");
				if (id == "PASS" && field_of(l, "IDX").empty()) continue;
				place(dict_str_, id, field_of(l, "IDX"));
			}
		}
		if (!idx_ok) { std::cerr << "BCF header: IDX= outside [0, " << (1 << 24) << "]" << std::endl; bad_ = true; return false; }
		return true;
	}
	// typed values (BCF2 spec 6.3): descriptor byte = length << 4 | type, length 15 = a typed integer follows
	struct Typed { int type = 0; uint32_t n = 0; const uint8_t* p = nullptr; };
	static int type_size(int t) { return t == 1 ? 1 : t == 2 ? 2 : t == 3 ? 4 : t == 5 ? 4 : t == 7 ? 1 : 0; }
	static bool typed_int(const uint8_t*& p, const uint8_t* end, int64_t& v) {
		if (p >= end) return false;
		const int t = *p & 15, n = *p >> 4; ++p;
		if (n != 1 || (t != 1 && t != 2 && t != 3) || p + type_size(t) > end) return false;
		if (t == 1) v = (int8_t)p[0]; else if (t == 2) { int16_t x; std::memcpy(&x, p, 2); v = x; } else { int32_t x; std::memcpy(&x, p, 4); v = x; }
		p += type_size(t);
		return true;
	}
	static bool typed(const uint8_t*& p, const uint8_t* end, Typed& out) {
		if (p >= end) return false;
		out.type = *p & 15; uint32_t n = *p >> 4; ++p;
		if (n == 15) { int64_t v; if (!typed_int(p, end, v) || v < 0) return false; n = (uint32_t)v; }
		out.n = n; out.p = p;
		const size_t bytes = (size_t)n * type_size(out.type);
		if ((size_t)(end - p) < bytes) return false;
		p += bytes;
		return true;
	}
	bool bcf_record(std::string& line) {
		uint32_t l_shared = 0, l_indiv = 0;
		if (!read_bytes(&l_shared, 4)) return false;                                // clean end of file
		if (!read_bytes(&l_indiv, 4) || l_shared < 24 || (uint64_t)l_shared + l_indiv > (1ull << 31)) { bad_ = true; return false; }
		rec_.resize((size_t)l_shared + l_indiv);
		if (!read_bytes(rec_.data(), rec_.size())) { bad_ = true; return false; }
		const uint8_t* p = rec_.data(); const uint8_t* const se = p + l_shared; const uint8_t* const ie = se + l_indiv;
		int32_t chrom, pos; uint32_t n_allele_info, n_fmt_sample;
		std::memcpy(&chrom, p, 4); std::memcpy(&pos, p + 4, 4); std::memcpy(&n_allele_info, p + 16, 4); std::memcpy(&n_fmt_sample, p + 20, 4);
		p += 24;
		const uint32_t n_allele = n_allele_info >> 16, n_fmt = n_fmt_sample >> 24, n_sample = n_fmt_sample & 0xFFFFFFu;
		Typed id;
		if (!typed(p, se, id)) { bad_ = true; return false; }
		line = (chrom >= 0 && (size_t)chrom < dict_ctg_.size()) ? dict_ctg_[chrom] : std::string("?");
		line += '\t'; line += std::to_string((int64_t)pos + 1); line += "\t.";
		std::string alt;
		for (uint32_t a = 0; a < n_allele; ++a) {
			Typed al;
			if (!typed(p, se, al) || al.type != 7) { bad_ = true; return false; }
			const std::string s((const char*)al.p, al.n);
			if (a == 0) { line += '\t'; line += s.empty() ? "." : s; }
			else { if (!alt.empty()) alt += ','; alt += s; }
		}
		if (n_allele == 0) line += "\t.";
		line += '\t'; line += alt.empty() ? "." : alt;
		line += "\t.\t.\t.";                                                      // QUAL FILTER INFO: not read by the importer
		if (n_fmt == 0) return true;                                                 // no FORMAT column: "no fmt" (importer.cpp:278-282)
		// FORMAT keys; the genotypes are written out only when GT leads (VCF requires it to; the importer looks there)
		p = se;
		std::string fmt; Typed gt; bool have_gt = false;
		for (uint32_t k = 0; k < n_fmt; ++k) {
			int64_t key;
			if (!typed_int(p, ie, key)) { bad_ = true; return false; }
			if (p >= ie) { bad_ = true; return false; }
			Typed v; v.type = *p & 15; uint32_t n = *p >> 4; ++p;
			if (n == 15) { int64_t x; if (!typed_int(p, ie, x) || x < 0) { bad_ = true; return false; } n = (uint32_t)x; }
			v.n = n; v.p = p;
			const size_t bytes = (size_t)n * type_size(v.type) * n_sample;
			if ((size_t)(ie - p) < bytes) { bad_ = true; return false; }
			p += bytes;
			const std::string name = (key >= 0 && (size_t)key < dict_str_.size()) ? dict_str_[key] : std::string("?");
			if (k) fmt += ':';
			fmt += name;
			if (k == 0 && name == "GT") { gt = v; have_gt = true; }
		}
		line += '\t'; line += fmt;
		const int es = type_size(gt.type);
		for (uint32_t s = 0; s < n_sample; ++s) {
			line += '\t';
			if (!have_gt || (gt.type != 1 && gt.type != 2 && gt.type != 3)) { line += '.'; continue; }
			bool any = false;
			for (uint32_t k = 0; k < gt.n; ++k) {
				const uint8_t* q = gt.p + ((size_t)s * gt.n + k) * es;
				int32_t v; bool vec_end;
				if (es == 1) { v = (int8_t)q[0]; vec_end = v == -127; } else if (es == 2) { int16_t x; std::memcpy(&x, q, 2); v = x; vec_end = x == -32767; }
				else { std::memcpy(&v, q, 4); vec_end = v == -2147483647; }
				if (vec_end) break;                                                  // lower ploidy than the widest sample
				if (k) line += (v & 1) ? '|' : '/';
				const int allele = (v >> 1) - 1;                                     // 0 = missing
				if (allele < 0) line += '.'; else line += std::to_string(allele);
				any = true;
			}
			if (!any) line += '.';
		}
		return true;
	}
};

// `##key=<a=b,c="d,e">` -> pairs (values keep their quotes, like htslib's hrec)
std::vector<std::pair<std::string, std::string>> structured_fields(const std::string& line) {
	std::vector<std::pair<std::string, std::string>> out;
	const size_t lt = line.find('<'), gt = line.rfind('>');
	if (lt == std::string::npos || gt == std::string::npos || gt <= lt) return out;
	size_t i = lt + 1;
	while (i < gt) {
		const size_t eq = line.find('=', i);
		if (eq == std::string::npos || eq > gt) break;
		const std::string key = line.substr(i, eq - i);
		size_t j = eq + 1; bool q = false;
		while (j < gt && (q || line[j] != ',')) { if (line[j] == '"' && line[j - 1] != '\\') q = !q; ++j; }
		out.emplace_back(key, line.substr(eq + 1, j - eq - 1));
		i = j + 1;
	}
	return out;
}

enum Drop { KEEP = -1, INVARIANT = 0, MISS_THRESHOLD = 1, FEW_SAMPLES = 2, MIXED_PLOIDY = 3, NO_GT = 4, NO_FORMAT = 5,
            NOT_BIALLELIC = 6, NOT_SNP = 7, HWE = 8, MALFORMED = 9 };
const char* const DROP_NAMES[9] = {"Invariant", "Missing threshold", "Insufficient samples", "Mixed ploidy", "No genotypes",
                                   "No FORMAT", "Not biallelic", "Not SNP", "Hardy-Weinberg threshold"};

struct Site {
	std::string chrom;
	uint32_t pos = 0;            // 0-based
	bool biallelic_snp = false;  // two alleles, both one of A, T, G, C (the duplicate-site message, importer.cpp:127-131)
	int drop = KEEP;
	Variant v;
};

inline bool canonical(const char* s, size_t n) { return n == 1 && (s[0] == 'A' || s[0] == 'T' || s[0] == 'G' || s[0] == 'C'); }   // tomahawk.h:56
inline uint8_t base_code(char c) { return c == 'T' ? 1 : c == 'G' ? 2 : c == 'C' ? 3 : c == 'N' ? 4 : 0; }                            // core.h:38-47

struct Scratch { std::vector<int8_t> gt; std::vector<uint8_t> phase; };

// One data line.  Order of the tests as importer.cpp:137-196 + GenotypeEncoder::Encode (genotype_encoder.h:197-275).
void parse_site(const std::string& line, uint32_t n_samples, const twk_vimport_settings& st, Site& out, Scratch& sc) {
	std::vector<int8_t>& gt = sc.gt;
	out = Site();
	const char* p = line.data();
	const char* const end = p + line.size();
	const char* f[10]; size_t fl[10]; int nf = 0;
	while (nf < 9 && p <= end) {
		const char* t = (const char*)std::memchr(p, '\t', (size_t)(end - p));
		if (!t) t = end;
		f[nf] = p; fl[nf] = (size_t)(t - p); ++nf;
		p = t + 1;
	}
	if (nf < 5) { out.drop = MALFORMED; return; }
	out.chrom.assign(f[0], fl[0]);
	out.pos = (uint32_t)(std::strtoull(std::string(f[1], fl[1]).c_str(), nullptr, 10) - 1);
	const bool multi_alt = std::memchr(f[4], ',', fl[4]) != nullptr;
	const bool no_alt = fl[4] == 1 && f[4][0] == '.';
	const bool two_alleles = !multi_alt && !no_alt;
	out.biallelic_snp = two_alleles && canonical(f[3], fl[3]) && canonical(f[4], fl[4]);
	if (nf < 9) { out.drop = NO_FORMAT; return; }                                   // importer.cpp:278-282
	if (!(fl[8] >= 2 && f[8][0] == 'G' && f[8][1] == 'T' && (fl[8] == 2 || f[8][2] == ':'))) { out.drop = NO_GT; return; }   // :272-276
	if (!two_alleles) { out.drop = NOT_BIALLELIC; return; }                         // :147-152
	if (!out.biallelic_snp) { out.drop = NOT_SNP; return; }                         // :155-165
	// genotypes: per sample (first, second) in {0 ref, 1 alt, 2 missing}; the phase is the separator
	gt.resize((size_t)2 * n_samples);
	sc.phase.resize(n_samples);
	uint64_t hap[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, cnt[3] = {0, 0, 0};
	bool non_diploid = false, have_phase = false, mixed = false;
	int phase_uniform = 0;
	uint32_t s = 0;
	for (; s < n_samples && p <= end; ++s) {
		const char* t = p;
		int g[2] = {2, 2}, k = 0, ph = 0;
		while (t < end && *t != '\t' && *t != ':') {
			int a;
			if (*t == '.') { a = 2; ++t; }
			else if (*t >= '0' && *t <= '9') { int x = 0; while (t < end && *t >= '0' && *t <= '9') x = x * 10 + (*t++ - '0'); a = x; if (a > 1) { out.drop = NOT_BIALLELIC; return; } }
			else { out.drop = MALFORMED; return; }
			if (k < 2) g[k] = a;
			++k;
			if (t < end && (*t == '|' || *t == '/')) { if (k == 1) ph = (*t == '|'); ++t; }
		}
		if (k != 2) non_diploid = true;
		gt[2 * s] = (int8_t)g[0]; gt[2 * s + 1] = (int8_t)g[1]; sc.phase[s] = (uint8_t)ph;
		++cnt[g[0]]; ++cnt[g[1]]; ++hap[g[0] * 3 + g[1]];
		if (g[0] != 2 && g[1] != 2 && !have_phase) { phase_uniform = ph; have_phase = true; }    // genotype_encoder.h:77-87
		while (t < end && *t != '\t') ++t;
		p = t + 1;
	}
	if (s < n_samples) { out.drop = MALFORMED; return; }
	if (non_diploid) { out.drop = MIXED_PLOIDY; return; }                           // importer.cpp:139-145, genotype_encoder.h:222-226
	// mixed phasing: a sample whose second allele is present and whose separator differs from the
	// first complete genotype's (:95-103)
	for (uint32_t i = 0; i < n_samples && !mixed; ++i) if (gt[2 * i + 1] != 2 && sc.phase[i] != phase_uniform) mixed = true;
	const uint64_t total_hap = hap[0] + hap[1] + hap[3] + hap[4];                 // complete genotypes
	if ((float)total_hap < st.threshold_miss * (float)n_samples) { out.drop = MISS_THRESHOLD; return; }   // genotype_encoder.h:206-211
	if (total_hap < 5) { out.drop = FEW_SAMPLES; return; }                          // :213-217
	if (hap[0] == total_hap || hap[1] == total_hap || hap[3] == total_hap || hap[4] == total_hap) {
		if (st.remove_univariate) { out.drop = INVARIANT; return; }                 // :225-236
	}
	// st.flip_major_minor can only be false (importer.h:34, import.h:92-94): alleles are never flipped
	out.v.encode(gt.data(), n_samples, mixed ? false : phase_uniform != 0);
	out.v.pos = out.pos;
	out.v.alleles = (uint8_t)(base_code(f[3][0]) << 4 | base_code(f[4][0]));       // core.h:271
	out.v.hwe = hardy_weinberg_exact(hap[0], hap[1] + hap[3], hap[4]);
	if (out.v.hwe < st.hwe) { out.drop = HWE; return; }                             // importer.cpp:180-186
}

}  // namespace

bool twk_variant_importer::Import(twk_vimport_settings& s) { settings = s; const bool ok = Import(); s = settings; return ok; }

bool twk_variant_importer::Import(void) {
	using clock = std::chrono::steady_clock;
	const auto t0 = clock::now();
	if (settings.input != "-") std::cerr << stamp("LOG", "READER") << "Opening " << settings.input << "..." << std::endl;
	LineReader in;
	if (!in.open(settings.input)) { std::cerr << "failed to get vcfreader" << std::endl; return false; }

	// ---- header (vcf_reader.h:107-180, header_internal.cpp:17-60) ----
	Header hdr;
	hdr.fileformat = "fileformat";           // vcf_reader.h:121 stores the key of the first header record
	std::unordered_map<std::string, uint32_t> contig_of;
	bool has_gt = false, has_pass = false, has_fileformat = false;
	std::vector<std::string> meta;
	std::string line;
	bool got_columns = false;
	while (in.next(line)) {
		if (line.size() >= 2 && line[0] == '#' && line[1] == '#') {
			if (meta.empty() && line.compare(0, 13, "##fileformat=") == 0) has_fileformat = true;
			if (line.compare(0, 10, "##contig=<") == 0) {
				Contig c; c.idx = (uint32_t)hdr.contigs.size();
				for (const auto& kv : structured_fields(line)) {
					if (kv.first == "ID") c.name = kv.second;
					else if (kv.first == "length") c.n_bases = std::atoll(kv.second.c_str());
					else if (kv.first == "IDX") c.idx = (uint32_t)std::atoi(kv.second.c_str());
					else c.extra.emplace_back(kv.first, kv.second);
				}
				if (contig_of.count(c.name)) { std::cerr << stamp("ERROR") << "Illegal: duplicated contig name" << std::endl; return false; }
				contig_of[c.name] = (uint32_t)hdr.contigs.size();
				hdr.contigs.push_back(c);
			} else if (line.compare(0, 10, "##FORMAT=<") == 0) {
				for (const auto& kv : structured_fields(line)) if (kv.first == "ID" && kv.second == "GT") has_gt = true;
			} else if (line.compare(0, 10, "##FILTER=<") == 0) {
				for (const auto& kv : structured_fields(line)) if (kv.first == "ID" && kv.second == "PASS") has_pass = true;
			}
			meta.push_back(line);
			continue;
		}
		if (!line.empty() && line[0] == '#') {
			size_t col = 0, b = 0;
			while (b <= line.size()) {
				size_t e = line.find('\t', b);
				if (e == std::string::npos) e = line.size();
				if (col >= 9) hdr.samples.push_back(line.substr(b, e - b));
				++col; b = e + 1;
			}
			got_columns = true;
			break;
		}
		break;
	}
	if (meta.empty() || !got_columns) { std::cerr << stamp("ERROR") << "Empty header, not a valid VCF." << std::endl; return false; }
	if (!has_fileformat) std::cerr << stamp("ERROR") << "Not a valid VCF, fileformat needed: " << settings.input << std::endl;
	if (!has_gt) { std::cerr << "Genotype data not set in this file" << std::endl; return false; }     // importer.cpp:40-43
	// literals: the header text up to the column line; htslib always carries the PASS filter
	for (size_t i = 0; i < meta.size(); ++i) {
		hdr.literals += meta[i] + "\n";
		if (i == 0 && !has_pass) hdr.literals += "##FILTER=<ID=PASS,Description=\"All filters passed\">\n";
	}
	const uint32_t n_samples = (uint32_t)hdr.samples.size();
	std::cerr << stamp("LOG", "VCF") << "Constructing lookup table for " << pretty(hdr.contigs.size()) << " contigs..." << std::endl;
	std::cerr << stamp("LOG", "VCF") << "Samples: " << pretty(n_samples) << "..." << std::endl;
	if (n_samples == 0) { std::cerr << stamp("ERROR") << "No samples in this file" << std::endl; return false; }

	// ---- output (importer.cpp:49-99) ----
	const bool to_stdout = settings.output.empty() || settings.output == "-";
	if (to_stdout) { std::cerr << stamp("ERROR", "WRITER") << "Writing a .twk to stdout is not supported: the index needs file offsets; give -o FILE" << std::endl; return false; }
	{
		const std::string ext = extension(settings.output);
		if (!(ext.size() == 3 && strncasecmp(ext.c_str(), "twk", 3) == 0)) {
			const std::string bp = base_path(settings.output);
			settings.output = (bp.size() ? bp + "/" : "") + base_name(settings.output) + ".twk";
		}
	}
	std::cerr << stamp("LOG", "WRITER") << "Opening " << settings.output << "..." << std::endl;
	hdr.literals += "##tomahawk_importVersion=" + std::string(TWK_AMD_VERSION) + "\n";
	hdr.literals += "##tomahawk_importCommand=" + command_line() + "; Date=" + datetime() + "\n";

	// Contigs the records name but the header does not declare are appended (htslib does the same,
	// with a warning); they must be known before the header is written, so the writer is opened
	// lazily at the first flush... the header precedes the blocks, hence: collect while parsing and
	// refuse undeclared contigs once a block is out.
	TwkWriter w;
	bool writer_open = false;
	auto open_writer = [&]() -> bool {
		if (writer_open) return true;
		if (!w.open(settings.output, hdr, settings.c_level)) { std::cerr << "failed to open" << std::endl; return false; }
		writer_open = true;
		return true;
	};

	// ---- records: read rounds of lines, parse them on T threads, consume in file order ----
	const int T = std::max(1, std::min(settings.n_threads > 0 ? settings.n_threads : (int)std::thread::hardware_concurrency(), util::usable_cpus()));
	const size_t round_bytes = 64u << 20;
	std::vector<std::string> lines, next_lines;
	std::vector<Site> sites;
	auto read_round = [&](std::vector<std::string>& dst) {
		dst.clear();
		size_t bytes = 0;
		std::string l;
		while (bytes < round_bytes && in.next(l)) { if (l.empty()) continue; bytes += l.size(); dst.push_back(std::move(l)); l.clear(); }
	};
	auto parse_round = [&](const std::vector<std::string>& src, std::vector<Site>& dst) {
		dst.assign(src.size(), Site());
		std::atomic<size_t> nxt{0};
		auto work = [&]() {
			Scratch sc;
			for (;;) { const size_t i = nxt.fetch_add(1); if (i >= src.size()) return; parse_site(src[i], n_samples, settings, dst[i], sc); }
		};
		std::vector<std::thread> th;
		const int nt = (int)std::min<size_t>((size_t)T, std::max<size_t>(1, src.size()));
		for (int t = 1; t < nt; ++t) th.emplace_back(work);
		work();
		for (auto& x : th) x.join();
	};

	Block block;
	bool first_block = true;
	uint32_t block_minpos = 0;
	struct { bool dropped = false; uint32_t rid = 0, pos = 0; } prev;
	std::memset(filtered, 0, sizeof(filtered));
	n_sites = n_written = n_duplicates = 0;
	uint64_t n_malformed = 0;
	// Finished blocks wait here until the end of the round, are compressed on the parser threads
	// and written in order (the reference compresses inline on its one thread).
	struct Pending { Block blk; int level; uint32_t minpos; TwkWriter::Packed packed; };
	std::vector<Pending> pending;
	auto flush = [&](int level) -> bool {
		pending.emplace_back();
		pending.back().blk.rid = block.rid; pending.back().blk.rcds.swap(block.rcds);
		pending.back().level = level; pending.back().minpos = block_minpos;
		block.rcds.clear();
		first_block = false;
		return true;
	};
	auto write_pending = [&]() -> bool {
		if (pending.empty()) return true;
		if (!open_writer()) return false;
		std::atomic<size_t> nxt{0};
		std::atomic<bool> ok{true};
		auto work = [&]() {
			for (;;) { const size_t i = nxt.fetch_add(1); if (i >= pending.size()) return;
				if (!TwkWriter::pack(pending[i].blk, pending[i].level, pending[i].minpos, pending[i].packed)) ok = false; }
		};
		std::vector<std::thread> th;
		const int nt = (int)std::min<size_t>((size_t)T, pending.size());
		for (int t = 1; t < nt; ++t) th.emplace_back(work);
		work();
		for (auto& x : th) x.join();
		if (!ok) return false;
		for (auto& pb : pending) if (!w.write_packed(pb.packed)) return false;
		pending.clear();
		return true;
	};

	read_round(lines);
	while (!lines.empty()) {
		std::thread reader([&] { read_round(next_lines); });     // overlap the (serial) inflate with the parse
		parse_round(lines, sites);
		reader.join();
		for (Site& sv : sites) {
			++n_sites;
			if (sv.drop == MALFORMED && sv.chrom.empty()) { ++n_malformed; continue; }
			auto it = contig_of.find(sv.chrom);
			if (it == contig_of.end()) {
				if (writer_open) { std::cerr << stamp("ERROR") << "Contig " << sv.chrom << " is not declared in the header" << std::endl; return false; }
				std::cerr << stamp("LOG", "VCF") << "Contig '" << sv.chrom << "' is not defined in the header. (Quick workaround: index the file with tabix.)" << std::endl;
				Contig c; c.idx = (uint32_t)hdr.contigs.size(); c.name = sv.chrom; c.n_bases = 0;
				it = contig_of.emplace(sv.chrom, c.idx).first;
				hdr.contigs.push_back(c);
			}
			const uint32_t rid = it->second;
			if (!prev.dropped && prev.rid == rid && prev.pos == sv.pos) {       // importer.cpp:107-136 (the state starts at contig 0, position 0)
				if (sv.biallelic_snp) std::cerr << stamp("LOG") << "Duplicate site dropped: " << sv.chrom << ":" << sv.pos + 1 << std::endl;
				prev.rid = rid; prev.pos = sv.pos; prev.dropped = true;
				++n_duplicates;
				continue;
			}
			prev.rid = rid; prev.pos = sv.pos; prev.dropped = false;
			if (sv.drop != KEEP) {
				if (sv.drop == MALFORMED) ++n_malformed; else ++filtered[sv.drop];
				if (sv.drop == NO_GT) std::cerr << "no genotypes" << std::endl;
				if (sv.drop == NO_FORMAT) std::cerr << "no fmt" << std::endl;
				prev.dropped = true;
				continue;
			}
			sv.v.rid = rid;
			if (!block.rcds.empty()) {                                                       // importer.cpp:188-262
				if (block.rid != rid) { if (!flush(settings.c_level)) return false; }
				if (block.rcds.size() == settings.block_size) { if (!flush(10)) return false; }    // full blocks: zstd level 10 (:232)
				if (block.rcds.empty()) { block.rid = rid; block_minpos = sv.pos + 1; }         // twk1_block_t::Add (core.cpp:221)
			} else {
				block.rid = rid; block_minpos = first_block ? sv.pos : sv.pos + 1;              // :262-264: 0-based for the very first block
			}
			block.rcds.push_back(std::move(sv.v));
			++n_written;
		}
		if (!write_pending()) { std::cerr << "failed to compress" << std::endl; return false; }
		lines.swap(next_lines);
	}
	if (in.bad()) { std::cerr << stamp("ERROR") << "Failed to parse VCF record: read error" << std::endl; return false; }
	if (!block.rcds.empty() && !flush(settings.c_level)) return false;
	if (!write_pending()) { std::cerr << "failed to compress" << std::endl; return false; }
	if (!open_writer() || !w.close()) { std::cerr << "failed to compress" << std::endl; return false; }

	const uint64_t n_out = w.n_variants();
	std::cerr << stamp("LOG") << "Wrote: " << pretty(n_out) << " variants to " << pretty(w.n_blocks()) << " blocks..." << std::endl;
	std::cerr << stamp("LOG") << "Finished: " << elapsed_string(std::chrono::duration<double>(clock::now() - t0).count()) << std::endl;
	const double tot = n_sites ? (double)n_sites : 1.0;
	std::cerr << stamp("LOG") << "Filtered out " << pretty(n_sites - n_out) << " sites (" << (float)((n_sites - n_out) / tot * 100) << "%):" << std::endl;
	for (int i = 0; i < 9; ++i)
		std::cerr << stamp("LOG") << "   " << DROP_NAMES[i] << ": " << pretty(filtered[i]) << " (" << (float)(filtered[i] / tot * 100) << "%)" << std::endl;
	if (n_malformed) std::cerr << stamp("LOG") << "   Malformed lines: " << pretty(n_malformed) << std::endl;
	return true;
}

}  // namespace tomahawk
