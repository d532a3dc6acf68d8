// Small C API over the host library for tests, tools and the benchmark harness
// (ctypes-friendly; plain pointers and sizes).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "twk_format.h"
#include "twk_ld.h"
#include "twk_hip.h"
#include "twk_two_tools.h"
#include "twk_import.h"
#include "twk_record_sink.h"
#include "twk_repcodec.h"

using namespace tomahawk;

// No C++ exception may cross the C boundary (ctypes callers would abort): every entry point is a
// function-try-block that turns one into -9.

extern "C" {

// Write a .twk from dense genotypes.  alleles: int8 [n_variants][2*n_samples] in {0,1,2};
// pos/rid: per variant (sorted by rid,pos); phased: per variant 0/1; hwe may be NULL (1.0);
// n_contigs contigs named "1".."n"; block_size variants per block (one contig per block).
int twk_file_write_twk(const char* path, uint32_t n_samples, uint32_t n_variants, const int8_t* alleles,
                       const uint32_t* pos, const uint32_t* rid, const uint8_t* phased, const double* hwe,
                       uint32_t n_contigs, uint32_t block_size, int c_level) try {
	if (!path || !alleles || !pos || !rid || !phased || n_samples == 0 || block_size == 0) return -1;
	Header hdr;
	hdr.literals = "##fileformat=VCFv4.2\n##source=tomahawk_amd synthetic writer";
	for (uint32_t s = 0; s < n_samples; ++s) hdr.samples.push_back("S" + std::to_string(s));
	for (uint32_t c = 0; c < n_contigs; ++c) { Contig k; k.idx = c; k.name = std::to_string(c + 1); k.n_bases = 250000000; hdr.contigs.push_back(k); }
	TwkWriter w;
	if (!w.open(path, hdr, c_level)) return -2;
	Block blk;
	for (uint32_t v = 0; v < n_variants; ++v) {
		if (rid[v] >= n_contigs) return -1;
		if (!blk.rcds.empty() && (blk.rid != rid[v] || blk.rcds.size() == block_size)) { if (!w.write_block(blk)) return -3; blk.rcds.clear(); }
		if (blk.rcds.empty()) blk.rid = rid[v];
		Variant x;
		x.encode(alleles + (size_t)v * 2 * n_samples, n_samples, phased[v] != 0);
		x.pos = pos[v]; x.rid = rid[v]; x.hwe = hwe ? hwe[v] : 1.0; x.alleles = 0x12;
		blk.rcds.push_back(std::move(x));
	}
	if (!blk.rcds.empty() && !w.write_block(blk)) return -3;
	return w.close() ? 0 : -3;
} catch (...) { return -9; }

// Write the first n_variants of the synthetic benchmark input as a .twk:
// the CPU-baseline sample shares its bits with twk_hip_generate_synthetic().
int twk_file_write_synthetic_twk(const char* path, uint32_t n_samples, uint32_t n_variants, uint64_t seed,
                                 int phased, uint32_t block_size, int c_level, int n_threads) try {
	if (!path || n_samples == 0 || n_variants == 0 || block_size == 0) return -1;
	Header hdr;
	hdr.literals = "##fileformat=VCFv4.2\n##source=tomahawk_amd synthetic benchmark input seed=" + std::to_string(seed);
	for (uint32_t s = 0; s < n_samples; ++s) hdr.samples.push_back("S" + std::to_string(s));
	Contig k; k.idx = 0; k.name = "1"; k.n_bases = 250000000; hdr.contigs.push_back(k);
	TwkWriter w;
	if (!w.open(path, hdr, c_level)) return -2;
	const size_t w64 = ((size_t)2 * n_samples + 63) / 64;
	const uint32_t T = (uint32_t)std::max(1, std::min(n_threads, util::usable_cpus()));
	for (uint32_t v0 = 0; v0 < n_variants; v0 += block_size) {
		const uint32_t nb = std::min(block_size, n_variants - v0);
		Block blk; blk.rid = 0; blk.rcds.resize(nb);
		auto job = [&](uint32_t t) {
			std::vector<uint64_t> bv(w64);
			std::vector<int8_t> al((size_t)2 * n_samples);
			for (uint32_t i = t; i < nb; i += T) {
				twk_synth_bitvector(seed, n_samples, v0 + i, bv.data());
				for (size_t p = 0; p < (size_t)2 * n_samples; ++p) al[p] = (int8_t)((bv[p >> 6] >> (p & 63)) & 1);
				Variant& x = blk.rcds[i];
				x.encode(al.data(), n_samples, phased != 0);
				x.pos = 1000u + 100u * (v0 + i); x.rid = 0; x.hwe = 1.0; x.alleles = 0x12;
			}
		};
		std::vector<std::thread> th;
		for (uint32_t t = 0; t < T; ++t) th.emplace_back(job, t);
		for (auto& t : th) t.join();
		if (!w.write_block(blk)) return -3;
	}
	return w.close() ? 0 : -3;
} catch (...) { return -9; }

// A .twk with the shape of real cohort data, for the end-to-end and allele-frequency-spectrum measurements
// (no counterpart in the reference; SURVEY 8(d)'s generator is iid): haplotypes are mosaics of `n_founders`
// founder haplotypes whose assignment is redrawn, per haplotype with probability `p_switch`, at every file
// block (so a block is a haplotype block); a fraction `rare_frac` of the variants is rare - ALT allele count
// drawn from a 1/x spectrum between 1 and max_rare_af * 2N, carried by haplotypes of one founder - the rest
// take their alleles from the founders (frequency U(0.05, 0.5)) plus `p_mut` noise; a fraction `miss_variants`
// of the variants has `miss_rate` missing samples.  Counter-based randomness: the file depends on the arguments
// only, not on the thread count.  Contigs "1".."n_contigs" of equal size, positions 1000 + spacing * i per contig.
namespace {
inline uint64_t cmix(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
inline double cunit(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }
}
int twk_file_write_cohort_twk(const char* path, uint32_t n_samples, uint32_t n_variants, uint64_t seed, uint32_t n_founders,
                              double p_switch, double p_mut, double rare_frac, double max_rare_af, double miss_variants,
                              double miss_rate, int phased, uint32_t n_contigs, uint32_t spacing, uint32_t block_size,
                              int c_level, int n_threads) try {
	if (!path || n_samples == 0 || n_variants == 0 || block_size == 0 || n_founders < 2 || n_founders > 255 || n_contigs == 0) return -1;
	Header hdr;
	hdr.literals = "##fileformat=VCFv4.2\n##source=tomahawk_amd cohort-shaped synthetic input seed=" + std::to_string(seed);
	for (uint32_t s = 0; s < n_samples; ++s) hdr.samples.push_back("S" + std::to_string(s));
	for (uint32_t c = 0; c < n_contigs; ++c) { Contig k; k.idx = c; k.name = std::to_string(c + 1); k.n_bases = 250000000; hdr.contigs.push_back(k); }
	TwkWriter w;
	if (!w.open(path, hdr, c_level)) return -2;
	const size_t H = (size_t)2 * n_samples;
	const uint32_t T = (uint32_t)std::max(1, std::min(n_threads, util::usable_cpus()));
	const uint32_t per_contig = (n_variants + n_contigs - 1) / n_contigs;
	std::vector<uint8_t> state(H);
	for (size_t h = 0; h < H; ++h) state[h] = (uint8_t)(cmix(seed ^ (0xA5A5ull << 32) ^ h) % n_founders);
	const uint64_t sw_thr = (uint64_t)(std::min(1.0, std::max(0.0, p_switch)) * 18446744073709551615.0);
	const uint64_t mut_thr = (uint64_t)(std::min(1.0, std::max(0.0, p_mut)) * 18446744073709551615.0);
	const uint64_t miss_thr = (uint64_t)(std::min(1.0, std::max(0.0, miss_rate)) * 18446744073709551615.0);
	uint32_t blk_no = 0;
	for (uint32_t v0 = 0; v0 < n_variants; ++blk_no) {
		const uint32_t contig = v0 / per_contig;
		const uint32_t nb = std::min(std::min(block_size, n_variants - v0), (contig + 1) * per_contig - v0);     // one contig per block
		// redraw founder assignments for this block
		{
			std::vector<std::thread> th;
			for (uint32_t t = 0; t < T; ++t) th.emplace_back([&, t] {
				for (size_t h = H * t / T, e = H * (t + 1) / T; h < e; ++h) {
					const uint64_t r = cmix(seed ^ ((uint64_t)(blk_no + 1) * 0xD6E8FEB86659FD93ull) ^ (h * 0x9E3779B97F4A7C15ull));
					if (r < sw_thr) state[h] = (uint8_t)(cmix(r) % n_founders);
				}
			});
			for (auto& t : th) t.join();
		}
		std::vector<size_t> founder_n(n_founders, 0);
		for (size_t h = 0; h < H; ++h) ++founder_n[state[h]];
		Block blk; blk.rid = contig; blk.rcds.resize(nb);
		auto job = [&](uint32_t t) {
			std::vector<int8_t> al(H);
			for (uint32_t i = t; i < nb; i += T) {
				const uint32_t v = v0 + i;
				const uint64_t vk = cmix(seed + 0x632BE59BD9B4E019ull * (uint64_t)(v + 1));
				const bool rare = cunit(cmix(vk ^ 1)) < rare_frac;
				uint8_t fa[256];
				uint64_t carrier_thr = 0; uint32_t rare_founder = 0;
				if (rare) {
					// 1/x spectrum: ac = exp(U(0, ln(max_ac)))
					const double max_ac = std::max(1.0, max_rare_af * (double)H);
					const double ac = std::exp(cunit(cmix(vk ^ 2)) * std::log(max_ac));
					rare_founder = (uint32_t)(cmix(vk ^ 3) % n_founders);
					const double q = std::min(1.0, ac / (double)std::max<size_t>(1, founder_n[rare_founder]));
					carrier_thr = (uint64_t)(q * 18446744073709551615.0);
				} else {
					const double p = 0.05 + 0.45 * cunit(cmix(vk ^ 4));
					uint32_t n1 = 0;
					for (uint32_t f = 0; f < n_founders; ++f) { fa[f] = cunit(cmix(vk ^ (0x100 + f))) < p ? 1 : 0; n1 += fa[f]; }
					if (n1 == 0) fa[cmix(vk ^ 5) % n_founders] = 1;
					if (n1 == n_founders) fa[cmix(vk ^ 6) % n_founders] = 0;
				}
				const bool miss = cunit(cmix(vk ^ 7)) < miss_variants;
				size_t n_alt = 0;
				for (size_t h = 0; h < H; ++h) {
					const uint64_t r = cmix(vk ^ (h * 0xD1342543DE82EF95ull));
					int8_t a;
					if (rare) a = (state[h] == rare_founder && r < carrier_thr) ? 1 : 0;
					else a = (int8_t)(fa[state[h]] ^ (r < mut_thr ? 1 : 0));
					al[h] = a; n_alt += (size_t)a;
				}
				if (n_alt == 0) { al[cmix(vk ^ 8) % H] = 1; }                        // keep every site polymorphic
				if (n_alt == H) { al[cmix(vk ^ 9) % H] = 0; }
				if (miss) for (uint32_t s2 = 0; s2 < n_samples; ++s2) if (cmix(vk ^ 0xABCDull ^ ((uint64_t)s2 << 20)) < miss_thr) { al[2 * (size_t)s2] = 2; al[2 * (size_t)s2 + 1] = 2; }
				Variant& x = blk.rcds[i];
				x.encode(al.data(), n_samples, phased != 0);
				x.pos = 1000u + spacing * (v - contig * per_contig); x.rid = contig; x.hwe = 1.0; x.alleles = 0x12;
			}
		};
		std::vector<std::thread> th;
		for (uint32_t t = 0; t < T; ++t) th.emplace_back(job, t);
		for (auto& t : th) t.join();
		if (!w.write_block(blk)) return -3;
		v0 += nb;
	}
	return w.close() ? 0 : -3;
} catch (...) { return -9; }

// Read a whole .twk: two-call pattern (first with data == NULL to get the sizes).
// meta: twk_hip_variant_meta[n_variants]; extra (may be NULL): uint32 [n_variants][4] = n_het, n_hom, gt_phase, n_runs.
int twk_file_read_twk(const char* path, uint32_t* n_samples, uint32_t* n_variants, uint64_t* data, uint64_t* mask,
                      twk_hip_variant_meta* meta, uint32_t* extra) try {
	TwkReader rd;
	if (!path || !rd.open(path)) return -2;
	uint32_t M = 0;
	for (const auto& e : rd.index.ent) M += e.n;
	const uint32_t N = (uint32_t)rd.hdr.samples.size();
	if (n_samples) *n_samples = N;
	if (n_variants) *n_variants = M;
	if (!data) return 0;
	const size_t w64 = ((size_t)2 * N + 63) / 64;
	uint32_t v = 0;
	for (size_t b = 0; b < rd.index.ent.size(); ++b) {
		Block blk;
		if (!rd.read_block(b, blk)) return -3;
		for (const auto& x : blk.rcds) {
			if (!x.build_bitvector(N, data + (size_t)v * w64, mask ? mask + (size_t)v * w64 : nullptr)) return -4;
			if (meta) { twk_hip_variant_meta& m = meta[v]; m.ac = x.ac; m.an = x.an; m.pos = x.pos; m.rid = x.rid; m.missing = x.gt_missing; m._pad = 0; m.hwe = x.hwe; }
			if (extra) { extra[4 * v] = x.n_het; extra[4 * v + 1] = x.n_hom; extra[4 * v + 2] = x.gt_phase; extra[4 * v + 3] = (uint32_t)x.runs.size(); }
			++v;
		}
	}
	return 0;
} catch (...) { return -9; }

// Read a whole .two: records are the 106-byte packed twk1_two_t.  Two-call pattern.
// info (may be NULL): [n_samples, n_contigs, n_index_blocks, index_state].
int twk_file_read_two(const char* path, void* records, uint64_t capacity, uint64_t* n_records, uint64_t* info) try {
	TwoReader rd;
	if (!path || !rd.open(path)) return -2;
	if (info) { info[0] = rd.hdr.samples.size(); info[1] = rd.hdr.contigs.size(); info[2] = rd.index.ent.size(); info[3] = rd.index.state; }
	std::vector<TwoRecord> blk;
	uint64_t n = 0;
	while (rd.next_block(blk)) {
		if (records) {
			if (n + blk.size() > capacity) return -4;
			std::memcpy((uint8_t*)records + n * sizeof(TwoRecord), blk.data(), blk.size() * sizeof(TwoRecord));
		}
		n += blk.size();
	}
	if (!rd.error.empty()) return -3;
	if (n_records) *n_records = n;
	// index consistency: entries must tile the block stream
	uint64_t idx_n = 0;
	for (const auto& e : rd.index.ent) idx_n += e.n;
	return idx_n == n ? 0 : -5;
} catch (...) { return -9; }

// Write an unsorted .two (the shape calc writes): 106-byte packed records in blocks of
// `block_records`; n_contigs contigs named "1".."n"; n_samples sample names "S<i>".
int twk_file_write_two(const char* path, const void* records, uint64_t n_records, uint32_t n_samples,
                       uint32_t n_contigs, uint32_t block_records, int c_level) try {
	if (!path || (!records && n_records) || block_records == 0) return -1;
	Header hdr;
	hdr.literals = "##fileformat=VCFv4.2\n##source=tomahawk_amd synthetic writer\n";
	for (uint32_t s = 0; s < n_samples; ++s) hdr.samples.push_back("S" + std::to_string(s));
	for (uint32_t c = 0; c < n_contigs; ++c) { Contig k; k.idx = c; k.name = std::to_string(c + 1); k.n_bases = 250000000; hdr.contigs.push_back(k); }
	TwoWriter w;
	if (!w.open(path, hdr, c_level)) return -2;
	const TwoRecord* r = (const TwoRecord*)records;
	for (uint64_t i = 0; i < n_records; i += block_records)
		if (!w.write_block(r + i, (uint32_t)std::min<uint64_t>(block_records, n_records - i))) return -3;
	return w.close() ? 0 : -3;
} catch (...) { return -9; }

// Index of a .two.  Two-call pattern: entries [n][6] = rid, ridB, n, minpos, maxpos, b_unc (as int64);
// contigs [m][5] = rid, n, minpos, maxpos, nn.  counts = [state, n, m].
int twk_file_two_index(const char* path, int64_t* entries, uint64_t cap_entries, int64_t* contigs, uint64_t cap_contigs,
                       uint64_t* counts) try {
	TwoReader rd;
	if (!path || !rd.open(path)) return -2;
	if (counts) { counts[0] = rd.index.state; counts[1] = rd.index.ent.size(); counts[2] = rd.index.meta.size(); }
	if (entries) {
		if (cap_entries < rd.index.ent.size()) return -4;
		for (size_t i = 0; i < rd.index.ent.size(); ++i) {
			const IndexEntryOutput& e = rd.index.ent[i];
			int64_t* o = entries + i * 6;
			o[0] = e.rid; o[1] = e.ridB; o[2] = e.n; o[3] = e.minpos; o[4] = e.maxpos; o[5] = e.b_unc;
		}
	}
	if (contigs) {
		if (cap_contigs < rd.index.meta.size()) return -4;
		for (size_t i = 0; i < rd.index.meta.size(); ++i) {
			const IndexEntryEntry& e = rd.index.meta[i];
			int64_t* o = contigs + i * 5;
			o[0] = e.rid; o[1] = e.n; o[2] = e.minpos; o[3] = e.maxpos; o[4] = (int64_t)e.nn;
		}
	}
	return 0;
} catch (...) { return -9; }

// two_reader::Sort through a flat argument list (lib/sort.h:93-123).
int twk_two_sort(const char* in, const char* out, double memory_limit_gb, int c_level, int n_threads) try {
	two_sorter_settings s;
	s.in = in ? in : ""; s.out = out ? out : "-";
	if (memory_limit_gb > 0) s.memory_limit = (float)memory_limit_gb;
	if (c_level > 0) s.c_level = c_level;
	if (n_threads > 0) s.n_threads = n_threads;
	return two_sort(s) ? 0 : 1;
} catch (...) { return -9; }

// twk_variant_importer::Import through a flat argument list (lib/import.h:46-128).
// counters (may be NULL): [0..8] sites dropped per reason (genotype_encoder.h:25-35), [9] duplicates,
// [10] sites read, [11] variants written.
int twk_import_vcf(const char* in, const char* out, double threshold_miss, double hwe, int remove_univariate,
                   uint32_t block_size, int c_level, int n_threads, uint64_t* counters) try {
	twk_vimport_settings s;
	s.input = in ? in : "-"; s.output = out ? out : "-";
	if (threshold_miss >= 0) s.threshold_miss = (float)threshold_miss;
	if (hwe >= 0) s.hwe = hwe;
	s.remove_univariate = remove_univariate != 0;
	if (block_size) s.block_size = block_size;
	if (c_level > 0) s.c_level = (uint8_t)c_level;
	s.n_threads = n_threads;
	twk_variant_importer imp;
	const bool ok = imp.Import(s);
	if (counters) {
		for (int i = 0; i < 9; ++i) counters[i] = imp.filtered[i];
		counters[9] = imp.n_duplicates; counters[10] = imp.n_sites; counters[11] = imp.n_written;
	}
	return ok ? 0 : 1;
} catch (...) { return -9; }

double twk_hwe_exact(uint64_t hom1, uint64_t het, uint64_t hom2) { return hardy_weinberg_exact(hom1, het, hom2); }

// Header literals of a .two / .twk (NUL terminated, truncated to cap).
int twk_file_header_literals(const char* path, int is_two, char* out, size_t cap) try {
	Header h;
	if (is_two) { TwoReader r; if (!r.open(path)) return -2; h = r.hdr; }
	else { TwkReader r; if (!r.open(path)) return -2; h = r.hdr; }
	if (out && cap) { std::strncpy(out, h.literals.c_str(), cap - 1); out[cap - 1] = 0; }
	return 0;
} catch (...) { return -9; }

// twk_ld::Compute through a flat argument list (what calc.h:96-238 builds).
int twk_ld_compute(const char* in, const char* out, int force_phased, int force_unphased, double minR2, double minP,
                   double minDprime, int window, int l_window, int n_chunks, int c_chunk, int n_threads, int c_level,
                   int b_size, uint64_t* n_pairs, uint64_t* n_records) try {
	twk_ld_settings s;
	s.in = in ? in : ""; s.out = out ? out : "-";
	s.force_phased = force_phased != 0; s.forced_unphased = force_unphased != 0;
	s.minR2 = minR2; s.minP = minP; s.minDprime = minDprime;
	s.window = window != 0; if (window) s.l_window = l_window;
	s.n_chunks = n_chunks; s.c_chunk = c_chunk;
	if (n_threads > 0) s.n_threads = n_threads;
	s.c_level = c_level; if (b_size > 0) s.b_size = b_size;
	twk_ld ld;
	const bool ok = ld.Compute(s);
	if (n_pairs) *n_pairs = ld.n_pairs();
	if (n_records) *n_records = ld.n_records();
	return ok ? 0 : 1;
} catch (...) { return -9; }

// The writer side of a calc run on its own: survivor records as the engine hands them out
// (twk_hip_record, variant indices into rid[] / pos[]) -> forward + reverse twk1_two_t blocks with the
// reference's flush rule -> a .two file (twk_record_sink.h).  This is what rank 0 of a multi-process
// run does with the records it gathered from the other ranks (bench.py); `tomahawk calc` uses the same
// classes directly.  Header: n_contigs contigs "1".."n", sample names "S<i>".
namespace {
struct TwoStream {
	TwoOutput out;
	std::vector<uint32_t> rid, pos;
	std::unique_ptr<RecordEmitter> emitter;
};
}
// CPUs the host side sizes its thread pools by: hardware threads, cut to the affinity mask and the container's CFS quota
// (twk_util.h usable_cpus).
int twk_usable_cpus(void) { return util::usable_cpus(); }

// The records' own zstd encoder (twk_repcodec.h) on a buffer: -> frame size, 0 if dst is too small (twk_record_codec_bound(n)
// always suffices).  Tests decode the frame with libzstd.
uint64_t twk_record_codec_bound(uint64_t n) { return repcodec::bound(n); }
uint64_t twk_record_codec_compress(const uint8_t* src, uint64_t n, uint32_t stride, uint8_t* dst, uint64_t cap) try {
	if ((!src && n) || !dst || stride == 0 || cap < repcodec::bound(n) || (n >> 32)) return 0;
	std::unique_ptr<repcodec::Work> w(new repcodec::Work);
	return repcodec::compress_frame(dst, src, n, stride, *w);
} catch (...) { return 0; }

void* twk_two_stream_open(const char* path, uint32_t n_samples, uint32_t n_contigs, const uint32_t* rid, const uint32_t* pos,
                          uint32_t n_variants, int c_level, uint32_t b_size, int n_threads, int map_output) try {
	if (!path || !rid || !pos || n_variants == 0 || b_size < 2) return nullptr;
	std::unique_ptr<TwoStream> st(new TwoStream);
	Header hdr;
	hdr.literals = "##fileformat=VCFv4.2\n##source=tomahawk_amd record stream\n";
	for (uint32_t s = 0; s < n_samples; ++s) hdr.samples.push_back("S" + std::to_string(s));
	for (uint32_t c = 0; c < n_contigs; ++c) { Contig k; k.idx = c; k.name = std::to_string(c + 1); k.n_bases = 250000000; hdr.contigs.push_back(k); }
	st->rid.assign(rid, rid + n_variants); st->pos.assign(pos, pos + n_variants);
	for (uint32_t v = 0; v < n_variants; ++v) if (st->rid[v] >= n_contigs) return nullptr;
	if (!st->out.writer.open(path, hdr, c_level > 0 ? c_level : 1)) return nullptr;
	if (map_output == 1) (void)st->out.writer.map_output();       // (0: through a stream also where the file could be mapped - A/B runs and tests of both paths)
	else if (map_output == 2) (void)st->out.writer.direct_output();      // (2: frames by pwritev, space reserved ahead)
	st->out.b_size = b_size; st->out.c_level = c_level > 0 ? c_level : 1;
	st->out.rid = st->rid.data(); st->out.pos = st->pos.data(); st->out.n_variants = st->rid.size();
	st->emitter.reset(new RecordEmitter(st->out, n_threads > 0 ? std::min(n_threads, util::usable_cpus()) : 1));
	return st.release();
} catch (...) { return nullptr; }

int twk_two_stream_append(void* h, const void* records, uint64_t n) try {
	auto* st = static_cast<TwoStream*>(h);
	if (!st || (!records && n)) return -1;
	const twk_hip_record* r = static_cast<const twk_hip_record*>(records);
	for (uint64_t i = 0; i < n; ++i) if (r[i].idxA >= st->rid.size() || r[i].idxB >= st->rid.size()) return -1;
	return st->emitter->emit(r, n, false) ? 0 : -3;
} catch (...) { return -9; }

int twk_two_stream_close(void* h, uint64_t* n_records) try {
	std::unique_ptr<TwoStream> st(static_cast<TwoStream*>(h));
	if (!st) return -1;
	bool ok = st->emitter->emit(nullptr, 0, true);
	ok = st->out.writer.close() && ok;
	if (n_records) *n_records = st->out.n_records;
	return ok ? 0 : -3;
} catch (...) { return -9; }

}  // extern "C"
