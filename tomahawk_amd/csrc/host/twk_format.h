// On-disk formats of the calc path: `.twk` in, `.two` out.
//
// Written from the format description in SURVEY.md Appendix B; every struct
// cites the reference serialiser it is byte-compatible with.  All integers are
// little-endian, packed, no alignment (reference: include/buffer.h:38-156 raw
// primitive append).  Compression: one-shot zstd frames (lib/zstd_codec.cpp:
// 136-168).
#pragma once
#include <atomic>
#include <cstdint>
#include <memory>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

namespace tomahawk {

// include/tomahawk.h:47-51,66-68
extern const char     TWK_MAGIC[9];           // "TOMAHAWK\1"
extern const char     TWO_MAGIC[4];           // "TWO\1"
extern const char     TWK_EOF_HEX[33];        // first 32 chars of the EOF digest
constexpr uint64_t    TWK_INDEX_START_MARKER = 1954702206512158641ull;

// ---- byte buffer ---------------------------------------------------------
class ByteBuf {
public:
	std::vector<uint8_t> v;
	size_t rd = 0;
	template <class T> void put(const T& x) { const size_t o = v.size(); v.resize(o + sizeof(T)); std::memcpy(&v[o], &x, sizeof(T)); }
	void put_bytes(const void* p, size_t n) { const size_t o = v.size(); v.resize(o + n); if (n) std::memcpy(&v[o], p, n); }
	void put_str(const std::string& s) { put<uint32_t>((uint32_t)s.size()); put_bytes(s.data(), s.size()); } // buffer.cpp:410-414
	template <class T> bool get(T& x) { if (rd + sizeof(T) > v.size()) return false; std::memcpy(&x, &v[rd], sizeof(T)); rd += sizeof(T); return true; }
	bool get_bytes(void* p, size_t n) { if (rd + n > v.size()) return false; if (n) std::memcpy(p, &v[rd], n); rd += n; return true; }
	bool get_str(std::string& s) { uint32_t n; if (!get(n) || rd + n > v.size()) return false; s.assign((const char*)&v[rd], n); rd += n; return true; }
	size_t size() const { return v.size(); }
	size_t remaining() const { return v.size() - rd; }     // bytes not read yet: bounds every count a file declares
	void clear() { v.clear(); rd = 0; }
};

bool zstd_compress(const uint8_t* src, size_t n, std::vector<uint8_t>& dst, int level);
// The records' own zstd encoder (twk_repcodec.h): frames any zstd decoder reads, made by comparing every byte with the byte one
// record back - 2-3 x as fast as libzstd's level 1 on the .two blocks of a large cohort and 5-8 % larger.  On records whose
// values repeat (small cohorts) or that do not follow each other (a shuffled file) a general match search finds far more, so
// every frame is first tried on its leading 32 KiB, both ways, and goes through libzstd when the encoder's sample is more than
// a quarter larger.  TwoWriter::pack_block takes this path when the compression level is RECORD_CODEC_LEVEL + k, k the libzstd
// level of the fallback (`tomahawk calc --engine-option record_codec=1`; -k keeps its meaning).
enum : int { RECORD_CODEC_LEVEL = 1 << 20 };
inline bool is_record_codec(int level) { return level >= RECORD_CODEC_LEVEL - 1000; }
bool record_codec_compress(const uint8_t* src, size_t n, std::vector<uint8_t>& dst, uint32_t stride, int fallback_level);
bool zstd_decompress(const uint8_t* src, size_t n, std::vector<uint8_t>& dst, size_t n_uncompressed);
// The same into caller-owned memory of exactly n_uncompressed bytes (quiet: returns false on any mismatch).
bool zstd_decompress_into(const uint8_t* src, size_t n, uint8_t* dst, size_t n_uncompressed);
const char* zstd_version();

// ---- header (include/header.h:115-128,312-416; lib/header.cpp:330-363) ----
struct Contig {
	uint32_t idx = 0;
	std::string name, description;
	int64_t n_bases = 0;
	std::vector<std::pair<std::string, std::string>> extra;
};
struct Header {
	std::string fileformat = "##fileformat=VCFv4.2";
	std::string literals;
	std::vector<std::string> samples;
	std::vector<Contig> contigs;
	void serialize(ByteBuf& b) const;
	bool deserialize(ByteBuf& b);
	int  contig_id(const std::string& name) const;
};

// ---- index (include/index.h:35-132; lib/index.cpp) ------------------------
struct IndexEntry {      // index.cpp:8-18 (40 bytes)
	int32_t  rid = 0;
	uint32_t n = 0, minpos = 0, maxpos = 0, b_unc = 0, b_cmp = 0;
	uint64_t foff = 0, fend = 0;
};
struct IndexEntryOutput : IndexEntry { int32_t ridB = -1; };  // index.cpp:41-52 (+4)
struct IndexEntryEntry {  // index.cpp:90-99: rid,n,minpos,maxpos,foff,fend,nn
	int32_t rid = 0; uint32_t n = 0, minpos = 0, maxpos = 0; uint64_t foff = 0, fend = 0, nn = 0;
};
struct TwkIndex {         // index.cpp:158-181
	std::vector<IndexEntry> ent;
	std::vector<IndexEntryEntry> meta;
	void serialize(ByteBuf& b) const;
	bool deserialize(ByteBuf& b);
};
struct TwoIndex {         // index.cpp:242-267
	uint8_t state = 0;    // 0 unsorted
	std::vector<IndexEntryOutput> ent;
	std::vector<IndexEntryEntry> meta;
	void serialize(ByteBuf& b) const;
	bool deserialize(ByteBuf& b);
};

// ---- .twk records (include/core.h:160-350; lib/core.cpp:59-101,245-261) ---
struct Variant {          // twk1_t
	uint8_t  gt_ptype = 1;      // bytes per RLE word: 1, 2 or 4
	bool     gt_flipped = false, gt_phase = false, gt_missing = false;
	uint8_t  alleles = 0;
	uint32_t pos = 0, ac = 0, an = 0, rid = 0, n_het = 0, n_hom = 0;
	double   hwe = 1.0;
	std::vector<uint32_t> runs; // RLE words, widened (core.h:195-198)
	uint32_t run_length(size_t i) const { return runs[i] >> (2 + 2 * (gt_missing ? 1 : 0)); }
	uint8_t  run_a(size_t i) const { const int m = gt_missing ? 1 : 0; return (runs[i] >> (1 + m)) & ((1u << (1 + m)) - 1); }
	uint8_t  run_b(size_t i) const { const int m = gt_missing ? 1 : 0; return runs[i] & ((1u << (1 + m)) - 1); }
	void serialize(ByteBuf& b) const;
	bool deserialize(ByteBuf& b);
	// RLE-encode a genotype vector (alleles[2*s], alleles[2*s+1] in {0,1,2}) the
	// way lib/genotype_encoder.h:277-343 does (minimum-cost word width :150-192,
	// run limit :17) and fill ac/an/n_het/n_hom/gt_missing.
	void encode(const int8_t* alleles, uint32_t n_samples, bool phased);
	// Expand to the reference bitvector layout (twk_igt_vec::Build,
	// lib/core.cpp:349-391).  mask may be nullptr.
	bool build_bitvector(uint32_t n_samples, uint64_t* data, uint64_t* mask) const;
};
struct Block {            // twk1_block_t (core.cpp:245-261)
	uint32_t rid = 0;
	std::vector<Variant> rcds;
	void serialize(ByteBuf& b) const;
	bool deserialize(ByteBuf& b);
};

// ---- .twk container (lib/importer.cpp:192-326, lib/twk_reader.cpp)
class TwkWriter {
public:
	bool open(const std::string& path, const Header& hdr, int c_level = 1);
	// Append a block (one contig per block, importer.cpp:192-260).
	bool write_block(const Block& blk);
	// The same with an explicit zstd level and index minpos (the importer's first block records
	// the 0-based position, importer.cpp:262-264; full blocks are compressed at level 10, :232).
	bool write_block(const Block& blk, int c_level, uint32_t minpos);
	// In two steps, so that blocks can be compressed on worker threads: pack() touches no writer
	// state; write_packed() appends in call order.
	struct Packed { std::vector<uint8_t> z; IndexEntry entry; };
	static bool pack(const Block& blk, int c_level, uint32_t minpos, Packed& out);
	bool write_packed(const Packed& p);
	bool close();
	uint64_t n_variants() const { uint64_t n = 0; for (const auto& e : index_.ent) n += e.n; return n; }
	size_t n_blocks() const { return index_.ent.size(); }
private:
	std::ofstream out_;
	TwkIndex index_;
	int c_level_ = 1;
};
class TwkReader {
public:
	Header hdr;
	TwkIndex index;
	bool open(const std::string& path);
	// Read block `i` of the index.
	bool read_block(size_t i, Block& blk);
	std::string error;
private:
	std::ifstream in_;
};

// ---- .two records / container --------------------------------
#pragma pack(push, 1)
struct TwoRecord {        // twk1_two_t serialised form, 106 bytes (core.cpp:470-490)
	uint16_t controller;
	uint32_t ridA, ridB;
	uint32_t packA, packB;   // pos << 2 | phased << 1 | miss
	double   cnt[4];
	double   D, Dprime, R, R2, P, ChiSqFisher, ChiSqModel;
	uint32_t Apos() const { return packA >> 2; }
	uint32_t Bpos() const { return packB >> 2; }
};
#pragma pack(pop)
static_assert(sizeof(TwoRecord) == 106, "twk1_two_t::packed_size");

class TwoWriter {         // include/writer.h:163-406 as used by calc
public:
	// path "-" or "" -> stdout (ld.cpp:585-587)
	bool open(const std::string& path, const Header& hdr, int c_level = 1);
	// One block: u32 n, u32 m, n records -> zstd -> [1][b_unc][b_cmp][bytes]
	// plus its IndexEntryOutput (ld_engine.cpp:1742-1802, writer.h:70-87).
	bool write_block(const TwoRecord* recs, uint32_t n);
	// The same in two steps so that compression can run on worker threads:
	// pack() is thread-safe and touches no writer state; write_packed() appends in call order.
	struct Packed { std::vector<uint8_t> z; IndexEntryOutput entry; uint32_t b_unc = 0; };
	static bool pack(const TwoRecord* recs, uint32_t n, int c_level, Packed& out);
	// the same from a block already laid out as it goes into the frame: u32 n, u32 n, n records
	static bool pack_block(const uint8_t* block, uint32_t n, int c_level, Packed& out);
	// The generic .two block of view -O b and sort (twk_two_writer_t::WriteBlockCompressedTWO,
	// writer.h:346-396): the index entry carries contigs and positions only in a sorted file.
	static bool pack_generic(const TwoRecord* recs, uint32_t n, int c_level, bool sorted, Packed& out);
	// Index state (0 unsorted, 2 sorted: index.h:103-105).  In a sorted file every block also
	// extends the per-contig entry of its ridA (writer.h:384-386, index.cpp:70-88).
	void set_state(uint8_t s) { index_.state = s; }
	bool write_packed(const Packed& p);
	// Mapped mode.  One thread appending through a stream moves ~5 GB/s into the page cache and that was what bound a
	// survivor-rich calc run (the reference has the same shape: every thread's flush goes through one spinlocked stream,
	// writer.h:70-87); positional pwrite()s from many threads are slower still (they serialise on the inode lock).  A shared
	// mapping has no such lock: map_output(), after open(), maps the file's block area window by window (1 GiB each, space
	// reserved with fallocate where the file system can, so that a full disk is an error here and not a SIGBUS later);
	// reserve() - called in the order the blocks are to appear, under the caller's lock - gives a frame its place and its
	// index entry, fill() - any thread, no lock - copies the frame there: the page faults and the copying run in parallel.
	// map_output() returns false, and the writer stays a stream, for stdout and for files that cannot be grown and mapped.
	struct Span { uint64_t off = 0; };
	bool map_output();
	bool mapped() const { return fd_ >= 0; }
	bool reserve(const Packed& p, Span& at);
	void fill(const Span& at, const Packed& p);
	// Direct mode (round 5).  With the blocks' compression cheap (record codec) the one stream into the file is what a
	// survivor-rich run waits for; direct_output(), after open(), takes the block frames out of the iostream: one pwritev()
	// per frame (head + payload) at the writer's own offset, with the file's space reserved 1 GiB ahead (fallocate, size kept)
	// where the file system can.  Same bytes as the stream; close() cuts the reservation and writes the index through the
	// stream again.  Returns false, and the writer stays a stream, for stdout and files that cannot be reopened.
	bool direct_output();
	bool direct() const { return dfd_ >= 0; }
	// Append an already compressed block under the given index entry (concat, lib/concat.h:160-175).
	bool write_raw(uint32_t b_unc, const std::vector<uint8_t>& z, IndexEntryOutput entry);
	int  compression_level() const { return c_level_; }
	bool close();             // writer.h:293-313
	uint64_t n_records = 0, n_blocks = 0;
private:
	std::ostream* os_ = nullptr;
	std::ofstream file_;
	uint64_t off_ = 0;        // bytes written so far (tellp is unusable on stdout)
	TwoIndex index_;
	int c_level_ = 1;
	bool put(const void* p, size_t n);
	void add_index_entry(const Packed& p, uint64_t foff, uint64_t fend);
	// mapped mode
	static constexpr uint64_t WIN_BYTES = 1ull << 30;
	static constexpr size_t MAX_WINDOWS = 1u << 14;                  // 16 TiB
	std::string path_;
	int fd_ = -1;
	int dfd_ = -1;                                                   // direct mode's descriptor
	uint64_t reserved_end_ = 0;                                      // file bytes below it are preallocated
	uint64_t mapped_end_ = 0;                                        // file bytes [0, mapped_end_) are reserved and mapped
	std::unique_ptr<std::atomic<uint8_t*>[]> win_;                   // window k = file bytes [k * WIN_BYTES, (k + 1) * WIN_BYTES)
	bool grow_to(uint64_t end);
	void unmap_all();
public:
	~TwoWriter();
	TwoWriter() = default;
	TwoWriter(const TwoWriter&) = delete;
	TwoWriter& operator=(const TwoWriter&) = delete;
};
class TwoReader {         // lib/two_reader.cpp:11-160
public:
	Header hdr;
	TwoIndex index;
	bool open(const std::string& path);
	// Next block of records; false at the end marker.
	bool next_block(std::vector<TwoRecord>& recs);
	// Next block as stored (still compressed): twk1_two_iterator::NextBlockRaw (two_reader.cpp:11-44).
	bool next_block_raw(uint32_t& b_unc, std::vector<uint8_t>& z);
	// Block at file offset `foff` (an index entry's foff) through a caller-owned stream, so that
	// worker threads can decode different blocks of one file at once.
	static bool read_block_at(std::ifstream& in, uint64_t foff, std::vector<TwoRecord>& recs);
	std::string error;
private:
	std::ifstream in_;
};

// Concatenate .two files of the same sample set by copying their compressed blocks and
// re-basing the index (lib/concat.h:63-251).  `note` is appended to the header literals.
bool two_concat(const std::vector<std::string>& inputs, const std::string& out, const std::string& note, std::string& error);

}  // namespace tomahawk
