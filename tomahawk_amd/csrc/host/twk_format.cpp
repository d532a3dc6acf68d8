// `.twk` / `.two` containers -- see twk_format.h for the reference citations.
#include "twk_format.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstring>
#include <iostream>
#include <memory>

#include "twk_repcodec.h"

// libzstd's stable one-shot API (zstd.h).  The image ships libzstd.so.1 but no
// system header, so the six prototypes we use are declared here.
extern "C" {
size_t ZSTD_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int compressionLevel);
size_t ZSTD_decompress(void* dst, size_t dstCapacity, const void* src, size_t compressedSize);
size_t ZSTD_compressBound(size_t srcSize);
typedef struct ZSTD_CCtx_s ZSTD_CCtx;
ZSTD_CCtx* ZSTD_createCCtx(void);
size_t ZSTD_freeCCtx(ZSTD_CCtx* cctx);
size_t ZSTD_compressCCtx(ZSTD_CCtx* cctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize, int compressionLevel);
unsigned ZSTD_isError(size_t code);
const char* ZSTD_getErrorName(size_t code);
const char* ZSTD_versionString(void);
unsigned long long ZSTD_getFrameContentSize(const void* src, size_t srcSize);
}

namespace tomahawk {

const char TWK_MAGIC[9]    = {'T','O','M','A','H','A','W','K','\1'};
const char TWO_MAGIC[4]    = {'T','W','O','\1'};
const char TWK_EOF_HEX[33] = "a4f54f39f5e251a6993796f48164ccf5";

// One compression context per thread, kept: ZSTD_compress() builds and frees one (its tables, ~1 MB at
// level 1) on every call, which costs as much as compressing a small block.
namespace {
struct CCtxHolder { ZSTD_CCtx* c = ZSTD_createCCtx(); ~CCtxHolder() { if (c) ZSTD_freeCCtx(c); } };
}
bool zstd_compress(const uint8_t* src, size_t n, std::vector<uint8_t>& dst, int level) {
	static thread_local CCtxHolder ctx;
	if (is_record_codec(level)) level -= RECORD_CODEC_LEVEL;      // (a writer opened with the records' codec: its header and index frames, its fallback)
	dst.resize(ZSTD_compressBound(n));
	const size_t r = ctx.c ? ZSTD_compressCCtx(ctx.c, dst.data(), dst.size(), src, n, level) : ZSTD_compress(dst.data(), dst.size(), src, n, level);
	if (ZSTD_isError(r)) { std::cerr << "[zstd] " << ZSTD_getErrorName(r) << std::endl; return false; }
	dst.resize(r);
	return true;
}
bool record_codec_compress(const uint8_t* src, size_t n, std::vector<uint8_t>& dst, uint32_t stride, int fallback_level) {
	static thread_local std::unique_ptr<repcodec::Work> work(new repcodec::Work);
	enum : size_t { SAMPLE = 32768 };
	// small blocks are not where the time goes; the frame header written by the encoder holds a 4-byte content size
	if (n < 4 * SAMPLE || (n >> 32)) return zstd_compress(src, n, dst, fallback_level);
	dst.resize(repcodec::bound(n));
	{                                                       // is this the kind of block the encoder is for?  Its head, both ways
		static thread_local std::vector<uint8_t> z;
		const size_t mine = repcodec::compress_frame(dst.data(), src, SAMPLE, stride, *work);
		if (!zstd_compress(src, SAMPLE, z, 1)) return false;
		if (4 * mine > 5 * z.size()) return zstd_compress(src, n, dst, fallback_level);
	}
	dst.resize(repcodec::compress_frame(dst.data(), src, n, stride, *work));
	return true;
}
bool zstd_decompress(const uint8_t* src, size_t n, std::vector<uint8_t>& dst, size_t n_unc) {
	// one-shot frames record their content size: a declared size that disagrees is a corrupt file,
	// found before anything is allocated for it
	if (ZSTD_getFrameContentSize(src, n) != (unsigned long long)n_unc) { std::cerr << "[zstd] decompress failed" << std::endl; return false; }
	dst.resize(n_unc);
	const size_t r = ZSTD_decompress(dst.data(), n_unc, src, n);
	if (ZSTD_isError(r) || r != n_unc) { std::cerr << "[zstd] decompress failed" << std::endl; return false; }
	return true;
}
bool zstd_decompress_into(const uint8_t* src, size_t n, uint8_t* dst, size_t n_unc) {
	if (ZSTD_getFrameContentSize(src, n) != (unsigned long long)n_unc) return false;
	const size_t r = ZSTD_decompress(dst, n_unc, src, n);
	return !ZSTD_isError(r) && r == n_unc;
}
const char* zstd_version() { return ZSTD_versionString(); }

// ---- Header ----------------------------------------------------------------
void Header::serialize(ByteBuf& b) const {
	b.put_str(fileformat);
	b.put_str(literals);
	b.put<uint32_t>((uint32_t)samples.size());
	for (const auto& s : samples) b.put_str(s);
	b.put<uint32_t>((uint32_t)contigs.size());
	for (const auto& c : contigs) {
		b.put<uint32_t>(c.idx); b.put_str(c.name); b.put_str(c.description); b.put<int64_t>(c.n_bases);
		b.put<uint32_t>((uint32_t)c.extra.size());
		for (const auto& e : c.extra) { b.put_str(e.first); b.put_str(e.second); }
	}
}
bool Header::deserialize(ByteBuf& b) {
	uint32_t n = 0;
	if (!b.get_str(fileformat) || !b.get_str(literals) || !b.get(n) || n > b.remaining() / 4) return false;
	samples.resize(n);
	for (auto& s : samples) if (!b.get_str(s)) return false;
	if (!b.get(n) || n > b.remaining() / 24) return false;
	contigs.resize(n);
	for (auto& c : contigs) {
		uint32_t ne = 0;
		if (!b.get(c.idx) || !b.get_str(c.name) || !b.get_str(c.description) || !b.get(c.n_bases) || !b.get(ne) || ne > b.remaining() / 8) return false;
		c.extra.resize(ne);
		for (auto& e : c.extra) if (!b.get_str(e.first) || !b.get_str(e.second)) return false;
	}
	return true;
}
int Header::contig_id(const std::string& name) const {
	for (size_t i = 0; i < contigs.size(); ++i) if (contigs[i].name == name) return (int)i;
	return -1;
}

// ---- Index -------------------------------------------------------------------
static void put_entry(ByteBuf& b, const IndexEntry& e) {
	b.put(e.rid); b.put(e.n); b.put(e.minpos); b.put(e.maxpos); b.put(e.b_unc); b.put(e.b_cmp); b.put(e.foff); b.put(e.fend);
}
static bool get_entry(ByteBuf& b, IndexEntry& e) {
	return b.get(e.rid) && b.get(e.n) && b.get(e.minpos) && b.get(e.maxpos) && b.get(e.b_unc) && b.get(e.b_cmp) && b.get(e.foff) && b.get(e.fend);
}
static void put_meta(ByteBuf& b, const IndexEntryEntry& e) {
	b.put(e.rid); b.put(e.n); b.put(e.minpos); b.put(e.maxpos); b.put(e.foff); b.put(e.fend); b.put(e.nn);
}
static bool get_meta(ByteBuf& b, IndexEntryEntry& e) {
	return b.get(e.rid) && b.get(e.n) && b.get(e.minpos) && b.get(e.maxpos) && b.get(e.foff) && b.get(e.fend) && b.get(e.nn);
}
void TwkIndex::serialize(ByteBuf& b) const {
	b.put<uint64_t>(TWK_INDEX_START_MARKER);
	b.put<uint64_t>(ent.size()); b.put<uint64_t>(std::max<size_t>(ent.size(), 1)); b.put<uint64_t>(meta.size());
	for (const auto& e : ent) put_entry(b, e);
	for (const auto& e : meta) put_meta(b, e);
}
bool TwkIndex::deserialize(ByteBuf& b) {
	uint64_t marker = 0, n = 0, m = 0, me = 0;
	if (!b.get(marker) || marker != TWK_INDEX_START_MARKER || !b.get(n) || !b.get(m) || !b.get(me)) return false;
	if (n > b.remaining() / 40 || me > b.remaining() / 40 || n * 40 + me * 40 > b.remaining()) return false;   // 40-byte entries
	ent.resize(n); meta.resize(me);
	for (auto& e : ent) if (!get_entry(b, e)) return false;
	for (auto& e : meta) if (!get_meta(b, e)) return false;
	return true;
}
void TwoIndex::serialize(ByteBuf& b) const {
	b.put<uint64_t>(TWK_INDEX_START_MARKER);
	b.put<uint8_t>(state);
	b.put<uint64_t>(ent.size()); b.put<uint64_t>(std::max<size_t>(ent.size(), 1)); b.put<uint64_t>(meta.size());
	for (const auto& e : ent) { put_entry(b, e); b.put(e.ridB); }
	for (const auto& e : meta) put_meta(b, e);
}
bool TwoIndex::deserialize(ByteBuf& b) {
	uint64_t marker = 0, n = 0, m = 0, me = 0;
	if (!b.get(marker) || marker != TWK_INDEX_START_MARKER || !b.get(state) || !b.get(n) || !b.get(m) || !b.get(me)) return false;
	if (n > b.remaining() / 44 || me > b.remaining() / 40 || n * 44 + me * 40 > b.remaining()) return false;   // 44- and 40-byte entries
	ent.resize(n); meta.resize(me);
	for (auto& e : ent) if (!get_entry(b, e) || !b.get(e.ridB)) return false;
	for (auto& e : meta) if (!get_meta(b, e)) return false;
	return true;
}

// ---- Variant -------------------------------------------------------------------
void Variant::serialize(ByteBuf& b) const { // core.cpp:59-73 + core.h:200-205
	const uint8_t pack = (uint8_t)(gt_ptype << 3 | (gt_flipped ? 4 : 0) | (gt_phase ? 2 : 0) | (gt_missing ? 1 : 0));
	b.put(pack); b.put(alleles); b.put(pos); b.put(ac); b.put(an); b.put(rid); b.put(n_het); b.put(n_hom); b.put(hwe);
	const uint32_t n_write = (uint32_t)runs.size() << 1 | (gt_missing ? 1 : 0);
	b.put(n_write);
	for (uint32_t r : runs) {
		if (gt_ptype == 1) b.put<uint8_t>((uint8_t)r);
		else if (gt_ptype == 2) b.put<uint16_t>((uint16_t)r);
		else b.put<uint32_t>(r);
	}
}
bool Variant::deserialize(ByteBuf& b) { // core.cpp:75-101
	uint8_t pack = 0;
	if (!b.get(pack)) return false;
	gt_ptype = pack >> 3; gt_flipped = (pack >> 2) & 1; gt_phase = (pack >> 1) & 1; gt_missing = pack & 1;
	if (!b.get(alleles) || !b.get(pos) || !b.get(ac) || !b.get(an) || !b.get(rid) || !b.get(n_het) || !b.get(n_hom) || !b.get(hwe)) return false;
	if (gt_ptype != 1 && gt_ptype != 2 && gt_ptype != 4) return false; // "illegal gt primitive type" core.cpp:95
	uint32_t n_write = 0;
	if (!b.get(n_write)) return false;
	const uint32_t n = n_write >> 1;
	// (n_write & 1) is the container's own miss bit; it equals gt_missing in valid files.
	gt_missing = n_write & 1;
	if (n > b.remaining() / gt_ptype) return false;
	runs.resize(n);
	for (uint32_t i = 0; i < n; ++i) {
		if (gt_ptype == 1) { uint8_t x; if (!b.get(x)) return false; runs[i] = x; }
		else if (gt_ptype == 2) { uint16_t x; if (!b.get(x)) return false; runs[i] = x; }
		else { uint32_t x; if (!b.get(x)) return false; runs[i] = x; }
	}
	return true;
}

void Variant::encode(const int8_t* al, uint32_t n_samples, bool phased) {
	bool miss = false;
	uint32_t cnt[3] = {0, 0, 0};
	n_het = n_hom = 0;
	for (uint32_t s = 0; s < n_samples; ++s) {
		const int a = al[2 * s], b = al[2 * s + 1];
		++cnt[a]; ++cnt[b];
		if (a == 2 || b == 2) miss = true;
		else if (a != b) ++n_het;
		else if (a == 1) ++n_hom;
	}
	gt_missing = miss; gt_phase = phased; gt_flipped = false;
	ac = cnt[1]; an = cnt[2];
	const int m = miss ? 1 : 0;
	// Count runs for each word width (genotype_encoder.h:150-192), pick min bytes.
	const uint32_t limit[3] = { (1u << (8 - 2 - 2 * m)) - 1, (1u << (16 - 2 - 2 * m)) - 1,
	                            (uint32_t)((1ull << (32 - 2 - 2 * m)) - 1) };
	uint32_t nruns[3] = {0, 0, 0};
	for (int w = 0; w < 3; ++w) {
		uint32_t len = 0; int ref = -1;
		for (uint32_t s = 0; s < n_samples; ++s) {
			const int cur = (al[2 * s] << (m + 1)) | al[2 * s + 1];
			if (cur != ref) { if (len) ++nruns[w]; ref = cur; len = 0; }
			if (len == limit[w]) { ++nruns[w]; len = 0; }
			++len;
		}
		if (len) ++nruns[w];
	}
	int best = 0; uint64_t cost = nruns[0];
	if ((uint64_t)nruns[1] * 2 < cost) { best = 1; cost = (uint64_t)nruns[1] * 2; }
	if ((uint64_t)nruns[2] * 4 < cost) { best = 2; }
	gt_ptype = (uint8_t)(1 << best);
	runs.clear(); runs.reserve(nruns[best]);
	uint32_t len = 0; int ref = -1;
	for (uint32_t s = 0; s < n_samples; ++s) {
		const int cur = (al[2 * s] << (m + 1)) | al[2 * s + 1];
		if (cur != ref) { if (len) runs.push_back(len << (2 + 2 * m) | (uint32_t)ref); ref = cur; len = 0; }
		if (len == limit[best]) { runs.push_back(len << (2 + 2 * m) | (uint32_t)ref); len = 0; }
		++len;
	}
	if (len) runs.push_back(len << (2 + 2 * m) | (uint32_t)ref);
}

bool Variant::build_bitvector(uint32_t n_samples, uint64_t* data, uint64_t* mask) const {
	const size_t n = ((size_t)2 * n_samples + 63) / 64;
	std::memset(data, 0, n * 8);
	if (mask) std::memset(mask, 0, n * 8);
	uint64_t cum = 0;
	for (size_t i = 0; i < runs.size(); ++i) {
		const uint64_t len = run_length(i);
		const uint8_t a = run_a(i), b = run_b(i);
		if (cum + 2 * len > (uint64_t)2 * n_samples) return false;
		if (a || b) {
			for (uint64_t j = 0; j < 2 * len; j += 2) {
				const uint64_t p0 = cum + j, p1 = p0 + 1;
				if (a == 1) data[p0 >> 6] |= 1ull << (p0 & 63);
				if (b == 1) data[p1 >> 6] |= 1ull << (p1 & 63);
				if ((a == 2 || b == 2) && mask) { mask[p0 >> 6] |= 1ull << (p0 & 63); mask[p1 >> 6] |= 1ull << (p1 & 63); }
			}
		}
		cum += 2 * len;
	}
	return cum == (uint64_t)2 * n_samples;
}

void Block::serialize(ByteBuf& b) const { // core.cpp:245-251
	const uint32_t n = (uint32_t)rcds.size();
	b.put(n); b.put<uint32_t>(std::max<uint32_t>(n, 1)); b.put(rid);
	for (const auto& r : rcds) r.serialize(b);
}
bool Block::deserialize(ByteBuf& b) {
	uint32_t n = 0, m = 0;
	if (!b.get(n) || !b.get(m) || !b.get(rid) || n > b.remaining() / 38) return false;
	rcds.resize(n);
	for (auto& r : rcds) if (!r.deserialize(b)) return false;
	return true;
}

// ---- .twk writer / reader --------------------------------------------------------
static bool write_header_frame(std::ostream& os, const Header& hdr, int level, uint64_t* written) {
	ByteBuf b; hdr.serialize(b);
	std::vector<uint8_t> z;
	if (!zstd_compress(b.v.data(), b.v.size(), z, level)) return false;
	const uint64_t unc = b.v.size(), cmp = z.size();
	os.write((const char*)&unc, 8); os.write((const char*)&cmp, 8); os.write((const char*)z.data(), z.size());
	if (written) *written += 16 + z.size();
	return os.good();
}

bool TwkWriter::open(const std::string& path, const Header& hdr, int c_level) {
	c_level_ = c_level;
	out_.open(path, std::ios::binary | std::ios::trunc);
	if (!out_.good()) return false;
	out_.write(TWK_MAGIC, 9);
	index_ = TwkIndex();
	index_.meta.resize(hdr.contigs.size());
	return write_header_frame(out_, hdr, c_level_, nullptr);
}
bool TwkWriter::write_block(const Block& blk) {
	if (blk.rcds.empty()) return true;
	return write_block(blk, c_level_, blk.rcds.front().pos + 1);
}
bool TwkWriter::write_block(const Block& blk, int c_level, uint32_t minpos) {
	if (blk.rcds.empty()) return true;
	Packed p;
	return pack(blk, c_level, minpos, p) && write_packed(p);
}
bool TwkWriter::pack(const Block& blk, int c_level, uint32_t minpos, Packed& out) {
	ByteBuf b; blk.serialize(b);
	if (b.size() >= (1ull << 32)) return false; // b_unc is u32 (core.cpp:266-271)
	if (!zstd_compress(b.v.data(), b.v.size(), out.z, c_level)) return false;
	IndexEntry& e = out.entry;
	e = IndexEntry();
	e.rid = (int32_t)blk.rid; e.n = (uint32_t)blk.rcds.size();
	e.minpos = minpos; e.maxpos = blk.rcds.back().pos + 1; // core.cpp:221-222
	e.b_unc = (uint32_t)b.size(); e.b_cmp = (uint32_t)out.z.size();
	return true;
}
bool TwkWriter::write_packed(const Packed& p) {
	IndexEntry e = p.entry;
	e.foff = (uint64_t)out_.tellp();
	const uint8_t marker = 1; const uint32_t unc = e.b_unc, cmp = e.b_cmp;
	out_.write((const char*)&marker, 1); out_.write((const char*)&unc, 4); out_.write((const char*)&cmp, 4);
	out_.write((const char*)p.z.data(), p.z.size());
	e.fend = (uint64_t)out_.tellp();
	index_.ent.push_back(e);
	if (e.rid >= 0 && (size_t)e.rid < index_.meta.size()) { // IndexEntryEntry::operator+= (index.cpp:69-86)
		IndexEntryEntry& m = index_.meta[e.rid];
		if (m.n == 0) { m.minpos = e.minpos; m.foff = e.foff; m.rid = e.rid; }
		m.n += e.n; m.maxpos = e.maxpos; m.fend = e.fend; ++m.nn;
	}
	return out_.good();
}
bool TwkWriter::close() { // importer.cpp:308-326
	ByteBuf b; index_.serialize(b);
	std::vector<uint8_t> z;
	if (!zstd_compress(b.v.data(), b.v.size(), z, c_level_)) return false;
	const uint64_t off = (uint64_t)out_.tellp(), unc = b.size(), cmp = z.size();
	const uint8_t marker = 0;
	out_.write((const char*)&marker, 1); out_.write((const char*)&unc, 8); out_.write((const char*)&cmp, 8);
	out_.write((const char*)z.data(), z.size());
	out_.write((const char*)&off, 8); out_.write(TWK_EOF_HEX, 32);
	out_.flush();
	const bool ok = out_.good();
	out_.close();
	return ok;
}

template <class IndexT>
static bool open_container(std::ifstream& in, const std::string& path, const char* magic, size_t nmagic,
                           Header& hdr, IndexT& index, std::string& error) {
	in.open(path, std::ios::binary | std::ios::ate);
	if (!in.good()) { error = "failed to open " + path; return false; }
	const uint64_t filesize = (uint64_t)in.tellg();
	in.seekg(0);
	char m[16];
	in.read(m, nmagic);
	if (!in.good() || std::memcmp(m, magic, nmagic) != 0) { error = "bad magic"; return false; }
	uint64_t unc = 0, cmp = 0;
	in.read((char*)&unc, 8); in.read((char*)&cmp, 8);
	if (!in.good() || cmp > filesize) { error = "bad header frame"; return false; }
	std::vector<uint8_t> z(cmp);
	in.read((char*)z.data(), cmp);
	ByteBuf b;
	if (!in.good() || !zstd_decompress(z.data(), cmp, b.v, unc) || !hdr.deserialize(b)) { error = "bad header"; return false; }
	const uint64_t data_start = (uint64_t)in.tellg();
	if (filesize < data_start + 40) { error = "truncated file"; return false; }
	in.seekg(filesize - 32 - 8);
	uint64_t off = 0;
	in.read((char*)&off, 8);
	char eof[32]; in.read(eof, 32);
	if (!in.good() || std::memcmp(eof, TWK_EOF_HEX, 32) != 0) { error = "missing EOF marker (truncated file?)"; return false; }
	if (off < data_start || off >= filesize) { error = "bad index offset"; return false; }
	in.seekg(off);
	uint8_t marker = 1;
	in.read((char*)&marker, 1); in.read((char*)&unc, 8); in.read((char*)&cmp, 8);
	if (!in.good() || marker != 0 || cmp > filesize) { error = "bad index frame"; return false; }
	z.resize(cmp);
	in.read((char*)z.data(), cmp);
	b.clear();
	if (!in.good() || !zstd_decompress(z.data(), cmp, b.v, unc) || !index.deserialize(b)) { error = "bad index"; return false; }
	in.seekg(data_start);
	return true;
}

bool TwkReader::open(const std::string& path) { return open_container(in_, path, TWK_MAGIC, 9, hdr, index, error); }

bool TwkReader::read_block(size_t i, Block& blk) { // twk_reader.cpp:8-44
	if (i >= index.ent.size()) return false;
	in_.clear();
	in_.seekg(index.ent[i].foff);
	uint8_t marker = 0; uint32_t unc = 0, cmp = 0;
	in_.read((char*)&marker, 1); in_.read((char*)&unc, 4); in_.read((char*)&cmp, 4);
	if (!in_.good() || marker != 1) { error = "bad block marker"; return false; }
	std::vector<uint8_t> z(cmp);
	in_.read((char*)z.data(), cmp);
	ByteBuf b;
	if (!in_.good() || !zstd_decompress(z.data(), cmp, b.v, unc) || !blk.deserialize(b)) { error = "bad block"; return false; }
	return true;
}

// ---- .two writer / reader ------------------------------------------------------------
bool TwoWriter::put(const void* p, size_t n) {
	if (direct()) {
		for (size_t done = 0; done < n;) {
			const ssize_t w = ::pwrite(dfd_, static_cast<const char*>(p) + done, n - done, (off_t)(off_ + done));
			if (w < 0) { if (errno == EINTR) continue; return false; }
			done += (size_t)w;
		}
		off_ += n;
		return true;
	}
	os_->write((const char*)p, n); off_ += n; return os_->good();
}

bool TwoWriter::open(const std::string& path, const Header& hdr, int c_level) {
	c_level_ = c_level; off_ = 0; n_records = n_blocks = 0; path_ = path;
	if (path.empty() || path == "-") os_ = &std::cout;
	else {
		file_.open(path, std::ios::binary | std::ios::trunc);
		if (!file_.good()) return false;
		os_ = &file_;
	}
	index_ = TwoIndex();
	index_.meta.resize(hdr.contigs.size()); // IndexOutput(n_contigs): all-zero meta (ld.cpp:621)
	if (!put(TWO_MAGIC, 4)) return false;
	uint64_t w = 0;
	if (!write_header_frame(*os_, hdr, c_level_, &w)) return false; // writer.h:225-242
	off_ += w;
	return true;
}

bool TwoWriter::pack(const TwoRecord* recs, uint32_t n, int c_level, Packed& out) {
	ByteBuf b;
	b.put<uint32_t>(n); b.put<uint32_t>(n);               // core.cpp:626-631 (n, m)
	b.put_bytes(recs, (size_t)n * sizeof(TwoRecord));
	return pack_block(b.v.data(), n, c_level, out);
}
bool TwoWriter::pack_block(const uint8_t* block, uint32_t n, int c_level, Packed& out) {
	const TwoRecord* recs = reinterpret_cast<const TwoRecord*>(block + 8);      // (packed struct: no alignment)
	const size_t bytes = 8 + (size_t)n * sizeof(TwoRecord);
	if (!(is_record_codec(c_level) ? record_codec_compress(block, bytes, out.z, (uint32_t)sizeof(TwoRecord), c_level - RECORD_CODEC_LEVEL) : zstd_compress(block, bytes, out.z, c_level))) return false;
	IndexEntryOutput& e = out.entry;                      // ld_engine.cpp:1270-1288,1757-1763
	e = IndexEntryOutput();
	e.rid = (int32_t)recs[0].ridA; e.ridB = (int32_t)recs[0].ridB;
	e.minpos = recs[0].Apos(); e.maxpos = recs[n - 1].Apos();
	for (uint32_t i = 1; i < n; ++i) if ((int32_t)recs[i].ridB != e.ridB) { e.ridB = -1; break; }
	e.n = n; e.b_unc = 106u * n + 8u; e.b_cmp = (uint32_t)out.z.size();
	out.b_unc = (uint32_t)bytes;
	return true;
}

bool TwoWriter::pack_generic(const TwoRecord* recs, uint32_t n, int c_level, bool sorted, Packed& out) {
	if (!pack(recs, n, c_level, out)) return false;
	if (!sorted) { out.entry.rid = -1; out.entry.ridB = -1; out.entry.minpos = 0; out.entry.maxpos = 0; }   // writer.h:377-382
	return true;
}

void TwoWriter::add_index_entry(const Packed& p, uint64_t foff, uint64_t fend) {
	IndexEntryOutput e = p.entry;
	e.foff = foff; e.fend = fend;
	index_.ent.push_back(e);
	if (index_.state == 2 && e.rid >= 0 && (size_t)e.rid < index_.meta.size()) {      // index.cpp:70-88
		IndexEntryEntry& m = index_.meta[e.rid];
		if (m.n == 0) { m.minpos = e.minpos; m.foff = e.foff; m.rid = e.rid; }
		m.n += e.n; m.maxpos = e.maxpos; m.fend = e.fend; ++m.nn;
	}
	n_records += e.n; ++n_blocks;
}

bool TwoWriter::write_packed(const Packed& p) {
	if (mapped()) { Span at; if (!reserve(p, at)) return false; fill(at, p); return true; }
	const uint64_t foff = off_;
	const uint8_t marker = 1; const uint32_t unc = p.b_unc, cmp = (uint32_t)p.z.size();
	if (direct()) {
		uint8_t head[9];
		head[0] = marker; std::memcpy(head + 1, &unc, 4); std::memcpy(head + 5, &cmp, 4);
		const uint64_t bytes = 9 + p.z.size();
		while (reserved_end_ < off_ + bytes) {               // space ahead of the writes: a hint only - where a gigabyte ahead cannot be had (a small
			// /dev/shm, a quota) the writes allocate as they go, and pwritev reports the ENOSPC that is real
			(void)::fallocate(dfd_, FALLOC_FL_KEEP_SIZE, (off_t)reserved_end_, (off_t)(1ull << 30));
			reserved_end_ += 1ull << 30;
		}
		struct iovec iov[2] = {{head, 9}, {const_cast<uint8_t*>(p.z.data()), p.z.size()}};
		uint64_t done = 0;
		while (done < bytes) {
			struct iovec part[2]; int n = 0;
			uint64_t skip = done;
			for (int k = 0; k < 2; ++k) {
				if (skip >= iov[k].iov_len) { skip -= iov[k].iov_len; continue; }
				part[n].iov_base = static_cast<uint8_t*>(iov[k].iov_base) + skip; part[n].iov_len = iov[k].iov_len - skip; ++n; skip = 0;
			}
			const ssize_t w = ::pwritev(dfd_, part, n, (off_t)(off_ + done));
			if (w < 0) { if (errno == EINTR) continue; return false; }
			done += (uint64_t)w;
		}
		off_ += bytes;
		add_index_entry(p, foff, off_);
		return true;
	}
	if (!put(&marker, 1) || !put(&unc, 4) || !put(&cmp, 4) || !put(p.z.data(), p.z.size())) return false;
	add_index_entry(p, foff, off_);
	return true;
}

// ---- mapped mode (twk_format.h) ----
TwoWriter::~TwoWriter() { unmap_all(); if (fd_ >= 0) ::close(fd_); if (dfd_ >= 0) ::close(dfd_); }

bool TwoWriter::direct_output() {
	if (direct()) return true;
	if (mapped() || !os_ || !file_.is_open() || path_.empty() || path_ == "-") return false;
	file_.flush();
	if (!file_.good()) return false;
	struct stat st;
	if (::stat(path_.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || (uint64_t)st.st_size != off_) return false;      // a regular file holding exactly what was written so far
	dfd_ = ::open(path_.c_str(), O_WRONLY);
	reserved_end_ = off_;
	return dfd_ >= 0;
}

void TwoWriter::unmap_all() {
	if (!win_) return;
	for (uint64_t k = 0; k * WIN_BYTES < mapped_end_; ++k)
		if (uint8_t* w = win_[k].load()) { ::munmap(w, WIN_BYTES); win_[k].store(nullptr); }
	mapped_end_ = 0;
}

bool TwoWriter::grow_to(uint64_t end) {
	while (mapped_end_ < end) {
		const uint64_t k = mapped_end_ / WIN_BYTES;
		if (k >= MAX_WINDOWS) return false;
		// space first: fallocate where the file system has it (an error now instead of a SIGBUS at the first touch of a page
		// the disk has no room for), else just the size
		if (::fallocate(fd_, 0, (off_t)mapped_end_, (off_t)WIN_BYTES) != 0) {
			if (errno != EOPNOTSUPP && errno != ENOSYS && errno != EINVAL) return false;
			if (::ftruncate(fd_, (off_t)(mapped_end_ + WIN_BYTES)) != 0) return false;
		}
		void* w = ::mmap(nullptr, WIN_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, (off_t)mapped_end_);
		if (w == MAP_FAILED) return false;
		win_[k].store(static_cast<uint8_t*>(w));
		mapped_end_ += WIN_BYTES;
	}
	return true;
}

bool TwoWriter::map_output() {
	if (mapped()) return true;
	if (!os_ || !file_.is_open() || path_.empty() || path_ == "-") return false;
	file_.flush();
	if (!file_.good()) return false;
	struct stat st;
	if (::stat(path_.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || (uint64_t)st.st_size != off_) return false;      // a regular file holding exactly what was written so far
	const int fd = ::open(path_.c_str(), O_RDWR);
	if (fd < 0) return false;
	fd_ = fd;
	win_.reset(new std::atomic<uint8_t*>[MAX_WINDOWS]);
	for (size_t k = 0; k < MAX_WINDOWS; ++k) win_[k].store(nullptr);
	mapped_end_ = 0;
	if (!grow_to(off_ + 1)) {               // the first window, which also holds the header: if this cannot be done, stay a stream
		unmap_all();
		if (::ftruncate(fd_, (off_t)off_) != 0) { /* the stream will overwrite what matters; the tail is cut at close */ }
		::close(fd_); fd_ = -1; win_.reset();
		return false;
	}
	return true;
}

bool TwoWriter::reserve(const Packed& p, Span& at) {
	if (!mapped()) return false;
	const uint64_t bytes = 9 + p.z.size();
	if (!grow_to(off_ + bytes)) return false;
	at.off = off_;
	off_ += bytes;
	add_index_entry(p, at.off, off_);
	return true;
}

void TwoWriter::fill(const Span& at, const Packed& p) {
	uint8_t head[9];
	head[0] = 1;
	const uint32_t unc = p.b_unc, cmp = (uint32_t)p.z.size();
	std::memcpy(head + 1, &unc, 4); std::memcpy(head + 5, &cmp, 4);
	auto copy = [&](uint64_t off, const uint8_t* src, size_t n) {          // a frame may straddle two windows
		while (n) {
			const uint64_t k = off / WIN_BYTES, in = off % WIN_BYTES;
			const size_t m = (size_t)std::min<uint64_t>(n, WIN_BYTES - in);
			std::memcpy(win_[k].load() + in, src, m);
			off += m; src += m; n -= m;
		}
	};
	copy(at.off, head, 9);
	copy(at.off + 9, p.z.data(), p.z.size());
}

bool TwoWriter::write_raw(uint32_t b_unc, const std::vector<uint8_t>& z, IndexEntryOutput e) {
	e.foff = off_;
	const uint8_t marker = 1; const uint32_t cmp = (uint32_t)z.size();
	if (!put(&marker, 1) || !put(&b_unc, 4) || !put(&cmp, 4) || !put(z.data(), z.size())) return false;
	e.fend = off_; e.b_cmp = cmp; e.b_unc = b_unc;      // concat.h:165-166 stores the frame sizes
	index_.ent.push_back(e);
	n_records += e.n; ++n_blocks;
	return true;
}

bool TwoWriter::write_block(const TwoRecord* recs, uint32_t n) {
	if (n == 0) return true;
	Packed p;
	return pack(recs, n, c_level_, p) && write_packed(p);
}

bool TwoWriter::close() { // writer.h:293-313
	if (!os_) return false;
	ByteBuf b; index_.serialize(b);
	std::vector<uint8_t> z;
	if (!zstd_compress(b.v.data(), b.v.size(), z, c_level_)) return false;
	const uint64_t off = off_, unc = b.size(), cmp = z.size();
	const uint8_t marker = 0;
	if (mapped()) {
		// the blocks are in place (every fill() has returned: the caller's business); cut the file to them and go on as a stream
		unmap_all();
		const bool cut = ::ftruncate(fd_, (off_t)off_) == 0;
		::close(fd_); fd_ = -1; win_.reset();
		file_.seekp((std::streamoff)off_);
		if (!cut || !file_.good()) { file_.close(); os_ = nullptr; return false; }
	}
	if (direct()) {
		// the frames are in the file; give back what was reserved beyond them and go on as a stream
		const bool cut = ::ftruncate(dfd_, (off_t)off_) == 0;
		::close(dfd_); dfd_ = -1;
		file_.seekp((std::streamoff)off_);
		if (!cut || !file_.good()) { file_.close(); os_ = nullptr; return false; }
	}
	put(&marker, 1); put(&unc, 8); put(&cmp, 8); put(z.data(), z.size()); put(&off, 8); put(TWK_EOF_HEX, 32);
	os_->flush();
	const bool ok = os_->good();
	if (file_.is_open()) file_.close();
	os_ = nullptr;
	return ok;
}

bool TwoReader::open(const std::string& path) { return open_container(in_, path, TWO_MAGIC, 4, hdr, index, error); }

bool TwoReader::next_block(std::vector<TwoRecord>& recs) { // two_reader.cpp:11-57
	uint8_t marker = 0;
	in_.read((char*)&marker, 1);
	if (!in_.good() || marker == 0) return false;
	if (marker != 1) { error = "bad block marker"; return false; }
	uint32_t unc = 0, cmp = 0;
	in_.read((char*)&unc, 4); in_.read((char*)&cmp, 4);
	std::vector<uint8_t> z(cmp);
	in_.read((char*)z.data(), cmp);
	ByteBuf b;
	if (!in_.good() || !zstd_decompress(z.data(), cmp, b.v, unc)) { error = "bad block"; return false; }
	uint32_t n = 0, m = 0;
	if (!b.get(n) || !b.get(m) || (size_t)n * sizeof(TwoRecord) + 8 > b.size()) { error = "bad block payload"; return false; }
	recs.resize(n);
	b.get_bytes(recs.data(), (size_t)n * sizeof(TwoRecord));
	return true;
}

bool TwoReader::read_block_at(std::ifstream& in, uint64_t foff, std::vector<TwoRecord>& recs) {
	in.clear();
	in.seekg((std::streamoff)foff);
	uint8_t marker = 0; uint32_t unc = 0, cmp = 0;
	in.read((char*)&marker, 1); in.read((char*)&unc, 4); in.read((char*)&cmp, 4);
	if (!in.good() || marker != 1) return false;
	std::vector<uint8_t> z(cmp);
	in.read((char*)z.data(), cmp);
	ByteBuf b;
	if (!in.good() || !zstd_decompress(z.data(), cmp, b.v, unc)) return false;
	uint32_t n = 0, m = 0;
	if (!b.get(n) || !b.get(m) || (size_t)n * sizeof(TwoRecord) + 8 > b.size()) return false;
	recs.resize(n);
	b.get_bytes(recs.data(), (size_t)n * sizeof(TwoRecord));
	return true;
}

bool TwoReader::next_block_raw(uint32_t& unc, std::vector<uint8_t>& z) {
	uint8_t marker = 0;
	in_.read((char*)&marker, 1);
	if (!in_.good() || marker == 0) return false;
	if (marker != 1) { error = "bad block marker"; return false; }
	uint32_t cmp = 0;
	in_.read((char*)&unc, 4); in_.read((char*)&cmp, 4);
	z.resize(cmp);
	in_.read((char*)z.data(), cmp);
	if (!in_.good()) { error = "truncated block"; return false; }
	return true;
}

bool two_concat(const std::vector<std::string>& inputs, const std::string& out, const std::string& note, std::string& error) {
	if (inputs.empty()) { error = "no input"; return false; }
	std::vector<std::unique_ptr<TwoReader>> rd;
	for (const auto& p : inputs) {
		rd.emplace_back(new TwoReader);
		if (!rd.back()->open(p)) { error = "Failed to open \"" + p + "\": " + rd.back()->error; return false; }
		if (rd.back()->hdr.samples != rd.front()->hdr.samples) { error = "Sample have different sample names or lengths: " + p; return false; }   // concat.h:141-154
		if (rd.back()->hdr.contigs.size() != rd.front()->hdr.contigs.size()) { error = "Files have different contigs: " + p; return false; }
	}
	Header hdr = rd.front()->hdr;
	hdr.literals += note;
	TwoWriter w;
	if (!w.open(out, hdr, 1)) { error = "failed to open " + out; return false; }
	for (auto& r : rd) {
		uint32_t unc = 0; std::vector<uint8_t> z; size_t k = 0;
		while (r->next_block_raw(unc, z)) {
			if (k >= r->index.ent.size()) { error = "index shorter than block stream"; return false; }
			if (!w.write_raw(unc, z, r->index.ent[k++])) { error = "write failed"; return false; }
		}
		if (!r->error.empty()) { error = r->error; return false; }
	}
	if (!w.close()) { error = "write failed"; return false; }
	return true;
}

}  // namespace tomahawk
