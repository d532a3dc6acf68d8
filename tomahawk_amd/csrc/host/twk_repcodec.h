// A zstd *encoder* for blocks of fixed-size records (RFC 8878 frames that any zstd decoder reads, the reference's
// ZSTDCodec::Decompress - lib/zstd_codec.cpp:156-168 - included), for the one thing CompressBlock (ld_engine.cpp:1742-1810)
// compresses: n x 106-byte twk1_two_t behind an 8-byte head.
//
// Why its own encoder: a survivor-rich run is bound by libzstd's level 1 on the host (~0.5 GB/s a core), and what level 1
// finds in these records is almost only this: byte runs that equal the *previous record's* bytes at the same place (contig
// ids, the A variant's position, the exponents and signs of neighbouring statistics).  So the match search collapses to one
// comparison per byte against the byte `stride` back - no hash table, no chain - and the literals (mantissa bytes, incompressible:
// Huffman gains 3 % on them) are stored raw.  Measured on the reference's own records (profiles/r05_rep_codec.txt).
//
// Format choices (all plain RFC 8878):
//   frame   magic, FHD = 0xA0 (single segment, 4-byte content size, no checksum, no dictionary), content size
//   blocks  <= 128 KiB of content each; Compressed_Block unless that would not be smaller, then Raw_Block
//   literals  Raw_Literals_Block (3-byte header)
//   sequences  literal-length and match-length codes through FSE tables made from the code histogram of the frame's first
//     block that has sequences (FSE_Compressed mode there, Repeat mode in the blocks behind it; every code keeps at least a
//     "less than one" slot, so any later block can be written with them); offset codes in RLE mode: every sequence of a block
//     has the same offset code - code(stride + 3) up to and including the first block that holds a sequence, "repeat offset 1"
//     (code 0, no extra bits) in all blocks behind it.  Repeat offset 1 means the previous offset only if the sequence has at
//     least one literal, so a match that would start a block, or follow another match directly, gives its first byte to the
//     literals.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace tomahawk {
namespace repcodec {

enum : uint32_t { BLOCK = 1u << 17, MIN_MATCH = 4, LL_SYMS = 36, ML_SYMS = 53, MAX_LOG = 9 };

inline uint32_t highbit(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }
// literal-length / match-length -> (code, number of extra bits); the extra bits are the value minus the code's baseline,
// which for these power-of-two ranges is the value's low bits (of match length - 3 for the match codes)
inline void ll_code(uint32_t v, uint32_t& code, uint32_t& nbits) {
	if (v < 16) { code = v; nbits = 0; return; }
	if (v < 24) { code = 16 + ((v - 16) >> 1); nbits = 1; return; }
	if (v < 32) { code = 20 + ((v - 24) >> 2); nbits = 2; return; }
	if (v < 48) { code = 22 + ((v - 32) >> 3); nbits = 3; return; }
	if (v < 64) { code = 24; nbits = 4; return; }
	const uint32_t hb = highbit(v);                        // 64..127 -> 25 (6 bits), 128..255 -> 26 (7 bits), ...
	code = 19 + hb; nbits = hb;
}
inline void ml_code(uint32_t b, uint32_t& code, uint32_t& nbits) {      // b = match length - 3
	if (b < 32) { code = b; nbits = 0; return; }
	if (b < 40) { code = 32 + ((b - 32) >> 1); nbits = 1; return; }      // baselines - 3: 32, 34, 36, 38, 40, 44, 48, 56, 64, 80, 96, 128, 256, ...
	if (b < 48) { code = 36 + ((b - 40) >> 2); nbits = 2; return; }
	if (b < 64) { code = 38 + ((b - 48) >> 3); nbits = 3; return; }
	if (b < 96) { code = 40 + ((b - 64) >> 4); nbits = 4; return; }
	if (b < 128) { code = 42; nbits = 5; return; }
	const uint32_t hb = highbit(b);                        // 128..255 -> 43 (7 bits), 256..511 -> 44 (8 bits), ...
	code = 36 + hb; nbits = hb;
}

// One FSE compression table (FSE_buildCTable's construction: spread by step (size >> 1) + (size >> 3) + 3, "less than one"
// symbols from the top) and its description as the sequences section carries it (FSE_writeNCount's bit layout).
struct CTable {
	uint32_t log = 0;
	uint16_t state[1 << MAX_LOG];
	int32_t dnb[64], dfs[64];
	uint8_t desc[96];
	uint32_t desc_bytes = 0;
};

// counts[0..n_sym) -> normalised counts that sum to 1 << log with a slot for every symbol: one slot each ("less than one", -1)
// and the rest in proportion to the counts, the rounding's remainder to the most frequent symbol.
inline void normalise(const uint32_t* counts, int n_sym, uint32_t log, int16_t* norm) {
	const uint32_t size = 1u << log, spare = size - (uint32_t)n_sym;
	uint64_t total = 0;
	for (int s = 0; s < n_sym; ++s) total += counts[s];
	uint32_t used = 0, best = 0;
	for (int s = 0; s < n_sym; ++s) {
		const uint32_t extra = total ? (uint32_t)((uint64_t)counts[s] * spare / total) : 0;
		norm[s] = (int16_t)(1 + extra); used += 1 + extra;
		if (counts[s] > counts[best]) best = (uint32_t)s;
	}
	norm[best] = (int16_t)(norm[best] + (int16_t)(size - used));
	for (int s = 0; s < n_sym; ++s) if (norm[s] == 1) norm[s] = -1;
}

inline void build_ctable(const int16_t* norm, int n_sym, uint32_t log, CTable& t) {
	const uint32_t size = 1u << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
	uint32_t cumul[65], high = size - 1;
	static thread_local uint8_t sym_of[1 << MAX_LOG];
	t.log = log;
	cumul[0] = 0;
	for (int s = 0; s < n_sym; ++s) {
		if (norm[s] == -1) { cumul[s + 1] = cumul[s] + 1; sym_of[high--] = (uint8_t)s; }
		else cumul[s + 1] = cumul[s] + (uint32_t)norm[s];
	}
	uint32_t pos = 0;
	for (int s = 0; s < n_sym; ++s)
		for (int k = 0; k < norm[s]; ++k) {
			sym_of[pos] = (uint8_t)s;
			pos = (pos + step) & mask;
			while (pos > high) pos = (pos + step) & mask;
		}
	for (uint32_t u = 0; u < size; ++u) t.state[cumul[sym_of[u]]++] = (uint16_t)(size + u);
	int32_t total = 0;
	for (int s = 0; s < n_sym; ++s) {
		const int c = norm[s];
		if (c == 0) { t.dnb[s] = (int32_t)(((log + 1) << 16) - (1u << log)); t.dfs[s] = 0; }
		else if (c == -1 || c == 1) { t.dnb[s] = (int32_t)((log << 16) - (1u << log)); t.dfs[s] = total - 1; ++total; }
		else {
			const uint32_t max_bits = log - highbit((uint32_t)(c - 1));
			t.dnb[s] = (int32_t)((max_bits << 16) - ((uint32_t)c << max_bits)); t.dfs[s] = total - c; total += c;
		}
	}
	// the description: 4 bits log - 5, then every symbol's count + 1 in a field that narrows as the remaining total shrinks
	// (no count is 0 here, so the zero-run flags of the format never occur)
	uint64_t acc = log - 5; uint32_t nb = 4; uint8_t* p = t.desc;
	int32_t remaining = (int32_t)size + 1, threshold = (int32_t)size; uint32_t bits = log + 1;
	for (int s = 0; s < n_sym && remaining > 1; ++s) {
		int32_t c = norm[s];
		const int32_t max = (2 * threshold - 1) - remaining;
		remaining -= c < 0 ? -c : c;
		++c;
		if (c >= threshold) c += max;
		acc |= (uint64_t)(uint32_t)c << nb; nb += bits; nb -= (c < max);
		while (remaining < threshold) { --bits; threshold >>= 1; }
		while (nb >= 8) { *p++ = (uint8_t)acc; acc >>= 8; nb -= 8; }
	}
	if (nb) *p++ = (uint8_t)acc;
	t.desc_bytes = (uint32_t)(p - t.desc);
}

// the backward bit stream of the sequences section: bits are appended at the low end of a 64-bit container and the bytes
// leave in memory order; the decoder starts at the last byte, below its highest set bit
struct BitOut {
	uint8_t* p;
	uint64_t acc = 0;
	uint32_t n = 0;
	explicit BitOut(uint8_t* at) : p(at) {}
	void add(uint32_t v, uint32_t bits) { acc |= ((uint64_t)v & ((1ull << bits) - 1)) << n; n += bits; }
	void flush() { std::memcpy(p, &acc, 8); p += n >> 3; acc >>= n & ~7u; n &= 7; }      // (little-endian host; 8 bytes of slack behind p)
	uint8_t* close() { add(1, 1); flush(); if (n) { *p++ = (uint8_t)acc; n = 0; } return p; }
};

struct Seq { uint32_t lit, match; };           // literals in front of the match, match length

// What a thread keeps while it encodes frames: the sequences and the equality mask of one block, the frame's two tables.
struct Work {
	Seq seqs[BLOCK / (MIN_MATCH + 1) + 2];
	uint64_t eq[BLOCK / 64 + 2];
	CTable ll, ml;
};

// Room a frame's destination needs: head + per block (3 header + content) when every block is stored raw, + what the last
// block may take while it is being tried (its literals and a sequences section that turns out not to pay: <= 18 bits of state
// for each of <= BLOCK / 5 sequences, two table descriptions) and the slack of the wide stores.
inline size_t bound(size_t n) { return 512 + n + 3 * (n / BLOCK + 1) + (BLOCK >> 1); }

// bit i of eq: src[lo + i] == src[lo + i - stride] (0 where lo + i < stride, and from n on)
inline void equality_mask(const uint8_t* src, size_t lo, uint32_t n, uint32_t stride, uint64_t* eq) {
	const uint32_t words = (n + 63) / 64;
	uint32_t i = 0;
	for (uint32_t w = 0; w < words; ++w, i += 64) {
		uint64_t m = 0;
		if (lo + i >= stride && i + 64 <= n) {
			const uint8_t* a = src + lo + i; const uint8_t* b = a - stride;
#if defined(__SSE2__)
			for (uint32_t k = 0; k < 4; ++k) {
				const __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i*>(a + 16 * k)), y = _mm_loadu_si128(reinterpret_cast<const __m128i*>(b + 16 * k));
				m |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(x, y)) << (16 * k);
			}
#else
			for (uint32_t k = 0; k < 8; ++k) {
				uint64_t x, y;
				std::memcpy(&x, a + 8 * k, 8); std::memcpy(&y, b + 8 * k, 8);
				x ^= y;                                      // 0x80 in every byte of x that is zero, exactly; then one bit per byte
				const uint64_t k7 = 0x7f7f7f7f7f7f7f7full;
				const uint64_t z = ~(((x & k7) + k7) | x | k7);
				m |= (((z >> 7) * 0x0102040810204080ull) >> 56) << (8 * k);
			}
#endif
		} else {
			for (uint32_t k = 0; k < 64 && i + k < n; ++k)
				if (lo + i + k >= stride && src[lo + i + k] == src[lo + i + k - stride]) m |= 1ull << k;
		}
		eq[w] = m;
	}
	eq[words] = 0;
}

// One block [lo, hi) of the frame's content src (matches look back `stride` bytes, also into the blocks in front).
// -> bytes written at dst (block header included).  rep_ready: a block in front holds sequences, so repeat offset 1 is
// `stride` and the tables are the decoder's already; set when this block is the first such.
inline size_t encode_block(uint8_t* dst, const uint8_t* src, size_t lo, size_t hi, uint32_t stride, bool last, bool& rep_ready, Work& w) {
	const uint32_t n = (uint32_t)(hi - lo);
	uint8_t* const head = dst;
	uint8_t* p = dst + 6;                                   // block header, literals header
	// ---- matches: maximal runs of equal bytes, at least MIN_MATCH long once a leading literal is taken off; the literals ----
	// ---- between them go straight to their place ---------------------------------------------------------------------
	equality_mask(src, lo, n, stride, w.eq);
	uint32_t n_seq = 0, anchor = 0, start = 0;
	{
		// the runs' boundaries are the bits of eq ^ (eq << 1): walked word by word, starts and ends alternating
		const uint32_t words = (n + 63) / 64;
		uint64_t carry = 0;
		bool in_run = false;
		for (uint32_t wd = 0; wd <= words; ++wd) {           // (eq[words] == 0 closes a run that reaches the block's end)
			const uint64_t m = w.eq[wd];
			uint64_t t = m ^ ((m << 1) | carry);
			carry = m >> 63;
			while (t) {
				const uint32_t pos = (wd << 6) + (uint32_t)__builtin_ctzll(t);
				t &= t - 1;
				if (!in_run) { start = pos; in_run = true; continue; }
				in_run = false;
				const uint32_t s = start == anchor ? start + 1 : start;      // no literal in front: the first byte becomes one
				if (pos >= s + MIN_MATCH) {
					const uint32_t lit = s - anchor;
					w.seqs[n_seq].lit = lit; w.seqs[n_seq].match = pos - s; ++n_seq;
					if (lit <= 16 && anchor + 16 <= n) std::memcpy(p, src + lo + anchor, 16); else std::memcpy(p, src + lo + anchor, lit);      // (slack behind p: bound())
					p += lit;
					anchor = pos;
				}
			}
		}
	}
	size_t body = n;
	if (n_seq) {
		std::memcpy(p, src + lo + anchor, n - anchor); p += n - anchor;
		const uint32_t lit_total = (uint32_t)(p - (dst + 6));
		const uint32_t lh = (lit_total << 4) | (3u << 2);       // Raw_Literals_Block, 20-bit size
		dst[3] = (uint8_t)lh; dst[4] = (uint8_t)(lh >> 8); dst[5] = (uint8_t)(lh >> 16);
		// ---- sequences section ----------------------------------------------------------------------------------------
		if (n_seq < 128) *p++ = (uint8_t)n_seq;
		else if (n_seq < 0x7F00) { *p++ = (uint8_t)((n_seq >> 8) + 128); *p++ = (uint8_t)n_seq; }
		else { *p++ = 255; *p++ = (uint8_t)(n_seq - 0x7F00); *p++ = (uint8_t)((n_seq - 0x7F00) >> 8); }
		const uint32_t of_value = rep_ready ? 1u : stride + 3u, of_code = highbit(of_value), of_extra = of_value - (1u << of_code);
		if (!rep_ready) {                                   // the frame's tables, from this block's codes
			uint32_t hl[LL_SYMS] = {0}, hm[ML_SYMS] = {0}, c, nb;
			for (uint32_t k = 0; k < n_seq; ++k) { ll_code(w.seqs[k].lit, c, nb); ++hl[c]; ml_code(w.seqs[k].match - 3, c, nb); ++hm[c]; }
			int16_t norm[64];
			normalise(hl, LL_SYMS, MAX_LOG, norm); build_ctable(norm, LL_SYMS, MAX_LOG, w.ll);
			normalise(hm, ML_SYMS, MAX_LOG, norm); build_ctable(norm, ML_SYMS, MAX_LOG, w.ml);
			*p++ = (uint8_t)((2u << 6) | (1u << 4) | (2u << 2));        // LL FSE_Compressed, OF RLE, ML FSE_Compressed
			std::memcpy(p, w.ll.desc, w.ll.desc_bytes); p += w.ll.desc_bytes;
			*p++ = (uint8_t)of_code;
			std::memcpy(p, w.ml.desc, w.ml.desc_bytes); p += w.ml.desc_bytes;
		} else {
			*p++ = (uint8_t)((3u << 6) | (1u << 4) | (3u << 2));        // LL Repeat, OF RLE, ML Repeat
			*p++ = (uint8_t)of_code;
		}
		const CTable& L = w.ll; const CTable& M = w.ml;
		BitOut out(p);
		uint32_t lc, lnb, mc, mnb, ll_state, ml_state;
		{                                                   // the last sequence initialises the states (FSE_initCState2) and leaves its extra bits
			const Seq& q = w.seqs[n_seq - 1];
			ml_code(q.match - 3, mc, mnb); ll_code(q.lit, lc, lnb);
			{ const uint32_t o = (uint32_t)(M.dnb[mc] + (1 << 15)) >> 16; const uint32_t v = (o << 16) - (uint32_t)M.dnb[mc]; ml_state = M.state[(int32_t)(v >> o) + M.dfs[mc]]; }
			{ const uint32_t o = (uint32_t)(L.dnb[lc] + (1 << 15)) >> 16; const uint32_t v = (o << 16) - (uint32_t)L.dnb[lc]; ll_state = L.state[(int32_t)(v >> o) + L.dfs[lc]]; }
			out.add(q.lit, lnb); out.add(q.match - 3, mnb); out.add(of_extra, of_code); out.flush();
		}
		for (uint32_t k = n_seq - 1; k-- > 0;) {
			const Seq& q = w.seqs[k];
			ml_code(q.match - 3, mc, mnb); ll_code(q.lit, lc, lnb);
			{ const uint32_t o = (ml_state + (uint32_t)M.dnb[mc]) >> 16; out.add(ml_state, o); ml_state = M.state[(int32_t)(ml_state >> o) + M.dfs[mc]]; }
			{ const uint32_t o = (ll_state + (uint32_t)L.dnb[lc]) >> 16; out.add(ll_state, o); ll_state = L.state[(int32_t)(ll_state >> o) + L.dfs[lc]]; }
			out.flush();                                      // (7 + 9 + 9 bits at most; then 7 + 16 + 16 + the offset's)
			out.add(q.lit, lnb); out.add(q.match - 3, mnb); out.add(of_extra, of_code); out.flush();
		}
		out.add(ml_state, M.log); out.add(ll_state, L.log);         // (the offset state of an RLE table has no bits)
		p = out.close();
		body = (size_t)(p - head) - 3;
	}
	uint32_t type = 2;
	if (!n_seq || body >= n) {                              // nothing found, or not smaller: the content as a raw block
		std::memcpy(head + 3, src + lo, n);
		body = n; type = 0;
	} else rep_ready = true;
	const uint32_t h = (last ? 1u : 0u) | (type << 1) | ((uint32_t)body << 3);
	head[0] = (uint8_t)h; head[1] = (uint8_t)(h >> 8); head[2] = (uint8_t)(h >> 16);
	return 3 + body;
}

// A whole frame.  dst must hold bound(n) bytes.  -> frame size.
inline size_t compress_frame(uint8_t* dst, const uint8_t* src, size_t n, uint32_t stride, Work& w) {
	uint8_t* p = dst;
	*p++ = 0x28; *p++ = 0xB5; *p++ = 0x2F; *p++ = 0xFD;
	*p++ = 0xA0;                                            // single segment, 4-byte Frame_Content_Size
	*p++ = (uint8_t)n; *p++ = (uint8_t)(n >> 8); *p++ = (uint8_t)(n >> 16); *p++ = (uint8_t)(n >> 24);
	bool rep_ready = false;
	if (n == 0) { *p++ = 1; *p++ = 0; *p++ = 0; return (size_t)(p - dst); }      // one empty raw block, last
	for (size_t lo = 0; lo < n; lo += BLOCK) {
		const size_t hi = lo + BLOCK < n ? lo + BLOCK : n;
		p += encode_block(p, src, lo, hi, stride, hi == n, rep_ready, w);
	}
	return (size_t)(p - dst);
}

}  // namespace repcodec
}  // namespace tomahawk
