// The output record types a libtomahawk client can name after `#include "ld.h"`: tomahawk::twk1_two_t (one LD record)
// and tomahawk::twk1_two_block_t (a growable array of them) - reference include/core.h:756-834 and :851-902, which the
// reference's ld.h pulls in through core.h (include/ld.h:30-31).  Same member names, types and order, same flag setters,
// same ordering and the same 106-byte serialised form (lib/core.cpp:470-518), so client code that fills, sorts or reads
// records compiles unchanged; the text printers (PrintLD / PrintLDJson need the reference's VcfHeader) and the
// twk_buffer_t stream operators are not declared - the byte form is reached through pack() / unpack() below, which is what
// those operators do.  Header-only: nothing here needs the GPU libraries.
#ifndef TWK_TWO_TYPES_H_
#define TWK_TWO_TYPES_H_

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace tomahawk {

struct twk1_two_t {
	// bytes of one record inside a .two block: u16 + 2 x u32 (contigs) + 2 x u32 (packed positions) + 11 doubles
	const static uint32_t packed_size = sizeof(uint16_t) + 2 * sizeof(int32_t) + 2 * sizeof(uint32_t) + 11 * sizeof(double);

	twk1_two_t() { clear(); }

	double& operator[](const uint32_t& p) { return cnt[p]; }
	const double& operator[](const uint32_t& p) const { return cnt[p]; }

	// controller bits, in the reference's order (core.h:773-786)
	void SetUsedPhasedMath(const bool yes = true)    { controller |= yes << 0; }
	void SetSameContig(const bool yes = true)        { controller |= yes << 1; }
	void SetLongRange(const bool yes = true)         { controller |= yes << 2; }
	void SetCompleteLD(const bool yes = true)        { controller |= yes << 3; }
	void SetPerfectLD(const bool yes = true)         { controller |= yes << 4; }
	void SetMultipleRoots(const bool yes = true)     { controller |= yes << 5; }
	void SetFastMode(const bool yes = true)          { controller |= yes << 6; }
	void SetSampled(const bool yes = true)           { controller |= yes << 7; }
	void SetHasMissingValuesA(const bool yes = true) { controller |= yes << 8; }
	void SetHasMissingValuesB(const bool yes = true) { controller |= yes << 9; }
	void SetLowACA(const bool yes = true)            { controller |= yes << 10; }
	void SetLowACB(const bool yes = true)            { controller |= yes << 11; }
	void SetInvalidHWEA(const bool yes = true)       { controller |= yes << 12; }
	void SetInvalidHWEB(const bool yes = true)       { controller |= yes << 13; }

	void clear() {
		controller = 0; ridA = ridB = 0;
		Amiss = Aphased = Apos = 0; Bmiss = Bphased = Bpos = 0;
		R = R2 = D = Dprime = P = ChiSqModel = ChiSqFisher = 0;
		cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
	}

	// (ridA, ridB, Apos, Bpos): the order of `tomahawk sort` (core.cpp:458-468)
	bool operator<(const twk1_two_t& o) const {
		if (ridA != o.ridA) return ridA < o.ridA;
		if (ridB != o.ridB) return ridB < o.ridB;
		if (Apos != o.Apos) return Apos < o.Apos;
		return Bpos < o.Bpos;
	}

	// The record as it lies in a .two block (little endian, no padding; what `twk_buffer_t << twk1_two_t` appends).
	void pack(uint8_t* out) const {
		const uint32_t pa = (uint32_t)Apos << 2 | (uint32_t)Aphased << 1 | (uint32_t)Amiss;
		const uint32_t pb = (uint32_t)Bpos << 2 | (uint32_t)Bphased << 1 | (uint32_t)Bmiss;
		const double tail[11] = {cnt[0], cnt[1], cnt[2], cnt[3], D, Dprime, R, R2, P, ChiSqFisher, ChiSqModel};
		std::memcpy(out, &controller, 2); std::memcpy(out + 2, &ridA, 4); std::memcpy(out + 6, &ridB, 4);
		std::memcpy(out + 10, &pa, 4); std::memcpy(out + 14, &pb, 4); std::memcpy(out + 18, tail, sizeof(tail));
	}
	void unpack(const uint8_t* in) {
		uint32_t pa, pb; double tail[11];
		std::memcpy(&controller, in, 2); std::memcpy(&ridA, in + 2, 4); std::memcpy(&ridB, in + 6, 4);
		std::memcpy(&pa, in + 10, 4); std::memcpy(&pb, in + 14, 4); std::memcpy(tail, in + 18, sizeof(tail));
		Amiss = pa & 1; Aphased = (pa >> 1) & 1; Apos = pa >> 2;
		Bmiss = pb & 1; Bphased = (pb >> 1) & 1; Bpos = pb >> 2;
		cnt[0] = tail[0]; cnt[1] = tail[1]; cnt[2] = tail[2]; cnt[3] = tail[3];
		D = tail[4]; Dprime = tail[5]; R = tail[6]; R2 = tail[7]; P = tail[8]; ChiSqFisher = tail[9]; ChiSqModel = tail[10];
	}

	uint16_t controller;
	uint32_t ridA, ridB;
	uint32_t Amiss: 1, Aphased: 1, Apos: 30;
	uint32_t Bmiss: 1, Bphased: 1, Bpos: 30;
	double R, R2, D, Dprime, P;
	double ChiSqModel;   // chi-squared of the 3x3 table (unphased math)
	double ChiSqFisher;  // chi-squared of the 2x2 table
	double cnt[4];       // REFREF, REFALT, ALTREF, ALTALT
};
static_assert(twk1_two_t::packed_size == 106, "a .two record is 106 bytes on disk (core.h:758-759)");
static_assert(offsetof(twk1_two_t, ridA) == 4 && offsetof(twk1_two_t, R) == 24 && offsetof(twk1_two_t, cnt) == 80 && sizeof(twk1_two_t) == 112,
              "in-memory layout of the reference's twk1_two_t on x86-64");

struct twk1_two_block_t {
	typedef twk1_two_block_t   self_type;
	typedef twk1_two_t         value_type;
	typedef value_type&        reference;
	typedef const value_type&  const_reference;
	typedef value_type*        pointer;
	typedef const value_type*  const_pointer;
	typedef std::ptrdiff_t     difference_type;
	typedef std::size_t        size_type;

	twk1_two_block_t() : n(0), m(0), rcds(nullptr) {}
	twk1_two_block_t(const uint32_t p) : n(0), m(p), rcds(new twk1_two_t[p]) {}
	~twk1_two_block_t() { delete[] rcds; }
	twk1_two_block_t(const twk1_two_block_t&) = delete;
	twk1_two_block_t& operator=(const twk1_two_block_t&) = delete;

	twk1_two_block_t& operator+=(const twk1_two_t& rec) { return Add(rec); }
	twk1_two_block_t& Add(const twk1_two_t& rec) {
		if (n == m) resize();
		rcds[n++] = rec;
		return *this;
	}

	// capacity p (never below the records held); without an argument: 500 records at first, then doubling
	void resize(const uint32_t p) {
		if (p < n) return;
		twk1_two_t* grown = new twk1_two_t[p];
		std::copy(rcds, rcds + n, grown);
		delete[] rcds;
		rcds = grown; m = p;
	}
	void resize(void) { resize(rcds ? m * 2 : 500); }
	void reserve(const uint32_t p) { resize(p); }

	const uint32_t& size(void) const { return n; }
	reference front(void) { return rcds[0]; }
	const_reference front(void) const { return rcds[0]; }
	reference back(void) { return rcds[n ? n - 1 : 0]; }
	const_reference back(void) const { return rcds[n ? n - 1 : 0]; }
	reference operator[](const uint32_t position) { return rcds[position]; }
	const_reference operator[](const uint32_t position) const { return rcds[position]; }
	reference at(const uint32_t position) { return rcds[position]; }
	const_reference at(const uint32_t position) const { return rcds[position]; }
	pointer start(void) { return rcds; }
	const_pointer start(void) const { return rcds; }
	pointer end(void) { return rcds + n; }
	const_pointer end(void) const { return rcds + n; }

	void reset() { n = 0; }
	void clear() { delete[] rcds; rcds = nullptr; n = m = 0; }
	bool Sort() { std::sort(start(), end()); return true; }

	// The block as it lies in a .two frame before compression: u32 n, u32 m, n packed records (core.cpp:626-631).
	size_t packed_bytes() const { return 8 + (size_t)n * twk1_two_t::packed_size; }
	void pack(uint8_t* out) const {
		std::memcpy(out, &n, 4); std::memcpy(out + 4, &m, 4);
		for (uint32_t i = 0; i < n; ++i) rcds[i].pack(out + 8 + (size_t)i * twk1_two_t::packed_size);
	}
	// -> false when `bytes` is too short for the counts it announces
	bool unpack(const uint8_t* in, size_t bytes) {
		uint32_t n_in, m_in;
		if (bytes < 8) return false;
		std::memcpy(&n_in, in, 4); std::memcpy(&m_in, in + 4, 4);
		if (m_in < n_in || bytes < 8 + (size_t)n_in * twk1_two_t::packed_size) return false;
		delete[] rcds;
		rcds = new twk1_two_t[m_in]; n = n_in; m = m_in;
		for (uint32_t i = 0; i < n; ++i) rcds[i].unpack(in + 8 + (size_t)i * twk1_two_t::packed_size);
		return true;
	}

	uint32_t n, m;
	twk1_two_t* rcds;
};

}  // namespace tomahawk
#endif
