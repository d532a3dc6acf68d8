// The output side of a calc run: survivor records (twk_hip_record, variant indices) from one or
// more producers -> forward + reverse twk1_two_t blocks -> one .two file.
//
// The reference keeps one forward and one reverse block per worker thread (ld_engine.h:321) and
// flushes both, forward first, into the one shared writer under its spinlock when the forward
// block is full or the contig pair of the next record differs from the block's first
// (ld_engine.cpp:1270-1281, CompressBlock :1804-1810, writer.h:70-87).  Same structure here with
// one producer per GPU: a TwoOutput is the shared writer, a RecordEmitter one producer's pair of
// open blocks.  The survivors of a tile arrive together, so an emitter puts them in (row, col)
// order with a parallel key sort, cuts them into blocks by the flush rule, expands and compresses
// the blocks on worker threads and appends them in order; only the append holds the lock.
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "twk_format.h"
#include "twk_hip.h"
#include "twk_parallel.h"

namespace tomahawk {

struct TwoOutput {
	TwoWriter writer;
	std::mutex mu;                     // guards writer
	uint32_t b_size = 10000;           // records per block (twk_ld_settings.b_size)
	int c_level = 1;
	const uint32_t* rid = nullptr;     // per uploaded variant
	const uint32_t* pos = nullptr;
	uint64_t n_records = 0;            // forward + reverse records written (under mu)
};

class RecordEmitter {
public:
	RecordEmitter(TwoOutput& out, int n_workers) : out_(out), n_workers_(n_workers < 1 ? 1 : n_workers) {}

	// Write the survivors recs[0..n) (any order) behind this producer's open block; final: close it too.
	bool emit(const twk_hip_record* recs, uint64_t n, bool final) {
		const uint32_t* rid = out_.rid;
		// (row, col) order: with one producer the file is deterministic (the reference's order is
		// thread-timing dependent)
		par::Raw<par::SortKey> keys;
		keys.alloc(n);
		{
			const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_workers_, n / 65536 + 1));
			std::vector<std::thread> th;
			for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
				for (uint64_t i = n * (uint64_t)t / T, e = n * (uint64_t)(t + 1) / T; i < e; ++i)
					keys[i] = par::SortKey{0, (uint64_t)recs[i].idxA << 32 | recs[i].idxB, (uint32_t)i};
			});
			for (auto& x : th) x.join();
		}
		par::parallel_sort(keys, n_workers_);
		// the sequence is carry[0..nc) followed by recs[keys[.].idx]
		const uint64_t nc = carry_.size(), total = nc + n;
		auto at = [&](uint64_t i) -> const twk_hip_record& { return i < nc ? carry_[i] : recs[keys[i - nc].idx]; };
		// cuts by the flush rule: a block ends when it holds b_size records or the next record's
		// (ridA, ridB) differs from its first record's
		std::vector<uint64_t> cut{0};
		if (total) {
			uint32_t fa = rid[at(0).idxA], fb = rid[at(0).idxB];
			for (uint64_t i = 1; i < total; ++i) {
				const twk_hip_record& r = at(i);
				const uint32_t ra = rid[r.idxA], rb = rid[r.idxB];
				if (i - cut.back() == out_.b_size || ra != fa || rb != fb) { cut.push_back(i); fa = ra; fb = rb; }
			}
		}
		// the block after the last cut stays open unless this is the end
		const size_t n_closed = total ? (final ? cut.size() : cut.size() - 1) : 0;
		if (final && total) cut.push_back(total);
		struct Slot { std::vector<TwoRecord> f, v; TwoWriter::Packed pf, pv; };
		const int level = out_.c_level;
		std::function<bool(size_t, Slot&)> produce = [&](size_t b, Slot& s) -> bool {
			const uint64_t lo = cut[b], hi = cut[b + 1];
			s.f.resize(hi - lo); s.v.resize(hi - lo);
			for (uint64_t i = lo; i < hi; ++i) expand(at(i), s.f[i - lo], s.v[i - lo]);
			return TwoWriter::pack(s.f.data(), (uint32_t)s.f.size(), level, s.pf) && TwoWriter::pack(s.v.data(), (uint32_t)s.v.size(), level, s.pv);
		};
		std::function<bool(size_t, Slot&)> consume = [&](size_t, Slot& s) -> bool {   // CompressBlock (:1804-1810): forward, then reverse
			std::lock_guard<std::mutex> lk(out_.mu);
			out_.n_records += 2 * (uint64_t)s.f.size();
			return out_.writer.write_packed(s.pf) && out_.writer.write_packed(s.pv);
		};
		if (n_closed && !par::ordered_parallel<Slot>(n_closed, n_workers_, produce, consume)) return false;
		std::vector<twk_hip_record> next;
		if (!final && total) { next.reserve(total - cut.back()); for (uint64_t i = cut.back(); i < total; ++i) next.push_back(at(i)); }
		carry_.swap(next);
		return true;
	}

private:
	TwoOutput& out_;
	int n_workers_;
	std::vector<twk_hip_record> carry_;           // records of the open block (< b_size), in order

	void expand(const twk_hip_record& r, TwoRecord& f, TwoRecord& v) const {
		f.controller = (uint16_t)r.flags;
		f.ridA = out_.rid[r.idxA]; f.ridB = out_.rid[r.idxB];
		f.packA = out_.pos[r.idxA] << 2; f.packB = out_.pos[r.idxB] << 2;
		std::memcpy(f.cnt, r.cnt, sizeof(f.cnt));
		f.D = r.D; f.Dprime = r.Dprime; f.R = r.R; f.R2 = r.R2; f.P = r.P;
		f.ChiSqFisher = r.ChiSqFisher; f.ChiSqModel = r.ChiSqModel;
		v = f;                           // reverse copy swaps (rid,pos) only; cnt is NOT transposed (:1292-1298)
		v.ridA = f.ridB; v.ridB = f.ridA; v.packA = f.packB; v.packB = f.packA;      // (no references into the packed struct)
	}
};

}  // namespace tomahawk
