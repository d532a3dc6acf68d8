// The output side of a calc run: survivor records (twk_hip_record, variant indices) from one or
// more producers -> forward + reverse twk1_two_t blocks -> one .two file.
//
// The reference keeps one forward and one reverse block per worker thread (ld_engine.h:321) and
// flushes both, forward first, into the one shared writer under its spinlock when the forward
// block is full or the contig pair of the next record differs from the block's first
// (ld_engine.cpp:1270-1281, CompressBlock :1804-1810, writer.h:70-87).  Same structure here with
// one producer per GPU: a TwoOutput is the shared writer, a RecordEmitter one producer's pair of
// open blocks.
//
// The survivors of a super-tile arrive together, in (idxA, idxB) order when they come from the
// engine (sorted on the device), so the emitter only cuts them into blocks by the flush rule; its
// worker threads expand the blocks into their forward and reverse forms straight out of the
// producer's buffer, and emit() returns as soon as that is done - the buffer is the engine's again.
// Compressing the blocks and putting them into the file in order goes on behind the producer's back, on
// the same workers.  The order is kept by a *placing* step that whichever worker finds the next block in
// line compressed runs, one block after the other: with a mapped output (TwoWriter::map_output) it only
// assigns the block's two frames their place in the file and their index entries - under the shared
// writer's lock, microseconds - and the frames are then copied into the mapping by the workers, in
// parallel (round 3's single writer thread moved 3.8 GB in 0.7 s of a 0.9 s run: one thread, one inode
// lock); with a stream output (stdout, a file that cannot be mapped) the placing step is the write
// itself.  At most `window` blocks are in flight.  (With 33 M survivors of a 2,504-sample run the
// producer thread spent 2.1 s of a 2.3 s run in here when every call compressed and wrote its own
// blocks before returning.)
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
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
	size_t n_variants = 0;             // length of rid / pos
	uint64_t n_records = 0;            // forward + reverse records written (under mu)
	uint64_t n_blocks = 0, bytes_packed = 0;   // blocks appended and their compressed size (under mu)
};

class RecordEmitter {
public:
	// backlog_bytes: how much expanded-but-not-yet-written output may wait in memory (the blocks between the producer and
	// the file) beyond the 6 blocks per worker that keep the workers busy.  The producer only waits until its records are
	// *expanded* out of its buffer - work the workers do before any compressing - so with room for a backlog a burst of
	// survivors costs the producer a copy, not the burst's compression.  Measured on the GPU box over the 2,504 x 531,500 run
	// (profiles/r04_band_sort_ab.txt): 0.25 - 2 GB change nothing (the run is bound by the compression itself, ~10 GB/s of
	// records on that host), 4 GB gain 0.2 s of 1.6 s for `-u` (13 M survivors of one launch no longer hold the launch loop
	// up) and lose 0.25 s of 0.55 s for `-p -w 4000000` (4 GB of freshly faulted memory) - hence none by default.
	RecordEmitter(TwoOutput& out, int n_workers, size_t backlog_bytes = 0)
	    : out_(out), n_workers_(std::max(1, std::min(n_workers, 64))), backlog_bytes_(backlog_bytes) {
		window_ = (size_t)n_workers_ * 6;
		slots_.resize(window_);
		for (int t = 0; t < n_workers_; ++t) th_.emplace_back([this] { worker(); });
	}
	~RecordEmitter() {
		{ std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
		cv_job_.notify_all();
		for (auto& t : th_) t.join();
	}
	RecordEmitter(const RecordEmitter&) = delete;
	RecordEmitter& operator=(const RecordEmitter&) = delete;

	// seconds the producer spent in emit(): ordering unsorted input / cutting + waiting for the expansion (and, at the end, for the drain)
	double t_sort = 0, t_blocks = 0;
	std::atomic<uint64_t> ns_expand{0}, ns_pack{0}, ns_write{0};   // worker / writer thread time, summed

	// Write the survivors recs[0..n) behind this producer's open block; final: close it too and wait
	// until everything is in the file.  presorted: the records are in (idxA, idxB) order already.
	bool emit(const twk_hip_record* recs, uint64_t n, bool final, bool presorted = false) {
		const auto t_begin = std::chrono::steady_clock::now();
		if (failed_.load()) return false;
		const uint32_t* rid = out_.rid;
		if (n && !presorted) {
			bool sorted = true;
			for (uint64_t i = 1; i < n && sorted; ++i)
				sorted = recs[i - 1].idxA < recs[i].idxA || (recs[i - 1].idxA == recs[i].idxA && recs[i - 1].idxB <= recs[i].idxB);
			if (!sorted) { order(recs, n); recs = sorted_.data(); }
		}
		const auto t_sorted = std::chrono::steady_clock::now();
		t_sort += std::chrono::duration<double>(t_sorted - t_begin).count();
		// the sequence is carry[0..nc) followed by recs[0..n)
		const uint64_t nc = carry_.size(), total = nc + n;
		auto at = [&](uint64_t i) -> const twk_hip_record& { return i < nc ? carry_[i] : recs[i - nc]; };
		// cuts by the flush rule: a block ends when it holds b_size records or the next record's
		// (ridA, ridB) differs from its first record's
		std::vector<uint64_t> cut{0};
		if (total) {
			if (single_contig()) {
				for (uint64_t i = out_.b_size; i < total; i += out_.b_size) cut.push_back(i);
			} else {
				uint32_t fa = rid[at(0).idxA], fb = rid[at(0).idxB];
				for (uint64_t i = 1; i < total; ++i) {
					const twk_hip_record& r = at(i);
					const uint32_t ra = rid[r.idxA], rb = rid[r.idxB];
					if (i - cut.back() == out_.b_size || ra != fa || rb != fb) { cut.push_back(i); fa = ra; fb = rb; }
				}
			}
		}
		// the block after the last cut stays open unless this is the end
		const size_t n_closed = total ? (final ? cut.size() : cut.size() - 1) : 0;
		if (final && total) cut.push_back(total);
		// hand the closed blocks to the workers; wait until they are out of the producer's buffer
		{
			std::unique_lock<std::mutex> lk(mu_);
			if (next_seq_ == 0 && n_closed) {       // first blocks: the window of blocks in flight, by the backlog it may hold (nobody is using the slots yet)
				const size_t per_block = 2 * (8 + (size_t)out_.b_size * sizeof(TwoRecord));
				window_ = std::max<size_t>((size_t)n_workers_ * 6, std::min<size_t>(backlog_bytes_ / std::max<size_t>(per_block, 1), 1u << 16));
				slots_.resize(window_);
			}
			expanding_ = 0;
			for (size_t b = 0; b < n_closed; ++b) {
				const uint64_t seq = next_seq_;
				cv_room_.wait(lk, [&] { return failed_.load() || seq < written_ + window_; });
				if (failed_.load()) break;
				Slot& s = slots_[seq % window_];
				const uint64_t lo = cut[b], hi = cut[b + 1];
				// the block's records: [lo, hi) of carry ++ recs
				s.src_a = lo < nc ? carry_.data() + lo : nullptr; s.n_a = lo < nc ? std::min(hi, nc) - lo : 0;
				s.src_b = hi > nc ? recs + (std::max(lo, nc) - nc) : nullptr; s.n_b = hi > nc ? hi - std::max(lo, nc) : 0;
				s.n = (uint32_t)(hi - lo); s.state = Slot::QUEUED;
				expand_jobs_.push_back(seq);
				++next_seq_; ++expanding_;
				cv_job_.notify_one();
			}
			// also on failure (disk full, a block that would not pack): the workers still hold pointers into the
			// producer's buffer until every queued block of this call is expanded - they count down whatever happened
			cv_expanded_.wait(lk, [&] { return expanding_ == 0; });
			if (failed_.load()) return false;
		}
		std::vector<twk_hip_record> next;
		if (!final && total) { next.reserve(total - cut.back()); for (uint64_t i = cut.back(); i < total; ++i) next.push_back(at(i)); }
		carry_.swap(next);
		if (final) {
			std::unique_lock<std::mutex> lk(mu_);
			cv_room_.wait(lk, [&] { return failed_.load() || written_ == next_seq_; });
		}
		t_blocks += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_sorted).count();
		return !failed_.load();
	}

private:
	struct Slot {
		enum State { FREE, QUEUED, EXPANDED, PACKED, PLACED, DONE } state = FREE;      // EXPANDED: out of the producer's buffer, waiting to be compressed; PLACED: its frames have their place in the file; DONE: and are there
		TwoWriter::Span at_f, at_v;
		const twk_hip_record* src_a = nullptr; const twk_hip_record* src_b = nullptr;   // the block = src_a[0..n_a) ++ src_b[0..n_b)
		uint64_t n_a = 0, n_b = 0;
		uint32_t n = 0;
		std::vector<uint8_t> f, v;        // the forward and the reverse block as they go into their frames
		TwoWriter::Packed pf, pv;
	};
	TwoOutput& out_;
	int n_workers_;
	size_t backlog_bytes_;
	size_t window_ = 6;
	std::vector<Slot> slots_;             // block seq lives in slot seq % window_
	std::vector<std::thread> th_;
	std::mutex mu_;
	std::condition_variable cv_job_, cv_expanded_, cv_room_;
	std::deque<uint64_t> expand_jobs_;    // under mu_: blocks to expand (they hold the producer up: taken first) ...
	std::deque<uint64_t> jobs_;           // ... expanded blocks to compress, and (with a mapped output) placed blocks to copy in
	uint64_t next_seq_ = 0, written_ = 0; // under mu_: blocks handed out / in the file (every block below written_ is)
	uint64_t placed_ = 0;                 // under mu_: blocks [0, placed_) have their place in the file
	bool placing_ = false;                // under mu_: some worker is running the placing step
	uint64_t expanding_ = 0;              // under mu_: blocks of the current emit() still reading the producer's buffer
	bool stop_ = false;
	std::atomic<bool> failed_{false};
	std::vector<twk_hip_record> carry_;   // records of the open block (< b_size), in order
	std::vector<twk_hip_record> sorted_;  // unsorted input, ordered
	int single_contig_ = -1;

	bool single_contig() {                // every variant on one contig: the flush rule only counts
		if (single_contig_ < 0) {
			single_contig_ = out_.n_variants ? 1 : 0;
			for (size_t i = 1; i < out_.n_variants; ++i) if (out_.rid[i] != out_.rid[0]) { single_contig_ = 0; break; }
		}
		return single_contig_ == 1;
	}

	// (row, col) order: with one producer the file is deterministic (the reference's order is thread-timing dependent)
	void order(const twk_hip_record* recs, uint64_t n) {
		par::Raw<par::SortKey> keys;
		keys.alloc(n);
		const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_workers_, n / 65536 + 1));
		auto run = [&](const std::function<void(int)>& f) {
			std::vector<std::thread> th;
			for (int t = 0; t < T; ++t) th.emplace_back(f, t);
			for (auto& x : th) x.join();
		};
		run([&](int t) {
			for (uint64_t i = n * (uint64_t)t / T, e = n * (uint64_t)(t + 1) / T; i < e; ++i)
				keys[i] = par::SortKey{0, (uint64_t)recs[i].idxA << 32 | recs[i].idxB, (uint32_t)i};
		});
		par::parallel_sort(keys, n_workers_);
		sorted_.resize(n);
		run([&](int t) {
			for (uint64_t i = n * (uint64_t)t / T, e = n * (uint64_t)(t + 1) / T; i < e; ++i) sorted_[i] = recs[keys[i].idx];
		});
	}

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

	void worker() {
		for (;;) {
			uint64_t seq;
			bool expand_it;
			{
				std::unique_lock<std::mutex> lk(mu_);
				cv_job_.wait(lk, [&] { return stop_ || !expand_jobs_.empty() || !jobs_.empty(); });
				expand_it = !expand_jobs_.empty();
				if (expand_it) { seq = expand_jobs_.front(); expand_jobs_.pop_front(); }
				else if (!jobs_.empty()) { seq = jobs_.front(); jobs_.pop_front(); }
				else return;                           // stop_
			}
			Slot& s = slots_[seq % window_];
			if (s.state == Slot::PLACED) { copy_in(s); continue; }      // (a slot's state only changes under mu_, and only this thread holds the job)
			const uint32_t m = s.n;
			if (expand_it) {
				s.f.resize(8 + (size_t)m * sizeof(TwoRecord)); s.v.resize(8 + (size_t)m * sizeof(TwoRecord));       // u32 n, u32 n, records (core.cpp:626-631)
				std::memcpy(s.f.data(), &m, 4); std::memcpy(s.f.data() + 4, &m, 4);
				std::memcpy(s.v.data(), &m, 4); std::memcpy(s.v.data() + 4, &m, 4);
				TwoRecord* f = reinterpret_cast<TwoRecord*>(s.f.data() + 8);
				TwoRecord* v = reinterpret_cast<TwoRecord*>(s.v.data() + 8);
				const auto w0 = std::chrono::steady_clock::now();
				for (uint64_t i = 0; i < s.n_a; ++i) expand(s.src_a[i], f[i], v[i]);
				for (uint64_t i = 0; i < s.n_b; ++i) expand(s.src_b[i], f[s.n_a + i], v[s.n_a + i]);
				ns_expand += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count();
				std::lock_guard<std::mutex> lk(mu_);
				if (--expanding_ == 0) cv_expanded_.notify_all();
				if (!expand_jobs_.empty()) {           // more blocks hold the producer up: this one waits its turn to be compressed
					s.state = Slot::EXPANDED;
					jobs_.push_back(seq);
					cv_job_.notify_one();
					continue;
				}
			}
			const auto w1 = std::chrono::steady_clock::now();
			const bool ok = TwoWriter::pack_block(s.f.data(), m, out_.c_level, s.pf) && TwoWriter::pack_block(s.v.data(), m, out_.c_level, s.pv);
			ns_pack += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w1).count();
			std::unique_lock<std::mutex> lk(mu_);
			if (!ok) fail_locked();
			s.state = Slot::PACKED;
			place_blocks(lk);
		}
	}
	// The placing step (CompressBlock's order, ld_engine.cpp:1804-1810: forward, then reverse, blocks in sequence).  Called
	// with mu_ held by a worker that has just marked a block PACKED; if the next block in line is compressed and nobody else
	// is placing, this worker places it and every compressed block behind it.
	void place_blocks(std::unique_lock<std::mutex>& lk) {
		if (placing_) return;
		placing_ = true;
		while (placed_ < next_seq_ && slots_[placed_ % window_].state == Slot::PACKED) {
			Slot& s = slots_[placed_ % window_];
			const uint64_t seq = placed_;
			lk.unlock();
			bool ok = true, copy_later = false;
			const auto w0 = std::chrono::steady_clock::now();
			if (!failed_.load()) {
				std::lock_guard<std::mutex> wl(out_.mu);
				out_.n_records += 2 * (uint64_t)s.n;
				out_.n_blocks += 2; out_.bytes_packed += s.pf.z.size() + s.pv.z.size();
				if (out_.writer.mapped()) { ok = out_.writer.reserve(s.pf, s.at_f) && out_.writer.reserve(s.pv, s.at_v); copy_later = ok; }
				else ok = out_.writer.write_packed(s.pf) && out_.writer.write_packed(s.pv);
			}
			ns_write += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count();
			lk.lock();
			if (!ok) fail_locked();
			++placed_;
			if (copy_later) { s.state = Slot::PLACED; jobs_.push_back(seq); cv_job_.notify_one(); }
			else { s.state = Slot::DONE; retire_locked(); }
		}
		placing_ = false;
	}
	void copy_in(Slot& s) {
		const auto w0 = std::chrono::steady_clock::now();
		out_.writer.fill(s.at_f, s.pf);
		out_.writer.fill(s.at_v, s.pv);
		ns_write += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count();
		std::lock_guard<std::mutex> lk(mu_);
		s.state = Slot::DONE;
		retire_locked();
	}
	void retire_locked() {                    // slots whose blocks are in the file become free, in order
		bool any = false;
		while (written_ < placed_ && slots_[written_ % window_].state == Slot::DONE) { slots_[written_ % window_].state = Slot::FREE; ++written_; any = true; }
		if (any) cv_room_.notify_all();
	}
	void fail_locked() {
		failed_.store(true);
		cv_room_.notify_all(); cv_expanded_.notify_all();
	}
};

// The hand-off between the thread that receives the engine's survivors and a RecordEmitter.  emit() holds its caller until
// the workers have taken the piece out of the caller's buffer - and when the workers are behind with their compressing, that
// is the speed of the compression (24 ms per 2^20 survivors on the GPU box), paid by the one thread that also keeps the device
// supplied with launches.  put() only copies the piece (3-4 ms) into a buffer of the queue's own and a second thread feeds
// the emitter, in order; at most `cap` such buffers (109 MB each) exist, and only as many as a burst ever needed.
// cap == 0: no queue (takes() is false, the caller feeds the emitter itself).
class RecordHandOff {
public:
	enum : uint64_t { PIECE = 1ull << 20 };        // records per buffer: what the engine hands over at most (twk_hip.h)
	RecordHandOff(RecordEmitter& emitter, size_t cap) : emitter_(emitter), cap_(cap) {
		if (cap_) feeder_ = std::thread([this] { feed(); });
	}
	~RecordHandOff() {
		if (feeder_.joinable()) {
			{ std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
			cv_.notify_all();
			feeder_.join();
		}
		for (auto& p : queue_) free(p.recs);
		for (auto* p : spare_) free(p);
	}
	RecordHandOff(const RecordHandOff&) = delete;
	RecordHandOff& operator=(const RecordHandOff&) = delete;

	double t_copy = 0;                            // seconds put() spent copying / waiting for a buffer
	bool takes(uint64_t n) const { return cap_ != 0 && n <= PIECE; }
	// Queue a copy of recs[0..n) (n <= PIECE); with `edit`, of the records it returns true for, as it leaves them.
	// -> false once the emitter has failed.
	bool put(const twk_hip_record* recs, uint64_t n) { return put_impl(recs, n, nullptr); }
	template <class Edit> bool put(const twk_hip_record* recs, uint64_t n, Edit edit) {
		const std::function<bool(twk_hip_record&)> f(edit);
		return put_impl(recs, n, &f);
	}
	// every piece handed over so far is with the emitter; -> false if the emitter failed
	bool drain() {
		if (!cap_) return true;
		std::unique_lock<std::mutex> lk(mu_);
		cv_.wait(lk, [&] { return queue_.empty() && !feeding_; });
		return !failed_.load();
	}

private:
	struct Piece { twk_hip_record* recs; uint64_t n; };
	RecordEmitter& emitter_;
	size_t cap_, made_ = 0;                     // buffers at most / allocated (under mu_)
	std::mutex mu_; std::condition_variable cv_;
	std::deque<Piece> queue_;                   // under mu_: copied pieces, in order
	std::vector<twk_hip_record*> spare_;        // under mu_: buffers not in use (last in, first out: the warm ones)
	bool feeding_ = false, stop_ = false;       // under mu_
	std::atomic<bool> failed_{false};
	std::thread feeder_;

	bool put_impl(const twk_hip_record* recs, uint64_t n, const std::function<bool(twk_hip_record&)>* edit) {
		if (failed_.load() || !takes(n)) return false;
		const auto t_in = std::chrono::steady_clock::now();
		twk_hip_record* buf = nullptr;
		{
			std::unique_lock<std::mutex> lk(mu_);
			cv_.wait(lk, [&] { return !spare_.empty() || made_ < cap_; });
			if (!spare_.empty()) { buf = spare_.back(); spare_.pop_back(); }
			else ++made_;
		}
		if (!buf) buf = static_cast<twk_hip_record*>(malloc(PIECE * sizeof(twk_hip_record)));
		if (!buf) {
			std::lock_guard<std::mutex> lk(mu_);
			--made_; failed_.store(true);
			return false;
		}
		uint64_t m = 0;
		if (edit) {
			for (uint64_t i = 0; i < n; ++i) {
				twk_hip_record r = recs[i];
				if ((*edit)(r)) buf[m++] = r;
			}
		} else {
			// four threads' worth of memcpy for a full piece (one thread moves ~10 GB/s on the GPU box's host)
			const int T = n >= (1u << 18) ? 4 : 1;
			std::vector<std::thread> th;
			for (int t = 1; t < T; ++t)
				th.emplace_back([=] { std::memcpy(buf + n * t / T, recs + n * t / T, (size_t)(n * (t + 1) / T - n * t / T) * sizeof(twk_hip_record)); });
			std::memcpy(buf, recs, (size_t)(n / T) * sizeof(twk_hip_record));
			for (auto& x : th) x.join();
			m = n;
		}
		{
			std::lock_guard<std::mutex> lk(mu_);
			queue_.push_back(Piece{buf, m});
		}
		cv_.notify_all();
		t_copy += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_in).count();
		return true;
	}
	void feed() {
		for (;;) {
			Piece p;
			{
				std::unique_lock<std::mutex> lk(mu_);
				cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
				if (queue_.empty()) return;
				p = queue_.front(); queue_.pop_front(); feeding_ = true;
			}
			const bool ok = !failed_.load() && emitter_.emit(p.recs, p.n, false, true);      // (the engine's survivors come sorted)
			std::lock_guard<std::mutex> lk(mu_);
			if (!ok) failed_.store(true);
			spare_.push_back(p.recs); feeding_ = false;
			cv_.notify_all();
		}
	}
};

}  // namespace tomahawk
