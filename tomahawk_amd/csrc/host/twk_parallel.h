// Thread-level building blocks shared by the host tools (view, sort, calc's output path):
// an ordered block-parallel pipeline, uninitialised scratch arrays and a stable parallel key sort.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "twk_util.h"

namespace tomahawk {
namespace par {

// produce(i, slot) on T worker threads for i in [0, n), consume(i, slot) on the calling thread in
// order; at most `window` results are in flight.
template <class Slot>
inline bool ordered_parallel(size_t n, int T, const std::function<bool(size_t, Slot&)>& produce,
                      const std::function<bool(size_t, Slot&)>& consume) {
	if (n == 0) return true;
	const int hw = util::usable_cpus();
	// more threads than cores only starves the consumer; beyond ~64 the single consumer (file
	// write) or the memory system is the limit and extra workers cost more than they add
	T = std::max(1, std::min<int>(std::min(std::min(T, hw), 64), (int)std::min<size_t>(n, 1024)));
	const size_t window = (size_t)T * 4;
	const size_t wake_every = std::max<size_t>(1, window / 4);
	std::vector<Slot> slots(window);
	std::unique_ptr<std::atomic<int>[]> ready(new std::atomic<int>[window]);
	for (size_t i = 0; i < window; ++i) ready[i].store(0);
	std::atomic<size_t> next{0}, consumed{0};
	std::atomic<bool> failed{false};
	// Two condition variables, each with at most a handful of sleepers: workers that ran a whole
	// window ahead of the consumer (rare), and the one consumer.  No broadcast per item.
	std::mutex mu; std::condition_variable cv_ready, cv_room;
	std::atomic<int> room_waiters{0};
	auto worker = [&]() {
		for (;;) {
			const size_t i = next.fetch_add(1);
			if (i >= n || failed.load()) return;
			if (i >= consumed.load() + window) {
				std::unique_lock<std::mutex> lk(mu);
				++room_waiters;
				cv_room.wait(lk, [&] { return failed.load() || i < consumed.load() + window; });
				--room_waiters;
				if (failed.load()) return;
			}
			const bool ok = produce(i, slots[i % window]);
			if (!ok) failed.store(true);
			ready[i % window].store(1);      // seq_cst with the load of `consumed` below (store-load pairing with the consumer)
			if (i == consumed.load() || !ok) { std::lock_guard<std::mutex> lk(mu); cv_ready.notify_one(); }
			if (!ok) { cv_room.notify_all(); return; }
		}
	};
	std::vector<std::thread> th;
	for (int t = 0; t < T; ++t) th.emplace_back(worker);
	bool ok = true;
	for (size_t i = 0; i < n && ok; ++i) {
		if (!ready[i % window].load()) {
			std::unique_lock<std::mutex> lk(mu);
			cv_ready.wait(lk, [&] { return failed.load() || ready[i % window].load() != 0; });
		}
		if (failed.load()) { ok = false; break; }
		if (!consume(i, slots[i % window])) ok = false;
		ready[i % window].store(0);
		{
			std::lock_guard<std::mutex> lk(mu);      // orders `consumed` against a worker about to sleep
			consumed.store(i + 1);
		}
		// Workers that ran a whole window ahead are woken in batches: the worker holding item j sleeps
		// only while j >= consumed + window and is woken by consumed = j - window + wake_every at the
		// latest, long before the consumer needs item j.
		if (room_waiters.load() > 0 && (i + 1) % wake_every == 0) cv_room.notify_all();
	}
	if (!ok) failed.store(true);
	{ std::lock_guard<std::mutex> lk(mu); }
	cv_room.notify_all();
	for (auto& t : th) t.join();
	return ok && !failed.load();
}

// Large scratch arrays without the single-threaded zero fill of std::vector (6.8 GB of records for a
// 64 M record file): memory is first touched by the threads that fill it.
template <class T> struct Raw {
	std::unique_ptr<T[]> p; size_t n = 0;
	void alloc(size_t m) { p.reset(new T[m]); n = m; }
	size_t size() const { return n; }
	T* data() { return p.get(); }
	T* begin() { return p.get(); }
	T* end() { return p.get() + n; }
	T& operator[](size_t i) { return p[i]; }
	const T& operator[](size_t i) const { return p[i]; }
	void swap(Raw& o) { p.swap(o.p); std::swap(n, o.n); }
};

// (hi, lo) compared lexicographically; idx = position in the input.
struct SortKey {
	uint64_t hi, lo;       // (ridA, ridB), (Apos, Bpos): twk1_two_t::operator< (core.cpp:458-468)
	uint32_t idx;
	bool operator<(const SortKey& o) const { return hi < o.hi || (hi == o.hi && lo < o.lo); }
};
// Sort keys on T threads: T sorted chunks, then a partitioned T-way merge -- splitters from a sample
// cut the key space into T ranges, and every thread merges its range of all chunks into place in a
// second buffer.  Stable: equal keys keep their input order (chunks are in input order, ties in the
// merge go to the lower chunk, and equal keys never straddle a splitter).
inline void parallel_sort(Raw<SortKey>& k, int T) {
	const size_t n = k.size();
	T = std::max(1, std::min<int>(T, (int)(n / 8192 + 1)));
	T = std::min<int>(T, util::usable_cpus());
	if (T == 1) { std::stable_sort(k.begin(), k.end()); return; }
	std::vector<size_t> cut(T + 1);
	for (int t = 0; t <= T; ++t) cut[t] = n * (size_t)t / T;
	auto run = [&](const std::function<void(int)>& f) {
		std::vector<std::thread> th;
		for (int t = 0; t < T; ++t) th.emplace_back(f, t);
		for (auto& x : th) x.join();
	};
	run([&](int t) { std::stable_sort(k.begin() + cut[t], k.begin() + cut[t + 1]); });
	// splitters: 64 evenly spaced keys per chunk, sorted, T-quantiles
	std::vector<SortKey> sample;
	for (int t = 0; t < T; ++t) {
		const size_t len = cut[t + 1] - cut[t];
		for (size_t j = 0; j < 64 && len; ++j) sample.push_back(k[cut[t] + len * j / 64]);
	}
	std::sort(sample.begin(), sample.end());
	std::vector<SortKey> split(T - 1);
	for (int p = 1; p < T; ++p) split[p - 1] = sample[sample.size() * (size_t)p / T];
	// bounds[p][t] = first element of chunk t that belongs to partition >= p
	std::vector<std::vector<size_t>> bounds(T + 1, std::vector<size_t>(T));
	for (int t = 0; t < T; ++t) { bounds[0][t] = cut[t]; bounds[T][t] = cut[t + 1]; }
	run([&](int t) {
		for (int p = 1; p < T; ++p)
			bounds[p][t] = (size_t)(std::lower_bound(k.begin() + cut[t], k.begin() + cut[t + 1], split[p - 1]) - k.begin());
	});
	std::vector<size_t> out_off(T + 1, 0);
	for (int p = 0; p < T; ++p) { size_t m = 0; for (int t = 0; t < T; ++t) m += bounds[p + 1][t] - bounds[p][t]; out_off[p + 1] = out_off[p] + m; }
	Raw<SortKey> out; out.alloc(n);
	run([&](int p) {
		struct Head { SortKey key; int chunk; };
		auto after = [](const Head& a, const Head& b) { return b.key < a.key || (!(a.key < b.key) && b.chunk < a.chunk); };
		std::vector<Head> heap;
		std::vector<size_t> pos(T), end(T);
		for (int t = 0; t < T; ++t) { pos[t] = bounds[p][t]; end[t] = bounds[p + 1][t]; if (pos[t] < end[t]) heap.push_back(Head{k[pos[t]], t}); }
		std::make_heap(heap.begin(), heap.end(), after);
		size_t o = out_off[p];
		while (!heap.empty()) {
			std::pop_heap(heap.begin(), heap.end(), after);
			Head h = heap.back();
			out[o++] = h.key;
			const int t = h.chunk;
			if (++pos[t] < end[t]) { heap.back() = Head{k[pos[t]], t}; std::push_heap(heap.begin(), heap.end(), after); }
			else heap.pop_back();
		}
	});
	k.swap(out);
}

}  // namespace par
}  // namespace tomahawk
