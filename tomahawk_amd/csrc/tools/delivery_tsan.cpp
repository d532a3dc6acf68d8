// ThreadSanitizer harness for the delivery queue of region calls (csrc/hip/twk_delivery.h) with the device operations stubbed: `make tsan`
// builds this file with -fsanitize=thread and runs it.  A producer stages "launches" of ragged sizes through DeliveryQueue while the
// delivery thread hands them to a sink that is slower than the producer (back-pressure: the pool may never exceed its bound), with
// allocations that fail now and then (the producer then delivers the launch itself, behind what is queued), a copy that fails, a sink
// that fails, buffers that must be replaced by larger ones, reclaim() in the middle of a run; every configuration checks that the sink
// saw every launch once, whole, and in the order staged.  Any data race TSan sees fails the run.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../hip/twk_delivery.h"

typedef int (*Sink)(void* user, const void* recs, uint64_t n);
constexpr size_t REC = 104;

struct Seen { std::vector<uint64_t> first_words; std::vector<uint64_t> sizes; int fail_at = -1; int slow_us = 0; std::atomic<int> calls{0}; };
static int sink(void* user, const void* recs, uint64_t n) {
	Seen* s = static_cast<Seen*>(user);
	const int k = s->calls.fetch_add(1);
	if (s->slow_us) std::this_thread::sleep_for(std::chrono::microseconds(s->slow_us));
	uint64_t w; std::memcpy(&w, recs, 8);
	// every record of the launch carries the launch's number in its first word
	for (uint64_t i = 0; i < n; i += (n / 7 + 1)) { uint64_t x; std::memcpy(&x, static_cast<const char*>(recs) + i * REC, 8); if (x != w) return 2; }
	s->first_words.push_back(w); s->sizes.push_back(n);
	return k == s->fail_at ? 1 : 0;
}

struct StubOps {
	std::atomic<long long> live_bytes{0}, peak_bytes{0};
	std::atomic<int> live_buffers{0}, peak_buffers{0};
	int fail_alloc_every = 0, fail_copy_at = 0, n_alloc = 0, n_copy = 0;
	void* alloc(size_t bytes) {
		if (fail_alloc_every && ++n_alloc % fail_alloc_every == 0) return nullptr;
		char* p = static_cast<char*>(std::malloc(bytes + 16));
		if (!p) return nullptr;
		std::memcpy(p, &bytes, sizeof(bytes));
		const long long now = live_bytes.fetch_add((long long)bytes) + (long long)bytes;
		long long pk = peak_bytes.load(); while (now > pk && !peak_bytes.compare_exchange_weak(pk, now)) {}
		const int nb = live_buffers.fetch_add(1) + 1;
		int pb = peak_buffers.load(); while (nb > pb && !peak_buffers.compare_exchange_weak(pb, nb)) {}
		return p + 16;
	}
	void release(void* q) {
		char* p = static_cast<char*>(q) - 16;
		size_t bytes; std::memcpy(&bytes, p, sizeof(bytes));
		live_bytes.fetch_sub((long long)bytes); live_buffers.fetch_sub(1);
		std::free(p);
	}
	int copy_aside(void* dst, const void* src, size_t bytes) {
		if (fail_copy_at && ++n_copy == fail_copy_at) return -3;
		std::memcpy(dst, src, bytes);
		return 0;
	}
	int deliver(const void* recs, uint64_t n, Sink sk, void* user, char* err, size_t err_len) {
		if (sk(user, recs, n)) { std::snprintf(err, err_len, "stub sink failed"); return -1; }
		return 0;
	}
	void thread_begin() {}
};
typedef twk::DeliveryQueue<StubOps, Sink> Queue;

int main() {
	int bad = 0, configs = 0;
	std::vector<char> src((size_t)200000 * REC);
	uint64_t x = 88172645463325252ull;
	auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
	for (int max_buffers : {1, 2, 3}) for (int slow_us : {0, 300}) for (int fail_alloc_every : {0, 3}) for (int scenario = 0; scenario < 4; ++scenario) {
		// scenario 0: plain; 1: the sink fails at its 5th call; 2: the 4th copy aside fails; 3: reclaim() every 7 launches
		++configs;
		StubOps ops; ops.fail_alloc_every = fail_alloc_every; ops.fail_copy_at = scenario == 2 ? 4 : 0;
		Seen seen; seen.slow_us = slow_us; seen.fail_at = scenario == 1 ? 4 : -1;
		Queue q(REC);
		if (!q.begin(&ops, (size_t)max_buffers)) { std::fprintf(stderr, "no thread\n"); return 1; }
		std::vector<uint64_t> order; int rc = 0; uint64_t launch = 0; bool failed_here = false;      // (the failing call may fall to this thread: then the queue has no text for it)
		for (; launch < 40 && !rc; ++launch) {
			const uint64_t n = 1 + rnd() % (launch < 10 ? 3000 : 150000);      // (later launches outgrow the first buffers)
			for (uint64_t i = 0; i < n; ++i) std::memcpy(src.data() + i * REC, &launch, 8);
			rc = q.stage(src.data(), n, sink, &seen);
			if (rc == Queue::STAGE_DELIVER_YOURSELF) {      // as the engine does: behind what is queued, on this thread
				rc = q.drain();
				if (!rc && sink(&seen, src.data(), n)) { rc = -1; failed_here = true; }
			}
			if (!rc) order.push_back(launch);
			if (scenario == 3 && launch % 7 == 6) (void)q.reclaim();
		}
		const int erc = q.end();
		if (!rc) rc = erc;
		bool ok = true;
		if (ops.live_buffers.load() != 0 || ops.live_bytes.load() != 0) { ok = false; std::fprintf(stderr, "leak: %d buffers\n", ops.live_buffers.load()); }
		if (ops.peak_buffers.load() > max_buffers) { ok = false; std::fprintf(stderr, "pool bound broken: %d > %d\n", ops.peak_buffers.load(), max_buffers); }
		if (scenario == 0 || scenario == 3) {
			if (rc || seen.first_words != order || order.size() != 40) { ok = false; std::fprintf(stderr, "rc %d, sink saw %zu of %zu launches\n", rc, seen.first_words.size(), order.size()); }
		} else {
			if (!rc) { ok = false; std::fprintf(stderr, "a failure was swallowed\n"); }
			// what reached the sink before the failure did so once and in order
			for (size_t i = 0; i < seen.first_words.size(); ++i) if (seen.first_words[i] != i) { ok = false; std::fprintf(stderr, "order broken at %zu\n", i); break; }
			if (scenario == 1 && !failed_here && !q.error()[0]) { ok = false; std::fprintf(stderr, "no error text\n"); }
		}
		// the queue is usable again for the next call
		StubOps ops2; Seen seen2; 
		if (!q.begin(&ops2, 2) || q.stage(src.data(), 10, sink, &seen2) != 0 || q.end() != 0 || seen2.sizes.size() != 1 || seen2.sizes[0] != 10) { ok = false; std::fprintf(stderr, "queue not reusable\n"); }
		if (!ok) { ++bad; std::fprintf(stderr, "  (max_buffers %d, slow %d us, failing allocations every %d, scenario %d; waits %llu)\n", max_buffers, slow_us, fail_alloc_every, scenario, (unsigned long long)q.waits()); }
	}
	std::printf("delivery_tsan: %d configurations, %d bad\n", configs, bad);
	return bad ? 1 : 0;
}
