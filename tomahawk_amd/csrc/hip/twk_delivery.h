// The delivery queue of a region call (twk_hip.hip: option "async_delivery").
//
// A launch rich in survivors holds the thread that finishes it for as long as the output side needs for them, and while it does
// no launch is enqueued: the device idles (profiles/r05_delivery_thread.txt).  So the finished launch's sorted survivors are copied
// aside - into a staging buffer of this queue - and a second thread hands the staged launches to the sink, in the order they were
// staged, one call at a time, while the calling thread goes on with the launches.  Reference analogue: the per-thread output block
// a slave flushes into the shared writer (lib/ld/ld_engine.cpp:1270-1281, 1742-1802) - there the compute thread itself blocks on the
// writer's spinlock.
//
// Bounded: at most `max_buffers` staging buffers exist at a time (round 5 allocated a new one whenever all were busy: with a sink
// slower than the launches - the zstd-bound run this was built for - copies of up to 7.5 GB a launch piled up until the device was
// out of memory).  stage() waits for a buffer instead; a buffer that is too small is freed and replaced by one at least half again
// as large, so a call replaces a buffer a handful of times at most.  Where no buffer is to be had at all (allocation failure) the
// caller delivers the launch itself, behind whatever is queued (STAGE_DELIVER_YOURSELF).
//
// Backend-free: everything that touches the device is behind Ops, so that the queue's locking can run under ThreadSanitizer on
// the CPU with stub operations (csrc/tools/delivery_tsan.cpp, `make tsan`).  Ops provides
//     void* alloc(size_t bytes)                      nullptr: no memory
//     void  release(void* p)
//     int   copy_aside(void* dst, const void* src, size_t bytes)                 0 or an error code
//     int   deliver(const void* recs, uint64_t n, Sink sink, void* user, char* err, size_t err_len)     (on the delivery thread)
//     void  thread_begin()                           (on the delivery thread, before its first item)
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace twk {

template <class Ops, class Sink>
class DeliveryQueue {
public:
	enum { STAGE_OK = 0, STAGE_DELIVER_YOURSELF = 1 };      // (negative: an error code of Ops or of an earlier item)

	explicit DeliveryQueue(size_t record_bytes) : rec_bytes(record_bytes) {}
	~DeliveryQueue() { (void)end(); }
	DeliveryQueue(const DeliveryQueue&) = delete;
	DeliveryQueue& operator=(const DeliveryQueue&) = delete;

	bool active() const { return is_active; }

	// Start the thread for one call.  false: no thread to be had (the caller delivers itself, as with the option off).
	bool begin(Ops* o, size_t max_buffers_) {
		if (is_active) return true;
		ops = o; max_buffers = max_buffers_ ? max_buffers_ : 1;
		stop = false; first_rc.store(0); err[0] = 0; n_staged = 0; n_waits = 0; peak_buffers = 0;
		try { th = std::thread([this] { loop(); }); } catch (...) { return false; }
		is_active = true;
		return true;
	}

	// Copy n records at src aside and queue them for (sink, user).  May wait for a staging buffer (back-pressure).
	int stage(const void* src, uint64_t n, Sink sink, void* user) {
		if (const int rc = first_rc.load()) return rc;
		if (!n) return STAGE_OK;
		size_t at = SIZE_MAX;
		void* doomed = nullptr;
		{
			std::unique_lock<std::mutex> lk(mu);
			for (;;) {
				size_t free_small = SIZE_MAX;
				for (size_t k = 0; k < pool.size(); ++k) {
					if (pool[k].busy) continue;
					if (pool[k].cap >= n) { if (at == SIZE_MAX || pool[k].cap < pool[at].cap) at = k; }
					else if (free_small == SIZE_MAX || pool[k].cap > pool[free_small].cap) free_small = k;
				}
				if (at != SIZE_MAX) { pool[at].busy = true; break; }
				if (pool.size() < max_buffers) { pool.push_back(Buf{nullptr, 0, true}); at = pool.size() - 1; break; }       // a slot of its own: filled below
				if (free_small != SIZE_MAX) { at = free_small; pool[at].busy = true; doomed = pool[at].p; pool[at].p = nullptr; break; }      // too small: replaced below
				++n_waits;
				cv_free.wait(lk);            // every buffer is on its way to the sink: wait for one
				if (const int rc = first_rc.load()) return rc;
			}
			if (pool.size() > peak_buffers) peak_buffers = pool.size();
		}
		if (doomed) ops->release(doomed);
		void* dst;
		{
			std::unique_lock<std::mutex> lk(mu);
			dst = pool[at].p;
			if (!dst) {
				const uint64_t grown = pool[at].cap + pool[at].cap / 2;
				uint64_t cap = n + n / 8;
				if (cap < grown) cap = grown;
				if (cap < (1ull << 16)) cap = 1ull << 16;
				lk.unlock();
				dst = ops->alloc((size_t)cap * rec_bytes);
				lk.lock();
				if (!dst) {              // no room for a copy: the slot stays, empty, for a later launch to fill
					pool[at].cap = 0; pool[at].busy = false;
					cv_free.notify_all();
					return STAGE_DELIVER_YOURSELF;
				}
				pool[at].p = dst; pool[at].cap = cap;
			}
		}
		if (const int rc = ops->copy_aside(dst, src, (size_t)n * rec_bytes)) {
			std::lock_guard<std::mutex> lk(mu);
			pool[at].busy = false;
			cv_free.notify_all();
			return rc;
		}
		{
			std::lock_guard<std::mutex> lk(mu);
			q.push_back(Item{at, dst, n, sink, user});
			++n_staged;
		}
		cv_work.notify_one();
		return STAGE_OK;
	}

	// Everything staged so far has reached its sink -> the first failure, if any.
	int drain() {
		if (!is_active) return first_rc.load();
		std::unique_lock<std::mutex> lk(mu);
		cv_idle.wait(lk, [&] { return q.empty() && !handing_over; });
		return first_rc.load();
	}

	// drain, then give the idle staging buffers back (the caller is out of device memory) -> bytes released
	size_t reclaim() {
		(void)drain();
		std::vector<void*> gone; size_t bytes = 0;
		{
			std::lock_guard<std::mutex> lk(mu);
			for (auto& b : pool) if (!b.busy && b.p) { gone.push_back(b.p); bytes += (size_t)b.cap * rec_bytes; b.p = nullptr; b.cap = 0; }
		}
		for (void* p : gone) ops->release(p);
		return bytes;
	}

	// Join the thread (after it has handed over what is queued), free the buffers -> the first failure of the call.
	int end() {
		if (!is_active) return first_rc.load();
		{ std::lock_guard<std::mutex> lk(mu); stop = true; }
		cv_work.notify_all();
		th.join();
		is_active = false;
		for (auto& b : pool) if (b.p) ops->release(b.p);
		pool.clear(); q.clear();
		return first_rc.load();
	}

	const char* error() const { return err; }
	// measurement / tests: launches staged, times stage() had to wait for a buffer, most buffers alive at once - of the last call
	uint64_t staged() const { return n_staged; }
	uint64_t waits() const { return n_waits; }
	size_t peak() const { return peak_buffers; }

private:
	struct Item { size_t buf; const void* p; uint64_t n; Sink sink; void* user; };
	struct Buf { void* p; uint64_t cap; bool busy; };

	void loop() {
		ops->thread_begin();
		for (;;) {
			Item it;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_work.wait(lk, [&] { return stop || !q.empty(); });
				if (q.empty()) return;
				it = q.front(); q.pop_front(); handing_over = true;
			}
			int rc = 0;
			char local[256]; local[0] = 0;
			if (!first_rc.load()) rc = ops->deliver(it.p, it.n, it.sink, it.user, local, sizeof(local));
			std::lock_guard<std::mutex> lk(mu);
			if (rc && !first_rc.load()) { std::snprintf(err, sizeof(err), "%s", local[0] ? local : "the record sink failed"); first_rc.store(rc); }
			pool[it.buf].busy = false; handing_over = false;
			cv_free.notify_all();
			if (q.empty()) cv_idle.notify_all();
		}
	}

	const size_t rec_bytes;
	Ops* ops = nullptr;
	size_t max_buffers = 3;
	std::thread th;
	std::mutex mu;
	std::condition_variable cv_work, cv_idle, cv_free;
	std::deque<Item> q;              // under mu: in the order of the launches
	std::vector<Buf> pool;           // under mu
	bool stop = false, handing_over = false;      // under mu
	bool is_active = false;          // calling thread only
	std::atomic<int> first_rc{0};
	char err[256] = {0};             // written under mu by the delivery thread, read by the calling thread after drain() / end()
	uint64_t n_staged = 0, n_waits = 0; size_t peak_buffers = 0;
};

}  // namespace twk
