// How fast can this machine take the .two file?  Writes `gb` GB of incompressible blocks of `block_kb` KB to `path` four ways and
// prints GB/s: one thread write(); the same after fallocate(); T threads pwrite() at disjoint offsets (T = 4, 16); T threads
// memcpy into a MAP_SHARED mapping of the fallocated file.  (The placing step of twk_record_sink.h is the first of these.)
//   write_probe <path> [gb=2] [block_kb=512]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
	if (argc < 2) { fprintf(stderr, "write_probe <path> [gb] [block_kb]\n"); return 2; }
	const char* path = argv[1];
	const size_t total = (size_t)((argc > 2 ? atof(argv[2]) : 2.0) * (1ull << 30));
	const size_t block = (size_t)(argc > 3 ? atoi(argv[3]) : 512) << 10;
	const size_t n_blocks = total / block;
	std::vector<uint8_t> src(block * 8);
	uint64_t x = 88172645463325252ull;
	for (size_t i = 0; i < src.size(); i += 8) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; memcpy(&src[i], &x, 8); }
	auto report = [&](const char* what, double s) { printf("%-46s %6.2f GB/s (%.3f s)\n", what, total / s / 1e9, s); fflush(stdout); };
	auto fresh = [&](bool prealloc) {
		unlink(path);
		const int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
		if (fd < 0) { perror("open"); exit(1); }
		if (prealloc && posix_fallocate(fd, 0, (off_t)(n_blocks * block))) perror("fallocate");
		return fd;
	};
	for (int prealloc = 0; prealloc < 2; ++prealloc) {
		const int fd = fresh(prealloc);
		const double t = now();
		for (size_t b = 0; b < n_blocks; ++b) if (write(fd, &src[(b & 7) * block], block) != (ssize_t)block) { perror("write"); return 1; }
		report(prealloc ? "1 thread write() after fallocate" : "1 thread write()", now() - t);
		close(fd);
	}
	for (int threads : {4, 16}) for (int prealloc = 0; prealloc < 2; ++prealloc) {
		const int fd = fresh(prealloc);
		const double t = now();
		std::vector<std::thread> pool;
		for (int w = 0; w < threads; ++w) pool.emplace_back([&, w] {
			for (size_t b = w; b < n_blocks; b += threads) if (pwrite(fd, &src[(b & 7) * block], block, (off_t)(b * block)) != (ssize_t)block) perror("pwrite");
		});
		for (auto& th : pool) th.join();
		char name[96]; snprintf(name, sizeof(name), "%d threads pwrite()%s", threads, prealloc ? " after fallocate" : "");
		report(name, now() - t);
		close(fd);
	}
	for (int threads : {4, 16}) {
		const int fd = fresh(true);
		const double t = now();
		uint8_t* m = (uint8_t*)mmap(nullptr, n_blocks * block, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		if (m == MAP_FAILED) { perror("mmap"); return 1; }
		std::vector<std::thread> pool;
		for (int w = 0; w < threads; ++w) pool.emplace_back([&, w] { for (size_t b = w; b < n_blocks; b += threads) memcpy(m + b * block, &src[(b & 7) * block], block); });
		for (auto& th : pool) th.join();
		munmap(m, n_blocks * block);
		char name[96]; snprintf(name, sizeof(name), "%d threads memcpy into a mapping", threads);
		report(name, now() - t);
		close(fd);
	}
	unlink(path);
	return 0;
}
