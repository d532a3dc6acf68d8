// Small host utilities shared by the CLI-facing translation units: log stamps, pretty numbers,
// path pieces (reference lib/utility.cpp:68-144) and the command line that ends up in headers.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <ctime>
#include <iomanip>
#include <sched.h>
#include <sstream>
#include <string>
#include <sys/time.h>
#include <thread>

#ifndef TWK_AMD_VERSION
#define TWK_AMD_VERSION "0.7.0-mi355x"
#endif

namespace tomahawk {

// Defined by the executable, as with the reference (lib/main.cpp:4).  A weak
// *reference*: clients that do not define it (ctypes, tests) get an empty string.
extern __attribute__((weak)) std::string LITERAL_COMMAND_LINE;

namespace util {

inline std::string command_line() {
	std::string* volatile p = &LITERAL_COMMAND_LINE;
	return p ? *p : std::string();
}
inline std::string datetime() {
	time_t t = time(nullptr);
	struct timeval tv; gettimeofday(&tv, nullptr);
	struct tm now; localtime_r(&t, &now);
	char buf[64];
	snprintf(buf, sizeof(buf), "%04u-%02u-%02u %02u:%02u:%02u,%03u", now.tm_year + 1900, now.tm_mon + 1,
	         now.tm_mday, now.tm_hour, now.tm_min, now.tm_sec, (unsigned)(tv.tv_usec / 1000));
	return std::string(buf, 23);
}
inline std::string stamp(const std::string& a) { return "[" + datetime() + "][" + a + "] "; }
inline std::string stamp(const std::string& a, const std::string& b) { return "[" + datetime() + "][" + a + "][" + b + "] "; }
inline std::string pretty(uint64_t v) {
	std::string s = std::to_string(v);
	for (int i = (int)s.size() - 3; i > 0; i -= 3) s.insert(i, ",");
	return s;
}
inline std::string base_path(const std::string& in) { const size_t f = in.find_last_of("/\\"); return f == std::string::npos ? std::string() : in.substr(0, f); }
inline std::string base_name(const std::string& in) { const size_t f = in.find_last_of("/\\"); return f == std::string::npos ? in : in.substr(f + 1); }
inline std::string extension(const std::string& in) {
	const std::string b = base_name(in);
	const size_t d = b.rfind('.');
	return d == std::string::npos ? std::string() : b.substr(d + 1);
}
inline std::string elapsed_string(double sec) {
	std::ostringstream o;
	const uint64_t s = (uint64_t)sec;
	if (s >= 3600) o << s / 3600 << "h";
	if (s >= 60) o << (s % 3600) / 60 << "m";
	o << std::fixed << std::setprecision(3) << (sec - (double)(s / 60 * 60)) << "s";
	return o.str();
}

// CPUs this process may really use: the hardware's threads, cut down to the scheduler affinity mask and to the
// container's CFS quota (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us).  Thread pools sized beyond it gain
// nothing and lose a lot: when the quota of a period is spent, *every* thread of the container is frozen until the next
// period - the one that feeds the GPU with launches included.  (The GPU boxes of this pool have 256 hardware threads and
// a quota of 16 CPUs: 16 emitter threads compress 3.95 GB in 6.3 CPU-seconds, 64 threads in 19.8-34 "CPU-seconds" of
// which most is waiting for the next period - profiles/r04_output_floor.txt.)
inline int usable_cpus() {
	static const int cached = [] {
		int n = (int)std::max(1u, std::thread::hardware_concurrency());
		cpu_set_t set;
		if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0) n = std::min(n, a); }
		auto read_two = [](const char* path, long long& a, long long& b) -> int {
			FILE* f = fopen(path, "r");
			if (!f) return 0;
			char w1[64] = {0}, w2[64] = {0};
			const int got = fscanf(f, "%63s %63s", w1, w2);
			fclose(f);
			if (got >= 1) a = (w1[0] == 'm') ? -1 : atoll(w1);      // "max"
			if (got >= 2) b = atoll(w2);
			return got;
		};
		// The quota may sit on the process's own cgroup or on any of its ancestors (a systemd slice, a container started with the host's
		// cgroup namespace): /proc/self/cgroup names the process's path below each hierarchy's mount, and the smallest quota from there
		// up to the root counts.  (Inside a cgroup namespace the path is "/" and the mount root is the container's own group.)
		auto apply = [&](long long quota, long long period) { if (quota > 0 && period > 0) n = (int)std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period)); };
		std::string v2_path = "/", v1_path = "/";
		if (FILE* f = fopen("/proc/self/cgroup", "r")) {
			char line[4096];
			while (fgets(line, sizeof(line), f)) {
				std::string l(line);
				while (!l.empty() && (l.back() == '\n' || l.back() == '\r')) l.pop_back();
				const size_t c1 = l.find(':'), c2 = c1 == std::string::npos ? c1 : l.find(':', c1 + 1);
				if (c2 == std::string::npos) continue;
				const std::string ctrl = l.substr(c1 + 1, c2 - c1 - 1), path = l.substr(c2 + 1);
				if (ctrl.empty()) v2_path = path;                                   // "0::/path" - the unified hierarchy
				else if (("," + ctrl + ",").find(",cpu,") != std::string::npos) v1_path = path;
			}
			fclose(f);
		}
		auto walk = [&](const std::string& mount, std::string path, bool v2) {
			for (;;) {
				const std::string dir = mount + (path == "/" ? std::string() : path);
				long long quota = -1, period = 0, dummy = 0;
				if (v2) { if (read_two((dir + "/cpu.max").c_str(), quota, period) >= 2) apply(quota, period); }
				else if (read_two((dir + "/cpu.cfs_quota_us").c_str(), quota, dummy) >= 1 && read_two((dir + "/cpu.cfs_period_us").c_str(), period, dummy) >= 1) apply(quota, period);
				if (path == "/" || path.empty()) break;
				const size_t slash = path.find_last_of('/');
				path = slash == 0 || slash == std::string::npos ? "/" : path.substr(0, slash);
			}
		};
		walk("/sys/fs/cgroup", v2_path, true);
		walk("/sys/fs/cgroup/cpu", v1_path, false);
		return std::max(1, n);
	}();
	return cached;
}

}  // namespace util
}  // namespace tomahawk
