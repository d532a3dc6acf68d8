// Small host utilities shared by the CLI-facing translation units: log stamps, pretty numbers,
// path pieces (reference lib/utility.cpp:68-144) and the command line that ends up in headers.
#pragma once
#include <cstdint>
#include <cstdio>
#include <ctime>
#include <iomanip>
#include <sstream>
#include <string>
#include <sys/time.h>

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

}  // namespace util
}  // namespace tomahawk
