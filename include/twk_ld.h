// C++ API of the MI355X LD engine: the same class and settings struct a
// libtomahawk client uses (reference include/ld.h:40-69, include/core.h:909-924).
// A reference client switches by including this header (an `ld.h` shim that
// includes it is all a source tree needs) and linking
// libtomahawk_amd.so instead of libtomahawk.so.
#ifndef TWK_LD_H_
#define TWK_LD_H_

#include <cstdint>
#include <string>
#include <vector>

namespace tomahawk {

// Process-global command line, defined by the executable (reference
// lib/main.cpp:4, include/tomahawk.h:34) and copied into the .two header
// (lib/ld/ld.cpp:610-612).  libtomahawk_amd only holds a weak reference to it: an
// executable that does not define it simply records an empty command line.
extern std::string LITERAL_COMMAND_LINE;

// Unpacking selector bits kept for source compatibility (core.h:84-88); the
// GPU engine always builds dense bit-planes in HBM.
#define TWK_LDD_NONE   0
#define TWK_LDD_VEC    1
#define TWK_LDD_LIST   2
#define TWK_LDD_BITMAP 4
#define TWK_LDD_ALL  ((TWK_LDD_VEC) | (TWK_LDD_LIST) | (TWK_LDD_BITMAP))

// Field-for-field twk_ld_settings (core.h:909-924; defaults core.cpp:297-306).
struct twk_ld_settings {
	twk_ld_settings();
	std::string GetString() const;

	bool square, window, low_memory, bitmaps, single;
	bool force_phased, forced_unphased, force_cross_intervals;
	int32_t c_level, bl_size, b_size, l_window;
	int32_t n_threads, cycle_threshold, ldd_load_type;
	int32_t l_surrounding;
	std::string in, out;
	double minP, minR2, maxR2, minDprime, maxDprime;
	int32_t n_chunks, c_chunk;
	std::vector<std::string> ival_strings;
};

class twk_ld {
public:
	twk_ld();
	~twk_ld();
	twk_ld(const twk_ld&) = delete;
	twk_ld& operator=(const twk_ld&) = delete;

	void operator=(const twk_ld_settings& s) { settings = s; }

	// ld.h:50-61.  Returns true on success; diagnostics go to std::cerr.
	bool Compute(const twk_ld_settings& settings);
	bool Compute();
	bool ComputeSingle(const twk_ld_settings& settings, bool verbose = false, bool progress = true);
	bool ComputeSingle(bool verbose = false, bool progress = true);
	// The reference's compile-time-gated micro-benchmark (ld.cpp:878-1057): not
	// available, returns false like a reference build without TWK_SLAVE_DEBUG_MODE.
	bool ComputePerformance();

	// Not in the reference: a switch of the GPU engine by name, applied to every engine context this object creates
	// (twk_hip_set_option, include/twk_hip.h - measurement and test switches; none changes a record), plus two of this
	// class's own: "force_device" (every context on GPU n: several contexts on one GPU), "progress_ms" (period of
	// the progress line) "map_output" (1: the .two is written through a shared mapping by the emitter's workers instead of a stream) and "emit_workers" (threads that
	// expand, compress and place the output blocks, per GPU; default min(-t, 32)) and "emit_backlog_mb" (expanded blocks that may wait in memory for those threads, per GPU;
	// default 0: six blocks per thread) and "emit_queue_pieces" (buffers of 2^20 survivors between the engine's thread and the emitter, per GPU; default 8, 0: none) and "record_codec" (1: output blocks compressed by the records' own zstd encoder instead of libzstd at level -k; default 0) and "direct_output" (1: block frames written with pwritev() and space reserved ahead instead of through the iostream; default 0).  `tomahawk calc --engine-option key=value` ends here.  Nothing is read from the environment
	// except TWK_HIP_DEVICE / TWK_HIP_GPUS / TWK_HIP_PART (placement), TWK_REF_COMPAT and TWK_HIP_NO_SCREEN.
	void SetEngineOption(const std::string& key, int64_t value);

	// Results of the last Compute(): pairs compared / records written (both copies).
	uint64_t n_pairs() const;
	uint64_t n_records() const;

private:
	class twk_ld_impl;
	twk_ld_settings settings;
	twk_ld_impl* mImpl;
};

}  // namespace tomahawk
#endif
