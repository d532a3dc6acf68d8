# Build everything in-tree for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   lib/libtwk_hip.so       HIP kernels + C ABI (include/twk_hip.h)
#   lib/libtomahawk_amd.so  host C++: .twk/.two formats, tomahawk::twk_ld, test C API
#   bin/tomahawk            `tomahawk calc` CLI
#   oracle/liboracle.so     CPU oracle (tests only)   oracle/_ref/  compiled reference (tests/baseline only)
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
PKG      := tomahawk_amd
LIBDIR   := $(PKG)/lib
BINDIR   := $(PKG)/bin
ZSTD_LIB ?= /usr/lib/x86_64-linux-gnu/libzstd.so.1
ZLIB     ?= /usr/lib/x86_64-linux-gnu/libz.so.1

HIP_SRC  := $(PKG)/csrc/hip/twk_hip.hip
HIP_DEPS := $(wildcard $(PKG)/csrc/hip/*.h) include/twk_hip.h
# -ffp-contract=off: the pair statistics must round like the reference's SSE4.2 build (ld_math.hip.h)
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function

HOST_SRC := $(wildcard $(PKG)/csrc/host/*.cpp)
HOST_LIB_SRC := $(filter-out %/calc_main.cpp,$(HOST_SRC))
HOST_DEPS := $(wildcard $(PKG)/csrc/host/*.h) include/twk_hip.h
CXXFLAGS := -O2 -std=c++17 -fPIC -Wall -pthread -Iinclude -I$(PKG)/csrc/host

.PHONY: all hip host cli oracle tools clean asan asan-test tsan
all: hip host cli oracle

hip: $(LIBDIR)/libtwk_hip.so
$(LIBDIR)/libtwk_hip.so: $(HIP_SRC) $(HIP_DEPS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared $(HIP_SRC) -o $@

host: $(LIBDIR)/libtomahawk_amd.so
$(LIBDIR)/libtomahawk_amd.so: $(HOST_LIB_SRC) $(HOST_DEPS) $(LIBDIR)/libtwk_hip.so
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared $(HOST_LIB_SRC) -o $@ -L$(LIBDIR) -ltwk_hip $(ZSTD_LIB) $(ZLIB) -Wl,-rpath,'$$ORIGIN'
	ln -sf libtomahawk_amd.so $(LIBDIR)/libtomahawk.so      # the reference library's name (makefile:162-182), for -ltomahawk

cli: $(BINDIR)/tomahawk
$(BINDIR)/tomahawk: $(PKG)/csrc/host/calc_main.cpp $(LIBDIR)/libtomahawk_amd.so
	@mkdir -p $(BINDIR)
	$(CXX) $(CXXFLAGS) $< -o $@ -L$(LIBDIR) -ltomahawk_amd -ltwk_hip $(ZSTD_LIB) -Wl,-rpath,'$$ORIGIN/../lib'

oracle:
	$(MAKE) -C oracle all

tools: build/count_microbench build/valu_rate build/hbm_read_bw build/issue_test2 build/fisher_probe build/bank_test2 build/list_vs_dense build/probe_vs_dense build/shift64_probe build/issue_test3 build/lds_dma3_probe build/bitop3_probe build/sgpr_probe
build/%: $(PKG)/csrc/tools/%.hip $(HIP_DEPS)
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -Ibuild $< -o $@
# (the probe's instruction streams with every register fixed by hand are written by a script)
build/bitop3_probe: build/bitop3_probe_gen.h
build/bitop3_probe_gen.h: $(PKG)/csrc/tools/bitop3_probe_gen.py
	@mkdir -p build
	python3 $< > $@
build/sgpr_probe: build/sgpr_probe_gen.h
build/sgpr_probe_gen.h: $(PKG)/csrc/tools/sgpr_probe_gen.py
	@mkdir -p build
	python3 $< > $@

# AddressSanitizer + UBSan build of the host side (the GPU side cannot be sanitised on this pool): same sources,
# separate output directory; `make asan-test` runs the CPU test suite against it (python is not instrumented, so
# libasan is preloaded; leak checking is off because the interpreter and the HIP runtime never free their arenas).
ASAN_DIR := $(PKG)/lib_asan
SANFLAGS := -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined
asan: $(LIBDIR)/libtwk_hip.so
	@mkdir -p $(ASAN_DIR)
	$(CXX) $(CXXFLAGS) $(SANFLAGS) -shared $(HOST_LIB_SRC) -o $(ASAN_DIR)/libtomahawk_amd.so -L$(LIBDIR) -ltwk_hip $(ZSTD_LIB) $(ZLIB) -Wl,-rpath,'$$ORIGIN/../lib'
	$(CXX) $(CXXFLAGS) $(SANFLAGS) $(PKG)/csrc/host/calc_main.cpp -o $(ASAN_DIR)/tomahawk -L$(ASAN_DIR) -ltomahawk_amd -L$(LIBDIR) -ltwk_hip $(ZSTD_LIB) -Wl,-rpath,'$$ORIGIN:$$ORIGIN/../lib'
asan-test: asan
	LD_PRELOAD=$$($(CXX) -print-file-name=libasan.so):$$($(CXX) -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
	TWK_HOST_LIB=$(abspath $(ASAN_DIR))/libtomahawk_amd.so TWK_CLI=$(abspath $(ASAN_DIR))/tomahawk python -m pytest tests -x -q -m "not gpu"

# ThreadSanitizer runs: the record emitter (worker pool, ordered placing step, backlog, mapped output) - the format and emitter sources with
# -fsanitize=thread around csrc/tools/emitter_tsan.cpp - and the engine's delivery queue (csrc/hip/twk_delivery.h) with the device operations
# stubbed (csrc/tools/delivery_tsan.cpp)
tsan:
	@mkdir -p build
	$(CXX) -O1 -g -std=c++17 -pthread -fsanitize=thread -Iinclude -I$(PKG)/csrc/host $(PKG)/csrc/tools/emitter_tsan.cpp $(PKG)/csrc/host/twk_format.cpp \
		-o build/emitter_tsan $(ZSTD_LIB) $(ZLIB)
	TSAN_OPTIONS=halt_on_error=1 ./build/emitter_tsan
	$(CXX) -O1 -g -std=c++17 -pthread -fsanitize=thread $(PKG)/csrc/tools/delivery_tsan.cpp -o build/delivery_tsan
	TSAN_OPTIONS=halt_on_error=1 ./build/delivery_tsan

clean:
	rm -rf $(LIBDIR) $(BINDIR) $(ASAN_DIR) build
	$(MAKE) -C oracle clean
