/*
 * twk_hip.h -- C ABI of the MI355X (gfx950) pairwise-LD engine.
 *
 * This is the drop-in boundary for the `tomahawk calc` hot path.  The
 * reference (mklarqvist/tomahawk) has no FFI of its own: its replacement point
 * is the per-thread block-pair loop `twk_ld_slave::Phased/Unphased`
 * (lib/ld/ld_engine.cpp:1898-2188) that calls one `twk_ld_engine` kernel per
 * variant pair (lib/ld/ld_engine.h:225, kernels ld_engine.cpp:185-1160) and
 * `PhasedMath`/`UnphasedMath` (ld_engine.cpp:1162-1740).  Each entry point
 * below names the reference interface it replaces.
 *
 * Conventions: plain C types only; every function returns 0 (TWK_HIP_OK) or a
 * negative TWK_HIP_E_* code and never throws; the caller owns every host
 * buffer; the library owns all device memory inside the ctx; one ctx per
 * device, used from one host thread at a time.  There is NO CPU fallback: if
 * no HIP device is usable the calls fail with TWK_HIP_E_DEVICE.
 */
#ifndef TWK_HIP_H_
#define TWK_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: twk_hip_timing grew (fused / carrier-list counters); new entry points twk_hip_set_device_sink,
 *    twk_hip_device_records, twk_hip_fisher_exact.  A caller built against another version must not pass
 *    its structs in: compare twk_hip_abi_version() with the header it was compiled with. */
/* 3: twk_hip_set_option / twk_hip_get_option; the library no longer reads environment variables. */
/* 4: twk_hip_timing grew (three-product launches); option "three". */
/* 5: twk_hip_generate_synthetic_planted / twk_synth_planted_bitvector / twk_synth_plant_source (synthetic input with planted LD pairs);
 *    twk_hip_option_describe; twk_hip_timing grew (finish_ms); twk_hip_gather_records / twk_hip_gather_backend /
 *    twk_hip_drain_device_sink (the RCCL gather of a one-process multi-GPU run). */
#define TWK_HIP_ABI_VERSION 5

enum {
	TWK_HIP_OK         =  0,
	TWK_HIP_E_INVALID  = -1, /* bad argument */
	TWK_HIP_E_NOMEM    = -2, /* host or device allocation failed */
	TWK_HIP_E_DEVICE   = -3, /* HIP runtime error / no device */
	TWK_HIP_E_OVERFLOW = -4, /* record buffer too small; *n_out holds the required count */
	TWK_HIP_E_STATE    = -5  /* call sequence error (e.g. compute before upload) */
};

/* Pair math selection; mirrors twk_ld_settings.force_phased / forced_unphased
 * (include/core.h:916) and the default per-pair rule of
 * twk_ld_slave::Calculate (ld_engine.cpp:2775,2803). */
enum {
	TWK_HIP_MODE_PHASED   = 1, /* calc -p : PhasedListVector/PhasedVectorized + PhasedMath   */
	TWK_HIP_MODE_UNPHASED = 2, /* calc -u : UnphasedVectorized(NoMissing) + UnphasedMath      */
	TWK_HIP_MODE_AUTO     = 3  /* calc    : unphased iff either variant has missing alleles   */
};

typedef struct twk_hip_ctx twk_hip_ctx;

/* Per-variant metadata: the fields of twk1_t (include/core.h:291-295) that the
 * pair loop and the math read (ac, an, pos, rid, hwe, gt_missing). */
typedef struct {
	uint32_t ac;      /* ALT allele count                       */
	uint32_t an;      /* number of MISSING alleles (sic)        */
	uint32_t pos;     /* 0-based position                       */
	uint32_t rid;     /* contig id                              */
	uint32_t missing; /* twk1_t.gt_missing (variant has a mask) */
	uint32_t _pad;
	double   hwe;     /* Hardy-Weinberg P                       */
} twk_hip_variant_meta;

/* Output filters: twk_ld_settings.minR2/maxR2/minDprime/maxDprime/minP
 * (include/core.h:921; defaults lib/core.cpp:304). */
typedef struct {
	double minR2, maxR2, minDprime, maxDprime, minP;
} twk_hip_filters;

/* A rectangular slice of the variant-pair space, in variant indices as
 * uploaded.  Replaces one ticket of twk_ld_dynamic_balancer::GetBlockPair
 * (lib/ld/ld_balancing.h:214-233): rows [rowA0,rowA0+nA) x cols [rowB0,rowB0+nB).
 * When diag != 0 (requires rowA0 == rowB0, nB >= nA) only pairs with col > row
 * are evaluated (ticket type 1, ld_engine.cpp:1913-1933); the columns beyond nA
 * form the rectangle to the right of the diagonal square. */
typedef struct {
	uint32_t rowA0, nA, rowB0, nB;
	int32_t  diag;
	int32_t  window;   /* option bits, see TWK_HIP_OPT_*                                */
	uint32_t l_window; /* base pairs (twk_ld_settings.l_window)                         */
	uint32_t _pad;
} twk_hip_tile_desc;

/* Option bits for twk_hip_tile_desc.window and the `window` argument of ld_all / ld_region. */
enum {
	TWK_HIP_OPT_WINDOW      = 1, /* keep only same-contig pairs with |posA-posB| <= l_window (calc -w) */
	TWK_HIP_OPT_KEEP_LOW_AC = 2  /* do not skip pairs with ac_A + ac_B <= 2: the single-site loop has the
	                                skip commented out (ld_engine.cpp:2267-2269, 2293-2295)            */,
	TWK_HIP_OPT_REF_COMPAT  = 4  /* reproduce what the reference's PhasedVectorized really returns for pairs with
	                                missing genotypes when 2N is not a multiple of 128: its scalar tail adds the
	                                (A ref, B alt) count to REFREF and swaps the two off-diagonal cells, and REFREF is
	                                reduced by half the padding (ld_engine.cpp:596-609).  Off by
	                                default: the engine returns the correct table. */,
	TWK_HIP_OPT_R2_SCREEN   = 8  /* whole-triangle runs with filters.minR2 >= 0.001: decide from the two allele counts alone
	                                which pairs can reach the r2 cut-off at all (|D| <= a(1-b) for minor allele frequencies
	                                a <= b, so r2 <= a(1-b)/((1-a)b)), walk the variants in order of minor allele count and
	                                never contract the tiles outside that band.  Same records as without it (pairs in the
	                                band still go through the reference's formula); on cohort data, where most variants
	                                are rare, most pairs fall outside the band.  This is the engine's answer to the
	                                reference's O(carriers) list kernels for rare variants (twk_igt_list,
	                                include/core.h:517-672, PhasedListVector ld_engine.cpp:185-267): not a faster way to
	                                count a rare pair, a proof that it need not be counted.  Off by default at the ABI
	                                (bench.py measures the full contraction); `tomahawk calc` switches it on. */
};

/* One surviving pair, device-compacted.  Field-for-field the payload of
 * twk1_two_t (include/core.h:826-833) with variant indices instead of
 * rid/pos (the host expands them and writes the forward + reverse copies,
 * ld_engine.cpp:1290-1298). */
typedef struct {
	uint32_t idxA, idxB; /* variant indices as uploaded           */
	uint32_t flags;      /* twk1_two_t.controller                 */
	uint32_t _pad;
	double   cnt[4];     /* REFREF, ALTREF-slot, REFALT-slot, ALTALT (core.h:833) */
	double   D, Dprime, R, R2, P, ChiSqFisher, ChiSqModel;
} twk_hip_record;

/* ---- device / context ------------------------------------------------- */
int twk_hip_abi_version(void);
int twk_hip_device_count(void);
const char* twk_hip_strerror(int code);
/* Last HIP runtime error text recorded on this ctx ("" if none). */
const char* twk_hip_last_error(const twk_hip_ctx* ctx);
int twk_hip_ctx_create(int device, twk_hip_ctx** out);
int twk_hip_ctx_destroy(twk_hip_ctx* ctx);

/* Page-locked host memory for upload staging (hipHostMalloc / hipHostFree): uploads from it run at
 * full PCIe rate and without a bounce copy.  No reference counterpart (the reference never leaves
 * host memory). */
int twk_hip_host_alloc(size_t bytes, void** out);
int twk_hip_host_free(void* p);

/* ---- input ------------------------------------------------------------ */
/* Declare the problem: n_samples diploid samples, n_variants variants.
 * Replaces twk_ld_engine::SetSamples (ld_engine.cpp:55-71) + the ldd block
 * arrays (ld.cpp:390-391). Frees any previous data. */
int twk_hip_set_problem(twk_hip_ctx* ctx, uint32_t n_samples, uint32_t n_variants);

/* Upload variants [first, first+count) as reference-layout bitvectors
 * (twk_igt_vec, include/core.h:724-753 built by core.cpp:349-438): for sample
 * s bit 2s = first allele is ALT, bit 2s+1 = second allele is ALT, little-
 * endian uint64 words, stride64 words per variant (>= ceil(2N/64)).  `mask`
 * may be NULL (no variant has missing data); otherwise same layout with both
 * bits of a sample set when either allele is missing (core.cpp:379-380).
 * Replaces twk1_ldd_blk::Inflate (ld_structs.cpp:125-203). */
int twk_hip_upload_bitvectors(twk_hip_ctx* ctx, uint32_t first, uint32_t count,
                              const uint64_t* data, const uint64_t* mask, size_t stride64,
                              const twk_hip_variant_meta* meta);

/* The same upload with the expansion done on the device: `bytes` holds the run-length genotype words of
 * `count` variants exactly as they are stored in a .twk block (twk1_t::gt, include/core.h:195-205:
 * 1, 2 or 4 bytes per run, little endian, run = length << (2+2m) | A << (1+m) | B with m = 1 and two bits
 * per allele when the variant has missing genotypes), desc[i] says where variant first+i's runs start.
 * A HIP kernel does what twk_igt_vec::Build does on the reference's unpack threads
 * (lib/core.cpp:349-391, lib/ld/ld_unpacker.h:44-123): bitvector + mask, padding bits zero - so the
 * PCIe link carries the compressed genotypes, not N/4 bytes per variant.  `bytes` may be pageable or
 * pinned host memory and is free for reuse when the call returns.  Runs that do not add up to the
 * problem's sample count fail with TWK_HIP_E_INVALID (found on the device: the rows of this call are
 * then undefined). */
typedef struct {
	uint64_t offset;   /* byte offset of the variant's first run word in `bytes` */
	uint32_t n_runs;
	uint8_t  width;    /* bytes per run word: 1, 2 or 4 (twk1_t.gt_ptype)        */
	uint8_t  missing;  /* twk1_t.gt_missing: two bits per allele in the run word */
	uint16_t _pad;
} twk_hip_rle_desc;
int twk_hip_upload_rle(twk_hip_ctx* ctx, uint32_t first, uint32_t count, const void* bytes, size_t n_bytes,
                       const twk_hip_rle_desc* desc, const twk_hip_variant_meta* meta);

/* Read variants [first, first+count) back in the reference layout (twk_igt_vec::data / ::mask,
 * include/core.h:724-753): the inverse of twk_hip_upload_bitvectors, whatever way the rows were put
 * there (bitvectors, run-length words, the synthetic generator).  mask may be NULL; variants without
 * missing genotypes get an all-zero mask.  Parity tests of the device-side T1 use it. */
int twk_hip_download_bitvectors(twk_hip_ctx* ctx, uint32_t first, uint32_t count,
                                uint64_t* data, uint64_t* mask, size_t stride64);

/* Fill the whole problem with the synthetic benchmark input of SURVEY 8(d)
 * directly in HBM (iid alleles, per-variant ALT frequency U(0.05,0.5), one
 * contig, pos = 1000+100*v, no missing data).  Bit-identical to the host
 * generator twk_synth_bitvector() below. */
int twk_hip_generate_synthetic(twk_hip_ctx* ctx, uint64_t seed);
/* The same for a slab: local variant v is global variant first_variant + v (bits, ALT
 * frequency and position all follow the global id), so that ranks that hold different
 * slabs of one synthetic data set agree on every variant they share (halo). */
int twk_hip_generate_synthetic_range(twk_hip_ctx* ctx, uint64_t seed, uint32_t first_variant);
/* Host twin of the device generator: writes variant v's bitvector
 * (ceil(2N/64) words) and returns its ALT allele count. */
uint32_t twk_synth_bitvector(uint64_t seed, uint32_t n_samples, uint32_t v, uint64_t* out_words);

/* The same input with LD planted in it: iid genotypes have no pair near the default r2 cut-off, so a run over them exercises
 * neither the recount of screened-in pairs nor the pair math, Fisher's test, the sort and the way out.  With a plant, odd variant
 * 2k + 1 (k < n_planted; global ids) is a noisy copy of the even variant 2 ((k mult + offset) mod half): every allele of the
 * source flipped with probability eps_k = max_eps u_k, u_k ~ U[0, 1) from the seed.  Sources are never copies, and with mult
 * coprime to half no two copies share a source: the data set then holds exactly n_planted pairs in LD - r2 from 1 down to about
 * (1 - 2 max_eps)^2 - spread over the whole pair triangle (mult large) or at a fixed distance 2 offset - 1 (mult = 1; window
 * runs), and nothing else above r2 ~ 30 / N.  plant == NULL or n_planted == 0: twk_hip_generate_synthetic_range.
 * Constraints (else TWK_HIP_E_INVALID): n_planted <= half, mult >= 1, 0 <= max_eps <= 0.5.  `half` describes the GLOBAL data
 * set (n_variants / 2): the slabs of one data set (first_variant) all pass the same plant and agree on every variant they share.
 * No reference counterpart (the reference's benchmarks read 1000 Genomes files, docs/tutorial.md:177-199). */
typedef struct {
	uint32_t n_planted; /* copies: global variants 1, 3, ..., 2 n_planted - 1                       */
	uint32_t half;      /* sources are the global variants 2 j, j < half                             */
	uint32_t mult;      /* j = (k mult + offset) mod half                                            */
	uint32_t offset;
	double   max_eps;   /* largest flip probability                                                  */
} twk_hip_plant;
int twk_hip_generate_synthetic_planted(twk_hip_ctx* ctx, uint64_t seed, uint32_t first_variant, const twk_hip_plant* plant);
/* Host twins: global variant v of the planted data set (ceil(2N/64) words; returns its ALT allele count), and whether v is a
 * copy (returns 1 and its source / flip probability; 0 and *src = v, *eps = 0 otherwise).  Bit-identical to the device. */
uint32_t twk_synth_planted_bitvector(uint64_t seed, uint32_t n_samples, uint32_t v, const twk_hip_plant* plant, uint64_t* out_words);
int twk_synth_plant_source(uint64_t seed, const twk_hip_plant* plant, uint32_t v, uint32_t* src, double* eps);

/* Copy back per-variant ALT allele count / het / hom-alt / missing-sample
 * counts as seen by the device (popcounts of the planes).  Any pointer may be
 * NULL. */
int twk_hip_get_marginals(twk_hip_ctx* ctx, uint32_t* ac, uint32_t* n_het, uint32_t* n_hom,
                          uint32_t* n_miss_samples);

/* ---- compute ---------------------------------------------------------- */
/* Raw contingency counts for one tile (debug/parity entry point; replaces the
 * `helper.alleleCounts` filled by the K1-K6 kernels before the math).
 * mode PHASED  : out[4*(i*nB+j)+{0,1,2,3}] = REFREF, ALTREF(1), REFALT(4), ALTALT(5)
 *                cells of twk_ld_count (ld_engine.h:27-30).
 * mode UNPHASED: out[9*(i*nB+j)+k] = cells {0,1+4,5,16+64,17+20+65+68,21+69,80,81+84,85}.
 * Pairs excluded by `diag` are zero-filled. */
int twk_hip_count_tile(twk_hip_ctx* ctx, int mode, const twk_hip_tile_desc* tile, uint64_t* out);

/* LD for one tile: counts -> math -> filters -> compacted survivors.
 * `out` is a HOST buffer of `capacity` records; *n_out receives the number of
 * survivors (on TWK_HIP_E_OVERFLOW: the number required, nothing guaranteed in
 * out).  *n_pairs (may be NULL) receives the number of pairs compared, counted
 * as the reference's progress counter does (ld_engine.cpp:1933,2015). */
int twk_hip_ld_tile(twk_hip_ctx* ctx, int mode, const twk_hip_tile_desc* tile,
                    const twk_hip_filters* filters, twk_hip_record* out, uint64_t capacity,
                    uint64_t* n_out, uint64_t* n_pairs);

/* All-vs-all LD over the upper triangle, restricted to shard `part` of
 * `n_parts`: shard k is the contiguous band of rows that holds the k-th n_parts-th
 * of the pairs (equal-area bands of the triangle, boundaries on multiples of 64
 * variants; every rank derives the same partition without communication).  This
 * replaces twk_ld_balancer::Build (ld_balancing.h:23-80) for multi-GPU sharding.
 * tile_variants = edge of a super-tile in variants (0 = choose: launches sized
 * by their work where no count matrix is kept, see the "band_launch" switch).
 * Survivors are handed to `sink` (may be NULL to discard) launch by launch, one
 * call at a time, all of them before this function returns - launches with
 * many survivors from a second thread of the engine while the calling thread
 * goes on with the launches, the others from the calling thread once that
 * thread has caught up (option "async_delivery", default 1; 0: always the
 * calling thread) -
 * a launch's survivors in pieces of at most 2^20 records; the
 * records of one sink call are in (idxA, idxB) order (sorted on the device), the
 * pieces of a launch follow each other in that order, and a piece stays valid
 * until the sink returns.  *n_pairs / *n_records (may be NULL)
 * receive totals for this shard. */
typedef int (*twk_hip_record_sink)(void* user, const twk_hip_record* recs, uint64_t n);
int twk_hip_ld_all(twk_hip_ctx* ctx, int mode, const twk_hip_filters* filters,
                   uint32_t part, uint32_t n_parts, uint32_t tile_variants,
                   int32_t window, uint32_t l_window,
                   twk_hip_record_sink sink, void* user, uint64_t* n_pairs, uint64_t* n_records);

/* Same over a sub-region of the pair space: rows [a0,a0+nA) x cols [b0,b0+nB).
 * triangle != 0 (requires a0 == b0, nA == nB): only col > row.  This is what a
 * `-c/-C` chunk of the reference is (ld_balancing.h:59-78: diagonal or square). */
int twk_hip_ld_region(twk_hip_ctx* ctx, int mode, const twk_hip_filters* filters,
                      uint32_t a0, uint32_t nA, uint32_t b0, uint32_t nB, int32_t triangle,
                      uint32_t part, uint32_t n_parts, uint32_t tile_variants,
                      int32_t window, uint32_t l_window,
                      twk_hip_record_sink sink, void* user, uint64_t* n_pairs, uint64_t* n_records);

/* Multi-GPU runs: keep the survivors of twk_hip_ld_all / twk_hip_ld_region on the device.  With on != 0 the
 * record sink of those calls is not invoked; the survivors of every tile are appended (each tile in (idxA, idxB)
 * order, tiles in the order they finish) to one device buffer that twk_hip_device_records() returns, so that a
 * gather over RCCL can send them GPU to GPU and only the writer rank copies them to host memory, once.  This is
 * the per-worker output block of the reference (twk_ld_engine::blk_f / blk_r, lib/ld/ld_engine.h:321, flushed
 * into the shared writer at ld_engine.cpp:1270-1281) held in HBM until the gather.  Every call of
 * twk_hip_set_device_sink (on or off) empties the buffer.  *records is device memory owned by the ctx, valid
 * until the next compute call or twk_hip_set_device_sink; NULL when *n == 0. */
int twk_hip_set_device_sink(twk_hip_ctx* ctx, int on);
int twk_hip_device_records(twk_hip_ctx* ctx, const twk_hip_record** records, uint64_t* n);

/* One process, several GPUs (`tomahawk calc` with TWK_HIP_GPUS = n and engine option "gather" = 1): gather the device sinks of n
 * contexts - one per GPU, each with twk_hip_set_device_sink on and its region calls done - into the device sink of ctxs[dst], GPU to
 * GPU over RCCL: one communicator clique per set of devices (ncclCommInitAll, kept for the life of the process), then ONE group of
 * exact-size ncclSend / ncclRecv on the contexts' copy streams - no padding to a common size, no host hop.  Afterwards ctxs[dst]'s
 * sink holds its own records followed by those of the other contexts in the order of ctxs[]; the others' sinks are empty.
 * This is the north star's "final RCCL gather of .two output blocks over xGMI" inside the C++ product; it replaces every slave's
 * flush of its output block into the shared writer (lib/ld/ld_engine.cpp:1742-1802).  librccl is opened on first use (dlopen), never
 * linked: twk_hip_gather_backend() says what was found (and NCCL_DEBUG is set to NONE unless the caller has set it: RCCL otherwise
 * prints a banner to stdout when its first communicator comes up).  n == 1: nothing to move - unless flags has TWK_HIP_GATHER_SELF_LOOP, which
 * sends the one context's records from its sink to itself through the same group of calls (what a one-GPU box can exercise of the
 * path).  *transfer_ms (may be NULL): the transfers' duration on the destination's stream (HIP events).  Everything gathered must fit
 * the destination's HBM beside its problem (TWK_HIP_E_NOMEM otherwise: the caller falls back to per-GPU copies to the host). */
enum { TWK_HIP_GATHER_SELF_LOOP = 1 };
int twk_hip_gather_records(twk_hip_ctx* const* ctxs, uint32_t n, uint32_t dst, int32_t flags, uint64_t* n_records, double* transfer_ms);
const char* twk_hip_gather_backend(void);      /* "rccl <version code>" or "unavailable (<why>)" */
/* Hand what the device sink holds to `sink` on the host - in pieces of at most 2^20 records through the engine's pinned staging, in the
 * order they lie in the sink - and empty it.  (The records of a gathered run leave the destination GPU this way, once.) */
int twk_hip_drain_device_sink(twk_hip_ctx* ctx, twk_hip_record_sink sink, void* user, uint64_t* n_records);

/* The row band [row_begin,row_end) that shard `part` of `n_parts` owns in a region of
 * n_rows x n_cols variants (triangle != 0: col > row only, n_rows == n_cols) and the
 * number of pairs in it.  Pure host arithmetic (no device needed): lets a launcher
 * or a test derive the multi-GPU partition that twk_hip_ld_all/ld_region use. */
int twk_hip_shard_rows(uint32_t n_rows, uint32_t n_cols, int32_t triangle, uint32_t part, uint32_t n_parts,
                       uint32_t* row_begin, uint32_t* row_end, uint64_t* n_pairs);

/* The planner behind twk_hip_ld_region on caller-supplied arrays - pure host arithmetic, no device, no ctx (csrc/hip/ld_plan.h): the
 * shard's rows, the columns every row reaches (window: same contig, |dpos| <= l_window; r2 band: from the allele counts `popc`, positions
 * in order of minor allele count), and the launches (tile descriptors) that cover them; the first *n_band_launches of them are band
 * launches (fused form: rows x all the columns they reach).  meta / popc: one entry per position of the index space [0, n_variants).
 * tiles: room for `capacity` descriptors (TWK_HIP_E_OVERFLOW with *n_tiles = the number needed); lo / hi: nA entries each or NULL (filled
 * in window / band mode: row a0 + r reaches columns [b0 + lo[r], b0 + hi[r])).  Replaces twk_ld_balancer::Build and the block-pair
 * ticker (lib/ld/ld_balancing.h:23-80, 176-233) for one GPU's share; exported so that the partition can be tested and inspected
 * without a GPU (tests/test_plan.py). */
typedef struct {
	uint32_t n_samples;
	int32_t  planes_per_variant;  /* plane rows per variant of the mode's plane set: 1 phased, 2 unphased (masked: 2 / 3) */
	uint32_t k_chunks;            /* 128-byte chunks of a plane row (band launches are sized by tiles x chunks)            */
	uint32_t resident_blocks;     /* count-kernel blocks the chip holds at once (2 per CU: 512)                          */
	int32_t  screen;              /* 0 none, 1 r2 band for PhasedMath, 2 for UnphasedMath                                 */
	int32_t  fused;               /* the mode's launches run the fused count -> screen form                              */
	int32_t  phased_math;
	int32_t  band_launch, band_reverse;   /* the options of the same names */
	int32_t  _pad;
	int64_t  band_work_log2, band_max_launches;
	double   minR2;
} twk_hip_plan_env;
int twk_hip_plan_region(const twk_hip_plan_env* env, const twk_hip_variant_meta* meta, const uint32_t* popc, uint32_t n_variants,
                        uint32_t a0, uint32_t nA, uint32_t b0, uint32_t nB, int32_t triangle, uint32_t part, uint32_t n_parts,
                        uint32_t tile_variants, int32_t window, uint32_t l_window,
                        twk_hip_tile_desc* tiles, uint32_t capacity, uint32_t* n_tiles, uint32_t* n_band_launches,
                        uint32_t* row_begin, uint32_t* row_end, uint64_t* n_pairs, uint32_t* lo, uint32_t* hi);

/* Fisher's exact test on n caller-supplied 2x2 tables (tables[4*i + {0,1,2,3}] = n11, n12, n21, n22), two-sided P
 * into p_two_sided[i]: kt_fisher_exact (lib/fisher_math.cpp:231-267) as the pair math calls it
 * (ld_engine.cpp:1222-1226, :1656-1658), through the engine's own Fisher kernels: the reference's walk, one table per
 * lane, behind the search for its starting points.  in_given_order == 0: as the pair math runs it, the walks in the order
 * of their length; != 0: in the order of `tables` (same P, bit for bit: a measurement switch).  Needs
 * twk_hip_set_problem first (the log-factorial table covers counts up to 2 * n_samples + 15; larger counts take
 * lgamma itself).  *kernel_ms (may be NULL): the kernels' duration (HIP events).  A parity and measurement entry
 * point: the pair math reaches the same kernels internally. */
int twk_hip_fisher_exact(twk_hip_ctx* ctx, const int32_t* tables, uint64_t n, double* p_two_sided,
                         int32_t in_given_order, float* kernel_ms);

/* ---- switches ---------------------------------------------------------- */
/* Measurement and test switches of one ctx, by name.  The library reads NO environment variable: a process that
 * embeds it gets the documented defaults unless it calls this.  There is no reference counterpart (the reference's
 * closest relative is the compile-time SLAVE_DEBUG_MODE / SIMD_AVAILABLE switches, lib/ld/ld_engine.h:20-24).
 * The keys, their defaults, ranges and meaning are ONE table in the engine (TWK_HIP_OPTIONS, csrc/hip/twk_hip.hip), readable through
 * twk_hip_option_describe() below and printed as the table of INTEGRATION.md 1 (tests/test_docs_consistency.py keeps the document in
 * step with it).  In short: "fused", "three" pick the form of the contraction; "lists", "list_max", "probe", "probe_zone",
 * "probe_lds" the carrier-list passes for rare variants; "band_*" the launches of fused runs; "patch_rows" / "patch_cols" / "seg" /
 * "xcd_queues" / "skip_pad" / "count_min_chunks" / "cand_chunk" the count kernel's work order; "fisher_*" Fisher's test;
 * "async_delivery", "deliver_buffers" the engine's delivery thread; "record_cap", "deliver_fail_*_at" are test hooks; "timeline" a log.
 * Keys of the host side (`tomahawk calc --engine-option`, twk_ld::SetEngineOption - handled in csrc/host/twk_ld.cpp, unknown to this
 * function): "force_device", "progress_ms", "map_output", "emit_workers", "emit_backlog_mb", "emit_queue_pieces", "record_codec", "direct_output" (INTEGRATION.md 1).
 * None of them changes a record (tests/test_gpu_fused.py, test_gpu_lists.py); "lists" / "list_max" drop the derived
 * plane sets, which are rebuilt on next use.  Unknown key or value out of range: TWK_HIP_E_INVALID. */
int twk_hip_set_option(twk_hip_ctx* ctx, const char* key, int64_t value);
int twk_hip_get_option(const twk_hip_ctx* ctx, const char* key, int64_t* value);
/* Entry `index` (0, 1, ... until TWK_HIP_E_INVALID) of the engine's option table: key, default, range, one line of meaning (static strings;
 * any pointer may be NULL).  Needs no device and no ctx. */
int twk_hip_option_describe(uint32_t index, const char** key, int64_t* dflt, int64_t* lo, int64_t* hi, const char** meaning);

/* ---- measurement ------------------------------------------------------ */
/* Cumulative device time (HIP events on the engine's own stream) and launch
 * count of the dominant kernel (count-tile) and of the math kernel since the
 * last reset, plus the algorithmic units they processed. */
typedef struct {
	double   count_ms;      /* sum of count-kernel durations                     */
	double   stats_ms;      /* sum of math/filter-kernel durations (a band launch: up to and including the sort of its survivors) */
	uint64_t count_launches;
	uint64_t stats_launches;
	uint64_t row_pairs;     /* plane-row pairs contracted by the count kernel    */
	uint64_t variant_pairs; /* variant pairs evaluated by the math kernel        */
	uint64_t words_per_row; /* 32-bit words contracted per row pair (unpadded)   */
	uint64_t fused_launches;/* count launches that ran the fused count -> r2 screen form (short rows,
	                           phased math: no count matrix, candidates only)    */
	uint64_t candidates;    /* list slots those launches handed to the math kernel:
	                           the pairs that passed the screen, plus the slots a wave
	                           had reserved and not used when its launch ended      */
	double   list_ms;       /* sum of carrier-list intersection kernel durations (rare variants at very large sample
	                           counts: the device's twk_igt_list / PhasedListVector, core.h:517-672, ld_engine.cpp:185-267) */
	uint64_t list_launches;
	uint64_t list_pairs;    /* variant pairs decided by list intersection instead of the dense contraction */
	double   probe_ms;      /* sum of the probe kernel's durations: a rare variant's carrier list against the bitvector rows of variants
	                           that keep no list (K1's asymmetric path, ld_engine.cpp:230-242) */
	uint64_t probe_launches;
	uint64_t probe_pairs;   /* variant pairs decided that way */
	uint64_t count_shader_cycles; /* what the count kernel's blocks lived for, summed over the blocks: shader clock cycles ...   */
	uint64_t count_wall_ticks;    /* ... and ticks of the constant 100 MHz counter.  cycles / ticks x 100 MHz = the clock the
	                                 launches really ran at (a chip that was idle needs ~20 ms of load to reach its 2.4 GHz) */
	uint64_t three_launches;      /* count launches that ran the three-product form of the plain unphased planes: per word and
	                                 variant pair HH = popc(H_A & H_B) and S = popc(Q_A & C_B) + popc(C_A & Q_B), C = H | Q - what
	                                 UnphasedMath's r2 screen reads (ld_engine.cpp:1363-1375 via minhap / maxhap) - instead of the four
	                                 products HH, HQ, QH, QQ; the reference's list kernel likewise takes one popcount and the other
	                                 cells from the margins (ld_engine.cpp:244-246) */
	uint64_t three_row_pairs;     /* ... and the plane-row pairs of those launches (of row_pairs): they executed 3/4 of the
	                                 AND+popcounts row_pairs x words_per_row stands for (one v_and / v_bitop3 + one v_bcnt each) */
	uint64_t recount_candidates;  /* pairs that passed the three-product screen and had their four products counted afresh */
	uint64_t outlier_launches;    /* count launches the outlier watch flagged (twk_hip_launch_log)                          */
	double   finish_ms;           /* (ABI 5) the calling thread's wall time between a launch's last kernel and the hand-over of its
	                                 records: device sort, copy to the host, the sink (launches the delivery thread takes: up to
	                                 their copy aside on the device) - summed over the launches                              */
} twk_hip_timing;
/* Progress of twk_hip_ld_all / twk_hip_ld_region: `cb` runs on the calling thread after every tile
 * with the variant pairs finished so far in the current call and the tile counts.  Replaces the
 * counters the reference's ticker thread polls every 30 s (lib/ld/ld_progress.h:40-86: n_var, n_out).
 * cb == NULL switches it off. */
typedef void (*twk_hip_progress_cb)(void* user, uint64_t pairs_done, uint32_t tiles_done, uint32_t tiles_total);
int twk_hip_set_progress(twk_hip_ctx* ctx, twk_hip_progress_cb cb, void* user);

/* The count launches one by one, since the last twk_hip_timing_reset (a ring of the last 4096): what the engine's outlier watch
 * looks at.  A launch is an outlier when its time per unit of work (row_pairs x words_per_row, the three-product form's at 3/4)
 * exceeds 1.4 x the median of the launches of its own kind and row length before it (at least 8 of them, launches of >= 0.3 ms);
 * option "timeline" >= 1 prints such a launch - with the shader clock its blocks ran at and how far apart the XCDs finished - on
 * stderr as it is found.  Measurement integrity (round 4 saw the same run take 1.7 x as long now and then); no reference counterpart. */
typedef struct {
	double   ms;                  /* HIP events around the count kernel                                        */
	double   shader_mhz;          /* clock the blocks ran at (0: not measured)                                 */
	double   xcd_finish_spread_us;/* last block of the last XCD to finish minus last block of the first one    */
	uint64_t row_pairs;           /* plane-row pairs contracted                                                */
	uint64_t candidates;          /* fused / three-product launches: pairs that passed the screen              */
	uint32_t words_per_row;
	uint32_t kind;                /* 0 four products -> matrix, 1 three products -> matrix, 2 fused PhasedMath,
	                                 3 fused UnphasedMath four products, 4 fused UnphasedMath three products   */
	uint32_t outlier;             /* != 0: flagged by the watch                                                */
	uint32_t _pad;
} twk_hip_launch_stat;
/* Copies the most recent min(capacity, launches kept) entries, oldest first; *n_total: launches seen since the reset. */
int twk_hip_launch_log(twk_hip_ctx* ctx, twk_hip_launch_stat* out, uint32_t capacity, uint32_t* n_copied, uint64_t* n_total);

int twk_hip_timing_reset(twk_hip_ctx* ctx);
int twk_hip_timing_get(twk_hip_ctx* ctx, twk_hip_timing* out);

#ifdef __cplusplus
}
#endif
#endif /* TWK_HIP_H_ */
