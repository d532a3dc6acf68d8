/*
 * TEST INFRASTRUCTURE ONLY -- the CPU oracle.
 *
 * A plain scalar C restatement of the reference's (mklarqvist/tomahawk v0.7.0)
 * pairwise-LD algorithm: RLE -> bitvector build, contingency counting, phased
 * and unphased statistics, Fisher's exact test, record packing and the pair
 * loop with its kernel-selection heuristics.  Every function cites the
 * reference file:line it follows.  Nothing here is shipped or measured: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it,
 * and only as the checker.
 *
 * Parity pinning: tests/test_oracle_golden.py checks this oracle against
 * golden records produced by the compiled reference itself (oracle/_ref, see
 * tests/golden/make_golden.py) and against the Fisher / record known-answer
 * values of SURVEY.md 8(c).
 */
#ifndef TWK_LD_ORACLE_H_
#define TWK_LD_ORACLE_H_
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* twk1_t fields used on the path (include/core.h:291-295). */
typedef struct {
	uint32_t ac, an, pos, rid;
	uint32_t gt_missing, gt_phase;
	double   hwe;
} orc_variant;

/* twk_ld_settings subset (include/core.h:909-924, defaults lib/core.cpp:297-306). */
typedef struct {
	double minR2, maxR2, minDprime, maxDprime, minP;
	int force_phased, forced_unphased;
	int keep_low_ac;   /* single-site loop: the ac skip is commented out (ld_engine.cpp:2267-2269) */
	int ref_compat;    /* restate PhasedVectorized with its scalar-tail and pad-correction slips too
	                      (ld_engine.cpp:596-609): what the reference really returns for
	                      -p pairs with missing data when 2N is not a multiple of 128 */
} orc_settings;

/* twk1_two_t (include/core.h:826-833). */
typedef struct {
	uint32_t controller;
	uint32_t ridA, ridB, Apos, Bpos;
	double cnt[4];
	double D, Dprime, R, R2, P, ChiSqFisher, ChiSqModel;
} orc_record;

void orc_default_settings(orc_settings* s);

/* words of 64 bits per variant: ceil(2N/64) (ld_engine.cpp:58). */
uint32_t orc_words64(uint32_t n_samples);

/* T1 twk_igt_vec::Build (lib/core.cpp:349-391).  Runs are (length, alleleA,
 * alleleB) with allele codes 0 ref / 1 alt / 2 missing
 * (lib/genotype_encoder.h:11-17).  mask may be NULL when no run is missing.
 * Returns 0, or -1 if the run lengths do not sum to n_samples. */
int orc_build_bitvector(const uint32_t* run_len, const uint8_t* run_a, const uint8_t* run_b,
                        uint32_t n_runs, uint32_t n_samples, uint64_t* data, uint64_t* mask);

/* Genotype of sample s out of a bitvector: alleles 0/1, 2 if the sample is masked. */
void orc_genotype(const uint64_t* data, const uint64_t* mask, uint32_t s, int* a, int* b);

/* 2x2 haplotype table, vector-kernel semantics (PhasedVectorized /
 * PhasedListVector, ld_engine.cpp:185-267,513-634): a sample masked in either
 * variant contributes nothing.  out = {REFREF(0), c[1], c[4], ALTALT(5)} with
 * the vector kernels' orientation c[1] = (A alt, B ref) (ld_engine.cpp:244-246,
 * 606-607). */
void orc_count_phased(const uint64_t* a, const uint64_t* ma, const uint64_t* b, const uint64_t* mb,
                      uint32_t n_samples, uint64_t out[4]);
/* Same table with the run-length kernel's semantics (PhasedRunlength,
 * ld_engine.cpp:1011-1091): per-allele exclusion of missing alleles and
 * c[1] = (A ref, B alt). */
/* K3 PhasedVectorized exactly as compiled, slips included (ld_engine.cpp:513-634). */
void orc_count_phased_k3_as_is(const uint64_t* a, const uint64_t* ma, const uint64_t* b, const uint64_t* mb,
                               uint32_t n_samples, uint64_t out[4]);
void orc_count_phased_rle(const uint64_t* a, const uint64_t* ma, const uint64_t* b, const uint64_t* mb,
                          uint32_t n_samples, uint64_t out[4]);
/* 3x3 genotype table (UnphasedVectorized / UnphasedRunlength,
 * ld_engine.cpp:709-1009,1093-1160) as the nine sums UnphasedMath uses:
 * out = {0, 1+4, 5, 16+64, 17+20+65+68, 21+69, 80, 81+84, 85}. */
void orc_count_unphased(const uint64_t* a, const uint64_t* ma, const uint64_t* b, const uint64_t* mb,
                        uint32_t n_samples, uint64_t out[9]);

/* M3 kt_fisher_exact (lib/fisher_math.cpp:231-267). Returns q. */
double orc_fisher_exact(int n11, int n12, int n21, int n22, double* left, double* right, double* two);

/* M1 PhasedMath (ld_engine.cpp:1162-1310).  c = {c[0], c[1], c[4], c[5]}.
 * Returns 1 and fills rec if the pair survives, 0 otherwise. */
int orc_phased_math(const uint64_t c[4], const orc_variant* A, const orc_variant* B,
                    const orc_settings* st, orc_record* rec);
/* M2 UnphasedMath + ChiSquaredUnphasedTable + ChooseF11Calculate
 * (ld_engine.cpp:1312-1740).  c9 as orc_count_unphased. */
int orc_unphased_math(const uint64_t c9[9], const orc_variant* A, const orc_variant* B,
                      const orc_settings* st, orc_record* rec);

/* S1: one pair as the slave loops treat it (ld_engine.cpp:1898-2188, 2740-2838):
 * ac skip, phased/unphased choice (forced or the default an-rule), kernel
 * choice by the ac thresholds (decides semantics / orientation only).
 * vector_only != 0 forces the vector kernels (the GPU contract). */
int orc_pair(const uint64_t* a, const uint64_t* ma, const orc_variant* A,
             const uint64_t* b, const uint64_t* mb, const orc_variant* B,
             uint32_t n_samples, const orc_settings* st, int vector_only, orc_record* rec);

/* All pairs i<j of M variants (bitvectors row-major, stride words64; mask may
 * be NULL).  recs must hold M*(M-1)/2 records.  Returns the number written. */
uint64_t orc_all_pairs(const uint64_t* data, const uint64_t* mask, const orc_variant* vars,
                       uint32_t n_variants, uint32_t n_samples, const orc_settings* st,
                       int vector_only, orc_record* recs);
/* The same for rows [row0, row1) of the triangle only (re-entrant: the test harness runs disjoint row
 * ranges on several host threads); recs must hold the pairs of those rows. */
uint64_t orc_all_pairs_rows(const uint64_t* data, const uint64_t* mask, const orc_variant* vars,
                            uint32_t n_variants, uint32_t n_samples, const orc_settings* st,
                            int vector_only, uint32_t row0, uint32_t row1, orc_record* recs);

/* O1 serialiser of twk1_two_t (lib/core.cpp:470-490): 106 bytes. */
void orc_pack_record(const orc_record* r, uint8_t out[106]);

#ifdef __cplusplus
}
#endif
#endif
