/*
 * satabsearch_debug.h - what is NOT part of the drop-in boundary of satabsearch.h: environment overrides of the
 * library's launch heuristics, for tuning runs and tests.  Results never depend on them (the random streams are
 * keyed by query / db ordinal / restart, not by the launch shape); they are read ONCE by sat_ctx_create, never on
 * the search path.  Nothing here is needed to use the library.
 *
 *   SAT_EXP_LPC = 0|1|2          log2 lanes per restart chain (default: by LDS occupancy, sat_capi.hip)
 *   SAT_EXP_LPC_WAVES = n        resident waves per CU at which that choice stops adding lanes (default 8; 12 for queries above 64 SSEs)
 *   SAT_EXP_CHAINS = 64|128|192 restart chains per workgroup (default: one per restart, at most 256)
 *   SAT_EXP_COMPACT = 0|1        wave-level work compaction of the SA step (default: exactly when LORDER)
 *   SAT_EXP_QLDS = 0|1           query cells staged in LDS (default: queries of up to 16 SSEs)
 *   SAT_EXP_LDS_PAD = bytes      unused LDS added per db entry (occupancy experiments)
 *   SAT_EXP_EPW = 1..8           db entries per workgroup (default: chosen per launch from the CU's LDS granules)
 *   SAT_EXP_GENERAL = 1          the general kernel instantiation instead of the option-specialised ones
 *   SAT_EXP_STREAMS = 0          queue the order buckets of a search one after the other instead of concurrently
 *   SAT_EXP_UPLOAD_THREADS = n   host threads slicing the database copy (default 4)
 *   SAT_EXP_UPLOAD_TIMING = 1    per-phase upload times on stderr
 *   SAT_EXP_UPLOAD_PIECES = n    pieces of the overlapped upload + search (default by size, at most 8)
 *   SAT_MULTI_GATHER = rccl|peer the gather of sat_multi_* (default: RCCL, falling back to peer copies)
 *   SAT_PARSE_THREADS = n        threads of the mmap reader (libsathost; default: cores, at most 16)
 *   SAT_DEVICE_LIB = path        Python wrapper only: load another build of libsatabsearch.so (A/B runs)
 *
 * Diagnostic BUILDS (-DSAT_DIAG ..., cuda_satabsearch_amd/csrc/diag/sat_diag.hpp: phase timers, issue-sensitivity
 * perturbations, duplicated LDS accesses, the per-move self-check) are separate libraries made by
 * scripts/exp/variant_lib.sh and tests/native; the shipped library contains none of that code.  They export one
 * extra symbol:
 */
#ifndef SATABSEARCH_DEBUG_H
#define SATABSEARCH_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif
/* diagnostic builds only: counters of the last search - [0..7] and [10] wave-cycles per phase of the kernel, [8] self-check
 * mismatches, [9] self-checks made */
void sat_diag_counters(unsigned long long out[16]);
#ifdef __cplusplus
}
#endif
#endif
