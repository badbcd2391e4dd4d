// sat_diag.hpp - the lab bench of the SA kernel: phase timers, issue-sensitivity perturbations, duplicated LDS
// accesses for the bank-conflict attribution, and the per-move self-check.  NOT part of the product: the shipped
// libsatabsearch.so is built without -DSAT_DIAG and never includes this file (sat_sa_kernel.hpp then defines every
// hook below as nothing).  Diagnostic builds (scripts/exp/variant_lib.sh, tests/native) pass -DSAT_DIAG and one of
//   -DSAT_DIAG_PHASE        wave-cycles per phase of the SA step (s_memtime), printed on stderr after each search
//   -DSAT_DIAG_PERTURB=k    extra instructions of one kind per SA step: 1: 40 full-rate VALU, 2: 40 SALU,
//                           3: 10 LDS reads + wait, 4: 40 s_nop, 5: 40 half-rate VALU
//   -DSAT_DIAG_DUP=k        ONE LDS access site issued twice, results unchanged: 1 db-cell gathers, 2 map words of the
//                           rounds, 3 accumulator atomics, 4 the proposal's own-map byte (scripts/exp/ablate_lds.sh)
//   -DSAT_DIAG_FS_ROWS      the rows-in-step form of the initial full score everywhere
//   -DSAT_DIAG_SELFCHECK    the reference's TESTING assertion (K.cu:1105-1134: score + delta == tmscord(...) on every
//                           move): after every proposal the full score of the PROPOSED map is recomputed from scratch
//                           and compared with score + delta; mismatches are counted in diag[8] (tests/test_gpu_parity.py)
// Counters: SatKernelArgs::diag points at 16 u64 - [0..7] and [10] phase wave-cycles, [8] self-check mismatches, [9] checks.
#pragma once

#define SAT_DIAG_ARGS unsigned long long *diag;

#ifdef SAT_DIAG_PHASE
#define SAT_PHASE_INIT unsigned long long ph_acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t0 = __builtin_amdgcn_s_memtime()
#define SAT_PHASE(k) do { const unsigned long long ph_t1 = __builtin_amdgcn_s_memtime(); ph_acc[k] += ph_t1 - ph_t0; ph_t0 = ph_t1; } while (0)
#define SAT_PHASE_FLUSH do { if ((threadIdx.x & 63) == 0) for (int k = 0; k < 11; k++) if (k < 8 || k == 10) atomicAdd(a.diag + k, ph_acc[k]); } while (0)
#endif

#ifdef SAT_DIAG_FS_ROWS
#define SAT_DIAG_FS_ROWS_ONLY 1
#endif

#ifdef SAT_DIAG_DUP
#if SAT_DIAG_DUP == 1
// every db-cell gather a second time (volatile, result dropped): the LDS counters grow by exactly this site's share
#define SAT_DIAG_DUP_CELLS(row, l0, l1, l2, l3) do { if constexpr (CELLS == SAT_CELLS_FULL8) {                                            \
        typedef const volatile __attribute__((address_space(3))) unsigned long long *lds_vu64;                          \
        const unsigned long long dup0 = *(lds_vu64)&(row).cells[l0], dup1 = *(lds_vu64)&(row).cells[l1],                  \
                                 dup2 = *(lds_vu64)&(row).cells[l2], dup3 = *(lds_vu64)&(row).cells[l3];                  \
        asm volatile("" : : "v"(dup0), "v"(dup1), "v"(dup2), "v"(dup3)); } } while (0)
#elif SAT_DIAG_DUP == 2
#define SAT_DIAG_DUP_MAPWORD(p) (void)*(const volatile __attribute__((address_space(3))) uint32_t *)(p)
#elif SAT_DIAG_DUP == 3
#define SAT_DIAG_DUP_ATOMIC(p) __hip_atomic_fetch_add((p), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT)
#elif SAT_DIAG_DUP == 4
#define SAT_DIAG_DUP_MAPBYTE(p) (void)*(const volatile __attribute__((address_space(3))) uint8_t *)(p)
#endif
#endif

#ifdef SAT_DIAG_PERTURB
#define SAT_R10(x) x x x x x x x x x x
namespace satk {
__device__ __forceinline__ void perturb(uint32_t (&d)[4])
{
#if SAT_DIAG_PERTURB == 1
    asm volatile(SAT_R10("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n")
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
#elif SAT_DIAG_PERTURB == 2
    uint32_t s0 = __builtin_amdgcn_readfirstlane(d[1]), s1 = __builtin_amdgcn_readfirstlane(d[2]);
    asm volatile(SAT_R10("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 3\n")
                 : "+s"(s0), "+s"(s1) : : "scc");
    d[1] = s0; d[2] = s1;
#elif SAT_DIAG_PERTURB == 3
    uint32_t addr = (threadIdx.x & 63u) << 2, t0, t1;
    asm volatile("ds_read_b32 %0, %2\n ds_read_b32 %1, %2 offset:256\n ds_read_b32 %0, %2 offset:512\n ds_read_b32 %1, %2 offset:768\n"
                 "ds_read_b32 %0, %2 offset:1024\n ds_read_b32 %1, %2 offset:1280\n ds_read_b32 %0, %2 offset:1536\n"
                 "ds_read_b32 %1, %2 offset:1792\n ds_read_b32 %0, %2 offset:2048\n ds_read_b32 %1, %2 offset:2304\n s_waitcnt lgkmcnt(0)\n"
                 : "=&v"(t0), "=&v"(t1) : "v"(addr) : "memory");
    d[0] ^= t0 & t1 & 0x80000000u;
#elif SAT_DIAG_PERTURB == 4
    asm volatile(SAT_R10("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"));
#elif SAT_DIAG_PERTURB == 5
    asm volatile(SAT_R10("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n")
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
#endif
}
}  // namespace satk
#define SAT_DIAG_PERTURB_INIT uint32_t pert[4] = { (uint32_t)lane_id, 1u, 2u, 3u }
#define SAT_DIAG_PERTURB_STEP satk::perturb(pert)
// (keeps the perturbed values alive to the end of the kernel)
#define SAT_DIAG_PERTURB_END do { if ((pert[0] ^ pert[1] ^ pert[2] ^ pert[3]) == 0xDEADBEEFu) Q.scores[e] = -1; } while (0)
#endif

#ifdef SAT_DIAG_SELFCHECK
// K.cu:1105-1134 (TESTING): the score after the move must equal the full score of the moved map.  The proposed
// image is written into the chain's map, the full score recomputed by the rows-in-step form (score_rows: every pair
// i < k from the map bytes, nothing shared with the step's delta), and the byte restored.
#define SAT_DIAG_SELFCHECK_STEP do {                                                                                     \
        const uint8_t keep_ = smap_b[map_byte_addr(ssei)];                                                               \
        smap_b[map_byte_addr(ssei)] = (uint8_t)newj;                                                                     \
        int chk_ = score_rows();                                                                                         \
        if (lpc >= 2) chk_ += __shfl_xor(chk_, 1, 64);                                                                   \
        if (lpc == 4) chk_ += __shfl_xor(chk_, 2, 64);                                                                   \
        smap_b[map_byte_addr(ssei)] = keep_;                                                                             \
        if (part == 0) { atomicAdd(a.diag + 9, 1ull); if (chk_ != newscore) atomicAdd(a.diag + 8, 1ull); }               \
    } while (0)
#endif

// ---- host side (sat_capi.hip): the counters' buffer, zeroed before the launches of a search, read back after
#if defined(SAT_DIAG_HOST)
#include <cstdio>
namespace satdiag {
inline unsigned long long *&buffer() { static unsigned long long *d = nullptr; return d; }
inline unsigned long long (&last())[16] { static unsigned long long h[16]; return h; }
inline hipError_t begin(hipStream_t stream, unsigned long long *&arg)
{
    if (!buffer()) { hipError_t e = hipMalloc(&buffer(), 16 * sizeof(unsigned long long)); if (e != hipSuccess) return e; }
    arg = buffer();
    return hipMemsetAsync(buffer(), 0, 16 * sizeof(unsigned long long), stream);
}
inline hipError_t end(hipStream_t stream)
{
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    e = hipMemcpy(last(), buffer(), sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost);
#ifdef SAT_DIAG_PHASE
    unsigned long long tot = 0;
    for (int k = 0; k < 11; k++) if (k < 8 || k == 10) tot += last()[k];
    static const char *nm[11] = { "draw+proposal", "compaction set-up", "compacted rounds", "read-back/static loops",
                                 "best tracking", "metropolis+update", "initial full score", "staging+restart loop", "", "",
                                 "thinit" };
    for (int k = 0; k < 11; k++)
        if (k < 8 || k == 10) fprintf(stderr, "phase %-24s %14llu wave-cycles %5.1f%%\n", nm[k], last()[k], tot ? 100.0 * last()[k] / tot : 0.0);
#endif
    return e;
}
}  // namespace satdiag
// diagnostic builds export the counters of the last search (not declared in include/satabsearch.h)
extern "C" void sat_diag_counters(unsigned long long out[16]) { for (int k = 0; k < 16; k++) out[k] = satdiag::last()[k]; }
#endif
