// sat_sa_kernel.hpp - the simulated-annealing tableau search kernel for gfx950 (CDNA4).
//
// One ENTRY SLOT of a workgroup scores ONE database structure against the query; every lane runs
// an independent restart chain (lane = restart, as the reference maps threadIdx to restarts,
// K.cu:1012-1015), 100 Metropolis steps each.  A workgroup is one slot, or several side by side
// (own threads, own LDS carve, shared barriers only) where that packs more entries into the CU's
// 128 LDS granules of 1280 bytes - the host decides per launch (sat_capi.hip pick_epw).  Written from scratch for
// 64-wide wavefronts and the 160 KB LDS; what it computes follows the reference
// kernel body K.cu:924-1233 (K.cu = nvcc_src_current/cudaSaTabsearch_kernel.cu).
//
// Data layout
//   Dc   (LDS) the db entry's cells {f32 distance, code byte}.  Entries of up to 32 SSEs: the full
//        n2 x (n2+1) matrix of 8-byte cells, expanded from the packed lower triangle in HBM (one
//        ds_read_b64 per pair); larger entries: the lower triangle as it is, distances and codes in two
//        arrays, addressed by (max, min) of the pair - half the LDS where the cells limit the
//        workgroups per CU (DbRow).  Column n2 (full matrix) / row n2 (triangle) is a "null"
//        SSE whose distance is the sentinel -1e30: an unmatched query SSE
//        is represented as matched to the null SSE, so |d1 - d2| <= 4 is false and the
//        pair scores 0 without any branch or predicate in the hot loop (the reference
//        tests l >= 0, old_j >= 0, k != sse_i per pair, K.cu:521-531).  The null SSE is never the
//        ROW of an evaluation: a null image contributes 0, the compacted rounds never list it, and the two
//        loops that may meet one (full score, static loops) walk row 0 and drop the sum.
//   Q    query, grouped by 4 consecutive query SSEs k (one "word" of the map) and
//        TRANSPOSED: qdist[kw*N1P + i] = float4 of dmat1[i][4kw..4kw+3],
//        qcode[kw*N1P + i] = the four code bytes tab1[i][4kw..4kw+3] packed in a dword.
//        The hot loop reads column i = the moved SSE (different per lane) of group kw
//        (same for all lanes): 16-byte and 4-byte loads from consecutive addresses.
//        Diagonal and padding distances are the sentinel +1e30, which removes the k == i
//        term (K.cu:524,530).  Read through L1/L2 (32+-SSE queries) or staged in LDS.
//        Behind the grouped arrays the blob holds the same cells once more as a dense
//        qpair[i * N1P + k] = {distance, code byte} (8 bytes): the full score of an initial map
//        walks the matched pairs (i, k) of each chain and fetches one such cell per pair.
//   qmask (LDS, launches with one-word db sets) per query SSE the db SSEs of its type,
//        tmask[qtypes[i]]: one read on the path of every SA step instead of two dependent ones.
//   smap (LDS) per-chain SSE map, one byte per query SSE, stored word-interleaved
//        smap[w*(T+1) + chain]: word w of every chain is contiguous, so a loop over a
//        uniform word reads it conflict free; the odd row stride T+1 spreads the words of
//        ONE chain over different banks for the compacted loop, where the lanes serving
//        a row read that chain's words in the same instruction.
//   bmap (LSOLN only) best map so far of every chain, same word-interleaved layout but in
//        GLOBAL memory (one slab per entry slot of the launch): it is written on improvements
//        only and read once by the winner, the resident slabs (~4 KB x a few thousand
//        slots) live in L2, and keeping it out of LDS keeps 12 entries per CU.
//
// Work compaction in the SA step (the db-scan regime is sparse: on random pairs ~25 % of
// the query SSEs are matched, the moved SSE has a real old image in 25 % and a real new
// one in 27 % of the steps, both in 10 %, neither in 58 %): instead of every lane scoring
// 2 rows x n1/4 map words for its own chain, the lanes of a wave list the rows that are real
// (ballot + mbcnt prefix -> a per-wave item table in LDS: row, moved SSE, owner chain,
// sign), then the WHOLE wave works through them in rounds: a row is served by lpi =
// ceil(n1w / 4) lanes, each taking up to four map words (word kw, kw + lpi, ...) whose loads
// are issued together, a round holds 64 / lpi rows, and each lane adds its signed sum to the
// row's accumulator (the item slot itself) with an LDS atomic.  The last few rows of a step
// go to a tail shape with one or two words per lane (more lanes per row) rather than a mostly
// empty full round.  Maps are padded to lpi * wpl words (padding = unmatched SSEs, which meet
// the query's sentinel cells), so the rounds carry no validity tests.  A wave-step then costs
// ~S*n1w/64 packed evaluations (S = real rows in the wave, ~37 of 128) instead of 2*n1w per
// lane.  With LORDER = F most rows are real and the static per-lane loops run instead.
//
// Lanes per chain (lpc = 1, 2 or 4): when the cells of a large db entry leave room for
// only a few workgroups per CU, lpc adjacent lanes run ONE chain together - every lane
// does the cheap per-step bookkeeping redundantly (same stream, same decisions), each
// takes every lpc-th map word of the pair loops and the partial sums are added across
// the lanes - so a workgroup has lpc x the waves for the same LDS.
//
// Pair scores, four at a time (quad_terms): gfx950 issues and/or/xor/add/sub/lshr/
// bitop3/f32 add at one wave64 op per ~2.4 clk and everything else (cmp, cndmask, bcnt,
// perm, shifts left, SDWA, mad) at ~4.2 clk (profiles/r01_gfx950_valu_opcode_cost.txt), so
// the four pairs of a map word are evaluated with packed byte arithmetic:
//   * code bytes are (hi << 4) | lo with hi, lo <= 7 (parsetableaux.c:13-33 uses 0..4):
//     bits 3 and 7 are free guard bits;  X = codes(db, 4 bytes) ^ codes(query, 4 bytes);
//     (X + 0x77777777) & 0x88888888 has bit 3 / bit 7 of byte s set iff the low / high
//     nibble of pair s differs;
//   * distance test: t = 4 - |d1 - d2| is >= 0 exactly when |d1 - d2| <= 4 in f32 (the
//     reference's test K.cu:432, 524, 530); the four sign bytes are gathered by v_perm;
//   * the 3-bit index (lo differs, hi differs, too far) of every pair selects one of
//     {2, 1, 1, -2, 0, 0, 0, 0} = tscord (K.cu:306-332) gated by distance, all four with
//     ONE v_perm_b32, and v_dot4_i32_i8 adds the four signed bytes to the running sum.
//
// Free-SSE bookkeeping uses bit masks instead of the reference's int revmap[] and
// 111-int candidate list (K.cu:677-714): occ = occupied db SSEs, mapped = matched
// query SSEs, tmask[t] = db SSEs of type t.
//
// Random numbers: Philox4x32-10 with rocRAND's counter layout (rocrand_init(seed, subsequence,
// offset) + rocrand4), written out in philox_block; one 4x32-bit block per TWO SA steps, addressed
// by (seed, query, db ordinal, restart, step pair): a step takes two words - one split into two
// 16-bit draws (which query SSE moves, which candidate is taken), one whole for the Metropolis
// test - see oracle/sa_oracle.h for the slot layout, which the CPU oracle restates bit for bit.
//
// Metropolis test: the reference evaluates expf((float)delta / temp) > u with glibc
// expf on the host path (K.cu:1166).  temp takes 100 values and delta is a small
// integer, so the host tabulates P[iter][-delta] = expf(-nd / temp_iter) with ITS
// libm and the kernel compares table entries: accept decisions are those of the
// host's expf, bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

// Hook points of the diagnostic builds (phase timers, perturbations, duplicated LDS accesses, the per-move
// self-check): their code lives in diag/sat_diag.hpp, which the shipped library does not include - every hook is
// nothing here.
#ifdef SAT_DIAG
#include "diag/sat_diag.hpp"
#endif
#ifndef SAT_DIAG_ARGS
#define SAT_DIAG_ARGS
#endif
#ifndef SAT_PHASE_INIT
#define SAT_PHASE_INIT
#define SAT_PHASE(k)
#define SAT_PHASE_FLUSH
#endif
#ifndef SAT_DIAG_FS_ROWS_ONLY
#define SAT_DIAG_FS_ROWS_ONLY 0
#endif
#ifndef SAT_DIAG_DUP_CELLS
#define SAT_DIAG_DUP_CELLS(row, l0, l1, l2, l3)
#endif
#ifndef SAT_DIAG_DUP_MAPWORD
#define SAT_DIAG_DUP_MAPWORD(p)
#endif
#ifndef SAT_DIAG_DUP_ATOMIC
#define SAT_DIAG_DUP_ATOMIC(p)
#endif
#ifndef SAT_DIAG_DUP_MAPBYTE
#define SAT_DIAG_DUP_MAPBYTE(p)
#endif
#ifndef SAT_DIAG_PERTURB_INIT
#define SAT_DIAG_PERTURB_INIT
#define SAT_DIAG_PERTURB_STEP
#define SAT_DIAG_PERTURB_END
#endif
#ifndef SAT_DIAG_SELFCHECK_STEP
#define SAT_DIAG_SELFCHECK_STEP
#endif

// Cell layouts of a launch, chosen from its largest entry (satk::cell_layout)
#define SAT_CELLS_FULL8 0             // full matrix, 8-byte cells {f32 distance, code}
#define SAT_CELLS_FULL5 1             // full matrix, distances and code bytes in two arrays
#define SAT_CELLS_TRI5  2             // lower triangle, distances and code bytes in two arrays
// register budget of the option-specialised kernels with one lane per chain, as resident waves per SIMD
#ifndef SAT_FAST_WAVES
#define SAT_FAST_WAVES 6
#endif
// the initial full score of two chains walked by a pair of lanes together (one-word sets, one lane per chain)
#ifndef SAT_FS_TEAMS
#define SAT_FS_TEAMS 1
#endif
#define SAT_K_MAXITER 100
#define SAT_FS_UNROLL 2               // pairs per lane and round of the full score of an initial map
#define SAT_K_STEP_BLOCK0 32          // Philox block of SA step 0 (oracle/sa_oracle.h)
#define SAT_K_EPS 1.1e-7              // K.cu:67
#define SAT_K_NO_SCORE (-99999)       // K.cu:1009
#define SAT_K_QSENT 1.0e30f            // query-side "never within 4 A" distance
#define SAT_K_DSENT (-1.0e30f)         // db-side sentinel (null SSE, non-finite input)

// One query of a batch; a launch covers (db entries of one size bucket) x (queries of one
// size class): blockIdx.x picks the entry, blockIdx.y the query.
struct SatQuery {
    const float4   *qdist;        // [N1P/4][N1P] distances of 4 consecutive query SSEs (transposed)
    const uint32_t *qcode;        // [N1P/4][N1P] their 4 code bytes
    const uint8_t  *qtypes;       // [N1P]
    const uint2    *qpair;        // [N1P][N1P] dense cells {distance, code byte} for the full score of an initial map
    int32_t         n1;
    uint32_t        pad_;
    uint64_t        seed_q;       // seed + (query ordinal << 32)
    int32_t        *scores;       // [N] this query's score row
    int8_t         *ssemaps;      // [N][n1] this query's maps, -1 = unmatched
};

struct SatKernelArgs {
    // database shard (HBM)
    const int32_t  *orders;       // [N]
    const int64_t  *cell_off;     // [N] first packed cell
    const uint8_t  *tab_tri;      // packed lower triangles, code bytes
    const float    *dist_tri;     // packed lower triangles, distances
    const uint32_t *ordinal;      // [N] db file-order ordinal (stream key)
    const int32_t  *entry_list;   // entries handled by this launch
    int32_t         n_list;       // how many
    // A workgroup holds `epw` entries side by side: entry slot s = threads [s * tpe, (s + 1) * tpe) with its
    // own LDS carve at s * lds_stride.  The slots share nothing but the three workgroup barriers; the
    // host picks epw so that the CU's 128 LDS granules of 1280 bytes hold the most entries.
    int32_t         epw, tpe;
    uint32_t        lds_stride;   // bytes, a multiple of 16
    // queries of this launch's size class
    const SatQuery *queries;
    // options
    int32_t         lorder, lsoln, maxstart;
    int32_t         lpc_shift;    // log2(lanes per chain): 0, 1 or 2
    int32_t         compact;      // 1: the SA step may use the wave-level work compaction (its LDS tables exist)
    uint32_t       *bmap_slabs;   // LSOLN: best-map slab of workgroup g at g * bmap_slab_words
    uint32_t        bmap_slab_words;
    // Metropolis table
    const float    *ptab;         // ragged rows { 2^33, 2^32 * expf(-nd / temp) for nd = 0 .. last, 0.0 }
    const int32_t  *prow;         // [100][2] = {row offset, largest tabulated -delta}
    SAT_DIAG_ARGS                 // diagnostic builds only (diag/sat_diag.hpp): their counters
};

namespace satk {

// ---------------------------------------------------------------- small bit sets
template <int W> struct Bits { uint32_t w[W]; };

template <int W> __device__ __forceinline__ Bits<W> bits_zero()
{
    Bits<W> b;
#pragma unroll
    for (int i = 0; i < W; i++) b.w[i] = 0u;
    return b;
}
// bits [0, pos) set; pos may be <= 0 or >= 32*W
template <int W> __device__ __forceinline__ Bits<W> bits_below(int pos)
{
    Bits<W> b;
    if constexpr (W == 1) {
        // one word, full-rate ops only: all-ones shifted right by 32 - clamp(pos, 0, 32), done as
        // two shifts of at most 16 so that a total of 32 really empties the word
        const int s = 32 - min(max(pos, 0), 32);          // v_med3_i32
        const int h = s >> 1;
        b.w[0] = (0xFFFFFFFFu >> h) >> (s - h);
        return b;
    }
    // several words: the word that holds `pos` gets the bits below it, the words under it are full, the rest
    // empty (pos < 0: no word is under or at it; pos >= 32 W: every word is under it) - two compares and two
    // selects per word where a clamped 64-bit shift per word cost twice that
    const int wi = pos >> 5;                                   // arithmetic shift: negative for pos < 0
    const uint32_t part = (1u << (pos & 31)) - 1u;
#pragma unroll
    for (int i = 0; i < W; i++) b.w[i] = i < wi ? 0xFFFFFFFFu : (i == wi ? part : 0u);
    return b;
}
template <int W> __device__ __forceinline__ void bits_set(Bits<W> &b, int pos)
{
#pragma unroll
    for (int i = 0; i < W; i++) b.w[i] |= ((pos >> 5) == i) ? (1u << (pos & 31)) : 0u;
}
template <int W> __device__ __forceinline__ void bits_clear(Bits<W> &b, int pos)
{
#pragma unroll
    for (int i = 0; i < W; i++) b.w[i] &= ~(((pos >> 5) == i) ? (1u << (pos & 31)) : 0u);
}
template <int W> __device__ __forceinline__ int bits_count(const Bits<W> &b)
{
    int c = 0;
#pragma unroll
    for (int i = 0; i < W; i++) c += __popc(b.w[i]);
    return c;
}
template <int W> __device__ __forceinline__ bool bits_any(const Bits<W> &b)
{
    uint32_t o = 0u;
#pragma unroll
    for (int i = 0; i < W; i++) o |= b.w[i];
    return o != 0u;
}
// clears the lowest set bit (no-op on an empty set)
template <int W> __device__ __forceinline__ void bits_drop_lowest(Bits<W> &b)
{
    bool done = false;
#pragma unroll
    for (int i = 0; i < W; i++) {
        const bool here = !done && b.w[i] != 0u;
        b.w[i] = here ? b.w[i] & (b.w[i] - 1u) : b.w[i];
        done = done || here;
    }
}
template <int W> __device__ __forceinline__ int bits_lowest(const Bits<W> &b)   // -1 if empty
{
    int r = -1;
#pragma unroll
    for (int i = W - 1; i >= 0; i--) r = b.w[i] ? 32 * i + (__ffs(b.w[i]) - 1) : r;
    return r;
}
template <int W> __device__ __forceinline__ int bits_highest(const Bits<W> &b)  // -1 if empty
{
    int r = -1;
#pragma unroll
    for (int i = 0; i < W; i++) r = b.w[i] ? 32 * i + (31 - __clz(b.w[i])) : r;
    return r;
}
// position of the r-th (0-based, ascending) set bit of a non-zero word with > r bits
__device__ __forceinline__ int word_select(uint32_t v, int r)
{
    // branch-free rank select by halving: counts of the low half decide the side
    uint32_t a = v - ((v >> 1) & 0x55555555u);
    uint32_t b = (a & 0x33333333u) + ((a >> 2) & 0x33333333u);
    uint32_t c = (b + (b >> 4)) & 0x0F0F0F0Fu;
    uint32_t d = (c + (c >> 8)) & 0x00FF00FFu;
    int pos = 0;
    int t = (int)(d & 0xFFu);
    if (r >= t) { pos = 16; r -= t; }
    t = (int)((c >> pos) & 0xFu);
    if (r >= t) { pos += 8; r -= t; }
    t = (int)((b >> pos) & 0x7u);
    if (r >= t) { pos += 4; r -= t; }
    t = (int)((a >> pos) & 0x3u);
    if (r >= t) { pos += 2; r -= t; }
    t = (int)((v >> pos) & 0x1u);
    if (r >= t) { pos += 1; }
    return pos;
}
template <int W> __device__ __forceinline__ int bits_select(const Bits<W> &b, int r)
{
    // the word that holds the r-th bit and the rank inside it first (popcounts), then ONE rank select
    uint32_t word = b.w[0];
    int base = 0, rr = r;
    bool found = false;
#pragma unroll
    for (int i = 0; i < W; i++) {
        const int c = __popc(b.w[i]);
        const bool here = !found && r < c;
        word = here ? b.w[i] : word;
        base = here ? 32 * i : base;
        rr = here ? r : rr;
        found = found || here;
        r -= c;
    }
    // (r beyond the set: callers never ask; an empty word would select position 31 of nothing)
    return found ? base + word_select(word, rr) : 0;
}

// p = the highest set bit of `mapped` at or below position `upto` (0 when there is none: `none`).  One and two
// words as 32- and 64-bit arithmetic, more words word by word.
template <int W> __device__ __forceinline__ void highest_mapped_upto(const Bits<W> &mapped, int upto, int &p, bool &none)
{
    if constexpr (W == 1) {
        const uint32_t low = mapped.w[0] & (0xFFFFFFFFu >> (31 - upto));
        p = 31 ^ __builtin_clz(low | 1u);
        none = low == 0u;
    } else if constexpr (W == 2) {
        const unsigned long long m = (unsigned long long)mapped.w[0] | ((unsigned long long)mapped.w[1] << 32);
        const unsigned long long low = m & (~0ull >> (63 - upto));
        p = 63 ^ __builtin_clzll(low | 1ull);
        none = low == 0ull;
    } else {
        Bits<W> lowpart, below = bits_below<W>(upto + 1);
#pragma unroll
        for (int w = 0; w < W; w++) lowpart.w[w] = mapped.w[w] & below.w[w];
        p = bits_highest<W>(lowpart);
        none = p < 0;
        p = none ? 0 : p;
    }
}

// ---------------------------------------------------------------- pair scores
// Sum of the four pair scores of one map word: query SSEs 4kw..4kw+3 against the db SSEs
// in `word` (one byte each), all on db row `row` (the image of the moved / anchor SSE).
//   qd, qc   the query group's distances and code bytes for this lane's column
//   force    0x04 in byte s forces pair s to score 0 (used by the full score for k <= i)
// Returns acc + sum.  See the file header for the arithmetic.
// A row of the db entry's cell matrix in LDS, in one of three layouts picked per LAUNCH from its largest entry
// (cell_layout):
//   FULL8  entries of up to 32 SSEs: the full matrix of 8-byte cells {f32 distance, code byte}, one ds_read_b64 per
//          pair at row base + image;
//   FULL5  up to 48 SSEs: the full matrix, distances and code bytes in two arrays (5 bytes per cell, two reads per
//          pair): 37 % less LDS where the cells start to limit the workgroups per CU;
//   TRI5   above 48 SSEs: only the lower TRIANGLE (the matrix is symmetric), two arrays: a 96-SSE entry takes 23.8 KB
//          where the full split matrix took 46.6 KB, i.e. 4-5 resident workgroups per CU instead of 2-3, for ~17 cycles
//          of index arithmetic per pair (tri_index).  Measured per entry order (profiles/r03_cost_by_order.txt): the
//          triangle loses 10-19 % at 40 and 48 SSEs, where the LDS does not limit the occupancy and the index
//          arithmetic is pure cost, and wins from 56 SSEs on (-5 % at 64, -12 % at 88, -21 % at 111 under a 32-SSE
//          query; up to -44 % under an 8-SSE query).  The null SSE is row n2 of the triangle (n2 + 1 sentinel
//          cells): an unmatched image l = n2 is the larger index of every pair it appears in.
template <int CELLS> struct DbRow;
template <> struct DbRow<SAT_CELLS_FULL8> { const uint2 *cells; };
template <> struct DbRow<SAT_CELLS_FULL5> { const float *dist; const uint8_t *code; };
template <> struct DbRow<SAT_CELLS_TRI5> { const float *dist; const uint8_t *code; int j; };
__host__ __device__ inline int cell_layout(int n2max) { return n2max <= 32 ? SAT_CELLS_FULL8 : (n2max <= 48 ? SAT_CELLS_FULL5 : SAT_CELLS_TRI5); }
// cell (j, l) of the lower triangle: row max(j, l), column min(j, l)
__device__ __forceinline__ int tri_index(int j, int l)
{
    const int mx = max(j, l), mn = min(j, l);
    return (int)((__umul24((uint32_t)mx, (uint32_t)mx) + (uint32_t)mx) >> 1) + mn;
}
__host__ __device__ inline uint32_t tri_cells(int n2) { return (uint32_t)(n2 + 1) * (uint32_t)(n2 + 2) / 2u; }   // rows 0 .. n2
// the distance bits and the code byte of cell (row, l)
template <int CELLS> __device__ __forceinline__ uint2 db_cell(const DbRow<CELLS> row, int l)
{
    if constexpr (CELLS == SAT_CELLS_FULL8) return row.cells[l];
    else if constexpr (CELLS == SAT_CELLS_FULL5) return uint2{ __float_as_uint(row.dist[l]), row.code[l] };
    else { const int c = tri_index(row.j, l); return uint2{ __float_as_uint(row.dist[c]), row.code[c] }; }
}

template <int CELLS>
__device__ __forceinline__ int quad_terms(const float4 qd, const uint32_t qc, const DbRow<CELLS> row,
                                          const uint32_t word, const uint32_t force, const int acc)
{
    const uint32_t l0 = word & 0xFFu, l1 = (word >> 8) & 0xFFu, l2 = (word >> 16) & 0xFFu, l3 = word >> 24;
    SAT_DIAG_DUP_CELLS(row, l0, l1, l2, l3);
    const uint2 d0 = db_cell<CELLS>(row, (int)l0), d1 = db_cell<CELLS>(row, (int)l1), d2 = db_cell<CELLS>(row, (int)l2),
                d3 = db_cell<CELLS>(row, (int)l3);
    // sign bit of t = "distances differ by more than 4 A"
    const float t0 = 4.0f - fabsf(qd.x - __uint_as_float(d0.x));
    const float t1 = 4.0f - fabsf(qd.y - __uint_as_float(d1.x));
    const float t2 = 4.0f - fabsf(qd.z - __uint_as_float(d2.x));
    const float t3 = 4.0f - fabsf(qd.w - __uint_as_float(d3.x));
    // v_perm_b32(S0, S1, sel): selector 0-3 = byte of S1, 4-7 = byte of S0, 0x0C = zero
    const uint32_t far = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x0C0C0703u) |
                         __builtin_amdgcn_perm(__float_as_uint(t3), __float_as_uint(t2), 0x07030C0Cu);
    const uint32_t x = __builtin_amdgcn_perm(d1.y, d0.y, 0x0C0C0400u) ^
                       __builtin_amdgcn_perm(d3.y, d2.y, 0x04000C0Cu) ^ qc;
    const uint32_t z = (x + 0x77777777u) & 0x88888888u;             // bit 3: low nibbles differ, bit 7: high
    // ((z >> 3) | (z >> 6)) & 0x03030303 and the merge of the distance bits as two v_bitop3_b32 (full
    // rate; the and-or / or3 forms the compiler picks for the plain expression issue at half rate)
    uint32_t sel = __builtin_amdgcn_bitop3_b32(z >> 3, z >> 6, 0x03030303u, 0xA8);        // (a | b) & c
    sel = __builtin_amdgcn_bitop3_b32(far >> 5, 0x04040404u, sel, 0xEA);                   // (a & b) | c
    sel |= force;
    const uint32_t terms = __builtin_amdgcn_perm(0u, 0xFE010102u, sel);   // {2, 1, 1, -2 | 0, 0, 0, 0}
    return __builtin_amdgcn_sdot4((int)terms, 0x01010101, acc, false);
}

// One pair score (the full score of an initial map walks the matched pairs one by one): query cell
// {distance, code byte}, db cell likewise; same arithmetic as one byte lane of quad_terms.
__device__ __forceinline__ int pair_term(const uint32_t qd_bits, const uint32_t qc, const uint32_t dd_bits, const uint32_t dc)
{
    const float t = 4.0f - fabsf(__uint_as_float(qd_bits) - __uint_as_float(dd_bits));   // sign bit: more than 4 A apart
    const uint32_t z = ((qc ^ dc) + 0x77u) & 0x88u;                      // bit 3: low nibbles differ, bit 7: high
    uint32_t sel = __builtin_amdgcn_bitop3_b32(z >> 3, z >> 6, 0x3u, 0xA8);                  // (a | b) & c
    sel = __builtin_amdgcn_bitop3_b32(__float_as_uint(t) >> 29, 0x4u, sel, 0xEA);             // (a & b) | c
    const uint32_t terms = __builtin_amdgcn_perm(0u, 0xFE010102u, sel);  // byte 0 = {2, 1, 1, -2 | 0, 0, 0, 0}[sel]
    return (int)(int8_t)(terms & 0xFFu);
}

// ---------------------------------------------------------------- random streams
// Block `block` of the chain's Philox stream, written out by hand (NOT a call into rocRAND), laid out as rocRAND's:
// key = seed_q, counter = (block, 0, subsequence lo, subsequence hi).
__device__ __forceinline__ uint4 philox_block(uint64_t seed_q, uint64_t subsequence, uint32_t block)
{
    // Philox4x32-10 written out (same words as rocrand_init(seed_q, subsequence, 4 * block) +
    // rocrand4 - checked on the device against rocRAND's own device API by
    // tests/test_gpu_parity.py::test_philox_block_is_rocrands_block): in every use here the block and the low
    // subsequence word are wave-uniform, so rounds 1-3 are left to the compiler (it keeps the
    // uniform half on the scalar unit); from round 4 on all four words are per lane and the two
    // three-way XORs of a round are one v_bitop3_b32 each.
    uint32_t c0 = block, c1 = 0u, c2 = (uint32_t)subsequence, c3 = (uint32_t)(subsequence >> 32);
    uint32_t k0 = (uint32_t)seed_q, k1 = (uint32_t)(seed_q >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0, n2;
        if (r < 3) {
            n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
            n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        } else {
            n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
            n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        }
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return uint4{ c0, c1, c2, c3 };
}
// 2^32 * (uniform draw of word v): rocRAND's uniform_distribution is 2^-32 + float(v) * 2^-32
// in (0, 1] (rocrand_uniform.h:65-68); scaling by a power of two is exact, so float(v) + 1.0f is
// that value times 2^32 with the same two roundings.
__device__ __forceinline__ float draw32(uint32_t v)
{
    return (float)v + 1.0f;
}
__device__ __forceinline__ float to_uniform(uint32_t v)
{
    return draw32(v) * 2.3283064365386963e-10f;
}
// Index draw from a 16-bit value v: u = (v + 1) * 2^-16 in (0, 1] (exact in float), index =
// (int)((u - EPS) * n) evaluated in double, as K.cu:1042 and K.cu:710 do.  For n <= 111 that equals the integer
// ((v + 1) * n - 1) >> 16: when (v + 1) * n is a multiple of 2^16 the EPS term drops the index by
// one, otherwise the fractional part is at least 2^-16 > EPS * n and nothing changes
// (tests/test_oracle_units.py checks all 65536 x 111 cases against the double expression).
// nm1 = max(n - 1, 0); n = 0 gives 0.
__device__ __forceinline__ int scaled_index16(uint32_t v16, int n, int nm1)
{
    return (int)(__umul24(v16, (uint32_t)n) + (uint32_t)nm1) >> 16;
}

// Work compaction (SA step): a listed row is served by `lpi` lanes, each taking `wpl` <= 4 map
// words (word kw, kw + lpi, ...), so the map of a chain is padded to wpl * lpi >= n1w words.
__host__ __device__ inline void compaction_shape(int n1w, int &lpi, int &wpl)
{
    lpi = (n1w + 3) >> 2;
    wpl = lpi > 0 ? (n1w + lpi - 1) / lpi : 1;
    if (lpi < 1) lpi = 1;
}
__host__ __device__ inline int map_words(int n1w)
{
    int lpi, wpl;
    compaction_shape(n1w, lpi, wpl);
    return lpi * wpl;
}

// LDS carve of one workgroup, byte offsets from the dynamic-LDS base.  ONE function for the kernel
// (its own entry's n2, its own query's padded map words) and for the host (the launch's largest):
// the two cannot disagree, every offset keeps the alignment its users need (cells 16, query cells
// 16, 64-bit reduction keys / the LSOLN leader key 8: a 64-bit LDS atomic on a 4-byte aligned
// address faults), and the total grows with n2 and with the map words, so a workgroup sized for
// the launch's largest member holds every member.
struct LdsLayout {
    uint32_t code;        // split cells only: the code bytes (distances start at 0)
    uint32_t qdist;       // query cells staged in LDS (QLDS): float4 groups ...
    uint32_t qcode;       // ... and their code dwords
    uint32_t smap;        // chain maps, word-interleaved [word][chain], row stride chains + 1
    uint32_t tmask;       // [4 types][4 words] db SSEs of a type
    uint32_t qtypes;      // query SSE types
    uint32_t qmask;       // one-word db sets only: per query SSE the mask of the db SSEs of its type (tmask[qtypes[i]])
    uint32_t leader;      // the LSOLN leader key (64-bit)
    uint32_t red;         // the waves' arg-max keys (64-bit), red_stride bytes apart
    uint32_t red_stride;
    uint32_t items;       // per-wave item tables of the work compaction
    uint32_t total;
};
// m2w = 32-bit words of a db-side bit set in this launch's size class (1, 2 or 4); cells = its cell layout
// (SAT_CELLS_*: DbRow).
__host__ __device__ inline LdsLayout lds_layout(int m2w, int cells, int n2, int words, int n1p, int chains, int threads,
                                                 bool q_in_lds, bool compact)
{
    LdsLayout L;
    const bool split = cells != SAT_CELLS_FULL8;
    // full matrix: rows 0 .. n2-1, columns 0 .. n2: the null SSE (index n2) has a column - map bytes
    // of unmatched query SSEs point at it - but no row: a null image scores 0 and its row is never summed
    uint32_t dcells = (uint32_t)n2 * (uint32_t)(n2 + 1);
    uint32_t off;
    if (split) {                                              // 4-byte distances + 1-byte codes; TRI5: lower triangle incl. the null row
        if (cells == SAT_CELLS_TRI5) dcells = tri_cells(n2);
        dcells = (dcells + 3u) & ~3u;
        L.code = dcells * 4u;
        off = L.code + ((dcells + 15u) & ~15u);
    } else {                                                  // 8-byte cells
        dcells = (dcells + 1u) & ~1u;
        L.code = 0u;
        off = dcells * 8u;
    }
    L.qdist = off;                                            // 16-byte aligned in both layouts
    if (q_in_lds) off += (uint32_t)words * (uint32_t)n1p * 16u;
    L.qcode = off;
    if (q_in_lds) off += (uint32_t)words * (uint32_t)n1p * 4u;
    L.smap = off;
    // an even word count keeps what follows 8-byte aligned
    off += (((uint32_t)words * (uint32_t)(chains + 1) + 1u) & ~1u) * 4u;
    L.tmask = off;
    off += 4u * (uint32_t)m2w * 4u;                           // [4 types][m2w words]
    L.qtypes = off;
    off += ((uint32_t)n1p + 15u) & ~15u;
    L.qmask = off;
    if (m2w == 1) off += (uint32_t)n1p * 4u;
    off = (off + 7u) & ~7u;
    L.leader = off;
    off += 8u;
    // the arg-max key of wave w: with item tables, the first 8 bytes of the wave's own table (it is done
    // with the table by then, and no other wave touches it); without, an array of 16 keys
    L.red = off;
    L.red_stride = compact ? 256u : 8u;
    if (!compact) off += 16u * 8u;
    L.items = off;
    if (compact) off += (uint32_t)((threads + 63) / 64) * 64u * 4u;      // compaction handles <= 64 rows per wave
    L.total = off;
    return L;
}

// LDS byte size of one workgroup (host side: the launch's largest query and entry)
__host__ __device__ inline size_t lds_bytes(int n1, int n1p, int n2, int chains, int threads, bool lsoln, bool q_in_lds,
                                             bool compact)
{
    (void)lsoln;                                              // the best maps live in global memory
    return lds_layout(n2 <= 32 ? 1 : (n2 <= 64 ? 2 : 4), cell_layout(n2), n2, map_words((n1 + 3) >> 2), n1p, chains, threads, q_in_lds, compact).total;
}

}  // namespace satk


// N1P: pitch of the query cell matrix (>= 4*ceil(n1/4)); M2W: 32-bit words of a db-side
// bit set (n2 <= 32*M2W); QLDS: query cells staged in LDS (else read through L1/L2).
// OPT: >= 0: bit 0 LORDER, bit 1 LSOLN, bits 2-3 log2(lanes per chain) as compile-time facts (lanes per
// chain above one only instantiated for the largest entries, M2W = 4, where that layout is the default).
// -1 = every option is read from the arguments (the general instantiation: forced layouts, several lanes
// per chain, forced layouts); otherwise the options are compile-time facts - bit 0 LORDER, bit 1
// LSOLN, one lane per chain, work compaction exactly when LORDER - and their tests leave the SA
// step loop (the general kernel spills ~90 SGPRs and is ~9 % slower on the bench shape).
// WPL: map words per lane in the compacted rounds (satk::compaction_shape) when every query of
// the launch has the same; 0 = read it from the query (a four-way switch per step).
// CELLS: the launch's cell layout (SAT_CELLS_*, satk::cell_layout of its largest entry).
template <int N1P, int M2W, bool QLDS, int OPT, int WPL, int CELLS>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu((OPT < 0 || OPT >= 4) ? 4 : SAT_FAST_WAVES)))
sat_sa_kernel(const SatKernelArgs a)
{
    using namespace satk;
    constexpr int M1W = (N1P + 31) / 32;
    extern __shared__ __align__(16) unsigned char lds_raw[];

    // entry slot of this wave (wave-uniform: tpe is a multiple of 64) and the lane inside it
    const int nthreads = a.tpe;
    const int wave_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int slot = a.epw > 1 ? wave_wg / (nthreads >> 6) : 0;
    const int wlane = (int)(threadIdx.x & 63u);
    const int lane_id = ((wave_wg - slot * (nthreads >> 6)) << 6) | wlane;
    __builtin_assume(lane_id >= 0 && lane_id < 1024);
    const uint32_t lds_base = (uint32_t)slot * a.lds_stride;
    unsigned char *const lds_slot = lds_raw + lds_base;
    // chain = restart slot of this lane; `part` of `lpc` adjacent lanes share one chain
    constexpr bool FAST = OPT >= 0;
    const int lpc_shift = FAST ? (OPT >> 2) : a.lpc_shift;      // OPT bits 2-3: log2(lanes per chain), 0..2
    const bool opt_lorder = FAST ? (OPT & 1) != 0 : a.lorder != 0;
    const bool opt_compact = FAST ? (OPT & 1) != 0 : a.compact != 0;
    const int lpc = 1 << lpc_shift;
    const int tid = lane_id >> lpc_shift;         // chain index inside the workgroup
    const int part = lane_id & (lpc - 1);
    const int T = nthreads >> lpc_shift;          // chains per workgroup
    // the last workgroup's spare slots repeat the last entry (same result, written twice)
    const int list_pos = (int)blockIdx.x * a.epw + slot;
    const int e = a.entry_list[min(list_pos, a.n_list - 1)];
    const SatQuery Q = a.queries[blockIdx.y];
    const int n1 = Q.n1;
    const int n2 = a.orders[e];
    const int n2p = n2 + 1;
    const int n1w = (n1 + 3) >> 2;
    const int NULLJ = n2;                       // the null db SSE
    const bool lsoln = FAST ? (OPT & 2) != 0 : a.lsoln != 0;

    int cmp_lpi_q, cmp_wpl_q;
    compaction_shape(n1w, cmp_lpi_q, cmp_wpl_q);
    // lanes per listed row = ceil(n1w / 4): a compile-time fact in the two small query classes (1 for up to
    // 16 SSEs, 2 for 17..32), which turns the word strides of the rounds into instruction offsets
    constexpr int LPI_CT = N1P == 16 ? 1 : (N1P == 32 ? 2 : 0);
    const int cmp_lpi = LPI_CT ? LPI_CT : cmp_lpi_q;
    const int cmp_wpl = WPL > 0 ? WPL : cmp_wpl_q;           // the host launches WPL > 0 only where it matches
    const int cmp_words = cmp_lpi * cmp_wpl;                 // words n1w .. cmp_words - 1 stay "unmatched"
    // ---- carve LDS: satk::lds_layout, the function the host sizes the workgroup with.  The cell layout
    // goes by the launch's size class, not by this entry's order (n2max > 32 <=> M2W > 1).
    constexpr bool SPLIT = CELLS != SAT_CELLS_FULL8;
    const LdsLayout lay = lds_layout(M2W, CELLS, n2, cmp_words, N1P, T, nthreads, QLDS, opt_compact);
    uint2 *Dc = reinterpret_cast<uint2 *>(lds_slot);                          // !SPLIT: 8-byte cells
    float *distL = reinterpret_cast<float *>(lds_slot);                       // SPLIT: distances ...
    uint8_t *codeL = lds_slot + lay.code;                                     // ... and code bytes
    auto db_row = [&](int j) -> DbRow<CELLS> {
        if constexpr (CELLS == SAT_CELLS_TRI5) return DbRow<SAT_CELLS_TRI5>{ distL, codeL, j };
        else if constexpr (CELLS == SAT_CELLS_FULL5) return DbRow<SAT_CELLS_FULL5>{ distL + __mul24(j, n2p), codeL + __mul24(j, n2p) };
        else return DbRow<SAT_CELLS_FULL8>{ Dc + __mul24(j, n2p) };
    };
    // query groups in LDS cover the padding words too (sentinel cells, like every group past n1w)
    float4 *qdistL = reinterpret_cast<float4 *>(lds_slot + lay.qdist);
    uint32_t *qcodeL = reinterpret_cast<uint32_t *>(lds_slot + lay.qcode);
    uint32_t *smap = reinterpret_cast<uint32_t *>(lds_slot + lay.smap);
    // map word w of chain c lives at w*TP + c with TP = T + 1: the odd stride puts the words of
    // one chain in different banks (the compacted loop reads them from several lanes at once) and
    // keeps word w of all chains contiguous for the static loops
    const int TP = T + 1;
    uint32_t *tmask = reinterpret_cast<uint32_t *>(lds_slot + lay.tmask);
    // best maps: word w of chain c at w*T + c of this workgroup's slab (global memory)
    uint32_t *bmap = lsoln ? a.bmap_slabs + ((size_t)blockIdx.y * gridDim.x * a.epw + list_pos) * a.bmap_slab_words : nullptr;
    uint8_t *qtypes = lds_slot + lay.qtypes;
    // M2W == 1: the candidate mask of query SSE i by ONE LDS read (tmask[qtypes[i]] is two, one after the other,
    // on the path of every SA step)
    uint32_t *qmask = reinterpret_cast<uint32_t *>(lds_slot + lay.qmask);
    unsigned char *red_b = lds_slot + lay.red;
    auto red_key = [&](int w) -> unsigned long long * { return reinterpret_cast<unsigned long long *>(red_b + (uint32_t)w * lay.red_stride); };
    constexpr int TMS = M2W;                                  // words per type of the type masks
    // explicit LDS address space: these two are written by some lanes and read by others of the
    // same wave between wavefront-scope fences, and must stay ds_* instructions
    typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
    typedef __attribute__((address_space(3))) int32_t lds_i32_t;
    const uint32_t items_off = lds_base + lay.items;
    // LSOLN: key (score, restart) of the best proposal any chain of the workgroup has recorded so
    // far, same form as the final arg-max key.  A chain copies its map out only when its new best
    // beats this leader: the map that is finally output belongs to the chain with the largest key,
    // and that chain's last own-best proposal always beats every key recorded before it (a stale,
    // lower leader only causes a spare copy).  ~1150 copies per workgroup become ~20.
    typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
    lds_u64_t *leader = (lds_u64_t *)(uintptr_t)(lds_base + lay.leader);
    auto beats_leader = [&](int sc, int restart_) -> bool {
        const unsigned long long key = (((unsigned long long)(uint32_t)(sc + 0x40000000)) << 32) | (0xFFFFFFFFu - (uint32_t)restart_);
        if (key <= *leader) return false;
        __hip_atomic_fetch_max(leader, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return true;
    };
    lds_u32_t *items = (lds_u32_t *)(uintptr_t)(uint32_t)(items_off + (uint32_t)(lane_id >> 6) * 256u);
    // query group (4 distances, 4 code bytes) of column `col`: from LDS, or from global memory
    // through L1 - the descriptor's pointers are cast to the global address space so that the
    // loads are global_load (a pointer read from memory is otherwise a generic "flat" pointer)
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) f32x4_t *gptr_f4;
    typedef const __attribute__((address_space(1))) uint32_t *gptr_u32;
    const gptr_f4 qdistG = (gptr_f4)(uintptr_t)Q.qdist;
    const gptr_u32 qcodeG = (gptr_u32)(uintptr_t)Q.qcode;
    typedef const __attribute__((address_space(1))) char *gptr_c;
    typedef const __attribute__((address_space(4))) int32_t *cptr_i32;
    const cptr_i32 prowC = (cptr_i32)(uintptr_t)a.prow;
    typedef const __attribute__((address_space(1))) float *gptr_f32;
    const gptr_f32 ptabG = (gptr_f32)(uintptr_t)a.ptab;
    // uniform 64-bit base + 32-bit byte offset: the saddr form of global_load, no 64-bit VALU math
    // (byte offsets: off16 = 16 * group index, off4 = 4 * group index.  The callers build them from a
    // per-lane base plus constants, so that the words of a round differ by instruction offsets only.)
    auto load_qdist = [&](uint32_t off16) -> float4 {
        if constexpr (QLDS) return *reinterpret_cast<const float4 *>(reinterpret_cast<const unsigned char *>(qdistL) + off16);
        else {
            const f32x4_t v = *(gptr_f4)((gptr_c)qdistG + off16);
            return float4{ v.x, v.y, v.z, v.w };
        }
    };
    auto load_qcode = [&](uint32_t off4) -> uint32_t {
        if constexpr (QLDS) return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(qcodeL) + off4);
        else return *(gptr_u32)((gptr_c)qcodeG + off4);
    };
    // the same cells for a wave-uniform index (the full score walks the query in step for all
    // chains): through the constant address space these are scalar loads into scalar registers
    typedef const __attribute__((address_space(4))) f32x4_t *cptr_f4;
    typedef const __attribute__((address_space(4))) uint32_t *cptr_u32;
    const cptr_f4 qdistC = (cptr_f4)(uintptr_t)Q.qdist;
    const cptr_u32 qcodeC = (cptr_u32)(uintptr_t)Q.qcode;
    auto load_qdist_uniform = [&](uint32_t idx) -> float4 {
        if constexpr (QLDS) return qdistL[idx];
        else {
            const f32x4_t v = qdistC[idx];
            return float4{ v.x, v.y, v.z, v.w };
        }
    };
    auto load_qcode_uniform = [&](uint32_t idx) -> uint32_t {
        if constexpr (QLDS) return qcodeL[idx];
        else return qcodeC[idx];
    };

    SAT_PHASE_INIT;
    // ---- stage the db entry: packed lower triangle (HBM) -> full cell matrix (LDS).  Row r of the triangle (r + 1
    // cells) and row n2 - 1 - r (n2 - r cells) are n2 + 1 cells together: the waves take such row pairs in turn
    // and the lanes the n2 + 1 positions, so consecutive lanes read consecutive triangle cells, every triangle cell
    // is read ONCE and written to both mirror positions, and no lane divides (the first version walked the
    // n2 (n2 + 1) cells of the full matrix: a division, and a gather of the mirrored triangle cell, per cell).
    // Launches with the triangle layout (DbRow) keep the triangle as it is: a straight copy.
    {
        const uint8_t *tt = a.tab_tri + a.cell_off[e];
        const float *dd = a.dist_tri + a.cell_off[e];
        // NaN / inf never pass the reference's |d1 - d2| <= 4 either: same as the sentinel
        auto clean = [](float v) -> uint32_t { return __float_as_uint(fabsf(v) <= 3.0e38f ? v : SAT_K_DSENT); };
        auto put = [&](int c, uint32_t dist_bits, uint32_t code) {
            if constexpr (SPLIT) {
                distL[c] = __uint_as_float(dist_bits);
                codeL[c] = (uint8_t)code;
            } else {
                Dc[c] = uint2{ dist_bits, code };
            }
        };
        if constexpr (CELLS == SAT_CELLS_TRI5) {
            // the triangle as it lies in HBM, then the null row: n2 + 1 cells that never pass the distance test
            // (four cells per lane and trip: the entry's first cell sits at any cell index, so the 16 bytes of
            // distances are only dword aligned and the 4 code bytes not at all - global memory takes both; their
            // LDS images start 16-byte aligned)
            const int ncell = (n2 * n2p) >> 1;
            typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(4)));
            typedef uint32_t u32u_t __attribute__((aligned(1)));
            for (int t = lane_id << 2; t + 3 < ncell; t += nthreads << 2) {
                const f32x4u_t v = *reinterpret_cast<const f32x4u_t *>(dd + t);
                const uint32_t c4 = *reinterpret_cast<const u32u_t *>(tt + t);
                *reinterpret_cast<uint4 *>(distL + t) = uint4{ clean(v.x), clean(v.y), clean(v.z), clean(v.w) };
                *reinterpret_cast<uint32_t *>(codeL + t) = c4;
            }
            for (int t = (ncell & ~3) + lane_id; t < ncell; t += nthreads) put(t, clean(dd[t]), tt[t]);
            for (int x = lane_id; x <= n2; x += nthreads) put(ncell + x, __float_as_uint(SAT_K_DSENT), 0u);
        } else {
            const int swave = lane_id >> 6, swaves = nthreads >> 6;
            const int pairs = (n2 + 1) >> 1;               // an odd order's middle row pairs with itself: taken once
            for (int r = swave; r < pairs; r += swaves) {
                const int rb = n2 - 1 - r;
                for (int x = wlane; x <= n2; x += 64) {
                    const bool first = x <= r;
                    if (!first && rb == r) continue;
                    const int hi = first ? r : rb, lo = first ? x : x - r - 1;
                    const int t = ((hi * (hi + 1)) >> 1) + lo;
                    const uint32_t dist_bits = clean(dd[t]), code = tt[t];
                    put(__mul24(hi, n2p) + lo, dist_bits, code);
                    if (lo != hi) put(__mul24(lo, n2p) + hi, dist_bits, code);
                }
            }
            // the null SSE's column: never passes the distance test
            for (int j = lane_id; j < n2; j += nthreads) put(__mul24(j, n2p) + n2, __float_as_uint(SAT_K_DSENT), 0u);
        }
        if (lane_id < 4 * TMS) tmask[lane_id] = 0u;
        if (lane_id == 0) *reinterpret_cast<unsigned long long *>(lds_slot + lay.leader) = 0ull;   // LSOLN leader key
        for (int i = lane_id; i < N1P; i += nthreads) qtypes[i] = Q.qtypes[i];
        if (QLDS) {
            const int groups = cmp_words * N1P;
            for (int c = lane_id; c < groups; c += nthreads) {
                qdistL[c] = Q.qdist[c];
                qcodeL[c] = Q.qcode[c];
            }
        }
    }
    __syncthreads();
    for (int j = lane_id; j < n2; j += nthreads) {
        int t = a.tab_tri[a.cell_off[e] + (int64_t)j * (j + 1) / 2 + j] & 3;   // diagonal = SSE type
        atomicOr(&tmask[t * TMS + (j >> 5)], 1u << (j & 31));
    }
    __syncthreads();
    if constexpr (M2W == 1) {
        for (int i = lane_id; i < N1P; i += nthreads) qmask[i] = tmask[qtypes[i] & 3];
        __syncthreads();
    }

    uint8_t *smap_b = reinterpret_cast<uint8_t *>(smap);
    uint8_t *bmap_b = reinterpret_cast<uint8_t *>(bmap);
    auto bmap_byte_addr = [&](int k) -> int { return (__mul24(k >> 2, T) + tid) * 4 + (k & 3); };
    // byte k of this lane's map lives at ((k>>2)*T + tid)*4 + (k&3)
    const int T4 = TP << 2, tid4 = tid << 2;
    auto map_byte_addr = [&](int k) -> int { return __mul24(k >> 2, T4) + tid4 + (k & 3); };

    // Full score of this lane's chain in the rows-in-step form (tmscord, K.cu:396-440): every lane walks all n1 rows of
    // its chain, the wave reads the query cells with scalar loads; lanes that share a chain split the words and
    // the caller adds their sums.
    auto score_rows = [&]() -> int {
        int total = 0;
        for (int i = 0; i < n1 - 1; i++) {
            // an unmatched SSE has no row in LDS: its lane walks row 0 and drops the sum
            const int j = smap_b[map_byte_addr(i)];
            const bool jreal = j != NULLJ;
            const DbRow<CELLS> drow = db_row(jreal ? j : 0);
            int rowsum = 0;
            auto row_group = [&](int kw) {
                // pairs with k <= i inside the first word are switched off (mask from i and kw)
                const int below = i + 1 - 4 * kw;
                const uint32_t force = below <= 0 ? 0u : (0x04040404u >> (8 * (4 - below)));
                const uint32_t qi = (uint32_t)(kw * N1P + i);
                rowsum = quad_terms(load_qdist(qi << 4), load_qcode(qi << 2), drow, smap[kw * TP + tid], force, rowsum);
            };
            // one lane per chain: the group index stays in scalar registers, and so do the query cells
            if (lpc == 1) {
                for (int kw = (i + 1) >> 2; kw < n1w; kw++) {
                    const int below = i + 1 - 4 * kw;
                    const uint32_t force = below <= 0 ? 0u : (0x04040404u >> (8 * (4 - below)));
                    const uint32_t qi = (uint32_t)(kw * N1P + i);
                    rowsum = quad_terms(load_qdist_uniform(qi), load_qcode_uniform(qi), drow, smap[kw * TP + tid], force, rowsum);
                }
            } else {
                for (int kw = ((i + 1) >> 2) + part; kw < n1w; kw += lpc) row_group(kw);
            }
            total += jreal ? rowsum : 0;
        }
        return total;
    };

    const uint64_t subseq_lo = (uint64_t)a.ordinal[e];
    int best = SAT_K_NO_SCORE;
    uint32_t best_restart = 0xFFFFFFFFu;
    bool any = false;

    // ---- work compaction constants (see the SA step).  A listed row is served by cmp_lpi lanes,
    // each taking cmp_wpl <= 4 map words: the loads of a lane's words are in flight together and a
    // round holds 64 / cmp_lpi rows, so a typical step is one or two rounds.  lane / cmp_lpi by a
    // 16-bit reciprocal (exact for lane <= 64); lane -> (item of the round, first map word).
    const int cmp_recip = (65536 + cmp_lpi - 1) / cmp_lpi;
    const int per_round = (64 * cmp_recip) >> 16;
    const int sub = __mul24(wlane, cmp_recip) >> 16, kw = wlane - __mul24(sub, cmp_lpi);
    const bool lane_ok = sub < per_round;
    // tail shapes: one word per lane (n1w lanes per row) and two words per lane, used for the last
    // rows of a step when they fit one round of that shape; the two-word shape only if its padded
    // word count stays inside the map's
    const int tail1_recip = (65536 + n1w - 1) / n1w, tail1_rows = (64 * tail1_recip) >> 16;
    const int tail2_lpi = (n1w + 1) >> 1;
    const int tail2_recip = (65536 + tail2_lpi - 1) / tail2_lpi;
    const int tail2_rows = 2 * tail2_lpi <= cmp_words ? (64 * tail2_recip) >> 16 : 0;
    const uint32_t nullword = (uint32_t)NULLJ * 0x01010101u;     // a map word of unmatched SSEs
    SAT_PHASE(7);                         // staging (and, in the restart loop, its own overhead)
    SAT_DIAG_PERTURB_INIT;
    for (int restart = tid; restart < a.maxstart; restart += T) {
        any = true;
        SAT_PHASE(7);
        const uint64_t subseq = subseq_lo | ((uint64_t)(uint32_t)restart << 32);

        // ---- random initial map (thinit, K.cu:588-648): order preserving, types respected
        Bits<M1W> mapped = bits_zero<M1W>();
        Bits<M2W> occ = bits_zero<M2W>();
        {
            {
                // (the word addresses are formed again every restart: kept, they are loop invariants that sit in
                // registers through the step loop - one of them ended up in scratch)
                int wa = tid;
                asm volatile("" : "+v"(wa));
                for (int w = 0; w < cmp_words; w++, wa += TP) smap[wa] = nullword;
            }
            int j = 0;
            bool stopped = false;
            for (int i0 = 0; i0 < n1; i0 += 4) {
                // a chain whose type search failed draws no more (K.cu:633-638): once that holds for every lane of
                // the wave the rest of the query is skipped - for long queries against short entries (the scan
                // position runs off the entry after ~2 n2 query SSEs) that is most of the loop and of its Philox blocks
                if constexpr (N1P > 32)
                    if (__builtin_amdgcn_ballot_w64(!stopped) == 0ull) break;
                uint4 r = philox_block(Q.seed_q, subseq, (uint32_t)(i0 >> 2));
                uint32_t rv[4] = { r.x, r.y, r.z, r.w };
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const int i = i0 + s;
                    if (i < n1) {
                        // u < 0.5 for u = 2^-32 + float(v) * 2^-32 (K.cu:625): float(v) < 2^31, i.e. v below the
                        // first value that rounds up to 2^31 (24-bit mantissa, ties to even)
                        if (!stopped && rv[s] < 0x7FFFFFC0u) {
                            Bits<M2W> cand, below = bits_below<M2W>(j);
                            if constexpr (M2W == 1) {
                                cand.w[0] = qmask[i] & ~below.w[0];
                            } else {
                                const int t = qtypes[i];
#pragma unroll
                                for (int w = 0; w < M2W; w++) cand.w[w] = tmask[t * TMS + w] & ~below.w[w];
                            }
                            int jj = bits_lowest<M2W>(cand);
                            if (jj < 0) {
                                stopped = true;              // K.cu:633-638: give up, no more draws used
                            } else {
                                smap_b[map_byte_addr(i)] = (uint8_t)jj;
                                bits_set<M1W>(mapped, i);
                                bits_set<M2W>(occ, jj);
                                j = jj + 1;
                            }
                        }
                    }
                }
            }
        }

        SAT_PHASE(10);                    // thinit
        // ---- full score of the initial map (tmscord, K.cu:396-440): pairs i < k, both matched
        int score = 0;
        constexpr bool FS_PAIRS = N1P > 16 && !SAT_DIAG_FS_ROWS_ONLY;
        // The pair walk below costs the wave what its busiest lane costs, m (m - 1) / 2 single pairs for m matched
        // SSEs (~27 instructions each), the rows-in-step form n1w * n1 / 2 packed evaluations (~31 each) whatever
        // the maps hold: a wave whose densest initial map would make the walk the dearer of the two takes the rows
        // (all-hit databases, where thinit matches 16+ of 32 SSEs: 5.7 -> 6.6 M scorings/s on scripts/exp/
        // dense_hits.py, its all-miss leg 7.6 -> 8.3 M).  With sets of several words a pop costs more, but pricing
        // the pair at 60 there sent the 101-SSE-query launches to the rows too early (-4 %): one price for all.
        bool walk_pairs = FS_PAIRS;
        if constexpr (FS_PAIRS) {
            constexpr int PAIR_COST = 27;
            const int m = bits_count<M1W>(mapped);
            walk_pairs = __builtin_amdgcn_ballot_w64(__mul24(__mul24(m, m - 1), PAIR_COST) > __mul24(__mul24(n1w, n1), 31)) == 0ull;
        }
        if (walk_pairs) {
            // Every lane walks the matched pairs of ITS chain (set bits of `mapped`: i ascending, k above i)
            // and the wave loops until its last lane is done.  An initial map matches ~8 query SSEs whatever
            // the query's size, so this is ~30-90 single pair evaluations per restart where walking the rows
            // in step for all lanes costs n1w * n1 / 2 packed ones: 136 for a 32-SSE query, 1313 for 101 SSEs
            // (half the run time of the 101-SSE query class before this loop).  Measured against the rows-in-step
            // form below: 101-SSE query x entries of 8..96 SSEs 1.66 -> 2.1 M scorings/s, BASELINE configs[4]
            // 1.60 -> 2.0 M, configs[2] 310 -> 443 k, 32-SSE query x entries of 8..32 SSEs +7 %, x 32-SSE
            // entries +-0; queries of up to 16 SSEs keep the rows in step (at most 32 packed evaluations: 1.5 %
            // faster there).
            typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
            typedef const __attribute__((address_space(1))) u32x2_t *gptr_u2;
            const gptr_c qpairG = (gptr_c)(uintptr_t)Q.qpair;
            // thinit's maps are order preserving whatever LORDER says (K.cu:588-648), so the c-th matched query
            // SSE has the c-th occupied db SSE as its image: with a one-word db set the two bit sets are popped
            // in step and the map bytes are never read (with more words the byte read is cheaper than the pop)
            constexpr bool POP_IMAGES = M2W == 1;
            Bits<M1W> ri = mapped;
            Bits<M2W> rj = occ;
            // pops the lowest set bit: its position (0 when the set is empty) and whether there was one
            auto pop = [](auto &b, bool &valid) -> int {
                constexpr int W = sizeof(b.w) / sizeof(b.w[0]);
                valid = bits_any<W>(b);
                const int pos = max(bits_lowest<W>(b), 0);
                bits_drop_lowest<W>(b);
                return pos;
            };
            // One-word sets on both sides, one lane per chain, full wave: the two lanes of a PAIR walk their two chains
            // together.  A walk costs the wave what its busiest lane costs, and the busiest chain of 64 has ~13
            // matched SSEs (78 pairs) where the mean has 7.7 (26): the rows of the pair's two chains (row a of a chain
            // with m matched SSEs = its a-th matched SSE against the m - 1 - a above it) are merged in order of
            // decreasing length and dealt to the two lanes alternately, so both lanes of the pair meet rows of nearly
            // equal length in the same trip and each does half of the pair's work.  A lane needs its partner's two
            // sets (two shuffles) and hands the partner's share of the sums back at the end (one more).  Bench shape
            // 10.85 -> 10.71 ms (+1.3 %): the walk's ~2300 VALU instructions per restart become ~1800 - the kernel is
            // issue bound, so that, not the shorter dependent chain of loads, is what the gain is.
            // (not with LSOLN: four more live registers there end up in scratch)
            constexpr bool FS_TEAMS = SAT_FS_TEAMS && M1W == 1 && M2W == 1 && FAST && (OPT == 0 || OPT == 1);
            bool teamed = false;
            if constexpr (FS_TEAMS) teamed = __builtin_amdgcn_ballot_w64(true) == ~0ull;
            if (FS_TEAMS && teamed) {
                const int pl = wlane & 1;
                const uint32_t om = (uint32_t)__shfl_xor((int)mapped.w[0], 1, 64), oo = (uint32_t)__shfl_xor((int)occ.w[0], 1, 64);
                const uint32_t m0 = pl ? om : mapped.w[0], o0 = pl ? oo : occ.w[0];       // chain 0: the even lane's
                const uint32_t m1 = pl ? mapped.w[0] : om, o1 = pl ? occ.w[0] : oo;
                const int c0n = __popc(m0), c1n = __popc(m1);
                const bool big1 = c1n > c0n;                                             // the longer chain leads the merged order
                const uint32_t mb = big1 ? m1 : m0, ob = big1 ? o1 : o0, msm = big1 ? m0 : m1, osm = big1 ? o0 : o1;
                const int rb = max((big1 ? c1n : c0n) - 1, 0), rs = max((big1 ? c0n : c1n) - 1, 0);   // rows with partners
                const int dlead = rb - rs, etot = rb + rs;
                uint32_t rm = mb, ro = ob;                 // the stream this lane is on: its sets with the rows below `cur` stripped
                int cur = 0, acc_b = 0, acc_s = 0;
                bool on_small = false;
                // element e of the merged order: the first dlead are rows 0 .. of the longer chain (lengths rb .. rs + 1),
                // then lengths rs .. 1 twice each, longer chain first.  This lane takes e = pl, pl + 2, ...: the leading
                // rows of the longer chain two apart, then ONE of the two chains row after row (e - dlead keeps its parity).
                for (int e = pl; __builtin_amdgcn_ballot_w64(e < etot) != 0ull; e += 2) {
                    const bool act = e < etot;
                    const int e2 = e - dlead;
                    const bool small = e2 >= 0 && (e2 & 1) != 0;
                    const int len = e2 < 0 ? rb - e : rs - (e2 >> 1);
                    const int a = (small ? rs : rb) - len;
                    if (small && !on_small) { rm = msm; ro = osm; cur = 0; on_small = true; }
#pragma unroll
                    for (int q = 0; q < 2; q++) {                           // at most two rows further on
                        const uint32_t go = (act && cur < a) ? 1u : 0u;
                        rm &= rm - go;
                        ro &= ro - go;
                        cur += (int)go;
                    }
                    const int i = act ? __ffs(rm) - 1 : 0, ji = act ? __ffs(ro) - 1 : 0;
                    Bits<1> rk, rl;
                    rk.w[0] = act ? rm & (rm - 1u) : 0u;
                    rl.w[0] = act ? ro & (ro - 1u) : 0u;
                    const DbRow<CELLS> drow = db_row(ji);
                    const uint32_t qrow = (uint32_t)__mul24(i, N1P * 8);
                    int rowsum = 0;
                    while (__builtin_amdgcn_ballot_w64(rk.w[0] != 0u) != 0ull) {
                        int ll[SAT_FS_UNROLL];
                        bool vv[SAT_FS_UNROLL];
                        u32x2_t qcell[SAT_FS_UNROLL];
#pragma unroll
                        for (int u = 0; u < SAT_FS_UNROLL; u++) {
                            bool vl;
                            const int k = pop(rk, vv[u]);
                            ll[u] = pop(rl, vl);
                            qcell[u] = *(gptr_u2)(qpairG + (qrow + ((uint32_t)k << 3)));
                        }
#pragma unroll
                        for (int u = 0; u < SAT_FS_UNROLL; u++) {
                            const uint2 c = db_cell<CELLS>(drow, ll[u]);
                            const int term = pair_term(qcell[u].x, qcell[u].y, c.x, c.y);
                            rowsum += vv[u] ? term : 0;
                        }
                    }
                    acc_s += small ? rowsum : 0;
                    acc_b += small ? 0 : rowsum;
                }
                const int acc0 = big1 ? acc_s : acc_b, acc1 = big1 ? acc_b : acc_s;          // by chain
                score = (pl ? acc1 : acc0) + __shfl_xor(pl ? acc0 : acc1, 1, 64);
            } else
            while (__builtin_amdgcn_ballot_w64(bits_any<M1W>(ri)) != 0ull) {
                bool ai, aj;
                // The lanes that share a chain take its ROWS in turn: every trip pops lpc matched SSEs, lane `part`
                // keeps the part-th as its row - with the matched SSEs above THAT one as the row's partners - and each
                // lane then walks its own row's partners alone.  (The first version shared every row: all lanes popped
                // the same partner sequence and kept every lpc-th, i.e. every lane paid every pop - and the pops of a
                // four-word set are most of a pair's instructions.)
                int i = 0, ji = 0;                                         // (a lane that is done walks row 0, sums nothing)
                ai = false;
                Bits<M1W> rk = bits_zero<M1W>();                           // the matched SSEs above i ...
                Bits<M2W> rl = bits_zero<M2W>();                           // ... and their images
                for (int p = 0; p < lpc; p++) {
                    bool v;
                    const int pos = pop(ri, v);
                    int img = 0;
                    if constexpr (POP_IMAGES) { bool vj; img = pop(rj, vj); }
                    if (p == part) {
                        i = pos;
                        ai = v;
                        ji = img;
                        rk = ri;
                        rl = rj;
                    }
                }
                (void)aj;
                if constexpr (!POP_IMAGES) { ji = smap_b[map_byte_addr(i)]; ji = ai ? ji : 0; }
                const DbRow<CELLS> drow = db_row(ji);
                const uint32_t qrow = (uint32_t)__mul24(i, N1P * 8);
                int rowsum = 0;
                // SAT_FS_UNROLL pairs per lane and round, their loads in flight together
                while (__builtin_amdgcn_ballot_w64(bits_any<M1W>(rk)) != 0ull) {
                    int ll[SAT_FS_UNROLL];
                    bool vv[SAT_FS_UNROLL];
                    u32x2_t qcell[SAT_FS_UNROLL];
#pragma unroll
                    for (int u = 0; u < SAT_FS_UNROLL; u++) {
                        const int ku = pop(rk, vv[u]);
                        ll[u] = 0;
                        if constexpr (POP_IMAGES) { bool vl; ll[u] = pop(rl, vl); }
                        // (none left: SSE 0's image, possibly the null column - it exists, and the term is dropped)
                        if constexpr (!POP_IMAGES) ll[u] = smap_b[map_byte_addr(ku)];
                        qcell[u] = *(gptr_u2)(qpairG + (qrow + ((uint32_t)ku << 3)));
                    }
#pragma unroll
                    for (int u = 0; u < SAT_FS_UNROLL; u++) {
                        const uint2 c = db_cell<CELLS>(drow, ll[u]);
                        const int term = pair_term(qcell[u].x, qcell[u].y, c.x, c.y);
                        rowsum += vv[u] ? term : 0;
                    }
                }
                score += rowsum;
            }
        } else {
            score = score_rows();
        }
        if (lpc >= 2) score += __shfl_xor(score, 1, 64);
        if (lpc == 4) score += __shfl_xor(score, 2, 64);
        const int best_before = best;
        if (score > best) {
            best = score;
            if (lsoln && beats_leader(score, restart))
                for (int w = 0; w < n1w; w++) bmap[w * T + tid] = smap[w * TP + tid];
        }

        // ---- 100 Metropolis steps, temperature 10 * 0.95^iter (K.cu:1030-1191)
        SAT_PHASE(6);                     // full score of the initial map
        uint4 blk = uint4{ 0u, 0u, 0u, 0u };
        for (int iter = 0; iter < SAT_K_MAXITER; iter++) {
            // this step's row of the Metropolis table.  The directory is read-only for the kernel's
            // lifetime: through the constant address space these are scalar loads (a plain global
            // pointer gets a vector load, whose latency would sit in front of the table load), asked
            // for here so that they are back long before the test at the end of the step
            const int rowoff = prowC[2 * iter], rowmax = prowC[2 * iter + 1];
            // The first 64 entries of this step's row, one per lane: a coalesced load asked for here, a whole step
            // before the test needs it; the test then fetches its entry from the lane that holds it (ds_bpermute)
            // instead of waiting for a dependent global load at the very end of the step's chain.  Bench shape +3 %
            // in same-box A/B runs, the other LORDER launches of the 32-SSE classes and up with one-word db sets
            // 0 .. +0.7 %; not where it measured slower: the 16 class (-5 %), the static loops of LORDER = F
            // (-1.6 %), 64-SSE entries (-0.8 %).
            constexpr bool ROW_IN_LANES = N1P >= 32 && M2W == 1 && (OPT < 0 || (OPT & 1) != 0);
            float rowv = 0.0f;
            if constexpr (ROW_IN_LANES)
                rowv = *(gptr_f32)((gptr_c)ptabG + (((uint32_t)rowoff + (uint32_t)min(wlane, rowmax + 2)) << 2));
            // one Philox block per two steps: the even step draws it and uses words 0, 1, the odd step
            // uses words 2, 3 (moved down).  Word a = two 16-bit draws (moved SSE: high half, candidate:
            // low half), word b = the Metropolis draw.
            if ((iter & 1) == 0) {
                blk = philox_block(Q.seed_q, subseq, (uint32_t)(SAT_K_STEP_BLOCK0 + (iter >> 1)));
            } else {
                blk.x = blk.z;
                blk.y = blk.w;
            }
            const uint32_t word_a = blk.x, word_b = blk.y;
            SAT_DIAG_PERTURB_STEP;

            // which query SSE moves (K.cu:1037-1042)
            const int ssei = scaled_index16(word_a >> 16, n1, n1 - 1);

            // candidate db SSEs: free, same type, inside the order window (K.cu:1053-1086)
            int oldj;
            Bits<M2W> cand;
            if (M2W == 1 && opt_lorder) {
                // LORDER maps are order preserving (thinit builds them so and every move stays
                // inside its window), so the images of the mapped query SSEs are the set bits of
                // `occ` in the same order.  With p = highest mapped query SSE <= ssei and A its
                // image, the window [startj, endj) of K.cu:1053-1077 is the run of free bits
                // between A and the next occupied bit above it: no second and third map read, no
                // range masks.  p == ssei exactly when ssei is mapped, so A is also its old image.
                int p;
                bool none;
                highest_mapped_upto(mapped, ssei, p, none);
                const int A = smap_b[map_byte_addr(p)];
                SAT_DIAG_DUP_MAPBYTE(&smap_b[map_byte_addr(p)]);
                oldj = p == ssei ? A : NULLJ;
                const uint32_t above = 0xFFFFFFFEu << (A & 31);          // bits A+1 .. 31
                const uint32_t y = occ.w[0] & above;                     // occupied above A
                const uint32_t gap = (y - 1u) & ~y & above;              // free run up to the next occupied bit
                // no mapped SSE at or below ssei: startj = n2, empty (K.cu:1060-1063); no mapped
                // successor: endj = -1, empty, unless ssei is the last query SSE (K.cu:1064-1077)
                const bool empty = none || (y == 0u && ssei != n1 - 1);
                cand.w[0] = empty ? 0u : (qmask[ssei] & gap);
            } else if (M2W == 2 && FAST && opt_lorder) {
                // The same for entries of 33..64 SSEs with the two words of the db-side sets taken as ONE 64-bit
                // word: p, A and the old image as above, the window is the run of free bits between A and the next
                // occupied bit - (y - 1) & ~y on 64 bits - where the general path below reads three map bytes (two of
                // them behind the first) and builds four range masks word by word.
                int p;
                bool none;
                highest_mapped_upto(mapped, ssei, p, none);
                const int t = qtypes[ssei];
                const int A = smap_b[map_byte_addr(p)];
                oldj = p == ssei ? A : NULLJ;
                const unsigned long long occ64 = (unsigned long long)occ.w[0] | ((unsigned long long)occ.w[1] << 32);
                const unsigned long long above = (~1ull) << (A & 63);             // bits A+1 .. 63 (A = 64: no mapped SSE, `empty`)
                const unsigned long long y = occ64 & above;                       // occupied above A
                const unsigned long long gap = (y - 1ull) & ~y & above;           // free run up to the next occupied bit
                const unsigned long long tm = *reinterpret_cast<const unsigned long long *>(&tmask[t * TMS]);
                const bool empty = none || (y == 0ull && ssei != n1 - 1);
                const unsigned long long c64 = empty ? 0ull : (tm & gap);
                cand.w[0] = (uint32_t)c64;
                cand.w[M2W - 1] = (uint32_t)(c64 >> 32);
            } else if (M2W == 4 && FAST && opt_lorder) {
                // ... and for entries above 64 SSEs as two 64-bit halves: y - 1 borrows from the upper half exactly
                // when no bit of the lower half is occupied above A.
                int p;
                bool none;
                highest_mapped_upto(mapped, ssei, p, none);
                const int t = qtypes[ssei];
                const int A = smap_b[map_byte_addr(p)];
                oldj = p == ssei ? A : NULLJ;
                const unsigned long long occ_lo = (unsigned long long)occ.w[0] | ((unsigned long long)occ.w[1] << 32),
                                         occ_hi = (unsigned long long)occ.w[2] | ((unsigned long long)occ.w[3] << 32);
                const unsigned long long base = (~1ull) << (A & 63);               // (A & 63 = 63: nothing above it in its half)
                const bool a_hi = A >= 64;
                const unsigned long long above_lo = a_hi ? 0ull : base, above_hi = a_hi ? base : ~0ull;
                const unsigned long long y_lo = occ_lo & above_lo, y_hi = occ_hi & above_hi;      // occupied above A
                const unsigned long long gap_lo = (y_lo - 1ull) & ~y_lo & above_lo;
                const unsigned long long gap_hi = (y_hi - (y_lo == 0ull ? 1ull : 0ull)) & ~y_hi & above_hi;
                const unsigned long long *tm = reinterpret_cast<const unsigned long long *>(&tmask[t * TMS]);
                const bool empty = none || ((y_lo | y_hi) == 0ull && ssei != n1 - 1);
                const unsigned long long c_lo = empty ? 0ull : (tm[0] & gap_lo), c_hi = empty ? 0ull : (tm[1] & gap_hi);
                cand.w[0] = (uint32_t)c_lo;
                cand.w[1 % M2W] = (uint32_t)(c_lo >> 32);
                cand.w[2 % M2W] = (uint32_t)c_hi;
                cand.w[3 % M2W] = (uint32_t)(c_hi >> 32);
            } else {
                oldj = smap_b[map_byte_addr(ssei)];
                int startj = 0, endj = n2;
                if (opt_lorder) {
                    Bits<M1W> upto = bits_below<M1W>(ssei + 1), lowpart, highpart;
#pragma unroll
                    for (int w = 0; w < M1W; w++) {
                        lowpart.w[w] = mapped.w[w] & upto.w[w];
                        highpart.w[w] = mapped.w[w] & ~upto.w[w];
                    }
                    const int p = bits_highest<M1W>(lowpart);
                    const int q = bits_lowest<M1W>(highpart);
                    const int pimg = smap_b[map_byte_addr(p < 0 ? 0 : p)];
                    const int qimg = smap_b[map_byte_addr(q < 0 ? 0 : q)];
                    startj = p < 0 ? n2 : pimg;                      // no mapped predecessor: empty window
                    endj = (ssei == n1 - 1) ? n2 : (q < 0 ? -1 : qimg);   // K.cu:1064-1077
                }
                const int t = qtypes[ssei];
                Bits<M2W> lo = bits_below<M2W>(startj), hi = bits_below<M2W>(endj);
#pragma unroll
                for (int w = 0; w < M2W; w++)
                    cand.w[w] = tmask[t * TMS + w] & ~occ.w[w] & hi.w[w] & ~lo.w[w];
            }
            // no candidate: the SSE becomes unmatched; one: it is taken without a draw
            // (K.cu:701-702); several: the draw picks the (u - EPS) * cnt -th (K.cu:705-711).
            // Branch-free: in a 64-lane wave every case occurs anyway.
            const int cnt = bits_count<M2W>(cand);
            const int pick = scaled_index16(word_a & 0xFFFFu, cnt, max(cnt - 1, 0));   // 0 for cnt <= 1: no draw used
            int sel;
            if (M2W == 1 && opt_lorder) {
                // inside an order window the picked rank is small (few free same-type SSEs): strip
                // the lowest set bit `pick` times, looping while any lane of the wave still has to
                uint32_t c = cand.w[0];
                int left = pick;
                {
                    // the first strip without the wave-level test (a ballot, a scalar branch and its wait per trip of
                    // the loop below; most waves need one or two strips): 32-SSE bench shape +1.5 %
                    const uint32_t go = left > 0 ? 1u : 0u;
                    c &= c - go;
                    left -= (int)go;
                }
                while (__builtin_amdgcn_ballot_w64(left > 0) != 0ull) {
                    const uint32_t go = left > 0 ? 1u : 0u;
                    c &= c - go;                                   // c & (c - 1) clears the lowest set bit
                    left -= (int)go;
                }
                sel = __ffs(c) - 1;
            } else if (M2W == 2 && FAST && opt_lorder && __builtin_amdgcn_ballot_w64(pick > 2) == 0ull) {
                // (wide windows - a short query against a long entry - hold many candidates: the strip loop runs as
                // often as the largest pick of the wave, so it is taken only while every pick is small; else the rank select)
                unsigned long long c = (unsigned long long)cand.w[0] | ((unsigned long long)cand.w[M2W - 1] << 32);
                int left = pick;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const unsigned long long go = left > 0 ? 1ull : 0ull;
                    c &= c - go;
                    left -= (int)go;
                }
                sel = __ffsll((long long)c) - 1;
            } else if (M2W == 4 && FAST && opt_lorder && __builtin_amdgcn_ballot_w64(pick > 2) == 0ull) {
                unsigned long long c_lo = (unsigned long long)cand.w[0] | ((unsigned long long)cand.w[1 % M2W] << 32),
                                   c_hi = (unsigned long long)cand.w[2 % M2W] | ((unsigned long long)cand.w[3 % M2W] << 32);
                int left = pick;
                // strips the lowest candidate: of the lower half while it has one, else of the upper half
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const bool go = left > 0, in_lo = c_lo != 0ull;
                    c_lo &= c_lo - ((go && in_lo) ? 1ull : 0ull);
                    c_hi &= c_hi - ((go && !in_lo) ? 1ull : 0ull);
                    left -= go ? 1 : 0;
                }
                sel = c_lo != 0ull ? __ffsll((long long)c_lo) - 1 : 63 + __ffsll((long long)c_hi);
            } else {
                sel = bits_select<M2W>(cand, pick);
            }
            const bool nreal = cnt != 0;
            const int newj = nreal ? sel : NULLJ;

            SAT_PHASE(0);                 // draw + proposal
            // score change (deltasd, K.cu:502-535)
            int delta;
            {
                // rows of this step that are real, listed once per chain (part 0 of its lanes)
                const bool oreal = oldj != NULLJ;
                const bool lists = part == 0;
                const int nitems = lists ? (int)oreal + (int)nreal : 0;
                // (two plain ballots and scalar logic: a ballot of a combined predicate goes through
                // a select and a compare per lane)
                const unsigned long long bo = __builtin_amdgcn_ballot_w64(lists && oreal),
                                         bn = __builtin_amdgcn_ballot_w64(lists && nreal);
                const unsigned long long m1 = bo | bn, m2 = bo & bn;
                const int total_items = __popcll(m1) + __popcll(m2);           // wave-uniform
                // only full waves compact (a wave's last lanes may have no restart left), so a lane's
                // rank among the consumers is its lane number; see cmp_* above the restart loop
                if (opt_compact && __builtin_amdgcn_ballot_w64(true) == ~0ull && total_items <= 64) {
                    const int pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0)) +
                                    __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0));
                    // item = row | moved SSE << 8 | owner chain << 16 | negate << 24.  The slot doubles as
                    // the row's accumulator: the lanes that serve an item all read it in one instruction,
                    // then add their signed sums to it; the owner subtracts what it wrote.
                    const uint32_t item1 = (uint32_t)(oreal ? oldj : newj) | ((uint32_t)ssei << 8) | ((uint32_t)tid << 16) |
                                           (oreal ? 1u << 24 : 0u);
                    const uint32_t item2 = (uint32_t)newj | ((uint32_t)ssei << 8) | ((uint32_t)tid << 16);
                    if (nitems >= 1) items[pre] = item1;
                    if (nitems == 2) items[pre + 1] = item2;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    SAT_PHASE(1);         // compaction set-up
                    // one round: the rows first .. first + 64 / lpi - 1 of the table, `lpi` lanes per row
                    // (lane `rsub` of the round serves row first + rsub, words rkw, rkw + lpi, ...)
                    auto one_round = [&](auto wtag, int first, int lpi, int rsub, int rkw, bool rlane_ok) {
                        constexpr int W = decltype(wtag)::value;
                        const int idx = first + rsub;
                        bool ok = rlane_ok && idx < total_items;
                        int v = 0;
                        if (ok) {
                            const uint32_t it = items[idx];
                            const int row = it & 0xFF, si = (it >> 8) & 0xFF, owner = (it >> 16) & 0xFF;
                            const DbRow<CELLS> drow = db_row(row);
                            float4 qd[W];
                            uint32_t qc[W], wd[W];
                            // byte offsets of (word rkw, column si) in the two query arrays; word rkw + u * lpi is
                            // u * lpi * N1P groups further on (an instruction offset where lpi is a constant)
                            const uint32_t qoff4 = (uint32_t)(rkw * N1P + si) << 2, qoff16 = qoff4 << 2;
#pragma unroll
                            for (int u = 0; u < W; u++) {
                                // words past the map (a lane's last one, when lpi does not divide n1w)
                                // are padding: unmatched SSEs against the query's sentinel cells
                                const int kwu = rkw + u * lpi;
                                wd[u] = smap[kwu * TP + owner];
                                SAT_DIAG_DUP_MAPWORD(&smap[kwu * TP + owner]);
                                qd[u] = load_qdist(qoff16 + (uint32_t)(u * lpi * N1P * 16));
                                qc[u] = load_qcode(qoff4 + (uint32_t)(u * lpi * N1P * 4));
                            }
#pragma unroll
                            for (int u = 0; u < W; u++) v = quad_terms(qd[u], qc[u], drow, wd[u], 0u, v);
                            v = (it >> 24) ? -v : v;
                        }
                        // signed sum of a lane's words -> the row's accumulator (its item slot)
                        if ((lpi & 3) == 0) {
                            // rows are aligned groups of 4m lanes: add up each quad of lanes with two
                            // DPP moves, so that a quarter of the lanes hit the accumulator
                            v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                            v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
                            ok = ok && (rkw & 3) == 0;
                        }
                        if (ok)
                            __hip_atomic_fetch_add((lds_i32_t *)(items + idx), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        if (ok) SAT_DIAG_DUP_ATOMIC((lds_i32_t *)(items + idx));
                    };
                    auto main_round = [&](int first) {
                        if constexpr (WPL > 0) one_round(std::integral_constant<int, WPL>{}, first, cmp_lpi, sub, kw, lane_ok);
                        else switch (cmp_wpl) {
                        case 1: one_round(std::integral_constant<int, 1>{}, first, cmp_lpi, sub, kw, lane_ok); break;
                        case 2: one_round(std::integral_constant<int, 2>{}, first, cmp_lpi, sub, kw, lane_ok); break;
                        case 3: one_round(std::integral_constant<int, 3>{}, first, cmp_lpi, sub, kw, lane_ok); break;
                        default: one_round(std::integral_constant<int, 4>{}, first, cmp_lpi, sub, kw, lane_ok); break;
                        }
                    };
                    // full rounds of the main shape while more rows remain than one round holds; the
                    // last rows go to the shape with the fewest words per lane that still takes them in
                    // one round (a step lists ~0.6 rows per chain: the tail is usually a few rows)
                    int first = 0;
                    for (; total_items - first > per_round; first += per_round) main_round(first);
                    const int rest = total_items - first;
                    if (rest > 0) {
                        if (cmp_wpl > 1 && rest <= tail1_rows) {
                            // (lane -> (row, word) of a tail shape is worked out here every time: hoisted out
                            // of the step loop these values would sit in registers the main shape needs)
                            int l = wlane;
                            asm volatile("" : "+v"(l));
                            const int rsub = __mul24(l, tail1_recip) >> 16;
                            one_round(std::integral_constant<int, 1>{}, first, n1w, rsub, l - __mul24(rsub, n1w), rsub < tail1_rows);
                        } else if (cmp_wpl > 2 && rest <= tail2_rows) {
                            int l = wlane;
                            asm volatile("" : "+v"(l));
                            const int rsub = __mul24(l, tail2_recip) >> 16;
                            one_round(std::integral_constant<int, 2>{}, first, tail2_lpi, rsub, l - __mul24(rsub, tail2_lpi), rsub < tail2_rows);
                        } else {
                            main_round(first);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    SAT_PHASE(2);         // compacted rounds
                    delta = 0;
                    if (nitems >= 1) delta = (int)(items[pre] - item1);
                    if (nitems == 2) delta += (int)(items[pre + 1] - item2);
                    // with several lanes per chain only part 0 listed rows: hand its sum to the others
                    if (lpc > 1) delta = __shfl(delta, wlane & ~(lpc - 1), 64);
                } else {
                    // dense regime: every lane scores its own two rows
                    // (a null image has no row: row 0 stands in and the sum is dropped)
                    const DbRow<CELLS> orow = db_row(oreal ? oldj : 0), nrow = db_row(nreal ? newj : 0);
                    int sum_new = 0, sum_old = 0;
                    auto move_group = [&](int kw) {
                        const uint32_t word = smap[kw * TP + tid];
                        const uint32_t qi = (uint32_t)(kw * N1P + ssei);      // 32-bit offsets from uniform bases
                        const float4 qd = load_qdist(qi << 4);
                        const uint32_t qc = load_qcode(qi << 2);
                        sum_new = quad_terms(qd, qc, nrow, word, 0u, sum_new);
                        sum_old = quad_terms(qd, qc, orow, word, 0u, sum_old);
                    };
                    if (lpc == 1) for (int kw = 0; kw < n1w; kw++) move_group(kw);
                    else for (int kw = part; kw < n1w; kw += lpc) move_group(kw);
                    delta = (nreal ? sum_new : 0) - (oreal ? sum_old : 0);
                    if (lpc >= 2) delta += __shfl_xor(delta, 1, 64);
                    if (lpc == 4) delta += __shfl_xor(delta, 2, 64);
                }
            }
            const int newscore = score + delta;
            SAT_PHASE(3);                 // read-back (compacted) or the static loops
            SAT_DIAG_SELFCHECK_STEP;

            // best-so-far from the PROPOSED state, before the accept test (K.cu:1136-1155)
            // (which restart holds the best is settled once per restart, below the step loop)
            if (lsoln && newscore > best) {
                if (beats_leader(newscore, restart)) {
                    for (int w = 0; w < n1w; w++) bmap[w * T + tid] = smap[w * TP + tid];
                    bmap_b[bmap_byte_addr(ssei)] = (uint8_t)newj;
                }
            }
            best = max(best, newscore);

            SAT_PHASE(4);                 // best tracking
            // Metropolis: accept iff expf(delta / temp) > u, via the host-built table
            // the table holds 2^32 * expf(.), compared with 2^32 * u: same decision, one multiply less
            const float u = draw32(word_b);
            // row = { 2^33 (any delta > 0: expf(x > 0) > 1 >= u), P[0], ..., P[rowmax], 0.0 (a larger
            // -delta can never be accepted) }, indexed by 1 - delta clamped to the row
            const uint32_t nd = (uint32_t)min(max(1 - delta, 0), rowmax + 2);
            float p;
            // (full waves only: a lane without a restart has not loaded its entry of the row; -delta beyond 62
            // anywhere in the wave: the load after all)
            if (ROW_IN_LANES && __builtin_amdgcn_ballot_w64(true) == ~0ull && __builtin_amdgcn_ballot_w64(nd >= 64u) == 0ull)
                p = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(nd << 2), __float_as_int(rowv)));
            else
                p = *(gptr_f32)((gptr_c)ptabG + (((uint32_t)rowoff + nd) << 2));
            const bool accept = p > u;
            if (accept) smap_b[map_byte_addr(ssei)] = (uint8_t)newj;
            score = accept ? newscore : score;
            {
                // an accepted move toggles the old image's bit (set) and the new image's bit (clear) of `occ`, and
                // the moved SSE's bit of `mapped` when it changes between matched and unmatched; the accept
                // decision is folded into the bits, the word index picks the word
                const bool oreal_ = oldj != NULLJ;
                if constexpr (M2W == 1) {
                    // bit n2 (the null SSE) must not be touched; n2 may be 32: mask by comparison.
                    const uint32_t oldbit = (accept && oreal_) ? (1u << (oldj & 31)) : 0u;
                    const uint32_t newbit = (accept && nreal) ? (1u << (newj & 31)) : 0u;
                    occ.w[0] = (occ.w[0] & ~oldbit) | newbit;
                } else {
                    const uint32_t oldbit = (accept && oreal_) ? (1u << (oldj & 31)) : 0u;
                    const uint32_t newbit = (accept && nreal) ? (1u << (newj & 31)) : 0u;
                    const int ow = oldj >> 5, nw = newj >> 5;
#pragma unroll
                    for (int w = 0; w < M2W; w++) occ.w[w] ^= (ow == w ? oldbit : 0u) ^ (nw == w ? newbit : 0u);
                }
                if constexpr (M1W == 1) {
                    const uint32_t ibit = 1u << ssei;
                    const uint32_t setbit = (accept && nreal) ? ibit : 0u, clrbit = (accept && !nreal) ? ibit : 0u;
                    mapped.w[0] = (mapped.w[0] & ~clrbit) | setbit;
                } else {
                    const uint32_t ibit = (accept && oreal_ != nreal) ? (1u << (ssei & 31)) : 0u;
                    const int iw = ssei >> 5;
#pragma unroll
                    for (int w = 0; w < M1W; w++) mapped.w[w] ^= iw == w ? ibit : 0u;
                }
            }
            SAT_PHASE(5);                 // Metropolis + state update
        }
        if (best > best_before) best_restart = (uint32_t)restart;
    }
    SAT_PHASE_FLUSH;
    SAT_DIAG_PERTURB_END;

    // ---- arg-max over restarts; ties go to the lowest restart index, which is the
    // first restart that reaches the maximum in the reference's sequential order
    // (strict '>' at K.cu:1024, 1137, 1211)
    unsigned long long key = any
        ? (((unsigned long long)(uint32_t)(best + 0x40000000)) << 32) | (0xFFFFFFFFu - best_restart)
        : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long other = __shfl_xor(key, off, 64);
        key = other > key ? other : key;
    }
    const int wave = lane_id >> 6, nwaves = (nthreads + 63) >> 6;
    if (wlane == 0) *red_key(wave) = key;
    __syncthreads();
    unsigned long long win = *red_key(0);
    for (int w = 1; w < nwaves; w++) win = *red_key(w) > win ? *red_key(w) : win;

    const uint32_t win_restart = 0xFFFFFFFFu - (uint32_t)(win & 0xFFFFFFFFu);
    if (lane_id == 0) Q.scores[e] = (int)(uint32_t)(win >> 32) - 0x40000000;
    if (lsoln && any && part == 0 && best_restart == win_restart &&
        ((((unsigned long long)(uint32_t)(best + 0x40000000)) << 32) | (0xFFFFFFFFu - best_restart)) == win) {
        int8_t *out = Q.ssemaps + (size_t)e * n1;
        for (int i = 0; i < n1; i++) {
            int j = bmap_b[bmap_byte_addr(i)];
            out[i] = (int8_t)(j == NULLJ ? -1 : j);
        }
    }
}
