// sat_sa_list_kernel.hpp - the SA search kernel for ORDER-PRESERVING searches (LORDER = T) of
// queries of up to 32 SSEs against entries of up to 32 SSEs: chains keep their map as a LIST of
// matched pairs, and a step scores only the matched pairs.
//
// Why: a database scan is sparse.  On random (query, entry) pairs a chain has m ~ 7.7 of 32 query
// SSEs matched (measured over the bench workload's 2.56 M steps: P(m <= 8) = 0.62, P(m <= 16) =
// 0.999), so the dense kernel (sat_sa_kernel.hpp), which scores a listed row against all n1 map
// bytes four at a time, spends three quarters of its pair evaluations on unmatched SSEs: 3.96
// packed evaluations per chain-step where 1.22 would do.  The kernel is VALU-issue bound
// (DESIGN.md section 4), so the evaluations are what to cut.
//
// Chain state (LDS, word-interleaved [word][chain], so that every access a lane makes to its OWN
// chain is bank-conflict free whatever byte it touches):
//   slot lists  K[s], L[s], s < m: query SSE K[s] is matched to db SSE L[s]; unordered; slots
//               s >= m hold L = the null SSE (whose db cells carry the distance sentinel, so a
//               padding pair scores 0 without any test), K = a valid index;
//   pos[k]      the slot of query SSE k (valid while k is matched).
//   map[k] of the reference (K.cu:1053-1086) is L[pos[k]]; a move touches at most five bytes:
//   re-map  L[pos[i]] = newj;   map  K[m] = i, L[m] = newj, pos[i] = m, m++;
//   unmap   (K, L)[pos[i]] = (K, L)[m-1], pos[K[m-1]] = pos[i], L[m-1] = null, m--.
//   The matched-SSE bit set `mapped`, the occupied-db-SSE bit set `occ` and m stay in registers.
//
// Step scoring: as in the dense kernel, the lanes of a wave list the rows of this step that are
// real (old image / new image of the moved SSE; 0.50 per chain-step) and the whole wave serves
// them - but a row now costs ceil(m / 8) lane-tasks of 8 slots ("octs": two packed evaluations
// whose sixteen loads are in flight together).  Rows of chains with m <= 8 take one lane and are
// listed from the front of the wave's item table, rows of longer chains take two lanes (each
// every second oct) and are listed from the back; lanes [0, S) serve the short rows and lanes
// [64 - 2M, 64) the long ones, so a typical step (20 short + 12 long rows) is ONE round of 44
// busy lanes x two packed evaluations.  The query cells are gathered per pair from the query's
// cell matrix qcell[i][k] = {distance, code} through L1 (8 KB for a 32-SSE query).
//
// The restart's first full score runs over the list too: sum over slot pairs a < b, i.e. m(m-1)/2
// pair evaluations instead of n1(n1-1)/2.
//
// Everything else (Philox streams and slot layout, order window from the occupied-bit set, the
// candidate pick, best-so-far from the proposed state, host-tabulated Metropolis test, arg-max
// with ties to the lowest restart, LSOLN leader key) is what sat_sa_kernel.hpp does, and the CPU
// oracle is the same: results are bit-identical between the two kernels.
#pragma once

#include "sat_sa_kernel.hpp"

namespace satk {

struct ListLds {
    uint32_t cells, pos, kw, lw, tmask, qtypes, red, items, total;
};

// LDS carve of one workgroup of the list kernel: used by the kernel (with its own query's n1 and
// its own entry's n2) and by the host (with the launch's largest), so the two cannot disagree.
__host__ __device__ inline ListLds list_lds_layout(int n1, int n1p, int n2, int chains)
{
    ListLds L;
    const uint32_t n1w = (uint32_t)((n1 + 3) >> 2);
    uint32_t dcells = (uint32_t)(n2 + 1) * (uint32_t)(n2 + 1);
    dcells = (dcells + 1u) & ~1u;                           // 16-byte multiple
    uint32_t off = 0;
    L.cells = off;  off += dcells * 8u;
    L.pos = off;    off += n1w * (uint32_t)chains * 4u;
    L.kw = off;     off += n1w * (uint32_t)chains * 4u;
    L.lw = off;     off += n1w * (uint32_t)chains * 4u;
    L.tmask = off;  off += 16u * 4u;
    L.qtypes = off; off += ((uint32_t)n1p + 15u) & ~15u;
    off = (off + 7u) & ~7u;                                 // 64-bit reduction keys / LSOLN leader key
    L.red = off;    off += 17u * 8u;
    L.items = off;  off += (uint32_t)((chains + 63) / 64) * 64u * 4u;
    L.total = off;
    return L;
}

// Sum of the pair scores of the four slots of one list quad against db row `drow` for moved /
// anchor query SSE row `qrow`: slot s pairs query cell (qrow, K[s]) with db cell (drow, L[s]).
//   kword, lword  the quad's four K bytes and four L bytes
//   qrow          byte offset of the query row in the cell matrix (global memory, through L1)
//   drow          LDS byte address of the db row's cells
//   force         0x04 in byte s switches slot s off (full score: slots a and below)
struct OctLoads {
    uint2 q[8], d[8];
};

}  // namespace satk

// N1P: pitch of the query cell matrix (16 or 32); LSOLN: solution maps wanted.
template <int N1P, bool LSOLN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
sat_sa_list_kernel(const SatKernelArgs a)
{
    using namespace satk;
    static_assert(N1P <= 32, "one 32-bit word of matched-SSE bits");
    extern __shared__ __align__(16) unsigned char lds_raw[];
    typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
    typedef __attribute__((address_space(3))) int32_t lds_i32_t;
    typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;

    const int lane_id = threadIdx.x;
    const int T = blockDim.x;                     // chains per workgroup = lanes (one lane per chain)
    const int tid = lane_id;
    const int e = a.entry_list[blockIdx.x];
    const SatQuery Q = a.queries[blockIdx.y];
    const int n1 = Q.n1;
    const double n1d = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint((double)n1)),
                                        __builtin_amdgcn_readfirstlane(__double2loint((double)n1)));
    const int n2 = a.orders[e];
    const int n2p = n2 + 1;
    const int n1w = (n1 + 3) >> 2;
    const int NULLJ = n2;

    const ListLds lay = list_lds_layout(n1, N1P, n2, T);
    uint2 *Dc = reinterpret_cast<uint2 *>(lds_raw + lay.cells);
    uint32_t *tmask = reinterpret_cast<uint32_t *>(lds_raw + lay.tmask);
    uint8_t *qtypes = lds_raw + lay.qtypes;
    unsigned long long *red = reinterpret_cast<unsigned long long *>(lds_raw + lay.red);
    lds_u64_t *leader = (lds_u64_t *)(uintptr_t)(lay.red + 16u * 8u);
    lds_u32_t *items = (lds_u32_t *)(uintptr_t)(lay.items + (uint32_t)(lane_id >> 6) * 256u);
    // word w of chain c of one of the three per-chain arrays lives at base + (w * T + c) * 4; byte x
    // of this lane's own chain at base + ((x >> 2) * T + tid) * 4 + (x & 3)
    const uint32_t T4 = (uint32_t)T << 2, tid4 = (uint32_t)tid << 2;
    auto own_byte = [&](uint32_t base, int x) -> uint8_t * {
        return lds_raw + base + (uint32_t)__mul24(x >> 2, (int)T4) + tid4 + (uint32_t)(x & 3);
    };
    auto chain_word = [&](uint32_t base, int w, int chain) -> uint32_t * {
        return reinterpret_cast<uint32_t *>(lds_raw + base + (((uint32_t)__mul24(w, T) + (uint32_t)chain) << 2));
    };

    auto beats_leader = [&](int sc, int restart_) -> bool {
        const unsigned long long key = (((unsigned long long)(uint32_t)(sc + 0x40000000)) << 32) | (0xFFFFFFFFu - (uint32_t)restart_);
        if (key <= *leader) return false;
        __hip_atomic_fetch_max(leader, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return true;
    };

    // query cells through the global address space with a scalar base and a 32-bit byte offset
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) u32x2_t *gptr_u2;
    typedef const __attribute__((address_space(1))) char *gptr_c;
    typedef const __attribute__((address_space(4))) int32_t *cptr_i32;
    typedef const __attribute__((address_space(1))) float *gptr_f32;
    const gptr_c qcellG = (gptr_c)(uintptr_t)Q.qcell;
    const cptr_i32 prowC = (cptr_i32)(uintptr_t)a.prow;
    const gptr_f32 ptabG = (gptr_f32)(uintptr_t)a.ptab;
    auto load_qcell = [&](uint32_t byte_off) -> uint2 {
        const u32x2_t v = *(gptr_u2)(qcellG + byte_off);
        return uint2{ v.x, v.y };
    };

    // ---- stage the db entry: packed lower triangle (HBM) -> full cell matrix (LDS)
    {
        const uint8_t *tt = a.tab_tri + a.cell_off[e];
        const float *dd = a.dist_tri + a.cell_off[e];
        const int total = n2p * n2p;
        for (int c = lane_id; c < total; c += T) {
            const int j = c / n2p, l = c - j * n2p;
            uint2 cell;
            if (j < n2 && l < n2) {
                const int hi = j > l ? j : l, lo = j > l ? l : j;
                const int t = hi * (hi + 1) / 2 + lo;
                const float v = dd[t];
                cell.x = __float_as_uint(fabsf(v) <= 3.0e38f ? v : SAT_K_DSENT);
                cell.y = tt[t];
            } else {
                cell.x = __float_as_uint(SAT_K_DSENT);        // the null SSE never passes the distance test
                cell.y = 0u;
            }
            Dc[c] = cell;
        }
        if (lane_id < 16) tmask[lane_id] = 0u;
        if (lane_id == 0) red[16] = 0ull;                      // LSOLN leader key
        for (int i = lane_id; i < N1P; i += T) qtypes[i] = Q.qtypes[i];
    }
    __syncthreads();
    for (int j = lane_id; j < n2; j += T) {
        const int t = a.tab_tri[a.cell_off[e] + (int64_t)j * (j + 1) / 2 + j] & 3;     // diagonal = SSE type
        atomicOr(&tmask[t * 4], 1u << (j & 31));
    }
    __syncthreads();

    // LSOLN: best list of this chain in this workgroup's global slab, [word][chain]: K words, L words,
    // then one word {m, moved SSE, its proposed image} (the best state is a PROPOSED state)
    uint32_t *bslab = LSOLN ? a.bmap_slabs + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * a.bmap_slab_words : nullptr;

    const uint32_t nullword = (uint32_t)NULLJ * 0x01010101u;
    const uint32_t cell_row_bytes = (uint32_t)n2p << 3;
    const uint64_t subseq_lo = (uint64_t)a.ordinal[e];
    int best = SAT_K_NO_SCORE;
    uint32_t best_restart = 0xFFFFFFFFu;
    bool any = false;

    // packed evaluation of one quad of slots (see sat_sa_kernel.hpp, quad_terms): cells already loaded
    auto quad_eval = [&](const uint2 *q, const uint2 *d, const uint32_t force, const int acc) -> int {
        const float t0 = 4.0f - fabsf(__uint_as_float(q[0].x) - __uint_as_float(d[0].x));
        const float t1 = 4.0f - fabsf(__uint_as_float(q[1].x) - __uint_as_float(d[1].x));
        const float t2 = 4.0f - fabsf(__uint_as_float(q[2].x) - __uint_as_float(d[2].x));
        const float t3 = 4.0f - fabsf(__uint_as_float(q[3].x) - __uint_as_float(d[3].x));
        const uint32_t far = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x0C0C0703u) |
                             __builtin_amdgcn_perm(__float_as_uint(t3), __float_as_uint(t2), 0x07030C0Cu);
        const uint32_t x = __builtin_amdgcn_perm(q[1].y ^ d[1].y, q[0].y ^ d[0].y, 0x0C0C0400u) |
                           __builtin_amdgcn_perm(q[3].y ^ d[3].y, q[2].y ^ d[2].y, 0x04000C0Cu);
        const uint32_t z = (x + 0x77777777u) & 0x88888888u;
        uint32_t sel = ((z >> 3) | (z >> 6)) & 0x03030303u;
        sel |= ((far >> 5) & 0x04040404u) | force;
        const uint32_t terms = __builtin_amdgcn_perm(0u, 0xFE010102u, sel);      // {2, 1, 1, -2 | 0, 0, 0, 0}
        return __builtin_amdgcn_sdot4((int)terms, 0x01010101, acc, false);
    };
    // issue the eight loads of a quad: query cells (qrow, K[s]) and db cells (drow, L[s])
    auto quad_load = [&](const uint32_t kword, const uint32_t lword, const uint32_t qrow, const uint32_t drow, uint2 *q, uint2 *d) {
        const uint32_t k0 = kword & 0xFFu, k1 = (kword >> 8) & 0xFFu, k2 = (kword >> 16) & 0xFFu, k3 = kword >> 24;
        const uint32_t l0 = lword & 0xFFu, l1 = (lword >> 8) & 0xFFu, l2 = (lword >> 16) & 0xFFu, l3 = lword >> 24;
        q[0] = load_qcell(qrow + (k0 << 3));
        q[1] = load_qcell(qrow + (k1 << 3));
        q[2] = load_qcell(qrow + (k2 << 3));
        q[3] = load_qcell(qrow + (k3 << 3));
        d[0] = *reinterpret_cast<const uint2 *>(lds_raw + drow + (l0 << 3));
        d[1] = *reinterpret_cast<const uint2 *>(lds_raw + drow + (l1 << 3));
        d[2] = *reinterpret_cast<const uint2 *>(lds_raw + drow + (l2 << 3));
        d[3] = *reinterpret_cast<const uint2 *>(lds_raw + drow + (l3 << 3));
    };

    for (int restart = tid; restart < a.maxstart; restart += T) {
        any = true;
        const uint64_t subseq = subseq_lo | ((uint64_t)(uint32_t)restart << 32);

        // ---- random initial map (thinit, K.cu:588-648): order preserving, types respected
        uint32_t mapped = 0u, occ = 0u;
        int m = 0;
        {
            for (int w = 0; w < n1w; w++) {
                *chain_word(lay.lw, w, tid) = nullword;
                *chain_word(lay.kw, w, tid) = 0u;
            }
            int j = 0;
            bool stopped = false;
            for (int i0 = 0; i0 < n1; i0 += 4) {
                const uint4 r = philox_block(Q.seed_q, subseq, (uint32_t)(i0 >> 2));
                const uint32_t rv[4] = { r.x, r.y, r.z, r.w };
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const int i = i0 + s;
                    if (i < n1) {
                        const float u = to_uniform(rv[s]);
                        if (!stopped && u < 0.5f) {
                            const int t = qtypes[i];
                            const uint32_t cand = tmask[t * 4] & ~bits_below<1>(j).w[0];
                            if (cand == 0u) {
                                stopped = true;              // K.cu:633-638: give up, no more draws used
                            } else {
                                const int jj = __ffs(cand) - 1;
                                *own_byte(lay.kw, m) = (uint8_t)i;
                                *own_byte(lay.lw, m) = (uint8_t)jj;
                                *own_byte(lay.pos, i) = (uint8_t)m;
                                mapped |= 1u << i;
                                occ |= 1u << (jj & 31);
                                j = jj + 1;
                                m++;
                            }
                        }
                    }
                }
            }
        }

        // ---- full score of the initial map (tmscord, K.cu:396-440) over the list: slot pairs a < b
        int score = 0;
        {
            const int nq = (m + 3) >> 2;
            for (int sa = 0; sa + 1 < m; sa++) {
                const uint32_t i = *own_byte(lay.kw, sa), j = *own_byte(lay.lw, sa);
                const uint32_t qrow = (uint32_t)__mul24((int)i, N1P * 8), drow = lay.cells + (uint32_t)__mul24((int)j, (int)cell_row_bytes);
                for (int qd = sa >> 2; qd < nq; qd++) {
                    // slots <= sa of the first quad are switched off
                    const int below = sa + 1 - 4 * qd;
                    const uint32_t force = below <= 0 ? 0u : (0x04040404u >> (8 * (4 - below)));
                    uint2 q[4], d[4];
                    quad_load(*chain_word(lay.kw, qd, tid), *chain_word(lay.lw, qd, tid), qrow, drow, q, d);
                    score = quad_eval(q, d, force, score);
                }
            }
        }
        const int best_before = best;
        if (score > best) {
            best = score;
            if (LSOLN && beats_leader(score, restart)) {
                for (int w = 0; w < n1w; w++) {
                    bslab[w * T + tid] = *chain_word(lay.kw, w, tid);
                    bslab[(n1w + w) * T + tid] = *chain_word(lay.lw, w, tid);
                }
                bslab[2 * n1w * T + tid] = (uint32_t)m | 0xFF00u;           // no pending move
            }
        }

        // ---- 100 Metropolis steps, temperature 10 * 0.95^iter (K.cu:1030-1191)
        for (int iter = 0; iter < SAT_K_MAXITER; iter++) {
            const int rowoff = prowC[2 * iter], rowmax = prowC[2 * iter + 1];
            const uint4 r = philox_block(Q.seed_q, subseq, (uint32_t)(SAT_K_STEP_BLOCK0 + iter));

            // which query SSE moves (K.cu:1037-1042)
            const int ssei = scaled_index(draw32(r.x), n1d);

            // order window from the occupied-bit set (see sat_sa_kernel.hpp): p = highest matched query
            // SSE <= ssei, A its image; the candidates are the free same-type db SSEs between A and the
            // next occupied one
            const uint32_t lowpart = mapped & (0xFFFFFFFFu >> (31 - ssei));
            const int p = 31 ^ __builtin_clz(lowpart | 1u);
            const bool none = lowpart == 0u;
            const int slot_p = *own_byte(lay.pos, p);                    // garbage when `none`: never used then
            const int A = none ? 0 : (int)*own_byte(lay.lw, none ? 0 : slot_p);
            const bool oreal = !none && p == ssei;
            const int oldj = oreal ? A : NULLJ;
            const uint32_t above = 0xFFFFFFFEu << (A & 31);
            const uint32_t y = occ & above;
            const uint32_t gap = (y - 1u) & ~y & above;
            const bool empty = none || (y == 0u && ssei != n1 - 1);
            const uint32_t cand = empty ? 0u : (tmask[qtypes[ssei] * 4] & gap);

            const int cnt = __popc(cand);
            const int pick = cnt > 1 ? scaled_index(draw32(r.y), (double)cnt) : 0;
            int sel;
            {
                uint32_t c = cand;
                int left = pick;
                while (__builtin_amdgcn_ballot_w64(left > 0) != 0ull) {
                    const uint32_t go = left > 0 ? 1u : 0u;
                    c &= c - go;
                    left -= (int)go;
                }
                sel = __ffs(c) - 1;
            }
            const bool nreal = cnt != 0;
            const int newj = nreal ? sel : NULLJ;

            // ---- score change (deltasd, K.cu:502-535) over the matched pairs only
            int delta;
            {
                const int nitems = (int)oreal + (int)nreal;
                const bool small = m <= 8;
                const unsigned long long bo = __builtin_amdgcn_ballot_w64(oreal), bn = __builtin_amdgcn_ballot_w64(nreal),
                                         bs = __builtin_amdgcn_ballot_w64(small);
                const unsigned long long m1s = (bo | bn) & bs, m2s = bo & bn & bs, m1l = (bo | bn) & ~bs, m2l = bo & bn & ~bs;
                const int nshort = __popcll(m1s) + __popcll(m2s), nlong = __popcll(m1l) + __popcll(m2l);
                if (__builtin_amdgcn_ballot_w64(true) == ~0ull && nshort + 2 * nlong <= 64) {
                    // this lane's rank among the rows of its class
                    const unsigned long long c1 = small ? m1s : m1l, c2 = small ? m2s : m2l;
                    const int pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(c1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)c1, 0)) +
                                    __builtin_amdgcn_mbcnt_hi((uint32_t)(c2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)c2, 0));
                    const int slot1 = small ? pre : 63 - pre, slot2 = small ? pre + 1 : 62 - pre;
                    // item = row | moved SSE << 8 | owner chain << 16 | negate << 24 | octs << 25; the slot doubles
                    // as the row's accumulator (the owner subtracts what it wrote)
                    const uint32_t common = ((uint32_t)ssei << 8) | ((uint32_t)tid << 16) | ((uint32_t)((m + 7) >> 3) << 25);
                    const uint32_t item1 = (uint32_t)(oreal ? oldj : newj) | common | (oreal ? 1u << 24 : 0u);
                    const uint32_t item2 = (uint32_t)newj | common;
                    if (nitems >= 1) items[slot1] = item1;
                    if (nitems == 2) items[slot2] = item2;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    // serve: lane l < nshort takes short row l whole; lanes from the top take the long rows in
                    // pairs, each every second oct
                    {
                        const int l = lane_id & 63, back = 63 - l;
                        const bool is_short = l < nshort, is_long = back < 2 * nlong;
                        if (is_short || is_long) {
                            const int slot = is_short ? l : 63 - (back >> 1);
                            const int first = is_short ? 0 : (back & 1), stride = is_short ? 1 : 2;
                            const uint32_t it = items[slot];
                            const uint32_t row = it & 0xFFu, si = (it >> 8) & 0xFFu, owner = (it >> 16) & 0xFFu, octs = it >> 25;
                            const uint32_t qrow = (uint32_t)__mul24((int)si, N1P * 8), drow = lay.cells + (uint32_t)__mul24((int)row, (int)cell_row_bytes);
                            int v = 0;
                            for (uint32_t o = (uint32_t)first; o < octs; o += (uint32_t)stride) {
                                uint2 q[8], d[8];
                                // the second quad of an oct may lie past the list (m <= 4 mod 8): its word is
                                // then outside the chain's n1w words only when n1w is odd - clamp to the last word,
                                // whose slots past m are padding, and switch the duplicate off
                                const int w0 = (int)(2 * o), w1 = min((int)(2 * o + 1), n1w - 1);
                                const uint32_t dup = (int)(2 * o + 1) > n1w - 1 ? 0x04040404u : 0u;
                                quad_load(*chain_word(lay.kw, w0, (int)owner), *chain_word(lay.lw, w0, (int)owner), qrow, drow, q, d);
                                quad_load(*chain_word(lay.kw, w1, (int)owner), *chain_word(lay.lw, w1, (int)owner), qrow, drow, q + 4, d + 4);
                                v = quad_eval(q, d, 0u, v);
                                v = quad_eval(q + 4, d + 4, dup, v);
                            }
                            v = (it & (1u << 24)) ? -v : v;
                            __hip_atomic_fetch_add((lds_i32_t *)(items + slot), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    delta = 0;
                    if (nitems >= 1) delta = (int)(items[slot1] - item1);
                    if (nitems == 2) delta += (int)(items[slot2] - item2);
                } else {
                    // partial wave or too many rows for one round: every lane scores its own two rows
                    const uint32_t qrow = (uint32_t)__mul24(ssei, N1P * 8);
                    const uint32_t orow = lay.cells + (uint32_t)__mul24(oldj, (int)cell_row_bytes),
                                   nrow = lay.cells + (uint32_t)__mul24(newj, (int)cell_row_bytes);
                    const int nq = (m + 3) >> 2;
                    int sum_new = 0, sum_old = 0;
                    for (int qd = 0; qd < nq; qd++) {
                        const uint32_t kword = *chain_word(lay.kw, qd, tid), lword = *chain_word(lay.lw, qd, tid);
                        uint2 q[4], d[4];
                        quad_load(kword, lword, qrow, nrow, q, d);
                        sum_new = quad_eval(q, d, 0u, sum_new);
                        quad_load(kword, lword, qrow, orow, q, d);
                        sum_old = quad_eval(q, d, 0u, sum_old);
                    }
                    delta = sum_new - sum_old;
                }
            }
            const int newscore = score + delta;

            // best-so-far from the PROPOSED state, before the accept test (K.cu:1136-1155)
            if (LSOLN && newscore > best) {
                if (beats_leader(newscore, restart)) {
                    for (int w = 0; w < n1w; w++) {
                        bslab[w * T + tid] = *chain_word(lay.kw, w, tid);
                        bslab[(n1w + w) * T + tid] = *chain_word(lay.lw, w, tid);
                    }
                    bslab[2 * n1w * T + tid] = (uint32_t)m | ((uint32_t)ssei << 8) | ((uint32_t)newj << 16);
                }
            }
            best = max(best, newscore);

            // Metropolis: accept iff expf(delta / temp) > u, via the host-built table (sat_sa_kernel.hpp)
            const float u = draw32(r.z);
            const uint32_t nd = (uint32_t)min(max(1 - delta, 0), rowmax + 2);
            const float pacc = *(gptr_f32)((gptr_c)ptabG + (((uint32_t)rowoff + nd) << 2));
            const bool accept = pacc > u;
            score = accept ? newscore : score;

            // the accepted move on the lists (at most five bytes), the bit sets and m
            {
                const bool change = accept && (oreal || nreal);
                const bool grow = accept && nreal && !oreal, shrink = accept && oreal && !nreal;
                const int last = m - 1;
                // unmap: the last slot's pair moves into the freed slot
                const uint32_t klast = *own_byte(lay.kw, shrink ? last : 0), llast = *own_byte(lay.lw, shrink ? last : 0);
                const int slot = oreal ? slot_p : m;                               // the slot that is written
                if (change) *own_byte(lay.lw, slot) = (uint8_t)(shrink ? llast : (uint32_t)newj);
                if (grow || shrink) {
                    const uint32_t kk = shrink ? klast : (uint32_t)ssei;
                    *own_byte(lay.kw, slot) = (uint8_t)kk;
                    *own_byte(lay.pos, (int)kk) = (uint8_t)slot;
                }
                if (shrink) *own_byte(lay.lw, last) = (uint8_t)NULLJ;
                m += (int)grow - (int)shrink;
                const uint32_t oldbit = (accept && oreal) ? (1u << (oldj & 31)) : 0u;
                const uint32_t newbit = (accept && nreal) ? (1u << (newj & 31)) : 0u;
                occ = (occ & ~oldbit) | newbit;
                const uint32_t ibit = 1u << ssei;
                mapped = (mapped & ~(shrink ? ibit : 0u)) | (grow ? ibit : 0u);
            }
        }
        if (best > best_before) best_restart = (uint32_t)restart;
    }

    // ---- arg-max over restarts; ties go to the lowest restart index (K.cu:1024, 1137, 1211)
    unsigned long long key = any
        ? (((unsigned long long)(uint32_t)(best + 0x40000000)) << 32) | (0xFFFFFFFFu - best_restart)
        : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long other = __shfl_xor(key, off, 64);
        key = other > key ? other : key;
    }
    const int wave = lane_id >> 6, nwaves = (T + 63) >> 6;
    if ((lane_id & 63) == 0) red[wave] = key;
    __syncthreads();
    unsigned long long win = red[0];
    for (int w = 1; w < nwaves; w++) win = red[w] > win ? red[w] : win;

    const uint32_t win_restart = 0xFFFFFFFFu - (uint32_t)(win & 0xFFFFFFFFu);
    if (lane_id == 0) Q.scores[e] = (int)(uint32_t)(win >> 32) - 0x40000000;
    if (LSOLN && any && best_restart == win_restart &&
        ((((unsigned long long)(uint32_t)(best + 0x40000000)) << 32) | (0xFFFFFFFFu - best_restart)) == win) {
        // decode the winner's best list: matched pairs, then the pending move of the proposed state
        int8_t *out = Q.ssemaps + (size_t)e * n1;
        for (int i = 0; i < n1; i++) out[i] = -1;
        const uint32_t tail = bslab[2 * n1w * T + tid];
        const int bm = (int)(tail & 0xFFu), mi = (int)((tail >> 8) & 0xFFu), mj = (int)(tail >> 16);
        const uint8_t *kb = reinterpret_cast<const uint8_t *>(bslab), *lb = reinterpret_cast<const uint8_t *>(bslab + (size_t)n1w * T);
        for (int s = 0; s < bm; s++) {
            const int k = kb[(((s >> 2) * T + tid) << 2) + (s & 3)], l = lb[(((s >> 2) * T + tid) << 2) + (s & 3)];
            out[k] = (int8_t)l;
        }
        if (mi != 0xFF) out[mi] = (int8_t)(mj == NULLJ ? -1 : mj);
    }
}
