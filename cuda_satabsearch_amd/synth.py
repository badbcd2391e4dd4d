"""Seeded synthetic tableau + SSE-distance-matrix databases.

There is no network and the ASTRAL-derived databases are not redistributable here, so
throughput is measured on synthetic structures of the named shape:

* order n: fixed, or uniform on [lo, hi] (optionally sorted ascending, like the
  reference's `convdb2.py -s`);
* SSE types i.i.d. e 0.51 / xa 0.35 / xg 0.13 / xi 0.01 (measured on the reference's
  586-entry example database);
* off-diagonal tableau codes drawn from the eight codes the reference's angle binning
  can emit (scripts/pttableau.py:434-469), with the frequencies measured on that
  database: PE .11 PD .07 RD .14 RT .18 OT .13 OS .07 LS .14 LE .17;
* distances: SSE midpoints follow a 3-D random walk with step length ~ N(10 A, 3 A),
  pairwise Euclidean distances rounded to 3 decimals (the ASCII format is %6.3f,
  scripts/convdb2.py:214-226) and kept below 100 A so the 7-column parse quirk of the
  reader is not triggered.

Everything is deterministic in (seed, index, orders of the whole database): structures are
generated in chunks of 1024 at the largest order of their chunk, so every rank of a multi-GPU
run can generate its own shard and gets the matching slice of the whole database.
"""
import numpy as np

from .structures import StructSet

# code byte = (hi << 4) | lo, hi: P0 R1 O2 L3, lo: E0 D1 S2 T3  (parsetableaux.c:13-33)
_CODES = np.array([0x00, 0x01, 0x11, 0x13, 0x23, 0x22, 0x32, 0x30], dtype=np.uint8)   # PE PD RD RT OT OS LS LE
_CODE_P = np.array([.11, .07, .14, .18, .13, .07, .14, .17])
_TYPES = np.array([0, 1, 3, 2], dtype=np.uint8)       # e xa xg xi
_TYPE_P = np.array([.51, .35, .13, .01])

DB_SEED = 0x5A7AB5EA
QUERY_SEED = 0xC0FFEE


def _orders(n, lo, hi, seed, sort):
    if lo == hi:
        return np.full(n, lo, np.int32)
    rng = np.random.Generator(np.random.Philox(key=seed ^ 0x0DDBA11))
    o = rng.integers(lo, hi + 1, size=n).astype(np.int32)
    return np.sort(o) if sort else o


CHUNK = 1024   # structures per generation chunk: chunk c is a pure function of (seed, c)


# 256-entry lookup tables: a uniform byte -> code / type with the measured frequencies (to 1/256)
def _lut(values, probs):
    edges = np.round(np.cumsum(probs / probs.sum()) * 256).astype(int)
    lut = np.empty(256, np.uint8)
    lo = 0
    for v, hi in zip(values, edges):
        lut[lo:hi] = v
        lo = hi
    lut[lo:] = values[-1]
    return lut


_CODE_LUT = _lut(_CODES, _CODE_P)
_TYPE_LUT = _lut(_TYPES, _TYPE_P)


def _chunk(c, nmax, seed):
    """Structures c*CHUNK .. c*CHUNK+CHUNK-1 at order nmax: (tabs, dists) [CHUNK, nmax, nmax]."""
    n = CHUNK
    rng = np.random.Generator(np.random.Philox(key=seed, counter=[0, 0, c, nmax]))
    types = _TYPE_LUT[rng.integers(0, 256, size=(n, nmax), dtype=np.uint8)]
    codes = _CODE_LUT[rng.integers(0, 256, size=(n, nmax, nmax), dtype=np.uint8)]
    step_len = np.clip(rng.normal(10.0, 3.0, size=(n, nmax)), 3.8, 16.0)
    direction = rng.normal(size=(n, nmax, 3))
    direction /= np.linalg.norm(direction, axis=2, keepdims=True)
    pos = np.cumsum(direction * step_len[:, :, None] * 0.75, axis=1)
    # pairwise distances through the Gram matrix (float64), rounded to the 3 decimals of the ASCII format
    sq = (pos * pos).sum(axis=2)
    d2 = sq[:, :, None] + sq[:, None, :] - 2.0 * np.matmul(pos, pos.transpose(0, 2, 1))
    d = np.sqrt(np.maximum(d2, 0.0))
    d = np.minimum(np.round(d, 3), 99.999).astype(np.float32)
    ii, jj = np.tril_indices(nmax)
    tabs = codes
    tabs[:, jj, ii] = codes[:, ii, jj]
    dists = d
    dists[:, jj, ii] = d[:, ii, jj]
    idx = np.arange(nmax)
    tabs[:, idx, idx] = types
    dists[:, idx, idx] = types.astype(np.float32)
    return tabs, dists


def orders_c5(total, seed=DB_SEED, sort=True):
    """Orders of the large-structure workload C5 of SURVEY.md section 8d: uniform on [8, 96], and
    1 % of the entries uniform on [97, 111] (the reference's second, "large" pass)."""
    rng = np.random.Generator(np.random.Philox(key=seed ^ 0xC5C5))
    o = rng.integers(8, 97, size=total).astype(np.int32)
    big = rng.random(total) < 0.01
    o[big] = rng.integers(97, 112, size=int(big.sum())).astype(np.int32)
    return np.sort(o) if sort else o


def make_db(n, order_lo=32, order_hi=None, seed=DB_SEED, sort=True, first_index=0, total=None, orders=None,
            name_format="s%07d"):
    """Synthetic database of n structures: entries first_index .. first_index+n-1 of a
    database of `total` structures, so every rank can generate its own shard.  A
    structure of order m is the leading m x m block of its chunk's sample.
    `orders` (length `total`) overrides the uniform draw of the orders."""
    order_hi = order_lo if order_hi is None else order_hi
    total = n + first_index if total is None else total
    if orders is not None:
        all_orders = np.ascontiguousarray(orders, dtype=np.int32)
        assert all_orders.shape[0] == total
        order_lo, order_hi = int(all_orders.min()), int(all_orders.max())
    else:
        all_orders = _orders(total, order_lo, order_hi, seed, sort)
    orders = all_orders[first_index:first_index + n]
    ncell = orders.astype(np.int64) * (orders + 1) // 2
    cell_off = np.concatenate([[0], np.cumsum(ncell)[:-1]]).astype(np.int64)
    tab = np.empty(int(ncell.sum()), np.uint8)
    dist = np.empty(int(ncell.sum()), np.float32)
    tril = {}
    for c in range(first_index // CHUNK, (first_index + n - 1) // CHUNK + 1):
        # a chunk is generated at the largest order it holds IN THE WHOLE DATABASE, so that a shard
        # generated on its own holds the same structures as the matching slice of the whole
        nmax = int(all_orders[c * CHUNK:(c + 1) * CHUNK].max())
        tabs, dists = _chunk(c, nmax, seed)
        k0 = max(first_index, c * CHUNK)
        k1 = min(first_index + n, (c + 1) * CHUNK)
        if order_lo == order_hi:
            ii, jj = tril.setdefault(nmax, np.tril_indices(nmax))
            o = int(cell_off[k0 - first_index])
            sl = slice(k0 - c * CHUNK, k1 - c * CHUNK)
            tab[o:o + (k1 - k0) * ii.size] = tabs[sl][:, ii, jj].reshape(-1)
            dist[o:o + (k1 - k0) * ii.size] = dists[sl][:, ii, jj].reshape(-1)
            continue
        # entries of one order at a time: one fancy-index gather per (chunk, order)
        local = orders[k0 - first_index:k1 - first_index]
        for m in np.unique(local):
            m = int(m)
            ii, jj = tril.setdefault(m, np.tril_indices(m))
            ks = np.nonzero(local == m)[0]
            src_t = tabs[ks + (k0 - c * CHUNK)][:, ii, jj]
            src_d = dists[ks + (k0 - c * CHUNK)][:, ii, jj]
            dst = (cell_off[ks + (k0 - first_index)][:, None] + np.arange(ii.size, dtype=np.int64)[None, :]).reshape(-1)
            tab[dst] = src_t.reshape(-1)
            dist[dst] = src_d.reshape(-1)
    names = [name_format % (first_index + k) for k in range(n)]
    return StructSet(orders, names, cell_off, tab, dist)


def make_query(order=32, seed=QUERY_SEED):
    """A random query structure; returns (tab[n,n], dist[n,n], ssetypes[n])."""
    tabs, dists = _chunk(0, order, seed)
    t, d = tabs[0].copy(), dists[0].copy()
    return t, d, np.diagonal(t).copy()


def planted_query(db: StructSet, s, keep=0.75, jitter=1.0, seed=QUERY_SEED):
    """Query derived from db structure s: a random `keep` fraction of its SSEs (order
    kept) with distances jittered by up to +-jitter A, so that a high-scoring match
    exists in the database."""
    rng = np.random.Generator(np.random.Philox(key=seed ^ (s + 1)))
    n = int(db.orders[s])
    m = max(2, int(round(n * keep)))
    sel = np.sort(rng.choice(n, size=m, replace=False))
    t, d = db.dense(s)
    t, d = t[np.ix_(sel, sel)].copy(), d[np.ix_(sel, sel)].copy()
    noise = rng.uniform(-jitter, jitter, size=(m, m)).astype(np.float32)
    noise = np.tril(noise, -1)
    noise = noise + noise.T
    types = np.diagonal(t).copy()
    d = np.round(np.abs(d + noise), 3).astype(np.float32)
    idx = np.arange(m)
    d[idx, idx] = types.astype(np.float32)
    return t, d, types


def write_ascii(db: StructSet, path):
    """Write the reference's ASCII format (scripts/convdb2.py:214-226): header
    '%-8s %4d', rows of 2-letter codes + blank, rows of '%6.3f ', blank line between.
    (Python loop, kept as the independent check of the C writer StructSet.write_ascii.)"""
    hi = "PROL?"
    lo = "EDST?"
    tname = {0: "e  ", 1: "xa ", 2: "xi ", 3: "xg "}
    with open(path, "w") as f:
        for s in range(len(db)):
            n = int(db.orders[s])
            t, d = db.dense(s)
            f.write("%-8s %4d\n" % (db.names[s], n))
            for i in range(n):
                row = []
                for j in range(i + 1):
                    if i == j:
                        row.append(tname[int(t[i, j])])
                    else:
                        c = int(t[i, j])
                        row.append(hi[c >> 4] + lo[c & 15] + " ")
                f.write("".join(row) + "\n")
            for i in range(n):
                f.write("".join("%6.3f " % d[i, j] for j in range(i + 1)) + "\n")
            f.write("\n")
