"""Structure sets: a database or a list of queries as packed lower triangles.

Mirrors csrc/host/sat_parse.h (which follows the reference reader
nvcc_src_current/parsetableaux.c): `orders[s]`, `names[s]`, `cell_off[s]` and the
packed `tab` (uint8 codes) / `dist` (float32 Angstrom) arrays, cell (i, j), j <= i, of
structure s at cell_off[s] + i*(i+1)/2 + j.  Parsing is done by the C reader.
"""
import ctypes as C

import numpy as np

from . import _native

MAXDIM = _native.MAXDIM
MAXDIM_SMALL = 96     # saparams.h:17 MAXDIM_GPU, boundary of the reference's two passes


class StructSet:
    def __init__(self, orders, names, cell_off, tab, dist):
        self.orders = np.ascontiguousarray(orders, dtype=np.int32)
        self.names = list(names)
        self.cell_off = np.ascontiguousarray(cell_off, dtype=np.int64)
        self.tab = np.ascontiguousarray(tab, dtype=np.uint8)
        self.dist = np.ascontiguousarray(dist, dtype=np.float32)

    def __len__(self):
        return int(self.orders.shape[0])

    # ---- construction -------------------------------------------------------
    @classmethod
    def _from_cset(cls, cset):
        host = _native.host_lib()
        count, cells = cset.count, cset.cells
        orders = np.ctypeslib.as_array(cset.order, shape=(max(count, 1),))[:count].copy()
        cell_off = np.ctypeslib.as_array(cset.cell_off, shape=(max(count, 1),))[:count].copy()
        tab = np.ctypeslib.as_array(cset.tab, shape=(max(cells, 1),))[:cells].copy()
        dist = np.ctypeslib.as_array(cset.dist, shape=(max(cells, 1),))[:cells].copy()
        raw = C.string_at(cset.name, count * (_native.LABELSIZE + 1)) if count else b""
        names = [raw[i * 9:(i + 1) * 9].split(b"\0", 1)[0].decode("latin-1") for i in range(count)]
        host.sat_set_free(C.byref(cset))
        return cls(orders, names, cell_off, tab, dist)

    @classmethod
    def read(cls, path, what="database", skip_header_lines=0, stdio=False):
        """Parse an ASCII tableau + distance-matrix file with the C reader.

        Whole files go through the mmap reader (sat_read_structures_file); with
        skip_header_lines=2 the 'dbfile' and 'LTYPE LORDER LSOLN' lines of a query file
        are skipped first (the reference reads them with fscanf before read_queries,
        cudaSaTabsearch.cu:667-684) and the stdio reader is used.  stdio=True forces the
        stdio reader (the one the reference's semantics were pinned with)."""
        host = _native.host_lib()
        cset = _native.StructSetC()
        host.sat_set_init(C.byref(cset))
        if skip_header_lines == 0 and not stdio:
            n = host.sat_read_structures_file(str(path).encode(), C.byref(cset), what.encode())
            if n < 0:
                host.sat_set_free(C.byref(cset))
                raise OSError(f"cannot read {path}")
            return cls._from_cset(cset)
        libc = _native.libc()
        fp = libc.fopen(str(path).encode(), b"r")
        if not fp:
            raise OSError(f"cannot open {path}")
        try:
            buf = C.create_string_buffer(4096)
            for _ in range(skip_header_lines):
                libc.fgets(buf, 4096, fp)
            n = host.sat_read_structures(fp, C.byref(cset), what.encode())
            if n < 0:
                host.sat_set_free(C.byref(cset))
                raise MemoryError("sat_read_structures failed")
        finally:
            libc.fclose(fp)
        return cls._from_cset(cset)

    def _to_cset(self):
        """A StructSetC viewing this set's arrays (valid while self is alive)."""
        cset = _native.StructSetC()
        names = b"".join(n.encode("latin-1")[:8].ljust(9, b"\0") for n in self.names)
        self._name_buf = C.create_string_buffer(names, len(names) + 1)
        cset.count = cset.capacity = len(self)
        cset.cells = cset.cells_cap = int(self.tab.size)
        cset.order = self.orders.ctypes.data_as(C.POINTER(C.c_int))
        cset.name = C.cast(self._name_buf, C.POINTER(C.c_char))
        cset.cell_off = self.cell_off.ctypes.data_as(C.POINTER(C.c_int64))
        cset.tab = self.tab.ctypes.data_as(C.POINTER(C.c_uint8))
        cset.dist = self.dist.ctypes.data_as(C.POINTER(C.c_float))
        return cset

    def save_binary(self, path):
        cset = self._to_cset()
        if _native.host_lib().sat_set_save_binary(C.byref(cset), str(path).encode()) != 0:
            raise OSError(f"cannot write {path}")

    def write_ascii(self, path):
        """Write the set in the reference's ASCII database format (C writer, sat_parse.h)."""
        cset = self._to_cset()
        if _native.host_lib().sat_set_write_ascii(C.byref(cset), str(path).encode()) != 0:
            raise OSError(f"cannot write {path}")

    @classmethod
    def load_binary(cls, path):
        host = _native.host_lib()
        cset = _native.StructSetC()
        host.sat_set_init(C.byref(cset))
        if host.sat_set_load_binary(str(path).encode(), C.byref(cset)) != 0:
            raise OSError(f"{path} is not a valid structure-set image")
        return cls._from_cset(cset)

    @classmethod
    def from_dense(cls, orders, tabs, dmats, names=None):
        """Build from dense [N, P, P] arrays (only the lower triangle is read)."""
        orders = np.asarray(orders, dtype=np.int32)
        ncell = orders.astype(np.int64) * (orders + 1) // 2
        cell_off = np.concatenate([[0], np.cumsum(ncell)[:-1]]).astype(np.int64)
        tab = np.empty(int(ncell.sum()), np.uint8)
        dist = np.empty(int(ncell.sum()), np.float32)
        for s, n in enumerate(orders):
            ii, jj = np.tril_indices(int(n))
            o = int(cell_off[s])
            tab[o:o + ii.size] = tabs[s][ii, jj]
            dist[o:o + ii.size] = dmats[s][ii, jj]
        if names is None:
            names = [f"s{i:07d}" for i in range(len(orders))]
        return cls(orders, names, cell_off, tab, dist)

    # ---- views --------------------------------------------------------------
    def dense(self, s, pitch=None):
        """(tab[P,P] uint8, dist[P,P] float32) symmetric expansion of structure s."""
        n = int(self.orders[s])
        pitch = pitch or n
        t = np.zeros((pitch, pitch), np.uint8)
        d = np.zeros((pitch, pitch), np.float32)
        ii, jj = np.tril_indices(n)
        o = int(self.cell_off[s])
        t[ii, jj] = self.tab[o:o + ii.size]
        t[jj, ii] = self.tab[o:o + ii.size]
        d[ii, jj] = self.dist[o:o + ii.size]
        d[jj, ii] = self.dist[o:o + ii.size]
        return t, d

    def ssetypes(self, s):
        n = int(self.orders[s])
        i = np.arange(n, dtype=np.int64)
        return self.tab[int(self.cell_off[s]) + i * (i + 1) // 2 + i].copy()

    def subset(self, index):
        """New set holding structures `index` (array of positions), cells re-packed."""
        index = np.asarray(index, dtype=np.int64)
        orders = self.orders[index]
        ncell = orders.astype(np.int64) * (orders + 1) // 2
        cell_off = np.concatenate([[0], np.cumsum(ncell)[:-1]]).astype(np.int64) if len(index) else np.zeros(0, np.int64)
        total = int(ncell.sum())
        tab = np.empty(total, np.uint8)
        dist = np.empty(total, np.float32)
        # gather cell ranges without a Python loop over cells
        src_start = self.cell_off[index]
        pos = np.repeat(src_start - cell_off, ncell) + np.arange(total, dtype=np.int64)
        tab[:] = self.tab[pos]
        dist[:] = self.dist[pos]
        return StructSet(orders, [self.names[i] for i in index], cell_off, tab, dist)
