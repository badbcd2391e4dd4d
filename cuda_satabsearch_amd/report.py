"""Result rows: `name rawscore norm2score z-score p-value` exactly as the reference
prints them (nvcc_src_current/cudaSaTabsearch.cu:415-420 header, :445-453 rows), with
the statistics of csrc/host/sat_gumbel.c (reference gumbelstats.c)."""
from . import _native


def _g(x):
    """C printf("%g") formatting (Python's %g follows the same rules)."""
    return "%g" % x


def stats(score, n1, n2):
    host = _native.host_lib()
    norm2 = host.sat_norm2(int(score), int(n1), int(n2))
    z = host.sat_z_gumbel_trunc(norm2)
    p = host.sat_pv_gumbel(z)
    return norm2, z, p


def header_lines(qid, dbfile, ltype=True, lorder=True, lsoln=False):
    tf = lambda b: "T" if b else "F"
    return [f"# cudaSaTabsearch LTYPE = {tf(ltype)} LORDER = {tf(lorder)} LSOLN = {tf(lsoln)}",
            "# QUERY ID = %-8s" % qid,
            "# DBFILE = %-80s" % dbfile]


def result_lines(names, orders, scores, n1, ssemaps=None, wide_pvalue_gap=False):
    """wide_pvalue_gap reproduces the two blanks before the p-value that the reference's
    GPU path prints for the large-structure pass (cudaSaTabsearch.cu:1261)."""
    out = []
    gap = "  " if wide_pvalue_gap else " "
    for e, name in enumerate(names):
        norm2, z, p = stats(scores[e], n1, orders[e])
        out.append("%-8s %d %s %s%s%s" % (name, int(scores[e]), _g(norm2), _g(z), gap, _g(p)))
        if ssemaps is not None:
            for i in range(n1):
                j = int(ssemaps[e][i])
                if j >= 0:
                    out.append("%3d %3d" % (i + 1, j + 1))
    return out
