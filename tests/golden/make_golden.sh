#!/bin/bash
# tests/golden/make_golden.sh - regenerate the golden fixtures.
#
# Runs ONLY in the build container, where the reference tree is mounted at
# /root/reference.  It (1) copies the reference's example DATA files (queries and
# tiny databases; MIT-licensed data, no source code) into inputs/, (2) runs the
# reference's own host path, compiled from its sources by oracle/Makefile into
# oracle/_ref/, on each of them and stores the stdout under expected/, and (3)
# copies the stdout the reference recorded from a real 2013 "-c -r4096" run.
# The GPU box never runs this; tests read only inputs/ and expected/.
set -euo pipefail
here=$(cd "$(dirname "$0")" && pwd)
repo=$(cd "$here/../.." && pwd)
ref=/root/reference/nvcc_src_current
old=/root/reference/old/nvcc_src_cuda5
make -s -C "$repo/oracle" ref

for f in d1ubia_.input d1ae6h1.input d2phlb1.input d2phlb1.input2 d2phlb1.input3 \
         1qlp_sheetbc.input d1twfa_.input multiquery.input \
         tableauxdistmatrixdb.test.ascii tableauxdistmatrixdb.test2.ascii \
         d1qlpa_.ascii d1qwra_.ascii d2pq6a1.ascii; do
  cp "$ref/$f" "$here/inputs/$f"
done
gzip -9 -n -c "$ref/tableauxdistmatrixdb.small.ascii" > "$here/inputs/tableauxdistmatrixdb.small.ascii.gz"

# config C1 (BASELINE.json configs[0]): the 8-SSE d1ubia_ query against the 586-entry db, T T F
{ echo tableauxdistmatrixdb.small.ascii; echo "T T F"; tail -n +3 "$ref/d1ubia_.input"; } > "$here/inputs/c1_d1ubia_small.input"
# LORDER = F and LSOLN = T variants of the 19-SSE query (exercise the [0,n2) branch, kernel.cu:1079-1083)
{ echo tableauxdistmatrixdb.small.ascii; echo "T F T"; tail -n +3 "$ref/d2phlb1.input"; } > "$here/inputs/d2phlb1_TFT.input"
{ echo tableauxdistmatrixdb.small.ascii; echo "T T T"; tail -n +3 "$ref/d2phlb1.input"; } > "$here/inputs/d2phlb1_TTT.input"

work=$(mktemp -d)
trap 'rm -rf "$work"' EXIT
cp "$here"/inputs/* "$work"/
gunzip "$work/tableauxdistmatrixdb.small.ascii.gz"
cd "$work"
for f in d1ubia_.input d1ae6h1.input d2phlb1.input d2phlb1.input2 d2phlb1.input3 \
         1qlp_sheetbc.input d1twfa_.input multiquery.input c1_d1ubia_small.input \
         d2phlb1_TFT.input d2phlb1_TTT.input; do
  "$repo/oracle/_ref/ref_oracle" -c -r128 < "$f" > "$here/expected/${f%.input*}${f##*.input}.r128.out" 2>/dev/null
done
"$repo/oracle/_ref/ref_oracle" -c -r16 < d1twfa_.input > "$here/expected/d1twfa_.r16.out" 2>/dev/null
# per-iteration trace of the reference's DEBUG build, one restart (step-level fixture)
"$repo/oracle/_ref/ref_oracle_debug" -c -r1 < d1ubia_.input > "$here/expected/d1ubia_.r1.trace.stdout" 2> "$here/expected/d1ubia_.r1.trace.stderr"
# the reference's own recorded run (2013 sources, MAXDIM_GPU = 32): "-c -r4096 < d2phlb1.input"
cp "$old/cpu_cudaSaTabsearch.o1462445" "$here/expected/recorded_2013_d2phlb1.r4096.out"
ls -la "$here/expected"

# The README's worked example (README_example_usage.txt:10-27 query, :43-49 printed rows): the query
# body is data; the rows the README prints are NOT what the current sources give at any drand48 seed
# (first row 6 there, 11 here for seeds 1234, 1..5) - kept as readme_1ubq.printed_head.txt and
# documented as a stale vector of an older build, not a parity target (tests/test_oracle_golden.py).
{ echo tableauxdistmatrixdb.small.ascii; echo "T T F"; sed -n 10,27p /root/reference/README_example_usage.txt | sed 's/^    //'; } > "$here/inputs/readme_1ubq.input"
cp "$here/inputs/readme_1ubq.input" "$work/"
"$repo/oracle/_ref/ref_oracle" -c -r128 < readme_1ubq.input > "$here/expected/readme_1ubq.r128.out" 2>/dev/null
sed -n 43,49p /root/reference/README_example_usage.txt | sed 's/^    //' > "$here/expected/readme_1ubq.printed_head.txt"

# -q mode (SIDs on stdin; the second one is upper case and longer than 7 characters: cut + case-insensitive
# lookup, cudaSaTabsearch.cu:657, 752): the reference's parser, kernel and statistics under ref_driver's
# replay of main's -q branch
printf 'd1kcul1\nD1NLDL1xyz\nd1lfwa2\n' > "$here/inputs/qmode_sids.txt"
"$repo/oracle/_ref/ref_oracle" -c -r16 -q tableauxdistmatrixdb.small.ascii < "$here/inputs/qmode_sids.txt" > "$here/expected/qmode_small.r16.out" 2>/dev/null
