"""End-to-end timing of the command line on a synthetic ASCII database (parse + upload +
search + print), GPU mode vs the kernel-only window it reports on stderr."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_satabsearch_amd as sat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cli = os.path.join(root, "cuda_satabsearch_amd", "bin", "satabsearch")
tmp = tempfile.mkdtemp()
db = sat.synth.make_db(n, 8, 32)
t = time.time(); sat.synth.write_ascii(db, os.path.join(tmp, "db.ascii")); print(f"wrote {os.path.getsize(os.path.join(tmp, 'db.ascii'))/1e6:.0f} MB ASCII in {time.time()-t:.1f}s")
qt, qd, qtypes = sat.synth.make_query(32)
qs = sat.StructSet.from_dense([32], [qt], [qd], ["SYNQ32"])
sat.synth.write_ascii(qs, os.path.join(tmp, "q.body"))
open(os.path.join(tmp, "q.input"), "w").write("db.ascii\nT T F\n" + open(os.path.join(tmp, "q.body")).read())
for args in (["-r", "128"], ["-r", "128", "-b"], ["-r", "128", "-b"], ["-r", "128", "-k", "10"]):
    t = time.time()
    p = subprocess.run([cli, *args], stdin=open(os.path.join(tmp, "q.input")), cwd=tmp, capture_output=True, text=True)
    wall = time.time() - t
    info = [l for l in p.stderr.splitlines() if "Loaded" in l or "GPU execution" in l or "Copied" in l]
    print(args, f"wall {wall:.2f}s rc={p.returncode} rows={p.stdout.count(chr(10))} |", " | ".join(info))
