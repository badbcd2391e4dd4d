"""End-to-end wall time of the command line on the paper's workload shape: 200 SIDs (-q) against a 15 000-entry
database (python scripts/cli_query_list_e2e.py; optionally a second binary name to compare with)."""
import os, sys, time, subprocess, tempfile, hashlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import cuda_satabsearch_amd as sat
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
d = tempfile.mkdtemp()
db = sat.synth.make_db(15000, 4, 40, sort=True, name_format="s%06d")
db.write_ascii(d + "/db.ascii")
pick = np.random.default_rng(5).choice(len(db), 200, replace=False)
sids = "".join(db.names[int(s)] + "\n" for s in pick).encode()
for exe in (sys.argv[1:] or ["satabsearch"]) * 2:
    t0 = time.time()
    p = subprocess.run([root + "/cuda_satabsearch_amd/bin/" + exe, "-r", "128", "-q", "db.ascii"], input=sids, cwd=d, capture_output=True)
    dt = time.time() - t0
    gpu = [l for l in p.stderr.decode().splitlines() if l.startswith("GPU execution time")]
    print(exe, "rc", p.returncode, "%.2f s end to end" % dt, len(p.stdout) // 1000000, "MB", hashlib.md5(p.stdout).hexdigest()[:10], gpu[:1])
