"""bench.py as the driver runs it, on the one-GPU box: the N = 1 line, and the N > 1 flow rehearsed with two
ranks on GPU 0 (gloo carries the gather: RCCL refuses two ranks on one device) - started by bench.py itself,
with no torch.distributed environment, which is how a plain `python bench.py --gpus N` must work."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                       timeout=timeout, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_one_gpu_line_carries_the_contract_keys():
    r = _bench("--steps", "5", "--warmup", "1", "--entries", "20000", "--no-cpu-baseline")
    assert r["n_gpus"] == 1 and r["ranks_seen"] == 1 and r["scaling"] == "weak" and r["steps"] == 5
    assert r["oracle_sample_ok"] is True and len(r["oracle_sample"]) == 16
    assert r["roofline"]["bound"] == "hbm" and 0 < r["roofline"]["frac"] < 1
    assert abs(r["value"] - 20000 * 5 / (r["ms_per_step"] * 5e-3)) < 1e-6 * r["value"]
    assert r["other_regimes"]["all_hit_scorings_per_sec"] > 0 and r["other_regimes"]["planted_query_scorings_per_sec"] > 0
    assert "sat_sa_kernel<32, 1, false, 1, 4, 0>" in r["roofline"]["kernel"]


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_started_by_bench_itself(scaling):
    argv = ["--gpus", "2", "--backend", "gloo", "--all-ranks-on-device0", "--steps", "5", "--warmup", "1",
            "--scaling", scaling]
    argv += ["--entries", "20000"] if scaling == "weak" else ["--total", "30001"]     # 30001: unequal shards, padded rows
    r = _bench(*argv)
    total = 40000 if scaling == "weak" else 30001
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["scaling"] == scaling
    assert r["gathered_scores"] == total and r["config"]["db_entries"] == total
    assert r["oracle_sample_ok"] is True
    assert len(r["kernel_ms_by_rank"]["all"]) == 2 and r["gather_ms_alone"] > 0
    assert abs(r["value"] - total * 5 / (r["ms_per_step"] * 5e-3)) < 1e-6 * r["value"]


def test_single_process_multi_gpu_entry_points():
    """--single-process: the product's sat_multi_* calls (two contexts on GPU 0 here: the peer-copy gather)."""
    r = _bench("--single-process", "--gpus", "2", "--all-ranks-on-device0", "--steps", "3", "--warmup", "1",
               "--entries", "10000")
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["oracle_sample_ok"] is True
    assert r["shards"][0] == 0 and r["shards"][-1] == 20000
