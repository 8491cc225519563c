"""bench.py's multi-GPU leg is "N independent replicas, barrier + max-over-ranks timing, no data-path
collective" (DESIGN.md section 7).  The rank bookkeeping is exercised here with world_size 2 on CPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_replica_timing_world_size_2_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.check_output(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "tests", "dist_worker.py")],
        env=env, text=True, stderr=subprocess.STDOUT, timeout=300)
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["world"] == 2 and abs(d["elapsed"] - 1.5) < 1e-12
    # whole-job throughput: both replicas' units over the slowest rank's time
    assert abs(d["value"] - 2 * 3 * 20 / 1.5) < 1e-9


def test_algorithmic_bytes_match_survey():
    sys.path.insert(0, ROOT)
    import bench
    ab = bench.algorithmic_bytes(256, 4)
    assert ab["nbrows"] == 16777216 and ab["nnzb"] == 117047296
    assert ab["ilu_pair"] == 18872795136
    assert ab["sgs_pair"] == 21020278784
    assert 2 * ab["sgs_relax_pass"] == 34255929344
    assert ab["factor_sweep"] == 46417838080
    assert ab["spmv"] == 16591093764
    assert ab["lower_sweep"] + ab["upper_sweep"] == ab["ilu_pair"]
