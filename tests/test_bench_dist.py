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


def _bench(args, extra_env=None, timeout=300):
    env = dict(os.environ, BLASTED_BENCH_DRYRUN="1", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, text=True,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def test_gpus_flag_launches_that_many_replicas():
    """`python bench.py --gpus 2` without a launcher starts two replica processes itself (before anything
    touches a GPU) and rank 0 prints ONE line with n_gpus = 2; the dry-run mode swaps the kernels for a sleep
    and RCCL for gloo so that the launcher, the rendezvous and the rank bookkeeping run here on CPU."""
    res = _bench(["--gpus", "2", "--steps", "5", "--warmup", "1"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["replicas"] == 2 and d["steps"] == 5
    assert d["config"]["workload"] == "poisson3d_256_bs4_async_block_ilu0_apply"
    # whole-job throughput: both replicas' sweeps over the slowest rank's time
    assert abs(d["value"] - 2 * 3 * 5 / (d["ms_per_step"] * 5e-3)) < 1e-6 * d["value"]
    one = json.loads([l for l in _bench(["--steps", "5", "--warmup", "1"]).stdout.splitlines() if l.startswith("{")][0])
    assert one["n_gpus"] == 1


def test_gpus_flag_must_match_the_launcher():
    res = _bench(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode != 0 and "does not match" in res.stderr


def test_gpus_flag_under_torchrun():
    env = dict(os.environ, BLASTED_BENCH_DRYRUN="1", MASTER_ADDR="127.0.0.1")
    out = subprocess.check_output(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
         "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "3"],
        env=env, text=True, stderr=subprocess.STDOUT, timeout=300)
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["workload"] == "poisson3d_256_bs4_async_block_sgs_relaxation"
    assert d["config"]["napplysweeps"] == 5


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
    # the same formulas from an actual pattern's counts (what the unstructured config uses)
    from blasted_amd import workloads
    m = workloads.poisson3d(8, 4, grid="uniform")
    nb, nnzb, nnzl, nnzu = bench.matrix_counts(m)
    assert bench.pattern_bytes(nb, nnzb, nnzl, nnzu, nnzl, 4, nfixed=nnzu) == bench.algorithmic_bytes(6, 4)
    # the bytes an in-place sweep touches once the pair-less upper blocks are left alone: 3 arrays x their blocks less
    ab6 = bench.algorithmic_bytes(6, 4)
    assert ab6["factor_sweep"] - ab6["factor_sweep_touched"] == 3 * nnzu * 128
