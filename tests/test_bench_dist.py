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
    # ... but each of them is still READ once as the u_kj operand of the pair of the diagonal block below it (VERDICT r03
    # item 3a): every-array-once less the A read and the F write of the fixed blocks
    ab6 = bench.algorithmic_bytes(6, 4)
    assert ab6["factor_sweep"] - ab6["factor_sweep_touched"] == 2 * nnzu * 128
    # 256^3 bs=4: 33.6 GB per in-place sweep, 37.9 GB for the fused first sweep of a build (reads A once, writes every
    # block, reads a row's own lower blocks back); the counters showed 39.06 GB per launch averaged over a 3-sweep build
    # (profiles/r03z_c2_pmc.json): the touched count of a build is within 15 % of it
    assert abs(ab["factor_sweep_touched"] - 33.6e9) < 0.1e9 and abs(ab["factor_sweep_fused_first"] - 37.85e9) < 0.1e9
    build = (ab["factor_sweep_fused_first"] + 2 * ab["factor_sweep_touched"]) / 3
    assert 0.85 < build / 39.06e9 < 1.0
    # a pattern whose fixed blocks are not all operands: counted separately
    pb = bench.pattern_bytes(nb, nnzb, nnzl, nnzu, nnzl, 4, nfixed=nnzu, nfixed_operands=nnzu - 10)
    assert ab6["factor_sweep_touched"] - pb["factor_sweep_touched"] == 10 * 128


def test_live_traffic_picks_the_dominant_kernel_from_a_counter_csv(tmp_path):
    """bench.py's roofline.traffic comes from rocprofv3 --pmc passes it starts itself: the parser must pick the kernel
    the roofline is quoted on (the descending sweep for the apply ops, by its PART template argument and family) and
    average its launches -- checked here on a counter CSV of the shape rocprofv3 writes."""
    import bench
    rows = [("void bhip::sweepw_kernel<4, 0, 0, 0, 128, true, 2, 1, false, false, false, true>(bhip::SweepArgs)", 4000000.0)] * 9
    rows += [("void bhip::sweepw_kernel<4, 1, 1, 1, 128, true, 2, 1, false, false, false, true>(bhip::SweepArgs)", 5000000.0)] * 8
    rows += [("void bhip::sweepw_kernel<4, 1, 1, 1, 128, true, 2, 1, false, false, false, true>(bhip::SweepArgs)", 5200000.0)]
    rows += [("void bhip::(anonymous namespace)::sfw_kernel<4, true, 1, true, 2, false>(bhip::SweepArgs, int const*)", 9e6)] * 2
    rows += [("bhip::(anonymous namespace)::factor4_kernel(bhip::FactorArgs)", 17000000.0)] * 3
    rows += [("void at::native::vectorized_elementwise_kernel<4, foo>", 1.0)] * 40
    rows += [("void bhip::sweep_kernel<1, false, 1, 1, 3, 1, false>(bhip::SweepArgs)", 10000.0)] * 2
    f = tmp_path / "1_counter_collection.csv"
    with open(f, "w") as fh:
        fh.write("Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n")
        for i, (k, v) in enumerate(rows):
            fh.write('%d,%d,0,1,1,1,256,7,"%s",256,0,0,64,0,32,FETCH_SIZE,%r,0,1\n' % (i, i, k, v))
    name, avg, n = bench.dominant_kernel_counter(str(f), "ilu_apply")
    assert "sweepw_kernel<4, 1, 1, 1" in name and n == 9 and abs(avg - (8 * 5e6 + 5.2e6) / 9) < 1e-6
    name, avg, n = bench.dominant_kernel_counter(str(f), "factor")
    assert "factor4_kernel" in name and n == 3 and avg == 17000000.0
    assert bench.dominant_kernel_counter(str(f), "spmv") is None
