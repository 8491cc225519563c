#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config[1]:
  async block-ILU(0) apply, BSR bs=4, 3-D Poisson 256^3 (16.7 M block-rows, 15 GB of factor), 1 GPU.

A "step" is one preconditioner application z = U^-1 L^-1 r through the C ABI
(blasted_hip_ilu0_apply, device pointers): y := 0, `s` asynchronous lower sweeps, z := 0, `s`
asynchronous upper sweeps (s = --sweeps, default 3 as in the reference's calibration runs).
value = L+U sweep pairs per second over the whole job (all ranks); every input is resident in HBM
before the timed region.  Multi-GPU = independent replicas (the operator is the subdomain-local
preconditioner: the matrix is replicated per GPU, there is no data-path collective).

python bench.py [--gpus N] [--steps K] [--warmup W] [--n 256] [--sweeps 3] [--op ilu_apply]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(n, bs):
    """compulsory bytes, every array touched once per sweep."""
    nb = n ** 3
    nnzb = 7 * n ** 3 - 6 * n ** 2
    nnzl = 3 * n ** 3 - 3 * n ** 2
    B, S, I = 8 * bs * bs, 8 * bs, 4
    lower = nnzl * (B + I) + 2 * nb * I + 3 * nb * S
    upper = (nnzl + nb) * B + nnzl * I + 2 * nb * I + 3 * nb * S
    return {
        "lower_sweep": lower, "upper_sweep": upper, "ilu_pair": lower + upper,
        "sgs_pair": 2 * nnzl * (B + I) + 2 * nb * B + 4 * nb * I + 6 * nb * S,
        "sgs_relax_pass": 2 * nnzl * (B + I) + nb * B + 2 * nb * I + 3 * nb * S,
        "factor_sweep": 3 * nnzb * B + 2 * nnzb * I + 2 * nb * I + 2 * nnzl * I,
        "spmv": nnzb * (B + I) + (nb + 1) * I + 2 * nb * S,
        "nbrows": nb, "nnzb": nnzb,
    }


def cpu_baseline(op, nsample, bs, sweeps, full_unit_bytes, units_per_call, budget_s=12.0):
    """The oracle's threaded port of the reference loop nest (omp for schedule(dynamic,256) nowait),
    timed on this box's host cores on a bounded sample of the same workload (nsample^3 instead of
    256^3), scaled to the metric's unit by algorithmic bytes."""
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    import oracle
    from blasted_amd import workloads
    # one thread per CPU this process is really granted (affinity mask cut by the cgroup quota): more
    # threads than that only thrash; the count is what `cores` reports
    budget = oracle.cpu_budget()
    oracle.set_num_threads(budget)
    m = workloads.poisson3d(nsample + 2, bs, grid="uniform")
    r = workloads.rhs_vector(m["nbrows"] * bs)
    kw = {}
    if op == "ilu_apply":
        kw["iluvals"] = oracle.ilu0_factorize(m, None, 1, mode=oracle.GS_SERIAL)["iluvals"]
    elif op == "factor":
        kw["plist"] = oracle.ilu_positions(m)
    elif op != "spmv":
        kw["dblocks"] = oracle.jacobi_compute(m)
    t1 = oracle.time_op(op, m, r, sweeps, 256, 2, **kw)
    reps = max(3, min(200, int(budget_s / max(t1, 1e-4))))
    t = oracle.time_op(op, m, r, sweeps, 256, reps, **kw)
    ab = algorithmic_bytes(nsample, bs)
    sample_unit = {"ilu_apply": ab["ilu_pair"], "sgs_apply": ab["sgs_pair"], "sgs_relax": 2 * ab["sgs_relax_pass"],
                   "spmv": ab["spmv"], "factor": ab["factor_sweep"]}[op]
    return {
        "value": (units_per_call / t) * sample_unit / full_unit_bytes,
        "unit": "sweeps/s", "cores": oracle.num_threads(), "kind": "port",
        "achieved_gbps": sample_unit * units_per_call / t / 1e9,
        "sample": "oracle ASYNC_OMP (reference loop nest, chunk 256) %s on Poisson %d^3 bs=%d, %d sweeps per call, "
                  "min of %d calls = %.1f ms, %d OpenMP threads = the CPUs granted to this process (%d hardware "
                  "threads visible); scaled to the full size by algorithmic bytes" %
                  (op, nsample, bs, sweeps, reps, t * 1e3, oracle.num_threads(), os.cpu_count() or 0),
    }


def max_over_ranks(seconds, device):
    """Whole-job time = the slowest rank's time (all_reduce MAX; identity for a single process)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(world, units_per_step, steps, elapsed):
    """Replicas only: every rank processes units_per_step * steps units of its own copy."""
    return world * units_per_step * steps / elapsed


def measured_copy_gbps(dev, nbytes=1 << 31, reps=5):
    """Device copy bandwidth of this box (read + write bytes / time): the practical HBM ceiling the
    roofline fraction can be read against (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy)."""
    import torch
    a = torch.empty(nbytes // 8, dtype=torch.float64, device=dev).normal_()
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=256, help="grid points per axis (256 = BASELINE config)")
    ap.add_argument("--bs", type=int, default=4)
    ap.add_argument("--sweeps", type=int, default=3, help="napplysweeps")
    ap.add_argument("--build-sweeps", type=int, default=3)
    ap.add_argument("--op", default="ilu_apply", choices=["ilu_apply", "sgs_apply", "sgs_relax", "spmv", "factor"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-n", type=int, default=96)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from blasted_amd import capi, workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n, bs, s = args.n, args.bs, args.sweeps
    ab = algorithmic_bytes(n, bs)

    # ---- workload resident in HBM
    m = workloads.poisson3d_device(n, bs, dev, grid="uniform")
    r = workloads.rhs_vector_device(m["nbrows"] * bs, dev)
    z = torch.zeros_like(r)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    p = capi.Prec(local_rank, stream)
    p.set_matrix(m)

    if args.op in ("ilu_apply", "factor"):
        p.ilu0_factorize(args.build_sweeps, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
    else:
        p.jacobi_compute()
    torch.cuda.synchronize()

    if args.op == "ilu_apply":
        step = lambda: p.ilu0_apply(r, s, init=capi.INIT_A_ZERO, mode=capi.ASYNC, out=z)
        unit_bytes, units_per_step, kernel, kbytes = ab["ilu_pair"], s, "upper", ab["upper_sweep"]
        workload = "poisson3d_%d_bs%d_async_block_ilu0_apply" % (n, bs)
    elif args.op == "sgs_apply":
        step = lambda: p.sgs_apply(r, s, init=capi.INIT_A_ZERO, mode=capi.ASYNC, out=z)
        unit_bytes, units_per_step, kernel, kbytes = ab["sgs_pair"], s, "upper", ab["sgs_pair"] / 2
        workload = "poisson3d_%d_bs%d_async_block_sgs_apply" % (n, bs)
    elif args.op == "sgs_relax":
        step = lambda: p.sgs_relax(r, z, s, mode=capi.ASYNC)
        unit_bytes, units_per_step, kernel, kbytes = 2 * ab["sgs_relax_pass"], s, "upper", ab["sgs_relax_pass"]
        workload = "poisson3d_%d_bs%d_async_block_sgs_relaxation" % (n, bs)
    elif args.op == "spmv":
        step = lambda: p.spmv(r, out=z)
        unit_bytes, units_per_step, kernel, kbytes = ab["spmv"], 1, "lower", ab["spmv"]
        workload = "poisson3d_%d_bs%d_bsr_spmv" % (n, bs)
    else:
        step = lambda: p.ilu0_factorize(s, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
        unit_bytes, units_per_step, kernel, kbytes = ab["factor_sweep"], s, "lower", ab["factor_sweep"]
        workload = "poisson3d_%d_bs%d_async_block_ilu0_factor" % (n, bs)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    p.set_timing(True)
    p.get_timing(reset=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    tm = p.get_timing(reset=True)
    p.set_timing(False)

    elapsed = max_over_ranks(t1 - t0, dev)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = job_throughput(world, units_per_step, args.steps, elapsed)
        copy_gbps = measured_copy_gbps(dev)
        # practical read ceiling of this device: the matrix's own value array through a read-only kernel
        read_gbps = capi.measure_read_stream(m["vals"], reps=10)
        kms = tm[kernel + "_ms"] / max(tm[kernel + "_launches"], 1)
        achieved = kbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(args.op, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "precond_apply_sweeps_per_sec", "value": value, "unit": "sweeps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "grid_points": n, "block_size": bs, "nbrows": ab["nbrows"],
                       "nnzb": ab["nnzb"], "napplysweeps": s, "nbuildsweeps": args.build_sweeps,
                       "sweep_mode": "async", "grid": "uniform", "replicas": world,
                       "unit_definition": "one L+U sweep pair = %d algorithmic bytes" % unit_bytes},
            "achieved_gbps": unit_bytes * units_per_step / (ms_per_step * 1e-3) / 1e9,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "measured_copy_gbps": copy_gbps, "frac_of_measured_copy": achieved / copy_gbps,
                         "measured_read_stream_gbps": read_gbps, "frac_of_read_stream": achieved / read_gbps,
                         "kernel": "%s (%s pass; bhip::sweepw_kernel<%d, ...> in the rocprofv3 summaries)" % (
                             {"ilu_apply": "upper triangular sweep z <- D^-1 (y - U z)",
                              "sgs_apply": "backward Gauss-Seidel sweep z <- y - D^-1 U z",
                              "sgs_relax": "relaxation pass x <- D^-1 (b - (A - D) x)", "spmv": "BSR SpMV",
                              "factor": "ILU(0) fixed-point sweep (bhip::factor4_kernel)"}[args.op],
                             "descending" if kernel == "upper" else "ascending", bs),
                         "kernel_ms": kms, "algorithmic_bytes_per_launch": kbytes,
                         "lower_ms": tm["lower_ms"] / max(tm["lower_launches"], 1),
                         "upper_ms": tm["upper_ms"] / max(tm["upper_launches"], 1),
                         "other_ms_per_step": tm["other_ms"] / args.steps},
        }
        if world == 1 and args.op == "ilu_apply":
            # side figure, never `value`: the exact (level-scheduled) solve the sweeps approximate
            def _t(fn, reps=5):
                fn()
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t) / reps * 1e3
            try:
                ex = _t(lambda: p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z))
                st = p.level_stats()
                sy = _t(lambda: p.ilu0_apply(r, s, init=capi.INIT_A_ZERO, mode=capi.JACOBI_SYNC, out=z))
                exf = _t(lambda: p.ilu0_factorize(-1), reps=2)
                asf = _t(lambda: p.ilu0_factorize(args.build_sweeps), reps=2)
                out["exact_apply"] = {"ms": ex, "levels": st["levels"], "syncfree_aborts": st["syncfree_aborts"],
                                      "exact_factor_ms": exf, "async_factor_ms": asf, "sync_sweeps_ms": sy,
                                      "note": "one exact L and U solve (mode LEVEL), and %d+%d SYNCHRONOUS sweeps "
                                              "(deterministic; the first sweep from zero needs no matrix), beside "
                                              "ms_per_step for %d+%d asynchronous sweeps" % (s, s, s, s)}
            except Exception as e:
                out["exact_apply"] = {"ms": None, "note": "failed: %r" % (e,)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.op, args.cpu_sample_n, bs, s, unit_bytes, units_per_step)
            except Exception as e:  # the baseline is a reported side figure, never the measurement
                out["cpu_baseline"] = {"value": None, "unit": "sweeps/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    p.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
