#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric (precond-apply sweeps/s + achieved HBM GB/s) on its configs.

Default (= --config 2, BASELINE.json configs[1], the configuration the metric is quoted on):
  async block-ILU(0) apply, BSR bs=4, 3-D Poisson 256^3 (16.7 M block-rows, 15 GB of factor), 1 GPU.

A "step" is one application of the operator through the C ABI with device pointers; for the ILU apply:
y := 0, `s` asynchronous lower sweeps, z := 0, `s` asynchronous upper sweeps (s = 3 as in the reference's
calibration runs).  value = sweeps (L+U pairs, or relaxation steps) per second over the whole job (all
ranks); every input is resident in HBM before the timed region.  Multi-GPU = independent replicas (the
operator is the subdomain-local preconditioner: the matrix is replicated per GPU, no data-path collective).

--config K selects the other BASELINE.json configurations (numbered 1..5 in the order of its `configs`):
  1  Poisson 64^3 scalar CSR (Chebyshev grid, the reference's own test problem), async ILU(0), 3 sweeps
  2  Poisson 256^3 bs=4, async block-ILU(0) apply (+ factor timing beside it)           [default]
  3  Poisson 256^3 bs=4, async block-SGS relaxation, 5 steps
  4  unstructured bs=5, 126^3 = 2.0 M block-rows (workloads.unstructured_bsr), async block-ILU(0)
  5  Poisson 100^3 pattern bs=8 (1.0 M block-rows), block-ILU(0) apply
Each prints the same JSON line with its own `roofline` (dominant kernel, HIP-event time inside the timed
region, algorithmic bytes from the actual pattern counts) and `cpu_baseline`.

python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1..5] [--n 256] [--bs 4] [--sweeps 3] [--op ilu_apply]

`--gpus N` without a launcher (WORLD_SIZE unset) starts N replica processes itself, one per GPU, before
anything in this process touches the GPU; under torchrun (WORLD_SIZE set) --gpus must equal WORLD_SIZE.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

CONFIGS = {
    1: dict(workload="poisson3d_64_csr_async_ilu0_apply", gen="poisson", n=64, bs=1, grid="chebyshev",
            op="ilu_apply", sweeps=3, build=3, cpu_n=64),
    # cpu_n: BASELINE.md 3's stated fall-back size for the CPU side (192^3); cpu_n_small: used instead when setting the
    # 192^3 sample up would not fit the time budget of a default run (cpu_baseline says which one ran)
    2: dict(workload="poisson3d_256_bs4_async_block_ilu0_apply", gen="poisson", n=256, bs=4, grid="uniform",
            op="ilu_apply", sweeps=3, build=3, cpu_n=192, cpu_n_small=96),
    3: dict(workload="poisson3d_256_bs4_async_block_sgs_relaxation", gen="poisson", n=256, bs=4, grid="uniform",
            op="sgs_relax", sweeps=5, build=3, cpu_n=192, cpu_n_small=96),
    4: dict(workload="unstructured_126_bs5_async_block_ilu0_apply", gen="unstructured", n=126, bs=5, grid="-",
            op="ilu_apply", sweeps=3, build=3, cpu_n=40),
    5: dict(workload="poisson3d_100_bs8_block_ilu0_apply", gen="poisson", n=100, bs=8, grid="uniform",
            op="ilu_apply", sweeps=3, build=3, cpu_n=56),
}


def pattern_bytes(nb, nnzb, nnzl, nnzu, npairs, bs, nfixed=0, nfixed_operands=None):
    """compulsory bytes, every array touched once per sweep, from the pattern's counts.
    nfixed = upper blocks without position pairs (in-place factorisation sweeps leave them alone); nfixed_operands = how
    many of THOSE are read as the u_kj operand of some other block's pair (default: all of them -- on a 7-point pattern
    every upper block is an operand of the diagonal block below it)."""
    if nfixed_operands is None:
        nfixed_operands = nfixed
    B, S, I = 8 * bs * bs, 8 * bs, 4
    lower = nnzl * (B + I) + 2 * nb * I + 3 * nb * S
    upper = (nnzu + nb) * B + nnzu * I + 2 * nb * I + 3 * nb * S
    return {
        "lower_sweep": lower, "upper_sweep": upper, "ilu_pair": lower + upper,
        "sgs_pair": (nnzl + nnzu) * (B + I) + 2 * nb * B + 4 * nb * I + 6 * nb * S,
        "sgs_bwd": nnzu * (B + I) + nb * B + 2 * nb * I + 3 * nb * S,
        "sgs_relax_pass": (nnzl + nnzu) * (B + I) + nb * B + 2 * nb * I + 3 * nb * S,
        "sgs_fwd": nnzl * (B + I) + nb * B + 2 * nb * I + 3 * nb * S,
        "factor_sweep": 3 * nnzb * B + 2 * nnzb * I + 2 * nb * I + 2 * npairs * I,
        # what an in-place sweep TOUCHES once the upper blocks without position pairs hold their value (all sweeps
        # after INIT_F_ORIGINAL, tuning factorskip=1): their matrix block is not read, their factor block neither
        # read nor written; every index array is still walked
        # ... but every fixed block that is the u_kj of a pair is still READ once (perfect-cache rule: each block once per
        # sweep, however many pairs name it); the diagonal blocks the lower updates multiply with are among the blocks
        # already counted
        "factor_sweep_touched": (3 * (nnzb - nfixed) + nfixed_operands) * B + 2 * nnzb * I + 2 * nb * I + 2 * npairs * I,
        # the FIRST sweep of an asynchronous build at bs >= 2 (round 3: the initialisation pass is fused into it): the
        # matrix is both right-hand side and iterate (read once), every factor block is written
        # (+ a row's own finished lower blocks, read back from the factor: FactorArgs::lrow_fresh)
        "factor_sweep_fused_first": (2 * nnzb + nnzl) * B + 2 * nnzb * I + 2 * nb * I + 2 * npairs * I,
        "spmv": nnzb * (B + I) + (nb + 1) * I + 2 * nb * S,
        "nbrows": nb, "nnzb": nnzb, "nnzl": nnzl, "nnzu": nnzu, "pairs": npairs, "fixed_upper": nfixed,
    }


def algorithmic_bytes(n, bs):
    """The N^3 7-point pattern (every bench config but the unstructured one)."""
    nb = n ** 3
    nnzb = 7 * n ** 3 - 6 * n ** 2
    nnzl = 3 * n ** 3 - 3 * n ** 2
    # (a 7-point row's three upper blocks have no position pairs: only diagonal blocks do)
    return pattern_bytes(nb, nnzb, nnzl, nnzl, nnzl, bs, nfixed=nnzl, nfixed_operands=nnzl)


def matrix_counts(m):
    """(nb, nnzb, nnzL, nnzU) of a matrix dict of numpy arrays or torch tensors."""
    nb, nnzb = int(m["nbrows"]), int(m["nnzb"])
    rp, dg = m["browptr"], m["diagind"]
    nnzl = int((dg.long() - rp[:-1].long()).sum().item()) if hasattr(dg, "long") else int((dg.astype("int64") - rp[:-1]).sum())
    return nb, nnzb, nnzl, nnzb - nb - nnzl


def unit_of(op, ab):
    """(bytes of one unit of `value`, bytes of one launch of the dominant kernel, which timing bucket it is in)"""
    return {"ilu_apply": (ab["ilu_pair"], ab["upper_sweep"], "upper"),
            # (ASYNC-mode SGS application = ONE exact forward pass + s backward sweeps: the unit is a backward sweep,
            # the forward pass is accounted per step in main())
            "sgs_apply": (ab["sgs_bwd"], ab["sgs_bwd"], "upper"),
            "sgs_relax": (2 * ab["sgs_relax_pass"], ab["sgs_relax_pass"], "upper"),
            "spmv": (ab["spmv"], ab["spmv"], "lower"),
            "factor": (ab["factor_sweep"], ab["factor_sweep"], "lower")}[op]


def cpu_sample(cfg, op, nsample, sweeps, budget_s, dev=None):
    """One timed sample of the oracle's threaded loop nest: (seconds per call, calls, unit bytes, description, set-up s).
    Large Poisson samples are GENERATED on the GPU when there is one (the numpy generator and the oracle's serial
    factorisation take minutes at 192^3) -- matrix by workloads.poisson3d_device, the factor the sweeps are timed on by
    the library's exact factorisation, both copied to the host; what is TIMED is the oracle alone."""
    import oracle
    from blasted_amd import workloads
    t_setup = time.perf_counter()
    bs = cfg["bs"]
    on_device = dev is not None and cfg["gen"] != "unstructured" and nsample >= 128
    iluvals = None
    if cfg["gen"] == "unstructured":
        m = workloads.to_numpy(workloads.unstructured_bsr(nsample, bs, device="cpu"))
        what = "unstructured %d^3" % nsample
    elif on_device:
        from blasted_amd import capi
        md = workloads.poisson3d_device(nsample, bs, dev, grid=cfg["grid"])
        if op == "ilu_apply":
            pg = capi.Prec(dev.index or 0)
            pg.set_matrix(md)
            pg.ilu0_factorize(-1)
            iluvals = pg.get_iluvals()
            pg.close()
        m = workloads.to_numpy(md)
        del md
        what = "Poisson %d^3 (%s grid; matrix and factor made on the GPU, copied to the host)" % (nsample, cfg["grid"])
    else:
        m = workloads.poisson3d(nsample + 2, bs, grid=cfg["grid"])
        what = "Poisson %d^3 (%s grid)" % (nsample, cfg["grid"])
    r = workloads.rhs_vector(m["nbrows"] * bs)
    kw = {}
    plist = oracle.ilu_positions(m)
    if op == "ilu_apply":
        kw["iluvals"] = iluvals if iluvals is not None else oracle.ilu0_factorize(m, plist, 1, mode=oracle.GS_SERIAL)["iluvals"]
    elif op == "factor":
        kw["plist"] = plist
    elif op != "spmv":
        kw["dblocks"] = oracle.jacobi_compute(m)
    t_setup = time.perf_counter() - t_setup
    t1 = oracle.time_op(op, m, r, sweeps, 256, 2, **kw)
    reps = max(3, min(200, int(budget_s / max(t1, 1e-4))))
    t = oracle.time_op(op, m, r, sweeps, 256, reps, **kw)
    nb, nnzb, nnzl, nnzu = matrix_counts(m)
    sample_unit = unit_of(op, pattern_bytes(nb, nnzb, nnzl, nnzu, int(plist[1].size), bs))[0]
    return t, reps, sample_unit, what, t_setup


def cpu_baseline(cfg, op, sweeps, full_unit_bytes, units_per_call, budget_s=12.0, setup_budget_s=150.0, dev=None):
    """The oracle's threaded port of the reference loop nest (omp for schedule(dynamic,256) nowait), timed on this
    box's host cores on a bounded sample of the same workload (a smaller grid of the same generator), scaled to the
    metric's unit by algorithmic bytes.  Configs 2 / 3: BASELINE.md 3's 192^3 when its set-up (numpy generator, position
    lists, one serial factorisation) fits `setup_budget_s` -- predicted from the 96^3 sample, which always runs first --
    otherwise the 96^3 sample stands."""
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    import oracle
    # one thread per CPU this process is really granted (affinity mask cut by the cgroup quota): more
    # threads than that only thrash; the count is what `cores` reports
    budget = oracle.cpu_budget()
    oracle.set_num_threads(budget)
    bs = cfg["bs"]
    small = cfg.get("cpu_n_small")
    note = ""
    if small and small < cfg["cpu_n"]:
        t, reps, sample_unit, what, t_setup = cpu_sample(cfg, op, small, sweeps, min(budget_s, 4.0))
        # (on the GPU the large sample's set-up is the position lists and two copies: a tenth of the host generator's)
        predicted = t_setup * (cfg["cpu_n"] / small) ** 3 * (0.1 if dev is not None else 1.0)
        if predicted <= setup_budget_s:
            t96 = (t, what)
            t, reps, sample_unit, what, t_setup2 = cpu_sample(cfg, op, cfg["cpu_n"], sweeps, budget_s, dev)
            note = "; set-up %.0f s (predicted %.0f s from the %s sample, which ran first: %.1f ms per call)" % (
                t_setup2, predicted, t96[1], t96[0] * 1e3)
        else:
            note = "; the %d^3 sample was NOT run: its set-up was predicted at %.0f s (budget %.0f s)" % (
                cfg["cpu_n"], predicted, setup_budget_s)
    else:
        t, reps, sample_unit, what, _ = cpu_sample(cfg, op, cfg["cpu_n"], sweeps, budget_s)
    return {
        "value": (units_per_call / t) * sample_unit / full_unit_bytes,
        "unit": "sweeps/s", "cores": oracle.num_threads(), "kind": "port",
        "achieved_gbps": sample_unit * units_per_call / t / 1e9,
        "sample": "oracle ASYNC_OMP (reference loop nest, chunk 256) %s on %s bs=%d, %d sweeps per call, "
                  "min of %d calls = %.1f ms, %d OpenMP threads = the CPUs granted to this process (%d hardware "
                  "threads visible); scaled to the full size by algorithmic bytes%s" %
                  (op, what, bs, sweeps, reps, t * 1e3, oracle.num_threads(), os.cpu_count() or 0, note),
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


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_replicas(ngpus, argv):
    """`python bench.py --gpus N` without a launcher: start N copies of this script, one per GPU, with the
    rendezvous variables torchrun would set.  Runs before this process has made any HIP / torch.cuda call
    (a process that has initialised the GPU must never be replaced or forked on this pool); the parent only
    waits, forwards rank 0's JSON line and returns the worst exit code."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for rank in range(ngpus):
        env = dict(os.environ, WORLD_SIZE=str(ngpus), RANK=str(rank), LOCAL_RANK=str(rank),
                   LOCAL_WORLD_SIZE=str(ngpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        out = subprocess.PIPE if rank == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    text = procs[0].communicate()[0].decode()
    rc = procs[0].returncode
    for pr in procs[1:]:
        rc = pr.wait() or rc
    sys.stdout.write(text)
    sys.stdout.flush()
    return rc


def quality_figures(p, capi, torch, r, z, s):
    """Side figure, never `value`: what the s+s asynchronous sweeps are worth as a preconditioner -- the
    relative distance of z to the exact triangular solves after s+s and 10+10 sweeps, the contraction per
    sweep between them, and from it the sweeps (and milliseconds) to reach 1e-2 and 1e-6."""
    import math
    ze = p.ilu0_apply(r, 1, mode=capi.LEVEL, out=torch.empty_like(z)).clone()
    nz = float(torch.linalg.vector_norm(ze))

    def dist_after(k):
        p.ilu0_apply(r, k, init=capi.INIT_A_ZERO, mode=capi.ASYNC, out=z)
        return float(torch.linalg.vector_norm(z - ze)) / nz

    def ms_of(k, reps=3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            p.ilu0_apply(r, k, init=capi.INIT_A_ZERO, mode=capi.ASYNC, out=z)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3
    d_s, d_10 = dist_after(s), dist_after(10)
    out = {"distance_after_%d+%d" % (s, s): d_s, "distance_after_10+10": d_10}
    if 0 < d_10 < d_s and s < 10:
        rho = (d_10 / d_s) ** (1.0 / (10 - s))
        per = ms_of(s) / s
        out["contraction_per_sweep"] = rho
        for tol, key in ((1e-2, "1e-2"), (1e-6, "1e-6")):
            k = max(s, s + math.ceil(math.log(tol / d_s) / math.log(rho)))
            out["sweeps_to_" + key] = k
            out["ms_to_" + key] = k * per
    return out


def product_default_block(m, r, z, cfg, s, dev_index, stream, capi, torch, ab):
    """What a caller gets WITHOUT any of this benchmark's settings: compact copies made when they pay (the 16th
    application since a factorisation at bs = 4: until then the sweeps read the factor in place) and the quick form of the
    class-aware placement.  An operator of its own, built and dropped after the measured one's timed region."""
    capi.set_tuning("compactafter=-1")
    capi.set_tuning("placement=1")
    capi.set_tuning("placeafter=%s" % os.environ.get("BLASTED_HIP_PLACE_AFTER", "256"))
    p0 = capi.Prec(dev_index, stream)
    p0.set_matrix(m)
    p0.ilu0_factorize(cfg["build"], init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)

    def timed(n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            p0.ilu0_apply(r, s, init=capi.INIT_A_ZERO, mode=capi.ASYNC, out=z)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3
    p0.ilu0_apply(r, s, init=capi.INIT_A_ZERO, mode=capi.ASYNC, out=z)     # application 1 (first-touch effects)
    in_place = timed(10)                                                     # applications 2 .. 11
    n_before = 1 + 10
    copies0 = p0.memory_stats()["derived_copies"]
    switch_ms, guard = None, 0
    while p0.memory_stats()["derived_copies"] == copies0 and guard < 40:     # ... until the copies are made
        switch_ms = timed(1)
        n_before += 1
        guard += 1
    n_copy = n_before
    placed0 = capi.placement_stats()["probes"]   # (the placing application looks at the plain copies first and may keep them)
    timed(3)
    plain = timed(10)                                                        # plain compact copies
    n_apps = n_copy + 3 + 10
    # the default places the copies (quick search) once the operator has been applied 256 times in its life
    place_ms, n_place, guard = None, None, 0
    while capi.placement_stats()["probes"] == placed0 and guard < 400:
        place_ms = timed(1)
        n_apps += 1
        guard += 1
    if capi.placement_stats()["probes"] != placed0:
        n_place = n_apps
    timed(3)
    p0.set_timing(True)
    p0.get_timing(reset=True)
    steady = timed(10)
    tm = p0.get_timing(reset=True)
    p0.set_timing(False)
    where = p0.placement_check(r, z)
    lo = tm["lower_ms"] / max(tm["lower_launches"], 1)
    up = tm["upper_ms"] / max(tm["upper_launches"], 1)
    p0.close()
    torch.cuda.synchronize()
    return {"in_place_apply_ms": in_place, "in_place_value": s / (in_place * 1e-3),
            "copies_made_with_application": n_copy, "that_application_ms": switch_ms,
            "plain_copies_apply_ms": plain, "plain_copies_value": s / (plain * 1e-3),
            "copies_placed_with_application": n_place, "placing_application_ms": place_ms if n_place else None,
            "steady_apply_ms": steady, "steady_value": s / (steady * 1e-3), "unit": "sweeps/s",
            "lower_ms": lo, "upper_ms": up, "upper_frac": ab["upper_sweep"] / (up * 1e-3) / 1e9 / HBM_PEAK_GBS if up > 0 else 0.0,
            "placement": "quick (default)", "where": where,
            "note": "product defaults (BLASTED_HIP_COMPACT_AFTER / BLASTED_HIP_PLACE_AFTER / BLASTED_HIP_PLACEMENT unset): "
                    "applications 2-11 read the factor in place; the 17th makes plain compact copies (that_application_ms: the "
                    "copy pass); once the operator has been applied 256 times in its life the quick placement search makes "
                    "placed copies beside them (placing_application_ms); steady_* is the rate after that.  `value` of this "
                    "line is the steady state of an operator whose copies were placed by the thorough search (config.placement)"}


def fixed_upper_blocks(p, m, gen):
    """Upper blocks without position pairs (what factorskip leaves alone).  A 7-point row: all its upper blocks."""
    nb, nnzb, nnzl, nnzu = matrix_counts(m)
    if gen != "unstructured":
        return nnzu, nnzu
    import numpy as np
    posptr, _, upperp = p.ilu0_positions()
    rp = m["browptr"].cpu().numpy().astype(np.int64)
    col = m["bcolind"].cpu().numpy()
    rowof = np.repeat(np.arange(nb, dtype=np.int64), rp[1:] - rp[:-1])
    fixed = (col > rowof) & (posptr[1:] == posptr[:-1])
    operand = np.zeros(nnzb, dtype=bool)
    operand[upperp] = True   # blocks that are the u_kj of some pair
    return int(fixed.sum()), int((fixed & operand).sum())


def short_run(k, dev, stream, capi, workloads, torch, steps=5, warmup=2, reuse=None):
    """A short measurement of BASELINE configuration k for the default run's `other_configs` block: the same
    timed-region protocol as main() (operator applied `steps` times after `warmup`, HIP events of the library
    around every sweep inside the region), reduced to value / kernel time / roofline fraction."""
    cfg = CONFIGS[k]
    n, bs, s, op = cfg["n"], cfg["bs"], cfg["sweeps"], cfg["op"]
    t_setup = time.perf_counter()
    if reuse is not None:
        p, m, r, z = reuse
    else:
        if cfg["gen"] == "unstructured":
            m = workloads.unstructured_bsr(n, bs, device=dev)
        else:
            m = workloads.poisson3d_device(n, bs, dev, grid=cfg["grid"])
        r = workloads.rhs_vector_device(m["nbrows"] * bs, dev)
        z = torch.zeros_like(r)
        p = capi.Prec(dev.index or 0, stream)
        p.set_matrix(m)
    nb, nnzb, nnzl, nnzu = matrix_counts(m)
    npairs = nnzl
    build_ms = None
    if op == "ilu_apply":
        p.ilu0_factorize(cfg["build"], init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
        npairs = p.ilu0_positions_size()
        # the build itself, once more (the first call also analysed the pattern): cfg["build"] asynchronous sweeps
        torch.cuda.synchronize()
        tb = time.perf_counter()
        p.ilu0_factorize(cfg["build"], init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
        torch.cuda.synchronize()
        build_ms = (time.perf_counter() - tb) * 1e3
    else:
        p.jacobi_compute()
    ab = pattern_bytes(nb, nnzb, nnzl, nnzu, npairs, bs)
    step = {"ilu_apply": lambda: p.ilu0_apply(r, s, init=capi.INIT_A_ZERO, mode=capi.ASYNC, out=z),
            "sgs_relax": lambda: p.sgs_relax(r, z, s, mode=capi.ASYNC)}[op]
    unit_bytes, kbytes, kernel = unit_of(op, ab)
    if op == "sgs_relax":
        warmup = max(warmup, 3)   # (thorough placement makes the relaxation's matrix copy once eight passes have run)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    # a launch-bound size (config 1: tens of microseconds per step) is not measured by five steps: 200 of them
    t1 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    if time.perf_counter() - t1 < 0.5e-3:
        steps, warmup = max(steps, 200), warmup + 21
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    p.set_timing(True)
    p.get_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tm = p.get_timing(reset=True)
    p.set_timing(False)
    kms = tm[kernel + "_ms"] / max(tm[kernel + "_launches"], 1)
    ach = kbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    plain = None
    if el / steps < 0.5e-3:
        plain = plain_rate(step, steps, s, torch)
    if reuse is None:
        p.close()
    return {"baseline_config": k, "without_event_instrumentation": plain, "workload": cfg["workload"], "value": s * steps / el, "unit": "sweeps/s", "steps": steps,
            "warmup": warmup, "ms_per_step": el / steps * 1e3, "napplysweeps": s, "nbrows": nb, "block_size": bs,
            "build_ms": build_ms, "nbuildsweeps": cfg["build"] if op == "ilu_apply" else None,
            "achieved_gbps": unit_bytes * s / (el / steps) / 1e9,
            "roofline": {"bound": "hbm", "kernel_ms": kms, "algorithmic_bytes_per_launch": kbytes, "achieved": ach,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "lower_ms": tm["lower_ms"] / max(tm["lower_launches"], 1),
                         "upper_ms": tm["upper_ms"] / max(tm["upper_launches"], 1)},
            "setup_s": t_setup}


def plain_rate(step, steps, units_per_step, torch):
    """Launch-bound sizes only: the same K steps once more WITHOUT the per-phase HIP events of the roofline measurement.
    On a 64^3 scalar problem a sweep is a 6 us kernel and the eight event records of an application cost the host
    about as much as its eight launches, so the instrumented `value` understates what a caller gets."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"value": units_per_step * steps / el, "unit": "sweeps/s", "ms_per_step": el / steps * 1e3,
            "note": "same steps, timing events off (blasted_hip_set_timing(0)): host launch cost of the events removed"}


def reference_side_quality(cfg, capi, workloads, torch, dev, s):
    """SURVEY 8(d) tier P4 beside the GPU figure, on the cpu_baseline sample (same generator, smaller grid): what
    s+s and 10+10 sweeps of the REFERENCE loop nest (oracle ASYNC_OMP, this box's host cores, chunk 256) leave of
    the distance to the exact triangular solves, against HIP ASYNC on the same matrix."""
    import numpy as np
    import oracle
    oracle.set_num_threads(oracle.cpu_budget())
    nq = cfg.get("cpu_n_small", cfg["cpu_n"])
    m = workloads.poisson3d(nq + 2, cfg["bs"], grid=cfg["grid"])
    r = workloads.rhs_vector(m["nbrows"] * cfg["bs"])
    f = oracle.ilu0_factorize(m, None, 1, mode=oracle.GS_SERIAL)["iluvals"]
    ze = oracle.ilu0_apply(m, f, r, 1, mode=oracle.GS_SERIAL)
    nz = float(np.linalg.norm(ze))
    dist = lambda z: float(np.linalg.norm(z - ze)) / nz
    ref = {k: dist(oracle.ilu0_apply(m, f, r, k, mode=oracle.ASYNC_OMP, init=oracle.INIT_A_ZERO, chunk=256)) for k in (s, 10)}
    p = capi.Prec(dev.index or 0)
    p.set_matrix(m)
    p.ilu0_factorize(-1)
    rd = torch.from_numpy(r).to(dev)
    hip = {k: dist(p.ilu0_apply(rd, k, init=capi.INIT_A_ZERO, mode=capi.ASYNC).cpu().numpy()) for k in (s, 10)}
    p.close()

    def rho(d):
        return (d[10] / d[s]) ** (1.0 / (10 - s)) if 0 < d[10] < d[s] and s < 10 and d[10] > 1e-14 else None
    return {"sample": "Poisson %d^3 bs=%d" % (nq, cfg["bs"]),
            "threads": oracle.num_threads(),
            "reference_loop_nest": {"distance_after_%d+%d" % (s, s): ref[s], "distance_after_10+10": ref[10],
                                    "contraction_per_sweep": rho(ref)},
            "hip_async": {"distance_after_%d+%d" % (s, s): hip[s], "distance_after_10+10": hip[10],
                          "contraction_per_sweep": rho(hip)},
            "note": "the reference's threaded sweep in natural order is nearly sequential Gauss-Seidel (few threads, "
                    "chunks of 256 consecutive rows); thousands of concurrent waves make the GPU's in-place sweep "
                    "Jacobi-like over the rows in flight: profiles/r03_async_vs_reference.txt"}


# ---- live HBM traffic of the dominant kernel (roofline.traffic) ----------------------------------------------------
# PMC counters can only be read from outside the process: before this process touches the GPU, rank 0 of a one-GPU
# run starts itself twice under `rocprofv3 --pmc` (one counter per pass, never combined with a trace; MI355X_MICROARCH
# guide, HBM section: bytes = 2 x FETCH_SIZE + WRITE_SIZE in KB on gfx950) with 2 steps of the same configuration and
# reads the dominant kernel's average per launch from the counter CSV.  Failure of any kind (no rocprofv3, a pass that
# times out, an unexpected CSV) falls back to the committed record in profiles/traffic.json, and says so.
PART_ARG = {"sweepw_kernel": 1, "sweepodd_kernel": 1, "sweepwr_kernel": 1, "sweep_kernel": 2, "sweep1_kernel": 0,
            "sweep1s_kernel": 0}
PART_OF_OP = {"ilu_apply": "1", "sgs_apply": "1", "sgs_relax": "2", "spmv": "3"}


def dominant_kernel_counter(csv_path, op):
    """(kernel name, average counter value per launch, launches) of the kernel roofline is quoted on."""
    import collections
    import csv
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(csv_path)):
        name = row["Kernel_Name"]
        if "bhip::" not in name:
            continue
        if op == "factor":
            if "factor" not in name or "kernel" not in name or "plan" in name or "fill" in name:
                continue
        else:
            fam = next((f for f in PART_ARG if ("::" + f + "<") in name), None)
            if fam is None or "<" not in name:
                continue
            targs = [t.strip() for t in name.split(fam + "<", 1)[1].split(">")[0].split(",")]
            if len(targs) <= PART_ARG[fam] or targs[PART_ARG[fam]] != PART_OF_OP[op]:
                continue
        agg[name].append(float(row["Counter_Value"]))
    if not agg:
        return None
    name = max(agg, key=lambda k: (len(agg[k]), sum(agg[k])))
    return name, sum(agg[name]) / len(agg[name]), len(agg[name])


def live_traffic(argv, op, budget_s=330.0):
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None, "rocprofv3 not found"
    t0 = time.perf_counter()
    keep = [a for a in argv if a not in ("--no-cpu-baseline", "--no-other-configs")]
    child = []
    skip = False
    for a in keep:  # drop --steps / --warmup / --live-traffic and their values
        if skip:
            skip = False
            continue
        if a in ("--steps", "--warmup", "--live-traffic", "--placement"):
            skip = True
            continue
        if a.startswith(("--steps=", "--warmup=", "--live-traffic=", "--placement=")):
            continue
        child.append(a)
    cmd_tail = [sys.executable, os.path.abspath(__file__)] + child + [
        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--live-traffic", "off",
        # (the bytes a kernel moves do not depend on where its buffers lie: no placement search in the counter passes)
        "--placement", "0", "--no-product-default"]
    env = dict(os.environ, TMPDIR="/tmp", BLASTED_BENCH_PMC_CHILD="1")
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        left = budget_s - (time.perf_counter() - t0)
        if left < 30:
            return None, "time budget spent before the %s pass" % counter
        d = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        try:
            # (own session: on a timeout the whole group goes, not only the profiler's launcher)
            proc = subprocess.Popen([rp, "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + cmd_tail, cwd="/tmp",
                                    env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=min(left, 200.0))
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                return None, "%s pass timed out" % counter
            if rc != 0:
                return None, "%s pass exited with %d" % (counter, rc)
            files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
            if not files:
                return None, "%s pass wrote no counter file" % counter
            dom = dominant_kernel_counter(files[0], op)
            if dom is None:
                return None, "no kernel of the %s family in the %s pass" % (op, counter)
            got[counter] = dom
        except Exception as e:  # noqa: BLE001 -- any failure means: use the committed record
            return None, "%s pass failed: %s" % (counter, type(e).__name__)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if got["FETCH_SIZE"][0] != got["WRITE_SIZE"][0]:
        return None, "the two passes disagree on the dominant kernel"
    nbytes = (2.0 * got["FETCH_SIZE"][1] + got["WRITE_SIZE"][1]) * 1024.0
    return {"hbm_bytes_per_launch": nbytes, "kernel": got["FETCH_SIZE"][0], "launches_per_pass": got["FETCH_SIZE"][2],
            "FETCH_SIZE_KB_avg": got["FETCH_SIZE"][1], "WRITE_SIZE_KB_avg": got["WRITE_SIZE"][1],
            "seconds": time.perf_counter() - t0}, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE.json configuration, numbered 1..5 (2 = the one the metric is quoted on)")
    ap.add_argument("--n", type=int, default=None, help="grid points per axis (overrides the config's)")
    ap.add_argument("--bs", type=int, default=None)
    ap.add_argument("--sweeps", type=int, default=None, help="napplysweeps / relaxation steps")
    ap.add_argument("--build-sweeps", type=int, default=None)
    ap.add_argument("--op", default=None, choices=["ilu_apply", "sgs_apply", "sgs_relax", "spmv", "factor"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run only: skip the short measurements of configurations 3, 4, 5, 1")
    ap.add_argument("--cpu-sample-n", type=int, default=None)
    ap.add_argument("--placement", default="2", choices=["0", "1", "2"],
                    help="class-aware placement of the measured operator's triangle copies: 2 thorough (default here), "
                         "1 quick (the product default), 0 off")
    ap.add_argument("--no-product-default", action="store_true",
                    help="skip the product_default block (an extra operator with the product's lazy copies and quick placement)")
    ap.add_argument("--live-traffic", default="auto", choices=["auto", "on", "off"],
                    help="roofline.traffic from two rocprofv3 --pmc passes of this command, started before the timed run "
                         "(auto: one-GPU runs when rocprofv3 is there; off: the committed record of profiles/traffic.json)")
    args = ap.parse_args()

    dry = os.environ.get("BLASTED_BENCH_DRYRUN") == "1"  # tests of the launcher: gloo, no GPU, no kernels
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            raise SystemExit(launch_replicas(args.gpus, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d does not match the launcher's WORLD_SIZE=%s" %
                         (args.gpus, os.environ["WORLD_SIZE"]))

    cfg = dict(CONFIGS[args.config])
    custom = False
    for key, val in (("n", args.n), ("bs", args.bs), ("sweeps", args.sweeps), ("build", args.build_sweeps),
                     ("op", args.op), ("cpu_n", args.cpu_sample_n)):
        if val is not None and val != cfg[key]:
            cfg[key] = val
            custom = custom or key != "cpu_n"
    n, bs, s, op = cfg["n"], cfg["bs"], cfg["sweeps"], cfg["op"]
    if custom:
        kind = {"ilu_apply": "async_ilu0_apply", "sgs_apply": "async_sgs_apply", "sgs_relax": "async_sgs_relaxation",
                "spmv": "spmv", "factor": "async_ilu0_factor"}[op]
        cfg["workload"] = "%s_%d_bs%d_%s" % ("unstructured" if cfg["gen"] == "unstructured" else "poisson3d", n, bs, kind)

    live, live_note, live_others, live_factor = None, "not asked for", {}, None
    under_profiler = any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_LIBRARY"))
    if under_profiler:
        live_note = "this process runs under a profiler itself"
    if (args.live_traffic != "off" and not dry and args.gpus == 1 and "WORLD_SIZE" not in os.environ
            and os.environ.get("BLASTED_BENCH_PMC_CHILD") != "1" and (args.live_traffic == "on" or not under_profiler)):
        # (before anything here touches the GPU: the passes are child processes with the device to themselves)
        live, live_note = live_traffic(sys.argv[1:], op)
        if live is not None and op == "ilu_apply":
            # the factorisation sweep timed beside the apply: the same two passes of this configuration with --op factor
            live_factor, _ = live_traffic([a for a in sys.argv[1:]] + ["--op", "factor"], "factor", budget_s=90.0)
        if live is not None and args.config == 2 and not custom and not args.no_other_configs:
            # the short runs of the other configurations get their dominant kernel's traffic the same way
            t_live = time.perf_counter()
            for k in (3, 4, 5, 1):
                if time.perf_counter() - t_live > 150.0:
                    break
                lk, _ = live_traffic(["--config", str(k)], CONFIGS[k]["op"], budget_s=90.0)
                if lk is not None:
                    live_others[k] = lk

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if dry:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
        dev = torch.device("cpu")
        sync = lambda: None
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync = torch.cuda.synchronize

    def barrier():
        if world > 1:
            dist.barrier()

    p = None
    if dry:
        ab = algorithmic_bytes(n, bs)
        step = lambda: time.sleep(0.002)
    else:
        from blasted_amd import capi, workloads
        # the steady state of a long solve: the compact triangle copies made with the first application (the product
        # makes them once they pay -- the 16th application since a factorisation at bs = 4 -- which would fall into
        # the timed region here)
        capi.set_tuning("compactafter=0")
        # ---- workload resident in HBM
        if cfg["gen"] == "unstructured":
            m = workloads.unstructured_bsr(n, bs, device=dev)
        else:
            m = workloads.poisson3d_device(n, bs, dev, grid=cfg["grid"])
        r = workloads.rhs_vector_device(m["nbrows"] * bs, dev)
        z = torch.zeros_like(r)
        sync()
        stream = torch.cuda.current_stream().cuda_stream
        product_default = None
        # (the product-default figures come from a SECOND operator with nothing set, made after the timed region below:
        # the measured operator is the first thing this process allocates after the matrix and the vectors)
        want_product_default = (op == "ilu_apply" and world == 1 and os.environ.get("BLASTED_BENCH_PMC_CHILD") != "1"
                                and not args.no_product_default)
        # the measured operator: the steady state of a long solve (copies made at once) with its triangle copies placed by
        # the THOROUGH search (every piece in the right address class; a one-time cost of 0.1 ... several seconds that the
        # quick default search does not spend -- DESIGN.md, address classes)
        capi.set_tuning("compactafter=0")
        capi.set_tuning("placement=%s" % args.placement)
        capi.set_tuning("placeafter=0")   # (the measured operator's copies are placed at once, by whichever search was asked for)
        p = capi.Prec(local_rank, stream)
        p.set_matrix(m)
        nb, nnzb, nnzl, nnzu = matrix_counts(m)
        npairs = nnzl
        nfixed, nfixed_ops = 0, 0
        if op in ("ilu_apply", "factor"):
            p.ilu0_factorize(cfg["build"], init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
            npairs = p.ilu0_positions_size()
            nfixed, nfixed_ops = fixed_upper_blocks(p, m, cfg["gen"])
        else:
            p.jacobi_compute()
        ab = pattern_bytes(nb, nnzb, nnzl, nnzu, npairs, bs, nfixed, nfixed_ops)
        sync()
        step = {
            "ilu_apply": lambda: p.ilu0_apply(r, s, init=capi.INIT_A_ZERO, mode=capi.ASYNC, out=z),
            "sgs_apply": lambda: p.sgs_apply(r, s, init=capi.INIT_A_ZERO, mode=capi.ASYNC, out=z),
            "sgs_relax": lambda: p.sgs_relax(r, z, s, mode=capi.ASYNC),
            "spmv": lambda: p.spmv(r, out=z),
            "factor": lambda: p.ilu0_factorize(s, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC),
        }[op]
    unit_bytes, kbytes, kernel = unit_of(op, ab)
    units_per_step = 1 if op == "spmv" else s
    # bytes one step moves by the algorithmic count (the ASYNC-mode SGS application is one exact forward pass
    # followed by s backward sweeps; the in-place factorisation sweeps leave the fixed upper blocks alone)
    step_bytes = unit_bytes * units_per_step
    if op == "sgs_apply":
        step_bytes = ab["sgs_fwd"] + s * ab["sgs_bwd"]
    if op == "factor":
        kbytes = ab["factor_sweep_touched"]
        if bs >= 2 and s >= 1:  # a build = one fused first sweep + in-place sweeps (see pattern_bytes)
            kbytes = (ab["factor_sweep_fused_first"] + (s - 1) * ab["factor_sweep_touched"]) / s

    for _ in range(args.warmup):
        step()
    sync()
    if p:
        p.set_timing(True)
        p.get_timing(reset=True)
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    barrier()
    t1 = time.perf_counter()
    tm = None
    if p:
        tm = p.get_timing(reset=True)
        p.set_timing(False)

    elapsed = max_over_ranks(t1 - t0, dev)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = job_throughput(world, units_per_step, args.steps, elapsed)
        out = {
            "metric": "precond_apply_sweeps_per_sec", "value": value, "unit": "sweeps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "dry-run (launcher test: no kernels ran)" if dry else "synthetic",
            "config": {"workload": cfg["workload"], "baseline_config": None if custom else args.config,
                       "generator": cfg["gen"], "grid_points": n, "block_size": bs, "nbrows": ab["nbrows"],
                       "nnzb": ab["nnzb"], "nnz_lower": ab["nnzl"], "nnz_upper": ab["nnzu"],
                       "napplysweeps": s, "nbuildsweeps": cfg["build"], "sweep_mode": "async",
                       "row_order_in_chunk": "natural (default)", "grid": cfg["grid"],
                       "compact_copies": "made with the first application (steady state; product default: once they pay, "
                                         "the 16th application since a factorisation at bs=4 -- see product_default)",
                       "placement": {"0": "off", "1": "quick (the product default)",
                                     "2": "thorough (BLASTED_HIP_PLACEMENT=2; the product default is the quick search -- "
                                          "see product_default)"}[getattr(args, "placement", "2")],
                       "replicas": world,
                       "unit_definition": "one %s = %d algorithmic bytes" % (
                           {"ilu_apply": "L+U sweep pair",
                            "sgs_apply": "backward sweep (every step also runs ONE exact forward pass, as the reference "
                                         "does: %d bytes, counted in achieved_gbps)" % ab["sgs_fwd"],
                            "sgs_relax": "relaxation step (ascending + descending pass)", "spmv": "product",
                            "factor": "factorisation sweep"}[op], unit_bytes)},
            "achieved_gbps": step_bytes * world / (ms_per_step * 1e-3) / 1e9,
        }
        if not dry:
            copy_gbps = measured_copy_gbps(dev)
            # practical read ceiling of this device: the matrix's own value array through a read-only kernel
            read_gbps = capi.measure_read_stream(m["vals"], reps=10)
            kms = tm[kernel + "_ms"] / max(tm[kernel + "_launches"], 1)
            achieved = kbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
            # HBM bytes per launch of the dominant kernel from the PMC passes of this configuration (separate
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this command, tools/profile_round.sh): a committed
            # record, not a measurement of this run -- traffic_source says which passes and which commit
            traffic, traffic_source = None, None
            tf = os.path.join(ROOT, "profiles", "traffic.json")
            if live is not None:
                traffic = live["hbm_bytes_per_launch"]
                traffic_source = {"live": True, "how": "two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this command "
                                  "with 2 steps, run by this process before its timed region on the same device; "
                                  "bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB, averaged over the kernel's launches",
                                  "kernel": live["kernel"], "launches_per_pass": live["launches_per_pass"],
                                  "FETCH_SIZE_KB_avg": live["FETCH_SIZE_KB_avg"], "WRITE_SIZE_KB_avg": live["WRITE_SIZE_KB_avg"],
                                  "seconds": round(live["seconds"], 1)}
            elif os.path.exists(tf):
                try:
                    tj = json.load(open(tf))
                    ent = tj.get(cfg["workload"]) or (tj.get(op) if args.config == 2 and not custom else None) or {}
                    traffic = ent.get("hbm_bytes_per_launch")
                    if traffic is not None:
                        traffic_source = {"live": False, "why_not_live": live_note, "profiles": ent.get("from"),
                                          "commit": ent.get("commit"), "kernel": ent.get("kernel")}
                except Exception:
                    traffic, traffic_source = None, None
            family = ("sweepw_kernel<%d, ...>" % bs if bs in (4, 8) else
                      "sweepodd_kernel<%d, ...>" % bs if bs in (3, 5, 7) else
                      # scalar rows: the product takes the LDS-staged form; in-place sweeps the general kernel
                      "sweep1s_kernel<...>" if (bs == 1 and op == "spmv") else "sweep_kernel<%d, ...>" % bs)
            # a 64^3 scalar problem is 22 MB: it lives in the L2s / the 256 MB Infinity Cache and a sweep is an
            # 8 us launch -- the step is bound by launch latency and cache bandwidth, not by HBM
            cache_resident = ab["ilu_pair"] < 128e6
            out["roofline"] = {
                "bound": "launch/L2" if cache_resident else "hbm",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "launches_per_step": (tm["lower_launches"] + tm["upper_launches"] + tm["other_launches"]) / args.steps,
                "measured_copy_gbps": copy_gbps, "frac_of_measured_copy": achieved / copy_gbps,
                "measured_read_stream_gbps": read_gbps, "frac_of_read_stream": achieved / read_gbps,
                "kernel": "%s (%s pass; bhip::%s in the rocprofv3 summaries)" % (
                    {"ilu_apply": "upper triangular sweep z <- D^-1 (y - U z)",
                     "sgs_apply": "backward Gauss-Seidel sweep z <- y - D^-1 U z",
                     "sgs_relax": "relaxation pass x <- D^-1 (b - (A - D) x)", "spmv": "BSR SpMV",
                     "factor": "ILU(0) fixed-point sweep"}[op],
                    "descending" if kernel == "upper" else "ascending",
                    "factor kernel" if op == "factor" else family),
                "kernel_ms": kms, "algorithmic_bytes_per_launch": kbytes,
                "lower_ms": tm["lower_ms"] / max(tm["lower_launches"], 1),
                "upper_ms": tm["upper_ms"] / max(tm["upper_launches"], 1),
                "other_ms_per_step": tm["other_ms"] / args.steps}
            out["placement"] = dict(capi.placement_stats(), note="class-aware placement of the triangle copies "
                                    "(blasted_hip_placement_stats; DESIGN.md, address classes): 1 GiB pieces checked with a "
                                    "read-beside-write probe against the vectors the sweeps read and write")
            if op == "ilu_apply":
                out["placement"]["where"] = p.placement_check(r, z)
            if want_product_default:
                try:
                    nb0, nnzb0, nnzl0, nnzu0 = matrix_counts(m)
                    product_default = product_default_block(m, r, z, cfg, s, local_rank, stream, capi, torch,
                                                            pattern_bytes(nb0, nnzb0, nnzl0, nnzu0, nnzl0, bs))
                except Exception as e:  # a side figure
                    product_default = {"failed": repr(e)}
                finally:  # back to this run's settings for what follows
                    capi.set_tuning("compactafter=0")
                    capi.set_tuning("placement=%s" % args.placement)
                    capi.set_tuning("placeafter=0")
            if product_default is not None:
                out["product_default"] = product_default
            if cache_resident:
                out["roofline"]["note"] = ("working set %.0f MB: cache-resident, launch-latency bound (%.1f us per step "
                                           "over %d launches); frac is against the HBM peak only for uniformity" % (
                                               ab["ilu_pair"] / 1e6, ms_per_step * 1e3,
                                               out["roofline"]["launches_per_step"]))
                if world == 1:
                    out["without_event_instrumentation"] = plain_rate(step, args.steps, units_per_step, torch)

        def _t(fn, reps=5):
            fn()
            sync()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            sync()
            return (time.perf_counter() - t) / reps * 1e3
        pmc_child = os.environ.get("BLASTED_BENCH_PMC_CHILD") == "1"  # a counter pass of live_traffic(): the timed region only
        if not dry and world == 1 and op == "ilu_apply" and not pmc_child:
            # side figures, never `value`: the factorisation next to the apply, the exact (level-scheduled)
            # solve the sweeps approximate, and what the sweeps are worth as a preconditioner
            try:
                p.set_timing(True)
                p.get_timing(reset=True)
                asf = _t(lambda: p.ilu0_factorize(cfg["build"]), reps=2)
                tf_ = p.get_timing(reset=True)
                p.set_timing(False)
                fms = tf_["lower_ms"] / max(tf_["lower_launches"], 1)
                tb = ab["factor_sweep_touched"]
                if bs >= 2 and cfg["build"] >= 1:
                    # (the timed sweeps are the build's: one fused first sweep + in-place sweeps; average per sweep)
                    tb = (ab["factor_sweep_fused_first"] + (cfg["build"] - 1) * ab["factor_sweep_touched"]) / cfg["build"]
                ftraffic = None
                if live_factor is not None:
                    ftraffic = {"hbm_bytes_per_launch": live_factor["hbm_bytes_per_launch"], "live": True,
                                "kernel": live_factor["kernel"], "launches_per_pass": live_factor["launches_per_pass"],
                                "seconds": round(live_factor["seconds"], 1)}
                try:
                    fent = {} if ftraffic else json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("factor", {}) if args.config == 2 and not custom else {}
                    if fent:
                        ftraffic = {"hbm_bytes_per_launch": fent.get("hbm_bytes_per_launch"), "live": False,
                                    "profiles": fent.get("from"), "commit": fent.get("commit")}
                except Exception:
                    pass
                out["factor"] = {"async_factor_ms": asf, "nbuildsweeps": cfg["build"], "sweep_ms": fms, "traffic": ftraffic,
                                 "algorithmic_bytes_per_sweep": ab["factor_sweep"],
                                 "touched_bytes_per_sweep": tb, "fixed_upper_blocks": ab["fixed_upper"],
                                 "achieved": tb / (fms * 1e-3) / 1e9 if fms > 0 else 0.0,
                                 "frac": tb / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS if fms > 0 else 0.0,
                                 "moved_gbps": (ftraffic["hbm_bytes_per_launch"] / (fms * 1e-3) / 1e9
                                                if ftraffic and ftraffic.get("hbm_bytes_per_launch") and fms > 0 else None),
                                 "frac_by_traffic": (ftraffic["hbm_bytes_per_launch"] / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                     if ftraffic and ftraffic.get("hbm_bytes_per_launch") and fms > 0 else None),
                                 "traffic_over_touched": (ftraffic["hbm_bytes_per_launch"] / tb
                                                          if ftraffic and ftraffic.get("hbm_bytes_per_launch") else None),
                                 "note": "in-place sweeps leave upper blocks without position pairs alone (their value "
                                         "is the matrix block; those that are the u_kj of a pair are still read once as "
                                         "operands): achieved / frac count the bytes the sweeps touch, not "
                                         "the every-array-once figure; at bs >= 2 the build's first sweep reads the matrix "
                                         "as its iterate and writes every block (the initialisation pass is fused into "
                                         "it), sweep_ms / touched bytes / traffic are averages over the build's sweeps"}
                ex = _t(lambda: p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z))
                st = p.level_stats()
                sy = _t(lambda: p.ilu0_apply(r, s, init=capi.INIT_A_ZERO, mode=capi.JACOBI_SYNC, out=z))
                out["exact_apply"] = {"ms": ex, "levels": st["levels"], "syncfree_aborts": st["syncfree_aborts"],
                                      "sync_sweeps_ms": sy,
                                      "note": "one exact L and U solve (mode LEVEL), and %d+%d SYNCHRONOUS sweeps "
                                              "(deterministic; the first sweep from zero needs no matrix), beside "
                                              "ms_per_step for %d+%d asynchronous sweeps" % (s, s, s, s)}
                # what the host C++ layer applies by default (BLASTED_HIP_SWEEP_MODE unset): timed beside the metric's mode
                dm = _t(lambda: p.ilu0_apply(r, s, init=capi.INIT_A_ZERO, mode=capi.ASYNC, out=z))
                out["default_mode_apply"] = {"mode": "async", "ms": dm,
                                             "note": "the host C++ layer's default sweep mode (operators.cpp, "
                                                     "HipOperator::sweep_mode) is the mode this line measures"}
                out["quality"] = quality_figures(p, capi, torch, r, z, s)
                out["quality"]["note"] = ("distance of z to the exact solves: a yardstick, not the goal -- in the reference's "
                                          "flexible solver the sweep variants rank by how REPEATABLE they are, the other way "
                                          "round (GCR(30) at 160^3, 3 sweeps: this default 684 iterations, rounds 1-2's sweep "
                                          "943-981, the interleaved order 1173-1196, synchronous sweeps 655: "
                                          "profiles/r03_sweep_order_quality.txt)")
                if bs in (4, 8):
                    # the other row order inside a chunk, measured beside the default: interleaved (a row's predecessor
                    # belongs to the step before) -- closer to the exact solves per sweep, 10 % dearer per sweep, and no
                    # better inside the reference's flexible solver (profiles/r03_sweep_order_quality.txt), hence not
                    # the default
                    try:
                        capi.set_tuning("interleave=1")
                        p.set_timing(True)
                        p.get_timing(reset=True)
                        alt_ms = _t(lambda: p.ilu0_apply(r, s, init=capi.INIT_A_ZERO, mode=capi.ASYNC, out=z))
                        ta = p.get_timing(reset=True)
                        p.set_timing(False)
                        ums = ta["upper_ms"] / max(ta["upper_launches"], 1)
                        alt = {"tuning": "interleave=1 (rows of a step four apart, the finished row forwarded in registers)",
                               "ms_per_step": alt_ms, "value": s / (alt_ms * 1e-3), "unit": "sweeps/s",
                               "lower_ms": ta["lower_ms"] / max(ta["lower_launches"], 1), "upper_ms": ums,
                               "upper_frac": ab["upper_sweep"] / (ums * 1e-3) / 1e9 / HBM_PEAK_GBS if ums > 0 else 0.0,
                               "gcr_iterations_160": {"default (late store, 2 steps in flight), 3 / 5 sweeps": "684 / 498",
                                                      "interleaved, 3 / 5 sweeps": "1173-1196 / 510-511",
                                                      "source": "profiles/r03_sweep_order_quality.txt"}}
                        alt.update(quality_figures(p, capi, torch, r, z, s))
                        out["sweep_order_alternative"] = alt
                    finally:
                        capi.set_tuning("interleave=0")
                if cfg["gen"] != "unstructured" and not args.no_cpu_baseline:
                    try:
                        out["quality"]["same_matrix_reference"] = reference_side_quality(cfg, capi, workloads, torch, dev, s)
                    except Exception as e:
                        out["quality"]["same_matrix_reference"] = {"failed": repr(e)}
                exf = _t(lambda: p.ilu0_factorize(-1), reps=2)
                out["exact_apply"]["exact_factor_ms"] = exf
            except Exception as e:
                out["exact_apply"] = {"ms": None, "note": "failed: %r" % (e,)}
        if not dry and world == 1 and args.config == 2 and not custom and not args.no_other_configs:
            # the other BASELINE configurations, briefly (5 steps each), so that the one line the driver records
            # carries a timed figure for every configuration; `python bench.py --config K` is the full run of each
            others = []
            try:
                others.append(short_run(3, dev, stream, capi, workloads, torch, reuse=(p, m, r, z)))
                p.close()
                p = None
                del m, r, z
                torch.cuda.empty_cache()
                for k in (4, 5, 1):
                    others.append(short_run(k, dev, stream, capi, workloads, torch))
                    torch.cuda.empty_cache()
            except Exception as e:
                others.append({"failed": repr(e)})
            for o in others:
                lk = live_others.get(o.get("baseline_config")) if isinstance(o, dict) else None
                if lk is not None and "roofline" in o:
                    o["roofline"]["traffic"] = lk["hbm_bytes_per_launch"]
                    o["roofline"]["traffic_over_algorithmic"] = lk["hbm_bytes_per_launch"] / o["roofline"]["algorithmic_bytes_per_launch"]
                    o["roofline"]["traffic_source"] = {"live": True, "kernel": lk["kernel"], "launches_per_pass": lk["launches_per_pass"],
                                                       "seconds": round(lk["seconds"], 1)}
            out["other_configs"] = others
        if not dry and world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, op, s, unit_bytes, units_per_step, dev=dev)
            except Exception as e:  # the baseline is a reported side figure, never the measurement
                out["cpu_baseline"] = {"value": None, "unit": "sweeps/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    if p:
        p.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
