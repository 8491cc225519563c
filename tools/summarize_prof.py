#!/usr/bin/env python3
"""Summarises gpurun_out/prof_<tag>_* (tools/profile_round.sh) into profiles/<tag>_*.{csv,json}.
HBM traffic per launch = 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes): on gfx950 FETCH_SIZE reports half
the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    f = glob.glob(os.path.join(ROOT, "gpurun_out", pattern))
    return f[0] if f else None


# bench config number -> (key in profiles/traffic.json = the workload name, name fragments of its dominant kernel)
CONFIG_KERNELS = {
    # (in-place scalar sweeps: the general kernel, "scalarlane=auto")
    "1": ("poisson3d_64_csr_async_ilu0_apply", ("sweep_kernel<1, false, 1,", "sweep1_kernel<1, 1, 3,")),
    # (bench.py also times the interleaved row order beside the default: the IW instantiation, last template argument true)
    # (default in-place sweep: natural order, late store, two row steps in flight = UNR 2, LS true)
    "2": ("ilu_apply", ("sweepw_kernel<4, 1, 1, 1, 128, true, 2, 1, false, false, false, true>", "sweepw_kernel<4, 1,", "sweep_kernel<")),
    "3": ("poisson3d_256_bs4_async_block_sgs_relaxation", ("sweepw_kernel<4, 2,",)),
    "4": ("unstructured_126_bs5_async_block_ilu0_apply", ("sweepodd_kernel<5, 1,", "sweepx_kernel<5, 1,")),
    "5": ("poisson3d_100_bs8_block_ilu0_apply", ("sweepw_kernel<8, 1, 1, 1, 128, true, 1, 1, false, false, false, true>", "sweepw_kernel<8, 1,")),
}


def main(tag, op="ilu_apply"):
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    stats = one("prof_%s_stats/*/*_kernel_stats.csv" % tag)
    if stats:
        rows = [r for r in csv.reader(open(stats))]
        keep = [rows[0]] + [r for r in rows[1:] if "bhip::" in r[0]]
        with open(os.path.join(prof, "%s_kernel_stats.csv" % tag), "w", newline="") as f:
            csv.writer(f).writerows(keep)
    b = os.path.join(ROOT, "gpurun_out", "prof_%s_bench.json" % tag)
    if os.path.exists(b):
        shutil.copy(b, os.path.join(prof, "%s_bench.json" % tag))
    pmc = {}
    for name, pat in (("FETCH_SIZE", "prof_%s_fetch/*/*_counter_collection.csv"), ("WRITE_SIZE", "prof_%s_write/*/*_counter_collection.csv")):
        f = one(pat % tag)
        if not f:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "bhip::" in r["Kernel_Name"]:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            pmc.setdefault(k, {})[name + "_KB_avg"] = sum(v) / len(v)
            pmc[k]["launches_" + name] = len(v)
    # the other counter passes (SQ, L2): per-kernel averages, and the ratios they are read for
    for pat in ("prof_%s_sq/*/*_counter_collection.csv", "prof_%s_tcc/*/*_counter_collection.csv"):
        f = one(pat % tag)
        if not f:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "bhip::" in r["Kernel_Name"]:
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            for cn, v in cs.items():
                pmc.setdefault(k, {})[cn + "_avg"] = sum(v) / len(v)
    for k, d in pmc.items():
        if d.get("SQ_WAVE_CYCLES_avg"):
            for cn in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if cn + "_avg" in d:
                    d[cn + "_frac_of_wave_cycles"] = d[cn + "_avg"] / d["SQ_WAVE_CYCLES_avg"]
        if d.get("TCC_HIT_sum_avg") is not None and d.get("TCC_MISS_sum_avg") is not None:
            tot = d["TCC_HIT_sum_avg"] + d["TCC_MISS_sum_avg"]
            d["L2_hit_rate"] = d["TCC_HIT_sum_avg"] / tot if tot else None
    for k, d in pmc.items():
        if "FETCH_SIZE_KB_avg" in d and "WRITE_SIZE_KB_avg" in d:
            d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE_KB_avg"] + d["WRITE_SIZE_KB_avg"]) * 1024.0
    json.dump(pmc, open(os.path.join(prof, "%s_pmc.json" % tag), "w"), indent=1)
    # bench.py reads the dominant kernel's traffic from profiles/traffic.json: the kernel bench.py's
    # roofline object is quoted on (the descending/upper sweep for the apply ops), i.e. the bhip::
    # sweep or factor kernel with the most launches in the FETCH pass, ties broken by traffic
    want = {"ilu_apply": ("sweepw_kernel<4, 1,", "sweep_kernel<"), "sgs_apply": ("sweepw_kernel<4, 1,", "sweep_kernel<"),
            "sgs_relax": ("sweepw_kernel<4, 2,", "sweep_kernel<"), "spmv": ("sweepw_kernel<4, 3,", "sweep_kernel<"),
            "factor": ("factor4_kernel", "factor_sweep_kernel"),
            # ad-hoc scalar profiles at 256^3 (bench.py --n 256 --bs 1 --op factor | ilu_apply)
            "scalar_factor": ("factor1p_kernel", "factor1_kernel"), "scalar_sweeps": ("sweep1_kernel<1, 1, 3,", "sweep_kernel<1, false, 1,"),
            "scalar_spmv": ("sweep1s_kernel<3, 3, 0>", "sweep_kernel<1, false, 3,")}.get(op, ("sweepw_kernel",))
    if op in CONFIG_KERNELS:  # "summarize_prof.py <tag> <config number>"
        op, want = CONFIG_KERNELS[op]
    dom = []
    for w in want:
        dom = sorted([k for k in pmc if w in k and "hbm_bytes_per_launch" in pmc[k]],
                     key=lambda k: (-pmc[k]["launches_FETCH_SIZE"], -pmc[k]["hbm_bytes_per_launch"]))
        if dom:
            break
    tf = os.path.join(prof, "traffic.json")
    cur = json.load(open(tf)) if os.path.exists(tf) else {}
    import subprocess
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
    except Exception:
        commit = None

    def entry(k):
        return {"kernel": k, "hbm_bytes_per_launch": pmc[k]["hbm_bytes_per_launch"], "from": tag, "commit": commit,
                "launches_in_fetch_pass": pmc[k].get("launches_FETCH_SIZE")}
    if dom:
        cur[op] = entry(dom[0])
    # side kernels of the default configuration's run: the factorisation sweep and the exact (level-scheduled) passes
    extras = {"factor": ("factor4_kernel",), "ilu_apply_interleaved_order": ("sweepw_kernel<4, 1, 1, 1, 128, true, 1, 1, false, false, true, false>",),
              "exact_lower_pass": ("sfw_kernel<4, false",), "exact_upper_pass": ("sfw_kernel<4, true",),
              "lower_sweep": ("sweepw_kernel<4, 0, 0, 0, 128, true, 2, 1, false, false, false, true>",)}
    if op == "ilu_apply":
        for name, frags in extras.items():
            ks = sorted([k for k in pmc if any(f in k for f in frags) and "hbm_bytes_per_launch" in pmc[k]],
                        key=lambda k: -pmc[k]["launches_FETCH_SIZE"])
            if ks:
                cur[name] = entry(ks[0])
        ex = {k: v for k, v in pmc.items() if "sfw_kernel" in k or "sf_fill_kernel" in k or "sf_sweep_kernel" in k}
        if ex:
            json.dump(ex, open(os.path.join(prof, "%s_exact_solve_pmc.json" % tag.split("_c")[0]), "w"), indent=1)
            if stats:
                keep = [rows[0]] + [r for r in rows[1:] if "sfw_kernel" in r[0] or "sf_fill_kernel" in r[0] or "sf_sweep_kernel" in r[0]]
                with open(os.path.join(prof, "%s_exact_solve_kernel_stats.csv" % tag.split("_c")[0]), "w", newline="") as f:
                    csv.writer(f).writerows(keep)
    json.dump(cur, open(tf, "w"), indent=1)
    print(json.dumps(pmc, indent=1)[:6000])


if __name__ == "__main__":
    main(*sys.argv[1:])
