#!/usr/bin/env python3
"""bench.py's `cpu_baseline` leg, run as a child process — TEST / BENCH INFRASTRUCTURE ONLY.

    python oracle/cpu_bench_child.py <frames.npy> <kind> <budget_s> [min_pairs]
        kind = cpu_tvl1 : oracle/cpu_tvl1_baseline.c, the restatement of CPU cv::optflow::DualTVL1OpticalFlow
                          (the comparator BASELINE.json's north_star names), built here with -O3 -march=native
               tvl1 | farn | brox : the parity oracle (cv::cuda semantics), timed as a second CPU point
    python oracle/cpu_bench_child.py <in.npz> parity:<tvl1|farn|brox> <out.npz>
        bench.py's `parity_check`: the parity oracle's flows (and, TVL1, executed iteration tables) of the listed pairs
        of the very frames the timed run used.  in.npz: frames (n,H,W) u8, pairs (k,2) indices into frames, params (JSON:
        TVL1 parameter overrides).  out.npz: flow_<i>, iters_<i>.  The checker, never the thing measured.

A child process so that the OpenMP runtime starts with a placement chosen HERE, before any library is loaded:
one thread per physical core of the CPUs this process may run on, pinned (GOMP_CPU_AFFINITY).  Round 1 timed the
oracle inside the bench process after the HIP runtime was up and got 0.21 and 0.65 pairs/s on two boxes of the same
class; unpinned 256-thread teams across both sockets were up to 100x slower still.  The sample is run three
times and the median is reported, with the spread, so that an unstable box is visible in the JSON line.
Prints one JSON object."""
import json
import os
import sys
import time


def physical_cores(allowed):
    """One logical CPU per physical core among `allowed`, sorted by (package, core)."""
    seen, pick = set(), []
    for cpu in sorted(allowed):
        base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
        try:
            with open(base + "physical_package_id") as f:
                pkg = int(f.read())
            with open(base + "core_id") as f:
                core = int(f.read())
        except (OSError, ValueError):
            pkg, core = 0, cpu
        if (pkg, core) not in seen:
            seen.add((pkg, core))
            pick.append((pkg, core, cpu))
    pick.sort()
    return [c for _, _, c in pick]


def cgroup_cpu_quota():
    """CPUs' worth of time this container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited.
    Measured on the MI355X boxes of this pool (profiles/round2/cpu_thread_scan.log): 256 logical CPUs visible,
    cpu.max = 16 CPUs — 16 threads give 0.54 pairs/s, 128 spinning threads 0.21 (throttled)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max":
            return max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = int(f.read())
        if q > 0:
            return max(1, q // p)
    except (OSError, ValueError):
        pass
    return None


def main():
    path, kind, budget_arg = sys.argv[1], sys.argv[2], sys.argv[3]
    budget = 0.0 if kind.startswith("parity:") else float(budget_arg)
    min_pairs = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # >= 10 pairs for the headline comparator
    allowed = os.sched_getaffinity(0)
    cpus = physical_cores(allowed)
    quota = cgroup_cpu_quota()
    if quota is not None:
        cpus = cpus[:quota]
    cap = int(os.environ.get("DFX_CPU_THREADS", "0"))
    if cap > 0:
        cpus = physical_cores(allowed)[:cap]
    os.environ["OMP_NUM_THREADS"] = str(len(cpus))
    os.environ["GOMP_CPU_AFFINITY"] = " ".join(str(c) for c in cpus)
    os.environ["OMP_PROC_BIND"] = "true"
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import numpy as np

    from oracle import oracle_py as O

    O.build()
    if kind.startswith("parity:"):
        algo = kind.split(":", 1)[1]
        z = np.load(path)
        frames, pairs = z["frames"], z["pairs"]
        over = json.loads(str(z["params"])) if "params" in z.files else {}
        out = {}
        O.tvl1_calc(np.zeros((160, 160), np.uint8), np.zeros((160, 160), np.uint8), threads=len(cpus))  # thread team up
        t0 = time.perf_counter()
        for i, (a, b) in enumerate(pairs):
            if algo == "tvl1":
                prm = O.tvl1_default_params()
                for k, v in over.items():
                    setattr(prm, k, type(getattr(prm, k))(v))
                flow, tr = O.tvl1_calc(frames[a], frames[b], prm, want_trace=True, threads=len(cpus))
                out[f"iters_{i}"] = np.array([r[:5] for r in tr.iters_table()[:tr.nscales]], np.int32)
            else:
                flow = {"farn": O.farneback_calc, "brox": O.brox_calc}[algo](frames[a], frames[b], threads=len(cpus))
            out[f"flow_{i}"] = flow
        dt = time.perf_counter() - t0
        np.savez(budget_arg, **out)
        print(json.dumps({"pairs": int(len(pairs)), "seconds": dt, "value": len(pairs) / max(dt, 1e-9),
                          "unit": "frame-pairs/s", "cores": len(cpus)}))
        return
    frames = np.load(path)
    if kind == "cpu_tvl1":
        fn = lambda a, b: O.cpu_tvl1_calc(a, b, native=True)
        what = "oracle/cpu_tvl1_baseline.c (CPU cv::optflow::DualTVL1OpticalFlow port), -O3 -march=native"
    else:
        base = {"tvl1": O.tvl1_calc, "farn": O.farneback_calc, "brox": O.brox_calc}[kind]
        fn = lambda a, b: base(a, b, threads=len(cpus))
        what = f"oracle/ ({kind}, cv::cuda semantics, the parity oracle), -O2"
    t0 = time.perf_counter()
    fn(frames[0], frames[1])  # warm-up: thread team, page faults, table construction
    t1 = time.perf_counter() - t0
    n = int(min(len(frames) - 1, max(min_pairs, 1, (budget / 3.0) // max(t1, 1e-3))))
    runs = []
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(n):
            fn(frames[i], frames[i + 1])
        runs.append(n / (time.perf_counter() - t0))
    runs.sort()
    h, w = frames[0].shape
    print(json.dumps({
        "value": runs[1],
        "unit": "frame-pairs/s",
        "cores": len(cpus),
        "spread": (runs[2] - runs[0]) / runs[1],
        "sample": f"median of 3 runs over {n} consecutive pairs of the same {w}x{h} clip; {what}; {len(cpus)} threads pinned "
                  f"one per core ({len(allowed)} CPUs visible, cgroup quota {quota if quota is not None else 'none'})",
    }))


if __name__ == "__main__":
    main()
