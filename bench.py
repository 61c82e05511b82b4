#!/usr/bin/env python3
"""bench.py -- coordinate updates per second of the full cyclic Lasso sweep.

One "step" = one full cyclic pass (`_cdPass!` over 1..p, src/coordinate_descent.jl:
94-110) = p coordinate updates on BASELINE.json's config: Lasso, dense Gaussian X,
n = 10,000,000, p = 1,000, fp64, generated in HBM (synthetic).  Each step
starts from beta = 0 with lambda = 1e-6 * lambda_max, so every coordinate moves (h != 0)
on its visit ("all-move") -- the most expensive regime: every
visit pays the residual update.  (At n = 1e7 the gradient noise is ~sigma/sqrt(n), so
SURVEY's 1e-3 * lambda_max would leave 90% of the coordinates at zero.)

N > 1 (launched by torch.distributed.run, one rank per GPU): the SAME 10M x 1000
problem with rows sharded across ranks (strong scaling), gradient scalars summed by an
RCCL all-reduce inside the library: that region is always timed first.  Both exchanges meet the
machine IN PROCESSES OF THEIR OWN first, before this one has touched a GPU
(coordinatedescent.jl_amd/p2p_probe.py rccl / p2p: whatever a transport that has never run on this
hardware can do -- a bring-up that never returns, a faulting peer store -- it does there, under a
timeout).  RCCL's probe failing: the sweep is measured over the host-staged exchange and the line says
why.  With --exchange auto (the default), only if every rank's direct-exchange probe validated are the same K steps timed
over it in this process, and `value` is the faster of the two if the direct one validated again
(adopt_direct_exchange) -- the other timing is in `exchange_trial`.  --exchange rccl:
RCCL only, nothing afterwards.

Prints ONE JSON line on rank 0.  `roofline` is computed from HIP events recorded on the
library's own stream around the sweep kernels; `cpu_baseline` times the CPU oracle's
restatement of the reference visit (kind "port": the reference is Julia, not runnable
here) on a bounded sample of the same data.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# dmabuf IPC (what this pool's driver supports) for RCCL and for the IPC-mapped inboxes of the direct
# exchange; must be in the environment before anything initialises HIP
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def profiled_traffic(kernel, rows, cols, dtype, block):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/hbm_traffic.json, made by tools/profile_round.sh: FETCH_SIZE x2 + WRITE_SIZE, KiB ->
    bytes, as MI355X_MICROARCH.md prescribes).  PMC counters cannot be read inside this process,
    so the figure is only reported for the exact configuration that was profiled; else None."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    except Exception:
        return None
    return tab.get(f"{kernel}:n{rows}:p{cols}:{dtype}:B{block}")


def measure_traffic_live(kernel, argv_tail, timeout_s=240):
    """HBM bytes per launch of the dominant kernel on THIS box, measured now: two child runs of this very
    script under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, no trace domains, as
    MI355X_MICROARCH.md prescribes), 1 warm-up + 2 timed steps each; KiB -> bytes, FETCH_SIZE doubled
    (gfx950 counts 64 B per 128-B request of a wide coalesced read).  Returns (bytes, detail) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    tmp = tempfile.mkdtemp(prefix="cdh_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    out = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "c", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                   "--no-sparse", "--no-cfg1", "--no-cfg3", "--no-live-traffic"] + argv_tail
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                    if row["Counter_Name"] == counter and kernel in row["Kernel_Name"]]
            if not vals:
                return None, f"no {kernel} launches in the {counter} pass"
            out[counter] = (sum(vals) / len(vals), len(vals))
        fetch = 2.0 * 1024.0 * out["FETCH_SIZE"][0]
        write = 1024.0 * out["WRITE_SIZE"][0]
        return fetch + write, {"fetch_bytes": fetch, "write_bytes": write, "launches_counted": out["FETCH_SIZE"][1]}
    except Exception as e:
        return None, str(e)[:160]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def esz_of(dtype):
    import numpy as np
    return np.dtype(dtype).itemsize


def host_threads(omp_max):
    """Threads this process may really use: affinity mask and cgroup CPU quota, not nproc."""
    n = min(omp_max, len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def host_cpu_model():
    """Model name of the box's CPU as /proc/cpuinfo gives it, and the logical CPUs the machine has in all."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return model, os.cpu_count()


def cpu_baseline(f, n, lam, visits_target_s=12.0):
    """CPU oracle (port of cd_differentiable_function.jl:96-99,107-109) on a 32-column
    slice of the same X, single thread (faithful: the reference has no threading) and
    all host cores (OpenMP over n: a ceiling the reference does not have)."""
    import numpy as np
    import coordinatedescent_jl_amd as cd
    import oracle as O
    L = O.lib()
    ncol = 32
    X = f.X_cols(0, ncol).astype(np.float64, copy=False)   # the device-generated data itself
    y = f.y.astype(np.float64, copy=False)
    out = {}
    beta_cpu = None
    cpu_model, machine_cpus = host_cpu_model()
    for threads in (1, host_threads(int(L.cdo_max_threads()))):
        r = y.copy()
        beta = np.zeros(ncol)
        # calibrate on one cycle over the slice, then size the sample to ~visits_target_s
        t0 = time.perf_counter()
        L.cdo_bench_ls_visits(n, ncol, O._ptr(X), n, O._ptr(r), O._ptr(beta), lam, ncol, threads)
        per = (time.perf_counter() - t0) / ncol
        if threads == 1:
            # exactly the first `ncol` visits of a cyclic sweep from beta = 0, r = y: what the timed GPU sweep's first
            # block of visits computes at full n (main() compares the two)
            out["first_cycle_beta"] = beta.copy()
        visits = int(max(ncol, min(4096, visits_target_s / per)) // ncol * ncol)
        t0 = time.perf_counter()
        L.cdo_bench_ls_visits(n, ncol, O._ptr(X), n, O._ptr(r), O._ptr(beta), lam, visits, threads)
        dt = time.perf_counter() - t0
        out[threads] = dict(value=visits / dt, unit="coord-updates/s", cores=threads, kind="port",
                            cpu_model=cpu_model, machine_logical_cpus=machine_cpus,
                            threads_allowed=len(os.sched_getaffinity(0)),
                            sample=f"{visits} visits cycling a {ncol}-column slice of the same X, n={n}, fp64; "
                                   f"{dt / visits * 1e3:.2f} ms/visit; p-sweep extrapolated x(p/visits)")
        if threads == 1:
            beta_cpu, cycles = beta.copy(), (visits + ncol) // ncol
    # parity at full n on the same slice: the CPU port has cycled the 32
    # columns to their fixed point; solve the same 32-column problem with the HIP path and compare
    fg = cd.CDLeastSquaresLoss(y, X)
    fg.set_sweep_mode("block", 16)
    xg = cd.SparseIterate(ncol)
    cd.coordinateDescent_(xg, fg, cd.ProxL1(lam), cd.CDOptions(maxIter=500, optTol=1e-13, randomize=False))
    out["parity"] = dict(max_abs_beta_diff=float(np.max(np.abs(xg.dense() - beta_cpu))), tolerance=1e-10,
                         sample=f"n={n}, first {ncol} columns of the same X, lambda as timed; HIP solve "
                                f"({fg.last_stats['passes']} passes, optTol 1e-13) vs {cycles} cyclic sweeps of the CPU port")
    fg.close()
    return out


def cfg1_cpu_vs_gpu(device=0):
    """BASELINE.json configs[0] run in full on both sides: lasso, n=1000, p=200, s=10,
    sigma=1, lambda=0.1, host-generated, ordered sweeps, optTol 1e-7 -- the reference's own CPU-runnable
    case.  CPU = the oracle's C restatement (kind "port", 1 thread); GPU = the same solve through the
    C ABI, upload excluded (data resident, as for `value`) and included.  At this size a streamed pass is 200
    visits of 8 KB columns, i.e. launch latency and nothing else (round 2: 0.71 ms per solve); since round 3 such a
    problem is solved in one launch on the resident Gram matrix (cdh_set_onchip_solve), reported as it is."""
    import numpy as np
    import coordinatedescent_jl_amd as cd
    import oracle as O
    rng = np.random.default_rng(123)
    n, p, s, lam = 1000, 200, 10, 0.1
    X = np.asfortranarray(rng.standard_normal((n, p)))
    y = X[:, :s] @ (rng.standard_normal(s) * (1.0 + rng.random(s))) + rng.standard_normal(n)
    o = dict(maxIter=2000, optTol=1e-7, randomize=False)
    reps = 20
    fo = O.CDLeastSquaresLoss(y, X)
    t0 = time.perf_counter()
    for _ in range(reps):
        xo = O.SparseIterate(p)
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
    t_cpu = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    f = cd.CDLeastSquaresLoss(y, X, device=device)
    t_up = time.perf_counter() - t0
    xg = cd.SparseIterate(p)
    cd.coordinateDescent_(xg, f, cd.ProxL1(lam), cd.CDOptions(**o))      # warm the launch path
    t0 = time.perf_counter()
    for _ in range(reps):
        xg = cd.SparseIterate(p)
        cd.coordinateDescent_(xg, f, cd.ProxL1(lam), cd.CDOptions(**o))
    t_gpu = (time.perf_counter() - t0) / reps
    passes, visits = f.last_stats["passes"], f.last_stats["visits"]
    onchip, last = f.onchip_stats(), f.onchip_last()
    # the reference's DEFAULT visit order is shuffled (CDOptions.randomize = true, src/utils.jl:18): the same solve with seeded
    # shuffled sweeps on both sides (the documented splitmix64 substitute for Julia's global RNG), same bar
    osh = dict(o, randomize=True, seed=7)
    t0 = time.perf_counter()
    for _ in range(reps):
        xso = O.SparseIterate(p)
        sts = O.coordinateDescent_(xso, fo, O.ProxL1(lam), O.CDOptions(**osh))
    t_cpu_sh = (time.perf_counter() - t0) / reps
    xs = cd.SparseIterate(p)
    cd.coordinateDescent_(xs, f, cd.ProxL1(lam), cd.CDOptions(**osh))
    t0 = time.perf_counter()
    for _ in range(reps):
        xs = cd.SparseIterate(p)
        cd.coordinateDescent_(xs, f, cd.ProxL1(lam), cd.CDOptions(**osh))
    t_gpu_sh = (time.perf_counter() - t0) / reps
    shuffled = {"cpu_port_ms": t_cpu_sh * 1e3, "gpu_ms": t_gpu_sh * 1e3, "passes": f.last_stats["passes"], "cpu_passes": sts["passes"],
                "visits": f.last_stats["visits"], "max_abs_beta_diff": float(np.max(np.abs(xs.dense() - xso.dense()))),
                "same_support_order": bool(xs.nzval2ind.tolist() == xso.nzval2ind.tolist()),
                "solve_kernel_us": f.onchip_last()["kernel_us"], "solve_kernel_steps": f.onchip_last()["steps"]}
    f.close()
    return {"workload": "lasso_n1000_p200_s10_lambda0.1_full_solve", "cpu_port_ms": t_cpu * 1e3, "cpu_cores": 1,
            "gpu_ms": t_gpu * 1e3, "gpu_ms_incl_upload_and_create": (t_gpu + t_up) * 1e3,
            "passes": passes, "visits": visits, "cpu_passes": st["passes"],
            "cpu_coord_updates_per_sec": st["visits"] / t_cpu, "gpu_coord_updates_per_sec": visits / t_gpu,
            "max_abs_beta_diff": float(np.max(np.abs(xg.dense() - xo.dense()))), "tolerance": 1e-10,
            # the GPU side is ONE launch per solve (csrc/small_solve.hpp): how many solves took it, and of the last one the
            # kernel's own time, its visit steps and the clock the chip held for a one-wave kernel
            "one_launch_solves": onchip["solves"], "solve_kernel_us": last["kernel_us"], "solve_kernel_steps": last["steps"],
            "solve_kernel_clock_GHz": last["clock_GHz"], "shuffled_sweeps": shuffled}


def cfg3_path(device=0, n=2_000_000, p=5000, nlam=100):
    """BASELINE.json configs[2] on one GPU (a parity-test case timed for context, never part of `value`):
    warm-started 100-lambda Lasso path, n = 2e6, p = 5000 (80 GB generated in HBM), omega = _stdX!, ordered
    sweeps, optTol 1e-7, log-spaced lambdas from lambda_max to 1e-2 lambda_max -- the loop of LassoPath
    (src/lasso.jl:250-252) as coordinatedescent_jl_amd.LassoPath runs it (carried residual reused, gradient
    cache from the first full pass).  Its parity and properties are tests/test_gpu_configs.py."""
    import numpy as np
    import coordinatedescent_jl_amd as cd
    f, _ = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=100, noise=6.0, device=device)
    try:
        x = cd.SparseIterate(p)
        cd.initialize_(f, x)
        om = cd.stdX(f)
        lmax = cd.findLambdaMax(x, f, cd.ProxL1(1.0, om))
        lams = np.exp(np.linspace(np.log(lmax), np.log(1e-2 * lmax), nlam))
        opt = cd.CDOptions(optTol=1e-7, randomize=False)
        cd._lib.check(f._L.cdh_set_reuse_residual(f._h, 1), f._h)
        f.set_gradient_cache(2)
        f._L.cdh_synchronize(f._h)
        t0 = time.perf_counter()
        passes = visits = 0
        for lam in lams:
            cd.coordinateDescent_(x, f, cd.ProxL1(lam, om), opt)
            passes += f.last_stats["passes"]
            visits += f.last_stats["visits"]
        f._L.cdh_synchronize(f._h)
        dt = time.perf_counter() - t0
        out = {"workload": f"lasso_path_{nlam}_lambdas_n{n}_p{p}_f64_warm_started", "seconds": dt, "passes": passes,
               "visits": visits, "visits_per_sec": visits / dt, "nnz_last": int(x.nnz), "gradient_cache": f.cache_stats()}
        # outside the timed region: what the path produced is CHECKED, not only timed.  (1) The cached gradient is taken
        # afresh from X (one dots-only pass) and its drift from the carried one measured in units of the thresholds: the
        # certificates' margin is 1e-9.  (2) The KKT conditions of the last lambda from X'r computed on the device:
        # |X_k'r| / n = lambda omega_k on the support (to what optTol 1e-7 allows), <= it off the support.
        out["cache_drift_over_threshold"] = f.cache_drift(rereference_now=True)
        xtr = np.zeros(p)
        cd._lib.check(f._L.cdh_xt_r(f._h, xtr.ctypes.data), f._h)
        grad, thr, beta = np.abs(xtr) / n, lams[-1] * om, x.dense()
        act = beta != 0
        out["kkt_last_lambda"] = {"max_rel_violation_on_support": float(np.max(np.abs(grad[act] - thr[act]) / thr[act])) if act.any() else 0.0,
                                  "max_ratio_off_support": float(np.max(grad[~act] / thr[~act])), "tolerance": "1e-5 / 1 + 1e-5 (optTol 1e-7)"}
        return out
    finally:
        f.close()


def ref_shape_path(device=0):
    """The reference's OWN benchmark shape (benchmark/cd_bench.jl:10-14: n = 3000, p = 5000, s = 100, noise 6) as a warm-started
    60-lambda LassoPath (src/lasso.jl:229-260; omega = _stdX!, optTol 1e-7, ordered sweeps) from 0.95 to 0.03 lambda_max, where
    the support reaches several hundred non-zeros -- timed for context, never part of `value`, and CHECKED against the CPU port
    of the oracle on the same data (beta at the last lambda, tolerance 1e-10).  The first run on the handle fetches the Gram
    columns of the coordinates that enter; `seconds` is the better of the two runs that follow."""
    import numpy as np
    import coordinatedescent_jl_amd as cd
    import oracle as O
    rng = np.random.default_rng(123)
    n, p, s = 3000, 5000, 100
    X = np.asfortranarray(rng.standard_normal((n, p)))
    Y = X[:, :s] @ (rng.standard_normal(s) * (1.0 + rng.random(s))) + 6.0 * rng.standard_normal(n)
    lmax = float(np.max(np.abs(X.T @ Y) / np.sqrt((X * X).mean(axis=0)))) / n
    lams = lmax * np.exp(np.linspace(np.log(0.95), np.log(0.03), 60))
    o = dict(maxIter=2000, optTol=1e-7, randomize=False)
    f = cd.CDLeastSquaresLoss(Y, X, device=device)
    try:
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            path = cd.LassoPath(f, None, lams, cd.CDOptions(**o))
            f._L.cdh_synchronize(f._h)
            times.append(time.perf_counter() - t0)
        beta = path.betapath[-1].dense()
        t0 = time.perf_counter()
        _, bo = O.LassoPath(X, Y, lams, O.CDOptions(**o))
        t_cpu = time.perf_counter() - t0
        ls = f.device_loop_stats()
        return {"workload": "lasso_path_60_lambdas_n3000_p5000_f64_warm_started (benchmark/cd_bench.jl's shape)", "seconds": min(times[1:]),
                "seconds_first_run_on_the_handle": times[0], "cpu_port_seconds": t_cpu, "cpu_cores": 1, "nnz_last": int(path.betapath[-1].nnz),
                "max_abs_beta_diff_last_lambda": float(np.max(np.abs(beta - bo[-1]))), "tolerance": 1e-10,
                "device_loop": {"launches": ls["launches"], "passes": ls["passes"], "table_passes": ls["table"]["passes"],
                                "helper_passes": ls["crew"]["passes"], "helper_jobs": ls["crew"]["jobs"]}}
    finally:
        f.close()


def isolated_exchange_probe(cp, device, what="p2p", timeout_s=150):
    """One child process per rank (same GPU, fresh rendezvous port) runs coordinatedescent.jl_amd/p2p_probe.py in mode
    `what` ("p2p": the direct exchange; "rccl": the communicator's bring-up and probe sums).  Returns (ok on every rank,
    info).  Nothing the children do can hurt this process: they are waited for with a timeout and killed by PID if they
    overstay."""
    import socket
    import struct
    import subprocess
    port = 0
    if cp.rank == 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    port = struct.unpack("<q", cp.broadcast_bytes(struct.pack("<q", port), 8, src=0))[0]
    env = dict(os.environ, RANK=str(cp.rank), LOCAL_RANK=str(cp.local_rank), WORLD_SIZE=str(cp.world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # the children rendezvous among themselves: nothing of the launcher's agent may leak in (with
    # TORCHELASTIC_USE_AGENT_STORE set, rank 0 would wait for a store nobody serves on the new port)
    for k in [k for k in env if k.startswith("TORCHELASTIC_") or k in ("GROUP_RANK", "ROLE_RANK", "ROLE_NAME",
                                                                      "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE", "LOCAL_WORLD_SIZE")]:
        env.pop(k, None)
    script = os.path.join(ROOT, "coordinatedescent.jl_amd", "p2p_probe.py")
    ok, info = False, {}
    try:
        tag = "RCCL_PROBE_" if what == "rccl" else "P2P_PROBE_"
        r = subprocess.run([sys.executable, script, str(device), what], env=env, capture_output=True, text=True,
                           timeout=timeout_s, cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith(tag)]
        ok = r.returncode == 0 and bool(line) and line[-1].startswith(tag + "OK")
        info = {"rc": r.returncode, "line": line[-1] if line else None}
        if ok:
            info["latency_us"] = float(line[-1].split()[1])
    except subprocess.TimeoutExpired:
        info = {"rc": None, "line": "timeout"}
    except Exception as e:             # pragma: no cover
        info = {"rc": None, "line": str(e)[:120]}
    all_ok = cp.sum_over_ranks(1.0 if ok else 0.0) == cp.world
    return all_ok, info


def adopt_direct_exchange(mode, selftest_ok, all_ranks_completed, max_abs_dbeta, t_direct, t_rccl):
    """--exchange auto: the direct exchange's timing becomes `value` only if it validated in this very
    run (self-test on every rank, every rank completed the K steps, beta within 1e-9 of the RCCL
    sweep) AND was faster than RCCL."""
    return bool(mode == "auto" and selftest_ok and all_ranks_completed and max_abs_dbeta <= 1e-9
                and 0.0 < t_direct < t_rccl)


def launcher_command(argv, gpus, port):
    """The command the parent of a launcher-less `bench.py --gpus N` starts: torch.distributed.run with one
    rank per GPU on this node, this very script and the caller's own arguments behind it."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(argv, gpus):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): this process becomes the
    launcher's parent.  It never touches the GPU -- no HIP call, no import of the library -- starts the N ranks
    as a CHILD process (never an exec), relays rank 0's single JSON line on stdout, everything else on stderr,
    and exits with the child's return code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    child = subprocess.Popen(launcher_command(argv, gpus, port), env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    lines = []
    for line in child.stdout:
        s = line.strip()
        if s.startswith("{") and '"metric"' in s:
            lines.append(s)
        else:
            sys.stderr.write(line)
    rc = child.wait()
    for s in lines[-1:]:
        print(s, flush=True)
    return rc if rc != 0 or lines else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--cols", type=int, default=1000)
    ap.add_argument("--planted", type=int, default=100)
    ap.add_argument("--noise", type=float, default=6.0)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--mode", default=os.environ.get("CDH_BENCH_MODE", "block"), choices=["coord", "block"])
    ap.add_argument("--block", type=int, default=None, choices=[2, 4, 8, 16, 32, 64],
                    help="visits per launch of the blocked sweep (default: 32 on one GPU; on row shards the faster of 32 "
                         "and 64 in one untimed sweep each -- 64 halves the exchanges per sweep)")
    ap.add_argument("--graph", action="store_true", help="replay each pass from a captured hipGraph")
    ap.add_argument("--lam-frac", type=float, default=1e-6, help="lambda / lambda_max")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sparse", action="store_true", help="skip the secondary sparse-regime timing")
    ap.add_argument("--exchange", default=os.environ.get("CDH_EXCHANGE", "auto"), choices=["auto", "rccl", "p2p"],
                    help="exchange when sharded.  auto (default): time K steps over RCCL; probe the direct exchange "
                         "in separate processes; if every rank's probe validated, time K steps over it here and "
                         "report the faster validated one (the other in exchange_trial).  rccl / p2p: that exchange only")
    ap.add_argument("--no-rccl", action="store_true",
                    help="TEST ONLY (ranks sharing one GPU, which RCCL refuses): build no communicator; with "
                         "--exchange rccl the timed region then has NO exchange and its numbers mean nothing")
    ap.add_argument("--no-rccl-probe", action="store_true",
                    help="at N > 1: build the RCCL communicator in this process without trying it in child processes first")
    ap.add_argument("--rccl-probe-timeout", type=float, default=120.0, help="seconds the isolated RCCL bring-up may take")
    ap.add_argument("--no-exchange-trial", action="store_true",
                    help="with --exchange auto: skip the second region (same as --exchange rccl)")
    ap.add_argument("--no-cfg1", action="store_true", help="skip the cfg1 (n=1000, p=200) CPU-vs-GPU solve timing")
    ap.add_argument("--no-cfg3", action="store_true", help="skip the cfg3 (100-lambda path, n=2e6, p=5000) timing and the 60-lambda path at the reference's benchmark shape")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 --pmc child runs (the committed figure is used)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], a.gpus))     # before anything below initialises HIP

    import numpy as np
    import coordinatedescent_jl_amd as cd
    from importlib import import_module
    sharded = import_module("coordinatedescent_jl_amd.sharded")

    cp = sharded.ControlPlane()
    if cp.world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={cp.world} ranks "
                 f"(run `python bench.py --gpus N` without a launcher, or make the two agree)")
    row0, n_local = sharded.shard_rows(a.rows, cp.rank, cp.world)
    dtype = np.float64 if a.dtype == "f64" else np.float32
    # At N > 1 the exchanges meet this machine in child processes first -- neither RCCL nor the direct exchange has ever run
    # across GPUs in this pipeline -- and they do so HERE, before this process has touched a GPU: a bring-up that never
    # returns or a faulting peer store costs a timeout there, not this process and its result line; and the box never
    # holds more than one process per rank on its GPUs at a time (nor the children's buffers next to this rank's shard).
    probes = {}
    if cp.world > 1:
        import torch
        dev0 = cp.local_rank % max(torch.cuda.device_count(), 1)     # (counts the devices without initialising one)
        if not a.no_rccl and not a.no_rccl_probe:
            probes["rccl"] = isolated_exchange_probe(cp, dev0, "rccl", timeout_s=a.rccl_probe_timeout)
        if a.exchange == "auto" and not a.no_exchange_trial and probes.get("rccl", (True, None))[0]:
            probes["p2p"] = isolated_exchange_probe(cp, dev0, "p2p")
    L = cd._lib.lib()
    ndev = cd._lib.C.c_int32()
    L.cdh_device_count(cd._lib.C.byref(ndev))
    device = cp.local_rank % max(ndev.value, 1)
    import struct
    # which card every rank drives, in rank order (one rank per GPU unless the box has fewer cards than ranks)
    devices = list(struct.unpack(f"<{cp.world}q", cp.all_gather_bytes(struct.pack("<q", device))))

    f, bstar = cd.CDLeastSquaresLoss.generate(n_local, a.cols, seed=123, s=a.planted, noise=a.noise, dtype=dtype,
                                              device=device, n_total=a.rows, row_offset=row0)
    exchange, exchange_fallback, rccl_probe = "rccl", None, None
    if a.no_rccl:
        exchange = "none(test-only)"
    else:
        # the communicator is built AND probed (exact sums of three records) before anything is timed over it.  A run
        # whose RCCL cannot be built or sums wrongly still measures the sharded sweep -- over the host-staged exchange,
        # which needs nothing from the interconnect and is far slower -- and the line says so instead of dying
        # ... and only if the same bring-up went through in the child processes above
        rccl_ok, why, rccl_probe = True, "", None
        if "rccl" in probes:
            rccl_ok, rccl_probe = probes["rccl"]
            if not rccl_ok:
                why = "isolated bring-up: " + str((rccl_probe or {}).get("line"))
        if rccl_ok:
            rccl_ok, why = sharded.connect_checked(f, cp)
        if not rccl_ok:
            f.comm_drop()
            sharded.connect_host(f, cp)
            exchange, exchange_fallback = "host(gloo)", why
            a.exchange, a.graph = "rccl", False      # no direct-exchange trial on top of a machine RCCL failed on; no graphs
    if cp.world > 1 and a.exchange == "p2p" and sharded.connect_p2p(f, cp):
        exchange = "p2p"
    block_auto = a.block is None
    if a.block is None:
        # B = 32 streams fastest per visit on long shards (measured 13.0 vs 13.8 us per visit at 5e6 rows);
        # from 2.5e6 rows down the two widths tie and B = 64 halves the exchanges.  fp32 B = 64 has no
        # LDS-transposed variant (it would spill) and runs far below B = 32.
        # (decided from a.rows // world, the same number on every rank: near-equal shards may straddle a cut on n_local)
        a.block = 32 if (cp.world == 1 or a.dtype == "f32" or a.rows // cp.world >= 4_000_000) else 64
    f.set_sweep_mode(a.mode, a.block)
    f.set_use_graph(a.graph)
    x = cd.SparseIterate(a.cols)
    cd.initialize_(f, x)
    lmax = cd.findLambdaMax(x, f, cd.ProxL1(1.0))
    g = cd.ProxL1(a.lam_frac * lmax)
    visit = np.arange(1, a.cols + 1, dtype=np.int64)

    def step():
        # one step = initialize!(f, 0) (beta = 0, r = y) + one full cyclic pass from there:
        # every coordinate moves (SURVEY.md 8d "all-move"); continuing instead would converge
        # within a few sweeps on this well-conditioned design and the visits would get cheap.
        x.fill_(0.0)
        cd.initialize_(f, x)
        return cd.cdPass_(x, f, g, visit)

    maxh = 0.0
    width_trial = None
    if block_auto and cp.world > 1 and a.mode == "block" and a.dtype == "f64" and exchange != "none(test-only)":
        # On row shards the width is a trade between the streaming kernel (B = 32 is up to 4 % faster per visit on long
        # shards) and the number of exchanges per sweep (B = 64 halves them), so what an exchange costs on THIS machine
        # decides -- which nobody has measured yet.  One untimed sweep per width after a first use of it; the slower rank's
        # time; every rank sees the same two numbers (collective max) and so makes the same choice.
        width_trial = {}
        for B in (32, 64):
            f.set_sweep_mode("block", B)
            step()
            cp.barrier()
            L.cdh_synchronize(f._h)
            tw = time.perf_counter()
            step()
            L.cdh_synchronize(f._h)
            width_trial[B] = cp.max_over_ranks(time.perf_counter() - tw)
        a.block = min(width_trial, key=width_trial.get)
        f.set_sweep_mode("block", a.block)
    for _ in range(a.warmup):
        step()
    cp.barrier()
    L.cdh_synchronize(f._h)
    f.profile_begin()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        maxh = step()
    L.cdh_synchronize(f._h)
    dt_local = time.perf_counter() - t0          # this rank's own K steps (reported as min / max over ranks)
    cp.barrier()
    dt = cp.max_over_ranks(time.perf_counter() - t0)
    ev_ms, launches, alg_bytes = f.profile_end()
    beta_timed, moved = x.dense().copy(), int(x.nnz)
    # the same K steps once more with every pass replayed from a captured hipGraph (north_star names it; measured
    # neutral on long columns: the stream is GPU-bound, not launch-bound).  Informational, never `value`; one GPU only.
    graph_trial = None
    if cp.world == 1 and not a.graph:
        try:
            f.set_use_graph(True)
            step()                          # the capture
            L.cdh_synchronize(f._h)
            tg = time.perf_counter()
            for _ in range(a.steps):
                step()
            L.cdh_synchronize(f._h)
            graph_trial = {"graph_ms_per_step": (time.perf_counter() - tg) / a.steps * 1e3,
                           "max_abs_dbeta_vs_timed": float(np.max(np.abs(x.dense() - beta_timed)))}
        except Exception as e:              # informational: never costs the result line
            graph_trial = {"error": str(e)[:200]}
        finally:
            f.set_use_graph(False)
    # the record one exchange carries in this sweep mode (4 doubles per coordinate; c, G, q per block)
    rec_doubles = {16: 273, 32: 801, 64: 2625}.get(a.block, 4 * a.block) if a.mode == "block" else 4
    exch_us = {}
    if cp.world > 1 and exchange != "none(test-only)":
        try:
            exch_us[exchange] = f.exchange_latency(rec_doubles, 200)  # back-to-back all-reduces, HIP events
        except Exception as e:                                        # informational: never costs the result line
            exch_us[exchange + "_error"] = str(e)[:200]

    # secondary, outside the timed region: the "sparse" regime of SURVEY 8d (lambda = 0.5 lambda_max,
    # few coordinates move, a visit is dots only).  Reported for context; never part of `value`.  Two ways:
    #  ms_per_sweep            full passes whose runs of non-moving visits are settled by dots-only screens
    #                          (cdh_set_screening 2, gradient cache off): X is read once per pass, the figure
    #                          to hold against the streaming rate
    #  certified_ms_per_sweep  the same passes once the gradient cache holds the Gram columns of the
    #                          support: X is not read at all for the settled visits (steady state of a
    #                          lambda path / sigma loop; same iterates)
    sparse = None
    # one GPU only: on row shards nothing that is not needed for `value` runs before the result line is out
    if not a.no_sparse and cp.world == 1:
        gs = cd.ProxL1(0.5 * lmax)
        nsp = max(2, min(a.steps, 5))

        def sparse_sweeps(warm):
            x.fill_(0.0)
            cd.initialize_(f, x)
            for _ in range(warm):
                cd.cdPass_(x, f, gs, visit)
            cp.barrier()
            L.cdh_synchronize(f._h)
            ts = time.perf_counter()
            for _ in range(nsp):
                cd.cdPass_(x, f, gs, visit)
            L.cdh_synchronize(f._h)
            cp.barrier()
            return cp.max_over_ranks(time.perf_counter() - ts)

        try:                   # secondary figures: a failure here must not cost the primary result
            f.set_screening(2)     # cdh_pass may settle runs of non-moving visits without visiting them one by one
            f.set_gradient_cache(0)
            dts = sparse_sweeps(1)
            sparse = {"lambda_over_lambda_max": 0.5, "ms_per_sweep": dts / nsp * 1e3,
                      "coord_updates_per_sec": nsp * a.cols / dts, "nnz": int(x.nnz),
                      "GBps_X_once": esz_of(dtype) * n_local * a.cols * nsp / dts / 1e9, "screened_pass": True}
            if a.dtype == "f64":
                f.set_gradient_cache(2)
                dtc = sparse_sweeps(3)
                sparse.update({"certified_ms_per_sweep": dtc / nsp * 1e3,
                               "certified_coord_updates_per_sec": nsp * a.cols / dtc, "cache": f.cache_stats()})
            f.set_gradient_cache(1)
            f.set_screening(1)
        except Exception as e:
            sparse = {"error": str(e)[:200]}

    esz = np.dtype(dtype).itemsize
    kernel = ("k_gramstep" if a.block >= 16 else "k_blockstep") if a.mode == "block" else "k_step"
    dt_loc_min, dt_loc_max = cp.min_over_ranks(dt_local), cp.max_over_ranks(dt_local)

    def result(exchange, dt, maxh, ev_ms, launches, alg_bytes):
        """The JSON line for a timed region (K steps, max-over-ranks wall time dt, HIP-event time ev_ms)."""
        updates = a.steps * a.cols
        achieved = alg_bytes / (ev_ms * 1e-3) / 1e9 if ev_ms > 0 else 0.0
        stream_model = esz * n_local * 5.0 * updates / (ev_ms * 1e-3) / 1e9 if ev_ms > 0 else 0.0
        st = f.exchange_stats()
        return {
            "metric": "coord_updates_per_sec", "value": updates / dt, "unit": "coord-updates/s",
            "n_gpus": cp.world, "devices_by_rank": devices, "visible_devices": ndev.value, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": a.dtype,
            "data": "synthetic",
            "config": {"workload": f"lasso_full_cyclic_sweep_gaussian_n{a.rows}_p{a.cols}_{a.dtype}_allmove",
                       "n": a.rows, "p": a.cols, "s": a.planted, "noise": a.noise, "lambda_over_lambda_max": a.lam_frac,
                       "sweep_mode": a.mode + (str(a.block) if a.mode == "block" else ""), "graph": bool(a.graph),
                       "parallelism": f"rows{cp.world}",
                       "exchange": exchange if (cp.world > 1 or st["rccl_calls"] > 0) else None,
                       "moved_per_sweep": moved, "last_maxH": maxh, "beta_abs_sum": float(np.abs(beta_timed).sum()),
                       **({"sweep_width_trial_ms": {str(k): v * 1e3 for k, v in width_trial.items()}} if width_trial else {})},
            # what the exchange itself reports: the communicator's rank count (ncclCommCount) and how many
            # all-reduces went through each transport since the handle was created (rank 0)
            "exchange_stats": st,
            "ms_per_step_ranks": {"min": dt_loc_min / a.steps * 1e3, "max": dt_loc_max / a.steps * 1e3},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": profiled_traffic(kernel, n_local, a.cols, a.dtype, a.block if a.mode == "block" else 1),
                         "traffic_source": "profiles/hbm_traffic.json: rocprofv3 --pmc passes of this exact "
                                           "configuration (committed), not counters of this run",
                         "kernel": kernel,
                         "launches": launches, "avg_launch_us": ev_ms * 1e3 / max(launches, 1),
                         "algorithmic_bytes_per_launch": alg_bytes / max(launches, 1),
                         "stream_model_5n_GBps": stream_model,
                         "floor_X_once_GBps": esz * n_local * updates / (ev_ms * 1e-3) / 1e9 if ev_ms > 0 else 0.0},
        }

    res = result(exchange, dt, maxh, ev_ms, launches, alg_bytes)
    if exchange_fallback:
        res["exchange_fallback"] = {"wanted": "rccl", "used": exchange, "why": exchange_fallback}
    if rccl_probe is not None:
        res["rccl_bring_up_probe"] = rccl_probe
    if sparse is not None:
        res["sparse_regime"] = sparse
    if exch_us:
        res["exchange_latency_us"] = dict(exch_us, doubles=rec_doubles, how="200 back-to-back all-reduces, HIP events")
    if graph_trial is not None:
        res["graph_ms_per_step"] = graph_trial.get("graph_ms_per_step")
        res["graph_trial"] = graph_trial

    # --exchange auto at N > 1.  The direct exchange has never run across GPUs in this pipeline, so its first
    # contact with this machine happens in child processes (isolated_exchange_probe); this process only touches it
    # after every rank's probe came back clean, and otherwise reports the RCCL region as it is.
    run_trial = cp.world > 1 and a.exchange == "auto" and exchange != "p2p" and not a.no_exchange_trial
    if run_trial:
        probe_ok, probe_info = probes["p2p"] if "p2p" in probes else isolated_exchange_probe(cp, device, "p2p")
        if not probe_ok:
            res["exchange_trial"] = {"exchange": "p2p", "probe": probe_info, "skipped": "the isolated probe did not validate on every rank"}
            run_trial = False
    if run_trial:
        trial, t_loc, ok_loc, err_loc, prof, maxh_p2p = {"exchange": "p2p", "probe": probe_info}, 0.0, False, 1e300, None, maxh
        try:
            connected = sharded.connect_p2p(f, cp)
        except Exception as e:          # pragma: no cover - connect_p2p is written not to raise
            connected, trial["error"] = False, str(e)[:200]
        trial["selftest"] = bool(connected)
        if connected:
            try:
                step()
                L.cdh_synchronize(f._h)
            except Exception as e:
                trial["error"] = str(e)[:200]
        cp.barrier()
        tt = time.perf_counter()
        if connected and "error" not in trial:
            try:
                f.profile_begin()
                for _ in range(a.steps):
                    maxh_p2p = step()
                L.cdh_synchronize(f._h)
                ok_loc = True
            except Exception as e:
                trial["error"] = str(e)[:200]
        cp.barrier()
        if ok_loc:                      # same bracket as the RCCL region: barrier, K steps, synchronize, barrier
            t_loc = time.perf_counter() - tt
            try:
                prof = f.profile_end()
                err_loc = float(np.max(np.abs(x.dense() - beta_timed)))
            except Exception as e:
                ok_loc, trial["error"] = False, str(e)[:200]
        all_ok = cp.sum_over_ranks(1.0 if ok_loc else 0.0) == cp.world
        if all_ok:
            try:
                exch_us["p2p"] = f.exchange_latency(rec_doubles, 200)
            except Exception as e:
                trial["error"] = str(e)[:200]
        t_max = cp.max_over_ranks(t_loc)
        err_max = cp.max_over_ranks(err_loc)
        if connected:
            trial.update({"completed_on_all_ranks": bool(all_ok),
                          "ms_per_step": t_max / a.steps * 1e3 if all_ok else None,
                          "max_abs_dbeta_vs_rccl": err_max if all_ok else None})
        t_min = cp.min_over_ranks(t_loc)
        if adopt_direct_exchange(a.exchange, connected, all_ok, err_max, t_max, dt):
            dt_loc_min, dt_loc_max = t_min, t_max
            rccl_line = res
            res = result("p2p", t_max, maxh_p2p, *prof)
            for k in ("sparse_regime",):
                if k in rccl_line:
                    res[k] = rccl_line[k]
            trial = {"exchange": "rccl", "ms_per_step": dt / a.steps * 1e3, "p2p_selftest": True,
                     "max_abs_dbeta_p2p_vs_rccl": err_max, "probe": probe_info}
        res["exchange_trial"] = trial
        if exch_us:
            res["exchange_latency_us"] = dict(exch_us, doubles=rec_doubles, how="200 back-to-back all-reduces, HIP events")
        try:
            f.p2p_enable(False)
        except Exception:
            pass

    if cp.rank == 0 and cp.world == 1 and not a.no_cpu_baseline:
        try:
            cb = cpu_baseline(f, n_local, g.lambda0)
            res["cpu_baseline"] = cb[1]
            res["parity_vs_cpu_port"] = cb.pop("parity")
            # the TIMED sweep's own output against the CPU port at full n: its first 32 visits (beta = 0, r = y, columns
            # 1..32 in order) are exactly the port's first cycle over the 32-column slice
            first = cb.pop("first_cycle_beta")
            nfirst = min(len(first), a.cols)
            diff = float(np.max(np.abs(beta_timed[:nfirst] - first[:nfirst])))
            res["timed_sweep_parity"] = {"max_abs_beta_diff_first_32_visits": diff, "tolerance": 1e-10, "ok": bool(diff <= 1e-10),
                                         "sample": f"beta of the timed sweep (n={n_local}, {a.mode}{a.block if a.mode == 'block' else ''}) after its "
                                                   f"first {nfirst} visits vs the CPU port's first cycle over the same columns from r = y"}
            mt = [v for k, v in cb.items() if k != 1]
            if mt:
                res["cpu_baseline_all_cores"] = mt[0]
        except Exception as e:
            res["cpu_baseline"] = {"error": str(e)[:200]}
    if cp.rank == 0 and cp.world == 1 and not a.no_cfg1 and not a.no_cpu_baseline:
        try:
            res["cfg1"] = cfg1_cpu_vs_gpu(device)
        except Exception as e:
            res["cfg1"] = {"error": str(e)[:200]}
    f.close()
    if cp.rank == 0 and cp.world == 1 and not a.no_live_traffic and not a.no_cpu_baseline:
        # the 80 GB are released: the two counter passes (child processes) have the device to themselves
        tail = ["--rows", str(a.rows), "--cols", str(a.cols), "--planted", str(a.planted), "--noise", str(a.noise),
                "--dtype", a.dtype, "--mode", a.mode, "--block", str(a.block), "--lam-frac", str(a.lam_frac)]
        live, detail = measure_traffic_live(kernel, tail)
        if live is not None:
            res["roofline"]["traffic"] = live
            res["roofline"]["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this script on this "
                                                 "box, just now (separate passes; KiB -> bytes, FETCH_SIZE x2)")
            res["roofline"]["traffic_detail"] = detail
        else:
            res["roofline"]["traffic_live_error"] = detail
    default_workload = (a.rows, a.cols, a.dtype) == (10_000_000, 1000, "f64")
    if cp.rank == 0 and cp.world == 1 and default_workload and not a.no_cfg3 and not a.no_cpu_baseline:
        try:                               # the 80 GB of cfg2 are released: cfg3's 80 GB fit
            res["cfg3_path"] = cfg3_path(device)
        except Exception as e:
            res["cfg3_path"] = {"error": str(e)[:200]}
    if cp.rank == 0 and cp.world == 1 and default_workload and not a.no_cfg3 and not a.no_cpu_baseline:
        try:
            res["ref_shape_path"] = ref_shape_path(device)
        except Exception as e:
            res["ref_shape_path"] = {"error": str(e)[:200]}
    if cp.rank == 0:
        print(json.dumps(res), flush=True)
    cp.shutdown()


if __name__ == "__main__":
    main()
