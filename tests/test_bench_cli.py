"""bench.py's command line survives torch.distributed.run's own argument parser (which matches
abbreviations: an option like `--n` would be swallowed as `--nnodes` / `--nproc-per-node`), and the
CPU-baseline timing loop of the oracle is the same arithmetic as the oracle's descendCoordinate!."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_options_do_not_collide_with_torchrun_abbreviations():
    from torch.distributed.run import get_args_parser
    src = open(os.path.join(ROOT, "bench.py")).read()
    opts = re.findall(r'add_argument\("(--[a-z0-9-]+)"', src)
    assert {"--gpus", "--steps", "--warmup"} <= set(opts)
    argv = ["--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29500",
            "bench.py"]
    for o in opts:
        argv += [o] if o in ("--graph", "--no-cpu-baseline", "--no-sparse", "--no-rccl", "--no-exchange-trial",
                             "--no-cfg1", "--no-cfg3", "--no-live-traffic") else [o, "1"]
    ns = get_args_parser().parse_args(argv)
    assert ns.training_script == "bench.py"
    assert ns.training_script_args == argv[argv.index("bench.py") + 1:]   # every option reached the script
    assert ns.nproc_per_node == "2" and ns.nnodes == "1"


def test_cpu_baseline_loop_is_the_oracle_visit():
    import oracle as O
    rng = np.random.default_rng(5)
    n, p, lam = 400, 6, 0.05
    X = np.asfortranarray(rng.standard_normal((n, p)))
    y = X[:, :2] @ np.array([1.5, -2.0]) + rng.standard_normal(n)
    r, beta = y.copy(), np.zeros(p)
    O.lib().cdo_bench_ls_visits(n, p, O._ptr(X), n, O._ptr(r), O._ptr(beta), lam, 5 * p, 1)
    f, g, x = O.CDLeastSquaresLoss(y, X), O.ProxL1(lam), O.SparseIterate(p)
    O.initialize_(f, x)
    for _ in range(5):
        O.cdPass_(x, f, g, np.arange(1, p + 1))
    np.testing.assert_allclose(beta, x.dense(), rtol=0, atol=1e-14)
    np.testing.assert_allclose(r, f.r, rtol=0, atol=1e-12)


def test_direct_exchange_is_adopted_only_when_validated_and_faster():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    adopt = bench.adopt_direct_exchange
    assert adopt("auto", True, True, 3e-13, 0.040, 0.043)
    assert not adopt("rccl", True, True, 3e-13, 0.040, 0.043)       # asked for RCCL only
    assert not adopt("auto", False, True, 3e-13, 0.040, 0.043)      # self-test failed somewhere
    assert not adopt("auto", True, False, 3e-13, 0.040, 0.043)      # a rank did not complete
    assert not adopt("auto", True, True, 2e-9, 0.040, 0.043)        # beta differs from the RCCL sweep
    assert not adopt("auto", True, True, 3e-13, 0.050, 0.043)       # slower
    assert not adopt("auto", True, True, float("nan"), 0.040, 0.043)


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_launcher_less_command_relays_every_argument():
    """`python bench.py --gpus N` with no launcher: the parent starts torch.distributed.run with N ranks on
    127.0.0.1 and hands the caller's own arguments to the script untouched."""
    from torch.distributed.run import get_args_parser
    bench = _bench_module()
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "3", "--exchange", "auto", "--no-cfg3"]
    cmd = bench.launcher_command(argv, 8, 29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    ns = get_args_parser().parse_args(cmd[3:])
    assert ns.nproc_per_node == "8" and ns.nnodes == "1" and ns.master_addr == "127.0.0.1" and ns.master_port == 29511
    assert os.path.samefile(ns.training_script, os.path.join(ROOT, "bench.py"))
    assert ns.training_script_args == argv


def test_self_launch_relays_one_json_line_and_the_return_code(monkeypatch, capfd):
    """The parent prints rank 0's JSON line (only that) on stdout, everything else on stderr, and exits with
    the child's return code; a child that prints no result line is a failure even with rc 0."""
    import sys
    bench = _bench_module()
    line = '{"metric": "coord_updates_per_sec", "value": 1.0, "n_gpus": 2}'

    def fake(rc, with_line):
        body = "print('rank noise'); print('{\"not\": \"the line\"}');" + (f"print('{line}');" if with_line else "")
        return lambda argv, gpus, port: [sys.executable, "-c", body + f"import sys; sys.exit({rc})"]

    monkeypatch.setattr(bench, "launcher_command", fake(0, True))
    assert bench.self_launch(["--gpus", "2"], 2) == 0
    out, err = capfd.readouterr()
    assert out.strip() == line and "rank noise" in err and "not" in err
    monkeypatch.setattr(bench, "launcher_command", fake(7, True))
    assert bench.self_launch(["--gpus", "2"], 2) == 7
    capfd.readouterr()
    monkeypatch.setattr(bench, "launcher_command", fake(0, False))
    assert bench.self_launch(["--gpus", "2"], 2) == 1
    assert capfd.readouterr().out.strip() == ""


def test_bench_gpus_n_without_a_launcher_starts_n_ranks_before_touching_the_gpu():
    """End to end on a box WITHOUT a GPU: `python bench.py --gpus 2` (WORLD_SIZE unset) must get as far as two
    ranks each failing in cdh_create ("no HIP device") -- the parent itself never loads the library -- and the
    parent must relay the failure as a non-zero exit code with no result line."""
    import subprocess
    import sys
    if os.path.exists("/dev/kfd"):
        import pytest
        pytest.skip("CPU-only check; the GPU box runs the launcher-less bench for real (-m gpu)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--rows", "1000", "--cols", "8", "--no-cpu-baseline", "--no-sparse"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.stderr.count("no HIP device") >= 2 or r.stderr.count("HipError") >= 2, r.stderr[-2000:]
