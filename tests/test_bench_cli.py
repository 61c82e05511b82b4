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
