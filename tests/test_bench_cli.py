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
        argv += [o] if o in ("--graph", "--no-cpu-baseline", "--no-sparse") else [o, "1"]
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
