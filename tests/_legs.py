"""Re-collection of the GPU parity cases under another execution path of the product.

The library reads its path switches from the environment when a handle is created
(`cdh_create`, csrc/cdhip.hip), so a *leg* is: the same test functions as
tests/test_gpu_parity.py + tests/test_gpu_configs.py + tests/test_gpu_edge_matrix.py, collected a second
time in a module whose autouse fixture exports the switches before each test builds its handles.
What stays out of a leg: tests that start other processes (they build their own environment and
test the launcher, not the sweep), and the few that assert the statistics of the very path the
leg overrides (named in each leg's `SKIP`, with the reason).
"""
import importlib

import pytest

SOURCES = ("test_gpu_parity", "test_gpu_configs", "test_gpu_edge_matrix")

# tests that spawn processes of their own (torch.distributed.run, bench.py): not about the path under test
OWN_PROCESSES = {
    "test_rccl_path_under_torchrun_matches_plain_run",
    "test_p2p_exchange_two_ranks_one_gpu",
    "test_sharded_launch_sequence_over_host_exchange",
    "test_bench_sharded_over_host_exchange_two_ranks_one_gpu",
    "test_bench_gpus_2_without_a_launcher_two_ranks_one_gpu",
    "test_bench_two_ranks_when_rccl_cannot_serve_them_still_measures_the_sharded_sweep",
    "test_bench_an_rccl_bring_up_that_never_returns_costs_a_timeout_not_the_result_line",
    "test_bench_a_direct_exchange_probe_that_dies_costs_nothing",
    "test_bench_sharded_with_p2p_exchange_two_ranks_one_gpu",
    "test_bench_exchange_trial_code_path_two_ranks_one_gpu",
    # 80 GB generated per run and no solve in it: once is enough
    "test_full_size_properties_cfg2",
    "test_handles_with_every_optional_buffer_are_created_and_destroyed_repeatedly",
}


def adopt(namespace, env, skip=None):
    """Put every test of SOURCES into `namespace` (a leg module's globals()) and install the
    autouse fixture that exports `env` around each of them."""
    skip = dict(skip or {})
    for mod in SOURCES:
        m = importlib.import_module(mod)
        for name, obj in vars(m).items():
            if not (name.startswith("test_") and callable(obj)) or name in OWN_PROCESSES:
                continue
            if name in skip:
                obj = _skipped(obj, skip[name])
            namespace[name] = obj

    @pytest.fixture(autouse=True)
    def _leg_environment(monkeypatch):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        yield

    namespace["_leg_environment"] = _leg_environment
    namespace["pytestmark"] = pytest.mark.gpu


def _skipped(fn, reason):
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **k):
        pytest.skip(reason)

    # parametrisation marks travel with the function object: keep them
    wrapper.pytestmark = list(getattr(fn, "pytestmark", []))
    return wrapper
