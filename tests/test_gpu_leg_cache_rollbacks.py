"""Leg 3: leg 2 with every second device-side pass declared failed after it ran
(`CDH_GC_INJECT_ROLLBACK=2`, tests only): the undo (snapshot restore) and the windowed walk that takes
over are exercised on every problem of the suite, not only where a certificate really breaks."""
from _legs import adopt

_WHY = "asserts how many passes ran whole on the device; this leg undoes every second one on purpose"
SKIP = {
    "test_gradient_cache_keeps_g_on_the_device_and_reports_its_drift": _WHY,
    "test_cfg5_one_rank_shard_full_size_properties": _WHY,
    "test_cfg3_full_size_path_properties": _WHY,
}
adopt(globals(), {"CDH_GRADIENT_CACHE": "3", "CDH_GC_INJECT_ROLLBACK": "2"}, SKIP)
