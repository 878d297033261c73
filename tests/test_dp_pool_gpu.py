"""The DP worker pool (MPA_DP_POOL=1: k_dp_worker, device-resident unit queues shared by the rounds in flight -- dp_kernels.hip) is
not the default executor of a DP round, but it is product code: the operator tests against the oracle and the whole-path golden
cases are run once more with it, in a process of their own (the switch is read when the library first runs a round)."""
import os
import subprocess
import sys
import pytest
import refbind

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("budget", ["768", "96"])          # 96: far fewer workers than units, every round lives on admission and stealing
def test_operator_and_whole_path_with_the_worker_pool(budget):
    env = dict(os.environ, MPA_DP_POOL="1", MPA_DP_WORKERS=budget)
    sel = ["tests/test_dp_gpu.py", "tests/test_map_gpu.py::test_dpp3_paf_identical", "tests/test_map_gpu.py::test_synthetic_paf_identical",
           "tests/test_map_gpu.py::test_batch_stream_matches_single_batch", "tests/test_scale_gpu.py::test_long_introns_and_the_widest_extension_classes"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + sel, cwd=refbind.ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
