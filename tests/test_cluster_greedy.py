"""Host logic (no GPU): the greedy clustering of the cluster sweep engine (csrc/ldu_cluster_greedy.hpp) against its round-3
form (tools/greedy_check/greedy_reference.hpp: candidates in one vector, linear scan per pick) - the clusters, the order of
their members, the internal levels and the cluster levels must be the same, cluster for cluster: the cluster plan of the
216^3 headline (and with it the measured 0.90 ms launch) is a function of them."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("greedy") / "harness")
    src = os.path.join(ROOT, "tools", "greedy_check", "harness.cpp")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, src], capture_output=True, text=True, cwd=os.path.dirname(src))
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


@pytest.mark.parametrize("n,irregular", [(8, 0), (21, 0), (32, 1), (45, 1)])
def test_same_clusters_as_the_linear_scan(harness, n, irregular):
    r = subprocess.run([harness, str(n), str(irregular)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "identical 1" in r.stdout, r.stdout + r.stderr
