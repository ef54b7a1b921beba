"""bench.py's other configurations at toy sizes: each must print one JSON line whose in-run comparison against the CPU oracle
says bit-exact (the default configuration is what the driver runs; these are the ones only the builder runs)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", *args], capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(450)
@pytest.mark.parametrize("extra", [(), ("--svdag-policy", "nonresident"), ("--kind", "any")], ids=["reference", "svdag_nonresident", "any"])
def test_two_level_config_small(extra):
    j = _bench("--config", "two-level", "--city-blocks", "4", "--side", "512", "--svdag", "32", *extra)
    assert j["cpu_baseline"]["gpu_matches_bit_exact"] is True
    assert j["config"]["batching_points"] >= 2 and j["value"] > 0 and j["roofline"]["frac"] > 0
    if "nonresident" in extra:
        assert j["config"]["svdag_tests"] == 0


@pytest.mark.timeout(450)
@pytest.mark.parametrize("frac", ["0", "0.25"], ids=["resident", "streamed"])
def test_landscape_config_small(frac):
    j = _bench("--config", "landscape", "--landscape-instances", "20000", "--landscape-prototypes", "8", "--landscape-terrain", "128", "--side", "384",
               "--leaf-cache-frac", frac)
    assert j["cpu_baseline"]["gpu_matches_bit_exact"] is True
    assert j["config"]["instances"] == 20000 and j["config"]["primary_rays_hitting_an_instance"] > 0.01
    if frac != "0":  # (loads of the last step: none when every waiting queue was sparse enough to be traced in host memory)
        assert j["config"]["leaf_loads_per_step"] >= 0 and j["config"]["rounds"] >= 1


@pytest.mark.timeout(450)
def test_single_level_shadow_rays_small():
    j = _bench("--tris", "20000", "--side", "512", "--rays", "shadow", "--kind", "any", "--no-two-level")
    assert j["cpu_baseline"]["gpu_matches_bit_exact"] is True and 0.05 < j["config"]["hit_fraction"] < 0.95 and "two_level" not in j


@pytest.mark.timeout(450)
def test_default_line_carries_the_two_level_record_small():
    """the driver's command (`bench.py --gpus 1`) at toy sizes: headline = configs[1], `two_level` = configs[2] with its own
    roofline, CPU baseline and bit-exact verdict"""
    j = _bench("--tris", "20000", "--side", "512", "--city-blocks", "4", "--svdag", "32", "--two-level-steps", "2")
    assert j["cpu_baseline"]["gpu_matches_bit_exact"] is True and "configs[1]" in j["config"]["workload"]
    t = j["two_level"]
    assert "error" not in t and t["gpu_matches_bit_exact"] is True and t["value"] > 0 and t["ms_per_step"] > 0
    assert "configs[2]" in t["config"]["workload"] and t["roofline"]["frac"] > 0 and t["cpu_baseline"]["value"] > 0
    assert "sharded" not in j


@pytest.mark.timeout(450)
@pytest.mark.parametrize("extra", [("--builder", "device"), ("--builder", "resident"), ("--builder", "resident", "--cut", "device")], ids=["device", "resident", "resident_device_cut"])
def test_two_level_config_with_trees_built_on_the_gpu_small(extra):
    j = _bench("--config", "two-level", "--city-blocks", "4", "--side", "512", "--svdag", "32", *extra)
    assert j["cpu_baseline"]["gpu_matches_bit_exact"] is True
    assert j["config"]["bvh8_builder"] == extra[1] and j["config"]["batching_points"] >= 2 and j["value"] > 0


@pytest.mark.timeout(450)
@pytest.mark.parametrize("builder", ["device", "resident"])
def test_single_level_config_with_a_tree_built_on_the_gpu_small(builder):
    j = _bench("--tris", "20000", "--side", "512", "--no-two-level", "--builder", builder)
    assert j["cpu_baseline"]["gpu_matches_bit_exact"] is True and j["config"]["bvh8_builder"] == builder and j["config"]["bvh8_build_device_ms"] > 0
