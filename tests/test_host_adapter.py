"""The C++ host adapter (include/pandora_amd/gpu_acceleration_structure.hpp) that mirrors the
reference's AccelerationStructure / ...Builder interface, driven by tests/cpp/test_adapter.cpp."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "pandora_amd", "lib", "test_adapter")


def _have_gpu():
    import torch
    return torch.cuda.is_available()


def test_adapter_reports_missing_gpu_cleanly():
    assert os.path.exists(EXE), "build() did not produce pandora_amd/lib/test_adapter"
    if _have_gpu():
        pytest.skip("GPU present: covered by the gpu test")
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=60)
    assert p.returncode == 2 and "pgpu_init" in p.stdout, p.stdout + p.stderr


@pytest.mark.gpu
def test_adapter_two_bounces_and_shadow_rays_on_gpu():
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "adapter ok" in p.stdout, p.stdout + p.stderr
