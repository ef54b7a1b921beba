"""Sanitizer fuzz of the on-disk readers (host code only): tests/cpp/fuzz_files.cpp built with ASan + UBSan."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_readers_survive_mutated_files(tmp_path):
    exe = str(tmp_path / "fuzz_files")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tests", "cpp", "fuzz_files.cpp"), "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)  # (the sanitizer runtime must come first in the child; nothing is changed for this process)
    env["ASAN_OPTIONS"] = "detect_leaks=0"
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and run.stdout.startswith("ok"), (run.stdout + run.stderr)[-3000:]
