"""Thread- and address-sanitizer builds of the threaded host code (csrc/sampler.hip is host C++: std::thread helpers inside a
call, and K pipeline threads calling dgll_host_sample_batch_seeded side by side -- pipeline.py, sampler_threads=K).  SURVEY.md
section 5 asks for a sanitizer story; GPU sanitizers are not available on this pool, so this is the CPU build, g++ only."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("target", ["tsan", "asan"])
def test_threaded_host_sampler_is_clean_under_the_sanitizer(target, tmp_path):
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.fail("g++ / make missing: the sanitizer build cannot run")
    proc = subprocess.run(["make", "-C", os.path.join(HERE, "c_abi"), target, "OUT=" + str(tmp_path)], capture_output=True, text=True,
                          timeout=600)
    out = proc.stdout + proc.stderr
    if "unexpected memory mapping" in out:      # the TSAN runtime could not lay out its shadow even with randomisation off
        pytest.skip("ThreadSanitizer's runtime refuses this host's address-space layout (not a finding about the code)")
    assert proc.returncode == 0, out[-4000:]
    assert "ThreadSanitizer" not in out and "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
    assert "equal the sequential draw" in out
