"""The drop-in shown on the reference's own caller (build container only): oracle/check_dropin_on_reference.py constructs the
reference's Qwen3Attention with the `hip` branch of INTEGRATION.md section 2 applied in memory and runs the reference's own
k_cache / v_cache binding loop (engine/model_runner.py:146-157) over it.  Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("NVH_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "nanovllm")), reason="the reference tree is only present in the build container")
def test_reference_constructs_and_binds_the_hip_attention():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MASTER_PORT="29573")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "check_dropin_on_reference.py")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "[dropin] ok" in r.stdout
