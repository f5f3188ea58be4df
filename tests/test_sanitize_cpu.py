"""CPU: the library's host C++ (context, workspace arena + dry-pass planner, model slots, weight store, graph builders, chunking,
C-ABI error paths) under AddressSanitizer + UndefinedBehaviorSanitizer + LeakSanitizer, on a host-memory stand-in for the HIP runtime
(tools/host_sanitize: sources compiled --cuda-host-only, kernel launches are no-ops).  SURVEY §5 asked for a sanitizer build of the
native host code; GPU sanitizers are not available on the pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("make") is None, reason="needs hipcc + make")
@pytest.mark.skipif(os.environ.get("SVG_SKIP_SANITIZE") == "1", reason="SVG_SKIP_SANITIZE=1")
def test_host_code_is_clean_under_asan_ubsan():
    d = os.path.join(ROOT, "tools", "host_sanitize")
    r = subprocess.run(["make", "-C", d, "-j", str(min(8, os.cpu_count() or 1))], capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "host-sanitize: OK" in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail and "LeakSanitizer" not in tail, tail
