"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side code: tools/sanitize.sh builds the oracle, the C++ host
mirror, the host-compiled device headers, the C++ tile exchange and the host BVH builders with -fsanitize=address,undefined in a
scratch copy of the repo and runs the CPU test-suite on them.  Never on the GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_code_is_sanitizer_clean(tmp_path):
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("this gcc has no libasan")
    env = {**os.environ, "PT_SANITIZE_DIR": str(tmp_path / "copy")}
    env.pop("LD_PRELOAD", None)
    res = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize.sh")], capture_output=True, text=True, env=env, timeout=900)
    shutil.rmtree(env["PT_SANITIZE_DIR"], ignore_errors=True)
    assert res.returncode == 0, (res.stdout[-3000:], res.stderr[-3000:])
    assert "sanitize: clean" in res.stdout and "builder cases, 0 failed" in res.stdout
