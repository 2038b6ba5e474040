"""CPU: the C-ABI library builds, loads and exports every symbol include/*.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from deflatedmlmc_schwinger_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "schwinger_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sw_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib_built):
    names = declared_symbols()
    assert len(names) >= 40
    lib = ctypes.CDLL(engine.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the Python binding lists the same set
    assert sorted(engine.EXPORTED_SYMBOLS) == names


def test_version_and_device_count(lib_built):
    assert b"gfx950" in lib_built.sw_version()
    assert engine.device_count() >= 0


def test_create_without_gpu_reports_error(lib_built):
    if engine.device_count() > 0:
        return
    h = ctypes.c_void_p()
    assert lib_built.sw_create(ctypes.byref(h), 0) != 0
    assert b"no HIP device" in lib_built.sw_last_error(None)


def test_host_code_under_address_and_ub_sanitizers():
    """SURVEY section 5 (sanitizers): the host-side packers and the MT19937 / GF(2) jump code
    build with -fsanitize=address,undefined and their driver runs clean (`make sanitize`)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no host toolchain")
    csrc = os.path.join(ROOT, "deflatedmlmc_schwinger_amd", "csrc")
    out = subprocess.run(["make", "-C", csrc, "sanitize"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "all checks passed" in out.stdout
