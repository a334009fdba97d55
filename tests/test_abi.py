"""The C-ABI library loads and exports every symbol include/dgll_hip.h declares (no compute: runs without a GPU)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "dgll_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(dgll_hip_\w+|dgll_host_\w+|launch_gcn_fused_kernel\w*)\s*\(", text)
    assert names, "no declarations parsed"
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from dgll_amd import _lib

    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), "libdgll_hip.so does not export %s" % name


def test_python_binding_covers_the_header():
    from dgll_amd import _lib

    assert sorted(_lib.SIGNATURES) == _declared_functions()


def test_abi_version_and_error_string():
    from dgll_amd import _lib

    assert _lib.lib.dgll_hip_abi_version() == 1
    assert isinstance(_lib.last_error(), str)


def test_argument_validation_needs_no_gpu():
    """Bad arguments are rejected on the host with an error code and text (the reference exit(1)s)."""
    from dgll_amd import _lib

    code = _lib.lib.dgll_hip_spmm_csr(None, None, None, None, None, None, 4, 0, None, 4, 0, 10, 10, 8, 0, 0, None, None, 0)
    assert code == -1 and "NULL" in _lib.last_error()
    code = _lib.lib.dgll_hip_spmm_csr(None, None, 16, 16, None, 16, 4, 7, 16, 4, 0, 10, 10, 4, 0, 0, None, None, 0)
    assert code == -1 and "x_dtype" in _lib.last_error()


def test_library_is_in_tree():
    from dgll_amd import _lib

    assert os.path.realpath(_lib.LIB_PATH).startswith(os.path.realpath(ROOT))


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch

    import dgll_amd
    from dgll_amd import ops

    g = dgll_amd.CSRGraph.from_coo(torch.tensor([0, 1]), torch.tensor([1, 0]), None, (2, 2))
    with pytest.raises(RuntimeError):
        ops.spmm(g, torch.ones(2, 4))


def _compile_c_client(tmp_path):
    import shutil
    import subprocess

    from dgll_amd import _lib

    exe = str(tmp_path / "spmm_smoke")
    libdir = os.path.dirname(_lib.LIB_PATH)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [shutil.which("gcc") or "gcc", "-std=c99", "-Wall", os.path.join(ROOT, "tests", "c_abi", "spmm_smoke.c"),
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"), "-D__HIP_PLATFORM_AMD__",
           "-L", libdir, "-ldgll_hip", "-L", os.path.join(rocm, "lib"), "-lamdhip64",
           "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(rocm, "lib"), "-lm", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """include/dgll_hip.h compiles as C99 and a C program links against libdgll_hip.so (no Python, no torch types)."""
    import subprocess

    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "dgll_hip.h")],
                   check=True, capture_output=True)
    assert os.path.exists(_compile_c_client(tmp_path))


import pytest  # noqa: E402


@pytest.mark.gpu
def test_c_client_runs_on_the_gpu(tmp_path):
    import subprocess

    res = subprocess.run([_compile_c_client(tmp_path)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "c abi smoke ok" in res.stdout
