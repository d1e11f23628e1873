"""The drop-in boundary from a compiled host: tests/c_host/c_abi_host.c (plain C, gcc, only include/rabitq_hip.h and
librabitq_hip.so) builds, dumps, reloads and queries an index and checks single == batch == reloaded itself."""
import subprocess

import pytest

from tests.test_abi import _build_c_host

pytestmark = pytest.mark.gpu


def test_c_host_program(tmp_path):
    exe = _build_c_host(tmp_path)
    p = subprocess.run([exe, str(tmp_path / "idx")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    assert "batch == single == reloaded: ok" in p.stdout
