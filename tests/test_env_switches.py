"""CPU suite: docs/ENVIRONMENT.md lists every CL_* environment switch the library reads (scripts/env_switches.py regenerates it)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_environment_table_is_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "env_switches.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
