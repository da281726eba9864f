"""The measurement tools only run on the GPU box; here they are at least compiled, so a broken edit shows up in the CPU suite
(one cost a profile collection this round)."""
import glob
import os
import py_compile
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_tool_compiles():
    tools = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))
    assert len(tools) >= 10
    for path in tools + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]:
        py_compile.compile(path, doraise=True)


def test_shell_tools_parse():
    for path in sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh"))):
        assert subprocess.run(["bash", "-n", path]).returncode == 0, path
