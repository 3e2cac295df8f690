"""The row text of `minimod freq` without printf (minimod_amd/csrc/fmt_core.h, what the device formatter's kernels are made of) against
glibc's snprintf on the CPU: "%f" of every n_mod / n_called up to a bound, random count pairs over 32 bits, doubles exactly on a rounding
tie and their neighbours, whole TSV / bedmethyl rows (reference src/mod.c:666-719)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fmt_core_equals_snprintf(tmp_path):
    exe = str(tmp_path / "fmt_check")
    subprocess.check_call(["g++", "-O2", "-I", os.path.join(ROOT, "minimod_amd", "csrc"), "-o", exe, os.path.join(ROOT, "tests", "fmt_check.cpp")])
    r = subprocess.run([exe, "700", "1000000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    assert b" 0 mismatches" in r.stdout
