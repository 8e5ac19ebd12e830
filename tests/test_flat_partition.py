"""CPU check of the flat-partition index arithmetic the coarse kernel and its launcher share
(rag_project_icd10_amd/csrc/flat_partition.hpp): plain C++, compiled here with g++ and run."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flat_partition_invariants(tmp_path):
    exe = tmp_path / "flat_partition_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "rag_project_icd10_amd", "csrc"),
                    os.path.join(ROOT, "tests", "flat_partition_check.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cases ok" in out.stdout
