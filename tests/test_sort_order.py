"""bmbs_sort.h reproduces libstdc++ std::sort's permutation for the reference's vote comparator
(Schema.cpp:560-563, 24986) -- the order decides second_best_diff / MAPQ."""
import os
import subprocess

from common import ROOT


def test_intro_sort_matches_std_sort(tmp_path):
    exe = str(tmp_path / "sort_check")
    subprocess.run(["g++", "-O2", "-o", exe, os.path.join(ROOT, "tests", "csrc", "sort_check.cpp")], check=True)
    out = subprocess.run([exe, "20000"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr


def test_data_parallel_partition_matches_std_sort(tmp_path):
    """the formulation the GPU runs (Lpos/Rpos lists, prefix of swaps, stable final pass) gives std::sort's permutation"""
    exe = str(tmp_path / "psort_check")
    subprocess.run(["g++", "-O2", "-o", exe, os.path.join(ROOT, "tests", "csrc", "psort_check.cpp")], check=True)
    out = subprocess.run([exe, "20000"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
