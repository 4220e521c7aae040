"""Runs the CPU-oracle test files against oracle/libgs_oracle_asan.so (`make -C oracle asan`) - ASan + UBSan.
  LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tests/tools/run_oracle_asan.py
Round 1: 36 tests, clean."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402

oracle_lib.ORACLE_SO = os.path.join(ROOT, "oracle", "libgs_oracle_asan.so")
oracle_lib.build = lambda: None
import pytest  # noqa: E402

T = os.path.join(ROOT, "tests")
sys.exit(pytest.main(["-x", "-q", "-p", "no:cacheprovider", "-k", "not saturated"] +
                     [os.path.join(T, f) for f in ("test_oracle_dense.py", "test_oracle_knn.py", "test_oracle_loss.py",
                                                   "test_fsgs_cpu.py", "test_api.py", "test_densify_cpu.py")]))
