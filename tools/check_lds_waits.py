"""Moved into the package: sonicdiffusionbayeslab_amd/asm_lint.py (build.py runs every rule on every translation unit).
usage kept: python tools/check_lds_waits.py file.s [...]      exit status 1 on a violation"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonicdiffusionbayeslab_amd import asm_lint

if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        rc |= 1 if asm_lint.lint_file(p) else 0
    sys.exit(rc)
