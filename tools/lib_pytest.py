"""Diagnostic: run pytest against an alternative build of the library: python tools/lib_pytest.py <lib.so> [pytest args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from azalea_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[1])
import pytest
sys.exit(pytest.main(sys.argv[2:]))
