"""Run pytest against another build of liborbfe.so (kernel-variant bisecting): 
   python tools/run_with_lib.py /abs/path/liborbfe_X.so <pytest args>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import orbfe
orbfe.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest
sys.exit(pytest.main(sys.argv[2:]))
