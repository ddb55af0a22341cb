"""tribench.py against another build of the library: tb_lib.py <library file name in slam_amd/lib> [tribench options]"""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[1])
sys.argv = ["tribench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tribench.py"), run_name="__main__")
