"""python scripts/bench_lib.py libadmpc_X.so [bench.py arguments]: bench.py against another build of the library in ad_mpc_amd/."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ad_mpc_amd._lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "ad_mpc_amd", sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
