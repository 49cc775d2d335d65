import faulthandler, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
log = open("gpurun_out/c5_dbg2.log", "a", buffering=1)
faulthandler.enable(file=log)
faulthandler.dump_traceback_later(float(sys.argv[2]) if len(sys.argv) > 2 else 90, exit=True, file=log)
from tests import inproc, test_gpu_config5 as T
name = sys.argv[1]
case = T.Case(**T.CASES[name])
ref = T._reference(case)
print("reference done", name, file=log)
print(inproc.run_threads(8, T._operator_worker, case, ref, timeout=80, wire_timeout=30), file=log)
print("ALL OK", name, file=log)
