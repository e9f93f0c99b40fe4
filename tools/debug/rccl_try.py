"""Debug: does the C-ABI RCCL communicator come up on this box?  argv[1] = 'torch' imports torch first."""
import faulthandler
import os
import sys
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd')]
if len(sys.argv) > 1 and sys.argv[1] == 'torch':
    import torch  # noqa: F401
    print('torch imported', flush=True)
from quantum_optimal_control.core import hip_engine
print('devices', hip_engine.device_count(), flush=True)
os.system("grep -E 'amdhip|rccl|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
uid = hip_engine.comm_unique_id()
print('uid ok', len(uid), flush=True)
os.system("grep -E 'amdhip|rccl|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
comm = hip_engine.QocComm(uid, 1, 0, 0)
print('comm ok', comm.library, flush=True)
import numpy as np
print(comm.all_gather([1.0, 2.0]), comm.all_reduce_max([3.0]), flush=True)
comm.barrier()
comm.close()
print('DONE', flush=True)
