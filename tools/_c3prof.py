import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/quantum-optimal-control_amd')
sys.argv = ['x']
from tools import bench_configs as bc
from tests.golden import cases
bc.run('C3', cases.case_c3(), 1, 20)
