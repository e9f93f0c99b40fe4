"""Row f3: HDF5 run log + Analysis read-back.  h5py is optional; when the interpreter running pytest lacks it
the bodies (tests/h5_scripts.py) run under the image's conda python3.9, which has h5py, numpy and scipy -- the product path
needs nothing else (ctypes over libqoc_hip.so)."""
import importlib.util
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CONDA = '/opt/conda/bin/python3.9'


def _run(func, tmp_path):
    if importlib.util.find_spec('h5py') is not None:
        exe = sys.executable
    elif os.path.exists(CONDA):
        exe = CONDA
    else:
        pytest.skip('no interpreter with h5py available')
    env = dict(os.environ)
    sys_cxx = '/usr/lib/x86_64-linux-gnu/libstdc++.so.6'
    if exe == CONDA and os.path.exists(sys_cxx):
        env['LD_PRELOAD'] = sys_cxx       # conda ships an older libstdc++ than libamdhip64 needs; take the system one
    r = subprocess.run([exe, '-W', 'ignore', os.path.join(HERE, 'h5_scripts.py'), func, str(tmp_path)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and ('OK ' + func) in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_run_log_add_append(tmp_path):
    _run('run_log', tmp_path)


def test_analysis_datasets(tmp_path):
    _run('analysis_log', tmp_path)


@pytest.mark.gpu
def test_grape_save_and_resimulation(tmp_path):
    _run('grape_save', tmp_path)
