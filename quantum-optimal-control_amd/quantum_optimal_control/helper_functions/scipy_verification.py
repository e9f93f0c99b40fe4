"""Independent re-simulation of a saved run (reference: helper_functions/qutip_verification.py:5-88).

The reference integrates the Schroedinger equation with QuTiP's sesolve and compares with the logged
`inter_vecs_raw_*`.  QuTiP is not a dependency here; for piecewise-constant controls the exact propagator of a
slice is expm(-i dt H_t), so the same check is done with scipy.linalg.expm.  Same inputs (run log path, atol), same
printed summary; also returns (max_abs_diff_list, all_close_list).
"""
import numpy as np
from scipy.linalg import expm


def scipy_verification(datafile, atol):
    import h5py
    with h5py.File(datafile, 'r') as hf:
        gate_time = float(np.array(hf.get('total_time')))
        gate_steps = int(np.array(hf.get('steps')))
        H0 = np.array(hf.get('H0'))
        Hops = np.array(hf.get('Hops'))
        initial_vectors_c = np.array(hf.get('initial_vectors_c'))
        uks = np.array(hf.get('uks'))[-1]
        raw = np.array(hf.get('inter_vecs_raw_real'))[-1] + 1j * np.array(hf.get('inter_vecs_raw_imag'))[-1]
    dt = gate_time / gate_steps
    max_abs_diff_list, all_close_list = [], []
    props = []
    for t in range(gate_steps):
        H = H0.astype(np.complex128)
        for kk in range(len(Hops)):
            H = H + uks[kk, t] * Hops[kk]
        props.append(expm(-1j * dt * H))
    for vec_id in range(len(initial_vectors_c)):
        print("Verifying init vector id: %d" % vec_id)
        psi = np.asarray(initial_vectors_c[vec_id], dtype=np.complex128).reshape(-1)
        traj = [psi]
        for K in props:
            psi = K @ psi
            traj.append(psi)
        traj = np.transpose(np.array(traj))                     # (n, steps+1)
        max_abs_diff_list.append(float(np.max(np.abs(traj) - np.abs(raw[vec_id]))))
        all_close_list.append(bool(np.allclose(traj, raw[vec_id], atol=atol)))
    print("SciPy simulation verification result for each initial state")
    print("================================================")
    print("max abs diff: " + str(max_abs_diff_list))
    print("all close: " + str(all_close_list))
    print("================================================")
    return max_abs_diff_list, all_close_list


qutip_verification = scipy_verification      # drop-in name for scripts written against the reference helper
