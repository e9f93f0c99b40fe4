"""NumPy helpers used on the host side of the path (complex<->real embedding, dressed-state bookkeeping).

Mirrors the parts of the reference's helper_functions/grape_functions.py that the hot path touches
(c_to_r_mat :211-213, c_to_r_vec :215-220, sort_ev :194-202, get_state_index :204-209, get_dressed_info :9-24);
outputs are checked against fixtures captured from the reference (tests/golden/helpers.npz).
"""
import numpy as np
import scipy.linalg as la


def c_to_r_mat(M):
    """Real 2n x 2n image [[Re, -Im], [Im, Re]] of a complex matrix."""
    M = np.asarray(M)
    top = np.concatenate([M.real, -M.imag], axis=1)
    bottom = np.concatenate([M.imag, M.real], axis=1)
    return np.concatenate([top, bottom], axis=0)


def c_to_r_vec(V):
    """Real 2n image [Re; Im] of a complex vector."""
    V = np.asarray(V)
    return np.reshape([V.real, V.imag], [2 * len(V)])


def r_to_c_mat(M, state_num):
    """Inverse of c_to_r_mat as Analysis.RtoCMat reads it (top-left + i * bottom-left block)."""
    return M[:state_num, :state_num] + 1j * M[state_num:2 * state_num, :state_num]


def get_state_index(bareindex, dressed_id):
    """Position of the dressed state that overlaps most with bare state `bareindex`."""
    if len(dressed_id) > 0:
        return list(dressed_id).index(bareindex)
    return bareindex


def sort_ev(v, dressed_id):
    """Eigenvector matrix with column i = dressed partner of bare state i."""
    count = len(dressed_id)
    picked = [v[:, get_state_index(bare, dressed_id)] for bare in range(count)]
    return np.transpose(np.reshape(picked, [count, count]))


def get_dressed_info(H0):
    """Eigen-decomposition of H0 plus a greedy one-to-one assignment dressed -> bare index by largest overlap."""
    w_c, v_c = la.eig(H0)
    dressed_id = []
    for col in range(len(v_c)):
        weights = np.abs(v_c[:, col])
        index = int(np.argmax(weights))
        if index in dressed_id:
            remaining = weights.tolist()
            while index in dressed_id:
                remaining[index] = 0
                index = int(np.argmax(remaining))
        dressed_id.append(index)
    return w_c, v_c, dressed_id


def dressed_unitary(U, v, dressed_id):
    """U expressed in the dressed basis."""
    conv = sort_ev(v, dressed_id)
    return np.dot(np.dot(conv, U), np.conjugate(np.transpose(conv)))
