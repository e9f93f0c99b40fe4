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


# ----------------------------------------------------------------------------------------------------------------
# Caller-side builders (reference helper_functions/grape_functions.py:26-191).  Independent
# re-implementations whose outputs are pinned to the reference by tests/golden/helpers.npz.
# ----------------------------------------------------------------------------------------------------------------

def qft(N):
    """Quantum Fourier transform on N qubits: F[a, b] = exp(2 pi i a b / 2^N) / sqrt(2^N)."""
    dim = 2 ** N
    idx = np.arange(dim)
    return np.exp(2.0j * np.pi / dim * np.outer(idx, idx)) / np.sqrt(dim)


def hamming_distance(x):
    """Number of set bits of the non-negative integer x."""
    return bin(int(x)).count('1')


def Hadamard(N=1):
    """N-qubit Hadamard: entries (-1)^{popcount(i & j)} / 2^{N/2}."""
    dim = 2 ** N
    signs = np.array([[(-1) ** hamming_distance(i & j) for i in range(dim)] for j in range(dim)])
    return (2.0 ** (-N / 2.0)) * signs


def baseN(num, b, numerals="0123456789abcdefghijklmnopqrstuvwxyz"):
    """Digits of num in base b as a string (no leading zeros; '0' for zero)."""
    if num == 0:
        return numerals[0]
    digits = []
    while num:
        num, rem = divmod(num, b)
        digits.append(numerals[rem])
    return ''.join(reversed(digits))


def Basis(a, N, r):
    """Base-r representation of a, left-padded with zeros to N digits."""
    return baseN(a, r).rjust(N, '0')


def Bin(a, N):
    """Binary representation of a, left-padded to N digits."""
    return np.binary_repr(a).rjust(N, '0')


def is_binary(num):
    """True when every digit of the string is 0 or 1."""
    return all(ch in '01' for ch in num)


def concerned(N, levels):
    """Indices of the computational (qubit) subspace of N `levels`-level systems."""
    return [idx for idx in range(levels ** N) if is_binary(Basis(idx, N, levels))]


def transmon_gate(gate, levels):
    """Embed a qubit gate into N multi-level transmons: identity outside the qubit subspace."""
    N = int(np.log2(len(gate)))
    dim = levels ** N
    out = np.identity(dim, dtype=complex)
    labels = [Basis(idx, N, levels) for idx in range(dim)]
    qubit_like = [idx for idx in range(dim) if is_binary(labels[idx])]
    for row in qubit_like:
        for col in qubit_like:
            out[row, col] = gate[int(labels[row], 2), int(labels[col], 2)]
    return out


def rz(theta):
    return [[np.exp(-1j * theta / 2), 0], [0, np.exp(1j * theta / 2)]]


def rx(theta):
    c, s = np.cos(theta / 2), np.sin(theta / 2)
    return [[c, -1j * s], [-1j * s, c]]


def multi_kron(op, num):
    """op (x) op (x) ... (num factors)."""
    out = op
    for _ in range(num - 1):
        out = np.kron(out, op)
    return out


def kron_all(op, num, op_2):
    """Kronecker strings with `op` moving through `op_2` factors.  NOTE: like the reference (:107-126) this returns
    the LAST string built, not the accumulated sum -- callers depend on the value, so the behaviour is kept."""
    last = op
    for lead in range(num):
        last = op if lead == 0 else op_2
        for pos in range(num - 1):
            last = np.kron(last, op if (lead - pos) == 1 else op_2)
    return last


def append_separate_krons(op, name, num, state_num, Hops, Hnames, ops_max_amp, amp=4.0):
    """Append op acting on each of `num` subsystems separately (op (x) I (x) I, I (x) op (x) I, ...), with names
    and amplitudes.  The naming follows the reference (:135-163): first string name+'i'*(num-1), then 'i'...name...'i'."""
    eye = np.identity(state_num)
    first = op
    for _ in range(num - 1):
        first = np.kron(first, eye)
    Hops.append(first)
    ops_max_amp.append(amp)
    Hnames.append(name + 'i' * (num - 1))
    for site in range(1, num):
        term, label = eye, 'i'
        for pos in range(1, num):
            if pos == site:
                term, label = np.kron(term, op), label + name
            else:
                term, label = np.kron(term, eye), label + 'i'
        Hops.append(term)
        ops_max_amp.append(amp)
        Hnames.append(label)
    return Hops, Hnames, ops_max_amp


def nn_chain_kron(op, op_I, qubit_num, qubit_state_num):
    """Nearest-neighbour coupling chain: op(x)op(x)I..I + I(x)op(x)op(x)I.. + ... (qubit_num - 1 terms)."""
    dim = qubit_state_num ** qubit_num
    total = np.zeros([dim, dim])
    for left in range(qubit_num - 1):
        term = None
        for site in range(qubit_num):
            factor = op if site in (left, left + 1) else op_I
            term = factor if term is None else np.kron(term, factor)
        total = total + term
    return total
