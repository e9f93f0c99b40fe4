"""Host-side pre-processing of one GRAPE problem (CPU, NumPy, once per Grape() call).

Same constructor signature and public attributes as the reference's SystemParameters
(core/system_parameters.py:10-284) so code that introspects ``sys_para`` keeps working; written for Python 3 and
organised around what the HIP engine consumes (complex n x n operators) with the reference's real-embedded
arrays (``matrix_list``, ``initial_vectors`` ...) kept as derived views.  Every numeric output is checked bit for
bit against fixtures produced by the reference itself (tests/golden/sysparams_*.npz).
"""
import numpy as np

from quantum_optimal_control.helper_functions.grape_functions import c_to_r_mat, c_to_r_vec, get_state_index, sort_ev

MAX_TAYLOR_TERMS = 20   # system_parameters.py:126
MIN_TAYLOR_TERMS = 3    # system_parameters.py:151
# increments applied to the squaring count between successive candidates (cumulative: s0, +1, +3, +6, +10, +15)
N_CANDIDATES = 6        # system_parameters.py:211-214


def taylor_propagator(M, n_terms, squarings):
    """Order (n_terms - 1) Taylor polynomial of exp(M / 2^squarings), squared `squarings` times (:88-103)."""
    dim = len(M)
    total = np.identity(dim, dtype=M.dtype)
    power = np.identity(dim, dtype=M.dtype)
    fact = 1.0
    for order in range(1, n_terms):
        fact *= order
        power = np.dot(power, M)
        total = total + power / ((2. ** float(order * squarings)) * fact)
    for _ in range(squarings):
        total = np.dot(total, total)
    return total


def taylor_scalar(x, n_terms, squarings):
    """Scalar stand-in for taylor_propagator used when the Hilbert space has >= 10 levels (:105-120)."""
    total, power, fact = 1.0, 1.0, 1.0
    for order in range(1, n_terms):
        fact *= order
        power = x * power
        total += power / ((2. ** float(order * squarings)) * fact)
    for _ in range(squarings):
        total = total * total
    return total


class SystemParameters(object):

    def __init__(self, H0, Hops, Hnames, U, U0, total_time, steps, states_concerned_list, dressed_info, maxA, draw,
                 initial_guess, show_plots, Unitary_error, state_transfer, no_scaling, reg_coeffs, save, file_path,
                 Taylor_terms, use_gpu, use_inter_vecs, sparse_H, sparse_U, sparse_K):
        self.sparse_U, self.sparse_H, self.sparse_K = sparse_U, sparse_H, sparse_K
        self.use_inter_vecs = use_inter_vecs
        self.use_gpu = use_gpu
        self.Taylor_terms = Taylor_terms
        self.dressed_info = dressed_info
        self.reg_coeffs = reg_coeffs
        self.file_path = file_path
        self.state_transfer = state_transfer
        self.no_scaling = no_scaling
        self.save = save
        self.H0_c = H0
        self.ops_c = Hops
        self.ops_max_amp = maxA
        self.Hnames = Hnames
        self.Hnames_original = Hnames
        self.total_time = total_time
        self.steps = steps
        self.show_plots = show_plots
        self.Unitary_error = Unitary_error
        self.states_concerned_list = states_concerned_list
        self.U0_c = U0
        self.draw_list, self.draw_names = (draw[0], draw[1]) if draw is not None else ([], [])

        self._transform_guess(initial_guess)
        self._unpack_dressed(dressed_info)
        self.initial_unitary = c_to_r_mat(U0)
        if not state_transfer:
            self.target_unitary = c_to_r_mat(U)
        else:
            self.target_vectors = [c_to_r_vec(np.asarray(vec)) for vec in U]

        self.dt = float(total_time) / steps
        self.state_num = len(H0)
        self._build_initial_vectors()
        self._build_operators()
        self._build_envelope()
        self._build_guess()
        if save:
            self._log_setup()

    # ------------------------------------------------------------------------------------------------------------
    def _transform_guess(self, initial_guess):
        """Physical amplitudes -> optimisation variable: base = arcsin(u / maxA)   (:38-46)."""
        if initial_guess is None:
            self.u0 = []
            self.u0_base = None
            return
        self.u0 = initial_guess
        ratio = np.zeros_like(np.asarray(initial_guess, dtype=np.float64))
        for row in range(len(ratio)):
            ratio[row] = np.asarray(initial_guess[row], dtype=np.float64) / self.ops_max_amp[row]
            if max(ratio[row]) > 1.0:
                raise ValueError('Initial guess has strength > max_amp for op %d' % (row))
        self.u0_base = np.arcsin(ratio)

    def _unpack_dressed(self, dressed_info):
        self.is_dressed = False
        if dressed_info is not None:
            self.v_c = dressed_info['eigenvectors']
            self.dressed_id = dressed_info['dressed_id']
            self.w_c = dressed_info['eigenvalues']
            self.is_dressed = dressed_info['is_dressed']
            self.H0_diag = np.diag(self.w_c)

    def _build_initial_vectors(self):
        """Columns that get propagated: user vectors, dressed eigenvectors or bare basis states (:168-187)."""
        self.initial_vectors, self.initial_vectors_c = [], []
        for entry in self.states_concerned_list:
            if self.state_transfer:
                vec = np.array(entry)
            elif self.is_dressed:
                vec = self.v_c[:, get_state_index(entry, self.dressed_id)]
            else:
                vec = np.zeros(self.state_num)
                vec[entry] = 1
            self.initial_vector_c = vec
            self.initial_vectors_c.append(vec)
            self.initial_vector = c_to_r_vec(vec)
            self.initial_vectors.append(self.initial_vector)

    # ---- Taylor order / squaring count heuristic (:122-158, :208-227) -----------------------------------------
    def _unitarity_metric(self, H_max, n_terms, U_running):
        if self.state_num < 10:
            step_prop = taylor_propagator((0 - 1j) * self.dt * H_max, n_terms, self.scaling)
            for _ in range(self.steps):
                U_running = np.dot(U_running, step_prop)
            metric = np.abs(np.trace(np.dot(np.conjugate(np.transpose(U_running)), U_running))) / self.state_num
            return metric, U_running
        x = np.max(np.abs(-(0 + 1j) * self.dt * H_max))
        metric = 1 + self.steps * np.abs((taylor_scalar(x, n_terms, self.scaling) - np.exp(x)) / np.exp(x))
        return metric, U_running

    def Choose_exp_terms(self, d):
        """Smallest Taylor term count (searched downwards from 20) that keeps the unitarity metric within
        Unitary_error for the current squaring count; returns the first count that FAILS (reference behaviour)."""
        H_max = self.H0_c
        for amp, op in zip(self.ops_max_amp, self.ops_c):
            H_max = H_max + amp * op
        if d == 0:
            self.scaling = max(int(2 * np.log2(np.max(np.abs(-(0 + 1j) * self.dt * H_max)))), 0)
        else:
            self.scaling += d           # cumulative on purpose: candidates s0, s0+1, s0+3, s0+6, s0+10, s0+15
        if self.state_transfer or self.no_scaling:
            self.scaling = 0
        n_terms = MAX_TAYLOR_TERMS
        U_running = self.U0_c           # NOT reset between trials of n_terms (reference behaviour)
        while True:
            metric, U_running = self._unitarity_metric(H_max, n_terms, U_running)
            if n_terms == MIN_TAYLOR_TERMS or not (np.abs(metric - 1.0) < self.Unitary_error):
                return n_terms
            n_terms -= 1

    def _build_operators(self):
        """-i*dt*H for the drift and every control, complex (engine) and real-embedded (reference layout) (:194-251)."""
        self.ops = [c_to_r_mat(-1j * self.dt * np.asarray(op)) for op in self.ops_c]
        self.ops_len = len(self.ops)
        self.H0 = c_to_r_mat(-1j * self.dt * np.asarray(self.H0_c))
        self.identity_c = np.identity(self.state_num)
        self.identity = c_to_r_mat(self.identity_c)
        # complex stack handed to the HIP engine: [-i dt H0, -i dt H_1, ...]
        self.Hs_c = np.stack([-1j * self.dt * np.asarray(self.H0_c)] +
                             [-1j * self.dt * np.asarray(op) for op in self.ops_c]).astype(np.complex128)

        if self.Taylor_terms is None:
            self.exps, self.scalings = [], []
            n_candidates = 1 if (self.state_transfer or self.no_scaling) else N_CANDIDATES
            for d in range(n_candidates):
                self.exps.append(self.Choose_exp_terms(d))
                self.scalings.append(self.scaling)
            self.complexities = np.add(self.exps, self.scalings)
            best = int(np.argmin(self.complexities))
            self.exp_terms, self.scaling = self.exps[best], self.scalings[best]
        else:
            self.exp_terms, self.scaling = self.Taylor_terms[0], self.Taylor_terms[1]

        print("Using " + str(self.exp_terms) + " Taylor terms and " + str(self.scaling) + " Scaling & Squaring terms")
        self.H_ops = list(self.ops)
        self.matrix_list = np.array([self.H0] + self.H_ops + [np.eye(2 * self.state_num).tolist()])

    def _build_envelope(self):
        """1 - Gaussian over [-2, 2], clipped at 0, lifted by 0.01; one row per control (:253-270)."""
        grid = np.linspace(-2, 2, self.steps)
        shape = np.ones(self.steps) - self.gaussian(grid) - 0.0
        shape = shape * (shape > 0)
        shape = shape + 0.01 * np.ones(self.steps)
        self.one_minus_gauss = np.array([shape for _ in range(self.ops_len)])

    @staticmethod
    def gaussian(x, mu=0., sig=1.):
        return np.exp(-np.power(x - mu, 2.) / (2 * np.power(sig, 2.)))

    def _build_guess(self):
        """Initial optimisation variable: transformed user guess or N(0, 1/sqrt(steps)) from NumPy's global RNG (:272-284)."""
        if self.u0_base is not None:
            self.ops_weight_base = np.reshape(self.u0_base, [self.ops_len, self.steps])
        else:
            self.ops_weight_base = np.random.normal(0, 1. / np.sqrt(self.steps), [self.ops_len, self.steps])
        self.raw_shape = np.shape(self.ops_weight_base)

    def _log_setup(self):
        from quantum_optimal_control.helper_functions.data_management import H5File
        with H5File(self.file_path) as hf:
            hf.add('initial_vectors_c', data=np.array(self.initial_vectors_c))
            hf.add('taylor_terms', data=self.exp_terms)
            hf.add('taylor_scaling', data=self.scaling)

    # ---- views consumed by the HIP engine ----------------------------------------------------------------------
    def engine_inputs(self):
        """(Hs, U0, V, W, Vs) as complex arrays in the engine's layout."""
        n = self.state_num
        V = np.stack([np.asarray(v, dtype=np.complex128) for v in self.initial_vectors_c], axis=1)
        U0 = np.asarray(self.U0_c, dtype=np.complex128)
        if self.state_transfer:
            W = np.stack([tv[:n] + 1j * tv[n:] for tv in self.target_vectors], axis=1)
        else:
            U = self.target_unitary[:n, :n] + 1j * self.target_unitary[n:, :n]
            W = U @ V                                               # tensorflow_state.py:164
        Vs = None
        rc = self.reg_coeffs or {}
        if self.is_dressed and rc.get('forbid_dressed', False) and 'forbidden_coeff_list' in rc:
            Vs = np.asarray(sort_ev(self.v_c, self.dressed_id), dtype=np.complex128).reshape(n, n)
        return self.Hs_c, U0, V, W, Vs
