"""Read-back of the optimised pulse and propagation (reference: core/analysis.py:8-101).

The reference evaluates TF tensors (`final_state`, `ops_weight`, `inter_vecs`) and converts the real-embedded
results; here the same quantities come from the engine's read-back entry points (qoc_get_final_unitary,
qoc_get_uks, qoc_get_inter_vecs) in complex fp64.  Array shapes returned to the caller and written to the HDF5
run log are the reference's: `final_state` is the real-embedded (2n, 2n) matrix, the `inter_vecs_*` datasets are
(m, n, steps+1).
"""
import numpy as np

from quantum_optimal_control.helper_functions.grape_functions import c_to_r_mat, sort_ev


class Analysis(object):

    def __init__(self, sys_para, engine, seed=0):
        self.sys_para = sys_para
        self.engine = engine
        self.seed = seed            # which control set of a restart batch is reported

    def RtoCMat(self, M):
        """Real-to-complex matrix isomorphism (analysis.py:18-24)."""
        n = self.sys_para.state_num
        M = np.asarray(M)
        return M[:n, :n] + 1j * M[n:2 * n, :n]

    def _log(self):
        from quantum_optimal_control.helper_functions.data_management import H5File
        return H5File(self.sys_para.file_path)

    def get_final_state(self, save=True):
        """Final evolved unitary, n x n complex (analysis.py:26-35)."""
        CMat = np.array(self.engine.get_final_unitary()[self.seed])
        if self.sys_para.save and save:
            with self._log() as hf:
                hf.append('final_state', np.array(c_to_r_mat(CMat)))
        return CMat

    def get_ops_weight(self):
        """sin(ops_weight_base), (k, steps) (analysis.py:37-41); physical amplitudes are maxA_k times this."""
        uks = np.array(self.engine.get_uks()[self.seed])
        return uks / np.asarray(self.sys_para.ops_max_amp, dtype=np.float64)[:, None]

    def get_inter_vecs(self):
        """Populations |<level|psi_j(t)>|^2, shape (m, n, steps+1), dressed basis if the system is dressed
        (analysis.py:44-101)."""
        sp = self.sys_para
        if not sp.use_inter_vecs:
            return None
        raw = np.array(self.engine.get_inter_vecs()[self.seed])        # (steps+1, n, m)
        raw = np.ascontiguousarray(np.transpose(raw, (2, 1, 0)))       # (m, n, steps+1) like tf.stack(inter_vecs)
        if sp.save:
            with self._log() as hf:
                hf.append('inter_vecs_raw_real', np.array(raw.real))
                hf.append('inter_vecs_raw_imag', np.array(raw.imag))
        vecs = raw
        if sp.is_dressed:
            v_sorted = sort_ev(sp.v_c, sp.dressed_id)
            vecs = np.einsum('ab,jat->jbt', v_sorted, raw)             # v_sorted^T psi (analysis.py:74)
        mag_squared = np.square(np.abs(vecs))
        if sp.save:
            with self._log() as hf:
                hf.append('inter_vecs_mag_squared', np.array(mag_squared))
                hf.append('inter_vecs_real', np.array(vecs.real))
                hf.append('inter_vecs_imag', np.array(vecs.imag))
        return list(mag_squared)
