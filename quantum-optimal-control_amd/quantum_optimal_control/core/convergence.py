"""Hyper-parameters of the optimisation loop and the progress figure (reference: core/convergence.py:16-222).

The numeric defaults are part of the hot path.  The reference redraws a matplotlib summary inside a Jupyter notebook at
every update step (`display.display` + `clear_output`, :121-222); `plot_summary` below draws the same panels -- error
curves, final operator, pulses, populations of the concerned states with the forbidden-level sum -- from the engine's
read-back.  It is shown live only when IPython is importable and a kernel is running (the reference's own condition of
use); scripts get the progress line, and can ask for a file with `plot_summary(..., filename=...)`.
"""
import time

import numpy as np

DEFAULTS = (('rate', 0.01), ('update_step', 100), ('evol_save_step', 100), ('conv_target', 1e-8),
            ('max_iterations', 5000), ('learning_rate_decay', 2500), ('min_grad', 1e-25))


class Convergence(object):
    def __init__(self, sys_para, time_unit, convergence):
        self.sys_para = sys_para
        self.time_unit = time_unit
        for key, default in DEFAULTS:
            setattr(self, key, convergence[key] if key in convergence else default)
        self.reset_convergence()

    def reset_convergence(self):
        self.costs = []
        self.reg_costs = []
        self.iterations = []
        self.learning_rate = []
        self.last_iter = 0
        self.accumulate_rate = 1.00

    def save_evol(self, anly):
        """Snapshot of the propagation at this point of the run (convergence.py:62-66); with save=True the Analysis
        calls append final_state / inter_vecs_* to the run log."""
        if not self.sys_para.state_transfer:
            self.final_state = anly.get_final_state()
        self.inter_vecs = anly.get_inter_vecs()

    # ---- progress figure (UI) -----------------------------------------------------------------------------------------
    @staticmethod
    def in_notebook():
        try:
            from IPython import get_ipython
            return get_ipython() is not None
        except ImportError:
            return False

    def plot_inter_vecs_general(self, plt, pop_inter_vecs, start):
        """Populations of one propagated state over time (convergence.py:85-118)."""
        sp = self.sys_para
        tlist = sp.dt * np.arange(sp.steps + 1)
        draw_list = getattr(sp, 'draw_list', [])
        if len(draw_list) > 0:
            for kk, level in enumerate(draw_list):
                plt.plot(tlist, np.array(pop_inter_vecs[level, :]), label=sp.draw_names[kk])
        else:
            if start > 4:
                plt.plot(tlist, np.array(pop_inter_vecs[start, :]), label='Starting level ' + str(start))
            for jj in range(min(4, pop_inter_vecs.shape[0])):
                plt.plot(tlist, np.array(pop_inter_vecs[jj, :]), label='level ' + str(jj))
        rc = sp.reg_coeffs or {}
        if 'states_forbidden_list' in rc:
            forbidden = np.zeros(sp.steps + 1)
            for forbid in rc['states_forbidden_list']:
                if sp.dressed_info is None or rc.get('forbid_dressed', False):
                    forbidden = forbidden + np.array(pop_inter_vecs[forbid, :])
                else:
                    from quantum_optimal_control.helper_functions.grape_functions import sort_ev
                    dressed = np.dot(sort_ev(sp.v_c, sp.dressed_id), np.sqrt(pop_inter_vecs))
                    forbidden = forbidden + np.square(np.abs(dressed[forbid, :]))
            plt.plot(tlist, forbidden, label='forbidden', linestyle='--', linewidth=4)
        plt.ylabel('Population')
        plt.ylim(-0.1, 1.1)
        plt.xlabel('Time (' + self.time_unit + ')')
        plt.legend(ncol=7)

    def plot_summary(self, last_cost, last_reg_cost, anly, unitary_metric=float('nan'), filename=None):
        """Draw the reference's summary figure (convergence.py:121-222).  Returns the matplotlib figure."""
        import matplotlib
        if filename is not None and not self.in_notebook():
            matplotlib.use('Agg')
        import matplotlib.pyplot as plt
        from matplotlib import gridspec
        sp = self.sys_para
        concerned = list(sp.states_concerned_list)
        if not hasattr(self, 'start_time'):
            self.start_time = time.time()
        runtime = time.time() - self.start_time
        last_iter = self.iterations[-1] if self.iterations else 0
        remaining_h = runtime * (self.max_iterations - last_iter) / last_iter / 3600.0 if last_iter else 0.0
        self.save_evol(anly)
        rows = 3 + len(concerned) - (1 if sp.state_transfer else 0)
        fig = plt.figure()
        gs = gridspec.GridSpec(rows, 2)
        index = 0
        plt.subplot(gs[index, :], title='Error = %1.2e; Other errors = %1.2e; Unitary Metric: %.5f; Runtime: %.1fs; '
                                        'Estimated Remaining Runtime: %.1fh' % (last_cost, last_reg_cost - last_cost,
                                                                                unitary_metric, runtime, remaining_h))
        index += 1
        plt.plot(np.array(self.iterations), np.array(self.costs), 'bx-', label='Fidelity Error')
        plt.plot(np.array(self.iterations), np.array(self.reg_costs), 'go-', label='All Penalties')
        plt.ylabel('Error')
        plt.xlabel('Iteration')
        if len(self.costs) and min(np.min(self.costs), np.min(self.reg_costs)) > 0:
            plt.yscale('log')
        plt.legend()
        if not sp.state_transfer:
            M = np.asarray(self.final_state)
            for col, (part, name) in enumerate(((M.real, 'real'), (M.imag, 'imaginary'))):
                plt.subplot(gs[index, col], title='operator: ' + name)
                plt.imshow(part, interpolation='none')
                plt.clim(-1, 1)
                plt.colorbar()
            index += 1
        plt.subplot(gs[index, :], title='Optimized pulse')
        ops_weight = anly.get_ops_weight()
        tlist = sp.dt * np.arange(sp.steps)
        for jj in range(sp.ops_len):
            plt.plot(tlist, sp.ops_max_amp[jj] * ops_weight[jj, :], label='u' + str(sp.Hnames[jj]))
        plt.ylabel('Amplitude')
        plt.xlabel('Time (' + self.time_unit + ')')
        plt.legend()
        index += 1
        if sp.use_inter_vecs and self.inter_vecs is not None:
            for ii in range(len(concerned)):
                plt.subplot(gs[index + ii, :], title='Evolution')
                start = concerned[ii] if np.isscalar(concerned[ii]) else 0
                self.plot_inter_vecs_general(plt, np.asarray(self.inter_vecs[ii]), start)
        fig.set_size_inches(15, 4 * rows)
        if filename is not None:
            fig.savefig(filename, dpi=60)
        if self.in_notebook():
            from IPython import display
            display.display(fig)
            display.clear_output(wait=True)
        plt.close(fig)
        return fig

    def record(self, iteration, cost, reg_cost):
        self.iterations.append(iteration)
        self.costs.append(cost)
        self.reg_costs.append(reg_cost)
