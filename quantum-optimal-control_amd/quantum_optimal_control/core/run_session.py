"""Optimiser drivers over the HIP engine (reference: core/run_session.py:11-199).

ADAM        device-resident loop (stop rule, LR schedule and Adam update run on the GPU; the host polls every
            update_step iterations to print the reference's progress line)
EVOLVE      one evaluation at the initial controls
anything else is handed to scipy.optimize.minimize with (reg_loss, gradient) from the engine, as the reference does.
"""
import time

import numpy as np
from scipy.optimize import minimize

from quantum_optimal_control.core.analysis import Analysis


class run_session(object):

    def __init__(self, tfs, graph, conv, sys_para, method, show_plots=True, single_simulation=False, use_gpu=True):
        self.tfs = tfs
        self.graph = graph
        self.engine = tfs.engine
        self.conv = conv
        self.sys_para = sys_para
        self.update_step = conv.update_step
        self.iterations = 0
        self.method = method.upper()
        self.show_plots = show_plots
        self.target = False
        self.end = False
        self.seed = 0          # which control set the scalar attributes (l, rl, ...) report
        print("Initialized")
        if self.method == 'EVOLVE':
            self.start_time = time.time()
            self.l, self.rl, self.grads, self.metric, self.g_squared = self.get_error(self.sys_para.ops_weight_base)
            self.get_end_results()
        elif self.method == 'ADAM':
            self.start_adam_optimizer()
        else:
            self.bfgs_optimize(method=self.method)

    # ---- Adam ---------------------------------------------------------------------------------------------------
    def start_adam_optimizer(self):
        eng, conv = self.engine, self.conv
        self.start_time = time.time()
        params = eng.adam_params(rate=conv.rate, learning_rate_decay=conv.learning_rate_decay,
                                 conv_target=conv.conv_target, min_grad=conv.min_grad,
                                 max_iterations=conv.max_iterations, poll_every=max(1, int(conv.update_step)))
        budget = int(conv.max_iterations) + 1          # evaluations: one per update + the one that trips the stop rule
        u, ev = max(1, int(conv.update_step)), max(1, int(conv.evol_save_step))
        launched = 0                                   # evaluations enqueued so far; the last one has index launched - 1
        while True:
            # run to the next evaluation whose index is a multiple of update_step OR of evol_save_step: the reference logs
            # at both, independently (run_session.py:75-91); evaluation 0 is both
            it = launched
            while it % u != 0 and it % ev != 0:
                it += 1
            n = min(it + 1, budget) - launched
            eng.iterate(params, n)
            launched += n
            s = eng.scalars()
            self._take_scalars(s)
            self.end = bool(np.all(s['done']))
            if self.end or launched >= budget:
                break
            self.update_and_save(launched - 1)
        self.end = True
        self.get_end_results()

    def _take_scalars(self, s, replace_last=False):
        b = self.seed
        self.l, self.rl = float(s['loss'][b]), float(s['reg_loss'][b])
        self.g_squared, self.metric = float(s['grad_squared'][b]), float(s['unitary_scale'][b])
        if 'iterations' in s:
            # the device counter is already one past the evaluation these scalars belong to, unless that evaluation
            # tripped the stop rule (run_session.py:53-60 evaluates, checks, then increments)
            self.iterations = int(s['iterations'][b]) - (0 if int(s['done'][b]) else 1)
        if replace_last and self.conv.iterations:
            self.conv.iterations[-1], self.conv.costs[-1], self.conv.reg_costs[-1] = self.iterations, self.l, self.rl
        else:
            self.conv.record(self.iterations, self.l, self.rl)

    def update_and_save(self, it=None):
        """Host stop at evaluation `it` (run_session.py:75-91): at multiples of update_step the progress line and a run-log
        row; at multiples of evol_save_step the propagation snapshot (plus a run-log row if update_step did not write one),
        unless the live figure of this same stop already took it."""
        it = self.iterations if it is None else it
        upd = it % max(1, int(self.conv.update_step)) == 0
        evo = it % max(1, int(self.conv.evol_save_step)) == 0
        self.anly = Analysis(self.sys_para, self.engine, self.seed)
        plotted = False
        if upd:
            self.save_data()
            self.display()
            if self.show_plots and self.conv.in_notebook():
                self.conv.plot_summary(self.l, self.rl, self.anly, self.metric)   # redraws the live figure (and snapshots)
                plotted = True
        if evo and not plotted:
            if not upd:
                self.save_data()
            self.conv.save_evol(self.anly)

    # ---- results ------------------------------------------------------------------------------------------------
    def get_end_results(self):
        if self.engine.n_seeds > 1:                       # report / return the best restart
            s = self.engine.scalars()
            self.seed = int(np.argmin(s['loss']))
            self._take_scalars(s, replace_last=True)      # the final point was recorded for seed 0: report the best restart instead
        self.anly = Analysis(self.sys_para, self.engine, self.seed)
        self.save_data()
        self.display()
        self.conv.save_evol(self.anly)                    # final_state / inter_vecs_* rows of the run log (:100-101)
        self.uks = self.Get_uks()
        if not self.sys_para.state_transfer:
            self.Uf = self.anly.get_final_state(save=False)
        else:
            self.Uf = []

    def Get_uks(self):
        """Physical pulse amplitudes maxA_k sin(base), (k, steps) (run_session.py:112-117): the controls the reported
        loss / final_state / inter_vecs were evaluated on."""
        return self.engine.get_uks(evaluated=True)[self.seed]

    def get_error(self, uks):
        """Loss, regularised loss, flattened gradient, unitary metric and grad_squared at controls `uks`
        (the variable is the pre-sin base, exactly as in the reference where get_error assigns ops_weight_base)."""
        eng = self.engine
        base = np.broadcast_to(np.reshape(np.asarray(uks, dtype=np.float64), (eng.k, eng.steps)),
                               (eng.n_seeds, eng.k, eng.steps))
        adam_state_reset = base      # set_base resets the optimiser slots, irrelevant for scipy drivers
        eng.set_base(adam_state_reset)
        r = eng.evaluate(want_grad=True)
        b = self.seed
        g = np.reshape(r['grad'][b], (eng.k * eng.steps))
        return float(r['loss'][b]), float(r['reg_loss'][b]), g, float(r['unitary_scale'][b]), float(r['grad_squared'][b])

    def save_data(self):
        if self.sys_para.save:
            from quantum_optimal_control.helper_functions.data_management import H5File
            self.elapsed = time.time() - self.start_time
            with H5File(self.sys_para.file_path) as hf:
                hf.append('error', np.array(self.l))
                hf.append('reg_error', np.array(self.rl))
                hf.append('uks', np.array(self.Get_uks()))
                hf.append('iteration', np.array(self.iterations))
                hf.append('run_time', np.array(self.elapsed))
                hf.append('unitary_scale', np.array(self.metric))

    def display(self):
        self.elapsed = time.time() - self.start_time
        print('Error = :%1.2e; Runtime: %.1fs; Iterations = %d, grads =  %10.3e, unitary_metric = %.5f' % (
            self.l, self.elapsed, self.iterations, self.g_squared, self.metric))

    # ---- scipy drivers ------------------------------------------------------------------------------------------
    def minimize_opt_fun(self, x):
        k = len(self.sys_para.ops_c)
        self.l, self.rl, self.grads, self.metric, self.g_squared = self.get_error(np.reshape(x, (k, len(x) // k)))
        if self.l < self.conv.conv_target:
            self.conv_time = time.time() - self.start_time
            self.conv_iter = self.iterations
            self.end = True
            print('Target fidelity reached')
            self.grads = 0 * self.grads          # zero gradient terminates the scipy optimisation
        if not self.end:
            if self.iterations % self.conv.update_step == 0 or self.iterations % self.conv.evol_save_step == 0:
                self.update_and_save()
            self.iterations += 1
        return np.float64(self.rl), np.asarray(self.grads, dtype=np.float64)

    def bfgs_optimize(self, method='L-BFGS-B', jac=True, options=None):
        self.conv.reset_convergence()
        self.first = True
        self.conv_time = 0.
        self.conv_iter = 0
        self.end = False
        print("Starting " + self.method + " Optimization")
        self.start_time = time.time()
        x0 = np.reshape(self.sys_para.ops_weight_base, -1)
        if method == 'L-BFGS-B':
            options = {'maxfun': self.conv.max_iterations, 'gtol': self.conv.min_grad, 'disp': False, 'maxls': 40}
        else:
            options = {'gtol': self.conv.min_grad, 'disp': False, 'maxiter': self.conv.max_iterations}
        res = minimize(self.minimize_opt_fun, x0, method=method, jac=jac, options=options)
        k = len(self.sys_para.ops_c)
        self.l, self.rl, self.grads, self.metric, self.g_squared = self.get_error(np.reshape(res['x'], (k, len(res['x']) // k)))
        print(self.method + ' optimization done')
        if not self.sys_para.show_plots:
            print(res.message)
            print("Error = %1.2e" % self.l)
            print("Total time is " + str(time.time() - self.start_time))
        self.get_end_results()
