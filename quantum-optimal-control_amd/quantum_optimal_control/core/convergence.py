"""Hyper-parameters of the optimisation loop (reference: core/convergence.py:16-49).

Only the numeric defaults are part of the hot path; the reference's matplotlib/IPython live plots are UI and are
not reproduced.
"""

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

    def record(self, iteration, cost, reg_cost):
        self.iterations.append(iteration)
        self.costs.append(cost)
        self.reg_costs.append(reg_cost)
