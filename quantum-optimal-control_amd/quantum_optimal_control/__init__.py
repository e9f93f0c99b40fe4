"""MI355X-native GRAPE engine behind the import path of SchusterLab/quantum-optimal-control.

    from quantum_optimal_control.main_grape.grape import Grape

The package is import-light on purpose: nothing here imports TensorFlow, matplotlib or IPython.
"""
__version__ = '0.1.0'
