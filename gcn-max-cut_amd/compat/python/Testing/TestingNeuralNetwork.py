"""Reference import root -> the same module object as ``gcn_max_cut_amd.Testing.TestingNeuralNetwork``."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("gcn_max_cut_amd.Testing.TestingNeuralNetwork")
