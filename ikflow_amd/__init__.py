"""ikflow_amd - MI355X-native engine for the IKFlow inference hot path
(IKFlowSolver.generate_ik_solutions / generate_exact_ik_solutions of jstmn/ikflow)."""
__version__ = "0.1.0"
