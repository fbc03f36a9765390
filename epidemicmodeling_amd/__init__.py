"""epidemicmodeling_amd -- MI355X-native EKF/EKS ensemble engine behind the
reference's Tools/ function signatures."""
__version__ = "0.1.0"
