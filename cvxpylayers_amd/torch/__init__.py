from cvxpylayers_amd.torch.cvxpylayer import CanonTemplate, CvxpyLayer, VariableRecovery  # noqa: F401
