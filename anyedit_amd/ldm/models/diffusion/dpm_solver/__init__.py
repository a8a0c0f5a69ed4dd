from .sampler import DPMSolverSampler
