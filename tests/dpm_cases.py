"""The DPM_Solver.sample argument sets tools/gen_golden.py::gen_dpm_solver_general ran in the reference (tests/golden/dpm_solver_general.npz):
tag -> (predict_x0, keyword arguments).  Shared by the oracle-vs-golden test (CPU) and the HIP-vs-golden test (GPU)."""
DPM_GENERAL_CASES = {  # tag -> (predict_x0, DPM_Solver.sample arguments) — the cases tools/gen_golden.py::gen_dpm_solver_general ran in the reference
    "m3.x0": (True, dict(steps=15, order=3, method="multistep", skip_type="time_uniform")),
    "m3.eps.taylor": (False, dict(steps=16, order=3, method="multistep", skip_type="logSNR", solver_type="taylor")),
    "m3.x0.nolof": (True, dict(steps=9, order=3, method="multistep", skip_type="time_quadratic", lower_order_final=False, denoise_to_zero=True)),
    "s3.eps": (False, dict(steps=10, order=3, method="singlestep", skip_type="logSNR")),
    "s3.x0.taylor": (True, dict(steps=9, order=3, method="singlestep", skip_type="logSNR", solver_type="taylor")),
    "s3.eps.taylor": (False, dict(steps=11, order=3, method="singlestep", skip_type="logSNR", solver_type="taylor", denoise_to_zero=True)),
    "s3.x0": (True, dict(steps=12, order=3, method="singlestep", skip_type="logSNR")),
    "s2.eps": (False, dict(steps=7, order=2, method="singlestep", skip_type="logSNR")),
    "s2.x0.taylor": (True, dict(steps=6, order=2, method="singlestep", skip_type="logSNR", solver_type="taylor")),
    "s2.eps.taylor": (False, dict(steps=8, order=2, method="singlestep", skip_type="logSNR", solver_type="taylor")),
    "f3.eps": (False, dict(steps=9, order=3, method="singlestep_fixed", skip_type="time_uniform")),
    "f2.x0": (True, dict(steps=8, order=2, method="singlestep_fixed", skip_type="logSNR")),
    "a2.eps": (False, dict(order=2, method="adaptive")),
    "a3.eps": (False, dict(order=3, method="adaptive", atol=0.01, rtol=0.1)),
    "a3.x0.taylor": (True, dict(order=3, method="adaptive", solver_type="taylor", t_end=0.01)),
}
