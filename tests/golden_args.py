"""Argument sets used when the golden fixtures were generated (tests/golden/make_golden.py: glow_args)."""
GLOW_DEFAULTS = dict(learn_prior=True, n_units_prior=16, make_conditional=True, base_norm="actnorm",
                     non_lin_glow="relu", split2d_act="softplus", L=2, K=2, n_bits=8, LU_decomposed=True,
                     n_units_affine=16, clamp_type="realnvp", flow_norm="actnorm", flow_batchnorm_momentum=0.0)
