"""Target registry: which `ln_like_fn` callables are evaluated on the device.  The reference's own target
objects (bipymc/utils/d100_gauss.py:14-35, dblgauss_rv.py:11-32, banana_rv.py:11-40) must be recognised, their parameter
block rebuilt from their attributes and verified against the callable; everything else is a host callback
(samplers.py:36-43)."""
import os
import sys

import numpy as np
import pytest

import _ref_lookalikes as LK
from bipymc_amd.utils import _target as T
from bipymc_amd.utils import banana_rv, d100_gauss, dblgauss_rv, mixture_nd

REF = "/root/reference"


def _closed_form_of(tid, blk):
    if tid == T.TARGET_GAUSS_EQUICORR:
        return lambda y: d100_gauss.equicorr_ln_like(blk, y)
    if tid == T.TARGET_MIXTURE_PAIRS:
        m = dblgauss_rv.BimodeGauss_2D()
        m._params = np.array(blk)
        return m.ln_like
    b = banana_rv.Banana_2D()
    b._params = np.array(blk)
    return b.ln_like


@pytest.mark.parametrize("obj,dim,tid,ours", [
    (LK.Gauss_100D(), 100, T.TARGET_GAUSS_EQUICORR, d100_gauss.Gauss_100D()),
    (LK.Gauss_100D(rho=0.3, dim=7), 7, T.TARGET_GAUSS_EQUICORR, d100_gauss.Gauss_100D(rho=0.3, dim=7)),
    (LK.BimodeGauss_2D(), 2, T.TARGET_MIXTURE_PAIRS, dblgauss_rv.BimodeGauss_2D()),
    (LK.BimodeGauss_2D(mu_g2=(1.5, -1.0), sigma_g1=(0.3, 0.2), rho_g1=0.5, w_g1=1.0, w_g2=1.0), 2, T.TARGET_MIXTURE_PAIRS,
     dblgauss_rv.BimodeGauss_2D(mu_g2=[1.5, -1.0], sigma_g1=[0.3, 0.2], rho_g1=0.5, w_g1=1.0, w_g2=1.0)),
    (LK.Banana_2D(), 2, T.TARGET_BANANA_2D, banana_rv.Banana_2D()),
    (LK.Banana_2D(mu1=0.2, sigma2=1.5, rho=0.5, a=1.3, b=0.25), 2, T.TARGET_BANANA_2D,
     banana_rv.Banana_2D(mu1=0.2, sigma2=1.5, rho=0.5, a=1.3, b=0.25)),
])
def test_lookalikes_of_the_reference_targets_resolve_to_device_targets(obj, dim, tid, ours):
    got_tid, blk, rule = T.resolve_info(obj.ln_like, {}, dim)
    assert (got_tid, rule) == (tid, "reference-lookalike")
    # the block rebuilt from the attributes is the block the shipped target publishes
    np.testing.assert_allclose(blk, ours._bpm_target_spec()[1], rtol=1e-13, atol=1e-15)
    f = _closed_form_of(tid, blk)
    rs = np.random.RandomState(3)
    for _ in range(5):
        y = np.asarray(ours.rvs(1))
        y = y.reshape(-1) if y.ndim == 2 and y.shape[0] == 1 else np.array([float(y[0][0]), float(y[1][0])])
        np.testing.assert_allclose(float(f(y)), float(obj.ln_like(y)), rtol=1e-9)


def test_verification_rejects_a_lookalike_with_another_density():
    o = LK.Gauss_100D_scaled(dim=6)
    assert type(o).__name__ == "Gauss_100D_scaled"
    assert T.resolve_info(o.ln_like, {}, 6)[2] == "host-callback"          # class name not in the registry
    Fake = type("Gauss_100D", (LK.Gauss_100D_scaled,), {})
    assert T.resolve_info(Fake(dim=6).ln_like, {}, 6)[2] == "host-callback"   # name matches, values do not
    o2 = LK.Gauss_100D(dim=6)
    o2.cov = o2.cov + np.diag(np.arange(6) * 0.1)                            # no longer equicorrelated
    assert T.resolve_info(o2.ln_like, {}, 6)[2] == "host-callback"
    b = LK.Banana_2D()
    b.a = 1.3                                                                  # attribute edited after construction: ln_like follows, block too
    assert T.resolve_info(b.ln_like, {}, 2)[2] == "reference-lookalike"
    m = LK.BimodeGauss_2D()
    m.w_g1 = 0.5                                                               # weights no longer normalised
    assert T.resolve_info(m.ln_like, {}, 2)[2] == "host-callback"


def test_a_users_variant_of_a_reference_class_stays_a_host_callback():
    """a subclass / a patched instance / a same-named class from another module keeps the attributes the look-alike rule reads and may
    change ln_like AWAY from the mode (a prior box, a truncation, a tempering factor): its likelihood must never be replaced by the closed form."""
    import types
    boxed = LK.Gauss_100D_boxed(dim=6)
    assert np.isfinite(boxed.ln_like(np.zeros(6))) and boxed.ln_like(np.full(6, 100.0)) == -np.inf
    assert T.resolve_info(boxed.ln_like, {}, 6)[2] == "host-callback"                 # a subclass under its own name
    Same = type("Gauss_100D", (LK.Gauss_100D_boxed,), {"__module__": "bipymc.utils.d100_gauss"})
    assert T.resolve_info(Same(dim=6).ln_like, {}, 6)[2] == "host-callback"           # ... under the reference's name and module: still a subclass
    Flat = type("Gauss_100D", (object,), dict(vars(LK.Gauss_100D_boxed), __init__=LK.Gauss_100D.__init__,
                                              __module__="bipymc.utils.d100_gauss"))
    Flat.ln_like = lambda self, y: (-np.inf if np.any(np.abs(np.asarray(y)) > 5.0 * self.var) else float(np.log(self.rv_100d.pdf(y))))
    assert T.resolve_info(Flat(dim=6).ln_like, {}, 6)[2] == "host-callback"           # no base class, same name and module: the TAIL probes catch the box
    g = LK.Gauss_100D(dim=6)
    g.ln_like = types.MethodType(lambda self, y: -0.5 * float(np.sum(np.asarray(y) ** 2)), g)
    assert T.resolve_info(g.ln_like, {}, 6)[2] == "host-callback"                     # patched on the instance
    Elsewhere = type("Gauss_100D", (object,), dict(vars(LK.Gauss_100D), __module__="my_models"))
    assert T.resolve_info(Elsewhere(dim=6).ln_like, {}, 6)[2] == "host-callback"      # the name alone is not enough any more
    assert T.resolve_info(LK.Gauss_100D(dim=6).ln_like, {}, 6)[2] == "reference-lookalike"


def test_promotion_to_the_device_path_is_logged(caplog):
    import logging
    with caplog.at_level(logging.INFO, logger="bipymc_amd"):
        assert T.resolve_info(LK.Banana_2D().ln_like, {}, 2)[2] == "reference-lookalike"
    assert any("evaluated on the GPU" in r.getMessage() and "Banana_2D" in r.getMessage() for r in caplog.records)


def test_rules_that_keep_the_host_callback():
    g = LK.Gauss_100D(dim=6)
    assert T.resolve_info(g.ln_like, {"extra": 1}, 6)[2] == "host-callback"      # frozen kwargs change the callable
    assert T.resolve_info(g.ln_like, {}, 5)[2] == "host-callback"                # dimension mismatch
    assert T.resolve_info(lambda y: -0.5 * float(np.sum(y * y)), {}, 6)[2] == "host-callback"
    assert T.resolve_info(g.rv_100d.logpdf, {}, 6)[2] == "host-callback"         # another bound method of a recognised object
    ours = mixture_nd.BimodeGauss_ND(8)
    assert T.resolve_info(ours.ln_like, {}, 8)[2] == "spec"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "bipymc", "utils")), reason="the reference is not on this machine")
def test_the_genuine_reference_objects_are_recognised(oracle_engine):
    os.environ.setdefault("MPLBACKEND", "agg")
    sys.path.insert(0, REF)
    sys.dont_write_bytecode, old = True, sys.dont_write_bytecode
    try:
        from bipymc.utils import banana_rv as rb, d100_gauss as rg, dblgauss_rv as rd
    finally:
        sys.path.remove(REF)
        sys.dont_write_bytecode = old
    for obj, dim, tid, ours in ((rg.Gauss_100D(), 100, T.TARGET_GAUSS_EQUICORR, d100_gauss.Gauss_100D()),
                                (rg.Gauss_100D(rho=0.2, dim=10), 10, T.TARGET_GAUSS_EQUICORR, d100_gauss.Gauss_100D(rho=0.2, dim=10)),
                                (rd.BimodeGauss_2D(), 2, T.TARGET_MIXTURE_PAIRS, dblgauss_rv.BimodeGauss_2D()),
                                (rb.Banana_2D(1, 1), 2, T.TARGET_BANANA_2D, banana_rv.Banana_2D(1, 1)),
                                (rb.Banana_2D(), 2, T.TARGET_BANANA_2D, banana_rv.Banana_2D())):
        got_tid, blk, rule = T.resolve_info(obj.ln_like, {}, dim)
        assert (got_tid, rule) == (tid, "reference-lookalike"), type(obj)
        np.testing.assert_allclose(blk, ours._bpm_target_spec()[1], rtol=1e-13, atol=1e-15)
    # and the sampler class takes the device path for it (oracle engine: no GPU here)
    from bipymc_amd.dream import DreamMpi
    s = DreamMpi(rd.BimodeGauss_2D().ln_like, np.zeros(2), n_chains=8, seed=1)
    assert s.uses_device_target and s.target_rule == "reference-lookalike"
    s.run_mcmc(8 * 5)
    assert s.param_est(0)[2].shape == (8 * 5, 2)
