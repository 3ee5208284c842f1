"""CPU test of the rule "every kernel instance a default plan can select runs in the default GPU suite" (tests/kernel_coverage.py).

The planner itself (passes.build_chain, FFTPlan._select_strategy on the model of the full MI355X) enumerates what default plans launch
for every power-of-two shape of up to 2^24 points in the four dtypes -- rows, strided passes, one-launch N-D shapes, pass pairs, persistent
launches by tuning rule, the several-work-groups-per-transform kernels of out-of-place executes -- and the same planner says what each
default-collected GPU test case launches.  The reference covers its kernels by shape sweeps too: test/test_errors.py:125-145."""
import numpy

import kernel_coverage as KC
from pyfft_amd import _native as N


def test_every_kernel_instance_a_default_plan_selects_has_a_default_gpu_test():
    uni = KC.universe()
    cov = KC.covered_keys()
    kinds = {}
    for k in uni:
        kinds[k[0]] = kinds.get(k[0], 0) + 1
    # the enumeration reaches every family (a planner change that silently drops one would make the rule vacuous)
    assert kinds["pairXY"] >= 15 and kinds["pairYZ"] >= 25 and kinds["persistent"] >= 60 and kinds["nd_oop"] >= 20 and kinds["nd_fixed"] >= 350 and kinds["nd2z"] >= 30 and kinds["nd_generic"] >= 20
    assert kinds["row"] >= 40 and kinds["col"] >= 80
    for prec in ("f32", "f64"):            # every instance of the several-work-groups tables is somebody's default choice (or dead code)
        for xyz in KC.nd2z_shapes(prec):
            assert ("nd2z", prec) + xyz in uni or ("nd_oop", prec) + xyz in uni, (prec, xyz)
    missing = sorted(((k, ex) for k, ex in uni.items() if k not in cov), key=str)
    assert not missing, "kernel instances no default-collected GPU test runs (key, an example plan): %r" % (missing[:40],)


def test_pair_kernel_table_and_reachable_pair_keys_agree():
    """mifft_pair_kernel_supported over its key space (include/mifft.h): every pair key a plan reaches has a kernel, and every kernel of
    the table is reached by some default plan -- no dead instance, no plan that would ask for a missing one."""
    have = set()
    pows = [1 << k for k in range(1, 15)]
    for prec, pname in ((N.F32, "f32"), (N.F64, "f64")):
        for lay, lname in ((N.INTERLEAVED, "interleaved"), (N.SPLIT, "split")):
            for kind, kname in ((0, "pairXY"), (1, "pairYZ")):
                for k0 in pows + [1 << 15, 1 << 16, 1 << 17, 1 << 18]:
                    for k1 in pows[:12]:
                        for k2 in pows[:12]:
                            if N.lib.mifft_pair_kernel_supported(prec, lay, kind, k0, k1, k2) == 0:
                                have.add((kname, pname, lname, k0, k1, k2))
    reached = set(k for k in KC.universe() if k[0] in ("pairXY", "pairYZ"))
    assert reached <= have, sorted(reached - have)
    # (instances only the persistent two-pair launch uses -- the {64, 128}^3 shapes whose chain is a plane pass + a z pass -- are reached
    # through their 'persistent' keys; what is left over here would be a kernel no plan selects)
    from pyfft_amd.plan import FFTPlan     # noqa: F401
    fusedp_only = set()
    for k, (shape, dtname, batch) in KC.universe().items():
        if k[0] == "persistent" and len(shape) == 3:
            plan = KC.plan_for(shape, numpy.dtype(dtname))
            chain = plan._pair_alt or plan._kernels
            fusedp_only |= KC._chain_keys(plan, chain)
    dead = sorted(k for k in have - reached - fusedp_only - KC.DEV_ONLY_PAIR_KEYS)
    assert not dead, "pair kernels no default plan reaches: %r" % (dead,)


def test_registry_names_every_gpu_test_function():
    """A GPU test added without saying what it covers fails here, not silently."""
    cases = KC.collected_cases()                 # raises on a missing / stale entry
    assert len(cases) > 1800
    mods = set(m for m, _, _ in cases)
    assert {"test_errors_gpu", "test_persistent_gpu", "test_pairs_gpu", "test_nd_gpu", "test_strided_gpu", "test_interop_gpu"} <= mods
